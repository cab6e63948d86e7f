// refine.hip -- the full-resolution part of the Refinement network (reference models/net.py:73-126) in two kernels.
//
// Layer by layer (conv0 on the image, the transposed convolution, torch.cat, conv3, res, and four element-wise passes) the
// full-resolution half of Refinement moves ~0.9 GB per 1600x1200 depth map through HBM in ten launches for 1.6 GFLOP of work.
//   pmn_refine_front   x16 = cat( relu(bn(deconv(t))), conv0(img) )         image + half-resolution features -> [B,H,W,16]
//   pmn_refine_tail    depth = (nearest_x2(d) + res(conv3(x16))) * span + min                    [B,H,W,16] -> [B,1,H,W]
// Both own a 16x16 tile of output pixels per 256-thread workgroup, stage what they read in LDS with every load of a thread
// in flight at once, and feed the FMAs wave-uniform SGPR weights ([ky][kx][ci][co] layouts of params.pack_conv / pack_deconv).
#include "pmn_common.hpp"

typedef const float __attribute__((address_space(4))) cfloat;

// ---- front: conv0 (3 -> 8, BN, ReLU; net.py:82,110) || ConvTranspose2d(8,8,3,s2,p1,op1) + BN + ReLU (net.py:86-88,114) -----------
// out[oy,ox] of the transposed convolution = sum over (ky,kx) with (oy+1-ky), (ox+1-kx) even of in[(oy+1-ky)/2,(ox+1-kx)/2] w[ky][kx]:
// even coordinates take tap 1 of in[y], odd ones tap 2 of in[y] and tap 0 of in[y+1].  Wave w of the workgroup owns parity class
// (py,px) = (w>>1, w&1) of the tile -- 8x8 pixels at stride 2 -- so the tap set is wave-uniform (scalar branches, SGPR weights).
__global__ __launch_bounds__(PMN_BLOCK) void refine_front_kernel(const float* __restrict__ img, const float* __restrict__ t2,
                                                                 const float* __restrict__ w0, const float* __restrict__ s0,
                                                                 const float* __restrict__ wd, const float* __restrict__ sd,
                                                                 float* __restrict__ x16, int B, int H, int W) {
    constexpr int TW = 16, TH = 16, IW = 18, IWP = 19, TP = 9, TPP = 12;  // image patch 18x18 (pitch 19), t2 patch 9x9 px x 12 words
    __shared__ float xin[3 * IW * IWP];
    __shared__ float4 tp4[TP * TP * TPP / 4];
    float* tp = reinterpret_cast<float*>(tp4);
    const cfloat* cw0 = (const cfloat*)w0;  // [3][3][3][8]
    const cfloat* cs0 = (const cfloat*)s0;
    const cfloat* cwd = (const cfloat*)wd;  // [3][3][8][8]
    const cfloat* csd = (const cfloat*)sd;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int bt = pmn_xcd_tile(blockIdx.x, B * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const int Hi = H / 2, Wi = W / 2, iy0 = oy0 / 2, ix0 = ox0 / 2;

    {
        constexpr int NL = (3 * IW * IW + PMN_BLOCK - 1) / PMN_BLOCK;  // 4
        float v[NL];
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * PMN_BLOCK;
            const int c = idx / (IW * IW), r = (idx / IW) % IW, q = idx % IW;
            const int gy = oy0 - 1 + r, gx = ox0 - 1 + q;
            v[k] = 0.0f;
            if (idx < 3 * IW * IW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v[k] = img[(((size_t)n * 3 + c) * H + gy) * W + gx];
        }
        const int upix = tid >> 1, uq = tid & 1;  // 81 pixels x 2 float4
        if (upix < TP * TP) {
            const int gy = iy0 + upix / TP, gx = ix0 + upix % TP;
            if (gy < Hi && gx < Wi) u = *reinterpret_cast<const float4*>(t2 + (((size_t)n * Hi + gy) * Wi + gx) * 8 + 4 * uq);
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * PMN_BLOCK;
            const int c = idx / (IW * IW), r = (idx / IW) % IW, q = idx % IW;
            if (idx < 3 * IW * IW) xin[(c * IW + r) * IWP + q] = v[k];
        }
        if (upix < TP * TP) *reinterpret_cast<float4*>(tp + upix * TPP + 4 * uq) = u;
    }
    __syncthreads();

    const int py = wave >> 1, px = wave & 1, yy = lane >> 3, xx = lane & 7;
    const int ty = 2 * yy + py, tx = 2 * xx + px;  // pixel inside the tile
    float up[8], f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { up[c] = 0.0f; f[c] = 0.0f; }

    // transposed convolution: taps valid for this wave's parity class
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        if ((py == 0) != (ky == 1)) continue;  // even rows: ky = 1; odd rows: ky = 0 (input row y+1) and ky = 2 (row y)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            if ((px == 0) != (kx == 1)) continue;
            const float* ip = tp + ((yy + (ky == 0 ? 1 : 0)) * TP + xx + (kx == 0 ? 1 : 0)) * TPP;
            const float4 a = *reinterpret_cast<const float4*>(ip), b = *reinterpret_cast<const float4*>(ip + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            const cfloat* wq = cwd + __builtin_amdgcn_readfirstlane((ky * 3 + kx) * 64);
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int c = 0; c < 8; ++c) up[c] = fmaf(v[ci], wq[ci * 8 + c], up[c]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // conv0 on the image
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {  // rolled: fully unrolled, hipcc emits scalar v_fma_f32 instead of packed FMAs for this block
        const cfloat* wq = cw0 + __builtin_amdgcn_readfirstlane(ky * 72);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = xin[(ci * IW + ty + ky) * IWP + tx + kx];
#pragma unroll
                for (int c = 0; c < 8; ++c) f[c] = fmaf(v, wq[(kx * 3 + ci) * 8 + c], f[c]);
            }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= H || ox >= W) return;
    float* op = x16 + (((size_t)n * H + oy) * W + ox) * 16;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        up[c] = fmaxf(up[c] + csd[c], 0.0f);
        f[c] = fmaxf(f[c] + cs0[c], 0.0f);
    }
    *reinterpret_cast<float4*>(op) = make_float4(up[0], up[1], up[2], up[3]);
    *reinterpret_cast<float4*>(op + 4) = make_float4(up[4], up[5], up[6], up[7]);
    *reinterpret_cast<float4*>(op + 8) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(op + 12) = make_float4(f[4], f[5], f[6], f[7]);
}

// img [B,3,H,W] planar; t2 [B,H/2,W/2,8] (conv2 output, channels-last); w0 [3][3][3][8] / s0 [8] (pack_conv of conv0);
// wd [3][3][8][8] / sd [8] (pack_deconv of deconv + bn) -> x16 [B,H,W,16] = (deconv branch | image branch) as net.py:117 concatenates
extern "C" int pmn_refine_front(const float* img, const float* t2, const float* w0, const float* s0, const float* wd,
                                const float* sd, float* x16, int B, int H, int W, void* stream) {
    if (!img || !t2 || !w0 || !s0 || !wd || !sd || !x16 || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return PMN_ERR_ARG;
    const int blocks = B * ((W + 15) / 16) * ((H + 15) / 16);
    PMN_LAUNCH(refine_front_kernel, dim3(blocks), dim3(PMN_BLOCK), 0, (hipStream_t)stream, img, t2, w0, s0, wd, sd, x16, B,
                       H, W);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// ---- tail: conv3 (16 -> 8, BN, ReLU; net.py:90,117) -> res (8 -> 1, no bias; net.py:92,117) -> residual + de-normalisation ---------
// conv3 is evaluated on the 18x18 halo patch into LDS (positions outside the image are ZERO: res pads conv3's output map) in
// 12 wave-passes of (one half of the 8 output channels) x (64 pixels) -- the half is wave-uniform, so the weights stay in
// SGPRs; then every thread forms its pixel's residual from ds_read_b128's.
__global__ __launch_bounds__(PMN_BLOCK) void refine_tail_kernel(const float* __restrict__ x16, const float* __restrict__ w3,
                                                                const float* __restrict__ s3, const float* __restrict__ wr,
                                                                const float* __restrict__ dnorm,
                                                                const float* __restrict__ dmin, const float* __restrict__ dmax,
                                                                float* __restrict__ out, int B, int H, int W) {
    // x16 patch 20x20 px x 20 words, row pitch 424 words; conv3 map 18x18 px x 12 words, row pitch 256 words: with these row
    // pitches the ds_read_b128 of both phases are bank-conflict free (brute-forced over the 16-lane service groups of a wave64
    // b128 read; the dense pitches 400 / 216 cost 1.8x / 2x the LDS cycles)
    constexpr int TW = 16, TH = 16, IW = 20, XP = 20, XRP = 424, MW = 18, MP = 12, MRP = 256;
    extern __shared__ float4 rt_lds4[];
    float* xs = reinterpret_cast<float*>(rt_lds4);
    float* mid = xs + IW * XRP;
    const cfloat* cw3 = (const cfloat*)w3;  // [2 halves][3][3][16][4]
    const cfloat* cs3 = (const cfloat*)s3;
    const cfloat* cwr = (const cfloat*)wr;  // [3][3][8]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, tx = tid % TW, ty = tid / TW;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int bt = pmn_xcd_tile(blockIdx.x, B * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;

    {
        constexpr int TOT = IW * IW * 4, NL = (TOT + PMN_BLOCK - 1) / PMN_BLOCK;  // 1600 float4, 7 per thread
        float4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * PMN_BLOCK, pix = idx >> 2, q = idx & 3;
            const int gy = oy0 - 2 + pix / IW, gx = ox0 - 2 + pix % IW;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < TOT && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v[k] = *reinterpret_cast<const float4*>(x16 + (((size_t)n * H + gy) * W + gx) * 16 + 4 * q);
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * PMN_BLOCK, pix = idx >> 2, q = idx & 3;
            if (idx < TOT) *reinterpret_cast<float4*>(xs + (pix / IW) * XRP + (pix % IW) * XP + 4 * q) = v[k];
        }
    }
    __syncthreads();

    // conv3: 2 halves (4 output channels each) x 6 groups of 64 pixels = 12 wave-passes over 4 waves
#pragma unroll 1
    for (int p = 0; p < 3; ++p) {
        const int wp = p * 4 + wave, half = wp / 6, m = (wp % 6) * 64 + lane;
        const int mc = min(m, MW * MW - 1), r = mc / MW, q = mc - r * MW;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const cfloat* wh = cw3 + __builtin_amdgcn_readfirstlane(half * 576);
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll 1
            for (int kx = 0; kx < 3; ++kx) {  // rolled: one tap's 64 weights in SGPRs at a time (unrolled, hipcc hoists all 576 and spills)
                const float* ip = xs + (r + ky) * XRP + (q + kx) * XP;
                const cfloat* wq = wh + __builtin_amdgcn_readfirstlane((ky * 3 + kx) * 64);
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float4 a = *reinterpret_cast<const float4*>(ip + 4 * c4);
                    const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = fmaf(v[k], wq[(4 * c4 + k) * 4 + c], acc[c]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        const int gy = oy0 - 1 + r, gx = ox0 - 1 + q;
        const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const cfloat* sh = cs3 + __builtin_amdgcn_readfirstlane(half * 4);
        float4 o;
        o.x = inside ? fmaxf(acc[0] + sh[0], 0.0f) : 0.0f;
        o.y = inside ? fmaxf(acc[1] + sh[1], 0.0f) : 0.0f;
        o.z = inside ? fmaxf(acc[2] + sh[2], 0.0f) : 0.0f;
        o.w = inside ? fmaxf(acc[3] + sh[3], 0.0f) : 0.0f;
        if (m < MW * MW) *reinterpret_cast<float4*>(mid + r * MRP + q * MP + half * 4) = o;
    }
    __syncthreads();

    float res = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* mp = mid + (ty + ky) * MRP + (tx + kx) * MP;
            const float4 a = *reinterpret_cast<const float4*>(mp), b = *reinterpret_cast<const float4*>(mp + 4);
            const cfloat* wq = cwr + (ky * 3 + kx) * 8;
            res = fmaf(a.x, wq[0], res); res = fmaf(a.y, wq[1], res); res = fmaf(a.z, wq[2], res); res = fmaf(a.w, wq[3], res);
            res = fmaf(b.x, wq[4], res); res = fmaf(b.y, wq[5], res); res = fmaf(b.z, wq[6], res); res = fmaf(b.w, wq[7], res);
        }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= H || ox >= W) return;
    {
#pragma clang fp contract(off)
        const float lo = dmin[n], span = dmax[n] - lo;
        const float d = dnorm[((size_t)n * (H / 2) + (oy >> 1)) * (W / 2) + (ox >> 1)] + res;  // nearest x2 + residual (net.py:119)
        out[((size_t)n * H + oy) * W + ox] = d * span + lo;                                     // net.py:122
    }
}

// x16 [B,H,W,16] (pmn_refine_front); w3 [2][3][3][16][4] (conv3, BN folded, output channels split in two halves of 4:
// params.pack_refine_tail) / s3 [8]; wr [3][3][8] (res weight, [ky][kx][ci]);
// dnorm [B,1,H/2,W/2] = (depth - depth_min) / (depth_max - depth_min); depth_min / depth_max DEVICE float[B] -> out [B,1,H,W]
extern "C" int pmn_refine_tail(const float* x16, const float* w3, const float* s3, const float* wr, const float* dnorm,
                               const float* depth_min, const float* depth_max, float* out, int B, int H, int W, void* stream) {
    if (!x16 || !w3 || !s3 || !wr || !dnorm || !depth_min || !depth_max || !out || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1))
        return PMN_ERR_ARG;
    const size_t lds = (size_t)(20 * 424 + 18 * 256) * sizeof(float);  // 52.4 KB
    if (pmn_raise_dynamic_lds(reinterpret_cast<const void*>(refine_tail_kernel), lds) != PMN_OK) return PMN_ERR_LAUNCH;
    const int blocks = B * ((W + 15) / 16) * ((H + 15) / 16);
    PMN_LAUNCH(refine_tail_kernel, dim3(blocks), dim3(PMN_BLOCK), lds, (hipStream_t)stream, x16, w3, s3, wr, dnorm,
                       depth_min, depth_max, out, B, H, W);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// =================================================================================================================================
// pmn_refine_fused: the two kernels above in ONE (round 3).  pmn_refine_front / pmn_refine_tail exchange x16 = cat(relu(bn(deconv(t2))),
// conv0(img)), [B,H,W,16] fp32 = 123 MB per 1600x1200 depth map, written once and read back with a 1.56x halo, and conv3 (16 -> 8, 144
// MACs per output) runs on the fp32 VALU.  Here a workgroup owns 16 x 16 output pixels and keeps everything in LDS:
//   (1) image patch 22 x 24 x 3 and t2 patch 11 x 11 x 8 -> LDS (aligned float4 loads when W % 4 == 0);
//   (2) x16 on the 20 x 20 halo patch: wave w owns parity class (w >> 1, w & 1) of the patch (the transposed convolution's tap set is
//       then wave-uniform, as in refine_front_kernel), 100 pixels = two passes; same FMA order as refine_front_kernel, so x16 has the same
//       bits; zero outside the image (conv3 pads x16); split into hi / lo fp16 planes [20][26 px][16 halves] (pitches from
//       scripts/experiments/refine_fused/lds_banks.py: the conv3 operand reads below are conflict-free to 10 %);
//   (3) conv3 on the 18 x 18 patch as split-operand fp16 MFMAs (x = hi + lo / 2048 as conv_f16s.hip; v_mfma_f32_16x16x32_f16, roles
//       swapped: rows = 8 output channels, columns = 16 consecutive pixels of the patch in linear order -- 21 M-tiles, six per wave, three
//       for wave 3 which carries the four-tap parity class in (2)); k = (tap, half of the 16 channels): 18 blocks = 5 k-steps; + shift,
//       ReLU, zero outside the image (res pads conv3's OUTPUT) -> fp32 map in LDS, laid out as refine_tail_kernel's;
//   (4) res (8 -> 1) + nearest-x2 residual + de-normalisation: refine_tail_kernel's last phase verbatim.
// HBM traffic per depth map: 23 MB image + 15 MB t2 + 2 MB depth + 7.7 MB out instead of 292 MB; 148 -> 97 us on the same box
// (scripts/experiments/refine_fused/ab.py).  conv3 carries the split-fp16 error (2-4e-7 of its output scale: both forms are 1.1e-6 of
// the depth span from a float64 evaluation); everything else is the arithmetic of the two kernels it replaces.
// =================================================================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define RF_LO_SCALE 2048.0f

template <bool VEC4>
__global__ __launch_bounds__(PMN_BLOCK, 3) void refine_fused_kernel(
    const float* __restrict__ img, const float* __restrict__ t2, const float* __restrict__ w0, const float* __restrict__ s0,
    const float* __restrict__ wd, const float* __restrict__ sd, const f16x8* __restrict__ w3A, const float* __restrict__ s3,
    const float* __restrict__ wr, const float* __restrict__ dnorm, const float* __restrict__ dmin, const float* __restrict__ dmax,
    float* __restrict__ out, int B, int H, int W) {
    constexpr int TW = 16, TH = 16;
    constexpr int IR = 22, IC = 24;             // image patch: rows oy0 - 3 .., columns ox0 - 4 .. ox0 + 19 (six aligned float4)
    constexpr int TP = 11, TPP = 12;            // t2 patch 11 x 11 px x 12 words, origin (oy0 / 2 - 1, ox0 / 2 - 1)
    constexpr int XW = 20, XROWP = 26, XPIX = 16;  // x16 planes: [20 rows][26 px pitch][16 halves]
    constexpr int XPLANE = XW * XROWP * XPIX;   // halves per plane
    constexpr int MW = 18, MP = 12, MRP = 256;  // conv3 map: 18 x 18 px x 12 words, row pitch 256 words (refine_tail_kernel's)
    // LDS: [x16 hi | x16 lo] 33,280 B, then one region that first holds the input patches and later the conv3 map (18,432 B)
    extern __shared__ float4 rf_lds4[];
    _Float16* xh = reinterpret_cast<_Float16*>(rf_lds4);
    float* region = reinterpret_cast<float*>(xh + 2 * XPLANE);
    float* xin = region;                        // 3 * 22 * 24 floats = 6,336 B
    float* tp = region + 3 * IR * IC;           // 121 * 12 floats = 5,808 B
    float* mid = region;                        // 18 * 256 floats
    const cfloat* cw0 = (const cfloat*)w0;      // [3][3][3][8]
    const cfloat* cs0 = (const cfloat*)s0;
    const cfloat* cwd = (const cfloat*)wd;      // [3][3][8][8]
    const cfloat* csd = (const cfloat*)sd;
    const cfloat* cwr = (const cfloat*)wr;      // [3][3][8]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, kb = lane >> 4;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int bt = pmn_xcd_tile(blockIdx.x, B * tiles_x * tiles_y);
    const int n = bt / (tiles_x * tiles_y), tr = bt - n * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const int Hi = H / 2, Wi = W / 2, iy0 = oy0 / 2 - 1, ix0 = ox0 / 2 - 1;

    // ---- (1) input patches -> LDS, every load of the thread in flight before the first LDS write ----------------------------------
    {
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
        const int upix = tid >> 1, uq = tid & 1;  // 121 pixels x 2 float4
        const int ur = upix / TP, uc = upix - ur * TP;
        if (upix < TP * TP) {
            const int gy = iy0 + ur, gx = ix0 + uc;
            if ((unsigned)gy < (unsigned)Hi && (unsigned)gx < (unsigned)Wi)
                u = *reinterpret_cast<const float4*>(t2 + (((size_t)n * Hi + gy) * Wi + gx) * 8 + 4 * uq);
        }
        if constexpr (VEC4) {
            float4 v[3];
            const int j = tid & 7, gx = ox0 - 4 + 4 * j;  // float4 j of the row (6 used)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int R = (tid >> 3) + 32 * k;       // row R of 66 = (channel, patch row)
                const int c = (R >= IR) + (R >= 2 * IR), r = R - IR * c, gy = oy0 - 3 + r;
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < 6 && R < 3 * IR && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                    v[k] = *reinterpret_cast<const float4*>(img + (((size_t)n * 3 + c) * H + gy) * W + gx);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int R = (tid >> 3) + 32 * k;
                if (j < 6 && R < 3 * IR) *reinterpret_cast<float4*>(xin + R * IC + 4 * j) = v[k];
            }
        } else {  // any width / base alignment: one float per thread, 8 rows of 32 columns (24 used) per pass
            float v[9];
            const int q = tid & 31, gx = ox0 - 4 + q;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int R = (tid >> 5) + 8 * k;
                const int c = (R >= IR) + (R >= 2 * IR), r = R - IR * c, gy = oy0 - 3 + r;
                v[k] = 0.0f;
                if (q < IC && R < 3 * IR && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                    v[k] = img[(((size_t)n * 3 + c) * H + gy) * W + gx];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int R = (tid >> 5) + 8 * k;
                if (q < IC && R < 3 * IR) xin[R * IC + q] = v[k];
            }
        }
        if (upix < TP * TP) *reinterpret_cast<float4*>(tp + upix * TPP + 4 * uq) = u;
    }
    // conv3's weights (A operands of the five k-steps, hi | lo): ten 1 KB loads per wave, in flight across phase (2)
    f16x8 wa[5][2];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        wa[ks][0] = w3A[(ks * 2 + 0) * 64 + lane];
        wa[ks][1] = w3A[(ks * 2 + 1) * 64 + lane];
    }
    __syncthreads();

    // ---- (2) x16 on the 20 x 20 patch (global origin (oy0 - 2, ox0 - 2)): wave = parity class, two passes of 64 / 36 pixels --------
    {
        const int py = wave >> 1, px = wave & 1;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int idx = lane + 64 * pass;
            if (idx >= 100) break;
            const int yy = idx / 10, xx = idx - yy * 10;
            const int r = 2 * yy + py, q = 2 * xx + px;      // patch position
            const int gy = oy0 - 2 + r, gx = ox0 - 2 + q;
            f16x8 h0 = {0, 0, 0, 0, 0, 0, 0, 0}, h1 = h0, l0 = h0, l1 = h0;  // channels 0-7 (deconv branch), 8-15 (image branch)
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                float up[8], f[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) { up[c] = 0.0f; f[c] = 0.0f; }
                // transposed convolution: even rows take tap 1 of t2 row y/2; odd rows tap 0 of row (y+1)/2 and tap 2 of row (y-1)/2
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    if ((py == 0) != (ky == 1)) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        if ((px == 0) != (kx == 1)) continue;
                        const float* ip = tp + ((yy + (ky == 0 ? 1 : 0)) * TP + xx + (kx == 0 ? 1 : 0)) * TPP;
                        const float4 a = *reinterpret_cast<const float4*>(ip), b = *reinterpret_cast<const float4*>(ip + 4);
                        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                        const cfloat* wq = cwd + __builtin_amdgcn_readfirstlane((ky * 3 + kx) * 64);
#pragma unroll
                        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                            for (int c = 0; c < 8; ++c) up[c] = fmaf(v[ci], wq[ci * 8 + c], up[c]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // conv0 on the image: input rows r + ky of the patch (origin oy0 - 3), columns q + kx + 1 (origin ox0 - 4)
                const float* xp = xin + r * IC + q + 1;
#pragma unroll 1
                for (int ky = 0; ky < 3; ++ky) {
                    const cfloat* wq = cw0 + __builtin_amdgcn_readfirstlane(ky * 72);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int ci = 0; ci < 3; ++ci) {
                            const float v = xp[(ci * IR + ky) * IC + kx];
#pragma unroll
                            for (int c = 0; c < 8; ++c) f[c] = fmaf(v, wq[(kx * 3 + ci) * 8 + c], f[c]);
                        }
                }
#pragma unroll
                for (int c = 0; c < 8; c += 2) {
                    const f32x2_t a = {fmaxf(up[c] + csd[c], 0.0f), fmaxf(up[c + 1] + csd[c + 1], 0.0f)};
                    const f32x2_t b = {fmaxf(f[c] + cs0[c], 0.0f), fmaxf(f[c + 1] + cs0[c + 1], 0.0f)};
                    const f16x2_t ah = __builtin_convertvector(a, f16x2_t), bh = __builtin_convertvector(b, f16x2_t);
                    const f16x2_t al = __builtin_convertvector((a - __builtin_convertvector(ah, f32x2_t)) * RF_LO_SCALE, f16x2_t);
                    const f16x2_t bl = __builtin_convertvector((b - __builtin_convertvector(bh, f32x2_t)) * RF_LO_SCALE, f16x2_t);
                    h0[c] = ah[0]; h0[c + 1] = ah[1]; l0[c] = al[0]; l0[c + 1] = al[1];
                    h1[c] = bh[0]; h1[c + 1] = bh[1]; l1[c] = bl[0]; l1[c + 1] = bl[1];
                }
            }
            _Float16* ph = xh + (r * XROWP + q) * XPIX;
            *reinterpret_cast<f16x8*>(ph) = h0;
            *reinterpret_cast<f16x8*>(ph + 8) = h1;
            *reinterpret_cast<f16x8*>(ph + XPLANE) = l0;
            *reinterpret_cast<f16x8*>(ph + XPLANE + 8) = l1;
        }
    }
    __syncthreads();  // x16 planes complete; the input patches are dead: their region becomes the conv3 map

    // ---- (3) conv3 on the 18 x 18 patch (global origin (oy0 - 1, ox0 - 1)): M-tile t = pixels 16 t .. 16 t + 15 in linear order ------
    {
        const int ntile = wave < 3 ? 6 : 3;  // 21 tiles: waves 0-2 six each, wave 3 (four-tap parity class in phase 2) three
        const f32x4_t sh = *reinterpret_cast<const f32x4_t*>(s3 + 4 * (kb & 1));
#pragma unroll 1
        for (int g = 0; g < ntile; g += 3) {  // three tiles at a time: three independent accumulator chains
            f32x4_t accM[3], accL[3];
            const _Float16* pb[3];
            int mm[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                accM[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                accL[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const int m = min((wave * 6 + g + j) * 16 + li, MW * MW - 1);
                mm[j] = m;
                const int r = m / MW, q = m - r * MW;
                pb[j] = xh + (r * XROWP + q) * XPIX;
            }
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                int q = 4 * ks + kb;
                q = q < 17 ? q : 17;  // padding blocks 18, 19 (zero weights) read block 17
                const int tap = q >> 1, cb = q & 1, dy = tap / 3, dx = tap - dy * 3;
                const int off = (dy * XROWP + dx) * XPIX + cb * 8;
                f16x8 bh[3], bl[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    bh[j] = *reinterpret_cast<const f16x8*>(pb[j] + off);
                    bl[j] = *reinterpret_cast<const f16x8*>(pb[j] + off + XPLANE);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) accM[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][0], bh[j], accM[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 3; ++j) accL[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][0], bl[j], accL[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 3; ++j) accL[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks][1], bh[j], accL[j], 0, 0, 0);
            }
            // D rows 4 kb + e = output channels: lanes with kb < 2 hold channels [4 kb, 4 kb + 4) of pixel mm[j]
            if (kb < 2) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int t = wave * 6 + g + j;
                    const int m = t * 16 + li;
                    if (m < MW * MW) {
                        const int r = mm[j] / MW, q = mm[j] - r * MW;
                        const int gy = oy0 - 1 + r, gx = ox0 - 1 + q;
                        f32x4_t v = f32x4_t{0.f, 0.f, 0.f, 0.f};
                        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                            v = accM[j] + accL[j] * (1.0f / RF_LO_SCALE) + sh;
                            v = __builtin_elementwise_max(v, f32x4_t{0.f, 0.f, 0.f, 0.f});
                        }
                        *reinterpret_cast<f32x4_t*>(mid + r * MRP + q * MP + 4 * kb) = v;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- (4) res (8 -> 1, no bias) + nearest-x2 residual + de-normalisation: refine_tail_kernel's last phase ------------------------
    const int tx = tid % TW, ty = tid / TW;
    float res = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* mp = mid + (ty + ky) * MRP + (tx + kx) * MP;
            const float4 a = *reinterpret_cast<const float4*>(mp), b = *reinterpret_cast<const float4*>(mp + 4);
            const cfloat* wq = cwr + (ky * 3 + kx) * 8;
            res = fmaf(a.x, wq[0], res); res = fmaf(a.y, wq[1], res); res = fmaf(a.z, wq[2], res); res = fmaf(a.w, wq[3], res);
            res = fmaf(b.x, wq[4], res); res = fmaf(b.y, wq[5], res); res = fmaf(b.z, wq[6], res); res = fmaf(b.w, wq[7], res);
        }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= H || ox >= W) return;
    {
#pragma clang fp contract(off)
        const float lo = dmin[n], span = dmax[n] - lo;
        const float d = dnorm[((size_t)n * (H / 2) + (oy >> 1)) * (W / 2) + (ox >> 1)] + res;  // nearest x2 + residual (net.py:119)
        out[((size_t)n * H + oy) * W + ox] = d * span + lo;                                     // net.py:122
    }
}

// img [B,3,H,W] planar; t2 [B,H/2,W/2,8] channels-last; w0 / s0, wd / sd as pmn_refine_front; w3a DEVICE fp16 [5][2][64][8]
// (params.pack_refine_conv3_f16s: conv3's weights as MFMA A operands, hi | lo, BatchNorm folded); s3 [8]; wr [3][3][8]; dnorm
// [B,1,H/2,W/2]; depth_min / depth_max DEVICE float[B] -> out [B,1,H,W]
extern "C" int pmn_refine_fused(const float* img, const float* t2, const float* w0, const float* s0, const float* wd, const float* sd,
                            const void* w3a, const float* s3, const float* wr, const float* dnorm, const float* depth_min,
                            const float* depth_max, float* out, int B, int H, int W, void* stream) {
    if (!img || !t2 || !w0 || !s0 || !wd || !sd || !w3a || !s3 || !wr || !dnorm || !depth_min || !depth_max || !out || B < 1 ||
        H < 2 || W < 2 || (H & 1) || (W & 1))
        return PMN_ERR_ARG;
    const size_t lds = (size_t)2 * 20 * 26 * 16 * sizeof(_Float16) + (size_t)18 * 256 * sizeof(float);  // 33,280 + 18,432 B
    const int blocks = B * ((W + 15) / 16) * ((H + 15) / 16);
    const bool vec4 = W % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0;
    const void* kern = vec4 ? reinterpret_cast<const void*>(refine_fused_kernel<true>) : reinterpret_cast<const void*>(refine_fused_kernel<false>);
    if (pmn_raise_dynamic_lds(kern, lds) != PMN_OK) return PMN_ERR_LAUNCH;
    if (vec4)
        PMN_LAUNCH(refine_fused_kernel<true>, dim3(blocks), dim3(PMN_BLOCK), lds, (hipStream_t)stream, img, t2, w0, s0, wd, sd,
                           reinterpret_cast<const f16x8*>(w3a), s3, wr, dnorm, depth_min, depth_max, out, B, H, W);
    else
        PMN_LAUNCH(refine_fused_kernel<false>, dim3(blocks), dim3(PMN_BLOCK), lds, (hipStream_t)stream, img, t2, w0, s0, wd, sd,
                           reinterpret_cast<const f16x8*>(w3a), s3, wr, dnorm, depth_min, depth_max, out, B, H, W);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}
