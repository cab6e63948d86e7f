// fusion.hip -- photometric + geometric consistency of one reference view against its source views and the fused 3-D
// points, in one kernel.
//
// Reference: eval.py:86-145 (reproject_with_depth), :148-190 (check_geometric_consistency), :207-281 (filter_depth: photo
// mask, geometric mask sum, averaged depth, final mask, world points).  The reference does this per (ref, src) pair in
// single-threaded numpy + cv2.remap over maps re-read from disk; here a thread owns one reference pixel and walks the source
// views with the running mask count / depth sum in registers, reading the per-scan slot buffer that the all-gather
// leaves in HBM (slot v = depth, confidence of view v, each view at its own size).  The numeric types follow the reference's numpy dtype flow (float32 camera matrices promoted to float64 in
// the products, float32 casts of the map coordinates / re-projected depth and positions, float32 threshold on the relative
// depth difference, float64 on the pixel distance) and cv2.remap's INTER_LINEAR is restated with its 1/32-pixel fixed-point
// coordinates (OpenCV imgwarp.cpp, INTER_BITS = 5; oracle/fusion_oracle.py carries the same restatement).
#include <cstring>

#include "pmn_common.hpp"

#define PMN_FUSE_REF_FLOATS 48
#define PMN_FUSE_SRC_FLOATS 64

struct FuseArgs {
    const float* maps;     // slot v = (depth [h_v][w_v], confidence [h_v][w_v]) packed at the start of the slot
    long long slot_stride; // floats between slots (>= 2*h*w of the largest view)
    const float* mats;     // device floats: ref block (48) + n_src blocks (64), layout in include/pmn_hip.h
    unsigned char* masks;  // [3][H][W]: photo, geo, final
    float* xyz;            // [H][W][3] world points (meaningful where final)
    double* depth_avg;     // [H][W] or null
    int* geo_sum;          // [H][W] or null
    int ref_slot, n_src, H, W, geo_mask_thres;
    float geo_pixel_thres, geo_depth_thres, photo_thres;
    int src_slot[PMN_MAX_FUSE_SRC];
    int src_h[PMN_MAX_FUSE_SRC], src_w[PMN_MAX_FUSE_SRC];  // every source map at its OWN size (reference eval.py:236-237 reads
                                                           // each view's file as it is; cv2.remap samples it at that size)
};

// cv2.remap(src, x, y, INTER_LINEAR), float32 single channel, BORDER_CONSTANT 0
__device__ __forceinline__ float remap_linear_cv2(const float* __restrict__ src, int H, int W, float x, float y) {
#pragma clang fp contract(off)
    const float mx = x * 32.0f, my = y * 32.0f;
    // (written so that NaN fails the test: a NaN coordinate samples nothing)
    if (!(fabsf(mx) < 1073741824.0f) || !(fabsf(my) < 1073741824.0f)) return 0.0f;
    const int sx = (int)rintf(mx), sy = (int)rintf(my);  // cvRound: to nearest, ties to even
    const int ix = sx >> 5, iy = sy >> 5;
    const float fx = (float)(sx & 31) * 0.03125f, fy = (float)(sy & 31) * 0.03125f;
    const float w00 = (1.0f - fy) * (1.0f - fx), w01 = (1.0f - fy) * fx, w10 = fy * (1.0f - fx), w11 = fy * fx;
    const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W, y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
    const float t00 = (x0 && y0) ? src[(size_t)iy * W + ix] : 0.0f;
    const float t01 = (x1 && y0) ? src[(size_t)iy * W + ix + 1] : 0.0f;
    const float t10 = (x0 && y1) ? src[(size_t)(iy + 1) * W + ix] : 0.0f;
    const float t11 = (x1 && y1) ? src[(size_t)(iy + 1) * W + ix + 1] : 0.0f;
    float out = t00 * w00;
    out = out + t01 * w01;
    out = out + t10 * w10;
    out = out + t11 * w11;
    return out;
}

// rows of a row-major float32 matrix times a float64 vector (numpy: the float32 matrix is promoted, the product is float64)
__device__ __forceinline__ double dot3(const float* __restrict__ m, double a, double b, double c) {
    return fma((double)m[2], c, fma((double)m[1], b, (double)m[0] * a));
}
__device__ __forceinline__ double dot4(const float* __restrict__ m, double a, double b, double c, double d) {
    return fma((double)m[3], d, fma((double)m[2], c, fma((double)m[1], b, (double)m[0] * a)));
}

__global__ __launch_bounds__(256) void fuse_view_kernel(const FuseArgs a) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.W || y >= a.H) return;
    const size_t hw = (size_t)a.H * a.W, p = (size_t)y * a.W + x;
    const float* ref = a.maps + (size_t)a.ref_slot * a.slot_stride;
    const float d_ref = ref[p], conf = ref[hw + p];
    const float* Kri = a.mats;        // inverse(K_ref), float32 as numpy computes it
    const float* Kr = a.mats + 9;     // K_ref
    const float* Eri = a.mats + 18;   // inverse(E_ref) [4][4]
    // reference 3-D point: inverse(K_ref) @ ((x, y, 1) * depth)   (int64 grid * float32 depth -> float64)
    const double dx = (double)x * (double)d_ref, dy = (double)y * (double)d_ref, dz = (double)d_ref;
    const double rx = dot3(Kri, dx, dy, dz), ry = dot3(Kri + 3, dx, dy, dz), rz = dot3(Kri + 6, dx, dy, dz);

    int geo_sum = 0;
    float acc = 0.0f;
    for (int s = 0; s < a.n_src; ++s) {
        const float* M = a.mats + PMN_FUSE_REF_FLOATS + s * PMN_FUSE_SRC_FLOATS;
        const float* T = M;          // E_src @ inverse(E_ref)   (float32 product, as numpy)
        const float* Ks = M + 16;    // K_src
        const float* Ksi = M + 25;   // inverse(K_src)
        const float* T2 = M + 34;    // E_ref @ inverse(E_src)
        const float* src = a.maps + (size_t)a.src_slot[s] * a.slot_stride;
        // source 3-D point and pixel
        const double sx = dot4(T, rx, ry, rz, 1.0), sy = dot4(T + 4, rx, ry, rz, 1.0), sz = dot4(T + 8, rx, ry, rz, 1.0);
        const double kx = dot3(Ks, sx, sy, sz), ky = dot3(Ks + 3, sx, sy, sz), kz = dot3(Ks + 6, sx, sy, sz);
        const double xs = kx / kz, ys = ky / kz;
        const float sampled = remap_linear_cv2(src, a.src_h[s], a.src_w[s], (float)xs, (float)ys);
        // back-project with the SAMPLED source depth (float64 coordinates, as the reference)
        const double bx = xs * (double)sampled, by = ys * (double)sampled, bz = (double)sampled;
        const double qx = dot3(Ksi, bx, by, bz), qy = dot3(Ksi + 3, bx, by, bz), qz = dot3(Ksi + 6, bx, by, bz);
        const double wx = dot4(T2, qx, qy, qz, 1.0), wy = dot4(T2 + 4, qx, qy, qz, 1.0), wz = dot4(T2 + 8, qx, qy, qz, 1.0);
        const float depth_rep = (float)wz;
        const double px = dot3(Kr, wx, wy, wz), py = dot3(Kr + 3, wx, wy, wz), pz = dot3(Kr + 6, wx, wy, wz);
        const float x_rep = (float)(px / pz), y_rep = (float)(py / pz);
        const double ex = (double)x_rep - (double)x, ey = (double)y_rep - (double)y;
        const double dist = sqrt(ex * ex + ey * ey);
        const float rel = fabsf(depth_rep - d_ref) / d_ref;
        const bool m = dist < (double)a.geo_pixel_thres && rel < a.geo_depth_thres;
        geo_sum += m ? 1 : 0;
        acc = acc + (m ? depth_rep : 0.0f);
    }
    const float total = acc + d_ref;  // float32 sums, as python's sum() over float32 arrays
    const double avg = (double)total / (double)(geo_sum + 1);
    const bool photo = conf > a.photo_thres, geo = geo_sum >= a.geo_mask_thres, fin = photo && geo;
    a.masks[p] = photo;
    a.masks[hw + p] = geo;
    a.masks[2 * hw + p] = fin;
    if (a.depth_avg) a.depth_avg[p] = avg;
    if (a.geo_sum) a.geo_sum[p] = geo_sum;
    // world point of the averaged depth
    const double ax = (double)x * avg, ay = (double)y * avg;
    const double cx = dot3(Kri, ax, ay, avg), cy = dot3(Kri + 3, ax, ay, avg), cz = dot3(Kri + 6, ax, ay, avg);
    a.xyz[3 * p + 0] = (float)dot4(Eri, cx, cy, cz, 1.0);
    a.xyz[3 * p + 1] = (float)dot4(Eri + 4, cx, cy, cz, 1.0);
    a.xyz[3 * p + 2] = (float)dot4(Eri + 8, cx, cy, cz, 1.0);
}

extern "C" int pmn_fuse_view(const float* maps, long long slot_stride, int ref_slot, const int* src_slots_host,
                             const int* src_hw_host, int n_src, const float* mats, int H, int W, float geo_pixel_thres, float geo_depth_thres, int geo_mask_thres,
                             float photo_thres, unsigned char* masks, float* xyz, double* depth_avg, int* geo_sum,
                             void* stream) {
    if (!maps || !mats || !masks || !xyz || (n_src > 0 && !src_slots_host)) return PMN_ERR_ARG;
    if (H < 1 || W < 1 || n_src < 0 || ref_slot < 0 || slot_stride < 2LL * H * W) return PMN_ERR_ARG;
    if (n_src > PMN_MAX_FUSE_SRC) return PMN_ERR_SHAPE;
    FuseArgs a;
    memset(&a, 0, sizeof(a));
    a.maps = maps;
    a.slot_stride = slot_stride;
    a.mats = mats;
    a.masks = masks;
    a.xyz = xyz;
    a.depth_avg = depth_avg;
    a.geo_sum = geo_sum;
    a.ref_slot = ref_slot;
    a.n_src = n_src;
    a.H = H;
    a.W = W;
    a.geo_mask_thres = geo_mask_thres;
    a.geo_pixel_thres = geo_pixel_thres;
    a.geo_depth_thres = geo_depth_thres;
    a.photo_thres = photo_thres;
    for (int i = 0; i < n_src; ++i) {
        if (src_slots_host[i] < 0) return PMN_ERR_ARG;
        a.src_slot[i] = src_slots_host[i];
        a.src_h[i] = src_hw_host ? src_hw_host[2 * i] : H;
        a.src_w[i] = src_hw_host ? src_hw_host[2 * i + 1] : W;
        if (a.src_h[i] < 1 || a.src_w[i] < 1 || slot_stride < 2LL * a.src_h[i] * a.src_w[i]) return PMN_ERR_ARG;
    }
    PMN_LAUNCH(fuse_view_kernel, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, (hipStream_t)stream, a);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}

// ---- point packing (round 5) --------------------------------------------------------------------------------------------
// The host half of reference eval.py:270-281 on the device: the pixels that survive the final mask become PLY vertex records
// (x, y, z little-endian float32 + red, green, blue uint8: the 15 bytes plyfile writes per vertex, eval.py:283-297) in row-major
// pixel order, appended to a per-scan record buffer at a device-resident cursor -- so a scan's fused.ply body is ONE contiguous
// device buffer that leaves through pinned memory in large chunks, instead of per view: boolean index on the device (a
// synchronising nonzero), two variable-size downloads, numpy's boolean index of the reference image and two strided byte copies.
// Three launches per view on the caller's stream: per-block counts (1024 pixels per block), an exclusive scan of the block counts
// that also advances the cursor, and the pack.  Colours: the decoded image bytes (uint8 [H,W,3]), or for a resized float image in
// [0,1] the reference's (color * 255).astype(uint8) = truncation of the float32 product.
#define PMN_PACK_PIX 4                        // consecutive pixels per thread
#define PMN_PACK_BLOCK (256 * PMN_PACK_PIX)   // pixels per workgroup

__device__ __forceinline__ int pack_flags(const unsigned char* __restrict__ mask, long long p0, long long n, bool (&keep)[PMN_PACK_PIX]) {
    int c = 0;
#pragma unroll
    for (int j = 0; j < PMN_PACK_PIX; ++j) {
        keep[j] = (p0 + j < n) && mask[p0 + j] != 0;
        c += keep[j] ? 1 : 0;
    }
    return c;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void pack_count_kernel(const unsigned char* __restrict__ mask, long long n, long long* __restrict__ blocks) {
    __shared__ int part[4];
    bool keep[PMN_PACK_PIX];
    const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * PMN_PACK_PIX;
    const int c = wave_sum(pack_flags(mask, p0, n, keep));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blocks[blockIdx.x] = (long long)(part[0] + part[1] + part[2] + part[3]);
}

// blocks[i] (counts) -> absolute record index of block i's first point; *cursor += total unless the buffer would overflow
// (then the view's count is reported as -1 and nothing is packed)
__global__ __launch_bounds__(1024) void pack_scan_kernel(long long* __restrict__ blocks, int nblocks, long long* __restrict__ cursor,
                                                         long long capacity, int* __restrict__ view_count) {
    __shared__ long long tmp[1024];
    __shared__ long long carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    const long long base = *cursor;
    for (int c0 = 0; c0 < nblocks; c0 += 1024) {
        const int i = c0 + t;
        const long long mine = i < nblocks ? blocks[i] : 0;
        tmp[t] = mine;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
            const long long add = t >= o ? tmp[t - o] : 0;
            __syncthreads();
            tmp[t] += add;
            __syncthreads();
        }
        const long long incl = tmp[t], before = carry;
        if (i < nblocks) blocks[i] = base + before + incl - mine;
        __syncthreads();
        if (t == 1023) carry = before + incl;
        __syncthreads();
    }
    if (t == 0) {
        const long long total = carry;
        if (base + total > capacity) {
            *view_count = -1;
            blocks[0] = -1;  // the pack launch reads this flag
        } else {
            *view_count = (int)total;
            *cursor = base + total;
        }
    }
}

__global__ __launch_bounds__(256) void pack_write_kernel(const unsigned char* __restrict__ mask, const float* __restrict__ xyz,
                                                         const void* __restrict__ image, int image_is_float, long long n,
                                                         const long long* __restrict__ blocks, unsigned char* __restrict__ records) {
#pragma clang fp contract(off)
    __shared__ int part[4];
    if (blocks[0] < 0) return;  // overflow reported by the scan
    bool keep[PMN_PACK_PIX];
    const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * PMN_PACK_PIX;
    const int c = pack_flags(mask, p0, n, keep);
    // exclusive prefix of c over the workgroup's threads (= row-major pixel order): inclusive scan inside the wave, wave totals via LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    int before = incl - c;
    for (int k = 0; k < wv; ++k) before += part[k];
    long long r = blocks[blockIdx.x] + before;
#pragma unroll
    for (int j = 0; j < PMN_PACK_PIX; ++j) {
        if (!keep[j]) continue;
        const long long p = p0 + j;
        unsigned int word[4];
        word[0] = __float_as_uint(xyz[3 * p + 0]);
        word[1] = __float_as_uint(xyz[3 * p + 1]);
        word[2] = __float_as_uint(xyz[3 * p + 2]);
        unsigned int cr, cg, cb;
        if (image_is_float) {
            const float* im = static_cast<const float*>(image) + 3 * p;
            cr = (unsigned char)(im[0] * 255.0f);
            cg = (unsigned char)(im[1] * 255.0f);
            cb = (unsigned char)(im[2] * 255.0f);
        } else {
            const unsigned char* im = static_cast<const unsigned char*>(image) + 3 * p;
            cr = im[0];
            cg = im[1];
            cb = im[2];
        }
        word[3] = cr | (cg << 8) | (cb << 16);
        unsigned char* o = records + r * 15;
#pragma unroll
        for (int i = 0; i < 15; ++i) o[i] = (unsigned char)(word[i >> 2] >> (8 * (i & 3)));
        ++r;
    }
}

extern "C" int pmn_pack_points(const unsigned char* final_mask, const float* xyz, const void* image_hwc, int image_is_float, int H,
                               int W, unsigned char* records, long long capacity_points, long long* cursor, int* view_count,
                               long long* scratch, void* stream) {
    if (!final_mask || !xyz || !image_hwc || !records || !cursor || !view_count || !scratch) return PMN_ERR_ARG;
    if (H < 1 || W < 1 || capacity_points < 0) return PMN_ERR_ARG;
    const long long n = (long long)H * W;
    const long long nb = (n + PMN_PACK_BLOCK - 1) / PMN_PACK_BLOCK;
    if (nb > 0x7fffffffLL) return PMN_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    PMN_LAUNCH(pack_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, final_mask, n, scratch);
    PMN_CHECK_LAUNCH();
    PMN_LAUNCH(pack_scan_kernel, dim3(1), dim3(1024), 0, st, scratch, (int)nb, cursor, capacity_points, view_count);
    PMN_CHECK_LAUNCH();
    PMN_LAUNCH(pack_write_kernel, dim3((unsigned)nb), dim3(256), 0, st, final_mask, xyz, image_hwc, image_is_float, n,
                       (const long long*)scratch, records);
    PMN_CHECK_LAUNCH();
    return PMN_OK;
}
