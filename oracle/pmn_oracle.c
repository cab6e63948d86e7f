/*
 * pmn_oracle.c -- CPU restatement (plain C, fp32) of the arithmetic on the learned-PatchMatch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP kernels in
 * patchmatchnet_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing in the product path links, imports or calls anything in oracle/.
 *
 * Each function cites the reference lines it follows (paths relative to the PatchmatchNet reference
 * checkout) and, where the arithmetic lives in PyTorch itself, the ATen semantics it restates
 * (ATen/native/GridSampler.h: grid_sampler_unnormalize / clip_coordinates / bilinear corner weights).
 *
 * Conventions: tensors are contiguous NCHW slices of ONE batch element; all arithmetic is IEEE fp32
 * evaluated in the order written (build with -ffp-contract=off).  Loops are OpenMP-parallel over pixels
 * so the same code can serve as the "port" CPU baseline of bench.py.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int pmo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void pmo_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- F.grid_sample coordinate helpers (ATen GridSampler.h semantics) ---------------------------------- */

/* grid_sampler_unnormalize: [-1,1] -> pixel index space */
static inline float gs_unnormalize(float coord, int size, int align_corners) {
    if (align_corners) {
        return ((coord + 1.0f) / 2.0f) * (float)(size - 1);
    }
    return ((coord + 1.0f) * (float)size - 1.0f) / 2.0f;
}

/* clip_coordinates (padding_mode="border") */
static inline float gs_clip(float v, int size) {
    float hi = (float)(size - 1);
    v = v > 0.0f ? v : 0.0f; /* max(v, 0)  */
    return v < hi ? v : hi;  /* min(., hi) */
}

/* Bilinear corner set for an un-normalised position. */
typedef struct {
    int x0, y0;            /* north-west corner (floor) */
    float nw, ne, sw, se;  /* corner weights, ATen order */
} bil_t;

static inline bil_t bil_setup(float ix, float iy) {
    bil_t b;
    float fx = floorf(ix), fy = floorf(iy);
    float x1 = fx + 1.0f, y1 = fy + 1.0f;
    b.x0 = (int)fx;
    b.y0 = (int)fy;
    b.nw = (x1 - ix) * (y1 - iy);
    b.ne = (ix - fx) * (y1 - iy);
    b.sw = (x1 - ix) * (iy - fy);
    b.se = (ix - fx) * (iy - fy);
    return b;
}

/* zeros padding: a corner contributes only when in bounds; accumulation order nw, ne, sw, se */
static inline float bil_fetch_zeros(const float *plane, int h, int w, const bil_t *b) {
    float out = 0.0f;
    int x0 = b->x0, y0 = b->y0, x1 = x0 + 1, y1 = y0 + 1;
    int x0in = (x0 >= 0 && x0 < w), x1in = (x1 >= 0 && x1 < w);
    int y0in = (y0 >= 0 && y0 < h), y1in = (y1 >= 0 && y1 < h);
    if (x0in && y0in) out += plane[(size_t)y0 * w + x0] * b->nw;
    if (x1in && y0in) out += plane[(size_t)y0 * w + x1] * b->ne;
    if (x0in && y1in) out += plane[(size_t)y1 * w + x0] * b->sw;
    if (x1in && y1in) out += plane[(size_t)y1 * w + x1] * b->se;
    return out;
}

/* ---- a2: differentiable_warping (models/module.py:130-181) --------------------------------------------- */

/* Un-normalised source-image sample position of reference pixel (x,y) at depth d.
 * module.py:161-164  p = (rot @ [x,y,1]) * d + trans
 * module.py:166-169  p.z <= 1e-3  ->  p = (w, h, 1)     (w,h = REFERENCE map size)
 * module.py:170-173  g = p.xy / p.z ; normalise with (w-1)/2, (h-1)/2
 * module.py:175-181  grid_sample(align_corners=True) un-normalises with the SOURCE map size */
static inline void warp_position(const float *rot, const float *trans, float x, float y, float d,
                                 int h, int w, int hs, int ws, float *ix, float *iy) {
    float rx = rot[0] * x + rot[1] * y + rot[2];
    float ry = rot[3] * x + rot[4] * y + rot[5];
    float rz = rot[6] * x + rot[7] * y + rot[8];
    float px = rx * d + trans[0];
    float py = ry * d + trans[1];
    float pz = rz * d + trans[2];
    if (pz <= 1e-3f) {
        px = (float)w;
        py = (float)h;
        pz = 1.0f;
    }
    float gx = px / pz, gy = py / pz;
    float xn = gx / ((float)(w - 1) / 2.0f) - 1.0f;
    float yn = gy / ((float)(h - 1) / 2.0f) - 1.0f;
    *ix = gs_unnormalize(xn, ws, 1);
    *iy = gs_unnormalize(yn, hs, 1);
}

/* Sample positions only (debug / unit tests): out_ix, out_iy [D,h,w]. */
void pmo_warp_positions(const float *rot, const float *trans, const float *depth, int D, int h, int w,
                        int hs, int ws, float *out_ix, float *out_iy) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int d = 0; d < D; ++d) {
                size_t o = ((size_t)d * h + y) * w + x;
                warp_position(rot, trans, (float)x, (float)y, depth[o], h, w, hs, ws, &out_ix[o], &out_iy[o]);
            }
}

/* warped [C,D,h,w] = bilinear/zeros sample of src [C,hs,ws]   (module.py:130-181) */
void pmo_differentiable_warping(const float *src, const float *rot, const float *trans, const float *depth,
                                int C, int D, int h, int w, int hs, int ws, float *warped) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int d = 0; d < D; ++d) {
                size_t o = ((size_t)d * h + y) * w + x;
                float ix, iy;
                warp_position(rot, trans, (float)x, (float)y, depth[o], h, w, hs, ws, &ix, &iy);
                bil_t b = bil_setup(ix, iy);
                for (int c = 0; c < C; ++c)
                    warped[(size_t)c * D * h * w + o] = bil_fetch_zeros(src + (size_t)c * hs * ws, hs, ws, &b);
            }
}

/* a2+a3: per-view group-wise correlation  sim[G,D,h,w] = mean_{c in group}(warped * ref)
 * (models/patchmatch.py:193,199-203; the warped volume is never materialised here) */
void pmo_warp_similarity(const float *ref, const float *src, const float *rot, const float *trans,
                         const float *depth, int C, int G, int D, int h, int w, int hs, int ws, float *sim) {
    const int cg = C / G;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int d = 0; d < D; ++d) {
                size_t o = ((size_t)d * h + y) * w + x;
                float ix, iy;
                warp_position(rot, trans, (float)x, (float)y, depth[o], h, w, hs, ws, &ix, &iy);
                bil_t b = bil_setup(ix, iy);
                for (int g = 0; g < G; ++g) {
                    float acc = 0.0f;
                    for (int k = 0; k < cg; ++k) {
                        int c = g * cg + k;
                        float wv = bil_fetch_zeros(src + (size_t)c * hs * ws, hs, ws, &b);
                        acc += wv * ref[((size_t)c * h + y) * w + x];
                    }
                    sim[(size_t)g * D * h * w + o] = acc / (float)cg;
                }
            }
}

/* ---- a15 / a4 / a6 / a11: pointwise 1x1x1 MLP  G -> 16 -> 8 -> 1 ---------------------------------------- */
/* ConvBnReLU3D (models/module.py:43-72) x2 then Conv3d(8,1,1)+bias (patchmatch.py:547-549, 597-599,
 * 690-692); BatchNorm3d in eval mode: (x-mean)/sqrt(var+eps)*gamma+beta, eps=1e-5.
 * x [G,M] -> out [M].  bn arrays hold gamma | beta | running_mean | running_var, each n long. */
void pmo_pointwise_mlp(const float *x, int G, int64_t M, const float *w0, const float *bn0, const float *w1,
                       const float *bn1, const float *w2, float b2, float eps, int apply_sigmoid, float *out) {
    float inv0[16], inv1[8];
    for (int j = 0; j < 16; ++j) inv0[j] = 1.0f / sqrtf(bn0[48 + j] + eps);
    for (int j = 0; j < 8; ++j) inv1[j] = 1.0f / sqrtf(bn1[24 + j] + eps);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        float h0[16], h1[8];
        for (int j = 0; j < 16; ++j) {
            float acc = 0.0f;
            for (int g = 0; g < G; ++g) acc += w0[j * G + g] * x[(size_t)g * M + m];
            acc = (acc - bn0[32 + j]) * inv0[j] * bn0[j] + bn0[16 + j];
            h0[j] = acc > 0.0f ? acc : 0.0f;
        }
        for (int j = 0; j < 8; ++j) {
            float acc = 0.0f;
            for (int i = 0; i < 16; ++i) acc += w1[j * 16 + i] * h0[i];
            acc = (acc - bn1[16 + j]) * inv1[j] * bn1[j] + bn1[8 + j];
            h1[j] = acc > 0.0f ? acc : 0.0f;
        }
        float acc = 0.0f;
        for (int i = 0; i < 8; ++i) acc += w2[i] * h1[i];
        acc += b2;
        out[m] = apply_sigmoid ? 1.0f / (1.0f + expf(-acc)) : acc;
    }
}

/* ---- a10 + neighbour gathers: get_grid (patchmatch.py:314-426) + grid_sample(border, align_corners=False)
 *      at patchmatch.py:117-123 (propagation), :569-575 (SimilarityNet), :615-617 (FeatureWeightNet),
 *      :659-661 (depth_weight). ------------------------------------------------------------------------- */

/* Position of neighbour k of pixel (x,y):
 *   patchmatch.py:409-412  X = x + dx_k + offset[2k],  Y = y + dy_k + offset[2k+1]   (table entries are [dy,dx])
 *   patchmatch.py:420-421  xn = X/((w-1)/2) - 1 ,  yn = Y/((h-1)/2) - 1
 *   grid_sample(align_corners=False): ix = ((xn+1)*w - 1)/2, then border clip to [0, w-1] */
static inline void neighbor_position(const float *offs, const int *base, int k, int h, int w, int y, int x,
                                     float *ix, float *iy) {
    size_t hw = (size_t)h * w, p = (size_t)y * w + x;
    float X = (float)x + ((float)base[2 * k + 1] + offs[(size_t)(2 * k) * hw + p]);
    float Y = (float)y + ((float)base[2 * k + 0] + offs[(size_t)(2 * k + 1) * hw + p]);
    float xn = X / ((float)(w - 1) / 2.0f) - 1.0f;
    float yn = Y / ((float)(h - 1) / 2.0f) - 1.0f;
    *ix = gs_clip(gs_unnormalize(xn, w, 0), w);
    *iy = gs_clip(gs_unnormalize(yn, h, 0), h);
}

void pmo_neighbor_positions(const float *offs, const int *base, int K, int h, int w, float *out_ix,
                            float *out_iy) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int k = 0; k < K; ++k) {
                size_t o = ((size_t)k * h + y) * w + x;
                neighbor_position(offs, base, k, h, w, y, x, &out_ix[o], &out_iy[o]);
            }
}

/* out[Cn,K,h,w] = bilinear/border sample of in[Cn,h,w] at the K neighbour positions of every pixel.
 * With border padding the clipped position always has its NW corner in bounds; the +1 corners may fall at
 * index == size (weight 0) and are skipped exactly as ATen's within_bounds test does. */
void pmo_neighbor_gather(const float *in, const float *offs, const int *base, int Cn, int K, int h, int w,
                         float *out) {
    size_t hw = (size_t)h * w;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int k = 0; k < K; ++k) {
                float ix, iy;
                neighbor_position(offs, base, k, h, w, y, x, &ix, &iy);
                bil_t b = bil_setup(ix, iy);
                for (int c = 0; c < Cn; ++c)
                    out[((size_t)c * K + k) * hw + (size_t)y * w + x] = bil_fetch_zeros(in + c * hw, h, w, &b);
            }
}

/* ---- element-wise glue of one PatchMatch iteration, OpenMP-parallel over pixels ------------------------------------------
 * The same IEEE fp32 operations, in the same order, as the numpy expressions they replaced in oracle.py (reductions over
 * a non-innermost axis add slab by slab, i.e. sequentially in the reduced index); kept in C so that bench.py's
 * cpu_baseline scales with the host's cores instead of idling on serial numpy temporaries. */

/* a11 (patchmatch.py:618-620): corr[G,K,hw] = mean over the C/G channels of a group of nb[C,K,hw] * ref[C,hw] */
void pmo_feature_corr(const float *nb, const float *ref, int C, int G, int K, int64_t hw, float *out) {
    const int cg = C / G;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < hw; ++p)
        for (int g = 0; g < G; ++g)
            for (int k = 0; k < K; ++k) {
                float acc = 0.0f;
                for (int i = 0; i < cg; ++i) {
                    int c = g * cg + i;
                    acc += nb[((size_t)c * K + k) * hw + p] * ref[(size_t)c * hw + p];
                }
                out[((size_t)g * K + k) * hw + p] = acc / (float)cg;
            }
}

/* a12 (patchmatch.py:650-669): x = (1/ds - 1/dmax) / (1/dmin - 1/dmax) for ds[D,hw] */
void pmo_normalised_inverse_depth(const float *ds, int64_t n, float inv_min, float inv_max, float *x) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] = (1.0f / ds[i] - inv_max) / (inv_min - inv_max);
}

/* a12 (patchmatch.py:662-669): w[D,K,hw] = sigmoid(4 - 2 * clamp(|x1 - x| / interval, 0, 4)), x1[D,K,hw], x[D,hw] */
void pmo_depth_weight(const float *x, const float *x1, int D, int K, int64_t hw, float interval, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < hw; ++p)
        for (int d = 0; d < D; ++d)
            for (int k = 0; k < K; ++k) {
                size_t o = ((size_t)d * K + k) * hw + p;
                float v = fabsf(x1[o] - x[(size_t)d * hw + p]) / interval;
                v = v < 0.0f ? 0.0f : (v > 4.0f ? 4.0f : v);
                float z = 4.0f - 2.0f * v;
                out[o] = 1.0f / (1.0f + expf(-z));
            }
}

/* a13 (patchmatch.py:502-510): weight = depth_weight * feature_weight; weight /= sum_k weight.   dw[D,K,hw] in place, fw[K,hw] */
void pmo_weight_normalise(float *dw, const float *fw, int D, int K, int64_t hw) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < hw; ++p)
        for (int d = 0; d < D; ++d) {
            float s = 0.0f;
            for (int k = 0; k < K; ++k) {
                size_t o = ((size_t)d * K + k) * hw + p;
                dw[o] = dw[o] * fw[(size_t)k * hw + p];
                s = k == 0 ? dw[o] : s + dw[o];
            }
            for (int k = 0; k < K; ++k) dw[((size_t)d * K + k) * hw + p] /= s;
        }
}

/* a6 (patchmatch.py:576-577): out[D,hw] = sum_k nb[D,K,hw] * weight[D,K,hw] */
void pmo_weighted_neighbor_sum(const float *nb, const float *weight, int D, int K, int64_t hw, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < hw; ++p)
        for (int d = 0; d < D; ++d) {
            float s = 0.0f;
            for (int k = 0; k < K; ++k) {
                size_t o = ((size_t)d * K + k) * hw + p;
                float t = nb[o] * weight[o];
                s = k == 0 ? t : s + t;
            }
            out[(size_t)d * hw + p] = s;
        }
}

/* a5 (patchmatch.py:209-217): sim_sum[GD,hw] += sim[GD,hw] * vw[hw]; weight_sum[hw] += vw[hw] */
void pmo_view_accumulate(const float *sim, const float *vw, int GD, int64_t hw, float *sim_sum, float *weight_sum) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < hw; ++p) {
        float v = vw[p];
        for (int j = 0; j < GD; ++j) sim_sum[(size_t)j * hw + p] += sim[(size_t)j * hw + p] * v;
        weight_sum[p] += v;
    }
}

/* a5 (patchmatch.py:223-224): similarity = sim_sum / weight_sum, in place */
void pmo_view_normalise(float *sim_sum, const float *weight_sum, int GD, int64_t hw) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < hw; ++p)
        for (int j = 0; j < GD; ++j) sim_sum[(size_t)j * hw + p] /= weight_sum[p];
}
