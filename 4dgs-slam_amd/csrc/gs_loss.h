// gs_loss.h -- fused photometric + depth L1 loss of the SLAM mapping / tracking steps:
//   L = alpha * mean_{3,H,W}( w_rgb * |exp(a) * I + b - I_gt| ) + (1 - alpha) * mean_{H,W}( w_d * |D - D_gt| )
// which is what utils/slam_utils.py:252-364 (get_loss_mapping*) and, with w_rgb *= rendered opacity and w_d *= (opacity > 0.95),
// :57-173 (get_loss_tracking*) evaluate with ~20 elementwise / reduction torch kernels per view plus their autograd replay: every pixel mask there (rgb boundary threshold, valid depth, motion masks, the x2 weighting
// of dynamic regions) is a constant of the keyframe and folds into the two weight images w_rgb, w_d in {0, 1, 2}.
// One pass computes the loss, one pass (in backward, scaled by the upstream gradient read from device memory) writes
// dL/dI, dL/dD -- exactly the cotangents the rasterizer's backward consumes -- and the exposure gradients.
// Sums are two-level with a fixed order (per-block partials, then one block), so the value is reproducible.
#pragma once
#include "gs_device.h"

namespace gsr {

constexpr int LOSS_BLOCKS = 256, LOSS_THREADS = 256;

struct LossArgs {
    int N;                                   // pixels
    const float* image; const float* depth;  // rendered [3,N], [N]
    const float* gt_image; const float* gt_depth;
    const float* w_rgb; const float* w_depth;   // [N] each or nullptr (= 1)
    const float* exposure_a; const float* exposure_b;   // device scalars or nullptr (a = 0, b = 0)
    const float* opacity; float opacity_thr;            // rendered opacity [N] or nullptr: w_rgb *= opacity, w_depth *= (opacity > thr)
    float c_rgb, c_depth;                    // alpha / (3N), (1 - alpha) / N
};

__device__ __forceinline__ float block_sum_fixed(float v, float* s_tmp /*[LOSS_THREADS / 64]*/)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane_id() == 0) s_tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < LOSS_THREADS / 64; w++) t += s_tmp[w];
    __syncthreads();
    return t;
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }   // d|v|/dv as torch defines it

__global__ void __launch_bounds__(LOSS_THREADS) l1_loss_fwd_kernel(LossArgs a, float* __restrict__ partials)
{
    __shared__ float s_tmp[LOSS_THREADS / 64];
    const float ea = a.exposure_a ? expf(a.exposure_a[0]) : 1.f, eb = a.exposure_b ? a.exposure_b[0] : 0.f;
    float acc = 0.f;
    for (int p = blockIdx.x * LOSS_THREADS + threadIdx.x; p < a.N; p += LOSS_BLOCKS * LOSS_THREADS) {
        float wr = a.w_rgb ? a.w_rgb[p] : 1.f, wd = a.w_depth ? a.w_depth[p] : 1.f;
        if (a.opacity) { const float op = a.opacity[p]; wr *= op; wd = op > a.opacity_thr ? wd : 0.f; }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) s += fabsf(ea * a.image[(size_t)c * a.N + p] + eb - a.gt_image[(size_t)c * a.N + p]);
        acc += a.c_rgb * wr * s + a.c_depth * wd * fabsf(a.depth[p] - a.gt_depth[p]);
    }
    const float t = block_sum_fixed(acc, s_tmp);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// out[k] = sum over blocks of partials[b * stride + k], k < stride, fixed order
__global__ void __launch_bounds__(64) loss_finalize_kernel(const float* __restrict__ partials, int stride, float* __restrict__ out)
{
    const int k = threadIdx.x;
    if (k >= stride) return;
    float t = 0.f;
    for (int b = 0; b < LOSS_BLOCKS; b++) t += partials[b * stride + k];
    out[k] = t;
}

__global__ void __launch_bounds__(LOSS_THREADS) l1_loss_bwd_kernel(LossArgs a, const float* __restrict__ upstream, float* __restrict__ dL_dimage,
                                                                   float* __restrict__ dL_ddepth, float* __restrict__ partials /*[blocks][2]*/)
{
    __shared__ float s_tmp[LOSS_THREADS / 64];
    const float g = upstream ? upstream[0] : 1.f;
    const float ea = a.exposure_a ? expf(a.exposure_a[0]) : 1.f, eb = a.exposure_b ? a.exposure_b[0] : 0.f;
    float da = 0.f, db = 0.f;
    for (int p = blockIdx.x * LOSS_THREADS + threadIdx.x; p < a.N; p += LOSS_BLOCKS * LOSS_THREADS) {
        float wr = (a.w_rgb ? a.w_rgb[p] : 1.f) * a.c_rgb * g, wd = (a.w_depth ? a.w_depth[p] : 1.f) * a.c_depth * g;
        if (a.opacity) { const float op = a.opacity[p]; wr *= op; wd = op > a.opacity_thr ? wd : 0.f; }   // opacity is a weight: the rasterizer
                                                                                                    // drops its cotangent anyway
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float I = a.image[(size_t)c * a.N + p];
            const float s = wr * sgn(ea * I + eb - a.gt_image[(size_t)c * a.N + p]);   // dL / d(exp(a) I + b)
            dL_dimage[(size_t)c * a.N + p] = s * ea;
            da += s * ea * I;
            db += s;
        }
        dL_ddepth[p] = wd * sgn(a.depth[p] - a.gt_depth[p]);
    }
    const float ta = block_sum_fixed(da, s_tmp), tb = block_sum_fixed(db, s_tmp);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = ta; partials[2 * blockIdx.x + 1] = tb; }
}

}  // namespace gsr

// ------------------------------------------------------------------------------------------------------------------
// Fused Adam step over several parameter tensors in ONE launch (second half): the reference steps six
// groups (xyz, f_dc, f_rest, opacity, scaling, rotation; scene/gaussian_model.py:404-447, torch.optim.Adam(lr=0, eps=1e-15))
// after every mapping iteration; torch's multi-tensor path needs ~10 launches for them. Same arithmetic as
// torch.optim.Adam's single-tensor path (no amsgrad, no weight decay):
//   m <- m + (g - m)(1 - b1);  v <- b2 v + (1 - b2) g^2;  p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// ------------------------------------------------------------------------------------------------------------------
constexpr int ADAM_MAX_SEGMENTS = 8;
struct AdamSegment { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; unsigned long long n; float step_size, inv_bc2_sqrt, eps, beta2, one_minus_beta1, one_minus_beta2; };   // 1 - beta evaluated in double on the host, as torch does
struct AdamArgs { int nseg; unsigned long long total; unsigned long long start[ADAM_MAX_SEGMENTS + 1]; AdamSegment seg[ADAM_MAX_SEGMENTS]; };

__global__ void __launch_bounds__(256) adam_step_kernel(AdamArgs a)
{
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x; e < a.total; e += (unsigned long long)gridDim.x * 256) {
        int s = 0;
#pragma unroll
        for (int k = 1; k < ADAM_MAX_SEGMENTS; k++) if (k < a.nseg && e >= a.start[k]) s = k;
        const AdamSegment& g = a.seg[s];
        const unsigned long long i = e - a.start[s];
        const float gr = g.grad[i];
        float m = g.exp_avg[i], v = g.exp_avg_sq[i];
        m = m + (gr - m) * g.one_minus_beta1;
        v = v * g.beta2 + g.one_minus_beta2 * gr * gr;
        g.exp_avg[i] = m; g.exp_avg_sq[i] = v;
        g.param[i] = g.param[i] - g.step_size * (m / (sqrtf(v) * g.inv_bc2_sqrt + g.eps));
    }
}
