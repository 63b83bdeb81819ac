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

constexpr int LOSS_BLOCKS = 1024, LOSS_THREADS = 256;          // (1.2 pixels per thread at 640 x 480: with 256 blocks a thread walked five pixels,
                                                               // each a round of dependent loads: 9.6 us for 12 MB)

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

__device__ __forceinline__ float sgn(float v) { return l1_sgn(v); }   // d|v|/dv as torch defines it (gs_device.h)

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
    // one wave: lane l adds the partials of blocks l, l + 64, l + 128, ... (independent loads), the lanes are added by xor shuffles -- a fixed
    // order. (One thread per k walking all partials was a chain of load latencies: 6.5 us per launch, nine launches per dynamic iteration.)
    static_assert(LOSS_BLOCKS % 64 == 0, "loss_finalize_kernel adds LOSS_BLOCKS / 64 partials per lane");
    const int lane = threadIdx.x;
    for (int k = 0; k < stride; k++) {
        float v[LOSS_BLOCKS / 64];
#pragma unroll
        for (int u = 0; u < LOSS_BLOCKS / 64; u++) v[u] = partials[(lane + 64 * u) * stride + k];
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < LOSS_BLOCKS / 64; u++) t += v[u];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane == 0) out[k] = t;
    }
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
        const float I[3] = {a.image[p], a.image[(size_t)a.N + p], a.image[2 * (size_t)a.N + p]};
        const float gt[3] = {a.gt_image[p], a.gt_image[(size_t)a.N + p], a.gt_image[2 * (size_t)a.N + p]};
        const L1PixelGrad o = l1_bwd_pixel(wr, wd, ea, eb, I, gt, a.depth[p], a.gt_depth[p], da, db);   // gs_device.h: shared with render_fwd's tracking epilogue
#pragma unroll
        for (int c = 0; c < 3; c++) dL_dimage[(size_t)c * a.N + p] = o.gi[c];
        dL_ddepth[p] = o.gd;
    }
    const float ta = block_sum_fixed(da, s_tmp), tb = block_sum_fixed(db, s_tmp);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = ta; partials[2 * blockIdx.x + 1] = tb; }
}



// ---- masked L1 against a constant target, several images per call (the optical-flow terms of the dynamic mapping loop) -----------
//   L = scale * sum_terms mean_{c < C, p}( | target[c,p] - image[c,p] * mask[p] | )
// utils/slam_backend.py:486-488,503-505 form this with bitwise_not / unsqueeze / 2 mul / permute / sub / abs / mean / mul per direction
// and as many autograd nodes back; a keyframe's two directions are two terms of one call here. The image has `Cimg` >= C channels
// (render_flow's third channel is the dynamic mask: no loss, zero gradient).
constexpr int MASKED_L1_MAX_TERMS = 4;
struct MaskedL1Args {
    int n_terms, N, C, Cimg;
    const float* image[MASKED_L1_MAX_TERMS]; const float* target[MASKED_L1_MAX_TERMS]; const float* mask[MASKED_L1_MAX_TERMS];
    float* dL_dimage[MASKED_L1_MAX_TERMS];
    float coeff;                              // scale / (C N)
};

__global__ void __launch_bounds__(LOSS_THREADS) masked_l1_fwd_kernel(MaskedL1Args a, float* __restrict__ partials)
{
    __shared__ float s_tmp[LOSS_THREADS / 64];
    float acc = 0.f;
    for (int t = 0; t < a.n_terms; t++) {
        for (int p = blockIdx.x * LOSS_THREADS + threadIdx.x; p < a.N; p += LOSS_BLOCKS * LOSS_THREADS) {
            const float m = a.mask[t][p];
            for (int c = 0; c < a.C; c++) acc += fabsf(a.target[t][(size_t)c * a.N + p] - a.image[t][(size_t)c * a.N + p] * m);
        }
    }
    const float s = block_sum_fixed(acc * a.coeff, s_tmp);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ void __launch_bounds__(LOSS_THREADS) masked_l1_bwd_kernel(MaskedL1Args a, const float* __restrict__ upstream)
{
    const float g = (upstream ? upstream[0] : 1.f) * a.coeff;
    float* const out = a.dL_dimage[blockIdx.y];
    const float* const img = a.image[blockIdx.y]; const float* const tgt = a.target[blockIdx.y]; const float* const msk = a.mask[blockIdx.y];
    for (int p = blockIdx.x * LOSS_THREADS + threadIdx.x; p < a.N; p += LOSS_BLOCKS * LOSS_THREADS) {
        const float m = msk[p];
        for (int c = 0; c < a.Cimg; c++)
            out[(size_t)c * a.N + p] = c < a.C ? -g * m * sgn(tgt[(size_t)c * a.N + p] - img[(size_t)c * a.N + p] * m) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused Adam step over several parameter tensors in ONE launch (second half): the reference steps six
// groups (xyz, f_dc, f_rest, opacity, scaling, rotation; scene/gaussian_model.py:404-447, torch.optim.Adam(lr=0, eps=1e-15))
// after every mapping iteration; torch's multi-tensor path needs ~10 launches for them. Same arithmetic as
// torch.optim.Adam's single-tensor path (no amsgrad, no weight decay):
//   m <- m + (g - m)(1 - b1);  v <- b2 v + (1 - b2) g^2;  p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// ------------------------------------------------------------------------------------------------------------------
constexpr int ADAM_MAX_SEGMENTS = 32;          // (the node network: 25 tensors; the Gaussian model: 6)
struct AdamSegment { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; unsigned long long n; float step_size, inv_bc2_sqrt, eps, beta2, one_minus_beta1, one_minus_beta2; };   // 1 - beta evaluated in double on the host, as torch does
struct AdamArgs { int nseg; unsigned long long total; unsigned long long start[ADAM_MAX_SEGMENTS + 1]; AdamSegment seg[ADAM_MAX_SEGMENTS];
                  const float* coef; };   // coef != nullptr (gsr_adam_step_scheduled): step_size / inv_bc2_sqrt of segment k are coef[2k], coef[2k+1] in DEVICE memory

// The coefficients of a step whose COUNT lives on the device (a torch.optim.Adam(capturable=True) state: one float32 scalar per parameter):
// thread k advances step[k] and writes lr / (1 - b1^t), 1 / sqrt(1 - b2^t) -- double arithmetic, as gsr_adam_coefficients -- for
// gsr_adam_step_scheduled. One tiny launch; with it the step is two launches inside a hipGraph that replays correctly (the count is read at
// replay time).
struct AdamDeviceSteps { int nseg; float* step[ADAM_MAX_SEGMENTS]; float lr[ADAM_MAX_SEGMENTS]; double beta1[ADAM_MAX_SEGMENTS], beta2[ADAM_MAX_SEGMENTS]; };
__global__ void __launch_bounds__(64) adam_device_coefficients_kernel(AdamDeviceSteps a, float* __restrict__ coef)
{
    static_assert(ADAM_MAX_SEGMENTS <= 64, "one thread per segment, one wave");
    const int k = threadIdx.x;
    const bool live = k < a.nseg;                              // (no early return: every thread reaches the barrier)
    bool first = live;                                         // (several parameters may share one counter: advance it once)
    float t = 0.f;
    if (live) {
        for (int j = 0; j < k; j++) first = first && a.step[j] != a.step[k];
        t = a.step[k][0] + 1.f;
        const double bc1 = 1.0 - pow(a.beta1[k], (double)t), bc2 = 1.0 - pow(a.beta2[k], (double)t);
        coef[2 * k] = (float)((double)a.lr[k] / bc1); coef[2 * k + 1] = (float)(1.0 / sqrt(bc2));
    }
    __syncthreads();                                           // every thread has read its counter
    if (first) a.step[k][0] = t;
}

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
        const float step_size = a.coef ? a.coef[2 * s] : g.step_size, inv_bc2_sqrt = a.coef ? a.coef[2 * s + 1] : g.inv_bc2_sqrt;
        g.param[i] = g.param[i] - step_size * (m / (sqrtf(v) * inv_bc2_sqrt + g.eps));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused SSIM ("optional SSIM"): gaussian_splatting/utils/loss_utils.py:46-111 as used by the colour
// refinement / mapping losses (utils/slam_backend.py:636,824-832): 11x11 Gaussian window (sigma 1.5), zero padding, per channel,
//   ssim_map = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)),   loss term = mean(ssim_map).
// torch evaluates it with five grouped conv2d + ~15 elementwise kernels and as many again in backward. Here: one kernel per
// direction, one 16x16 pixel tile per block and channel, separable window through LDS.
//   forward : the five windowed means (x, y, x^2, y^2, xy) -> ssim value (block-summed, fixed order) and, for the backward pass,
//             the three partial derivatives of the map w.r.t. mu1, E[x^2], E[xy] (img2 is the ground truth: no gradient);
//   backward: dL/dx(p) = sum_q w(q - p) [ dm/dmu1(q) + 2 x(p) dm/dE11(q) + y(p) dm/dE12(q) ] * upstream / (C H W): three more
//             separable convolutions of those maps.
// An optional pixel mask zeroes both images first (loss_utils.py:66-68), i.e. masks the gradient.
// ------------------------------------------------------------------------------------------------------------------
constexpr int SSIM_WIN = 11, SSIM_R = 5, SSIM_T = 16, SSIM_P = SSIM_T + 2 * SSIM_R;   // window, radius, tile, padded tile (26)

struct SsimWindow { float w[SSIM_WIN]; };

__device__ __forceinline__ float ssim_load(const float* __restrict__ img, const unsigned char* __restrict__ mask, int W, int H, int x, int y)
{
    if (x < 0 || y < 0 || x >= W || y >= H) return 0.f;                          // conv2d zero padding
    if (mask && !mask[(size_t)y * W + x]) return 0.f;                             // torch.where(mask, img, 0)
    return img[(size_t)y * W + x];
}

__global__ void __launch_bounds__(SSIM_T * SSIM_T) ssim_fwd_kernel(int W, int H, const float* __restrict__ img1, const float* __restrict__ img2,
                                                                   const unsigned char* __restrict__ mask, SsimWindow win,
                                                                   float* __restrict__ dmaps /*[C][3][H*W]*/, float* __restrict__ partials)
{
    __shared__ float s_x[SSIM_P][SSIM_P + 1], s_y[SSIM_P][SSIM_P + 1];
    __shared__ float s_h[5][SSIM_P][SSIM_T + 1];          // horizontally filtered x, y, xx, yy, xy
    __shared__ float s_tmp[SSIM_T * SSIM_T / 64];
    const int c = blockIdx.z, tx = threadIdx.x % SSIM_T, ty = threadIdx.x / SSIM_T;
    const int x0 = blockIdx.x * SSIM_T - SSIM_R, y0 = blockIdx.y * SSIM_T - SSIM_R;
    const size_t N = (size_t)W * H;
    const float* a = img1 + c * N; const float* b = img2 + c * N;
    for (int i = threadIdx.x; i < SSIM_P * SSIM_P; i += SSIM_T * SSIM_T) {
        const int r = i / SSIM_P, q = i % SSIM_P;
        s_x[r][q] = ssim_load(a, mask, W, H, x0 + q, y0 + r);
        s_y[r][q] = ssim_load(b, mask, W, H, x0 + q, y0 + r);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SSIM_P * SSIM_T; i += SSIM_T * SSIM_T) {      // horizontal pass: 26 rows x 16 columns
        const int r = i / SSIM_T, q = i % SSIM_T;
        float hx = 0.f, hy = 0.f, hxx = 0.f, hyy = 0.f, hxy = 0.f;
#pragma unroll
        for (int k = 0; k < SSIM_WIN; k++) {
            const float u = s_x[r][q + k], v = s_y[r][q + k], w = win.w[k];
            hx += w * u; hy += w * v; hxx += w * u * u; hyy += w * v * v; hxy += w * u * v;
        }
        s_h[0][r][q] = hx; s_h[1][r][q] = hy; s_h[2][r][q] = hxx; s_h[3][r][q] = hyy; s_h[4][r][q] = hxy;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;                // vertical pass: this thread's pixel
#pragma unroll
    for (int k = 0; k < SSIM_WIN; k++) {
        const float w = win.w[k];
        mu1 += w * s_h[0][ty + k][tx]; mu2 += w * s_h[1][ty + k][tx]; e11 += w * s_h[2][ty + k][tx];
        e22 += w * s_h[3][ty + k][tx]; e12 += w * s_h[4][ty + k][tx];
    }
    const int px = blockIdx.x * SSIM_T + tx, py = blockIdx.y * SSIM_T + ty;
    const bool inside = px < W && py < H;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
    const float inv = 1.0f / (Cc * D);
    const float m = A * B * inv;
    if (inside && dmaps) {
        float* dm = dmaps + (size_t)c * 3 * N + (size_t)py * W + px;
        // d ssim / d mu1 (mu1 also enters s1 and s12), d / d E[x^2], d / d E[xy]
        dm[0] = (2.f * mu2 * (B - A)) * inv - m * (2.f * mu1 * (D - Cc)) * inv;
        dm[N] = -m / D;
        dm[2 * N] = 2.f * A * inv;
    }
    const float t = block_sum_fixed(inside ? m : 0.f, s_tmp);
    if (threadIdx.x == 0) partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
}

// out[0] = scale * sum of n partials, fixed order (one block)
__global__ void __launch_bounds__(256) sum_partials_kernel(int n, const float* __restrict__ partials, float scale, float* __restrict__ out)
{
    __shared__ float s_tmp[4];
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    const float t = block_sum_fixed(v, s_tmp);
    if (threadIdx.x == 0) out[0] = t * scale;
}

__global__ void __launch_bounds__(SSIM_T * SSIM_T) ssim_bwd_kernel(int W, int H, const float* __restrict__ img1, const float* __restrict__ img2,
                                                                   const unsigned char* __restrict__ mask, SsimWindow win,
                                                                   const float* __restrict__ dmaps, const float* __restrict__ upstream,
                                                                   float scale, float* __restrict__ dL_dimg1)
{
    __shared__ float s_m[3][SSIM_P][SSIM_P + 1];
    __shared__ float s_h[3][SSIM_P][SSIM_T + 1];
    const int c = blockIdx.z, tx = threadIdx.x % SSIM_T, ty = threadIdx.x / SSIM_T;
    const int x0 = blockIdx.x * SSIM_T - SSIM_R, y0 = blockIdx.y * SSIM_T - SSIM_R;
    const size_t N = (size_t)W * H;
    for (int i = threadIdx.x; i < SSIM_P * SSIM_P; i += SSIM_T * SSIM_T) {
        const int r = i / SSIM_P, q = i % SSIM_P, x = x0 + q, y = y0 + r;
        const bool in = x >= 0 && y >= 0 && x < W && y < H;
        const float* dm = dmaps + (size_t)c * 3 * N + (size_t)y * W + x;
        s_m[0][r][q] = in ? dm[0] : 0.f; s_m[1][r][q] = in ? dm[N] : 0.f; s_m[2][r][q] = in ? dm[2 * N] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SSIM_P * SSIM_T; i += SSIM_T * SSIM_T) {
        const int r = i / SSIM_T, q = i % SSIM_T;
        float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
        for (int k = 0; k < SSIM_WIN; k++) { const float w = win.w[k]; h0 += w * s_m[0][r][q + k]; h1 += w * s_m[1][r][q + k]; h2 += w * s_m[2][r][q + k]; }
        s_h[0][r][q] = h0; s_h[1][r][q] = h1; s_h[2][r][q] = h2;
    }
    __syncthreads();
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int k = 0; k < SSIM_WIN; k++) { const float w = win.w[k]; g0 += w * s_h[0][ty + k][tx]; g1 += w * s_h[1][ty + k][tx]; g2 += w * s_h[2][ty + k][tx]; }
    const int px = blockIdx.x * SSIM_T + tx, py = blockIdx.y * SSIM_T + ty;
    if (px < W && py < H) {
        const size_t p = (size_t)py * W + px;
        const bool on = !mask || mask[p];
        const float x = on ? img1[c * N + p] : 0.f, y = on ? img2[c * N + p] : 0.f;
        const float g = (upstream ? upstream[0] : 1.f) * scale;
        dL_dimg1[c * N + p] = on ? g * (g0 + 2.f * x * g1 + y * g2) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Densification statistics of one rendered view in one launch (utils/slam_backend.py:712-720 + scene/gaussian_model.py:973-977):
//   visible = radii > 0;  max_radii2D[visible] = max(max_radii2D, radii);  xyz_gradient_accum[visible] += |grad_mean2D[:, :2]|;
//   denom[visible] += 1
// The reference does this with boolean-mask indexing: ~10 torch kernels and three host synchronisations per view.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) densification_stats_kernel(int P, const int* __restrict__ radii, const float* __restrict__ grad_mean2D,
                                                                  float* __restrict__ max_radii2D, float* __restrict__ grad_accum,
                                                                  float* __restrict__ denom)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    const float gx = grad_mean2D[3 * (size_t)i], gy = grad_mean2D[3 * (size_t)i + 1];
    grad_accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
}

}  // namespace gsr
