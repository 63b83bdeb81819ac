// gs_linear.h -- weight gradient of a dense layer over a very long batch:  dW[o][c] = sum_p dY[p][o] X[p][c],  db[o] = sum_p dY[p][o].
//
// The deformation network (utils/deformation.py:58-70) is seven nn.Linear layers of width <= 128 applied to every Gaussian, so its
// weight gradients are GEMMs with a 64x128-or-smaller OUTPUT and a reduction length of n = 10^5..10^6 points.  The vendor GEMM
// picks macro-tiles for that shape that leave almost all of the chip idle (measured at n = 200k: 465-524 us per layer, 3.4 of the
// 4.5 ms the whole MLP takes forward+backward).  Here the reduction is split over ~1000 blocks; each wave owns one 16-row tile of
// dW, streams its 16 dY columns and all X columns straight from global memory in MFMA fragment order (lane l: A = dY[p + (l>>4)]
// [o0 + (l&15)], B = X[p + (l>>4)][c0 + (l&15)] -- 64-byte runs, no LDS staging needed) and accumulates with
// v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain).  Block partials go to a workspace; a second kernel sums them in a
// fixed order, so the result is deterministic (no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WGRAD_BLOCK = 256;      // 4 waves = 4 row tiles (64 rows of dW) per block
constexpr int WGRAD_UNROLL = 4;       // k-steps (of 4 points) in flight per wave

// partial: [gridDim.x][out_dim][in_dim + 1]  (last column: the bias gradient)
template <int NT>
__global__ void __launch_bounds__(WGRAD_BLOCK)
linear_wgrad_kernel(const int64_t n, const int in_dim, const int out_dim, const float* __restrict__ x, const int64_t x_stride,
                    const float* __restrict__ dy, const int64_t dy_stride, float* __restrict__ partial, const int64_t chunk)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mt = blockIdx.y * (WGRAD_BLOCK / 64) + wave;
    if (mt * 16 >= out_dim) return;                               // no barrier below: a wave may leave on its own
    const int i = lane & 15, kk = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * chunk, p1 = min(n, p0 + chunk);
    const int o = mt * 16 + i;
    const bool o_ok = o < out_dim;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int64_t p = p0; p < p1; p += 4 * WGRAD_UNROLL) {
        float a[WGRAD_UNROLL], b[WGRAD_UNROLL][NT];
#pragma unroll
        for (int u = 0; u < WGRAD_UNROLL; u++) {
            const int64_t pt = p + 4 * u + kk;
            const bool ok = pt < p1;
            a[u] = ok && o_ok ? dy[pt * dy_stride + o] : 0.f;
#pragma unroll
            for (int t = 0; t < NT; t++) b[u][t] = ok && 16 * t + i < in_dim ? x[pt * x_stride + 16 * t + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < WGRAD_UNROLL; u++) {
            bsum += a[u];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][t], acc[t], 0, 0, 0);
        }
    }
    // C layout of 16x16x4: column = lane & 15 (input index), row = (lane >> 4) * 4 + r (output index inside the tile)
    float* out = partial + (size_t)blockIdx.x * out_dim * (in_dim + 1);
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int c = 16 * t + i;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = mt * 16 + kk * 4 + r;
            if (row < out_dim && c < in_dim) out[(size_t)row * (in_dim + 1) + c] = acc[t][r];
        }
    }
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (kk == 0 && o_ok) out[(size_t)o * (in_dim + 1) + in_dim] = bsum;
}

// dW[o][c] = sum_b partial[b][o][c] (c < in_dim);  db[o] = sum_b partial[b][o][in_dim].  A block sums 32 outputs: 8 slices of
// the partial blocks per output in parallel (each thread a strided subset, four independent accumulators), then the slices in a
// fixed order through LDS -- deterministic, and ~260 blocks instead of ~33 for a 64x128 layer.
constexpr int WRED_OUT = 32, WRED_SLICES = 8;
__global__ void __launch_bounds__(WRED_OUT * WRED_SLICES)
linear_wgrad_reduce_kernel(const int nblocks, const int in_dim, const int out_dim, const float* __restrict__ partial,
                           float* __restrict__ dW, float* __restrict__ db)
{
    __shared__ float s_part[WRED_SLICES][WRED_OUT];
    const int j = threadIdx.x % WRED_OUT, slice = threadIdx.x / WRED_OUT;
    const int idx = blockIdx.x * WRED_OUT + j;
    const int per = out_dim * (in_dim + 1);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < per) {
        int b = slice;
        for (; b + 3 * WRED_SLICES < nblocks; b += 4 * WRED_SLICES) {
            s0 += partial[(size_t)b * per + idx];
            s1 += partial[(size_t)(b + WRED_SLICES) * per + idx];
            s2 += partial[(size_t)(b + 2 * WRED_SLICES) * per + idx];
            s3 += partial[(size_t)(b + 3 * WRED_SLICES) * per + idx];
        }
        for (; b < nblocks; b += WRED_SLICES) s0 += partial[(size_t)b * per + idx];
    }
    s_part[slice][j] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice != 0 || idx >= per) return;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < WRED_SLICES; k++) s += s_part[k][j];
    const int o = idx / (in_dim + 1), c = idx % (in_dim + 1);
    if (c < in_dim) { if (dW) dW[(size_t)o * in_dim + c] = s; }
    else if (db) db[o] = s;
}

}  // namespace gsr
