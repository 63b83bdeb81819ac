// gs_nodes.h -- SC-GS control-node warp (include/control_nodes.h): K nearest control nodes, normalised RBF weights, blend of the
// nodes' translation / rotation / scale predictions, and the gradients back to the per-node quantities.
//
// Reference: utils/time_utils.py ControlNodeWarp.cal_nn_weight :981-1011 (pytorch3d.ops.knn_points + exp(-d / 2 r^2) * node
// weight + 1e-7, normalised over K) and ControlNodeWarp.forward :1192-1258.  There, one call is a pytorch3d CUDA kernel plus ~20
// gathers / broadcasts over [N, K, .] intermediates and their autograd twins (index_put_ with accumulation into [M, .] tensors).
//
// Here the node positions (a few hundred) sit in LDS, each thread owns one Gaussian, scans every node with a register top-K and
// finishes the blend from the K winners' attributes (L2-resident); backward accumulates the 21 floats per node in LDS with
// ds_add_f32 and each block writes ONE partial row that a second kernel sums in a fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/control_nodes.h"

namespace gsr {

constexpr int NODE_BLOCK = 256;
constexpr int NODE_CHUNK = 4096;          // node positions staged per pass: 48 KB of LDS
constexpr int NODE_GRAD = 21;             // per node: trans 3, rot 4, scale 3, frame 9, radius 1, weight 1
constexpr int NODE_LDS_MAX = 720;         // backward keeps m * 21 floats in LDS up to this many nodes (< 64 KB)

// sorted insertion of (d, j) into the ascending list bd[0..K): strict <, so among equal distances the earlier index stays first
template <int KMAX>
__device__ __forceinline__ void topk_insert(float (&bd)[KMAX], int (&bi)[KMAX], const int K, float d, int j)
{
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        if (k < K) {
            const bool sw = d < bd[k];
            const float td = sw ? bd[k] : d;
            const int ti = sw ? bi[k] : j;
            bd[k] = sw ? d : bd[k];
            bi[k] = sw ? j : bi[k];
            d = td;
            j = ti;
        }
    }
}

template <int KMAX>
__device__ __forceinline__ float topk_worst(const float (&bd)[KMAX], const int K)
{
    float w = bd[0];
#pragma unroll
    for (int k = 1; k < KMAX; k++) w = k < K ? bd[k] : w;
    return w;
}

// ---- pytorch3d.ops.knn_points for one batch element -------------------------------------------------------------------------
template <int DMAX, int KMAX>
__global__ void __launch_bounds__(NODE_BLOCK)
knn_points_kernel(const int64_t n, const int64_t m, const int D, const int K, const float* __restrict__ p1, const float* __restrict__ p2,
                  float* __restrict__ dist2, int64_t* __restrict__ idx)
{
    extern __shared__ float s_p2[];                               // [rows][D]
    const int64_t i = (int64_t)blockIdx.x * NODE_BLOCK + threadIdx.x;
    float q[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; d++) q[d] = (i < n && d < D) ? p1[i * D + d] : 0.f;
    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) { bd[k] = INFINITY; bi[k] = -1; }
    float worst = INFINITY;
    const int rows = (int)min((int64_t)(NODE_CHUNK * 3 / D), m);  // rows per pass (same LDS budget as the 3-D case)
    for (int64_t base = 0; base < m; base += rows) {
        const int cnt = (int)min((int64_t)rows, m - base);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * D; e += NODE_BLOCK) s_p2[e] = p2[base * D + e];
        __syncthreads();
        if (i < n) {
            for (int j = 0; j < cnt; j++) {
                float d2 = 0.f;
#pragma unroll
                for (int d = 0; d < DMAX; d++) {
                    if (d < D) { const float t = q[d] - s_p2[j * D + d]; d2 = fmaf(t, t, d2); }
                }
                if (d2 < worst) {
                    topk_insert<KMAX>(bd, bi, K, d2, (int)(base + j));
                    worst = topk_worst<KMAX>(bd, K);
                }
            }
        }
    }
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        if (k < K) {
            const bool ok = bi[k] >= 0;
            dist2[i * K + k] = ok ? bd[k] : 0.f;
            idx[i * K + k] = ok ? bi[k] : 0;
        }
    }
}

// ---- cal_nn_weight + blend ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NODE_BLOCK)
node_blend_fwd_kernel(const gsr_node_blend a, float* __restrict__ nn_weight, float* __restrict__ nn_dist, int64_t* __restrict__ nn_idx,
                      float* __restrict__ d_xyz, float* __restrict__ d_rotation, float* __restrict__ d_scaling)
{
    constexpr int KMAX = GSR_BLEND_MAX_K;
    __shared__ float s_pos[NODE_CHUNK * 3];
    const int64_t i = (int64_t)blockIdx.x * NODE_BLOCK + threadIdx.x;
    const int K = a.K;
    float x[3] = {0.f, 0.f, 0.f};
    if (i < a.n) { x[0] = a.x[3 * i]; x[1] = a.x[3 * i + 1]; x[2] = a.x[3 * i + 2]; }
    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) { bd[k] = INFINITY; bi[k] = -1; }
    float worst = INFINITY;
    for (int base = 0; base < a.m; base += NODE_CHUNK) {
        const int cnt = min(NODE_CHUNK, a.m - base);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 3; e += NODE_BLOCK) s_pos[e] = a.nodes[(size_t)(base + e / 3) * a.node_stride + e % 3];
        __syncthreads();
        if (i < a.n) {
            for (int j = 0; j < cnt; j++) {                       // every lane reads the same node: an LDS broadcast
                const float tx = x[0] - s_pos[3 * j], ty = x[1] - s_pos[3 * j + 1], tz = x[2] - s_pos[3 * j + 2];
                const float d2 = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
                if (d2 < worst) {
                    topk_insert<KMAX>(bd, bi, K, d2, base + j);
                    worst = topk_worst<KMAX>(bd, K);
                }
            }
        }
    }
    if (i >= a.n) return;
    // weights: exp(-d / (2 r^2)) [* node weight] + 1e-7, normalised over the K (:1000-1006)
    float w[KMAX], S = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        w[k] = 0.f;
        if (k < K) {
            const int j = max(bi[k], 0);
            if (bi[k] < 0) bd[k] = 0.f;                           // fewer nodes than K: pytorch3d pads with index 0, distance 0
            bi[k] = j;
            const float r = a.node_radius[j];
            float u = expf(-bd[k] / (2.f * r * r));
            if (a.node_weight) u *= a.node_weight[j];
            w[k] = u + 1e-7f;
            S += w[k];
        }
    }
    const float invS = 1.f / S;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        if (k < K) {
            w[k] *= invS;
            nn_weight[i * K + k] = w[k];
            nn_dist[i * K + k] = bd[k];
            nn_idx[i * K + k] = bi[k];
        }
    }
    if (!a.node_trans) return;
    const float mask = a.motion_mask ? a.motion_mask[i] : 1.f;
    float t[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        if (k < K) {
            const int j = bi[k];
            const float* tr = a.node_trans + 3 * (size_t)j;
            if (a.local_frame) {                                  // R (x - node) + node + trans (:1209)
                const float* R = a.node_frame + 9 * (size_t)j;
                const float* nd = a.nodes + (size_t)j * a.node_stride;
                const float ox = x[0] - nd[0], oy = x[1] - nd[1], oz = x[2] - nd[2];
#pragma unroll
                for (int c = 0; c < 3; c++) t[c] += w[k] * ((R[3 * c] * ox + R[3 * c + 1] * oy + R[3 * c + 2] * oz) + nd[c] + tr[c]);
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) t[c] += w[k] * tr[c];  // :1213
            }
            const float* qr = a.node_rot + 4 * (size_t)j;
#pragma unroll
            for (int c = 0; c < 4; c++) q[c] += w[k] * (qr[c] + ((!a.rot_as_residual && c == 0) ? 1.f : 0.f));
            const float* sc = a.node_scale + 3 * (size_t)j;
#pragma unroll
            for (int c = 0; c < 3; c++) s[c] += w[k] * sc[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        d_xyz[3 * i + c] = (a.local_frame ? t[c] - x[c] : t[c]) * mask;                 // :1211,1214
        d_scaling[3 * i + c] = s[c] * mask;                                            // :1248,1257
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float b = c == 0 ? 1.f : 0.f;
        d_rotation[4 * i + c] = a.rot_as_residual ? q[c] * mask : (q[c] - b) * mask + b;   // :1255-1256 / :1231
    }
}

// partial: [gridDim.x][m * NODE_GRAD]; use_lds = 0: every block adds into row 0 with global atomics (caller zeroed it)
__global__ void __launch_bounds__(NODE_BLOCK)
node_blend_bwd_kernel(const gsr_node_blend a, const float* __restrict__ nn_weight, const float* __restrict__ nn_dist,
                      const int64_t* __restrict__ nn_idx, const float* __restrict__ g_xyz, const float* __restrict__ g_rotation,
                      const float* __restrict__ g_scaling, const float* __restrict__ g_nn_weight, float* __restrict__ partial, const int use_lds)
{
    constexpr int KMAX = GSR_BLEND_MAX_K;
    extern __shared__ float s_acc[];                              // [m][NODE_GRAD] when use_lds
    const int K = a.K;
    const int total = a.m * NODE_GRAD;
    if (use_lds) {
        for (int e = threadIdx.x; e < total; e += NODE_BLOCK) s_acc[e] = 0.f;
        __syncthreads();
    }
    float* acc = use_lds ? s_acc : partial;
    auto add = [&](int e, float v) {
        if (v == 0.f) return;
        if (use_lds) __hip_atomic_fetch_add(&s_acc[e], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else unsafeAtomicAdd(&acc[e], v);
    };
    for (int64_t i = (int64_t)blockIdx.x * NODE_BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * NODE_BLOCK) {
        const float mask = a.motion_mask ? a.motion_mask[i] : 1.f;
        const float x[3] = {a.x[3 * i], a.x[3 * i + 1], a.x[3 * i + 2]};
        float gx[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f};
        const bool blend = a.node_trans != nullptr;
        if (blend) {                                              // the motion mask multiplies every output (:1214,1231,1248,1255-1257)
#pragma unroll
            for (int c = 0; c < 3; c++) { gx[c] = g_xyz ? g_xyz[3 * i + c] * mask : 0.f; gs[c] = g_scaling ? g_scaling[3 * i + c] * mask : 0.f; }
#pragma unroll
            for (int c = 0; c < 4; c++) gq[c] = g_rotation ? g_rotation[4 * i + c] * mask : 0.f;
        }
        float w[KMAX], G[KMAX], Gw = 0.f, S = 0.f;                // G_k = dL/dw_k
        int idx[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            w[k] = G[k] = 0.f;
            idx[k] = 0;
            if (k < K) {
                const int j = (int)nn_idx[i * K + k];
                idx[k] = j;
                w[k] = nn_weight[i * K + k];
                float g = g_nn_weight ? g_nn_weight[i * K + k] : 0.f;
                if (blend) {
                    const float* tr = a.node_trans + 3 * (size_t)j;
                    const float* qr = a.node_rot + 4 * (size_t)j;
                    const float* sc = a.node_scale + 3 * (size_t)j;
                    if (a.local_frame) {
                        const float* R = a.node_frame + 9 * (size_t)j;
                        const float* nd = a.nodes + (size_t)j * a.node_stride;
                        const float o[3] = {x[0] - nd[0], x[1] - nd[1], x[2] - nd[2]};
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            g += gx[c] * ((R[3 * c] * o[0] + R[3 * c + 1] * o[1] + R[3 * c + 2] * o[2]) + nd[c] + tr[c]);
#pragma unroll
                            for (int b = 0; b < 3; b++) add(j * NODE_GRAD + 10 + 3 * c + b, w[k] * gx[c] * o[b]);   // dL/dR[c][b]
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; c++) g += gx[c] * tr[c];
                    }
#pragma unroll
                    for (int c = 0; c < 3; c++) { add(j * NODE_GRAD + c, w[k] * gx[c]); add(j * NODE_GRAD + 7 + c, w[k] * gs[c]); g += gs[c] * sc[c]; }
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        add(j * NODE_GRAD + 3 + c, w[k] * gq[c]);
                        g += gq[c] * (qr[c] + ((!a.rot_as_residual && c == 0) ? 1.f : 0.f));
                    }
                }
                G[k] = g;
                Gw += g * w[k];
            }
        }
        // w_k = u_k / S, u_k = e_k nw_k + 1e-7:  dL/du_j = (G_j - sum_k G_k w_k) / S;  S is rebuilt from the stored distances
        float e[KMAX], nw[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            e[k] = nw[k] = 0.f;
            if (k < K) {
                const float r = a.node_radius[idx[k]];
                e[k] = expf(-nn_dist[i * K + k] / (2.f * r * r));
                nw[k] = a.node_weight ? a.node_weight[idx[k]] : 1.f;
                S += e[k] * nw[k] + 1e-7f;
            }
        }
        const float invS = 1.f / S;
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            if (k < K) {
                const float du = (G[k] - Gw) * invS;
                const float r = a.node_radius[idx[k]];
                add(idx[k] * NODE_GRAD + 19, du * nw[k] * e[k] * nn_dist[i * K + k] / (r * r * r));      // d e / d r = e d / r^3
                if (a.node_weight) add(idx[k] * NODE_GRAD + 20, du * e[k]);
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        float* row = partial + (size_t)blockIdx.x * total;
        for (int e = threadIdx.x; e < total; e += NODE_BLOCK) row[e] = s_acc[e];
    }
}

// out component c of node j = sum over the G partial rows, fixed order
__global__ void __launch_bounds__(256)
node_grad_reduce_kernel(const int G, const int m, const float* __restrict__ partial, float* __restrict__ g_trans, float* __restrict__ g_rot,
                        float* __restrict__ g_scale, float* __restrict__ g_frame, float* __restrict__ g_radius, float* __restrict__ g_weight)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int total = m * NODE_GRAD;
    if (e >= total) return;
    float s0 = 0.f, s1 = 0.f;
    int b = 0;
    for (; b + 1 < G; b += 2) { s0 += partial[(size_t)b * total + e]; s1 += partial[(size_t)(b + 1) * total + e]; }
    if (b < G) s0 += partial[(size_t)b * total + e];
    const float s = s0 + s1;
    const int j = e / NODE_GRAD, c = e % NODE_GRAD;
    if (c < 3) { if (g_trans) g_trans[3 * j + c] = s; }
    else if (c < 7) { if (g_rot) g_rot[4 * j + c - 3] = s; }
    else if (c < 10) { if (g_scale) g_scale[3 * j + c - 7] = s; }
    else if (c < 19) { if (g_frame) g_frame[9 * j + c - 10] = s; }
    else if (c == 19) { if (g_radius) g_radius[j] = s; }
    else if (g_weight) g_weight[j] = s;
}

}  // namespace gsr
