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
constexpr int NODE_CHUNK = 4096;          // generic kNN: floats*3 staged per pass (48 KB of LDS)
constexpr int NODE_CHUNK4 = 2048;         // 3-D scans: nodes staged per pass as float4 (32 KB of LDS)
constexpr int NODE_BATCH = 8;             // LDS reads in flight per lane in the 3-D scan
constexpr int NODE_GRAD = 21;             // per node: trans 3, rot 4, scale 3, frame 9, radius 1, weight 1
constexpr int NODE_DET_MAX_K = 4;       // the ordered (bit-reproducible) backward route covers K <= 4: every call of the SLAM loop (K = 3)
constexpr int NODE_LDS_MAX = 720;         // backward keeps m * 21 floats in LDS up to this many nodes (< 64 KB)

// sorted insertion of (d, j) into the ascending list bd[0..K): strict < for the new entry, so among equal distances the earlier index
// stays first; from the slot it takes on, every older entry moves down one place unconditionally (comparing the displaced entry
// again would let it fall behind a later entry of the same distance)
template <int KMAX>
__device__ __forceinline__ void topk_insert(float (&bd)[KMAX], int (&bi)[KMAX], const int K, float d, int j)
{
    bool placed = false;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        if (k < K) {
            const bool sw = placed || d < bd[k];
            placed = sw;
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
    // the list is ascending (unfilled places hold +inf), so its K-th entry is the maximum of the first K; written as a maximum because the
    // select chain `k < K ? bd[k] : w` is recognised as bd[K - 1] -- a run-time index that sent the whole list to scratch memory (32-144 B per lane)
    float w = bd[0];
#pragma unroll
    for (int k = 1; k < KMAX; k++) w = fmaxf(w, k < K ? bd[k] : w);
    return w;
}

// the same with the list length known at compile time: K compare-swaps, no predicates, no branch
template <int K>
__device__ __forceinline__ void topk_insert_exact(float (&bd)[K], int (&bi)[K], float d, int j)
{
    bool placed = false;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const bool sw = placed || d < bd[k];
        placed = sw;
        const float td = sw ? bd[k] : d;
        const int ti = sw ? bi[k] : j;
        bd[k] = sw ? d : bd[k];
        bi[k] = sw ? j : bi[k];
        d = td;
        j = ti;
    }
}

// Stage `cnt` 3-D points (row stride `stride` floats, at most NODE_CHUNK4) as float4 and pad the tail of the last batch with +inf,
// which no list accepts.  Every lane of a wave then reads the same address: an LDS broadcast, NODE_BATCH reads in flight.
__device__ __forceinline__ void stage_points3(float4* s_pos4, const float* __restrict__ pts, int64_t base, int cnt, int stride, int D)
{
    const int padded = (cnt + NODE_BATCH - 1) / NODE_BATCH * NODE_BATCH;
    for (int e = threadIdx.x; e < padded; e += NODE_BLOCK) {
        float4 p = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
        if (e < cnt) {
            const float* r = pts + (size_t)(base + e) * stride;
            p = make_float4(r[0], D > 1 ? r[1] : 0.f, D > 2 ? r[2] : 0.f, D > 3 ? r[3] : 0.f);
        }
        s_pos4[e] = p;
    }
}

// EXACT: the list length is KMAX (compile time) and every candidate goes through the branch-free insertion -- with a short list
// that is cheaper than a divergent `if (d < worst)` some lane of the wave takes on most iterations anyway.  Otherwise K <= KMAX
// is a run-time value and the insertion is guarded.
template <int KMAX, bool EXACT, bool FOURTH>
__device__ __forceinline__ void scan_points3(const float4* s_pos4, int cnt, int base, const float (&x)[4], float (&bd)[KMAX], int (&bi)[KMAX],
                                             int K, float& worst)
{
    for (int j0 = 0; j0 < cnt; j0 += NODE_BATCH) {
        float4 p[NODE_BATCH];
#pragma unroll
        for (int u = 0; u < NODE_BATCH; u++) p[u] = s_pos4[j0 + u];
#pragma unroll
        for (int u = 0; u < NODE_BATCH; u++) {
            const float tx = x[0] - p[u].x, ty = x[1] - p[u].y, tz = x[2] - p[u].z;
            float d2 = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
            if (FOURTH) { const float tw = x[3] - p[u].w; d2 = fmaf(tw, tw, d2); }
            if (EXACT) {
                topk_insert_exact<KMAX>(bd, bi, d2, base + j0 + u);
            } else if (d2 < worst) {
                topk_insert<KMAX>(bd, bi, K, d2, base + j0 + u);
                worst = topk_worst<KMAX>(bd, K);
            }
        }
    }
}

// ---- pytorch3d.ops.knn_points for one batch element -------------------------------------------------------------------------
// D <= 4 (the control-node case is D = 3): float4 staging, batched LDS reads, exact short lists
template <int KMAX, bool EXACT>
__global__ void __launch_bounds__(NODE_BLOCK)
knn_points3_kernel(const int64_t n, const int64_t m, const int D, const int K, const float* __restrict__ p1, const float* __restrict__ p2,
                   float* __restrict__ dist2, int64_t* __restrict__ idx)
{
    __shared__ float4 s_pos4[NODE_CHUNK4];
    const int64_t i = (int64_t)blockIdx.x * NODE_BLOCK + threadIdx.x;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
#pragma unroll
        for (int d = 0; d < 4; d++) x[d] = d < D ? p1[i * D + d] : 0.f;
    }
    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) { bd[k] = INFINITY; bi[k] = -1; }
    float worst = INFINITY;
    for (int64_t base = 0; base < m; base += NODE_CHUNK4) {
        const int cnt = (int)min((int64_t)NODE_CHUNK4, m - base);
        __syncthreads();
        stage_points3(s_pos4, p2, base, cnt, D, D);
        __syncthreads();
        if (i < n) scan_points3<KMAX, EXACT, true>(s_pos4, cnt, (int)base, x, bd, bi, K, worst);
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

// The same for small candidate sets and longer lists (m <= 64 * CMAX, K > 4), batched: ONE WAVE per query. The thread-per-query kernel
// above needs n >> 10^4 queries to fill the chip and pays a divergent register insertion per accepted candidate; the ARAP term of the
// dynamic mapping loop asks for the 11 nearest of 512 nodes for 512 queries x ~10 time samples per iteration (deform_utils.py:74) --
// 2 blocks of work per call, 385 us each. Here a lane holds candidates lane, lane + 64, ... as 64-bit keys (distance bits << 32 | index:
// distances are >= +0, so their bit patterns order like the values, and the index breaks ties the way the insertion lists do), and
// the K nearest are K rounds of "lane minimum -> wave minimum -> knock out". Same distance arithmetic, same order, same padding.
template <int CMAX>
__global__ void __launch_bounds__(NODE_BLOCK)
knn_points3_wave_kernel(const int64_t n, const int64_t m, const int D, const int K, const float* __restrict__ p1, const float* __restrict__ p2,
                        float* __restrict__ dist2, int64_t* __restrict__ idx)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (NODE_BLOCK / 64) + (threadIdx.x >> 6);
    if (i >= n) return;                                            // wave-uniform
    const int64_t b = blockIdx.y;
    p1 += (b * n + i) * D; p2 += b * m * D; dist2 += (b * n + i) * K; idx += (b * n + i) * K;
    float x[4];
#pragma unroll
    for (int d = 0; d < 4; d++) x[d] = d < D ? p1[d] : 0.f;
    constexpr unsigned long long NONE = ~0ull;
    unsigned long long key[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; c++) {
        const int j = c * 64 + lane;
        key[c] = NONE;
        if (j < m) {
            float q[4];
#pragma unroll
            for (int d = 0; d < 4; d++) q[d] = d < D ? p2[(int64_t)j * D + d] : 0.f;
            const float tx = x[0] - q[0], ty = x[1] - q[1], tz = x[2] - q[2], tw = x[3] - q[3];
            const float d2 = fmaf(tw, tw, fmaf(tz, tz, fmaf(ty, ty, tx * tx)));
            if (d2 == d2) key[c] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j;        // a NaN distance is never accepted
        }
    }
    unsigned long long mine = NONE;                                // lane k keeps the k-th nearest
    for (int k = 0; k < K; k++) {
        unsigned long long w = key[0];
#pragma unroll
        for (int c = 1; c < CMAX; c++) w = key[c] < w ? key[c] : w;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const unsigned long long o = __shfl_xor(w, s);
            w = o < w ? o : w;
        }
#pragma unroll
        for (int c = 0; c < CMAX; c++) key[c] = key[c] == w ? NONE : key[c];
        if (lane == k) mine = w;
    }
    if (lane < K) {
        const bool ok = mine != NONE;
        dist2[lane] = ok ? __uint_as_float((unsigned)(mine >> 32)) : 0.f;
        idx[lane] = ok ? (int64_t)(mine & 0xffffffffull) : 0;
    }
}

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

struct NodeFrame { float R[9]; };

// floats between two nodes' rows of node_trans / node_scale (3 packed) and node_rot / node_local_rotation (4 packed): attr_stride != 0 when the
// four are column ranges of one [., m, attr_stride] matrix (the node network's heads as it produces them)
__device__ __forceinline__ size_t node_s3(const gsr_node_blend& a) { return a.attr_stride ? (size_t)a.attr_stride : 3; }
__device__ __forceinline__ size_t node_s4(const gsr_node_blend& a) { return a.attr_stride ? (size_t)a.attr_stride : 4; }

// quaternion_to_matrix(local_rotation + (1,0,0,0)), utils/time_utils.py:115-133,1207-1208
__device__ __forceinline__ NodeFrame node_frame_of(const gsr_node_blend& a, int j)
{
    NodeFrame f;
    if (!a.node_local_rotation) {
#pragma unroll
        for (int c = 0; c < 9; c++) f.R[c] = a.node_frame[9 * (size_t)j + c];
        return f;
    }
    const float* q = a.node_local_rotation + node_s4(a) * (size_t)j;
    const float r = q[0] + 1.f, x = q[1], y = q[2], z = q[3];
    const float s = 2.0f / (r * r + x * x + y * y + z * z);
    f.R[0] = 1.f - s * (y * y + z * z); f.R[1] = s * (x * y - z * r); f.R[2] = s * (x * z + y * r);
    f.R[3] = s * (x * y + z * r); f.R[4] = 1.f - s * (x * x + z * z); f.R[5] = s * (y * z - x * r);
    f.R[6] = s * (x * z - y * r); f.R[7] = s * (y * z + x * r); f.R[8] = 1.f - s * (x * x + y * y);
    return f;
}

__device__ __forceinline__ float node_radius_of(const gsr_node_blend& a, int j)
{
    const float v = a.node_radius[j];
    return (a.flags & GSR_NODE_RADIUS_IS_LOG) ? expf(v) : v;
}

__device__ __forceinline__ float node_weight_of(const gsr_node_blend& a, int j)
{
    if (!a.node_weight) return 1.f;
    const float v = a.node_weight[j];
    return (a.flags & GSR_NODE_WEIGHT_IS_LOGIT) ? 1.f / (1.f + expf(-v)) : v;
}

// Batched blends (gsr_node_blend_*_batch): blockIdx.y selects one of B sets of node attributes [B, m, .] -- the time samples of one
// mapping iteration, same Gaussians, same nodes. With gridDim.y = 1 this is the plain call.
__device__ __forceinline__ gsr_node_blend batch_element(gsr_node_blend a, int b)
{
    const size_t m = (size_t)a.m;
    if (a.node_trans) a.node_trans += (size_t)b * m * node_s3(a);
    if (a.node_rot) a.node_rot += (size_t)b * m * node_s4(a);
    if (a.node_scale) a.node_scale += (size_t)b * m * node_s3(a);
    if (a.node_frame) a.node_frame += (size_t)b * m * 9;
    if (a.node_local_rotation) a.node_local_rotation += (size_t)b * m * node_s4(a);
    return a;
}

// ---- cal_nn_weight + blend ------------------------------------------------------------------------------------------------------
template <int KMAX, bool EXACT>
__global__ void __launch_bounds__(NODE_BLOCK)
node_blend_fwd_kernel(const gsr_node_blend a_, float* __restrict__ nn_weight, float* __restrict__ nn_dist, int64_t* __restrict__ nn_idx,
                      float* __restrict__ d_xyz, float* __restrict__ d_rotation, float* __restrict__ d_scaling)
{
    __shared__ float4 s_pos4[NODE_CHUNK4];
    const gsr_node_blend a = batch_element(a_, (int)blockIdx.y);
    const bool first = blockIdx.y == 0;                  // the neighbours and weights are the same for every batch element: written once
    if (d_xyz) { d_xyz += (size_t)blockIdx.y * a.n * 3; d_rotation += (size_t)blockIdx.y * a.n * 4; d_scaling += (size_t)blockIdx.y * a.n * 3; }
    const int64_t i = (int64_t)blockIdx.x * NODE_BLOCK + threadIdx.x;
    const int K = a.K;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < a.n) { x[0] = a.x[3 * i]; x[1] = a.x[3 * i + 1]; x[2] = a.x[3 * i + 2]; }
    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) { bd[k] = INFINITY; bi[k] = -1; }
    float worst = INFINITY;
    for (int base = 0; base < a.m; base += NODE_CHUNK4) {
        const int cnt = min(NODE_CHUNK4, a.m - base);
        __syncthreads();
        stage_points3(s_pos4, a.nodes, base, cnt, a.node_stride, 3);
        __syncthreads();
        if (i < a.n) scan_points3<KMAX, EXACT, false>(s_pos4, cnt, base, x, bd, bi, K, worst);
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
            const float r = node_radius_of(a, j);
            const float u = expf(-bd[k] / (2.f * r * r)) * node_weight_of(a, j);
            w[k] = u + 1e-7f;
            S += w[k];
        }
    }
    const float invS = 1.f / S;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        if (k < K) {
            w[k] *= invS;
            if (first) {
                nn_weight[i * K + k] = w[k];
                nn_dist[i * K + k] = bd[k];
                nn_idx[i * K + k] = bi[k];
            }
        }
    }
    if (!a.node_trans) return;
    const float mask = a.motion_mask ? a.motion_mask[i] : 1.f;
    float t[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        if (k < K) {
            const int j = bi[k];
            const float* tr = a.node_trans + node_s3(a) * (size_t)j;
            if (a.local_frame) {                                  // R (x - node) + node + trans (:1209)
                const NodeFrame F = node_frame_of(a, j);
                const float* R = F.R;
                const float* nd = a.nodes + (size_t)j * a.node_stride;
                const float ox = x[0] - nd[0], oy = x[1] - nd[1], oz = x[2] - nd[2];
#pragma unroll
                for (int c = 0; c < 3; c++) t[c] += w[k] * ((R[3 * c] * ox + R[3 * c + 1] * oy + R[3 * c + 2] * oz) + nd[c] + tr[c]);
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) t[c] += w[k] * tr[c];  // :1213
            }
            const float* qr = a.node_rot + node_s4(a) * (size_t)j;
#pragma unroll
            for (int c = 0; c < 4; c++) q[c] += w[k] * (qr[c] + ((!a.rot_as_residual && c == 0) ? 1.f : 0.f));
            const float* sc = a.node_scale + node_s3(a) * (size_t)j;
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

// ---- deterministic scatter-add: the reverse lists of an index array, then ordered segment sums (round 4) ----------------------------------
// Several backward passes of the dynamic mapping loop are scatter-adds through an index array -- a Gaussian's gradient goes to its K nearest
// control nodes (node_blend_bwd), a node's to its K nearest neighbours (the ARAP / elastic regularisers' gathers). Float atomics (LDS or
// global, here or inside torch's index_put / scatter_add backward) add in whatever order the hardware schedules them: the sums differ in the
// last bits from run to run, and Adam with eps = 1e-15 turns that into visibly different maps. Instead:
//   index_csr_kernel     one WAVE per (index set s, target v): scans idx[s][0..E) in order, counts its matches, reserves a segment of
//                        order[s][..] (an integer atomic: WHERE the segment lies varies, its content does not) and writes the matching
//                        positions e in increasing order;  seg[s][v] = {begin, count}
//   segment_sum_kernel   one thread per (batch b, target v, channel c): out[b][v][c] = sum over the segment, in that order, of g[b][e][c]
// -- a fixed summation order, bit-reproducible results.
__global__ void __launch_bounds__(256)
index_csr_kernel(const int E, const int Nv, const int64_t* __restrict__ idx, int* __restrict__ order, int* __restrict__ seg, int* __restrict__ cursor)
{
    // U chunks of 64 entries per trip, all loads of a trip issued before the first compare: a trip costs ONE L2 latency instead of U (the
    // scan is a chain of dependent loads otherwise: 145 us per mapping iteration before, at 512 targets x 6 000 entries)
    constexpr int U = 8;
    const int lane = threadIdx.x & 63, v = blockIdx.x * 4 + (threadIdx.x >> 6), s = blockIdx.y;
    if (v >= Nv) return;
    idx += (size_t)s * E; order += (size_t)s * E;
    int count = 0;
    for (int e0 = 0; e0 < E; e0 += 64 * U) {
        int64_t w[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int e = e0 + 64 * u + lane; w[u] = e < E ? idx[e] : (int64_t)-1; }
#pragma unroll
        for (int u = 0; u < U; u++) count += (int)__popcll(__ballot(w[u] == (int64_t)v));
    }
    int base = 0;
    if (lane == 0) base = count ? atomicAdd(&cursor[s], count) : 0;
    base = __shfl(base, 0, 64);
    int running = 0;
    for (int e0 = 0; e0 < E && running < count; e0 += 64 * U) {
        int64_t w[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int e = e0 + 64 * u + lane; w[u] = e < E ? idx[e] : (int64_t)-1; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool hit = w[u] == (int64_t)v;
            const unsigned long long m = __ballot(hit);
            if (hit) order[base + running + (int)__popcll(m & ((1ull << lane) - 1ull))] = e0 + 64 * u + lane;
            running += (int)__popcll(m);
        }
    }
    if (lane == 0) { seg[2 * ((size_t)s * Nv + v)] = base; seg[2 * ((size_t)s * Nv + v) + 1] = count; }
}

// The same lists for SMALL index sets (E <= 10 240 entries, fewer than 65 535 targets: the regularisers' neighbour sets -- 512 nodes x 10
// neighbours --, the warp's Gaussian -> node lists of a few thousand dynamic Gaussians). The wave-per-target kernel above is a chain of
// memory latencies: ten trips per pass, two passes, 3.5 waves per SIMD -- 20-35 us per call. Here the set is first packed to 16-bit indices
// (index_csr_pack_kernel), then a wave fetches the WHOLE set with one 16-byte load per lane and 512 entries (all loads in flight at once) and
// counts, reserves and places from registers: one memory latency per wave.
constexpr int CSR_REG_TRIP = 512;                        // entries per wave-wide 16-byte load: lane l holds entries 8 l .. 8 l + 7 of the trip
__global__ void __launch_bounds__(256)
index_csr_pack_kernel(const int E, const int Epad, const int64_t* __restrict__ idx, unsigned short* __restrict__ packed)
{
    const int e = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (e < Epad) packed[(size_t)s * Epad + e] = e < E ? (unsigned short)idx[(size_t)s * E + e] : (unsigned short)0xFFFF;
}

template <int TRIPS>
__global__ void __launch_bounds__(256)
index_csr_reg_kernel(const int E, const int Epad, const int Nv, const unsigned short* __restrict__ packed, int* __restrict__ order, int* __restrict__ seg,
                     int* __restrict__ cursor)
{
    const int lane = threadIdx.x & 63, v = blockIdx.x * 4 + (threadIdx.x >> 6), s = blockIdx.y;
    if (v >= Nv) return;
    packed += (size_t)s * Epad; order += (size_t)s * E;
    uint4 w[TRIPS];
#pragma unroll
    for (int r = 0; r < TRIPS; r++)
        w[r] = r * CSR_REG_TRIP < Epad ? *reinterpret_cast<const uint4*>(packed + r * CSR_REG_TRIP + 8 * lane) : make_uint4(~0u, ~0u, ~0u, ~0u);
    const unsigned int vv = (unsigned int)v;
    auto hits = [&](const uint4& q) __attribute__((always_inline)) {      // bit j: entry 8 lane + j of the trip points at v
        return (unsigned int)((q.x & 0xFFFFu) == vv) | ((unsigned int)((q.x >> 16) == vv) << 1) | ((unsigned int)((q.y & 0xFFFFu) == vv) << 2) |
               ((unsigned int)((q.y >> 16) == vv) << 3) | ((unsigned int)((q.z & 0xFFFFu) == vv) << 4) | ((unsigned int)((q.z >> 16) == vv) << 5) |
               ((unsigned int)((q.w & 0xFFFFu) == vv) << 6) | ((unsigned int)((q.w >> 16) == vv) << 7);
    };
    int count = 0;
#pragma unroll
    for (int r = 0; r < TRIPS; r++) {
        const unsigned int h = hits(w[r]);
        int c = __popc(h);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        count += c;
    }
    int base = 0;
    if (lane == 0) base = count ? atomicAdd(&cursor[s], count) : 0;
    base = __shfl(base, 0, 64);
    int running = 0;
#pragma unroll
    for (int r = 0; r < TRIPS; r++) {
        const unsigned int h = hits(w[r]);
        if (!__ballot(h != 0)) continue;
        const int mine = __popc(h);
        int incl = mine;                                  // inclusive prefix over the lanes: the hits of lower lanes come first (positions 8 l + j)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        int pos = base + running + incl - mine;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (h & (1u << j)) order[pos++] = r * CSR_REG_TRIP + 8 * lane + j;
        running += __shfl(incl, 63, 64);
    }
    if (lane == 0) { seg[2 * ((size_t)s * Nv + v)] = base; seg[2 * ((size_t)s * Nv + v) + 1] = count; }
}

// SEG_LANES adjacent lanes share one (batch, target, channel): lane j adds the entries j, j + SEG_LANES, ... of the segment in that order, the
// lanes' sums are added pairwise ((0 + 1) + (2 + 3)) + ... by xor shuffles -- a fixed order. (One thread per output walked the whole segment:
// the busiest control node has hundreds of Gaussians, and every step of the walk is an index load followed by a dependent gradient load.)
constexpr int SEG_LANES = 8;
__global__ void __launch_bounds__(256)
segment_sum_kernel(const int B, const int E, const int C, const int Nv, const float* __restrict__ g, const size_t g_batch_stride,
                   const int* __restrict__ order, const int* __restrict__ seg, const int* __restrict__ set_of_b, float* __restrict__ out,
                   const size_t out_batch_stride)
{
    const size_t tt = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t t = tt / SEG_LANES;
    const int j = (int)(tt % SEG_LANES);
    const bool live = t < (size_t)B * Nv * C;
    float acc = 0.f;
    int c = 0, v = 0, b = 0;
    if (live) {
        c = (int)(t % C); v = (int)((t / C) % Nv); b = (int)(t / ((size_t)C * Nv));
        const int s = set_of_b ? set_of_b[b] : 0;
        const int begin = seg[2 * ((size_t)s * Nv + v)], count = seg[2 * ((size_t)s * Nv + v) + 1];
        const int* o = order + (size_t)s * E + begin;
        const float* gb = g + (size_t)b * g_batch_stride;
        for (int k = j; k < count; k += SEG_LANES) acc += gb[(size_t)o[k] * C + c];
    }
#pragma unroll
    for (int off = 1; off < SEG_LANES; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (live && j == 0) out[(size_t)b * out_batch_stride + (size_t)v * C + c] = acc;
}

// ---- several small sums in one launch (round 5) -------------------------------------------------------------------------------------------
// A row of the control-node warp's output (d_xyz / d_rotation / d_scaling of one time sample) is read by up to three rasterizer calls of a
// dynamic mapping iteration (the keyframe's render, its flow render, the partner's flow render); autograd adds the calls' gradients of that
// row pairwise -- 32 launches of 5 000-element kernels per iteration plus the re-stacking of the rows. multi_add_kernel forms every row of
// the stacked gradient in one launch: dst[j] = src0[j] + src1[j] + src2[j] + src3[j] (null sources skipped, that order), zero without sources.
constexpr int MULTI_ADD_MAX = 64, MULTI_ADD_SOURCES = 4;
struct MultiAddItem { float* dst; const float* src[MULTI_ADD_SOURCES]; int count; };
struct MultiAddItems { MultiAddItem item[MULTI_ADD_MAX]; };
__global__ void __launch_bounds__(256) multi_add_kernel(const MultiAddItems items)
{
    const MultiAddItem& it = items.item[blockIdx.y];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < it.count; j += gridDim.x * 256) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < MULTI_ADD_SOURCES; s++) if (it.src[s]) t += it.src[s][j];
        it.dst[j] = t;
    }
}

// ---- the node network's input (DeformNetwork's embedders, utils/time_utils.py:208-273: include_input, log-spaced sin / cos) for n time
// samples x M nodes in one launch: row (i, m) = [x_m, sin(2^0 x_m), cos(2^0 x_m), ..., | t_i, sin(2^0 t_i), cos(2^0 t_i), ...]. As tensor
// ops that is two embeddings of four launches each, two expands and a concatenation on the way to the same [n M, 3 (1 + 2 Fx) + 1 + 2 Ft]
// matrix. sinf / cosf (not the fast intrinsics: the values feed a network whose state_dict is the reference's).
// two launches: the embeddings of the M nodes and of the n times once each ([M][Wx] and [n][Wt] in `tables`), then their broadcast
__global__ void __launch_bounds__(256)
node_embedding_tables_kernel(const int n, const int M, const int Fx, const int Ft, const float* __restrict__ nodes, const int node_stride,
                             const float* __restrict__ tt, float* __restrict__ tables)
{
    const int Wx = 3 * (1 + 2 * Fx), Wt = 1 + 2 * Ft;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= M * Wx + n * Wt) return;
    int k, C;
    float x;
    if (e < M * Wx) { k = e % Wx; C = 3; x = nodes[(size_t)(e / Wx) * node_stride + k % 3]; }
    else { const int r = e - M * Wx; k = r % Wt; C = 1; x = tt[r / Wt]; }
    const int blk = k / C;                                 // blk 0: the input itself; 1 + 2 f: sin of frequency f; 2 + 2 f: cos
    float v = x;
    if (blk) {
        const float a = x * (float)(1u << ((blk - 1) >> 1));
        v = ((blk - 1) & 1) ? cosf(a) : sinf(a);
    }
    tables[e] = v;
}
__global__ void __launch_bounds__(256)
node_embedding_expand_kernel(const int n, const int M, const int Wx, const int Wt, const float* __restrict__ tables, float* __restrict__ out)
{
    const int Wd = Wx + Wt;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * M * Wd) return;
    const int col = (int)(e % Wd);
    const size_t row = e / Wd;
    const int m = (int)(row % M), i = (int)(row / M);
    out[e] = col < Wx ? tables[(size_t)m * Wx + col] : tables[(size_t)M * Wx + (size_t)i * Wt + (col - Wx)];
}

// ---- the node network's trunk layer, backward half that is not a GEMM (slam/deform_model.py NodeNetwork.trunk) -------------------------------
// A layer is y = relu(x W^T + b) on ~50 000 rows x 256 columns. Its backward pass needs G = dY . [y > 0] (then dW = G^T x and dx = G W are
// GEMMs) and db = column sums of G. As torch ops that is threshold_backward (read dY, y; write G) plus a column reduction that reads G again
// and runs at 1.7 TB/s: 8 layers x (35 + 30) us per mapping iteration. relu_bwd_bias_kernel forms G and the column sums of a 64-row band in
// one pass (a thread owns four adjacent columns of every (256 / (cols / 4))-th row of its band, the waves' partial sums are added in wave
// order through LDS); colsum_finalize_kernel adds the bands. Fixed summation order: bit-reproducible.
constexpr int RELU_BAND = 64;
__global__ void __launch_bounds__(256)
relu_bwd_bias_kernel(const int rows, const int cols, const float* __restrict__ dY, const float* __restrict__ Y, float* __restrict__ G,
                     float* __restrict__ partial /* [bands][cols] */)
{
    __shared__ float4 s_sum[256];
    const int tpr = cols >> 2;                                    // threads per row
    const int c4 = threadIdx.x % tpr, r_in = threadIdx.x / tpr, rstep = 256 / tpr;
    const int r0 = blockIdx.x * RELU_BAND, r1 = min(rows, r0 + RELU_BAND);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0 + r_in; r < r1; r += rstep) {
        const size_t o = ((size_t)r * cols >> 2) + c4;
        const float4 d = reinterpret_cast<const float4*>(dY)[o], y = reinterpret_cast<const float4*>(Y)[o];
        const float4 g = make_float4(y.x > 0.f ? d.x : 0.f, y.y > 0.f ? d.y : 0.f, y.z > 0.f ? d.z : 0.f, y.w > 0.f ? d.w : 0.f);
        reinterpret_cast<float4*>(G)[o] = g;
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
    }
    s_sum[threadIdx.x] = acc;
    __syncthreads();
    if (r_in == 0) {
        float4 t = acc;
        for (int k = 1; k < rstep; k++) { const float4 u = s_sum[k * tpr + c4]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        reinterpret_cast<float4*>(partial)[((size_t)blockIdx.x * cols >> 2) + c4] = t;
    }
}
// 16 columns per block, 16 threads per column: thread (c, p) adds the bands p, p + 16, ... in that order (independent loads, 64-byte
// segments), the 16 partial sums of a column are then added in the order of p
__global__ void __launch_bounds__(256)
colsum_finalize_kernel(const int bands, const int cols, const float* __restrict__ partial, float* __restrict__ out)
{
    __shared__ float s_p[16][17];
    const int cl = threadIdx.x & 15, p = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    float acc = 0.f;
    if (c < cols) {
        int b = p;
        for (; b + 7 * 16 < bands; b += 8 * 16) {            // eight loads in flight per trip (the adds stay in band order)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = partial[(size_t)(b + 16 * u) * cols + c];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u];
        }
        for (; b < bands; b += 16) acc += partial[(size_t)b * cols + c];
    }
    s_p[p][cl] = acc;
    __syncthreads();
    if (p == 0 && c < cols) {
        float t = s_p[0][cl];
#pragma unroll
        for (int k = 1; k < 16; k++) t += s_p[k][cl];
        out[c] = t;
    }
}

// partial: [gridDim.x][m * NODE_GRAD]; use_lds = 0: every block adds into row 0 with global atomics (caller zeroed it).
// contrib != nullptr (the deterministic route, K <= 4): nothing is accumulated here; the 21 values a Gaussian sends to its k-th node are
// WRITTEN to contrib[b][(i K + k)][0..21) and summed per node by index_csr_kernel + segment_sum_kernel over nn_idx.
__global__ void __launch_bounds__(NODE_BLOCK)
node_blend_bwd_kernel(const gsr_node_blend a_, const float* __restrict__ nn_weight, const float* __restrict__ nn_dist,
                      const int64_t* __restrict__ nn_idx, const float* __restrict__ g_xyz, const float* __restrict__ g_rotation,
                      const float* __restrict__ g_scaling, const float* __restrict__ g_nn_weight, float* __restrict__ partial, const int use_lds,
                      const size_t batch_stride /* floats of workspace per batch element */, float* __restrict__ contrib, const size_t contrib_stride)
{
    constexpr int KMAX = GSR_BLEND_MAX_K;
    extern __shared__ float s_acc[];                              // [m][NODE_GRAD] when use_lds
    const gsr_node_blend a = batch_element(a_, (int)blockIdx.y);
    if (g_xyz) g_xyz += (size_t)blockIdx.y * a.n * 3;
    if (g_rotation) g_rotation += (size_t)blockIdx.y * a.n * 4;
    if (g_scaling) g_scaling += (size_t)blockIdx.y * a.n * 3;
    partial += (size_t)blockIdx.y * batch_stride;
    if (contrib) contrib += (size_t)blockIdx.y * contrib_stride;
    const int K = a.K;
    const int total = a.m * NODE_GRAD;
    if (use_lds && !contrib) {
        for (int e = threadIdx.x; e < total; e += NODE_BLOCK) s_acc[e] = 0.f;
        __syncthreads();
    }
    float* acc = use_lds ? s_acc : partial;
    float* crow = nullptr;                                         // contrib row of the (Gaussian, k) pair being processed
    auto add = [&](int e, float v) {
        if (contrib) { crow[e % NODE_GRAD] = v; return; }          // every component is written exactly once per pair (zero rows are pre-filled below)
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
                if (contrib) {
                    crow = contrib + ((size_t)i * K + k) * NODE_GRAD;
#pragma unroll
                    for (int c = 0; c < NODE_GRAD; c++) crow[c] = 0.f;
                }
                if (blend) {
                    const float* tr = a.node_trans + node_s3(a) * (size_t)j;
                    const float* qr = a.node_rot + node_s4(a) * (size_t)j;
                    const float* sc = a.node_scale + node_s3(a) * (size_t)j;
                    if (a.local_frame) {
                        const NodeFrame F = node_frame_of(a, j);
                        const float* R = F.R;
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
                const float r = node_radius_of(a, idx[k]);
                e[k] = expf(-nn_dist[i * K + k] / (2.f * r * r));
                nw[k] = node_weight_of(a, idx[k]);
                S += e[k] * nw[k] + 1e-7f;
            }
        }
        const float invS = 1.f / S;
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            if (k < K) {
                const float du = (G[k] - Gw) * invS;
                const float r = node_radius_of(a, idx[k]);
                if (contrib) crow = contrib + ((size_t)i * K + k) * NODE_GRAD;
                add(idx[k] * NODE_GRAD + 19, du * nw[k] * e[k] * nn_dist[i * K + k] / (r * r * r));      // d e / d r = e d / r^3
                if (a.node_weight) add(idx[k] * NODE_GRAD + 20, du * e[k]);
            }
        }
    }
    if (use_lds && !contrib) {
        __syncthreads();
        float* row = partial + (size_t)blockIdx.x * total;
        for (int e = threadIdx.x; e < total; e += NODE_BLOCK) row[e] = s_acc[e];
    }
}

// summed[e] = sum over the G partial rows of element e (node j, component c = e % 21), fixed order
__global__ void __launch_bounds__(256)
node_grad_reduce_kernel(const int G, const int total, const float* __restrict__ partial, float* __restrict__ summed, const size_t batch_stride)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    partial += (size_t)blockIdx.y * batch_stride;
    summed += (size_t)blockIdx.y * batch_stride;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < G; b += 4) {
        s0 += partial[(size_t)b * total + e];
        s1 += partial[(size_t)(b + 1) * total + e];
        s2 += partial[(size_t)(b + 2) * total + e];
        s3 += partial[(size_t)(b + 3) * total + e];
    }
    for (; b < G; b++) s0 += partial[(size_t)b * total + e];
    summed[e] = (s0 + s1) + (s2 + s3);
}

// one thread per node: hand the 21 sums out, through the chain rules of the per-node activations where the inputs were raw
__global__ void __launch_bounds__(256)
node_grad_finalize_kernel(const gsr_node_blend a_, const float* __restrict__ summed, float* __restrict__ g_trans, float* __restrict__ g_rot,
                          float* __restrict__ g_scale, float* __restrict__ g_frame, float* __restrict__ g_radius, float* __restrict__ g_weight,
                          const size_t batch_stride)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const gsr_node_blend a = batch_element(a_, (int)blockIdx.y);
    if (j >= a.m) return;
    // (grad_stride != 0: the four attribute gradients are column ranges of one [B, m, grad_stride] matrix, like the attributes with attr_stride)
    const size_t g3 = a.grad_stride ? (size_t)a.grad_stride : 3, g4 = a.grad_stride ? (size_t)a.grad_stride : 4;
    {   // every output is [B, m, .]: radius / weight gradients per batch element too (the caller sums them over B)
        const size_t b = blockIdx.y, m = (size_t)a.m;
        summed += b * batch_stride;
        if (g_trans) g_trans += b * m * g3;
        if (g_rot) g_rot += b * m * g4;
        if (g_scale) g_scale += b * m * g3;
        if (g_frame) g_frame += b * m * (a.node_local_rotation ? g4 : 9);
        if (g_radius) g_radius += b * m;
        if (g_weight) g_weight += b * m;
    }
    const float* s = summed + (size_t)j * NODE_GRAD;
    if (g_trans) { g_trans[g3 * j] = s[0]; g_trans[g3 * j + 1] = s[1]; g_trans[g3 * j + 2] = s[2]; }
    if (g_rot) { g_rot[g4 * j] = s[3]; g_rot[g4 * j + 1] = s[4]; g_rot[g4 * j + 2] = s[5]; g_rot[g4 * j + 3] = s[6]; }
    if (g_scale) { g_scale[g3 * j] = s[7]; g_scale[g3 * j + 1] = s[8]; g_scale[g3 * j + 2] = s[9]; }
    if (g_radius) g_radius[j] = (a.flags & GSR_NODE_RADIUS_IS_LOG) ? s[19] * expf(a.node_radius[j]) : s[19];       // d exp(v) = exp(v) dv
    if (g_weight && a.node_weight) {
        float g = s[20];
        if (a.flags & GSR_NODE_WEIGHT_IS_LOGIT) { const float w = 1.f / (1.f + expf(-a.node_weight[j])); g *= w * (1.f - w); }
        g_weight[j] = g;
    }
    if (!g_frame) return;
    const float* Gm = s + 10;                                         // dL/dR, row-major
    if (!a.node_local_rotation) {
#pragma unroll
        for (int c = 0; c < 9; c++) g_frame[9 * j + c] = Gm[c];
        return;
    }
    // R = I + s A(q), s = 2 / |q|^2, A homogeneous quadratic (time_utils.py:115-133); q = local_rotation + (1,0,0,0)
    const float* q = a.node_local_rotation + node_s4(a) * (size_t)j;
    const float r = q[0] + 1.f, x = q[1], y = q[2], z = q[3];
    const float n2 = r * r + x * x + y * y + z * z, sc = 2.0f / n2;
    const float A[9] = {-(y * y + z * z), x * y - z * r, x * z + y * r, x * y + z * r, -(x * x + z * z), y * z - x * r,
                        x * z - y * r, y * z + x * r, -(x * x + y * y)};
    float GA = 0.f;
#pragma unroll
    for (int c = 0; c < 9; c++) GA += Gm[c] * A[c];
    const float dr = -z * Gm[1] + y * Gm[2] + z * Gm[3] - x * Gm[5] - y * Gm[6] + x * Gm[7];
    const float dx = y * Gm[1] + z * Gm[2] + y * Gm[3] - 2.f * x * Gm[4] - r * Gm[5] + z * Gm[6] + r * Gm[7] - 2.f * x * Gm[8];
    const float dy = -2.f * y * Gm[0] + x * Gm[1] + r * Gm[2] + x * Gm[3] + z * Gm[5] - r * Gm[6] + z * Gm[7] - 2.f * y * Gm[8];
    const float dz = -2.f * z * Gm[0] - r * Gm[1] + x * Gm[2] + r * Gm[3] - 2.f * z * Gm[4] + y * Gm[5] + x * Gm[6] + y * Gm[7];
    const float k = -sc * sc * GA;                                    // d s / d q = -s^2 q
    g_frame[g4 * j] = sc * dr + k * r;
    g_frame[g4 * j + 1] = sc * dx + k * x;
    g_frame[g4 * j + 2] = sc * dy + k * y;
    g_frame[g4 * j + 3] = sc * dz + k * z;
}

}  // namespace gsr
