// gs_hexplane_binned.h -- HexPlane backward with the points SORTED by texel cell, so that points sharing a cell are merged in
// registers and the plane gradients receive one atomic per run of points instead of one per point.
//
// Why: the direct kernel (hexplane_bwd_lane_kernel) issues n x 96 texel-wide float atomics; the atomic path of the chip sustains
// ~1.2 TB/s of payload when the points are spread over the whole plane and less when they are not (config #3: the scene fills a
// part of the aabb, every view has one time -> 8.2 ms at 500k points, 65 % of the whole mapping iteration).  Accumulating plane
// regions in LDS instead was measured and rejected: ds_add_f32 runs at ~0.3 lanes per clock per CU on this part (3.3 ms for the
// 614 M lane-adds of a 200k-point batch).
//
// How: per plane family (xy, xz, xt, yz, yt, zt) a counting sort of the points by the Morton code of their finest-level cell
// (one histogram over all families, one scan, one scatter that also lays the coordinates out in sorted order).  Phase 1 (per point)
// gathers the samples, forms dL/dsample for all 24 planes and writes each family's slice to the point's SORTED slot, plus dL/dxyz.
// Phase 2: a group of C lanes (one channel each) walks HEXSORT_CHUNK consecutive sorted points of one family, reading coordinates
// and dL/dsample sequentially; per level it keeps the four corner sums of the current cell in registers and flushes them with four
// texel-wide atomics when the cell changes.  Because aligned 2^k x 2^k blocks of fine cells are contiguous in Morton order, the
// coarser levels' cells form long runs too.  Atomics fall from n x 96 to about the number of distinct cells touched; a crowded
// plane gets cheaper, not dearer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gs_hexplane.h"

namespace gsr {

constexpr int HEXSORT_CHUNK = 64;               // consecutive sorted points per group in phase 2 (32: 795 us, 64: 756, 256: 1328 at 200k points)
constexpr int HEXSORT_SCAN_ITEMS = 8;           // counters per thread in the scan kernels (1024 threads -> 8192 per block)

struct HexSortPlan {
    int fine[4];                                // finest resolution along x, y, z, t
    int bits[4];                                // bits of a cell index along each coordinate
    int sub_bits[6];                            // log2 of the sub-counters per cell (spreads the histogram / cursor atomics of a crowded cell)
    int key_off[7];                             // first counter of each family; [6] = total number of counters
};

struct HexSortWs {                              // device pointers carved from the caller's workspace
    float4* coords;                             // [n]       normalised (x, y, z) and raw t
    uint32_t* key;                              // [6][n]    counter index of the point in each family
    uint32_t* count;                            // [NB]      histogram, then (after the scan) the running cursor
    uint32_t* block_sums;                       // [ceil(NB / 8192)]
    uint32_t* header;                           // [0] = 6 x (number of points with a non-zero cotangent row)
    int* rank;                                  // [6][n]    sorted slot of the point in each family (0 .. 6n)
    float4* scoords;                            // [6n]      coordinates in sorted order
    float* gs;                                  // [6n][L][C]  dL/dsample in sorted order
};

__device__ __forceinline__ uint32_t morton_spread(uint32_t v)   // 0000 abcd -> 0a0b 0c0d (up to 16 bits)
{
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// cell key of family pl: the Morton code of the two cell indices when both axes have the same number of bits, otherwise the
// shorter axis goes on top (the time axis has 25 rows against 512 columns: rows of cells stay contiguous)
__device__ __forceinline__ uint32_t hexsort_key(const HexSortPlan& P, int c0, int c1, int ia, int ib)
{
    if (P.bits[c0] == P.bits[c1]) return morton_spread((uint32_t)ia) | (morton_spread((uint32_t)ib) << 1);
    return ((uint32_t)ib << P.bits[c0]) | (uint32_t)ia;
}

// ---- phase 0a: normalised coordinates, cell keys, histogram ----------------------------------------------------------------------
__global__ void __launch_bounds__(256)
hexsort_count_kernel(const gsr_hexplane_field f, const HexSortPlan P, const HexSortWs ws, const int64_t n, const float* __restrict__ xyz,
                     const int64_t xyz_stride, const float* __restrict__ time, const int64_t time_stride, const float* __restrict__ dL_dfeatures)
{
    // eight lanes per point: they read the point's cotangent row together (coalesced float4 loads), lane 0 of the group then bins it
    const int sub = threadIdx.x & 7;
    const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (i >= n) return;
    // A point whose cotangent row is exactly zero (a Gaussian the view does not see) adds nothing to any plane: it is left out of the
    // sort altogether, so everything after this kernel costs in proportion to the points that carry a gradient.
    const float4* row = reinterpret_cast<const float4*>(dL_dfeatures + i * ((int64_t)f.num_levels * f.feat_dim));
    int active = 0;
    for (int e = sub; e < f.num_levels * f.feat_dim / 4; e += 8) {
        const float4 v = row[e];
        active |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    }
    active |= __shfl_xor(active, 1, 64);
    active |= __shfl_xor(active, 2, 64);
    active |= __shfl_xor(active, 4, 64);
    if (sub != 0) return;
    if (!active) {
#pragma unroll
        for (int pl = 0; pl < 6; pl++) ws.key[(size_t)pl * n + i] = 0xFFFFFFFFu;
        return;
    }
    const HexPoint p = hex_point(f.aabb, xyz + i * xyz_stride, time + i * time_stride);
    ws.coords[i] = make_float4(p.c[0], p.c[1], p.c[2], p.c[3]);
    int i0[4];
#pragma unroll
    for (int k = 0; k < 4; k++) i0[k] = hex_axis(p.c[k], P.fine[k]).i0;
#pragma unroll
    for (int pl = 0; pl < 6; pl++) {
        // the sub-counter (low bits of the point index) only decides the order INSIDE a cell
        const uint32_t k = P.key_off[pl] + ((hexsort_key(P, hex_c0(pl), hex_c1(pl), i0[hex_c0(pl)], i0[hex_c1(pl)]) << P.sub_bits[pl])
                                            | ((uint32_t)i & ((1u << P.sub_bits[pl]) - 1u)));
        ws.key[(size_t)pl * n + i] = k;
        atomicAdd(&ws.count[k], 1u);
    }
}

// ---- phase 0b: exclusive scan of the histogram, in place (three small kernels) ----------------------------------------------------
__global__ void __launch_bounds__(1024)
hexsort_scan_sums_kernel(const uint32_t* __restrict__ count, const int nb, uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t s_w[16];
    const int base = (blockIdx.x * 1024 + threadIdx.x) * HEXSORT_SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < HEXSORT_SCAN_ITEMS; k++) s += base + k < nb ? count[base + k] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; w++) t += s_w[w];
        block_sums[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(1024)
hexsort_scan_top_kernel(uint32_t* __restrict__ block_sums, const int nblocks, uint32_t* __restrict__ header)
{
    __shared__ uint32_t s_tmp[17];
    const uint32_t total = block_exclusive_scan_1024(nblocks, [&](int b) { return block_sums[b]; },
                                                     [&](int b, uint32_t ex, uint32_t) { block_sums[b] = ex; }, s_tmp);
    if (threadIdx.x == 0) header[0] = total;                      // 6 x active points
}

__global__ void __launch_bounds__(1024)
hexsort_scan_apply_kernel(uint32_t* __restrict__ count, const int nb, const uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t s_w[17];
    const int base = (blockIdx.x * 1024 + threadIdx.x) * HEXSORT_SCAN_ITEMS;
    uint32_t v[HEXSORT_SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < HEXSORT_SCAN_ITEMS; k++) { v[k] = base + k < nb ? count[base + k] : 0u; s += v[k]; }
    const uint32_t incl = wave_inclusive_scan(s);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = block_sums[blockIdx.x];
        for (int w = 0; w < 16; w++) { const uint32_t t = s_w[w]; s_w[w] = run; run += t; }
    }
    __syncthreads();
    uint32_t ex = s_w[threadIdx.x >> 6] + incl - s;
#pragma unroll
    for (int k = 0; k < HEXSORT_SCAN_ITEMS; k++) {
        if (base + k < nb) count[base + k] = ex;                  // the cell's first slot; the scatter pass advances it
        ex += v[k];
    }
}

// ---- phase 0c: sorted slots -----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
hexsort_scatter_kernel(const HexSortWs ws, const int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (ws.key[i] == 0xFFFFFFFFu) {                               // no gradient reaches this point
        ws.rank[i] = -1;
        return;
    }
    const float4 c = ws.coords[i];
#pragma unroll
    for (int pl = 0; pl < 6; pl++) {
        const uint32_t slot = atomicAdd(&ws.count[ws.key[(size_t)pl * n + i]], 1u);
        ws.rank[(size_t)pl * n + i] = (int)slot;
        ws.scoords[slot] = c;
    }
}

// ---- phase 1: dL/dsample of all 24 planes (written to the sorted slots) and dL/dxyz ------------------------------------------------
template <int C>
__global__ void __launch_bounds__(HEX_BLOCK)
hexsort_phase1_kernel(const gsr_hexplane_field f, const HexSortWs ws, const int64_t n, const float* __restrict__ xyz, const int64_t xyz_stride,
                      const float* __restrict__ time, const int64_t time_stride, const float* __restrict__ dL_dfeatures,
                      float* __restrict__ dL_dxyz)
{
    const int ch = threadIdx.x % C;
    const int64_t i = (int64_t)blockIdx.x * (HEX_BLOCK / C) + threadIdx.x / C;
    if (i >= n) return;
    if (ws.rank[i] < 0) {                                         // zero cotangent: nothing to gather, nothing to hand over
        if (dL_dxyz && ch < 3) dL_dxyz[3 * i + ch] = 0.f;
        return;
    }
    const HexPoint p = hex_point(f.aabb, xyz + i * xyz_stride, time + i * time_stride);
    const float (&c)[4] = p.c;
    const int L = f.num_levels;
    const float* gout = dL_dfeatures + i * ((int64_t)L * C) + ch;
    size_t slot[6];
#pragma unroll
    for (int pl = 0; pl < 6; pl++) slot[pl] = (size_t)ws.rank[(size_t)pl * n + i] * L * C + ch;
    float gc[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < L; l++) {
        const gsr_hexplane_level& Lv = f.levels[l];
        HexAxis ax[4];
#pragma unroll
        for (int k = 0; k < 4; k++) ax[k] = hex_axis(c[k], Lv.res[k]);
        const float g = gout[(size_t)l * C];
        float s[6], corner[6][4];
#pragma unroll
        for (int pl = 0; pl < 6; pl++) {
            const int W = Lv.res[hex_c0(pl)];
            const HexAxis& X = ax[hex_c0(pl)];
            const HexAxis& Y = ax[hex_c1(pl)];
            const int x1 = X.has1 ? X.i0 + 1 : X.i0, y1 = Y.has1 ? Y.i0 + 1 : Y.i0;
            const float* plane = Lv.planes[pl];
            corner[pl][0] = plane[((size_t)Y.i0 * W + X.i0) * C + ch];
            corner[pl][1] = plane[((size_t)Y.i0 * W + x1) * C + ch];
            corner[pl][2] = plane[((size_t)y1 * W + X.i0) * C + ch];
            corner[pl][3] = plane[((size_t)y1 * W + x1) * C + ch];
        }
#pragma unroll
        for (int pl = 0; pl < 6; pl++) {
            const HexAxis& X = ax[hex_c0(pl)];
            const HexAxis& Y = ax[hex_c1(pl)];
            const float wx1 = X.has1 ? X.w1 : 0.f, wy1 = Y.has1 ? Y.w1 : 0.f;
            float v = corner[pl][0] * (X.w0 * Y.w0);
            v = fmaf(corner[pl][1], wx1 * Y.w0, v);
            v = fmaf(corner[pl][2], X.w0 * wy1, v);
            s[pl] = fmaf(corner[pl][3], wx1 * wy1, v);
        }
        float suffix[6];
        suffix[5] = 1.f;
#pragma unroll
        for (int pl = 4; pl >= 0; pl--) suffix[pl] = suffix[pl + 1] * s[pl + 1];
        float prefix = g;
#pragma unroll
        for (int pl = 0; pl < 6; pl++) {
            const int c0 = hex_c0(pl), c1 = hex_c1(pl);
            const HexAxis& X = ax[c0];
            const HexAxis& Y = ax[c1];
            const float gs = prefix * suffix[pl];
            prefix *= s[pl];
            __builtin_nontemporal_store(gs, &ws.gs[slot[pl] + (size_t)l * C]);   // streamed once: keep the planes in L2 (331 -> 299 us)
            if (dL_dxyz) {
                const float nw = corner[pl][0] * gs, ne = X.has1 ? corner[pl][1] * gs : 0.f, sw = Y.has1 ? corner[pl][2] * gs : 0.f;
                const float se = X.has1 && Y.has1 ? corner[pl][3] * gs : 0.f;
                if (c0 < 3) gc[c0] += ((ne - nw) * Y.w0 + (se - sw) * Y.w1) * X.dmult;
                if (c1 < 3) gc[c1] += ((sw - nw) * X.w0 + (se - ne) * X.w1) * Y.dmult;
            }
        }
    }
    if (dL_dxyz) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int d = 1; d < C; d <<= 1) gc[k] += __shfl_xor(gc[k], d, 64);
        }
        if (ch == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) dL_dxyz[3 * i + k] = gc[k] * p.dscale[k];
        }
    }
}

// ---- phase 2: run-length accumulation along the sorted order --------------------------------------------------------------------------
template <int C, int LMAX>
__global__ void __launch_bounds__(256)
hexsort_phase2_kernel(const gsr_hexplane_field f, const HexSortWs ws, const int64_t n)
{
    constexpr int GROUPS = 256 / C;
    const int ch = threadIdx.x % C;
    const int64_t na = ws.header[0] / 6;                          // points that were sorted (non-zero cotangent); the grid covers n
    const int64_t chunks_per_family = (na + HEXSORT_CHUNK - 1) / HEXSORT_CHUNK;
    const int64_t gid = (int64_t)blockIdx.x * GROUPS + threadIdx.x / C;
    if (gid >= 6 * chunks_per_family) return;
    const int pl = (int)(gid / chunks_per_family);
    const int64_t first = (gid - pl * chunks_per_family) * HEXSORT_CHUNK;
    const int cnt = (int)min((int64_t)HEXSORT_CHUNK, na - first);
    const int c0 = hex_c0(pl), c1 = hex_c1(pl);
    const int L = f.num_levels;
    const float4* sc = ws.scoords + (size_t)pl * na + first;
    const float* gsrow = ws.gs + ((size_t)pl * na + first) * L * C + ch;

    float acc[LMAX][4];
    int cx[LMAX], cy[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; l++) { cx[l] = cy[l] = -1; acc[l][0] = acc[l][1] = acc[l][2] = acc[l][3] = 0.f; }

    auto flush = [&](int l) {                                     // four texel-wide atomics for the cell (cx, cy) of level l
        const gsr_hexplane_level& Lv = f.levels[l];
        float* gp = Lv.grad_planes[pl];
        const int W = Lv.res[c0], H = Lv.res[c1];
        if (gp && cx[l] >= 0) {
            float* t = gp + ((size_t)cy[l] * W + cx[l]) * C + ch;
            const bool x1 = cx[l] + 1 < W, y1 = cy[l] + 1 < H;     // safe_add_2d: corners outside the plane receive nothing
            if (acc[l][0] != 0.f) unsafeAtomicAdd(t, acc[l][0]);
            if (x1 && acc[l][1] != 0.f) unsafeAtomicAdd(t + C, acc[l][1]);
            if (y1 && acc[l][2] != 0.f) unsafeAtomicAdd(t + (size_t)W * C, acc[l][2]);
            if (x1 && y1 && acc[l][3] != 0.f) unsafeAtomicAdd(t + (size_t)(W + 1) * C, acc[l][3]);
        }
        acc[l][0] = acc[l][1] = acc[l][2] = acc[l][3] = 0.f;
    };

    // the loads of point k + 1 (coordinates, dL/dsample of every level) are issued before point k is processed: the walk is
    // sequential, and without this each step waits for its own memory round trip (416 us of the kernel at 200k points)
    float4 cc_next = sc[0];
    float g_next[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; l++) g_next[l] = l < L ? gsrow[(size_t)l * C] : 0.f;
    for (int k = 0; k < cnt; k++) {
        const float4 cc = cc_next;
        float g_cur[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; l++) g_cur[l] = g_next[l];
        const int kn = min(k + 1, cnt - 1);
        cc_next = sc[kn];
#pragma unroll
        for (int l = 0; l < LMAX; l++) g_next[l] = l < L ? gsrow[((size_t)kn * L + l) * C] : 0.f;
        const float c[4] = {cc.x, cc.y, cc.z, cc.w};
#pragma unroll
        for (int l = 0; l < LMAX; l++) {
            if (l < L) {
                const gsr_hexplane_level& Lv = f.levels[l];
                const HexAxis X = hex_axis(c[c0], Lv.res[c0]), Y = hex_axis(c[c1], Lv.res[c1]);
                const float gs = g_cur[l];
                if (X.i0 != cx[l] || Y.i0 != cy[l]) {
                    flush(l);
                    cx[l] = X.i0;
                    cy[l] = Y.i0;
                }
                acc[l][0] = fmaf(gs, X.w0 * Y.w0, acc[l][0]);
                acc[l][1] = fmaf(gs, X.w1 * Y.w0, acc[l][1]);
                acc[l][2] = fmaf(gs, X.w0 * Y.w1, acc[l][2]);
                acc[l][3] = fmaf(gs, X.w1 * Y.w1, acc[l][3]);
            }
        }
    }
#pragma unroll
    for (int l = 0; l < LMAX; l++) {
        if (l < L) flush(l);
    }
}

}  // namespace gsr
