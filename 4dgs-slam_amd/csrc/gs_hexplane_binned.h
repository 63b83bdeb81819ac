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

#include <type_traits>

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
    uint32_t* header;                           // [0] = 6 x (number of points with a non-zero cotangent row); [1] = bits of the largest |dL/dsample| (ordered mode)
    int* rank;                                  // [6][n]    sorted slot of the point in each family (0 .. 6n)
    float4* scoords;                            // [6n]      coordinates in sorted order
    float* gs;                                  // [6n][L][C]  dL/dsample in sorted order
};

// ---- ordered accumulation (gsr_set_option("hex_ordered", 1), the default): plane gradients that are bitwise the same run to run ------------
// Float atomics add in whatever order the waves arrive, and the counting sort's cursors hand out the slots of a cell in arrival order too, so
// the float sums of rounds 1-5 were reproducible to rounding only. Here every contribution  dL/dsample x corner weight  is rounded ONCE to a
// multiple of a power-of-two quantum (2^-40 of the largest |dL/dsample| of the call, found by phase 1) and from there on only integers are
// added -- in registers as integer-valued doubles (exact below 2^53: a walk adds at most HEXSORT_CHUNK values below 2^40), in LDS and in
// memory as 64-bit integer atomics. Integer addition is associative: neither the order inside a cell, nor where the chunks split, nor
// the order of the atomics can change a bit of the result. One pass at the end converts the sums and adds them to the float planes. The time
// families of the batched-views path keep their sums per (view, column) and that pass applies the two time-row weights, views in
// their order.
struct HexOrd {
    unsigned long long* acc;                        // fixed-point sums of the planes the walks scatter into (null: float atomics, rounds 1-5)
    unsigned long long* acc_t;                      // batched views: the time families' column sums [family][view][column of every level][C]
    uint64_t off[GSR_HEXPLANE_MAX_LEVELS][6];       // first element of plane (level, pl) in acc
    uint32_t tcol[3][GSR_HEXPLANE_MAX_LEVELS];      // first column of a level inside one view's row of family j
    uint32_t tcols[3];                              // columns of one view of family j (all levels)
    uint32_t tbase[3];                              // first column row of family j: V x the earlier families' tcols
    int budget;                                     // the largest |dL/dsample| is scaled to just below 2^budget
};

// S = the largest power of two with  max x S < 2^budget  (max given by its bit pattern; 0 when nothing was written), and its inverse
__device__ __forceinline__ int hexord_shift(uint32_t maxbits, int budget)
{
    const int e = (int)(maxbits >> 23) - 127;       // max in [2^e, 2^(e + 1))
    return min(127, max(-126, budget - 1 - e));
}
__device__ __forceinline__ float hexord_scale(uint32_t maxbits, int budget)
{
    return maxbits == 0u ? 0.f : __uint_as_float((uint32_t)(hexord_shift(maxbits, budget) + 127) << 23);
}
__device__ __forceinline__ double hexord_inverse(uint32_t maxbits, int budget)
{
    return maxbits == 0u ? 0.0 : __longlong_as_double((long long)(1023 - hexord_shift(maxbits, budget)) << 52);
}
__device__ __forceinline__ void hexord_add(unsigned long long* p, double v)
{
    if (v != 0.0) atomicAdd(p, (unsigned long long)(long long)v);
}
// the wave's largest finite |dL/dsample| -> header[1] (a maximum is order-free); once the word has grown, almost no wave still has to write.
// A NaN or an infinity among the values (fmaxf drops a NaN, and neither survives the conversion to an integer) raises header[2] instead: the
// last pass then makes every gradient of the call NaN -- as loud as the float path, where the value itself would have reached its texels.
__device__ __forceinline__ float hexord_track(float mx, float v)
{
    const float a = fabsf(v);
    return a <= 3.4028234664e38f ? fmaxf(mx, a) : __uint_as_float(0x7f800000u);          // +inf marks "something was not finite"
}
__device__ __forceinline__ void hexord_publish_max(float mx, uint32_t* header)
{
    const bool bad = !(mx <= 3.4028234664e38f);
    if (bad) mx = 0.f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    const bool any_bad = __any(bad);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t b = __float_as_uint(mx);
        if (b > __hip_atomic_load(header + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(header + 1, b);
        if (any_bad) atomicOr(header + 2, 1u);
    }
}

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
                     const int64_t xyz_stride, const float* __restrict__ time, const int64_t time_stride, const float* __restrict__ dL_dfeatures,
                     const uint32_t* __restrict__ view_mask)
{
    // eight lanes per point: they read the point's cotangent row together (coalesced float4 loads), lane 0 of the group then bins it
    const int sub = threadIdx.x & 7;
    const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (i >= n) return;
    // A point whose cotangent row is exactly zero (a Gaussian the view does not see) adds nothing to any plane: it is left out of the
    // sort altogether, so everything after this kernel costs in proportion to the points that carry a gradient.
    // (the batched-views caller passes no cotangent but, optionally, the views whose cotangent row of the point is not zero, bit v = view v)
    const uint32_t vbits = view_mask ? view_mask[i] : 0xFFFFFFFFu;
    int active = dL_dfeatures ? 0 : (vbits != 0u);
    if (dL_dfeatures) {
        const float4* row = reinterpret_cast<const float4*>(dL_dfeatures + i * ((int64_t)f.num_levels * f.feat_dim));
        for (int e = sub; e < f.num_levels * f.feat_dim / 4; e += 8) {
            const float4 v = row[e];
            active |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
        }
        active |= __shfl_xor(active, 1, 64);
        active |= __shfl_xor(active, 2, 64);
        active |= __shfl_xor(active, 4, 64);
    }
    if (sub != 0) return;
    if (!active) {
#pragma unroll
        for (int pl = 0; pl < 6; pl++) ws.key[(size_t)pl * n + i] = 0xFFFFFFFFu;
        return;
    }
    const float zero_time = 0.f;       // time == nullptr (batched views): the time families are sorted along their spatial coordinate only
    const HexPoint p = hex_point(f.aabb, xyz + i * xyz_stride, time ? time + i * time_stride : &zero_time);
    // batched views: the fourth component carries the point's view bits to the sorted order instead of a time (the walks take the time from
    // the view)
    ws.coords[i] = make_float4(p.c[0], p.c[1], p.c[2], time ? p.c[3] : __uint_as_float(vbits));
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
// (returns the largest |dL/dsample| this lane wrote: the ordered mode's scale, hexord_publish_max)
template <int C>
__device__ __forceinline__ float hexsort_phase1_body(const gsr_hexplane_field& f, const HexSortWs& ws, const int64_t n, const float* __restrict__ xyz,
                                                     const int64_t xyz_stride, const float* __restrict__ time, const int64_t time_stride,
                                                     const float* __restrict__ dL_dfeatures, float* __restrict__ dL_dxyz)
{
    const int ch = threadIdx.x % C;
    const int64_t i = (int64_t)blockIdx.x * (HEX_BLOCK / C) + threadIdx.x / C;
    if (i >= n) return 0.f;
    if (ws.rank[i] < 0) {                                         // zero cotangent: nothing to gather, nothing to hand over
        if (dL_dxyz && ch < 3) dL_dxyz[3 * i + ch] = 0.f;
        return 0.f;
    }
    float mx = 0.f;
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
            mx = hexord_track(mx, gs);
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
    return mx;
}

template <int C>
__global__ void __launch_bounds__(HEX_BLOCK)
hexsort_phase1_kernel(const gsr_hexplane_field f, const HexSortWs ws, const int64_t n, const float* __restrict__ xyz, const int64_t xyz_stride,
                      const float* __restrict__ time, const int64_t time_stride, const float* __restrict__ dL_dfeatures,
                      float* __restrict__ dL_dxyz, const int ordered)
{
    const float mx = hexsort_phase1_body<C>(f, ws, n, xyz, xyz_stride, time, time_stride, dL_dfeatures, dL_dxyz);
    if (ordered) hexord_publish_max(mx, ws.header);
}

// ---- phase 2: run-length accumulation along the sorted order --------------------------------------------------------------------------
// One group of C lanes (one channel each) walks `cnt` consecutive sorted points of plane family `pl`: coordinates `sc[k]` (with the time
// replaced by `t_fixed` when `fix_t`: the batched-views caller sorts once for all views), dL/dsample rows `gsrow[k * L * C + l * C]`.
template <int C, int LMAX, bool ORD>
__device__ __forceinline__ void hexsort_phase2_walk(const gsr_hexplane_field& f, const int pl, const float4* __restrict__ sc,
                                                    const float* __restrict__ gsrow, const int cnt, const bool fix_t, const float t_fixed,
                                                    const HexOrd& o, const float S)
{
    using Acc = std::conditional_t<ORD, double, float>;           // ORD: integer-valued doubles (see HexOrd)
    const int ch = threadIdx.x % C;
    const int c0 = hex_c0(pl), c1 = hex_c1(pl);
    const int L = f.num_levels;
    Acc acc[LMAX][4];
    int cx[LMAX], cy[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; l++) { cx[l] = cy[l] = -1; acc[l][0] = acc[l][1] = acc[l][2] = acc[l][3] = 0; }

    auto flush = [&](int l) {                                     // four texel-wide atomics for the cell (cx, cy) of level l
        const gsr_hexplane_level& Lv = f.levels[l];
        float* gp = Lv.grad_planes[pl];
        const int W = Lv.res[c0], H = Lv.res[c1];
        if (gp && cx[l] >= 0) {
            const size_t at = ((size_t)cy[l] * W + cx[l]) * C + ch;
            const bool x1 = cx[l] + 1 < W, y1 = cy[l] + 1 < H;     // safe_add_2d: corners outside the plane receive nothing
            if constexpr (ORD) {
                unsigned long long* t = o.acc + o.off[l][pl] + at;
                hexord_add(t, acc[l][0]);
                if (x1) hexord_add(t + C, acc[l][1]);
                if (y1) hexord_add(t + (size_t)W * C, acc[l][2]);
                if (x1 && y1) hexord_add(t + (size_t)(W + 1) * C, acc[l][3]);
            } else {
                float* t = gp + at;
                if (acc[l][0] != 0.f) unsafeAtomicAdd(t, acc[l][0]);
                if (x1 && acc[l][1] != 0.f) unsafeAtomicAdd(t + C, acc[l][1]);
                if (y1 && acc[l][2] != 0.f) unsafeAtomicAdd(t + (size_t)W * C, acc[l][2]);
                if (x1 && y1 && acc[l][3] != 0.f) unsafeAtomicAdd(t + (size_t)(W + 1) * C, acc[l][3]);
            }
        }
        acc[l][0] = acc[l][1] = acc[l][2] = acc[l][3] = 0;
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
        const float c[4] = {cc.x, cc.y, cc.z, fix_t ? t_fixed : cc.w};
#pragma unroll
        for (int l = 0; l < LMAX; l++) {
            if (l < L) {
                const gsr_hexplane_level& Lv = f.levels[l];
                const HexAxis X = hex_axis(c[c0], Lv.res[c0]), Y = hex_axis(c[c1], Lv.res[c1]);
                if (X.i0 != cx[l] || Y.i0 != cy[l]) {
                    flush(l);
                    cx[l] = X.i0;
                    cy[l] = Y.i0;
                }
                if constexpr (ORD) {                              // every product rounded once to the call's quantum, then integers only
                    const float gs = g_cur[l] * S;
                    acc[l][0] += (double)rintf(gs * (X.w0 * Y.w0));
                    acc[l][1] += (double)rintf(gs * (X.w1 * Y.w0));
                    acc[l][2] += (double)rintf(gs * (X.w0 * Y.w1));
                    acc[l][3] += (double)rintf(gs * (X.w1 * Y.w1));
                } else {
                    const float gs = g_cur[l];
                    acc[l][0] = fmaf(gs, X.w0 * Y.w0, acc[l][0]);
                    acc[l][1] = fmaf(gs, X.w1 * Y.w0, acc[l][1]);
                    acc[l][2] = fmaf(gs, X.w0 * Y.w1, acc[l][2]);
                    acc[l][3] = fmaf(gs, X.w1 * Y.w1, acc[l][3]);
                }
            }
        }
    }
#pragma unroll
    for (int l = 0; l < LMAX; l++) {
        if (l < L) flush(l);
    }
}

template <int C, int LMAX, bool ORD>
__global__ void __launch_bounds__(256)
hexsort_phase2_kernel(const gsr_hexplane_field f, const HexSortWs ws, const int64_t n, const HexOrd o)
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
    const float S = ORD ? hexord_scale(ws.header[1], o.budget) : 0.f;
    hexsort_phase2_walk<C, LMAX, ORD>(f, pl, ws.scoords + (size_t)pl * na + first, ws.gs + ((size_t)pl * na + first) * f.num_levels * C + ch, cnt, false, 0.f,
                                      o, S);
}

// ---- ordered mode, last pass: the fixed-point sums -> the float gradient planes (ACCUMULATED into, like the atomics they replace) -----------
// grid (elements / 512, level x 6 + plane); a thread converts two neighbouring elements. Every element has one owner: no atomics.
__global__ void __launch_bounds__(256)
hexord_convert_kernel(const gsr_hexplane_field f, const HexOrd o, const uint32_t* __restrict__ header, const int plane_mask)
{
    const int l = blockIdx.y / 6, pl = blockIdx.y % 6;
    if (!((plane_mask >> pl) & 1)) return;
    const gsr_hexplane_level& Lv = f.levels[l];
    float* gp = Lv.grad_planes[pl];
    const size_t count = (size_t)Lv.res[hex_c0(pl)] * Lv.res[hex_c1(pl)] * f.feat_dim;      // even: the channel counts are multiples of 8
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (!gp || e >= count) return;
    if (header[2]) { gp[e] = gp[e + 1] = __uint_as_float(0x7fc00000u); return; }      // a non-finite dL/dsample somewhere in the call (hexord_publish_max)
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(o.acc + o.off[l][pl] + e);
    if ((v.x | v.y) == 0ull) return;
    const double inv = hexord_inverse(header[1], o.budget);
    if (v.x) gp[e] += (float)((double)(long long)v.x * inv);
    if (v.y) gp[e + 1] += (float)((double)(long long)v.y * inv);
}

// ---- which rows of a batch's cotangent are not zero (gsr_row_mask) ---------------------------------------------------------------------
// g [V][n][width]: bit v of view_mask[i] = some element of row (v, i) is non-zero; rows[] = the flat indices v n + i of those rows in
// ascending order (deterministic: per-(view, block) counts, one scan, ranks inside the block). A mapping iteration's Gaussians mostly
// receive NO gradient from a given view (BASELINE config #3: 63 % of the 8 x 500k rows -- outside the frustum, or behind a saturated pixel):
// the deformation MLP's backward then runs over the listed rows only, and the field's backward skips the others by the bit.
constexpr int ROWMASK_BLOCK = 256;
__device__ __forceinline__ bool rowmask_nonzero(const float* __restrict__ r, int width)
{
    bool nz = false;
    for (int k = 0; k < width; k++) nz |= r[k] != 0.f;
    return nz;
}
__global__ void __launch_bounds__(ROWMASK_BLOCK)
rowmask_count_kernel(const int V, const int64_t n, const int width, const float* __restrict__ g, uint32_t* __restrict__ view_mask,
                     uint32_t* __restrict__ block_counts /*[V][gridDim.x]*/)
{
    __shared__ uint32_t s_cnt[ROWMASK_BLOCK / 64];
    const int64_t i = (int64_t)blockIdx.x * ROWMASK_BLOCK + threadIdx.x;
    uint32_t bits = 0;
    for (int v = 0; v < V; v++) {
        const bool nz = i < n && rowmask_nonzero(g + ((size_t)v * n + i) * width, width);
        bits |= (nz ? 1u : 0u) << v;
        const unsigned long long b = __ballot(nz);
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(b);
        __syncthreads();
        if (threadIdx.x == 0) block_counts[(size_t)v * gridDim.x + blockIdx.x] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        __syncthreads();
    }
    if (i < n) view_mask[i] = bits;
}
__global__ void __launch_bounds__(1024)
rowmask_scan_kernel(uint32_t* __restrict__ block_counts, const int count, int32_t* __restrict__ n_rows)
{
    __shared__ uint32_t s_tmp[17];
    const uint32_t total = block_exclusive_scan_1024(count, [&](int b) { return block_counts[b]; },
                                                     [&](int b, uint32_t ex, uint32_t) { block_counts[b] = ex; }, s_tmp);
    if (threadIdx.x == 0) n_rows[0] = (int32_t)total;
}
__global__ void __launch_bounds__(ROWMASK_BLOCK)
rowmask_list_kernel(const int V, const int64_t n, const uint32_t* __restrict__ view_mask, const uint32_t* __restrict__ block_offsets,
                    int32_t* __restrict__ rows)
{
    __shared__ uint32_t s_cnt[ROWMASK_BLOCK / 64];
    const int64_t i = (int64_t)blockIdx.x * ROWMASK_BLOCK + threadIdx.x;
    const uint32_t bits = i < n ? view_mask[i] : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int v = 0; v < V; v++) {
        const bool nz = (bits >> v) & 1u;
        const unsigned long long b = __ballot(nz);
        if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += s_cnt[w];
        if (nz) rows[block_offsets[(size_t)v * gridDim.x + blockIdx.x] + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = (int32_t)((int64_t)v * n + i);
        __syncthreads();
    }
}

// ---- the views of one mapping iteration: one sort, one spatial scatter (gsr_hexplane_backward_views) ----------------------------------
// Every keyframe of an iteration evaluates the field at the same positions with its own time. The cells a point falls into therefore do
// not depend on the view: ONE counting sort serves all of them (the time families are keyed by their spatial coordinate alone -- a view's
// time selects the same two rows of cells for every point). Phase 1 handles a point for all views at once: the three spatial planes'
// corners are gathered once per level, dL/dsample of a SPATIAL plane is summed over the views in registers (view order) before it is
// written to the point's sorted slot -- one spatial scatter per iteration instead of one per view, and an eighth of the atomics --, and only
// the time planes' dL/dsample is staged per view. Phase 2 is the same run-length walk over 3 + 3 V streams.
struct HexViewsWs {
    float* gs_sp;     // [3][n][L][C]     dL/dsample of the xy, xz, yz planes, summed over the views, in the family's sorted order
    float* gs_t;      // [3][V][n][L][C]  dL/dsample of the xt, yt, zt planes per view, in the order of the x / y / z family
};

template <int C>
__device__ __forceinline__ float hexsort_phase1_views_body(const gsr_hexplane_field& f, const HexSortWs& ws, const HexViewsWs& vw, const HexTimes& tv,
                                                           const int64_t n, const float* __restrict__ xyz, const int64_t xyz_stride,
                                                           const float* __restrict__ dL_dfeatures, float* __restrict__ dL_dxyz)
{
    const int ch = threadIdx.x % C;
    const int64_t i = (int64_t)blockIdx.x * (HEX_BLOCK / C) + threadIdx.x / C;
    if (i >= n) return 0.f;
    if (ws.rank[i] < 0) {                                         // no view's cotangent reaches this point: nothing to gather, nothing to hand over
        if (dL_dxyz && ch < 3) dL_dxyz[3 * i + ch] = 0.f;
        return 0.f;
    }
    float mx = 0.f;
    const uint32_t vbits = __float_as_uint(ws.coords[i].w);       // views whose cotangent row of this point is not zero
    const float zero_time = 0.f;
    const HexPoint p = hex_point(f.aabb, xyz + i * xyz_stride, &zero_time);
    const int L = f.num_levels, V = tv.V;
    const size_t row = (size_t)L * C;
    const size_t na = ws.header[0] / 6;                           // points that were sorted: family pl owns the slots [pl na, (pl + 1) na)
    constexpr int SP[3] = {0, 1, 3}, TP[3] = {2, 4, 5};           // plane numbers of the spatial / time families
    size_t sp_slot[3], t_slot[3];                                 // this lane's element of the point's row in each family's sorted order
#pragma unroll
    for (int j = 0; j < 3; j++) {
        sp_slot[j] = ((size_t)j * n + ((size_t)ws.rank[(size_t)SP[j] * n + i] - (size_t)SP[j] * na)) * row + ch;
        t_slot[j] = ((size_t)j * V * n + ((size_t)ws.rank[(size_t)TP[j] * n + i] - (size_t)TP[j] * na)) * row + ch;
    }
    const float* gout = dL_dfeatures + (size_t)i * row + ch;
    float gc[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < L; l++) {
        const gsr_hexplane_level& Lv = f.levels[l];
        HexAxis ax[3];
#pragma unroll
        for (int k = 0; k < 3; k++) ax[k] = hex_axis(p.c[k], Lv.res[k]);
        float cs[3][4], ss[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int c0 = hex_c0(SP[j]), c1 = hex_c1(SP[j]);
            const int W = Lv.res[c0];
            const HexAxis& X = ax[c0];
            const HexAxis& Y = ax[c1];
            const int x1 = X.has1 ? X.i0 + 1 : X.i0, y1 = Y.has1 ? Y.i0 + 1 : Y.i0;
            const float* plane = Lv.planes[SP[j]];
            cs[j][0] = plane[((size_t)Y.i0 * W + X.i0) * C + ch];
            cs[j][1] = plane[((size_t)Y.i0 * W + x1) * C + ch];
            cs[j][2] = plane[((size_t)y1 * W + X.i0) * C + ch];
            cs[j][3] = plane[((size_t)y1 * W + x1) * C + ch];
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const HexAxis& X = ax[hex_c0(SP[j])];
            const HexAxis& Y = ax[hex_c1(SP[j])];
            const float wx1 = X.has1 ? X.w1 : 0.f, wy1 = Y.has1 ? Y.w1 : 0.f;
            float v = cs[j][0] * (X.w0 * Y.w0);
            v = fmaf(cs[j][1], wx1 * Y.w0, v);
            v = fmaf(cs[j][2], X.w0 * wy1, v);
            ss[j] = fmaf(cs[j][3], wx1 * wy1, v);
        }
        float Gs[3] = {0.f, 0.f, 0.f};
        const int Wt = Lv.res[3];
        for (int v = 0; v < V; v++) {
            if (!((vbits >> v) & 1u)) continue;                   // a zero row: nothing to add, and phase 2 skips the slot by the same bit
            const HexAxis T = hex_axis(tv.t[v], Wt);
            const int t1 = T.has1 ? T.i0 + 1 : T.i0;
            const float wt1 = T.has1 ? T.w1 : 0.f;
            float ct[3][4], st[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {                         // plane (k, t): width res[k], height res[t]
                const int W = Lv.res[j];
                const HexAxis& X = ax[j];
                const int x1 = X.has1 ? X.i0 + 1 : X.i0;
                const float* plane = Lv.planes[TP[j]];
                ct[j][0] = plane[((size_t)T.i0 * W + X.i0) * C + ch];
                ct[j][1] = plane[((size_t)T.i0 * W + x1) * C + ch];
                ct[j][2] = plane[((size_t)t1 * W + X.i0) * C + ch];
                ct[j][3] = plane[((size_t)t1 * W + x1) * C + ch];
            }
            const float g = gout[(size_t)v * n * row + (size_t)l * C];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const HexAxis& X = ax[j];
                const float wx1 = X.has1 ? X.w1 : 0.f;
                float s_ = ct[j][0] * (X.w0 * T.w0);
                s_ = fmaf(ct[j][1], wx1 * T.w0, s_);
                s_ = fmaf(ct[j][2], X.w0 * wt1, s_);
                st[j] = fmaf(ct[j][3], wx1 * wt1, s_);
            }
            // plane order of the product (hexplane.py:93-103): xy, xz, xt, yz, yt, zt
            const float s[6] = {ss[0], ss[1], st[0], ss[2], st[1], st[2]};
            float suffix[6];
            suffix[5] = 1.f;
#pragma unroll
            for (int pl = 4; pl >= 0; pl--) suffix[pl] = suffix[pl + 1] * s[pl + 1];
            float prefix = g, gs[6];
#pragma unroll
            for (int pl = 0; pl < 6; pl++) { gs[pl] = prefix * suffix[pl]; prefix *= s[pl]; }
            Gs[0] += gs[0]; Gs[1] += gs[1]; Gs[2] += gs[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const float gt = gs[TP[j]];
                __builtin_nontemporal_store(gt, &vw.gs_t[t_slot[j] + (size_t)v * n * row + (size_t)l * C]);
                mx = hexord_track(mx, gt);
                if (dL_dxyz) {                                    // the time itself receives no gradient
                    const HexAxis& X = ax[j];
                    const float nw = ct[j][0] * gt, ne = X.has1 ? ct[j][1] * gt : 0.f, sw = T.has1 ? ct[j][2] * gt : 0.f;
                    const float se = X.has1 && T.has1 ? ct[j][3] * gt : 0.f;
                    gc[j] += ((ne - nw) * T.w0 + (se - sw) * T.w1) * X.dmult;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int c0 = hex_c0(SP[j]), c1 = hex_c1(SP[j]);
            const HexAxis& X = ax[c0];
            const HexAxis& Y = ax[c1];
            const float gsp = Gs[j];
            __builtin_nontemporal_store(gsp, &vw.gs_sp[sp_slot[j] + (size_t)l * C]);
            mx = hexord_track(mx, gsp);
            if (dL_dxyz) {
                const float nw = cs[j][0] * gsp, ne = X.has1 ? cs[j][1] * gsp : 0.f, sw = Y.has1 ? cs[j][2] * gsp : 0.f;
                const float se = X.has1 && Y.has1 ? cs[j][3] * gsp : 0.f;
                gc[c0] += ((ne - nw) * Y.w0 + (se - sw) * Y.w1) * X.dmult;
                gc[c1] += ((sw - nw) * X.w0 + (se - ne) * X.w1) * Y.dmult;
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
    return mx;
}

template <int C>
__global__ void __launch_bounds__(HEX_BLOCK)
hexsort_phase1_views_kernel(const gsr_hexplane_field f, const HexSortWs ws, const HexViewsWs vw, const HexTimes tv, const int64_t n,
                            const float* __restrict__ xyz, const int64_t xyz_stride, const float* __restrict__ dL_dfeatures,
                            float* __restrict__ dL_dxyz, const int ordered)
{
    const float mx = hexsort_phase1_views_body<C>(f, ws, vw, tv, n, xyz, xyz_stride, dL_dfeatures, dL_dxyz);
    if (ordered) hexord_publish_max(mx, ws.header);
}

// ---- phase 2 of the TIME families, all views ----------------------------------------------------------------------------------------
// The weights along a time family's spatial coordinate are the same for every view, and a view's time is the same for every point: per
// (view, level) the sums  sum gs * w0 -> column cx,  sum gs * w1 -> column cx + 1  over the points of a spatial cell are all there is; the two
// time rows' weights are applied once, at the very end. Because the points are sorted along that coordinate, a block that walks
// HEXT_ROUNDS * GROUPS consecutive chunks stays inside a few columns of every level: it sums into an LDS window of HEXT_WIN columns per
// (view, level) -- a run's two sums with ds_add_f32, a few per chunk -- and issues the global float atomics ONCE per column it touched.
// Measured first (config #3, 500k points x 8 views): the generic walk with its 4 atomics per run spent 6.4 ms in these streams at 1 TB/s --
// a view's rows of one time are 2 x 960 texels that every wave in flight was hitting at once, and on this part a wave's loads queue behind
// its outstanding atomics.
constexpr int HEXT_VB = 4;        // views per block
constexpr int HEXT_ROUNDS = 4;    // chunks per group
constexpr int HEXT_WIN = 16;      // columns of the LDS window per level (runs outside it go to global memory directly)
constexpr int HEXT_WIN_ORD = 8;   // ... of the ordered mode's 64-bit window (the same LDS footprint; a block's 2 048 sorted points span 3-4 columns of the finest level at config #3)

template <int C, int LMAX, bool ORD>
__global__ void __launch_bounds__(256)
hexsort_phase2_time_kernel(const gsr_hexplane_field f, const HexSortWs ws, const HexViewsWs vw, const HexTimes tv, const int64_t n, const HexOrd o)
{
    constexpr int GROUPS = 256 / C, VB = HEXT_VB, WIN = ORD ? HEXT_WIN_ORD : HEXT_WIN;
    using Win = std::conditional_t<ORD, unsigned long long, float>;
    using Acc = std::conditional_t<ORD, double, float>;
    __shared__ Win s_acc[VB][LMAX][WIN][C];
    const int ch = threadIdx.x % C, grp = threadIdx.x / C;
    const int j = blockIdx.y % 3, v0 = (blockIdx.y / 3) * VB;     // family (x, y, z) and first view of this block
    const int nv = min(VB, tv.V - v0);
    const int pl = j == 0 ? 2 : 3 + j;                            // planes (x,t) = 2, (y,t) = 4, (z,t) = 5
    const int L = f.num_levels;
    const size_t row = (size_t)L * C, view_stride = (size_t)n * row;
    const int64_t na = ws.header[0] / 6;                          // points that were sorted (some view's cotangent reaches them); the grid covers n
    const int64_t chunks = (na + HEXSORT_CHUNK - 1) / HEXSORT_CHUNK;
    const int64_t chunk0 = (int64_t)blockIdx.x * (GROUPS * HEXT_ROUNDS);
    if (chunk0 >= chunks) return;
    const float S = ORD ? hexord_scale(ws.header[1], o.budget) : 0.f;
    const float4* __restrict__ sc = ws.scoords + (size_t)pl * na;
    const float* __restrict__ gs = vw.gs_t + ((size_t)j * tv.V + v0) * view_stride + ch;
    auto coord = [&](const float4& c) { return j == 0 ? c.x : (j == 1 ? c.y : c.z); };
    for (int e = threadIdx.x; e < VB * LMAX * WIN * C; e += 256) (&s_acc[0][0][0][0])[e] = 0;
    // ordered mode: element of (view v0 + v, level l, column col) in the family's column sums
    auto ord_at = [&](int v, int l, int col) { return o.acc_t + ((size_t)o.tbase[j] + (size_t)(v0 + v) * o.tcols[j] + o.tcol[j][l] + col) * C + ch; };
    int col0[LMAX];                                               // first column of the window: the cell of the block's first point
    {
        const float c_first = coord(sc[chunk0 * HEXSORT_CHUNK]);
#pragma unroll
        for (int l = 0; l < LMAX; l++) col0[l] = l < L ? hex_axis(c_first, f.levels[l].res[j]).i0 : 0;
    }
    __syncthreads();
    Acc acc[VB][LMAX][2];
    int cx[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; l++) {
        cx[l] = -1;
#pragma unroll
        for (int v = 0; v < VB; v++) acc[v][l][0] = acc[v][l][1] = 0;
    }
    auto flush = [&](int l) {                                     // the run of level l ends: its sums go to the window (or, outside it, to memory)
        if (cx[l] >= 0) {
            const gsr_hexplane_level& Lv = f.levels[l];
            const int W = Lv.res[j], rel = cx[l] - col0[l];
            const bool x1 = cx[l] + 1 < W;                          // safe_add_2d: a column outside the plane receives nothing
            if (rel >= 0 && rel + 1 < WIN) {
#pragma unroll
                for (int v = 0; v < VB; v++) {
                    if constexpr (ORD) {
                        hexord_add(&s_acc[v][l][rel][ch], acc[v][l][0]);
                        if (x1) hexord_add(&s_acc[v][l][rel + 1][ch], acc[v][l][1]);
                    } else {
                        if (acc[v][l][0] != 0.f) atomicAdd(&s_acc[v][l][rel][ch], acc[v][l][0]);
                        if (x1 && acc[v][l][1] != 0.f) atomicAdd(&s_acc[v][l][rel + 1][ch], acc[v][l][1]);
                    }
                }
            } else {
                float* gp = Lv.grad_planes[pl];
#pragma unroll
                for (int v = 0; v < VB; v++) {
                    if (gp && v < nv) {
                        if constexpr (ORD) {
                            hexord_add(ord_at(v, l, cx[l]), acc[v][l][0]);
                            if (x1) hexord_add(ord_at(v, l, cx[l] + 1), acc[v][l][1]);
                        } else {
                            const HexAxis T = hex_axis(tv.t[v0 + v], Lv.res[3]);
                            float* t = gp + ((size_t)T.i0 * W + cx[l]) * C + ch;
                            const float a0 = acc[v][l][0], a1 = acc[v][l][1];
                            if (a0 != 0.f) unsafeAtomicAdd(t, a0 * T.w0);
                            if (x1 && a1 != 0.f) unsafeAtomicAdd(t + C, a1 * T.w0);
                            if (T.has1 && a0 != 0.f) unsafeAtomicAdd(t + (size_t)W * C, a0 * T.w1);
                            if (T.has1 && x1 && a1 != 0.f) unsafeAtomicAdd(t + (size_t)(W + 1) * C, a1 * T.w1);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VB; v++) acc[v][l][0] = acc[v][l][1] = 0;
    };
    for (int r = 0; r < HEXT_ROUNDS; r++) {
        const int64_t chunk = chunk0 + (int64_t)r * GROUPS + grp;  // the groups of a block walk neighbouring chunks at the same time
        if (chunk >= chunks) break;
        const int64_t first = chunk * HEXSORT_CHUNK;
        const int cnt = (int)min((int64_t)HEXSORT_CHUNK, na - first);
        const float4* __restrict__ scc = sc + first;
        const float* __restrict__ gsrow = gs + (size_t)first * row;
        // the loads of point k + 1 (one coordinate + view bits, VB * L rows) are in flight while point k is processed; the rows of views whose
        // bit is clear were never written (phase 1 skipped them) and are not read
        float4 cc0 = scc[0];
        float c_next = coord(cc0);
        uint32_t bits = __float_as_uint(cc0.w) >> v0;
        float g_next[VB][LMAX];
#pragma unroll
        for (int v = 0; v < VB; v++) {
#pragma unroll
            for (int l = 0; l < LMAX; l++) g_next[v][l] = (l < L && v < nv && ((bits >> v) & 1u)) ? gsrow[(size_t)v * view_stride + (size_t)l * C] : 0.f;
        }
        for (int k = 0; k < cnt; k++) {
            const float c = c_next;
            float g_cur[VB][LMAX];
#pragma unroll
            for (int v = 0; v < VB; v++) {
#pragma unroll
                for (int l = 0; l < LMAX; l++) g_cur[v][l] = g_next[v][l];
            }
            const int kn = min(k + 1, cnt - 1);
            const float4 ccn = scc[kn];
            c_next = coord(ccn);
            bits = __float_as_uint(ccn.w) >> v0;
#pragma unroll
            for (int v = 0; v < VB; v++) {
#pragma unroll
                for (int l = 0; l < LMAX; l++)
                    g_next[v][l] = (l < L && v < nv && ((bits >> v) & 1u)) ? gsrow[(size_t)v * view_stride + ((size_t)kn * L + l) * C] : 0.f;
            }
#pragma unroll
            for (int l = 0; l < LMAX; l++) {
                if (l < L) {
                    const HexAxis X = hex_axis(c, f.levels[l].res[j]);
                    if (X.i0 != cx[l]) {
                        flush(l);
                        cx[l] = X.i0;
                    }
#pragma unroll
                    for (int v = 0; v < VB; v++) {
                        if constexpr (ORD) {
                            const float g = g_cur[v][l] * S;
                            acc[v][l][0] += (double)rintf(g * X.w0);
                            acc[v][l][1] += (double)rintf(g * X.w1);
                        } else {
                            acc[v][l][0] = fmaf(g_cur[v][l], X.w0, acc[v][l][0]);
                            acc[v][l][1] = fmaf(g_cur[v][l], X.w1, acc[v][l][1]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int l = 0; l < LMAX; l++) {
            if (l < L) { flush(l); cx[l] = -1; }
        }
    }
    __syncthreads();
    // the window -> (ordered) the family's column sums / (float) the two time rows of every view: once per (view, level, column) the block touched
    for (int e = grp; e < VB * LMAX * WIN; e += GROUPS) {
        const int v = e / (LMAX * WIN), l = (e / WIN) % LMAX, rel = e % WIN;
        if (v >= nv || l >= L) continue;
        const Win a = s_acc[v][l][rel][ch];
        const gsr_hexplane_level& Lv = f.levels[l];
        float* gp = Lv.grad_planes[pl];
        const int W = Lv.res[j], col = col0[l] + rel;
        if (a == 0 || !gp || col >= W) continue;
        if constexpr (ORD) {
            atomicAdd(ord_at(v, l, col), a);
        } else {
            const HexAxis T = hex_axis(tv.t[v0 + v], Lv.res[3]);
            float* t = gp + ((size_t)T.i0 * W + col) * C + ch;
            unsafeAtomicAdd(t, a * T.w0);
            if (T.has1) unsafeAtomicAdd(t + (size_t)W * C, a * T.w1);
        }
    }
}

// ordered mode: the time families' column sums -> the planes. grid (columns x C / 256, family x levels + level); a thread owns one (column,
// channel) of one level of one family and adds the views in their order, each to the view's two time rows with its weights.
__global__ void __launch_bounds__(256)
hexord_time_convert_kernel(const gsr_hexplane_field f, const HexTimes tv, const HexOrd o, const uint32_t* __restrict__ header)
{
    const int L = f.num_levels, C = f.feat_dim;
    const int j = blockIdx.y / L, l = blockIdx.y % L;
    const int pl = j == 0 ? 2 : 3 + j;
    const gsr_hexplane_level& Lv = f.levels[l];
    float* gp = Lv.grad_planes[pl];
    const int W = Lv.res[j];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (!gp || e >= W * C) return;
    const int col = e / C, ch = e % C;
    if (header[2]) {                                              // a non-finite dL/dsample somewhere in the call: every row of the plane
        for (int r = 0; r < Lv.res[3]; r++) gp[((size_t)r * W + col) * C + ch] = __uint_as_float(0x7fc00000u);
        return;
    }
    const double inv = hexord_inverse(header[1], o.budget);
    for (int v = 0; v < tv.V; v++) {
        const unsigned long long a = o.acc_t[((size_t)o.tbase[j] + (size_t)v * o.tcols[j] + o.tcol[j][l] + col) * C + ch];
        if (a == 0ull) continue;
        const float val = (float)((double)(long long)a * inv);
        const HexAxis T = hex_axis(tv.t[v], Lv.res[3]);
        float* t = gp + ((size_t)T.i0 * W + col) * C + ch;
        *t += val * T.w0;
        if (T.has1) t[(size_t)W * C] += val * T.w1;
    }
}

// the spatial families (planes 0, 1, 3), whose dL/dsample phase 1 summed over the views: the generic walk, one stream per family
template <int C, int LMAX, bool ORD>
__global__ void __launch_bounds__(256)
hexsort_phase2_views_kernel(const gsr_hexplane_field f, const HexSortWs ws, const HexViewsWs vw, const int64_t n, const HexOrd o)
{
    constexpr int GROUPS = 256 / C;
    constexpr int GPW = C >= 64 ? 1 : 64 / C;                     // groups per wave: they take the same stream (one code path per wave) ...
    const int ch = threadIdx.x % C;
    const int64_t na = ws.header[0] / 6;                          // points that were sorted; the grid covers n
    const int64_t chunks = (na + HEXSORT_CHUNK - 1) / HEXSORT_CHUNK;
    const int64_t gid = (int64_t)blockIdx.x * GROUPS + threadIdx.x / C;
    const int64_t wave_chunks = (chunks + GPW - 1) / GPW;
    const int64_t w = gid / GPW;
    if (w >= 3 * wave_chunks) return;
    // ... and consecutive waves different streams of the same chunks: the waves in flight at one moment spread their atomics over the planes
    const int st = (int)(w % 3);
    const int64_t chunk = (w / 3) * GPW + gid % GPW;
    if (chunk >= chunks) return;
    const int64_t first = chunk * HEXSORT_CHUNK;
    const int cnt = (int)min((int64_t)HEXSORT_CHUNK, na - first);
    const size_t row = (size_t)f.num_levels * C;
    const int pl = st == 2 ? 3 : st;
    const float S = ORD ? hexord_scale(ws.header[1], o.budget) : 0.f;
    hexsort_phase2_walk<C, LMAX, ORD>(f, pl, ws.scoords + (size_t)pl * na + first, vw.gs_sp + ((size_t)st * n + first) * row + ch, cnt, false, 0.f, o, S);
}

}  // namespace gsr
