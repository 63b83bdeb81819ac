// gs_knn.h -- exact 3-nearest-neighbour mean squared distance on a uniform spatial-hash grid (gfx950).
// Replaces SimpleKNN::knn (submodules/simple-knn/simple_knn.cu:185-221; KNN = submodules/simple-knn).
#pragma once
#include <float.h>
#include "gs_forward.h"

namespace gsr {

struct KnnGrid {          // lives in device memory; written by knn_setup_kernel
    uint32_t bbox[6];     // order-preserving uint encodings: min xyz (init ~0), max xyz (init 0)
    float origin[3];
    float cell;           // cell edge length (> 0)
    float inv_cell;
    int dims[3];
    int ncells;
};

__device__ __forceinline__ uint32_t float_to_ordered(float f)
{
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t u)
{
    return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

__global__ void __launch_bounds__(256) knn_bbox_kernel(int P, const float* __restrict__ pts, KnnGrid* g)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; a++) { const float v = pts[3 * (size_t)i + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64)); }
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { atomicMin(&g->bbox[a], float_to_ordered(mn[a])); atomicMax(&g->bbox[3 + a], float_to_ordered(mx[a])); }
    }
}

// One thread: pick the grid. Target ~2 points per cell, never more than max_cells cells in total.
__global__ void knn_setup_kernel(int P, int max_cells, KnnGrid* g)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lo[3], ext[3];
    for (int a = 0; a < 3; a++) {
        lo[a] = ordered_to_float(g->bbox[a]);
        const float hi = ordered_to_float(g->bbox[3 + a]);
        ext[a] = fmaxf(hi - lo[a], 0.f);
        g->origin[a] = lo[a];
    }
    const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
    float cell = emax > 0.f ? emax : 1.f;
    if (emax > 0.f) {
        // volume of the occupied box, ignoring degenerate axes
        double vol = 1.0; int nd = 0;
        for (int a = 0; a < 3; a++) if (ext[a] > 1e-6f * emax) { vol *= ext[a]; nd++; }
        const double target = fmax(1.0, (double)P / 2.0);
        cell = (float)pow(vol / target, 1.0 / (double)(nd > 0 ? nd : 1));
        if (!(cell > 0.f)) cell = emax;
    }
    int d[3];
    for (int it = 0; it < 64; it++) {
        long long tot = 1;
        for (int a = 0; a < 3; a++) { d[a] = (int)fminf(floorf(ext[a] / cell) + 1.f, 2048.f); if (d[a] < 1) d[a] = 1; tot *= d[a]; }
        if (tot <= (long long)max_cells) break;
        cell *= 1.26f;
    }
    for (int a = 0; a < 3; a++) g->dims[a] = d[a];
    g->cell = cell; g->inv_cell = 1.0f / cell; g->ncells = d[0] * d[1] * d[2];
}

__device__ __forceinline__ void knn_cell_of(const KnnGrid* g, float x, float y, float z, int& cx, int& cy, int& cz)
{
    cx = min(g->dims[0] - 1, max(0, (int)((x - g->origin[0]) * g->inv_cell)));
    cy = min(g->dims[1] - 1, max(0, (int)((y - g->origin[1]) * g->inv_cell)));
    cz = min(g->dims[2] - 1, max(0, (int)((z - g->origin[2]) * g->inv_cell)));
}

__global__ void __launch_bounds__(256) knn_count_kernel(int P, const float* __restrict__ pts, const KnnGrid* g, uint32_t* cell_of,
                                                        uint32_t* cell_count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    int cx, cy, cz;
    knn_cell_of(g, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], cx, cy, cz);
    const uint32_t c = (uint32_t)((cz * g->dims[1] + cy) * g->dims[0] + cx);
    cell_of[i] = c;
    atomicAdd(&cell_count[c], 1u);
}

__global__ void __launch_bounds__(1024) knn_scan_kernel(const KnnGrid* g, const uint32_t* cell_count, uint32_t* cell_start, uint32_t* cursor)
{
    __shared__ uint32_t s_tmp[17];
    const int n = g->ncells;
    const uint32_t total = block_exclusive_scan_1024(
        n, [&](int i) { return cell_count[i]; }, [&](int i, uint32_t excl, uint32_t) { cell_start[i] = excl; cursor[i] = excl; }, s_tmp);
    if (threadIdx.x == 0) cell_start[n] = total;
}

__global__ void __launch_bounds__(256) knn_scatter_kernel(int P, const float* __restrict__ pts, const uint32_t* cell_of, uint32_t* cursor,
                                                          float4* sorted_pts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t pos = atomicAdd(&cursor[cell_of[i]], 1u);
    sorted_pts[pos] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __int_as_float(i));
}

// KNN/simple_knn.cu:131-145
__device__ __forceinline__ void knn_update3(float dist, float* best)
{
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
    }
}

// One thread per (cell-sorted) point. Searches the cube of cells within Chebyshev radius r = 1, 2, ... around the
// point's cell; after radius r every unvisited point is at least `bound` away (distance from the point to the
// nearest face of the visited cube), so the search stops once best[2] <= bound^2 or the cube covers the grid.
__global__ void __launch_bounds__(256) knn_query_kernel(int P, const KnnGrid* g, const uint32_t* __restrict__ cell_start,
                                                        const float4* __restrict__ sorted_pts, float* __restrict__ out)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= P) return;
    const float4 me = sorted_pts[s];
    const int self = __float_as_int(me.w);
    const int dx = g->dims[0], dy = g->dims[1], dz = g->dims[2];
    int cx, cy, cz;
    knn_cell_of(g, me.x, me.y, me.z, cx, cy, cz);
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    const float cell = g->cell;
    const int rmax = max(dx, max(dy, dz));
    for (int r = 0; r <= rmax; r++) {
        const int x0 = cx - r, x1 = cx + r, y0 = cy - r, y1 = cy + r, z0 = cz - r, z1 = cz + r;
        for (int z = max(z0, 0); z <= min(z1, dz - 1); z++) {
            for (int y = max(y0, 0); y <= min(y1, dy - 1); y++) {
                const bool shell_zy = (z == z0 || z == z1 || y == y0 || y == y1);
                // on a z/y face the whole x-row is new; otherwise only the two end cells are
                const int xs = shell_zy ? 1 : (x1 - x0 > 0 ? x1 - x0 : 1);
                for (int x = x0; x <= x1; x += xs) {
                    if (x < 0 || x >= dx) continue;
                    const uint32_t c = (uint32_t)((z * dy + y) * dx + x);
                    const uint32_t b = cell_start[c], e = cell_start[c + 1];
                    for (uint32_t k = b; k < e; k++) {
                        const float4 q = sorted_pts[k];
                        if (__float_as_int(q.w) == self) continue;   // self excluded by index, simple_knn.cu:157,175
                        const float ddx = q.x - me.x, ddy = q.y - me.y, ddz = q.z - me.z;
                        knn_update3(ddx * ddx + ddy * ddy + ddz * ddz, best);
                    }
                }
            }
        }
        const bool covers = x0 <= 0 && y0 <= 0 && z0 <= 0 && x1 >= dx - 1 && y1 >= dy - 1 && z1 >= dz - 1;
        if (covers) break;
        // distance to the nearest face of the visited cube that is not also a grid boundary
        float bound = FLT_MAX;
        const float ox = g->origin[0], oy = g->origin[1], oz = g->origin[2];
        if (x0 > 0) bound = fminf(bound, me.x - (ox + x0 * cell));
        if (x1 < dx - 1) bound = fminf(bound, (ox + (x1 + 1) * cell) - me.x);
        if (y0 > 0) bound = fminf(bound, me.y - (oy + y0 * cell));
        if (y1 < dy - 1) bound = fminf(bound, (oy + (y1 + 1) * cell) - me.y);
        if (z0 > 0) bound = fminf(bound, me.z - (oz + z0 * cell));
        if (z1 < dz - 1) bound = fminf(bound, (oz + (z1 + 1) * cell) - me.z);
        // shave a little for the rounding of the cell assignment / face positions
        bound = fmaxf(bound - 1e-5f * fmaxf(cell, fabsf(bound)), 0.f);
        if (best[2] <= bound * bound) break;
    }
    out[self] = (best[0] + best[1] + best[2]) / 3.0f;   // simple_knn.cu:182
}

}  // namespace gsr
