// gs_views.h -- the multi-view entry point: ONE launch per pipeline stage for all views of a mapping iteration.
//
// The mapping back-end renders the SAME Gaussians from up to ten keyframes per iteration (utils/slam_backend.py:357,526,657) and
// back-propagates all of them before one optimizer step. View by view that is ~12 dependent launches per view, each of which -- at
// SLAM scale, 30-50k Gaussians -- fills a tenth of the chip and costs mostly its launch / drain latency (profiles/r03_phase_cycles.json:
// ~5 us of every small kernel is outside any wave's lifetime). Here blockIdx.y (blockIdx.z for the long-list sort) selects the view:
// every view keeps its own geometry / image / binning buffers (the layout of the single-view path, carved on the device from the
// buffers' base addresses), the per-view pointers travel by value in the kernel arguments (ViewTable, <= 12 views, 2.6 KB), and the
// kernels are the single-view BODIES of gs_forward.h / gs_render.h / gs_backward.h, unchanged.
//
// Parameter gradients: the views' geometry kernels run concurrently, so they cannot add to one gradient buffer. Each view writes
// its parameter gradients to a scratch row; views_reduce_kernel then adds them to the target in VIEW ORDER, visible views only, one
// separately rounded addition per view -- the arithmetic of V consecutive single-view calls in accumulate mode, bit for bit.
#pragma once
#include "gs_backward.h"

namespace gsr {

constexpr int MAX_VIEWS = 12;

struct ViewSlot {
    const float* viewmatrix; const float* projmatrix; const float* projmatrix_raw; const float* cam_pos;
    const float* dx; const float* ds; const float* dr;                     // this view's deltas of the dynamic subset (raw mode), or null
    char* geom; char* image; char* binning;                                 // the view's scratch buffers, as the allocators returned them
    float* out_color; float* out_depth; float* out_opacity; int* radii; int* n_touched;
    const float* dL_dpix; const float* dL_dpix_depth;
    float* dL_dmean2D; float* ddx; float* dds; float* ddr; float* tau_sum;  // per-view gradients (screen space, deltas, pose)
    float* part;                                                            // scratch row of this view's parameter gradients
    uint32_t* mailbox; uint32_t cap; uint32_t cap_tile; uint32_t seq;
    // flow mode (render_flow, gs_rasterizer.h gsr_raw_inputs.flow_*): this view's second displacement and the two projections; ddx2 its gradient
    const float* flow_dx2; const float* flow_proj1; const float* flow_proj2; float* ddx2;
    const int* flow_clip;                                                   // optional tile rectangle of a flow view (gsr_view.flow_clip)
};
struct ViewTable { ViewSlot v[MAX_VIEWS]; };
static_assert(sizeof(ViewTable) + sizeof(PreprocessArgs) + 64 <= 4096 && sizeof(ViewTable) + sizeof(GeomBwdArgs) + 64 <= 4096,
              "the view table travels by value in the kernel arguments (4 KB)");

struct ViewDims { int P, W, H, gx, gy, T, nblocks; };

__device__ __forceinline__ GeomState view_geom(const ViewSlot& s, int P) { char* p = s.geom; return GeomState::from(p, (size_t)P); }
__device__ __forceinline__ ImageState view_image(const ViewSlot& s, const ViewDims& d)
{
    char* p = s.image;
    return ImageState::from(p, (size_t)d.W * d.H, (size_t)d.T, (size_t)d.P);
}
__device__ __forceinline__ BinningPtrs view_binning(const ViewSlot& s, const ViewDims& d) { return carve_binning(s.binning, s.cap, s.cap, (size_t)d.T); }

template <bool RAW, bool PRE = false>
__global__ void __launch_bounds__(GB) preprocess_views_kernel(PreprocessArgs a, ViewTable t, ViewDims d)
{
    const ViewSlot& s = t.v[blockIdx.y];
    const GeomState geom = view_geom(s, d.P);
    const ImageState img = view_image(s, d);
    a.viewmatrix = s.viewmatrix; a.projmatrix = s.projmatrix; a.cam_pos = s.cam_pos;
    a.raw.dx = s.dx; a.raw.ds = s.ds; a.raw.dr = s.dr;
    a.raw.flow_dx2 = s.flow_dx2; a.raw.flow_proj1 = s.flow_proj1; a.raw.flow_proj2 = s.flow_proj2; a.raw.flow_clip = s.flow_clip;
    a.radii = s.radii; a.n_touched = s.n_touched;
    a.rec = geom.rec; a.cov3D = geom.cov3D; a.clamped = geom.clamped; a.tiles_touched = geom.tiles_touched; a.block_sums = geom.block_sums;
    a.tile_count = img.tile_count; a.flags = img.tile_count + (size_t)d.T * CTR_STRIDE; a.block_tile_base = img.block_tile_base;
    preprocess_fwd_body<RAW, PRE>(a);
}

template <int SEGS, int RMAX>
__global__ void __launch_bounds__(TO_COLS * SEGS) tile_offsets_views_kernel(ViewTable t, ViewDims d)
{
    const ImageState img = view_image(t.v[blockIdx.y], d);
    tile_offsets_body<SEGS, RMAX>(d.nblocks, d.T, img.block_tile_base, img.tile_count);
}

__global__ void __launch_bounds__(1024) scan_views_kernel(ViewTable t, ViewDims d)
{
    const ViewSlot& s = t.v[blockIdx.y];
    const GeomState geom = view_geom(s, d.P);
    const ImageState img = view_image(s, d);
    scan_body(d.nblocks, geom.block_sums, geom.block_base, d.T, img.tile_count, img.ranges, img.tile_cursor, nullptr, s.cap, s.cap_tile, img.chunk_base,
              geom.header, s.mailbox, s.seq);
}

__global__ void __launch_bounds__(GB) scatter_views_kernel(ViewTable t, ViewDims d, int eager, int order_tiles)
{
    const ViewSlot& s = t.v[blockIdx.y];
    const GeomState geom = view_geom(s, d.P);
    const ImageState img = view_image(s, d);
    const BinningPtrs bin = view_binning(s, d);
    if (order_tiles && blockIdx.x == gridDim.x - 1) {                // the launch's extra block per view (gs_forward.h F3b)
        extern __shared__ uint32_t s_dyn[];
        if (!(geom.header[HDR_FLAGS] & FLAG_OVERFLOW)) {
            order_tiles_body(d.T, img.ranges, s_dyn, img.tile_count, (order_tiles & 2) != 0, geom.header[HDR_MAX_TILE], (uint32_t)(order_tiles >> 2));
        }
        return;
    }
    scatter_instances_body(d.P, d.gx, d.gy, s.radii, geom.rec, geom.tiles_touched, geom.block_base, geom.point_offsets, img.tile_cursor, img.ranges,
                           img.block_tile_base, bin.keys, bin.inst_gauss, geom.header, 1, s.cap, s.cap, eager, s.flow_clip);
}

template <int CAP, int LOWER>
__global__ void __launch_bounds__(256) sort_tiles_views_kernel(ViewTable t, ViewDims d)
{
    const ViewSlot& s = t.v[blockIdx.y];
    const GeomState geom = view_geom(s, d.P);
    const ImageState img = view_image(s, d);
    const BinningPtrs bin = view_binning(s, d);
    sort_tiles_body<CAP, LOWER>(d.T, img.ranges, bin.keys, bin.inst_gauss, bin.sorted, geom.header);
}

template <int CHUNKK>
__global__ void __launch_bounds__(256) sort_long_chunks_views_kernel(ViewTable t, ViewDims d, uint32_t lower)
{
    const ViewSlot& s = t.v[blockIdx.z];
    const GeomState geom = view_geom(s, d.P);
    const ImageState img = view_image(s, d);
    const BinningPtrs bin = view_binning(s, d);
    sort_long_chunks_body<CHUNKK>(img.ranges, bin.keys, geom.header, lower);
}
template <int CHUNKK>
__global__ void __launch_bounds__(256) rank_long_chunks_views_kernel(ViewTable t, ViewDims d, uint32_t lower)
{
    const ViewSlot& s = t.v[blockIdx.z];
    const GeomState geom = view_geom(s, d.P);
    const ImageState img = view_image(s, d);
    const BinningPtrs bin = view_binning(s, d);
    rank_long_chunks_body<CHUNKK>(img.ranges, bin.keys, bin.inst_gauss, bin.sorted, geom.header, lower);
}

__global__ void __launch_bounds__(RB) render_fwd_views_kernel(ViewTable t, ViewDims d, const float* __restrict__ bg, int fuse_sort, int order_tiles)
{
    const ViewSlot& s = t.v[blockIdx.y];
    const GeomState geom = view_geom(s, d.P);
    const ImageState img = view_image(s, d);
    const BinningPtrs bin = view_binning(s, d);
    render_fwd_body(d.T, d.gx, img.ranges, bin.sorted, d.W, d.H, geom.rec, bg, img.final_T, img.n_contrib, s.out_color, s.out_depth, s.out_opacity, s.n_touched,
                    img.final_C, bin.ckpt, geom.header, fuse_sort ? (const uint64_t*)bin.keys : nullptr, (const uint32_t*)bin.inst_gauss, bin.sorted,
                    (const uint32_t*)img.chunk_base, bin.chunk_info, order_tiles ? (const uint32_t*)img.tile_count : nullptr, (order_tiles & 2) ? 1 : 0, TrackLossArgs{});
}

__global__ void __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(GSR_BWD_WAVES, 8))) render_bwd_views_kernel(ViewTable t, ViewDims d, const float* __restrict__ bg)
{
    const ViewSlot& s = t.v[blockIdx.y];
    if (blockIdx.x >= s.cap / (uint32_t)CHUNK + (uint32_t)d.T) return;   // the grid is sized for the largest view (s.cap: this view's num_rendered):
                                                                        // beyond its own bound a view has no work-item table entry to read
    const GeomState geom = view_geom(s, d.P);
    const ImageState img = view_image(s, d);
    render_bwd_body(d.T, d.gx, (const char*)s.binning, (const uint32_t*)geom.header, d.W, d.H, bg, geom.rec, img.final_T, img.final_C, img.n_contrib,
                    s.dL_dpix, s.dL_dpix_depth);
}

// Layout of a view's scratch row (floats): the optimizer's parameter order, each tensor whole -- xyz[3P] f_dc[3P] f_rest[3(M-1)P] opacity[P]
// scaling[SP] rotation[4P].
struct PartLayout { size_t xyz, f_dc, f_rest, opacity, scaling, rotation, total; };
__host__ __device__ inline PartLayout part_layout(size_t P, int M, int S)
{
    PartLayout L;
    L.xyz = 0; L.f_dc = 3 * P; L.f_rest = L.f_dc + 3 * P; L.opacity = L.f_rest + 3 * (size_t)(M - 1) * P; L.scaling = L.opacity + P;
    L.rotation = L.scaling + (size_t)S * P; L.total = L.rotation + 4 * P;
    return L;
}

template <bool RAW, bool PRE = false>
__global__ void __launch_bounds__(256) geometry_bwd_views_kernel(GeomBwdArgs a, ViewTable t, ViewDims d)
{
    const ViewSlot& s = t.v[blockIdx.y];
    const GeomState geom = view_geom(s, d.P);
    a.viewmatrix = s.viewmatrix; a.projmatrix = s.projmatrix; a.projmatrix_raw = s.projmatrix_raw; a.campos = s.cam_pos;
    a.raw.dx = s.dx; a.raw.ds = s.ds; a.raw.dr = s.dr;
    a.raw.flow_dx2 = s.flow_dx2; a.raw.flow_proj1 = s.flow_proj1; a.raw.flow_proj2 = s.flow_proj2;
    a.radii = s.radii; a.clamped = geom.clamped; a.cov3Ds = geom.cov3D; a.tiles_touched = geom.tiles_touched; a.point_offsets = geom.point_offsets;
    a.bin_base = s.binning; a.header = geom.header;
    a.dL_dmean2D = s.dL_dmean2D;
    a.rawg.ddx = s.ddx; a.rawg.dds = s.dds; a.rawg.ddr = s.ddr; a.rawg.ddx2 = s.ddx2;
    a.tau_partials = s.tau_sum ? geom.tau_partials : nullptr;
    a.accumulate = 0;                                    // a full row per view; views_reduce_kernel does the (ordered) accumulation
    if (!a.pose_only) {
        const PartLayout L = part_layout((size_t)a.P, a.M, a.raw.scale_dim);
        a.dL_dmean3D = s.part + L.xyz; a.rawg.f_dc = s.flow_proj1 ? nullptr : s.part + L.f_dc; a.rawg.f_rest = a.M > 1 ? s.part + L.f_rest : nullptr;
        a.dL_dopacity = s.part + L.opacity; a.dL_dscale = s.part + L.scaling; a.dL_drot = s.part + L.rotation;
    }
    geometry_bwd_body<RAW, PRE>(a);
}

__global__ void __launch_bounds__(384) tau_sum_views_kernel(ViewTable t, ViewDims d)
{
    const ViewSlot& s = t.v[blockIdx.y];
    if (!s.tau_sum) return;
    tau_sum_body((d.P + 255) / 256, view_geom(s, d.P).tau_partials, s.tau_sum);
}

// target[i] (+)= part_0[i] + part_1[i] + ..., views in order, only the views that saw Gaussian i (radii > 0), every addition its own
// rounding -- the result of V single-view backward passes in accumulate mode (accumulate != 0), or of autograd summing V dense
// gradients whose invisible rows are zero (accumulate == 0: the first visible view assigns).
struct ReduceTargets { float* xyz; float* f_dc; float* f_rest; float* opacity; float* scaling; float* rotation; };
__global__ void __launch_bounds__(256) views_reduce_kernel(int V, ViewTable t, int P, int M, int S, ReduceTargets out, int accumulate)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const PartLayout L = part_layout((size_t)P, M, S);
    uint32_t seen = 0;
#pragma unroll
    for (int v = 0; v < MAX_VIEWS; v++) seen |= (v < V && t.v[v].radii[i] > 0 ? 1u : 0u) << v;
    if (!seen && accumulate) return;                     // accumulate mode leaves the rows of Gaussians no view saw alone
    // Four elements of the row at a time: the views' values are all requested first (and the old value, in accumulate mode), then added in
    // view order. (Element by element, view by view, every term waited for its own load: V x 14 dependent round trips per thread in a kernel
    // of 76 blocks at SLAM sizes.)
    auto fold = [&](float* dst, size_t base, int width) {
        if (!dst) return;                                 // a gradient the caller did not ask for (flow mode: everything but the positions)
        for (int k0 = 0; k0 < width; k0 += 4) {
            const int w = min(4, width - k0);
            const size_t at = (size_t)i * width + k0;
            float g[MAX_VIEWS][4], acc[4];
#pragma unroll
            for (int v = 0; v < MAX_VIEWS; v++) {
                const bool on = v < V && ((seen >> v) & 1u);
                const float* row = t.v[v].part + base + at;
#pragma unroll
                for (int k = 0; k < 4; k++) g[v][k] = on && k < w ? row[k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = accumulate && k < w ? dst[at + k] : 0.f;
            bool first = !accumulate;
#pragma unroll
            for (int v = 0; v < MAX_VIEWS; v++) {
                if (!(v < V && ((seen >> v) & 1u))) continue;
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] = first ? g[v][k] : add_separately(acc[k], g[v][k]);
                first = false;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) if (k < w) dst[at + k] = acc[k];
        }
    };
    fold(out.xyz, L.xyz, 3);
    fold(out.f_dc, L.f_dc, 3);
    if (M > 1) fold(out.f_rest, L.f_rest, 3 * (M - 1));
    fold(out.opacity, L.opacity, 1);
    fold(out.scaling, L.scaling, S);
    fold(out.rotation, L.rotation, 4);
}

}  // namespace gsr
