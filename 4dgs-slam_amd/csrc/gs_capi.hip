// gs_capi.hip -- host orchestration + the extern "C" boundary declared in include/gs_rasterizer.h.
// Mirrors CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (DGR/cuda_rasterizer/rasterizer_impl.cu:141-153,198-344,348-455) with an MI355X-first pipeline:
//
//   forward : F1 preprocess (+ per-block tile histogram rows, + block sums) -> F1b column scan -> F2 single-pass scans (+ host mailbox)
//             -> F3 scatter -> [F4 sort of lists > 1024] -> F5 tile kernel (sorts its own list, composites, checkpoints),
//             all enqueued speculatively on a binning buffer sized from the previous frame; the host then reads R from the
//             mailbox and only redoes F3..F5 if the scan kernel flagged the capacity as too small
//   backward: B1 gradient pass, one block per 128-entry chunk of a tile list (swap/DPP reductions, per-instance slots)
//             -> B2 per-Gaussian gather + geometry -> pose-gradient sum
//   extras  : raw-parameter entry points (fused prologue), fused L1 losses, fused Adam, exact 3-NN (simple_knn)
//
// No global 64-bit radix sort, no float atomics, no cooperative-groups block trees.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <sched.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <atomic>

#include "../../include/gs_rasterizer.h"
#include "../../include/simple_knn.h"
#include "gs_backward.h"
#include "gs_views.h"
#include "gs_knn.h"
#include "gs_loss.h"
#include "gs_hexplane.h"
#include "gs_hexplane_binned.h"
#include "gs_linear.h"
#include "gs_mlp.h"
#include "gs_dense.h"
#include "../../include/dense_layers.h"
#include "gs_nodes.h"
#include "../../include/slam_losses.h"
#include "../../include/slam_map.h"
#include "gs_map.h"

namespace gsr {

static thread_local std::string g_last_error;

#define GSR_HIP_CHECK(expr)                                                                              \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                            \
            return GSR_ERR_HIP;                                                                          \
        }                                                                                                \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: `done` holds one bit per device ordinal, so a process
// that drives several GPUs sets it on each of them (a process-wide flag left the second device at the 64 KB default and the 156 KB launch
// failed there); atomic: the entry points may be called from several host threads.
static int ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done)
{
    int dev = 0;
    GSR_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        GSR_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        done.fetch_or(bit, std::memory_order_release);
    }
    return 0;
}

static size_t binning_bytes(size_t carve_R, size_t cap_sorted, size_t ntiles)
{
    return (size_t)(carve_binning(nullptr, carve_R, cap_sorted, ntiles).end - (char*)nullptr) + 256;
}
template <typename F>
static size_t required(F&& f)
{
    char* p = nullptr;
    f(p);
    return reinterpret_cast<size_t>(p) + 256;
}

// ---- per-kernel timing with events on the launch stream ---------------------------------------------------------
enum KernelId { K_PREPROCESS = 0, K_SCAN, K_SCATTER, K_SORT, K_RENDER_FWD, K_RENDER_BWD, K_GEOM_BWD, K_KNN, K_COUNT };
static const char* const kKernelNames[K_COUNT] = {"preprocess_fwd", "scan", "scatter_instances", "sort_tiles",
                                                  "render_fwd", "render_bwd", "geometry_bwd", "knn_total"};
struct Profiler {
    unsigned mask = 0;   // bit i set => kernel id i is timed
    struct Rec { hipEvent_t a, b; int id; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[K_COUNT] = {0};
    int calls[K_COUNT] = {0};
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    void drain()
    {
        for (auto& r : pending) {
            (void)hipEventSynchronize(r.b);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { total_ms[r.id] += ms; calls[r.id]++; }
            pool.push_back(r.a); pool.push_back(r.b);
        }
        pending.clear();
    }
};
static Profiler g_prof;
struct ScopedKernelTimer {
    int id; hipStream_t s; hipEvent_t a{}, b{}; bool on;
    ScopedKernelTimer(int id_, hipStream_t s_) : id(id_), s(s_), on((g_prof.mask >> id_) & 1u)
    {
        if (on) { a = g_prof.get(); b = g_prof.get(); (void)hipEventRecord(a, s); }
    }
    ~ScopedKernelTimer()
    {
        if (on) { (void)hipEventRecord(b, s); g_prof.pending.push_back({a, b, id}); if (g_prof.pending.size() > 4096) g_prof.drain(); }
    }
};

static int debug_sync(int debug, hipStream_t s, const char* what)
{
    // CHECK_CUDA(..., debug) of the reference (auxiliary.h:166-173): synchronise after every stage when debugging.
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && debug) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { g_last_error = std::string(what) + ": " + hipGetErrorString(e); return GSR_ERR_HIP; }
    return 0;
}
#define GSR_STAGE(what) do { int _r = debug_sync(debug, stream, what); if (_r) return _r; } while (0)

// Host mailbox: 8 words of pinned, host-coherent memory per host thread; word 4 carries the sequence number of the
// forward call whose header (words 0-3) is valid. Plus the allocation size that was enough last time (speculative
// binning allocation while the GPU is still busy with the preprocess).
// ... all of it per (host thread, device): a thread that renders on several GPUs gets one mailbox and one capacity estimate per device
// (the mailbox's device pointer is the one hipHostGetDevicePointer returns for THAT device).
// ... and per view SLOT of the multi-view entry point (slot 0 = the single-view calls): the v-th view of consecutive mapping iterations
// looks alike, the views of one iteration do not.
struct SpecState { uint32_t* mailbox = nullptr; uint32_t* mailbox_dev = nullptr; uint32_t seq = 0; size_t last_R_alloc = 0; uint32_t last_max_tile = 0; };
constexpr int VIEW_SLOT_GROUPS = 4;                             // gsr_set_option("view_slot_group", g): a caller that splits one iteration's views over several calls names the call
constexpr int SPEC_SLOTS = 1 + 2 * MAX_VIEWS * VIEW_SLOT_GROUPS;
static thread_local SpecState t_spec[16][SPEC_SLOTS];   // [device][0 = single-view calls | per group: 1..V views of a batch | MAX_VIEWS+1.. views of a flow batch]
static thread_local int t_view_slot_group = 0;
// a flow view's tile rectangle (gsr_view.flow_clip) for a view that goes through the single-view path inside a gsr_forward_views call
static thread_local const int* t_clip_single = nullptr;
// gsr_track_step (include/slam_map.h): the loss epilogue of the forward pass's render_fwd launches, and the launch that replaces tau_sum_kernel
static thread_local const TrackLossArgs* t_track_loss = nullptr;
struct TrackTail { const float* exposure_partials; float* dL_dexposure; CameraStepArgs step; };
static thread_local const TrackTail* t_track_tail = nullptr;
static thread_local SpecState* t_cur = &t_spec[0][0];
static int select_device_state(int slot = 0)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    t_cur = &t_spec[dev][slot];
    return dev;
}
#define t_mailbox (t_cur->mailbox)
#define t_mailbox_dev (t_cur->mailbox_dev)
#define t_seq (t_cur->seq)
#define t_last_R_alloc (t_cur->last_R_alloc)
#define t_last_max_tile (t_cur->last_max_tile)
static thread_local bool t_use_mailbox = true;
static thread_local bool t_speculate = true;
// Lazy mode (gsr_set_option("lazy", 1)): a speculative forward pass returns WITHOUT waiting for the scan kernel's header -- no host
// synchronisation at all, which is also what makes the call capturable in a hipGraph. The value returned in place of num_rendered is
// then the capacity the binning buffer was laid out for (an upper bound of R; gsr_backward only sizes its grid from it, the kernels
// read the true counts from the device header). If a frame outgrows the capacity, the speculative kernels and both backward kernels
// of that frame exit at once (outputs / gradients of THAT call are undefined) and the sticky overflow counter in the mailbox is
// bumped: a caller that uses lazy mode polls gsr_forward_status() at a convenient point and repeats the affected work eagerly.
static thread_local bool t_lazy = false;
static thread_local int t_cap_margin_permille = 125;   // head room of a speculative binning buffer over the last frame's count (gsr_set_option)
static thread_local int t_cap_tile_margin_permille = -1;  // head room of the longest tile list (which picks the sort kernels); < 0: max(250, cap_margin_permille)
static inline unsigned tile_margin() { return (unsigned)(t_cap_tile_margin_permille >= 0 ? t_cap_tile_margin_permille : std::max(250, t_cap_margin_permille)); }
static thread_local int t_cap_test_shrink_permille = 0; // TEST facility: > 0 lays speculative buffers out for that fraction of the last count (forces overflows)
static thread_local int t_cap_floor = 0;               // smallest speculative capacity, in instances (gsr_set_option "cap_floor")
static inline size_t spec_capacity(size_t last)
{
    if (t_cap_test_shrink_permille > 0) return (size_t)((unsigned long long)last * (unsigned)t_cap_test_shrink_permille / 1000ull) + 1;
    return std::max(last + (size_t)((unsigned long long)last * (unsigned)t_cap_margin_permille / 1000ull) + 4096, (size_t)t_cap_floor);
}
static thread_local bool t_options_read = false;
static thread_local unsigned t_views_batched = 0;
static bool t_fuse_sort = getenv("GSR_FUSE_SORT") ? getenv("GSR_FUSE_SORT")[0] != '0' : true;   // sort short tile lists inside render_fwd
// render_bwd's work items: full pieces in tile order, then the partial pieces longest first at the end of every XCD's sequence (gs_device.h:
// item_block_*; one extra block of the scatter launch ranks them). GSR_ORDER_ITEMS=0: tile order, as rounds 2-5 (A/B runs; the results are
// bit-identical either way)
static bool t_order_items = getenv("GSR_ORDER_ITEMS") ? getenv("GSR_ORDER_ITEMS")[0] != '0' : true;
// SH coefficient rows move through LDS in preprocess_fwd / geometry_bwd (gs_backward.h); GSR_SH_ROWS=0: per lane (bit-identical results)
static bool t_sh_rows = getenv("GSR_SH_ROWS") ? getenv("GSR_SH_ROWS")[0] != '0' : true;
// render_fwd's blocks take the tiles of their XCD band by list length, dealt over the band's CUs (gs_forward.h F3c). GSR_ORDER_TILES=0: band order
static bool t_order_tiles = getenv("GSR_ORDER_TILES") ? getenv("GSR_ORDER_TILES")[0] != '0' : true;
static const int t_deal_heavy = getenv("GSR_DEAL_HEAVY") ? atoi(getenv("GSR_DEAL_HEAVY")) : 1;      // lists per CU held back for the CUs with one block less (gs_forward.h)
// 0: band order; 1 + heavy otherwise (the scatter launch's argument)
static int order_fwd_tiles(int T, bool lds_hist) { return t_order_items && t_order_tiles && lds_hist && T / 8 >= ORDER_FWD_MIN_BAND ? 1 + std::max(0, std::min(7, t_deal_heavy)) : 0; }
// HexPlane plane gradients as fixed-point integer sums (gs_hexplane_binned.h: HexOrd): bitwise the same run to run. 0: float atomics (rounds 1-5)
static std::atomic<int> g_hex_ordered{getenv("GSR_HEX_ORDERED") ? (getenv("GSR_HEX_ORDERED")[0] != '0' ? 1 : 0) : 1};
static void read_option_env()
{
    if (t_options_read) return;
    t_options_read = true;
    if (const char* e = getenv("GSR_MAILBOX")) t_use_mailbox = e[0] != '0';
    if (const char* e = getenv("GSR_SPECULATE")) t_speculate = e[0] != '0';
    if (const char* e = getenv("GSR_LAZY")) t_lazy = e[0] != '0';
}

static int ensure_mailbox()
{
    if (!t_mailbox) {
        GSR_HIP_CHECK(hipHostMalloc((void**)&t_mailbox, 8 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable));
        memset(t_mailbox, 0, 8 * sizeof(uint32_t));
        GSR_HIP_CHECK(hipHostGetDevicePointer((void**)&t_mailbox_dev, t_mailbox, 0));
    }
    return 0;
}

static int wait_for_header(hipStream_t stream, const uint32_t* device_header, uint32_t seq, uint32_t out[4])
{
    if (t_use_mailbox) {
        // spin on the mailbox (the scan kernel is a few microseconds away when the GPU is not backed up); after ~1e5 polls the wait is
        // evidently long, so the core is yielded between polls; the stream is checked now and then so that a faulted / finished stream
        // cannot hang us
        for (unsigned long long spins = 0;; spins++) {
            if (__atomic_load_n(&t_mailbox[4], __ATOMIC_ACQUIRE) == seq) {
                for (int i = 0; i < 4; i++) out[i] = __atomic_load_n(&t_mailbox[i], __ATOMIC_RELAXED);
                return 0;
            }
            if (spins > 100000ull) sched_yield();
            if ((spins & 0xFFFFF) == 0xFFFFF) {
                hipError_t q = hipStreamQuery(stream);
                if (q == hipSuccess) {   // stream drained: the store must be visible by now, otherwise fall back for good
                    if (__atomic_load_n(&t_mailbox[4], __ATOMIC_ACQUIRE) == seq) continue;
                    t_use_mailbox = false;
                    break;
                }
                if (q != hipErrorNotReady) { g_last_error = std::string("stream error while waiting for the header: ") + hipGetErrorString(q); return GSR_ERR_HIP; }
            }
        }
    }
    GSR_HIP_CHECK(hipMemcpyAsync(out, device_header, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GSR_HIP_CHECK(hipStreamSynchronize(stream));
    return 0;
}

}  // namespace gsr

using namespace gsr;

extern "C" {

const char* gsr_last_error(void) { return g_last_error.c_str(); }
const char* gsr_version(void) { return "gs_rasterizer_hip 0.1 (gfx950)"; }

size_t gsr_geometry_buffer_size(int P) { return required([&](char*& p) { GeomState::from(p, (size_t)P); }); }
size_t gsr_image_buffer_size(int width, int height, int P)
{
    const size_t T = (size_t)((width + TILE_X - 1) / TILE_X) * ((height + TILE_Y - 1) / TILE_Y);
    return required([&](char*& p) { ImageState::from(p, (size_t)width * height, T, (size_t)P); });
}
size_t gsr_binning_buffer_size(int R_alloc) { return binning_bytes((size_t)R_alloc, (size_t)R_alloc, (size_t)65536); }   // sized for frames of up to 65 536 tiles (4096 x 4096 pixels)

int gsr_set_option(const char* name, int value)
{
    read_option_env();
    if (!name) { g_last_error = "gsr_set_option: null name"; return GSR_ERR_INVALID_ARGUMENT; }
    const std::string n(name);
    if (n == "views_batched") return (int)t_views_batched;     // read-only: multi-view calls of this thread that took the one-launch-per-stage path
    if (n == "hex_ordered") {                                  // process-wide (the workspace sizes depend on it); value < 0 only reads
        const int old_ordered = g_hex_ordered.load();
        if (value >= 0) g_hex_ordered.store(value ? 1 : 0);
        return old_ordered;
    }
    if (n == "cap_margin_permille") {
        const int old_margin = t_cap_margin_permille;
        if (value >= 0) t_cap_margin_permille = value > 4000 ? 4000 : value;
        return old_margin;
    }
    if (n == "cap_tile_margin_permille") {
        const int old_margin = t_cap_tile_margin_permille < 0 ? 1000000 : t_cap_tile_margin_permille;      // 1000000 = "follow cap_margin_permille"
        if (value >= 1000000) t_cap_tile_margin_permille = -1;
        else if (value >= 0) t_cap_tile_margin_permille = value > 16000 ? 16000 : value;
        return old_margin;
    }
    if (n == "view_slot_group") {
        const int old_group = t_view_slot_group;
        if (value >= 0) t_view_slot_group = value >= VIEW_SLOT_GROUPS ? VIEW_SLOT_GROUPS - 1 : value;
        return old_group;
    }
    if (n == "cap_floor") {
        const int old_floor = t_cap_floor;
        if (value >= 0) t_cap_floor = value > (1 << 26) ? (1 << 26) : value;
        return old_floor;
    }
    if (n == "cap_test_shrink_permille") {
        const int old_shrink = t_cap_test_shrink_permille;
        if (value >= 0) t_cap_test_shrink_permille = value > 1000 ? 1000 : value;
        return old_shrink;
    }
    bool* opt = n == "speculate" ? &t_speculate : n == "lazy" ? &t_lazy : n == "mailbox" ? &t_use_mailbox : n == "order_items" ? &t_order_items : n == "sh_rows" ? &t_sh_rows : nullptr;
    if (!opt) { g_last_error = "gsr_set_option: unknown option '" + n + "' (speculate, lazy, mailbox, order_items, sh_rows, cap_margin_permille, cap_tile_margin_permille, cap_floor, view_slot_group)"; return GSR_ERR_INVALID_ARGUMENT; }
    const int old = *opt ? 1 : 0;
    if (value >= 0) *opt = value != 0;
    return old;
}

int gsr_forward_status(unsigned int* overflow_count, unsigned int* last_num_rendered)
{
    select_device_state();
    if (overflow_count) *overflow_count = t_mailbox ? __atomic_load_n(&t_mailbox[5], __ATOMIC_ACQUIRE) : 0u;
    if (last_num_rendered) *last_num_rendered = t_mailbox ? __atomic_load_n(&t_mailbox[0], __ATOMIC_ACQUIRE) : 0u;
    return 0;
}

int gsr_forward_status_views(unsigned int* overflow_count_total)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    unsigned int total = 0;
    for (int slot = 0; slot < SPEC_SLOTS; slot++)
        if (const uint32_t* mb = t_spec[dev][slot].mailbox) total += __atomic_load_n(&mb[5], __ATOMIC_ACQUIRE);
    if (overflow_count_total) *overflow_count_total = total;
    return 0;
}

/* Dev: per view slot of this thread and device, 8 words: mailbox R, flags, R_alloc, max tile, seq, overflow count | the host's estimates
 * last_R_alloc, last_max_tile. out: [slots][8]; returns the number of slots. */
int gsr_debug_view_slots(unsigned int* out, int max_slots)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    const int n = max_slots < SPEC_SLOTS ? max_slots : SPEC_SLOTS;
    for (int slot = 0; slot < n; slot++) {
        const SpecState& s = t_spec[dev][slot];
        for (int k = 0; k < 6; k++) out[slot * 8 + k] = s.mailbox ? __atomic_load_n(&s.mailbox[k], __ATOMIC_ACQUIRE) : 0u;
        out[slot * 8 + 6] = (unsigned int)s.last_R_alloc; out[slot * 8 + 7] = s.last_max_tile;
    }
    return n;
}

int gsr_profile_enable(int kernel_mask) { g_prof.mask = (unsigned)kernel_mask; return K_COUNT; }
void gsr_profile_reset(void)
{
    g_prof.drain();
    for (int i = 0; i < K_COUNT; i++) { g_prof.total_ms[i] = 0; g_prof.calls[i] = 0; }
}
int gsr_profile_read(const char** names, float* total_ms, int* calls, int cap)
{
    g_prof.drain();
    int n = 0;
    for (int i = 0; i < K_COUNT && n < cap; i++, n++) {
        if (names) names[n] = kKernelNames[i];
        if (total_ms) total_ms[n] = (float)g_prof.total_ms[i];
        if (calls) calls[n] = g_prof.calls[i];
    }
    return n;
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, unsigned char* present, void* stream_)
{
    (void)projmatrix;
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) { g_last_error = "gsr_mark_visible: null argument"; return GSR_ERR_INVALID_ARGUMENT; }
    if (P == 0) return 0;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, viewmatrix, present);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"

static RawInputs to_device_view(const gsr_raw_inputs* in)
{
    RawInputs r{};
    if (in) {
        r.xyz = in->xyz; r.log_scales = in->log_scales; r.scale_dim = in->scale_dim; r.raw_rot = in->raw_rotations;
        r.logit_opacity = in->logit_opacity; r.f_dc = in->features_dc; r.f_rest = in->features_rest;
        r.dyn_slot = in->dyn_slot; r.dx = in->dx; r.ds = in->ds; r.dr = in->dr; r.gather = in->gather;
        r.flow_dx2 = in->flow_dx2; r.flow_proj1 = in->flow_proj1; r.flow_proj2 = in->flow_proj2;
        r.flow_clip = in->flow_proj1 ? t_clip_single : nullptr;
        r.delta_mode = in->delta_mode; r.delta_stride = in->delta_stride;
    }
    return r;
}
static bool raw_inputs_ok(const gsr_raw_inputs* in, int M)
{
    const bool flow = in->flow_proj1 != nullptr;
    if (flow && (!in->flow_proj2 || M != 1 || (in->flow_dx2 && !in->dyn_slot) || in->delta_mode || in->delta_stride)) return false;
    if ((in->delta_mode != 0 && in->delta_mode != 1) || in->delta_stride < 0 || (in->delta_stride && (in->delta_stride < 4 || !in->delta_mode)) ||
        (in->delta_mode && in->gather)) return false;
    return in->xyz && in->log_scales && (in->scale_dim == 1 || in->scale_dim == 3) && in->raw_rotations && in->logit_opacity &&
           (flow || (in->features_dc && (M == 1 || in->features_rest))) && ((!in->dx && !in->ds && !in->dr) || in->dyn_slot || in->delta_mode);
}

static int forward_impl(gsr_alloc_fn geometry_alloc, void* geometry_user, gsr_alloc_fn binning_alloc, void* binning_user,
                gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
                const float* means3D, const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                float* out_depth, float* out_opacity, int* radii, int* n_touched, int debug, void* stream_, const gsr_raw_inputs* raw, int slot = 0)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || width <= 0 || height <= 0 || !geometry_alloc || !binning_alloc || !image_alloc) {
        g_last_error = "gsr_forward: invalid size or missing allocator"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (!background || !out_color || !out_depth || !out_opacity || !n_touched) { g_last_error = "gsr_forward: null output/background"; return GSR_ERR_INVALID_ARGUMENT; }
    if (P > 0 && raw) {
        if (!viewmatrix || !projmatrix || !cam_pos || M <= 0 || !raw_inputs_ok(raw, M)) { g_last_error = "gsr_forward_raw: null / inconsistent input"; return GSR_ERR_INVALID_ARGUMENT; }
        if (D < 0 || D > 3 || (D + 1) * (D + 1) > M) { g_last_error = "gsr_forward_raw: sh degree out of range for M"; return GSR_ERR_INVALID_ARGUMENT; }
    } else if (P > 0) {
        if (!means3D || !opacities || !viewmatrix || !projmatrix || !cam_pos) { g_last_error = "gsr_forward: null input"; return GSR_ERR_INVALID_ARGUMENT; }
        if (!cov3D_precomp && (!scales || !rotations)) { g_last_error = "gsr_forward: need scales+rotations or cov3D_precomp"; return GSR_ERR_INVALID_ARGUMENT; }
        // rasterizer_impl.cu:245-248 analogue: colours must come from somewhere
        if (!colors_precomp && (!shs || M <= 0)) { g_last_error = "gsr_forward: need shs (M>0) or colors_precomp"; return GSR_ERR_INVALID_ARGUMENT; }
        if (!colors_precomp && (D < 0 || D > 3 || (D + 1) * (D + 1) > M)) { g_last_error = "gsr_forward: sh degree out of range for M"; return GSR_ERR_INVALID_ARGUMENT; }
    }
    const int gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y, T = gx * gy;
    const size_t N = (size_t)width * height;
    select_device_state(slot);

    char* gchunk = geometry_alloc(geometry_user, gsr_geometry_buffer_size(P));
    char* ichunk = image_alloc(image_user, gsr_image_buffer_size(width, height, P));
    if (!gchunk || !ichunk) { g_last_error = "gsr_forward: allocation callback returned NULL"; return GSR_ERR_ALLOC; }
    GeomState geom = GeomState::from(gchunk, (size_t)P);
    ImageState img = ImageState::from(ichunk, N, (size_t)T, (size_t)P);
    const bool lds_hist = use_lds_hist((size_t)T);
    const size_t hist_lds_bytes = lds_hist ? (size_t)T * sizeof(uint32_t) : 0;
    const bool order_items = t_order_items && lds_hist;        // (the ordering block keeps the tiles' list lengths in the same dynamic LDS)
    if (!radii) radii = geom.internal_radii;   // rasterizer_impl.cu:232-235

    // Zero-filled scratch: the flag word only when someone can raise it (prefiltered; the reference's callers never set it), the
    // per-tile counters only when they are accumulated with atomics (the fallback for frames with more tiles than the LDS
    // histogram holds; otherwise tile_offsets_kernel writes them). The common case needs no memset at all.
    uint32_t* const flags = img.tile_count + (size_t)T * CTR_STRIDE;
    if (!lds_hist)
        GSR_HIP_CHECK(hipMemsetAsync(img.tile_count, 0, ((size_t)T * CTR_STRIDE + 1) * sizeof(uint32_t), stream));
    else if (prefiltered)
        GSR_HIP_CHECK(hipMemsetAsync(flags, 0, sizeof(uint32_t), stream));

    const int nblocks = (P + GB - 1) / GB;
    // small launches are latency-bound: the per-Gaussian kernels then request every row they may need up front (gs_forward.h)
    static const int eager_max = getenv("GSR_EAGER_MAX") ? atoi(getenv("GSR_EAGER_MAX")) : 512 * 1024;
    const int eager = P <= eager_max ? 1 : 0;
    if (P > 0) {
        PreprocessArgs a;
        a.P = P; a.D = D; a.M = M; a.W = width; a.H = height; a.gx = gx; a.gy = gy;
        a.means3D = means3D; a.scales = scales; a.scale_modifier = scale_modifier; a.rotations = rotations; a.opacities = opacities;
        a.shs = shs; a.cov3D_precomp = cov3D_precomp; a.colors_precomp = colors_precomp;
        a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.cam_pos = cam_pos;
        a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
        a.focal_y = height / (2.0f * tan_fovy); a.focal_x = width / (2.0f * tan_fovx);   // rasterizer_impl.cu:225-226
        a.prefiltered = prefiltered; a.radii = radii; a.n_touched = n_touched;
        a.rec = geom.rec; a.cov3D = geom.cov3D;
        a.clamped = geom.clamped; a.tiles_touched = geom.tiles_touched; a.block_sums = geom.block_sums; a.tile_count = img.tile_count;
        a.flags = flags; a.block_tile_base = lds_hist ? img.block_tile_base : nullptr;
        a.raw = to_device_view(raw);
        {
            ScopedKernelTimer tm(K_PREPROCESS, stream);
            a.eager = eager;
            // SH rows through LDS (gs_device.h: stage_rows): one 6.25 KB window per wave behind the tile histogram, reserved only when coefficients
            // above the DC band are read (the LDS histogram path: the windows sit in the same dynamic allocation)
            const bool sh_win = t_sh_rows && lds_hist && D > 0 && !colors_precomp && (M == 9 || M == 16) && !(raw && raw->flow_proj1);
            a.sh_win_offset = sh_win ? (int)((hist_lds_bytes + 15) & ~size_t(15)) : 0;
            const size_t pre_lds = sh_win ? (size_t)a.sh_win_offset + (size_t)(GB / 64) * SH_WIN_FLOATS * sizeof(float) : hist_lds_bytes;
            if (sh_win) {
                static std::atomic<unsigned long long> attr_set[3];
                const void* fn = raw && raw->delta_mode ? reinterpret_cast<const void*>(preprocess_fwd_kernel<true, true>)
                               : raw ? reinterpret_cast<const void*>(preprocess_fwd_kernel<true>) : reinterpret_cast<const void*>(preprocess_fwd_kernel<false>);
                const int rc = ensure_dynamic_lds(fn, (int)pre_lds, attr_set[raw && raw->delta_mode ? 2 : raw ? 1 : 0]);
                if (rc) return rc;
            }
            if (raw && raw->delta_mode) hipLaunchKernelGGL((preprocess_fwd_kernel<true, true>), dim3(nblocks), dim3(GB), pre_lds, stream, a);
            else if (raw) hipLaunchKernelGGL(preprocess_fwd_kernel<true>, dim3(nblocks), dim3(GB), pre_lds, stream, a);
            else hipLaunchKernelGGL(preprocess_fwd_kernel<false>, dim3(nblocks), dim3(GB), pre_lds, stream, a);
        }
        GSR_STAGE("preprocess_fwd");
        if (lds_hist) {
            ScopedKernelTimer tm(K_SCAN, stream);
            if (nblocks <= TO_SEGS * 16)
                hipLaunchKernelGGL((tile_offsets_kernel<TO_SEGS, 16>), dim3((T + TO_COLS - 1) / TO_COLS), dim3(TO_COLS * TO_SEGS), 0, stream, nblocks, T,
                               img.block_tile_base, img.tile_count, img.tile_cursor);      // (tile_cursor: free on this path; holds the dense list lengths)
            else
                hipLaunchKernelGGL((tile_offsets_kernel<TO_SEGS_BIG, 32>), dim3((T + TO_COLS - 1) / TO_COLS), dim3(TO_COLS * TO_SEGS_BIG), 0, stream, nblocks, T,
                               img.block_tile_base, img.tile_count, img.tile_cursor);
        }
        GSR_STAGE("tile_offsets");
    } else if (lds_hist) {
        GSR_HIP_CHECK(hipMemsetAsync(img.tile_count, 0, (size_t)T * CTR_STRIDE * sizeof(uint32_t), stream));   // no Gaussians: no columns to scan
    }
    // Speculation: a SLAM loop renders nearly the same scene again and again, so the binning buffer is allocated for what
    // sufficed last time plus slack and scatter / sort / render are enqueued right behind the scan, WITHOUT waiting for R;
    // the host reads R from the mailbox afterwards, while the GPU is already busy with them. The scan kernel compares the
    // frame's real needs with the speculative capacity and raises FLAG_OVERFLOW if they do not fit: the speculative kernels
    // then exit at once and the host redoes them on an exact-size buffer (also the path of the first call and of debug mode).
    read_option_env();
    if (t_lazy && t_mailbox) {
        // lazy mode never waits, so the capacity estimate is refreshed from whatever header the GPU published last (a few calls old)
        const uint32_t s0 = __atomic_load_n(&t_mailbox[4], __ATOMIC_ACQUIRE);
        if (s0 != 0) {
            const uint32_t r = __atomic_load_n(&t_mailbox[0], __ATOMIC_RELAXED), ra = __atomic_load_n(&t_mailbox[2], __ATOMIC_RELAXED);
            const uint32_t mx = __atomic_load_n(&t_mailbox[3], __ATOMIC_RELAXED);
            if (__atomic_load_n(&t_mailbox[4], __ATOMIC_ACQUIRE) == s0 && r <= 0x7fffffffu && ra <= 0x7fffffffu) {
                const size_t need = ra > r ? ra : r;
                // grow at once, shrink slowly: a view that briefly needs less must not take the slack away from the next one
                t_last_R_alloc = need > t_last_R_alloc ? need : t_last_R_alloc - (t_last_R_alloc - need) / 16;
                t_last_max_tile = mx > t_last_max_tile ? mx : t_last_max_tile;
            }
        }
    }
    const bool speculate = t_speculate && t_last_R_alloc && !debug && P > 0;
    const size_t cap = speculate ? spec_capacity(t_last_R_alloc) : 0;
    // longest tile list the speculative launches are sized for: decides which sort kernels run (and the chunk grid of the long-list sort)
    const uint32_t want_tile = t_last_max_tile + (uint32_t)((unsigned long long)t_last_max_tile * tile_margin() / 1000ull);
    const uint32_t cap_tile = want_tile <= (uint32_t)SORT_SMALL_CAP ? (uint32_t)SORT_SMALL_CAP
                            : want_tile <= (uint32_t)SORT_MID_CAP ? (uint32_t)SORT_MID_CAP
                            : want_tile <= (uint32_t)SORT_LDS_CAP ? (uint32_t)SORT_LDS_CAP
                            : (want_tile + (uint32_t)SORT_LDS_CAP - 1) / (uint32_t)SORT_LDS_CAP * (uint32_t)SORT_LDS_CAP;
    // F2b (gs_forward.h): on a speculative frame of the LDS-histogram path the scatter launch does the scan's work itself -- one launch less
    // (GSR_SCAN_IN_SCATTER=0: the scan as its own launch, as rounds 1-5)
    static const bool scan_in_scatter_ok = !(getenv("GSR_SCAN_IN_SCATTER") && getenv("GSR_SCAN_IN_SCATTER")[0] == '0');
    const bool scan_in_scatter = scan_in_scatter_ok && speculate && lds_hist && nblocks <= GB;
    {
        ScopedKernelTimer tm(K_SCAN, stream);
        { const int rc = ensure_mailbox(); if (rc) return rc; }
        if (++t_seq == 0) t_seq = 1;
        if (!scan_in_scatter)
            hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, nblocks, geom.block_sums, geom.block_base, T, img.tile_count,
                               img.ranges, img.tile_cursor, prefiltered ? flags : (const uint32_t*)nullptr, (uint32_t)cap, cap_tile, img.chunk_base,
                               geom.header,
                               t_use_mailbox ? t_mailbox_dev : nullptr, t_seq);
    }
    GSR_STAGE("scan");

    // scatter -> sort -> render on a binning buffer laid out for carve_R instances / cap_sorted sorted entries
    auto enqueue_binning_and_render = [&](char* chunk, size_t carve_R, size_t cap_sorted, bool spec, bool any_padding,
                                          uint32_t longest_list) -> int {
        const bool long_lists = longest_list > (uint32_t)SORT_SMALL_CAP;
        const BinningPtrs bin = carve_binning(chunk, carve_R, cap_sorted, (size_t)T);
        uint32_t* const chk = spec ? geom.header : nullptr;
        if (any_padding) GSR_HIP_CHECK(hipMemsetAsync(bin.keys, 0xFF, cap_sorted * sizeof(uint64_t), stream));   // sort padding
        {
            ScopedKernelTimer tm(K_SCATTER, stream);
            ScanInScatter sis{};
            if (spec && scan_in_scatter) {
                sis.dense_total = img.tile_cursor; sis.block_sums = geom.block_sums; sis.nblocks = nblocks; sis.ranges = img.ranges; sis.block_base = geom.block_base;
                sis.chunk_base = img.chunk_base; sis.flags = prefiltered ? flags : (const uint32_t*)nullptr; sis.cap_R = (uint32_t)cap; sis.cap_tile_list = cap_tile;
                sis.host_mailbox = t_use_mailbox ? t_mailbox_dev : nullptr; sis.seq = t_seq;
            }
            hipLaunchKernelGGL(scatter_instances_kernel, dim3(nblocks + (order_items || sis.dense_total ? 1 : 0)), dim3(GB), hist_lds_bytes, stream, P, gx, gy, radii, geom.rec,
                               geom.tiles_touched, geom.block_base, geom.point_offsets, img.tile_cursor, img.ranges,
                               lds_hist ? img.block_tile_base : nullptr, bin.keys, bin.inst_gauss, geom.header, spec ? 1 : 0,
                               (uint32_t)carve_R, (uint32_t)cap_sorted, eager, (raw && raw->flow_proj1) ? t_clip_single : (const int*)nullptr,
                               order_items ? img.tile_count : (uint32_t*)nullptr, order_fwd_tiles(T, lds_hist), sis);
        }
        GSR_STAGE("scatter_instances");
        if (!t_fuse_sort || long_lists) {   // lists of up to SORT_SMALL_CAP entries are sorted inside render_fwd (fused); longer ones here
            ScopedKernelTimer tm(K_SORT, stream);
            if (!t_fuse_sort)
                hipLaunchKernelGGL((sort_tiles_kernel<SORT_SMALL_CAP, 0>), dim3(T), dim3(256), 0, stream, T, img.ranges, bin.keys,
                                   bin.inst_gauss, bin.sorted, chk);
            // dev knobs: GSR_LONG_FROM = list length above which the chunk + rank path is taken (default SORT_LDS_CAP; 1024 drops the
            // one-block-per-tile LDS sort of the 1024..4096 lists), GSR_LONG_CHUNK = its chunk size (1024 / 2048 / 4096)
            static const uint32_t long_from = getenv("GSR_LONG_FROM") ? (uint32_t)atoi(getenv("GSR_LONG_FROM")) : (uint32_t)SORT_LDS_CAP;
            static const int long_chunk = getenv("GSR_LONG_CHUNK") ? atoi(getenv("GSR_LONG_CHUNK")) : SORT_LDS_CAP;
            if (long_lists && long_from > (uint32_t)SORT_SMALL_CAP) {
                // one launch: 18 KiB blocks (eight per CU) when no list exceeds 2048 keys, 36 KiB blocks (four per CU) otherwise. Splitting
                // the tiles between two launches by length was measured: the two tails cost more than the occupancy returns (99 -> 131 us).
                if (longest_list <= (uint32_t)SORT_MID_CAP)
                    hipLaunchKernelGGL((sort_tiles_kernel<SORT_MID_CAP, SORT_SMALL_CAP>), dim3(T), dim3(256), 0, stream, T, img.ranges,
                                       bin.keys, bin.inst_gauss, bin.sorted, chk);
                else
                    hipLaunchKernelGGL((sort_tiles_kernel<SORT_LDS_CAP, SORT_SMALL_CAP>), dim3(T), dim3(256), 0, stream, T, img.ranges,
                                       bin.keys, bin.inst_gauss, bin.sorted, chk);
            }
            if (longest_list > long_from) {   // chunk-wise LDS sort + rank by counting (gs_forward.h F4b)
#define GSR_LONG(CK)                                                                                                                   \
    do {                                                                                                                              \
        const dim3 g((unsigned)T, (longest_list + (uint32_t)(CK) - 1) / (uint32_t)(CK));                                               \
        hipLaunchKernelGGL((sort_long_chunks_kernel<CK>), g, dim3(256), 0, stream, img.ranges, bin.keys, chk, long_from);              \
        hipLaunchKernelGGL((rank_long_chunks_kernel<CK>), g, dim3(256), 0, stream, img.ranges, (const uint64_t*)bin.keys,               \
                           (const uint32_t*)bin.inst_gauss, bin.sorted, chk, long_from);                                               \
    } while (0)
                if (long_chunk == 1024) GSR_LONG(1024); else if (long_chunk == 2048) GSR_LONG(2048); else GSR_LONG(4096);
#undef GSR_LONG
            }
        }
        GSR_STAGE("sort_tiles");
        {   // Tiles with an empty range still run and write the background (forward.cu:297-299,382-391; Q21).
            ScopedKernelTimer tm(K_RENDER_FWD, stream);
            if (t_track_loss)
                hipLaunchKernelGGL(render_fwd_track_kernel, dim3(T), dim3(RB), 0, stream, T, gx, img.ranges, bin.sorted, width, height, geom.rec,
                                   background, img.final_T, img.n_contrib, out_color, out_depth,
                                   out_opacity, n_touched, img.final_C, bin.ckpt, chk, t_fuse_sort ? (const uint64_t*)bin.keys : nullptr,
                                   (const uint32_t*)bin.inst_gauss, bin.sorted, (const uint32_t*)img.chunk_base, bin.chunk_info,
                                   order_items ? (const uint32_t*)img.tile_count : (const uint32_t*)nullptr, order_fwd_tiles(T, lds_hist) ? 1 : 0, *t_track_loss);
            else
            hipLaunchKernelGGL(render_fwd_kernel, dim3(T), dim3(RB), 0, stream, T, gx, img.ranges, bin.sorted, width, height, geom.rec,
                               background, img.final_T, img.n_contrib, out_color, out_depth,
                               out_opacity, n_touched, img.final_C, bin.ckpt, chk, t_fuse_sort ? (const uint64_t*)bin.keys : nullptr,
                               (const uint32_t*)bin.inst_gauss, bin.sorted, (const uint32_t*)img.chunk_base, bin.chunk_info,
                               order_items ? (const uint32_t*)img.tile_count : (const uint32_t*)nullptr, order_fwd_tiles(T, lds_hist) ? 1 : 0);
        }
        GSR_STAGE("render_fwd");
        return 0;
    };

    char* bchunk = nullptr;
    if (speculate) {
        bchunk = binning_alloc(binning_user, binning_bytes(cap, cap, (size_t)T));
        if (!bchunk) { g_last_error = "gsr_forward: binning allocation callback returned NULL"; return GSR_ERR_ALLOC; }
        const int rc = enqueue_binning_and_render(bchunk, cap, cap, true, false, cap_tile);
        if (rc) return rc;
    }
    if (speculate && t_lazy) return (int)(cap > 0x7fffffffull ? 0x7fffffffull : cap);   // no wait: see t_lazy
    // The one host wait of the forward pass (the reference's is the blocking cudaMemcpy at rasterizer_impl.cu:283-284).
    uint32_t hdr[4];
    { const int rc = wait_for_header(stream, geom.header, t_seq, hdr); if (rc) return rc; }
    const uint32_t R = hdr[HDR_R], flg = hdr[HDR_FLAGS], R_alloc = hdr[HDR_R_ALLOC], max_tile_list = hdr[HDR_MAX_TILE];
    if (flg & FLAG_PREFILTERED) { g_last_error = "Point is filtered although prefiltered is set. This shouldn't happen!"; return GSR_ERR_PREFILTERED; }
    if (R > 0x7fffffffu || R_alloc > 0x7fffffffu) { g_last_error = "gsr_forward: more than 2^31 instances"; return GSR_ERR_INVALID_ARGUMENT; }
    t_last_R_alloc = std::max<size_t>(1, R_alloc > R ? R_alloc : R);     // (never 0: 0 means "no estimate yet"; a frame may legitimately have no instance)
    t_last_max_tile = max_tile_list;
    if (speculate && !(flg & FLAG_OVERFLOW)) return (int)R;

    if (R > 0) {
        bchunk = binning_alloc(binning_user, binning_bytes((size_t)R, (size_t)R_alloc, (size_t)T));
        if (!bchunk) { g_last_error = "gsr_forward: binning allocation callback returned NULL"; return GSR_ERR_ALLOC; }
        const int rc = enqueue_binning_and_render(bchunk, (size_t)R, (size_t)R_alloc, false, R_alloc != R, max_tile_list);
        if (rc) return rc;
    } else {
        if (!bchunk) bchunk = binning_alloc(binning_user, binning_bytes(0, 0, (size_t)T));
        if (!bchunk) { g_last_error = "gsr_forward: binning allocation callback returned NULL"; return GSR_ERR_ALLOC; }
        // keep point_offsets defined for debug readers / backward even when nothing is visible
        if (P > 0) GSR_HIP_CHECK(hipMemsetAsync(geom.point_offsets, 0, (size_t)P * sizeof(uint32_t), stream));
        ScopedKernelTimer tm(K_RENDER_FWD, stream);
        if (t_track_loss)
            hipLaunchKernelGGL(render_fwd_track_kernel, dim3(T), dim3(RB), 0, stream, T, gx, img.ranges, (const uint2*)nullptr, width, height,
                               geom.rec, background, img.final_T, img.n_contrib, out_color,
                               out_depth, out_opacity, n_touched, img.final_C, (float*)nullptr, (const uint32_t*)nullptr,
                               (const uint64_t*)nullptr, (const uint32_t*)nullptr, (uint2*)nullptr, (const uint32_t*)nullptr, (uint4*)nullptr,
                               (const uint32_t*)nullptr, 0, *t_track_loss);
        else
        hipLaunchKernelGGL(render_fwd_kernel, dim3(T), dim3(RB), 0, stream, T, gx, img.ranges, (const uint2*)nullptr, width, height,
                           geom.rec, background, img.final_T, img.n_contrib, out_color,
                           out_depth, out_opacity, n_touched, img.final_C, (float*)nullptr, (const uint32_t*)nullptr,
                           (const uint64_t*)nullptr, (const uint32_t*)nullptr, (uint2*)nullptr, (const uint32_t*)nullptr, (uint4*)nullptr,
                           (const uint32_t*)nullptr, 0);
    }
    GSR_STAGE("render_fwd");
    return (int)R;
}

extern "C" {

int gsr_forward(gsr_alloc_fn geometry_alloc, void* geometry_user, gsr_alloc_fn binning_alloc, void* binning_user,
                gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
                const float* means3D, const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                float* out_depth, float* out_opacity, int* radii, int* n_touched, int debug, void* stream)
{
    return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background, width, height,
                        means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                        cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, out_opacity, radii, n_touched, debug, stream, nullptr);
}

int gsr_forward_raw(gsr_alloc_fn geometry_alloc, void* geometry_user, gsr_alloc_fn binning_alloc, void* binning_user,
                    gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
                    const gsr_raw_inputs* in, float scale_modifier, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                    float tan_fovx, float tan_fovy, float* out_color, float* out_depth, float* out_opacity, int* radii, int* n_touched,
                    int debug, void* stream)
{
    if (!in) { g_last_error = "gsr_forward_raw: null input descriptor"; return GSR_ERR_INVALID_ARGUMENT; }
    return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background, width, height,
                        nullptr, nullptr, nullptr, nullptr, nullptr, scale_modifier, nullptr, nullptr, viewmatrix, projmatrix, cam_pos,
                        tan_fovx, tan_fovy, 0, out_color, out_depth, out_opacity, radii, n_touched, debug, stream, in);
}

}  // extern "C"

// ---- multi-view entry point (include/gs_rasterizer.h, csrc/gs_views.h) ------------------------------------------------------------------
namespace {
struct CapturedAlloc { gsr_alloc_fn fn; void* user; char* got; };
char* captured_alloc(void* u, size_t bytes)
{
    CapturedAlloc* c = static_cast<CapturedAlloc*>(u);
    c->got = c->fn(c->user, bytes);
    return c->got;
}
ViewDims view_dims(int P, int width, int height)
{
    ViewDims d;
    d.P = P; d.W = width; d.H = height; d.gx = (width + TILE_X - 1) / TILE_X; d.gy = (height + TILE_Y - 1) / TILE_Y; d.T = d.gx * d.gy;
    d.nblocks = (P + GB - 1) / GB;
    return d;
}
}  // namespace

static int forward_one_view(gsr_view& vw, int slot, gsr_alloc_fn geometry_alloc, gsr_alloc_fn binning_alloc, gsr_alloc_fn image_alloc, int P, int D, int M,
                            const float* background, int width, int height, const gsr_raw_inputs* in, float scale_modifier, float tan_fovx, float tan_fovy,
                            int debug, void* stream)
{
    gsr_raw_inputs one = *in;
    one.dx = vw.dx; one.ds = vw.ds; one.dr = vw.dr;
    one.flow_dx2 = vw.flow_dx2; one.flow_proj1 = vw.flow_proj1; one.flow_proj2 = vw.flow_proj2;
    CapturedAlloc g{geometry_alloc, vw.geometry_user, nullptr}, b{binning_alloc, vw.binning_user, nullptr}, i{image_alloc, vw.image_user, nullptr};
    const int rc = forward_impl(captured_alloc, &g, captured_alloc, &b, captured_alloc, &i, P, D, M, background, width, height, nullptr, nullptr, nullptr, nullptr,
                                nullptr, scale_modifier, nullptr, nullptr, vw.viewmatrix, vw.projmatrix, vw.cam_pos, tan_fovx, tan_fovy, 0, vw.out_color,
                                vw.out_depth, vw.out_opacity, vw.radii, vw.n_touched, debug, stream, &one, slot);
    if (rc < 0) return rc;
    vw.geom_buffer = g.got; vw.binning_buffer = b.got; vw.image_buffer = i.got; vw.num_rendered = rc;
    return 0;
}

extern "C" int gsr_forward_views(int V, gsr_view* views, gsr_alloc_fn geometry_alloc, gsr_alloc_fn binning_alloc, gsr_alloc_fn image_alloc, int P, int D, int M,
                                 const float* background, int width, int height, const gsr_raw_inputs* in, float scale_modifier, float tan_fovx,
                                 float tan_fovy, int debug, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (V < 1 || V > MAX_VIEWS || !views || !in || P <= 0 || width <= 0 || height <= 0 || !geometry_alloc || !binning_alloc || !image_alloc || !background ||
        M <= 0 || D < 0 || D > 3 || (D + 1) * (D + 1) > M || in->gather || in->flow_proj1) {
        g_last_error = "gsr_forward_views: invalid argument (1 <= V <= GSR_MAX_VIEWS, P > 0, raw inputs without gather; flow mode is per view)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const bool flow = views[0].flow_proj1 != nullptr;      // render_flow views: every view of the call, or none
    for (int v = 0; v < V; v++) {
        const gsr_view& w = views[v];
        gsr_raw_inputs probe = *in; probe.dx = w.dx; probe.ds = w.ds; probe.dr = w.dr;
        probe.flow_dx2 = w.flow_dx2; probe.flow_proj1 = w.flow_proj1; probe.flow_proj2 = w.flow_proj2;
        if (!w.viewmatrix || !w.projmatrix || !w.cam_pos || !w.out_color || !w.out_depth || !w.out_opacity || !w.radii || !w.n_touched ||
            (w.flow_proj1 != nullptr) != flow || !raw_inputs_ok(&probe, M)) {
            g_last_error = "gsr_forward_views: null / inconsistent view descriptor"; return GSR_ERR_INVALID_ARGUMENT;
        }
    }
    // flow batches keep their own capacity estimates (their v-th view is another camera), and so does every call of an iteration that needs
    // several (view_slot_group): a slot shared by two calls per iteration sees its estimate flip between two cameras -- eager calls then redo
    // the view through the single-view path every time, captured ones overflow at every replay
    const int slot0 = 1 + t_view_slot_group * 2 * MAX_VIEWS + (flow ? MAX_VIEWS : 0);
    const int* clips[MAX_VIEWS];                             // a flow view's tile rectangle: gsr_view.flow_clip
    for (int v = 0; v < MAX_VIEWS; v++) clips[v] = (flow && v < V) ? views[v].flow_clip : nullptr;
    const ViewDims d = view_dims(P, width, height);
    read_option_env();
    // the batched path needs a capacity estimate for every slot (the first iteration of a window goes view by view and leaves one)
    const int dev = select_device_state(0);
    bool batched = !debug && t_speculate && use_lds_hist((size_t)d.T) && V > 1;
    for (int v = 0; v < V && batched; v++) batched = t_spec[dev][slot0 + v].last_R_alloc != 0;
    if (!batched) {
        for (int v = 0; v < V; v++) {
            t_clip_single = clips[v];
            const int rc = forward_one_view(views[v], slot0 + v, geometry_alloc, binning_alloc, image_alloc, P, D, M, background, width, height, in, scale_modifier,
                                            tan_fovx, tan_fovy, debug, stream_);
            t_clip_single = nullptr;
            if (rc) return rc;
        }
        return 0;
    }
    static const int eager_max = getenv("GSR_EAGER_MAX") ? atoi(getenv("GSR_EAGER_MAX")) : 512 * 1024;
    t_views_batched++;
    ViewTable t;
    memset(&t, 0, sizeof(t));
    uint32_t want_tile = 0;
    for (int v = 0; v < V; v++) want_tile = std::max(want_tile, t_spec[dev][slot0 + v].last_max_tile);
    want_tile += (uint32_t)((unsigned long long)want_tile * tile_margin() / 1000ull);
    const uint32_t cap_tile = want_tile <= (uint32_t)SORT_SMALL_CAP ? (uint32_t)SORT_SMALL_CAP
                            : want_tile <= (uint32_t)SORT_MID_CAP ? (uint32_t)SORT_MID_CAP
                            : want_tile <= (uint32_t)SORT_LDS_CAP ? (uint32_t)SORT_LDS_CAP
                            : (want_tile + (uint32_t)SORT_LDS_CAP - 1) / (uint32_t)SORT_LDS_CAP * (uint32_t)SORT_LDS_CAP;
    for (int v = 0; v < V; v++) {
        gsr_view& w = views[v];
        select_device_state(slot0 + v);
        { const int rc = ensure_mailbox(); if (rc) return rc; }
        if (++t_seq == 0) t_seq = 1;
        const size_t cap = spec_capacity(t_last_R_alloc);
        w.geom_buffer = geometry_alloc(w.geometry_user, gsr_geometry_buffer_size(P));
        w.image_buffer = image_alloc(w.image_user, gsr_image_buffer_size(width, height, P));
        w.binning_buffer = binning_alloc(w.binning_user, binning_bytes(cap, cap, (size_t)d.T));
        if (!w.geom_buffer || !w.image_buffer || !w.binning_buffer) { g_last_error = "gsr_forward_views: allocation callback returned NULL"; return GSR_ERR_ALLOC; }
        ViewSlot& s = t.v[v];
        s.viewmatrix = w.viewmatrix; s.projmatrix = w.projmatrix; s.projmatrix_raw = w.projmatrix_raw; s.cam_pos = w.cam_pos;
        s.dx = w.dx; s.ds = w.ds; s.dr = w.dr;
        s.flow_dx2 = w.flow_dx2; s.flow_proj1 = w.flow_proj1; s.flow_proj2 = w.flow_proj2; s.flow_clip = clips[v];
        s.geom = w.geom_buffer; s.image = w.image_buffer; s.binning = w.binning_buffer;
        s.out_color = w.out_color; s.out_depth = w.out_depth; s.out_opacity = w.out_opacity; s.radii = w.radii; s.n_touched = w.n_touched;
        s.mailbox = t_use_mailbox ? t_mailbox_dev : nullptr; s.cap = (uint32_t)std::min<size_t>(cap, 0x7fffffffu); s.cap_tile = cap_tile; s.seq = t_seq;
    }
    PreprocessArgs a;
    memset(&a, 0, sizeof(a));
    a.P = P; a.D = D; a.M = M; a.W = width; a.H = height; a.gx = d.gx; a.gy = d.gy;
    a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
    a.focal_y = height / (2.0f * tan_fovy); a.focal_x = width / (2.0f * tan_fovx);
    a.prefiltered = 0; a.eager = P <= eager_max ? 1 : 0;
    a.raw = to_device_view(in);
    const dim3 gv((unsigned)d.nblocks, (unsigned)V), tv((unsigned)d.T, (unsigned)V);
    const size_t hist_lds_bytes = (size_t)d.T * sizeof(uint32_t);
    {
        ScopedKernelTimer tm(K_PREPROCESS, stream);
        if (in->delta_mode) hipLaunchKernelGGL((preprocess_views_kernel<true, true>), gv, dim3(GB), hist_lds_bytes, stream, a, t, d);
        else hipLaunchKernelGGL(preprocess_views_kernel<true>, gv, dim3(GB), hist_lds_bytes, stream, a, t, d);
    }
    {
        ScopedKernelTimer tm(K_SCAN, stream);
        const dim3 go((unsigned)((d.T + TO_COLS - 1) / TO_COLS), (unsigned)V);
        if (d.nblocks <= TO_SEGS * 16) hipLaunchKernelGGL((tile_offsets_views_kernel<TO_SEGS, 16>), go, dim3(TO_COLS * TO_SEGS), 0, stream, t, d);
        else hipLaunchKernelGGL((tile_offsets_views_kernel<TO_SEGS_BIG, 32>), go, dim3(TO_COLS * TO_SEGS_BIG), 0, stream, t, d);
        hipLaunchKernelGGL(scan_views_kernel, dim3(1, (unsigned)V), dim3(1024), 0, stream, t, d);
    }
    {
        ScopedKernelTimer tm(K_SCATTER, stream);
        hipLaunchKernelGGL(scatter_views_kernel, dim3((unsigned)d.nblocks + (t_order_items ? 1u : 0u), (unsigned)V), dim3(GB), hist_lds_bytes, stream, t, d, a.eager,
                           t_order_items ? (order_fwd_tiles(d.T, true) ? 3 + 4 * (order_fwd_tiles(d.T, true) - 1) : 1) : 0);
    }
    if (!t_fuse_sort || cap_tile > (uint32_t)SORT_SMALL_CAP) {
        ScopedKernelTimer tm(K_SORT, stream);
        if (!t_fuse_sort) hipLaunchKernelGGL((sort_tiles_views_kernel<SORT_SMALL_CAP, 0>), tv, dim3(256), 0, stream, t, d);
        if (cap_tile > (uint32_t)SORT_SMALL_CAP) {
            if (cap_tile <= (uint32_t)SORT_MID_CAP) hipLaunchKernelGGL((sort_tiles_views_kernel<SORT_MID_CAP, SORT_SMALL_CAP>), tv, dim3(256), 0, stream, t, d);
            else hipLaunchKernelGGL((sort_tiles_views_kernel<SORT_LDS_CAP, SORT_SMALL_CAP>), tv, dim3(256), 0, stream, t, d);
        }
        if (cap_tile > (uint32_t)SORT_LDS_CAP) {
            const dim3 gl((unsigned)d.T, (cap_tile + (uint32_t)SORT_LDS_CAP - 1) / (uint32_t)SORT_LDS_CAP, (unsigned)V);
            hipLaunchKernelGGL((sort_long_chunks_views_kernel<SORT_LDS_CAP>), gl, dim3(256), 0, stream, t, d, (uint32_t)SORT_LDS_CAP);
            hipLaunchKernelGGL((rank_long_chunks_views_kernel<SORT_LDS_CAP>), gl, dim3(256), 0, stream, t, d, (uint32_t)SORT_LDS_CAP);
        }
    }
    {
        ScopedKernelTimer tm(K_RENDER_FWD, stream);
        hipLaunchKernelGGL(render_fwd_views_kernel, tv, dim3(RB), 0, stream, t, d, background, t_fuse_sort ? 1 : 0, t_order_items ? (order_fwd_tiles(d.T, true) ? 3 + 4 * (order_fwd_tiles(d.T, true) - 1) : 1) : 0);
    }
    GSR_HIP_CHECK(hipGetLastError());
    // one wait per view (they are all long done by the time the host has enqueued the tile kernels); a view that outgrew its capacity is
    // redone through the single-view path, which allocates exactly
    for (int v = 0; v < V; v++) {
        gsr_view& w = views[v];
        select_device_state(slot0 + v);
        if (t_lazy) { w.num_rendered = (int)t.v[v].cap; continue; }
        uint32_t hdr[4];
        char* gp = w.geom_buffer;
        const GeomState geom = GeomState::from(gp, (size_t)P);
        { const int rc = wait_for_header(stream, geom.header, t.v[v].seq, hdr); if (rc) return rc; }
        const uint32_t R = hdr[HDR_R], flg = hdr[HDR_FLAGS], R_alloc = hdr[HDR_R_ALLOC];
        if (R > 0x7fffffffu || R_alloc > 0x7fffffffu) { g_last_error = "gsr_forward_views: more than 2^31 instances"; return GSR_ERR_INVALID_ARGUMENT; }
        t_last_R_alloc = std::max<size_t>(1, R_alloc > R ? R_alloc : R);
        t_last_max_tile = hdr[HDR_MAX_TILE];
        w.num_rendered = (int)R;
        if (flg & FLAG_OVERFLOW) {
            t_clip_single = clips[v];
            const int rc = forward_one_view(w, slot0 + v, geometry_alloc, binning_alloc, image_alloc, P, D, M, background, width, height, in, scale_modifier, tan_fovx,
                                            tan_fovy, debug, stream_);
            t_clip_single = nullptr;
            if (rc) return rc;
        }
    }
    return 0;
}

extern "C" size_t gsr_views_scratch_size(int V, int P, int M, int scale_dim)
{
    return (size_t)V * (part_layout((size_t)P, M, scale_dim).total * sizeof(float) + 256) + 256;
}

extern "C" int gsr_backward_views(int V, gsr_view* views, int P, int D, int M, const float* background, int width, int height, const gsr_raw_inputs* in,
                                  float scale_modifier, float tan_fovx, float tan_fovy, const gsr_raw_grads* out, char* scratch, int debug, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    const bool accumulate = (debug & GSR_BACKWARD_ACCUMULATE) != 0, pose_only = (debug & GSR_BACKWARD_POSE_ONLY) != 0;
    const bool flow = views && V >= 1 && views[0].flow_proj1 != nullptr;      // flow views: only out->xyz is required (constants otherwise, :307,326-334)
    if (V < 1 || V > MAX_VIEWS || !views || !in || P <= 0 || width <= 0 || height <= 0 || !background || in->gather || in->flow_proj1 ||
        (!pose_only && (!out || !scratch || !out->xyz ||
                        (!flow && (!out->log_scales || !out->raw_rotations || !out->logit_opacity || !out->features_dc || (M > 1 && !out->features_rest)))))) {
        g_last_error = "gsr_backward_views: invalid argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const ViewDims d = view_dims(P, width, height);
    ViewTable t;
    memset(&t, 0, sizeof(t));
    const PartLayout L = part_layout((size_t)P, M, in->scale_dim);
    const size_t row_bytes = (L.total * sizeof(float) + 255) & ~size_t(255);
    char* sp = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(scratch) + 255) & ~uintptr_t(255));
    int max_R = 0;
    for (int v = 0; v < V; v++) {
        const gsr_view& w = views[v];
        if (!w.geom_buffer || !w.binning_buffer || !w.image_buffer || !w.dL_dcolor || !w.dL_ddepth || !w.dL_dmean2D || !w.viewmatrix || !w.projmatrix ||
            !w.projmatrix_raw || !w.cam_pos || !w.radii || (w.flow_proj1 != nullptr) != flow || (flow && !w.flow_proj2)) {
            g_last_error = "gsr_backward_views: null / inconsistent view argument"; return GSR_ERR_INVALID_ARGUMENT;
        }
        ViewSlot& s = t.v[v];
        s.viewmatrix = w.viewmatrix; s.projmatrix = w.projmatrix; s.projmatrix_raw = w.projmatrix_raw; s.cam_pos = w.cam_pos;
        s.dx = w.dx; s.ds = w.ds; s.dr = w.dr;
        s.flow_dx2 = w.flow_dx2; s.flow_proj1 = w.flow_proj1; s.flow_proj2 = w.flow_proj2; s.ddx2 = w.ddx2;
        s.geom = w.geom_buffer; s.image = w.image_buffer; s.binning = w.binning_buffer; s.radii = w.radii;
        s.dL_dpix = w.dL_dcolor; s.dL_dpix_depth = w.dL_ddepth; s.dL_dmean2D = w.dL_dmean2D; s.ddx = w.ddx; s.dds = w.dds; s.ddr = w.ddr; s.tau_sum = w.dL_dtau_sum;
        s.part = pose_only ? nullptr : reinterpret_cast<float*>(sp + (size_t)v * row_bytes);
        s.cap = (uint32_t)std::max(0, w.num_rendered);
        max_R = std::max(max_R, w.num_rendered);
    }
    {
        ScopedKernelTimer tm(K_RENDER_BWD, stream);
        if (max_R > 0) {
            hipLaunchKernelGGL(render_bwd_views_kernel, dim3((unsigned)(max_R / CHUNK + d.T), (unsigned)V), dim3(RB), 0, stream, t, d, background);
        }
    }
    GeomBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.P = P; a.D = D; a.M = M; a.W = width; a.H = height; a.scale_modifier = scale_modifier;
    a.focal_y = height / (2.0f * tan_fovy); a.focal_x = width / (2.0f * tan_fovx); a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
    a.pose_only = pose_only ? 1 : 0;
    a.sh_rows = t_sh_rows ? 1 : 0;
    a.raw = to_device_view(in);
    a.rawg = RawGrads{};
    a.rawg.scale_dim = in->scale_dim;
    {
        ScopedKernelTimer tm(K_GEOM_BWD, stream);
        if (in->delta_mode) hipLaunchKernelGGL((geometry_bwd_views_kernel<true, true>), dim3((unsigned)((P + 255) / 256), (unsigned)V), dim3(256), 0, stream, a, t, d);
        else hipLaunchKernelGGL(geometry_bwd_views_kernel<true>, dim3((unsigned)((P + 255) / 256), (unsigned)V), dim3(256), 0, stream, a, t, d);
        hipLaunchKernelGGL(tau_sum_views_kernel, dim3(1, (unsigned)V), dim3(384), 0, stream, t, d);
        if (!pose_only) {
            ReduceTargets r{out->xyz, out->features_dc, out->features_rest, out->logit_opacity, out->log_scales, out->raw_rotations};
            hipLaunchKernelGGL(views_reduce_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, V, t, P, M, in->scale_dim, r, accumulate ? 1 : 0);
        }
    }
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

static int backward_impl(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* projmatrix_raw,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                 char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth, float* dL_dmean2D, float* dL_dconic,
                 float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscale, float* dL_drot, float* dL_dtau, float* dL_dtau_sum, int debug, void* stream_,
                 const gsr_raw_inputs* raw, const gsr_raw_grads* rawg)
{
    hipStream_t stream = (hipStream_t)stream_;
    const bool accumulate = (debug & GSR_BACKWARD_ACCUMULATE) != 0;     // include/gs_rasterizer.h
    const bool pose_only = (debug & GSR_BACKWARD_POSE_ONLY) != 0;
    debug &= 1;
    if (P < 0 || R < 0 || width <= 0 || height <= 0) { g_last_error = "gsr_backward: invalid size"; return GSR_ERR_INVALID_ARGUMENT; }
    if (P == 0) { if (dL_dtau_sum) GSR_HIP_CHECK(hipMemsetAsync(dL_dtau_sum, 0, 6 * sizeof(float), stream)); return 0; }
    // Intermediate gradients the caller does not want may be NULL (dL_dconic, dL_dcolor, dL_ddepth, dL_dcov3D; dL_dtau when
    // dL_dtau_sum is given): the kernel then keeps them in registers only. gsr_backward itself requires all of them.
    if (!geom_buffer || !binning_buffer || !image_buffer || !dL_dpix || !dL_dpix_depth || !background || (!means3D && !raw) || !viewmatrix ||
        !projmatrix || !projmatrix_raw || !campos || !dL_dmean2D || (!pose_only && (!dL_dopacity || !dL_dmean3D)) || (!dL_dtau && !dL_dtau_sum) ||
        (raw && (!raw_inputs_ok(raw, M) || !rawg ||
                 (!pose_only && (!dL_dscale || !dL_drot || (!raw->flow_proj1 && (!rawg->features_dc || (M > 1 && !rawg->features_rest)))))))) {
        g_last_error = "gsr_backward: null argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const int gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y, T = gx * gy;
    char* gp = geom_buffer; char* ip = image_buffer;
    GeomState geom = GeomState::from(gp, (size_t)P);
    ImageState img = ImageState::from(ip, (size_t)width * height, (size_t)T, 0);
    if (!radii) radii = geom.internal_radii;   // rasterizer_impl.cu:387-390
    if (R > 0) {
        // one block per CHUNK entries of a tile list; sum over tiles of ceil(n / CHUNK) <= R / CHUNK + T, surplus blocks exit
        ScopedKernelTimer tm(K_RENDER_BWD, stream);
        const int grid = R / CHUNK + T;
        hipLaunchKernelGGL(render_bwd_kernel, dim3(grid), dim3(RB), 0, stream, T, gx, (const char*)binning_buffer,
                           (const uint32_t*)geom.header, width, height, background, geom.rec,
                           img.final_T, img.final_C, img.n_contrib, dL_dpix, dL_dpix_depth);
    }
    GSR_STAGE("render_bwd");
    GeomBwdArgs a;
    a.P = P; a.D = D; a.M = M; a.W = width; a.H = height;
    a.means3D = means3D; a.radii = radii; a.shs = shs; a.clamped = geom.clamped; a.scales = scales; a.rotations = rotations;
    a.scale_modifier = scale_modifier; a.cov3Ds = cov3D_precomp ? cov3D_precomp : geom.cov3D;   // rasterizer_impl.cu:429
    a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.projmatrix_raw = projmatrix_raw; a.campos = campos;
    a.focal_y = height / (2.0f * tan_fovy); a.focal_x = width / (2.0f * tan_fovx); a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
    a.tiles_touched = geom.tiles_touched; a.point_offsets = geom.point_offsets; a.bin_base = binning_buffer; a.header = geom.header;
    a.dL_dmean2D = dL_dmean2D; a.dL_dconic = dL_dconic; a.dL_dopacity = dL_dopacity; a.dL_dcolor = dL_dcolor; a.dL_ddepth = dL_ddepth;
    a.dL_dmean3D = dL_dmean3D; a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscale = dL_dscale; a.dL_drot = dL_drot; a.dL_dtau = dL_dtau;
    a.accumulate = accumulate ? 1 : 0;
    a.pose_only = pose_only ? 1 : 0;
    a.sh_rows = t_sh_rows ? 1 : 0;
    a.tau_partials = dL_dtau_sum ? geom.tau_partials : nullptr;
    a.raw = to_device_view(raw);
    a.rawg = RawGrads{};
    if (raw) { a.rawg.f_dc = rawg->features_dc; a.rawg.f_rest = rawg->features_rest; a.rawg.ddx = rawg->dx; a.rawg.dds = rawg->ds; a.rawg.ddr = rawg->dr; a.rawg.scale_dim = raw->scale_dim; a.rawg.ddx2 = rawg->dx2; }
    {
        ScopedKernelTimer tm(K_GEOM_BWD, stream);
        if (raw && raw->delta_mode) hipLaunchKernelGGL((geometry_bwd_kernel<true, true>), dim3((P + 255) / 256), dim3(256), 0, stream, a);
        else if (raw) hipLaunchKernelGGL(geometry_bwd_kernel<true>, dim3((P + 255) / 256), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(geometry_bwd_kernel<false>, dim3((P + 255) / 256), dim3(256), 0, stream, a);
        if (dL_dtau_sum && t_track_tail)     // gsr_track_step: pose-gradient sum + exposure-gradient sum + camera step in one launch
            hipLaunchKernelGGL(track_tail_kernel, dim3(1), dim3(384), 0, stream, (P + 255) / 256, (const float*)geom.tau_partials, dL_dtau_sum, T,
                               t_track_tail->exposure_partials, t_track_tail->dL_dexposure, t_track_tail->step);
        else if (dL_dtau_sum)
            hipLaunchKernelGGL(tau_sum_kernel, dim3(1), dim3(384), 0, stream, (P + 255) / 256, geom.tau_partials, dL_dtau_sum);
    }
    GSR_STAGE("geometry_bwd");
    return 0;
}

// ---- HexPlane feature field (include/deformation_field.h) -------------------------------------------------------------------
static int hexplane_check(const gsr_hexplane_field* f, int64_t n, const float* xyz, const float* time, const char* who)
{
    static thread_local std::string msg;
    auto fail = [&](const char* what) { msg = std::string(who) + ": " + what; g_last_error = msg.c_str(); return GSR_ERR_INVALID_ARGUMENT; };
    if (!f) return fail("null field descriptor");
    if (n < 0) return fail("negative point count");
    if (f->num_levels < 1 || f->num_levels > GSR_HEXPLANE_MAX_LEVELS) return fail("num_levels outside 1..8");
    if (f->feat_dim != 8 && f->feat_dim != 16 && f->feat_dim != 32 && f->feat_dim != 64) return fail("feat_dim must be 8, 16, 32 or 64");
    for (int l = 0; l < f->num_levels; l++) {
        for (int k = 0; k < 4; k++) if (f->levels[l].res[k] < 1) return fail("resolution < 1");
        for (int p = 0; p < 6; p++) if (!f->levels[l].planes[p]) return fail("null plane");
    }
    if (n > 0 && (!xyz || !time)) return fail("null xyz / time");
    return 0;
}

template <bool BWD, typename... Args>
static void hexplane_launch(const gsr_hexplane_field& f, int64_t n, hipStream_t stream, Args... args)
{
    const int lpp = f.feat_dim / 4;
    if constexpr (BWD) {   // channels-last planes: one channel per lane, a whole texel per atomic instruction (gs_hexplane.h)
        if (f.channels_last) {
            const int ppb = HEX_BLOCK / f.feat_dim;
            const dim3 g((unsigned)((n + ppb - 1) / ppb)), b(HEX_BLOCK);
            switch (f.feat_dim) {
            case 8: hipLaunchKernelGGL((hexplane_bwd_lane_kernel<8>), g, b, 0, stream, f, n, args...); break;
            case 16: hipLaunchKernelGGL((hexplane_bwd_lane_kernel<16>), g, b, 0, stream, f, n, args...); break;
            case 32: hipLaunchKernelGGL((hexplane_bwd_lane_kernel<32>), g, b, 0, stream, f, n, args...); break;
            case 64: hipLaunchKernelGGL((hexplane_bwd_lane_kernel<64>), g, b, 0, stream, f, n, args...); break;
            }
            return;
        }
    }
    const dim3 grid((unsigned)((n + HEX_BLOCK / lpp - 1) / (HEX_BLOCK / lpp))), block(HEX_BLOCK);
#define GSR_HEX_CASE(LPP)                                                                                                       \
    case LPP:                                                                                                                   \
        if constexpr (BWD) { if (f.channels_last) hipLaunchKernelGGL((hexplane_bwd_kernel<LPP, HEX_VEC>), grid, block, 0, stream, f, n, args...);   \
                   else hipLaunchKernelGGL((hexplane_bwd_kernel<LPP, HEX_PLANAR>), grid, block, 0, stream, f, n, args...); }        \
        else     { if (f.channels_last) hipLaunchKernelGGL((hexplane_fwd_kernel<LPP, HEX_VEC>), grid, block, 0, stream, f, n, args...);   \
                   else hipLaunchKernelGGL((hexplane_fwd_kernel<LPP, HEX_PLANAR>), grid, block, 0, stream, f, n, args...); }        \
        break;
    switch (lpp) { GSR_HEX_CASE(2) GSR_HEX_CASE(4) GSR_HEX_CASE(8) GSR_HEX_CASE(16) }
#undef GSR_HEX_CASE
}


extern "C" {

int gsr_backward_fused(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* projmatrix_raw,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                 char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth, float* dL_dmean2D, float* dL_dconic,
                 float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscale, float* dL_drot, float* dL_dtau, float* dL_dtau_sum, int debug, void* stream)
{
    return backward_impl(P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp,
                         viewmatrix, projmatrix, projmatrix_raw, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer,
                         dL_dpix, dL_dpix_depth, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh,
                         dL_dscale, dL_drot, dL_dtau, dL_dtau_sum, debug, stream, nullptr, nullptr);
}

int gsr_backward_raw(int P, int D, int M, int R, const float* background, int width, int height, const gsr_raw_inputs* in,
                     float scale_modifier, const float* viewmatrix, const float* projmatrix, const float* projmatrix_raw,
                     const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                     char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth, float* dL_dmean2D, const gsr_raw_grads* out,
                     float* dL_dtau_sum, int debug, void* stream)
{
    if (!in || !out) { g_last_error = "gsr_backward_raw: null descriptor"; return GSR_ERR_INVALID_ARGUMENT; }
    return backward_impl(P, D, M, R, background, width, height, nullptr, nullptr, nullptr, nullptr, scale_modifier, nullptr, nullptr, viewmatrix,
                         projmatrix, projmatrix_raw, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix,
                         dL_dpix_depth, dL_dmean2D, nullptr, out->logit_opacity, nullptr, nullptr, out->xyz, nullptr, nullptr,
                         out->log_scales, out->raw_rotations, nullptr, dL_dtau_sum, debug, stream, in, out);
}

int gsr_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* projmatrix_raw,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                 char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth, float* dL_dmean2D, float* dL_dconic,
                 float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscale, float* dL_drot, float* dL_dtau, int debug, void* stream)
{
    if (P > 0 && (!dL_dconic || !dL_dcolor || !dL_ddepth || !dL_dcov3D || !dL_dtau)) { g_last_error = "gsr_backward: null argument"; return GSR_ERR_INVALID_ARGUMENT; }
    return gsr_backward_fused(P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                              cov3D_precomp, viewmatrix, projmatrix, projmatrix_raw, campos, tan_fovx, tan_fovy, radii, geom_buffer,
                              binning_buffer, image_buffer, dL_dpix, dL_dpix_depth, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth,
                              dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, dL_dtau, nullptr, debug, stream);
}

#if GSR_TIMELINE
// dev builds only (-DGSR_TIMELINE=1): {start, end (100 MHz ticks), HW_ID, XCC_ID} of every block of the last render_fwd (which = 0, by tile) /
// render_bwd (which = 1, by block) launch
int gsr_debug_spans(unsigned int* out, int nwords, int which)
{
    GSR_HIP_CHECK(hipDeviceSynchronize());
    GSR_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spans), (size_t)nwords * sizeof(uint32_t), (size_t)which * 8192 * 4 * sizeof(uint32_t)));
    return 0;
}
#endif
#if GSR_FWD_TIMING
// dev builds only (-DGSR_FWD_TIMING=1): per-wave cycle accounting of the last render_fwd launch, 8 words per (tile, quadrant wave)
int gsr_debug_fwd_timing(unsigned int* out, int nwords)
{
    GSR_HIP_CHECK(hipDeviceSynchronize());
    GSR_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_timing), (size_t)nwords * sizeof(uint32_t)));
    return 0;
}
int gsr_debug_pre_timing(unsigned int* out, int nwords, int which)
{
    GSR_HIP_CHECK(hipDeviceSynchronize());
    if (which == 0) GSR_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pre_timing), (size_t)nwords * sizeof(uint32_t)));
    else GSR_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sca_timing), (size_t)nwords * sizeof(uint32_t)));
    return 0;
}
int gsr_debug_geo_timing(unsigned int* out, int nwords)
{
    GSR_HIP_CHECK(hipDeviceSynchronize());
    GSR_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_geo_timing), (size_t)nwords * sizeof(uint32_t)));
    return 0;
}
int gsr_debug_bwd_timing(unsigned int* out, int nwords)
{
    GSR_HIP_CHECK(hipDeviceSynchronize());
    GSR_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_timing), (size_t)nwords * sizeof(uint32_t)));
    return 0;
}
#endif
int gsr_debug_read_state(int P, int R, int width, int height, const char* geom_buffer, const char* binning_buffer,
                         const char* image_buffer, float* depths, float* means2D, float* conic_opacity, float* rgb, float* cov3D,
                         unsigned char* clamped, uint32_t* tiles_touched, uint32_t* point_offsets, float* final_T, uint32_t* n_contrib,
                         uint32_t* ranges, uint32_t* point_list, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    GSR_HIP_CHECK(hipStreamSynchronize(stream));
    const int gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y, T = gx * gy;
    const size_t N = (size_t)width * height;
    char* gp = const_cast<char*>(geom_buffer); char* bp = const_cast<char*>(binning_buffer); char* ip = const_cast<char*>(image_buffer);
    GeomState geom = GeomState::from(gp, (size_t)P);
    ImageState img = ImageState::from(ip, N, (size_t)T, 0);
    uint32_t hdr[HDR_WORDS] = {0};
    if (P > 0) GSR_HIP_CHECK(hipMemcpy(hdr, geom.header, sizeof(hdr), hipMemcpyDeviceToHost));
    const BinningPtrs bin = carve_binning(bp, hdr[HDR_CARVE_R], hdr[HDR_CAP_SORTED], (size_t)T);
#define D2H(dst, src, bytes) do { if ((dst) && (bytes)) GSR_HIP_CHECK(hipMemcpy((dst), (src), (bytes), hipMemcpyDeviceToHost)); } while (0)
    if (P && (depths || means2D || conic_opacity || rgb)) {   // the packed per-Gaussian rows, handed out in the reference's four arrays
        std::vector<TileRec> rows((size_t)P);
        GSR_HIP_CHECK(hipMemcpy(rows.data(), geom.rec, (size_t)P * sizeof(TileRec), hipMemcpyDeviceToHost));
        for (int i = 0; i < P; i++) {
            const TileRec& r = rows[i];
            if (depths) depths[i] = r.q0.z;
            if (means2D) { means2D[2 * i] = r.q0.x; means2D[2 * i + 1] = r.q0.y; }
            if (conic_opacity) { conic_opacity[4 * i] = r.q1.x; conic_opacity[4 * i + 1] = r.q1.y; conic_opacity[4 * i + 2] = r.q1.z; conic_opacity[4 * i + 3] = r.q0.w; }
            if (rgb) { rgb[3 * i] = r.q2.x; rgb[3 * i + 1] = r.q2.y; rgb[3 * i + 2] = r.q2.z; }
        }
    }
    D2H(cov3D, geom.cov3D, P * 6 * sizeof(float));
    if (clamped && P) {
        std::vector<uint8_t> bits(P);
        GSR_HIP_CHECK(hipMemcpy(bits.data(), geom.clamped, P, hipMemcpyDeviceToHost));
        for (int i = 0; i < P; i++) for (int k = 0; k < 3; k++) clamped[3 * i + k] = (bits[i] >> k) & 1;
    }
    D2H(tiles_touched, geom.tiles_touched, P * sizeof(uint32_t));
    D2H(point_offsets, geom.point_offsets, P * sizeof(uint32_t));
    D2H(final_T, img.final_T, N * sizeof(float));
    D2H(n_contrib, img.n_contrib, N * sizeof(uint32_t));
    std::vector<uint32_t> rg((size_t)T * 2);
    GSR_HIP_CHECK(hipMemcpy(rg.data(), img.ranges, (size_t)T * 8, hipMemcpyDeviceToHost));
    // Padded segments (huge tiles) are compacted so the caller sees the reference's packed layout.
    std::vector<uint32_t> packed((size_t)T * 2);
    uint32_t run = 0, maxend = 0;
    for (int t = 0; t < T; t++) {
        const uint32_t n = rg[2 * t + 1] - rg[2 * t];
        packed[2 * t] = n ? run : 0; packed[2 * t + 1] = n ? run + n : 0;   // empty tiles stay (0,0) like the memset at rasterizer_impl.cu:313
        run += n;
        if (rg[2 * t + 1] > maxend) maxend = rg[2 * t + 1];
    }
    if (ranges) memcpy(ranges, packed.data(), (size_t)T * 8);
    if (point_list && R > 0) {
        std::vector<uint2> srt(maxend);
        GSR_HIP_CHECK(hipMemcpy(srt.data(), bin.sorted, (size_t)maxend * sizeof(uint2), hipMemcpyDeviceToHost));
        size_t o = 0;
        for (int t = 0; t < T; t++)
            for (uint32_t k = rg[2 * t]; k < rg[2 * t + 1]; k++) point_list[o++] = srt[k].x;
    }
#undef D2H
    return 0;
}


// ---- simple_knn.h ---------------------------------------------------------------------------------------------------
struct KnnWorkspace {
    KnnGrid* grid; uint32_t* cell_of; uint32_t* cell_count; uint32_t* cell_start; uint32_t* cursor; float4* sorted_pts;
    static KnnWorkspace from(char*& p, size_t P, size_t C)
    {
        KnnWorkspace w;
        carve(p, w.grid, 1); carve(p, w.cell_of, P); carve(p, w.cell_count, C); carve(p, w.cell_start, C + 1); carve(p, w.cursor, C);
        carve(p, w.sorted_pts, P);
        return w;
    }
};
static inline size_t knn_max_cells(int P) { return (size_t)(P < 64 ? 64 : P); }

size_t gsr_knn_workspace_size(int P)
{
    return required([&](char*& p) { KnnWorkspace::from(p, (size_t)(P > 0 ? P : 1), knn_max_cells(P)); });
}

int gsr_knn_mean_dist2(int P, const float* points, float* mean_dists, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || (P > 0 && (!points || !mean_dists || !workspace))) { g_last_error = "gsr_knn_mean_dist2: null argument"; return GSR_ERR_INVALID_ARGUMENT; }
    if (P == 0) return 0;
    const size_t C = knn_max_cells(P);
    char* wp = workspace;
    KnnWorkspace w = KnnWorkspace::from(wp, (size_t)P, C);
    ScopedKernelTimer tm(K_KNN, stream);
    GSR_HIP_CHECK(hipMemsetAsync(w.grid, 0xFF, 3 * sizeof(uint32_t), stream));                       // bbox min = ~0
    GSR_HIP_CHECK(hipMemsetAsync(reinterpret_cast<char*>(w.grid) + 12, 0x00, 3 * sizeof(uint32_t), stream));   // bbox max = 0
    GSR_HIP_CHECK(hipMemsetAsync(w.cell_count, 0, C * sizeof(uint32_t), stream));
    const int nb = (P + 255) / 256;
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, stream, P, points, w.grid);
    hipLaunchKernelGGL(knn_setup_kernel, dim3(1), dim3(1), 0, stream, P, (int)C, w.grid);
    hipLaunchKernelGGL(knn_count_kernel, dim3(nb), dim3(256), 0, stream, P, points, w.grid, w.cell_of, w.cell_count);
    hipLaunchKernelGGL(knn_scan_kernel, dim3(1), dim3(1024), 0, stream, w.grid, w.cell_count, w.cell_start, w.cursor);
    hipLaunchKernelGGL(knn_scatter_kernel, dim3(nb), dim3(256), 0, stream, P, points, w.cell_of, w.cursor, w.sorted_pts);
    hipLaunchKernelGGL(knn_query_kernel, dim3(nb), dim3(256), 0, stream, P, w.grid, w.cell_start, w.sorted_pts, mean_dists);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}


// Test hook: runs gsr::wave_sum10_transposed on one wave. in: device float[64][10]; out: device float[64] (what each lane holds).
__global__ void debug_reduce10_kernel(const float* in, float* out)
{
    const int l = threadIdx.x;
    const float* v = in + l * 10;
    unsigned long long proc = 0; uint32_t addr;
    out[l] = wave_sum10_transposed(wave_select_masks(), v[0], f2v{v[1], v[2]}, f2v{v[3], v[4]}, v[5], f2v{v[6], v[7]}, f2v{v[8], v[9]}, proc, 0, 0u, 0, addr);
}
// the work-item map of gs_device.h on the host (no device needed): block index of the full piece of rank `rank` (partial = 0) or of the
// partial piece of rank `rank` (partial = 1) in a frame of n_items pieces, n_partial of them partial
int gsr_debug_item_block(unsigned int n_items, unsigned int n_partial, unsigned int rank, int partial)
{
    if (n_partial > n_items || rank >= (partial ? n_partial : n_items - n_partial)) return GSR_ERR_INVALID_ARGUMENT;
    return (int)(partial ? item_block_partial(n_items, n_partial, rank) : item_block_full(n_items, n_partial, rank));
}
int gsr_debug_wave_reduce10(const float* in, float* out, void* stream_)
{
    hipLaunchKernelGGL(debug_reduce10_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, in, out);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}


// ---- fused L1 loss (include/slam_losses.h) ------------------------------------------------------------------------
size_t gsr_l1_loss_workspace_size(void) { return (size_t)LOSS_BLOCKS * 2 * sizeof(float); }

static LossArgs make_loss_args(int width, int height, const float* image, const float* depth, const float* gt_image, const float* gt_depth,
                               const float* w_rgb, const float* w_depth, const float* exposure_a, const float* exposure_b, float alpha,
                               const float* opacity, float opacity_thr)
{
    LossArgs a;
    a.opacity = opacity; a.opacity_thr = opacity_thr;
    a.N = width * height; a.image = image; a.depth = depth; a.gt_image = gt_image; a.gt_depth = gt_depth; a.w_rgb = w_rgb; a.w_depth = w_depth;
    a.exposure_a = exposure_a; a.exposure_b = exposure_b;
    a.c_rgb = alpha / (3.0f * (float)a.N); a.c_depth = (1.0f - alpha) / (float)a.N;
    return a;
}

int gsr_l1_loss_forward(int width, int height, const float* image, const float* depth, const float* gt_image, const float* gt_depth,
                        const float* w_rgb, const float* w_depth, const float* exposure_a, const float* exposure_b, float alpha,
                        const float* opacity, float opacity_depth_threshold, float* loss, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (width <= 0 || height <= 0 || !image || !depth || !gt_image || !gt_depth || !loss || !workspace) {
        g_last_error = "gsr_l1_loss_forward: null / invalid argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const LossArgs a = make_loss_args(width, height, image, depth, gt_image, gt_depth, w_rgb, w_depth, exposure_a, exposure_b, alpha, opacity,
                                      opacity_depth_threshold);
    float* partials = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(l1_loss_fwd_kernel, dim3(LOSS_BLOCKS), dim3(LOSS_THREADS), 0, stream, a, partials);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, stream, (const float*)partials, 1, loss);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_l1_loss_backward(int width, int height, const float* image, const float* depth, const float* gt_image, const float* gt_depth,
                         const float* w_rgb, const float* w_depth, const float* exposure_a, const float* exposure_b, float alpha,
                         const float* opacity, float opacity_depth_threshold, const float* upstream, float* dL_dimage, float* dL_ddepth, float* dL_dexposure, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (width <= 0 || height <= 0 || !image || !depth || !gt_image || !gt_depth || !dL_dimage || !dL_ddepth || !workspace) {
        g_last_error = "gsr_l1_loss_backward: null / invalid argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const LossArgs a = make_loss_args(width, height, image, depth, gt_image, gt_depth, w_rgb, w_depth, exposure_a, exposure_b, alpha, opacity,
                                      opacity_depth_threshold);
    float* partials = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(l1_loss_bwd_kernel, dim3(LOSS_BLOCKS), dim3(LOSS_THREADS), 0, stream, a, upstream, dL_dimage, dL_ddepth, partials);
    if (dL_dexposure) hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, stream, (const float*)partials, 2, dL_dexposure);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}


static int masked_l1_args(MaskedL1Args& a, int n_terms, const gsr_masked_l1_term* terms, int width, int height, int channels, int image_channels,
                          float scale, bool backward, const char* who)
{
    static thread_local std::string msg;
    bool ok = n_terms >= 1 && n_terms <= MASKED_L1_MAX_TERMS && terms && width > 0 && height > 0 && channels >= 1 && channels <= image_channels;
    for (int t = 0; ok && t < n_terms; t++) ok = terms[t].image && terms[t].target && terms[t].mask && (!backward || terms[t].dL_dimage);
    if (!ok) { msg = std::string(who) + ": null / invalid argument (1 <= n_terms <= 4, 1 <= channels <= image_channels)"; g_last_error = msg.c_str(); return GSR_ERR_INVALID_ARGUMENT; }
    a = MaskedL1Args{};
    a.n_terms = n_terms; a.N = width * height; a.C = channels; a.Cimg = image_channels;
    for (int t = 0; t < n_terms; t++) { a.image[t] = terms[t].image; a.target[t] = terms[t].target; a.mask[t] = terms[t].mask; a.dL_dimage[t] = terms[t].dL_dimage; }
    a.coeff = scale / ((float)channels * (float)a.N);
    return 0;
}

int gsr_masked_l1_forward(int n_terms, const gsr_masked_l1_term* terms, int width, int height, int channels, int image_channels, float scale,
                          float* loss, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    MaskedL1Args a;
    const int rc = masked_l1_args(a, n_terms, terms, width, height, channels, image_channels, scale, false, "gsr_masked_l1_forward");
    if (rc < 0) return rc;
    if (!loss || !workspace) { g_last_error = "gsr_masked_l1_forward: null loss / workspace"; return GSR_ERR_INVALID_ARGUMENT; }
    float* partials = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(masked_l1_fwd_kernel, dim3(LOSS_BLOCKS), dim3(LOSS_THREADS), 0, stream, a, partials);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, stream, (const float*)partials, 1, loss);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_masked_l1_backward(int n_terms, const gsr_masked_l1_term* terms, int width, int height, int channels, int image_channels, float scale,
                           const float* upstream, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    MaskedL1Args a;
    const int rc = masked_l1_args(a, n_terms, terms, width, height, channels, image_channels, scale, true, "gsr_masked_l1_backward");
    if (rc < 0) return rc;
    hipLaunchKernelGGL(masked_l1_bwd_kernel, dim3(LOSS_BLOCKS, (unsigned)n_terms), dim3(LOSS_THREADS), 0, stream, a, upstream);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- fused Adam (include/slam_losses.h) ---------------------------------------------------------------------------
void gsr_adam_coefficients(double lr, double beta1, double beta2, int step, float out[2])
{
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    out[0] = (float)(lr / bc1); out[1] = (float)(1.0 / sqrt(bc2));
}

static int adam_step_impl(int nseg, const gsr_adam_segment* segs, const float* coefficients, bool scheduled, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (nseg < 0 || nseg > ADAM_MAX_SEGMENTS || (nseg > 0 && !segs)) { g_last_error = "gsr_adam_step: 0..32 segments"; return GSR_ERR_INVALID_ARGUMENT; }
    if (scheduled && nseg > 0 && !coefficients) { g_last_error = "gsr_adam_step_scheduled: null coefficients"; return GSR_ERR_INVALID_ARGUMENT; }
    AdamArgs a;
    a.nseg = nseg; a.total = 0; a.coef = scheduled ? coefficients : nullptr;
    for (int k = 0; k < nseg; k++) {
        const gsr_adam_segment& h = segs[k];
        if (h.n && (!h.param || !h.grad || !h.exp_avg || !h.exp_avg_sq)) { g_last_error = "gsr_adam_step: null pointer"; return GSR_ERR_INVALID_ARGUMENT; }
        if (!scheduled && h.step < 1) { g_last_error = "gsr_adam_step: step counts from 1"; return GSR_ERR_INVALID_ARGUMENT; }
        a.start[k] = a.total;
        AdamSegment& d = a.seg[k];
        d.param = h.param; d.grad = h.grad; d.exp_avg = h.exp_avg; d.exp_avg_sq = h.exp_avg_sq; d.n = h.n;
        float c[2] = {0.f, 0.f};
        if (!scheduled) gsr_adam_coefficients((double)h.lr, h.beta1_d, h.beta2_d, h.step, c);
        d.step_size = c[0]; d.inv_bc2_sqrt = c[1]; d.eps = h.eps; d.beta2 = h.beta2;
        d.one_minus_beta1 = (float)(1.0 - h.beta1_d); d.one_minus_beta2 = (float)(1.0 - h.beta2_d);
        a.total += h.n;
    }
    a.start[nseg] = a.total;
    if (a.total == 0) return 0;
    const unsigned long long blocks = (a.total + 255) / 256;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, a);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}
int gsr_adam_step(int nseg, const gsr_adam_segment* segs, void* stream) { return adam_step_impl(nseg, segs, nullptr, false, stream); }
int gsr_adam_step_device_count(int nseg, const gsr_adam_segment* segs, float* const* step_counts, float* coefficients, void* stream_)
{
    if (nseg < 0 || nseg > ADAM_MAX_SEGMENTS || (nseg > 0 && (!segs || !step_counts || !coefficients))) {
        g_last_error = "gsr_adam_step_device_count: 0..32 segments, no null arguments"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (nseg == 0) return 0;
    AdamDeviceSteps d;
    d.nseg = nseg;
    for (int k = 0; k < nseg; k++) {
        if (!step_counts[k]) { g_last_error = "gsr_adam_step_device_count: null step count"; return GSR_ERR_INVALID_ARGUMENT; }
        d.step[k] = step_counts[k]; d.lr[k] = segs[k].lr; d.beta1[k] = segs[k].beta1_d; d.beta2[k] = segs[k].beta2_d;
    }
    hipLaunchKernelGGL(adam_device_coefficients_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, d, coefficients);
    GSR_HIP_CHECK(hipGetLastError());
    return adam_step_impl(nseg, segs, coefficients, true, stream_);
}
int gsr_adam_step_scheduled(int nseg, const gsr_adam_segment* segs, const float* coefficients, void* stream)
{
    return adam_step_impl(nseg, segs, coefficients, true, stream);
}


// ---- fused SSIM (include/slam_losses.h) ---------------------------------------------------------------------------
static SsimWindow ssim_window()
{
    SsimWindow w;                                          // loss_utils.py:46-53: exp(-(x - 5)^2 / (2 * 1.5^2)), normalised
    double sum = 0.0, g[SSIM_WIN];
    for (int k = 0; k < SSIM_WIN; k++) { g[k] = exp(-(double)((k - SSIM_R) * (k - SSIM_R)) / (2.0 * 1.5 * 1.5)); sum += g[k]; }
    for (int k = 0; k < SSIM_WIN; k++) w.w[k] = (float)((float)g[k] / (float)sum);
    return w;
}
static inline dim3 ssim_grid(int width, int height, int channels)
{
    return dim3((width + SSIM_T - 1) / SSIM_T, (height + SSIM_T - 1) / SSIM_T, channels);
}
size_t gsr_ssim_workspace_size(int width, int height, int channels)
{
    const dim3 g = ssim_grid(width, height, channels);
    return (size_t)g.x * g.y * g.z * sizeof(float) + 256 + (size_t)3 * channels * width * height * sizeof(float);
}

int gsr_ssim_forward(int width, int height, int channels, const float* img1, const float* img2, const unsigned char* mask, float* ssim_mean,
                     char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (width <= 0 || height <= 0 || channels <= 0 || !img1 || !img2 || !ssim_mean || !workspace) {
        g_last_error = "gsr_ssim_forward: null / invalid argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const dim3 g = ssim_grid(width, height, channels);
    const int nblocks = (int)(g.x * g.y * g.z);
    float* partials = reinterpret_cast<float*>(workspace);
    float* dmaps = reinterpret_cast<float*>(workspace + (((size_t)nblocks * sizeof(float) + 255) & ~(size_t)255));
    hipLaunchKernelGGL(ssim_fwd_kernel, g, dim3(SSIM_T * SSIM_T), 0, stream, width, height, img1, img2, mask, ssim_window(), dmaps, partials);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, stream, nblocks, (const float*)partials,
                       1.0f / ((float)channels * (float)width * (float)height), ssim_mean);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_ssim_backward(int width, int height, int channels, const float* img1, const float* img2, const unsigned char* mask,
                      const float* upstream, float* dL_dimg1, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (width <= 0 || height <= 0 || channels <= 0 || !img1 || !img2 || !dL_dimg1 || !workspace) {
        g_last_error = "gsr_ssim_backward: null / invalid argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const dim3 g = ssim_grid(width, height, channels);
    const int nblocks = (int)(g.x * g.y * g.z);
    const float* dmaps = reinterpret_cast<const float*>(workspace + (((size_t)nblocks * sizeof(float) + 255) & ~(size_t)255));
    hipLaunchKernelGGL(ssim_bwd_kernel, g, dim3(SSIM_T * SSIM_T), 0, stream, width, height, img1, img2, mask, ssim_window(), dmaps, upstream,
                       1.0f / ((float)channels * (float)width * (float)height), dL_dimg1);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}


int gsr_densification_stats(int P, const int* radii, const float* grad_mean2D, float* max_radii2D, float* xyz_gradient_accum, float* denom,
                            void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || (P > 0 && (!radii || !grad_mean2D || !max_radii2D || !xyz_gradient_accum || !denom))) {
        g_last_error = "gsr_densification_stats: null / invalid argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return 0;
    hipLaunchKernelGGL(densification_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, radii, grad_mean2D, max_radii2D,
                       xyz_gradient_accum, denom);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}


int gsr_hexplane_forward(const gsr_hexplane_field* field, int64_t n, const float* xyz, int64_t xyz_stride, const float* time,
                         int64_t time_stride, float* features, void* stream_)
{
    if (int rc = hexplane_check(field, n, xyz, time, "gsr_hexplane_forward")) return rc;
    if (n == 0) return 0;
    if (!features) { g_last_error = "gsr_hexplane_forward: null features"; return GSR_ERR_INVALID_ARGUMENT; }
    hexplane_launch<false>(*field, n, (hipStream_t)stream_, xyz, xyz_stride, time, time_stride, features);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

static void hexsort_plan(const gsr_hexplane_field& f, HexSortPlan* P)
{
    for (int k = 0; k < 4; k++) {
        int r = 1;
        for (int l = 0; l < f.num_levels; l++) r = std::max(r, (int)f.levels[l].res[k]);
        P->fine[k] = r;
        int b = 0;
        while ((1 << b) < r) b++;
        P->bits[k] = b;
    }
    static const int C0[6] = {0, 0, 0, 1, 1, 2}, C1[6] = {1, 2, 3, 2, 3, 3};   // itertools.combinations(range(4), 2)
    int off = 0;
    for (int pl = 0; pl < 6; pl++) {
        // every call of the reference passes ONE time for all points (gaussian_renderer/__init__.py:112): the time families then use a
        // single row of cells, n / 512 points per cell -- give them 16 sub-counters per cell (the spatial families: 2), up to 2^20 counters
        const int cell_bits = P->bits[C0[pl]] + P->bits[C1[pl]];
        P->sub_bits[pl] = std::max(0, std::min(C1[pl] == 3 ? 4 : 1, 20 - cell_bits));
        P->key_off[pl] = off;
        off += 1 << (cell_bits + P->sub_bits[pl]);
    }
    P->key_off[6] = off;
}

// Ordered mode (HexOrd): where the planes' fixed-point sums live. `plane_mask`: the planes the generic walk scatters into (all six, or the three
// spatial ones of the batched-views path, whose time families keep column sums per view instead: V > 0). Returns the elements of acc / acc_t.
static void hexord_plan(const gsr_hexplane_field& f, int64_t n, int V, int plane_mask, HexOrd* o, size_t* acc_elems, size_t* acct_elems)
{
    static const int C0[6] = {0, 0, 0, 1, 1, 2}, C1[6] = {1, 2, 3, 2, 3, 3};
    HexOrd h{};
    size_t off = 0;
    for (int l = 0; l < f.num_levels; l++)
        for (int pl = 0; pl < 6; pl++) {
            h.off[l][pl] = off;
            if ((plane_mask >> pl) & 1) off +=          // (whether or not the plane's gradient is asked for: the size must not depend on it)
                (size_t)f.levels[l].res[C0[pl]] * f.levels[l].res[C1[pl]] * f.feat_dim;
        }
    size_t rows = 0;
    if (V > 0)
        for (int j = 0; j < 3; j++) {
            uint32_t cols = 0;
            for (int l = 0; l < f.num_levels; l++) { h.tcol[j][l] = cols; cols += (uint32_t)f.levels[l].res[j]; }
            h.tcols[j] = cols;
            h.tbase[j] = (uint32_t)rows;
            rows += (size_t)V * cols;
        }
    // a texel receives at most 4 n contributions (four cells, every point once: the views are summed before): the largest one is scaled to just
    // below 2^budget, the sum stays below 2^62; 40 bits at most, so that HEXSORT_CHUNK of them stay exact in a double
    int lg = 0;
    while (((int64_t)1 << lg) < 4 * std::max<int64_t>(n, 1)) lg++;
    h.budget = std::min(40, 62 - lg);
    if (o) *o = h;
    if (acc_elems) *acc_elems = off;
    if (acct_elems) *acct_elems = rows * f.feat_dim;
}

static size_t hexsort_carve(const gsr_hexplane_field& f, const HexSortPlan& P, int64_t n, char* base, HexSortWs* ws, HexOrd* ord = nullptr,
                            size_t* ord_bytes = nullptr)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
    const size_t nb = (size_t)P.key_off[6];
    HexSortWs w;
    w.count = reinterpret_cast<uint32_t*>(take(nb * sizeof(uint32_t)));
    w.block_sums = reinterpret_cast<uint32_t*>(take((nb / (1024 * HEXSORT_SCAN_ITEMS) + 1) * sizeof(uint32_t)));
    w.header = reinterpret_cast<uint32_t*>(take(256));
    w.coords = reinterpret_cast<float4*>(take((size_t)n * sizeof(float4)));
    w.key = reinterpret_cast<uint32_t*>(take((size_t)6 * n * sizeof(uint32_t)));
    w.rank = reinterpret_cast<int*>(take((size_t)6 * n * sizeof(int)));
    w.scoords = reinterpret_cast<float4*>(take((size_t)6 * n * sizeof(float4)));
    w.gs = reinterpret_cast<float*>(take((size_t)6 * n * f.num_levels * f.feat_dim * sizeof(float)));
    if (g_hex_ordered.load()) {
        HexOrd h;
        size_t elems;
        hexord_plan(f, n, 0, 63, &h, &elems, nullptr);
        h.acc = reinterpret_cast<unsigned long long*>(take(elems * sizeof(unsigned long long)));
        if (ord) *ord = h;
        if (ord_bytes) *ord_bytes = elems * sizeof(unsigned long long);
    } else if (ord) {
        *ord = HexOrd{};
    }
    if (ws) *ws = w;
    return off + 256;
}

static void hexord_convert(const gsr_hexplane_field& f, const HexOrd& ord, const uint32_t* header, int plane_mask, hipStream_t stream)
{
    size_t largest = 0;
    for (int l = 0; l < f.num_levels; l++)
        for (int a = 0; a < 4; a++)
            for (int b = a + 1; b < 4; b++) largest = std::max(largest, (size_t)f.levels[l].res[a] * f.levels[l].res[b] * f.feat_dim);
    hipLaunchKernelGGL(hexord_convert_kernel, dim3((unsigned)((largest + 511) / 512), (unsigned)(6 * f.num_levels)), dim3(256), 0, stream, f, ord, header,
                       plane_mask);
}

static bool hexsort_supported(const gsr_hexplane_field& f, int64_t n)
{
    if (!f.channels_last || n * 6 >= ((int64_t)1 << 31)) return false;
    for (int l = 0; l < f.num_levels; l++)
        for (int k = 0; k < 4; k++) if (f.levels[l].res[k] > 1024) return false;   // 2 x 10 key bits per family
    return true;
}

size_t gsr_hexplane_backward_workspace_size(const gsr_hexplane_field* field, int64_t n)
{
    if (!field || n <= 0 || field->num_levels < 1 || field->num_levels > GSR_HEXPLANE_MAX_LEVELS || !hexsort_supported(*field, n)) return 256;
    HexSortPlan P;
    hexsort_plan(*field, &P);
    return hexsort_carve(*field, P, n, nullptr, nullptr);
}

int gsr_hexplane_backward(const gsr_hexplane_field* field, int64_t n, const float* xyz, int64_t xyz_stride, const float* time,
                          int64_t time_stride, const float* dL_dfeatures, float* dL_dxyz, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = hexplane_check(field, n, xyz, time, "gsr_hexplane_backward")) return rc;
    if (n == 0) return 0;
    if (!dL_dfeatures) { g_last_error = "gsr_hexplane_backward: null dL_dfeatures"; return GSR_ERR_INVALID_ARGUMENT; }
    const gsr_hexplane_field& f = *field;
    if (!workspace || !hexsort_supported(f, n)) {
        hexplane_launch<true>(f, n, stream, xyz, xyz_stride, time, time_stride, dL_dfeatures, dL_dxyz);
        GSR_HIP_CHECK(hipGetLastError());
        return 0;
    }
    HexSortPlan P;
    hexsort_plan(f, &P);
    HexSortWs ws;
    HexOrd ord;
    size_t ord_bytes = 0;
    hexsort_carve(f, P, n, workspace, &ws, &ord, &ord_bytes);
    const int ordered = ord.acc != nullptr;
    const int nb = P.key_off[6], scan_blocks = (nb + 1024 * HEXSORT_SCAN_ITEMS - 1) / (1024 * HEXSORT_SCAN_ITEMS);
    GSR_HIP_CHECK(hipMemsetAsync(ws.count, 0, (size_t)nb * sizeof(uint32_t), stream));
    if (ordered) {
        GSR_HIP_CHECK(hipMemsetAsync(ws.header, 0, 256, stream));
        GSR_HIP_CHECK(hipMemsetAsync(ord.acc, 0, ord_bytes, stream));
    }
    const dim3 per_point((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(hexsort_count_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, stream, f, P, ws, n, xyz, xyz_stride, time, time_stride,
                       dL_dfeatures, (const uint32_t*)nullptr);
    hipLaunchKernelGGL(hexsort_scan_sums_kernel, dim3(scan_blocks), dim3(1024), 0, stream, (const uint32_t*)ws.count, nb, ws.block_sums);
    hipLaunchKernelGGL(hexsort_scan_top_kernel, dim3(1), dim3(1024), 0, stream, ws.block_sums, scan_blocks, ws.header);
    hipLaunchKernelGGL(hexsort_scan_apply_kernel, dim3(scan_blocks), dim3(1024), 0, stream, ws.count, nb, (const uint32_t*)ws.block_sums);
    hipLaunchKernelGGL(hexsort_scatter_kernel, per_point, dim3(256), 0, stream, ws, n);
    const int C = f.feat_dim, ppb = HEX_BLOCK / C;
    const dim3 g1((unsigned)((n + ppb - 1) / ppb));
    const int64_t groups = 6 * ((n + HEXSORT_CHUNK - 1) / HEXSORT_CHUNK);
    const dim3 g2((unsigned)((groups + 256 / C - 1) / (256 / C)));
#define GSR_HEXSORT_CASE(CC)                                                                                                              \
    case CC:                                                                                                                               \
        hipLaunchKernelGGL((hexsort_phase1_kernel<CC>), g1, dim3(HEX_BLOCK), 0, stream, f, ws, n, xyz, xyz_stride, time, time_stride,       \
                           dL_dfeatures, dL_dxyz, ordered);                                                                                 \
        if (f.num_levels <= 4) {                                                                                                           \
            if (ordered) hipLaunchKernelGGL((hexsort_phase2_kernel<CC, 4, true>), g2, dim3(256), 0, stream, f, ws, n, ord);                 \
            else hipLaunchKernelGGL((hexsort_phase2_kernel<CC, 4, false>), g2, dim3(256), 0, stream, f, ws, n, ord);                        \
        } else {                                                                                                                           \
            if (ordered) hipLaunchKernelGGL((hexsort_phase2_kernel<CC, GSR_HEXPLANE_MAX_LEVELS, true>), g2, dim3(256), 0, stream, f, ws, n, ord);   \
            else hipLaunchKernelGGL((hexsort_phase2_kernel<CC, GSR_HEXPLANE_MAX_LEVELS, false>), g2, dim3(256), 0, stream, f, ws, n, ord);          \
        }                                                                                                                                  \
        break;
    switch (C) { GSR_HEXSORT_CASE(8) GSR_HEXSORT_CASE(16) GSR_HEXSORT_CASE(32) GSR_HEXSORT_CASE(64) }
#undef GSR_HEXSORT_CASE
    if (ordered) hexord_convert(f, ord, ws.header, 63, stream);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- the views of one mapping iteration (include/deformation_field.h) ----------------------------------------------------------------
static int hexplane_views_check(const gsr_hexplane_field* f, int64_t n, const float* xyz, int V, const float* times, const char* who)
{
    static const float some_time = 0.f;
    if (int rc = hexplane_check(f, n, xyz, &some_time, who)) return rc;
    if (V < 1 || V > GSR_HEXPLANE_MAX_VIEWS || !times) {
        static thread_local std::string msg;
        msg = std::string(who) + ": 1 <= V <= GSR_HEXPLANE_MAX_VIEWS times (host pointer) expected"; g_last_error = msg.c_str();
        return GSR_ERR_INVALID_ARGUMENT;
    }
    return 0;
}
static HexTimes hex_times(int V, const float* times)
{
    HexTimes tv{};
    tv.V = V;
    for (int v = 0; v < V; v++) tv.t[v] = times[v];
    return tv;
}

int gsr_hexplane_forward_views(const gsr_hexplane_field* field, int64_t n, const float* xyz, int64_t xyz_stride, int V, const float* times,
                               float* features, void* stream_)
{
    if (int rc = hexplane_views_check(field, n, xyz, V, times, "gsr_hexplane_forward_views")) return rc;
    if (n == 0) return 0;
    if (!features) { g_last_error = "gsr_hexplane_forward_views: null features"; return GSR_ERR_INVALID_ARGUMENT; }
    const gsr_hexplane_field& f = *field;
    const HexTimes tv = hex_times(V, times);
    const int lpp = f.feat_dim / 4;
    const dim3 grid((unsigned)((n + HEX_BLOCK / lpp - 1) / (HEX_BLOCK / lpp))), block(HEX_BLOCK);
    hipStream_t stream = (hipStream_t)stream_;
#define GSR_HEXV_CASE(LPP)                                                                                                              \
    case LPP:                                                                                                                           \
        if (f.channels_last) hipLaunchKernelGGL((hexplane_fwd_views_kernel<LPP, HEX_VEC>), grid, block, 0, stream, f, n, xyz, xyz_stride, tv, features);   \
        else hipLaunchKernelGGL((hexplane_fwd_views_kernel<LPP, HEX_PLANAR>), grid, block, 0, stream, f, n, xyz, xyz_stride, tv, features);               \
        break;
    switch (lpp) { GSR_HEXV_CASE(2) GSR_HEXV_CASE(4) GSR_HEXV_CASE(8) GSR_HEXV_CASE(16) }
#undef GSR_HEXV_CASE
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

static size_t hexviews_carve(const gsr_hexplane_field& f, const HexSortPlan& P, int64_t n, int V, char* base, HexSortWs* ws, HexViewsWs* vw,
                             HexOrd* ord = nullptr, size_t* ord_bytes = nullptr)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
    const size_t nb = (size_t)P.key_off[6], row = (size_t)f.num_levels * f.feat_dim;
    HexSortWs w{};
    HexViewsWs v{};
    w.count = reinterpret_cast<uint32_t*>(take(nb * sizeof(uint32_t)));
    w.block_sums = reinterpret_cast<uint32_t*>(take((nb / (1024 * HEXSORT_SCAN_ITEMS) + 1) * sizeof(uint32_t)));
    w.header = reinterpret_cast<uint32_t*>(take(256));
    w.coords = reinterpret_cast<float4*>(take((size_t)n * sizeof(float4)));
    w.key = reinterpret_cast<uint32_t*>(take((size_t)6 * n * sizeof(uint32_t)));
    w.rank = reinterpret_cast<int*>(take((size_t)6 * n * sizeof(int)));
    w.scoords = reinterpret_cast<float4*>(take((size_t)6 * n * sizeof(float4)));
    v.gs_sp = reinterpret_cast<float*>(take((size_t)3 * n * row * sizeof(float)));
    v.gs_t = reinterpret_cast<float*>(take((size_t)3 * V * n * row * sizeof(float)));
    if (g_hex_ordered.load()) {                                     // one region (one memset): the spatial planes' sums, then the time families' column sums
        HexOrd h;
        size_t elems, telems;
        hexord_plan(f, n, V, (1 << 0) | (1 << 1) | (1 << 3), &h, &elems, &telems);
        h.acc = reinterpret_cast<unsigned long long*>(take((elems + telems) * sizeof(unsigned long long)));
        h.acc_t = h.acc ? h.acc + elems : nullptr;
        if (ord) *ord = h;
        if (ord_bytes) *ord_bytes = (elems + telems) * sizeof(unsigned long long);
    } else if (ord) {
        *ord = HexOrd{};
    }
    if (ws) *ws = w;
    if (vw) *vw = v;
    return off + 256;
}

static bool hexviews_supported(const gsr_hexplane_field& f, int64_t n)
{
    // (the time families' LDS window is [4 views][4 or 8 levels][16 columns][C] floats: 64 channels with more than four levels would be 128 KB)
    return hexsort_supported(f, n) && !(f.feat_dim == 64 && f.num_levels > 4);
}

size_t gsr_hexplane_backward_views_workspace_size(const gsr_hexplane_field* field, int64_t n, int V)
{
    if (!field || n <= 0 || V < 1 || V > GSR_HEXPLANE_MAX_VIEWS || field->num_levels < 1 || field->num_levels > GSR_HEXPLANE_MAX_LEVELS ||
        !hexviews_supported(*field, n)) return 0;
    HexSortPlan P;
    hexsort_plan(*field, &P);
    return hexviews_carve(*field, P, n, V, nullptr, nullptr, nullptr);
}

int gsr_hexplane_backward_views(const gsr_hexplane_field* field, int64_t n, const float* xyz, int64_t xyz_stride, int V, const float* times,
                                const float* dL_dfeatures, const uint32_t* view_mask, float* dL_dxyz, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = hexplane_views_check(field, n, xyz, V, times, "gsr_hexplane_backward_views")) return rc;
    if (n == 0) return 0;
    const gsr_hexplane_field& f = *field;
    if (!dL_dfeatures || !workspace || !hexviews_supported(f, n)) {
        g_last_error = "gsr_hexplane_backward_views: null cotangent / workspace, or a geometry the sorted algorithm does not cover (see gsr_hexplane_backward_views_workspace_size)";
        return GSR_ERR_INVALID_ARGUMENT;
    }
    const HexTimes tv = hex_times(V, times);
    HexSortPlan P;
    hexsort_plan(f, &P);
    HexSortWs ws;
    HexViewsWs vw;
    HexOrd ord;
    size_t ord_bytes = 0;
    hexviews_carve(f, P, n, V, workspace, &ws, &vw, &ord, &ord_bytes);
    const int ordered = ord.acc != nullptr;
    const int nb = P.key_off[6], scan_blocks = (nb + 1024 * HEXSORT_SCAN_ITEMS - 1) / (1024 * HEXSORT_SCAN_ITEMS);
    GSR_HIP_CHECK(hipMemsetAsync(ws.count, 0, (size_t)nb * sizeof(uint32_t), stream));
    if (ordered) {
        GSR_HIP_CHECK(hipMemsetAsync(ws.header, 0, 256, stream));
        GSR_HIP_CHECK(hipMemsetAsync(ord.acc, 0, ord_bytes, stream));
    }
    hipLaunchKernelGGL(hexsort_count_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, stream, f, P, ws, n, xyz, xyz_stride, (const float*)nullptr, (int64_t)0,
                       (const float*)nullptr, view_mask);
    hipLaunchKernelGGL(hexsort_scan_sums_kernel, dim3(scan_blocks), dim3(1024), 0, stream, (const uint32_t*)ws.count, nb, ws.block_sums);
    hipLaunchKernelGGL(hexsort_scan_top_kernel, dim3(1), dim3(1024), 0, stream, ws.block_sums, scan_blocks, ws.header);
    hipLaunchKernelGGL(hexsort_scan_apply_kernel, dim3(scan_blocks), dim3(1024), 0, stream, ws.count, nb, (const uint32_t*)ws.block_sums);
    hipLaunchKernelGGL(hexsort_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ws, n);
    const int C = f.feat_dim, ppb = HEX_BLOCK / C;
    const dim3 g1((unsigned)((n + ppb - 1) / ppb));
    const int dbg = getenv("GSR_HEXV_DEBUG") ? atoi(getenv("GSR_HEXV_DEBUG")) : 0;   // development: 1 = skip the spatial streams, 2 = skip the time streams
    const int gpw = C >= 64 ? 1 : 64 / C;                        // groups per wave (hexsort_phase2_views_kernel)
    const int64_t chunks2 = (n + HEXSORT_CHUNK - 1) / HEXSORT_CHUNK;
    const int64_t groups = (int64_t)3 * ((chunks2 + gpw - 1) / gpw) * gpw;
    const dim3 g2((unsigned)((groups + 256 / C - 1) / (256 / C)));
    const int per_block = (256 / C) * HEXT_ROUNDS;               // chunks per block of the time families' kernel
    const dim3 g3((unsigned)((chunks2 + per_block - 1) / per_block), (unsigned)(3 * ((V + HEXT_VB - 1) / HEXT_VB)));
#define GSR_HEXVB_CASE(CC)                                                                                                              \
    case CC:                                                                                                                             \
        hipLaunchKernelGGL((hexsort_phase1_views_kernel<CC>), g1, dim3(HEX_BLOCK), 0, stream, f, ws, vw, tv, n, xyz, xyz_stride, dL_dfeatures, dL_dxyz, ordered);   \
        GSR_HEXVB_PHASE2(CC, 4, f.num_levels <= 4)                                                                                       \
        GSR_HEXVB_PHASE2(CC, GSR_HEXPLANE_MAX_LEVELS, f.num_levels > 4)                                                                  \
        break;
#define GSR_HEXVB_PHASE2(CC, LM, when)                                                                                                  \
        if (when) {                                                                                                                     \
            if (ordered) {                                                                                                              \
                if (!(dbg & 1)) hipLaunchKernelGGL((hexsort_phase2_views_kernel<CC, LM, true>), g2, dim3(256), 0, stream, f, ws, vw, n, ord);        \
                if (!(dbg & 2)) hipLaunchKernelGGL((hexsort_phase2_time_kernel<CC, LM, true>), g3, dim3(256), 0, stream, f, ws, vw, tv, n, ord);     \
            } else {                                                                                                                    \
                if (!(dbg & 1)) hipLaunchKernelGGL((hexsort_phase2_views_kernel<CC, LM, false>), g2, dim3(256), 0, stream, f, ws, vw, n, ord);       \
                if (!(dbg & 2)) hipLaunchKernelGGL((hexsort_phase2_time_kernel<CC, LM, false>), g3, dim3(256), 0, stream, f, ws, vw, tv, n, ord);    \
            }                                                                                                                           \
        }
    switch (C) { GSR_HEXVB_CASE(8) GSR_HEXVB_CASE(16) GSR_HEXVB_CASE(32) GSR_HEXVB_CASE(64) }
#undef GSR_HEXVB_PHASE2
#undef GSR_HEXVB_CASE
    if (ordered) {
        hexord_convert(f, ord, ws.header, (1 << 0) | (1 << 1) | (1 << 3), stream);
        int wmax = 1;
        for (int l = 0; l < f.num_levels; l++)
            for (int k = 0; k < 3; k++) wmax = std::max(wmax, (int)f.levels[l].res[k]);
        hipLaunchKernelGGL(hexord_time_convert_kernel, dim3((unsigned)(((size_t)wmax * C + 255) / 256), (unsigned)(3 * f.num_levels)), dim3(256), 0, stream, f, tv,
                           ord, (const uint32_t*)ws.header);
    }
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- dense-layer weight gradient (include/deformation_field.h) -----------------------------------------------------------------
static void wgrad_plan(int64_t n, int64_t* chunk, int* nblocks)
{
    int64_t c = (n + 511) / 512;                                  // ~512 blocks along the reduction (2 per CU)
    c = ((c + 4 * WGRAD_UNROLL - 1) / (4 * WGRAD_UNROLL)) * (4 * WGRAD_UNROLL);
    if (c < 64) c = 64;
    *chunk = c;
    *nblocks = (int)((n + c - 1) / c);
}

size_t gsr_linear_wgrad_workspace_size(int64_t n, int in_dim, int out_dim)
{
    if (n <= 0 || in_dim <= 0 || out_dim <= 0) return 256;
    int64_t chunk; int nb;
    wgrad_plan(n, &chunk, &nb);
    return (size_t)nb * out_dim * (in_dim + 1) * sizeof(float) + 256;
}

int gsr_linear_wgrad(int64_t n, int in_dim, int out_dim, const float* x, int64_t x_stride, const float* dy, int64_t dy_stride,
                     float* dW, float* db, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || in_dim <= 0 || in_dim > 128 || out_dim <= 0 || (n > 0 && (!x || !dy || !workspace)) || (!dW && !db)) {
        g_last_error = "gsr_linear_wgrad: null / invalid argument (in_dim must be 1..128)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) {
        if (dW) GSR_HIP_CHECK(hipMemsetAsync(dW, 0, (size_t)out_dim * in_dim * sizeof(float), stream));
        if (db) GSR_HIP_CHECK(hipMemsetAsync(db, 0, (size_t)out_dim * sizeof(float), stream));
        return 0;
    }
    int64_t chunk; int nb;
    wgrad_plan(n, &chunk, &nb);
    float* partial = reinterpret_cast<float*>(workspace);
    const dim3 grid((unsigned)nb, (unsigned)((out_dim + 63) / 64)), block(WGRAD_BLOCK);
    const int nt = (in_dim + 15) / 16;
    switch (nt) {
#define GSR_WGRAD_CASE(NT) case NT: hipLaunchKernelGGL((linear_wgrad_kernel<NT>), grid, block, 0, stream, n, in_dim, out_dim, x, x_stride, dy, dy_stride, partial, chunk); break;
        GSR_WGRAD_CASE(1) GSR_WGRAD_CASE(2) GSR_WGRAD_CASE(3) GSR_WGRAD_CASE(4) GSR_WGRAD_CASE(5) GSR_WGRAD_CASE(6) GSR_WGRAD_CASE(7) GSR_WGRAD_CASE(8)
#undef GSR_WGRAD_CASE
    }
    const int per = out_dim * (in_dim + 1);
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((per + WRED_OUT - 1) / WRED_OUT), dim3(WRED_OUT * WRED_SLICES), 0, stream, nb, in_dim, out_dim,
                       (const float*)partial, dW, db);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- fused deformation MLP (include/deformation_field.h) ------------------------------------------------------------------------------
static int mlp_fill(const gsr_deform_mlp* m, MlpWeights* w, const char* who)
{
    static thread_local std::string msg;
    auto fail = [&](const char* what) { msg = std::string(who) + ": " + what; g_last_error = msg.c_str(); return GSR_ERR_INVALID_ARGUMENT; };
    if (!m) return fail("null descriptor");
    if (m->in_dim < 16 || m->in_dim > 128 || m->in_dim % 16) return fail("in_dim must be a multiple of 16 in 16..128");
    if (!m->W0 || !m->b0) return fail("null W0 / b0");
    static const int od[3] = {3, 3, 4}, oo[3] = {0, 3, 6};
    w->W0 = m->W0; w->b0 = m->b0; w->in_dim = m->in_dim;
    for (int j = 0; j < 3; j++) {
        if (!m->W1[j] || !m->b1[j] || !m->W2[j] || !m->b2[j]) return fail("null head weights");
        w->W1[j] = m->W1[j]; w->b1[j] = m->b1[j]; w->W2[j] = m->W2[j]; w->b2[j] = m->b2[j];
        w->out_dim[j] = od[j]; w->out_off[j] = oo[j];
    }
    return 0;
}

static dim3 mlp_grid(int64_t n)
{
    const int64_t tiles = (n + MLP_TILE - 1) / MLP_TILE;
    return dim3((unsigned)std::min<int64_t>(tiles, 256));        // persistent: one block per CU (the weights live in its LDS)
}

constexpr int MLP_BWD_BLOCKS = 256;      // persistent: one block per CU (103 KB of LDS each)

int gsr_deform_mlp_forward(const gsr_deform_mlp* mlp, int64_t n, const float* features, float* out, void* stream_)
{
    MlpWeights w;
    if (int rc = mlp_fill(mlp, &w, "gsr_deform_mlp_forward")) return rc;
    if (n < 0 || (n > 0 && (!features || !out))) { g_last_error = "gsr_deform_mlp_forward: null / invalid argument"; return GSR_ERR_INVALID_ARGUMENT; }
    if (n == 0) return 0;
    // round 6: the bf16-split kernel (gs_mlp.h: three terms per operand, six products, weights stationary in registers) for the widths the
    // shipped networks use; GSR_MLP_FP32=1: the fp32-MFMA kernel of rounds 1-5 (also every other width)
    static const bool fp32_only = getenv("GSR_MLP_FP32") && getenv("GSR_MLP_FP32")[0] == '1';
    static const int rt = getenv("GSR_MLP_RT") ? atoi(getenv("GSR_MLP_RT")) : 2;      // row tiles per block: 2 (two blocks per CU: 1.37 ms at 4 M rows) or 4 (one: 1.65)
    const int nt = w.in_dim / 16;
    if (!fp32_only && (nt == 2 || nt == 4 || nt == 8)) {
        static std::atomic<unsigned long long> attr_set[6];
#define GSR_MLP3_LAUNCH(NT, RT, SLOT)                                                                                                             \
        do {                                                                                                                                      \
            using L = Mlp3Layout<NT, RT>;                                                                                                         \
            { const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(deform_mlp_fwd3_kernel<NT, RT>), L::BYTES, attr_set[SLOT]); if (rc) return rc; } \
            const int64_t tiles = (n + L::R - 1) / L::R;                                                                                          \
            hipLaunchKernelGGL((deform_mlp_fwd3_kernel<NT, RT>), dim3((unsigned)std::min<int64_t>(tiles, 256 * (RT >= 4 ? 1 : 2))), dim3(MLP3_THREADS), L::BYTES, \
                               (hipStream_t)stream_, n, features, w, out);                                                                        \
        } while (0)
        if (rt == 2) { if (nt == 2) GSR_MLP3_LAUNCH(2, 2, 0); else if (nt == 4) GSR_MLP3_LAUNCH(4, 2, 1); else GSR_MLP3_LAUNCH(8, 2, 2); }
        else { if (nt == 2) GSR_MLP3_LAUNCH(2, 4, 3); else if (nt == 4) GSR_MLP3_LAUNCH(4, 4, 4); else GSR_MLP3_LAUNCH(8, 4, 5); }
#undef GSR_MLP3_LAUNCH
        GSR_HIP_CHECK(hipGetLastError());
        return 0;
    }
    switch (w.in_dim / 16) {
#define GSR_MLPF_CASE(NT) case NT: hipLaunchKernelGGL((deform_mlp_fwd_kernel<NT>), mlp_grid(n), dim3(MLP_BLOCK), 0, (hipStream_t)stream_, n, features, w, out); break;
        GSR_MLPF_CASE(1) GSR_MLPF_CASE(2) GSR_MLPF_CASE(3) GSR_MLPF_CASE(4) GSR_MLPF_CASE(5) GSR_MLPF_CASE(6) GSR_MLPF_CASE(7) GSR_MLPF_CASE(8)
#undef GSR_MLPF_CASE
    }
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t gsr_deform_mlp_grad_count(int in_dim)
{
    static const int od[3] = {3, 3, 4};
    return (size_t)mlp_grad_layout(in_dim, od).total;
}

size_t gsr_deform_mlp_workspace_size(int in_dim)
{
    return (size_t)MLP_BWD_BLOCKS * gsr_deform_mlp_grad_count(in_dim) * sizeof(float) + 256;
}

size_t gsr_row_mask_workspace_size(int V, int64_t n)
{
    if (V < 1 || n < 0) return 256;
    return (size_t)V * (size_t)((n + ROWMASK_BLOCK - 1) / ROWMASK_BLOCK) * sizeof(uint32_t) + 256;
}

int gsr_row_mask(int V, int64_t n, int width, const float* g, uint32_t* view_mask, int32_t* rows, int32_t* n_rows, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (V < 1 || V > 32 || n < 0 || width < 1 || (int64_t)V * n >= ((int64_t)1 << 31) || !n_rows || !workspace || (n > 0 && (!g || !view_mask || !rows))) {
        g_last_error = "gsr_row_mask: invalid argument (1 <= V <= 32, V n < 2^31, non-null pointers)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) { GSR_HIP_CHECK(hipMemsetAsync(n_rows, 0, sizeof(int32_t), stream)); return 0; }
    const int nb = (int)((n + ROWMASK_BLOCK - 1) / ROWMASK_BLOCK);
    uint32_t* counts = reinterpret_cast<uint32_t*>(workspace);
    hipLaunchKernelGGL(rowmask_count_kernel, dim3(nb), dim3(ROWMASK_BLOCK), 0, stream, V, n, width, g, view_mask, counts);
    hipLaunchKernelGGL(rowmask_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, V * nb, n_rows);
    hipLaunchKernelGGL(rowmask_list_kernel, dim3(nb), dim3(ROWMASK_BLOCK), 0, stream, V, n, (const uint32_t*)view_mask, (const uint32_t*)counts, rows);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_deform_mlp_backward(const gsr_deform_mlp* mlp, int64_t n, const float* features, const float* dout, float* dfeatures,
                            float* grads, char* workspace, void* stream_)
{
    return gsr_deform_mlp_backward_rows(mlp, n, features, dout, dfeatures, grads, workspace, nullptr, nullptr, stream_);
}

int gsr_deform_mlp_backward_rows(const gsr_deform_mlp* mlp, int64_t n, const float* features, const float* dout, float* dfeatures,
                                 float* grads, char* workspace, const int32_t* rows, const int32_t* n_rows, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    MlpWeights w;
    if (int rc = mlp_fill(mlp, &w, "gsr_deform_mlp_backward")) return rc;
    if ((rows != nullptr) != (n_rows != nullptr)) { g_last_error = "gsr_deform_mlp_backward_rows: rows and n_rows go together"; return GSR_ERR_INVALID_ARGUMENT; }
    if (n < 0 || !grads || !workspace || (n > 0 && (!features || !dout || !dfeatures))) {
        g_last_error = "gsr_deform_mlp_backward: null / invalid argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const int total = (int)gsr_deform_mlp_grad_count(w.in_dim);
    if (n == 0) { GSR_HIP_CHECK(hipMemsetAsync(grads, 0, (size_t)total * sizeof(float), stream)); return 0; }
    const int blocks = (int)std::min<int64_t>(MLP_BWD_BLOCKS, (n + MLPB_TILE - 1) / MLPB_TILE);
    float* partial = reinterpret_cast<float*>(workspace);
    switch (w.in_dim / 16) {
#define GSR_MLPB_CASE(NT) case NT: hipLaunchKernelGGL((deform_mlp_bwd_kernel<NT>), dim3(blocks), dim3(MLPB_BLOCK), 0, stream, n, features, dout, w, dfeatures, partial, rows, n_rows); break;
        GSR_MLPB_CASE(1) GSR_MLPB_CASE(2) GSR_MLPB_CASE(3) GSR_MLPB_CASE(4) GSR_MLPB_CASE(5) GSR_MLPB_CASE(6) GSR_MLPB_CASE(7) GSR_MLPB_CASE(8)
#undef GSR_MLPB_CASE
    }
    hipLaunchKernelGGL(mlp_grad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, blocks, total, (const float*)partial, grads);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- fp32-accurate dense layers on the bf16 matrix cores (include/dense_layers.h) ---------------------------------------------------------
static inline int round_up_int(int v, int m) { return (v + m - 1) / m * m; }

size_t gsr_dense_planes_size(int rows, int cols)
{
    if (rows < 1 || cols < 1) return 0;
    return (size_t)3 * round_up_int(rows, DENSE_BN) * round_up_int(cols, DENSE_BK) * sizeof(unsigned short);
}

int gsr_dense_split(int N, int K, const float* W, int ldw, int k0, int transposed, void* planes, void* stream_)
{
    if (N < 1 || K < 1 || !W || !planes || ldw < k0 + K || k0 < 0) { g_last_error = "gsr_dense_split: invalid argument"; return GSR_ERR_INVALID_ARGUMENT; }
    const int rows = transposed ? K : N, cols = transposed ? N : K;
    const int rows_pad = round_up_int(rows, DENSE_BN), cols_pad = round_up_int(cols, DENSE_BK);
    const int64_t count = (int64_t)rows_pad * cols_pad;
    hipLaunchKernelGGL(dense_split_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, N, K, W, ldw, k0, transposed ? 1 : 0,
                       reinterpret_cast<unsigned short*>(planes), rows_pad, cols_pad);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

static int dense_forward_launch(const char* who, int M, int N, int K, const float* X, int ldx, const float* gate, int ldgate, const void* planes, const float* bias,
                                int relu, float* Y, int ldy, const float* mask, int ldmask, float* colsum_out, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (M < 0 || N < 1 || K < 1 || !planes || (M > 0 && (!X || !Y)) || ldx < K || ldy < N || (gate && ldgate < K) || (mask && ldmask < N)) {
        g_last_error = std::string(who) + ": invalid argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (M == 0) {
        if (colsum_out) GSR_HIP_CHECK(hipMemsetAsync(colsum_out, 0, (size_t)N * sizeof(float), stream));
        return 0;
    }
    const int Npad = round_up_int(N, DENSE_BN), Kpad = round_up_int(K, DENSE_BK);
    const int vec = (ldx % 4 == 0) && (reinterpret_cast<uintptr_t>(X) % 16 == 0) && (!gate || ((ldgate % 4 == 0) && reinterpret_cast<uintptr_t>(gate) % 16 == 0));
    // 16-byte output stores through an LDS transposition when every row piece is whole and aligned (the trunk's layers), else element stores
    const int vec_out = (N % 4 == 0) && (ldy % 4 == 0) && (reinterpret_cast<uintptr_t>(Y) % 16 == 0) && (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0) &&
                        (!mask || ((ldmask % 4 == 0) && reinterpret_cast<uintptr_t>(mask) % 16 == 0));
    if (colsum_out && (!vec_out || !workspace || (reinterpret_cast<uintptr_t>(workspace) % 16))) {
        g_last_error = std::string(who) + ": the column sums need N % 4 == 0, 16-byte aligned rows of Y / mask and a 16-byte aligned workspace"; return GSR_ERR_INVALID_ARGUMENT;
    }
    static const int forced_bt = getenv("GSR_DENSE_ROW_TILES") ? atoi(getenv("GSR_DENSE_ROW_TILES")) : 0;      // (development: 2 .. 10)
    float* partial = colsum_out ? reinterpret_cast<float*>(workspace) : nullptr;
    // full-width outputs (the network's layers): one block of eight waves per CU, two LDS stages (dense_fwd8_kernel); GSR_DENSE_WIDE=0: the
    // 128-column kernel
    static const bool wide_ok = !(getenv("GSR_DENSE_WIDE") && getenv("GSR_DENSE_WIDE")[0] == '0');
    if (wide_ok && vec_out && N % DENSE8_BN == 0) {
        const int bt8 = forced_bt >= 2 && forced_bt <= DENSE_MAX_BT ? forced_bt : dense8_row_tiles(M, N / DENSE8_BN);
        const dim3 grid8((unsigned)((M + 16 * bt8 - 1) / (16 * bt8)), (unsigned)(N / DENSE8_BN));
        Dense8Layer L;
        L.X = X; L.gate = gate; L.planes = reinterpret_cast<const unsigned short*>(planes); L.bias = bias; L.Y = Y; L.mask = mask; L.colsum = partial;
        L.N = N; L.K = K; L.ldx = ldx; L.ldgate = ldgate; L.Npad = Npad; L.Kpad = Kpad; L.relu = relu ? 1 : 0; L.ldy = ldy; L.vec = vec ? 1 : 0; L.ldmask = ldmask;
        static std::atomic<unsigned long long> attr_set[DENSE_MAX_BT + 1];
#define GSR_DENSE8_LAUNCH(BT)                                                                                                                          \
        case BT:                                                                                                                                       \
            { const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(dense_fwd8_kernel<BT>), DENSE8_LDS_BYTES, attr_set[BT]); if (rc) return rc; } \
            hipLaunchKernelGGL(dense_fwd8_kernel<BT>, grid8, dim3(DENSE8_THREADS), DENSE8_LDS_BYTES, stream, M, L);                                    \
            break;
        switch (bt8) {
            GSR_DENSE8_LAUNCH(2) GSR_DENSE8_LAUNCH(3) GSR_DENSE8_LAUNCH(4) GSR_DENSE8_LAUNCH(5) GSR_DENSE8_LAUNCH(6) GSR_DENSE8_LAUNCH(7)
            GSR_DENSE8_LAUNCH(8) GSR_DENSE8_LAUNCH(9) GSR_DENSE8_LAUNCH(10)
        }
#undef GSR_DENSE8_LAUNCH
        if (colsum_out) hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, stream, (int)grid8.x, N, (const float*)partial, colsum_out);
        GSR_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const int bt = forced_bt >= 2 && forced_bt <= DENSE_MAX_BT ? forced_bt : dense_row_tiles(M, Npad / DENSE_BN);
    const dim3 grid((unsigned)((M + 16 * bt - 1) / (16 * bt)), (unsigned)(Npad / DENSE_BN));
    hipLaunchKernelGGL(dense_fwd_kernel, grid, dim3(DENSE_THREADS), 0, stream, M, N, K, X, ldx, gate, ldgate,
                       reinterpret_cast<const unsigned short*>(planes), Npad, Kpad, bias, relu ? 1 : 0, Y, ldy, vec ? 1 : 0, vec_out ? 1 : 0, bt, mask, ldmask, partial);
    if (colsum_out) hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, stream, (int)grid.x, N, (const float*)partial, colsum_out);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_dense_forward(int M, int N, int K, const float* X, int ldx, const float* gate, int ldgate, const void* planes, const float* bias, int relu,
                      float* Y, int ldy, void* stream_)
{
    return dense_forward_launch("gsr_dense_forward", M, N, K, X, ldx, gate, ldgate, planes, bias, relu, Y, ldy, nullptr, 0, nullptr, nullptr, stream_);
}

size_t gsr_dense_backward_input_workspace_size(int M, int N)
{
    if (M < 1 || N < 1) return 256;
    return (size_t)((M + 31) / 32) * (size_t)N * sizeof(float) + 256;        // (at least two row tiles of 16 per block)
}

int gsr_dense_backward_input(int M, int N, int K, const float* G, int ldg, const void* planes_t, const float* mask, int ldmask, float* dX, int lddx,
                             float* dbias, char* workspace, void* stream_)
{
    return dense_forward_launch("gsr_dense_backward_input", M, N, K, G, ldg, nullptr, 0, planes_t, nullptr, 0, dX, lddx, mask, ldmask, dbias, workspace, stream_);
}

size_t gsr_dense_chain_workspace_size(int M, int N, int count)
{
    if (M < 1 || N < 1 || count < 1) return 256;
    return (size_t)count * ((size_t)((M + 31) / 32) * (size_t)N * sizeof(float) + 256) + 256;
}

int gsr_dense_chain(int M, int N, int count, const gsr_dense_chain_op* ops, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (M < 0 || N < 1 || N != DENSE8_BN || count < 1 || count > DENSE8_CHAIN_MAX || !ops) {
        g_last_error = "gsr_dense_chain: 1..8 products of 256 output columns each"; return GSR_ERR_INVALID_ARGUMENT;
    }
    bool sums = false;
    for (int l = 0; l < count; l++) {
        const gsr_dense_chain_op& q = ops[l];
        const bool aligned = (q.ldx % 4 == 0) && (q.ldy % 4 == 0) && !(reinterpret_cast<uintptr_t>(q.X) % 16) && !(reinterpret_cast<uintptr_t>(q.Y) % 16) &&
                             (!q.bias || !(reinterpret_cast<uintptr_t>(q.bias) % 16)) && (!q.mask || ((q.ldmask % 4 == 0) && !(reinterpret_cast<uintptr_t>(q.mask) % 16)));
        if (q.K < 1 || !q.planes || (M > 0 && (!q.X || !q.Y)) || q.ldx < q.K || q.ldy < N || (q.mask && q.ldmask < N) || !aligned) {
            g_last_error = "gsr_dense_chain: invalid product (16-byte aligned rows of X / Y / mask, ldx >= K, ldy >= 256)"; return GSR_ERR_INVALID_ARGUMENT;
        }
        sums = sums || q.dbias;
    }
    if (sums && (!workspace || (reinterpret_cast<uintptr_t>(workspace) % 16))) { g_last_error = "gsr_dense_chain: the bias gradients need a 16-byte aligned workspace"; return GSR_ERR_INVALID_ARGUMENT; }
    if (M == 0) {
        for (int l = 0; l < count; l++) if (ops[l].dbias) GSR_HIP_CHECK(hipMemsetAsync(ops[l].dbias, 0, (size_t)N * sizeof(float), stream));
        return 0;
    }
    static const int forced_bt = getenv("GSR_DENSE_ROW_TILES") ? atoi(getenv("GSR_DENSE_ROW_TILES")) : 0;
    const int bt = forced_bt >= 2 && forced_bt <= DENSE_MAX_BT ? forced_bt : dense8_row_tiles(M, 1);
    const unsigned blocks = (unsigned)((M + 16 * bt - 1) / (16 * bt));
    const size_t per_op = ((size_t)((M + 31) / 32) * (size_t)N * sizeof(float) + 255) & ~size_t(255);
    Dense8Chain c;
    c.M = M; c.count = count;
    float* partial[DENSE8_CHAIN_MAX];
    for (int l = 0; l < count; l++) {
        const gsr_dense_chain_op& q = ops[l];
        Dense8Layer& L = c.layer[l];
        partial[l] = q.dbias ? reinterpret_cast<float*>(workspace + (size_t)l * per_op) : nullptr;
        L.X = q.X; L.gate = nullptr; L.planes = reinterpret_cast<const unsigned short*>(q.planes); L.bias = q.bias; L.Y = q.Y; L.mask = q.mask; L.colsum = partial[l];
        L.N = N; L.K = q.K; L.ldx = q.ldx; L.ldgate = 0; L.Npad = round_up_int(N, DENSE_BN); L.Kpad = round_up_int(q.K, DENSE_BK); L.relu = q.relu ? 1 : 0;
        L.ldy = q.ldy; L.vec = 1; L.ldmask = q.ldmask;
    }
    static std::atomic<unsigned long long> attr_set[DENSE_MAX_BT + 1];
#define GSR_CHAIN8_LAUNCH(BT)                                                                                                                          \
    case BT:                                                                                                                                           \
        { const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(dense_chain8_kernel<BT>), DENSE8_LDS_BYTES, attr_set[BT]); if (rc) return rc; } \
        hipLaunchKernelGGL(dense_chain8_kernel<BT>, dim3(blocks), dim3(DENSE8_THREADS), DENSE8_LDS_BYTES, stream, c);                                  \
        break;
    switch (bt) {
        GSR_CHAIN8_LAUNCH(2) GSR_CHAIN8_LAUNCH(3) GSR_CHAIN8_LAUNCH(4) GSR_CHAIN8_LAUNCH(5) GSR_CHAIN8_LAUNCH(6) GSR_CHAIN8_LAUNCH(7)
        GSR_CHAIN8_LAUNCH(8) GSR_CHAIN8_LAUNCH(9) GSR_CHAIN8_LAUNCH(10)
    }
#undef GSR_CHAIN8_LAUNCH
    for (int l = 0; l < count; l++)
        if (ops[l].dbias) hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, stream, (int)blocks, N, (const float*)partial[l], ops[l].dbias);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_dense_split_many(int count, const gsr_dense_split_item* items, void* stream_)
{
    if (count < 0 || count > DENSE_SPLIT_MAX || (count > 0 && !items)) { g_last_error = "gsr_dense_split_many: 0 <= count <= 24 items"; return GSR_ERR_INVALID_ARGUMENT; }
    if (count == 0) return 0;
    DenseSplitItems a;
    int64_t largest = 0;
    for (int i = 0; i < count; i++) {
        const gsr_dense_split_item& q = items[i];
        if (q.N < 1 || q.K < 1 || !q.W || !q.planes || q.k0 < 0 || q.ldw < q.k0 + q.K) { g_last_error = "gsr_dense_split_many: invalid item"; return GSR_ERR_INVALID_ARGUMENT; }
        const int rows = q.transposed ? q.K : q.N, cols = q.transposed ? q.N : q.K;
        DenseSplitItem& d = a.item[i];
        d.W = q.W; d.planes = reinterpret_cast<unsigned short*>(q.planes); d.N = q.N; d.K = q.K; d.ldw = q.ldw; d.k0 = q.k0; d.transposed = q.transposed ? 1 : 0;
        d.rows_pad = round_up_int(rows, DENSE_BN); d.cols_pad = round_up_int(cols, DENSE_BK);
        largest = std::max(largest, (int64_t)d.rows_pad * d.cols_pad);
    }
    const unsigned bx = (unsigned)std::min<int64_t>((largest + 255) / 256, 128);
    hipLaunchKernelGGL(dense_split_many_kernel, dim3(bx, (unsigned)count), dim3(256), 0, (hipStream_t)stream_, a);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

static void dense_wgrad_plan(int M, int N, int K, int* slices, int* rows_per_slice)
{
    const int tiles = ((N + DENSE_BM - 1) / DENSE_BM) * ((K + DENSE_BN - 1) / DENSE_BN);
    int s = std::max(1, 512 / std::max(1, tiles));                             // ~two blocks per CU
    s = std::min(s, std::max(1, (M + 4 * DENSE_WG_ROWS - 1) / (4 * DENSE_WG_ROWS)));   // at least four steps per slice
    const int rps = round_up_int((M + s - 1) / s, DENSE_WG_ROWS);
    *rows_per_slice = std::max(rps, DENSE_WG_ROWS);
    *slices = std::max(1, (M + *rows_per_slice - 1) / *rows_per_slice);
}

size_t gsr_dense_wgrad_workspace_size(int M, int N, int K)
{
    if (M < 1 || N < 1 || K < 1) return 256;
    int slices, rps;
    dense_wgrad_plan(M, N, K, &slices, &rps);
    return (size_t)slices * N * K * sizeof(float) + 256;
}

int gsr_dense_wgrad(int M, int N, int K, const float* G, int ldg, const float* gate, int ldgate, const float* X, int ldx, float* dW, int lddw,
                    char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (M < 0 || N < 1 || K < 1 || !dW || lddw < K || (M > 0 && (!G || !X || !workspace)) || ldg < N || ldx < K || (gate && ldgate < N)) {
        g_last_error = "gsr_dense_wgrad: invalid argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (M == 0) { GSR_HIP_CHECK(hipMemset2DAsync(dW, (size_t)lddw * sizeof(float), 0, (size_t)K * sizeof(float), (size_t)N, stream)); return 0; }
    int slices, rps;
    dense_wgrad_plan(M, N, K, &slices, &rps);
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
    const dim3 grid((unsigned)((N + DENSE_BM - 1) / DENSE_BM), (unsigned)((K + DENSE_BN - 1) / DENSE_BN), (unsigned)slices);
    hipLaunchKernelGGL(dense_wgrad_kernel, grid, dim3(DENSE_THREADS), 0, stream, M, N, K, G, ldg, gate, ldgate, X, ldx, rps, partial);
    hipLaunchKernelGGL(dense_wgrad_sum_kernel, dim3((unsigned)((N * K + 255) / 256)), dim3(256), 0, stream, slices, N * K, (const float*)partial, K, dW, lddw);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

// several weight gradients over the same rows in one launch (gs_dense.h, round 6)
static int dense_wgrad_many_plan(int M, int count, const gsr_dense_wgrad_item* items, DenseWgradItems* out, int* slices, int* rows_per_slice)
{
    if (count < 1 || count > DENSE_WGM_MAX || !items) return -1;
    int tiles = 0;
    for (int i = 0; i < count; i++) {
        const gsr_dense_wgrad_item& q = items[i];
        if (q.N < 1 || q.K < 1 || q.ldg < q.N || q.ldx < q.K || q.lddw < q.K) return -1;
        if (out) {
            DenseWgradItem& d = out->item[i];
            d.G = q.G; d.X = q.X; d.dW = q.dW; d.ldg = q.ldg; d.ldx = q.ldx; d.lddw = q.lddw; d.N = q.N; d.K = q.K;
            d.tiles_k = (q.K + DENSE_BN - 1) / DENSE_BN; d.tile0 = tiles;
            d.vec = (q.N % 4 == 0 && q.K % 4 == 0 && q.ldg % 4 == 0 && q.ldx % 4 == 0 && reinterpret_cast<uintptr_t>(q.G) % 16 == 0 && reinterpret_cast<uintptr_t>(q.X) % 16 == 0) ? 1 : 0;
        }
        tiles += ((q.N + DENSE_BM - 1) / DENSE_BM) * ((q.K + DENSE_BN - 1) / DENSE_BN);
    }
    if (out) { out->count = count; out->total_tiles = tiles; }
    // every block resident at once (two per CU), at least four 32-row steps per slice; GSR_WGM_BLOCKS: dev knob
    static const int blocks = getenv("GSR_WGM_BLOCKS") ? std::max(1, atoi(getenv("GSR_WGM_BLOCKS"))) : 512;
    int s = std::max(1, blocks / std::max(1, tiles));
    s = std::min(s, std::max(1, (M + 4 * DENSE_WG_ROWS - 1) / (4 * DENSE_WG_ROWS)));
    const int rps = std::max(DENSE_WG_ROWS, round_up_int((std::max(M, 1) + s - 1) / s, DENSE_WG_ROWS));
    *rows_per_slice = rps;
    *slices = std::max(1, (M + rps - 1) / rps);
    return tiles;
}

size_t gsr_dense_wgrad_many_workspace_size(int M, int count, const gsr_dense_wgrad_item* items)
{
    int slices, rps;
    const int tiles = dense_wgrad_many_plan(M, count, items, nullptr, &slices, &rps);
    if (tiles < 0 || M < 1) return 256;
    return (size_t)slices * tiles * (DENSE_BM * DENSE_BN) * sizeof(float) + 256;
}

int gsr_dense_wgrad_many(int M, int count, const gsr_dense_wgrad_item* items, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DenseWgradItems t;
    memset(&t, 0, sizeof(t));
    int slices, rps;
    const int tiles = M < 0 ? -1 : dense_wgrad_many_plan(M, count, items, &t, &slices, &rps);
    bool ok = tiles > 0;
    for (int i = 0; ok && i < count; i++) ok = items[i].dW && (M == 0 || (items[i].G && items[i].X));
    if (!ok || (M > 0 && !workspace)) {
        g_last_error = "gsr_dense_wgrad_many: invalid argument (1..12 items, ldg >= N, ldx >= K, lddw >= K, non-null operands and workspace)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (M == 0) {
        for (int i = 0; i < count; i++)
            GSR_HIP_CHECK(hipMemset2DAsync(items[i].dW, (size_t)items[i].lddw * sizeof(float), 0, (size_t)items[i].K * sizeof(float), (size_t)items[i].N, stream));
        return 0;
    }
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
    hipLaunchKernelGGL(dense_wgrad_many_kernel, dim3((unsigned)tiles, (unsigned)slices), dim3(DENSE_THREADS), 0, stream, M, t, rps, partial);
    hipLaunchKernelGGL(dense_wgrad_many_sum_kernel, dim3((unsigned)tiles, 16), dim3(256), 0, stream, t, slices, (const float*)partial);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_trunk_forward(const gsr_trunk* t, int R, const float* emb, float* const* outs, const int* ldo, float* heads, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!t || R < 0 || t->E < 1 || t->E > TR_EPAD || t->n_head_outputs < 1 || t->n_head_outputs > 16 || !outs || !ldo || (R > 0 && (!emb || !heads))) {
        g_last_error = "gsr_trunk_forward: invalid argument (embedding width 1..96, 1..16 head outputs)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    TrunkArgs a{};
    a.R = R; a.E = t->E; a.NH = t->n_head_outputs; a.emb = emb; a.heads = heads;
    for (int k = 0; k < 10; k++) { if (!t->planes[k]) { g_last_error = "gsr_trunk_forward: null weight planes"; return GSR_ERR_INVALID_ARGUMENT; } a.planes[k] = reinterpret_cast<const unsigned short*>(t->planes[k]); }
    for (int k = 0; k < 9; k++) { if (!t->bias[k] || reinterpret_cast<uintptr_t>(t->bias[k]) % 16) { g_last_error = "gsr_trunk_forward: null / unaligned bias"; return GSR_ERR_INVALID_ARGUMENT; } a.bias[k] = t->bias[k]; }
    for (int l = 0; l < TR_LAYERS; l++) {
        if (!outs[l] || ldo[l] < TR_W || ldo[l] % 4 || reinterpret_cast<uintptr_t>(outs[l]) % 16) { g_last_error = "gsr_trunk_forward: layer outputs must be 16-byte aligned rows of >= 256 floats"; return GSR_ERR_INVALID_ARGUMENT; }
        a.outs[l] = outs[l]; a.ldo[l] = ldo[l];
    }
    if (R == 0) return 0;
    static std::atomic<unsigned long long> attr_set;
    { const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(trunk_fwd_kernel), TR_LDS_BYTES, attr_set); if (rc) return rc; }
    hipLaunchKernelGGL(trunk_fwd_kernel, dim3((unsigned)((R + TR_BM - 1) / TR_BM)), dim3(256), TR_LDS_BYTES, stream, a);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- SC-GS control nodes (include/control_nodes.h) ------------------------------------------------------------------------------
int gsr_knn_points_batch(int64_t B, int64_t n, int64_t m, int D, int K, const float* p1, const float* p2, float* dist2, int64_t* idx, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (B < 0 || B > 65535 || n < 0 || m < 0 || D < 1 || D > GSR_KNN_MAX_DIM || K < 1 || K > GSR_KNN_MAX_K || (B > 0 && n > 0 && (!p1 || !dist2 || !idx)) ||
        (B > 0 && m > 0 && !p2)) {
        g_last_error = "gsr_knn_points_batch: null / invalid argument (0 <= B <= 65535, 1 <= D <= 32, 1 <= K <= 32)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (B == 0 || n == 0) return 0;
    if (D <= 4 && K > 4 && m <= 64 * 16) {            // small candidate sets, longer lists: one wave per query, all batch elements in one launch
        const dim3 grid((unsigned)((n + NODE_BLOCK / 64 - 1) / (NODE_BLOCK / 64)), (unsigned)B), block(NODE_BLOCK);
#define GSR_KNNW(C) hipLaunchKernelGGL((knn_points3_wave_kernel<C>), grid, block, 0, stream, n, m, D, K, p1, p2, dist2, idx)
        if (m <= 64 * 4) GSR_KNNW(4); else if (m <= 64 * 8) GSR_KNNW(8); else GSR_KNNW(16);
#undef GSR_KNNW
        GSR_HIP_CHECK(hipGetLastError());
        return 0;
    }
    for (int64_t b = 0; b < B; b++) {
        const int rc = gsr_knn_points(n, m, D, K, p1 + b * n * D, p2 ? p2 + b * m * D : p2, dist2 + b * n * K, idx + b * n * K, stream_);
        if (rc < 0) return rc;
    }
    return 0;
}

int gsr_knn_points(int64_t n, int64_t m, int D, int K, const float* p1, const float* p2, float* dist2, int64_t* idx, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || m < 0 || D < 1 || D > GSR_KNN_MAX_DIM || K < 1 || K > GSR_KNN_MAX_K || (n > 0 && (!p1 || !dist2 || !idx)) || (m > 0 && !p2)) {
        g_last_error = "gsr_knn_points: null / invalid argument (1 <= D <= 32, 1 <= K <= 32)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return 0;
    const int64_t rows = std::min<int64_t>((int64_t)(NODE_CHUNK * 3 / D), std::max<int64_t>(m, 1));
    const size_t lds = (size_t)rows * D * sizeof(float);
    const dim3 grid((unsigned)((n + NODE_BLOCK - 1) / NODE_BLOCK)), block(NODE_BLOCK);
#define GSR_KNN_LAUNCH(DMAX, KMAX) hipLaunchKernelGGL((knn_points_kernel<DMAX, KMAX>), grid, block, lds, stream, n, m, D, K, p1, p2, dist2, idx)
#define GSR_KNN_K(DMAX) do { if (K <= 4) GSR_KNN_LAUNCH(DMAX, 4); else if (K <= 8) GSR_KNN_LAUNCH(DMAX, 8); else if (K <= 16) GSR_KNN_LAUNCH(DMAX, 16); else GSR_KNN_LAUNCH(DMAX, 32); } while (0)
    if (D <= 4) {
#define GSR_KNN3(KMAX, EXACT) hipLaunchKernelGGL((knn_points3_kernel<KMAX, EXACT>), grid, block, 0, stream, n, m, D, K, p1, p2, dist2, idx)
        switch (K) {
        case 1: GSR_KNN3(1, true); break;
        case 2: GSR_KNN3(2, true); break;
        case 3: GSR_KNN3(3, true); break;
        case 4: GSR_KNN3(4, true); break;
        default: if (K <= 8) GSR_KNN3(8, false); else if (K <= 16) GSR_KNN3(16, false); else GSR_KNN3(32, false);
        }
#undef GSR_KNN3
    } else if (D <= 8) GSR_KNN_K(8); else GSR_KNN_K(32);
#undef GSR_KNN_K
#undef GSR_KNN_LAUNCH
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

static int node_blend_check(const gsr_node_blend* a, const char* who)
{
    static thread_local std::string msg;
    auto fail = [&](const char* what) { msg = std::string(who) + ": " + what; g_last_error = msg.c_str(); return GSR_ERR_INVALID_ARGUMENT; };
    if (!a) return fail("null descriptor");
    if (a->n < 0 || a->m < 1) return fail("n < 0 or no control nodes");
    if (a->K < 1 || a->K > GSR_BLEND_MAX_K) return fail("K outside 1..8");
    if (a->node_stride < 3) return fail("node_stride < 3");
    if (a->n > 0 && !a->x) return fail("null x");
    if (!a->nodes || !a->node_radius) return fail("null nodes / node_radius");
    if (a->node_trans && (!a->node_rot || !a->node_scale)) return fail("node_trans without node_rot / node_scale");
    if (a->node_trans && a->local_frame && !a->node_frame && !a->node_local_rotation) return fail("local_frame without node_frame / node_local_rotation");
    if (a->attr_stride < 0 || a->grad_stride < 0 || (a->attr_stride > 0 && (a->attr_stride < 4 || a->node_frame)) || (a->grad_stride > 0 && a->grad_stride < 4))
        return fail("attr_stride / grad_stride: 0 (packed) or >= 4 floats per node, and no node_frame with strided attributes");
    return 0;
}

int gsr_node_blend_forward(const gsr_node_blend* a, float* nn_weight, float* nn_dist, int64_t* nn_idx, float* d_xyz, float* d_rotation,
                           float* d_scaling, void* stream_)
{
    return gsr_node_blend_forward_batch(a, 1, nn_weight, nn_dist, nn_idx, d_xyz, d_rotation, d_scaling, stream_);
}

int gsr_node_blend_forward_batch(const gsr_node_blend* a, int B, float* nn_weight, float* nn_dist, int64_t* nn_idx, float* d_xyz, float* d_rotation,
                                 float* d_scaling, void* stream_)
{
    if (int rc = node_blend_check(a, "gsr_node_blend_forward")) return rc;
    if (B < 1 || B > 65535 || (B > 1 && !a->node_trans)) { g_last_error = "gsr_node_blend_forward_batch: 1 <= B <= 65535, B > 1 needs node attributes"; return GSR_ERR_INVALID_ARGUMENT; }
    if (a->n == 0) return 0;
    if (!nn_weight || !nn_dist || !nn_idx || (a->node_trans && (!d_xyz || !d_rotation || !d_scaling))) {
        g_last_error = "gsr_node_blend_forward: null output"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const dim3 grid((unsigned)((a->n + NODE_BLOCK - 1) / NODE_BLOCK), (unsigned)B), block(NODE_BLOCK);
#define GSR_BLEND_FWD(KMAX, EXACT) hipLaunchKernelGGL((node_blend_fwd_kernel<KMAX, EXACT>), grid, block, 0, (hipStream_t)stream_, *a, nn_weight, nn_dist, nn_idx, d_xyz, d_rotation, d_scaling)
    switch (a->K) {
    case 1: GSR_BLEND_FWD(1, true); break;
    case 2: GSR_BLEND_FWD(2, true); break;
    case 3: GSR_BLEND_FWD(3, true); break;
    case 4: GSR_BLEND_FWD(4, true); break;
    default: GSR_BLEND_FWD(GSR_BLEND_MAX_K, false);
    }
#undef GSR_BLEND_FWD
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

static int node_bwd_blocks(int64_t n) { return (int)std::min<int64_t>(256, std::max<int64_t>(1, (n + NODE_BLOCK - 1) / NODE_BLOCK)); }

static size_t node_ws_floats(int64_t n, int32_t m)      // per batch element: block partials + the summed row, a multiple of 64 floats
{
    const size_t rows = m <= NODE_LDS_MAX ? (size_t)node_bwd_blocks(n) : 1;
    const size_t atomics_route = (rows + 1) * (size_t)m * NODE_GRAD;
    const size_t ordered_route = ((size_t)n * NODE_DET_MAX_K + (size_t)m) * NODE_GRAD;        // per-(Gaussian, k) contributions + the summed row
    return (std::max(atomics_route, ordered_route) + 63) & ~size_t(63);
}
static size_t index_csr_ints(int S, int E, int Nv) { return (((size_t)S * ((size_t)E + 2 * (size_t)Nv + 1) + 64) * sizeof(int) + 255) & ~size_t(255); }
static int index_csr_epad(int E) { return round_up_int(E, CSR_REG_TRIP); }
static size_t node_ws_shared_bytes(int64_t n, int32_t m)  // once per call (any batch size): the reverse lists of nn_idx (gsr_index_csr's workspace for one set)
{
    const int E = (int)std::min<int64_t>(n * NODE_DET_MAX_K, INT32_MAX / 2);
    return index_csr_ints(1, E, m) + (size_t)index_csr_epad(E > 0 ? E : 1) * sizeof(unsigned short) + 512;
}

size_t gsr_node_blend_workspace_size(int64_t n, int32_t m)
{
    if (m < 1) return 256;
    return node_ws_floats(n, m) * sizeof(float) + node_ws_shared_bytes(n, m) + 512;
}

size_t gsr_node_blend_workspace_size_batch(int64_t n, int32_t m, int B)
{
    if (m < 1 || B < 1) return 256;
    return (size_t)B * node_ws_floats(n, m) * sizeof(float) + node_ws_shared_bytes(n, m) + 512;
}

/* ---- deterministic scatter-add through an index array (include/control_nodes.h) ------------------------------------------------------ */
size_t gsr_index_csr_workspace_size(int S, int E, int Nv)
{
    return index_csr_ints(S, E, Nv) + (size_t)S * index_csr_epad(E > 0 ? E : 1) * sizeof(unsigned short) + 512;      // the lists + the set packed to 16 bits
}

static void index_csr_carve(char* workspace, int S, int E, int Nv, int*& order, int*& seg, int*& cursor)
{
    int* p = reinterpret_cast<int*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
    cursor = p; order = p + ((S + 63) & ~63); seg = order + (size_t)S * E;
}

int gsr_index_csr(int S, int E, int Nv, const int64_t* idx, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (S < 1 || S > 65535 || E < 0 || Nv < 1 || !workspace || (E > 0 && !idx)) { g_last_error = "gsr_index_csr: invalid argument"; return GSR_ERR_INVALID_ARGUMENT; }
    int *order, *seg, *cursor;
    index_csr_carve(workspace, S, E, Nv, order, seg, cursor);
    GSR_HIP_CHECK(hipMemsetAsync(cursor, 0, (size_t)S * sizeof(int), stream));
    static const bool reg_ok = !(getenv("GSR_CSR_REGISTERS") && getenv("GSR_CSR_REGISTERS")[0] == '0');
    if (reg_ok && E >= 1 && E <= 20 * CSR_REG_TRIP && Nv < 65535) {          // small sets: packed to 16 bits, a wave holds the whole set in registers
        const int Epad = index_csr_epad(E);
        unsigned short* packed = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(cursor) + index_csr_ints(S, E, Nv));
        hipLaunchKernelGGL(index_csr_pack_kernel, dim3((unsigned)((Epad + 255) / 256), (unsigned)S), dim3(256), 0, stream, E, Epad, idx, packed);
        const dim3 grid((unsigned)((Nv + 3) / 4), (unsigned)S);
        if (Epad <= 10 * CSR_REG_TRIP) hipLaunchKernelGGL(index_csr_reg_kernel<10>, grid, dim3(256), 0, stream, E, Epad, Nv, (const unsigned short*)packed, order, seg, cursor);
        else hipLaunchKernelGGL(index_csr_reg_kernel<20>, grid, dim3(256), 0, stream, E, Epad, Nv, (const unsigned short*)packed, order, seg, cursor);
        GSR_HIP_CHECK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(index_csr_kernel, dim3((unsigned)((Nv + 3) / 4), (unsigned)S), dim3(256), 0, stream, E, Nv, idx, order, seg, cursor);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t gsr_node_embedding_workspace_size(int n, int M, int Fx, int Ft)
{
    return n >= 0 && M >= 0 && Fx >= 0 && Ft >= 0 ? ((size_t)M * 3 * (1 + 2 * Fx) + (size_t)n * (1 + 2 * Ft)) * sizeof(float) + 16 : 0;
}

int gsr_node_embedding(int n, int M, int Fx, int Ft, const float* nodes, int node_stride, const float* times, float* out, char* workspace, void* stream_)
{
    if (n < 0 || M < 0 || Fx < 0 || Fx > 24 || Ft < 0 || Ft > 24 || node_stride < 3 || ((size_t)n * M > 0 && (!nodes || !times || !out || !workspace))) {
        g_last_error = "gsr_node_embedding: invalid argument (0 <= frequencies <= 24, node_stride >= 3, no null buffers)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const int Wx = 3 * (1 + 2 * Fx), Wt = 1 + 2 * Ft;
    const size_t total = (size_t)n * M * (Wx + Wt), small = (size_t)M * Wx + (size_t)n * Wt;
    if (total) {
        float* tables = reinterpret_cast<float*>(workspace);
        hipLaunchKernelGGL(node_embedding_tables_kernel, dim3((unsigned)((small + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, n, M, Fx, Ft, nodes, node_stride, times, tables);
        hipLaunchKernelGGL(node_embedding_expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, n, M, Wx, Wt, (const float*)tables, out);
    }
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t gsr_relu_backward_bias_workspace_size(int rows, int cols)
{
    return rows > 0 && cols > 0 ? (size_t)((rows + RELU_BAND - 1) / RELU_BAND) * (size_t)cols * sizeof(float) : 0;
}

int gsr_relu_backward_bias(int rows, int cols, const float* dY, const float* Y, float* G, float* dbias, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    const bool cols_ok = cols == 64 || cols == 128 || cols == 256 || cols == 512 || cols == 1024;
    if (rows < 0 || !cols_ok || !dbias || (rows > 0 && (!dY || !Y || !G || !workspace))) {
        g_last_error = "gsr_relu_backward_bias: invalid argument (cols in {64, 128, 256, 512, 1024}; 16-byte aligned rows)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (((uintptr_t)dY | (uintptr_t)Y | (uintptr_t)G | (uintptr_t)workspace) & 15) { g_last_error = "gsr_relu_backward_bias: buffers must be 16-byte aligned"; return GSR_ERR_INVALID_ARGUMENT; }
    const int bands = (rows + RELU_BAND - 1) / RELU_BAND;
    if (bands > 0) hipLaunchKernelGGL(relu_bwd_bias_kernel, dim3((unsigned)bands), dim3(256), 0, stream, rows, cols, dY, Y, G, reinterpret_cast<float*>(workspace));
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, stream, bands, cols, reinterpret_cast<const float*>(workspace), dbias);
    const int debug = 0;
    GSR_STAGE("gsr_relu_backward_bias");
    return 0;
}

int gsr_multi_add(int count, const gsr_multi_add_item* items, void* stream_)
{
    if (count < 0 || count > MULTI_ADD_MAX || (count > 0 && !items)) { g_last_error = "gsr_multi_add: 0 <= count <= 64 items"; return GSR_ERR_INVALID_ARGUMENT; }
    if (count == 0) return 0;
    MultiAddItems a;
    int largest = 0;
    for (int i = 0; i < count; i++) {
        const gsr_multi_add_item& q = items[i];
        if (q.count < 0 || (q.count > 0 && !q.dst)) { g_last_error = "gsr_multi_add: invalid item"; return GSR_ERR_INVALID_ARGUMENT; }
        a.item[i].dst = q.dst; a.item[i].count = q.count;
        for (int s = 0; s < MULTI_ADD_SOURCES; s++) a.item[i].src[s] = q.src[s];
        largest = std::max(largest, q.count);
    }
    if (largest == 0) return 0;
    hipLaunchKernelGGL(multi_add_kernel, dim3((unsigned)std::min((largest + 255) / 256, 64), (unsigned)count), dim3(256), 0, (hipStream_t)stream_, a);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_segment_sum(int B, int S, int E, int C, int Nv, const float* g, const char* csr_workspace, const int* set_of_b, float* out, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (B < 0 || S < 1 || E < 0 || C < 1 || Nv < 1 || !csr_workspace || (B > 0 && (!out || (E > 0 && !g)))) { g_last_error = "gsr_segment_sum: invalid argument"; return GSR_ERR_INVALID_ARGUMENT; }
    if (B == 0) return 0;
    int *order, *seg, *cursor;
    index_csr_carve(const_cast<char*>(csr_workspace), S, E, Nv, order, seg, cursor);
    const size_t total = (size_t)B * Nv * C * SEG_LANES;
    hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, B, E, C, Nv, g, (size_t)E * C, (const int*)order, (const int*)seg,
                       set_of_b, out, (size_t)Nv * C);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_node_blend_backward(const gsr_node_blend* a, const float* nn_weight, const float* nn_dist, const int64_t* nn_idx,
                            const float* g_xyz, const float* g_rotation, const float* g_scaling, const float* g_nn_weight,
                            float* g_node_trans, float* g_node_rot, float* g_node_scale, float* g_node_frame, float* g_node_radius,
                            float* g_node_weight, char* workspace, void* stream_)
{
    return gsr_node_blend_backward_batch(a, 1, nn_weight, nn_dist, nn_idx, g_xyz, g_rotation, g_scaling, g_nn_weight, g_node_trans, g_node_rot, g_node_scale,
                                         g_node_frame, g_node_radius, g_node_weight, workspace, stream_);
}

int gsr_node_blend_backward_batch(const gsr_node_blend* a, int B, const float* nn_weight, const float* nn_dist, const int64_t* nn_idx,
                                  const float* g_xyz, const float* g_rotation, const float* g_scaling, const float* g_nn_weight,
                                  float* g_node_trans, float* g_node_rot, float* g_node_scale, float* g_node_frame, float* g_node_radius,
                                  float* g_node_weight, char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = node_blend_check(a, "gsr_node_blend_backward")) return rc;
    if (B < 1 || B > 65535 || (B > 1 && (g_nn_weight || !a->node_trans))) {
        g_last_error = "gsr_node_blend_backward_batch: 1 <= B <= 65535; B > 1 needs node attributes and takes no direct cotangent of nn_weight"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (!workspace || (a->n > 0 && (!nn_weight || !nn_dist || !nn_idx))) { g_last_error = "gsr_node_blend_backward: null workspace / saved tensors"; return GSR_ERR_INVALID_ARGUMENT; }
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
    const int total = a->m * NODE_GRAD;
    const bool use_lds = a->m <= NODE_LDS_MAX;
    const size_t stride = node_ws_floats(a->n, a->m);
    // K <= 4 (every shipped call): the ordered route -- contributions written per (Gaussian, k), summed per node in the order of the Gaussians
    // (index_csr_kernel + segment_sum_kernel above): bit-reproducible. GSR_NODE_ATOMICS=1 keeps round 3's LDS-atomic accumulation.
    static const bool force_atomics = getenv("GSR_NODE_ATOMICS") && getenv("GSR_NODE_ATOMICS")[0] == '1';
    const bool ordered = a->K <= NODE_DET_MAX_K && a->n > 0 && !force_atomics;
    float* summed;
    if (ordered) {
        const int E = (int)(a->n * a->K);
        float* contrib = partial;                                  // [B][E][21], batch stride `stride`
        summed = partial + (size_t)E * NODE_GRAD;
        char* shared = reinterpret_cast<char*>(partial + (size_t)B * stride);
        hipLaunchKernelGGL(node_blend_bwd_kernel, dim3((unsigned)node_bwd_blocks(a->n), (unsigned)B), dim3(NODE_BLOCK), 0, stream, *a, nn_weight,
                           nn_dist, nn_idx, g_xyz, g_rotation, g_scaling, g_nn_weight, partial, 0, stride, contrib, stride);
        if (int rc = gsr_index_csr(1, E, a->m, nn_idx, shared, stream_)) return rc;
        int *order, *seg, *cursor;
        index_csr_carve(shared, 1, E, a->m, order, seg, cursor);
        const size_t nthreads = (size_t)B * a->m * NODE_GRAD * SEG_LANES;
        hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, B, E, NODE_GRAD, a->m, (const float*)contrib, stride,
                           (const int*)order, (const int*)seg, (const int*)nullptr, summed, stride);
    } else {
        const int G = use_lds ? node_bwd_blocks(a->n) : 1;
        if (!use_lds || a->n == 0) {
            for (int b = 0; b < B; b++) GSR_HIP_CHECK(hipMemsetAsync(partial + (size_t)b * stride, 0, (size_t)total * sizeof(float), stream));
        }
        if (a->n > 0) {
            const int blocks = node_bwd_blocks(a->n);
            hipLaunchKernelGGL(node_blend_bwd_kernel, dim3(blocks, (unsigned)B), dim3(NODE_BLOCK), use_lds ? (size_t)total * sizeof(float) : 0, stream, *a, nn_weight,
                               nn_dist, nn_idx, g_xyz, g_rotation, g_scaling, g_nn_weight, partial, use_lds ? 1 : 0, stride, (float*)nullptr, (size_t)0);
        }
        summed = partial + (size_t)G * total;
        hipLaunchKernelGGL(node_grad_reduce_kernel, dim3((total + 255) / 256, (unsigned)B), dim3(256), 0, stream, G, total, (const float*)partial, summed, stride);
    }
    hipLaunchKernelGGL(node_grad_finalize_kernel, dim3((a->m + 255) / 256, (unsigned)B), dim3(256), 0, stream, *a, (const float*)summed, g_node_trans, g_node_rot,
                       g_node_scale, g_node_frame, g_node_radius, g_node_weight, stride);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"

// ---- map maintenance + camera step (include/slam_map.h) -----------------------------------------------------------------------
extern "C" {

size_t gsr_seed_workspace_size(int n)
{
    const size_t N = (size_t)(n > 0 ? n : 1);
    return gsr_knn_workspace_size(n) + N * sizeof(float) + 512;      // k-NN scratch + the mean squared distances
}

int gsr_seed_from_rgbd(int n, const int* pix, int width, int height, const float* depth, const float* image, const float* exposure_a,
                       const float* exposure_b, float fx, float fy, float cx, float cy, const float* R, const float* T, float point_size,
                       int scale_dim, float* xyz, float* features_dc, float* log_scales, float* rotations, float* logit_opacity,
                       char* workspace, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || width <= 0 || height <= 0 || (scale_dim != 1 && scale_dim != 3) || !(fx != 0.f) || !(fy != 0.f)) {
        g_last_error = "gsr_seed_from_rgbd: invalid size / intrinsics / scale_dim"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return 0;
    if (!pix || !depth || !image || !R || !T || !xyz || !features_dc || !log_scales || !rotations || !logit_opacity || !workspace) {
        g_last_error = "gsr_seed_from_rgbd: null argument"; return GSR_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(seed_backproject_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, pix, width, height, depth, image, exposure_a,
                       exposure_b, fx, fy, cx, cy, R, T, xyz, features_dc, rotations, logit_opacity);
    uintptr_t a = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255);
    float* dist2 = reinterpret_cast<float*>(a);
    char* knn_ws = reinterpret_cast<char*>((a + (size_t)n * sizeof(float) + 255) & ~uintptr_t(255));
    if (int rc = gsr_knn_mean_dist2(n, xyz, dist2, knn_ws, stream_)) return rc;
    hipLaunchKernelGGL(seed_scales_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, (const float*)dist2, point_size, scale_dim, log_scales);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_densify_select(int P, const float* xyz_gradient_accum, const float* denom, const float* log_scales, int scale_dim,
                       const float* logit_opacity, float grad_threshold, float dense_scale, float min_opacity, float big_scale, int* flags,
                       void* stream_)
{
    if (P < 0 || (scale_dim != 1 && scale_dim != 3)) { g_last_error = "gsr_densify_select: invalid size / scale_dim"; return GSR_ERR_INVALID_ARGUMENT; }
    if (P == 0) return 0;
    if (!xyz_gradient_accum || !denom || !log_scales || !logit_opacity || !flags) { g_last_error = "gsr_densify_select: null argument"; return GSR_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(densify_select_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, P, xyz_gradient_accum, denom, log_scales,
                       scale_dim, logit_opacity, grad_threshold, dense_scale, min_opacity, big_scale, flags);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_densify_apply(int P, const int* flags, const int* offsets, int n_keep, int n_clone, int n_split, int n_child, int ntensors,
                      const gsr_densify_tensor* tensors, const float* xyz, const float* log_scales, int scale_dim, const float* raw_rotations,
                      const float* noise, void* stream_)
{
    if (P < 0 || n_keep < 0 || n_clone < 0 || n_split < 0 || n_child < 0 || ntensors < 0 || ntensors > DENSIFY_MAX_TENSORS || (ntensors > 0 && !tensors)) {
        g_last_error = "gsr_densify_apply: invalid counts (at most 32 tensors)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (P == 0 || ntensors == 0) return 0;
    if (!flags || !offsets) { g_last_error = "gsr_densify_apply: null flags / offsets"; return GSR_ERR_INVALID_ARGUMENT; }
    if (n_child > 0 && (!xyz || !log_scales || !raw_rotations || !noise || (scale_dim != 1 && scale_dim != 3))) {
        g_last_error = "gsr_densify_apply: children need xyz, log_scales, raw_rotations and noise"; return GSR_ERR_INVALID_ARGUMENT;
    }
    DensifyArgs a;
    a.P = P; a.n_keep = n_keep; a.n_clone = n_clone; a.n_split = n_split; a.n_child = n_child; a.ntensors = ntensors; a.scale_dim = scale_dim;
    a.flags = flags; a.offsets = offsets; a.xyz = xyz; a.log_scales = log_scales; a.raw_rot = raw_rotations; a.noise = noise;
    const long long n_out = (long long)n_keep + n_clone + 2LL * n_child;
    for (int k = 0; k < ntensors; k++) {
        const gsr_densify_tensor& t = tensors[k];
        if (t.width <= 0 || !t.src || (n_out > 0 && !t.dst) || t.kind < GSR_DENSIFY_COPY || t.kind > GSR_DENSIFY_SCALE ||
            (t.kind == GSR_DENSIFY_XYZ && t.width != 3)) {
            g_last_error = "gsr_densify_apply: bad tensor descriptor"; return GSR_ERR_INVALID_ARGUMENT;
        }
        a.t[k] = DensifyTensor{t.src, t.dst, t.width, t.kind};
    }
    hipLaunchKernelGGL(densify_apply_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, a);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

static int camera_step_args(const gsr_camera_step* s, CameraStepArgs& a)
{
    if (!s || !s->R || !s->T || !s->viewmatrix || (s->full_proj && !s->projmatrix)) { g_last_error = "gsr_camera_step_launch: null pose / output"; return GSR_ERR_INVALID_ARGUMENT; }
    const bool any_grad = s->g_rot_delta || s->g_trans_delta || s->g_exposure_a || s->g_exposure_b;
    if (any_grad && (!s->exp_avg || !s->exp_avg_sq || !s->step)) { g_last_error = "gsr_camera_step_launch: Adam state missing"; return GSR_ERR_INVALID_ARGUMENT; }
    if ((s->g_rot_delta && !s->rot_delta) || (s->g_trans_delta && !s->trans_delta) || (s->g_exposure_a && !s->exposure_a) || (s->g_exposure_b && !s->exposure_b) ||
        (s->do_pose && (!s->rot_delta || !s->trans_delta))) {
        g_last_error = "gsr_camera_step_launch: gradient / pose step without its parameter"; return GSR_ERR_INVALID_ARGUMENT;
    }
    a.p[0] = s->rot_delta; a.g[0] = s->g_rot_delta; a.n[0] = 3; a.lr[0] = s->lr_rot;
    a.p[1] = s->trans_delta; a.g[1] = s->g_trans_delta; a.n[1] = 3; a.lr[1] = s->lr_trans;
    a.p[2] = s->exposure_a; a.g[2] = s->g_exposure_a; a.n[2] = 1; a.lr[2] = s->lr_exposure;
    a.p[3] = s->exposure_b; a.g[3] = s->g_exposure_b; a.n[3] = 1; a.lr[3] = s->lr_exposure;
    a.exp_avg = s->exp_avg; a.exp_avg_sq = s->exp_avg_sq; a.step = s->step; a.beta1 = s->beta1; a.beta2 = s->beta2; a.eps = s->eps;
    a.R = s->R; a.T = s->T; a.proj = s->projmatrix; a.view = s->viewmatrix; a.full = s->full_proj; a.campos = s->campos;
    a.converged = s->converged; a.thr = s->converged_threshold; a.do_pose = s->do_pose; a.latch = s->latch;
    return 0;
}

int gsr_camera_step_launch(const gsr_camera_step* s, void* stream_)
{
    CameraStepArgs a;
    { const int rc = camera_step_args(s, a); if (rc) return rc; }
    hipLaunchKernelGGL(camera_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, a);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t gsr_track_workspace_size(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    const size_t N = (size_t)width * height, T = (size_t)((width + TILE_X - 1) / TILE_X) * ((height + TILE_Y - 1) / TILE_Y);
    return (4 * N + 2 * T + 8) * sizeof(float);
}

int gsr_track_step(gsr_alloc_fn geometry_alloc, void* geometry_user, gsr_alloc_fn binning_alloc, void* binning_user, gsr_alloc_fn image_alloc,
                   void* image_user, int P, int D, int M, const float* background, int width, int height, const gsr_raw_inputs* in,
                   float scale_modifier, const float* projmatrix_raw, float tan_fovx, float tan_fovy, float* out_color, float* out_depth,
                   float* out_opacity, int* radii, int* n_touched, const gsr_track_loss* loss, const gsr_camera_step* step, float* dL_dmean2D,
                   char* workspace, void* stream)
{
    if (!in || !loss || !step || !workspace || !dL_dmean2D || !projmatrix_raw || P <= 0 || width <= 0 || height <= 0 || !loss->gt_image || !loss->gt_depth ||
        !step->full_proj || !step->campos || !step->rot_delta || !step->trans_delta || (step->exposure_a == nullptr) != (step->exposure_b == nullptr) ||
        (reinterpret_cast<uintptr_t>(workspace) & 15)) {
        g_last_error = "gsr_track_step: null / invalid argument (P > 0; loss targets, pose deltas, full_proj, campos, a 16-byte aligned workspace of gsr_track_workspace_size bytes)";
        return GSR_ERR_INVALID_ARGUMENT;
    }
    const size_t N = (size_t)width * height, T = (size_t)((width + TILE_X - 1) / TILE_X) * ((height + TILE_Y - 1) / TILE_Y);
    float* const ws = reinterpret_cast<float*>(workspace);
    float* const g_image = ws, * const g_depth = ws + 3 * N, * const partials = ws + 4 * N, * const tau6 = partials + 2 * T, * const g_exp = tau6 + 6;
    gsr_camera_step st = *step;                   // which tensors are stepped: the pose always, the exposure pair when it is given
    st.g_rot_delta = tau6 + 3; st.g_trans_delta = tau6;
    st.g_exposure_a = st.exposure_a ? g_exp : nullptr; st.g_exposure_b = st.exposure_b ? g_exp + 1 : nullptr;
    TrackTail tail;
    { const int rc = camera_step_args(&st, tail.step); if (rc) return rc; }
    tail.exposure_partials = partials; tail.dL_dexposure = g_exp;
    TrackLossArgs tl;
    tl.gt_image = loss->gt_image; tl.gt_depth = loss->gt_depth; tl.w_rgb = loss->w_rgb; tl.w_depth = loss->w_depth;
    tl.exposure_a = step->exposure_a; tl.exposure_b = step->exposure_b;
    tl.opacity_thr = loss->opacity_depth_threshold; tl.use_opacity = loss->opacity_weights ? 1 : 0;
    tl.c_rgb = loss->alpha / (3.0f * (float)N); tl.c_depth = (1.0f - loss->alpha) / (float)N;          // (make_loss_args)
    tl.dL_dimage = g_image; tl.dL_ddepth = g_depth; tl.partials = partials;
    CapturedAlloc g{geometry_alloc, geometry_user, nullptr}, b{binning_alloc, binning_user, nullptr}, i{image_alloc, image_user, nullptr};
    t_track_loss = &tl;
    const int R = forward_impl(captured_alloc, &g, captured_alloc, &b, captured_alloc, &i, P, D, M, background, width, height, nullptr, nullptr, nullptr,
                               nullptr, nullptr, scale_modifier, nullptr, nullptr, step->viewmatrix, step->full_proj, step->campos, tan_fovx, tan_fovy, 0,
                               out_color, out_depth, out_opacity, radii, n_touched, 0, stream, in);
    t_track_loss = nullptr;
    if (R < 0) return R;
    gsr_raw_grads none;
    memset(&none, 0, sizeof(none));
    t_track_tail = &tail;
    const int rc = backward_impl(P, D, M, R, background, width, height, nullptr, nullptr, nullptr, nullptr, scale_modifier, nullptr, nullptr, step->viewmatrix,
                                 step->full_proj, projmatrix_raw, step->campos, tan_fovx, tan_fovy, radii, g.got, b.got, i.got, g_image, g_depth, dL_dmean2D,
                                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, tau6, GSR_BACKWARD_POSE_ONLY, stream,
                                 in, &none);
    t_track_tail = nullptr;
    if (rc < 0) return rc;
    GSR_HIP_CHECK(hipGetLastError());
    return R;
}

int gsr_camera_steps_launch(int n, const gsr_camera_step* steps, void* stream_)
{
    if (n < 0 || n > CAMERA_STEPS_MAX || (n > 0 && !steps)) { g_last_error = "gsr_camera_steps_launch: 0..12 cameras"; return GSR_ERR_INVALID_ARGUMENT; }
    if (n == 0) return 0;
    CameraStepsArgs all;
    memset(&all, 0, sizeof(all));
    for (int k = 0; k < n; k++) { const int rc = camera_step_args(&steps[k], all.c[k]); if (rc) return rc; }
    hipLaunchKernelGGL(camera_steps_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream_, all);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_schedule_advance(int* counter, const unsigned int* table, int row_words, int rows, unsigned int* current, void* stream_)
{
    if (!counter || !table || !current || row_words < 1 || rows < 1) { g_last_error = "gsr_schedule_advance: null / empty argument"; return GSR_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(schedule_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, counter, (const uint32_t*)table, row_words, rows, (uint32_t*)current);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_slot_gather(int n_slots, const gsr_keyframe_entry* table, const int* index, const gsr_keyframe_entry* dst, int pixels, void* stream_)
{
    if (n_slots < 0 || n_slots > SLOTS_MAX || pixels < 1 || (n_slots > 0 && (!table || !index || !dst))) {
        g_last_error = "gsr_slot_gather: 0..4 slots, device table / index and a host array of destinations"; return GSR_ERR_INVALID_ARGUMENT;
    }
    if (n_slots == 0) return 0;
    static_assert(sizeof(gsr_keyframe_entry) == sizeof(KeyframeEntry), "gsr_keyframe_entry layout");
    SlotGatherArgs a;
    memset(&a, 0, sizeof(a));
    a.n_slots = n_slots; a.pixels = pixels; a.table = reinterpret_cast<const KeyframeEntry*>(table); a.index = index;
    for (int s = 0; s < n_slots; s++) {
        if (!dst[s].viewmatrix || !dst[s].full_proj || !dst[s].campos) { g_last_error = "gsr_slot_gather: a slot needs viewmatrix / full_proj / campos buffers"; return GSR_ERR_INVALID_ARGUMENT; }
        memcpy(&a.dst[s], &dst[s], sizeof(KeyframeEntry));
    }
    hipLaunchKernelGGL(slot_gather_kernel, dim3(SLOT_GATHER_BLOCKS, (unsigned)n_slots), dim3(256), 0, (hipStream_t)stream_, a);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_edge_mask(const float* image, int height, int width, float edge_threshold, float eps, float* intensity, float* median,
                  unsigned char* mask, void* stream_)
{
    if (!image || !intensity || !median || !mask || height < 2 || width < 2) { g_last_error = "gsr_edge_mask: null argument or image smaller than 2x2"; return GSR_ERR_INVALID_ARGUMENT; }
    hipStream_t stream = (hipStream_t)stream_;
    const int n = height * width;
    hipLaunchKernelGGL(edge_intensity_kernel, dim3((width + 15) / 16, (height + 15) / 16), dim3(256), 0, stream, image, height, width, eps, intensity);
    // scratch of the radix select (260 words per device, allocated once; the selects of one device are stream-ordered by their callers)
    static uint32_t* select_state[16] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!select_state[dev]) GSR_HIP_CHECK(hipMalloc((void**)&select_state[dev], 260 * sizeof(uint32_t)));
    GSR_HIP_CHECK(hipMemsetAsync(select_state[dev], 0, 260 * sizeof(uint32_t), stream));
    for (int shift = 24; shift >= 0; shift -= 8)      // torch.median: the lower one of the two middle elements
        hipLaunchKernelGGL(radix_select_pass_kernel, dim3(SELECT_BLOCKS), dim3(1024), 0, stream, (const float*)intensity, n, (n - 1) / 2, shift, select_state[dev], median);
    hipLaunchKernelGGL(edge_compare_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float*)intensity, n, (const float*)median, edge_threshold, mask);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t gsr_isotropic_loss_workspace_size(int P) { return P > 0 ? (size_t)((P + 255) / 256) * sizeof(float) + 16 : 16; }

int gsr_isotropic_loss_forward(int P, const float* raw_scales, float* loss, char* workspace, void* stream_)
{
    if (P < 0 || !loss || !workspace || (P > 0 && !raw_scales)) { g_last_error = "gsr_isotropic_loss_forward: invalid argument"; return GSR_ERR_INVALID_ARGUMENT; }
    const int nb = (P + 255) / 256;
    float* partial = reinterpret_cast<float*>(workspace);
    if (nb) hipLaunchKernelGGL(isotropic_partial_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream_, P, raw_scales, partial);
    hipLaunchKernelGGL(isotropic_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, nb, P, (const float*)partial, loss);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_isotropic_loss_backward(int P, const float* raw_scales, const float* g_loss, float* d_raw_scales, void* stream_)
{
    if (P < 0 || (P > 0 && (!raw_scales || !g_loss || !d_raw_scales))) { g_last_error = "gsr_isotropic_loss_backward: invalid argument"; return GSR_ERR_INVALID_ARGUMENT; }
    if (P) hipLaunchKernelGGL(isotropic_backward_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, P, raw_scales, g_loss, d_raw_scales);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_arap_forward(int V, int T, int M, int K, const float* p, const float* nb, const float* keep, float* R, float* partial, void* stream_)
{
    if (V < 0 || T < 2 || M < 0 || K < 1 || ((size_t)V * M > 0 && (!p || !nb || !keep || !R || !partial))) {
        g_last_error = "gsr_arap_forward: invalid argument (T >= 2, K >= 1, no null buffers)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const size_t n = (size_t)V * (T - 1) * M;
    if (n > 0x7fffffffull) { g_last_error = "gsr_arap_forward: too many (view, sample, node) triples"; return GSR_ERR_INVALID_ARGUMENT; }
    if (n) hipLaunchKernelGGL(arap_forward_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream_, V, T, M, K, p, nb, keep, R, partial);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_arap_backward(int V, int T, int M, int K, const float* p, const float* nb, const float* keep, const float* R, const float* g_partial,
                      float* dp, float* dnb, void* stream_)
{
    if (V < 0 || T < 2 || M < 0 || K < 1 || ((size_t)V * M > 0 && (!p || !nb || !keep || !R || !g_partial || !dp || !dnb))) {
        g_last_error = "gsr_arap_backward: invalid argument (T >= 2, K >= 1, no null buffers)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const size_t n = (size_t)V * M;
    if (n) hipLaunchKernelGGL(arap_backward_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream_, V, T, M, K, p, nb, keep, R, g_partial, dp, dnb);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_elastic_forward(int V, int M, int K, int T, const float* x, const float* nb, float* ratio, void* stream_)
{
    if (V < 0 || M < 0 || K < 1 || T < 2 || T > ELASTIC_MAX_T || ((size_t)V * M > 0 && (!x || !nb || !ratio))) {
        g_last_error = "gsr_elastic_forward: invalid argument (2 <= T <= 16, K >= 1, no null buffers)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const size_t n = (size_t)V * M * K;
    if (n) hipLaunchKernelGGL(elastic_forward_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream_, (int)n, M, K, T, x, nb, ratio);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_elastic_backward(int V, int M, int K, int T, const float* x, const float* nb, const float* g_ratio, float* dx, float* dnb, void* stream_)
{
    if (V < 0 || M < 0 || K < 1 || T < 2 || T > ELASTIC_MAX_T || ((size_t)V * M > 0 && (!x || !nb || !g_ratio || !dx || !dnb))) {
        g_last_error = "gsr_elastic_backward: invalid argument (2 <= T <= 16, K >= 1, no null buffers)"; return GSR_ERR_INVALID_ARGUMENT;
    }
    const size_t n = (size_t)V * M * K, nx = (size_t)V * M * T * 3;
    if (n) {
        hipLaunchKernelGGL(elastic_backward_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream_, (int)n, M, K, T, x, nb, g_ratio, dnb);
        hipLaunchKernelGGL(elastic_backward_self_kernel, dim3((unsigned)((nx + 63) / 64)), dim3(64), 0, (hipStream_t)stream_, (int)nx, K, T * 3, dnb, dx);
    }
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

int gsr_kabsch_rotations(int n, const float* S, float* R, void* stream_)
{
    if (n < 0 || (n > 0 && (!S || !R))) { g_last_error = "gsr_kabsch_rotations: null argument"; return GSR_ERR_INVALID_ARGUMENT; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(kabsch_rotation_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream_, n, S, R);
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
