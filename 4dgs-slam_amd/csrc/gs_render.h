// gs_render.h -- the two per-tile kernels: front-to-back compositing (forward) and the back-to-front gradient pass.
// gfx950 / wave64. DGR = submodules/diff-gaussian-rasterization.
//
// Work decomposition (both kernels): 256-thread blocks on a 16x16 tile (the reference's binning granularity, which is part
// of the result), four wavefronts, wave w owning the 8x8-pixel quadrant (w&1, w>>1), one pixel per lane. The forward kernel
// runs one block per tile (and sorts the tile's list first), the backward kernel one block per 128-entry chunk of a tile list.
// List entries are staged through LDS (gathered by Gaussian id). While staging, each thread tests its entry against the four
// quadrants -- the region where alpha = o*exp(power) can reach 1/255 is an ellipse, and the minimum of its quadratic form over
// a quadrant's pixel rectangle has a closed form -- and four 64-bit wave ballots per staging wave give every quadrant a bit
// mask of the entries that can touch it at all. Each wave then walks ONLY the set bits of its masks (scalar s_ff1 loop), so a
// (wave, Gaussian) pair that cannot contribute costs nothing: no exp, no reduction. The cull is conservative (a superset of
// the pairs the reference blends; verified bit-identical against a build without it), so results are unchanged.
#pragma once
#include <type_traits>
#include "gs_forward.h"

namespace gsr {

constexpr int RB = 256;   // entries per staged batch == threads per block
constexpr int IDX_STRIDE = 70;   // per (quadrant wave, 64-entry group) index list: 64 rows + 4 the prefetch may touch + the read-ahead of the list itself
typedef float f2 __attribute__((ext_vector_type(2)));   // arithmetic on f2 lowers to v_pk_{add,mul,fma}_f32: two fp32 ops per issue slot

// Which of the four 8x8 quadrants of tile (tx,ty) can a Gaussian contribute to at all?
// alpha = o*exp(power) reaches 1/255 only where d^T Q d <= 2 tau, tau = ln(255 o), Q = [[a,b],[b,c]] (the conic): an ellipse
// around the mean. The test is the exact minimum of that quadratic form over the quadrant's pixel-centre rectangle: a convex
// function on a convex set, so the minimum is 0 if the mean lies inside and otherwise sits on an edge FACING the mean (every
// segment from the mean into the rectangle enters through one), where it is a clamped 1-D parabola. Two candidates cover
// all cases: the point on the line x = clamp(mean.x) with the best y, and the point on y = clamp(mean.y) with the best x.
// Slack (absolute, relative to tau, and relative to the magnitude of the cancelling terms) makes fp rounding only ever ADD
// quadrants: the per-pixel test of the reference stays in the kernels, so results are unchanged.
// Returns a 4-bit mask (bit q = quadrant q: 0 = top-left, 1 = top-right, 2 = bottom-left, 3 = bottom-right).
__device__ __forceinline__ uint32_t quadrant_mask(float gx, float gy, float a, float b, float c, float o, int tx, int ty)
{
    const float tau = __logf(255.0f * o);           // alpha >= 1/255  <=>  power >= -tau
    if (!(tau >= 0.f)) return 0u;                   // o < 1/255 (or NaN): can never be blended
    if (!(a * c - b * b > 0.f && a > 0.f && c > 0.f)) return 15u;          // degenerate conic: never cull
    const float thr = 2.0f * tau * 1.0005f + 2e-3f;
    const float nboa = -b * __builtin_amdgcn_rcpf(a), nboc = -b * __builtin_amdgcn_rcpf(c), b2 = 2.0f * b;
    const float X0 = (float)(tx * TILE_X) - gx, Y0 = (float)(ty * TILE_Y) - gy;
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float dx0 = X0 + (q & 1) * 8.f, dx1 = dx0 + 7.f, dy0 = Y0 + (q >> 1) * 8.f, dy1 = dy0 + 7.f;
        const float dxc = __builtin_amdgcn_fmed3f(0.f, dx0, dx1), dyc = __builtin_amdgcn_fmed3f(0.f, dy0, dy1);
        const float dyv = __builtin_amdgcn_fmed3f(nboc * dxc, dy0, dy1);   // best y on the line x = clamp(mean.x)
        const float dxh = __builtin_amdgcn_fmed3f(nboa * dyc, dx0, dx1);   // best x on the line y = clamp(mean.y)
        const float v1 = a * dxc * dxc, v2 = b2 * dxc * dyv, v3 = c * dyv * dyv;
        const float h1 = a * dxh * dxh, h2 = b2 * dxh * dyc, h3 = c * dyc * dyc;
        const float qv = (v1 + v2 + v3) - 1e-5f * (v1 + fabsf(v2) + v3);
        const float qh = (h1 + h2 + h3) - 1e-5f * (h1 + fabsf(h2) + h3);
        if (fminf(qv, qh) <= thr) mask |= 1u << q;
    }
    return mask;
}

// GSR_EXACT_MATH (csrc/build.sh --exact -> libgs_rasterizer_hip_exact.so, selected with GSR_EXACT_MATH=1): the two tile kernels evaluate a
// pair exactly like the reference / the fp32 oracle -- power = -0.5 (a dx^2 + c dy^2) - b dx dy in that order, alpha = min(0.99, o *
// exp(power)) with a correctly rounded exponential (evaluated in double), true division in the backward recurrence -- and the whole
// library is compiled without floating-point contraction. The default build folds log2(e) and log2(o) into the exponent and uses the
// hardware's v_exp_f32 / v_rcp_f32 (1 ulp); tests/test_hip_exact_math.py measures what that changes (discrete outputs: none or a
// handful of threshold flips; images 1e-7). Staging layout in exact mode: s_a = {mean.x, mean.y, a, b}, s_b = {c, o}.
#ifndef GSR_EXACT_MATH
#define GSR_EXACT_MATH 0
#endif

__device__ __forceinline__ float exact_power(float dx, float dy, float a, float b, float c) { return -0.5f * (a * dx * dx + c * dy * dy) - b * dx * dy; }
__device__ __forceinline__ float exact_exp(float x) { return (float)exp((double)x); }


// pop_lowest_bit + row = j * 16 in a VGPR: the byte offset of staged entry j inside every 16-byte-per-entry LDS array. One VALU
// instruction (the compiler's s_lshl + v_mov pair costs an SALU slot more) and, being volatile, it is neither recomputed nor
// re-materialised: all LDS reads of a pair use this one register with constant offsets. One asm block: between two separate blocks
// the hazard recogniser puts an s_nop it cannot prove unnecessary.
__device__ __forceinline__ int pop_lowest_bit_row16(unsigned long long& m, uint32_t& row)
{
    int j;
    asm volatile("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0\n\tv_lshlrev_b32 %2, 4, %0" : "=&s"(j), "+s"(m), "=v"(row));
    return j;
}
// mask ? a : b per lane, as the one VALU instruction it is (opaque to the optimiser: see render_fwd_kernel's pair loop)
typedef unsigned long long lanemask;
__device__ __forceinline__ float lane_select(lanemask m, float a, float b)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
template <typename T>
__device__ __forceinline__ T lds_at(const void* array, uint32_t byte_offset)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(array) + byte_offset);
}

__device__ __forceinline__ unsigned long long lds_mask_uniform(const unsigned long long* p)
{
    const unsigned long long m = *p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)m), hi = __builtin_amdgcn_readfirstlane((uint32_t)(m >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// GSR_FWD_TIMING (dev builds only): per-wave cycle accounting of render_fwd_kernel, read back with gsr_debug_fwd_timing().
#ifndef GSR_FWD_TIMING
#define GSR_FWD_TIMING 0
#endif
#if GSR_FWD_TIMING
__device__ uint32_t g_fwd_timing[8 * 4 * 8192];
__device__ uint32_t g_bwd_timing[8 * 4 * 8192];
#define FWD_TICK() ((uint32_t)__builtin_amdgcn_s_memtime())
#define FWD_T(...) __VA_ARGS__
#else
#define FWD_T(...)
#endif
// GSR_FWD_TIMING == 2: render_fwd's batch loop in finer pieces instead of the sort's sub-phases (report slots 2, 3, 6, 7: staging, the two
// block barriers, index lists, epilogue)
#if GSR_FWD_TIMING == 2
#define FWD_T2(...) __VA_ARGS__
#else
#define FWD_T2(...)
#endif
// GSR_TIMELINE (dev builds only): per-block start / end on the chip-wide 100 MHz clock (s_memrealtime) and the block's HW_ID / XCC_ID, for
// both tile kernels -> tools/tile_timeline.py reconstructs residency over time, dispatch gaps and the tail (gsr_debug_spans()).
#ifndef GSR_TIMELINE
#define GSR_TIMELINE 0
#endif
// dev A/B knobs (tools/dev_ab.sh): wave priority by list length in render_fwd; cache policy of the large streaming stores
#ifndef GSR_FWD_PRIO
#define GSR_FWD_PRIO 0
#endif
#ifndef GSR_FWD_PRIO_T1
#define GSR_FWD_PRIO_T1 8
#define GSR_FWD_PRIO_T2 9
#define GSR_FWD_PRIO_T3 10
#endif
#ifndef GSR_FWD_PREFETCH
#define GSR_FWD_PREFETCH 0
#endif
#ifndef GSR_CKPT_STORE
#define GSR_CKPT_STORE 0      // render_fwd's checkpoints: 0 plain, 1 nontemporal (dev A/B: no difference measured)
#endif
__device__ __forceinline__ void ckpt_store(float* p, float v)
{
#if GSR_CKPT_STORE == 1
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
#if GSR_TIMELINE
__device__ uint32_t g_spans[2][8192][4];
#define TL_BEGIN() const uint32_t tl_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime()
#define TL_END(which, id)                                                                                                             \
    do {                                                                                                                              \
        if (threadIdx.x == 0 && (id) < 8192) {                                                                                        \
            g_spans[which][id][0] = tl_t0; g_spans[which][id][1] = (uint32_t)__builtin_amdgcn_s_memrealtime();                        \
            g_spans[which][id][2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); g_spans[which][id][3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   \
        }                                                                                                                             \
    } while (0)
#else
#define TL_BEGIN()
#define TL_END(which, id)
#endif
// ------------------------------------------------------------------------------------------------------------------
// F5: tile compositing, DGR/cuda_rasterizer/forward.cu:263-392.
// ------------------------------------------------------------------------------------------------------------------
template <bool TRACK = false>     // TRACK: the tracking loss's cotangents in the epilogue (gsr_track_step); a kernel of its own, so the plain one keeps its registers
__device__ __forceinline__ void render_fwd_body(int ntiles, int gx, const uint2* __restrict__ ranges,
                                                        const uint2* sorted /* may alias sorted_out */, int W, int H,
                                                        const TileRec* __restrict__ rec,
                                                        const float* __restrict__ bg, float* __restrict__ final_T,
                                                        uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
                                                        float* __restrict__ out_depth, float* __restrict__ out_opacity,
                                                        int* __restrict__ n_touched, float4* __restrict__ final_C,
                                                        float* __restrict__ ckpt, const uint32_t* __restrict__ spec_header,
                                                        const uint64_t* __restrict__ keys, const uint32_t* __restrict__ inst_gauss,
                                                        uint2* sorted_out, const uint32_t* __restrict__ chunk_base, uint4* __restrict__ chunk_info,
                                                        const uint32_t* __restrict__ tile_pos, int fwd_order, const TrackLossArgs& tl)
{
    if (spec_header && (spec_header[HDR_FLAGS] & FLAG_OVERFLOW)) return;   // speculative launch on a buffer that turned out too small
    TL_BEGIN();
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[3 * RB * sizeof(float4)];
    float4* const s_a = reinterpret_cast<float4*>(s_raw);   // {mean.x, mean.y, A, B}   power*log2e = dx*(A*dx + B*dy) + C*dy*dy
    float4* const s_b = s_a + RB;                            // {C, log2 opacity, -, gaussian id bits}
    float4* const s_c = s_b + RB;                            // {r, g, b, depth}: two packed FMAs per blended entry
    __shared__ unsigned long long s_mask[4][4];   // [quadrant][staging wave]
    __shared__ int s_nt[RB];                      // per-entry n_touched increments of this tile, flushed once per batch
    __shared__ __attribute__((aligned(8))) uint32_t s_idx[4][4][IDX_STRIDE];   // [quadrant wave][64-entry group]: LDS row offsets of the entries its mask keeps

    int tile = xcd_tile_of_block(blockIdx.x, ntiles);
    if (fwd_order) tile = (int)__builtin_amdgcn_readfirstlane((int)tile_pos[(size_t)tile * CTR_STRIDE + POS_FWD_TILE]);   // gs_forward.h F3c: same band, dealt by length
    const int tx = tile % gx, ty = tile / gx;
    const int t = threadIdx.x, lane = lane_id(), wave = t >> 6;
    const int px = tx * TILE_X + (wave & 1) * 8 + (lane & 7);
    const int py = ty * TILE_Y + (wave >> 1) * 8 + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const bool inside = px < W && py < H;
    // A pixel that is done (saturated, forward.cu:358-362, or outside the image, :292) is encoded in its alpha threshold:
    // 1/255 while it still blends, +inf afterwards. One VGPR compare then replaces the lane-mask bookkeeping a `bool done`
    // costs in the loop (six SALU instructions per pair).
    const float INF = __builtin_inff();
    float thr = inside ? 1.0f / 255.0f : INF;
    // TRACK: the loss's operands of this pixel are requested now and consumed in the epilogue (there they would be a memory round trip at the
    // end of every block of a launch that lasts 15 us at tracking sizes)
    float tl_gt[3] = {0.f, 0.f, 0.f}, tl_gtd = 0.f, tl_wr = 1.f, tl_wd = 1.f, tl_a = 0.f, tl_b = 0.f;
    if (TRACK && tl.gt_image != nullptr && inside) {
        const size_t N = (size_t)H * W, pix = (size_t)py * W + px;
        tl_gt[0] = tl.gt_image[pix]; tl_gt[1] = tl.gt_image[N + pix]; tl_gt[2] = tl.gt_image[2 * N + pix];
        tl_gtd = tl.gt_depth[pix];
        if (tl.w_rgb) tl_wr = tl.w_rgb[pix];
        if (tl.w_depth) tl_wd = tl.w_depth[pix];
        if (tl.exposure_a) tl_a = tl.exposure_a[0];
        if (tl.exposure_b) tl_b = tl.exposure_b[0];
    }
    s_nt[t] = 0;
    float T = 1.0f;
    f2 acc_rg = {0.f, 0.f}, acc_bd = {0.f, 0.f};   // (C.r, C.g) and (C.b, D): accumulated with v_pk_fma_f32
    const f2 pxy = {pxf, pyf};
    uint32_t last = 0;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    // (read here, with the range, for the epilogue's work-item table: at the end they would be one more memory round trip of the last block)
    uint32_t nfull = 0, fbase = 0, prank = 0, part_home = 0, fs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (tile_pos != nullptr) {
        const uint32_t* const frame = tile_pos + (size_t)ntiles * CTR_STRIDE;
        nfull = frame[POS_TOTAL_FULL]; part_home = frame[POS_PART_HOME];
#pragma unroll
        for (int x = 0; x < 9; x++) fs[x] = frame[POS_FULL_START + x];
        fbase = tile_pos[(size_t)tile * CTR_STRIDE + POS_FULL_BASE]; prank = tile_pos[(size_t)tile * CTR_STRIDE + POS_PART_RANK];
    }
#if GSR_FWD_PRIO
    {   // The launch lasts as long as its longest tile's dependent chain (every tile of the frame is resident at once, 4.7 waves per SIMD share the
        // issue slots): the waves of the long lists take the arbitration (s_setprio), the short ones -- which finish early anyway -- the rest.
        const int mean = (int)(ranges[ntiles - 1].y / (uint32_t)ntiles);            // segments are exact: the last range ends at R
        if (n * 8 > mean * GSR_FWD_PRIO_T3) __builtin_amdgcn_s_setprio(3);
        else if (n * 8 > mean * GSR_FWD_PRIO_T2) __builtin_amdgcn_s_setprio(2);
        else if (n * 8 > mean * GSR_FWD_PRIO_T1) __builtin_amdgcn_s_setprio(1);
    }
#endif
    FWD_T(uint32_t tk0 = FWD_TICK(); uint32_t tk_sort = 0, tk_stage = 0, tk_list = 0, tk_pair = 0, tk_wait = 0, n_pairs = 0, n_batches = 0; uint32_t tk_mark = tk0;)
    // keys != nullptr: this block first sorts its own tile list (the staging arrays double as the key buffer) -- one kernel and
    // one GPU drain/fill less per frame than a separate sort launch; lists beyond the LDS capacity were sorted by
    // sort_tiles_kernel<SORT_LDS_CAP, SORT_SMALL_CAP> before.
    static_assert(3 * RB * sizeof(float4) >= padded_keys_size(SORT_SMALL_CAP) * sizeof(uint64_t), "key buffer aliases the staging arrays");
    if (keys != nullptr && n > 0 && n <= SORT_SMALL_CAP) {
#if GSR_FWD_TIMING
        uint32_t st[2] = {0, 0};
        sort_tile_in_lds<0>(range, keys, inst_gauss, sorted_out, PaddedKeys{reinterpret_cast<uint64_t*>(s_raw)}, st);
        const uint32_t tk_s2 = FWD_TICK();
        __syncthreads();
        tk_list = st[0] - tk0; n_batches = st[1] - st[0]; tk_wait = tk_s2 - st[1];      // dev: key load | network | rank + gather + store (reuses three report slots)
#else
        sort_tile_in_lds<0>(range, keys, inst_gauss, sorted_out, PaddedKeys{reinterpret_cast<uint64_t*>(s_raw)});
        __syncthreads();                                   // the sorted list (global) and the LDS buffer are reused below
#endif
    }
    FWD_T(tk_sort = FWD_TICK() - tk0; tk_mark = FWD_TICK();)
    // Checkpoint of the per-pixel compositing state in front of list entry `boundary` (a multiple of CHUNK): lets the backward
    // pass start at any chunk of the list instead of walking the whole list from its end (render_bwd_kernel). Five coalesced
    // 256-byte stores per wave and CHUNK entries. A wave whose pixels are all saturated stops writing them: the backward
    // pass uses the final values for a pixel that blended nothing behind the boundary.
    auto write_checkpoint = [&](int boundary) {
        float* c = ckpt + (size_t)((range.x >> 7) + (uint32_t)(boundary >> 7)) * CKPT_FLOATS + t;
        ckpt_store(c, T); ckpt_store(c + 256, acc_rg.x); ckpt_store(c + 512, acc_rg.y); ckpt_store(c + 768, acc_bd.x); ckpt_store(c + 1024, acc_bd.y);
    };
    static_assert(CHUNK == 128 && RB == 2 * CHUNK, "checkpoint cadence: one at the top of a batch, one in its middle");

#if GSR_FWD_PREFETCH
    // Round 6: the NEXT batch's list entries and records are requested while the current batch is composited. The launch is as long as its
    // longest tile's dependent chain (tools/tile_timeline.py: every block starts within 0.5 us, the longest lasts the whole launch, nothing
    // queues behind it), and that chain paid two dependent global latencies (entry -> record) in front of every 256-entry batch with the
    // block's four waves parked on them. pf_gid2: Gaussian id of this thread's entry of the batch AFTER the next one (requested a whole pair
    // phase before its record load needs it), pf_q*: this thread's record of the batch about to be staged. 13 more registers (88 -> 5 waves
    // per SIMD allowed; a 640x480 frame is 4.7).
    uint32_t pf_gid = 0, pf_gid2 = 0;
    float4 pf_q0 = make_float4(0.f, 0.f, 0.f, 0.f), pf_q1 = pf_q0, pf_q2 = pf_q0;
    if (t < n) pf_gid = sorted[range.x + t].x;
    if (RB + t < n) pf_gid2 = sorted[range.x + RB + t].x;
    if (t < n) { const TileRec* const g = rec + pf_gid; pf_q0 = g->q0; pf_q1 = g->q1; pf_q2 = g->q2; }
#endif
    FWD_T2(uint32_t t2_stage = 0, t2_bar = 0, t2_list = 0, t2_epi = 0;)
    for (int base = 0; base < n; base += RB) {
        FWD_T2(tk_mark = FWD_TICK();)
        const int all_done = __syncthreads_and(thr > 1.0f);       // forward.cu:318-320 (also orders the LDS reuse below)
        FWD_T2(t2_bar += FWD_TICK() - tk_mark;)
        FWD_T(tk_mark = FWD_TICK();)
        {   // flush the previous batch's n_touched increments: one global atomic per (tile, Gaussian), off the hot loop
            const int c = s_nt[t];
            if (c) { atomicAdd(&n_touched[__float_as_uint(s_b[t].w)], c); s_nt[t] = 0; }
        }
        if (all_done) break;
        if (base > 0) write_checkpoint(base);
        uint32_t qm = 0;
        if (base + t < n) {
#if GSR_FWD_PREFETCH
            const uint2 e = make_uint2(pf_gid, 0u);
            const float4 q0 = pf_q0, q1 = pf_q1, q2 = pf_q2;
#else
            const uint2 e = sorted[range.x + base + t];
            const TileRec* const g = rec + e.x;
            const float4 q0 = g->q0, q1 = g->q1, q2 = g->q2;
#endif
            const float2 xy = make_float2(q0.x, q0.y);
            const float4 co = make_float4(q1.x, q1.y, q1.z, q0.w);
            qm = quadrant_mask(xy.x, xy.y, co.x, co.y, co.z, co.w, tx, ty);
#if GSR_EXACT_MATH
            s_a[t] = make_float4(xy.x, xy.y, co.x, co.y);
            s_b[t] = make_float4(co.z, co.w, 0.f, __uint_as_float(e.x));
#else
            s_a[t] = make_float4(xy.x, xy.y, -0.5f * LOG2E * co.x, -LOG2E * co.y);
            s_b[t] = make_float4(-0.5f * LOG2E * co.z, __log2f(co.w), 0.f, __uint_as_float(e.x));   // log2(opacity): folded into the exponent
#endif
            s_c[t] = make_float4(q2.x, q2.y, q2.z, q0.z);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned long long m = __ballot((qm >> q) & 1u);
            if (lane == 0) s_mask[q][wave] = m;
        }
        FWD_T2(t2_stage += FWD_TICK() - tk_mark; tk_mark = FWD_TICK();)
        __syncthreads();
        FWD_T2(t2_bar += FWD_TICK() - tk_mark; tk_mark = FWD_TICK();)
#if GSR_FWD_PREFETCH
        pf_gid = pf_gid2;
        if (base + RB + t < n) { const TileRec* const g = rec + pf_gid; pf_q0 = g->q0; pf_q1 = g->q1; pf_q2 = g->q2; }
        if (base + 2 * RB + t < n) pf_gid2 = sorted[range.x + base + 2 * RB + t].x;
#endif
        // The saturation test is per 64-entry group, not per entry: a per-entry wave vote + branch serialises the loop on the
        // VALU->SALU round trip (measured: 111 -> 86 us), and pixels that are done blend nothing anyway.
        // Per 64-entry group the wave re-votes two things: whether any of its pixels is still unsaturated (else it skips the
        // group), and whether any pixel still has T > 0.5 -- only then can an entry bump n_touched (forward.cu:369-371), and
        // the loop variant without that bookkeeping is shorter.
        // n_touched bookkeeping: the wave visits each entry of the group once, so the count of entry jj is WRITTEN into lane jj
        // of a VGPR (s_bcnt1, then a plain lane == jj select: v_writelane would need m0, which inline asm must not clobber) and the
        // 64 counts go to LDS with one ds_add per group.
        //
        // The pair loop runs over a compacted INDEX LIST and is software-pipelined: each quadrant wave first turns its four ballot masks
        // into four lists of LDS row offsets (v_mbcnt ranks, one ds_write per group); the loop is then counted, keeps two entries in
        // flight (register sets A / B, each refilled right after its pair has been evaluated) and reads the rows two entries ahead,
        // so no pair waits for its ds_reads. What that bought is small (69 -> 67 us): per-wave cycle accounting (tools/phase_cycles.py,
        // profiles/r03_phase_cycles.json) shows this phase at ~310 cycles per pair and wave with 4.7 waves per SIMD all in it at once
        // = ~2 cycles per wave-instruction per SIMD: the SIMDs' issue rate, not latency, bounds the pair phase. Chunk-parallel
        // compositing (one block per 128 entries) would therefore add its 40 % of extra pair work on top of an
        // already saturated phase; what idles the chip is the SORT phase in front of it (every tile sorts at the same time).
        int cnt4[4];
#pragma unroll
        for (int sw = 0; sw < 4; sw++) {
            const unsigned long long mk = lds_mask_uniform(&s_mask[wave][sw]);
            cnt4[sw] = (int)__popcll(mk);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            uint32_t* const L = &s_idx[wave][sw][0];
            if ((mk >> lane) & 1ull) L[pos] = (uint32_t)(sw * 64 + lane) * 16u;
            if (lane < 4) L[cnt4[sw] + lane] = 0u;             // rows the prefetch may touch behind the list's end (never consumed)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // the lists are wave-private: DS operations of one wave execute in order
        __builtin_amdgcn_wave_barrier();
        struct Entry { float4 A4; float2 B2; float4 C4; uint32_t row; };
        auto fetch = [&](uint32_t row) {
            Entry e;
            e.A4 = lds_at<float4>(s_a, row);
            e.B2 = lds_at<float2>(s_b, row);
            e.C4 = lds_at<float4>(s_c, row);
            e.row = row;
            return e;
        };
        auto composite = [&](auto COUNT_TOUCHED, unsigned long long m, int sw, int cnt) {
            int counts = 0;
            uint32_t last_row = ~0u;                       // LDS row offset of the last entry this group blended into the pixel
            const uint32_t* const L = &s_idx[wave][sw][0];
            auto pair = [&](const Entry& e) {
                const float4 A4 = e.A4; const float2 B2 = e.B2; const float4 C4 = e.C4;
                const f2 d = f2{A4.x, A4.y} - pxy;
                // alpha = o exp(power) = exp2(power log2e + log2 o): the opacity rides in the exponent (one multiply less per pair)
#if GSR_EXACT_MATH
                const float pw = exact_power(d.x, d.y, A4.z, A4.w, B2.x);                           // forward.cu:345
                const float alpha = fminf(0.99f, B2.y * exact_exp(pw));                             // :353
                const lanemask validm = __builtin_amdgcn_ballot_w64(pw <= 0.0f) & __builtin_amdgcn_ballot_w64(alpha >= thr);   // :346, :354 and "not done"
#else
                const float pw = d.x * (A4.z * d.x + A4.w * d.y) + (B2.x * d.y * d.y + B2.y);      // forward.cu:345 (times log2 e) + log2 o
                const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(pw));                      // :353
                const lanemask validm = __builtin_amdgcn_ballot_w64(pw <= B2.y) & __builtin_amdgcn_ballot_w64(alpha >= thr);   // :346 (power <= 0), :354 and "not done"
#endif
                // lane masks and selects spelled out (v_cmp -> SGPR pair, s_and / s_xor, v_cndmask): left to itself hipcc turns the selects of
                // the loop's last pair into divergent branches (s_and_saveexec + two blocks)
                const float test_T = T * (1.0f - alpha);
                const lanemask stopm = validm & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);    // :358-362
                const lanemask blendm = validm ^ stopm;
                thr = lane_select(stopm, INF, thr);
                const float w = lane_select(blendm, alpha * T, 0.0f);
                acc_rg += f2{C4.x, C4.y} * w;                                                       // :364-367
                acc_bd += f2{C4.z, C4.w} * w;
                T = lane_select(blendm, test_T, T);
                last_row = __float_as_uint(lane_select(blendm, __uint_as_float(e.row), __uint_as_float(last_row)));   // `contributor`, :338,:376 (resolved below)
                if (COUNT_TOUCHED.value) {
                    const int jj = pop_lowest_bit(m);       // the list is in bit order: the mask walk names the same entry
                    const lanemask tm = blendm & __builtin_amdgcn_ballot_w64(test_T > 0.5f);             // blend && test_T > 0.5, :369-371
                    counts = lane == jj ? (int)__popcll(tm) : counts;      // plain VALU select: no m0 / v_writelane (whose two scalar operands exceed the constant bus)
                }
            };
            // ping-pong: entry k in register set A, entry k + 1 in set B; each set is refilled right after its pair has been evaluated,
            // i.e. one full pair evaluation before it is needed again; the rows come from the list two entries ahead
            uint2 ic = *reinterpret_cast<const uint2*>(L);          // rows of entries 0, 1
            uint2 in = *reinterpret_cast<const uint2*>(L + 2);      // rows of entries 2, 3
            Entry e0 = fetch(ic.x), e1 = fetch(ic.y);
            for (int k = 0; k + 1 < cnt; k += 2) {
                __builtin_amdgcn_sched_barrier(0);
                const uint2 in2 = *reinterpret_cast<const uint2*>(L + k + 4);   // issued first: DS results return in order, so the copy below
                __builtin_amdgcn_sched_barrier(0);                              // waits for this read only, not for the entry reads behind it
                pair(e0);
                e0 = fetch(in.x);
                __builtin_amdgcn_sched_barrier(0);
                pair(e1);
                e1 = fetch(in.y);
                in = in2;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (cnt & 1) pair(e0);
            // entries are visited in list order, so the group's last blended entry is the pixel's new `last` (1-based list position)
            last = last_row != ~0u ? (uint32_t)(base + 1) + (last_row >> 4) : last;
            if (COUNT_TOUCHED.value) {
                if (counts) __hip_atomic_fetch_add(&s_nt[sw * 64 + lane], counts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        FWD_T2(t2_list += FWD_TICK() - tk_mark;)
        FWD_T(tk_mark = FWD_TICK();)
        for (int sw = 0; sw < 4; sw++) {
            if (__all(thr > 1.0f)) break;
            if (sw == 2 && base + CHUNK < n) write_checkpoint(base + CHUNK);
            const int cnt = sw == 0 ? cnt4[0] : sw == 1 ? cnt4[1] : sw == 2 ? cnt4[2] : cnt4[3];
            if (cnt == 0) continue;
            FWD_T(n_pairs += cnt;)
            if (__any(thr < 1.0f && T > 0.5f)) composite(std::true_type{}, lds_mask_uniform(&s_mask[wave][sw]), sw, cnt);
            else composite(std::false_type{}, 0ull, sw, cnt);
        }
        FWD_T(tk_pair += FWD_TICK() - tk_mark; tk_mark = FWD_TICK();)
    }
#if GSR_FWD_TIMING
    if (lane == 0 && tile < 8192) {
        uint32_t* o = g_fwd_timing + (size_t)(tile * 4 + wave) * 8;
        o[0] = FWD_TICK() - tk0; o[1] = tk_sort; o[2] = tk_stage; o[3] = tk_list; o[4] = tk_pair; o[5] = n_pairs; o[6] = n_batches; o[7] = tk_wait;
        FWD_T2(o[2] = t2_stage; o[3] = t2_bar; o[6] = t2_list; o[7] = 0;)
    }
#endif
    FWD_T2(tk_mark = FWD_TICK();)
    __syncthreads();
    {
        const int c = s_nt[t];
        if (c) atomicAdd(&n_touched[__float_as_uint(s_b[t].w)], c);
    }
    if (chunk_info != nullptr) {
        // Work items of render_bwd_kernel: this tile's CHUNK-entry pieces (gs_device.h), written at the block index that kernel's XCD
        // banding gives the chunk. Bit 16: more entries behind (a checkpoint exists at the chunk's back end); bit 17: NO pixel of the
        // tile blended an entry at or behind the chunk's first -- the backward block then only zeroes the chunk's slots and never
        // requests pixel state, checkpoint or records (on long lists most chunks lie behind every pixel's saturation point: issuing
        // the whole prologue for them doubled render_bwd's time at 3.5-14 M instances).
        __shared__ int s_deepest[4];
        int wm = (int)last;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wm = max(wm, __shfl_xor(wm, d, 64));
        if (lane == 0) s_deepest[wave] = wm;
        __syncthreads();
        const int deepest = max(max(s_deepest[0], s_deepest[1]), max(s_deepest[2], s_deepest[3]));
        const uint32_t cb = chunk_base[tile], nchunks = chunk_base[ntiles];
        // where the pieces go: tile order inside XCD bands (tile_pos == nullptr, rounds 2-5), or full pieces in tile order + partial pieces
        // longest first at the end of every XCD's sequence (gs_device.h: item_block_*)
        for (int c = t; c * CHUNK < n; c += RB) {
            const int cstart = c * CHUNK, m = min(CHUNK, n - cstart);
            uint32_t block;
            if (tile_pos == nullptr) block = (uint32_t)xcd_block_of_tile((int)(cb + (uint32_t)c), (int)nchunks);
            else if (!part_home) block = m == CHUNK ? item_block_full(nchunks, nchunks - nfull, fbase + (uint32_t)c) : item_block_partial(nchunks, nchunks - nfull, prank);
            else if (m == CHUNK) {                      // full piece of rank f: the XCD whose range holds f
                const uint32_t f = fbase + (uint32_t)c;
                uint32_t x = 0, start = fs[0];          // (selects, not fs[x]: a run-time index would send the table to scratch memory)
#pragma unroll
                for (int k = 1; k < 8; k++) if (fs[k] <= f) { x = (uint32_t)k; start = fs[k]; }
                block = 8u * (f - start) + x;
            } else {                                    // partial piece: behind the full pieces of the XCD of this tile's band
                const int q8 = ntiles >> 3, r8 = ntiles & 7, head8 = r8 * (q8 + 1);
                const uint32_t x = (uint32_t)(tile < head8 ? tile / (q8 + 1) : r8 + (tile - head8) / q8);
                uint32_t lo = fs[0], hi = fs[1];
#pragma unroll
                for (int k = 1; k < 8; k++) if (x >= (uint32_t)k) { lo = fs[k]; hi = fs[k + 1]; }
                block = 8u * ((hi - lo) + prank) + x;
            }
            chunk_info[block] = make_uint4((uint32_t)tile, range.x + (uint32_t)cstart,
                                           (uint32_t)m | (cstart + m < n ? 0x10000u : 0u) | (deepest <= cstart ? 0x20000u : 0u), (uint32_t)cstart);
        }
    }
    const float out_rgb[3] = {acc_rg.x + T * bg[0], acc_rg.y + T * bg[1], acc_bd.x + T * bg[2]};    // forward.cu:384-390
    const float out_op = 1.0f - T;
    if (inside) {
        const size_t pix = (size_t)py * W + px;
        final_T[pix] = T;
        n_contrib[pix] = last;
        final_C[pix] = make_float4(acc_rg.x, acc_rg.y, acc_bd.x, acc_bd.y);   // colour / depth without the background term
        out_color[pix] = out_rgb[0];
        out_color[(size_t)H * W + pix] = out_rgb[1];
        out_color[2 * (size_t)H * W + pix] = out_rgb[2];
        out_depth[pix] = acc_bd.y;
        out_opacity[pix] = out_op;
    }
    if (TRACK && tl.gt_image != nullptr) {
        // Tracking (gsr_track_step): the weighted L1 loss's cotangents of this pixel, formed from the values still in registers -- what
        // l1_loss_bwd_kernel computes from the stored image one launch later (same function, same bits) -- and the tile's share of the two
        // exposure gradients (256 pixels in a fixed order; the tail kernel adds the tiles).
        const size_t N = (size_t)H * W, pix = (size_t)py * W + px;
        float da = 0.f, db = 0.f;
        if (inside) {
            const float ea = tl.exposure_a ? expf(tl_a) : 1.f, eb = tl_b;
            float wr = tl_wr * tl.c_rgb, wd = tl_wd * tl.c_depth;
            if (tl.use_opacity) { wr *= out_op; wd = out_op > tl.opacity_thr ? wd : 0.f; }
            const L1PixelGrad o = l1_bwd_pixel(wr, wd, ea, eb, out_rgb, tl_gt, acc_bd.y, tl_gtd, da, db);
            tl.dL_dimage[pix] = o.gi[0]; tl.dL_dimage[N + pix] = o.gi[1]; tl.dL_dimage[2 * N + pix] = o.gi[2];
            tl.dL_ddepth[pix] = o.gd;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { da += __shfl_xor(da, d, 64); db += __shfl_xor(db, d, 64); }
        __shared__ float s_exp[4][2];
        if (lane == 0) { s_exp[wave][0] = da; s_exp[wave][1] = db; }
        __syncthreads();
        if (t < 2) tl.partials[2 * (size_t)tile + t] = (s_exp[0][t] + s_exp[1][t]) + (s_exp[2][t] + s_exp[3][t]);
    }
    FWD_T2(if (lane == 0 && tile < 8192) { uint32_t* o = g_fwd_timing + (size_t)(tile * 4 + wave) * 8; o[7] = FWD_TICK() - tk_mark; o[0] = FWD_TICK() - tk0; })
    TL_END(0, tile);
}

// ------------------------------------------------------------------------------------------------------------------
// B1: render backward, DGR/cuda_rasterizer/backward.cu:563-787, restructured:
//  * per-(quadrant, Gaussian) sums over the quadrant's 64 pixels use the transposed butterfly of gs_device.h (37 VALU
//    instructions for all ten values); the reference runs a 256-thread shared-memory tree with 8 block barriers x 5
//    arrays for every listed Gaussian (backward.cu:541-559,759-765);
//  * the four quadrant waves drop their totals into LDS (ten lanes, one ds_write each); after the batch the block adds
//    the quadrants in a fixed order and writes one 40-byte slot per (tile, Gaussian) instance, coalesced. The reference
//    issues 10 float atomics per instance instead (backward.cu:774-783). B2 then sums a Gaussian's consecutive
//    instance slots in a fixed order: bit-reproducible gradients, nothing to zero-fill, no atomics;
//  * besides the quadrant cull, entries behind the deepest contributor of the quadrant (max n_contrib) are dropped
//    at staging time, so saturated regions skip their occluded tail entirely;
//  * one block per CHUNK entries of a tile list, not per tile: the forward pass checkpoints the per-pixel compositing state
//    every CHUNK entries, so every chunk can be differentiated on its own (details at the state set-up below).
// Slot (SLOT_FLOATS = 10): dmean2D.x, dmean2D.y, dconic.x, dconic.y, dconic.w, dopacity, dcolor.r, dcolor.g, dcolor.b, ddepth
// ------------------------------------------------------------------------------------------------------------------
constexpr int TB = 8;       // entries per batch of the LDS-transposed reduction
// value of lane (lane & ~7) + I: the I-th lane of this lane's group of eight (ds_swizzle in bit-mask mode: and_mask 0x18, or_mask I inside
// each half of the wave; the LDS crossbar moves the data, no LDS memory is touched)
template <int I>
__device__ __forceinline__ float row_broadcast(float v)
{
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x18 | (I << 5)));
}
constexpr int BB = CHUNK;   // entries per block of the backward kernel
constexpr int GRP = 64;     // entries per group: the staging arrays and the quadrant totals in LDS cover one group at a time
constexpr int PART_STRIDE = 10;   // floats per (quadrant, entry) in s_part: {M1x, M1y, M2xx, M2xy | M2yy, sum q, r, g | b, depth} (40-byte rows: 8-byte aligned pieces)

__device__ __forceinline__ void render_bwd_body(int ntiles, int gx, const char* bin_base,
                                                        const uint32_t* __restrict__ header, int W, int H,
                                                        const float* __restrict__ bg, const TileRec* __restrict__ rec,
                                                        const float* __restrict__ final_T,
                                                        const float4* __restrict__ final_C, const uint32_t* __restrict__ n_contrib,
                                                        const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_depth)
{
    // ---- which (tile, chunk) is this block? The grid is an upper bound (R / CHUNK + tiles); surplus blocks leave.
    // The XCD banding is computed over the REAL number of chunks: banding over the grid (an upper bound) put every surplus id
    // into the last XCD's band, which then ran out of work while the other seven still had a quarter of theirs.
    FWD_T(const uint32_t tk0 = FWD_TICK(); uint32_t tk_search = 0, tk_state = 0, tk_stage = 0, tk_pair = 0, tk_epi = 0, n_pairs = 0; uint32_t tk_mark = tk0;)
    TL_BEGIN();
    // The block's work item and the header are two INDEPENDENT loads (the item table sits at offset 0 of the binning buffer and is indexed by
    // the block id: the forward pass applied the XCD banding when it wrote it). Round 2 searched chunk_base for the tile -- eleven dependent
    // scalar loads -- and then read ranges / chunk_base: seven global latencies before the first list entry could be requested, now two.
    const uint4 item = reinterpret_cast<const uint4*>(((reinterpret_cast<uintptr_t>(bin_base) + 255) & ~uintptr_t(255)))[blockIdx.x];
    const uint32_t nchunks = header[HDR_CHUNKS];
    if (blockIdx.x >= nchunks || (header[HDR_FLAGS] & FLAG_OVERFLOW)) return;   // surplus block (its item is stale), or a lazy forward pass whose lists were never built
    const BinningPtrs bin = carve_binning(const_cast<char*>(bin_base), header[HDR_CARVE_R], header[HDR_CAP_SORTED], (size_t)ntiles);   // uniform: SALU
    const int tile = (int)__builtin_amdgcn_readfirstlane(item.x);
    FWD_T(tk_search = FWD_TICK() - tk_mark; tk_mark = FWD_TICK();)
    const uint2* __restrict__ sorted = bin.sorted;
    float* __restrict__ partials = reinterpret_cast<float*>(bin.partials);
    __shared__ float4 s_a[GRP];  // {mean.x, mean.y, A, B}   power*log2e = dx*(A*dx + B*dy) + C*dy*dy  (A = -a/2 log2e, B = -b log2e)
    __shared__ float4 s_b[GRP];  // {C, log2 opacity, -, -}                                             (C = -c/2 log2e)
    __shared__ float4 s_c[GRP];  // {r, g, b, depth}
    // The three arrays share one index scale, so a pair addresses all of them from ONE VGPR (j * 16 + constant offset): a float2 s_b cost
    // the loop a second shift + move per pair.
    // LDS budget (round 4): the staging arrays hold ONE 64-entry group. Thread t < 128 loads and prepares entry t of the chunk up front
    // (one global round trip for the whole chunk) but only the first group goes to LDS at once; the threads of wave 1 keep their entry in
    // twelve registers and store it when the first group's pair loops are done. Together with the transposition area below that is
    // 31.1 KiB per block: FIVE blocks share a CU. With both groups staged (34.1 KiB, four blocks) the kernel took 102-104 us; the footprint
    // alone is worth 10 % (profiles/r04_*: half of this kernel's wave-cycles are parked on memory, LDS or barriers, and a single wave issues
    // one instruction per ~8 cycles, so every resident wave counts). Splitting the TOTALS into 32-entry groups instead was built and
    // measured: bit-identical, 5 blocks per CU, but four more barriers and twice the partial batches -- 108 us.
    // Quadrant totals of ONE 64-entry group (10 floats per (quadrant, entry): the values of the three 16-byte pieces of a slot).
    __shared__ __attribute__((aligned(16))) float s_part[4][GRP][PART_STRIDE];
    __shared__ unsigned long long s_mask[4][2];
    __shared__ unsigned long long s_proc[4];      // [quadrant]: entries of the current group whose totals the quadrant wave actually wrote
    __shared__ int s_wmax[4];
    // Wave-private transposition area (round 4): the per-(pixel, entry) weights of a batch of TB entries, written with lane = pixel and
    // read back with lane = (pixel row, entry) -- see the pair loop. Row stride TB + 1 float2: both access patterns are bank-conflict free.
    __shared__ __attribute__((aligned(16))) float2 s_wq[4][64 * (TB + 1)];

    const int tx = tile % gx, ty = tile / gx;
    const int t = threadIdx.x, lane = lane_id(), wave = t >> 6;
    const int px = tx * TILE_X + (wave & 1) * 8 + (lane & 7);
    const int py = ty * TILE_Y + (wave >> 1) * 8 + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t first = __builtin_amdgcn_readfirstlane(item.y);          // position of list entry cstart in sorted[]
    const int m = (int)(__builtin_amdgcn_readfirstlane(item.z) & 0xFFFFu);
    const bool more_behind = (__builtin_amdgcn_readfirstlane(item.z) & 0x10000u) != 0;
    const int cstart = (int)__builtin_amdgcn_readfirstlane(item.w), cend = cstart + m;   // list positions [cstart, cend) of this tile, front to back

    if (__builtin_amdgcn_readfirstlane(item.z) & 0x20000u) {                 // nothing of this chunk was blended by any pixel of the tile: its
        if (t < m) {                                                         // instances' slots must still be written (zero), nothing else is read
            const uint2 e0 = sorted[first + (uint32_t)(m - 1 - t)];
            slot_store(partials, e0.y, make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float2(0.f, 0.f));
        }
        TL_END(1, blockIdx.x);
        return;
    }
    const bool inside = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    // Every global load of the prologue is issued here, in one go, before anything waits: the list entry (whose Gaussian record is the only
    // load that depends on another one), the pixel's forward results and cotangents, and the checkpoint at the chunk's back end. Taken in
    // program order (contributor count -> block vote -> pixel state -> list entry -> record) they cost five global latencies in a row.
    uint2 e = make_uint2(0u, 0u);
    if (t < m) e = sorted[first + (uint32_t)(m - 1 - t)];                    // 0-based list position cend-1-t, back to front (:656,:677)
    const int last_contrib = inside ? (int)n_contrib[pix] : 0;
    const float Tfin = inside ? final_T[pix] : 0.f;                          // backward.cu:617-623
    const float gr = inside ? dL_dpix[pix] : 0.f;                            // :629-635
    const float gg = inside ? dL_dpix[(size_t)H * W + pix] : 0.f;
    const float gb = inside ? dL_dpix[2 * (size_t)H * W + pix] : 0.f;
    const float gd = inside ? dL_dpix_depth[pix] : 0.f;
    const bool have_ckpt = inside && more_behind;                            // a checkpoint exists in front of entry cend (it may be stale: see below)
    float4 Cf = make_float4(0.f, 0.f, 0.f, 0.f);
    float ck[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (have_ckpt) {
        const float* c = bin.ckpt + (size_t)(((first - (uint32_t)cstart) >> 7) + (uint32_t)(cend >> 7)) * CKPT_FLOATS + t;
        Cf = final_C[pix];
        ck[0] = c[0]; ck[1] = c[256]; ck[2] = c[512]; ck[3] = c[768]; ck[4] = c[1024];
    }
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
    if (t < m) { const TileRec* const g = rec + e.x; q0 = g->q0; q1 = g->q1; q2 = g->q2; }
    {   // deepest list position any pixel of this quadrant blended; a chunk behind all four has nothing to do (its instances'
        // slots must still be written: zero)
        int wm = last_contrib;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wm = max(wm, __shfl_xor(wm, d, 64));
        if (lane == 0) s_wmax[wave] = wm;
    }
    __syncthreads();
    if (max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])) <= cstart) {
        if (t < m) {
            slot_store(partials, e.y, make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float2(0.f, 0.f));
        }
        TL_END(1, blockIdx.x);
        return;
    }

    // ---- per-pixel state at the BACK end of the chunk (the pass walks the chunk back to front, backward.cu:656,677) --------
    //   T  = transmittance in front of entry `cend`:  the forward pass's checkpoint there, or the final value for a pixel that
    //        blended nothing at or behind cend (in particular for the last chunk of the list; the forward pass stops writing
    //        checkpoints once a quadrant is saturated, so the loaded one is only used when the pixel blended behind it);
    //   Sb = T_final (bg . g) + sum over the blended entries at or behind cend of alpha_k T_k (c_k . g)
    //      = (C_final - C_checkpoint) . g + T_final (bg . g),   C = the forward pass's running colour / depth sums.
    // The reference carries the normalised "colour behind" accum_rec[3] + accum_rec_depth and last_alpha/last_color (:714-728)
    // from the END of the list and adds the background term separately (:738-743); with acc_i = S_i / (T_i (1 - alpha_i)) its
    //   dL_dalpha_i = (c_i - acc_i).g T_i - T_final/(1 - alpha_i) bg.g   becomes   (c_i . g) T_i - Sb_i / (1 - alpha_i),
    // which needs no per-channel state and -- because an invalid pair simply has alpha = 0 -- no selects on the state; and
    // since both T and Sb are available at every chunk boundary, every chunk of a list is an independent block: ~5000 short
    // blocks instead of 1200 long ones, which the hardware dispatcher balances over the CUs (with one block per tile the
    // kernel lasted as long as its slowest tile, 1.5x the mean).
    const float bgdot = bg[0] * gr + bg[1] * gg + bg[2] * gb;                // :738-742
    float T = Tfin, Sb = Tfin * bgdot;
    if (have_ckpt && last_contrib > cend) {
        T = ck[0];
        Sb += (Cf.x - ck[1]) * gr + (Cf.y - ck[2]) * gg + (Cf.z - ck[3]) * gb + (Cf.w - ck[4]) * gd;
    }
    const f2 pxy = {pxf, pyf}, g_rg = {gr, gg}, g_bd = {gb, gd};

    FWD_T(tk_state = FWD_TICK() - tk_mark; tk_mark = FWD_TICK();)
    uint32_t qm = 0;
    float4 st_a = make_float4(0.f, 0.f, 0.f, 0.f), st_b = st_a, st_c = st_a;   // this thread's entry, prepared: what s_a / s_b / s_c hold + {instance id, opacity}
    if (t < m) {
        const int pos = cend - 1 - t;
        const float2 xy = make_float2(q0.x, q0.y);
        const float4 co = make_float4(q1.x, q1.y, q1.z, q0.w);
        qm = quadrant_mask(xy.x, xy.y, co.x, co.y, co.z, co.w, tx, ty);
#pragma unroll
        for (int q = 0; q < 4; q++) if (pos >= s_wmax[q]) qm &= ~(1u << q);   // behind everything this quadrant blended (:678)
#if GSR_EXACT_MATH
        st_a = make_float4(xy.x, xy.y, co.x, co.y);
        st_b = make_float4(co.z, co.w, __uint_as_float(e.y), co.w);
#else
        st_a = make_float4(xy.x, xy.y, -0.5f * LOG2E * co.x, -LOG2E * co.y);
        st_b = make_float4(-0.5f * LOG2E * co.z, __log2f(co.w), __uint_as_float(e.y), co.w);   // log2(opacity): folded into the exponent
#endif
        st_c = make_float4(q2.x, q2.y, q2.z, q0.z);
        if (t < GRP) { s_a[t] = st_a; s_b[t] = st_b; s_c[t] = st_c; }
    }
    // pos < last_contrib (:678)  <=>  j >= cend - last_contrib, with j the index inside this chunk
    const int j_thr = cend - last_contrib;
    if (wave < 2) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned long long mk = __ballot((qm >> q) & 1u);
            if (lane == 0) s_mask[q][wave] = mk;
        }
    }
    __syncthreads();
    FWD_T(tk_stage = FWD_TICK() - tk_mark; tk_mark = FWD_TICK();)
    unsigned long long proc = 0;
    int group_base = 0;
    // ---- round 4: the per-entry sums over the quadrant's 64 pixels WITHOUT cross-lane traffic in the pair loop -----------------------------
    // The reduction above costs a pair 23 VALU instructions + 8 permlane swaps (5-8 issue cycles each, profiles/r02_ubench_issue.json) -- 40 %
    // of the loop. Instead, a pair only WRITES its two per-pixel weights (q = o G dL_dalpha and wv = alpha T) to a wave-private LDS matrix
    // [pixel][entry of the batch]; after TB = 8 entries the wave reads the matrix back TRANSPOSED -- lane (row h, entry c) walks the 8 pixels
    // of quadrant row h for entry c -- and accumulates the ten sums as plain per-lane FMAs: every moment is q times a polynomial in
    // (dx, dy), dy is constant along a row, so the row needs S0 = sum q, S1 = sum q dx, S2 = sum q dx^2 (M1y = dy S0, M2xy = dy S1,
    // M2yy = dy^2 S0) and the four colour sums wv * cotangent: 10 instructions per (pixel, entry), i.e. per PAIR (8 lanes share an entry).
    // The eight rows of an entry are then added with ONE halving butterfly per batch (9 swaps + 12 adds for 8 entries, not per entry).
    // Sums are formed in a fixed order: bit-reproducible like before (the order differs from round 3's, so the last bits do too).
    const int c7 = lane & 7, hrow = lane >> 3;
    // two planes (q, wv) of [pixel][TB + 1] floats: a pair writes one float per plane (ds_write2_b32), the transposed pass reads the values of
    // two neighbouring pixels with one ds_read2_b32 into a register PAIR -- its arithmetic is packed (v_pk_*: two pixels per instruction)
    constexpr int PLANE = 64 * (TB + 1);
    float* const wq_base = reinterpret_cast<float*>(&s_wq[wave][0]);
    char* const wq_write = reinterpret_cast<char*>(wq_base + lane * (TB + 1));
    const float* const wq_read = wq_base + hrow * 8 * (TB + 1) + c7;
    // the cotangents of the eight pixels of this lane's row, fetched ONCE per block from the lanes that own the pixels (ds_swizzle: the LDS
    // crossbar, no LDS memory), as pixel PAIRS per channel: 32 registers instead of a second LDS matrix
    f2 grow_r[4], grow_g[4], grow_b[4], grow_d[4];
#define GSR_ROW_COT(k)                                                                                                                        \
    grow_r[k] = f2{row_broadcast<2 * k>(gr), row_broadcast<2 * k + 1>(gr)}; grow_g[k] = f2{row_broadcast<2 * k>(gg), row_broadcast<2 * k + 1>(gg)};   \
    grow_b[k] = f2{row_broadcast<2 * k>(gb), row_broadcast<2 * k + 1>(gb)}; grow_d[k] = f2{row_broadcast<2 * k>(gd), row_broadcast<2 * k + 1>(gd)};
    GSR_ROW_COT(0) GSR_ROW_COT(1) GSR_ROW_COT(2) GSR_ROW_COT(3)
#undef GSR_ROW_COT
    const float x0qf = (float)(tx * TILE_X + (wave & 1) * 8), pyrow = (float)(ty * TILE_Y + (wave >> 1) * 8 + hrow);
    const uint32_t part_gid = (uint32_t)(((lane >> 5) & 1) * 2 + ((lane >> 4) & 1));                   // which three of the twelve row values this lane ends with
    char* const part_write = reinterpret_cast<char*>(&s_part[wave][0][0]) + part_gid * 12u;
    uint32_t jrow = 0;          // lanes with (lane & 7) == c: 16 x the group-relative index of the entry in batch slot c (its LDS row offset)
    int nslot = 0;              // filled slots of the current batch (uniform)
    auto bwd_pair_t = [&](int jj, uint32_t row) {
        const int j = group_base + jj;
        const float4 A4 = lds_at<float4>(s_a, row);
        const float2 B2 = lds_at<float2>(s_b, row);
        const f2 d = f2{A4.x, A4.y} - pxy;
#if GSR_EXACT_MATH
        const float pw = exact_power(d.x, d.y, A4.z, A4.w, B2.x);                            // :684
        const float G = B2.y * exact_exp(pw);
        const float alpha = fminf(0.99f, G);                                                  // :688
        const bool valid = j >= j_thr && pw <= 0.0f && alpha >= 1.0f / 255.0f;              // :678,:685,:689
#else
        const float pw = d.x * (A4.z * d.x + A4.w * d.y) + (B2.x * d.y * d.y + B2.y);       // :684 (times log2 e) + log2 o
        const float G = __builtin_amdgcn_exp2f(pw);
        const float alpha = fminf(0.99f, G);                                                  // :688 (clamp has no gradient mask, Q23)
        const bool valid = j >= j_thr && pw <= B2.y && alpha >= 1.0f / 255.0f;              // :678,:685 (power <= 0),:689
#endif
        if (!__any(valid)) return;            // the reference's skip_counter shortcut (:691-697); the entry stays out of `proc`
        const float4 C4 = lds_at<float4>(s_c, row);                              // {r, g, b, depth}
        const float av = valid ? alpha : 0.f;                                                 // an invalid pair blends nothing
        const float Gv = valid ? G : 0.f;             // G may be +inf where power > 0: keep it out of the products
#if GSR_EXACT_MATH
        const float inv1ma = 1.0f / (1.f - av);                                                // :700 true division
#else
        const float inv1ma = __builtin_amdgcn_rcpf(1.f - av);
#endif
        T *= inv1ma;                                                                           // :700
        const float wv = av * T;                                                               // :701 dchannel_dcolor
        const f2 cgp = f2{C4.x, C4.y} * g_rg + f2{C4.z, C4.w} * g_bd;
        const float cg = cgp.x + cgp.y;                                                        // colour.dL_dpixel + depth*dL_ddepth
        const float dL_dalpha = cg * T - Sb * inv1ma;                                          // :718-743, see above
        Sb += wv * cg;
        const float q = Gv * dL_dalpha;       // = o G dL_dalpha
        float* const w = reinterpret_cast<float*>(wq_write + nslot * 4);
        w[0] = q; w[PLANE] = wv;                                                               // one ds_write2_b32
        jrow = c7 == nslot ? row : jrow;
        asm("s_bitset1_b64 %0, %1" : "+s"(proc) : "s"(jj));
        nslot++;
    };
    auto flush_batch = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the matrix is wave-private: DS operations of one wave execute in order
        __builtin_amdgcn_wave_barrier();
        const float2 mean = lds_at<float2>(s_a, jrow);
        const float dxb = mean.x - x0qf, dy = mean.y - pyrow;
        f2 S0 = {0.f, 0.f}, S1 = {0.f, 0.f}, S2 = {0.f, 0.f}, cr = {0.f, 0.f}, cg2 = {0.f, 0.f}, cb = {0.f, 0.f}, cd = {0.f, 0.f};
        f2 dx2 = {dxb, dxb - 1.0f};
#pragma unroll
        for (int k = 0; k < 4; k++) {                  // pixels 2k, 2k + 1 of the row
            const float* r = wq_read + 2 * k * (TB + 1);
            const f2 q2 = {r[0], r[TB + 1]}, w2 = {r[PLANE], r[PLANE + TB + 1]};
            const f2 tq = q2 * dx2;
            S0 += q2; S1 += tq; S2 += tq * dx2;
            cr += grow_r[k] * w2; cg2 += grow_g[k] * w2; cb += grow_b[k] * w2; cd += grow_d[k] * w2;
            dx2 -= f2{2.0f, 2.0f};
        }
        const float s0 = S0.x + S0.y, s1 = S1.x + S1.y, s2 = S2.x + S2.y;
        // the row's twelve values in s_part order: {M1x, M1y, M2xx | M2xy, M2yy, sum q | r, g, b | depth, -, -}
        float V[12] = {s1, dy * s0, s2, dy * s1, dy * dy * s0, s0, cr.x + cr.y, cg2.x + cg2.y, cb.x + cb.y, cd.x + cd.y, 0.f, 0.f};
        // add the eight rows (lane bits 3-5): halve the value set at bits 5 and 4 (a lane keeps 6, then 3 of the 12), plain add at bit 3
#pragma unroll
        for (int k = 0; k < 6; k++) { lane_swap32(V[k], V[k + 6]); V[k] += V[k + 6]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { lane_swap16(V[k], V[k + 3]); V[k] += V[k + 3]; }
#pragma unroll
        for (int k = 0; k < 3; k++) V[k] += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(V[k]), 0x128, 0xf, 0xf, false));   // row_ror:8
        if ((lane & 8) == 0 && c7 < nslot) {
            float* o = reinterpret_cast<float*>(part_write + (jrow >> 4) * (uint32_t)(PART_STRIDE * 4));
            o[0] = V[0];
            if (part_gid != 3u) { o[1] = V[1]; o[2] = V[2]; }        // (the fourth triple is {depth, -, -}: the row ends behind its first value)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        nslot = 0;
    };
    for (int sw = 0; sw < BB / GRP; sw++) {
        unsigned long long mk = lds_mask_uniform(&s_mask[wave][sw]);
        proc = 0;
        group_base = sw * GRP;
        FWD_T(n_pairs += (uint32_t)__popcll(mk);)
        for (int left = (int)__popcll(mk); left > 0; left--) {          // counted: a scalar compare per trip (`while (mk)` compiled to a VALU 64-bit compare)
            uint32_t row;
            const int jj = pop_lowest_bit_row16(mk, row);
            bwd_pair_t(jj, row);
            if (nslot == TB) flush_batch();
        }
        if (nslot) flush_batch();
        if (lane == 0) s_proc[wave] = proc;
        FWD_T(tk_pair += FWD_TICK() - tk_mark; tk_mark = FWD_TICK();)
        __syncthreads();
        // Group epilogue, by the wave whose threads prepared the group's entries (wave sw: thread = entry, its conic, opacity and instance
        // id are still in registers): add the four quadrants in a fixed order, turn the moments into the reference's gradients and write the
        // entry's instance slot (12 floats, 48 B). Meanwhile wave 1 stages the second group (the first one's rows are dead: every wave has
        // left its pair loop).
        if (wave == sw) {
            if (t < m) {
                float sum[PART_STRIDE];
#pragma unroll
                for (int k = 0; k < PART_STRIDE; k++) sum[k] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if ((s_proc[q] >> lane) & 1ull) {
                        const float2* r = reinterpret_cast<const float2*>(&s_part[q][lane][0]);          // 40-byte rows: five 8-byte reads
#pragma unroll
                        for (int k = 0; k < PART_STRIDE / 2; k++) { const float2 v = r[k]; sum[2 * k] += v.x; sum[2 * k + 1] += v.y; }
                    }
                }
#if GSR_EXACT_MATH
                const float4 K4 = make_float4(st_a.z, st_a.w, st_b.x, st_b.w);                                   // {conic.x, conic.y, conic.z, opacity}
#else
                // the unscaled conic: undoes the scaling of A, B, C (one rounding, <= 1.5 ulp on the factor of dL_dmean2D)
                const float4 K4 = make_float4(st_a.z * (-2.0f / LOG2E), st_a.w * (-1.0f / LOG2E), st_b.x * (-2.0f / LOG2E), st_b.w);
#endif
                // {M1x, M1y, M2xx, M2xy} -> dL_dmean2D (:749-753 with ddelx_dx, :643), dL_dconic.x, .y (:754-755)
                // {M2yy, sum q, r, g} -> dL_dconic.w (:756), dL_dopacity = sum q / o = sum G dL_dalpha (:757; o = 0 blends nowhere), colour r, g (:719)
                // {b, depth} (:719,:729)
                slot_store(partials, __float_as_uint(st_b.z),
                           make_float4(-(K4.x * sum[0] + K4.y * sum[1]) * (0.5f * W), -(K4.z * sum[1] + K4.y * sum[0]) * (0.5f * H), -0.5f * sum[2], -0.5f * sum[3]),
                           make_float4(-0.5f * sum[4], K4.w > 0.f ? sum[5] / K4.w : 0.f, sum[6], sum[7]), make_float2(sum[8], sum[9]));
            }
        } else if (sw == 0 && wave == 1 && t < m) {
            s_a[lane] = st_a; s_b[lane] = st_b; s_c[lane] = st_c;
        }
        FWD_T(tk_epi += FWD_TICK() - tk_mark; tk_mark = FWD_TICK();)
        if (m <= GRP) break;            // a short chunk: no second group
        if (sw == 0) __syncthreads();   // the second group is staged; s_part and s_proc are free again
    }
#if GSR_FWD_TIMING
    if (lane == 0 && blockIdx.x < 8192) {
        uint32_t* o = g_bwd_timing + (size_t)(blockIdx.x * 4 + wave) * 8;
        o[0] = FWD_TICK() - tk0; o[1] = tk_search; o[2] = tk_state; o[3] = tk_stage; o[4] = tk_pair; o[5] = n_pairs; o[6] = tk_epi; o[7] = tk0;
    }
#endif
    TL_END(1, blockIdx.x);
}

// ---- the single-view kernels: the bodies above with their arguments passed by value (gs_views.h launches the same bodies once for
// several views) ----------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RB) render_fwd_kernel(int ntiles, int gx, const uint2* __restrict__ ranges,
                                                        const uint2* sorted /* may alias sorted_out */, int W, int H,
                                                        const TileRec* __restrict__ rec,
                                                        const float* __restrict__ bg, float* __restrict__ final_T,
                                                        uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
                                                        float* __restrict__ out_depth, float* __restrict__ out_opacity,
                                                        int* __restrict__ n_touched, float4* __restrict__ final_C,
                                                        float* __restrict__ ckpt, const uint32_t* __restrict__ spec_header,
                                                        const uint64_t* __restrict__ keys, const uint32_t* __restrict__ inst_gauss,
                                                        uint2* sorted_out, const uint32_t* __restrict__ chunk_base, uint4* __restrict__ chunk_info,
                                                        const uint32_t* __restrict__ tile_pos, int fwd_order)
{
    render_fwd_body<false>(ntiles, gx, ranges, sorted, W, H, rec, bg, final_T, n_contrib, out_color, out_depth, out_opacity, n_touched, final_C, ckpt, spec_header, keys, inst_gauss, sorted_out, chunk_base, chunk_info, tile_pos, fwd_order, TrackLossArgs{});
}

// the same tile kernel with the tracking loss's cotangents formed in its epilogue (gsr_track_step)
__global__ void __launch_bounds__(RB) render_fwd_track_kernel(int ntiles, int gx, const uint2* __restrict__ ranges,
                                                        const uint2* sorted /* may alias sorted_out */, int W, int H,
                                                        const TileRec* __restrict__ rec,
                                                        const float* __restrict__ bg, float* __restrict__ final_T,
                                                        uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
                                                        float* __restrict__ out_depth, float* __restrict__ out_opacity,
                                                        int* __restrict__ n_touched, float4* __restrict__ final_C,
                                                        float* __restrict__ ckpt, const uint32_t* __restrict__ spec_header,
                                                        const uint64_t* __restrict__ keys, const uint32_t* __restrict__ inst_gauss,
                                                        uint2* sorted_out, const uint32_t* __restrict__ chunk_base, uint4* __restrict__ chunk_info,
                                                        const uint32_t* __restrict__ tile_pos, int fwd_order, TrackLossArgs tl)
{
    render_fwd_body<true>(ntiles, gx, ranges, sorted, W, H, rec, bg, final_T, n_contrib, out_color, out_depth, out_opacity, n_touched, final_C, ckpt, spec_header, keys, inst_gauss, sorted_out, chunk_base, chunk_info, tile_pos, fwd_order, tl);
}

#ifndef GSR_BWD_WAVES
#define GSR_BWD_WAVES 5
#endif
__global__ void __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(GSR_BWD_WAVES, 8))) render_bwd_kernel(int ntiles, int gx, const char* bin_base,
                                                        const uint32_t* __restrict__ header, int W, int H,
                                                        const float* __restrict__ bg, const TileRec* __restrict__ rec,
                                                        const float* __restrict__ final_T,
                                                        const float4* __restrict__ final_C, const uint32_t* __restrict__ n_contrib,
                                                        const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_depth)
{
    render_bwd_body(ntiles, gx, bin_base, header, W, H, bg, rec, final_T, final_C, n_contrib, dL_dpix, dL_dpix_depth);
}


}  // namespace gsr
