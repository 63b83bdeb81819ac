// gs_device.h -- device-side helpers shared by the gfx950 rasterizer kernels.
// wave64 only: every cross-lane helper hard-codes 64 lanes (CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

constexpr int TILE_X = 16;   // DGR/cuda_rasterizer/config.h:15-17 (the binning granularity is part of the result)
constexpr int TILE_Y = 16;
constexpr int TILE_PIX = TILE_X * TILE_Y;
constexpr float LOG2E = 1.4426950408889634f;

// Spherical-harmonics constants, DGR/cuda_rasterizer/auxiliary.h:22-39
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                       -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- transposed ("butterfly") wave reduction of ten values -------------------------------------------------------
// Sums ten per-lane values over the 64 lanes in 23 VALU instructions (a plain butterfly: 60). At every stage a lane keeps
// only HALF of its values, adding the partner lane's copy of that half, instead of carrying all ten through all six stages;
// on return every lane holds the wave-wide total of ONE of the ten (wave_sum10_slot_of_lane() tells which), so the totals
// leave the wave with a single store instruction.
//   xor 32, xor 16: gfx950's v_permlane32_swap / v_permlane16_swap exchange half of register A with the other half of
//     register B; A + B then holds A's pair sums in one half of the lanes and B's in the other: 2 instructions per pair
//     of values, 1.5 with v_pk_add_f32 when both registers of two pairs sit in aligned register pairs -- which is how the
//     tile kernel produces them ((M1x,M1y), (M2xx,M2xy), (r,g), (b,depth) are f2 values). Ten values -> five -> three.
//   xor 1, xor 2: DPP quad_perm adds; keep/send selection with v_cndmask (3 instructions per pair). Three values -> one.
//   xor 8, xor 4: nothing left to select, plain DPP adds (row_ror:8, then row_ror:4 of the now 8-periodic value).
// The swaps use the compiler builtins (hipcc places the VALU-write -> permlane-read wait states itself); the DPP part is
// inline asm because hipcc does not fold DPP into the adds (it emits v_mov_dpp + v_add), with s_nop where a DPP source was
// written less than two issue slots earlier.
// Which total a lane ends with: bit 1 set -> bit 5 ? M2yy : s_op; else bit 0 ? (second components) : (first components),
// bit 4 ? colours : moments, bit 5 ? (q2 / c_bd) : (q1 / c_rg).
//   slots: 0 s_op, 1 q1.x, 2 q1.y, 3 q2.x, 4 q2.y, 5 m2yy, 6 c_rg.x, 7 c_rg.y, 8 c_bd.x, 9 c_bd.y
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int wave_sum10_slot_of_lane(int lane)
{
    const int b0 = lane & 1, b1 = (lane >> 1) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
    if (b1) return b5 ? 5 : 0;
    return (b4 ? 6 : 1) + b0 + 2 * b5;
}

// The two lane-select masks (lanes whose bit 0 / bit 1 is set). Fetch them ONCE outside the loop that reduces: the empty asm
// makes them opaque, so they live in SGPR pairs; as visible constants hipcc notices that both halves of each are equal,
// keeps one half and re-materialises the pair with an s_mov in front of every reduction.
struct WaveSelectMasks { unsigned long long m0, m1; };
__device__ __forceinline__ WaveSelectMasks wave_select_masks()
{
    WaveSelectMasks w{0xAAAAAAAAAAAAAAAAull, 0xCCCCCCCCCCCCCCCCull};
    asm volatile("" : "+s"(w.m0), "+s"(w.m1));
    return w;
}

__device__ __forceinline__ void lane_swap32(float& a, float& b)   // lanes 32-63 of a <-> lanes 0-31 of b
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void lane_swap16(float& a, float& b)   // odd 16-lane rows of a <-> even rows of b
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}

__device__ __forceinline__ void lane_swap32(f2v& a, f2v& b)
{
    float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    lane_swap32(ax, bx); lane_swap32(ay, by);
    a = f2v{ax, ay}; b = f2v{bx, by};
}
__device__ __forceinline__ void lane_swap16(f2v& a, f2v& b)
{
    float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    lane_swap16(ax, bx); lane_swap16(ay, by);
    a = f2v{ax, ay}; b = f2v{bx, by};
}

// `proc`/`jj`, `lds_lane`/`j`: the caller's bookkeeping for this entry (mark bit jj in the processed mask, LDS byte address
// lds_lane + ROW_BYTES j of the total's slot) rides in issue slots the DPP hazards would otherwise fill with s_nop.
template <int ROW_BYTES = 40>
__device__ __forceinline__ float wave_sum10_transposed(const WaveSelectMasks& w, float s_op, f2v q1, f2v q2, float m2yy, f2v c_rg, f2v c_bd,
                                                       unsigned long long& proc, int jj, uint32_t lds_lane, int j, uint32_t& lds_addr)
{
    // xor 32
    lane_swap32(q1, q2);
    f2v P = q1 + q2;
    lane_swap32(c_rg, c_bd);
    f2v C = c_rg + c_bd;
    lane_swap32(s_op, m2yy);
    float S = s_op + m2yy;
    // xor 16
    lane_swap16(P, C);
    const f2v PC = P + C;
    float S2 = S;
    lane_swap16(S, S2);
    S = S + S2;
    // xor 1, xor 2 (selecting), xor 8, xor 4 (plain). %[s] is read by a DPP two instructions into the block, so whatever wrote
    // it is at least two issue slots away.
    float keep, send, u, s1, t;
    uint32_t joff;
    asm volatile(
        "v_cndmask_b32_e64 %[send], %[y], %[x], %[m0]\n\t"      // bit 0 ? x : y
        "v_cndmask_b32_e64 %[keep], %[x], %[y], %[m0]\n\t"      // bit 0 ? y : x
        "v_add_f32_dpp %[s1], %[s], %[s] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[u], %[send], %[keep] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[send], %[s1], %[u], %[m1]\n\t"     // bit 1 ? u : s1
        "v_cndmask_b32_e64 %[keep], %[u], %[s1], %[m1]\n\t"     // bit 1 ? s1 : u
        "s_mul_i32 %[joff], %[j], %[rowb]\n\t"
        "v_add_f32_dpp %[t], %[send], %[keep] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_bitset1_b64 %[proc], %[jj]\n\t"
        "v_add_u32 %[addr], %[joff], %[lane]\n\t"
        "v_add_f32_dpp %[t], %[t], %[t] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[t], %[t], %[t] row_ror:4 row_mask:0xf bank_mask:0xf"
        : [keep] "=&v"(keep), [send] "=&v"(send), [u] "=&v"(u), [s1] "=&v"(s1), [t] "=&v"(t), [joff] "=&s"(joff), [addr] "=&v"(lds_addr),
          [proc] "+s"(proc)
        : [x] "v"(PC.x), [y] "v"(PC.y), [s] "v"(S), [m0] "s"(w.m0), [m1] "s"(w.m1), [j] "s"(j), [jj] "s"(jj), [lane] "v"(lds_lane), [rowb] "n"(ROW_BYTES));
    return t;
}

// Inclusive prefix sum across the wave (6 shuffle steps); used for instance expansion.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Expand per-lane item counts into a lane-parallel loop over all items of the wave, 64 items at a time:
// f(src_lane, k, active) is called by ALL 64 lanes each round (so f may shuffle from src_lane -- a DS
// permute only returns data of lanes that are active at the call); its side effects must be guarded by `active`.
template <typename F>
__device__ __forceinline__ void wave_expand(uint32_t cnt, F&& f)
{
    const int lane = lane_id();
    const uint32_t incl = wave_inclusive_scan(cnt);
    const uint32_t total = __shfl(incl, 63, 64);
    for (uint32_t base = 0; base < total; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        int lo = 0;  // smallest lane with incl > i
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const uint32_t v = __shfl(incl, lo + step - 1, 64);
            if (v <= i) lo += step;
        }
        lo = lo > 63 ? 63 : lo;
        const uint32_t src_incl = __shfl(incl, lo, 64);
        const uint32_t src_cnt = __shfl(cnt, lo, 64);
        f(lo, i - (src_incl - src_cnt), i < total);
    }
}

// Visit the (Gaussian, tile) instances of the wave's 64 Gaussians (lane l: `cnt` tiles, the rectangle [x0, x0 + w) x [y0, ...), row
// major) without wave_expand's search: every 64 instances of its lane-parallel walk over the concatenated lists cost a six-step binary
// search of dependent cross-lane reads (~800 cycles), and a SLAM map's 30-60 tiles per Gaussian made that loop two thirds of
// preprocess_fwd and nine tenths of scatter_instances.
//   wave_visit_small: Gaussians of up to INSTANCES_SMALL tiles -- every lane walks its OWN rectangle, incrementally, no cross-lane
//       traffic: f(own_lane, lane, k, tx, ty), <= INSTANCES_SMALL trips. Consecutive lanes then touch unrelated addresses, so this is
//       for callbacks that only hit LDS (the tile histogram); scatter_instances, which writes inst_gauss[u], keeps wave_expand for them.
//   wave_visit_large: the larger ones, one at a time -- the source's fields are wave-uniform (v_readlane), the 64 lanes take 64
//       consecutive tiles per trip: f(uniform_source, src, k, tx, ty).
// f runs only for real instances (no `active` guard needed) and must not contain wave-level operations.
constexpr uint32_t INSTANCES_SMALL = 32;
struct own_lane { static constexpr bool uniform = false; };
struct uniform_source { static constexpr bool uniform = true; };
template <typename TAG, typename T>
__device__ __forceinline__ T of_source(TAG, T v, int src)
{
    if constexpr (TAG::uniform) {
        static_assert(sizeof(T) == 4, "32-bit fields only");
        return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
    } else {
        return v;
    }
}
template <typename F>
__device__ __forceinline__ void wave_visit_small(uint32_t cnt, int x0, int y0, int w, F&& f)
{
    const uint32_t n = cnt <= INSTANCES_SMALL ? cnt : 0u;
    int tx = x0, ty = y0;
    for (uint32_t k = 0; __builtin_amdgcn_ballot_w64(k < n) != 0ull; k++) {
        if (k < n) f(own_lane{}, lane_id(), k, tx, ty);
        if (++tx == x0 + w) { tx = x0; ty++; }
    }
}
template <typename F>
__device__ __forceinline__ void wave_visit_large(uint32_t cnt, int x0, int y0, int w, F&& f)
{
    const int lane = lane_id();
    unsigned long long large = __builtin_amdgcn_ballot_w64(cnt > INSTANCES_SMALL);
    while (large) {
        const int src = (int)__builtin_ctzll(large);
        large &= large - 1;
        const uint32_t c = of_source(uniform_source{}, cnt, src);
        const int sx0 = of_source(uniform_source{}, x0, src), sy0 = of_source(uniform_source{}, y0, src), sw = max(1, of_source(uniform_source{}, w, src));
        const float inv = 1.0f / (float)sw;
        for (uint32_t k = (uint32_t)lane; k < c; k += 64) {
            int q = (int)(((float)k + 0.5f) * inv);            // k / sw for k < 2^22, one step of correction either way
            q -= (uint32_t)(q * sw) > k ? 1 : 0;
            q += (uint32_t)((q + 1) * sw) <= k ? 1 : 0;
            f(uniform_source{}, src, k, sx0 + (int)(k - (uint32_t)(q * sw)), sy0 + q);
        }
    }
}

// ---- small geometry helpers (reference semantics cited at the call sites) --------------------------------------
// DGR/cuda_rasterizer/auxiliary.h:41-44 evaluates in double because of its 1.0 / 0.5 literals; so do we.
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:46-56: tile rectangle with int truncation, clamped to the grid.
__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1)
{
    x0 = min(gx, max(0, (int)((px - radius) / TILE_X)));
    y0 = min(gy, max(0, (int)((py - radius) / TILE_Y)));
    x1 = min(gx, max(0, (int)((px + radius + TILE_X - 1) / TILE_X)));
    y1 = min(gy, max(0, (int)((py + radius + TILE_Y - 1) / TILE_Y)));
}

// ---- the binning scratch buffer (the reference's BinningState, rasterizer_impl.h:55-66) --------------------------------
// chunk_info | inst_gauss[carve_R] | partials[carve_R] x 40 B (forward: aliased by the sort keys) | sorted[cap] | ckpt, each 256-byte
// aligned. ckpt = per-pixel compositing state (T, r, g, b, depth) at every CHUNK-th entry of every tile list, written by the
// forward tile kernel: with it the backward pass can start anywhere in a list (see render_bwd_kernel).
// carve_R is the instance count the buffer was LAID OUT for: the exact R when the host waited for it before allocating, or
// the speculative capacity when the forward pass was enqueued without waiting. It lives in the geometry header (word 4),
// so the backward kernels derive their pointers on the device and the host never has to know which of the two it was.
constexpr int CHUNK = 128;              // entries of a tile list per backward work item
// One gradient slot per (tile, Gaussian) instance, written by render_bwd and summed per Gaussian by geometry_bwd: ten floats, 40 bytes (round 6;
// 48 with two dead floats before: 10 MB less to write and read again per frame at config #2)
//   {dmean2D.x, dmean2D.y, dconic.x, dconic.y | dconic.w, dopacity, r, g | b, depth}
constexpr int SLOT_FLOATS = 10;
typedef float4 __attribute__((aligned(8))) float4_a8;       // a slot's 16-byte pieces sit at 8-byte aligned addresses (global memory: any dword alignment is legal)
__device__ __forceinline__ void slot_store(float* partials, size_t instance, float4 a, float4 b, float2 c)
{
    float* const s = partials + instance * SLOT_FLOATS;
    *reinterpret_cast<float4_a8*>(s) = a; *reinterpret_cast<float4_a8*>(s + 4) = b; *reinterpret_cast<float2*>(s + 8) = c;
}
__device__ __forceinline__ void slot_load(const float* partials, size_t instance, float4& a, float4& b, float2& c)
{
    const float* const s = partials + instance * SLOT_FLOATS;
    a = *reinterpret_cast<const float4_a8*>(s); b = *reinterpret_cast<const float4_a8*>(s + 4); c = *reinterpret_cast<const float2*>(s + 8);
}
constexpr int CKPT_FLOATS = 5 * TILE_X * TILE_Y;   // one checkpoint: 5 planes of 256 pixels
// Work items of render_bwd_kernel: one 16-byte record per CHUNK-entry piece of a tile list, {tile, position of the piece in sorted[],
// entries in it | bit 16 = more entries follow in the list, list position of its first entry}, written by the forward tile kernel at
// the index of the backward BLOCK that will take the piece (the XCD banding is applied by the writer). A frame has at most
// R / CHUNK + tiles pieces.
// Which block takes which piece (round 6): render_bwd runs ~4.3 generations of blocks per CU slot; with the pieces in tile order it ended in a
// ~20 us tail in which the CUs ran dry one by one behind whichever pieces happened to start last (tools/tile_timeline.py,
// profiles/r06_tile_timeline.json: mean residency 3.9 of 5 blocks per CU, the last block started at 76 of 96 us). Now every XCD's sequence
// (block b runs on XCD b % 8, in the order of b / 8) is: first the FULL pieces (CHUNK entries) of a contiguous range of tiles, in tile order
// -- neighbouring tiles share Gaussians and one tile's pieces share its pixel state: same L2 --, then the PARTIAL last pieces of the lists of
// ITS band of tiles, in descending length, so that the launch ends on short blocks everywhere at once (96 -> 87 us first block to last,
// residency 4.4) and a tile's partial piece runs where the tile's pixel state already is (dealt round robin over all XCDs instead -- the map
// below, kept as the fallback for frames in which a band holds more partial pieces than its XCD runs blocks -- they cost 15 MB of fabric
// traffic per frame). item_block() is that map; its inputs (per tile: full pieces in front of it, rank of its partial
// piece; the frame's number of full pieces) come from order_tiles_body, one extra block of the scatter launch.
__host__ __device__ inline uint32_t items_below(uint32_t M, uint32_t x) { return x * (M >> 3) + (x < (M & 7u) ? x : (M & 7u)); }   // #{b < M : b % 8 < x}
// first full-piece rank of XCD x when the frame has N pieces, Pn of them partial: XCD x runs #{b < N : b % 8 == x} blocks, #{r < Pn : r % 8 == x} of them partial
__host__ __device__ inline uint32_t full_start(uint32_t N, uint32_t Pn, uint32_t x) { return items_below(N, x) - items_below(Pn, x); }
__host__ __device__ inline uint32_t item_block_full(uint32_t N, uint32_t Pn, uint32_t f /* rank among the full pieces, tile order */)
{
    uint32_t x = 0;
    while (x < 7u && full_start(N, Pn, x + 1u) <= f) x++;
    return 8u * (f - full_start(N, Pn, x)) + x;
}
__host__ __device__ inline uint32_t item_block_partial(uint32_t N, uint32_t Pn, uint32_t r /* rank among the partial pieces, longest first */)
{
    const uint32_t x = r & 7u;
    return 8u * ((full_start(N, Pn, x + 1u) - full_start(N, Pn, x)) + (r >> 3)) + x;
}
// where order_tiles_body leaves its results: the padding words of the per-tile counters (CTR_STRIDE words per tile, word 0 in use)
constexpr int POS_FULL_BASE = 1, POS_PART_RANK = 2, POS_TOTAL_FULL = 1;   // tile_count[t * CTR_STRIDE + 1 / + 2]; tile_count[T * CTR_STRIDE + 1]
// tile_count[T * CTR_STRIDE + 2]: 1 = a tile's partial piece runs on the XCD of the tile's band (POS_PART_RANK is its rank inside that band,
// tile_count[T * CTR_STRIDE + 32 + x], x = 0 .. 8: first full-piece rank of XCD x), 0 = the round-robin map of item_block_*
constexpr int POS_PART_HOME = 2, POS_FULL_START = 32;
constexpr int POS_FWD_TILE = 3;     // tile_count[i * CTR_STRIDE + 3]: the tile that render_fwd's block takes in place of tile i (order_fwd_tiles_body)
struct BinningPtrs { uint4* chunk_info; uint32_t* inst_gauss; float4* partials; uint64_t* keys; uint2* sorted; float* ckpt; char* end; };
__host__ __device__ inline BinningPtrs carve_binning(char* base, size_t carve_R, size_t cap_sorted, size_t ntiles)
{
    BinningPtrs b;
    uintptr_t p = (reinterpret_cast<uintptr_t>(base) + 255) & ~uintptr_t(255);
    b.chunk_info = reinterpret_cast<uint4*>(p);      // FIRST, at an offset that depends on nothing: render_bwd reads its entry without the header
    p = (p + (cap_sorted / CHUNK + ntiles + 2) * sizeof(uint4) + 255) & ~uintptr_t(255);
    b.inst_gauss = reinterpret_cast<uint32_t*>(p);
    p = (p + carve_R * sizeof(uint32_t) + 255) & ~uintptr_t(255);
    b.partials = reinterpret_cast<float4*>(p);
    b.keys = reinterpret_cast<uint64_t*>(p);
    p = (p + carve_R * SLOT_FLOATS * sizeof(float) + 255) & ~uintptr_t(255);
    b.sorted = reinterpret_cast<uint2*>(p);
    p = (p + cap_sorted * sizeof(uint2) + 255) & ~uintptr_t(255);
    b.ckpt = reinterpret_cast<float*>(p);            // checkpoint id = sorted position / CHUNK, ids 1 .. cap/CHUNK + 1
    b.end = reinterpret_cast<char*>(p + (cap_sorted / CHUNK + 2) * (size_t)CKPT_FLOATS * sizeof(float));
    return b;
}
// geometry header words (uint32): R, flags, R_alloc, longest tile list, carve_R, capacity of sorted[], number of list chunks
enum { HDR_R = 0, HDR_FLAGS = 1, HDR_R_ALLOC = 2, HDR_MAX_TILE = 3, HDR_CARVE_R = 4, HDR_CAP_SORTED = 5, HDR_CHUNKS = 6, HDR_WORDS = 8 };
enum { FLAG_PREFILTERED = 1u, FLAG_OVERFLOW = 2u };   // FLAG_OVERFLOW: the speculative binning capacity did not suffice

// ---- how the kernels read a Gaussian: the reference's activated tensors, or the model's raw parameters ------------------
// Raw mode (xyz != nullptr) folds the prologue of the reference's render() into the kernels' loads
// (gaussian_splatting/gaussian_renderer/__init__.py:108-127,159-174; scene/gaussian_model.py:60-68,100-128):
//   means3D = _xyz (+ dx[slot]);  scales = exp(_scaling) (+ ds[slot], isotropic models repeat the one value);
//   rotations = _rotation / max(|_rotation|, 1e-12) (+ dr[slot]);  opacity = sigmoid(_opacity);  shs = cat(_features_dc, _features_rest)
// where slot = dyn_slot[i] >= 0 marks the dynamic subset (pc.dygs) the control-node deltas apply to. The backward kernel
// applies the matching chain rules on its stores (geometry_bwd_kernel).
// What preprocess_fwd leaves per visible Gaussian for the binning and tile kernels (scatter_instances, render_fwd, render_bwd): ONE
// 48-byte row -- three 16-byte loads that touch at most two 128-byte lines -- instead of four arrays (means2D, conic_opacity, rgb,
// depths) = four cache lines per gathered list entry. The gathers were the largest part of both tile kernels' fabric traffic.
struct __attribute__((aligned(16))) TileRec {
    float4 q0;   // mean2D.x, mean2D.y, view-space depth, opacity
    float4 q1;   // conic a, b, c (inverse 2D covariance), unused
    float4 q2;   // r, g, b (or colors_precomp / the flow colour), unused
};
static_assert(sizeof(TileRec) == 48, "TileRec is three float4");

struct RawInputs {
    const float* xyz; const float* log_scales; int scale_dim; const float* raw_rot; const float* logit_opacity;
    const float* f_dc; const float* f_rest; const int* dyn_slot; const float* dx; const float* ds; const float* dr;
    const int* gather;   // optional: rasterized Gaussian i reads row gather[i] of the raw tensors (render()'s boolean mask, :179-191)
    const float* flow_dx2; const float* flow_proj1; const float* flow_proj2;   // flow mode (render_flow, :229-361): see include/gs_rasterizer.h
    const int* flow_clip;        // flow mode, optional: tile rectangle [x0, y0, x1, y1) outside of which nothing is needed (gsr_view.flow_clip)
    // delta_mode 1 (kernels instantiated with PRE = true): dx / ds / dr are the outputs of the 4DGaussians deformation network
    // (gaussian_renderer/__init__.py:149-157, utils/deformation.py:113-149), added to the RAW parameters IN FRONT of the activations:
    // scales = exp(_scaling + ds), rotations = normalize(_rotation + dr) -- not the control-node deltas of :159-174, which are added behind
    // them (mode 0). Then dyn_slot may be null (Gaussian i takes row i) and delta_stride is the number of floats between consecutive rows of
    // dx, ds AND dr (0: compact 3 / 3 / 4), so that the three can be column ranges of one [K, 10] network output.
    int delta_mode; int delta_stride;
};
struct RawGrads { float* f_dc; float* f_rest; float* ddx; float* dds; float* ddr; int scale_dim; float* ddx2; };

struct ShView {    // SH coefficients of one Gaussian: [k] with k = 3 * coefficient + channel
    const float* dc; const float* rest;
    __device__ __forceinline__ float operator[](int k) const { return k < 3 ? dc[k] : rest[k - 3]; }
};
// a + b as its own instruction: `a + (x * y)` would otherwise be contracted into one fma (HIP's __fadd_rn is a plain `+` and does not
// prevent that); the accumulate mode of the backward pass must round the gradient first, as autograd's own accumulation does.
__device__ __forceinline__ float add_separately(float a, float b)
{
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct ShOut {
    float* dc; float* rest;
    bool acc;        // accumulate mode (GSR_BACKWARD_ACCUMULATE): assignments add to what the caller's gradient buffer already holds
    float old_dc[3]; // ... whose DC part the kernel loaded up front (zero when the mode is off); higher bands are read here
    bool skip;       // pose-only backward (GSR_BACKWARD_POSE_ONLY): the coefficient gradients are not stored at all
    struct Ref {
        float* p; bool acc; float old; bool have_old; bool skip;
        __device__ __forceinline__ void operator=(float v) const
        {
            if (!skip) *p = have_old ? add_separately(old, v) : (acc ? add_separately(*p, v) : v);
        }
    };
    __device__ __forceinline__ Ref operator[](int k) const
    {
        return k < 3 ? Ref{dc + k, acc, old_dc[k], true, skip} : Ref{rest + (k - 3), acc, 0.f, false, skip};   // k: a compile-time constant at every call site
    }
};

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// ndc(p; M).xy of render_flow (gaussian_renderer/__init__.py:268-282): row-vector convention, w + 1e-7; optionally the 2x3 Jacobian
// d ndc / d p applied to a cotangent (gu, gv): returns J^T (gu, gv).
__device__ __forceinline__ void flow_ndc(const float* __restrict__ M, f3 p, float& u, float& v)
{
    const float hx = p.x * M[0] + p.y * M[4] + p.z * M[8] + M[12], hy = p.x * M[1] + p.y * M[5] + p.z * M[9] + M[13];
    const float hw = p.x * M[3] + p.y * M[7] + p.z * M[11] + M[15];
    const float iw = 1.0f / (hw + 0.0000001f);
    u = hx * iw; v = hy * iw;
}
__device__ __forceinline__ f3 flow_ndc_vjp(const float* __restrict__ M, f3 p, float gu, float gv)
{
    float u, v;
    flow_ndc(M, p, u, v);
    const float hw = p.x * M[3] + p.y * M[7] + p.z * M[11] + M[15];
    const float iw = 1.0f / (hw + 0.0000001f);
    // d u / d p_i = (M[4 i + 0] - u M[4 i + 3]) iw, same for v with column 1
    return mk3(((M[0] - u * M[3]) * gu + (M[1] - v * M[3]) * gv) * iw, ((M[4] - u * M[7]) * gu + (M[5] - v * M[7]) * gv) * iw,
               ((M[8] - u * M[11]) * gu + (M[9] - v * M[11]) * gv) * iw);
}

__device__ __forceinline__ size_t raw_row(const RawInputs& r, size_t i) { return r.gather ? (size_t)r.gather[i] : i; }   // row of the raw tensors
__device__ __forceinline__ int raw_slot(const RawInputs& r, size_t row) { return r.dyn_slot ? r.dyn_slot[row] : -1; }
// PRE (delta_mode 1): row of the network's output that belongs to raw row `row`, and the distance between rows of a delta tensor
__device__ __forceinline__ int pre_slot(const RawInputs& r, size_t row) { return r.dyn_slot ? r.dyn_slot[row] : (int)row; }
__device__ __forceinline__ int pre_stride(const RawInputs& r, int width) { return r.delta_stride ? r.delta_stride : width; }
template <bool PRE = false>
__device__ __forceinline__ f3 load_mean(const float* means3D, const RawInputs& r, size_t i)
{
    if (!r.xyz) return mk3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    i = raw_row(r, i);
    f3 m = mk3(r.xyz[3 * i], r.xyz[3 * i + 1], r.xyz[3 * i + 2]);
    if constexpr (PRE) {
        const int sl = pre_slot(r, i);
        if (sl >= 0 && r.dx) { const float* d = r.dx + (size_t)pre_stride(r, 3) * sl; m.x += d[0]; m.y += d[1]; m.z += d[2]; }
        return m;
    }
    const int sl = raw_slot(r, i);
    if (sl >= 0 && r.dx) { m.x += r.dx[3 * sl]; m.y += r.dx[3 * sl + 1]; m.z += r.dx[3 * sl + 2]; }
    return m;
}
template <bool PRE = false>
__device__ __forceinline__ void load_scale(const float* scales, const RawInputs& r, size_t i, float s[3])
{
    if (!r.xyz) { s[0] = scales[3 * i]; s[1] = scales[3 * i + 1]; s[2] = scales[3 * i + 2]; return; }
    i = raw_row(r, i);
    if constexpr (PRE) {                         // exp(_scaling.repeat(1, 3) + ds) (gaussian_renderer/__init__.py:150-155)
        const int sl = pre_slot(r, i);
        const float* d = sl >= 0 && r.ds ? r.ds + (size_t)pre_stride(r, 3) * sl : nullptr;
#pragma unroll
        for (int k = 0; k < 3; k++) s[k] = expf(r.log_scales[r.scale_dim == 1 ? i : 3 * i + k] + (d ? d[k] : 0.f));
        return;
    }
    if (r.scale_dim == 1) { s[0] = s[1] = s[2] = expf(r.log_scales[i]); }
    else { s[0] = expf(r.log_scales[3 * i]); s[1] = expf(r.log_scales[3 * i + 1]); s[2] = expf(r.log_scales[3 * i + 2]); }
    const int sl = raw_slot(r, i);
    if (sl >= 0 && r.ds) { s[0] += r.ds[3 * sl]; s[1] += r.ds[3 * sl + 1]; s[2] += r.ds[3 * sl + 2]; }
}
template <bool PRE = false>
__device__ __forceinline__ void load_rot(const float* rotations, const RawInputs& r, size_t i, float q[4])
{
    if (!r.xyz) { q[0] = rotations[4 * i]; q[1] = rotations[4 * i + 1]; q[2] = rotations[4 * i + 2]; q[3] = rotations[4 * i + 3]; return; }
    i = raw_row(r, i);
    float a = r.raw_rot[4 * i], b = r.raw_rot[4 * i + 1], c = r.raw_rot[4 * i + 2], d = r.raw_rot[4 * i + 3];
    if constexpr (PRE) {                         // normalize(_rotation + dr) (:156)
        const int sl = pre_slot(r, i);
        if (sl >= 0 && r.dr) { const float* e = r.dr + (size_t)pre_stride(r, 4) * sl; a += e[0]; b += e[1]; c += e[2]; d += e[3]; }
    }
    const float inv = 1.0f / fmaxf(sqrtf(a * a + b * b + c * c + d * d), 1e-12f);     // torch.nn.functional.normalize
    q[0] = a * inv; q[1] = b * inv; q[2] = c * inv; q[3] = d * inv;
    if constexpr (!PRE) {
        const int sl = raw_slot(r, i);
        if (sl >= 0 && r.dr) { q[0] += r.dr[4 * sl]; q[1] += r.dr[4 * sl + 1]; q[2] += r.dr[4 * sl + 2]; q[3] += r.dr[4 * sl + 3]; }
    }
}
__device__ __forceinline__ float load_opacity(const float* opacities, const RawInputs& r, size_t i)
{
    return r.xyz ? 1.0f / (1.0f + expf(-r.logit_opacity[raw_row(r, i)])) : opacities[i];           // torch.sigmoid
}
__device__ __forceinline__ ShView sh_view(const float* shs, const RawInputs& r, size_t i, int M)
{
    if (!r.xyz) { const float* p = shs + i * M * 3; return ShView{p, p + 3}; }
    i = raw_row(r, i);
    return ShView{r.f_dc + 3 * i, r.f_rest ? r.f_rest + i * (size_t)(M - 1) * 3 : nullptr};
}

// Matrices arrive as the reference passes them: row-major memory of the TRANSPOSED maths matrix,
// i.e. m[0],m[4],m[8],m[12] is row 0 (auxiliary.h:58-77).
__device__ __forceinline__ f3 xform_point_4x3(f3 p, const float* __restrict__ m)
{
    return mk3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
               m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}

// j = index of the lowest set bit of m; clears it.  Two SALU instructions (the C idiom m &= m - 1 costs three plus the ff1).
__device__ __forceinline__ int pop_lowest_bit(unsigned long long& m)
{
    int j;
    asm("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(j), "+s"(m));
    return j;
}

// ---- SH coefficient rows through LDS (round 6) -----------------------------------------------------------------------------------------
// A Gaussian's coefficients are one contiguous row ([M, 3] floats); read or written per lane -- lane = Gaussian, one coefficient per
// instruction -- every load / store touches 64 different rows: 48 + 48 instructions of 64 memory transactions each at SH degree 3, which
// made geometry_bwd the one bandwidth-bound kernel of the step and ran it at ~1 TB/s (96 us against 24 at M = 1, profiles/r05_long_lists.json;
// preprocess_fwd: 35 against 15).
// Here the wave moves ROWS: stage_rows() copies a column range of the rows of the lanes in `rows` into a wave-private LDS window, one
// LDS-DMA instruction per row (global_load_lds: lane c fetches column c, the row lands contiguously, nothing passes through registers and
// all rows are in flight at once); flush_rows() writes a window back, a row per store instruction. The window's row stride is odd, so lane =
// row accesses are bank-conflict free. Same values, same arithmetic as the per-lane path -- only the data movement differs.
constexpr int SH_WIN_STRIDE = 25;                 // floats per window row: up to 24 coefficients of a Gaussian (degrees 1 + 2, or degree 3's 21)
constexpr int SH_WIN_FLOATS = 64 * SH_WIN_STRIDE;
__device__ __forceinline__ const float* lane_pointer(const float* p, int src)      // lane src's pointer, wave-uniform
{
    const uintptr_t v = reinterpret_cast<uintptr_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return reinterpret_cast<const float*>(((uintptr_t)hi << 32) | lo);
}
// stage_rows = stage_rows_issue (the DMA instructions; returns at once) + stage_rows_wait (everything landed, visible to the wave)
__device__ __forceinline__ void stage_rows_issue(float* win, unsigned long long rows, const float* row_ptr, int col0, int width)
{
    const int lane = lane_id();
    typedef __attribute__((address_space(3))) float lds_float;
    // the DMA's LDS address travels in M0: a scalar. The window pointer is wave-uniform but derived from the thread index, so it is made
    // provably uniform here (readfirstlane of its 32-bit LDS offset)
    const uint32_t win_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_float*)win);
    if (lane < width) {                            // (the lane mask is set once around the loop: inside it, it cost the loop five more instructions per row)
        while (rows) {
            const int g = pop_lowest_bit(rows);
            const float* src = lane_pointer(row_ptr, g) + col0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane),
                                             (__attribute__((address_space(3))) void*)(lds_float*)(uintptr_t)(win_off + (uint32_t)(g * SH_WIN_STRIDE * 4)), 4, 0, 0);
        }
    }
}
__device__ __forceinline__ void stage_rows_wait()
{
    __builtin_amdgcn_s_waitcnt(0);                 // the DMA writes are ordered by vmcnt only
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void stage_rows(float* win, unsigned long long rows, const float* row_ptr, int col0, int width)
{
    stage_rows_issue(win, rows, row_ptr, col0, width);
    stage_rows_wait();
}
// acc: the stores ADD to what the destination holds (GSR_BACKWARD_ACCUMULATE); four rows per trip, so that their LDS reads (and old values)
// are in flight together
__device__ __forceinline__ void flush_rows(const float* win, unsigned long long rows, float* row_ptr, int col0, int width, bool acc)
{
    const int lane = lane_id();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the lanes' own row writes (the window is wave-private: DS operations of one wave execute in order)
    __builtin_amdgcn_wave_barrier();
    if (lane < width) {
        while (rows) {
            float* dst[4]; float v[4], old[4]; bool on[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                on[u] = rows != 0ull;
                const int g = on[u] ? pop_lowest_bit(rows) : 0;
                dst[u] = const_cast<float*>(lane_pointer(row_ptr, g)) + col0;
                v[u] = win[g * SH_WIN_STRIDE + lane];
                old[u] = (acc && on[u]) ? dst[u][lane] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (on[u]) dst[u][lane] = acc ? add_separately(old[u], v[u]) : v[u];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// XCD-aware tile order: block b runs on XCD b % 8 (observed, speed only); give each XCD a contiguous band of tiles
// so neighbouring tiles (which share Gaussians) hit the same L2. Bijective for any tile count.
__device__ __forceinline__ int xcd_tile_of_block(int b, int ntiles)
{
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = b & 7, k = b >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

// ... and its inverse: the block that xcd_tile_of_block() sends to tile / work item `id`
__device__ __forceinline__ int xcd_block_of_tile(int id, int ntiles)
{
    const int q = ntiles >> 3, r = ntiles & 7;
    const int head = r * (q + 1);
    if (id < head) return (id % (q + 1)) * 8 + id / (q + 1);
    const int rest = id - head;
    return (rest % q) * 8 + r + rest / q;
}

// ---- the weighted L1 loss's cotangents of ONE pixel (gs_loss.h: l1_loss_bwd_kernel; gs_render.h: the tracking epilogue of render_fwd) ----
// wr, wd: the pixel's weights with the loss coefficients and the upstream gradient already multiplied in (and, for the tracking loss,
// the rendered opacity / the opacity test: utils/slam_utils.py:118-135). One function for both callers, so that a pixel's cotangents are the
// same bits whichever kernel forms them.
__device__ __forceinline__ float l1_sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }   // d|v|/dv as torch defines it
struct L1PixelGrad { float gi[3]; float gd; };
__device__ __forceinline__ L1PixelGrad l1_bwd_pixel(float wr, float wd, float ea, float eb, const float (&I)[3], const float (&gt)[3], float depth, float gt_depth,
                                                    float& da, float& db /* running exposure-gradient sums of the calling thread */)
{
    L1PixelGrad o;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float s = wr * l1_sgn(ea * I[c] + eb - gt[c]);     // dL / d(exp(a) I + b)
        o.gi[c] = s * ea;
        da += s * ea * I[c];
        db += s;
    }
    o.gd = wd * l1_sgn(depth - gt_depth);
    return o;
}

// The tracking loss fused into render_fwd's epilogue (include/slam_map.h: gsr_track_step): gt_image == nullptr switches it off.
struct TrackLossArgs {
    const float* gt_image; const float* gt_depth;      // [3, N], [N]
    const float* w_rgb; const float* w_depth;          // [N] each or nullptr (= 1)
    const float* exposure_a; const float* exposure_b;  // device scalars or nullptr
    float opacity_thr; int use_opacity;                // tracking: w_rgb *= rendered opacity, w_depth *= (opacity > thr)
    float c_rgb, c_depth;                              // alpha / (3 N), (1 - alpha) / N
    float* dL_dimage; float* dL_ddepth;                // out: [3, N], [N]
    float* partials;                                   // out: [tiles][2] exposure-gradient partial sums (d/da, d/db)
};

}  // namespace gsr
