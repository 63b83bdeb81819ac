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

// ---- wave reduction of ten values with DPP adds ------------------------------------------------------------------
// One v_add_f32 with a DPP source modifier per value and butterfly stage (6 stages), written as inline asm: through
// the update_dpp builtin hipcc emits v_mov_b32 (old = 0) + v_mov_b32_dpp + half a v_pk_add_f32 per step (its SLP
// vectoriser packs the adds, and VOP3P cannot take a DPP operand), 2.5x the instructions.
// The ten chains are interleaved stage by stage, so every DPP read is ten instructions behind the write of its
// source (the gfx9 "VALU write -> DPP read" hazard needs two wait states; hipcc does not pad inside asm, and the
// leading s_nop covers values produced right before the statement). Totals are valid in lanes 48..63 (row 3).
#define GSR_DPP_STAGE(ctrl)                                   \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\t"                   \
    "v_add_f32_dpp %1, %1, %1 " ctrl "\n\t"                   \
    "v_add_f32_dpp %2, %2, %2 " ctrl "\n\t"                   \
    "v_add_f32_dpp %3, %3, %3 " ctrl "\n\t"                   \
    "v_add_f32_dpp %4, %4, %4 " ctrl "\n\t"                   \
    "v_add_f32_dpp %5, %5, %5 " ctrl "\n\t"                   \
    "v_add_f32_dpp %6, %6, %6 " ctrl "\n\t"                   \
    "v_add_f32_dpp %7, %7, %7 " ctrl "\n\t"                   \
    "v_add_f32_dpp %8, %8, %8 " ctrl "\n\t"                   \
    "v_add_f32_dpp %9, %9, %9 " ctrl "\n\t"
__device__ __forceinline__ void wave_sum10_to_row3(float& v0, float& v1, float& v2, float& v3, float& v4, float& v5, float& v6,
                                                   float& v7, float& v8, float& v9)
{
    asm volatile("s_nop 1\n\t"
                 GSR_DPP_STAGE("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 GSR_DPP_STAGE("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 GSR_DPP_STAGE("row_half_mirror row_mask:0xf bank_mask:0xf")
                 GSR_DPP_STAGE("row_mirror row_mask:0xf bank_mask:0xf")
                 GSR_DPP_STAGE("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 GSR_DPP_STAGE("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9));
}
#undef GSR_DPP_STAGE

// ---- transposed ("butterfly") wave reduction of ten values -------------------------------------------------------
// Sums v0..v9 over the 64 lanes in 37 VALU instructions instead of 60: at every butterfly stage a lane keeps only
// half of its values (adding the partner lane's copy of the same half) instead of carrying all ten through all six
// stages. On return every lane l holds the wave-wide total of ONE value, selected by p = l & 15:
//     p:      0   1   2   3   4   5   6   7   8,10,12,14   9,11,13,15
//     value:  v0  v5  v3  v8  v1  v6  v4  v9      v2            v7
// (wave_sum10_slot_of_lane() returns that index), so ten lanes can store the ten totals with a single instruction.
// Stage partners: xor 1 / xor 2 by DPP quad_perm, xor 4 by row_shl:4 / row_shr:4 under bank masks, xor 8 by
// row_ror:8, xor 16 / xor 32 by gfx950's v_permlane16_swap / v_permlane32_swap. Inline asm because hipcc neither
// folds DPP into the adds (see above) nor knows the swap instructions; instructions are ordered so that every DPP /
// permlane source was written at least two issue slots earlier (gfx9 VALU-write -> DPP-read hazard), with s_nop
// where the stage is too short.
__device__ __forceinline__ int wave_sum10_slot_of_lane(int lane) { return (int)((0x7272727294618350ull >> (4 * (lane & 15))) & 15ull); }

// The four lane-select masks (lanes whose bit 0 / 1 / 2 / 3 is set). Fetch them ONCE outside the loop that reduces: the
// empty asm makes them opaque, so they live in four SGPR pairs; as visible constants hipcc notices that both halves of
// each are equal, keeps one half and re-materialises the pair with an s_mov in front of every reduction.
struct WaveSelectMasks { unsigned long long m0, m1, m2, m3; };
__device__ __forceinline__ WaveSelectMasks wave_select_masks()
{
    WaveSelectMasks w{0xAAAAAAAAAAAAAAAAull, 0xCCCCCCCCCCCCCCCCull, 0xF0F0F0F0F0F0F0F0ull, 0xFF00FF00FF00FF00ull};
    asm volatile("" : "+s"(w.m0), "+s"(w.m1), "+s"(w.m2), "+s"(w.m3));
    return w;
}

__device__ __forceinline__ float wave_sum10_transposed(const WaveSelectMasks& w, float v0, float v1, float v2, float v3, float v4,
                                                       float v5, float v6, float v7, float v8, float v9)
{
    const unsigned long long m0 = w.m0, m1 = w.m1, m2 = w.m2, m3 = w.m3;
    float k0, k1, k2, k3, k4, s0, s1, s2, s3, s4;
    asm volatile(
        // stage 1 (xor 1): pairs (v0,v5) (v1,v6) (v2,v7) (v3,v8) (v4,v9): keep = bit0 ? second : first, send the other
        "v_cndmask_b32_e64 %[k0], %[v0], %[v5], %[m0]\n\t"
        "v_cndmask_b32_e64 %[s0], %[v5], %[v0], %[m0]\n\t"
        "v_cndmask_b32_e64 %[k1], %[v1], %[v6], %[m0]\n\t"
        "v_cndmask_b32_e64 %[s1], %[v6], %[v1], %[m0]\n\t"
        "v_cndmask_b32_e64 %[k2], %[v2], %[v7], %[m0]\n\t"
        "v_cndmask_b32_e64 %[s2], %[v7], %[v2], %[m0]\n\t"
        "v_cndmask_b32_e64 %[k3], %[v3], %[v8], %[m0]\n\t"
        "v_cndmask_b32_e64 %[s3], %[v8], %[v3], %[m0]\n\t"
        "v_cndmask_b32_e64 %[k4], %[v4], %[v9], %[m0]\n\t"
        "v_cndmask_b32_e64 %[s4], %[v9], %[v4], %[m0]\n\t"
        "v_add_f32_dpp %[k0], %[s0], %[k0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[k1], %[s1], %[k1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[k2], %[s2], %[k2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[k3], %[s3], %[k3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[k4], %[s4], %[k4] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        // stage 2 (xor 2): pairs (k0,k3) (k1,k4), k2 alone.  results: s0, s1, k2
        "v_cndmask_b32_e64 %[s0], %[k0], %[k3], %[m1]\n\t"      // keep A
        "v_cndmask_b32_e64 %[s2], %[k3], %[k0], %[m1]\n\t"      // send A
        "v_cndmask_b32_e64 %[s1], %[k1], %[k4], %[m1]\n\t"      // keep B
        "v_cndmask_b32_e64 %[s3], %[k4], %[k1], %[m1]\n\t"      // send B
        "v_add_f32_dpp %[k2], %[k2], %[k2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s0], %[s2], %[s0] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s1], %[s3], %[s1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        // stage 3 (xor 4): pair (s0,s1), k2 alone.  results: k0 (pair), k1 (single)
        "v_cndmask_b32_e64 %[k3], %[s0], %[s1], %[m2]\n\t"      // keep
        "v_cndmask_b32_e64 %[k4], %[s1], %[s0], %[m2]\n\t"      // send
        "v_add_f32_dpp %[k1], %[k2], %[k2] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[k1], %[k2], %[k2] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[k0], %[k4], %[k3] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[k0], %[k4], %[k3] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        // stage 4 (xor 8): pair (k0,k1).  result: s0
        "v_cndmask_b32_e64 %[s0], %[k0], %[k1], %[m3]\n\t"      // keep
        "v_cndmask_b32_e64 %[s1], %[k1], %[k0], %[m3]\n\t"      // send
        "s_nop 1\n\t"
        "v_add_f32_dpp %[s0], %[s1], %[s0] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        // stages 5, 6 (xor 16, xor 32): every lane of a column ends with the column total
        "v_mov_b32 %[s1], %[s0]\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %[s0], %[s1]\n\t"
        "s_nop 1\n\t"
        "v_add_f32 %[s0], %[s0], %[s1]\n\t"
        "v_mov_b32 %[s1], %[s0]\n\t"
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %[s0], %[s1]\n\t"
        "s_nop 1\n\t"
        "v_add_f32 %[s0], %[s0], %[s1]\n\t"
        "s_nop 0"
        : [k0] "=&v"(k0), [k1] "=&v"(k1), [k2] "=&v"(k2), [k3] "=&v"(k3), [k4] "=&v"(k4), [s0] "=&v"(s0), [s1] "=&v"(s1),
          [s2] "=&v"(s2), [s3] "=&v"(s3), [s4] "=&v"(s4)
        : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [v4] "v"(v4), [v5] "v"(v5), [v6] "v"(v6), [v7] "v"(v7),
          [v8] "v"(v8), [v9] "v"(v9), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3));
    return s0;
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
// inst_gauss[carve_R] | partials[carve_R] x 48 B (forward: aliased by the sort keys) | sorted[cap] | ckpt, each 256-byte
// aligned. ckpt = per-pixel compositing state (T, r, g, b, depth) at every CHUNK-th entry of every tile list, written by the
// forward tile kernel: with it the backward pass can start anywhere in a list (see render_bwd_kernel).
// carve_R is the instance count the buffer was LAID OUT for: the exact R when the host waited for it before allocating, or
// the speculative capacity when the forward pass was enqueued without waiting. It lives in the geometry header (word 4),
// so the backward kernels derive their pointers on the device and the host never has to know which of the two it was.
constexpr int CHUNK = 128;              // entries of a tile list per backward work item
constexpr int CKPT_FLOATS = 5 * TILE_X * TILE_Y;   // one checkpoint: 5 planes of 256 pixels
struct BinningPtrs { uint32_t* inst_gauss; float4* partials; uint64_t* keys; uint2* sorted; float* ckpt; char* end; };
__host__ __device__ inline BinningPtrs carve_binning(char* base, size_t carve_R, size_t cap_sorted)
{
    BinningPtrs b;
    uintptr_t p = (reinterpret_cast<uintptr_t>(base) + 255) & ~uintptr_t(255);
    b.inst_gauss = reinterpret_cast<uint32_t*>(p);
    p = (p + carve_R * sizeof(uint32_t) + 255) & ~uintptr_t(255);
    b.partials = reinterpret_cast<float4*>(p);
    b.keys = reinterpret_cast<uint64_t*>(p);
    p = (p + carve_R * 3 * sizeof(float4) + 255) & ~uintptr_t(255);
    b.sorted = reinterpret_cast<uint2*>(p);
    p = (p + cap_sorted * sizeof(uint2) + 255) & ~uintptr_t(255);
    b.ckpt = reinterpret_cast<float*>(p);            // checkpoint id = sorted position / CHUNK, ids 1 .. cap/CHUNK + 1
    b.end = reinterpret_cast<char*>(p + (cap_sorted / CHUNK + 2) * (size_t)CKPT_FLOATS * sizeof(float));
    return b;
}
// geometry header words (uint32): R, flags, R_alloc, longest tile list, carve_R, capacity of sorted[], number of list chunks
enum { HDR_R = 0, HDR_FLAGS = 1, HDR_R_ALLOC = 2, HDR_MAX_TILE = 3, HDR_CARVE_R = 4, HDR_CAP_SORTED = 5, HDR_CHUNKS = 6, HDR_WORDS = 8 };
enum { FLAG_PREFILTERED = 1u, FLAG_OVERFLOW = 2u };   // FLAG_OVERFLOW: the speculative binning capacity did not suffice

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Matrices arrive as the reference passes them: row-major memory of the TRANSPOSED maths matrix,
// i.e. m[0],m[4],m[8],m[12] is row 0 (auxiliary.h:58-77).
__device__ __forceinline__ f3 xform_point_4x3(f3 p, const float* __restrict__ m)
{
    return mk3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
               m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}

// XCD-aware tile order: block b runs on XCD b % 8 (observed, speed only); give each XCD a contiguous band of tiles
// so neighbouring tiles (which share Gaussians) hit the same L2. Bijective for any tile count.
__device__ __forceinline__ int xcd_tile_of_block(int b, int ntiles)
{
#if defined(GSR_NO_BANDING) && GSR_NO_BANDING
    return b;
#endif
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = b & 7, k = b >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

}  // namespace gsr
