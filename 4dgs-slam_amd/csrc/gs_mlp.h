// gs_mlp.h -- the deformation MLP of the 4D-Gaussians network as two fused kernels (forward; backward w.r.t. activations).
//
// Reference: utils/deformation.py:58-70 (create_net) and :101-149 (forward_dynamic) with the shipped flags (defor_depth 1, width 64,
// no_dx / no_ds / no_dr False, no_do / no_dshs True):
//     h0 = F W0^T + b0                       F  [n, in]   HexPlane features (in = 128)
//     a  = relu(h0)                          shared first ReLU of the three heads
//     u_j = a W1j^T + b1j,  v_j = relu(u_j)  j = position, scale, rotation           [n, 64] each
//     o_j = v_j W2j^T + b2j                  [n, 3], [n, 3], [n, 4]
// The reference runs this as 7 GEMMs + 6 ReLUs + 7 bias adds per call (and twice that on the way back), each a round trip of
// [n, 64] activations through HBM; the vendor GEMM handles the 3- and 4-column layers badly.
//
// Here one wave owns 16 points and carries them through all layers: v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain, so
// results match an fp32 GEMM to rounding order), weights straight from L2 in MFMA fragment order (the whole network is 85 KB), the
// activations of a layer handed to the next through an LDS tile (the MFMA result layout is rows = points in registers, the A
// operand wants rows = points in lanes).  The forward kernel writes only the 10 outputs per point; the backward kernel recomputes the
// activations from the features and produces dF and ALL parameter gradients (persistent register accumulators, see below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gs_linear.h"
#include "gs_dense.h"

namespace gsr {

constexpr int MLP_W = 64;                 // network width (arguments/__init__.py: net_width = 64)
constexpr int MLP_HEADS = 3;
constexpr int MLP_OUT = 10;               // 3 + 3 + 4
constexpr int MLP_BLOCK = 512;            // forward: 8 waves x 16 points per iteration, one persistent block per CU (weights in LDS; two waves per
                                          // SIMD: one runs MFMAs while the other sits at a barrier or waits for its tile -- 4 waves: 2.0 ms at config #3's 4 M rows)
constexpr int MLP_TILE = MLP_BLOCK / 64 * 16;
constexpr int MLP_LDA = MLP_W + 4;        // LDS row stride of a [16][64] tile (floats): 16-byte aligned, conflict-free float4 reads

struct MlpWeights {
    const float* W0; const float* b0;                      // [64, in], [64]
    const float* W1[MLP_HEADS]; const float* b1[MLP_HEADS]; // [64, 64], [64]
    const float* W2[MLP_HEADS]; const float* b2[MLP_HEADS]; // [o_j, 64], [o_j]
    int in_dim;                                             // multiple of 16, <= 128
    int out_dim[MLP_HEADS];                                 // <= 4 each
    int out_off[MLP_HEADS];                                 // column of head j in the [n, 10] output
};

__device__ __forceinline__ f32x4 mfma4(const float4 a, const float4 b, f32x4 c)   // four k-steps of one accumulator
{
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// four k-steps of four independent accumulators, interleaved so that consecutive MFMAs never depend on each other (a dependent
// 16x16x4 issues after 40 cycles instead of 32)
__device__ __forceinline__ void mfma4x4(const float4 a, const float4 (&b)[4], f32x4 (&c)[4])
{
#pragma unroll
    for (int t = 0; t < 4; t++) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, c[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, c[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t].z, c[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t].w, c[t], 0, 0, 0);
}

// Lane l = (i = l & 15, q = l >> 4).  A / B fragments are loaded as float4: K indices 16 S + 4 q + e, e = 0..3 (any assignment of K
// indices to k-steps is fine as long as A and B agree).  C layout: column = i, rows = 4 q + r.
template <int NT_IN>
__global__ void __launch_bounds__(MLP_BLOCK)
deform_mlp_fwd_kernel(const int64_t n, const float* __restrict__ feat, const MlpWeights w, float* __restrict__ out)
{
    constexpr int IN = 16 * NT_IN, LDW0 = IN + 4;
    __shared__ __attribute__((aligned(16))) float s_a[MLP_BLOCK / 64][16][MLP_LDA];
    __shared__ __attribute__((aligned(16))) float s_v[MLP_BLOCK / 64][16][MLP_LDA];   // one head at a time
    // the whole network in LDS (89 KB at in_dim 128), rows padded by 4 floats so that the 16 rows a quarter-wave reads as float4
    // fall into 16 different bank groups; read from L2 per tile the kernel ran at a third of the MFMA rate
    __shared__ __attribute__((aligned(16))) float s_W0[MLP_W][LDW0];
    __shared__ __attribute__((aligned(16))) float s_W1[MLP_HEADS][MLP_W][MLP_LDA];
    __shared__ __attribute__((aligned(16))) float s_W2[MLP_HEADS][4][MLP_LDA];
    for (int e = threadIdx.x; e < MLP_W * IN / 4; e += MLP_BLOCK) {
        const int r = e / (IN / 4), c = e - r * (IN / 4);
        *reinterpret_cast<float4*>(&s_W0[r][4 * c]) = *reinterpret_cast<const float4*>(w.W0 + (size_t)r * IN + 4 * c);
    }
#pragma unroll
    for (int j = 0; j < MLP_HEADS; j++) {
        for (int e = threadIdx.x; e < MLP_W * MLP_W / 4; e += MLP_BLOCK) {
            const int r = e / (MLP_W / 4), c = e - r * (MLP_W / 4);
            *reinterpret_cast<float4*>(&s_W1[j][r][4 * c]) = *reinterpret_cast<const float4*>(w.W1[j] + (size_t)r * MLP_W + 4 * c);
        }
        for (int e = threadIdx.x; e < 4 * MLP_W; e += MLP_BLOCK) {
            const int r = e / MLP_W, c = e - r * MLP_W;
            s_W2[j][r][c] = r < w.out_dim[j] ? w.W2[j][(size_t)r * MLP_W + c] : 0.f;
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int64_t tiles = (n + MLP_TILE - 1) / MLP_TILE;
    auto load_features = [&](int64_t tile, float4 (&dst)[NT_IN]) {   // this lane's A fragments of a tile: point i, 4 features per k-step group
        const int64_t prow = tile * MLP_TILE + wave * 16 + i;
#pragma unroll
        for (int S = 0; S < NT_IN; S++)
            dst[S] = (tile < tiles && prow < n) ? *reinterpret_cast<const float4*>(feat + prow * IN + 16 * S + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float bias0[4], bias1[MLP_HEADS][4], bias2[MLP_HEADS];        // this lane's columns, read once
#pragma unroll
    for (int t = 0; t < 4; t++) bias0[t] = w.b0[16 * t + i];
#pragma unroll
    for (int j = 0; j < MLP_HEADS; j++) {
#pragma unroll
        for (int t = 0; t < 4; t++) bias1[j][t] = w.b1[j][16 * t + i];
        bias2[j] = i < w.out_dim[j] ? w.b2[j][i] : 0.f;
    }
    float4 cur[NT_IN], nxt[NT_IN];
    load_features(blockIdx.x, cur);
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t p0 = tile * MLP_TILE + wave * 16;
        load_features(tile + gridDim.x, nxt);                     // in flight while this tile runs through the layers
        // ---- h0 = F W0^T + b0 ----
        f32x4 acc1[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc1[t] = f32x4{bias0[t], bias0[t], bias0[t], bias0[t]};
#pragma unroll
        for (int S = 0; S < NT_IN; S++) {
            const float4 a4 = cur[S];
            float4 b4[4];
#pragma unroll
            for (int t = 0; t < 4; t++) b4[t] = *reinterpret_cast<const float4*>(&s_W0[16 * t + i][16 * S + 4 * q]);
            mfma4x4(a4, b4, acc1);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                s_a[wave][4 * q + r][16 * t + i] = fmaxf(acc1[t][r], 0.f);
            }
        }
        __syncthreads();
        // ---- per head: u_j = a W1j^T + b1j, v_j = relu(u_j), o_j = v_j W2j^T + b2j (one 16-column tile, columns >= out_dim zero) ----
#pragma unroll
        for (int j = 0; j < MLP_HEADS; j++) {
            f32x4 acc2[4];
#pragma unroll
            for (int t = 0; t < 4; t++) acc2[t] = f32x4{bias1[j][t], bias1[j][t], bias1[j][t], bias1[j][t]};
#pragma unroll
            for (int S = 0; S < MLP_W / 16; S++) {
                const float4 a4 = *reinterpret_cast<const float4*>(&s_a[wave][i][16 * S + 4 * q]);
                float4 b4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) b4[t] = *reinterpret_cast<const float4*>(&s_W1[j][16 * t + i][16 * S + 4 * q]);
                mfma4x4(a4, b4, acc2);
            }
            if (j > 0) __syncthreads();                           // the previous head's v tile has been read
#pragma unroll
            for (int t = 0; t < 4; t++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    s_v[wave][4 * q + r][16 * t + i] = fmaxf(acc2[t][r], 0.f);
                }
            }
            __syncthreads();
            const bool col_ok = i < w.out_dim[j];
            f32x4 acc3 = f32x4{bias2[j], bias2[j], bias2[j], bias2[j]};
#pragma unroll
            for (int S = 0; S < MLP_W / 16; S++) {
                const float4 a4 = *reinterpret_cast<const float4*>(&s_v[wave][i][16 * S + 4 * q]);
                const float4 b4 = *reinterpret_cast<const float4*>(&s_W2[j][i & 3][16 * S + 4 * q]) * (col_ok ? 1.f : 0.f);
                acc3 = mfma4(a4, b4, acc3);
            }
            if (col_ok) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t p = p0 + 4 * q + r;
                    if (p < n) out[p * MLP_OUT + w.out_off[j] + i] = acc3[r];
                }
            }
        }
        __syncthreads();                                          // the tiles are rewritten by the next iteration
#pragma unroll
        for (int S = 0; S < NT_IN; S++) cur[S] = nxt[S];
    }
}

// ---- round 6: the same network on the bf16 matrix cores (three-term split, six products: gs_dense.h) -------------------------------------
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 matrix rate; with every fp32 operand split into three bf16 terms and the six cross terms
// of weight >= 2^-16 kept (gs_dense.h: fp32-GEMM accuracy, 1.8e-7 relative) a product costs 6/16 of its fp32 time. What bounds such a kernel is
// no longer the matrix pipe but its operands: a 16-byte fragment per lane and MFMA is 64 B / cycle / SIMD, twice what the LDS delivers to a CU.
// So the WEIGHTS ARE STATIONARY IN REGISTERS: wave w of the block's four holds the fragments of output features 16 w .. 16 w + 15 of W0 and
// of every W1j (120 registers, split once per kernel from the fp32 weights: no workspace, no extra launch), and all R = 16 RT rows of the
// block's tile stream through every wave; only the activations -- the second MFMA operand -- come from LDS, as bf16 planes [plane][row][k]
// written once by the layer that produced them. Products are formed TRANSPOSED (weight fragment first): a lane ends with FOUR CONSECUTIVE
// output features of one row -- one 8-byte LDS store per plane, already in the next layer's operand layout. The heads run one after the other
// through two v buffers (one barrier per head); W2j (3-4 rows) comes from a 4.6 KB LDS copy. Five block barriers per 16 RT rows.
constexpr int MLP3_THREADS = 256;
template <int NT_IN, int RT>
struct Mlp3Layout {
    static constexpr int IN = 16 * NT_IN, KS0 = (IN + 31) / 32, KP0 = 32 * KS0, R = 16 * RT;
    // bytes per row of a plane; the 16-byte chunks of a row are permuted (mlp3_off) so that the fragment reads are bank-conflict free
    static constexpr int LDF = 2 * KP0, LDA = 2 * MLP_W;
    static constexpr int OFF_F = 0, OFF_A = OFF_F + 3 * R * LDF, OFF_V = OFF_A + 3 * R * LDA, OFF_W2 = OFF_V + 2 * 3 * R * LDA;
    static constexpr int BYTES = OFF_W2 + MLP_HEADS * 3 * 4 * (2 * MLP_W);
    static constexpr int U = R * (KP0 / 4) / MLP3_THREADS;      // float4 of features per thread and tile
};

// Byte offset of the 16-byte chunk `c` (eight k) of row `row` in a plane whose rows hold CH chunks (4, 8 or 16: 32, 64 or 128 k). ds_read_b128
// serves a wave in four groups of sixteen lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same in the upper half (MI355X_MICROARCH.md,
// LDS) -- i.e. with lane = (row fi, chunk fq) the rows {0-3, 12-15} of one chunk together with the rows 4-11 of its neighbour: the XOR below
// sends those sixteen accesses to the sixteen 16-byte slots of a 256-byte bank row (enumerated for the three widths). Padding the rows by 16
// bytes -- the first layout -- left every group with 2-way conflicts.
template <int CH>
__device__ __forceinline__ int mlp3_off(int row, int c)
{
    const int r = row & 15;
    const int swz = CH == 16 ? r : CH == 8 ? (r >> 1) & 7 : ((0x1230 >> (4 * ((r >> 2) & 3))) & 3);      // CH == 4: rows 0-3 -> 0, 4-7 -> 3, 8-11 -> 2, 12-15 -> 1
    return row * (16 * CH) + 16 * (c ^ swz);
}

__device__ __forceinline__ void mlp3_split4(const float (&x)[4], uint2& h, uint2& m, uint2& l)
{
    uint32_t a[4], b[4], c[4];
#pragma unroll
    for (int e = 0; e < 4; e++) dense_split(x[e], a[e], b[e], c[e]);
    h = make_uint2(dense_pack(a[0], a[1]), dense_pack(a[2], a[3]));
    m = make_uint2(dense_pack(b[0], b[1]), dense_pack(b[2], b[3]));
    l = make_uint2(dense_pack(c[0], c[1]), dense_pack(c[2], c[3]));
}

// eight consecutive weights W[row][k0 .. k0 + 7] (zero beyond K) -> this lane's fragment of each plane
__device__ __forceinline__ void mlp3_weight_frag(const float* __restrict__ W, int ld, int K, int row, int k0, dense_frag (&dst)[3])
{
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; e++) x[e] = k0 + e < K ? W[(size_t)row * ld + k0 + e] : 0.f;
    dense_u4 h, m, l;
    dense_split8(x, h, m, l);
    dst[0] = __builtin_bit_cast(dense_frag, h); dst[1] = __builtin_bit_cast(dense_frag, m); dst[2] = __builtin_bit_cast(dense_frag, l);
}

// the six products of ONE weight fragment set with RT activation fragment sets, smallest terms first. Every accumulator is kept as TWO partial
// sums (c0: lo hi + mid mid + hi mid; c1: hi lo + mid hi + hi hi), added by the caller at the end: 2 RT independent chains, so that a dependent
// v_mfma_f32_16x16x32_bf16 is at least 2 RT - 1 issues behind its predecessor (back to back it waits for the predecessor's passes: with two
// row tiles and one chain each the matrix phases ran at half rate).
template <int RT>
__device__ __forceinline__ void mlp3_mfma6(const dense_frag (&a)[3], const dense_frag (&b)[RT][3], dense_acc (&c0)[RT], dense_acc (&c1)[RT])
{
#define GSR_MLP3_TERM(C, PA, PB)                                                                                  \
    _Pragma("unroll") for (int t = 0; t < RT; t++) C[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA], b[t][PB], C[t], 0, 0, 0);
    GSR_MLP3_TERM(c0, 2, 0) GSR_MLP3_TERM(c1, 0, 2) GSR_MLP3_TERM(c0, 1, 1) GSR_MLP3_TERM(c1, 1, 0) GSR_MLP3_TERM(c0, 0, 1) GSR_MLP3_TERM(c1, 0, 0)
#undef GSR_MLP3_TERM
}

template <int NT_IN, int RT>
__global__ void __launch_bounds__(MLP3_THREADS) __attribute__((amdgpu_waves_per_eu(RT >= 4 ? 1 : 2, RT >= 4 ? 1 : 2)))
deform_mlp_fwd3_kernel(const int64_t n, const float* __restrict__ feat, const MlpWeights w, float* __restrict__ out)
{
    using L = Mlp3Layout<NT_IN, RT>;
    constexpr int IN = L::IN, KS0 = L::KS0, KP0 = L::KP0, R = L::R, LDF = L::LDF, LDA = L::LDA, U = L::U;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    unsigned char* const s_f = s_mem + L::OFF_F;          // [3][R][LDF]   the tile's features
    unsigned char* const s_a = s_mem + L::OFF_A;          // [3][R][LDA]   a = relu(h0)
    unsigned char* const s_v = s_mem + L::OFF_V;          // [2][3][R][LDA] v_j = relu(u_j), heads alternate
    unsigned char* const s_w2 = s_mem + L::OFF_W2;        // [head][plane][4 rows][64] W2j (rows >= o_j zero)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fq = lane >> 4;

    // ---- this wave's share of the network, once per kernel: fragments (output feature 16 wave + fi, k = 32 s + 8 fq ..) of W0 and the W1j ----
    dense_frag w0[KS0][3], w1[MLP_HEADS][2][3];
#pragma unroll
    for (int s = 0; s < KS0; s++) mlp3_weight_frag(w.W0, IN, IN, 16 * wave + fi, 32 * s + 8 * fq, w0[s]);
#pragma unroll
    for (int j = 0; j < MLP_HEADS; j++)
#pragma unroll
        for (int s = 0; s < 2; s++) mlp3_weight_frag(w.W1[j], MLP_W, MLP_W, 16 * wave + fi, 32 * s + 8 * fq, w1[j][s]);
    for (int e = tid; e < MLP_HEADS * 4 * MLP_W; e += MLP3_THREADS) {
        const int j = e / (4 * MLP_W), row = (e / MLP_W) & 3, k = e & (MLP_W - 1);
        uint32_t hi, mid, lo;
        dense_split(row < w.out_dim[j] ? w.W2[j][(size_t)row * MLP_W + k] : 0.f, hi, mid, lo);
        unsigned short* dst = reinterpret_cast<unsigned short*>(s_w2 + ((j * 3) * 4 + row) * (2 * MLP_W)) + k;
        dst[0] = (unsigned short)(hi >> 16); dst[4 * MLP_W] = (unsigned short)(mid >> 16); dst[8 * MLP_W] = (unsigned short)(lo >> 16);
    }
    float b0v[4], b1v[MLP_HEADS][4], b2v[MLP_HEADS][4];     // biases of this lane's four output features (transposed result: features 4 fq + r)
#pragma unroll
    for (int r = 0; r < 4; r++) b0v[r] = w.b0[16 * wave + 4 * fq + r];
#pragma unroll
    for (int j = 0; j < MLP_HEADS; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            b1v[j][r] = w.b1[j][16 * wave + 4 * fq + r];
            b2v[j][r] = (fq == 0 && r < w.out_dim[j]) ? w.b2[j][r] : 0.f;
        }

    const int64_t tiles = (n + R - 1) / R;
    auto load_tile = [&](int64_t tile, float4 (&dst)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + MLP3_THREADS * u, row = e / (KP0 / 4), k = 4 * (e % (KP0 / 4));
            const int64_t p = tile * R + row;
            dst[u] = (tile < tiles && p < n && k < IN) ? *reinterpret_cast<const float4*>(feat + p * IN + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto relu_split_store = [&](const dense_acc& acc, unsigned char* base, int row) {     // four features 16 wave + 4 fq .. of row `row` -> three planes
        const float x[4] = {fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f)};
        uint2 h, m, l;
        mlp3_split4(x, h, m, l);
        unsigned char* d = base + mlp3_off<MLP_W / 8>(row, 2 * wave + (fq >> 1)) + 8 * (fq & 1);
        *reinterpret_cast<uint2*>(d) = h; *reinterpret_cast<uint2*>(d + R * LDA) = m; *reinterpret_cast<uint2*>(d + 2 * R * LDA) = l;
    };
    float4 cur[U], nxt[U];
    load_tile(blockIdx.x, cur);
    __syncthreads();                                              // s_w2 is complete
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        // ---- stage the tile's features as planes; request the next tile's ----
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + MLP3_THREADS * u, row = e / (KP0 / 4), k = 4 * (e % (KP0 / 4));
            const float x[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
            uint2 h, m, l;
            mlp3_split4(x, h, m, l);
            unsigned char* d = s_f + mlp3_off<KP0 / 8>(row, k >> 3) + 2 * (k & 7);
            *reinterpret_cast<uint2*>(d) = h; *reinterpret_cast<uint2*>(d + R * LDF) = m; *reinterpret_cast<uint2*>(d + 2 * R * LDF) = l;
        }
        load_tile(tile + gridDim.x, nxt);
        __syncthreads();
        // ---- h0 = F W0^T + b0 -> a = relu(h0): this wave's 16 features, all rows ----
        {
            dense_acc acc[RT], acc1[RT];
#pragma unroll
            for (int t = 0; t < RT; t++) { acc[t] = dense_acc{0.f, 0.f, 0.f, 0.f}; acc1[t] = dense_acc{b0v[0], b0v[1], b0v[2], b0v[3]}; }
#pragma unroll
            for (int s = 0; s < KS0; s++) {
                dense_frag b[RT][3];
#pragma unroll
                for (int t = 0; t < RT; t++)
#pragma unroll
                    for (int p = 0; p < 3; p++) b[t][p] = dense_ld_frag(s_f + p * (R * LDF) + mlp3_off<KP0 / 8>(16 * t + fi, 4 * s + fq));
                mlp3_mfma6<RT>(w0[s], b, acc, acc1);
            }
#pragma unroll
            for (int t = 0; t < RT; t++) relu_split_store(acc[t] + acc1[t], s_a, 16 * t + fi);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MLP_HEADS; j++) {
            unsigned char* const sv = s_v + (j & 1) * (3 * R * LDA);
            // ---- u_j = a W1j^T + b1j -> v_j = relu(u_j) ----
            {
                dense_acc acc[RT], acc1[RT];
#pragma unroll
                for (int t = 0; t < RT; t++) { acc[t] = dense_acc{0.f, 0.f, 0.f, 0.f}; acc1[t] = dense_acc{b1v[j][0], b1v[j][1], b1v[j][2], b1v[j][3]}; }
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    dense_frag b[RT][3];
#pragma unroll
                    for (int t = 0; t < RT; t++)
#pragma unroll
                        for (int p = 0; p < 3; p++) b[t][p] = dense_ld_frag(s_a + p * (R * LDA) + mlp3_off<MLP_W / 8>(16 * t + fi, 4 * s + fq));
                    mlp3_mfma6<RT>(w1[j][s], b, acc, acc1);
                }
#pragma unroll
                for (int t = 0; t < RT; t++) relu_split_store(acc[t] + acc1[t], sv, 16 * t + fi);
            }
            __syncthreads();
            // ---- o_j = v_j W2j^T + b2j: one 16-feature tile (o_j valid), the row tiles dealt to the waves ----
            for (int t = wave; t < RT; t += MLP3_THREADS / 64) {
                // (the two k steps as the "row tiles" of mlp3_mfma6: four independent chains of three products)
                dense_frag a[2][3], b[2][3];
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int p = 0; p < 3; p++) {
                        a[s][p] = fi < 4 ? dense_ld_frag(s_w2 + ((j * 3 + p) * 4 + fi) * (2 * MLP_W) + 2 * (32 * s + 8 * fq)) : dense_frag{0, 0, 0, 0, 0, 0, 0, 0};
                        b[s][p] = dense_ld_frag(sv + p * (R * LDA) + mlp3_off<MLP_W / 8>(16 * t + fi, 4 * s + fq));
                    }
                dense_acc c0[2] = {dense_acc{0.f, 0.f, 0.f, 0.f}, dense_acc{0.f, 0.f, 0.f, 0.f}};
                dense_acc c1[2] = {dense_acc{b2v[j][0], b2v[j][1], b2v[j][2], b2v[j][3]}, dense_acc{0.f, 0.f, 0.f, 0.f}};
#define GSR_MLP3_TERM2(C, PA, PB)                                                                                  \
    _Pragma("unroll") for (int s = 0; s < 2; s++) C[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][PA], b[s][PB], C[s], 0, 0, 0);
                GSR_MLP3_TERM2(c0, 2, 0) GSR_MLP3_TERM2(c1, 0, 2) GSR_MLP3_TERM2(c0, 1, 1) GSR_MLP3_TERM2(c1, 1, 0) GSR_MLP3_TERM2(c0, 0, 1) GSR_MLP3_TERM2(c1, 0, 0)
#undef GSR_MLP3_TERM2
                const dense_acc acc = (c0[0] + c1[0]) + (c0[1] + c1[1]);
                const int64_t p = tile * R + 16 * t + fi;
                if (fq == 0 && p < n) {
#pragma unroll
                    for (int r = 0; r < 4; r++) if (r < w.out_dim[j]) out[p * MLP_OUT + w.out_off[j] + r] = acc[r];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) cur[u] = nxt[u];
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------------------------
// One block = 4 waves = 64 points per iteration, persistent over the batch.  Nothing of the forward pass is read back: the block
// recomputes h0, a, u_j, v_j from the features (cheaper than the 1 KB per point of pre-activations the forward pass would have to
// store and three later kernels re-read), forms du_j, dh0, dF, and accumulates EVERY weight gradient in registers across its
// iterations -- each wave owns a quarter of the output tiles of dW0 / dW1j / dW2j and reduces over the block's 64 points, with the
// activation tiles of all four waves shared through LDS.  W0 and the W1j live in LDS too (86 KB of the block's 141 KB); the features
// -- B operand of dW0 -- are re-read from L2, the block having read them a moment before.  At the end the block writes one row of
// partial sums; a second kernel adds the rows in a fixed order.  HBM traffic: features + dout in, dF out.
//
// flat gradient layout (also of a partial row): W0 [64][in] | b0 [64] | per head j: W1j [64][64] | b1j [64] | W2j [o_j][64] | b2j [o_j]
struct MlpGradLayout {
    int W0, b0, W1[MLP_HEADS], b1[MLP_HEADS], W2[MLP_HEADS], b2[MLP_HEADS], total;
};

__host__ __device__ inline MlpGradLayout mlp_grad_layout(int in_dim, const int* out_dim)
{
    MlpGradLayout g;
    int off = 0;
    g.W0 = off; off += MLP_W * in_dim;
    g.b0 = off; off += MLP_W;
    for (int j = 0; j < MLP_HEADS; j++) {
        g.W1[j] = off; off += MLP_W * MLP_W;
        g.b1[j] = off; off += MLP_W;
        g.W2[j] = off; off += out_dim[j] * MLP_W;
        g.b2[j] = off; off += out_dim[j];
    }
    g.total = off;
    return g;
}

constexpr int MLPB_BLOCK = 256;           // 4 waves
constexpr int MLPB_TILE = 64;             // points per block iteration

template <int NT_IN>                      // in_dim / 16
__global__ void __launch_bounds__(MLPB_BLOCK)
deform_mlp_bwd_kernel(const int64_t n_all, const float* __restrict__ feat, const float* __restrict__ dout, const MlpWeights w,
                      float* __restrict__ dfeat, float* __restrict__ partial, const int32_t* __restrict__ rows, const int32_t* __restrict__ n_rows)
{
    // rows != nullptr: only the listed rows of feat / dout / dfeat take part (gsr_deform_mlp_backward_rows: the rows of a batch whose
    // cotangent is not zero, in ascending order); the block then walks the LIST 64 entries at a time and every row access goes through it.
    const int64_t n = rows ? (int64_t)*n_rows : n_all;
    constexpr int IN = 16 * NT_IN;
    __shared__ int64_t s_row[MLPB_TILE];                                          // row of feat / dout / dfeat of the tile's points (-1: none)
    __shared__ __attribute__((aligned(16))) float s_W0[MLP_W][IN + 4];          // W0 (h0 recompute and dF); the features for dW0 are re-read from L2
    __shared__ __attribute__((aligned(16))) float s_a[MLPB_TILE][MLP_LDA];      // a = relu(h0)
    __shared__ __attribute__((aligned(16))) float s_v[MLPB_TILE][MLP_LDA];      // v_j = relu(u_j), one head at a time
    __shared__ __attribute__((aligned(16))) float s_d[MLPB_TILE][MLP_LDA];      // du_j, one head at a time; then dh0
    __shared__ float s_o[MLPB_TILE][12];                                          // dout of the block's points (10 used)
    __shared__ __attribute__((aligned(16))) float s_W1[MLP_HEADS][MLP_W][MLP_LDA];  // u_j recompute and dL/da, 52 KB
    for (int e = threadIdx.x; e < MLP_W * IN / 4; e += MLPB_BLOCK) {
        const int r = e / (IN / 4), c = e - r * (IN / 4);
        *reinterpret_cast<float4*>(&s_W0[r][4 * c]) = *reinterpret_cast<const float4*>(w.W0 + (size_t)r * IN + 4 * c);
    }
#pragma unroll
    for (int j = 0; j < MLP_HEADS; j++) {
        for (int e = threadIdx.x; e < MLP_W * MLP_W / 4; e += MLPB_BLOCK) {
            const int r = e / (MLP_W / 4), c = e - r * (MLP_W / 4);
            *reinterpret_cast<float4*>(&s_W1[j][r][4 * c]) = *reinterpret_cast<const float4*>(w.W1[j] + (size_t)r * MLP_W + 4 * c);
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int bc = threadIdx.x & 63, bg = threadIdx.x >> 6;                       // bias sums: column bc over rows 16 bg .. 16 bg + 15

    // persistent weight-gradient accumulators (this wave's share) ...
    f32x4 gW0[NT_IN];                     // rows 16 wave .. +15 of dW0, all in-tiles
    f32x4 gW1[MLP_HEADS][4];              // rows 16 wave .. +15 of dW1j
    f32x4 gW2[MLP_HEADS];                 // rows 0 .. 15 (o_j valid) x columns 16 wave .. +15 of dW2j
#pragma unroll
    for (int t = 0; t < NT_IN; t++) gW0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MLP_HEADS; j++) {
        gW2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; t++) gW1[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float gb0 = 0.f, gb1[MLP_HEADS] = {0.f, 0.f, 0.f}, gb2 = 0.f;                 // ... and bias sums (this thread's column / row group)
    float bias0[4], bias1[MLP_HEADS][4];                                          // this lane's columns of b0 / b1j, read once
#pragma unroll
    for (int t = 0; t < 4; t++) bias0[t] = w.b0[16 * t + i];
#pragma unroll
    for (int j = 0; j < MLP_HEADS; j++)
#pragma unroll
        for (int t = 0; t < 4; t++) bias1[j][t] = w.b1[j][16 * t + i];

    const int64_t tiles = (n + MLPB_TILE - 1) / MLPB_TILE;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t pblk = tile * MLPB_TILE;
        const int r0 = wave * 16;                                 // this wave's rows of the block tile
        if (threadIdx.x < MLPB_TILE) {
            const int64_t e = pblk + threadIdx.x;
            s_row[threadIdx.x] = e < n ? (rows ? (int64_t)rows[e] : e) : -1;
        }
        __syncthreads();
        const int64_t prow = s_row[r0 + i];
        // stage dout (zero beyond n: those rows then contribute nothing anywhere)
        int nonzero = 0;
        for (int e = threadIdx.x; e < MLPB_TILE * MLP_OUT; e += MLPB_BLOCK) {
            const int r = e / MLP_OUT, c = e - r * MLP_OUT;
            const int64_t pr = s_row[r];
            const float v = pr >= 0 ? dout[pr * MLP_OUT + c] : 0.f;
            s_o[r][c] = v;
            nonzero |= v != 0.f;
        }
        if (!__syncthreads_or(nonzero)) {                         // 64 points no gradient reaches (Gaussians the view does not see): dF = 0
            for (int e = threadIdx.x; e < MLPB_TILE * IN / 4; e += MLPB_BLOCK) {
                const int r = e / (IN / 4), c = e - r * (IN / 4);
                const int64_t pr = s_row[r];
                if (pr >= 0) *reinterpret_cast<float4*>(dfeat + pr * IN + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __syncthreads();                                      // s_row is rewritten by the next iteration
            continue;
        }
        // ---- recompute h0 = F W0^T + b0 for the wave's 16 points; keep the features and a = relu(h0) in LDS ----
        f32x4 acc1[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc1[t] = f32x4{bias0[t], bias0[t], bias0[t], bias0[t]};
#pragma unroll
        for (int S = 0; S < NT_IN; S++) {
            const float4 a4 = prow >= 0 ? *reinterpret_cast<const float4*>(feat + prow * IN + 16 * S + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 b4[4];
#pragma unroll
            for (int t = 0; t < 4; t++) b4[t] = *reinterpret_cast<const float4*>(&s_W0[16 * t + i][16 * S + 4 * q]);
            mfma4x4(a4, b4, acc1);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) s_a[r0 + 4 * q + r][16 * t + i] = fmaxf(acc1[t][r], 0.f);
        }
        __syncthreads();
        if (bc < MLP_OUT) {
#pragma unroll
            for (int r = 0; r < 16; r++) gb2 += s_o[16 * bg + r][bc];
        }
        f32x4 da[4];                                              // dL/da of the wave's points
#pragma unroll
        for (int t = 0; t < 4; t++) da[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < MLP_HEADS; j++) {
            // u_j = a W1j^T + b1j (wave's points)
            f32x4 acc2[4];
#pragma unroll
            for (int t = 0; t < 4; t++) acc2[t] = f32x4{bias1[j][t], bias1[j][t], bias1[j][t], bias1[j][t]};
#pragma unroll
            for (int S = 0; S < MLP_W / 16; S++) {
                const float4 a4 = *reinterpret_cast<const float4*>(&s_a[r0 + i][16 * S + 4 * q]);
                float4 b4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) b4[t] = *reinterpret_cast<const float4*>(&s_W1[j][16 * t + i][16 * S + 4 * q]);
                mfma4x4(a4, b4, acc2);
            }
            // dv_j = dout_j W2j (one k-step), du_j = dv_j * [u_j > 0]; v_j = relu(u_j)
            const bool k_ok = q < w.out_dim[j];
            const float ao = k_ok ? s_o[r0 + i][w.out_off[j] + q] : 0.f;
            if (j > 0) __syncthreads();                           // every wave is done with the previous head's s_v / s_d
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float b = k_ok ? w.W2[j][(size_t)q * MLP_W + 16 * t + i] : 0.f;
                const f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x4f32(ao, b, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    s_v[r0 + 4 * q + r][16 * t + i] = fmaxf(acc2[t][r], 0.f);
                    s_d[r0 + 4 * q + r][16 * t + i] = acc2[t][r] > 0.f ? dv[r] : 0.f;
                }
            }
            __syncthreads();
            // bias gradient b1j: column sums of du_j
#pragma unroll
            for (int r = 0; r < 16; r++) gb1[j] += s_d[16 * bg + r][bc];
            // dW2j += dout_j^T v_j over the block's 64 points: rows = outputs (o_j valid), this wave's 16 columns
            // dW1j += du_j^T a: this wave's 16 rows, all 64 columns.   A[i][k = point], B[k = point][column]: dword LDS reads
#pragma unroll
            for (int S = 0; S < MLPB_TILE / 4; S++) {
                const int pt = 4 * S + q;
                const float ao2 = i < w.out_dim[j] ? s_o[pt][w.out_off[j] + i] : 0.f;
                gW2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ao2, s_v[pt][16 * wave + i], gW2[j], 0, 0, 0);
                const float ad = s_d[pt][16 * wave + i];
#pragma unroll
                for (int t = 0; t < 4; t++) gW1[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ad, s_a[pt][16 * t + i], gW1[j][t], 0, 0, 0);
            }
            // dL/da += du_j W1j (wave's points):  B[k][col] = W1j[k][col]
#pragma unroll
            for (int S = 0; S < MLP_W / 16; S++) {
                const float4 a4 = *reinterpret_cast<const float4*>(&s_d[r0 + i][16 * S + 4 * q]);
                const float* wrow = &s_W1[j][16 * S + 4 * q][i];     // B[k][col] = W1j[k][col]: four rows of the LDS copy per float4 of A
                float4 b4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) b4[t] = make_float4(wrow[16 * t], wrow[MLP_LDA + 16 * t], wrow[2 * MLP_LDA + 16 * t], wrow[3 * MLP_LDA + 16 * t]);
                mfma4x4(a4, b4, da);
            }
        }
        __syncthreads();                                          // the last head's s_d has been read by every wave
        // ---- dh0 = dL/da * [a > 0] -> s_d ----
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) s_d[r0 + 4 * q + r][16 * t + i] = s_a[r0 + 4 * q + r][16 * t + i] > 0.f ? da[t][r] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) gb0 += s_d[16 * bg + r][bc];
        // ---- dW0 += dh0^T F over the block's points: this wave's 16 rows, all in-tiles ----
#pragma unroll
        for (int S = 0; S < MLPB_TILE / 4; S++) {
            const int pt = 4 * S + q;
            const float ad = s_d[pt][16 * wave + i];
            const int64_t pg = s_row[pt] >= 0 ? s_row[pt] : s_row[0];  // rows beyond n have dh0 = 0: any valid row will do (the tile's first is one)
            float fb[NT_IN];                                          // the block read these rows a moment ago: L2
#pragma unroll
            for (int t = 0; t < NT_IN; t++) fb[t] = feat[pg * IN + 16 * t + i];
#pragma unroll
            for (int t = 0; t < NT_IN; t++) gW0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ad, fb[t], gW0[t], 0, 0, 0);
        }
        // ---- dF = dh0 W0 (wave's points), four column tiles at a time ----
#pragma unroll
        for (int t0 = 0; t0 < NT_IN; t0 += 4) {
            f32x4 accf[4];
#pragma unroll
            for (int t = 0; t < 4; t++) accf[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < MLP_W / 16; S++) {
                const float4 a4 = *reinterpret_cast<const float4*>(&s_d[r0 + i][16 * S + 4 * q]);
                const float* wrow = &s_W0[16 * S + 4 * q][16 * t0 + i];   // B[k][col] = W0[k][col]: four rows of the LDS copy
                constexpr int LDW = IN + 4;
                float4 b4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int c = t0 + t < NT_IN ? 16 * t : 0;    // in_dim not a multiple of 64: the surplus tiles redo tile t0 and are dropped
                    b4[t] = make_float4(wrow[c], wrow[LDW + c], wrow[2 * LDW + c], wrow[3 * LDW + c]);
                }
                mfma4x4(a4, b4, accf);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (t0 + t < NT_IN) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int64_t p = s_row[r0 + 4 * q + r];
                        if (p >= 0) dfeat[p * IN + 16 * (t0 + t) + i] = accf[t][r];
                    }
                }
            }
        }
        __syncthreads();                                          // tiles are rewritten by the next iteration
    }

    // ---- the block's partial row ----
    const MlpGradLayout g = mlp_grad_layout(IN, w.out_dim);
    float* row = partial + (size_t)blockIdx.x * g.total;
#pragma unroll
    for (int t = 0; t < NT_IN; t++) {
#pragma unroll
        for (int r = 0; r < 4; r++) row[g.W0 + (size_t)(16 * wave + 4 * q + r) * IN + 16 * t + i] = gW0[t][r];
    }
#pragma unroll
    for (int j = 0; j < MLP_HEADS; j++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) row[g.W1[j] + (16 * wave + 4 * q + r) * MLP_W + 16 * t + i] = gW1[j][t][r];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (4 * q + r < w.out_dim[j]) row[g.W2[j] + (4 * q + r) * MLP_W + 16 * wave + i] = gW2[j][r];
        }
    }
    // bias sums: add the four row groups through LDS (s_a is free now)
    __syncthreads();
    float* s_b = &s_a[0][0];                                      // [4][64 * 5]: b0, b1 x3, b2
    s_b[bg * 320 + bc] = gb0;
#pragma unroll
    for (int j = 0; j < MLP_HEADS; j++) s_b[bg * 320 + 64 * (1 + j) + bc] = gb1[j];
    s_b[bg * 320 + 256 + bc] = gb2;
    __syncthreads();
    if (bg == 0) {
        auto tot = [&](int k) { return (s_b[k] + s_b[320 + k]) + (s_b[640 + k] + s_b[960 + k]); };
        row[g.b0 + bc] = tot(bc);
#pragma unroll
        for (int j = 0; j < MLP_HEADS; j++) row[g.b1[j] + bc] = tot(64 * (1 + j) + bc);
        if (bc < MLP_OUT) {
            int j = 0;
            while (j + 1 < MLP_HEADS && bc >= w.out_off[j + 1]) j++;
            row[g.b2[j] + bc - w.out_off[j]] = tot(256 + bc);
        }
    }
}

// grads[e] = sum over the G partial rows, fixed order
__global__ void __launch_bounds__(256)
mlp_grad_reduce_kernel(const int G, const int total, const float* __restrict__ partial, float* __restrict__ grads)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < G; b += 4) {
        s0 += partial[(size_t)b * total + e];
        s1 += partial[(size_t)(b + 1) * total + e];
        s2 += partial[(size_t)(b + 2) * total + e];
        s3 += partial[(size_t)(b + 3) * total + e];
    }
    for (; b < G; b++) s0 += partial[(size_t)b * total + e];
    grads[e] = (s0 + s1) + (s2 + s3);
}

}  // namespace gsr
