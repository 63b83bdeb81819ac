// gs_dense.h -- fp32-accurate dense layers on the bf16 matrix cores (include/dense_layers.h): the node network's trunk.
//
// Reference: utils/time_utils.py:327-476 (DeformNetwork: D = 8 layers of W = 256 on 20-70 000 rows per mapping iteration, the embedding
// re-injected behind layer 4). As fp32 GEMMs (hipBLASLt: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate, 1/16 of the bf16 matrix rate)
// they were 1.0 of the dynamic mapping iteration's 2.7 ms of device time, at 82-98 TFLOP/s.
//
// Here every fp32 operand is split into THREE bf16 terms, x = hi + mid + lo with hi = x truncated to bf16 (8 significant bits), mid = (x - hi)
// truncated, lo = (x - hi - mid) truncated: together the 24 bits of an fp32 significand, the remainders exact. A product x w is evaluated as
// the six cross terms of weight >= 2^-16 -- hi hi, hi mid, mid hi, hi lo, mid mid, lo hi -- on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation; the three dropped terms are below 2^-24 |x w| each (1.8e-7 relative in all: one and a half fp32 ulps, the size of an fp32
// GEMM's own rounding). Six MFMA products at 16x the fp32 rate: 2.7x the fp32 matrix peak at equal utilisation, and the result is an fp32 GEMM
// for every purpose of the parity tests (values 1e-6, the golden node losses unchanged at their tolerances).
//
// Kernels:  dense_split_kernel / dense_split_many_kernel   W [N, K] fp32 -> bf16 planes [3][Npad][Kpad] (and / or of W^T), zero padded, once per
//                                  optimizer step (all of a network's weights in one launch);
//           dense_fwd8_kernel<BT>  Y = act(X W^T + b) [x (mask > 0), + column sums] for full-width outputs (N % 256 == 0): ONE block of eight
//                                  waves per CU, 16 BT x 256 tile, two LDS stages -- the node network's forward and input gradient;
//           dense_fwd_kernel       the same product on 128-column tiles, two blocks per CU (other widths; the first form);
//           dense_wgrad_kernel     dW = G^T X over a slice of the rows (both operands split while staged, transposed through LDS);
//           dense_wgrad_sum_kernel the slices' partial results added in a fixed order;
//           trunk_fwd_kernel       the eight layers + heads in one launch (opt-in).
//
// STATUS (round 5, tools/dev_dense.py / dev_dense_chain.py on an MI355X, profiles/r05_dense_layers.jsonl; library = hipBLASLt fp32 through torch):
// correct to fp32-GEMM accuracy everywhere (tests/test_hip_dense.py), and with dense_fwd8_kernel faster than the library at the node network's
// batch --
//     rows      forward 256x256     input gradient (masked, + bias sums)          weight gradient (library: row groups of ~2000)
//     33 280    32.4 vs 46.1 us     35.0 vs 46.4 (+ 21 for the library's pass)    83.6 vs 43.7
//     66 560    60.4 vs 113.8       68.0 vs 94.3 (+ 34)                           144 vs 77
// slam/deform_model._FusedTrunk uses the forward / input-gradient kernels by default and keeps the library for the weight gradients. The
// layer-fused trunk_fwd_kernel (a 64-row tile's activations resident in LDS across the eight layers) is slower than eight chained
// dense_fwd8 launches (420 vs 277 us at 33 k rows): bound by re-reading the network's planes from L2 per 64 rows at one wave per SIMD.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

typedef short dense_frag __attribute__((ext_vector_type(8)));   // 8 bf16 = one A / B operand of v_mfma_f32_16x16x32_bf16
typedef float dense_acc __attribute__((ext_vector_type(4)));
typedef uint32_t dense_u4 __attribute__((ext_vector_type(4)));     // 16 bytes as a native vector (arrays of HIP's uint4 struct went to scratch memory)

constexpr int DENSE_BM = 128, DENSE_BN = 128, DENSE_BK = 32;   // block tile; a K step is one MFMA deep
constexpr int DENSE_THREADS = 256;                              // four waves, 2 x 2, 64 x 64 outputs each
constexpr int DENSE_ROW_B = 64;                                 // bytes per staged row: 32 bf16 in four 16-byte chunks
// byte offset of chunk `q` (eight k) of staged row `row`: the chunks of a row are permuted by (row >> 1) & 3 -- with that the 16-byte fragment
// reads of a wave (ds_read_b128 serves lanes {0-3, 12-15, 20-27}, ... together: rows and chunks mixed) and the staging stores (eight
// consecutive lanes: two rows x four chunks, or eight rows of one chunk) are all bank-conflict free without padding (checked by enumeration)
__device__ __forceinline__ int dense_off(int row, int q) { return row * DENSE_ROW_B + 16 * (q ^ ((row >> 1) & 3)); }

// x -> (hi, mid, lo) as the upper halves of three fp32 words (truncation: every remainder is exact)
__device__ __forceinline__ void dense_split(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo)
{
    hi = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mid);
    lo = __float_as_uint(r2) & 0xFFFF0000u;
}
__device__ __forceinline__ uint32_t dense_pack(uint32_t even, uint32_t odd) { return (even >> 16) | odd; }   // two bf16: element 2j low, 2j + 1 high

// eight consecutive fp32 -> three 16-byte fragments (one per plane)
__device__ __forceinline__ void dense_split8(const float (&x)[8], dense_u4& h, dense_u4& m, dense_u4& l)
{
    uint32_t a[8], b[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; e++) dense_split(x[e], a[e], b[e], c[e]);
    h = dense_u4{dense_pack(a[0], a[1]), dense_pack(a[2], a[3]), dense_pack(a[4], a[5]), dense_pack(a[6], a[7])};
    m = dense_u4{dense_pack(b[0], b[1]), dense_pack(b[2], b[3]), dense_pack(b[4], b[5]), dense_pack(b[6], b[7])};
    l = dense_u4{dense_pack(c[0], c[1]), dense_pack(c[2], c[3]), dense_pack(c[4], c[5]), dense_pack(c[6], c[7])};
}

__device__ __forceinline__ dense_frag dense_ld_frag(const unsigned char* p)
{
    return *reinterpret_cast<const dense_frag*>(p);
}

// the six products of one row tile with FOUR column tiles for one K step: consecutive MFMAs go to different accumulators (a dependent
// v_mfma_f32_16x16x32_bf16 waits for its predecessor's passes; four in between hide that), smallest terms first
__device__ __forceinline__ void dense_mfma6x4(const dense_frag (&a)[3], const dense_frag (&b)[4][3], dense_acc (&c)[4])
{
#define GSR_DENSE_TERM(PA, PB)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; j++) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA], b[j][PB], c[j], 0, 0, 0);
    GSR_DENSE_TERM(2, 0) GSR_DENSE_TERM(0, 2) GSR_DENSE_TERM(1, 1) GSR_DENSE_TERM(1, 0) GSR_DENSE_TERM(0, 1) GSR_DENSE_TERM(0, 0)
#undef GSR_DENSE_TERM
}

// ---- weights -> bf16 planes ------------------------------------------------------------------------------------------------------------
// planes[p][n][k], n < Npad, k < Kpad (multiples of 128 / 32), = plane p of W[n][k0 + k] (zero beyond N / K); transposed = 1 writes the planes of
// the TRANSPOSE instead: planes[p][k][n] with k < Kpad' = round_up(K, 128) rows and n < Npad' = round_up(N, 32) columns.
__global__ void __launch_bounds__(256)
dense_split_kernel(const int N, const int K, const float* __restrict__ W, const int ldw, const int k0, const int transposed,
                   unsigned short* __restrict__ planes, const int rows_pad, const int cols_pad)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)rows_pad * cols_pad) return;
    const int r = (int)(e / cols_pad), c = (int)(e % cols_pad);
    const int n = transposed ? c : r, k = transposed ? r : c;
    const float x = (n < N && k < K) ? W[(size_t)n * ldw + k0 + k] : 0.f;
    uint32_t hi, mid, lo;
    dense_split(x, hi, mid, lo);
    const size_t plane = (size_t)rows_pad * cols_pad;
    planes[e] = (unsigned short)(hi >> 16);
    planes[plane + e] = (unsigned short)(mid >> 16);
    planes[2 * plane + e] = (unsigned short)(lo >> 16);
}

// several weights in one launch (the network's layers, both orientations, once per optimizer step): blockIdx.y = item
constexpr int DENSE_SPLIT_MAX = 24;
struct DenseSplitItem { const float* W; unsigned short* planes; int N, K, ldw, k0, transposed, rows_pad, cols_pad; };
struct DenseSplitItems { DenseSplitItem item[DENSE_SPLIT_MAX]; };
__global__ void __launch_bounds__(256)
dense_split_many_kernel(const DenseSplitItems items)
{
    const DenseSplitItem& it = items.item[blockIdx.y];
    const size_t plane = (size_t)it.rows_pad * it.cols_pad;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)plane; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / it.cols_pad), c = (int)(e % it.cols_pad);
        const int n = it.transposed ? c : r, k = it.transposed ? r : c;
        const float x = (n < it.N && k < it.K) ? it.W[(size_t)n * it.ldw + it.k0 + k] : 0.f;
        uint32_t hi, mid, lo;
        dense_split(x, hi, mid, lo);
        it.planes[e] = (unsigned short)(hi >> 16);
        it.planes[plane + e] = (unsigned short)(mid >> 16);
        it.planes[2 * plane + e] = (unsigned short)(lo >> 16);
    }
}

// ---- Y [M, N] = act(X [M, K] W^T + bias), W given as planes [3][Npad][Kpad] -----------------------------------------------------------------
// X: row stride ldx floats, optional gate: X is read as x * (gate > 0) (the ReLU mask of the layer that produced the cotangent, so that the
// input-gradient product consumes dY and the layer's output directly). relu: max(., 0) on the way out.
constexpr int DENSE_MAX_BT = 10;           // row tiles of 16 per block: 2 .. 10 (the two wave rows take ceil / floor of them, at most five each)
__global__ void __launch_bounds__(DENSE_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))      // two blocks per CU: 256 registers per lane, no spills
dense_fwd_kernel(const int M, const int N, const int K, const float* __restrict__ X, const int ldx, const float* __restrict__ gate, const int ldgate,
                 const unsigned short* __restrict__ planes, const int Npad, const int Kpad, const float* __restrict__ bias, const int relu,
                 float* __restrict__ Y, const int ldy, const int vec, const int vec_out, const int bt,
                 const float* __restrict__ mask, const int ldmask, float* __restrict__ colsum)
{
    // mask [M, N] (optional): the result is written as y * (mask > 0) -- the input-gradient product hands its result straight to the ReLU of
    // the layer below (G_{l-1} = (G_l W_l) [y_{l-1} > 0]); colsum [row blocks][N] (optional, 16-byte output path only): the column sums of
    // what this block wrote, added in a fixed order (that layer's bias gradient after dense_colsum's pass over the row blocks).
    // The block tile is 16 bt rows x 128 columns with bt chosen per launch (dense_row_tiles): with a fixed 128-row tile the node network's
    // batch of 33 280 rows is 520 blocks for the 512 block slots of the chip -- eight blocks run a second round alone and the launch takes
    // two block lifetimes (measured: 41 us; 32 768 rows, exactly 512 blocks: see tools/dev_dense.py). Nine row tiles per block are 464
    // blocks, one round.
    __shared__ __attribute__((aligned(16))) unsigned char s_a[3][16 * DENSE_MAX_BT * DENSE_ROW_B];
    __shared__ __attribute__((aligned(16))) unsigned char s_b[3][DENSE_BN * DENSE_ROW_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows_a = 16 * bt;
    const int m0 = blockIdx.x * rows_a, n0 = blockIdx.y * DENSE_BN;
    const int na = (bt + 1) >> 1;                                   // row tiles of the first wave row; the second takes the rest
    const int nt = (wave & 1) ? bt - na : na;                       // this wave's row tiles (wave-uniform) ...
    const int wm = (wave & 1) ? 16 * na : 0, wn = (wave >> 1) * 64; // ... starting at row wm of the block tile; its 64 columns
    const int fi = lane & 15, fq = lane >> 4;                      // fragment row / column and its group of eight k
    dense_acc acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = dense_acc{0.f, 0.f, 0.f, 0.f};

    // staging assignment: item = (row, group of eight k); 16 bt rows x 4 groups of X (up to three per thread), 128 x 4 of the weight (two per
    // thread). The next step's operands are requested before this step's products are issued. (Requesting X two steps ahead -- a second set of
    // staging registers, the loop unrolled by two -- was built and measured: 256 registers, spills, 41 -> 54 us at 33k rows.)
    const int steps = Kpad / DENSE_BK;
    const int items_a = rows_a * 4;
    float xa[3][8];
    dense_u4 wb[2][3];
    auto fetch = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int item = tid + it * DENSE_THREADS, row = item >> 2, g = item & 3;
            if (item >= items_a) continue;
            const int m = m0 + row, k = s * DENSE_BK + 8 * g;
            const float* src = X + (size_t)m * ldx + k;
            const float* gsrc = gate ? gate + (size_t)m * ldgate + k : nullptr;
            if (vec && m < M && k + 8 <= K) {                        // the usual case: two 16-byte loads (the host checked the alignment)
                const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                xa[it][0] = v0.x; xa[it][1] = v0.y; xa[it][2] = v0.z; xa[it][3] = v0.w;
                xa[it][4] = v1.x; xa[it][5] = v1.y; xa[it][6] = v1.z; xa[it][7] = v1.w;
                if (gate) {
                    const float4 g0 = *reinterpret_cast<const float4*>(gsrc), g1 = *reinterpret_cast<const float4*>(gsrc + 4);
                    xa[it][0] = g0.x > 0.f ? xa[it][0] : 0.f; xa[it][1] = g0.y > 0.f ? xa[it][1] : 0.f;
                    xa[it][2] = g0.z > 0.f ? xa[it][2] : 0.f; xa[it][3] = g0.w > 0.f ? xa[it][3] : 0.f;
                    xa[it][4] = g1.x > 0.f ? xa[it][4] : 0.f; xa[it][5] = g1.y > 0.f ? xa[it][5] : 0.f;
                    xa[it][6] = g1.z > 0.f ? xa[it][6] : 0.f; xa[it][7] = g1.w > 0.f ? xa[it][7] : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const bool ok = m < M && k + e < K;
                    float v = ok ? src[e] : 0.f;
                    if (gate) v = (ok && gsrc[e] > 0.f) ? v : 0.f;
                    xa[it][e] = v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int item = tid + it * DENSE_THREADS, row = item >> 2, g = item & 3;
            const size_t off = (size_t)(n0 + row) * Kpad + s * DENSE_BK + 8 * g;        // planes are padded: always in range
            const size_t plane = (size_t)Npad * Kpad;
#pragma unroll
            for (int p = 0; p < 3; p++) wb[it][p] = *reinterpret_cast<const dense_u4*>(planes + p * plane + off);
        }
    };
    auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int item = tid + it * DENSE_THREADS, row = item >> 2, g = item & 3;
            if (item >= items_a) continue;
            dense_u4 h, m_, l;
            dense_split8(xa[it], h, m_, l);
            const int o = dense_off(row, g);
            *reinterpret_cast<dense_u4*>(&s_a[0][o]) = h;
            *reinterpret_cast<dense_u4*>(&s_a[1][o]) = m_;
            *reinterpret_cast<dense_u4*>(&s_a[2][o]) = l;
        }
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int item = tid + it * DENSE_THREADS, row = item >> 2, g = item & 3;
            const int o = dense_off(row, g);
#pragma unroll
            for (int p = 0; p < 3; p++) *reinterpret_cast<dense_u4*>(&s_b[p][o]) = wb[it][p];
        }
    };
    fetch(0);
    for (int s = 0; s < steps; s++) {
        __syncthreads();                      // the previous step's fragments have been read
        stage();
        __syncthreads();
        if (s + 1 < steps) fetch(s + 1);      // in flight while this step's products run
        dense_frag b[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) b[j][p] = dense_ld_frag(&s_b[p][dense_off(wn + 16 * j + fi, fq)]);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (i < nt) {
                dense_frag a[3];
#pragma unroll
                for (int p = 0; p < 3; p++) a[p] = dense_ld_frag(&s_a[p][dense_off(wm + 16 * i + fi, fq)]);
                dense_mfma6x4(a, b, acc[i]);
            }
        }
    }
    // C layout: column = lane & 15, rows 4 (lane >> 4) + r. A wave's 16 x 64 slab goes through a wave-private LDS tile so that the stores are
    // 16 bytes per lane and 256 contiguous bytes per row (straight from the accumulators a store instruction wrote four 64-byte pieces)
    if (vec_out) {
        __syncthreads();                                          // the last step's fragments have been read: the staging arrays are free
        float* slab = reinterpret_cast<float*>(&s_a[0][0]) + wave * (16 * 68);          // 4 x 4.25 KB inside s_a
        const int col4 = 4 * (lane & 15), nn = n0 + wn + col4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && nn + 3 < N) bv = *reinterpret_cast<const float4*>(bias + nn);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (i < nt) {                                         // (the slab is the wave's own: no block barrier between its store and its reads)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) slab[(4 * fq + r) * 68 + 16 * j + fi] = acc[i][j][r];
                __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's LDS stores have landed
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int row = 4 * t + (lane >> 4), m = m0 + wm + 16 * i + row;
                    float4 v = *reinterpret_cast<const float4*>(&slab[row * 68 + col4]);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (m < M && nn + 3 < N) {
                        if (mask) {
                            const float4 mv = *reinterpret_cast<const float4*>(mask + (size_t)m * ldmask + nn);
                            v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f; v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
                        }
                        *reinterpret_cast<float4*>(Y + (size_t)m * ldy + nn) = v;
                        cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;          // rows in increasing order within the lane
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (colsum) {
            // lanes l, l + 16, l + 32, l + 48 hold rows 4 t + {0, 1, 2, 3} of the same four columns: (0 + 1) + (2 + 3), then the two wave rows
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                cs.x += __shfl_xor(cs.x, off, 64); cs.y += __shfl_xor(cs.y, off, 64); cs.z += __shfl_xor(cs.z, off, 64); cs.w += __shfl_xor(cs.w, off, 64);
            }
            __syncthreads();                                      // every wave is done with its slab
            float4* s_cs = reinterpret_cast<float4*>(&s_b[0][0]);   // [4 waves][16 lanes]
            if (lane < 16) s_cs[wave * 16 + lane] = cs;
            __syncthreads();
            if ((wave & 1) == 0 && lane < 16 && nn + 3 < N) {
                const float4 o = s_cs[(wave + 1) * 16 + lane];
                *reinterpret_cast<float4*>(colsum + (size_t)blockIdx.x * N + nn) = make_float4(cs.x + o.x, cs.y + o.y, cs.z + o.z, cs.w + o.w);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = n0 + wn + 16 * j + fi;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (i >= nt) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + wm + 16 * i + 4 * fq + r;
                if (m < M) {
                    float v = acc[i][j][r] + bv;
                    if (relu) v = fmaxf(v, 0.f);
                    if (mask) v = mask[(size_t)m * ldmask + n] > 0.f ? v : 0.f;
                    Y[(size_t)m * ldy + n] = v;
                }
            }
        }
    }
}

// ---- the same product, one block per CU (round 5, second form) --------------------------------------------------------------------------------
// Ablations of dense_fwd_kernel at 32 768 rows x 256 x 256: 10 us of launch + first fetch + epilogue, +5 us staging (split + LDS stores), +5 us
// global fetches, +4 us fragment reads, +8 us MFMA = 32 us -- the phases ADD UP: two blocks per CU that start together run in lock step (both
// stage, both multiply), a barrier on either side of every phase. dense_fwd8_kernel removes the lock step:
//   * ONE block of eight waves per CU and a 16 BT x 256 tile (the whole width of the network: X is read once, not once per column tile);
//     every wave owns ALL rows of the tile and 32 columns: equal work for every wave at every BT (2 .. 10, a template parameter: the
//     multiply phase is straight-line code whose fragment reads run one row tile ahead of its products);
//   * the LDS holds TWO stages (2 x (30 + 48) KB): step s + 1 is staged while step s is multiplied, one barrier per step;
//   * the two waves of a SIMD (w and w + 4) take the phases in OPPOSITE order -- one multiplies while the other splits and stores;
//   * the products are formed TRANSPOSED (weight fragment as the first MFMA operand): a lane ends up with four consecutive columns of one
//     row -- 16-byte stores straight from the accumulators, no transposition through LDS.
constexpr int DENSE8_THREADS = 512, DENSE8_BN = 256;
constexpr int DENSE8_STAGE_A = 3 * 16 * DENSE_MAX_BT * DENSE_ROW_B, DENSE8_STAGE_B = 3 * DENSE8_BN * DENSE_ROW_B;       // 30 720 + 49 152 bytes
constexpr int DENSE8_LDS_BYTES = 2 * (DENSE8_STAGE_A + DENSE8_STAGE_B);                                                // 159 744 of 163 840
// one product of the kernel below (and of dense_chain8_kernel: several in a row on the same rows)
struct Dense8Layer {
    const float* X; const float* gate; const unsigned short* planes; const float* bias; float* Y; const float* mask; float* colsum;
    int N, K, ldx, ldgate, Npad, Kpad, relu, ldy, vec, ldmask;
};

template <int BT>
__device__ __forceinline__ void dense8_layer(const int M, const Dense8Layer& L, unsigned char* s_dense8)
{
    const int N = L.N, K = L.K, ldx = L.ldx, ldgate = L.ldgate, Npad = L.Npad, Kpad = L.Kpad, relu = L.relu, ldy = L.ldy, vec = L.vec, ldmask = L.ldmask;
    const float* __restrict__ X = L.X; const float* __restrict__ gate = L.gate; const unsigned short* __restrict__ planes = L.planes;
    const float* __restrict__ bias = L.bias; float* __restrict__ Y = L.Y; const float* __restrict__ mask = L.mask; float* __restrict__ colsum = L.colsum;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int rows_a = 16 * BT;
    const int m0 = blockIdx.x * rows_a, n0 = blockIdx.y * DENSE8_BN;
    const int wn = wave * 32;                                       // this wave's 32 columns (two column tiles), all BT row tiles
    const int stage_first = wave >> 2;                              // waves w and w + 4 share a SIMD: opposite phase order
    const int fi = lane & 15, fq = lane >> 4;
    dense_acc acc[BT][2];
#pragma unroll
    for (int i = 0; i < BT; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = dense_acc{0.f, 0.f, 0.f, 0.f};
    auto sa = [&](int buf, int p) __attribute__((always_inline)) { return s_dense8 + buf * (DENSE8_STAGE_A + DENSE8_STAGE_B) + p * (16 * DENSE_MAX_BT * DENSE_ROW_B); };
    auto sb = [&](int buf, int p) __attribute__((always_inline)) { return s_dense8 + buf * (DENSE8_STAGE_A + DENSE8_STAGE_B) + DENSE8_STAGE_A + p * (DENSE8_BN * DENSE_ROW_B); };

    const int steps = Kpad / DENSE_BK;
    constexpr int items_a = rows_a * 4;                             // (row, group of eight k): up to 640, two per thread; the weight: 1024, two per thread
    constexpr int A_ITERS = (items_a + DENSE8_THREADS - 1) / DENSE8_THREADS;
    float xa[A_ITERS][8];
    dense_u4 wb[2][3];
    auto fetch = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < A_ITERS; it++) {
            const int item = tid + it * DENSE8_THREADS, row = item >> 2, g = item & 3;
            if (item >= items_a) continue;
            const int m = m0 + row, k = s * DENSE_BK + 8 * g;
            const float* src = X + (size_t)m * ldx + k;
            const float* gsrc = gate ? gate + (size_t)m * ldgate + k : nullptr;
            if (vec && m < M && k + 8 <= K) {
                const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                xa[it][0] = v0.x; xa[it][1] = v0.y; xa[it][2] = v0.z; xa[it][3] = v0.w;
                xa[it][4] = v1.x; xa[it][5] = v1.y; xa[it][6] = v1.z; xa[it][7] = v1.w;
                if (gate) {
                    const float4 g0 = *reinterpret_cast<const float4*>(gsrc), g1 = *reinterpret_cast<const float4*>(gsrc + 4);
                    xa[it][0] = g0.x > 0.f ? xa[it][0] : 0.f; xa[it][1] = g0.y > 0.f ? xa[it][1] : 0.f;
                    xa[it][2] = g0.z > 0.f ? xa[it][2] : 0.f; xa[it][3] = g0.w > 0.f ? xa[it][3] : 0.f;
                    xa[it][4] = g1.x > 0.f ? xa[it][4] : 0.f; xa[it][5] = g1.y > 0.f ? xa[it][5] : 0.f;
                    xa[it][6] = g1.z > 0.f ? xa[it][6] : 0.f; xa[it][7] = g1.w > 0.f ? xa[it][7] : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const bool ok = m < M && k + e < K;
                    float v = ok ? src[e] : 0.f;
                    if (gate) v = (ok && gsrc[e] > 0.f) ? v : 0.f;
                    xa[it][e] = v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int item = tid + it * DENSE8_THREADS, row = item >> 2, g = item & 3;
            const size_t off = (size_t)(n0 + row) * Kpad + s * DENSE_BK + 8 * g;        // N % 256 == 0 here: always in range
            const size_t plane = (size_t)Npad * Kpad;
#pragma unroll
            for (int p = 0; p < 3; p++) wb[it][p] = *reinterpret_cast<const dense_u4*>(planes + p * plane + off);
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < A_ITERS; it++) {
            const int item = tid + it * DENSE8_THREADS, row = item >> 2, g = item & 3;
            if (item >= items_a) continue;
            dense_u4 h, m_, l;
            dense_split8(xa[it], h, m_, l);
            const int o = dense_off(row, g);
            *reinterpret_cast<dense_u4*>(sa(buf, 0) + o) = h;
            *reinterpret_cast<dense_u4*>(sa(buf, 1) + o) = m_;
            *reinterpret_cast<dense_u4*>(sa(buf, 2) + o) = l;
        }
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int item = tid + it * DENSE8_THREADS, row = item >> 2, g = item & 3;
            const int o = dense_off(row, g);
#pragma unroll
            for (int p = 0; p < 3; p++) *reinterpret_cast<dense_u4*>(sb(buf, p) + o) = wb[it][p];
        }
    };
    // the six products of TWO row tiles with the wave's two column tiles, the weight fragment FIRST (the accumulator is the transposed tile:
    // lane (fi, fq) holds columns 4 fq .. 4 fq + 3 of row fi). Four independent accumulators per term (consecutive MFMAs never share one);
    // the next pair's fragments are read while this pair's products run
    auto multiply = [&](int buf) __attribute__((always_inline)) {
        dense_frag b[2][3], a[2][2][3];                         // a[parity of the pair][tile within the pair][plane]
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) b[j][p] = dense_ld_frag(sb(buf, p) + dense_off(wn + 16 * j + fi, fq));
#pragma unroll
        for (int t = 0; t < 2; t++)
            if (t < BT) {
#pragma unroll
                for (int p = 0; p < 3; p++) a[0][t][p] = dense_ld_frag(sa(buf, p) + dense_off(16 * t + fi, fq));
            }
#pragma unroll
        for (int i = 0; i < BT; i += 2) {
            const int par = (i >> 1) & 1;
#pragma unroll
            for (int t = 0; t < 2; t++)
                if (i + 2 + t < BT) {
#pragma unroll
                    for (int p = 0; p < 3; p++) a[par ^ 1][t][p] = dense_ld_frag(sa(buf, p) + dense_off(16 * (i + 2 + t) + fi, fq));
                }
#define GSR_DENSE8_TERM(PA, PB)                                                                                                            \
            _Pragma("unroll") for (int t = 0; t < 2; t++)                                                                                  \
                if (i + t < BT) {                                                                                                          \
                    _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                          \
                        acc[i + t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][PB], a[par][t][PA], acc[i + t][j], 0, 0, 0);           \
                }
            GSR_DENSE8_TERM(2, 0) GSR_DENSE8_TERM(0, 2) GSR_DENSE8_TERM(1, 1) GSR_DENSE8_TERM(1, 0) GSR_DENSE8_TERM(0, 1) GSR_DENSE8_TERM(0, 0)
#undef GSR_DENSE8_TERM
        }
    };
    fetch(0);
    stage(0);
    if (steps > 1) fetch(1);
    __syncthreads();
    for (int s = 0; s < steps; s++) {
        const int cur = s & 1;
        if (stage_first) {
            if (s + 1 < steps) { stage(cur ^ 1); if (s + 2 < steps) fetch(s + 2); }
            multiply(cur);
        } else {
            multiply(cur);
            if (s + 1 < steps) { stage(cur ^ 1); if (s + 2 < steps) fetch(s + 2); }
        }
        __syncthreads();                       // stage cur ^ 1 is complete, stage cur has been read
    }
    // epilogue: bias, ReLU, mask; 16-byte stores from the accumulators; the column sums of what was written (rows in increasing order per
    // lane, then over the 16 lanes of a column group by xor shuffles: a fixed order)
    float4 cs[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int nn = n0 + wn + 16 * j + 4 * fq;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + nn);
        cs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < BT; i++) {
            const int m = m0 + 16 * i + fi;
            float4 v = make_float4(acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (m < M) {
                if (mask) {
                    const float4 mv = *reinterpret_cast<const float4*>(mask + (size_t)m * ldmask + nn);
                    v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f; v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4*>(Y + (size_t)m * ldy + nn) = v;
                cs[j].x += v.x; cs[j].y += v.y; cs[j].z += v.z; cs[j].w += v.w;
            }
        }
    }
    if (colsum) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                cs[j].x += __shfl_xor(cs[j].x, off, 64); cs[j].y += __shfl_xor(cs[j].y, off, 64);
                cs[j].z += __shfl_xor(cs[j].z, off, 64); cs[j].w += __shfl_xor(cs[j].w, off, 64);
            }
            if (fi == 0) *reinterpret_cast<float4*>(colsum + (size_t)blockIdx.x * N + n0 + wn + 16 * j + 4 * fq) = cs[j];
        }
    }
}

template <int BT>
__global__ void __launch_bounds__(DENSE8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
dense_fwd8_kernel(const int M, const Dense8Layer L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dense8_single[];
    dense8_layer<BT>(M, L, s_dense8_single);
}

// Several products in a row on the SAME rows in one launch: product l + 1 reads what product l wrote (a layer's output is the next layer's
// input; an input gradient is the next input-gradient product's cotangent). A block owns its 16 BT rows through all of them -- rows are
// independent, so nothing is exchanged between blocks; its own stores become visible to its own loads through the block barrier between two
// products (the addresses were never read before in this launch: no stale line in the CU's cache). Against one launch per product this
// saves, per boundary, the launch gap, the wait for EVERY block's output stores before any block may start, and most of the first fetch:
// the node network's forward is seven boundaries, its input-gradient chain six.
constexpr int DENSE8_CHAIN_MAX = 8;
struct Dense8Chain { int M, count; Dense8Layer layer[DENSE8_CHAIN_MAX]; };
template <int BT>
__global__ void __launch_bounds__(DENSE8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
dense_chain8_kernel(const Dense8Chain c)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dense8_chain[];
    for (int l = 0; l < c.count; l++) {
        dense8_layer<BT>(c.M, c.layer[l], s_dense8_chain);
        // this block's stores of product l are visible to ITS OWN waves after the block barrier (the programming model's guarantee; the CU's
        // cache is write-through) before any of them fetches them as product l + 1's operand -- and the LDS stages are free. (A device-scope
        // fence here is an L2 write-back per block and boundary on this multi-XCD part: measured 86 us per product instead of 36.)
        if (l + 1 < c.count) __syncthreads();
    }
}

// row tiles of 16 per block for dense_fwd8_kernel: one block per CU, 256 slots per round
inline int dense8_row_tiles(int M, int col_tiles)
{
    const long units = ((long)M + 15) / 16;
    int best = 8; long best_cost = -1;
    for (int bt = 2; bt <= DENSE_MAX_BT; bt++) {
        const long blocks = ((units + bt - 1) / bt) * col_tiles, rounds = (blocks + 255) / 256;
        const long cost = rounds * (bt + 3);
        if (best_cost < 0 || cost < best_cost) { best = bt; best_cost = cost; }
    }
    return best;
}

// row tiles of 16 per block for a forward launch: the count that needs the fewest rounds of the chip's 512 block slots (two blocks per CU),
// each round weighed by the tile's rows plus a fixed share (weight staging, barriers, the epilogue)
inline int dense_row_tiles(int M, int col_tiles)
{
    const long units = ((long)M + 15) / 16;
    int best = 8; long best_cost = -1;
    for (int bt = 2; bt <= DENSE_MAX_BT; bt++) {
        const long blocks = ((units + bt - 1) / bt) * col_tiles, rounds = (blocks + 511) / 512;
        const long cost = rounds * (bt + 3);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && bt == 8)) { best = bt; best_cost = cost; }
    }
    return best;
}

// ---- dW [N, K] = G^T X over the rows [r0, r1) of this block's slice ---------------------------------------------------------------------------
// G [M, N] (row stride ldg; optional gate as above: G = g * (gate > 0)), X [M, K] (ldx). grid = (N tiles of 128, K tiles of 128, slices);
// partial [slices][N][K] fp32; dense_wgrad_sum_kernel adds the slices in order. The MFMA's reduction index is the ROW: both operands are
// staged transposed -- a thread reads sixteen rows of one column (coalesced across the wave: consecutive columns) and writes them as two
// 16-byte fragments per plane.
constexpr int DENSE_WG_ROWS = 32;       // rows per step = one MFMA's depth
__global__ void __launch_bounds__(DENSE_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
dense_wgrad_kernel(const int M, const int N, const int K, const float* __restrict__ G, const int ldg, const float* __restrict__ gate, const int ldgate,
                   const float* __restrict__ X, const int ldx, const int rows_per_slice, float* __restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_g[3][DENSE_BM * DENSE_ROW_B];   // [plane][n][32 rows]
    __shared__ __attribute__((aligned(16))) unsigned char s_x[3][DENSE_BN * DENSE_ROW_B];   // [plane][k][32 rows]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * DENSE_BM, k0 = blockIdx.y * DENSE_BN;
    const int r_begin = blockIdx.z * rows_per_slice, r_end = min(M, r_begin + rows_per_slice);
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    const int fi = lane & 15, fq = lane >> 4;
    dense_acc acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = dense_acc{0.f, 0.f, 0.f, 0.f};
    // staging: thread = (column c = tid & 127, half h = tid >> 7): rows 16 h .. 16 h + 15 of the step, of G's column n0 + c and X's column k0 + c
    const int c = tid & 127, h = tid >> 7;
    float gv[16], xv[16];
    auto fetch = [&](int r0) {
        const int n = n0 + c, k = k0 + c;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int r = r0 + 16 * h + e;
            const bool okr = r < r_end;
            float g = (okr && n < N) ? G[(size_t)r * ldg + n] : 0.f;
            if (gate) g = (okr && n < N && gate[(size_t)r * ldgate + n] > 0.f) ? g : 0.f;
            gv[e] = g;
            xv[e] = (okr && k < K) ? X[(size_t)r * ldx + k] : 0.f;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            float a8[8], b8[8];
#pragma unroll
            for (int e = 0; e < 8; e++) { a8[e] = gv[8 * half + e]; b8[e] = xv[8 * half + e]; }
            dense_u4 hh, mm, ll;
            const int o = dense_off(c, 2 * h + half);                 // rows 16 h + 8 half .. + 7 of this column
            dense_split8(a8, hh, mm, ll);
            *reinterpret_cast<dense_u4*>(&s_g[0][o]) = hh; *reinterpret_cast<dense_u4*>(&s_g[1][o]) = mm; *reinterpret_cast<dense_u4*>(&s_g[2][o]) = ll;
            dense_split8(b8, hh, mm, ll);
            *reinterpret_cast<dense_u4*>(&s_x[0][o]) = hh; *reinterpret_cast<dense_u4*>(&s_x[1][o]) = mm; *reinterpret_cast<dense_u4*>(&s_x[2][o]) = ll;
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += DENSE_WG_ROWS) {
        __syncthreads();
        stage();
        __syncthreads();
        if (r0 + DENSE_WG_ROWS < r_end) fetch(r0 + DENSE_WG_ROWS);
        dense_frag b[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) b[j][p] = dense_ld_frag(&s_x[p][dense_off(wn + 16 * j + fi, fq)]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            dense_frag a[3];
#pragma unroll
            for (int p = 0; p < 3; p++) a[p] = dense_ld_frag(&s_g[p][dense_off(wm + 16 * i + fi, fq)]);
            dense_mfma6x4(a, b, acc[i]);
        }
    }
    float* out = partial + (size_t)blockIdx.z * N * K;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = k0 + wn + 16 * j + fi;
        if (k >= K) continue;
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int n = n0 + wm + 16 * i + 4 * fq + r;
                if (n < N) out[(size_t)n * K + k] = acc[i][j][r];
            }
        }
    }
}

__global__ void __launch_bounds__(256)
dense_wgrad_sum_kernel(const int slices, const int count, const float* __restrict__ partial, const int K, float* __restrict__ dW, const int lddw)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < slices; b += 4) {
        s0 += partial[(size_t)b * count + e];
        s1 += partial[(size_t)(b + 1) * count + e];
        s2 += partial[(size_t)(b + 2) * count + e];
        s3 += partial[(size_t)(b + 3) * count + e];
    }
    for (; b < slices; b++) s0 += partial[(size_t)b * count + e];
    dW[(size_t)(e / K) * lddw + e % K] = (s0 + s1) + (s2 + s3);
}

// ---- round 6: SEVERAL weight gradients in one launch (gsr_dense_wgrad_many) ----------------------------------------------------------------------
// The node network's backward leaves every layer's G and input in memory (the chained input-gradient launch), so the D + 1 products
// dW_i = G_i^T X_i over the same M rows are independent: ONE launch of (all 128 x 128 result tiles) x (row slices), sized so that the chip holds
// every block at once (two per CU), instead of one launch per layer whose 256 x 256 result gives the chip four tiles to spread over 256 CUs.
// What the single-product kernel above spent besides its matrix instructions is cut here:
//  * operands come in as float4 (a thread: four rows x four consecutive columns of G and of X per 32-row step -- 8 loads of 16 bytes instead of
//    32 of 4), are split once per block and step, and go to LDS TRANSPOSED as 8-byte pieces (four k of one column, one ds_write_b64 per plane).
//    Thread <-> (rows, columns) and the order in which a thread writes its four columns are chosen so that the sixteen lanes the LDS serves
//    together always write sixteen different 8-byte slots of a 128-byte bank row: lane bits {0} = row group, {1} = column order flipped (column
//    parity), {2, 3} = column group mod 4 (which selects the chunk swizzle g below) -- conflict-free stores AND conflict-free ds_read_b128
//    fragment reads (the read pattern is mlp3_off<4>'s: g = 0, 3, 2, 1 for columns 0-3, 4-7, 8-11, 12-15 of a 16-column tile);
//  * the slices' partial tiles are stored tile-local ([slice][tile][128][128]) and summed by float4 in a fixed order.
// Items whose rows are not 16-byte aligned (the heads' [M, 14] cotangent) take scalar loads.
constexpr int DENSE_WGM_MAX = 12;
struct DenseWgradItem { const float* G; const float* X; float* dW; int ldg, ldx, lddw, N, K, tiles_k, tile0, vec; };   // tile0: first tile of the item; vec: float4 loads allowed
struct DenseWgradItems { DenseWgradItem item[DENSE_WGM_MAX]; int count, total_tiles; };

// byte offset of the 8-byte piece s (k = 4 s .. 4 s + 3) of column c in a [128][32 k] bf16 plane
__device__ __forceinline__ int dense_wgm_off(int c, int s) { return c * DENSE_ROW_B + 16 * ((s >> 1) ^ ((0 - (c >> 2)) & 3)) + 8 * (s & 1); }

template <bool VEC>
__device__ __forceinline__ void dense_wgrad_many_body(const int M, const DenseWgradItems& items, const int it, const int rows_per_slice, float* __restrict__ partial,
                                                      unsigned char (&s_g)[3][DENSE_BM * DENSE_ROW_B], unsigned char (&s_x)[3][DENSE_BN * DENSE_ROW_B])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const DenseWgradItem& q = items.item[it];
    const int tl = (int)blockIdx.x - q.tile0, n0 = (tl / q.tiles_k) * DENSE_BM, k0 = (tl % q.tiles_k) * DENSE_BN;
    const int N = q.N, K = q.K, ldg = q.ldg, ldx = q.ldx;
    const float* __restrict__ G = q.G;
    const float* __restrict__ X = q.X;
    const int r_begin = blockIdx.y * rows_per_slice, r_end = min(M, r_begin + rows_per_slice);
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    const int fi = lane & 15, fq = lane >> 4;
    dense_acc acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = dense_acc{0.f, 0.f, 0.f, 0.f};
    // staging: thread = (row group rg: rows 4 rg .. 4 rg + 3 of the step, column group cg: columns 4 cg .. 4 cg + 3 of the tile)
    const int rg = (lane & 1) | (wave << 1);
    const int cg = ((lane >> 2) & 3) | (((lane >> 1) & 1) << 2) | ((lane >> 4) << 3);
    const bool flip = ((lane >> 1) & 1) != 0;
    const int st_base = dense_wgm_off(4 * cg, rg);                 // + 64 per column (the swizzle depends on cg only)
    float4 gv[4], xv[4];
    auto load4 = [&](const float* __restrict__ P, int ld, int r, int c, int C) -> float4 {
        if (r >= r_end || c >= C) return make_float4(0.f, 0.f, 0.f, 0.f);
        const float* p = P + (size_t)r * ld + c;
        if (VEC) return *reinterpret_cast<const float4*>(p);      // (C % 4 == 0, 16-byte aligned rows)
        return make_float4(p[0], c + 1 < C ? p[1] : 0.f, c + 2 < C ? p[2] : 0.f, c + 3 < C ? p[3] : 0.f);
    };
    auto fetch = [&](int r0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            gv[i] = load4(G, ldg, r0 + 4 * rg + i, n0 + 4 * cg, N);
            xv[i] = load4(X, ldx, r0 + 4 * rg + i, k0 + 4 * cg, K);
        }
    };
    auto stage_one = [&](const float4 (&v)[4], unsigned char (&dst)[3][DENSE_BM * DENSE_ROW_B]) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // column 4 cg + (j ^ flip): the four rows of that column
            float x[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float a = j == 0 ? v[i].x : j == 1 ? v[i].y : j == 2 ? v[i].z : v[i].w;
                const float b = j == 0 ? v[i].y : j == 1 ? v[i].x : j == 2 ? v[i].w : v[i].z;
                x[i] = flip ? b : a;
            }
            uint2 h, m, l;
            {
                uint32_t a[4], b[4], c[4];
#pragma unroll
                for (int e = 0; e < 4; e++) dense_split(x[e], a[e], b[e], c[e]);
                h = make_uint2(dense_pack(a[0], a[1]), dense_pack(a[2], a[3]));
                m = make_uint2(dense_pack(b[0], b[1]), dense_pack(b[2], b[3]));
                l = make_uint2(dense_pack(c[0], c[1]), dense_pack(c[2], c[3]));
            }
            const int o = st_base + DENSE_ROW_B * (flip ? (j ^ 1) : j);
            *reinterpret_cast<uint2*>(&dst[0][o]) = h; *reinterpret_cast<uint2*>(&dst[1][o]) = m; *reinterpret_cast<uint2*>(&dst[2][o]) = l;
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += DENSE_WG_ROWS) {
        __syncthreads();
        stage_one(gv, s_g);
        stage_one(xv, s_x);
        __syncthreads();
        if (r0 + DENSE_WG_ROWS < r_end) fetch(r0 + DENSE_WG_ROWS);
        const int rsw = 16 * (fq ^ ((0 - (fi >> 2)) & 3));        // = dense_wgm_off(16 t + fi, 2 fq) - (16 t + fi) * 64
        dense_frag b[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) b[j][p] = dense_ld_frag(&s_x[p][(wn + 16 * j + fi) * DENSE_ROW_B + rsw]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            dense_frag a[3];
#pragma unroll
            for (int p = 0; p < 3; p++) a[p] = dense_ld_frag(&s_g[p][(wm + 16 * i + fi) * DENSE_ROW_B + rsw]);
            dense_mfma6x4(a, b, acc[i]);
        }
    }
    // the block's partial tile, tile-local [128 n][128 k]
    float* out = partial + ((size_t)blockIdx.y * items.total_tiles + blockIdx.x) * (DENSE_BM * DENSE_BN);
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int r = 0; r < 4; r++) out[(wm + 16 * i + 4 * fq + r) * DENSE_BN + wn + 16 * j + fi] = acc[i][j][r];
        }
    }
}

__global__ void __launch_bounds__(DENSE_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
dense_wgrad_many_kernel(const int M, const DenseWgradItems items, const int rows_per_slice, float* __restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_g[3][DENSE_BM * DENSE_ROW_B];   // [plane][n][32 rows]
    __shared__ __attribute__((aligned(16))) unsigned char s_x[3][DENSE_BN * DENSE_ROW_B];   // [plane][k][32 rows]
    int it = 0;
#pragma unroll 1
    for (int i = 1; i < items.count; i++) if ((int)blockIdx.x >= items.item[i].tile0) it = i;
    // two copies of the body: with the flag tested per load the compiler merged both forms into a dword + dwordx3 pair
    if (items.item[it].vec) dense_wgrad_many_body<true>(M, items, it, rows_per_slice, partial, s_g, s_x);
    else dense_wgrad_many_body<false>(M, items, it, rows_per_slice, partial, s_g, s_x);
}

// dW tiles = the slices' partial tiles added in slice order; grid = (tiles, 16): a block sums 8 rows of 128 of a tile per round, float4 per thread
__global__ void __launch_bounds__(256)
dense_wgrad_many_sum_kernel(const DenseWgradItems items, const int slices, const float* __restrict__ partial)
{
    int it = 0;
#pragma unroll 1
    for (int i = 1; i < items.count; i++) if ((int)blockIdx.x >= items.item[i].tile0) it = i;
    const DenseWgradItem& q = items.item[it];
    const int tl = (int)blockIdx.x - q.tile0, n0 = (tl / q.tiles_k) * DENSE_BM, k0 = (tl % q.tiles_k) * DENSE_BN;
    const size_t tile_stride = (size_t)items.total_tiles * (DENSE_BM * DENSE_BN);
    const float* src = partial + (size_t)blockIdx.x * (DENSE_BM * DENSE_BN);
    const int e4 = blockIdx.y * 256 + threadIdx.x;                // float4 index inside the tile: 4096 per tile, 16 blocks of 256
    const int row = e4 >> 5, col = (e4 & 31) * 4;
    const float4* p = reinterpret_cast<const float4*>(src) + e4;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    int b = 0;
    for (; b + 3 < slices; b += 4) {
        const float4 v0 = p[(size_t)b * (tile_stride / 4)], v1 = p[(size_t)(b + 1) * (tile_stride / 4)];
        const float4 v2 = p[(size_t)(b + 2) * (tile_stride / 4)], v3 = p[(size_t)(b + 3) * (tile_stride / 4)];
        add(s0, v0); add(s1, v1); add(s2, v2); add(s3, v3);
    }
    for (; b < slices; b++) add(s0, p[(size_t)b * (tile_stride / 4)]);
    const float o[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
    const int n = n0 + row;
    if (n >= q.N) return;
#pragma unroll
    for (int e = 0; e < 4; e++) if (k0 + col + e < q.K) q.dW[(size_t)n * q.lddw + k0 + col + e] = o[e];
}

// ---- the node network's trunk, forward, layer-fused -------------------------------------------------------------------------------------------
// utils/time_utils.py:428-452 with the shipped structure: eight layers y = relu(x W^T + b) of width 256, the embedding (width E <= 96)
// re-injected behind layer 4 (layer 5 reads [emb | h]), then all heads as one linear layer. One block carries a 64-row tile through ALL
// layers: the tile's activations stay in LDS as three bf16 planes (96 KB, XOR-swizzled rows), the embedding's planes beside them (42 KB);
// a wave owns 64 output columns and fetches its weight fragments straight from L2 in MFMA layout (the planes of gsr_dense_split, 3.1 MB for
// the whole network: no LDS, no barrier on the weight side, one step of prefetch in registers); after a layer's products every wave turns its
// 64 x 64 accumulators into bias + ReLU'd fp32 rows through a private LDS slab (16-byte stores of the layer's output, which the backward pass
// reads) and into the next layer's bf16 planes. Per layer: two block barriers. No operand comes from HBM inside a K loop -- what the per-layer
// kernels above wait for at this batch size.
constexpr int TR_BM = 64, TR_W = 256, TR_EPAD = 96, TR_LAYERS = 8, TR_SKIP = 4;
constexpr int TR_ACT_ROW_B = 512, TR_ACT_PLANE_B = TR_BM * TR_ACT_ROW_B;          // 32 KB per plane
constexpr int TR_EMB_ROW_B = 224, TR_EMB_PLANE_B = TR_BM * TR_EMB_ROW_B;          // 12 chunks + 2 of padding: conflict-free fragment reads
constexpr int TR_SLAB_FLOATS = 16 * 68;
constexpr int TR_LDS_BYTES = 3 * TR_ACT_PLANE_B + 3 * TR_EMB_PLANE_B + 4 * TR_SLAB_FLOATS * 4;

struct TrunkArgs {
    int R, E, NH;                                   // rows, embedding width, head outputs (<= 16)
    const float* emb;                               // [R, E]
    const unsigned short* planes[10];               // L0 (K = 96), L1 .. L4, L5 embedding columns (K = 96), L5 trunk columns, L6, L7, heads
    const float* bias[9];                           // eight layers, heads
    float* outs[TR_LAYERS]; int ldo[TR_LAYERS];     // the layers' outputs (post-ReLU), row strides in floats
    float* heads;                                   // [R, NH]
};

__device__ __forceinline__ int tr_act_off(int row, int chunk) { return row * TR_ACT_ROW_B + 16 * (chunk ^ (row & 15)); }

// acc += A(src planes in LDS) x B(planes in L2)^T over `steps` K steps; EMB selects the embedding's planes as the A operand
template <bool EMB>
__device__ __forceinline__ void trunk_products(dense_acc (&acc)[4][4], const unsigned char* __restrict__ src, const unsigned short* __restrict__ planes,
                                               const int Kpad, const int steps, const int wn, const int fi, const int fq)
{
    const size_t plane = (size_t)TR_W * Kpad;
    dense_frag b[4][3], bn[4][3];
    auto load_b = [&](int s, dense_frag (&dst)[4][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) dst[j][p] = *reinterpret_cast<const dense_frag*>(planes + p * plane + (size_t)(wn + 16 * j + fi) * Kpad + 32 * s + 8 * fq);
    };
    load_b(0, bn);
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) b[j][p] = bn[j][p];
        if (s + 1 < steps) load_b(s + 1, bn);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            dense_frag a[3];
            const int row = 16 * i + fi;
#pragma unroll
            for (int p = 0; p < 3; p++)
                a[p] = EMB ? dense_ld_frag(src + p * TR_EMB_PLANE_B + row * TR_EMB_ROW_B + 16 * (4 * s + fq))
                           : dense_ld_frag(src + p * TR_ACT_PLANE_B + tr_act_off(row, 4 * s + fq));
            dense_mfma6x4(a, b, acc[i]);
        }
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
trunk_fwd_kernel(const TrunkArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tr_lds[];
    unsigned char* const act = tr_lds;
    unsigned char* const embp = tr_lds + 3 * TR_ACT_PLANE_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* const slab = reinterpret_cast<float*>(tr_lds + 3 * TR_ACT_PLANE_B + 3 * TR_EMB_PLANE_B) + wave * TR_SLAB_FLOATS;
    const int m0 = blockIdx.x * TR_BM, wn = 64 * wave, fi = lane & 15, fq = lane >> 4;

    // ---- the tile's embedding rows -> bf16 planes (12 chunks of eight k per row; zero beyond E and beyond R) ----
    for (int item = tid; item < TR_BM * (TR_EPAD / 8); item += 256) {
        const int row = item / (TR_EPAD / 8), ch = item % (TR_EPAD / 8), m = m0 + row;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = (m < a.R && 8 * ch + e < a.E) ? a.emb[(size_t)m * a.E + 8 * ch + e] : 0.f;
        dense_u4 h, md, l;
        dense_split8(x, h, md, l);
        const int o = row * TR_EMB_ROW_B + 16 * ch;
        *reinterpret_cast<dense_u4*>(embp + o) = h;
        *reinterpret_cast<dense_u4*>(embp + TR_EMB_PLANE_B + o) = md;
        *reinterpret_cast<dense_u4*>(embp + 2 * TR_EMB_PLANE_B + o) = l;
    }
    __syncthreads();

    for (int layer = 0; layer < TR_LAYERS; layer++) {
        dense_acc acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = dense_acc{0.f, 0.f, 0.f, 0.f};
        if (layer == 0) {
            trunk_products<true>(acc, embp, a.planes[0], TR_EPAD, TR_EPAD / 32, wn, fi, fq);
        } else if (layer == TR_SKIP + 1) {                        // [emb | h]: the embedding columns of the weight, then the trunk columns
            trunk_products<true>(acc, embp, a.planes[5], TR_EPAD, TR_EPAD / 32, wn, fi, fq);
            trunk_products<false>(acc, act, a.planes[6], TR_W, TR_W / 32, wn, fi, fq);
        } else {
            trunk_products<false>(acc, act, a.planes[layer <= TR_SKIP ? layer : layer + 1], TR_W, TR_W / 32, wn, fi, fq);
        }
        __syncthreads();                                          // every wave has read the activations this layer consumed
        // ---- bias + ReLU; the layer's output to memory (fp32) and to the planes of the next layer's input ----
        const int col4 = 4 * (lane & 15);
        const float4 bv = *reinterpret_cast<const float4*>(a.bias[layer] + wn + col4);
        float* const out = a.outs[layer];
        const int ldo = a.ldo[layer];
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) slab[(4 * fq + r) * 68 + 16 * j + fi] = acc[i][j][r];
            __builtin_amdgcn_wave_barrier();                      // (the slab is this wave's own: LDS accesses of one wave are served in order)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int row = 16 * i + 4 * t + (lane >> 4), m = m0 + row;
                float4 v = *reinterpret_cast<const float4*>(&slab[(4 * t + (lane >> 4)) * 68 + col4]);
                v.x = fmaxf(v.x + bv.x, 0.f); v.y = fmaxf(v.y + bv.y, 0.f); v.z = fmaxf(v.z + bv.z, 0.f); v.w = fmaxf(v.w + bv.w, 0.f);
                if (m < a.R) *reinterpret_cast<float4*>(out + (size_t)m * ldo + wn + col4) = v;
                uint32_t h0, m_0, l0, h1, m_1, l1, h2, m_2, l2, h3, m_3, l3;
                dense_split(v.x, h0, m_0, l0); dense_split(v.y, h1, m_1, l1); dense_split(v.z, h2, m_2, l2); dense_split(v.w, h3, m_3, l3);
                const int o = tr_act_off(row, (wn + col4) >> 3) + 2 * (col4 & 7);          // four bf16 = 8 bytes inside the chunk of eight
                *reinterpret_cast<uint2*>(act + o) = make_uint2(dense_pack(h0, h1), dense_pack(h2, h3));
                *reinterpret_cast<uint2*>(act + TR_ACT_PLANE_B + o) = make_uint2(dense_pack(m_0, m_1), dense_pack(m_2, m_3));
                *reinterpret_cast<uint2*>(act + 2 * TR_ACT_PLANE_B + o) = make_uint2(dense_pack(l0, l1), dense_pack(l2, l3));
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                                          // the next layer's input is complete
    }

    // ---- the heads: one 16-column tile; wave w takes rows 16 w .. 16 w + 15 ----
    {
        dense_acc acc = dense_acc{0.f, 0.f, 0.f, 0.f};
        const unsigned short* planes = a.planes[9];
        const size_t plane = (size_t)DENSE_BN * TR_W;             // the heads' planes are padded to 128 rows
        for (int s = 0; s < TR_W / 32; s++) {
            dense_frag af[3], bf[3];
            const int row = 16 * wave + fi;
#pragma unroll
            for (int p = 0; p < 3; p++) {
                af[p] = dense_ld_frag(act + p * TR_ACT_PLANE_B + tr_act_off(row, 4 * s + fq));
                bf[p] = *reinterpret_cast<const dense_frag*>(planes + p * plane + (size_t)fi * TR_W + 32 * s + 8 * fq);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[0], acc, 0, 0, 0);
        }
        if (fi < a.NH) {
            const float bh = a.bias[8][fi];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + 16 * wave + 4 * fq + r;
                if (m < a.R) a.heads[(size_t)m * a.NH + fi] = acc[r] + bh;
            }
        }
    }
}

}  // namespace gsr
