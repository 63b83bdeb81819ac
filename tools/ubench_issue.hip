// ubench_issue.hip -- what does one wave64 instruction cost a gfx950 SIMD?  (tools/, not product)
//
// Settles the question VERDICT r01 raised about render_fwd / render_bwd: DESIGN.md modelled "every instruction = one 4-cycle
// issue slot per SIMD", MI355X_MICROARCH.md says a wave64 VALU op occupies the SIMD-32 for 2 cycles. This measures it:
// for each instruction class, W waves per SIMD (1, 2, 4, 8) each run a loop of independent (or dependent) instructions and
// time themselves with s_memtime; reported = shader cycles per instruction per SIMD (wall cycles of the slowest wave /
// (instructions per wave x waves per SIMD)). One workgroup per CU (big LDS footprint), 256 workgroups.
//
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o gpurun_out/ubench_issue && gpurun_out/ubench_issue > gpurun_out/ubench_issue.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum Op { FMA_IND = 0, FMA_DEP, PK_FMA_IND, EXP_IND, DPP_ADD_IND, DPP_ADD_DEP, SWAP32_IND, CNDMASK_IND, SALU_IND, LDS_READ128, LDS_READ64,
          MIX_VALU_SALU, MIX_VALU_LDS, MIX_FMA_EXP, FMA_IND_2SRC_SGPR, OP_COUNT };
static const char* kNames[OP_COUNT] = {"v_fma_f32 independent x8", "v_fma_f32 dependent chain", "v_pk_fma_f32 independent x8",
                                       "v_exp_f32 independent x8", "v_add_f32_dpp quad_perm independent x8", "v_add_f32_dpp row_ror dependent chain",
                                       "v_permlane32_swap independent x4", "v_cndmask_b32 (sgpr mask) independent x8",
                                       "s_add_u32 independent x8", "ds_read_b128 x8 + waitcnt", "ds_read_b64 x8 + waitcnt",
                                       "4 v_fma + 4 s_add interleaved", "6 v_fma + 2 ds_read_b128", "6 v_fma + 2 v_exp",
                                       "v_fma_f32 with sgpr operand x8"};
static const int kInstrPerIter[OP_COUNT] = {8, 8, 8, 8, 8, 8, 4, 8, 8, 8, 8, 8, 8, 8, 8};

template <int OP>
__global__ void __launch_bounds__(1024) bench_kernel(int iters, unsigned long long* cycles, float* sink)
{
    extern __shared__ float4 lds[];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0, p5 = p1, p6 = p2, p7 = p3;
    const float x = 1.0000001f, y = 1e-9f;
    const f2 px = {x, x}, py = {y, y};
    unsigned long long m0 = 0xAAAAAAAAAAAAAAAAull;
    asm volatile("" : "+s"(m0));
    uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
    float sx = x;
    asm volatile("" : "+s"(sx));
    const uint32_t laddr = (threadIdx.x & 63) * 16;
    float4 r0, r1, r2, r3, r4, r5, r6, r7;
    r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = make_float4(0, 0, 0, 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (OP == FMA_IND) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (OP == FMA_IND_2SRC_SGPR) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sx), "v"(y));
        } else if (OP == FMA_DEP) {
            asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                         "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                         : "+v"(a0) : "v"(x), "v"(y));
        } else if (OP == PK_FMA_IND) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                         "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(px), "v"(py));
        } else if (OP == EXP_IND) {
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                         "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == DPP_ADD_IND) {
            asm volatile("v_add_f32_dpp %0, %8, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %1, %8, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %2, %8, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %3, %8, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %4, %8, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %5, %8, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %6, %8, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %7, %8, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
        } else if (OP == DPP_ADD_DEP) {
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                         "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0));
        } else if (OP == SWAP32_IND) {
            asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == CNDMASK_IND) {
            asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n\tv_cndmask_b32_e64 %1, %1, %8, %9\n\tv_cndmask_b32_e64 %2, %2, %8, %9\n\t"
                         "v_cndmask_b32_e64 %3, %3, %8, %9\n\tv_cndmask_b32_e64 %4, %4, %8, %9\n\tv_cndmask_b32_e64 %5, %5, %8, %9\n\t"
                         "v_cndmask_b32_e64 %6, %6, %8, %9\n\tv_cndmask_b32_e64 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "s"(m0));
        } else if (OP == SALU_IND) {
            asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3\n\t"
                         "s_add_u32 %4, %4, 3\n\ts_add_u32 %5, %5, 3\n\ts_add_u32 %6, %6, 3\n\ts_add_u32 %7, %7, 3"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
        } else if (OP == LDS_READ128) {
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                         "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(laddr) : "memory");
        } else if (OP == LDS_READ64) {
            typedef float f2b __attribute__((ext_vector_type(2)));
            f2b q0, q1, q2, q3, q4, q5, q6, q7;
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:1024\n\tds_read_b64 %2, %8 offset:2048\n\tds_read_b64 %3, %8 offset:3072\n\t"
                         "ds_read_b64 %4, %8 offset:4096\n\tds_read_b64 %5, %8 offset:5120\n\tds_read_b64 %6, %8 offset:6144\n\tds_read_b64 %7, %8 offset:7168\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(laddr) : "memory");
            a0 += q0.x + q1.x + q2.x + q3.x + q4.x + q5.x + q6.x + q7.x;
        } else if (OP == MIX_VALU_SALU) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\ts_add_u32 %4, %4, 3\n\tv_fma_f32 %1, %1, %8, %9\n\ts_add_u32 %5, %5, 3\n\t"
                         "v_fma_f32 %2, %2, %8, %9\n\ts_add_u32 %6, %6, 3\n\tv_fma_f32 %3, %3, %8, %9\n\ts_add_u32 %7, %7, 3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(x), "v"(y) : "scc");
        } else if (OP == MIX_VALU_LDS) {
            asm volatile("ds_read_b128 %6, %10\n\tv_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t"
                         "ds_read_b128 %7, %10 offset:1024\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "=v"(r0), "=v"(r1) : "v"(x), "v"(y), "v"(laddr) : "memory");
        } else if (OP == MIX_FMA_EXP) {
            asm volatile("v_exp_f32 %6, %6\n\tv_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t"
                         "v_exp_f32 %7, %7\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    float acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + r0.x + r1.y + r2.z + r3.w +
                r4.x + r5.y + r6.z + r7.w + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
    if (acc == 123.456f) sink[0] = acc;
}

template <int OP>
static int run(int wps, int iters, double* cyc_per_instr_per_simd, double* wall_us)
{
    const int threads = 64 * 4 * wps, blocks = 256;
    unsigned long long* d_cycles; float* d_sink;
    CK(hipMalloc(&d_cycles, sizeof(unsigned long long) * blocks * 16));
    CK(hipMalloc(&d_sink, 64));
    const size_t lds = 96 * 1024;   // one workgroup per CU
    CK(hipFuncSetAttribute((const void*)bench_kernel<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench_kernel<OP>, dim3(blocks), dim3(threads), lds, 0, iters, d_cycles, d_sink);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench_kernel<OP>, dim3(blocks), dim3(threads), lds, 0, iters, d_cycles, d_sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 4 * wps);
    CK(hipMemcpy(h.data(), d_cycles, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    *cyc_per_instr_per_simd = med / ((double)iters * kInstrPerIter[OP] * wps);
    *wall_us = ms * 1e3;
    CK(hipFree(d_cycles)); CK(hipFree(d_sink));
    return 0;
}

template <int OP>
static int sweep(bool last)
{
    printf("  {\"op\": \"%s\", \"cycles_per_wave_instruction_per_simd\": {", kNames[OP]);
    const int wpss[4] = {1, 2, 3, 4};
    for (int k = 0; k < 4; k++) {
        double c, us;
        if (run<OP>(wpss[k], 20000, &c, &us)) return 1;
        printf("\"%d waves/SIMD\": %.2f%s", wpss[k], c, k < 3 ? ", " : "");
    }
    printf("}}%s\n", last ? "" : ",");
    return 0;
}

int main()
{
    // s_memtime runs at a fixed 100 MHz on some parts; calibrate it against hipEvent wall time and the reported clock
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int wall_clock_khz = 0; (void)hipDeviceGetAttribute(&wall_clock_khz, hipDeviceAttributeWallClockRate, 0);
    printf("{\"device\": \"%s\", \"clockRate_kHz\": %d, \"wallClockRate_kHz\": %d, \"note\": \"cycles = __builtin_readcyclecounter (s_memtime) ticks of the median wave / (instructions per wave x waves per SIMD); "
           "1 workgroup per CU, waves spread over the 4 SIMDs\",\n \"results\": [\n", prop.name, prop.clockRate, wall_clock_khz);
    if (sweep<FMA_IND>(false) || sweep<FMA_IND_2SRC_SGPR>(false) || sweep<FMA_DEP>(false) || sweep<PK_FMA_IND>(false) || sweep<EXP_IND>(false) || sweep<DPP_ADD_IND>(false) ||
        sweep<DPP_ADD_DEP>(false) || sweep<SWAP32_IND>(false) || sweep<CNDMASK_IND>(false) || sweep<SALU_IND>(false) || sweep<LDS_READ128>(false) ||
        sweep<LDS_READ64>(false) || sweep<MIX_VALU_SALU>(false) || sweep<MIX_VALU_LDS>(false) || sweep<MIX_FMA_EXP>(true))
        return 1;
    // tick calibration: a known-duration kernel
    {
        double c, us;
        if (run<FMA_IND>(1, 200000, &c, &us)) return 1;
        printf(" ],\n \"calibration\": {\"ticks_per_wave\": %.0f, \"wall_us\": %.1f, \"ticks_per_us\": %.1f}}\n", c * 200000 * 8, us, c * 200000 * 8 / us);
    }
    return 0;
}
