// Calibration of the SQ "VALU busy" counters on gfx950 (round 5): kernels whose vector-ALU load is known by construction, run under
//   rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES ...
// so that the same counters on the shipped kernels can be read as a fraction of a SATURATED vector ALU (profiles/r05_full_issue_breakdown.txt).
//   hipcc --offload-arch=gfx950 -O3 pmc_calib.hip -o pmc_calib && ./pmc_calib
// KIND 0: v_fma_f32 only, 8 independent chains (2-cycle class)      4 waves / SIMD  -> saturated
// KIND 1: v_cvt_f32_i32 only (4-cycle class)                        4 waves / SIMD  -> saturated
// KIND 2: alternating v_fma_f32 / v_cvt_f32_i32                     4 waves / SIMD  -> saturated
// KIND 3: v_fma_f32, ONE dependent chain                            1 wave  / SIMD  -> latency-bound, ALU mostly idle
// KIND 4: v_fma_f32 only                                            1 wave  / SIMD
// KIND 5: 1 v_fma_f32 per 3 s_nop 0                                 4 waves / SIMD  -> a quarter of the issue slots
// KIND 6: v_pk_fma_f32 only (slow class, two lane-operations)        4 waves / SIMD
// KIND 7: alternating v_pk_fma_f32 / v_fma_f32                       4 waves / SIMD  -> do packed and scalar FMAs share quad-cycles?
// KIND 8: alternating v_max_f32_dpp row_shr:1 (+ s_nop 1) / v_fma_f32
// KIND 9: alternating v_cvt_f32_i32 / v_perm_b32 (slow / slow)
// KIND 10: alternating v_sqrt_f32 / v_fma_f32 (transcendental / fast)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define ITERS 4000

template <int KIND>
__global__ void calib(float *out, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    float a[8], b = 1.0001f, c = 0.5f;
    f2 pk[8], pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    for (int i = 0; i < 8; i++) { a[i] = (float)threadIdx.x + i; pk[i] = f2{(float)i, (float)threadIdx.x}; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0 || KIND == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 1) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
                if (KIND == 2) { if (i & 1) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i])); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c)); }
                if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(b), "v"(c));
                if (KIND == 5) asm volatile("v_fma_f32 %0, %1, %2, %0\n s_nop 0\n s_nop 0\n s_nop 0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pk[i]) : "v"(pb), "v"(pc));
                if (KIND == 7) { if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pk[i]) : "v"(pb), "v"(pc)); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c)); }
                if (KIND == 8) { if (i & 1) asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i])); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c)); }
                if (KIND == 9) { if (i & 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); else asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i])); }
                if (KIND == 10) { if (i & 1) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i])); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c)); }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + pk[i].x + pk[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, float *d, int wpe)
{
    const int threads = 256, blocks = 256 * wpe;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    calib<KIND><<<blocks, threads>>>(d, 200);              // clocks up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    calib<KIND><<<blocks, threads>>>(d, ITERS);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("calib<%d> %-44s waves/SIMD=%d  %.3f ms  %.1f wave-instructions per SIMD and us\n", KIND, name, wpe, ms,
           (double)wpe * ITERS * REP / (ms * 1e3));
}

int main()
{
    float *d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("v_fma_f32 x8 chains", d, 4);
    run<1>("v_cvt_f32_i32", d, 4);
    run<2>("v_fma_f32 / v_cvt_f32_i32 alternating", d, 4);
    run<3>("v_fma_f32 one dependent chain", d, 1);
    run<4>("v_fma_f32 x8 chains", d, 1);
    run<5>("v_fma_f32 + 3 s_nop", d, 4);
    run<6>("v_pk_fma_f32", d, 4);
    run<7>("v_pk_fma_f32 / v_fma_f32 alternating", d, 4);
    run<8>("v_max_f32_dpp / v_fma_f32 alternating", d, 4);
    run<9>("v_cvt_f32_i32 / v_perm_b32 alternating", d, 4);
    run<10>("v_sqrt_f32 / v_fma_f32 alternating", d, 4);
    return 0;
}
