// Issue-rate microbenchmark for the instruction kinds of the float64 waterfall kernel (ssdr_wf_exact.hip) on gfx950:
// cycles per wave-instruction per SIMD at 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 f64_rate.hip -o f64_rate && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define ITERS 2000

template <int KIND>
__global__ void k(double *out, int iters)
{
    double a[8], b = 1.0000001, c = 0.25;
    float fa[8];
    int ia[8];
    for (int i = 0; i < 8; i++) { a[i] = (double)threadIdx.x + i; fa[i] = (float)i + threadIdx.x; ia[i] = i * 77 + threadIdx.x; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 1) asm volatile("v_add_f64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (KIND == 2) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (KIND == 3) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(ia[i]));
                if (KIND == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(fa[i]));
                if (KIND == 5) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fa[i]) : "v"(a[i]));
                if (KIND == 6) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ia[i]), "+v"(ia[(i + 1) & 7]));
                if (KIND == 7) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(ia[i]), "+v"(ia[(i + 1) & 7]));
                if (KIND == 8) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));
                if (KIND == 9) asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 10) asm volatile("v_fma_f64 %0, %2, %3, %0\n v_fma_f32 %1, %4, %4, %1" : "+v"(a[i]), "+v"(fa[i]) : "v"(b), "v"(c), "v"(fa[(i + 1) & 7]));
                if (KIND == 11) asm volatile("v_bfe_i32 %0, %1, 0, 16" : "=v"(ia[i]) : "v"(ia[(i + 1) & 7]));
                if (KIND == 12) asm volatile("v_ashrrev_i32 %0, 16, %1" : "=v"(ia[i]) : "v"(ia[(i + 1) & 7]));
                if (KIND == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(ia[i]) : "v"(ia[(i + 1) & 7]));
                if (KIND == 14) asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 15) asm volatile("v_add_f64 %0, %1, -%0" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (double)fa[i] + (double)ia[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, double *d)
{
    int cus = 256;
    for (int wpe : {2, 4}) {
        int threads = 256, blocks = cus * wpe;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, threads>>>(d, 200);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, threads>>>(d, ITERS);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst_per_simd = (double)wpe * ITERS * REP;
        printf("%-40s waves/SIMD=%d  %.3f ms  %.2f cyc/inst/SIMD @2.4GHz\n", name, wpe, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
    }
}

int main()
{
    double *d; hipMalloc(&d, 256 * 8 * 256 * 8 * 2);
    run<13>("v_mov_b32 (reference: 2 cyc class)", d);
    run<0>("v_fma_f64", d);
    run<9>("v_fma_f64 neg src", d);
    run<1>("v_add_f64", d);
    run<15>("v_add_f64 neg src", d);
    run<2>("v_mul_f64", d);
    run<3>("v_cvt_f64_i32", d);
    run<4>("v_cvt_f64_f32", d);
    run<5>("v_cvt_f32_f64", d);
    run<6>("v_permlane32_swap_b32", d);
    run<7>("v_permlane16_swap_b32", d);
    run<8>("v_mov_b32_dpp quad_perm (+s_nop 1)", d);
    run<10>("v_fma_f64 + v_fma_f32 pair", d);
    run<11>("v_bfe_i32", d);
    run<12>("v_ashrrev_i32", d);
    run<14>("v_pk_mov_b32", d);
    return 0;
}
