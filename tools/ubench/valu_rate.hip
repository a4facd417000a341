// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction per SIMD for the
// instruction kinds the FFT kernel is made of, at 1/2/4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
#define ITERS 2000

template <int KIND>
__global__ void k(float *out, int iters)
{
    f2 a[8], b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int i = 0; i < 8; i++) a[i] = f2{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[1,1] op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(b));
                if (KIND == 5) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 6) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i].x));
                if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i].x) : "v"(b.x));
                if (KIND == 8) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 9) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[i].x));
                if (KIND == 10) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i].x));
                if (KIND == 11) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(b), "v"(c));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, float *d)
{
    int cus = 256;
    for (int wpe : {1, 2, 4, 8}) {
        int threads = 256, blocks = cus * wpe;            // wpe waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, threads>>>(d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, threads>>>(d, ITERS);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst_per_simd = (double)wpe * ITERS * REP;    // wave-instructions per SIMD
        double cyc = ms * 1e-3 * 2.4e9 / inst_per_simd;
        printf("%-34s waves/SIMD=%d  %.3f ms  %.2f cyc/inst/SIMD @2.4GHz\n", name, wpe, ms, cyc);
    }
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
    run<0>("v_fma_f32 (8 indep chains)", d);
    run<1>("v_pk_fma_f32", d);
    run<11>("v_pk_fma_f32 sgpr src0", d);
    run<2>("v_add_f32", d);
    run<3>("v_pk_add_f32", d);
    run<4>("v_pk_mul_f32 op_sel", d);
    run<5>("v_mul_f32", d);
    run<6>("v_cvt_f32_i32", d);
    run<7>("v_mov_b32", d);
    run<8>("v_med3_i32", d);
    run<9>("v_lshlrev_b32", d);
    run<10>("v_floor_f32", d);
    return 0;
}
