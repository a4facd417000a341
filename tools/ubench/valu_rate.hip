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
    unsigned long long mask = __ballot(threadIdx.x & 1);
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
                if (KIND == 12) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i].x) : "v"(b.x) : "vcc");
                if (KIND == 13) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(a[i].x), "v"(b.x) : "vcc");
                if (KIND == 14) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 15) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 16) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 17) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 18) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 19) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i].x));
                if (KIND == 20) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x) : "vcc");
                if (KIND == 21) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i].x));
                if (KIND == 22) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].x));
                if (KIND == 23) asm volatile("v_bfe_u32 %0, %0, 21, 11" : "+v"(a[i].x));
                if (KIND == 24) asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2 clamp" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 25) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 26) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 27) asm volatile("v_dot2c_i32_i16 %0, %1, %1" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 28) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i].x));
                if (KIND == 29) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 30) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 31) asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x));
                if (KIND == 32) asm volatile("v_readlane_b32 s20, %0, 63" : : "v"(a[i].x) : "s20");
                if (KIND == 33) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i].x) : "v"(b.x));
                if (KIND == 34) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 35) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 36) asm volatile("v_fmac_f32 %0, %1, %2\n v_sqrt_f32 %3, %3" : "+v"(a[i].x), "+v"(a[(i + 4) & 7].y) : "v"(b.x), "v"(c.x));
                if (KIND == 37) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 38) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i].x));
                if (KIND == 40) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "s"(mask));
                if (KIND == 41) asm volatile("v_cmp_ge_f32 vcc, %1, %0\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x) : "vcc");
                if (KIND == 42) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 43) asm volatile("v_cmp_ge_f32 vcc, %1, %0\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x), "+v"(a[(i + 3) & 7].y), "+v"(a[(i + 5) & 7].y) : "v"(b.x) : "vcc");
                if (KIND == 44) asm volatile("v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2" : "+v"(a[i].x), "+v"(a[(i + 3) & 7].y) : "v"(b.x));
                if (KIND == 45) asm volatile("s_nop 0");
                if (KIND == 46) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 47) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 48) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f706e44" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 49) asm volatile("v_mul_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 50) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 51) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i].x) : : "vcc");
                if (KIND == 52) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 53) asm volatile("v_cmp_ge_f32_e64 s[20:21], %1, %0\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i].x) : "v"(b.x) : "s20", "s21");
                if (KIND == 54) asm volatile("v_cmp_ge_f32_e64 s[20:21], %1, %0\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i].x), "+v"(a[(i + 3) & 7].y), "+v"(a[(i + 5) & 7].y) : "v"(b.x) : "s20", "s21");
                if (KIND == 55) asm volatile("v_cmp_ge_f32 vcc, %1, %0\n s_nop 1\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i].x) : "v"(b.x) : "vcc");
                if (KIND == 56) asm volatile("s_mov_b64 vcc, %2\n s_nop 3\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x), "s"(mask) : "vcc");
                if (KIND == 39) asm volatile("v_fma_f32 %0, %1, %2, %0\n v_cvt_f32_i32 %3, %3" : "+v"(a[i].x), "+v"(a[(i + 4) & 7].y) : "v"(b.x), "v"(c.x));
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
    for (int wpe : {2, 4}) {
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
    run<12>("v_cmp_ge_f32 + v_addc (2 instr)", d);
    run<13>("v_cmp_ge_f32", d);
    run<14>("v_add_u32", d);
    run<15>("v_and_b32", d);
    run<16>("v_lshl_add_u32", d);
    run<17>("v_med3_f32", d);
    run<18>("v_max_f32", d);
    run<19>("v_cvt_i32_f32", d);
    run<20>("v_cndmask_b32", d);
    run<21>("v_sqrt_f32", d);
    run<22>("v_rcp_f32", d);
    run<23>("v_bfe_u32", d);
    run<24>("v_sub_f32 + v_mul_f32 clamp (2 instr)", d);
    run<25>("v_perm_b32", d);
    run<26>("v_mad_u32_u24", d);
    run<27>("v_dot2c_i32_i16", d);
    run<28>("v_cvt_f32_u32", d);
    run<29>("v_max3_f32 abs abs", d);
    run<37>("v_max3_f32", d);
    run<30>("s_nop 1 + v_mov_b32_dpp wave_shr:1", d);
    run<31>("s_nop 1 + v_max_f32_dpp row_shr:1", d);
    run<32>("v_readlane_b32", d);
    run<33>("v_cvt_f32_i32_sdwa WORD_1 sext", d);
    run<34>("v_cvt_pk_i16_i32", d);
    run<35>("v_fmac_f32", d);
    run<36>("v_fmac_f32 + v_sqrt_f32 (pair)", d);
    run<39>("v_fma_f32 + v_cvt_f32_i32 (pair)", d);
    run<38>("v_rndne_f32", d);
    run<40>("v_cndmask_b32_e64 sgpr mask", d);
    run<50>("v_cndmask_b32_e64 vcc", d);
    run<51>("v_addc_co_u32 vcc in/out", d);
    run<52>("v_cndmask_b32 vcc, dst != src", d);
    run<53>("v_cmp_e64 sgpr + s_nop 1 + v_cndmask_e64 (2 instr)", d);
    run<54>("v_cmp_e64 sgpr + 2 v_add + v_cndmask_e64 (4 instr)", d);
    run<55>("v_cmp vcc + s_nop 1 + v_addc (2 instr)", d);
    run<56>("s_mov vcc + s_nop 3 + v_cndmask vcc (1 valu)", d);
    run<41>("v_cmp + s_nop 1 + v_cndmask vcc (2 instr)", d);
    run<42>("v_cndmask_b32 vcc (no clobber)", d);
    run<43>("v_cmp + 2 v_add_f32 + v_cndmask (4 instr)", d);
    run<44>("2 v_add_f32 (2 instr)", d);
    run<45>("s_nop 0", d);
    run<46>("v_add_u32_sdwa WORD_0", d);
    run<47>("v_min3_u32", d);
    run<48>("v_fmaak_f32", d);
    run<49>("v_mul_f32_dpp row_shr:1", d);
    return 0;
}
