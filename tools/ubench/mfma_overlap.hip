// How much VALU work hides beside f32 MFMAs on one SIMD of gfx950?  Every wave runs the same stream: one MFMA followed by
// N independent v_fmac_f32 (no register shared with the MFMA), 4 accumulators round-robin.
//   hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap && ./mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int N>     // SHAPE 0: 4x4x1_16b, 1: 16x16x4, 2: no MFMA
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, int iters)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 acc[4];
    float v[8];
    for (int i = 0; i < 4; i++) acc[i] = f32x4{(float)l, 1.0f, 2.0f, (float)i};
    for (int i = 0; i < 8; i++) v[i] = (float)(l + i);
    float a = 1.0001f + l * 1e-6f, b = 0.9999f, c = 0.5f + l, d = 0.25f;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    f32x4 a4 = {a, b, a, b}, b4 = {b, a, b, a};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 big[2];
    for (int i = 0; i < 16; i++) { big[0][i] = (float)i; big[1][i] = (float)(i + l); }
    if (SHAPE == 4) { a4 = f32x4{0, 0, 0, 0}; b4 = a4; for (int i = 0; i < 4; i++) acc[i] = f32x4{0, 0, 0, 0}; }
    asm volatile("" : "+v"(a4), "+v"(b4));
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            if (SHAPE == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
            if (SHAPE == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
            if (SHAPE == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a4), "v"(b4));
            if (SHAPE == 4) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a4), "v"(b4));
            if (SHAPE == 5) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(big[g & 1]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < N; q++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[(g * N + q) & 7]) : "v"(c), "v"(d));
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i];
    if (SHAPE == 5) for (int i = 0; i < 16; i++) s += big[0][i] + big[1][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (l == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <int SHAPE, int N>
static void run()
{
    float *out; long long *cyc;
    (void)hipMalloc(&out, 1024 * 512 * 4); (void)hipMalloc(&cyc, 1024 * 8 * 8);
    printf("%-10s + %2d fmac:", SHAPE == 0 ? "4x4x1_16b" : SHAPE == 1 ? "16x16x4" : SHAPE == 3 ? "bf16 16x16x32" : SHAPE == 4 ? "i8 16x16x64" : SHAPE == 5 ? "f32 32x32x2" : "(none)", N);
    for (int wpe : {1, 2, 4}) {
        const int threads = wpe == 1 ? 256 : 512, blocks = 256 * (wpe == 4 ? 2 : 1), iters = 3000;
        hipLaunchKernelGGL((k<SHAPE, N>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);   // warm, clocks up
        hipLaunchKernelGGL((k<SHAPE, N>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(blocks * 8);
        (void)hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
        double mx = 0, sum = 0; const int nw = threads / 64;
        for (int bI = 0; bI < blocks; bI++) for (int w = 0; w < nw; w++) { sum += (double)h[bI * 8 + w]; if (h[bI * 8 + w] > mx) mx = (double)h[bI * 8 + w]; }
        // SIMD cycles per (MFMA + N fmac) group = slowest wave's cycles / groups per wave / ... the wpe waves of a SIMD run together
        printf("   %dw/SIMD: %6.1f cyc/group/wave, %5.1f per SIMD", wpe, sum / (blocks * nw) / iters / 16, mx / iters / 16 / wpe);
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<3, 0>(); run<3, 1>(); run<3, 2>(); run<3, 3>(); run<3, 4>(); run<3, 6>(); run<3, 8>();
    run<4, 0>(); run<4, 1>(); run<4, 2>(); run<4, 3>(); run<4, 4>(); run<4, 6>(); run<4, 8>();
    run<5, 0>(); run<5, 4>(); run<5, 8>(); run<5, 16>();
    run<2, 1>(); run<2, 4>(); run<2, 8>();
    run<0, 0>(); run<0, 1>(); run<0, 2>(); run<0, 3>(); run<0, 4>(); run<0, 6>();
    run<1, 0>(); run<1, 2>(); run<1, 4>(); run<1, 6>(); run<1, 8>(); run<1, 10>(); run<1, 12>(); run<1, 16>();
    return 0;
}
