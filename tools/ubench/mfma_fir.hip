// f32 MFMA microbenchmark for gfx950: operand layouts, bitwise equality with an fmaf chain, issue rates alone and
// beside VALU waves -- the facts the matrix-pipe FIR of ssdr_audio.hip is built on.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_fir.hip -o mfma_fir && ./mfma_fir
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- 1. layouts and the k order
__global__ void k_4x4(const float *a, const float *b, float *d, int steps)
{
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; s++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s * 64 + l], b[s * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[r * 64 + l] = acc[r];
}
__global__ void k_16x16(const float *a, const float *b, float *d, int steps)
{
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s * 64 + l], b[s * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[r * 64 + l] = acc[r];
}

// ---- 2. rates.  KIND 0: 4x4x1_16b, 1: 16x16x4, 2: v_fmac only, 3: waves 0..3 of a 512-thread group 4x4x1, 4..7 v_fmac,
//         4: the same with 16x16x4, 5: every wave interleaves 144 4x4x1 + 320 v_fmac per iteration, 6: 48 16x16x4 + 320 v_fmac
template <int KIND>
__global__ __launch_bounds__(512) void k_rate(float *out, long long *cyc, int iters)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 acc[4];
    float v[8];
    for (int i = 0; i < 4; i++) acc[i] = f32x4{(float)l, 1.0f, 2.0f, (float)i};
    for (int i = 0; i < 8; i++) v[i] = (float)(l + i);
    const float a = 1.0001f + l * 1e-6f, b = 0.9999f;
    const bool mf = KIND == 0 || KIND == 1 || ((KIND == 3 || KIND == 4) && w < 4);
    const bool va = KIND == 2 || ((KIND == 3 || KIND == 4) && w >= 4);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (KIND == 5 || KIND == 6) {
#pragma unroll
            for (int g = 0; g < (KIND == 5 ? 36 : 12); g++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (KIND == 5) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v[(g + i) & 7], acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[(g + i) & 7], acc[i], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < (KIND == 5 ? 2 : 6); q++)
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[(2 * i + q) & 7]) : "v"(a), "v"(b));
                }
                if (KIND == 5 && (g & 3) == 3) {
#pragma unroll
                    for (int q = 0; q < 8; q++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[q]) : "v"(a), "v"(b));
                }
                if (KIND == 6 && (g & 3) == 3) {
#pragma unroll
                    for (int q = 0; q < 8; q++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[q]) : "v"(a), "v"(b));
                }
            }
        } else if (mf) {
#pragma unroll
            for (int g = 0; g < 16; g++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (KIND == 0 || KIND == 3) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                }
        } else if (va) {
#pragma unroll
            for (int g = 0; g < 8; g++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (l == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <int KIND>
static void rate(const char *name, int per_iter_mfma, int per_iter_valu)
{
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&cyc, 4096 * 8 * 8);
    for (int wpe : {1, 2, 4}) {
        // wpe waves per SIMD: 512-thread groups put 2 waves on each SIMD
        const int threads = wpe == 1 ? 256 : 512, blocks = 256 * (wpe == 4 ? 2 : 1), iters = 400;
        if ((KIND == 3 || KIND == 4) && wpe == 1) continue;
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * 8);
        hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
        const int nw = threads / 64;
        double lo[2] = {0, 0}; int n[2] = {0, 0};
        for (int bI = 0; bI < blocks; bI++) for (int w = 0; w < nw; w++) { const int hf = (w >= 4); lo[hf] += (double)h[bI * 8 + w]; n[hf]++; }
        printf("%-34s %d waves/SIMD  %.3f ms  cycles/iter/wave: waves0-3 %.0f", name, wpe, ms, lo[0] / n[0] / iters);
        if (n[1]) printf("  waves4-7 %.0f", lo[1] / n[1] / iters);
        printf("   (per iter: %d mfma, %d valu)\n", per_iter_mfma, per_iter_valu);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    const int S = 40;
    std::vector<float> a(S * 64), b(S * 64), d(256);
    srand(7);
    for (auto &x : a) x = (float)(rand() % 2001 - 1000) / 997.0f;
    for (auto &x : b) x = (float)(rand() % 65536 - 32768) * 1.37f;
    float *da, *db, *dd;
    hipMalloc(&da, S * 256); hipMalloc(&db, S * 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), S * 256, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), S * 256, hipMemcpyHostToDevice);
    // 4x4x1_16b: hypotheses for D[reg r][lane L], block = L >> 2
    for (int steps : {1, 36}) {
        hipLaunchKernelGGL(k_4x4, dim3(1), dim3(64), 0, 0, da, db, dd, steps);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        int ok1 = 0, ok2 = 0;
        for (int r = 0; r < 4; r++) for (int L = 0; L < 64; L++) {
            float c1 = 0, c2 = 0;
            for (int s = 0; s < steps; s++) {
                c1 = fmaf(a[s * 64 + (L & ~3) + r], b[s * 64 + L], c1);       // D_b[i = r][j = L & 3] = A_b[i] B_b[j]
                c2 = fmaf(a[s * 64 + L], b[s * 64 + (L & ~3) + r], c2);       // transposed
            }
            ok1 += !memcmp(&c1, &d[r * 64 + L], 4);
            ok2 += !memcmp(&c2, &d[r * 64 + L], 4);
        }
        printf("4x4x1_16b steps=%d: D[r][L] == chain a[blk*4+r]*b[L] bitwise: %d/256; transposed: %d/256\n", steps, ok1, ok2);
    }
    for (int steps : {1, 12}) {
        hipLaunchKernelGGL(k_16x16, dim3(1), dim3(64), 0, 0, da, db, dd, steps);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        int ok = 0, okrev = 0;
        for (int r = 0; r < 4; r++) for (int L = 0; L < 64; L++) {
            const int row = (L >> 4) * 4 + r, col = L & 15;
            float c = 0, cr = 0;
            for (int s = 0; s < steps; s++) {
                for (int k = 0; k < 4; k++) c = fmaf(a[s * 64 + row + 16 * k], b[s * 64 + col + 16 * k], c);
                for (int k = 3; k >= 0; k--) cr = fmaf(a[s * 64 + row + 16 * k], b[s * 64 + col + 16 * k], cr);
            }
            ok += !memcmp(&c, &d[r * 64 + L], 4);
            okrev += !memcmp(&cr, &d[r * 64 + L], 4);
        }
        printf("16x16x4 steps=%d: D == k-ascending fmaf chain bitwise: %d/256; k-descending: %d/256\n", steps, ok, okrev);
    }
    rate<0>("4x4x1_16b only", 64, 0);
    rate<1>("16x16x4 only", 64, 0);
    rate<2>("v_fmac only", 0, 64);
    rate<3>("w0-3 4x4x1 | w4-7 v_fmac", 64, 64);
    rate<4>("w0-3 16x16x4 | w4-7 v_fmac", 64, 64);
    rate<5>("interleaved 144 4x4x1 + 360 fmac", 144, 360);
    rate<6>("interleaved 48 16x16x4 + 312 fmac", 48, 312);
    return 0;
}
