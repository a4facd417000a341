// Exhaustive check of cheaper correctly-rounded-sqrt candidates against sqrtf() over every positive normal float.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off sqrt_variants.hip -o sqrt_variants && ./sqrt_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define NV 5
__device__ float cand(int v, float p)
{
    if (v == 0) {           // sqrt, one residual, correction through rcp
        const float s = __builtin_amdgcn_sqrtf(p), h = 0.5f * __builtin_amdgcn_rcpf(s);
        return fmaf(fmaf(-s, s, p), h, s);
    }
    if (v == 1) {           // rsq only
        const float y = __builtin_amdgcn_rsqf(p), s = p * y, h = 0.5f * y;
        return fmaf(fmaf(-s, s, p), h, s);
    }
    if (v == 2) {           // rsq, two corrections
        const float y = __builtin_amdgcn_rsqf(p), s = p * y, h = 0.5f * y;
        const float s1 = fmaf(fmaf(-s, s, p), h, s);
        return fmaf(fmaf(-s1, s1, p), h, s1);
    }
    if (v == 3) {           // sqrt + rsq side by side
        const float s = __builtin_amdgcn_sqrtf(p), h = 0.5f * __builtin_amdgcn_rsqf(p);
        return fmaf(fmaf(-s, s, p), h, s);
    }
    {                       // rsq, refined h (Markstein's full sequence)
        const float y = __builtin_amdgcn_rsqf(p);
        float g = p * y, h = 0.5f * y;
        const float r = fmaf(-h, g, 0.5f);
        g = fmaf(g, r, g); h = fmaf(h, r, h);
        return fmaf(fmaf(-g, g, p), h, g);
    }
}
__global__ void k(unsigned long long *bad, unsigned *ex, uint32_t lo, uint32_t hi)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long b[NV] = {0};
    for (uint64_t u = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < hi; u += stride) {
        const float p = __uint_as_float((uint32_t)u), t = sqrtf(p);
        for (int v = 0; v < NV; v++) {
            const float c = cand(v, p);
            if (__float_as_uint(c) != __float_as_uint(t)) { atomicMax(&ex[v], (uint32_t)u); b[v]++; }
        }
    }
    for (int v = 0; v < NV; v++) if (b[v]) atomicAdd(&bad[v], b[v]);
}
int main()
{
    unsigned long long *bad, h[NV]; unsigned *ex, he[NV];
    hipMalloc(&bad, sizeof h); hipMalloc(&ex, sizeof he);
    const char *names[NV] = {"sqrt + rcp, one correction", "rsq, one correction", "rsq, two corrections", "sqrt + rsq, one correction", "rsq, refined g and h, one correction"};
    struct { const char *what; uint32_t lo, hi; } ranges[] = {{"all positive normal floats", 0x00800000u, 0x7F800000u},
                                                              {"1 <= p < 2^33 (integer powers)", 0x3F800000u, 0x50000000u}};
    for (auto &r : ranges) {
        hipMemset(bad, 0, sizeof h); hipMemset(ex, 0, sizeof he);
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, ex, r.lo, r.hi);
        hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(he, ex, sizeof he, hipMemcpyDeviceToHost);
        printf("%s:\n", r.what);
        for (int v = 0; v < NV; v++) printf("  %-40s mismatches %llu  (largest failing bits %08x)\n", names[v], h[v], he[v]);
    }
    return 0;
}
