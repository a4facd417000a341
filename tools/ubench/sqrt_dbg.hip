#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
__device__ float sqrt_rn(float p, float* raw){
    const float s = __builtin_amdgcn_sqrtf(p); *raw = s;
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-sd, s, p), ru = fmaf(-su, s, p);
    float r = (rd <= 0.0f) ? sd : s;
    r = (ru > 0.0f) ? su : r;
    return r;
}
__global__ void k(unsigned* out, unsigned* cnt, unsigned long long* hist){
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t u = 0x00800000ull + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < 0x7F800000ull; u += stride) {
        float p = __uint_as_float((uint32_t)u), raw;
        float a = sqrt_rn(p, &raw), b = sqrtf(p);
        int d = (int)(__float_as_uint(raw) - __float_as_uint(b));
        atomicAdd(&hist[d+4 < 0 ? 0 : (d+4 > 8 ? 8 : d+4)], 1ull);
        if (__float_as_uint(a) != __float_as_uint(b)) { unsigned i = atomicAdd(cnt, 1u); if (i < 16) { out[4*i]=(unsigned)u; out[4*i+1]=__float_as_uint(raw); out[4*i+2]=__float_as_uint(a); out[4*i+3]=__float_as_uint(b);} }
    }
}
int main(){ unsigned *o,*c; unsigned long long* h; hipMalloc(&o,1024); hipMalloc(&c,4); hipMalloc(&h,72); hipMemset(c,0,4); hipMemset(h,0,72);
  k<<<2048,256>>>(o,c,h); unsigned ho[64], hc; unsigned long long hh[9]; hipMemcpy(ho,o,256,hipMemcpyDeviceToHost); hipMemcpy(&hc,c,4,hipMemcpyDeviceToHost); hipMemcpy(hh,h,72,hipMemcpyDeviceToHost);
  printf("mismatches %u\nraw-minus-true ulp histogram (-4..+4):", hc); for(int i=0;i<9;i++) printf(" %llu", hh[i]); printf("\n");
  for (int i=0;i<16 && i<(int)hc;i++){ float p; memcpy(&p,&ho[4*i],4); printf("p=%08x (%g) raw=%08x mine=%08x true=%08x host_sqrtf=%08x\n", ho[4*i], p, ho[4*i+1], ho[4*i+2], ho[4*i+3], ({float r=sqrtf(p); unsigned b; memcpy(&b,&r,4); b;})); }
  return 0; }
