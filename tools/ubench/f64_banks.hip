// Does a v_fma_f64 with three distinct VGPR-pair sources issue at the rate of one whose sources repeat?  (ssdr_wf_exact.hip runs
// at half the rate its instruction count predicts.)  Explicit registers: VGPR banks are (index mod 4).
//   hipcc --offload-arch=gfx950 -O3 f64_banks.hip -o f64_banks && ./f64_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4000
#define CLOB "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75"

template <int KIND>
__global__ void k(double *out, int iters)
{
    // v[20:43] hold data; eight independent destinations v[60:75]
    asm volatile(
        "v_cvt_f64_i32 v[20:21], v0\n v_cvt_f64_i32 v[22:23], v0\n v_cvt_f64_i32 v[24:25], v0\n v_cvt_f64_i32 v[26:27], v0\n"
        "v_cvt_f64_i32 v[28:29], v0\n v_cvt_f64_i32 v[30:31], v0\n v_cvt_f64_i32 v[32:33], v0\n v_cvt_f64_i32 v[34:35], v0\n"
        "v_cvt_f64_i32 v[36:37], v0\n v_cvt_f64_i32 v[38:39], v0\n v_cvt_f64_i32 v[40:41], v0\n v_cvt_f64_i32 v[42:43], v0\n"
        "v_cvt_f64_i32 v[60:61], v0\n v_cvt_f64_i32 v[62:63], v0\n v_cvt_f64_i32 v[64:65], v0\n v_cvt_f64_i32 v[66:67], v0\n"
        "v_cvt_f64_i32 v[68:69], v0\n v_cvt_f64_i32 v[70:71], v0\n v_cvt_f64_i32 v[72:73], v0\n v_cvt_f64_i32 v[74:75], v0\n" :::
        "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43", CLOB);
    for (int it = 0; it < iters; it++) {
        if (KIND == 0)      // two sources shared by every instruction (the f64_rate.hip pattern), accumulate in place
            asm volatile("v_fma_f64 v[60:61], v[20:21], v[22:23], v[60:61]\n v_fma_f64 v[62:63], v[20:21], v[22:23], v[62:63]\n"
                         "v_fma_f64 v[64:65], v[20:21], v[22:23], v[64:65]\n v_fma_f64 v[66:67], v[20:21], v[22:23], v[66:67]\n"
                         "v_fma_f64 v[68:69], v[20:21], v[22:23], v[68:69]\n v_fma_f64 v[70:71], v[20:21], v[22:23], v[70:71]\n"
                         "v_fma_f64 v[72:73], v[20:21], v[22:23], v[72:73]\n v_fma_f64 v[74:75], v[20:21], v[22:23], v[74:75]\n" ::: CLOB);
        if (KIND == 1)      // three distinct sources per instruction, different ones each time, banks 0-1 / 2-3 / 0-1
            asm volatile("v_fma_f64 v[60:61], v[20:21], v[26:27], v[28:29]\n v_fma_f64 v[62:63], v[24:25], v[30:31], v[32:33]\n"
                         "v_fma_f64 v[64:65], v[28:29], v[34:35], v[36:37]\n v_fma_f64 v[66:67], v[32:33], v[38:39], v[40:41]\n"
                         "v_fma_f64 v[68:69], v[36:37], v[42:43], v[20:21]\n v_fma_f64 v[70:71], v[40:41], v[22:23], v[24:25]\n"
                         "v_fma_f64 v[72:73], v[20:21], v[30:31], v[36:37]\n v_fma_f64 v[74:75], v[24:25], v[34:35], v[40:41]\n" ::: CLOB);
        if (KIND == 2)      // three distinct sources, all on banks 0-1
            asm volatile("v_fma_f64 v[60:61], v[20:21], v[24:25], v[28:29]\n v_fma_f64 v[62:63], v[24:25], v[28:29], v[32:33]\n"
                         "v_fma_f64 v[64:65], v[28:29], v[32:33], v[36:37]\n v_fma_f64 v[66:67], v[32:33], v[36:37], v[40:41]\n"
                         "v_fma_f64 v[68:69], v[36:37], v[40:41], v[20:21]\n v_fma_f64 v[70:71], v[40:41], v[20:21], v[24:25]\n"
                         "v_fma_f64 v[72:73], v[20:21], v[28:29], v[36:37]\n v_fma_f64 v[74:75], v[24:25], v[32:33], v[40:41]\n" ::: CLOB);
        if (KIND == 3)      // two distinct VGPR sources + an inline constant
            asm volatile("v_fma_f64 v[60:61], v[20:21], 2.0, v[28:29]\n v_fma_f64 v[62:63], v[24:25], 2.0, v[32:33]\n"
                         "v_fma_f64 v[64:65], v[28:29], 2.0, v[36:37]\n v_fma_f64 v[66:67], v[32:33], 2.0, v[40:41]\n"
                         "v_fma_f64 v[68:69], v[36:37], 2.0, v[20:21]\n v_fma_f64 v[70:71], v[40:41], 2.0, v[24:25]\n"
                         "v_fma_f64 v[72:73], v[20:21], 2.0, v[36:37]\n v_fma_f64 v[74:75], v[24:25], 2.0, v[40:41]\n" ::: CLOB);
        if (KIND == 4)      // v_add_f64, two distinct sources
            asm volatile("v_add_f64 v[60:61], v[20:21], v[26:27]\n v_add_f64 v[62:63], v[24:25], v[30:31]\n"
                         "v_add_f64 v[64:65], v[28:29], v[34:35]\n v_add_f64 v[66:67], v[32:33], v[38:39]\n"
                         "v_add_f64 v[68:69], v[36:37], v[42:43]\n v_add_f64 v[70:71], v[40:41], v[22:23]\n"
                         "v_add_f64 v[72:73], v[20:21], v[30:31]\n v_add_f64 v[74:75], v[24:25], v[34:35]\n" ::: CLOB);
        if (KIND == 5)      // butterfly-like: pairs of dependent FMAs (s = fma(w, v, u); a = fma(w, v', s); b = fma(2, u, -a))
            asm volatile("v_fma_f64 v[60:61], v[20:21], v[26:27], v[28:29]\n v_fma_f64 v[62:63], v[24:25], v[30:31], v[32:33]\n"
                         "v_fma_f64 v[64:65], v[20:21], v[34:35], v[60:61]\n v_fma_f64 v[66:67], v[24:25], v[38:39], v[62:63]\n"
                         "v_fma_f64 v[68:69], v[28:29], 2.0, -v[64:65]\n v_fma_f64 v[70:71], v[32:33], 2.0, -v[66:67]\n"
                         "v_fma_f64 v[72:73], v[36:37], v[30:31], v[40:41]\n v_fma_f64 v[74:75], v[72:73], v[34:35], v[42:43]\n" ::: CLOB);
    }
    double r;
    asm volatile("v_add_f64 %0, v[60:61], v[74:75]" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char *name, double *d)
{
    for (int wpe : {1, 2, 3}) {
        int threads = 256, blocks = 256 * wpe;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, threads>>>(d, 400);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, threads>>>(d, ITERS);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-62s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instruction per SIMD\n", name, wpe, ms, ms * 1e6 / ((double)wpe * ITERS * 8));
    }
}

int main()
{
    double *d; hipMalloc(&d, 256 * 3 * 256 * 8);
    run<0>("v_fma_f64, two sources shared, accumulate in place", d);
    run<1>("v_fma_f64, three distinct sources (banks 01 / 23 / 01)", d);
    run<2>("v_fma_f64, three distinct sources (all banks 01)", d);
    run<3>("v_fma_f64, two distinct sources + inline constant", d);
    run<4>("v_add_f64, two distinct sources", d);
    run<5>("v_fma_f64, butterfly-like dependent pairs", d);
    return 0;
}
