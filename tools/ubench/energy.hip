// Energy per wave-instruction on gfx950, from the board's energy accumulator (rocm_smi): the shipped kernels run at the
// 1400 W cap, where time = energy / cap, so joules per instruction -- not cycles -- is the cost that matters.
//   hipcc --offload-arch=gfx950 -O3 energy.hip -o energy -L/opt/rocm/lib -lrocm_smi64 && ./energy
// Every variant: 4 waves per SIMD on all CUs, 64 instructions of one kind per loop trip on 8 independent chains, launches
// back to back for ~1.5 s after a 0.5 s warm-up; reports W, G wave-instr/s, nJ per wave-instruction (whole board) and the
// same after subtracting the s_nop variant's power ("what the instruction itself adds at that clock").
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP 64

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ float lds[256 * 4 * 2];
    f2 a[8], b, c;
    const unsigned t = threadIdx.x + blockIdx.x * 977u;
    for (int i = 0; i < 8; i++) a[i] = f2{(float)((t * 2654435761u + i * 40503u) >> 8) * 0x1p-12f - 2048.0f, (float)((t * 40503u + i * 2654435761u) >> 8) * 0x1p-12f - 2048.0f};
    b = f2{-1.0f, -1.0f};                                      // a <- -a + c: alternates between two unrelated values
    c = f2{(float)(t % 8191u) * 0.37f - 1500.0f, (float)(t % 4093u) * 0.91f - 1800.0f};
    lds[threadIdx.x * 4 + 0] = a[0].x; lds[threadIdx.x * 4 + 1] = a[0].y; lds[threadIdx.x * 4 + 2] = a[1].x; lds[threadIdx.x * 4 + 3] = a[1].y;
    lds[1024 + threadIdx.x * 4] = a[2].x;
    __syncthreads();
    const unsigned laddr = threadIdx.x * 16;
    f4 q = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("s_nop 0");
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 3) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 5) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(c.x));
                if (KIND == 6) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i].y) : "v"(a[i].x));
                if (KIND == 7) asm volatile("v_cmp_ge_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(c.x) : "vcc");
                if (KIND == 8) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i].x) : "v"(c.x));
                if (KIND == 9) asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(a[i].y) : "v"(a[i].x));
                if (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i].y) : "v"(a[(i + 1) & 7].x));
                if (KIND == 11) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(c.x), "v"(c.y));
                if (KIND == 12) asm volatile("ds_read_b64 %0, %1" : "=v"(a[i]) : "v"(laddr));
                if (KIND == 13) asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(laddr));
                if (KIND == 14) asm volatile("ds_write_b64 %0, %1" : : "v"(laddr), "v"(a[i]));
                if (KIND == 15) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[i].y) : "v"(a[i].x));
                if (KIND == 16) asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
                if (KIND == 17) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 18) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i].x) : "v"(laddr));
                if (KIND == 19) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %4" : "+v"(a[i].x), "+v"(a[i].y) : "v"(b.x), "v"(c.x), "v"(c.y));
                if (KIND == 20) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(a[i]) : "v"(laddr));
            }
        }
        if (KIND >= 12 && KIND <= 14 || KIND == 18 || KIND == 20) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = q[0] + q[1] + q[2] + q[3];
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
}

// streaming kernels: read-only and copy, 16 B per lane, grid-stride
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ src, float *out, size_t n)
{
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(src + i);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}
__global__ __launch_bounds__(256) void k_copy(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double energy_j()
{
    uint64_t e = 0, ts = 0; float res = 0;
    if (rsmi_dev_energy_count_get(0, &e, &res, &ts) != RSMI_STATUS_SUCCESS) return -1.0;
    return (double)e * res * 1e-6;
}
static double g_base_w = 0.0, g_base_rate = 0.0;

template <typename F>
static void measure(const char *name, double units_per_launch, const char *unit, F launch, int instr_per_unit = 1)
{
    const double t_w = now();
    while (now() - t_w < 0.5) { for (int i = 0; i < 8; i++) launch(); hipDeviceSynchronize(); }
    const double e0 = energy_j(), t0 = now();
    long launches = 0;
    while (now() - t0 < 1.5) { for (int i = 0; i < 8; i++) launch(); hipDeviceSynchronize(); launches += 8; }
    const double e1 = energy_j(), t1 = now();
    const double w = (e1 - e0) / (t1 - t0), rate = launches * units_per_launch / (t1 - t0);
    if (g_base_w == 0.0) { g_base_w = w; g_base_rate = rate; }
    printf("%-34s %7.1f W  %9.3f G%s/s  %8.3f nJ/%s  above s_nop power: %8.3f nJ/%s\n", name, w, rate * 1e-9, unit, w / rate * 1e9, unit,
           (w - g_base_w) / rate * 1e9, unit);
    (void)instr_per_unit;
}

template <int KIND>
static void run(const char *name, float *d, int per_block = 1)
{
    const int iters = 4000;
    measure(name, 256.0 * 4 * 4 * (double)iters * REP * per_block, "winstr",
            [&] { hipLaunchKernelGGL(k<KIND>, dim3(256 * 4), dim3(256), 0, 0, d, iters); });
}

int main()
{
    if (rsmi_init(0) != RSMI_STATUS_SUCCESS || energy_j() < 0) { printf("no energy counter\n"); return 1; }
    float *d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("s_nop 0 (baseline)", d);
    run<1>("v_fma_f32", d);
    run<19>("v_fma_f32 x2 (per pair)", d);
    run<2>("v_pk_fma_f32", d);
    run<3>("v_fmac_f32", d);
    run<4>("v_mul_f32", d);
    run<5>("v_sub_f32", d);
    run<16>("v_pk_add_f32", d);
    run<17>("v_pk_mul_f32", d);
    run<6>("v_cvt_f32_i32", d);
    run<7>("v_cmp + v_cndmask (per pair)", d);
    run<8>("v_sub_u32", d);
    run<9>("v_lshrrev_b32", d);
    run<10>("v_mov_b32", d);
    run<11>("v_perm_b32", d);
    run<15>("v_sqrt_f32", d);
    run<18>("ds_read_b32", d);
    run<20>("ds_read2_b32", d);
    run<12>("ds_read_b64", d);
    run<13>("ds_read_b128", d);
    run<14>("ds_write_b64", d);
    const size_t n = (size_t)1 << 28;                            // 4 GiB of float4
    f4 *src, *dst; hipMalloc(&src, n * 16); hipMalloc(&dst, n * 16);
    hipMemset(src, 1, n * 16);
    measure("HBM read (nontemporal, 4 GiB)", (double)n * 16, "B", [&] { hipLaunchKernelGGL(k_read, dim3(256 * 8), dim3(256), 0, 0, src, d, n); });
    measure("HBM copy (4 GiB in + 4 GiB out)", (double)n * 32, "B", [&] { hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, src, dst, n); });
    return 0;
}
