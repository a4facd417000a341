// What the HBM of one MI355X sustains for the read : write mixes of this repo's kernels (16 bytes per lane and instruction,
// 1 KB contiguous per wave instruction, buffers far beyond the 256 MB Infinity Cache):
//   read only, write only (plain and streaming stores), copy 1:1, 2:1 (audio stage), 1:2 (spectrum_db2col), 1:8 (play_buffer).
//   hipcc --offload-arch=gfx950 -O3 hbm_stream.hip -o hbm_stream && ./hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// R reads and W writes of 16 B per lane and trip; every wave walks its own contiguous 1 KB pieces, grid-strided
template <int R, int W, bool NT, int U>
__global__ __launch_bounds__(256) void stream(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, uint64_t trips)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n = (uint64_t)gridDim.x * blockDim.x;
    u32x4 keep = {0, 0, 0, 0};
    for (uint64_t t = tid; t < trips * n; t += U * n) {          // U trips in flight per lane (trips is a multiple of U)
        u32x4 acc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc[u] = u32x4{(uint32_t)t, 1, 2, 3};
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u32x4 *p = src + (uint64_t)r * trips * n + t + (uint64_t)u * n;
                acc[u] += NT ? __builtin_nontemporal_load(p) : *p;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int w = 0; w < W; w++) {
                u32x4 v = acc[u]; v.x += w;
                u32x4 *p = dst + (uint64_t)w * trips * n + t + (uint64_t)u * n;
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            }
            keep += acc[u];
        }
    }
    if (W == 0 && keep.x == 0x12345678u) dst[tid] = keep;
}

template <int R, int W, bool NT, int U = 4>
static void run(const char *name, const u32x4 *src, u32x4 *dst, uint64_t total_bytes, int blocks)
{
    const uint64_t n = (uint64_t)blocks * 256;
    const uint64_t trips = total_bytes / ((uint64_t)(R + W) * 16 * n) / U * U;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((stream<R, W, NT, U>), dim3(blocks), dim3(256), 0, 0, src, dst, trips);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((stream<R, W, NT, U>), dim3(blocks), dim3(256), 0, 0, src, dst, trips);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)trips * n * 16 * (R + W);
    printf("%-34s U %d grid %6d  %7.3f ms  %6.0f GB/s  (read %5.0f, write %5.0f)\n", name, U, blocks, ms / reps, bytes * reps / ms / 1e6,
           bytes * R / (R + W) * reps / ms / 1e6, bytes * W / (R + W) * reps / ms / 1e6);
}

int main()
{
    const uint64_t half = 12ull << 30;                           // 12 GB each way
    u32x4 *src, *dst;
    if (hipMalloc(&src, half) != hipSuccess || hipMalloc(&dst, half) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, half); hipMemset(dst, 2, half);
    // spin-up: the clocks of an idle board need ~0.5 s of load
    for (int i = 0; i < 40; i++) hipLaunchKernelGGL((stream<1, 1, true, 1>), dim3(4096), dim3(256), 0, 0, src, dst, (uint64_t)256);
    hipDeviceSynchronize();
    const uint64_t T = 8ull << 30;                               // bytes moved per launch
    for (int blocks : {2048, 8192, 65536}) {
        run<1, 0, false>("read only", src, dst, T, blocks);
        run<1, 0, true>("read only, streaming loads", src, dst, T, blocks);
        run<0, 1, false>("write only", src, dst, T, blocks);
        run<0, 1, true>("write only, streaming stores", src, dst, T, blocks);
        run<1, 1, true>("copy 1:1, streaming", src, dst, T, blocks);
        run<1, 1, false>("copy 1:1, plain", src, dst, T, blocks);
        run<2, 1, true>("2 read : 1 write, streaming", src, dst, T, blocks);
        run<1, 2, true>("1 read : 2 write, streaming", src, dst, T, blocks);
        run<1, 8, true>("1 read : 8 write, streaming", src, dst, 9ull << 30, blocks);
        run<1, 1, true, 1>("copy 1:1, streaming", src, dst, T, blocks);
        run<1, 1, true, 8>("copy 1:1, streaming", src, dst, T, blocks);
    }
    return 0;
}
