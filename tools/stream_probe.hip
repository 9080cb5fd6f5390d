// Development probe: how fast the chip reads a buffer far larger than its caches, by load path and bytes in flight.
// Every workgroup streams its own contiguous share of a 4 GiB buffer once per launch and keeps nothing.
//   path 0: 16-byte non-temporal loads into registers, U in flight per lane      (what the attention decode kernels do)
//   path 1: the same loads without the non-temporal hint
//   path 2: global_load_lds_dwordx4 into an LDS ring, U pieces in flight per wave (what the tile GEMM's operand stream does)
//   path 3: both at once: even waves take path 0, odd waves path 2
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iwhisper_amd/csrc -o tools/_build/stream_probe tools/stream_probe.hip
#include "uw_common.h"
#include <stdio.h>
#include <stdlib.h>
void uw_set_error(const char*, ...) {}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PATH, int U>
__global__ __launch_bounds__(1024) void stream_kernel(const char* src, long bytes_per_wg, unsigned* sink, int never)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_waves = blockDim.x >> 6;
    const char* base = src + (long)blockIdx.x * bytes_per_wg;
    const long per_round = (long)blockDim.x * 16 * U;          // bytes the workgroup requests per round
    unsigned acc = 0;
    const bool dma = PATH == 2 || (PATH == 3 && (wave & 1));
    if (!dma) {
        for (long off = 0; off + per_round <= bytes_per_wg; off += per_round) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32x4* ptr = (const u32x4*)(base + off + ((long)u * blockDim.x + tid) * 16);
                v[u] = PATH == 1 ? *ptr : __builtin_nontemporal_load(ptr);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].w;
        }
    } else {
        const unsigned lds_base = lds_addr_of(smem) + wave * (U * 1024);
        long off = 0;
        for (; off + per_round <= bytes_per_wg; off += per_round) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                glds16_sv(base + off, (unsigned)(((long)u * blockDim.x + wave * 64) * 16 + lane * 16), lds_base + u * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a wave's own pieces; the other waves keep the CU's queues full
        }
        acc = smem[tid];
    }
    if (never) sink[blockIdx.x * blockDim.x + tid] = acc + n_waves;
}

template <int PATH, int U> float run(const char* buf, long total, int wgs, int threads, unsigned* sink)
{
    const long per_wg = (total / wgs) & ~((long)threads * 16 * U - 1);
    const size_t lds = (size_t)(threads / 64) * U * 1024;
    hipFuncSetAttribute((const void*)stream_kernel<PATH, U>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_kernel<PATH, U>), dim3(wgs), dim3(threads), lds, 0, buf, per_wg, sink, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int iters = 5;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((stream_kernel<PATH, U>), dim3(wgs), dim3(threads), lds, 0, buf, per_wg, sink, 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double tbs = (double)per_wg * wgs * iters / (ms * 1e-3) / 1e12;
    printf("path %d  U=%2d  %4d workgroups x %4d threads  (%5.1f MB in flight)  %.2f TB/s  [%s]\n", PATH, U, wgs, threads,
           (double)wgs * threads * 16 * U / 1e6, tbs, hipGetErrorString(hipGetLastError()));
    return (float)tbs;
}

int main()
{
    const long total = 4l << 30;
    char* buf; unsigned* sink;
    if (hipMalloc(&buf, total) != hipSuccess || hipMalloc(&sink, 64 << 20) != hipSuccess) return 1;
    hipMemset(buf, 1, total);
    for (int wgs : {32, 64, 128, 192, 256, 512, 1024, 2048}) {   // up to 256: one workgroup per CU on that many CUs
        for (int threads : {256, 512, 1024}) {
            if ((long)wgs * threads > 1024 * 1024) continue;
            run<0, 4>(buf, total, wgs, threads, sink);
            run<0, 8>(buf, total, wgs, threads, sink);
            run<0, 16>(buf, total, wgs, threads, sink);
            run<1, 8>(buf, total, wgs, threads, sink);
            run<2, 8>(buf, total, wgs, threads, sink);
            run<3, 8>(buf, total, wgs, threads, sink);
        }
    }
    return 0;
}
