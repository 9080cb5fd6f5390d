// Development probe: what a CU takes in through global_load_lds_dwordx4 for the access shapes a GEMM tile loop can use.
// Two 512-thread workgroups per CU stream "tiles" of a [rows][K] bf16 matrix (rows 64-byte or 128-byte K segments, or
// pre-blocked 1 KiB pieces) into a 3-slot LDS ring with counted waits, nothing else.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iwhisper_amd/csrc -o tools/_build/dma_probe tools/dma_probe.hip
#include "uw_common.h"
#include <stdio.h>
#include <stdlib.h>
void uw_set_error(const char*, ...) {}

// mode 0: 16 rows x 64 B per wave instruction (row pitch = K bytes); mode 1: 8 rows x 128 B; mode 2: 1 KiB contiguous
template <int MODE>
__global__ __launch_bounds__(512) void dma_kernel(const char* src, long row_bytes, int rows_per_wg, int steps, int pieces, int share)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = lds_addr_of(smem) + wave * 1024;
    const char* base = src + (long)(blockIdx.x % share) * rows_per_wg * row_bytes;   // share = 1: every workgroup reads the same rows (L2 hits)
    // a step moves `pieces` wave instructions per wave = pieces * 8 KiB per workgroup
    for (int st = 0; st < steps; ++st) {
        for (int i = 0; i < pieces; ++i) {
            const int L = i * 512 + tid;
            unsigned off;
            if (MODE == 0) { const int row = L >> 2, c = L & 3; off = (unsigned)(row * row_bytes + st * 64 + c * 16); }
            else if (MODE == 1) { const int row = L >> 3, c = L & 7; off = (unsigned)(row * row_bytes + st * 128 + c * 16); }
            else { off = (unsigned)((long)st * pieces * 8192 + L * 16); }
            glds16_sv(base, off, lds_base + ((st % 3) * pieces + i) * 8192);
        }
        if (st >= 2) {
            if (pieces == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (smem[tid * 16] == 123) ((volatile char*)src)[0] = 1;
}

int main()
{
    const long K = 3072, row_bytes = K * 2;
    const int wgs = 512, rows64 = 384, iters = 20;
    char* buf;
    const size_t bytes = (size_t)wgs * rows64 * row_bytes;       // every workgroup its own rows: 1.2 GB
    if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int share : {1, 8, 512})
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: 384 rows x 64 B per step (3 pieces), K/32 steps; mode 1: 384 rows x 128 B per step (6 pieces), K/64 steps
        const int pieces = mode == 1 ? 6 : 3, steps = mode == 1 ? (int)(K / 64) : (int)(K / 32);
        const size_t lds = (size_t)3 * pieces * 8192;
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(dma_kernel<0>, dim3(wgs), dim3(512), lds, 0, buf, row_bytes, rows64, steps, pieces, share);
            else if (mode == 1) hipLaunchKernelGGL(dma_kernel<1>, dim3(wgs), dim3(512), lds, 0, buf, row_bytes, rows64, steps, pieces, share);
            else hipLaunchKernelGGL(dma_kernel<2>, dim3(wgs), dim3(512), lds, 0, buf, row_bytes, rows64, steps, pieces, share);
        };
        if (mode == 1) hipFuncSetAttribute((const void*)dma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (mode == 0) hipFuncSetAttribute((const void*)dma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (mode == 2) hipFuncSetAttribute((const void*)dma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        const double moved = (double)wgs * steps * pieces * 8192 * iters;
        printf("distinct sources %d, mode %d (%s): %.2f TB/s chip, %.1f GB/s per CU  [%s]\n", share, mode,
               mode == 0 ? "16 rows x 64 B" : mode == 1 ? "8 rows x 128 B" : "1 KiB contiguous", moved / (ms * 1e-3) / 1e12,
               moved / (ms * 1e-3) / 1e9 / 256, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
