// Development probe: HBM write rate of the GEMM epilogue's store pattern versus a contiguous stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;
// tile pattern: WG = 256 rows x (128 cols bf16 = 256 B); wave (wr, wc) stores rows wr*64.., 128 B per row per instruction
template <int NT>
__global__ __launch_bounds__(512) void tile_store(char* C, int tiles_n, long pitch)
{
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const f32x4 v = {1.f, 2.f, 3.f, (float)tid};
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const long row = (long)tm * 256 + wr * 64 + it * 8 + (lane >> 3);
        char* p = C + row * pitch + tn * 256 + wc * 128 + (lane & 7) * 16;
        if (NT) __builtin_nontemporal_store(v, (f32x4*)p); else *(f32x4*)p = v;
    }
}
// same, but a wave stores full 256-B tile rows (4 rows per instruction)
__global__ __launch_bounds__(512) void tile_store_wide(char* C, int tiles_n, long pitch)
{
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const f32x4 v = {1.f, 2.f, 3.f, (float)tid};
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const long row = (long)tm * 256 + wave * 32 + it * 4 + (lane >> 4);
        char* p = C + row * pitch + tn * 256 + (lane & 15) * 16;
        *(f32x4*)p = v;
    }
}
__global__ __launch_bounds__(512) void flat_store(char* C, long n16)
{
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n16; i += (long)gridDim.x * 512) ((f32x4*)C)[i] = v;
}
int main()
{
    const long M = 192000, N = 3072, pitch = N * 2, bytes = M * pitch;
    char* C; CK(hipMalloc(&C, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int tiles_n = N / 128, tiles = (M / 256) * tiles_n;
    for (int k = 0; k < 4; ++k) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) {
                if (k == 0) hipLaunchKernelGGL(tile_store<0>, dim3(tiles), dim3(512), 0, 0, C, tiles_n, pitch);
                if (k == 1) hipLaunchKernelGGL(tile_store<1>, dim3(tiles), dim3(512), 0, 0, C, tiles_n, pitch);
                if (k == 2) hipLaunchKernelGGL(tile_store_wide, dim3(tiles), dim3(512), 0, 0, C, tiles_n, pitch);
                if (k == 3) hipLaunchKernelGGL(flat_store, dim3(4096), dim3(512), 0, 0, C, bytes / 16);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const char* names[] = {"tile 8x128B", "tile 8x128B nt", "tile 4x256B", "flat"};
        printf("%-16s %.1f us  %.2f TB/s\n", names[k], 1e3 * ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}
