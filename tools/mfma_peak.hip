// Development probe: the matrix-core rate this chip sustains from registers alone (no LDS, no memory), as a ceiling
// for the GEMM kernels.  W waves per SIMD, ACC independent 16x16 accumulators per wave, v_mfma_f32_16x16x32_bf16.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip ; run: /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int ACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters)
{
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < ACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int ACC>
__global__ __launch_bounds__(256) void mfma32_loop(float* out, int iters)
{
    f32x16 acc[ACC];
    for (int i = 0; i < ACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < ACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ACC> void run32(int wgs_per_cu, float* out)
{
    const int iters = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mfma32_loop<ACC><<<grid, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma32_loop<ACC><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * ACC * 32 * 32 * 16 * 2;
    printf("32x32x16: waves/SIMD %d, %d accumulators: %.0f TFLOP/s (%.2f ms)\n", wgs_per_cu, ACC, fl / (ms * 1e-3) / 1e12, ms);
}

template <int ACC> void run(int wgs_per_cu, float* out)
{
    const int iters = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<ACC><<<grid, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<ACC><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * ACC * 16 * 16 * 32 * 2;
    printf("waves/SIMD %d, %d accumulators: %.0f TFLOP/s (%.2f ms)\n", wgs_per_cu, ACC, fl / (ms * 1e-3) / 1e12, ms);
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<4>(1, out); run<16>(1, out); run<4>(2, out); run<16>(2, out); run<16>(4, out); run<8>(4, out); run<4>(4, out); run<2>(4, out); run<1>(4, out);
    run32<1>(4, out); run32<2>(4, out); run32<4>(4, out); run32<1>(2, out); run32<2>(2, out); run32<4>(2, out); run32<4>(1, out);
    return 0;
}
