// Development micro-benchmark of the encoder-sized GEMM shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iwhisper_amd/csrc tools/gemm_bench.hip -o tools/_build/gemm_bench
#include "../whisper_amd/csrc/uw_gemm.hip"
#include <stdarg.h>
void uw_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 192000;
    struct Shape { int K, N, out_f32, act, res; const char* name; } shapes[] = {
        {768, 3072, 0, 1, 0, "fc1"}, {768, 1536, 0, 0, 0, "qk"}, {768, 768, 1, 0, 1, "out+res"}, {3072, 768, 1, 0, 1, "fc2"}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& sh : shapes) {
        char *W, *A; float *C, *bias;
        CK(hipMalloc(&W, (size_t)sh.N * sh.K * 2)); CK(hipMalloc(&A, (size_t)M * sh.K * 2)); CK(hipMalloc(&C, (size_t)M * sh.N * 4)); CK(hipMalloc(&bias, sh.N * 4));
        CK(hipMemset(W, 0, (size_t)sh.N * sh.K * 2)); CK(hipMemset(A, 0, (size_t)M * sh.K * 2)); CK(hipMemset(C, 0, (size_t)M * sh.N * 4)); CK(hipMemset(bias, 0, sh.N * 4));
        GemmParams p; p.A = A; p.B = W; p.C = C; p.bias = bias; p.lda = sh.K; p.ldb = sh.K; p.ldc = sh.N; p.ldr = sh.N; p.M = M; p.N = sh.N; p.K = sh.K;
        p.bias_mode = 1; p.act = getenv("NOACT") ? 0 : sh.act; p.residual = sh.res ? C : nullptr;
        const int iters = 5;
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) if (uw_gemm(UW_BF16, sh.out_f32, p, s)) return 1;
        }
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / iters;
        printf("%-8s M=%d K=%d N=%d  %.1f us  %.0f TFLOP/s\n", sh.name, M, sh.K, sh.N, us, 2.0 * M * sh.N * sh.K / us * 1e-6);
        CK(hipFree(W)); CK(hipFree(A)); CK(hipFree(C)); CK(hipFree(bias));
    }
    return 0;
}
