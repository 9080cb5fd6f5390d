// Development micro-benchmark of the decode-step GEMM shapes with cold weights (rotating copies larger than the
// Infinity Cache).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iwhisper_amd/csrc tools/skinny_bench.hip -o tools/_build/skinny_bench
#include "../whisper_amd/csrc/uw_gemm.hip"
#include <stdarg.h>
#include <vector>
void uw_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 128;
    struct Shape { int K, N, out_f32, act, res; const char* name; } shapes[] = {
        {768, 2304, 0, 0, 0, "qkv"}, {768, 768, 1, 0, 1, "out+res"}, {768, 768, 0, 0, 0, "crossq"}, {768, 3072, 0, 1, 0, "fc1"}, {3072, 768, 1, 0, 1, "fc2"}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& sh : shapes) {
        const size_t wbytes = (size_t)sh.N * sh.K * 2;
        const int copies = (int)((600u << 20) / wbytes) + 1;
        char *W, *A; float *C, *bias;
        CK(hipMalloc(&W, wbytes * copies)); CK(hipMalloc(&A, (size_t)M * sh.K * 4)); CK(hipMalloc(&C, (size_t)M * sh.N * 4)); CK(hipMalloc(&bias, sh.N * 4));
        CK(hipMemset(W, 0, wbytes * copies)); CK(hipMemset(A, 0, (size_t)M * sh.K * 4)); CK(hipMemset(C, 0, (size_t)M * sh.N * 4)); CK(hipMemset(bias, 0, sh.N * 4));
        GemmParams p; p.A = A; p.C = C; p.bias = bias; p.lda = sh.K; p.ldb = sh.K; p.ldc = sh.N; p.ldr = sh.N; p.M = M; p.N = sh.N; p.K = sh.K;
        p.bias_mode = 1; p.act = sh.act; p.ab_packed = getenv("PACKED") ? 1 : 0; p.c_packed = (p.ab_packed && sh.act) ? 1 : 0;
        if (getenv("LN") && !sh.res) { p.ln_fused = 1; p.wsum = bias; p.ab_packed = 1; }
        if (p.ab_packed && sh.res) p.c_xlayout = 1; p.residual = sh.res ? C : nullptr;
        const int iters = 400;
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) {
                p.B = W + (size_t)(i % copies) * wbytes;
                if (uw_gemm_skinny(UW_BF16, sh.out_f32, p, s)) return 1;
            }
        }
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-8s M=%d K=%d N=%d  %.2f us/launch (cold W, %d copies)\n", sh.name, M, sh.K, sh.N, 1e3 * ms / iters, copies);
        CK(hipFree(W)); CK(hipFree(A)); CK(hipFree(C)); CK(hipFree(bias));
    }
    return 0;
}
