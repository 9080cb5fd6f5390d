// Development micro-benchmark of the encoder-sized GEMM shapes on random operands (zero-filled operands let the chip
// hold a higher clock and flatter every variant).  UW_GEMM_BIG selects the kernel (2: 256x128 two workgroups per CU,
// 3: 256x256).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iwhisper_amd/csrc tools/gemm_bench.hip -o tools/_build/gemm_bench
#include "../whisper_amd/csrc/uw_gemm.hip"
#include <stdarg.h>
#include <vector>
void uw_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static void fill_bf16(std::vector<uint16_t>& v, uint32_t seed, float scale)
{
    uint32_t s = seed;
    for (auto& x : v) {
        s = s * 1664525u + 1013904223u;
        const float f = ((int)(s >> 8) % 2001 - 1000) * 1e-3f * scale;
        uint32_t u; memcpy(&u, &f, 4);
        x = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
}
int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 192000;
    struct Shape { int K, N, out_f32, act, res; const char* name; } shapes[] = {
        {768, 3072, 0, 1, 0, "fc1"}, {768, 1536, 0, 0, 0, "qk"}, {768, 768, 1, 0, 1, "out+res"}, {3072, 768, 1, 0, 1, "fc2"}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total_us = 0;
    for (auto& sh : shapes) {
        char *W, *A; float *C, *bias;
        CK(hipMalloc(&W, (size_t)sh.N * sh.K * 2)); CK(hipMalloc(&A, (size_t)M * sh.K * 2)); CK(hipMalloc(&C, (size_t)M * sh.N * 4)); CK(hipMalloc(&bias, sh.N * 4));
        std::vector<uint16_t> hw((size_t)sh.N * sh.K), ha((size_t)M * sh.K);
        fill_bf16(hw, 1, 0.03f); fill_bf16(ha, 2, 1.0f);
        CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(C, 0, (size_t)M * sh.N * 4)); CK(hipMemset(bias, 0, sh.N * 4));
        GemmParams p; p.A = A; p.B = W; p.C = C; p.bias = bias; p.lda = sh.K; p.ldb = sh.K; p.ldc = sh.N; p.ldr = sh.N; p.M = M; p.N = sh.N; p.K = sh.K;
        if (getenv("KBLOCK")) { const int kb = atoi(getenv("KBLOCK")); if (kb & 1) { p.lda = 32; p.kstep_a = (long)M * 32; } if (kb & 2) { p.ldb = 32; p.kstep_b = (long)sh.N * 32; } }
        p.bias_mode = 1; p.act = getenv("NOACT") ? 0 : sh.act;
        if (getenv("DMA_MODE")) p.act |= atoi(getenv("DMA_MODE"));   // -DUW_TUNE builds: 8 = B pieces only (x3), 16 = one A piece (x6) p.residual = sh.res ? C : nullptr;
        const int iters = 6;
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) if (uw_gemm(UW_BF16, sh.out_f32, p, s)) return 1;
        }
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / iters;
        total_us += us;
        printf("%-8s M=%d K=%d N=%d  %.1f us  %.0f TFLOP/s\n", sh.name, M, sh.K, sh.N, us, 2.0 * M * sh.N * sh.K / us * 1e-6);
        CK(hipFree(W)); CK(hipFree(A)); CK(hipFree(C)); CK(hipFree(bias));
    }
    printf("sum %.1f us (x12 layers = %.1f ms)\n", total_us, total_us * 12e-3);
    return 0;
}
