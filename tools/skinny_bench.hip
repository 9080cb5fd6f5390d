// Development micro-benchmark of the decode-step products as the step launches them (fragment-major operands, split
// activations, LayerNorm-fused forms) with cold weights (rotating copies larger than the Infinity Cache).
// UW_SK_TILE / UW_LN_TILE (e.g. 21 = 2 row tiles x 1 column tile per workgroup) override the tile shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DUW_TUNE -Iwhisper_amd/csrc tools/skinny_bench.hip -o tools/_build/skinny_bench
#include "../whisper_amd/csrc/uw_gemm.hip"
#include <stdarg.h>
#include <vector>
void uw_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// SK_TOUCH=1: before every product a kernel touches one dword per 128-byte line of the weights it is about to read, workgroup c
// the block workgroup c of the product will read (same XCD); SK_TOUCH=2: the block of workgroup c + 1 (another XCD, so the
// lines can only be found in the memory-side cache).  Read the product's duration from a kernel trace.
__global__ void touch_kernel(const char* w, int block_bytes, int blocks, int shift, int never, unsigned* sink)
{
    const int c = (blockIdx.x + shift) % blocks;
    unsigned v = 0;
    for (int l = threadIdx.x; l * 128 < block_bytes; l += blockDim.x) v |= *(const unsigned*)(w + (size_t)c * block_bytes + (size_t)l * 128);
    if (never) sink[threadIdx.x] = v;
}

int main(int argc, char** argv)
{
    const int touch = getenv("SK_TOUCH") ? atoi(getenv("SK_TOUCH")) : 0;
    const int touch_nt = getenv("SK_TOUCH_NT") ? atoi(getenv("SK_TOUCH_NT")) : 2;   // column tiles per workgroup of the product
    const int M = argc > 1 ? atoi(argv[1]) : 128;
    const int d = argc > 2 ? atoi(argv[2]) : 768;
    struct Shape { int K, N, ln, act, res; const char* name; } shapes[] = {
        {d, 3 * d, 1, 0, 0, "qkv (LN)"}, {d, d, 0, 0, 1, "out+res (hi/lo)"}, {d, d, 1, 0, 0, "crossq (LN)"}, {d, 4 * d, 1, 1, 0, "fc1 (LN, hi/lo out)"},
        {4 * d, d, 0, 0, 1, "fc2 (hi/lo)"}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total = 0;
    for (auto& sh : shapes) {
        const size_t wbytes = (size_t)sh.N * sh.K * 2;
        const int copies = getenv("SK_COPIES") ? atoi(getenv("SK_COPIES")) : (int)((600u << 20) / wbytes) + 1;   // SK_COPIES=1: warm weights
        char *W, *A, *A2; float *C, *bias; char* C2;
        const size_t abytes = (size_t)((M + 15) & ~15) * sh.K * 4;
        CK(hipMalloc(&W, wbytes * copies)); CK(hipMalloc(&A, abytes)); CK(hipMalloc(&A2, abytes)); CK(hipMalloc(&C, (size_t)128 * sh.N * 4)); CK(hipMalloc(&C2, (size_t)128 * sh.N * 4));
        CK(hipMalloc(&bias, sh.N * 4 + sh.K * 4));
        CK(hipMemset(W, 0x11, wbytes * copies)); CK(hipMemset(A, 0x11, abytes)); CK(hipMemset(A2, 0x11, abytes)); CK(hipMemset(C, 0, (size_t)128 * sh.N * 4)); CK(hipMemset(bias, 0, sh.N * 4 + sh.K * 4));
        GemmParams p; p.A = A; p.C = C; p.bias = bias; p.ldc = sh.N; p.ldr = sh.N; p.M = M; p.N = sh.N; p.K = sh.K;
        p.bias_mode = 1; p.act = sh.act; p.ab_packed = 1;
        int out_f32 = 0;
        if (sh.ln) { p.ln_fused = 1; p.wsum = bias; p.gamma = bias + sh.N; p.c_packed = sh.act ? 1 : 0; p.C_lo = sh.act ? C2 : nullptr; }
        else { p.A_lo = A2; p.c_xlayout = 1; p.residual = C; out_f32 = 1; }
        const int iters = 400;
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) {
                p.B = W + (size_t)(i % copies) * wbytes;
                if (getenv("SK_PF")) {   // the launch requests the weights of the next one (GemmParams::pf)
                    p.pf = W + (size_t)((i + 1) % copies) * wbytes; p.pf_n = sh.N; p.pf_k = sh.K;
                    p.pf_cols = uw_gemm_skinny_block_columns(UW_BF16, M, sh.N, sh.ln, sh.ln ? 0 : 1);
                }
                if (touch) {
                    const int bb = touch_nt * 16 * sh.K * 2, blocks = sh.N / (touch_nt * 16);
                    hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(512), 0, s, (const char*)p.B, bb, blocks, touch == 2 ? 1 : 0, 0, (unsigned*)C2);
                }
                if (uw_gemm_skinny(UW_BF16, out_f32, p, s)) return 1;
            }
        }
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-22s M=%d K=%d N=%d  %.2f us/launch\n", sh.name, M, sh.K, sh.N, 1e3 * ms / iters);
        total += 1e3 * ms / iters;
        CK(hipFree(W)); CK(hipFree(A)); CK(hipFree(A2)); CK(hipFree(C)); CK(hipFree(C2)); CK(hipFree(bias));
    }
    printf("layer total (out+res twice): %.2f us\n", total + 0);
    return 0;
}
