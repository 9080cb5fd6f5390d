// C[m][n] = act(sum_k A[m][k] * B[n][k] + bias) (+ residual), both operands K-contiguous ("TN").
//
// One kernel serves every dense contraction of the transcription path: the conv stem (as a strided-row
// GEMM over a time-major, zero-padded activation buffer), the encoder/decoder projections and MLPs, the
// cross-attention K/V projection and the tied logits product.  Element type T is bf16 (MFMA 16x16x32) or
// f32 (MFMA 16x16x4, exact fp32 FMA chain) -- the two share all staging code because operands are moved
// and addressed as 16-byte K-chunks.
//
// Tile: 128 x 128 x (128 bytes of K) per 256-thread workgroup, 2 x 2 waves of 64 x 64.  Staging is
// global_load_lds (16 B/lane) into a double-buffered, XOR-swizzled LDS image: the LDS side of the DMA is
// lane-linear, so the swizzle is applied to each lane's SOURCE chunk and again on the ds_read_b128.
// Accumulators hold C transposed (MFMA "A" operand = rows of B), so each lane owns 4 consecutive n of one
// row m and the epilogue stores 8/16 contiguous bytes per lane.
#include "uw_common.h"
#include "uw_kernels.h"

namespace {

constexpr int BM = 128, BN = 128, KB = 128;  // KB: bytes of K per tile row
constexpr int TILE_BYTES = BM * KB;          // 16 KiB per operand per buffer

template <typename T, typename TC>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int KT = KB / (int)sizeof(T);  // K elements per tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile = blockIdx.x;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const long bz = blockIdx.y;
    const T* __restrict__ A = (const T*)p.A + bz * p.sA;
    const T* __restrict__ B = (const T*)p.B + bz * p.sB;

    // per-thread staging sources: 4 chunks of A and 4 of B per K-tile
    const T* srcA[4];
    const T* srcB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int L = i * 256 + tid, row = L >> 3, c = (L & 7) ^ (row & 7);
        const int gm = min(m0 + row, p.M - 1), gn = min(n0 + row, p.N - 1);
        srcA[i] = A + (long)gm * p.lda + c * E;
        srcB[i] = B + (long)gn * p.ldb + c * E;
    }
    auto stage = [&](int kt, int buf) {
        char* a_dst = smem + buf * 2 * TILE_BYTES + wave * 1024;
        char* b_dst = a_dst + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(srcA[i] + (long)kt * KT), LDS_PTR(a_dst + i * 4096), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(srcB[i] + (long)kt * KT), LDS_PTR(b_dst + i * 4096), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / KT;
    const int frow = lane & 15, fg = lane >> 4;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* As = smem + (kt & 1) * 2 * TILE_BYTES;
        const char* Bs = As + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pos = ((ks * 4 + fg) ^ (frow & 7)) << 4;
            uint4 af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *(const uint4*)(As + (wr * 64 + t * 16 + frow) * KB + pos);
                bf[t] = *(const uint4*)(Bs + (wc * 64 + t * 16 + frow) * KB + pos);
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) mma16<T>(bf[tn], af[tm], acc[tm][tn]);
        }
    }

    // epilogue: lane holds C[m][n..n+3] per (tm, tn)
    TC* __restrict__ C = (TC*)p.C + bz * p.sC;
    const float* __restrict__ R = p.residual ? p.residual + bz * p.sR : nullptr;
    const bool vec_ok = ((p.ldc & 3) == 0) && (!R || (p.ldr & 3) == 0);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const int m = m0 + wr * 64 + tm * 16 + frow;
        if (m >= p.M) continue;
        const float bias_m = (p.bias_mode == 2) ? p.bias[m] : 0.f;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int n = n0 + wc * 64 + tn * 16 + fg * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
            const bool full = vec_ok && (n + 3 < p.N);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (p.bias_mode == 1) v[r] += (n + r < p.N) ? p.bias[n + r] : 0.f;
                else v[r] += bias_m;
                if (p.act == 1) v[r] = gelu_erf(v[r]);
            }
            if (full) {
                if (R) {
                    const f32x4 rv = *(const f32x4*)(R + (long)m * p.ldr + n);
                    v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
                }
                if constexpr (sizeof(TC) == 4) {
                    *(f32x4*)((float*)C + (long)m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    union { bf16_t h[4]; uint2 u; } pk;
                    pk.h[0] = (bf16_t)v[0]; pk.h[1] = (bf16_t)v[1]; pk.h[2] = (bf16_t)v[2]; pk.h[3] = (bf16_t)v[3];
                    *(uint2*)((bf16_t*)C + (long)m * p.ldc + n) = pk.u;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r >= p.N) break;
                    float x = v[r];
                    if (R) x += R[(long)m * p.ldr + n + r];
                    C[(long)m * p.ldc + n + r] = from_f32<TC>(x);
                }
            }
        }
    }
}

template <typename T, typename TC> int launch(const GemmParams& p, hipStream_t s)
{
    constexpr int KT = KB / (int)sizeof(T);
    UW_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.batch > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    UW_REQUIRE(p.K % KT == 0, "gemm: K=%d must be a multiple of %d", p.K, KT);
    UW_REQUIRE((p.lda * sizeof(T)) % 16 == 0 && (p.ldb * sizeof(T)) % 16 == 0 && (p.sA * sizeof(T)) % 16 == 0 &&
                   (p.sB * sizeof(T)) % 16 == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0,
               "gemm: operand rows must be 16-byte aligned");
    const long tiles = (long)ceil_div(p.M, BM) * ceil_div(p.N, BN);
    UW_REQUIRE(tiles < (1l << 31) && p.batch < 65536, "gemm: grid too large");
    dim3 grid((unsigned)tiles, (unsigned)p.batch);
    hipLaunchKernelGGL((gemm_tn_kernel<T, TC>), grid, dim3(256), 4 * TILE_BYTES, s, p);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

}  // namespace

int uw_gemm(int dtype, int out_f32, const GemmParams& p, hipStream_t s)
{
    if (dtype == UW_F32) return launch<float, float>(p, s);
    if (out_f32) return launch<bf16_t, float>(p, s);
    return launch<bf16_t, bf16_t>(p, s);
}
