// Multi-query attention for the encoder (1500 x 1500, no mask), the decoder's multi-token passes (causal
// self-attention, cross-attention over the 1500 encoder frames) and the softmax-probability rows that word
// alignment reads from the selected cross-attention heads.  head_dim is 64 for every Whisper size.
//
// Layout of one workgroup: 4 waves x 32 queries, K / V^T tiles of 64 keys double-buffered in LDS through
// global_load_lds with the XOR swizzle applied to the source chunk (the DMA's LDS side is lane-linear).
// Scores are produced TRANSPOSED, S^T = K Q^T with the 32x32 MFMA, so a lane owns one query column: the
// row maximum / sum are register reductions plus one cross-half shuffle, the rescale factor is lane-local
// for O^T = V^T P^T as well, and the P accumulators feed the second MFMA as its B operand without leaving
// registers.  K rows are fed to the MFMA with bits 2 and 3 of the row index swapped, which makes the eight
// P values a lane holds per 16-key step eight CONSECUTIVE keys -- so the matching V^T operand is one
// ordinary 16-byte K-chunk.  bf16 and f32 (exact-fp32 MFMA) share the code via 16-byte chunks.
#include "uw_common.h"
#include "uw_kernels.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

template <typename T> struct Lay {
    static constexpr int E = 16 / (int)sizeof(T);  // elements per chunk
    static constexpr int CPR = 64 / E;             // chunks per 64-element row
    static constexpr int RB = 64 * (int)sizeof(T); // row bytes
    static constexpr int TILE = 64 * RB;           // one 64-row tile
    static constexpr int NQ = CPR / 2;             // mma32 calls over head_dim
    static constexpr int CH = TILE / 16 / 256;     // chunks staged per thread per tile
    __device__ static __forceinline__ int swz(int row) { return CPR == 8 ? ((row >> 1) & 7) : (row & 15); }
};

// source pointers of this thread's chunks of a [64 rows][64 elements] tile whose rows are `ld` apart
template <typename T>
__device__ __forceinline__ const char* lds_chunk(const char* tile, int row, int chunk)
{
    return tile + row * Lay<T>::RB + ((chunk ^ Lay<T>::swz(row)) << 4);
}

__device__ __forceinline__ int key_of(int sub, int rho, int hl) { return sub * 32 + 16 * (rho >> 3) + 8 * hl + (rho & 7); }
// MFMA row i of the S^T tile reads K row pi(i): bits 2 and 3 swapped
__device__ __forceinline__ int pi_row(int i) { return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <typename T> __device__ __forceinline__ uint4 pack_p(const f32x16& p, int s8);
template <> __device__ __forceinline__ uint4 pack_p<bf16_t>(const f32x16& p, int s8)
{
    // pairs are converted together so that each v_cvt_pk_bf16_f32 fills a whole dword
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    union { bf16x2 h[4]; uint4 u; } pk;
#pragma unroll
    for (int k = 0; k < 4; ++k) pk.h[k] = __builtin_convertvector(f32x2{p[s8 + 2 * k], p[s8 + 2 * k + 1]}, bf16x2);
    return pk.u;
}

template <typename T, bool PROBS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attention_kernel(AttnParams p, ProbParams pp)
{
    using L = Lay<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    int head, Sq, Sk, q_base = 0, q_count;
    const T *Qb, *Kb;
    long ldq, ldk, sKh;
    if constexpr (PROBS) {
        head = pp.heads[blockIdx.y];
        Sq = pp.Sq; Sk = pp.Sk;
        q_base = pp.q_first ? pp.q_first[b] : 0;
        q_count = pp.q_count ? pp.q_count[b] : Sq;
        Qb = (const T*)pp.Q + b * pp.sQ; Kb = (const T*)pp.K + b * pp.sK;
        ldq = pp.ldq; ldk = pp.ldk; sKh = pp.sKh;
    } else {
        head = blockIdx.y;
        Sq = p.Sq; Sk = p.Sk;
        q_count = p.sq_len ? p.sq_len[b] : Sq;
        Qb = (const T*)p.Q + b * p.sQ; Kb = (const T*)p.K + b * p.sK;
        ldq = p.ldq; ldk = p.ldk; sKh = p.sKh;
    }
    const int q0 = blockIdx.x * 128;          // first query of the workgroup, relative to q_base
    if (q0 >= q_count) return;
    const int causal = PROBS ? 0 : p.causal;
    const int coff = Sk - Sq;
    const T* VTb = PROBS ? nullptr : (const T*)p.VT + b * p.sVT + (long)head * 64 * p.ldvt;

    // staging sources
    const T* srcK[L::CH];
    const T* srcV[L::CH];
    int krow[L::CH];
#pragma unroll
    for (int i = 0; i < L::CH; ++i) {
        const int Lc = i * 256 + tid, row = Lc / L::CPR, c = (Lc % L::CPR) ^ L::swz(row);
        krow[i] = row;
        srcK[i] = Kb + head * sKh + c * L::E;
        if constexpr (!PROBS) srcV[i] = VTb + (long)row * p.ldvt + c * L::E;
    }
    const unsigned lds_base = lds_addr_of(smem) + __builtin_amdgcn_readfirstlane(wave) * 1024;
    auto stage = [&](int key0, int buf) {
        const unsigned kd = lds_base + buf * 2 * L::TILE, vd = kd + L::TILE;
#pragma unroll
        for (int i = 0; i < L::CH; ++i) {
            const int kr = min(key0 + krow[i], Sk - 1);
            glds16(srcK[i] + (long)kr * ldk, kd + i * 4096);
            if constexpr (!PROBS) glds16(srcV[i] + key0, vd + i * 4096);
        }
    };

    // this lane's query (column of S^T) and its Q chunks, kept in registers
    const int q_rel = q0 + wave * 32 + l31;
    const int q_abs = q_base + q_rel;
    const bool q_ok = q_rel < q_count && q_abs < Sq;
    uint4 qf[L::NQ];
    {
        const T* qrow = Qb + (long)min(q_abs, Sq - 1) * ldq + head * 64;
#pragma unroll
        for (int i = 0; i < L::NQ; ++i) qf[i] = *(const uint4*)(qrow + (2 * i + hl) * L::E);
    }
    retire_vector_loads();   // otherwise every loop iteration waits on vmcnt for Q, and with it for the DMA just issued

    int n_tiles = (Sk + 63) / 64;
    if (causal) {
        const int last_q = min(q0 + 127, Sq - 1);
        n_tiles = min(n_tiles, (last_q + coff) / 64 + 1);
    }
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 ot[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ot[0][r] = 0.f; ot[1][r] = 0.f; }

    auto scores = [&](const char* Ks, int key0, f32x16 (&st)[2]) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[sub][r] = 0.f;
            const int row = sub * 32 + pi_row(l31);
#pragma unroll
            for (int i = 0; i < L::NQ; ++i) {
                const uint4 a = *(const uint4*)lds_chunk<T>(Ks, row, 2 * i + hl);
                mma32<T>(a, qf[i], st[sub]);
            }
        }
        const bool edge = (key0 + 64 > Sk) || (causal && key0 + 63 > q0 + wave * 32 + coff);
        if (edge) {
            const int lim = causal ? min(Sk - 1, q_abs + coff) : Sk - 1;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + key_of(sub, r, hl) > lim) st[sub][r] = -INFINITY;
        }
    };

    constexpr int PASSES = PROBS ? 2 : 1;
    float inv_l = 0.f;
    for (int pass = 0; pass < PASSES; ++pass) {
        __syncthreads();
        stage(0, 0);
        for (int j = 0; j < n_tiles; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (j + 1 < n_tiles) stage((j + 1) * 64, (j + 1) & 1);
            const char* Ks = smem + (j & 1) * 2 * L::TILE;
            const char* Vs = Ks + L::TILE;
            f32x16 st[2];
            scores(Ks, j * 64, st);
            if (PROBS && pass == 1) {
                // second sweep: normalised probabilities, 8 consecutive keys per (sub, rho >> 3)
                if (q_ok) {
                    float* prow = pp.P + (long)b * pp.sP_b + (long)pp.sel_slot[blockIdx.y] * pp.sP_sel + (long)q_rel * Sk;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const int key = j * 64 + key_of(sub, g * 8, hl);
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                                if (key + k < Sk) prow[key + k] = __builtin_amdgcn_exp2f((st[sub][g * 8 + k] - m_run) * LOG2E) * inv_l;
                        }
                }
                continue;
            }
            // running maximum in raw score units; exponentials as exp2(s * log2e - m * log2e): one FMA per score
            float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, st[0][r]), st[1][r]);   // folds to v_max3_f32
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);  // first tile: exp2(-inf) = 0
            const float neg_m = -m_new * LOG2E;
            // two scores per instruction for the scale-and-shift and for the row sum (v_pk_fma_f32 / v_pk_add_f32): the
            // vector ALU, not the matrix pipe, is what this loop keeps busiest
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            const f32x2 l2e = {LOG2E, LOG2E}, nm2 = {neg_m, neg_m};
            f32x2 rs2 = {0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 t = __builtin_elementwise_fma(f32x2{st[sub][r], st[sub][r + 1]}, l2e, nm2);
                    t = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    st[sub][r] = t[0];
                    st[sub][r + 1] = t[1];
                    rs2 += t;
                }
            const float rs = rs2[0] + rs2[1];
            l_run = l_run * alpha + rs;
            const bool rescale = __any(m_new != m_run);   // wave-uniform: most tiles leave every maximum unchanged
            m_run = m_new;
            if constexpr (!PROBS) {
                if (rescale) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { ot[0][r] *= alpha; ot[1][r] *= alpha; }
                }
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        if constexpr (sizeof(T) == 2) {
                            const uint4 pb = pack_p<bf16_t>(st[sub], s * 8);
                            const int chunk = sub * 4 + 2 * s + hl;
#pragma unroll
                            for (int m = 0; m < 2; ++m) {
                                const uint4 a = *(const uint4*)lds_chunk<T>(Vs, m * 32 + l31, chunk);
                                mma32<T>(a, pb, ot[m]);
                            }
                        } else {
                            const int chunk = sub * 8 + 4 * s + 2 * hl;
                            uint4 pb0, pb1;
                            pb0.x = __float_as_uint(st[sub][s * 8 + 0]); pb0.y = __float_as_uint(st[sub][s * 8 + 1]);
                            pb0.z = __float_as_uint(st[sub][s * 8 + 2]); pb0.w = __float_as_uint(st[sub][s * 8 + 3]);
                            pb1.x = __float_as_uint(st[sub][s * 8 + 4]); pb1.y = __float_as_uint(st[sub][s * 8 + 5]);
                            pb1.z = __float_as_uint(st[sub][s * 8 + 6]); pb1.w = __float_as_uint(st[sub][s * 8 + 7]);
#pragma unroll
                            for (int m = 0; m < 2; ++m) {
                                const uint4 a0 = *(const uint4*)lds_chunk<T>(Vs, m * 32 + l31, chunk);
                                const uint4 a1 = *(const uint4*)lds_chunk<T>(Vs, m * 32 + l31, chunk + 1);
                                mma32<T>(a0, pb0, ot[m]);
                                mma32<T>(a1, pb1, ot[m]);
                            }
                        }
                    }
            }
        }
        if (pass == 0) {
            const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
            inv_l = 1.0f / l_tot;
        }
    }

    if constexpr (!PROBS) {
        if (q_ok) {
            constexpr int KT = 64 / (int)sizeof(T);
            T* orow = (T*)p.O + b * p.sO + (long)q_abs * p.ldo + head * 64;
            const long krow = (long)b * Sq + q_abs;     // K-blocked output: row of the [o_kblock][H * 64] matrix
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int hd = 32 * m + 8 * a + 4 * hl;
                    const float v0 = ot[m][4 * a + 0] * inv_l, v1 = ot[m][4 * a + 1] * inv_l;
                    const float v2 = ot[m][4 * a + 2] * inv_l, v3 = ot[m][4 * a + 3] * inv_l;
                    T* dst = orow + hd;
                    if (p.o_kblock) {
                        const int k = head * 64 + hd;
                        dst = (T*)p.O + ((long)(k / KT) * p.o_kblock + krow) * KT + k % KT;
                    }
                    if constexpr (sizeof(T) == 4) {
                        *(f32x4*)dst = f32x4{v0, v1, v2, v3};
                    } else {
                        union { bf16_t h[4]; uint2 u; } pk;
                        pk.h[0] = (bf16_t)v0; pk.h[1] = (bf16_t)v1; pk.h[2] = (bf16_t)v2; pk.h[3] = (bf16_t)v3;
                        *(uint2*)dst = pk.u;
                        if (p.o_split) {   // split activations: lo half of the row, H * 64 columns further
                            union { bf16_t h[4]; uint2 u; } lo;
                            lo.h[0] = (bf16_t)(v0 - (float)pk.h[0]); lo.h[1] = (bf16_t)(v1 - (float)pk.h[1]);
                            lo.h[2] = (bf16_t)(v2 - (float)pk.h[2]); lo.h[3] = (bf16_t)(v3 - (float)pk.h[3]);
                            *(uint2*)((bf16_t*)orow + p.H * 64 + hd) = lo.u;
                        }
                    }
                }
        }
    }
}

template <typename T> int check_common(const void* Q, const void* K, long ldq, long ldk, long sQ, long sK)
{
    UW_REQUIRE(Q && K, "attention: null pointer");
    UW_REQUIRE(((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0 && (ldq * sizeof(T)) % 16 == 0 &&
                   (ldk * sizeof(T)) % 16 == 0 && (sQ * sizeof(T)) % 16 == 0 && (sK * sizeof(T)) % 16 == 0,
               "attention: operands must be 16-byte aligned");
    return UW_OK;
}

template <typename T> int launch_flash(const AttnParams& p, hipStream_t s)
{
    int rc = check_common<T>(p.Q, p.K, p.ldq, p.ldk, p.sQ, p.sK);
    if (rc) return rc;
    UW_REQUIRE(p.VT && p.O && p.B > 0 && p.H > 0 && p.Sq > 0 && p.Sk > 0, "attention: bad arguments");
    UW_REQUIRE(p.ldvt >= ((p.Sk + 63) / 64) * 64 && (p.ldvt * sizeof(T)) % 16 == 0 && ((uintptr_t)p.VT % 16) == 0 &&
                   (p.sVT * sizeof(T)) % 16 == 0,
               "attention: V^T rows must be padded to a multiple of 64 keys and 16-byte aligned (ldvt=%ld Sk=%d)", p.ldvt, p.Sk);
    UW_REQUIRE(!p.causal || p.Sk >= p.Sq, "attention: causal needs Sk >= Sq");
    UW_REQUIRE((p.ldo * sizeof(T)) % 8 == 0, "attention: output rows must be 8-byte aligned");
    UW_REQUIRE(!p.o_split || (sizeof(T) == 2 && p.ldo >= 2l * p.H * 64), "attention: split output is a bf16 row of 2 * H * 64 elements");
    UW_REQUIRE(!p.o_kblock || (!p.o_split && !p.sq_len && p.o_kblock >= (long)p.B * p.Sq), "attention: K-blocked output covers B * Sq rows, unsplit");
    const dim3 grid((p.Sq + 127) / 128, p.H, p.B);
    hipLaunchKernelGGL((attention_kernel<T, false>), grid, dim3(256), 4 * Lay<T>::TILE, s, p, ProbParams());
    UW_LAUNCH_CHECK();
    return UW_OK;
}

template <typename T> int launch_probs(const ProbParams& pp, hipStream_t s)
{
    int rc = check_common<T>(pp.Q, pp.K, pp.ldq, pp.ldk, pp.sQ, pp.sK);
    if (rc) return rc;
    UW_REQUIRE(pp.P && pp.B > 0 && pp.Sq > 0 && pp.Sk > 0 && pp.n_heads_sel > 0 && pp.n_heads_sel <= 32,
               "attention_probs: bad arguments");
    const dim3 grid((pp.Sq + 127) / 128, pp.n_heads_sel, pp.B);
    hipLaunchKernelGGL((attention_kernel<T, true>), grid, dim3(256), 4 * Lay<T>::TILE, s, AttnParams(), pp);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

}  // namespace

int uw_flash_attention(int dtype, const AttnParams& p, hipStream_t s)
{
    return dtype == UW_F32 ? launch_flash<float>(p, s) : launch_flash<bf16_t>(p, s);
}

int uw_attention_probs(int dtype, const ProbParams& pp, hipStream_t s)
{
    return dtype == UW_F32 ? launch_probs<float>(pp, s) : launch_probs<bf16_t>(pp, s);
}
