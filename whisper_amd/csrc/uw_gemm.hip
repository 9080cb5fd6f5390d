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
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace {

// element offset of C[m][n] inside one batch item (see GemmParams::c_group)
__device__ __forceinline__ long c_off(const GemmParams& p, long m, int n)
{
    return p.c_group ? (long)(n >> 6) * p.c_group + m * p.ldc + (n & 63) : m * p.ldc + n;
}

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
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
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
    const unsigned lds_base = lds_addr_of(smem) + __builtin_amdgcn_readfirstlane(wave) * 1024;
    // split activations: the other operand (the weights) repeats after kwrap elements of K
    const int wrap_a = p.kwrap_a ? p.kwrap_a / KT : 0x7fffffff, wrap_b = p.kwrap_b ? p.kwrap_b / KT : 0x7fffffff;
    auto stage = [&](int kt, int buf) {
        const unsigned a_dst = lds_base + buf * 2 * TILE_BYTES, b_dst = a_dst + TILE_BYTES;
        const int kta = kt >= wrap_a ? kt - wrap_a : kt, ktb = kt >= wrap_b ? kt - wrap_b : kt;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(srcA[i] + (long)kta * KT, a_dst + i * 4096);
            glds16(srcB[i] + (long)ktb * KT, b_dst + i * 4096);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kt landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                       // ... everyone's; and tile kt-1 is no longer being read
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);       // overlaps the MFMAs below
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
                    *(f32x4*)((float*)C + c_off(p, m, n)) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    union { bf16_t h[4]; uint2 u; } pk, lo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pk.h[r] = (bf16_t)v[r];
                        lo.h[r] = (bf16_t)(v[r] - (float)pk.h[r]);
                    }
                    *(uint2*)((bf16_t*)C + c_off(p, m, n)) = pk.u;
                    if (p.c_split) *(uint2*)((bf16_t*)C + c_off(p, m, n) + p.N) = lo.u;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r >= p.N) break;
                    float x = v[r];
                    if (R) x += R[(long)m * p.ldr + n + r];
                    const TC hi = from_f32<TC>(x);
                    C[c_off(p, m, n) + r] = hi;
                    if (p.c_split) C[c_off(p, m, n) + r + p.N] = from_f32<TC>(x - to_f32<TC>(hi));
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
    UW_REQUIRE(p.kwrap_a % KT == 0 && p.kwrap_b % KT == 0 && (!p.kwrap_a || p.K == 2 * p.kwrap_a) && (!p.kwrap_b || p.K == 2 * p.kwrap_b),
               "gemm: split operands need K=%d to be twice the wrap length, a multiple of %d", p.K, KT);
    UW_REQUIRE(!p.c_split || (sizeof(TC) == 2 && !p.c_group && !p.residual && p.N % 4 == 0 && p.ldc >= 2 * (long)p.N),
               "gemm: split output is a bf16 row of 2N=%d elements without residual", 2 * p.N);
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

// ------------------------------------------------------------------------------------------------
// Large-M form: 256 x 128 tile, 8 waves (4 x 2, 64 x 64 each), three-stage LDS ring filled by global_load_lds.
// Two K-tiles stay in flight while one is consumed: the wait before each barrier is a COUNTED vmcnt that
// retires only the oldest stage, and the barrier is the raw s_barrier (a __syncthreads() would drain the DMA).
// The epilogue parks the f32 tile in the (now idle) ring and leaves as whole 128/256-byte row segments.
constexpr int B2M = 256, B2N = 128;
constexpr int A2_BYTES = B2M * KB, B2_BYTES = B2N * KB, STAGE2 = A2_BYTES + B2_BYTES;   // 48 KiB
constexpr int NSTAGE2 = 3;
constexpr int EP_PITCH = 68;   // floats per staged row (64 + 4: conflict-free 16-byte LDS writes)

template <typename T, typename TC>
__global__ __launch_bounds__(512) void gemm_tn_big_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int KT = KB / (int)sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (p.N + B2N - 1) / B2N;
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * B2M, n0 = (tile % tiles_n) * B2N;
    const long bz = blockIdx.y;
    const T* __restrict__ A = (const T*)p.A + bz * p.sA;
    const T* __restrict__ B = (const T*)p.B + bz * p.sB;

    const T* srcA[4];
    const T* srcB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int L = i * 512 + tid, row = L >> 3, c = (L & 7) ^ (row & 7);
        srcA[i] = A + (long)min(m0 + row, p.M - 1) * p.lda + c * E;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = i * 512 + tid, row = L >> 3, c = (L & 7) ^ (row & 7);
        srcB[i] = B + (long)min(n0 + row, p.N - 1) * p.ldb + c * E;
    }
    const unsigned lds_base = lds_addr_of(smem) + wave * 1024;
    auto stage = [&](int kt, int st) {
        const unsigned a_dst = lds_base + st * STAGE2, b_dst = a_dst + A2_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(srcA[i] + (long)kt * KT, a_dst + i * 8192);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(srcB[i] + (long)kt * KT, b_dst + i * 8192);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / KT;
    const int frow = lane & 15, fg = lane >> 4;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's six DMA pieces of tile kt are older than the six of tile kt+1
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < nk) stage(kt + 2, (kt + 2) % NSTAGE2);
        const char* As = smem + (kt % NSTAGE2) * STAGE2;
        const char* Bs = As + A2_BYTES;
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
    // ---- epilogue: bias / activation in registers, f32 tile through LDS, coalesced residual add + store ----
    __syncthreads();
    float* ep = (float*)smem + wave * (64 * EP_PITCH);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const int ml = tm * 16 + frow;
        const int m = m0 + wr * 64 + ml;
        const float bias_m = (p.bias_mode == 2 && m < p.M) ? p.bias[m] : 0.f;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int nl = tn * 16 + fg * 4;
            const int n = n0 + wc * 64 + nl;
            f32x4 v = acc[tm][tn];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (p.bias_mode == 1) v[r] += (n + r < p.N) ? p.bias[n + r] : 0.f;
                else v[r] += bias_m;
                if (p.act == 1) v[r] = gelu_erf(v[r]);
            }
            *(f32x4*)(ep + ml * EP_PITCH + nl) = v;
        }
    }
    // the wave reads back only what it wrote: no workgroup barrier needed, LDS ops of a wave are ordered
    TC* __restrict__ C = (TC*)p.C + bz * p.sC;
    const float* __restrict__ R = p.residual ? p.residual + bz * p.sR : nullptr;
    const int nbase = n0 + wc * 64, mbase = m0 + wr * 64;
    if constexpr (sizeof(TC) == 4) {
        const bool vec_ok = ((p.ldc & 3) == 0) && (!R || (p.ldr & 3) == 0);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int ml = it * 4 + (lane >> 4), nl = (lane & 15) * 4;
            const int m = mbase + ml, n = nbase + nl;
            if (m >= p.M || n >= p.N) continue;
            f32x4 v = *(const f32x4*)(ep + ml * EP_PITCH + nl);
            if (vec_ok && n + 3 < p.N) {
                if (R) v += *(const f32x4*)(R + (long)m * p.ldr + n);
                *(f32x4*)((float*)C + c_off(p, m, n)) = v;
            } else {
                for (int r = 0; r < 4 && n + r < p.N; ++r)
                    ((float*)C)[c_off(p, m, n) + r] = v[r] + (R ? R[(long)m * p.ldr + n + r] : 0.f);
            }
        }
    } else {
        const bool vec_ok = ((p.ldc & 7) == 0) && (((uintptr_t)C & 15) == 0) && !R;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int ml = it * 8 + (lane >> 3), nl = (lane & 7) * 8;
            const int m = mbase + ml, n = nbase + nl;
            if (m >= p.M || n >= p.N) continue;
            const f32x4 v0 = *(const f32x4*)(ep + ml * EP_PITCH + nl), v1 = *(const f32x4*)(ep + ml * EP_PITCH + nl + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (vec_ok && n + 7 < p.N) {
                union { bf16_t h[8]; uint4 u; } pk;
#pragma unroll
                for (int r = 0; r < 8; ++r) pk.h[r] = (bf16_t)v[r];
                *(uint4*)((bf16_t*)C + c_off(p, m, n)) = pk.u;
            } else {
                for (int r = 0; r < 8 && n + r < p.N; ++r)
                    C[c_off(p, m, n) + r] = from_f32<TC>(v[r] + (R ? R[(long)m * p.ldr + n + r] : 0.f));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Large-M form for TWO co-resident workgroups per CU: same 256 x 128 tile and 8 waves, but 64-byte K rows
// (three 24 KiB stages = 72 KiB) and at most 128 registers, so that one workgroup's epilogue (bias,
// activation, the HBM-bound store of the tile) overlaps the other workgroup's MFMA loop.  With K = 768 a tile
// holds only a few microseconds of MFMA work, so an epilogue that nothing overlaps costs as much as the loop.
// LDS image: row pitch 64 bytes; the 16-byte chunk k of row r sits at position k ^ ((-(r >> 2)) & 3), which
// spreads each 16-lane group of a ds_read_b128 over all 64 banks.
// the output tile is written once and read by a later kernel from HBM: non-temporal store
#define STORE16_NT(ptr, val) __builtin_nontemporal_store((val), (f32x4*)(ptr))
constexpr int K3_ROW = 64;                                         // bytes of K per tile row and stage
constexpr int A3_BYTES = B2M * K3_ROW, B3_BYTES = B2N * K3_ROW, STAGE3 = A3_BYTES + B3_BYTES;   // 24 KiB
constexpr int NSTAGE3 = 3;

// ---- pieces of the two-workgroup tile kernel (prologue, epilogue, tile order), templated on the wave's row tiles so that the
// development variants in tools/gemm_variants.hip share them ----

// tile of this workgroup: batch item and origin.  false: the block lies in the padding of a panel-ordered grid.
__device__ __forceinline__ bool tile_origin_v(const GemmParams& p, int vt, int bn, int tiles_n, int tiles_mn, int total, long& bz,
                                              int& m0, int& n0)
{
    if (p.panel_n > 0) {
        // Column-panel order (batch 1): the row bands are cut into eight runs, one per XCD (blocks b and b + 8 share
        // an L2), and inside its run an XCD sweeps one panel of panel_n column tiles over all its row bands before it
        // moves to the next panel.  The panel's weight rows then stay in that XCD's L2 while the activation bands
        // stream past; in plain row-major order a weight matrix larger than the L2 is fetched again for every row band.
        const int tiles_m = tiles_mn / tiles_n, bands = (tiles_m + 7) >> 3;
        const int xcd = vt & 7, idx = vt >> 3;
        const int band0 = xcd * bands, nb = min(bands, tiles_m - band0);
        if (nb <= 0 || idx >= nb * tiles_n) return false;      // the index space is padded to 8 * bands * tiles_n
        const int per_panel = nb * p.panel_n;
        const int pn = idx / per_panel, rem = idx - pn * per_panel;
        const int pw = min(p.panel_n, tiles_n - pn * p.panel_n);    // the last panel may be narrower
        bz = 0;
        m0 = (band0 + rem / pw) * B2M;
        n0 = (pn * p.panel_n + rem % pw) * bn;
    } else {
        const int ct = xcd_contiguous_tile(vt, total);
        bz = ct / tiles_mn;
        const int tile = ct % tiles_mn;
        m0 = (tile / tiles_n) * B2M;
        n0 = (tile % tiles_n) * bn;
    }
    return true;
}
__device__ __forceinline__ bool tile_origin(const GemmParams& p, int tiles_n, int tiles_mn, int total, long& bz, int& m0, int& n0)
{
    return tile_origin_v(p, (int)blockIdx.x, B2N, tiles_n, tiles_mn, total, bz, m0, n0);
}

// The accumulators start at bias + residual: those loads ride behind the first K-stages instead of opening the
// epilogue (a tile's residual is read before any workgroup writes that tile, so C may alias the residual).
// mw0 / nw0: first row / column of the wave's block of TM x 4 MFMA tiles.
template <int TM>
__device__ __forceinline__ void tile_acc_init(const GemmParams& p, f32x4 (&acc)[TM][4], int mw0, int nw0, int frow, int fg,
                                              const float* __restrict__ R, bool r_late, bool r_vec)
{
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mw0 + i * 16 + frow;
        const float bias_m = (p.bias_mode == 2 && m < p.M) ? p.bias[m] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nw0 + j * 16 + fg * 4;
            if (p.bias_mode == 1) {
                if (n + 3 < p.N) acc[i][j] = *(const f32x4*)(p.bias + n);
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = (n + r < p.N) ? p.bias[n + r] : 0.f;
            } else {
                acc[i][j] = f32x4{bias_m, bias_m, bias_m, bias_m};
            }
            if (R && !r_late && m < p.M) {
                if (r_vec && n + 3 < p.N) acc[i][j] += *(const f32x4*)(R + (long)m * p.ldr + n);
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += (n + r < p.N) ? R[(long)m * p.ldr + n + r] : 0.f;
            }
        }
    }
}

// Epilogue of one wave: its TM x 4 accumulator tiles leave in parts of 32 rows through `ep` (the wave's own 32 x EP_PITCH
// floats of the idle ring), as whole row segments.  The wave reads back only what it wrote: LDS operations of one wave
// are ordered, no barrier needed between the parts.
template <typename T, typename TC, int TM>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, f32x4 (&acc)[TM][4], float* ep, int mw0, int nbase, long bz,
                                              int lane, const float* __restrict__ R, bool r_late)
{
    const int frow = lane & 15, fg = lane >> 4;
    TC* __restrict__ C = (TC*)p.C + bz * p.sC;
    constexpr int KTC = K3_ROW / (int)sizeof(TC);
    // element offset of C[m][n]; K-blocked output: the 4 or 8 consecutive columns a lane stores stay inside one block
    auto coff = [&](int m, int n) -> long {
        return p.c_kblock ? ((long)(n / KTC) * p.c_kblock + m) * KTC + (n % KTC) : c_off(p, m, n);
    };
#pragma unroll
    for (int part = 0; part < TM / 2; ++part) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int tm = part * 2 + t2;
            const int ml = t2 * 16 + frow;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                const int nl = tn * 16 + fg * 4;
                f32x4 v = acc[tm][tn];
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
                *(f32x4*)(ep + ml * EP_PITCH + nl) = v;
            }
        }
        const int mbase = mw0 + part * 32;
        if (p.C2 && nbase >= p.c2_col0) {
            // transposed copy: lane -> four consecutive rows m of one column n; eight lanes fill 64 bytes of a C2 row
            TC* __restrict__ C2 = (TC*)p.C2 + bz * p.sC2;
            const int m4 = (lane & 7) * 4;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int nl = it * 8 + (lane >> 3);
                const int m = mbase + m4, n = nbase + nl;
                if (n >= p.N || m >= p.M) continue;
                union { TC h[4]; uint2 u2; f32x4 u4; } pk;
                TC* const h = pk.h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = from_f32<TC>(ep[(m4 + r) * EP_PITCH + nl]);
                TC* dst = C2 + (long)(n - p.c2_col0) * p.ldc2 + m;
                if (m + 3 < p.M && (p.ldc2 & 3) == 0) {
                    if constexpr (sizeof(TC) == 4) *(f32x4*)dst = pk.u4;
                    else *(uint2*)dst = pk.u2;
                } else {
                    for (int r = 0; r < 4 && m + r < p.M; ++r) dst[r] = h[r];
                }
            }
        }
        if constexpr (sizeof(TC) == 4) {
            const bool vec_ok = (p.ldc & 3) == 0;
            f32x4 rr[8];
            if (r_late) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = mbase + it * 4 + (lane >> 4), n = nbase + (lane & 15) * 4;
                    rr[it] = (m < p.M && n + 3 < p.N) ? *(const f32x4*)(R + (long)m * p.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int ml = it * 4 + (lane >> 4), nl = (lane & 15) * 4;
                const int m = mbase + ml, n = nbase + nl;
                if (m >= p.M || n >= p.N) continue;
                f32x4 v = *(const f32x4*)(ep + ml * EP_PITCH + nl);
                if (r_late) {
                    if (n + 3 < p.N) v += rr[it];
                    else
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (n + r < p.N) ? R[(long)m * p.ldr + n + r] : 0.f;
                }
                if (vec_ok && n + 3 < p.N) {
                    STORE16_NT((float*)C + coff(m, n), v);
                } else {
                    for (int r = 0; r < 4 && n + r < p.N; ++r) ((float*)C)[coff(m, n) + r] = v[r];
                }
            }
        } else {
            const bool vec_ok = ((p.ldc & 7) == 0) && (((uintptr_t)C & 15) == 0);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int ml = it * 8 + (lane >> 3), nl = (lane & 7) * 8;
                const int m = mbase + ml, n = nbase + nl;
                if (m >= p.M || n >= p.N) continue;
                const f32x4 v0 = *(const f32x4*)(ep + ml * EP_PITCH + nl), v1 = *(const f32x4*)(ep + ml * EP_PITCH + nl + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (vec_ok && n + 7 < p.N) {
                    union { bf16_t h[8]; uint4 u; } pk;
#pragma unroll
                    for (int r = 0; r < 8; ++r) pk.h[r] = (bf16_t)v[r];
                    STORE16_NT((bf16_t*)C + coff(m, n), __builtin_bit_cast(f32x4, pk.u));
                    if (p.c_split) {
                        union { bf16_t h[8]; uint4 u; } lo;
#pragma unroll
                        for (int r = 0; r < 8; ++r) lo.h[r] = (bf16_t)(v[r] - (float)pk.h[r]);
                        STORE16_NT((bf16_t*)C + coff(m, n) + p.N, __builtin_bit_cast(f32x4, lo.u));
                    }
                } else {
                    for (int r = 0; r < 8 && n + r < p.N; ++r) {
                        const TC hi = from_f32<TC>(v[r]);
                        C[coff(m, n) + r] = hi;
                        if (p.c_split) C[coff(m, n) + r + p.N] = from_f32<TC>(v[r] - to_f32<TC>(hi));
                    }
                }
            }
        }
    }
}

template <typename T, typename TC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_tn_duo_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int KT = K3_ROW / (int)sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (p.N + B2N - 1) / B2N;
    const int tiles_mn = tiles_n * ((p.M + B2M - 1) / B2M), total = tiles_mn * p.batch;
    const int frow = lane & 15, fg = lane >> 4;
    const unsigned lds_base = lds_addr_of(smem) + wave * 1024;
    long bz;
    int m0, n0;
    if (!tile_origin(p, tiles_n, tiles_mn, total, bz, m0, n0)) return;
    const T* __restrict__ A = (const T*)p.A + bz * p.sA;
    const T* __restrict__ B = (const T*)p.B + bz * p.sB;

    // one DMA instruction of a wave fills 16 rows x 64 bytes; lane -> (row, position), source chunk = position ^ g(row).
    // Per-lane byte offsets inside the tile are loop-invariant; tile origin and K position go into the scalar base.
    unsigned offA[2], offB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = i * 512 + tid, row = L >> 2, c = (L & 3) ^ ((-(row >> 2)) & 3);
        offA[i] = (unsigned)(((long)(min(m0 + row, p.M - 1) - m0) * p.lda + c * E) * (long)sizeof(T));
    }
    {
        const int row = tid >> 2, c = (tid & 3) ^ ((-(row >> 2)) & 3);
        offB = (unsigned)(((long)(min(n0 + row, p.N - 1) - n0) * p.ldb + c * E) * (long)sizeof(T));
    }
    const T* baseA = A + (long)m0 * p.lda;
    const T* baseB = B + (long)n0 * p.ldb;
    // split activations: the other operand (the weights) repeats after kwrap elements of K
    const int wrap_a = p.kwrap_a ? p.kwrap_a / KT : 0x7fffffff, wrap_b = p.kwrap_b ? p.kwrap_b / KT : 0x7fffffff;
    auto stage = [&](int kt, int st) {
        const unsigned a_dst = lds_base + st * STAGE3, b_dst = a_dst + A3_BYTES;
        const T* ka = baseA + (long)(kt >= wrap_a ? kt - wrap_a : kt) * (p.kstep_a ? p.kstep_a : KT);
        const T* kb = baseB + (long)(kt >= wrap_b ? kt - wrap_b : kt) * (p.kstep_b ? p.kstep_b : KT);
        glds16_sv(ka, offA[0], a_dst);
        glds16_sv(ka, offA[1], a_dst + 8192);
        glds16_sv(kb, offB, b_dst);
    };

    const int nk = p.K / KT;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    // the accumulators start at bias + residual: those loads ride behind the first K-stage instead of opening the
    // epilogue (a tile's residual is read before any workgroup writes that tile, so C may alias the residual)
    const float* __restrict__ R = p.residual ? p.residual + bz * p.sR : nullptr;
    const bool r_vec = (p.ldr & 3) == 0;
    // f32 output with a residual of the same row pitch (the residual stream updated in place): the residual is read in
    // the epilogue, by the lanes and in the whole-row segments that store the result, instead of as 64-byte pieces of
    // sixteen rows per instruction in front of the loop
    const bool r_late = sizeof(TC) == 4 && R && r_vec && p.ldr == p.ldc && !p.c_group && !p.c_kblock && (p.ldc & 3) == 0;
    f32x4 acc[4][4];
    tile_acc_init<4>(p, acc, m0 + wr * 64, n0 + wc * 64, frow, fg, R, r_late, r_vec);

    retire_vector_loads();   // bias / residual are in the accumulators (and the first two stages have landed)
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's three DMA pieces of tile kt are older than the three of tile kt+1
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < nk) stage(kt + 2, (kt + 2) % NSTAGE3);
        const char* As = smem + (kt % NSTAGE3) * STAGE3;
        const char* Bs = As + A3_BYTES;
        uint4 af[4], bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // rows t*16 + frow of a 64-row block: (row >> 2) & 3 == (frow >> 2) & 3
            const int pos = (fg ^ ((-(frow >> 2)) & 3)) << 4;
            af[t] = *(const uint4*)(As + (wr * 64 + t * 16 + frow) * K3_ROW + pos);
            bf[t] = *(const uint4*)(Bs + (wc * 64 + t * 16 + frow) * K3_ROW + pos);
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) mma16<T>(bf[tn], af[tm], acc[tm][tn]);
    }
    // ---- epilogue: two halves of 32 rows per wave through the idle ring (69.6 KiB of its 72) ----
    __syncthreads();
    tile_epilogue<T, TC, 4>(p, acc, (float*)smem + wave * (32 * EP_PITCH), m0 + wr * 64, n0 + wc * 64, bz, lane, R, r_late);
}

// Column tiles per panel for the panel order of gemm_tn_duo_kernel, or 0 for the plain order.  Model of the bytes an
// XCD pulls through the fabric beyond the algorithmic ones: in plain order a weight matrix that does not fit the 4 MiB
// L2 beside the activation bands is fetched again for every row band when K is short (tiles of one band do not stay in
// step); with n panels the activation matrix is read n times instead of once.  Panels are used when that is less.
static int choose_panel(long es, const GemmParams& p, int bn = B2N)
{
    static const int forced = getenv("UW_GEMM_PANEL") ? atoi(getenv("UW_GEMM_PANEL")) : -1;   // development knob
    const int tiles_m = ceil_div(p.M, B2M), tiles_n = ceil_div(p.N, bn);
    if (p.batch != 1 || tiles_m < 64 || tiles_n < 2) return 0;
    if (forced >= 0) return forced >= tiles_n ? 0 : forced;
    const long w_bytes = (long)p.N * p.K * es, a_bytes = (long)p.M * p.K * es;
    if (w_bytes <= 3l << 20) return 0;                          // the whole weight matrix stays in L2 anyway
    const long panel_cap = 5l << 19;                            // 2.5 MiB of weights per panel
    int pw = (int)std::max<long>(1, panel_cap / ((long)bn * p.K * es));
    const int n_panels = ceil_div(tiles_n, pw);
    pw = ceil_div(tiles_n, n_panels);                           // even panels
    if (n_panels < 2) return 0;
    const long refetch_plain = (long)tiles_m * w_bytes, extra_panels = (long)(n_panels - 1) * a_bytes;
    // a long K keeps the tiles of a band in step, and then the plain order fetches the weights once per wave of
    // co-resident tiles only (measured: K = 3072, N = 768 pulls 1.3 GB extra, not 3.5): leave those alone
    if (p.K * es > 2048) return 0;
    return extra_panels * 2 < refetch_plain ? pw : 0;
}

template <typename T, typename TC> int launch_duo(const GemmParams& p, hipStream_t s)
{
    constexpr int KT = K3_ROW / (int)sizeof(T);
    UW_REQUIRE(p.K % KT == 0, "gemm: K=%d must be a multiple of %d", p.K, KT);
    UW_REQUIRE(p.kwrap_a % KT == 0 && p.kwrap_b % KT == 0 && (!p.kwrap_a || p.K == 2 * p.kwrap_a) && (!p.kwrap_b || p.K == 2 * p.kwrap_b),
               "gemm: split operands need K=%d to be twice the wrap length, a multiple of %d", p.K, KT);
    UW_REQUIRE(!p.c_split || (sizeof(TC) == 2 && !p.c_group && !p.residual && p.N % 8 == 0 && p.ldc >= 2 * (long)p.N),
               "gemm: split output is a bf16 row of 2N=%d elements without residual", 2 * p.N);
    UW_REQUIRE((p.lda * sizeof(T)) % 16 == 0 && (p.ldb * sizeof(T)) % 16 == 0 && (p.sA * sizeof(T)) % 16 == 0 &&
                   (p.sB * sizeof(T)) % 16 == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0,
               "gemm: operand rows must be 16-byte aligned");
    const long tiles = (long)ceil_div(p.M, B2M) * ceil_div(p.N, B2N);
    UW_REQUIRE(tiles * p.batch < (1l << 31), "gemm: too many tiles");
    UW_REQUIRE((long)B2M * p.lda * (long)sizeof(T) < (1l << 32) && (long)B2N * p.ldb * (long)sizeof(T) < (1l << 32),
               "gemm: operand row stride too large for 32-bit tile offsets");
    const size_t lds = (size_t)NSTAGE3 * STAGE3;
    static PerDeviceFlag attr_set;
    if (attr_set.pending()) {
        UW_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_duo_kernel<T, TC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.mark();
    }
    const long total = tiles * p.batch;
    unsigned grid = (unsigned)total;
    GemmParams q = p;
    q.panel_n = choose_panel((long)sizeof(T), p);
    if (q.panel_n > 0) grid = 8u * (unsigned)ceil_div(ceil_div(p.M, B2M), 8) * (unsigned)ceil_div(p.N, B2N);
    hipLaunchKernelGGL((gemm_tn_duo_kernel<T, TC>), dim3(grid), dim3(512), lds, s, q);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

// ------------------------------------------------------------------------------------------------
// Few rows against a tall weight matrix with split activations: C[m][n] = sum_k (A_hi[m][k] + A_lo[m][k]) * B[n][k],
// M <= 128, bf16 operands, f32 C -- the tied output projection of a decode step (N = vocabulary).  One workgroup per
// 256 columns, 8 waves of 64 x 64: a stage holds the hi rows, the lo rows and 256 weight rows of one 32-deep K-step,
// and every weight fragment is multiplied with both.  Against running the tile kernel over [hi | lo] with the weights
// repeating, the weight tile crosses into LDS once instead of twice and a workgroup's copy of the activations serves
// 256 columns instead of 128: the product is bound by what a CU can take in through the LDS DMA, and this halves it.
constexpr int HLM = 128, HLN = 256;
constexpr int HL_A = HLM * K3_ROW, HL_B = HLN * K3_ROW, HL_STAGE = 2 * HL_A + HL_B;   // 8 + 8 + 16 KiB
constexpr int HL_NSTAGE = 3;

__global__ __launch_bounds__(512) void gemm_rows_hilo_kernel(GemmParams p)
{
    using T = bf16_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 8, KT = K3_ROW / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * HLN;
    const unsigned lds_base = lds_addr_of(smem) + wave * 1024;
    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ B = (const T*)p.B + (long)n0 * p.ldb;
    unsigned offA, offB[2];
    {
        const int row = tid >> 2, c = (tid & 3) ^ ((-(row >> 2)) & 3);
        offA = (unsigned)(((long)min(row, p.M - 1) * p.lda + c * E) * 2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = i * 512 + tid, row = L >> 2, c = (L & 3) ^ ((-(row >> 2)) & 3);
        offB[i] = (unsigned)(((long)(min(n0 + row, p.N - 1) - n0) * p.ldb + c * E) * 2);
    }
    auto stage = [&](int kt, int st) {
        const unsigned hi_dst = lds_base + st * HL_STAGE, lo_dst = hi_dst + HL_A, b_dst = lo_dst + HL_A;
        const T* ka = A + (long)kt * KT;
        const T* kb = B + (long)kt * KT;
        glds16_sv(ka, offA, hi_dst);
        glds16_sv(ka + p.K, offA, lo_dst);       // the lo half of a row starts K elements after the hi half
        glds16_sv(kb, offB[0], b_dst);
        glds16_sv(kb, offB[1], b_dst + 8192);
    };
    const int nk = p.K / KT;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < nk) stage(kt + 2, (kt + 2) % HL_NSTAGE);
        const char* Hs = smem + (kt % HL_NSTAGE) * HL_STAGE;
        const char* Ls = Hs + HL_A;
        const char* Bs = Ls + HL_A;
        const int pos = (fg ^ ((-(frow >> 2)) & 3)) << 4;
        uint4 ah[4], al[4], bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ah[t] = *(const uint4*)(Hs + (wr * 64 + t * 16 + frow) * K3_ROW + pos);
            al[t] = *(const uint4*)(Ls + (wr * 64 + t * 16 + frow) * K3_ROW + pos);
            bf[t] = *(const uint4*)(Bs + (wc * 64 + t * 16 + frow) * K3_ROW + pos);
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                mma16<T>(bf[tn], ah[tm], acc[tm][tn]);
                mma16<T>(bf[tn], al[tm], acc[tm][tn]);
            }
    }
    // epilogue: two halves of 32 rows per wave through the idle ring, whole 256-byte row segments to memory
    __syncthreads();
    float* ep = (float*)smem + wave * (32 * EP_PITCH);
    float* __restrict__ C = (float*)p.C;
    const int nbase = n0 + wc * 64;
    const bool vec_ok = (p.ldc & 3) == 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int tm = half * 2 + t2, ml = t2 * 16 + frow;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) *(f32x4*)(ep + ml * EP_PITCH + tn * 16 + fg * 4) = acc[tm][tn];
        }
        const int mbase = wr * 64 + half * 32;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int ml = it * 4 + (lane >> 4), nl = (lane & 15) * 4;
            const int m = mbase + ml, n = nbase + nl;
            if (m >= p.M || n >= p.N) continue;
            const f32x4 v = *(const f32x4*)(ep + ml * EP_PITCH + nl);
            if (vec_ok && n + 3 < p.N) {
                *(f32x4*)(C + (long)m * p.ldc + n) = v;
            } else {
                for (int r = 0; r < 4 && n + r < p.N; ++r) C[(long)m * p.ldc + n + r] = v[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Logits of a decode step for at most 16 windows (see LogitsRows16Params).  Against LayerNorm + gemm_rows_hilo_kernel +
// a scan of the logits this is one launch instead of two, no 128-row tile for a handful of rows, and the greedy step
// reads one candidate per workgroup and row instead of the whole row.
// One workgroup = 256 vocabulary columns, 8 waves x two 16-column tiles.  Phase 1: every workgroup normalises the 16
// rows itself (16 x d floats from L2): wave w takes the 32-column K-steps w, w + 8, ..., a lane holds the 8 columns of
// its MFMA operand chunk, so the normalised values go to LDS as ready operand blocks, hi and lo plane.  Phase 2: the
// fragment-major embedding streams from memory straight into registers (1 KiB of consecutive bytes per wave
// instruction; with row-major rows, 16 x 64 bytes per instruction, the same kernel reached 2.7 TB/s), LR_U K-steps in
// flight per buffer, two buffers.
constexpr int LR_WAVES = 8, LR_COLS = 256, LR_U = 6, LR_MAXS = 5;   // LR_MAXS: K-steps per wave in phase 1 (d <= 1280)

__global__ __launch_bounds__(64 * LR_WAVES) void logits_rows16_kernel(LogitsRows16Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [d / 32][hi, lo][64 lanes][16 B]
    __shared__ float stat[LR_WAVES][16][2];
    __shared__ float c_val[LR_WAVES][16];
    __shared__ int c_idx[LR_WAVES][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int steps = p.d >> 5;
    // the embedding does not depend on the LayerNorm: the first two buffers of the stream are requested before its
    // arithmetic and arrive while the rows are being normalised
    const int n0 = blockIdx.x * LR_COLS + wave * 32;
    const int n_blocks = (p.V + 15) >> 4;
    const char* wrow[2];        // this lane's 16 bytes of the first operand block of each tile; blocks of a tile follow every 1 KiB
#pragma unroll
    for (int t = 0; t < 2; ++t) wrow[t] = (const char*)p.W + ((long)min((n0 >> 4) + t, n_blocks - 1) * steps * 64 + lane) * 16;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    uint4 wv[2][LR_U][2];
    auto load_round = [&](int buf, int s0) {
#pragma unroll
        for (int u = 0; u < LR_U; ++u) {
            const int st = min(s0 + u, steps - 1);      // steps past the end re-read the last one (unused)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                const u32x4 v = __builtin_nontemporal_load((const u32x4*)(wrow[t] + st * 1024));   // read once per step
                wv[buf][u][t] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
    };
    // ---- phase 1: LayerNorm of the 16 rows into LDS.  Its operand is requested first (loads return in order), the
    // stream right behind it.
    {
        f32x4 xv[LR_MAXS][2];
#pragma unroll
        for (int i = 0; i < LR_MAXS; ++i) {
            const int st = min(wave + LR_WAVES * i, steps - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) xv[i][h] = *(const f32x4*)(p.x + ((long)(st * 2 + h) * 64 + lane) * 4);
        }
        load_round(0, 0);
        if (LR_U < steps) load_round(1, LR_U);
        __builtin_amdgcn_sched_barrier(0);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LR_MAXS; ++i) {
            if (wave + LR_WAVES * i < steps) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 v = xv[i][h];
                    s1 += (v[0] + v[1]) + (v[2] + v[3]);
                    s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
            }
        }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (fg == 0) { stat[wave][fr][0] = s1; stat[wave][fr][1] = s2; }
        __syncthreads();
        float S1 = 0.f, S2 = 0.f;
#pragma unroll
        for (int w = 0; w < LR_WAVES; ++w) { S1 += stat[w][fr][0]; S2 += stat[w][fr][1]; }
        const float inv_d = 1.0f / (float)p.d;
        const float mean = S1 * inv_d;
        const float rstd = 1.0f / sqrtf(fmaxf(S2 * inv_d - mean * mean, 0.f) + 1e-5f);
#pragma unroll
        for (int i = 0; i < LR_MAXS; ++i) {
            const int st = wave + LR_WAVES * i;
            if (st < steps) {
                union { bf16_t h[8]; uint4 u; } hi, lo;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 g = *(const f32x4*)(p.gamma + st * 32 + fg * 8 + h * 4), b = *(const f32x4*)(p.beta + st * 32 + fg * 8 + h * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = (xv[i][h][e] - mean) * rstd * g[e] + b[e];
                        hi.h[h * 4 + e] = (bf16_t)y;
                        lo.h[h * 4 + e] = (bf16_t)(y - (float)hi.h[h * 4 + e]);
                    }
                }
                *(uint4*)(smem + ((st * 2 + 0) * 64 + lane) * 16) = hi.u;
                *(uint4*)(smem + ((st * 2 + 1) * 64 + lane) * 16) = lo.u;
            }
        }
    }
    __syncthreads();
    // ---- phase 2: the product
    auto mma_round = [&](int buf, int s0) {
#pragma unroll
        for (int u = 0; u < LR_U; ++u) {
            if (s0 + u < steps) {
                const uint4 ah = *(const uint4*)(smem + (((s0 + u) * 2 + 0) * 64 + lane) * 16);
                const uint4 al = *(const uint4*)(smem + (((s0 + u) * 2 + 1) * 64 + lane) * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    mma16<bf16_t>(wv[buf][u][t], ah, acc[t]);
                    mma16<bf16_t>(wv[buf][u][t], al, acc[t]);
                }
            }
        }
    };
    for (int s0 = 0; s0 < steps; s0 += 2 * LR_U) {
        if (s0 > 0 && s0 + LR_U < steps) load_round(1, s0 + LR_U);
        __builtin_amdgcn_sched_barrier(0);
        mma_round(0, s0);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 + LR_U < steps) {
            if (s0 + 2 * LR_U < steps) load_round(0, s0 + 2 * LR_U);
            __builtin_amdgcn_sched_barrier(0);
            mma_round(1, s0 + LR_U);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- epilogue: acc[t][r] = logit of row fr, column n0 + 16 t + 4 fg + r
    const int m = fr;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    bool bad = false;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + t * 16 + fg * 4;
        if (p.logits && m < p.M && n < p.V) {
            float* dst = p.logits + (long)m * p.ldl + n;
            if (n + 3 < p.V && (p.ldl & 3) == 0) *(f32x4*)dst = acc[t];
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.V) dst[r] = acc[t][r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[t][r];
            if (n + r < p.V) {
                bad |= (v != v);
                if (v > best) { best = v; arg = n + r; }     // ascending n: the first maximum stays
            }
        }
    }
    if (!p.cand_val) return;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(arg, o, 64);
        const int ob = __shfl_xor(bad ? 1 : 0, o, 64);
        if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
        bad |= ob != 0;
    }
    if (fg == 0) { c_val[wave][fr] = bad ? __builtin_nanf("") : best; c_idx[wave][fr] = arg; }
    __syncthreads();
    if (tid < 16) {
        float bv = c_val[0][tid];
        int bi = c_idx[0][tid];
        bool nb = bv != bv;
#pragma unroll
        for (int w = 1; w < LR_WAVES; ++w) {
            const float ov = c_val[w][tid];
            const int oi = c_idx[w][tid];
            nb |= ov != ov;
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        const long at = (long)tid * gridDim.x + blockIdx.x;
        p.cand_val[at] = nb ? __builtin_nanf("") : bv;
        p.cand_idx[at] = bi;
    }
}

template <typename T, typename TC> int launch_big(const GemmParams& p, hipStream_t s)
{
    constexpr int KT = KB / (int)sizeof(T);
    UW_REQUIRE(p.K % KT == 0, "gemm: K=%d must be a multiple of %d", p.K, KT);
    UW_REQUIRE(!p.kwrap_a && !p.kwrap_b && !p.c_split, "gemm: split operands are not supported with an activation plus residual");
    UW_REQUIRE((p.lda * sizeof(T)) % 16 == 0 && (p.ldb * sizeof(T)) % 16 == 0 && (p.sA * sizeof(T)) % 16 == 0 &&
                   (p.sB * sizeof(T)) % 16 == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0,
               "gemm: operand rows must be 16-byte aligned");
    const long tiles = (long)ceil_div(p.M, B2M) * ceil_div(p.N, B2N);
    UW_REQUIRE(tiles < (1l << 31) && p.batch < 65536, "gemm: grid too large");
    const size_t lds = (size_t)NSTAGE2 * STAGE2;
    static PerDeviceFlag attr_set;
    if (attr_set.pending()) {
        UW_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_big_kernel<T, TC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.mark();
    }
    hipLaunchKernelGGL((gemm_tn_big_kernel<T, TC>), dim3((unsigned)tiles, (unsigned)p.batch), dim3(512), lds, s, p);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

// ------------------------------------------------------------------------------------------------
// Skinny form for the decode step (M = windows in flight <= 128): the weight matrix is streamed exactly
// once, 16 output columns per workgroup so that even N = d_model gives dozens of workgroups.  The step is
// latency-bound, not bandwidth-bound, so the 8 waves of a workgroup split K and every wave issues the
// loads of a whole round of three 32-deep K-steps (27 x 16 B per lane) before its first MFMA; partial tiles
// are combined through LDS in a fixed order (results are run-to-run identical).  Activations (<= 128 rows)
// stay L2-resident and are read as MFMA fragments straight from global memory.
constexpr int SK_WAVES = 8;
constexpr int SK_ROUND = 3;      // K-steps per register buffer; two buffers are in flight
constexpr int SK_ROUND_LN = 2;   // same for the LayerNorm-fused form, whose f32 operand takes twice the registers

// Workgroup `i` of the extra ones a launch carries for GemmParams::pf: requests one dword of every 128-byte line of the
// weight block that workgroup (i - pf_shift) of the following launch will read.  pf_shift makes the two land on the
// same XCD (workgroups are dealt to the 8 XCDs round-robin by their linear index).  Nothing is kept: the point is
// that the lines have left HBM by the time the following launch asks for them.  pf_never is always 0; the store under
// it only keeps the loads alive.
template <typename T>
__device__ __forceinline__ void prefetch_following_weights(const GemmParams& p, int i)
{
    const int c = i - p.pf_shift, blocks = (p.pf_n + p.pf_cols - 1) / p.pf_cols;
    if (blockIdx.y != 0 || c < 0 || c >= blocks) return;
    const long rows16 = (p.pf_n + 15) >> 4, first16 = (long)c * (p.pf_cols >> 4);
    const long bytes16 = (long)p.pf_k * 16 * (long)sizeof(T);                        // one 16-row block of the packed matrix
    const long lines = min((long)(p.pf_cols >> 4), rows16 - first16) * bytes16 / 128;
    const char* base = (const char*)p.pf + first16 * bytes16;
    unsigned v = 0;
    for (long l0 = threadIdx.x; l0 < lines; l0 += 8 * (long)blockDim.x) {
        unsigned t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long l = l0 + u * (long)blockDim.x;
            t[u] = l < lines ? *(const unsigned*)(base + l * 128) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v |= t[u];
    }
    if (p.pf_never) *(unsigned*)p.C = v;
}

// grid of a skinny launch: the extra workgroups of GemmParams::pf are appended along x, and with several rows of
// workgroups the row length is padded to a multiple of 8, so that block c of this launch is read from XCD c % 8 in
// every row (which is where the launch before it left it)
static dim3 grid_with_prefetch(GemmParams& q, int gx, int gy)
{
    q.pf_first = gx;
    q.pf_never = 0;
    int extra = 0;
    if (q.pf && q.pf_cols >= 16 && q.pf_n > 0 && q.pf_k > 0) {
        q.pf_shift = (8 - gx % 8) % 8;      // extra workgroup i has linear index gx + i: on the XCD of block i - pf_shift
        extra = q.pf_shift + (q.pf_n + q.pf_cols - 1) / q.pf_cols;
    } else {
        q.pf = nullptr;
    }
    if (gy > 1) extra += (8 - (gx + extra) % 8) % 8;
    return dim3(gx + extra, gy);
}

// HILO: A arrives as two fragment-major planes, hi and lo (split activations, see GemmParams); every weight block is
// multiplied with both, the matrix cores being idle in this kernel anyway.
template <typename T, typename TC, int MT, int NT, bool PACKED, bool HILO = false>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_skinny_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int KS = 4 * E;                 // K elements per step (one MFMA chunk group)
    f32x4* part = (f32x4*)smem;               // [SK_WAVES][MT * NT][64]
    if ((int)blockIdx.x >= p.pf_first) {
        if (p.pf) prefetch_following_weights<T>(p, (int)blockIdx.x - p.pf_first);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: K offsets stay scalar
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * (NT * 16), m0 = blockIdx.y * (MT * 16);
    const int steps_total = p.K / KS;
    const int s_begin = (steps_total * wave) / SK_WAVES, s_end = (steps_total * (wave + 1)) / SK_WAVES;
    // buffer descriptors keep the loads in the global address space with 32-bit per-lane offsets; the K
    // position of a round is a scalar offset and the step inside the round an immediate
    constexpr int ES = (int)sizeof(T);
    // bytes from one K-step to the next: 64 along a row, or one 1 KiB block in the fragment-major layout
    constexpr int STEP_BYTES = PACKED ? 1024 : KS * ES;
    const long w_bytes = PACKED ? (long)((p.N + 15) & ~15) * p.K * ES : (long)p.N * p.ldb * ES;
    const long a_bytes = PACKED ? (long)((p.M + 15) & ~15) * p.K * ES : (long)p.M * p.lda * ES;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(HILO ? p.A_lo : p.A), 0, (int)a_bytes, 0x00020000);
    int woff[NT], aoff[MT];
    if constexpr (PACKED) {
        const int n_blocks = (p.N + 15) >> 4, m_blocks = (p.M + 15) >> 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) woff[nt] = min((n0 >> 4) + nt, n_blocks - 1) * steps_total * 1024 + lane * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aoff[mt] = min((m0 >> 4) + mt, m_blocks - 1) * steps_total * 1024 + lane * 16;
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) woff[nt] = (int)(((long)min(n0 + nt * 16 + fr, p.N - 1) * p.ldb + fg * E) * ES);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aoff[mt] = (int)(((long)min(m0 + mt * 16 + fr, p.M - 1) * p.lda + fg * E) * ES);
    }

    // wave w finishes output tile w after the reduction; its bias and residual are requested now so that they
    // arrive during the K loop instead of stalling the epilogue
    static_assert(MT * NT <= SK_WAVES, "one output tile per wave");
    TC* __restrict__ C = (TC*)p.C;
    const float* __restrict__ R = p.residual;
    const int e_mt = wave / NT, e_nt = wave % NT;
    const int e_m = m0 + e_mt * 16 + fr, e_n = n0 + e_nt * 16 + fg * 4;
    const bool e_live = wave < MT * NT && e_m < p.M && e_n < p.N;
    const bool vec_ok = ((p.ldc & 3) == 0) && (!R || (p.ldr & 3) == 0) && (e_n + 3 < p.N);
    f32x4 e_bias = {0.f, 0.f, 0.f, 0.f}, e_res = {0.f, 0.f, 0.f, 0.f};
    if (e_live) {
        if (p.bias_mode == 1) {
            if (e_n + 3 < p.N) e_bias = *(const f32x4*)(p.bias + e_n);
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) e_bias[r] = (e_n + r < p.N) ? p.bias[e_n + r] : 0.f;
        } else if (p.bias_mode == 2) {
            const float b = p.bias[e_m];
            e_bias = f32x4{b, b, b, b};
        }
        if (R) {
            if (vec_ok) e_res = *(const f32x4*)(R + (p.c_xlayout ? x_elem_offset<T>(e_m, e_n, p.N) : (long)e_m * p.ldr + e_n));
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) e_res[r] = (e_n + r < p.N) ? R[(long)e_m * p.ldr + e_n + r] : 0.f;
        }
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto ld = [](const __amdgpu_buffer_rsrc_t& r, int voff, int soff) -> uint4 {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    // Rounds of SK_ROUND K-steps, double-buffered in registers: the loads of round r+1 are issued before the
    // MFMAs of round r, so a wave pays the memory latency once, not once per round.
    uint4 wv[2][SK_ROUND][NT], av[2][SK_ROUND][HILO ? 2 * MT : MT];
    auto load_round = [&](int buf, int s) {
#pragma unroll
        for (int r = 0; r < SK_ROUND; ++r) {
            const int koff = min(s + r, s_end - 1) * STEP_BYTES;   // steps past the end re-read the last one (unused)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wv[buf][r][nt] = ld(rsW, woff[nt], koff);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                av[buf][r][mt] = ld(rsA, aoff[mt], koff);
                if constexpr (HILO) av[buf][r][MT + mt] = ld(rsA2, aoff[mt], koff);
            }
        }
    };
    auto mma_round = [&](int buf, int s) {
#pragma unroll
        for (int r = 0; r < SK_ROUND; ++r) {
            if (s + r < s_end) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        mma16<T>(wv[buf][r][nt], av[buf][r][mt], acc[mt][nt]);
                        if constexpr (HILO) mma16<T>(wv[buf][r][nt], av[buf][r][MT + mt], acc[mt][nt]);
                    }
            }
        }
    };
    if (s_begin < s_end) {
        load_round(0, s_begin);
        for (int s = s_begin; s < s_end; s += 2 * SK_ROUND) {
            if (s + SK_ROUND < s_end) load_round(1, s + SK_ROUND);
            __builtin_amdgcn_sched_barrier(0);
            mma_round(0, s);
            __builtin_amdgcn_sched_barrier(0);
            if (s + SK_ROUND < s_end) {
                if (s + 2 * SK_ROUND < s_end) load_round(0, s + 2 * SK_ROUND);
                __builtin_amdgcn_sched_barrier(0);
                mma_round(1, s + SK_ROUND);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) part[(wave * MT * NT + mt * NT + nt) * 64 + lane] = acc[mt][nt];
    __syncthreads();
    // wave w < MT * NT sums the 8 partials of tile w in wave order (run-to-run identical)
    if (!e_live) return;
    f32x4 t = part[wave * 64 + lane];
#pragma unroll
    for (int w = 1; w < SK_WAVES; ++w) t += part[(w * MT * NT + wave) * 64 + lane];
    f32x4 v = t + e_bias;
    if (p.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    }
    v += e_res;
    if (vec_ok) {
        const long at = p.c_packed ? frag_elem_offset<TC>(e_m, e_n, p.N) : (p.c_xlayout ? x_elem_offset<T>(e_m, e_n, p.N) : c_off(p, e_m, e_n));
        if constexpr (sizeof(TC) == 4) {
            *(f32x4*)((float*)C + at) = v;
        } else {
            union { bf16_t h[4]; uint2 u; } pk;
            pk.h[0] = (bf16_t)v[0]; pk.h[1] = (bf16_t)v[1]; pk.h[2] = (bf16_t)v[2]; pk.h[3] = (bf16_t)v[3];
            *(uint2*)((bf16_t*)C + at) = pk.u;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (e_n + r >= p.N) break;
            C[c_off(p, e_m, e_n) + r] = from_f32<TC>(v[r]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Skinny form with the LayerNorm in front of the linear layer fused in.  A is the f32 residual stream x in the
// step's x layout, W the layer's own weights (fragment-major, compute dtype), and
//     LN(x) W^T + b  =  rstd * ((x * gamma) W^T - mean * wsum) + b',      b' = b + W beta,  wsum[n] = sum_k gamma[k] W[n][k]
// so the product runs on x scaled by gamma as it is loaded, and the row statistics -- which the workgroup gets for
// free from the very operand blocks it loads, every wave summing its K slice -- enter only in the epilogue.  gamma
// is applied to the f32 operand, not folded into the weights: W stays the rounding of the model's weight, the same
// matrix the multi-token kernels multiply with.  For bf16 compute x * gamma is split into a high and a low bf16 part
// (two MFMAs per block, the matrix cores idle in this kernel anyway), which keeps ~16 mantissa bits: more than
// rounding a normalised row to bf16 keeps.
template <typename T, typename TC, int MT, int NT>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_step_ln_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int KS = 4 * E;
    constexpr int AH = sizeof(T) == 2 ? 2 : 1;          // 16-byte pieces of x per operand chunk
    constexpr int A_STEP = 1024 * AH;                    // bytes of x per (row block, K step)
    f32x4* part = (f32x4*)smem;                          // [SK_WAVES][MT * NT][64]
    float* stat = (float*)(part + SK_WAVES * MT * NT * 64);   // [SK_WAVES][MT][16][2]
    float* gam = stat + SK_WAVES * MT * 16 * 2;               // [K] LayerNorm scale
    if ((int)blockIdx.x >= p.pf_first) {
        if (p.pf) prefetch_following_weights<T>(p, (int)blockIdx.x - p.pf_first);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * (NT * 16), m0 = blockIdx.y * (MT * 16);
    const int steps_total = p.K / KS;
    const int s_begin = (steps_total * wave) / SK_WAVES, s_end = (steps_total * (wave + 1)) / SK_WAVES;
    const int n_blocks = (p.N + 15) >> 4, m_blocks = (p.M + 15) >> 4;
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((long)n_blocks * 16 * p.K * (int)sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((long)m_blocks * 16 * p.K * 4), 0x00020000);
    int woff[NT], aoff[MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) woff[nt] = min((n0 >> 4) + nt, n_blocks - 1) * steps_total * 1024 + lane * 16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) aoff[mt] = min((m0 >> 4) + mt, m_blocks - 1) * steps_total * A_STEP + lane * 16;

    // epilogue operands of the tile this wave finishes, requested up front
    static_assert(MT * NT <= SK_WAVES, "one output tile per wave");
    TC* __restrict__ C = (TC*)p.C;
    const int e_mt = wave / NT, e_nt = wave % NT;
    const int e_m = m0 + e_mt * 16 + fr, e_n = n0 + e_nt * 16 + fg * 4;
    const bool e_live = wave < MT * NT && e_m < p.M && e_n < p.N;
    f32x4 e_bias = {0.f, 0.f, 0.f, 0.f}, e_wsum = {0.f, 0.f, 0.f, 0.f};
    if (e_live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e_bias[r] = (p.bias && e_n + r < p.N) ? p.bias[e_n + r] : 0.f;
            e_wsum[r] = (e_n + r < p.N) ? p.wsum[e_n + r] : 0.f;
        }
    }

    f32x4 acc[MT][NT];
    float sx[MT], sxx[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        sx[mt] = 0.f; sxx[mt] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto ld = [](const __amdgpu_buffer_rsrc_t& r, int voff, int soff) -> uint4 {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    uint4 wv[2][SK_ROUND_LN][NT], xv[2][SK_ROUND_LN][MT][AH];
    auto load_round = [&](int buf, int s) {
#pragma unroll
        for (int r = 0; r < SK_ROUND_LN; ++r) {
            const int st = min(s + r, s_end - 1);   // steps past the end re-read the last one (unused)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wv[buf][r][nt] = ld(rsW, woff[nt], st * 1024);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int h = 0; h < AH; ++h) xv[buf][r][mt][h] = ld(rsA, aoff[mt] + h * 1024, st * A_STEP);
        }
    };
    auto mma_round = [&](int buf, int s) {
#pragma unroll
        for (int r = 0; r < SK_ROUND_LN; ++r) {
            if (s + r < s_end) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if constexpr (sizeof(T) == 4) {
                        const uint4 a = xv[buf][r][mt][0];
                        const float f0 = __uint_as_float(a.x), f1 = __uint_as_float(a.y), f2 = __uint_as_float(a.z), f3 = __uint_as_float(a.w);
                        sx[mt] += (f0 + f1) + (f2 + f3);
                        sxx[mt] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
                        const f32x4 g = *(const f32x4*)(gam + (s + r) * KS + fg * 4);
                        const uint4 ag = make_uint4(__float_as_uint(f0 * g[0]), __float_as_uint(f1 * g[1]), __float_as_uint(f2 * g[2]),
                                                    __float_as_uint(f3 * g[3]));
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) mma16<T>(wv[buf][r][nt], ag, acc[mt][nt]);
                    } else {
                        const uint4 a0 = xv[buf][r][mt][0], a1 = xv[buf][r][mt][AH - 1];
                        const float f[8] = {__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z), __uint_as_float(a0.w),
                                            __uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z), __uint_as_float(a1.w)};
                        const f32x4 g0 = *(const f32x4*)(gam + (s + r) * KS + fg * 8), g1 = *(const f32x4*)(gam + (s + r) * KS + fg * 8 + 4);
                        const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                        union { bf16_t h[8]; uint4 u; } hi, lo;
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float fgm = f[e] * g[e];
                            hi.h[e] = (bf16_t)fgm;
                            lo.h[e] = (bf16_t)(fgm - (float)hi.h[e]);
                            s1 += f[e];
                            s2 = fmaf(f[e], f[e], s2);
                        }
                        sx[mt] += s1;
                        sxx[mt] += s2;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            mma16<T>(wv[buf][r][nt], hi.u, acc[mt][nt]);
                            mma16<T>(wv[buf][r][nt], lo.u, acc[mt][nt]);
                        }
                    }
                }
            }
        }
    };
    if (s_begin < s_end) load_round(0, s_begin);
    // the scale vector goes through LDS (every lane needs the 4 or 8 entries of its K chunk at every step); the
    // first round of operand loads is already in flight behind this barrier
    for (int i = tid; i < (p.K >> 2); i += 64 * SK_WAVES) ((f32x4*)gam)[i] = ((const f32x4*)p.gamma)[i];
    __syncthreads();
    if (s_begin < s_end) {
        for (int s = s_begin; s < s_end; s += 2 * SK_ROUND_LN) {
            if (s + SK_ROUND_LN < s_end) load_round(1, s + SK_ROUND_LN);
            __builtin_amdgcn_sched_barrier(0);
            mma_round(0, s);
            __builtin_amdgcn_sched_barrier(0);
            if (s + SK_ROUND_LN < s_end) {
                if (s + 2 * SK_ROUND_LN < s_end) load_round(0, s + 2 * SK_ROUND_LN);
                __builtin_amdgcn_sched_barrier(0);
                mma_round(1, s + SK_ROUND_LN);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) part[(wave * MT * NT + mt * NT + nt) * 64 + lane] = acc[mt][nt];
        // row sums over the four 16-lane groups (the four chunks of a step), one (sum, sum of squares) per row
        float a = sx[mt], b = sxx[mt];
        a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
        a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
        if (fg == 0) {
            stat[((wave * MT + mt) * 16 + fr) * 2 + 0] = a;
            stat[((wave * MT + mt) * 16 + fr) * 2 + 1] = b;
        }
    }
    __syncthreads();
    if (!e_live) return;
    f32x4 t = part[wave * 64 + lane];
#pragma unroll
    for (int w = 1; w < SK_WAVES; ++w) t += part[(w * MT * NT + wave) * 64 + lane];
    float S1 = 0.f, S2 = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) {
        S1 += stat[((w * MT + e_mt) * 16 + fr) * 2 + 0];
        S2 += stat[((w * MT + e_mt) * 16 + fr) * 2 + 1];
    }
    const float inv_k = 1.0f / (float)p.K;
    const float mean = S1 * inv_k;
    const float var = fmaxf(S2 * inv_k - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = rstd * (t[r] - mean * e_wsum[r]) + e_bias[r];
        if (p.act == 1) v[r] = gelu_erf(v[r]);
    }
    if (e_n + 3 < p.N && (p.c_packed || (p.ldc & 3) == 0)) {
        const long at = p.c_packed ? frag_elem_offset<TC>(e_m, e_n, p.N) : c_off(p, e_m, e_n);
        if constexpr (sizeof(TC) == 4) {
            *(f32x4*)((float*)C + at) = v;
        } else {
            union { bf16_t h[4]; uint2 u; } pk;
            pk.h[0] = (bf16_t)v[0]; pk.h[1] = (bf16_t)v[1]; pk.h[2] = (bf16_t)v[2]; pk.h[3] = (bf16_t)v[3];
            *(uint2*)((bf16_t*)C + at) = pk.u;
            if (p.C_lo) {   // split output: what rounding to bf16 dropped goes to the lo plane
                union { bf16_t h[4]; uint2 u; } lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) lo.h[r] = (bf16_t)(v[r] - (float)pk.h[r]);
                *(uint2*)((bf16_t*)p.C_lo + at) = lo.u;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (e_n + r >= p.N) break;
            C[c_off(p, e_m, e_n) + r] = from_f32<TC>(v[r]);
        }
    }
}

template <typename T>
__global__ void pack_frag_kernel(const T* __restrict__ src, long ld, T* __restrict__ dst, int R, int K)
{
    constexpr int E = 16 / (int)sizeof(T);
    const int chunks = K / E;
    const long n = (long)((R + 15) & ~15) * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / chunks;
        const int c = (int)(i % chunks);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < R) v = *(const uint4*)(src + r * ld + c * E);
        *(uint4*)(dst + frag_elem_offset<T>(r, c * E, K)) = v;
    }
}

template <typename T, typename TC, int MT, int NT> int launch_skinny_mt(const GemmParams& p, hipStream_t s)
{
    const size_t lds = (size_t)SK_WAVES * MT * NT * 64 * sizeof(f32x4);
    GemmParams q = p;
    const dim3 grid = grid_with_prefetch(q, (p.N + NT * 16 - 1) / (NT * 16), (p.M + MT * 16 - 1) / (MT * 16));
    if (p.ab_packed) hipLaunchKernelGGL((gemm_skinny_kernel<T, TC, MT, NT, true>), grid, dim3(64 * SK_WAVES), lds, s, q);
    else hipLaunchKernelGGL((gemm_skinny_kernel<T, TC, MT, NT, false>), grid, dim3(64 * SK_WAVES), lds, s, q);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

template <typename T, typename TC, int MT, int NT> int launch_skinny_hilo(const GemmParams& p, hipStream_t s)
{
    const size_t lds = (size_t)SK_WAVES * MT * NT * 64 * sizeof(f32x4);
    GemmParams q = p;
    const dim3 grid = grid_with_prefetch(q, (p.N + NT * 16 - 1) / (NT * 16), (p.M + MT * 16 - 1) / (MT * 16));
    hipLaunchKernelGGL((gemm_skinny_kernel<T, TC, MT, NT, true, true>), grid, dim3(64 * SK_WAVES), lds, s, q);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

template <typename T, typename TC, int MT, int NT> int launch_step_ln_mt(const GemmParams& p, hipStream_t s)
{
    const size_t lds = (size_t)SK_WAVES * MT * NT * 64 * sizeof(f32x4) + (size_t)SK_WAVES * MT * 16 * 2 * sizeof(float) + (size_t)p.K * sizeof(float);
    GemmParams q = p;
    const dim3 grid = grid_with_prefetch(q, (p.N + NT * 16 - 1) / (NT * 16), (p.M + MT * 16 - 1) / (MT * 16));
    static PerDeviceFlag attr_set;
    if (lds > 64 * 1024 && attr_set.pending()) {
        UW_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_step_ln_kernel<T, TC, MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set.mark();
    }
    hipLaunchKernelGGL((gemm_step_ln_kernel<T, TC, MT, NT>), grid, dim3(64 * SK_WAVES), lds, s, q);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

// 16-column tiles per workgroup of the LayerNorm-fused form.  Measured at M = 128, K = 768: 16 rows x 48 columns per
// workgroup for the wide products (6.8 / 7.5 us at N = 2304 / 3072 against 7.6 / 8.2 us with 32 x 64), 16 x 32 for
// N = K, which is also best for 32 rows at every width tried (512, 768, 1280)
static int step_ln_column_tiles(int M, int N)
{
    if (M <= 16) return N >= 2048 ? 2 : 1;
    return (N >= 2048 && M > 64) ? 3 : 2;
}
// ... of the split-activation form: up to 32 rows 16 x 16 tiles are fastest (the grid is still under one workgroup per
// CU); at M = 128 16 rows x 32 columns beats 32 x 16 by 8-11 %
static int hilo_column_tiles(int M) { return M <= 64 ? 1 : 2; }
// ... of the plain form: few rows per workgroup keep its activation traffic small; wide N takes more tiles per
// workgroup so the grid stays near one workgroup per CU
static int plain_column_tiles(int M, int N)
{
    if (M <= 16) return N >= 2048 ? 2 : 1;
    return (N >= 2048 && M > 32) ? 2 : 1;
}

template <typename T, typename TC> int launch_step_ln(const GemmParams& p, hipStream_t s)
{
    constexpr int KS = 64 / (int)sizeof(T);
    UW_REQUIRE(p.M > 0 && p.M <= 128 && p.N > 0 && p.K > 0 && p.K <= 4096 && p.batch == 1 && p.ab_packed && p.wsum && p.gamma && !p.residual && !p.c_group,
               "LayerNorm-fused gemm: unsupported arguments (M=%d)", p.M);
    UW_REQUIRE(p.K % KS == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0, "LayerNorm-fused gemm: K=%d or alignment", p.K);
    UW_REQUIRE(!p.c_packed || (sizeof(TC) == sizeof(T) && p.N % KS == 0), "LayerNorm-fused gemm: packed output needs N=%d a multiple of %d", p.N, KS);
    UW_REQUIRE(!p.C_lo || (p.c_packed && sizeof(TC) == 2), "LayerNorm-fused gemm: a lo plane goes with packed bf16 output");
    // measured on MI355X at M = 128: the f32 operand and its bf16 split are amortised over four column tiles when
    // N is wide; a narrow N keeps 16-row workgroups so that the grid still covers most CUs
#ifdef UW_TUNE   // development builds only (tools/skinny_bench.hip): tile shape from the environment
    if (getenv("UW_LN_TILE")) {
        const int t = atoi(getenv("UW_LN_TILE"));
        if (t == 11) return launch_step_ln_mt<T, TC, 1, 1>(p, s);
        if (t == 12) return launch_step_ln_mt<T, TC, 1, 2>(p, s);
        if (t == 13) return launch_step_ln_mt<T, TC, 1, 3>(p, s);
        if (t == 14) return launch_step_ln_mt<T, TC, 1, 4>(p, s);
        if (t == 23) return launch_step_ln_mt<T, TC, 2, 3>(p, s);
        if (t == 21) return launch_step_ln_mt<T, TC, 2, 1>(p, s);
        if (t == 22) return launch_step_ln_mt<T, TC, 2, 2>(p, s);
        if (t == 24) return launch_step_ln_mt<T, TC, 2, 4>(p, s);
        if (t == 41) return launch_step_ln_mt<T, TC, 4, 1>(p, s);
        if (t == 42) return launch_step_ln_mt<T, TC, 4, 2>(p, s);
    }
#endif
    const int nt = step_ln_column_tiles(p.M, p.N);
    if (nt == 1) return launch_step_ln_mt<T, TC, 1, 1>(p, s);
    if (nt == 3) return launch_step_ln_mt<T, TC, 1, 3>(p, s);
    return launch_step_ln_mt<T, TC, 1, 2>(p, s);
}

template <typename T, typename TC> int launch_skinny(const GemmParams& p, hipStream_t s)
{
    constexpr int KS = 64 / (int)sizeof(T);
    UW_REQUIRE(p.M > 0 && p.M <= 128 && p.N > 0 && p.K > 0 && p.batch == 1, "skinny gemm: M=%d outside 1..128", p.M);
    UW_REQUIRE(p.K % KS == 0, "skinny gemm: K=%d must be a multiple of %d", p.K, KS);
    UW_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0 &&
                   (p.ab_packed || ((p.lda * sizeof(T)) % 16 == 0 && (p.ldb * sizeof(T)) % 16 == 0)),
               "skinny gemm: operand rows must be 16-byte aligned");
    const long w_rows = p.ab_packed ? (long)((p.N + 15) & ~15) * p.K : (long)p.N * p.ldb;
    const long a_rows = p.ab_packed ? (long)((p.M + 15) & ~15) * p.K : (long)p.M * p.lda;
    UW_REQUIRE(w_rows * (long)sizeof(T) < (1l << 31) && a_rows * (long)sizeof(T) < (1l << 31), "skinny gemm: operands must be under 2 GiB");
    UW_REQUIRE(!p.c_xlayout || (sizeof(TC) == 4 && p.N % KS == 0 && !p.c_group && !p.c_packed && (!p.residual || p.ldr == p.ldc)),
               "skinny gemm: x-layout output needs f32 C, N=%d a multiple of %d and a residual of the same shape", p.N, KS);
    UW_REQUIRE(!p.c_packed || (sizeof(TC) == sizeof(T) && p.N % KS == 0 && !p.c_group),
               "skinny gemm: packed output needs the compute dtype and N=%d a multiple of %d", p.N, KS);
    if (p.A_lo) {
        if constexpr (sizeof(T) == 2) {
            UW_REQUIRE(p.ab_packed && ((uintptr_t)p.A_lo % 16) == 0, "skinny gemm: split A needs fragment-major operands");
#ifdef UW_TUNE
            if (getenv("UW_SK_TILE")) {
                const int t = atoi(getenv("UW_SK_TILE"));
                if (t == 11) return launch_skinny_hilo<T, TC, 1, 1>(p, s);
                if (t == 12) return launch_skinny_hilo<T, TC, 1, 2>(p, s);
                if (t == 13) return launch_skinny_hilo<T, TC, 1, 3>(p, s);
                if (t == 14) return launch_skinny_hilo<T, TC, 1, 4>(p, s);
                if (t == 21) return launch_skinny_hilo<T, TC, 2, 1>(p, s);
                if (t == 22) return launch_skinny_hilo<T, TC, 2, 2>(p, s);
                if (t == 41) return launch_skinny_hilo<T, TC, 4, 1>(p, s);
            }
#endif
            if (hilo_column_tiles(p.M) == 1) return launch_skinny_hilo<T, TC, 1, 1>(p, s);
            return launch_skinny_hilo<T, TC, 1, 2>(p, s);
        } else {
            UW_REQUIRE(false, "skinny gemm: split A is a bf16 form");
        }
    }
    const int nt = plain_column_tiles(p.M, p.N);
    if (p.M <= 16) return nt == 2 ? launch_skinny_mt<T, TC, 1, 2>(p, s) : launch_skinny_mt<T, TC, 1, 1>(p, s);
    if (nt == 2) return launch_skinny_mt<T, TC, 4, 2>(p, s);
    return launch_skinny_mt<T, TC, 2, 1>(p, s);
}

}  // namespace

int uw_gemm_skinny_block_columns(int dtype, int M, int N, int ln_fused, int hilo)
{
    if (ln_fused) return 16 * step_ln_column_tiles(M, N);
    if (hilo && dtype == UW_BF16) return 16 * hilo_column_tiles(M);
    return 16 * plain_column_tiles(M, N);
}

int uw_gemm_skinny(int dtype, int out_f32, const GemmParams& p, hipStream_t s)
{
    if (p.ln_fused) {
        if (dtype == UW_F32) return launch_step_ln<float, float>(p, s);
        if (out_f32) return launch_step_ln<bf16_t, float>(p, s);
        return launch_step_ln<bf16_t, bf16_t>(p, s);
    }
    if (dtype == UW_F32) return launch_skinny<float, float>(p, s);
    if (out_f32) return launch_skinny<bf16_t, float>(p, s);
    return launch_skinny<bf16_t, bf16_t>(p, s);
}

int uw_gemm_rows_hilo(const GemmParams& p, hipStream_t s)
{
    constexpr int KT = K3_ROW / 2;
    UW_REQUIRE(p.A && p.B && p.C && p.M > 0 && p.M <= HLM && p.N > 0 && p.K > 0 && p.K % KT == 0 && p.lda >= 2l * p.K && p.ldb >= p.K,
               "gemm_rows_hilo: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    UW_REQUIRE((p.lda * 2) % 16 == 0 && (p.ldb * 2) % 16 == 0 && (p.K * 2) % 16 == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0,
               "gemm_rows_hilo: operand rows must be 16-byte aligned");
    UW_REQUIRE((long)HLM * p.lda * 2 < (1l << 32) && (long)HLN * p.ldb * 2 < (1l << 32), "gemm_rows_hilo: row stride too large");
    const size_t lds = (size_t)HL_NSTAGE * HL_STAGE;
    static PerDeviceFlag attr_set;
    if (attr_set.pending()) {
        UW_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_rows_hilo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.mark();
    }
    hipLaunchKernelGGL(gemm_rows_hilo_kernel, dim3((p.N + HLN - 1) / HLN), dim3(512), lds, s, p);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_logits_rows16_blocks(int V) { return (V + LR_COLS - 1) / LR_COLS; }

int uw_logits_rows16(const LogitsRows16Params& p, hipStream_t s)
{
    UW_REQUIRE(p.x && p.gamma && p.beta && p.W && (p.logits || p.cand_val) && p.M > 0 && p.M <= 16 && p.V > 0 && p.d > 0 && p.d % 32 == 0 &&
                   p.d <= 32 * LR_WAVES * LR_MAXS,
               "logits_rows16: bad arguments (M=%d V=%d d=%d)", p.M, p.V, p.d);
    UW_REQUIRE(!p.cand_val || p.cand_idx, "logits_rows16: candidate values without indices");
    UW_REQUIRE(((uintptr_t)p.x % 16) == 0 && ((uintptr_t)p.W % 16) == 0 && ((uintptr_t)p.gamma % 16) == 0 && ((uintptr_t)p.beta % 16) == 0,
               "logits_rows16: operands must be 16-byte aligned");
    const size_t lds = (size_t)(p.d / 32) * 2048;
    static PerDeviceFlag attr_set;
    if (lds > 48 * 1024 && attr_set.pending()) {
        UW_HIP_CHECK(hipFuncSetAttribute((const void*)logits_rows16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * LR_WAVES * LR_MAXS * 64));
        attr_set.mark();
    }
    hipLaunchKernelGGL(logits_rows16_kernel, dim3(uw_logits_rows16_blocks(p.V)), dim3(64 * LR_WAVES), lds, s, p);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_pack_frag(int dtype, const void* src, long ld, void* dst, int R, int K, hipStream_t s)
{
    const int E = dtype == UW_F32 ? 4 : 8;
    UW_REQUIRE(src && dst && R > 0 && K > 0 && K % (4 * E) == 0 && (ld * (16 / E)) % 16 == 0, "pack_frag: bad shape R=%d K=%d", R, K);
    const long n = (long)((R + 15) & ~15) * (K / E);
    const int grid = (int)std::min<long>((n + 255) / 256, 65535);
    if (dtype == UW_F32) hipLaunchKernelGGL(pack_frag_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, ld, (float*)dst, R, K);
    else hipLaunchKernelGGL(pack_frag_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)src, ld, (bf16_t*)dst, R, K);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

template <typename T>
__global__ void pack_kblock_kernel(const T* __restrict__ src, T* __restrict__ dst, long R, int K)
{
    constexpr int E = 16 / (int)sizeof(T), KT = 4 * E;
    const long n = R * (K / E);                      // 16-byte chunks
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / (K / E);
        const int k = (int)(i % (K / E)) * E;
        *(uint4*)(dst + ((long)(k / KT) * R + r) * KT + k % KT) = *(const uint4*)(src + r * K + k);
    }
}

int uw_pack_kblock(int dtype, const void* src, void* dst, long R, int K, hipStream_t s)
{
    const int kt = dtype == UW_F32 ? 16 : 32;
    UW_REQUIRE(src && dst && src != dst && R > 0 && K > 0 && K % kt == 0, "pack_kblock: K=%d must be a multiple of %d", K, kt);
    const long n = R * (K / (kt / 4));
    const int grid = (int)std::min<long>((n + 255) / 256, 65535);
    if (dtype == UW_F32) hipLaunchKernelGGL(pack_kblock_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, R, K);
    else hipLaunchKernelGGL(pack_kblock_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, R, K);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_gemm(int dtype, int out_f32, const GemmParams& p, hipStream_t s)
{
    UW_REQUIRE(!p.C2 || ((p.kstep_a || p.kstep_b) && !out_f32 && p.c2_col0 % 64 == 0 && p.ldc2 > 0),
               "gemm: the transposed second output exists in the K-blocked two-workgroup form only, from a 64-column boundary");
    if (p.kstep_a || p.kstep_b || p.c_kblock) {
        // K-blocked operands exist only in the two-workgroup tile kernel
        const int kt = dtype == UW_F32 ? 16 : 32;
        UW_REQUIRE(!(p.act && p.residual) && !p.kwrap_a && !p.kwrap_b && !p.c_split && (!p.c_kblock || (!p.c_group && p.batch == 1)) &&
                       (!p.kstep_a || p.lda == kt) && (!p.kstep_b || p.ldb == kt),
                   "gemm: unsupported combination with K-blocked operands");
        if (dtype == UW_F32) return launch_duo<float, float>(p, s);
        if (out_f32) return launch_duo<bf16_t, float>(p, s);
        return launch_duo<bf16_t, bf16_t>(p, s);
    }
    static const int big_mode = getenv("UW_GEMM_BIG") ? atoi(getenv("UW_GEMM_BIG")) : 2;   // 0: 128x128 only, 1: one WG/CU, 2: two WGs/CU
    const bool use_big = big_mode != 0;
    // the two-workgroup kernel folds the residual into the accumulator start, which is only right without an activation
    if (big_mode == 2 && p.M >= 512 && !(p.act && p.residual)) {
        if (dtype == UW_F32) return launch_duo<float, float>(p, s);
        if (out_f32) return launch_duo<bf16_t, float>(p, s);
        return launch_duo<bf16_t, bf16_t>(p, s);
    }
    if (use_big && p.M >= 512) {   // enough rows to fill 256-row tiles
        if (dtype == UW_F32) return launch_big<float, float>(p, s);
        if (out_f32) return launch_big<bf16_t, float>(p, s);
        return launch_big<bf16_t, bf16_t>(p, s);
    }
    if (dtype == UW_F32) return launch<float, float>(p, s);
    if (out_f32) return launch<bf16_t, float>(p, s);
    return launch<bf16_t, bf16_t>(p, s);
}
