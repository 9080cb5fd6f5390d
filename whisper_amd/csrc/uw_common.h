// Shared device/host helpers for the gfx950 transcription kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define UW_OK 0
#define UW_ERR_ARG 1
#define UW_ERR_HIP 2
#define UW_ERR_STATE 3
#define UW_ERR_NOPATH 4

void uw_set_error(const char* fmt, ...);

#define UW_HIP_CHECK(expr)                                                                            \
    do {                                                                                              \
        hipError_t e__ = (expr);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            uw_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return UW_ERR_HIP;                                                                        \
        }                                                                                             \
    } while (0)

#define UW_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            uw_set_error(__VA_ARGS__); \
            return UW_ERR_ARG;         \
        }                              \
    } while (0)

#define UW_LAUNCH_CHECK() UW_HIP_CHECK(hipGetLastError())

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// LDS byte address of a __shared__ pointer, as a provably wave-uniform scalar
__device__ __forceinline__ unsigned lds_addr_of(const void* p)
{
    return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(p));
}
// 16-byte-per-lane global -> LDS DMA (lane l lands at lds_addr + 16 * l).  Issued from inline asm so that the
// compiler does not track it: with the builtin form hipcc drains every outstanding DMA (vmcnt(0)) before the
// next ds_read, which serialises a multi-stage ring.  The caller orders it: counted s_waitcnt vmcnt(N), then a
// barrier, then the reads.  M0 is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

// Same DMA with the address split into a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset.
// hipcc tracks the VGPR inputs of an asm statement as pending until vmcnt retires them; a per-lane 64-bit address
// recomputed every iteration lives in temporaries that the next ds_read overwrites, so the compiler inserts a
// vmcnt(0) there and the ring loses its prefetch.  With this form the only VGPR input is a loop-invariant offset
// that is never overwritten; the advancing part of the address lives in SGPRs.
__device__ __forceinline__ void glds16_sv(const void* sbase, unsigned voff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

// s_waitcnt vmcnt(0) that the COMPILER sees (expcnt and lgkmcnt left alone).  Ordinary loads issued before a loop
// whose first use is inside the loop make hipcc place a vmcnt wait at that use in every iteration -- and since it
// does not count the asm-issued LDS DMAs, that wait is vmcnt(0) and drains the ring each time round.  Retiring
// such loads with this builtin in front of the loop removes the in-loop waits.
__device__ __forceinline__ void retire_vector_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2).  This bijection hands each
// XCD a CONTIGUOUS run of tile indices, so tiles that share an operand panel hit the same L2 (speed only).
__device__ __forceinline__ int xcd_contiguous_tile(int b, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// low / high bf16 of a packed 32-bit word, widened exactly
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// A 16-byte K-chunk of an MFMA operand: 8 bf16 or 4 f32.  One call consumes the chunk held by every
// lane: for the 16x16 shapes the four 16-lane groups hold four consecutive chunks of K, for the 32x32
// shapes the two 32-lane halves hold two.  The f32 forms split a chunk over four K=4 / K=2 MFMAs, which
// permutes K inside the chunk group identically for both operands.
template <typename T> __device__ __forceinline__ void mma16(const uint4& a, const uint4& b, f32x4& acc);
template <> __device__ __forceinline__ void mma16<bf16_t>(const uint4& a, const uint4& b, f32x4& acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(const uint4& a, const uint4& b, f32x4& acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ void mma32(const uint4& a, const uint4& b, f32x16& acc);
template <> __device__ __forceinline__ void mma32<bf16_t>(const uint4& a, const uint4& b, f32x16& acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma32<float>(const uint4& a, const uint4& b, f32x16& acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// "Fragment-major" operand layout of the decode-step GEMMs.  A [R][K] matrix is cut into blocks of 16 rows x
// 64 bytes of K; inside a block the 16-byte chunk c (0..3) of row r sits at lane position c * 16 + (r & 15), i.e.
// exactly where the MFMA 16x16 operand wants it, so a wave fetches one operand block as 1 KiB of consecutive
// bytes instead of sixteen 64-byte pieces of sixteen rows.  Blocks are ordered [row block][K block]; a band of
// 16 rows occupies the same bytes as in the row-major layout.  Returns the element offset of X[r][k].
template <typename T> __host__ __device__ __forceinline__ long frag_elem_offset(long r, int k, int K)
{
    constexpr int E = 16 / (int)sizeof(T);
    const int chunk = k / E, steps = K / (4 * E);
    return (((((r >> 4) * steps + (chunk >> 2)) << 6) + ((chunk & 3) << 4) + (r & 15)) * E) + (k % E);
}

// Layout of the f32 residual stream inside the decode step.  The step's LayerNorm-fused GEMMs read x itself as
// their MFMA operand (normalisation is applied to the product, see gemm_step_ln_kernel), so x is kept the way
// an operand block is fetched: for f32 compute the plain fragment-major layout; for bf16 compute a lane needs 8
// consecutive columns (32 bytes of f32) per 32-column step, stored as two 1 KiB halves of 4 columns per lane.
template <typename T> __host__ __device__ __forceinline__ long x_elem_offset(long m, int k, int K)
{
    if constexpr (sizeof(T) == 4) {
        return frag_elem_offset<float>(m, k, K);
    } else {
        const int steps = K >> 5;
        return ((((((m >> 4) * steps + (k >> 5)) << 1) + ((k >> 2) & 1)) << 6) + (((k >> 3) & 3) << 4) + (m & 15)) * 4 + (k & 3);
    }
}

// erf by the rational form of Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 before rounding): one reciprocal,
// one exponential and five FMAs instead of the library routine's branches; the GELU of a GEMM epilogue is
// evaluated once per output element, which made the library erf as expensive as the tile's MFMAs.
__device__ __forceinline__ float erf_as(float x)
{
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    const float r = 1.0f - poly * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// First index of the maximum of row[0..V) (numpy argmax tie rule), computed by a 1024-thread workgroup: 16-byte loads,
// eight independent loads in flight per thread.  sv / si: 16-entry LDS scratch.  The result is valid in thread 0 (after
// the internal barrier); row must be 16-byte aligned for the vector path.  A NaN never wins the comparison; *any_nan
// (valid in every thread) tells the caller that the row held one, so it can refuse the row instead of trusting the index.
// tail of the block-wide arg-max (1024 threads): every thread brings its best (value, index, saw-a-NaN); the first index wins ties
__device__ __forceinline__ int block_argmax_finish(float best, int arg, bool bad, float* sv, int* si, bool* any_nan)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(arg, o, 64);
        if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = arg; }
    const int nan_seen = __syncthreads_or(bad ? 1 : 0);
    if (any_nan) *any_nan = nan_seen != 0;
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < arg)) { best = sv[w]; arg = si[w]; }
    }
    return (arg == 0x7fffffff) ? 0 : arg;
}

__device__ __forceinline__ int block_argmax_1024(const float* __restrict__ row, int V, float* sv, int* si, bool* any_nan = nullptr)
{
    float best = -INFINITY;
    int arg = 0x7fffffff;
    bool bad = false;
    auto take = [&](float v, int i) { bad |= (v != v); if (v > best || (v == best && i < arg)) { best = v; arg = i; } };
    const bool vec = (((uintptr_t)row & 15) == 0);
    const int V4 = vec ? (V >> 2) : 0;
    for (int i0 = threadIdx.x; i0 < V4; i0 += 8192) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 1024;
            v[u] = (i < V4) ? ((const f32x4*)row)[i] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) take(v[u][k], (i0 + u * 1024) * 4 + k);
    }
    for (int i = V4 * 4 + threadIdx.x; i < V; i += 1024) take(row[i], i);
    return block_argmax_finish(best, arg, bad, sv, si, any_nan);
}

// the same over n (value, index) candidates, one per column block of the row (a NaN value marks a block that holds one)
__device__ __forceinline__ int block_argmax_candidates(const float* __restrict__ val, const int32_t* __restrict__ idx, int n, float* sv,
                                                       int* si, bool* any_nan = nullptr)
{
    float best = -INFINITY;
    int arg = 0x7fffffff;
    bool bad = false;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float v = val[i];
        const int a = idx[i];
        bad |= (v != v);
        if (v > best || (v == best && a < arg)) { best = v; arg = a; }
    }
    return block_argmax_finish(best, arg, bad, sv, si, any_nan);
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Function attributes (the dynamic LDS limit) are kept per device: one flag per device ordinal, so a process that
// owns contexts on several GPUs sets the attribute on each of them.
struct PerDeviceFlag {
    std::atomic<bool> done[64];
    PerDeviceFlag() { for (auto& d : done) d.store(false); }
    int dev() const
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return -1;
        return d & 63;
    }
    // true until mark() has been called on the current device; contexts on several host threads may both see true and
    // both set the attribute (harmless), but none launches before its own call has returned
    bool pending() const { const int d = dev(); return d < 0 || !done[d].load(std::memory_order_acquire); }
    void mark() { const int d = dev(); if (d >= 0) done[d].store(true, std::memory_order_release); }
};
