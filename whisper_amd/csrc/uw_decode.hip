// Single-token (KV-cached) decoder attention and the greedy bookkeeping step.
//
// The reference re-runs the whole decoder over the whole prefix for every token (urro_whisper/core.py:938-968)
// and once more per text token to read the cross-attention weights (urro_whisper/align.py:607-623).  Here one
// step touches each cached K/V element once: per generated token and window the cross-attention K/V of all
// layers (Ld * 2 * 1500 * d elements) is the dominant HBM stream, and the cross-attention kernel writes the
// softmax row of the alignment heads straight into the rows buffer that DTW consumes.
//
// Both kernels are bandwidth kernels on the vector ALU: a K/V row of one head is 64 elements; LPR lanes
// (8 for bf16, 16 for f32) each own one 16-byte chunk of a row, so one wave instruction moves 1 KiB.
#include "uw_common.h"
#include "uw_kernels.h"
#include <algorithm>

namespace {

template <typename T> struct Row {
    static constexpr int E = 16 / (int)sizeof(T);   // elements per lane
    static constexpr int LPR = 64 / E;              // lanes per row
    static constexpr int RPI = 64 / LPR;            // rows per wave instruction
};

template <typename T> __device__ __forceinline__ void unpack(const uint4& u, float* f);
template <> __device__ __forceinline__ void unpack<float>(const uint4& u, float* f)
{
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}
template <> __device__ __forceinline__ void unpack<bf16_t>(const uint4& u, float* f)
{
    f[0] = bf16lo(u.x); f[1] = bf16hi(u.x); f[2] = bf16lo(u.y); f[3] = bf16hi(u.y);
    f[4] = bf16lo(u.z); f[5] = bf16hi(u.z); f[6] = bf16lo(u.w); f[7] = bf16hi(u.w);
}

template <typename T> __device__ __forceinline__ float dot_chunk(const uint4& u, const float* q)
{
    float f[Row<T>::E];
    unpack<T>(u, f);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < Row<T>::E; ++e) s = fmaf(f[e], q[e], s);
    return s;
}

// 16-byte load of data that is read once per launch and is far larger than the caches: non-temporal, so it
// does not evict what the other kernels of the step keep in L2 / Infinity Cache
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ uint4 ld_stream(const void* p)
{
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// attention output element: the compute-dtype value, plus (split activations) what rounding it dropped in the lo plane
template <typename T> __device__ __forceinline__ void store_split(void* out, void* out_lo, long at, float v)
{
    const T hi = from_f32<T>(v);
    ((T*)out)[at] = hi;
    if (out_lo) ((T*)out_lo)[at] = from_f32<T>(v - to_f32<T>(hi));
}

// sum over the LPR lanes that share a row
template <typename T> __device__ __forceinline__ float row_sum(float v)
{
#pragma unroll
    for (int o = 1; o < Row<T>::LPR; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Data handed from one workgroup to another INSIDE a launch (the part records and raw scores of the row-split
// cross-attention) goes through device-scope stores and loads, which pass the per-XCD L2 instead of being kept in
// it; the producer waits for its stores before it takes the ticket.  A device-scope fence would do too, but it
// writes back and invalidates the whole L2 of the XCD for every workgroup that executes it (measured on a full
// batch: decode steps 5x slower).  Such a hand-over costs three dependent round trips through the fabric (store
// acknowledged, ticket, load) -- about what the launch it replaces costs, so it only pays where it replaces a launch
// outright and adds no work in front of it.
__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// true for the workgroup that arrives last of `n`; the counter is left at zero for the next launch.  All threads call.
__device__ __forceinline__ bool last_arrival(int32_t* counter, int n, int tid, int* flag)
{
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's stores have been acknowledged
    __syncthreads();
    if (tid == 0) {
        const bool last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n - 1;
        if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}

// ------------------------------------------------------------------------------------------------
// cross-attention for one new token: grid (H, B), NW waves per workgroup (4 when the grid fills the chip several
// times over, more when it does not: the bytes in flight per CU are what covers the HBM latency)
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void cross_decode_kernel(CrossDecodeParams p)
{
    using R = Row<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = (float*)smem;                 // [S rounded up] scores, then probabilities
    const int S = p.S, Spad = (S + 63) & ~63;
    float* red = sc + Spad;                   // [2 * NW]
    float* part = red + 2 * NW;               // [NW waves][64]
    const int h = blockIdx.x, b = blockIdx.y;
    if (p.active && !p.active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane % R::LPR, rin = lane / R::LPR;
    const long ld = p.ld_kv;
    const T* Kh = (const T*)p.K + (long)b * p.kv_sB + (long)h * p.kv_sH + sub * R::E;
    const T* Vh = (const T*)p.V + (long)b * p.kv_sB + (long)h * p.kv_sH + sub * R::E;

    float q[R::E];
    {
        const T* qp = (const T*)p.q + (long)b * p.d + h * 64 + sub * R::E;
#pragma unroll
        for (int e = 0; e < R::E; ++e) q[e] = to_f32<T>(qp[e]);
    }
    // ---- scores ----
    constexpr int U = 8;                               // loads in flight per lane
    const int rows_per_round = NW * R::RPI * U;         // rows covered by the workgroup per unrolled round
    for (int r0 = 0; r0 < S; r0 += rows_per_round) {
        uint4 kv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + (u * NW + wave) * R::RPI + rin;
            kv[u] = ld_stream(Kh + (long)min(r, S - 1) * ld);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + (u * NW + wave) * R::RPI + rin;
            const float s = row_sum<T>(dot_chunk<T>(kv[u], q));
            if (sub == 0 && r < S) sc[r] = s;
        }
    }
    // the first round of V rows is requested now, so HBM stays busy while the workgroup does its softmax
    uint4 vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int r = (u * NW + wave) * R::RPI + rin;
        vv[u] = ld_stream(Vh + (long)min(r, S - 1) * ld);
    }
    __syncthreads();
    // ---- softmax over S ----
    float mx = -INFINITY;
    for (int j = tid; j < S; j += 64 * NW) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.f;
    for (int j = tid; j < S; j += 64 * NW) {
        const float e = __expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[NW + wave] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[NW + w];
    const float inv = 1.0f / tot;
    const int slot = p.sel_slot[h];
    float* row_out = nullptr;
    if (p.rows && slot >= 0) {
        const int ri = p.row_index[b];
        if (ri >= 0 && ri < p.n_rows_max) row_out = p.rows + (long)b * p.rows_sB + (long)slot * p.rows_sSel + (long)ri * S;
    }
    for (int j = tid; j < S; j += 64 * NW) {
        const float pr = sc[j] * inv;
        sc[j] = pr;
        if (row_out) row_out[j] = pr;
    }
    __syncthreads();
    // ---- weighted sum of V ----
    float acc[R::E];
#pragma unroll
    for (int e = 0; e < R::E; ++e) acc[e] = 0.f;
    for (int r0 = 0; r0 < S; r0 += rows_per_round) {
        if (r0 > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + (u * NW + wave) * R::RPI + rin;
                vv[u] = ld_stream(Vh + (long)min(r, S - 1) * ld);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + (u * NW + wave) * R::RPI + rin;
            const float w = (r < S) ? sc[r] : 0.f;
            float f[R::E];
            unpack<T>(vv[u], f);
#pragma unroll
            for (int e = 0; e < R::E; ++e) acc[e] = fmaf(w, f[e], acc[e]);
        }
    }
    // reduce over the RPI row slots of the wave (lanes with equal `sub`), then over the 4 waves
#pragma unroll
    for (int e = 0; e < R::E; ++e) {
#pragma unroll
        for (int o = R::LPR; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (rin == 0) {
#pragma unroll
        for (int e = 0; e < R::E; ++e) part[wave * 64 + sub * R::E + e] = acc[e];
    }
    __syncthreads();
    if (tid < 64) {
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) o += part[w * 64 + tid];
        const long at = p.out_packed ? frag_elem_offset<T>(b, h * 64 + tid, p.d) : (long)b * p.d + h * 64 + tid;
        store_split<T>(p.out, p.out_lo, at, o);
    }
}

// End of a part's workgroup (256 threads): the one that takes the last ticket of its (window, head) merges the parts in
// part order -- maxima, exp-sums, the 64 accumulators -- and turns an alignment head's raw scores into probabilities;
// everything it needs is requested in one go.
template <typename T>
__device__ __forceinline__ void split_finish(const CrossDecodeParams& p, int b, int h, int tid)
{
    __shared__ float s_m[16], s_l[16];
    __shared__ int s_flag;
    if (!last_arrival(p.split_ticket + (long)b * p.H + h, p.n_split, tid, &s_flag)) return;
    const float* ws = p.split_ws + ((long)b * p.H + h) * p.n_split * 66;
    if (tid < p.n_split) { s_m[tid] = ld_dev(ws + tid * 66); s_l[tid] = ld_dev(ws + tid * 66 + 1); }
    float part[16];
#pragma unroll
    for (int z = 0; z < 16; ++z) part[z] = (tid < 64 && z < p.n_split) ? ld_dev(ws + z * 66 + 2 + tid) : 0.f;
    const int slot = p.sel_slot[h];
    float* row = nullptr;
    if (p.rows && slot >= 0) {
        const int ri = p.row_index[b];
        if (ri >= 0 && ri < p.n_rows_max) row = p.rows + (long)b * p.rows_sB + (long)slot * p.rows_sSel + (long)ri * p.S;
    }
    float rv[8];      // raw scores of an alignment head's row, 256 threads x 8 >= 1500 frames (longer rows loop below)
#pragma unroll
    for (int k = 0; k < 8; ++k) rv[k] = (row && tid + k * 256 < p.S) ? ld_dev(row + tid + k * 256) : 0.f;
    __syncthreads();
    float M = -INFINITY;
    for (int z = 0; z < p.n_split; ++z) M = fmaxf(M, s_m[z]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int z = 0; z < 16; ++z) {
        if (z < p.n_split) {
            const float sc = __expf(s_m[z] - M);
            L += s_l[z] * sc;
            o += part[z] * sc;
        }
    }
    const float inv = 1.0f / L;
    if (row) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (tid + k * 256 < p.S) row[tid + k * 256] = __expf(rv[k] - M) * inv;
        for (int j = tid + 2048; j < p.S; j += 256) row[j] = __expf(ld_dev(row + j) - M) * inv;
    }
    if (tid < 64) {
        const long at = p.out_packed ? frag_elem_offset<T>(b, h * 64 + tid, p.d) : (long)b * p.d + h * 64 + tid;
        store_split<T>(p.out, p.out_lo, at, o * inv);
    }
}

// ------------------------------------------------------------------------------------------------
// Small batches: (H, B) workgroups cannot fill 256 CUs, so the 1500 rows of a (window, head) are cut over
// n_split workgroups.  Each leaves (local max, local exp-sum, 64 unnormalised accumulators) and, for an alignment
// head, its raw scores in the rows buffer; the workgroup that finishes last merges the parts (split_finish): no
// second launch, and the merge order does not depend on which one that is.  grid (H, B, n_split), 256 threads.
template <typename T>
__global__ __launch_bounds__(256) void cross_decode_split_kernel(CrossDecodeParams p)
{
    using R = Row<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = (float*)smem;                 // [rows of this part, rounded up]
    const int h = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
    if (p.active && !p.active[b]) return;
    const int per = (((p.S + p.n_split - 1) / p.n_split) + 31) & ~31;   // rows per part, a multiple of 32
    const int s0 = z * per, S = min(p.S - s0, per);                      // this part: rows [s0, s0 + S)
    float* red = sc + per;                    // [8]
    float* part = red + 8;                    // [4 waves][64]
    float* ws = p.split_ws + (((long)b * p.H + h) * p.n_split + z) * 66;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (S <= 0) {
        if (tid < 66) st_dev(ws + tid, tid == 0 ? -INFINITY : 0.f);
        split_finish<T>(p, b, h, tid);
        return;
    }
    const int sub = lane % R::LPR, rin = lane / R::LPR;
    const long ld = p.ld_kv;
    const T* Kh = (const T*)p.K + (long)b * p.kv_sB + (long)h * p.kv_sH + (long)s0 * ld + sub * R::E;
    const T* Vh = (const T*)p.V + (long)b * p.kv_sB + (long)h * p.kv_sH + (long)s0 * ld + sub * R::E;
    float q[R::E];
    {
        const T* qp = (const T*)p.q + (long)b * p.d + h * 64 + sub * R::E;
#pragma unroll
        for (int e = 0; e < R::E; ++e) q[e] = to_f32<T>(qp[e]);
    }
    constexpr int U = 8;
    const int rows_per_round = 4 * R::RPI * U;
    for (int r0 = 0; r0 < S; r0 += rows_per_round) {
        uint4 kv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + (u * 4 + wave) * R::RPI + rin;
            kv[u] = ld_stream(Kh + (long)min(r, S - 1) * ld);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + (u * 4 + wave) * R::RPI + rin;
            const float s = row_sum<T>(dot_chunk<T>(kv[u], q));
            if (sub == 0 && r < S) sc[r] = s;
        }
    }
    // the first round of V rows is requested before the softmax (for up to 8 parts it is the only round)
    uint4 vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int r = (u * 4 + wave) * R::RPI + rin;
        vv[u] = ld_stream(Vh + (long)min(r, S - 1) * ld);
    }
    __syncthreads();
    const int slot = p.sel_slot[h];
    float* row_out = nullptr;
    if (p.rows && slot >= 0) {
        const int ri = p.row_index[b];
        if (ri >= 0 && ri < p.n_rows_max) row_out = p.rows + (long)b * p.rows_sB + (long)slot * p.rows_sSel + (long)ri * p.S + s0;
    }
    float mx = -INFINITY;
    for (int j = tid; j < S; j += 256) {
        mx = fmaxf(mx, sc[j]);
        if (row_out) st_dev(row_out + j, sc[j]);      // raw score; the merge makes it a probability
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < S; j += 256) {
        const float e = __expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    float acc[R::E];
#pragma unroll
    for (int e = 0; e < R::E; ++e) acc[e] = 0.f;
    for (int r0 = 0; r0 < S; r0 += rows_per_round) {
        if (r0 > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + (u * 4 + wave) * R::RPI + rin;
                vv[u] = ld_stream(Vh + (long)min(r, S - 1) * ld);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + (u * 4 + wave) * R::RPI + rin;
            const float w = (r < S) ? sc[r] : 0.f;
            float f[R::E];
            unpack<T>(vv[u], f);
#pragma unroll
            for (int e = 0; e < R::E; ++e) acc[e] = fmaf(w, f[e], acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < R::E; ++e) {
#pragma unroll
        for (int o = R::LPR; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (rin == 0) {
#pragma unroll
        for (int e = 0; e < R::E; ++e) part[wave * 64 + sub * R::E + e] = acc[e];
    }
    __syncthreads();
    if (tid < 64) st_dev(ws + 2 + tid, (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]));
    if (tid == 64) st_dev(ws, mx);
    if (tid == 65) st_dev(ws + 1, (red[4] + red[5]) + (red[6] + red[7]));
    split_finish<T>(p, b, h, tid);
}

// ------------------------------------------------------------------------------------------------
// causal self-attention for one new token: grid (H, B), 256 threads; also appends this token's K/V to the cache.
// Same structure as the cross kernel over the pos+1 cached rows; the row being appended is taken from registers.
// A (window, head) has at most a few hundred cached rows, so a workgroup lives for little more than its memory
// round trips: with NU > 0 the workgroup covers NU x (4 waves x RPI rows) rows (the launcher picks the smallest NU
// that covers the longest window) and requests ALL of its K rows at once, then all of its V rows as soon as the K
// registers are free -- two round trips, the second one under the softmax -- instead of one per 4-load round
// (single-window decoding: 8.5 -> about 5 us per launch; full batches are bandwidth-bound either way).
// NU == 0 is the general loop over rounds.
template <typename T, int NU>
__global__ __launch_bounds__(256) void self_decode_kernel(SelfDecodeParams p)
{
    using R = Row<T>;
    __shared__ float sc[512];
    __shared__ float red[8];
    __shared__ float part[4 * 64];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.active && !p.active[b]) return;
    const int pos = p.pos[b];
    if (pos < 0 || pos >= p.n_ctx || pos >= 512) return;
    const int sub = lane % R::LPR, rin = lane / R::LPR;
    const long ld = p.ld_cache;
    const T* qkv = (const T*)p.qkv + (long)b * 3 * p.d + h * 64 + sub * R::E;
    T* Kc = (T*)p.kcache + (long)b * p.cache_sB + (long)h * p.cache_sH + sub * R::E;
    T* Vc = (T*)p.vcache + (long)b * p.cache_sB + (long)h * p.cache_sH + sub * R::E;
    const uint4 knew = *(const uint4*)(qkv + p.d), vnew = *(const uint4*)(qkv + 2 * p.d);
    if (wave == 0 && rin == 0) {
        *(uint4*)(Kc + (long)pos * ld) = knew;
        *(uint4*)(Vc + (long)pos * ld) = vnew;
    }
    float q[R::E];
#pragma unroll
    for (int e = 0; e < R::E; ++e) q[e] = to_f32<T>(qkv[e]);
    const int n = pos + 1;
    // loads in flight per lane; measured at 128 windows x 12 heads (1536 workgroups of about 60 KB each): 8 per lane, as
    // the cross kernel keeps, is slower (23.1 against 21.6 us per launch), and so is requesting the first V round
    // before the softmax (22.3 us) -- the workgroups are short enough that occupancy counts for more
    constexpr int U = NU > 0 ? NU : 4;
    const int rows_per_round = 4 * R::RPI * U;
    // NU > 0: one round covers max_rows rows; a window beyond the launcher's bound (never in the greedy loop, whose
    // bound is exact) simply takes further rounds
    uint4 vv[U];
    for (int r0 = 0; r0 < n; r0 += rows_per_round) {
        uint4 kv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + (u * 4 + wave) * R::RPI + rin;
            kv[u] = knew;
            if (r < pos) kv[u] = ld_stream(Kc + (long)r * ld);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + (u * 4 + wave) * R::RPI + rin;
            const float s = row_sum<T>(dot_chunk<T>(kv[u], q));
            if (sub == 0 && r < n) sc[r] = s;
        }
    }
    const bool one_round = NU > 0 && n <= rows_per_round;   // block-uniform
    if (one_round) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = (u * 4 + wave) * R::RPI + rin;
            vv[u] = vnew;
            if (r < pos) vv[u] = ld_stream(Vc + (long)r * ld);
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < n; j += 256) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < n; j += 256) {
        const float e = __expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    float acc[R::E];
#pragma unroll
    for (int e = 0; e < R::E; ++e) acc[e] = 0.f;
    for (int r0 = 0; r0 < n; r0 += rows_per_round) {
        if (!one_round) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + (u * 4 + wave) * R::RPI + rin;
                vv[u] = vnew;
                if (r < pos) vv[u] = ld_stream(Vc + (long)r * ld);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + (u * 4 + wave) * R::RPI + rin;
            const float w = (r < n) ? sc[r] * inv : 0.f;
            float f[R::E];
            unpack<T>(vv[u], f);
#pragma unroll
            for (int e = 0; e < R::E; ++e) acc[e] = fmaf(w, f[e], acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < R::E; ++e) {
#pragma unroll
        for (int o = R::LPR; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (rin == 0) {
#pragma unroll
        for (int e = 0; e < R::E; ++e) part[wave * 64 + sub * R::E + e] = acc[e];
    }
    __syncthreads();
    if (tid < 64) {
        const float o = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
        const long at = p.out_packed ? frag_elem_offset<T>(b, h * 64 + tid, p.d) : (long)b * p.d + h * 64 + tid;
        store_split<T>(p.out, p.out_lo, at, o);
    }
}

// ------------------------------------------------------------------------------------------------
// End of a greedy step for one window per workgroup: argmax of its logits (np.argmax, first index on ties),
// urro_whisper/core.py:959-968 (stop on eot without appending, budget) and :991 (text filter), then -- if the window
// goes on -- the embedding of the token just chosen, written as the next step's residual-stream row.
template <typename T>
__global__ __launch_bounds__(1024) void greedy_next_kernel(GreedyState st, const float* __restrict__ logits, long ld, int V, int eot,
                                                           int sot, int ts_begin, int ts_end, float* __restrict__ x,
                                                           const float* __restrict__ tok_emb, const float* __restrict__ pos_emb, int d,
                                                           const float* __restrict__ cand_val, const int32_t* __restrict__ cand_idx, int n_cand)
{
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ int s_next[2];
    const int b = blockIdx.x;
    if (!st.active[b]) return;
    // the window's counters and the next position's embedding row do not depend on the arg-max: they are requested
    // first, so their round trips pass under the scan of the logits row instead of following it one by one
    const int pos0 = st.pos[b], nn0 = st.n_new[b], nt0 = st.n_text[b], dirty0 = st.dirty_from[b], budget0 = st.max_new[b];
    const int n_ctx_rows = st.ld_tok;               // positions 0 .. ld_tok - 1 exist
    f32x4 pe_next = {0.f, 0.f, 0.f, 0.f};
    if ((int)threadIdx.x < (d >> 2)) pe_next = ((const f32x4*)(pos_emb + (long)min(pos0 + 1, n_ctx_rows - 1) * d))[threadIdx.x];
    bool nan_row = false;
    // with candidates (one per 256-column block of the row, left by the logits kernel) the row itself is not read
    const int tok = cand_val ? block_argmax_candidates(cand_val + (long)b * n_cand, cand_idx + (long)b * n_cand, n_cand, sv, si, &nan_row)
                             : block_argmax_1024(logits + (long)b * ld, V, sv, si, &nan_row);
    if (threadIdx.x == 0) {
        bool alive = true;
        int idx = -1;
        if (nan_row) {
            // np.argmax would return the NaN's index and the reference would go on decoding garbage; here the window
            // stops like a decoder run that raised (core.py:978-980): the tokens generated so far are kept
            alive = false;
            if (st.failed) st.failed[b] = 1;
        } else if (tok == eot) {
            alive = false;
        } else {
            idx = pos0 + 1;   // pos = index of the token that produced these logits
            st.tokens[(long)b * st.ld_tok + idx] = tok;
            st.pos[b] = idx;
            const int nn = nn0 + 1;
            st.n_new[b] = nn;
            const bool is_text = tok < sot && !(tok >= ts_begin && tok <= ts_end);
            if (is_text) {
                st.text_ids[(long)b * st.ld_tok + nt0] = tok;
                st.n_text[b] = nt0 + 1;
            } else if (dirty0 < 0) {
                st.dirty_from[b] = nt0;  // rows captured from here on see a context the reference never builds
            }
            if (nn >= budget0) alive = false;
        }
        if (!alive) {
            st.active[b] = 0;
            atomicSub(st.n_active, 1);
        }
        s_next[0] = alive ? idx : -1;
        s_next[1] = tok;
    }
    __syncthreads();
    const int pos = s_next[0];
    if (pos < 0) return;
    const f32x4* e = (const f32x4*)(tok_emb + (long)min(max(s_next[1], 0), V - 1) * d);   // pos == pos0 + 1 here
    if ((int)threadIdx.x < (d >> 2)) *(f32x4*)(x + x_elem_offset<T>(b, threadIdx.x * 4, d)) = e[threadIdx.x] + pe_next;
}

}  // namespace

int uw_cross_attn_decode(int dtype, const CrossDecodeParams& p, hipStream_t s)
{
    UW_REQUIRE(p.q && p.K && p.V && p.out && p.B > 0 && p.H > 0 && p.d == p.H * 64 && p.S > 0 && p.S <= 4096 && p.ld_kv >= 64,
               "cross_attn_decode: bad arguments");
    UW_REQUIRE(!p.rows || p.row_index, "cross_attn_decode: rows without row_index");
    if (p.n_split > 1) {
        UW_REQUIRE(p.split_ws && p.split_ticket && p.n_split <= 16, "cross_attn_decode: split needs a workspace, tickets and at most 16 parts");
        const int per = (((p.S + p.n_split - 1) / p.n_split) + 31) & ~31;
        const size_t lds = (size_t)(per + 8 + 256) * 4;
        const dim3 grid(p.H, p.B, p.n_split);
        if (dtype == UW_F32) hipLaunchKernelGGL(cross_decode_split_kernel<float>, grid, dim3(256), lds, s, p);
        else hipLaunchKernelGGL(cross_decode_split_kernel<bf16_t>, grid, dim3(256), lds, s, p);
        UW_LAUNCH_CHECK();
        return UW_OK;
    }
    const dim3 grid(p.H, p.B);
    const int wgs = p.H * p.B;
#define UW_CROSS_LAUNCH(NW)                                                                                     \
    do {                                                                                                        \
        const size_t lds = (size_t)(((p.S + 63) & ~63) + 2 * NW + NW * 64) * 4;                                 \
        if (dtype == UW_F32) hipLaunchKernelGGL((cross_decode_kernel<float, NW>), grid, dim3(64 * NW), lds, s, p); \
        else hipLaunchKernelGGL((cross_decode_kernel<bf16_t, NW>), grid, dim3(64 * NW), lds, s, p);               \
    } while (0)
    if (wgs >= 512) UW_CROSS_LAUNCH(4);    // measured: 640 workgroups are faster with 4 waves, 256 with 16
    else UW_CROSS_LAUNCH(16);
#undef UW_CROSS_LAUNCH
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_self_attn_decode(int dtype, const SelfDecodeParams& p, hipStream_t s)
{
    UW_REQUIRE(p.qkv && p.kcache && p.vcache && p.out && p.pos && p.B > 0 && p.H > 0 && p.d == p.H * 64 && p.n_ctx <= 512 && p.ld_cache >= 64,
               "self_attn_decode: bad arguments");
    const dim3 grid(p.H, p.B);
#define UW_SELF_LAUNCH(T, NU) hipLaunchKernelGGL((self_decode_kernel<T, NU>), grid, dim3(256), 0, s, p)
    if (dtype == UW_F32) {
        UW_SELF_LAUNCH(float, 0);
    } else {
        // bf16: one round of NU loads per lane covers NU * 32 rows; max_rows bounds pos + 1 over the windows of the call
        const int rows = p.max_rows > 0 ? std::min(p.max_rows, p.n_ctx) : p.n_ctx;
        const int nu = (rows + 31) / 32;
        // a grid that fills the chip several times over is bandwidth-bound and runs faster with the short rounds
        // (measured at 128 windows x 12 heads: 18.4 us against 19.6 us per launch)
        if (p.H * p.B >= 512) UW_SELF_LAUNCH(bf16_t, 0);
        else if (nu <= 4) UW_SELF_LAUNCH(bf16_t, 4);
        else if (nu <= 6) UW_SELF_LAUNCH(bf16_t, 6);
        else if (nu <= 8) UW_SELF_LAUNCH(bf16_t, 8);
        else if (nu <= 10) UW_SELF_LAUNCH(bf16_t, 10);
        else if (nu <= 12) UW_SELF_LAUNCH(bf16_t, 12);
        else UW_SELF_LAUNCH(bf16_t, 14);
    }
#undef UW_SELF_LAUNCH
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_greedy_next(int dtype, const GreedyState& st, const float* logits, long ld, int B, int V, int eot, int sot, int ts_begin,
                   int ts_end, float* x, const float* tok_emb, const float* pos_emb, int d, hipStream_t s, const float* cand_val,
                   const int32_t* cand_idx, int n_cand)
{
    UW_REQUIRE((logits || cand_val) && x && tok_emb && pos_emb && B > 0 && d % 4 == 0 && d <= 4096, "greedy_next: bad arguments");
    UW_REQUIRE(!cand_val || (cand_idx && n_cand > 0), "greedy_next: candidate values without indices");
    if (dtype == UW_F32)
        hipLaunchKernelGGL(greedy_next_kernel<float>, dim3(B), dim3(1024), 0, s, st, logits, ld, V, eot, sot, ts_begin, ts_end, x, tok_emb, pos_emb, d,
                           cand_val, cand_idx, n_cand);
    else
        hipLaunchKernelGGL(greedy_next_kernel<bf16_t>, dim3(B), dim3(1024), 0, s, st, logits, ld, V, eot, sot, ts_begin, ts_end, x, tok_emb, pos_emb, d,
                           cand_val, cand_idx, n_cand);
    UW_LAUNCH_CHECK();
    return UW_OK;
}
