// Word-alignment numerics on the GPU.
//
//   cost matrix  (urro_whisper/align.py:758-811,818): mean over the selected heads' attention rows, width-7
//                median filter with scipy's 'reflect' edge rule (d c b a | a b c d | d c b a), softmax over
//                frames, negate, widen to f64.
//   DTW          (urro_whisper/align.py:818-853): dtw-python's `symmetric1` recurrence in f64 with its
//                tie order (diagonal, then left, then up; sums formed before the strict '<' scan), backtrack,
//                and the reference's boundary fix-up (n_tok + 1 entries, last one repeated and clamped).
//
// The DTW kernel is an anti-diagonal wavefront: one workgroup per window, one wave per band of 64 token
// rows, lane l of a band working on column t - l at step t.  A lane's "up"/"diagonal" inputs come from its
// neighbour lane by one wave shuffle per step; band w gets the last row of band w-1 through an LDS row plus
// a progress counter, so bands run skewed behind one another without workgroup barriers.  Directions are
// packed 2 bits per cell.  Every add and compare is the f64 operation the CPU performs (no contraction is
// possible: the recurrence has no multiply), so paths are bit-identical for equal cost matrices.
#include "uw_common.h"
#include "uw_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cswap(float& a, float& b)
{
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo;
    b = hi;
}
// median of 7: 13 compare-exchanges, element 3 ends up holding the 4th smallest
__device__ __forceinline__ float median7(float p0, float p1, float p2, float p3, float p4, float p5, float p6)
{
    cswap(p0, p5); cswap(p0, p3); cswap(p1, p6);
    cswap(p2, p4); cswap(p0, p1); cswap(p3, p5);
    cswap(p2, p6); cswap(p2, p3); cswap(p3, p6);
    cswap(p4, p5); cswap(p1, p4); cswap(p1, p3);
    cswap(p3, p4);
    return p3;
}

__global__ __launch_bounds__(256) void cost_matrix_kernel(const float* __restrict__ rows, long sB, long sSel, int n_sel,
                                                          const int32_t* __restrict__ n_tok, int n_rows_max, int S,
                                                          double* __restrict__ cost)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* w = (float*)smem;         // [S] head mean
    float* red = w + ((S + 3) & ~3); // [8] block reductions
    const int b = blockIdx.y, n = blockIdx.x, tid = threadIdx.x;
    if (n >= n_tok[b]) return;
    const float* src = rows + (long)b * sB + (long)n * S;
    const float inv_cnt = (float)n_sel;
    for (int j = tid; j < S; j += 256) {
        float s = src[j];
        for (int h = 1; h < n_sel; ++h) s += src[(long)h * sSel + j];
        w[j] = s / inv_cnt;  // torch: sum over the stacked heads, then true division by the count
    }
    __syncthreads();
    // median over frames, then softmax; each thread owns frames tid, tid+256, ...
    float v[8];
    int cnt = 0;
    float mx = -INFINITY;
    for (int j = tid; j < S; j += 256, ++cnt) {
        float t[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            int idx = j + k - 3;
            if (idx < 0) idx = -idx - 1;            // reflect about the left edge, edge sample repeated
            if (idx >= S) idx = 2 * S - 1 - idx;    // same on the right
            t[k] = w[idx];
        }
        const float m = (S > 7) ? median7(t[0], t[1], t[2], t[3], t[4], t[5], t[6]) : w[j];
        v[cnt] = m;
        mx = fmaxf(mx, m);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int k = 0; k < cnt; ++k) {
        v[k] = expf(v[k] - mx);
        sum += v[k];
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = 1.0f / sum;
    double* dst = cost + ((long)b * n_rows_max + n) * S;
    cnt = 0;
    for (int j = tid; j < S; j += 256, ++cnt) dst[j] = (double)(-(v[cnt] * inv));
}

// ------------------------------------------------------------------------------------------------
constexpr int DTW_MAX_BANDS = 8;    // up to 512 token rows
constexpr int DTW_PUBLISH = 16;     // a band publishes its progress every 16 columns

__global__ __launch_bounds__(64 * DTW_MAX_BANDS) void dtw_kernel(const double* __restrict__ cost_all,
                                                                  const int32_t* __restrict__ n_tok, int n_rows_max,
                                                                  int M, uint32_t* __restrict__ dir_all, int dir_words,
                                                                  int32_t* __restrict__ boundaries,
                                                                  int32_t* __restrict__ path_i, int32_t* __restrict__ path_j,
                                                                  int32_t* __restrict__ path_len)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: last row of each band (f64 [bands-1][M]) | progress counters | reversed path scratch
    const int b = blockIdx.x;
    const int N = n_tok[b];
    const int lane = threadIdx.x & 63;
    const int band = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_bands = (N + 63) >> 6;
    double* edge = (double*)smem;
    int* progress = (int*)(edge + (size_t)(DTW_MAX_BANDS - 1) * M);
    int* s_len = progress + DTW_MAX_BANDS;                   // [0] path length, [1] spin time-out flag
    uint16_t* rev = (uint16_t*)(progress + DTW_MAX_BANDS + 4);  // [2][n_rows_max + M]
    const double* cost = cost_all + (size_t)b * n_rows_max * M;
    uint32_t* dir = dir_all + (size_t)b * n_rows_max * dir_words;
    int32_t* bnd = boundaries + (size_t)b * (n_rows_max + 1);

    if (threadIdx.x < DTW_MAX_BANDS + 4) progress[threadIdx.x] = 0;
    __syncthreads();

    if (N > 0 && band < n_bands) {
        const int i = band * 64 + lane;
        const bool row_ok = i < N;
        const double* crow = cost + (size_t)min(i, N - 1) * M;
        const double* up_edge = edge + (size_t)(band > 0 ? band - 1 : 0) * M;
        double* my_edge = edge + (size_t)band * M;
        const bool writes_edge = (lane == 63) && (band + 1 < n_bands);
        double left = 0.0, up_prev = 0.0, mine_prev = 0.0;
        uint32_t dword = 0;
        int known = 0;  // columns of the band above known to be published
        const int steps = M + 63;
        // The cost value of a step is an 8-byte read from this lane's own row, 64 different cache lines per wave
        // instruction: far too slow to sit on the recurrence's dependency chain.  Each lane keeps the values of its next
        // PF steps in registers, loaded PF steps ahead (out-of-range steps read a valid cell and ignore it).
        constexpr int PF = 8;
        double cbuf[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) cbuf[u] = crow[min(max(u - lane, 0), M - 1)];
        // the step count is rounded up to a multiple of PF: in the extra steps every lane is past its last column
        for (int t0 = 0; t0 < steps; t0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 + u;
                const int j = t - lane;
                // the wave needs column min(t, M-1) of the band above for lane 0
                if (band > 0) {
                    const int need = min(t, M - 1) + 1;
                    if (known < need) {
                        int spins = 0;
                        do {
                            // progress counter and edge row both live in LDS, and a wave's LDS operations execute in
                            // order: a relaxed load (no vector-memory wait, which would drain the prefetched cost
                            // values) followed by the edge read is enough
                            known = __hip_atomic_load(&progress[band - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            asm volatile("" ::: "memory");
                            if (known >= need) break;
                            __builtin_amdgcn_s_sleep(1);
                        } while (++spins < (1 << 24));
                        if (known < need) s_len[1] = 1;  // never expected: the band above always advances
                    }
                }
                // D[i-1][j] : neighbour lane's value from the previous step, or the band edge for lane 0
                double up = __shfl_up(mine_prev, 1, 64);
                if (lane == 0 && band > 0 && j >= 0 && j < M) up = up_edge[j];
                const bool active = row_ok && j >= 0 && j < M;
                const double c = cbuf[u];
                cbuf[u] = crow[min(max(j + PF, 0), M - 1)];
                double mine = mine_prev;
                if (active) {
                    double best = INFINITY;
                    uint32_t arg = 0;
                    if (i == 0 && j == 0) {
                        best = c;
                    } else {
                        if (i > 0 && j > 0) { const double v = up_prev + c; if (v < best) { best = v; arg = 1; } }
                        if (j > 0)          { const double v = left + c;    if (v < best) { best = v; arg = 2; } }
                        if (i > 0)          { const double v = up + c;      if (v < best) { best = v; arg = 3; } }
                    }
                    mine = best;
                    left = best;
                    dword |= arg << ((j & 15) * 2);
                    if ((j & 15) == 15 || j == M - 1) {
                        dir[(size_t)i * dir_words + (j >> 4)] = dword;
                        dword = 0;
                    }
                    if (writes_edge) my_edge[j] = best;
                }
                up_prev = up;   // becomes D[i-1][j-1] for the next step
                mine_prev = mine;
                if (band + 1 < n_bands) {
                    // lane 63 finished column t-63; publish in batches
                    const int done = t - 63 + 1;
                    if (done > 0 && done <= M && ((done % DTW_PUBLISH) == 0 || done == M)) {
                        asm volatile("" ::: "memory");   // the edge writes above are issued before the counter (same wave, LDS order)
                        if (lane == 63)
                            __hip_atomic_store(&progress[band], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
    }
    __threadfence();
    __syncthreads();

    // backtrack (align.py:830-853) by one lane; reversed path parked in LDS
    const int cap = n_rows_max + M;
    if (threadIdx.x == 0) {
        int len = 0;
        if (N > 0) {
            int i = N - 1, j = M - 1;
            uint32_t word = dir[(size_t)i * dir_words + (j >> 4)];
            bnd[0] = 0;
            rev[len] = (uint16_t)i; rev[cap + len] = (uint16_t)j; ++len;
            while (!(i == 0 && j == 0)) {
                const uint32_t s = (word >> ((j & 15) * 2)) & 3u;
                const int oi = i, oj = j;
                if (s == 1) { --i; --j; } else if (s == 2) { --j; } else if (s == 3) { --i; } else { len = -1; break; }
                if (i != oi) bnd[oi] = oj;  // (oi, oj) is the first path cell of row oi
                if (i != oi || (j >> 4) != (oj >> 4)) word = dir[(size_t)i * dir_words + (j >> 4)];
                rev[len] = (uint16_t)i; rev[cap + len] = (uint16_t)j; ++len;
            }
            if (len > 0) {
                // the un-fixed boundary list has N entries; the reference pads it to N + 1 with its last value
                // and clamps that last entry to the frame count
                const int last = (N > 1) ? bnd[N - 1] : 0;
                bnd[N] = min(last, M);
            }
        }
        if (s_len[1]) len = -2;
        // no warping path (a NaN / Inf cost leaves cells without a predecessor) or a stalled band: the window gets no
        // boundaries at all, as the reference returns no timestamps when dtw raises (align.py:823-827)
        if (len < 0) bnd[0] = -1;
        s_len[0] = len;
        if (path_len) path_len[b] = len;
    }
    __syncthreads();
    if (path_i && path_j && s_len[0] > 0) {
        const int len = s_len[0];
        int32_t* pi = path_i + (size_t)b * cap;
        int32_t* pj = path_j + (size_t)b * cap;
        for (int k = threadIdx.x; k < len; k += blockDim.x) {
            pi[k] = rev[len - 1 - k];
            pj[k] = rev[cap + len - 1 - k];
        }
    }
}

}  // namespace

int uw_cost_matrix_run(const float* rows, long sB, long sSel, int n_sel, const int32_t* n_tok, int B, int n_rows_max,
                       int S, double* cost, hipStream_t s)
{
    UW_REQUIRE(rows && n_tok && cost, "cost_matrix: null pointer");
    UW_REQUIRE(B > 0 && n_sel > 0 && n_rows_max > 0 && S > 0 && S <= 2048, "cost_matrix: bad sizes B=%d n_sel=%d rows=%d S=%d",
               B, n_sel, n_rows_max, S);
    const size_t lds = (size_t)((S + 3) & ~3) * 4 + 64;
    hipLaunchKernelGGL(cost_matrix_kernel, dim3(n_rows_max, B), dim3(256), lds, s, rows, sB, sSel, n_sel, n_tok,
                       n_rows_max, S, cost);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

size_t uw_dtw_dir_words(int S) { return (size_t)(S + 15) / 16; }

int uw_dtw_run(const double* cost, const int32_t* n_tok, int B, int n_rows_max, int S, uint8_t* dir_ws,
               int32_t* boundaries, int32_t* path_i, int32_t* path_j, int32_t* path_len, hipStream_t s)
{
    UW_REQUIRE(cost && n_tok && dir_ws && boundaries, "dtw: null pointer");
    UW_REQUIRE(B > 0 && n_rows_max > 0 && n_rows_max <= 64 * DTW_MAX_BANDS && S > 0 && S < 65536,
               "dtw: bad sizes B=%d rows=%d S=%d (rows <= %d)", B, n_rows_max, S, 64 * DTW_MAX_BANDS);
    const int dir_words = (int)uw_dtw_dir_words(S);
    const size_t lds = (size_t)(DTW_MAX_BANDS - 1) * S * 8 + (DTW_MAX_BANDS + 4) * 4 + (size_t)2 * (n_rows_max + S) * 2 + 16;
    UW_REQUIRE(lds <= 160 * 1024, "dtw: S=%d needs %zu bytes of LDS", S, lds);
    static PerDeviceFlag attr_set;
    if (attr_set.pending()) {
        UW_HIP_CHECK(hipFuncSetAttribute((const void*)dtw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.mark();
    }
    hipLaunchKernelGGL(dtw_kernel, dim3(B), dim3(64 * DTW_MAX_BANDS), lds, s, cost, n_tok, n_rows_max, S,
                       (uint32_t*)dir_ws, dir_words, boundaries, path_i, path_j, path_len);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

// ------------------------------------------------------------------------------------------------
// DTW without a caller-provided workspace and without any allocation: the direction matrix lives in a fixed
// device array.  Each GPU holds its own copy of the array at its own address; the address is looked up once per
// device (with that device current) and kept.  Calls on one GPU share the array, so calls that overlap on one GPU race.
constexpr size_t DTW_STATIC_BYTES = (size_t)16 << 20;
static __device__ __align__(256) unsigned char scratch[DTW_STATIC_BYTES];

int uw_dtw_static_run(const double* cost, const int32_t* n_tok, int B, int n_rows_max, int S, int32_t* boundaries,
                      int32_t* path_i, int32_t* path_j, int32_t* path_len, hipStream_t s)
{
    static std::atomic<void*> dev_ptr[64] = {};   // contexts on several host threads may resolve it twice: same value
    UW_REQUIRE(B > 0 && n_rows_max > 0 && S > 0, "dtw: bad sizes B=%d rows=%d S=%d", B, n_rows_max, S);
    const size_t need = (size_t)B * n_rows_max * uw_dtw_dir_words(S) * 4;
    UW_REQUIRE(need <= DTW_STATIC_BYTES,
               "uw_dtw_static: B=%d x rows=%d x S=%d needs %zu bytes of direction matrix, the fixed device array holds %zu; "
               "use uw_dtw with a workspace",
               B, n_rows_max, S, need, DTW_STATIC_BYTES);
    int dev = 0;
    UW_HIP_CHECK(hipGetDevice(&dev));
    UW_REQUIRE(dev >= 0 && dev < 64, "uw_dtw_static: device ordinal %d out of range", dev);
    // the array used is the CURRENT device's copy: the kernel must run there too, i.e. the caller's buffers (and the
    // stream they are used on) must belong to the current device
    hipPointerAttribute_t at;
    if (cost && hipPointerGetAttributes(&at, cost) == hipSuccess) {
        UW_REQUIRE(at.device == dev, "uw_dtw_static: the cost matrix lives on device %d but device %d is current; make the "
                                     "buffers' device current before the call", at.device, dev);
    } else {
        (void)hipGetLastError();   // not a pointer the runtime knows: let the kernel launch report it
    }
    void* ptr = dev_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {
        UW_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        UW_REQUIRE(ptr, "uw_dtw_static: no device address for the scratch array");
        dev_ptr[dev].store(ptr, std::memory_order_release);
    }
    // every direction word of rows < n_tok[b] is written by the forward sweep before the backtrack reads it, so what
    // an earlier call left in the array does not need clearing
    return uw_dtw_run(cost, n_tok, B, n_rows_max, S, (uint8_t*)ptr, boundaries, path_i, path_j, path_len, s);
}
