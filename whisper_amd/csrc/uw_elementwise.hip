// Row-wise and elementwise kernels around the GEMMs: LayerNorm, dtype conversion / weight packing, the
// mel relayout feeding the conv stem, position adds, token embedding, and the greedy argmax.
// All are HBM-bound streaming kernels: 16-byte accesses, one wave per row where a row reduction is needed.
#include "uw_common.h"
#include "uw_kernels.h"

namespace {

// one wave per row; NI = float4 groups per lane (d <= 256 * NI), d % 4 == 0
template <typename T, int NI>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, T* __restrict__ y,
                                                        float* __restrict__ y32, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, long rows, int d, int packed, int x_layout,
                                                        int split)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const f32x4* xr = (const f32x4*)(x + row * d);
    const int nv = d >> 2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[NI], g[NI], bb[NI];
    // scale and shift are requested together with the row: one memory round trip, not two
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + i * 64;
        const bool in = c < nv;
        v[i] = in ? (x_layout ? *(const f32x4*)(x + x_elem_offset<T>(row, c * 4, d)) : xr[c]) : zero;
        g[i] = in ? ((const f32x4*)gamma)[c] : zero;
        bb[i] = in ? ((const f32x4*)beta)[c] : zero;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (lane + i * 64 < nv) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = v[i][k] - mean;
                q += t * t;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (v[i][k] - mean) * rstd * g[i][k] + bb[i][k];
            if (y32) ((f32x4*)(y32 + row * d))[c] = o;
            if (y) {
                // four consecutive columns never straddle a 16-byte chunk, so they stay contiguous when packed
                constexpr int KT = 64 / (int)sizeof(T);
                T* dst = packed == 2 ? y + ((long)((c * 4) / KT) * rows + row) * KT + (c * 4) % KT
                       : packed      ? y + frag_elem_offset<T>(row, c * 4, d)
                                     : y + row * (split ? 2 * d : d) + c * 4;
                if constexpr (sizeof(T) == 4) {
                    *(f32x4*)dst = o;
                } else {
                    union { bf16_t h[4]; uint2 u; } pk, lo;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        pk.h[k] = (bf16_t)o[k];
                        lo.h[k] = (bf16_t)(o[k] - (float)pk.h[k]);
                    }
                    *(uint2*)dst = pk.u;
                    if (split) *(uint2*)(dst + d) = lo.u;
                }
            }
        }
    }
}

// one wave per weight row: the row in the compute dtype, the LayerNorm shift folded into the bias, and the
// gamma-weighted sum of the rounded row for the mean correction
template <typename T>
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ w, const float* __restrict__ bias_in,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float scale, T* __restrict__ dst, float* __restrict__ bias_out,
                                                             float* __restrict__ wsum, int N, int K)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float sb = 0.f, sw = 0.f;
    for (int k = lane; k < K; k += 64) {
        const T r = from_f32<T>(w[(long)n * K + k] * scale);
        dst[(long)n * K + k] = r;
        sw += to_f32<T>(r) * gamma[k];
        sb += to_f32<T>(r) * beta[k];
    }
    sb = wave_sum(sb);
    sw = wave_sum(sw);
    if (lane == 0) {
        bias_out[n] = scale * (bias_in ? bias_in[n] : 0.f) + sb;
        wsum[n] = sw;
    }
}

template <typename T>
__global__ void convert_kernel(const float* __restrict__ src, T* __restrict__ dst, long n, float scale)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = from_f32<T>(src[i] * scale);
}

template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int co, int ci, int kpad)
{
    const long n = (long)co * kpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i / kpad), k = (int)(i % kpad);
        float v = 0.f;
        if (k < 3 * ci) {
            const int tap = k / ci, c = k % ci;
            v = w[((long)o * ci + c) * 3 + tap];
        }
        dst[i] = from_f32<T>(v);
    }
}

// mel [B][n_mels][F] -> dst[b][(t+1)][m]; rows 0 and F+1 (and the slack after) stay zero from allocation
template <typename T>
__global__ __launch_bounds__(256) void mel_time_major_kernel(const float* __restrict__ mel, T* __restrict__ dst,
                                                             int n_mels, int F, long dst_stride)
{
    __shared__ float tile[64][65];
    const int b = blockIdx.z, t0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
    const float* src = mel + (long)b * n_mels * F;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int m = i >> 6, t = i & 63;
        tile[m][t] = (m0 + m < n_mels && t0 + t < F) ? src[(long)(m0 + m) * F + t0 + t] : 0.f;
    }
    __syncthreads();
    T* out = dst + (long)b * dst_stride;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int t = i >> 6, m = i & 63;
        if (m0 + m < n_mels && t0 + t < F) out[(long)(t0 + t + 1) * n_mels + m0 + m] = from_f32<T>(tile[m][t]);
    }
}

__global__ void add_positions_kernel(float* __restrict__ x, const float* __restrict__ pos, long n4, long per_window4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 a = ((f32x4*)x)[i];
        const f32x4 p = ((const f32x4*)pos)[i % per_window4];
        a += p;
        ((f32x4*)x)[i] = a;
    }
}

__global__ __launch_bounds__(256) void embed_tokens_kernel(float* __restrict__ x, const float* __restrict__ tok_emb,
                                                           const float* __restrict__ pos_emb,
                                                           const int32_t* __restrict__ ids, int ld_ids,
                                                           const int32_t* __restrict__ pos0, int T, int d, int vocab,
                                                           int n_ctx)
{
    const int b = blockIdx.y, j = blockIdx.x;
    int id = ids[(long)b * ld_ids + j];
    id = min(max(id, 0), vocab - 1);
    const int pos = min((pos0 ? pos0[b] : 0) + j, n_ctx - 1);
    const f32x4* e = (const f32x4*)(tok_emb + (long)id * d);
    const f32x4* p = (const f32x4*)(pos_emb + (long)pos * d);
    f32x4* o = (f32x4*)(x + ((long)b * T + j) * d);
    for (int c = threadIdx.x; c < (d >> 2); c += 256) o[c] = e[c] + p[c];
}

// first index of the maximum (numpy argmax tie rule); one 1024-thread workgroup per row
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const float* __restrict__ logits, long ld, int V,
                                                           int32_t* __restrict__ out)
{
    __shared__ float sv[16];
    __shared__ int si[16];
    const int arg = block_argmax_1024(logits + (long)blockIdx.x * ld, V, sv, si);
    if (threadIdx.x == 0) out[blockIdx.x] = arg;
}

inline int stream_grid(long n, int block) { return (int)(n / block + 1 < 4096 ? n / block + 1 : 4096); }

}  // namespace

int uw_layernorm(int dtype, const float* x, void* y, float* y32, const float* gamma, const float* beta, long rows,
                 int d, hipStream_t s, int packed, int x_layout, int split)
{
    UW_REQUIRE(!split || (dtype == UW_BF16 && !packed), "layernorm: split output is a row-major bf16 form");
    UW_REQUIRE(d % 4 == 0 && d <= 2048 && rows > 0, "layernorm: d=%d rows=%ld unsupported", d, rows);
    UW_REQUIRE(!packed || d % (dtype == UW_F32 ? 16 : 32) == 0, "layernorm: packed output needs d=%d to be a multiple of 64 bytes", d);
    const dim3 grid((unsigned)((rows + 3) / 4));
    const int ni = (d / 4 + 63) / 64;
#define UW_LN_LAUNCH(NI)                                                                                                         \
    do {                                                                                                                         \
        if (dtype == UW_F32) hipLaunchKernelGGL((layernorm_kernel<float, NI>), grid, dim3(256), 0, s, x, (float*)y, y32, gamma, beta, rows, d, packed, x_layout, split); \
        else hipLaunchKernelGGL((layernorm_kernel<bf16_t, NI>), grid, dim3(256), 0, s, x, (bf16_t*)y, y32, gamma, beta, rows, d, packed, x_layout, split);    \
    } while (0)
    if (ni <= 1) UW_LN_LAUNCH(1);
    else if (ni <= 2) UW_LN_LAUNCH(2);
    else if (ni <= 3) UW_LN_LAUNCH(3);
    else if (ni <= 4) UW_LN_LAUNCH(4);
    else if (ni <= 6) UW_LN_LAUNCH(6);
    else UW_LN_LAUNCH(8);
#undef UW_LN_LAUNCH
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_fold_layernorm(int dtype, const float* w, const float* bias_in, const float* gamma, const float* beta, float scale,
                      void* dst, float* bias_out, float* wsum, int N, int K, hipStream_t s)
{
    UW_REQUIRE(w && gamma && beta && dst && bias_out && wsum && N > 0 && K > 0, "fold_layernorm: missing argument");
    const dim3 grid((N + 3) / 4);
    if (dtype == UW_F32) hipLaunchKernelGGL(fold_layernorm_kernel<float>, grid, dim3(256), 0, s, w, bias_in, gamma, beta, scale, (float*)dst, bias_out, wsum, N, K);
    else hipLaunchKernelGGL(fold_layernorm_kernel<bf16_t>, grid, dim3(256), 0, s, w, bias_in, gamma, beta, scale, (bf16_t*)dst, bias_out, wsum, N, K);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_convert(int dtype, const float* src, void* dst, long n, float scale, hipStream_t s)
{
    if (n <= 0) return UW_OK;
    if (dtype == UW_F32) hipLaunchKernelGGL(convert_kernel<float>, dim3(stream_grid(n, 256)), dim3(256), 0, s, src, (float*)dst, n, scale);
    else hipLaunchKernelGGL(convert_kernel<bf16_t>, dim3(stream_grid(n, 256)), dim3(256), 0, s, src, (bf16_t*)dst, n, scale);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_pack_conv_weight(int dtype, const float* w, void* dst, int co, int ci, int kpad, hipStream_t s)
{
    UW_REQUIRE(kpad >= 3 * ci, "pack_conv_weight: kpad=%d < 3*ci=%d", kpad, 3 * ci);
    const long n = (long)co * kpad;
    if (dtype == UW_F32) hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(stream_grid(n, 256)), dim3(256), 0, s, w, (float*)dst, co, ci, kpad);
    else hipLaunchKernelGGL(pack_conv_weight_kernel<bf16_t>, dim3(stream_grid(n, 256)), dim3(256), 0, s, w, (bf16_t*)dst, co, ci, kpad);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_mel_to_time_major(int dtype, const float* mel, void* dst, int B, int n_mels, int n_frames, long dst_stride,
                         hipStream_t s)
{
    const dim3 grid((n_frames + 63) / 64, (n_mels + 63) / 64, B);
    if (dtype == UW_F32) hipLaunchKernelGGL(mel_time_major_kernel<float>, grid, dim3(256), 0, s, mel, (float*)dst, n_mels, n_frames, dst_stride);
    else hipLaunchKernelGGL(mel_time_major_kernel<bf16_t>, grid, dim3(256), 0, s, mel, (bf16_t*)dst, n_mels, n_frames, dst_stride);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_add_positions(float* x, const float* pos, int B, int T, int d, hipStream_t s)
{
    UW_REQUIRE(d % 4 == 0, "add_positions: d=%d", d);
    const long per = (long)T * d / 4, n4 = per * B;
    hipLaunchKernelGGL(add_positions_kernel, dim3(stream_grid(n4, 256)), dim3(256), 0, s, x, pos, n4, per);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_embed_tokens(float* x, const float* tok_emb, const float* pos_emb, const int32_t* ids, int ld_ids,
                    const int32_t* pos0, int B, int T, int d, int vocab, int n_ctx, hipStream_t s)
{
    UW_REQUIRE(d % 4 == 0 && B > 0 && T > 0, "embed_tokens: bad sizes");
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(T, B), dim3(256), 0, s, x, tok_emb, pos_emb, ids, ld_ids, pos0, T, d,
                       vocab, n_ctx);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

int uw_argmax_rows(const float* logits, long ld, int B, int V, int32_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(B), dim3(1024), 0, s, logits, ld, V, out);
    UW_LAUNCH_CHECK();
    return UW_OK;
}
