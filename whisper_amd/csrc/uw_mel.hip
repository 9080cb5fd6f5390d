// Log-mel front end (urro_whisper/audio/load.py:67-121 -> whisper.log_mel_spectrogram(audio, n_mels,
// padding=480000)[:, :3000]).
//
// Per frame: 400 samples (hop 160, centred, left edge reflected, right side zero padded), periodic Hann,
// 400-point real DFT -> 201 power bins, sparse triangular mel filters, log10(max(., 1e-10)).  Then the whole
// window is floored at (global max over EVERY frame the reference computes) - 8 and mapped by (x + 4) / 4;
// frames past 3000 are never stored but still feed that maximum, as in the reference.
//
// The 400-point real transform is a 200-point complex Stockham FFT (radix 8, 5, 5) in LDS on the packed
// even/odd samples plus an untangling pass; 40 lanes per frame, 8 frames per pass, 32 frames per workgroup
// so the [n_mels][32] output tile leaves as 128-byte row segments.  Twiddles, window and filters are f32
// tables rounded from f64 on the host.
#include "uw_common.h"
#include "uw_kernels.h"
#include <math.h>
#include <vector>

struct MelTables {
    int n_mels = 0;
    float* hann = nullptr;        // [400]
    float2* tw200 = nullptr;      // [200]  e^{-2 pi i k / 200}
    float2* tw400 = nullptr;      // [201]  e^{-2 pi i k / 400}
    int* band_start = nullptr;    // [n_mels] first bin with non-zero weight
    int* band_count = nullptr;    // [n_mels]
    int* band_off = nullptr;      // [n_mels] offset into weights
    float* weights = nullptr;     // concatenated non-zero runs
};

namespace {

constexpr int NFFT = 400, NH = 200, NBIN = 201, HOP = 160, NFRAMES = 3000, PAD_SAMPLES = 480000;
constexpr int FPP = 8;    // frames per pass
constexpr int FPW = 32;   // frames per workgroup
constexpr int MEL_THREADS = 320;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward transform rotation by -90 degrees)
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

__device__ __forceinline__ void dft8(float2* v)
{
    const float r = 0.70710678118654752440f;
    // three radix-2 passes, decimation in time on bit-reversed input order handled by index shuffling below
    float2 a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]);
    float2 a2 = cadd(v[2], v[6]), a3 = mul_mi(csub(v[2], v[6]));
    float2 a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    float2 a6 = cadd(v[3], v[7]), a7 = mul_mi(csub(v[3], v[7]));
    float2 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = csub(a1, a3);
    float2 b4 = cadd(a4, a6), b6 = mul_mi(csub(a4, a6)), b5 = cadd(a5, a7), b7 = csub(a5, a7);
    // twiddles W8^1 = (1 - i) r, W8^3 = (-1 - i) r
    const float2 t5 = make_float2((b5.x + b5.y) * r, (b5.y - b5.x) * r);
    const float2 t7 = make_float2((-b7.x + b7.y) * r, (-b7.y - b7.x) * r);
    v[0] = cadd(b0, b4); v[4] = csub(b0, b4);
    v[1] = cadd(b1, t5); v[5] = csub(b1, t5);
    v[2] = cadd(b2, b6); v[6] = csub(b2, b6);
    v[3] = cadd(b3, t7); v[7] = csub(b3, t7);
}

__device__ __forceinline__ void dft5(float2* v)
{
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;  // cos(2pi/5), cos(4pi/5)
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;   // sin(2pi/5), sin(4pi/5)
    const float2 p14 = cadd(v[1], v[4]), m14 = csub(v[1], v[4]);
    const float2 p23 = cadd(v[2], v[3]), m23 = csub(v[2], v[3]);
    const float2 x0 = v[0];
    const float2 r1 = make_float2(x0.x + c1 * p14.x + c2 * p23.x, x0.y + c1 * p14.y + c2 * p23.y);
    const float2 r2 = make_float2(x0.x + c2 * p14.x + c1 * p23.x, x0.y + c2 * p14.y + c1 * p23.y);
    // forward transform: X1 = r1 - i (s1 m14 + s2 m23), X4 = r1 + i (...), X2 = r2 - i (s2 m14 - s1 m23), X3 = conj-pair
    const float2 q1 = make_float2(s1 * m14.x + s2 * m23.x, s1 * m14.y + s2 * m23.y);
    const float2 q2 = make_float2(s2 * m14.x - s1 * m23.x, s2 * m14.y - s1 * m23.y);
    v[0] = make_float2(x0.x + p14.x + p23.x, x0.y + p14.y + p23.y);
    v[1] = make_float2(r1.x + q1.y, r1.y - q1.x);
    v[4] = make_float2(r1.x - q1.y, r1.y + q1.x);
    v[2] = make_float2(r2.x + q2.y, r2.y - q2.x);
    v[3] = make_float2(r2.x - q2.y, r2.y + q2.x);
}

__device__ __forceinline__ int float_key(float f)
{
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key_float(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

__global__ void mel_init_max_kernel(int* __restrict__ keys, int B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) keys[i] = float_key(-10.0f);  // log10 of the clamp floor: every window has all-zero padding frames
}

__global__ __launch_bounds__(MEL_THREADS) void mel_frames_kernel(MelTables tb, const float* __restrict__ audio,
                                                                 long stride, const int32_t* __restrict__ n_samples,
                                                                 float* __restrict__ mel, int* __restrict__ max_keys)
{
    __shared__ float2 bufA[FPP][NH];
    __shared__ float2 bufB[FPP][NH];
    __shared__ float power[FPP][NBIN + 3];
    __shared__ float tile[128][FPW + 1];
    __shared__ float2 s_tw200[NH];
    __shared__ float s_hann[NFFT];

    const int b = blockIdx.y, t_base = blockIdx.x * FPW, tid = threadIdx.x;
    const int n = n_samples[b];
    const int total_frames = (n + PAD_SAMPLES) / HOP;                 // frames the reference computes
    const int live = min(total_frames, (n + NH + HOP - 1) / HOP);     // frames that can see audio
    if (t_base >= live && t_base >= NFRAMES) return;
    const float* x = audio + (long)b * stride;
    for (int i = tid; i < NH; i += MEL_THREADS) s_tw200[i] = tb.tw200[i];
    for (int i = tid; i < NFFT; i += MEL_THREADS) s_hann[i] = tb.hann[i];
    const int n_mels = tb.n_mels;
    float wg_max = -INFINITY;

    for (int pass = 0; pass < FPW / FPP; ++pass) {
        const int t0 = t_base + pass * FPP;
        __syncthreads();
        // windowed samples, packed even/odd into complex
        for (int i = tid; i < FPP * NH; i += MEL_THREADS) {
            const int f = i / NH, m = i % NH, t = t0 + f;
            float2 z = make_float2(0.f, 0.f);
            if (t < live) {
                int k0 = t * HOP + 2 * m - NH, k1 = k0 + 1;
                if (k0 < 0) k0 = -k0;
                if (k1 < 0) k1 = -k1;
                const float a0 = (k0 < n) ? x[k0] : 0.f, a1 = (k1 < n) ? x[k1] : 0.f;
                z = make_float2(a0 * s_hann[2 * m], a1 * s_hann[2 * m + 1]);
            }
            bufA[f][m] = z;
        }
        __syncthreads();
        const int f = tid / 40, j = tid % 40;
        // stage 1: radix 8, Ns = 1 (25 butterflies)
        if (j < 25) {
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = bufA[f][j + r * 25];
            dft8(v);
#pragma unroll
            for (int r = 0; r < 8; ++r) bufB[f][j * 8 + r] = v[r];
        }
        __syncthreads();
        // stage 2: radix 5, Ns = 8 (40 butterflies)
        {
            const int k = j & 7;
            float2 v[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                float2 a = bufB[f][j + r * 40];
                if (r) a = cmul(a, s_tw200[r * k * 5]);
                v[r] = a;
            }
            dft5(v);
            const int j0 = (j >> 3) * 40 + k;
#pragma unroll
            for (int r = 0; r < 5; ++r) bufA[f][j0 + r * 8] = v[r];
        }
        __syncthreads();
        // stage 3: radix 5, Ns = 40
        {
            const int k = j;  // j % 40
            float2 v[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                float2 a = bufA[f][j + r * 40];
                if (r) a = cmul(a, s_tw200[r * k]);
                v[r] = a;
            }
            dft5(v);
#pragma unroll
            for (int r = 0; r < 5; ++r) bufB[f][k + r * 40] = v[r];
        }
        __syncthreads();
        // untangle the packed transform into the 201 one-sided bins and take |X|^2
        for (int i = tid; i < FPP * NBIN; i += MEL_THREADS) {
            const int ff = i / NBIN, k = i % NBIN;
            const float2 zk = bufB[ff][k == NH ? 0 : k];
            const float2 zc = bufB[ff][k == 0 ? 0 : NH - k];
            const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
            const float2 o = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
            const float2 w = tb.tw400[k];
            const float2 xk = cadd(e, cmul(w, o));
            power[ff][k] = xk.x * xk.x + xk.y * xk.y;
        }
        __syncthreads();
        for (int i = tid; i < FPP * n_mels; i += MEL_THREADS) {
            const int ff = i / n_mels, m = i % n_mels, t = t0 + ff;
            float v = -10.0f;
            if (t < live) {
                const float* w = tb.weights + tb.band_off[m];
                const float* p = &power[ff][tb.band_start[m]];
                float acc = 0.f;
                const int cnt = tb.band_count[m];
                for (int q = 0; q < cnt; ++q) acc += w[q] * p[q];
                v = log10f(fmaxf(acc, 1e-10f));
                wg_max = fmaxf(wg_max, v);
            }
            tile[m][pass * FPP + ff] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < n_mels * FPW; i += MEL_THREADS) {
        const int m = i / FPW, c = i % FPW, t = t_base + c;
        if (t < NFRAMES) mel[((long)b * n_mels + m) * NFRAMES + t] = tile[m][c];
    }
    wg_max = wave_max(wg_max);
    if ((tid & 63) == 0 && wg_max > -INFINITY) atomicMax(&max_keys[b], float_key(wg_max));
}

__global__ void mel_finish_kernel(float* __restrict__ mel, const int* __restrict__ max_keys, long per_window, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float floor_v = key_float(max_keys[i / per_window]) - 8.0f;
        mel[i] = (fmaxf(mel[i], floor_v) + 4.0f) / 4.0f;
    }
}

template <typename T> int upload(T** dst, const std::vector<T>& src)
{
    UW_HIP_CHECK(hipMalloc((void**)dst, src.size() * sizeof(T)));
    UW_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return UW_OK;
}

}  // namespace

int uw_mel_tables_create(int n_mels, const float* filters_host, MelTables** out)
{
    UW_REQUIRE(n_mels > 0 && n_mels <= 128 && filters_host && out, "mel tables: n_mels=%d unsupported", n_mels);
    MelTables* t = new MelTables();
    t->n_mels = n_mels;
    const double PI = 3.14159265358979323846;
    std::vector<float> hann(NFFT);
    for (int i = 0; i < NFFT; ++i) hann[i] = (float)(0.5 - 0.5 * cos(2.0 * PI * i / NFFT));
    std::vector<float2> tw200(NH), tw400(NBIN);
    for (int k = 0; k < NH; ++k) tw200[k] = make_float2((float)cos(2.0 * PI * k / NH), (float)-sin(2.0 * PI * k / NH));
    for (int k = 0; k < NBIN; ++k) tw400[k] = make_float2((float)cos(2.0 * PI * k / NFFT), (float)-sin(2.0 * PI * k / NFFT));
    std::vector<int> start(n_mels), count(n_mels), off(n_mels);
    std::vector<float> w;
    for (int m = 0; m < n_mels; ++m) {
        int lo = NBIN, hi = -1;
        for (int k = 0; k < NBIN; ++k)
            if (filters_host[m * NBIN + k] != 0.f) { lo = k < lo ? k : lo; hi = k; }
        start[m] = hi < 0 ? 0 : lo;
        count[m] = hi < 0 ? 0 : hi - lo + 1;
        off[m] = (int)w.size();
        for (int k = 0; k < count[m]; ++k) w.push_back(filters_host[m * NBIN + start[m] + k]);
    }
    if (w.empty()) w.push_back(0.f);
    int rc = upload(&t->hann, hann);
    if (!rc) rc = upload(&t->tw200, tw200);
    if (!rc) rc = upload(&t->tw400, tw400);
    if (!rc) rc = upload(&t->band_start, start);
    if (!rc) rc = upload(&t->band_count, count);
    if (!rc) rc = upload(&t->band_off, off);
    if (!rc) rc = upload(&t->weights, w);
    if (rc) { uw_mel_tables_destroy(t); return rc; }
    *out = t;
    return UW_OK;
}

void uw_mel_tables_destroy(MelTables* t)
{
    if (!t) return;
    (void)hipFree(t->hann); (void)hipFree(t->tw200); (void)hipFree(t->tw400);
    (void)hipFree(t->band_start); (void)hipFree(t->band_count); (void)hipFree(t->band_off); (void)hipFree(t->weights);
    delete t;
}

int uw_mel_run(const MelTables* t, const float* audio, long stride, const int32_t* n_samples, int B, float* mel,
               float* frame_max_ws, hipStream_t s)
{
    UW_REQUIRE(t && audio && n_samples && mel && frame_max_ws, "mel: null pointer");
    UW_REQUIRE(B > 0 && stride > 0 && stride < (1l << 30), "mel: bad sizes B=%d stride=%ld", B, stride);
    int* keys = (int*)frame_max_ws;
    hipLaunchKernelGGL(mel_init_max_kernel, dim3((B + 255) / 256), dim3(256), 0, s, keys, B);
    UW_LAUNCH_CHECK();
    long frames = (stride + NH + HOP - 1) / HOP;   // upper bound on frames that can see audio
    if (frames < NFRAMES) frames = NFRAMES;
    const dim3 grid((unsigned)((frames + FPW - 1) / FPW), B);
    hipLaunchKernelGGL(mel_frames_kernel, grid, dim3(MEL_THREADS), 0, s, *t, audio, stride, n_samples, mel, keys);
    UW_LAUNCH_CHECK();
    const long per = (long)t->n_mels * NFRAMES, n = per * B;
    hipLaunchKernelGGL(mel_finish_kernel, dim3(2048), dim3(256), 0, s, mel, keys, per, n);
    UW_LAUNCH_CHECK();
    return UW_OK;
}
