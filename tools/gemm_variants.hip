// Development probe: two variants of the encoder's tile GEMM that were built, measured and NOT adopted, kept here so that the
// measurements can be repeated.  Both reuse the product kernel's staging layout, tile order and epilogue pieces
// (this file includes whisper_amd/csrc/uw_gemm.hip, so the helpers of its unnamed namespace are visible).
//   quad: 256 x 128 tile, four waves of 128 x 64, operand fragments of the next K-step read into a second register set
//   oct : persistent 256 x 256 tile, eight waves of 128 x 64, operand stream running four K-steps ahead across tile
//         boundaries, epilogue through its own LDS, fragments refreshed between the MFMAs
//   hex : oct with FOUR waves of 128 x 128 (the vendor library's wave tile for these shapes), one wave per SIMD:
//         bit-identical results, 476 / 588 / 261 / 635 TFLOP/s (fc1 / qk / out+res / fc2) against duo's 749 / 755 / 516 / 844;
//         without operand requests 822 / 806 / 277 / 743 -- with one wave per SIMD nothing covers the epilogue and the
//         residual's read-modify-write
// Timing-only knobs (results are wrong with them): DMA_MODE bits -- 8 / 16 (quad): request one operand piece again and
// again instead of new ones; 32: no wait for the DMA pieces; 64: no barrier; 128: no operand requests; 512: no A-fragment
// reads (oct).  What they showed on MI355X (random bf16 operands, M = 192000): every loop structure lands at 0.72-0.96
// PFLOP/s; without operand requests 0.87-1.2; with nothing but MFMAs and five LDS reads per step 0.92-1.22 -- the chip's
// sustained rate for this MFMA stream on random data, far below the 2.5 PFLOP/s of the data sheet.  Waiting (vmcnt,
// barriers) is not what costs: leaving both out changes the time by 2 %.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iwhisper_amd/csrc tools/gemm_variants.hip -o tools/_build/gemm_variants
// Run:   KBLOCK=3 VARIANT=duo|quad|oct|hex [NOACT=1] [DMA_MODE=n] [CHECK=1] tools/_build/gemm_variants [M]
#define UW_TUNE
#include "../whisper_amd/csrc/uw_gemm.hip"
#include <stdarg.h>
#include <type_traits>
#include <vector>
void uw_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

namespace {

// s_waitcnt lgkmcnt(0) that the compiler sees: fragments read from LDS one loop iteration ahead are then known to be in
// their registers, so no wait is inserted in front of the instructions that use them
__device__ __forceinline__ void retire_lds_reads() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// six / four LDS-DMA pieces in one statement (M0 saved once): a wave's share of one ring stage
__device__ __forceinline__ void glds16_sv_4a2b(const void* abase, const unsigned (&aoff)[4], unsigned a_lds, const void* bbase,
                                               const unsigned (&boff)[2], unsigned b_lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %9\n\t"
                 "s_add_u32 m0, %7, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %9\n\t"
                 "s_add_u32 m0, %7, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %9\n\t"
                 "s_add_u32 m0, %7, 0x3000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %9\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %10\n\t"
                 "s_add_u32 m0, %8, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %10\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(aoff[0]), "v"(aoff[1]), "v"(aoff[2]), "v"(aoff[3]), "v"(boff[0]), "v"(boff[1]), "s"(a_lds), "s"(b_lds), "s"(abase),
                   "s"(bbase)
                 : "memory", "scc");
}
__device__ __forceinline__ void glds16_sv_2a2b(const void* abase, const unsigned (&aoff)[2], unsigned a_lds, const void* bbase,
                                               const unsigned (&boff)[2], unsigned b_lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %7\n\t"
                 "s_add_u32 m0, %5, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %8\n\t"
                 "s_add_u32 m0, %6, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %8\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(aoff[0]), "v"(aoff[1]), "v"(boff[0]), "v"(boff[1]), "s"(a_lds), "s"(b_lds), "s"(abase), "s"(bbase)
                 : "memory", "scc");
}

// ------------------------------------------------------------------------------------------------
// Same 256 x 128 tile, stage layout and epilogue as gemm_tn_duo_kernel, with FOUR waves of 128 x 64 instead of eight of
// 64 x 64 and the operand fragments of the NEXT K-step fetched from LDS into a second register set while the matrix
// cores work on the current one.  A 64 x 64 wave tile reads 8 KiB of LDS per 16 MFMAs, which with sixteen waves on a CU
// keeps the LDS busy for as many cycles as the matrix cores; the 128 x 64 tile reads 12 KiB per 32 MFMAs (a quarter
// less per flop), and because its reads run one K-step ahead no MFMA waits on an LDS round trip.  Two workgroups per CU
// (2 waves per SIMD, up to 256 registers each), so one workgroup's epilogue still overlaps the other's loop.
// Ring protocol (3 stages): at the top of step k the wave waits for its own DMA pieces of stage k+1 and for its fragment
// reads of stage k, then the barrier makes stage k+1 visible and guarantees that every wave is done reading stage k from
// LDS -- so stage k+3 may be requested into that slot -- then fragments k+1 are read while step k multiplies.
constexpr int Q_WAVES = 4, Q_TM = 8;

template <typename T, typename TC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn_quad_kernel(GemmParams p)
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

    unsigned offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int L = i * 256 + tid, row = L >> 2, c = (L & 3) ^ ((-(row >> 2)) & 3);
        offA[i] = (unsigned)(((long)(min(m0 + row, p.M - 1) - m0) * p.lda + c * E) * (long)sizeof(T));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = i * 256 + tid, row = L >> 2, c = (L & 3) ^ ((-(row >> 2)) & 3);
        offB[i] = (unsigned)(((long)(min(n0 + row, p.N - 1) - n0) * p.ldb + c * E) * (long)sizeof(T));
    }
    const T* baseA = A + (long)m0 * p.lda;
    const T* baseB = B + (long)n0 * p.ldb;
    const int wrap_a = p.kwrap_a ? p.kwrap_a / KT : 0x7fffffff, wrap_b = p.kwrap_b ? p.kwrap_b / KT : 0x7fffffff;
    const long step_a = p.kstep_a ? p.kstep_a : KT, step_b = p.kstep_b ? p.kstep_b : KT;
    const int nk = p.K / KT;
    // the stage to request next: its K index, ring slot and operand addresses advance together (a split operand's
    // partner starts over after the wrap length), so a request costs a handful of scalar instructions
    int kt_next = 0, slot_next = 0;
    const T* ka = baseA;
    const T* kb = baseB;
    auto stage_next = [&]() {
        const unsigned a_dst = lds_base + slot_next * STAGE3;
#ifdef UW_TUNE   // timing-only builds (tools/gemm_bench.hip): leave out part of the operand traffic to see what bounds the loop
        if (p.act & 8) { glds16_sv(kb, offB[0], a_dst + A3_BYTES); glds16_sv(kb, offB[1], a_dst + A3_BYTES + 4096);
                         glds16_sv(kb, offB[0], a_dst + A3_BYTES); glds16_sv(kb, offB[1], a_dst + A3_BYTES + 4096);
                         glds16_sv(kb, offB[0], a_dst + A3_BYTES); glds16_sv(kb, offB[1], a_dst + A3_BYTES + 4096); }
        else if (p.act & 16) { glds16_sv(ka, offA[0], a_dst); glds16_sv(ka, offA[0], a_dst); glds16_sv(ka, offA[0], a_dst);
                               glds16_sv(ka, offA[0], a_dst); glds16_sv(ka, offA[0], a_dst); glds16_sv(ka, offA[0], a_dst); }
        else
#endif
        glds16_sv_4a2b(ka, offA, a_dst, kb, offB, a_dst + A3_BYTES);
        ++kt_next;
        slot_next = slot_next == NSTAGE3 - 1 ? 0 : slot_next + 1;
        ka = kt_next == wrap_a ? baseA : ka + step_a;
        kb = kt_next == wrap_b ? baseB : kb + step_b;
    };
    stage_next();
    if (nk > 1) stage_next();
    if (nk > 2) stage_next();
    const float* __restrict__ R = p.residual ? p.residual + bz * p.sR : nullptr;
    const bool r_vec = (p.ldr & 3) == 0;
    const bool r_late = sizeof(TC) == 4 && R && r_vec && p.ldr == p.ldc && !p.c_group && !p.c_kblock && (p.ldc & 3) == 0;
    f32x4 acc[Q_TM][4];
    tile_acc_init<Q_TM>(p, acc, m0 + wr * 128, n0 + wc * 64, frow, fg, R, r_late, r_vec);
    retire_vector_loads();   // bias / residual are in the accumulators, and the first three stages have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // lane's fragment address inside a stage: rows t*16 + frow of the wave's block, chunk position by the row's swizzle
    const int pos = (fg ^ ((-(frow >> 2)) & 3)) << 4;
    const char* a_lane = smem + (wr * 128 + frow) * K3_ROW + pos;
    const char* b_lane = smem + A3_BYTES + (wc * 64 + frow) * K3_ROW + pos;
    uint4 af[2][Q_TM], bf[2][4];
    auto read_frags = [&](int st, uint4 (&a)[Q_TM], uint4 (&b)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = *(const uint4*)(b_lane + st * STAGE3 + t * 16 * K3_ROW);
#pragma unroll
        for (int t = 0; t < Q_TM; ++t) a[t] = *(const uint4*)(a_lane + st * STAGE3 + t * 16 * K3_ROW);
    };
    auto mma_step = [&](const uint4 (&a)[Q_TM], const uint4 (&b)[4]) {
#pragma unroll
        for (int tm = 0; tm < Q_TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) mma16<T>(b[tn], a[tm], acc[tm][tn]);
    };
    int slot_read = 1;           // ring slot of stage kt + 1, the one whose fragments are read next
    // one K-step that is not the last: `ca / cb` hold the fragments of step kt (read during the previous step),
    // `na / nb` receive those of kt + 1 while step kt multiplies
    auto k_step = [&](int kt, const uint4 (&ca)[Q_TM], const uint4 (&cb)[4], uint4 (&na)[Q_TM], uint4 (&nb)[4]) {
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // all but the six pieces of stage kt + 2
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        retire_lds_reads();                                                 // this wave's reads of stage kt are in registers
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_frags(slot_read, na, nb);
        slot_read = slot_read == NSTAGE3 - 1 ? 0 : slot_read + 1;
        if (kt_next < nk) stage_next();                                     // into the slot of stage kt: nobody reads it any more
        mma_step(ca, cb);
    };
    // nk - 1 steps fetch ahead, the last one only multiplies.  The fragments alternate between the two register sets; an
    // odd number of fetching steps starts in set 1, so that the last step always finds its fragments in set 0.
    int kt = 0;
    if ((nk - 1) & 1) {
        read_frags(0, af[1], bf[1]);
        k_step(0, af[1], bf[1], af[0], bf[0]);
        kt = 1;
    } else {
        read_frags(0, af[0], bf[0]);
    }
    for (; kt + 1 < nk; kt += 2) {
        k_step(kt, af[0], bf[0], af[1], bf[1]);
        k_step(kt + 1, af[1], bf[1], af[0], bf[0]);
    }
    retire_lds_reads();
    mma_step(af[0], bf[0]);
    // ---- epilogue: four quarters of 32 rows per wave through the idle ring ----
    __syncthreads();
    tile_epilogue<T, TC, Q_TM>(p, acc, (float*)smem + wave * (32 * EP_PITCH), m0 + wr * 128, n0 + wc * 64, bz, lane, R, r_late);
}

// ------------------------------------------------------------------------------------------------
// Persistent 256 x 256 tile kernel: one workgroup of eight waves (2 x 4, each 128 x 64) per CU, walking over tiles.
// Why this shape: with 256 x 128 tiles a CU takes in 48 KiB of operands per 32-deep K-step, which at the rate a CU
// ingests from L2 through the LDS DMA (about 47 bytes per clock) takes as long as the step's MFMAs; timing builds that
// re-read one operand piece instead of fetching new ones run a quarter faster, and builds that add traffic run slower.
// A 256 x 256 tile needs 32 KiB for twice the flops.  What the second co-resident workgroup gave the 256 x 128 kernels --
// something to run while a tile stores its result and the next tile waits for its first operands -- comes from the
// structure instead:
//  * the operand stream does not stop at a tile's end: the ring (4 stages of 32 KiB) is always requested four K-steps
//    ahead of the multiplication, so the next tile's first four stages are in flight before the epilogue starts;
//  * the epilogue goes through 32 KiB of LDS of its own (16 rows x 64 columns of f32 per wave, chunk-swizzled), so it
//    does not touch the ring; bias, activation and residual are applied there, in the lanes that store;
//  * inside a step the fragments of the next step are read from LDS between the MFMAs: the four B fragments into a second
//    register set at the start, each A fragment into its own registers right behind the four MFMAs that were its last
//    use -- so the MFMA stream of a wave never waits for an LDS round trip except once per tile.
// Ring protocol at the top of step g (a running count over all tiles of the workgroup): the wave waits for its pieces of
// stage g + 1 (all but the eight youngest vector memory operations: the shares of stages g + 2 and g + 3) and for its
// fragment reads of stage g, the barrier publishes stage g + 1 and retires stage g's slot, stage g + 4 is requested into
// that slot, fragments g + 1 are read while step g multiplies.  The first three steps of a tile wait for nothing: the
// epilogue before them drained the wave's pieces (a vmcnt count cannot look past the epilogue's stores).
constexpr int O_BN = 256;
constexpr int O_A = B2M * K3_ROW, O_STAGE = O_A + O_BN * K3_ROW;   // 16 + 16 KiB
constexpr int O_NSTAGE = 4;
constexpr int O_EP_WAVE = 16 * 64;                                  // floats of epilogue staging per wave
constexpr int O_LDS = O_NSTAGE * O_STAGE + 8 * O_EP_WAVE * 4;       // 160 KiB

template <typename T, typename TC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn_oct_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int KT = K3_ROW / (int)sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_n = (p.N + O_BN - 1) / O_BN, tiles_m = (p.M + B2M - 1) / B2M;
    const int tiles_mn = tiles_n * tiles_m, total = tiles_mn * p.batch;
    // virtual tile indices: the panel order pads them to eight equal runs (tile_origin_v skips the padding)
    const int total_v = p.panel_n > 0 ? 8 * ((tiles_m + 7) >> 3) * tiles_n : total;
    const int nk = p.K / KT;
    const unsigned lds_base = lds_addr_of(smem) + wave * 1024;
    const long step_a = p.kstep_a ? p.kstep_a : KT, step_b = p.kstep_b ? p.kstep_b : KT;

    auto next_valid = [&](int vt) {     // first tile index >= vt (in steps of the grid) that is not padding
        long bz; int m0, n0;
        while (vt < total_v && !tile_origin_v(p, vt, O_BN, tiles_n, tiles_mn, total, bz, m0, n0)) vt += gridDim.x;
        return vt;
    };
    // ---- the operand stream: tile, K index and ring slot of the stage to request next ----
    int dma_vt = next_valid(blockIdx.x), dma_kt = 0, slot_next = 0;
    bool dma_live = dma_vt < total_v;
    unsigned offA[2] = {0, 0}, offB[2] = {0, 0};
    const T *ka = nullptr, *kb = nullptr;
    auto dma_tile = [&]() {
        long bz; int m0, n0;
        tile_origin_v(p, dma_vt, O_BN, tiles_n, tiles_mn, total, bz, m0, n0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int L = i * 512 + tid, row = L >> 2, c = (L & 3) ^ ((-(row >> 2)) & 3);
            offA[i] = (unsigned)(((long)(min(m0 + row, p.M - 1) - m0) * p.lda + c * E) * (long)sizeof(T));
            offB[i] = (unsigned)(((long)(min(n0 + row, p.N - 1) - n0) * p.ldb + c * E) * (long)sizeof(T));
        }
        ka = (const T*)p.A + bz * p.sA + (long)m0 * p.lda;
        kb = (const T*)p.B + bz * p.sB + (long)n0 * p.ldb;
        dma_kt = 0;
    };
    auto stage_next = [&]() {
        const unsigned a_dst = lds_base + slot_next * O_STAGE;
        glds16_sv_2a2b(ka, offA, a_dst, kb, offB, a_dst + O_A);
        slot_next = (slot_next + 1) & (O_NSTAGE - 1);
        ka += step_a;
        kb += step_b;
        if (++dma_kt == nk) {
            dma_vt = next_valid(dma_vt + gridDim.x);
            dma_live = dma_vt < total_v;
            if (dma_live) dma_tile();
        }
    };
    if (dma_live) dma_tile();
#pragma unroll
    for (int i = 0; i < O_NSTAGE; ++i)
        if (dma_live) stage_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int pos = (fg ^ ((-(frow >> 2)) & 3)) << 4;
    const char* a_lane = smem + (wr * 128 + frow) * K3_ROW + pos;
    const char* b_lane = smem + O_A + (wc * 64 + frow) * K3_ROW + pos;
    int slot_read = 0;
    // fragments: A tiles 0..6 in af; the four B tiles and A tile 7 twice (xf[set][0..3] = B, xf[set][4] = A tile 7)
    uint4 af[7], xf[2][5];
    f32x4 acc[8][4];
    // One K-step: multiplies with af and xc.  MORE: a further step of this tile follows; its B fragments and A tile 7 go to
    // xn at the start of the step, A tiles 0..6 replace af one by one, each right behind the four MFMAs that were its last
    // use -- the last read is then issued four MFMAs before the step ends, not after it.
    auto k_step = [&](int kt, auto more_tag, const uint4 (&xc)[5], uint4 (&xn)[5]) {
        constexpr bool MORE = decltype(more_tag)::value;
#ifdef UW_TUNE   // timing-only builds: leave out the wait for the DMA pieces (32) or the barrier (64) to see which one parks the waves
        if (!(p.act & 32))
#endif
        if (kt >= 3) {
            if (dma_live) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stream has ended: fewer pieces are outstanding
        }
        retire_lds_reads();
#ifdef UW_TUNE
        if (!(p.act & 64))
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef UW_TUNE   // 128: no operand requests at all (stale LDS contents); 256: every fragment read comes from one stage
        if (!(p.act & 128))
#endif
        if (dma_live) stage_next();
        const int at = slot_read * O_STAGE;
        if constexpr (MORE) {
#pragma unroll
            for (int t = 0; t < 4; ++t) xn[t] = *(const uint4*)(b_lane + at + t * 16 * K3_ROW);
            xn[4] = *(const uint4*)(a_lane + at + 7 * 16 * K3_ROW);
        }
#pragma unroll
        for (int tm = 0; tm < 8; ++tm) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) mma16<T>(xc[tn], tm < 7 ? af[tm] : xc[4], acc[tm][tn]);
#ifdef UW_TUNE   // 512: no A-fragment reads (the first step's fragments are used throughout)
            if (!(p.act & 512))
#endif
            if constexpr (MORE)
                if (tm < 7) af[tm] = *(const uint4*)(a_lane + at + tm * 16 * K3_ROW);
        }
        if constexpr (MORE) {
            slot_read = (slot_read + 1) & (O_NSTAGE - 1);
            // pin the order above: 5 LDS reads, then 7 x (4 MFMAs, 1 LDS read), then the last 4 MFMAs
            constexpr int N_MFMA = sizeof(T) == 2 ? 4 : 16;           // instructions per group (an f32 chunk is four MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA, 0);
        }
    };
    const std::true_type more;
    const std::false_type last;

    float* ep = (float*)(smem + O_NSTAGE * O_STAGE) + wave * O_EP_WAVE;
    constexpr int KTC = K3_ROW / (int)sizeof(TC);
    for (int vt = next_valid(blockIdx.x); vt < total_v; vt = next_valid(vt + gridDim.x)) {
        long bz; int m0, n0;
        tile_origin_v(p, vt, O_BN, tiles_n, tiles_mn, total, bz, m0, n0);
        // every wave has drained its pieces of this tile's first four stages (prologue, or the epilogue before)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto first_frags = [&](uint4 (&x)[5]) {
            const int at = slot_read * O_STAGE;
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = *(const uint4*)(b_lane + at + t * 16 * K3_ROW);
#pragma unroll
            for (int t = 0; t < 7; ++t) af[t] = *(const uint4*)(a_lane + at + t * 16 * K3_ROW);
            x[4] = *(const uint4*)(a_lane + at + 7 * 16 * K3_ROW);
            slot_read = (slot_read + 1) & (O_NSTAGE - 1);
        };
        // nk - 1 steps read ahead, the last one only multiplies; an odd count starts in set 1 so that the last step
        // always finds its fragments in set 0
        int kt = 0;
        if ((nk - 1) & 1) {
            first_frags(xf[1]);
            k_step(0, more, xf[1], xf[0]);
            kt = 1;
        } else {
            first_frags(xf[0]);
        }
        for (; kt + 1 < nk; kt += 2) {
            k_step(kt, more, xf[0], xf[1]);
            k_step(kt + 1, more, xf[1], xf[0]);
        }
        k_step(kt, last, xf[0], xf[1]);

        // ---- epilogue: eight parts of 16 rows per wave through the wave's own staging slab ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next tile's first stages have landed
        TC* __restrict__ C = (TC*)p.C + bz * p.sC;
        const float* __restrict__ R = p.residual ? p.residual + bz * p.sR : nullptr;
        const int mw0 = m0 + wr * 128, nbase = n0 + wc * 64;
        auto coff = [&](int m, int n) -> long {
            return p.c_kblock ? ((long)(n / KTC) * p.c_kblock + m) * KTC + (n % KTC) : c_off(p, m, n);
        };
        constexpr int CPL = sizeof(TC) == 4 ? 4 : 8;         // columns a lane stores per row
        constexpr int RPI = sizeof(TC) == 4 ? 4 : 8;         // rows per store instruction
        const int nl = (lane % (64 / CPL)) * CPL, n_lane = nbase + nl;
        float bias_n[CPL];
#pragma unroll
        for (int r = 0; r < CPL; ++r) bias_n[r] = (p.bias_mode == 1 && n_lane + r < p.N) ? p.bias[n_lane + r] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 8; ++tm) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) *(f32x4*)(ep + frow * 64 + (((tn * 4 + fg) ^ frow) << 2)) = acc[tm][tn];
            const int mbase = mw0 + tm * 16;
            if (p.C2 && nbase >= p.c2_col0) {
                // transposed copy: lane -> four consecutive rows of one column; four lanes fill 32 bytes of a C2 row
                TC* __restrict__ C2 = (TC*)p.C2 + bz * p.sC2;
                const int m4 = (lane & 3) * 4;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int cl = it * 16 + (lane >> 2);
                    const int m = mbase + m4, n = nbase + cl;
                    if (n >= p.N || m >= p.M) continue;
                    const float bn = p.bias_mode == 1 ? p.bias[n] : 0.f;
                    union { TC h[4]; uint2 u2; f32x4 u4; } pk;
                    TC* const h = pk.h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = from_f32<TC>(ep[(m4 + r) * 64 + ((((cl >> 2) ^ (m4 + r)) << 2) | (cl & 3))] + bn);
                    TC* dst = C2 + (long)(n - p.c2_col0) * p.ldc2 + m;
                    if (m + 3 < p.M && (p.ldc2 & 3) == 0) {
                        if constexpr (sizeof(TC) == 4) *(f32x4*)dst = pk.u4;
                        else *(uint2*)dst = pk.u2;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (m + r < p.M) dst[r] = h[r];
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 16 / RPI; ++it) {
                const int ml = it * RPI + lane / (64 / CPL);
                const int m = mbase + ml, n = n_lane;
                if (m >= p.M || n >= p.N) continue;
                float v[CPL];
#pragma unroll
                for (int q = 0; q < CPL / 4; ++q) {
                    const f32x4 t = *(const f32x4*)(ep + ml * 64 + ((((nl >> 2) + q) ^ ml) << 2));
                    v[q * 4 + 0] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
                }
                const float bias_m = p.bias_mode == 2 ? p.bias[m] : 0.f;
#pragma unroll
                for (int r = 0; r < CPL; ++r) {
                    v[r] += bias_n[r] + bias_m;
                    if (p.act == 1) v[r] = gelu_erf(v[r]);
                }
                const bool full = n + CPL - 1 < p.N;
                if constexpr (sizeof(TC) == 4) {
                    if (R) {
                        if (full && (p.ldr & 3) == 0) {
                            const f32x4 rv = *(const f32x4*)(R + (long)m * p.ldr + n);
                            v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (n + r < p.N) ? R[(long)m * p.ldr + n + r] : 0.f;
                        }
                    }
                    if (full && (p.ldc & 3) == 0) {
                        STORE16_NT((float*)C + coff(m, n), (f32x4{v[0], v[1], v[2], v[3]}));
                    } else {
                        for (int r = 0; r < 4 && n + r < p.N; ++r) ((float*)C)[coff(m, n) + r] = v[r];
                    }
                } else {
                    if (full && (p.ldc & 7) == 0 && (((uintptr_t)C & 15) == 0)) {
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
}

template <typename T, typename TC> int launch_oct(const GemmParams& p, hipStream_t s)
{
    const long tiles = (long)ceil_div(p.M, B2M) * ceil_div(p.N, O_BN) * p.batch;
    UW_REQUIRE(tiles < (1l << 28), "gemm: too many tiles");
    GemmParams q = p;
    q.panel_n = choose_panel((long)sizeof(T), p, O_BN);
    static PerDeviceFlag attr_set;
    static int n_cu[64];
    if (attr_set.pending()) {
        UW_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_oct_kernel<T, TC>, hipFuncAttributeMaxDynamicSharedMemorySize, O_LDS));
        int dev = 0, cus = 0;
        UW_HIP_CHECK(hipGetDevice(&dev));
        UW_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        n_cu[dev & 63] = cus > 0 ? cus : 256;
        attr_set.mark();
    }
    int dev = 0;
    UW_HIP_CHECK(hipGetDevice(&dev));
    // one workgroup per CU (the 160 KiB of LDS admit no second one); every workgroup walks over tiles blockIdx.x,
    // blockIdx.x + grid, ... -- the grid is a multiple of 8 so that a workgroup stays on the tiles of one XCD's run
    const long vt_total = q.panel_n > 0 ? 8l * ceil_div(ceil_div(p.M, B2M), 8) * ceil_div(p.N, O_BN) : tiles;
    const unsigned grid = (unsigned)std::min<long>(n_cu[dev & 63] & ~7, (vt_total + 7) & ~7l);
    hipLaunchKernelGGL((gemm_tn_oct_kernel<T, TC>), dim3(grid), dim3(512), O_LDS, s, q);
    UW_LAUNCH_CHECK();
    return UW_OK;
}


// eight LDS-DMA pieces in one statement: a wave's share of one ring stage of the four-wave kernel below
__device__ __forceinline__ void glds16_sv_4a4b(const void* abase, const unsigned (&aoff)[4], unsigned a_lds, const void* bbase,
                                               const unsigned (&boff)[4], unsigned b_lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %11\n\t"
                 "s_add_u32 m0, %9, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %11\n\t"
                 "s_add_u32 m0, %9, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %11\n\t"
                 "s_add_u32 m0, %9, 0x3000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %11\n\t"
                 "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %12\n\t"
                 "s_add_u32 m0, %10, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %12\n\t"
                 "s_add_u32 m0, %10, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %12\n\t"
                 "s_add_u32 m0, %10, 0x3000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %12\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(aoff[0]), "v"(aoff[1]), "v"(aoff[2]), "v"(aoff[3]), "v"(boff[0]), "v"(boff[1]), "v"(boff[2]), "v"(boff[3]), "s"(a_lds),
                   "s"(b_lds), "s"(abase), "s"(bbase)
                 : "memory", "scc");
}

// ------------------------------------------------------------------------------------------------
// hex: the oct kernel with FOUR waves of 128 x 128 (the wave tile of the vendor library's kernel for these shapes): half
// the LDS fragment reads per MFMA of a 128 x 64 wave tile, one wave per SIMD with the accumulators in 256 registers.
constexpr int X_BN = 256;
constexpr int X_A = B2M * K3_ROW, X_STAGE = X_A + X_BN * K3_ROW;   // 16 + 16 KiB
constexpr int X_NSTAGE = 4;
constexpr int X_EP_WAVE = 16 * 64;                                  // floats of epilogue staging per wave
constexpr int X_LDS = X_NSTAGE * X_STAGE + 4 * X_EP_WAVE * 4;       // 144 KiB

template <typename T, typename TC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_tn_hex_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int KT = K3_ROW / (int)sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_n = (p.N + X_BN - 1) / X_BN, tiles_m = (p.M + B2M - 1) / B2M;
    const int tiles_mn = tiles_n * tiles_m, total = tiles_mn * p.batch;
    // virtual tile indices: the panel order pads them to eight equal runs (tile_origin_v skips the padding)
    const int total_v = p.panel_n > 0 ? 8 * ((tiles_m + 7) >> 3) * tiles_n : total;
    const int nk = p.K / KT;
    const unsigned lds_base = lds_addr_of(smem) + wave * 1024;
    const long step_a = p.kstep_a ? p.kstep_a : KT, step_b = p.kstep_b ? p.kstep_b : KT;

    auto next_valid = [&](int vt) {     // first tile index >= vt (in steps of the grid) that is not padding
        long bz; int m0, n0;
        while (vt < total_v && !tile_origin_v(p, vt, X_BN, tiles_n, tiles_mn, total, bz, m0, n0)) vt += gridDim.x;
        return vt;
    };
    // ---- the operand stream: tile, K index and ring slot of the stage to request next ----
    int dma_vt = next_valid(blockIdx.x), dma_kt = 0, slot_next = 0;
    bool dma_live = dma_vt < total_v;
    unsigned offA[4] = {0, 0, 0, 0}, offB[4] = {0, 0, 0, 0};
    const T *ka = nullptr, *kb = nullptr;
    auto dma_tile = [&]() {
        long bz; int m0, n0;
        tile_origin_v(p, dma_vt, X_BN, tiles_n, tiles_mn, total, bz, m0, n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int L = i * 256 + tid, row = L >> 2, c = (L & 3) ^ ((-(row >> 2)) & 3);
            offA[i] = (unsigned)(((long)(min(m0 + row, p.M - 1) - m0) * p.lda + c * E) * (long)sizeof(T));
            offB[i] = (unsigned)(((long)(min(n0 + row, p.N - 1) - n0) * p.ldb + c * E) * (long)sizeof(T));
        }
        ka = (const T*)p.A + bz * p.sA + (long)m0 * p.lda;
        kb = (const T*)p.B + bz * p.sB + (long)n0 * p.ldb;
        dma_kt = 0;
    };
    auto stage_next = [&]() {
        const unsigned a_dst = lds_base + slot_next * X_STAGE;
        glds16_sv_4a4b(ka, offA, a_dst, kb, offB, a_dst + X_A);
        slot_next = (slot_next + 1) & (X_NSTAGE - 1);
        ka += step_a;
        kb += step_b;
        if (++dma_kt == nk) {
            dma_vt = next_valid(dma_vt + gridDim.x);
            dma_live = dma_vt < total_v;
            if (dma_live) dma_tile();
        }
    };
    if (dma_live) dma_tile();
#pragma unroll
    for (int i = 0; i < X_NSTAGE; ++i)
        if (dma_live) stage_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int pos = (fg ^ ((-(frow >> 2)) & 3)) << 4;
    const char* a_lane = smem + (wr * 128 + frow) * K3_ROW + pos;
    const char* b_lane = smem + X_A + (wc * 128 + frow) * K3_ROW + pos;
    int slot_read = 0;
    // fragments: A tiles 0..6 in af; the four B tiles and A tile 7 twice (xf[set][0..3] = B, xf[set][4] = A tile 7)
    uint4 af[7], xf[2][9];
    f32x4 acc[8][8];
    // One K-step: multiplies with af and xc.  MORE: a further step of this tile follows; its B fragments and A tile 7 go to
    // xn at the start of the step, A tiles 0..6 replace af one by one, each right behind the four MFMAs that were its last
    // use -- the last read is then issued four MFMAs before the step ends, not after it.
    auto k_step = [&](int kt, auto more_tag, const uint4 (&xc)[9], uint4 (&xn)[9]) {
        constexpr bool MORE = decltype(more_tag)::value;
#ifdef UW_TUNE   // timing-only builds: leave out the wait for the DMA pieces (32) or the barrier (64) to see which one parks the waves
        if (!(p.act & 32))
#endif
        if (kt >= 3) {
            if (dma_live) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stream has ended: fewer pieces are outstanding
        }
        retire_lds_reads();
#ifdef UW_TUNE
        if (!(p.act & 64))
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef UW_TUNE   // 128: no operand requests at all (stale LDS contents); 256: every fragment read comes from one stage
        if (!(p.act & 128))
#endif
        if (dma_live) stage_next();
        const int at = slot_read * X_STAGE;
        if constexpr (MORE) {
#pragma unroll
            for (int t = 0; t < 8; ++t) xn[t] = *(const uint4*)(b_lane + at + t * 16 * K3_ROW);
            xn[8] = *(const uint4*)(a_lane + at + 7 * 16 * K3_ROW);
        }
#pragma unroll
        for (int tm = 0; tm < 8; ++tm) {
#pragma unroll
            for (int tn = 0; tn < 8; ++tn) mma16<T>(xc[tn], tm < 7 ? af[tm] : xc[8], acc[tm][tn]);
#ifdef UW_TUNE   // 512: no A-fragment reads (the first step's fragments are used throughout)
            if (!(p.act & 512))
#endif
            if constexpr (MORE)
                if (tm < 7) af[tm] = *(const uint4*)(a_lane + at + tm * 16 * K3_ROW);
        }
        if constexpr (MORE) {
            slot_read = (slot_read + 1) & (X_NSTAGE - 1);
            // pin the order above: 9 LDS reads, then 7 x (8 MFMAs, 1 LDS read), then the last 8 MFMAs
            constexpr int N_MFMA = sizeof(T) == 2 ? 8 : 32;           // instructions per group (an f32 chunk is four MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA, 0);
        }
    };
    const std::true_type more;
    const std::false_type last;

    float* ep = (float*)(smem + X_NSTAGE * X_STAGE) + wave * X_EP_WAVE;   // 4 waves x 4 KiB
    constexpr int KTC = K3_ROW / (int)sizeof(TC);
    for (int vt = next_valid(blockIdx.x); vt < total_v; vt = next_valid(vt + gridDim.x)) {
        long bz; int m0, n0;
        tile_origin_v(p, vt, X_BN, tiles_n, tiles_mn, total, bz, m0, n0);
        // every wave has drained its pieces of this tile's first four stages (prologue, or the epilogue before)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto first_frags = [&](uint4 (&x)[9]) {
            const int at = slot_read * X_STAGE;
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = *(const uint4*)(b_lane + at + t * 16 * K3_ROW);
#pragma unroll
            for (int t = 0; t < 7; ++t) af[t] = *(const uint4*)(a_lane + at + t * 16 * K3_ROW);
            x[8] = *(const uint4*)(a_lane + at + 7 * 16 * K3_ROW);
            slot_read = (slot_read + 1) & (X_NSTAGE - 1);
        };
        // nk - 1 steps read ahead, the last one only multiplies; an odd count starts in set 1 so that the last step
        // always finds its fragments in set 0
        int kt = 0;
        if ((nk - 1) & 1) {
            first_frags(xf[1]);
            k_step(0, more, xf[1], xf[0]);
            kt = 1;
        } else {
            first_frags(xf[0]);
        }
        for (; kt + 1 < nk; kt += 2) {
            k_step(kt, more, xf[0], xf[1]);
            k_step(kt + 1, more, xf[1], xf[0]);
        }
        k_step(kt, last, xf[0], xf[1]);

        // ---- epilogue: eight parts of 16 rows per wave through the wave's own staging slab ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next tile's first stages have landed
        TC* __restrict__ C = (TC*)p.C + bz * p.sC;
        const float* __restrict__ R = p.residual ? p.residual + bz * p.sR : nullptr;
        const int mw0 = m0 + wr * 128;
        auto epilogue_half = [&](auto half_tag) {      // a compile-time half: the accumulators must stay in registers
        constexpr int half = decltype(half_tag)::value;
        const int nbase = n0 + wc * 128 + half * 64;
        auto coff = [&](int m, int n) -> long {
            return p.c_kblock ? ((long)(n / KTC) * p.c_kblock + m) * KTC + (n % KTC) : c_off(p, m, n);
        };
        constexpr int CPL = sizeof(TC) == 4 ? 4 : 8;         // columns a lane stores per row
        constexpr int RPI = sizeof(TC) == 4 ? 4 : 8;         // rows per store instruction
        const int nl = (lane % (64 / CPL)) * CPL, n_lane = nbase + nl;
        float bias_n[CPL];
#pragma unroll
        for (int r = 0; r < CPL; ++r) bias_n[r] = (p.bias_mode == 1 && n_lane + r < p.N) ? p.bias[n_lane + r] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 8; ++tm) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) *(f32x4*)(ep + frow * 64 + (((tn * 4 + fg) ^ frow) << 2)) = acc[tm][half * 4 + tn];
            const int mbase = mw0 + tm * 16;
            if (p.C2 && nbase >= p.c2_col0) {
                // transposed copy: lane -> four consecutive rows of one column; four lanes fill 32 bytes of a C2 row
                TC* __restrict__ C2 = (TC*)p.C2 + bz * p.sC2;
                const int m4 = (lane & 3) * 4;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int cl = it * 16 + (lane >> 2);
                    const int m = mbase + m4, n = nbase + cl;
                    if (n >= p.N || m >= p.M) continue;
                    const float bn = p.bias_mode == 1 ? p.bias[n] : 0.f;
                    union { TC h[4]; uint2 u2; f32x4 u4; } pk;
                    TC* const h = pk.h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = from_f32<TC>(ep[(m4 + r) * 64 + ((((cl >> 2) ^ (m4 + r)) << 2) | (cl & 3))] + bn);
                    TC* dst = C2 + (long)(n - p.c2_col0) * p.ldc2 + m;
                    if (m + 3 < p.M && (p.ldc2 & 3) == 0) {
                        if constexpr (sizeof(TC) == 4) *(f32x4*)dst = pk.u4;
                        else *(uint2*)dst = pk.u2;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (m + r < p.M) dst[r] = h[r];
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 16 / RPI; ++it) {
                const int ml = it * RPI + lane / (64 / CPL);
                const int m = mbase + ml, n = n_lane;
                if (m >= p.M || n >= p.N) continue;
                float v[CPL];
#pragma unroll
                for (int q = 0; q < CPL / 4; ++q) {
                    const f32x4 t = *(const f32x4*)(ep + ml * 64 + ((((nl >> 2) + q) ^ ml) << 2));
                    v[q * 4 + 0] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
                }
                const float bias_m = p.bias_mode == 2 ? p.bias[m] : 0.f;
#pragma unroll
                for (int r = 0; r < CPL; ++r) {
                    v[r] += bias_n[r] + bias_m;
                    if (p.act == 1) v[r] = gelu_erf(v[r]);
                }
                const bool full = n + CPL - 1 < p.N;
                if constexpr (sizeof(TC) == 4) {
                    if (R) {
                        if (full && (p.ldr & 3) == 0) {
                            const f32x4 rv = *(const f32x4*)(R + (long)m * p.ldr + n);
                            v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (n + r < p.N) ? R[(long)m * p.ldr + n + r] : 0.f;
                        }
                    }
                    if (full && (p.ldc & 3) == 0) {
                        STORE16_NT((float*)C + coff(m, n), (f32x4{v[0], v[1], v[2], v[3]}));
                    } else {
                        for (int r = 0; r < 4 && n + r < p.N; ++r) ((float*)C)[coff(m, n) + r] = v[r];
                    }
                } else {
                    if (full && (p.ldc & 7) == 0 && (((uintptr_t)C & 15) == 0)) {
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
        };
        epilogue_half(std::integral_constant<int, 0>{});
        epilogue_half(std::integral_constant<int, 1>{});
    }
}

template <typename T, typename TC> int launch_hex(const GemmParams& p, hipStream_t s)
{
    const long tiles = (long)ceil_div(p.M, B2M) * ceil_div(p.N, X_BN) * p.batch;
    UW_REQUIRE(tiles < (1l << 28), "gemm: too many tiles");
    GemmParams q = p;
    q.panel_n = choose_panel((long)sizeof(T), p, X_BN);
    static PerDeviceFlag attr_set;
    static int n_cu[64];
    if (attr_set.pending()) {
        UW_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_hex_kernel<T, TC>, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS));
        int dev = 0, cus = 0;
        UW_HIP_CHECK(hipGetDevice(&dev));
        UW_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        n_cu[dev & 63] = cus > 0 ? cus : 256;
        attr_set.mark();
    }
    int dev = 0;
    UW_HIP_CHECK(hipGetDevice(&dev));
    // one workgroup per CU (the 160 KiB of LDS admit no second one); every workgroup walks over tiles blockIdx.x,
    // blockIdx.x + grid, ... -- the grid is a multiple of 8 so that a workgroup stays on the tiles of one XCD's run
    const long vt_total = q.panel_n > 0 ? 8l * ceil_div(ceil_div(p.M, B2M), 8) * ceil_div(p.N, X_BN) : tiles;
    const unsigned grid = (unsigned)std::min<long>(n_cu[dev & 63] & ~7, (vt_total + 7) & ~7l);
    hipLaunchKernelGGL((gemm_tn_hex_kernel<T, TC>), dim3(grid), dim3(256), X_LDS, s, q);
    UW_LAUNCH_CHECK();
    return UW_OK;
}


template <typename T, typename TC> int launch_quad(const GemmParams& p, hipStream_t s)
{
    GemmParams q = p;
    q.panel_n = choose_panel((long)sizeof(T), p);
    const long tiles = (long)ceil_div(p.M, B2M) * ceil_div(p.N, B2N) * p.batch;
    unsigned grid = (unsigned)tiles;
    if (q.panel_n > 0) grid = 8u * (unsigned)ceil_div(ceil_div(p.M, B2M), 8) * (unsigned)ceil_div(p.N, B2N);
    const size_t lds = (size_t)NSTAGE3 * STAGE3;
    UW_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_quad_kernel<T, TC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gemm_tn_quad_kernel<T, TC>), dim3(grid), dim3(256), lds, s, q);
    UW_LAUNCH_CHECK();
    return UW_OK;
}

}  // namespace

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
    const char* variant = getenv("VARIANT") ? getenv("VARIANT") : "duo";
    struct Shape { int K, N, out_f32, act, res; const char* name; } shapes[] = {
        {768, 3072, 0, 1, 0, "fc1"}, {768, 1536, 0, 0, 0, "qk"}, {768, 768, 1, 0, 1, "out+res"}, {3072, 768, 1, 0, 1, "fc2"}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total_us = 0;
    printf("variant %s\n", variant);
    for (auto& sh : shapes) {
        char *W, *A; float *C, *bias;
        CK(hipMalloc(&W, (size_t)sh.N * sh.K * 2)); CK(hipMalloc(&A, (size_t)M * sh.K * 2)); CK(hipMalloc(&C, (size_t)M * sh.N * 4)); CK(hipMalloc(&bias, sh.N * 4));
        std::vector<uint16_t> hw((size_t)sh.N * sh.K), ha((size_t)M * sh.K);
        fill_bf16(hw, 1, 0.03f); fill_bf16(ha, 2, 1.0f);
        CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(C, 0, (size_t)M * sh.N * 4)); CK(hipMemset(bias, 0, sh.N * 4));
        GemmParams p; p.A = A; p.B = W; p.C = C; p.bias = bias; p.lda = sh.K; p.ldb = sh.K; p.ldc = sh.N; p.ldr = sh.N; p.M = M; p.N = sh.N; p.K = sh.K;
        if (getenv("KBLOCK")) { const int kb = atoi(getenv("KBLOCK")); if (kb & 1) { p.lda = 32; p.kstep_a = (long)M * 32; } if (kb & 2) { p.ldb = 32; p.kstep_b = (long)sh.N * 32; } }
        p.bias_mode = 1; p.act = getenv("NOACT") ? 0 : sh.act; p.residual = sh.res ? C : nullptr;
        if (getenv("DMA_MODE")) p.act |= atoi(getenv("DMA_MODE"));
        auto run = [&]() -> int {
            if (!strcmp(variant, "quad")) return sh.out_f32 ? launch_quad<bf16_t, float>(p, s) : launch_quad<bf16_t, bf16_t>(p, s);
            if (!strcmp(variant, "oct")) return sh.out_f32 ? launch_oct<bf16_t, float>(p, s) : launch_oct<bf16_t, bf16_t>(p, s);
            if (!strcmp(variant, "hex")) return sh.out_f32 ? launch_hex<bf16_t, float>(p, s) : launch_hex<bf16_t, bf16_t>(p, s);
            return uw_gemm(UW_BF16, sh.out_f32, p, s);
        };
        if (getenv("CHECK")) {   // the variant against the product kernel on the same operands (no residual: C is not an input then)
            p.residual = nullptr;
            const size_t cbytes = (size_t)M * sh.N * (sh.out_f32 ? 4 : 2);
            std::vector<char> c0(cbytes), c1(cbytes);
            CK(hipMemset(C, 0, cbytes));
            if (uw_gemm(UW_BF16, sh.out_f32, p, s)) return 1;
            CK(hipStreamSynchronize(s)); CK(hipMemcpy(c0.data(), C, cbytes, hipMemcpyDeviceToHost));
            CK(hipMemset(C, 0, cbytes));
            if (run()) return 1;
            CK(hipStreamSynchronize(s)); CK(hipMemcpy(c1.data(), C, cbytes, hipMemcpyDeviceToHost));
            size_t diff = 0;
            for (size_t i = 0; i < cbytes; ++i) diff += c0[i] != c1[i];
            printf("%-8s check: %zu of %zu bytes differ from the product kernel\n", sh.name, diff, cbytes);
            CK(hipFree(W)); CK(hipFree(A)); CK(hipFree(C)); CK(hipFree(bias));
            continue;
        }
        const int iters = 6;
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) if (run()) return 1;
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
