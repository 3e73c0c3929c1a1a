// Microbenchmark for VERDICT r5 item 1: can the filter kernels' epilogue (vector ALU, matrix pipe idle) run under the PARTNER
// wave's MFMAs?  The stage of csrc/pstat_engine.h (P block stationary in registers, Q slabs through an LDS ring by LDS-DMA,
// one s_barrier per stage) with a SYNTHETIC epilogue behind every unit (EPI passes of sub + max over the accumulators against
// thresholds read from LDS, one scalar gate per accumulator tile, start values of the next unit read from LDS), in three
// structures:
//   lockstep   NW = 8, SK = 0   all eight waves multiply, then all eight run their epilogue (the shipped schedule)
//   skewed     NW = 8, SK > 0   waves 4-7 work SK stages behind waves 0-3 on the same Q stream (ring SK slots deeper); a wave's
//                               epilogue runs BEHIND the barrier that ends its unit, so its SIMD partner - in the middle of its
//                               own unit - is multiplying meanwhile
//   two WGs    NW = 4, WGS = 2  two independent 256-thread workgroups per CU (own ring, own barriers, P block of 128 rows):
//                               nothing couples the two waves of a SIMD
// Part 1 (ovl<>) is the elementary question: waves 0-3 issue only v_mfma_f32_32x32x16_f16, waves 4-7 only vector ALU work -
// does the pair take max(a, b) or a + b, in cycles (s_memtime) and in wall time (events; the clock follows the power draw)?
// Unit of part 2: us per 256 MFMAs of a CU (8.4 MFLOP; tools/ubench/pstat.hip's unit).  Data are meaningless.
// Build: hipcc --offload-arch=gfx950 -O3 pskew.hip -o pskew
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <initializer_list>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int WROW = 32;                 // LDS row: 32 words = 128 B = 64 f16

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N < 64, "vmcnt");
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));      // vmcnt(N), lgkmcnt(0)
}
__device__ __forceinline__ unsigned long long lanes_ge(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 3); }

// where a wave runs: HW_REG_XCC_ID (20) bits 3:0, HW_REG_HW_ID (4): cu_id 11:8, sh_id 12, se_id 15:13
__device__ __forceinline__ unsigned hw_where() {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20), hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    return (xcc << 16) | (hw & 0xffffu);
}
__device__ __forceinline__ void poll_ge(const int* p, int need) {       // LDS flag: wait until *p >= need
    while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int*>(p)) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void signal_add(int* p, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------- part 1
// mode bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run vector work; bit 2: ALL waves interleave (4 MFMAs + VPER vector ops)
// mode bit 3: roles swapped (waves 0-3 vector work, 4-7 MFMAs); bits 5:4: s_setprio of the vector waves
template <int VPER>
__global__ void __launch_bounds__(512, 1) ovl(int mode, int iters, const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ clk) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (tid * 8 + blockIdx.x * 4096) % 65536), b0 = *reinterpret_cast<const f32x4*>(src + (tid * 8 + 4 + blockIdx.x * 4096) % 65536);
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    float r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = a0.x + j;
    const float th = b0.y;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    if (mode & 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), acc[m], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < VPER / 4; ++j) r[(m * (VPER / 4) + j) & 15] = fmaxf(r[(m * (VPER / 4) + j) & 15] - th, r[(m * (VPER / 4) + j + 1) & 15]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if ((wave < 4) != ((mode & 8) != 0)) {
        // bit 6: ONE accumulator (every MFMA waits for its predecessor's result: is a not-yet-ready MFMA kinder to the partner's
        // vector issue than a ready one queued behind a busy pipe?), bit 7: two accumulators alternating
        if ((mode & 1) && (mode & 64))
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), acc[0], 0, 0, 0);
        else if ((mode & 1) && (mode & 128))
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), acc[m & 1], 0, 0, 0);
        else if (mode & 1)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), acc[m], 0, 0, 0);
    } else {
        if (((mode >> 4) & 3) == 1) __builtin_amdgcn_s_setprio(1);
        if (((mode >> 4) & 3) == 2) __builtin_amdgcn_s_setprio(2);
        if (((mode >> 4) & 3) == 3) __builtin_amdgcn_s_setprio(3);
        if (mode & 2)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int j = 0; j < VPER; ++j) r[j & 15] = fmaxf(r[j & 15] - th, r[(j + 1) & 15]);
    }
    float sink = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) sink += acc[m][3];
#pragma unroll
    for (int j = 0; j < 16; ++j) sink += r[j];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    if (sink == 12345.678f) out[0] = sink;
    if ((tid & 63) == 0 && (wave == 0 || wave == 4)) {           // per block: waves 0 and 4 (one SIMD's pair)
        unsigned long long* rec = clk + ((int64_t)blockIdx.x * 2 + (((wave >> 2) != 0) != ((mode & 8) != 0))) * 4;   // class 0: the MFMA side
        rec[0] = t1 - t0;
        rec[1] = w0;
        rec[2] = w1;
        rec[3] = hw_where();
    }
}

// ---------------------------------------------------------------------------------------------------------------- part 2
// Pieces shared by the two stage kernels.  A work item is `units` units; the grid holds ROUNDS work items per resident slot, so
// the hardware dispatcher balances the CUs as it does for the real kernels (CUs do not all run at the same clock).
struct Rec {
    unsigned long long t0, w0;
    __device__ __forceinline__ void start() { t0 = __builtin_amdgcn_s_memtime(); w0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void stop(unsigned long long* clk, int idx) {
        unsigned long long* rec = clk + (int64_t)idx * 4;
        rec[0] = __builtin_amdgcn_s_memtime() - t0;
        rec[1] = w0;
        rec[2] = __builtin_amdgcn_s_memrealtime();
        rec[3] = hw_where();
    }
};

// The synthetic epilogue of one unit: EPI passes (+ EVAR more on a pseudo-random half of the (wave, unit) pairs: the real detail
// path is data dependent) of 16 sub + 32 max per accumulator tile against thresholds from LDS, one scalar gate per tile.
template <int MT, int NP, int EPI, int EVAR, int PRIO = 0>
struct Epi {
    float run_max, xs;
    int hits;
    const float* aux;
    int h, wave;
    __device__ __forceinline__ void pass(f32x16 (&acc)[MT][NP], int u, int p) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 th[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) th[g4] = *reinterpret_cast<const f32x4*>(aux + (p + 1) * 256 + (u & 1) * 128 + (m & 3) * 32 + g4 * 8 + h * 4);
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                float am = -INFINITY, wm = -INFINITY;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        am = fmaxf(am, acc[m][n][g4 * 4 + e]);
                        wm = fmaxf(wm, acc[m][n][g4 * 4 + e] - th[g4][e]);
                    }
                run_max = fmaxf(run_max, am);
                if (lanes_ge(wm, xs) != 0ull) ++hits;          // never taken (xs ~ 1e30): the gate's cost, not the detail path's
            }
        }
    }
    __device__ __forceinline__ void run(f32x16 (&acc)[MT][NP], int u) {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NP; ++n) run_max = fmaxf(run_max, acc[m][n][5]);      // (EPI = 0: the unit's MFMAs stay alive)
#pragma unroll
        for (int p = 0; p < EPI; ++p) pass(acc, u, p);
        if (EVAR > 0 && ((((unsigned)u * 2654435761u + (unsigned)wave * 0x9E3779B1u) >> 13) & 1u)) {
#pragma unroll
            for (int p = 0; p < EVAR; ++p) pass(acc, u, EPI + p);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    __device__ __forceinline__ void init(f32x16 (&acc)[MT][NP], int u) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(aux + ((u & 1) * 128 + (m & 3) * 32 + g4 * 8 + h * 4));
#pragma unroll
                for (int n = 0; n < NP; ++n) { acc[m][n][g4 * 4 + 0] = v.x; acc[m][n][g4 * 4 + 1] = v.y; acc[m][n][g4 * 4 + 2] = v.z; acc[m][n][g4 * 4 + 3] = v.w; }
            }
    }
};

// ---- barrier-synchronised stage (the shipped structure and its variants)
// NW waves per workgroup, QR Q rows per stage / unit (MT = QR / 32 accumulator tiles per P tile), NP 32-row P tiles per wave,
// KS 64-element slabs per unit, SK stage lag of waves 4-7 (NW = 8 only), WGS workgroups per CU
template <int NW, int QR, int NP, int KS, int EPI, int EVAR, int SK, int WGS, int PRIO = 0>
__global__ void __launch_bounds__(NW * 64, (WGS * NW) / 4) stage(const float* __restrict__ src, int64_t src_rows, int ld_words, int units,
                                                                  float* __restrict__ out, unsigned long long* __restrict__ clk) {
    constexpr int MT = QR / 32;
    constexpr int STAGE_WORDS = QR * WROW;
    constexpr int PIECES = QR / 8 / NW;               // 1 KB LDS-DMA pieces per wave and stage
    constexpr int DEPTH = 3;                          // stages in flight ahead of the leading group
    constexpr int RING = SK == 0 ? 4 : (DEPTH + SK + 1 <= 4 ? 4 : (DEPTH + SK + 1 <= 8 ? 8 : 16));
    static_assert(PIECES >= 1, "pieces");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* aux = lds + RING * STAGE_WORDS;            // [EPI + EVAR + 1][256] thresholds / start values
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int64_t nblk = src_rows / 512;
    for (int i = tid; i < (EPI + EVAR + 1) * 256; i += NW * 64) aux[i] = src[i] * 1e-3f;

    constexpr int PREG = (KS < 6 ? KS : 6) * 4;       // (the shipped engine keeps slabs 7-8 in LDS: the register budget of D = 512)
    f32x4 pf[NP][PREG];
    {
        const int64_t pblk = ((int64_t)blockIdx.x * 5 + 3) % nblk;
        const float* prow = src + (pblk * 512 + (wave * NP * 32) % 512 + r) * (int64_t)ld_words + h * 4;
#pragma unroll
        for (int n = 0; n < NP; ++n)
#pragma unroll
            for (int s = 0; s < PREG; ++s) pf[n][s] = *reinterpret_cast<const f32x4*>(prow + (int64_t)n * 32 * ld_words + s * 8);
    }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x7fffffff, 0x00020000);
    const int lr = lane >> 3, slot8 = lane & 7;
    unsigned vo[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = par * 8 + lr;
        vo[par] = (unsigned)((row * ld_words + (slot8 ^ ((row >> 1) & 7)) * 4) * 4);
    }
    const unsigned nb = (unsigned)nblk, blk_bytes = (unsigned)(512 * ld_words * 4);
    unsigned fblk = (unsigned)(((int64_t)blockIdx.x * 7) % nblk);
    int fk = 0, fslot = 0;                            // fetch cursor: slab of the unit, ring slot
    auto fetch = [&](int i) {
        const int p = wave * PIECES + i;
        float* dst = lds + fslot * STAGE_WORDS + p * 8 * WROW;
        const unsigned so = fblk * blk_bytes + (unsigned)(fk * 128) + (unsigned)((p & ~1) * 8 * ld_words * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)vo[p & 1], (int)so, 0, 0);
    };
    auto advance = [&]() {
        fslot = (fslot + 1) & (RING - 1);
        if (++fk == KS) {
            fk = 0;
            fblk = fblk + 13 >= nb ? fblk + 13 - nb : fblk + 13;
        }
    };
    const int sw = (r >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = r * WROW + ((2 * c + h) ^ sw) * 4;
    auto qfrag = [&](int ring_slot, int c, int m) -> f32x4 { return *reinterpret_cast<const f32x4*>(lds + ring_slot * STAGE_WORDS + m * 32 * WROW + coff[c]); };

    f32x16 acc[MT][NP];
    Epi<MT, NP, EPI, EVAR, PRIO> epi;
    epi.run_max = -1e30f; epi.xs = aux[tid & 255] + 1e30f; epi.hits = 0; epi.aux = aux; epi.h = h; epi.wave = wave + blockIdx.x * 8;

    const int lag = (NW == 8 && wave >= 4) ? SK : 0;
#pragma unroll
    for (int g = 0; g < DEPTH; ++g) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) fetch(i);
        advance();
    }
    wait_vm<PIECES>();
    __builtin_amdgcn_s_barrier();
    Rec rec;
    rec.start();
    for (int i = 0; i < lag; ++i) {                   // idle periods of the lagging group: fetch, wait, barrier
#pragma unroll
        for (int p = 0; p < PIECES; ++p) fetch(p);
        advance();
        wait_vm<PIECES>();
        __builtin_amdgcn_s_barrier();
    }
    int slot = 0;
    f32x4 q[MT];
    epi.init(acc, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) q[m] = qfrag(0, 0, m);
    for (int u = 0; u < units; ++u) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool last = ks == KS - 1;
            const int next_slot = (slot + 1) & (RING - 1);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int n = 0; n < NP; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, q[m]), __builtin_bit_cast(f16x8, pf[n][(ks * 4 + c) % PREG]), acc[m][n], 0, 0, 0);
                    if (c < 3) q[m] = qfrag(slot, c + 1, m);
                    else if (!last) q[m] = qfrag(next_slot, 0, m);
                    if (PIECES == 4 && m == 1) fetch(c);
                    if (PIECES == 2 && m == 1 && (c & 1) == 0) fetch(c >> 1);
                    if (PIECES == 1 && m == 1 && c == 0) fetch(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            advance();
            if (SK == 0 && last) epi.run(acc, u);              // lockstep: in front of the barrier (the shipped order)
            wait_vm<PIECES>();
            if (PRIO != 9 || last) __builtin_amdgcn_s_barrier();   // (PRIO = 9: timing experiment, ONE barrier per unit; data meaningless)
            slot = next_slot;
            if (last) {
                if (SK != 0) {                                 // skewed: behind it - the partner wave is in the middle of its unit
                    epi.run(acc, u);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) q[m] = qfrag(slot, 0, m);
                epi.init(acc, u + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    for (int i = 0; i < SK - lag; ++i) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) fetch(p);
        advance();
        wait_vm<PIECES>();
        __builtin_amdgcn_s_barrier();
    }
    if (lane == 0 && (wave == 0 || wave == NW - 1)) rec.stop(clk, blockIdx.x * 2 + (wave != 0));
    __builtin_amdgcn_s_waitcnt(0x0F70);
    float sink = epi.run_max + (float)epi.hits + q[0].x;
#pragma unroll
    for (int m = 0; m < MT; ++m) sink += acc[m][0][5];
    if (sink == 12345.678f) out[0] = sink;
}

// ---- free-running waves: NO s_barrier in the main loop.  Eight waves, each 32 P rows (all slabs in registers) x 64 Q rows per
// unit (MT = 2: 32 accumulator registers - what makes room for the P fragments of D = 512).  A stage is 64 Q rows x 128 elements
// = two 8-KB half-slots ([64 rows][128 B], the swizzle of the shipped engine) = 16 MFMAs per wave; the ring holds RS stages,
// DEPTH of them in flight.  Per ring slot two LDS counters: `filled` (a wave adds 1 when ITS two pieces of the stage have landed -
// its own vmcnt tells it) and `freed` (a wave adds 1 behind its last fragment read of the stage; LDS operations of a wave execute
// in order, so the add cannot overtake the reads).  A wave reads stage s when filled[slot] = 8 (s / RS + 1) and fetches stage
// s + DEPTH when freed[slot'] = 8 ((s + DEPTH) / RS).  A wave in its epilogue holds nobody up until the others are DEPTH - 1
// stages ahead (its pieces are in flight / confirmed that far) or RS - DEPTH stages behind.
template <int KS, int EPI, int EVAR, int RS, int DEPTH>
__global__ void __launch_bounds__(512, 1) freerun(const float* __restrict__ src, int64_t src_rows, int ld_words, int units,
                                                  float* __restrict__ out, unsigned long long* __restrict__ clk) {
    constexpr int MT = 2, NP = 1, PIECES = 2;
    constexpr int HALF_WORDS = 64 * WROW, STAGE_WORDS = 2 * HALF_WORDS;
    constexpr int SPU = KS / 2;                       // stages per unit
    static_assert(KS % 2 == 0 && (RS & (RS - 1)) == 0 && DEPTH < RS, "shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* aux = lds + RS * STAGE_WORDS;
    int* filled = reinterpret_cast<int*>(aux + (EPI + EVAR + 1) * 256);
    int* freed = filled + RS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int64_t nblk = src_rows / 512;
    for (int i = tid; i < (EPI + EVAR + 1) * 256; i += 512) aux[i] = src[i] * 1e-3f;
    if (tid < 2 * RS) filled[tid] = 0;

    f32x4 pf[KS * 4];
    {
        const int64_t pblk = ((int64_t)blockIdx.x * 5 + 3) % nblk;
        const float* prow = src + (pblk * 512 + wave * 32 + r) * (int64_t)ld_words + h * 4;
#pragma unroll
        for (int s = 0; s < KS * 4; ++s) pf[s] = *reinterpret_cast<const f32x4*>(prow + s * 8);
    }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x7fffffff, 0x00020000);
    const int lr = lane >> 3, slot8 = lane & 7;
    // this wave's two pieces of a stage: half-slot wave >> 2 (the stage's even / odd slab), rows 16 (wave & 3) .. + 15
    unsigned vo[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = par * 8 + lr;
        vo[par] = (unsigned)((row * ld_words + (slot8 ^ ((row >> 1) & 7)) * 4) * 4);
    }
    const unsigned nb = (unsigned)nblk, blk_bytes = (unsigned)(512 * ld_words * 4);
    unsigned fblk = (unsigned)(((int64_t)blockIdx.x * 7) % nblk);
    int fk = 0, fs = 0;                               // fetch cursor: stage of the unit, stage number
    auto fetch = [&](int i) {
        float* dst = lds + (fs & (RS - 1)) * STAGE_WORDS + (wave >> 2) * HALF_WORDS + ((wave & 3) * 2 + i) * 8 * WROW;
        const unsigned so = fblk * blk_bytes + (unsigned)((fk * 2 + (wave >> 2)) * 128) + (unsigned)((wave & 3) * 16 * ld_words * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)vo[i], (int)so, 0, 0);
    };
    auto advance = [&]() {
        ++fs;
        if (++fk == SPU) {
            fk = 0;
            fblk = fblk + 13 >= nb ? fblk + 13 - nb : fblk + 13;
        }
    };
    const int sw = (r >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = r * WROW + ((2 * c + h) ^ sw) * 4;
    auto qfrag = [&](int ring_slot, int c8, int m) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(lds + ring_slot * STAGE_WORDS + (c8 >> 2) * HALF_WORDS + m * 32 * WROW + coff[c8 & 3]);
    };
    f32x16 acc[MT][NP];
    Epi<MT, NP, EPI, EVAR> epi;
    epi.run_max = -1e30f; epi.hits = 0; epi.aux = aux; epi.h = h; epi.wave = wave + blockIdx.x * 8;
    __syncthreads();                                  // counters zeroed, aux written
    epi.xs = aux[tid & 255] + 1e30f;
#pragma unroll
    for (int g = 0; g < DEPTH; ++g) {
        fetch(0);
        fetch(1);
        advance();
    }
    wait_vm<(DEPTH - 1) * PIECES>();
    signal_add(filled + 0, lane);                     // my pieces of stage 0 are in LDS
    Rec rec;
    rec.start();
    int s = 0;                                        // stage being multiplied
    epi.init(acc, 0);
    for (int u = 0; u < units; ++u) {
#pragma unroll
        for (int st = 0; st < SPU; ++st) {
            const int slot = s & (RS - 1);
            poll_ge(freed + (fs & (RS - 1)), 8 * (fs / RS));            // the slot stage s + DEPTH goes into: its last occupant read by all
            poll_ge(filled + slot, 8 * (s / RS + 1));                    // stage s complete
            f32x4 q[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) q[m] = qfrag(slot, 0, m);
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, q[m]), __builtin_bit_cast(f16x8, pf[st * 8 + c8]), acc[m][0], 0, 0, 0);
                    if (c8 < 7) q[m] = qfrag(slot, c8 + 1, m);
                    if (m == 1 && (c8 == 0 || c8 == 4)) fetch(c8 >> 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            advance();
            signal_add(freed + slot, lane);           // behind my last read of the stage (LDS operations of a wave execute in order)
            wait_vm<(DEPTH - 1) * PIECES>();           // my pieces of stage s + 1 have landed
            signal_add(filled + ((s + 1) & (RS - 1)), lane);
            ++s;
            if (st == SPU - 1) {
                epi.run(acc, u);
                __builtin_amdgcn_sched_barrier(0);
                epi.init(acc, u + 1);
            }
        }
    }
    if (lane == 0 && (wave == 0 || wave == 7)) rec.stop(clk, blockIdx.x * 2 + (wave != 0));
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();                                  // (nobody leaves while a DMA is in flight into the workgroup's LDS)
    float sink = epi.run_max + (float)epi.hits;
#pragma unroll
    for (int m = 0; m < MT; ++m) sink += acc[m][0][5];
    if (sink == 12345.678f) out[0] = sink;
}

// ---------------------------------------------------------------------------------------------------------------- host
static const float* g_src;
static int64_t g_rows;
static float* g_out;
static unsigned long long* g_clk;
constexpr int ROUNDS = 8;                             // work items per resident slot
constexpr long TOTAL_MFMA = 1L << 21;                 // MFMAs per CU in every structure (8192 units of 256)

static void report(const char* name, const char* shape, size_t lds_bytes, float ms, int blocks, long mfma_per_block) {
    static unsigned long long h[4 * 2 * 256 * 8 * 4];
    (void)hipMemcpy(h, g_clk, sizeof(unsigned long long) * 4 * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, ghz = 0, dmin = 1e30, dmax = 0, dsum = 0;
    unsigned long long first = ~0ull, last = 0;
    bool cu_seen[8 * 65536 / 8] = {false};
    int cus = 0;
    double xcd_us[8] = {0};
    int xcd_n[8] = {0};
    for (int b = 0; b < blocks; ++b) {
        const unsigned long long* r = h + (size_t)b * 8;           // record of wave 0
        const double us = (double)(r[2] - r[1]) * 0.01;
        cyc += (double)r[0];
        ghz += (double)r[0] / ((double)(r[2] - r[1]) * 10.0);
        dmin = us < dmin ? us : dmin; dmax = us > dmax ? us : dmax; dsum += us;
        first = r[1] < first ? r[1] : first; last = r[2] > last ? r[2] : last;
        const unsigned where = (unsigned)r[3], xcc = (where >> 16) & 7, cu = ((where >> 8) & 15) | (((where >> 12) & 1) << 4) | (((where >> 13) & 7) << 5);
        if (!cu_seen[xcc * 256 + cu]) { cu_seen[xcc * 256 + cu] = true; ++cus; }
        xcd_us[xcc] += us; ++xcd_n[xcc];
    }
    const double units256 = (double)TOTAL_MFMA / 256;
    printf("%-9s %-34s lds %6zu: %.3f us per 256 MFMA/CU (%4.0f TF) | per item: %.0f cyc per 256 MFMA of its share, %.2f GHz, %.0f/%.0f/%.0f us min/mean/max | span %.2f ms, %d CUs | XCD mean us:", name, shape,
           lds_bytes, ms * 1e3 / units256, 2.0 * 256 * 256 * 64 * 256 / (ms * 1e3 / units256) * 1e-6, cyc / blocks / ((double)mfma_per_block / 256), ghz / blocks, dmin, dsum / blocks, dmax,
           (double)(last - first) * 1e-5, cus);
    for (int x = 0; x < 8; ++x) printf(" %.0f", xcd_n[x] ? xcd_us[x] / xcd_n[x] : 0.0);
    printf("  [%s]\n", hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

template <class K>
static void time_kernel(K k, const char* name, const char* shape, size_t lds_bytes, int threads, int wgs, int units_per_cu_slot, long mfma_per_unit_block) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
        printf("%-9s %-34s: LDS %zu B not available\n", name, shape, lds_bytes);
        (void)hipGetLastError();
        return;
    }
    const int blocks = 256 * wgs * ROUNDS, units = units_per_cu_slot / ROUNDS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_bytes, 0, g_src, g_rows, 256, 8, g_out, g_clk);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_bytes, 0, g_src, g_rows, 256, units, g_out, g_clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    report(name, shape, lds_bytes, best, blocks, (long)units * mfma_per_unit_block);
}

template <int NW, int QR, int NP, int KS, int EPI, int EVAR, int SK, int WGS, int PRIO = 0>
static void run(const char* name) {
    constexpr int MT = QR / 32;
    constexpr int RING = SK == 0 ? 4 : (3 + SK + 1 <= 4 ? 4 : (3 + SK + 1 <= 8 ? 8 : 16));
    const size_t lds_bytes = (size_t)(RING * QR * WROW + (EPI + EVAR + 1) * 256) * 4;
    const long per_unit = (long)NW * KS * 4 * MT * NP;                     // MFMAs of a workgroup per unit
    char shape[96];
    snprintf(shape, sizeof shape, "NW %d QR %3d NP %d KS %d EPI %d+%d SK %d WGS %d prio %d", NW, QR, NP, KS, EPI, EVAR, SK, WGS, PRIO);
    time_kernel(stage<NW, QR, NP, KS, EPI, EVAR, SK, WGS, PRIO>, name, shape, lds_bytes, NW * 64, WGS, (int)(TOTAL_MFMA / (WGS * per_unit)), per_unit);
}

template <int KS, int EPI, int EVAR, int RS, int DEPTH>
static void run_free(const char* name) {
    const size_t lds_bytes = (size_t)(RS * 2 * 64 * WROW + (EPI + EVAR + 1) * 256 + 2 * RS) * 4;
    const long per_unit = 8L * KS * 4 * 2;
    char shape[96];
    snprintf(shape, sizeof shape, "NW 8 QR  64 NP 1 KS %d EPI %d+%d ring %d depth %d", KS, EPI, EVAR, RS, DEPTH);
    time_kernel(freerun<KS, EPI, EVAR, RS, DEPTH>, name, shape, lds_bytes, 512, 1, (int)(TOTAL_MFMA / per_unit), per_unit);
}

template <int VPER>
static void run_ovl(bool part4 = false) {
    const int iters = 200000;
    static unsigned long long h[256 * 2 * 4];
    for (int mode : part4 ? std::initializer_list<int>{1, 1 + 64, 1 + 128, 2, 3, 3 + 64, 3 + 128, 3 + 8, 3 + 8 + 64} : std::initializer_list<int>{1, 2, 3, 3 + 16, 3 + 32, 3 + 48, 3 + 8, 3 + 8 + 16, 4}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(ovl<VPER>, dim3(256), dim3(512), 0, 0, mode, 1000, g_src, g_out, g_clk);
        hipEventRecord(e0);
        hipLaunchKernelGGL(ovl<VPER>, dim3(256), dim3(512), 0, 0, mode, iters, g_src, g_out, g_clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, g_clk, sizeof h, hipMemcpyDeviceToHost);
        // s_memrealtime ticks at 100 MHz.  Per wave class (0: waves 0-3, 1: waves 4-7): cycles per iteration, clock, duration over
        // the 256 blocks; span = first start to last end over all blocks
        printf("ovl VPER %2d mode %d (%s): kernel %.2f ms", VPER, mode,
               mode == 1 ? "waves 0-3 MFMA x4, 4-7 idle" : mode == 2 ? "waves 4-7 VALU, 0-3 idle" : (mode & 7) == 3 ? ((mode & 8) ? "4-7 MFMA beside 0-3 VALU (vector waves OLDER)" : "0-3 MFMA beside 4-7 VALU") : "all 8: 4 MFMA + VPER VALU interleaved", ms);
        if ((mode & 7) == 3) printf(" prio %d", (mode >> 4) & 3);
        if (mode & 64) printf(" [ONE accumulator: dependent MFMA chain]");
        if (mode & 128) printf(" [two accumulators]");
        unsigned long long first = ~0ull, last = 0;
        for (int cls = 0; cls < 2; ++cls) {
            double cyc = 0, dmin = 1e30, dmax = 0, dsum = 0, ghz = 0;
            for (int b = 0; b < 256; ++b) {
                const unsigned long long* r = h + ((size_t)b * 2 + cls) * 4;
                const double d = (double)(r[2] - r[1]) * 1e-5;      // ms
                cyc += (double)r[0] / iters; dsum += d; dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax;
                ghz += r[2] > r[1] ? (double)r[0] / ((double)(r[2] - r[1]) * 10.0) : 0.0;
                first = r[1] < first ? r[1] : first; last = r[2] > last ? r[2] : last;
            }
            printf(" | %s: %.1f cyc/iter %.2f GHz %.2f/%.2f/%.2f ms", cls == 0 ? "MFMA side" : "vector side", cyc / 256, ghz / 256, dmin, dsum / 256, dmax);
        }
        printf(" | span %.2f ms\n", (double)(last - first) * 1e-5);
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const int src_mb = 200;
    g_rows = (int64_t)src_mb * 1024;
    float* src;
    (void)hipMalloc(&src, g_rows * 1024);
    (void)hipMalloc(&g_out, 4);
    (void)hipMalloc(&g_clk, sizeof(unsigned long long) * 4 * 2 * 256 * 8 * 4);
    {
        uint16_t* hsrc = (uint16_t*)malloc(g_rows * 1024);
        uint32_t x = 12345u;
        for (int64_t i = 0; i < g_rows * 512; ++i) {
            x = x * 1664525u + 1013904223u;
            hsrc[i] = (uint16_t)(((x >> 16) & 0x83ffu) | 0x3400u | ((x >> 8) & 0x0400u));
        }
        (void)hipMemcpy(src, hsrc, g_rows * 1024, hipMemcpyHostToDevice);
        free(hsrc);
    }
    g_src = src;
    const int part = argc > 1 ? atoi(argv[1]) : 3;
    if (part & 1) {
        run_ovl<16>();
        run_ovl<64>();
    }
    if (part & 8) {                                   // one barrier per unit instead of one per stage (timing only)
        run<4, 64, 2, 2, 0, 0, 0, 2>("2wg-np2q64");
        run<4, 64, 2, 2, 0, 0, 0, 2, 9>("2wg-np2q64-bar1");
        run<4, 64, 2, 2, 4, 0, 0, 2>("2wg-np2q64");
        run<4, 64, 2, 2, 4, 0, 0, 2, 9>("2wg-np2q64-bar1");
        run<4, 64, 2, 2, 6, 0, 0, 2>("2wg-np2q64");
        run<4, 64, 2, 2, 6, 0, 0, 2, 9>("2wg-np2q64-bar1");
        run<8, 128, 1, 8, 4, 0, 0, 1>("lockstep");
        run<8, 128, 1, 8, 4, 0, 0, 1, 9>("lockstep-bar1");
    }
    if (part & 4) {                                   // dependent MFMA chains beside the partner's vector work
        run_ovl<16>(true);
        run_ovl<64>(true);
    }
    if (part & 2) {
        // ---- D = 512 (KS = 8): lockstep, two workgroups (with / without priority in the epilogue), variable epilogues
        run<8, 128, 1, 8, 0, 0, 0, 1>("lockstep");
        run<8, 128, 1, 8, 4, 0, 0, 1>("lockstep");
        run<8, 128, 1, 8, 6, 0, 0, 1>("lockstep");
        run<8, 128, 1, 8, 2, 4, 0, 1>("lockstep");
        run<4, 128, 1, 8, 4, 0, 0, 2>("2wg");
        run<4, 128, 1, 8, 4, 0, 0, 2, 1>("2wg");
        run<4, 128, 1, 8, 4, 0, 0, 2, 3>("2wg");
        run<4, 128, 1, 8, 2, 4, 0, 2>("2wg");
        run<4, 128, 1, 8, 2, 4, 0, 2, 1>("2wg");
        // ---- D = 128 (KS = 2)
        run<8, 128, 1, 2, 0, 0, 0, 1>("lockstep");
        run<8, 128, 1, 2, 4, 0, 0, 1>("lockstep");
        run<8, 128, 1, 2, 6, 0, 0, 1>("lockstep");
        run<8, 128, 1, 2, 2, 4, 0, 1>("lockstep");
        run<8, 128, 1, 2, 4, 4, 0, 1>("lockstep");
        run<4, 128, 1, 2, 4, 0, 0, 2, 1>("2wg");
        run<4, 128, 2, 2, 0, 0, 0, 2>("2wg-np2");
        run<4, 128, 2, 2, 4, 0, 0, 2>("2wg-np2");
        run<4, 128, 2, 2, 6, 0, 0, 2>("2wg-np2");
        run<4, 128, 2, 2, 4, 0, 0, 2, 1>("2wg-np2");
        run<4, 128, 2, 2, 6, 0, 0, 2, 1>("2wg-np2");
        run<4, 128, 2, 2, 2, 4, 0, 2>("2wg-np2");
        run<4, 128, 2, 2, 4, 4, 0, 2>("2wg-np2");
        run<4, 64, 2, 2, 0, 0, 0, 2>("2wg-np2q64");
        run<4, 64, 2, 2, 4, 0, 0, 2>("2wg-np2q64");
        run<4, 64, 2, 2, 6, 0, 0, 2>("2wg-np2q64");
        run<4, 64, 2, 2, 6, 0, 0, 2, 1>("2wg-np2q64");
        run<4, 64, 2, 2, 2, 4, 0, 2>("2wg-np2q64");
        run<4, 64, 2, 2, 4, 4, 0, 2>("2wg-np2q64");
        run<4, 64, 2, 2, 4, 0, 0, 3>("3wg-np2q64");
        run<4, 64, 2, 2, 6, 0, 0, 3>("3wg-np2q64");
        run<4, 64, 2, 2, 4, 4, 0, 3>("3wg-np2q64");
        run<8, 64, 2, 2, 4, 0, 0, 1>("1wg8-np2q64");
        run<8, 64, 2, 2, 6, 0, 0, 1>("1wg8-np2q64");
        // ---- D = 192 / 256 (KS = 3, 4)
        run<8, 128, 1, 4, 4, 0, 0, 1>("lockstep");
        run<8, 128, 1, 4, 6, 0, 0, 1>("lockstep");
        run<4, 64, 2, 4, 4, 0, 0, 2>("2wg-np2q64");
        run<4, 64, 2, 4, 6, 0, 0, 2>("2wg-np2q64");
        run<8, 128, 1, 3, 6, 0, 0, 1>("lockstep");
        run<4, 64, 2, 3, 6, 0, 0, 2>("2wg-np2q64");
        // ---- D = 64 (KS = 1)
        run<8, 128, 1, 1, 6, 0, 0, 1>("lockstep");
        run<4, 64, 2, 1, 6, 0, 0, 2>("2wg-np2q64");
        run<4, 64, 2, 1, 6, 0, 0, 3>("3wg-np2q64");
    }
    return 0;
}
