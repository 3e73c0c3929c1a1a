// Microbenchmark of an OPERAND-STATIONARY stage for the f16 filter kernels: a workgroup's P block
// (256 rows x D = 512 f16) is loaded ONCE into registers as MFMA B fragments and never moves again; only 128-row Q slabs
// (128 rows x 64 f16 = 16 KB per stage) stream through an LDS ring by LDS-DMA.  Per 8.4 MFLOP of a CU (what one stage of
// csrc/wide_engine.h multiplies) the load path moves 32 KB instead of 64 KB and a wave reads Q fragments only.
//   NW = 4: four waves (one per SIMD, 512 registers), each 64 P rows (256 fragment registers) x 128 Q rows (128 accumulators)
//   NW = 8: eight waves (two per SIMD, 256 registers), each 32 P rows (128 fragment registers) x 128 Q rows (64 accumulators)
//   LAG = 1: the barrier at the end of stage g publishes stage g+1 (first fragment reads of a stage wait behind the barrier)
//   LAG = 2: it publishes stage g+2, so the first fragments of stage g+1 are read under the last MFMAs of stage g
// Data are meaningless; only the instruction mix, the LDS traffic and the fill traffic count.  Same source patterns and the
// same unit (us per 8.4 MFLOP per CU) as stage_sched.hip, so the numbers compare directly (adopted schedule there:
// 1.73-1.85 us L2-resident, 2.00-2.17 us from a 200 MB source).
// Build: hipcc --offload-arch=gfx950 -O3 pstat.hip -o pstat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int WROW = 32;                 // LDS row: 32 words = 128 B = 64 f16
constexpr int QROWS = 128;               // Q rows per stage
constexpr int STAGE_WORDS = QROWS * WROW;   // 16 KB
constexpr int KSTEPS = 32;               // D = 512 f16 = 32 MFMA k-steps of 16

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N < 64, "vmcnt");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// ABL (ablation bits, timing only - results are meaningless with any of 1 / 2 / 4 set): 1 no LDS-DMA, 2 no fragment reads behind
// the first, 4 no barrier, 8 waves 4-7 issue their pieces one k-step behind waves 0-3, 16 s_setprio 1 around the MFMAs,
// 32 a barrier every SECOND stage only
template <int NW, int LAG, int ROT, int ABL = 0>
__global__ void __launch_bounds__(NW * 64, 1) pstat(const float* __restrict__ src, int64_t src_rows, int ld_words, int tiles,
                                                      float* __restrict__ out) {
    constexpr int NP = 8 / NW;                        // 32-row P tiles per wave: 2 (four waves) or 1 (eight waves)
    constexpr int PIECES = 16 / NW;                   // 1 KB LDS-DMA pieces per wave and stage
    constexpr int DEPTH = LAG + 1;                    // stages in flight ahead of the one being multiplied
    constexpr int RING = 4;
    static_assert(DEPTH < RING, "ring");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int64_t nblk = src_rows / 512;

    // ---- the stationary operand: NP x 32 fragments of 16 B per lane
    f32x4 pf[NP][KSTEPS];
    {
        const int64_t pblk = ((int64_t)blockIdx.x * 5 + 3) % nblk;
        const float* prow = src + (pblk * 512 + wave * (32 * NP) + r) * (int64_t)ld_words + h * 4;
#pragma unroll
        for (int n = 0; n < NP; ++n)
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) pf[n][s] = *reinterpret_cast<const f32x4*>(prow + (int64_t)n * 32 * ld_words + s * 8);
    }

    // ---- Q slabs by LDS-DMA: piece p = 8 rows x 128 B; this wave's pieces are p = wave * PIECES + i
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x7fffffff, 0x00020000);
    const int lr = lane >> 3, slot = lane & 7;
    unsigned vo[2];                                    // even / odd piece (the swizzle of a row depends on (row >> 1) & 7)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = par * 8 + lr;
        vo[par] = (unsigned)((row * ld_words + (slot ^ ((row >> 1) & 7)) * 4) * 4);
    }
    // the 512-row source block a Q tile reads: a 32-bit walk (a 64-bit modulo per stage would cost more SALU than the stage)
    const unsigned nb = (unsigned)nblk, blk_bytes = (unsigned)(512 * ld_words * 4);
    unsigned blk_cur = (unsigned)(((int64_t)blockIdx.x * 7) % nblk), blk_next = blk_cur;
    auto piece = [&](int ring_slot, bool next_tile, int kslab, int i) {
        if (ABL & 1) return;
        const int p = wave * PIECES + i;
        float* dst = lds + ring_slot * STAGE_WORDS + p * 8 * WROW;
        const unsigned so = (next_tile ? blk_next : blk_cur) * blk_bytes + (unsigned)(kslab * 128) + (unsigned)((p & ~1) * 8 * ld_words * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)vo[p & 1], (int)so, 0, 0);
    };

    const int sw = (r >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = ((2 * c + h) ^ sw) * 4;
    const int qrow = r * WROW;
    auto qfrags = [&](f32x4 (&q)[4], int ring_slot, int c) {
        const float* st = lds + ring_slot * STAGE_WORDS;
#pragma unroll
        for (int m = 0; m < 4; ++m) q[m] = *reinterpret_cast<const f32x4*>(st + qrow + m * 32 * WROW + coff[c]);
    };

    f32x16 acc[4][NP];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NP; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int G = tiles * 8;
    // prologue: DEPTH stages in flight, the first LAG of them landed and published
#pragma unroll
    for (int g = 0; g < DEPTH; ++g)
#pragma unroll
        for (int i = 0; i < PIECES; ++i) piece(g % RING, false, g, i);
    wait_vm<(DEPTH - LAG) * PIECES>();
    __builtin_amdgcn_s_barrier();

    f32x4 qa[4], qb[4];
    qfrags(qa, 0, 0);
    auto qfrag1 = [&](int ring_slot, int c, int m) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(lds + ring_slot * STAGE_WORDS + qrow + m * 32 * WROW + coff[c]);
    };
    for (int t = 0; t < tiles; ++t) {
        blk_cur = blk_next;
        blk_next = blk_cur + 13 >= nb ? blk_cur + 13 - nb : blk_cur + 13;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            if (ABL & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (ROT) {
                    // ONE fragment set: the fragment of row tile m for the next k-step is read right behind the MFMAs that used it
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
#pragma unroll
                        for (int n = 0; n < NP; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, qa[m]),
                                                                                 __builtin_bit_cast(f16x8, pf[n][s8 * 4 + c]), acc[m][n], 0, 0, 0);
                        if (!(ABL & 2)) {
                            if (c < 3) qa[m] = qfrag1(s8 % RING, c + 1, m);
                            else if (LAG == 2) qa[m] = qfrag1((s8 + 1) % RING, 0, m);
                        }
                        if (PIECES == 4 && m == 1) piece((s8 + DEPTH) % RING, s8 + DEPTH >= 8, (s8 + DEPTH) & 7, c);
                        if (PIECES == 2 && !(ABL & 8) && m == 1 && (c & 1) == 0) piece((s8 + DEPTH) % RING, s8 + DEPTH >= 8, (s8 + DEPTH) & 7, c >> 1);
                        if (PIECES == 2 && (ABL & 8) && m == 1) {
                            if (wave < 4 ? (c & 1) == 0 : (c & 1) == 1) piece((s8 + DEPTH) % RING, s8 + DEPTH >= 8, (s8 + DEPTH) & 7, c >> 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    continue;
                }
                f32x4 (&cur)[4] = (c & 1) ? qb : qa;
                f32x4 (&nxt)[4] = (c & 1) ? qa : qb;
                if (c < 3) qfrags(nxt, s8 % RING, c + 1);
                else if (LAG == 2) qfrags(nxt, (s8 + 1) % RING, 0);          // published by the barrier of stage g - 1
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
#pragma unroll
                    for (int n = 0; n < NP; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur[m]),
                                                                             __builtin_bit_cast(f16x8, pf[n][s8 * 4 + c]), acc[m][n], 0, 0, 0);
                    if (PIECES == 4 && m == 1) piece((s8 + DEPTH) % RING, s8 + DEPTH >= 8, (s8 + DEPTH) & 7, c);
                    if (PIECES == 2 && m == 1 && (c & 1) == 0) piece((s8 + DEPTH) % RING, s8 + DEPTH >= 8, (s8 + DEPTH) & 7, c >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (ABL & 16) __builtin_amdgcn_s_setprio(0);
            // stage g + LAG has landed (this wave's pieces); the barrier publishes it
            wait_vm<(DEPTH - LAG) * PIECES>();
            if (!(ABL & 4) && (!(ABL & 32) || (s8 & 1))) __builtin_amdgcn_s_barrier();
            if (LAG == 1) {
                qfrags(qa, (s8 + 1) % RING, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (G < 0) break;
    }
    float sink = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NP; ++b) sink += acc[a][b][5];
    sink += qa[0].x;
    if (!ROT) sink += qb[0].x;
    if (sink == 12345.678f) out[0] = sink;
}

template <int NW, int LAG, int ROT, int ABL = 0>
static void run(const float* src, int64_t rows, int ld_words, float* out, int src_mb) {
    const int tiles = 1024;                          // 8192 stages of 16 KB = 4096 units of 8.4 MFLOP per CU
    const size_t lds_bytes = 4 * STAGE_WORDS * 4;
    hipFuncSetAttribute((const void*)pstat<NW, LAG, ROT, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((pstat<NW, LAG, ROT, ABL>), dim3(256), dim3(NW * 64), lds_bytes, 0, src, rows, ld_words, 16, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pstat<NW, LAG, ROT, ABL>), dim3(256), dim3(NW * 64), lds_bytes, 0, src, rows, ld_words, tiles, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / (tiles * 4);        // per 8.4 MFLOP per CU (two 16-KB stages)
    printf("src %3d MB  waves %d lag %d rot %d abl %2d: %.3f us per 8.4 MFLOP/CU -> %.0f TFLOP/s of 2500 (%.2f)   [%s]\n", src_mb, NW, LAG, ROT, ABL, us,
           2.0 * 256 * 256 * 64 * 256 / us * 1e-6, 2.0 * 256 * 256 * 64 * 256 / us * 1e-6 / 2500.0, hipGetErrorString(hipGetLastError()));
}

int main() {
    for (int src_mb : {16, 200}) {
        const int ld_words = 256;
        const int64_t rows = (int64_t)src_mb * 1024;
        float *src, *out;
        (void)hipMalloc(&src, rows * 1024);
        (void)hipMalloc(&out, 4);
        // random f16 bit patterns in a sane exponent range (the clock follows the operand bits: never bench on constants)
        {
            uint16_t* hsrc = (uint16_t*)malloc(rows * 1024);
            uint32_t x = 12345u;
            for (int64_t i = 0; i < rows * 512; ++i) {
                x = x * 1664525u + 1013904223u;
                hsrc[i] = (uint16_t)(((x >> 16) & 0x83ffu) | 0x3400u | ((x >> 8) & 0x0400u));
            }
            (void)hipMemcpy(src, hsrc, rows * 1024, hipMemcpyHostToDevice);
            free(hsrc);
        }
        run<4, 2, 1>(src, rows, ld_words, out, src_mb);
        run<8, 1, 0>(src, rows, ld_words, out, src_mb);
        run<8, 2, 0>(src, rows, ld_words, out, src_mb);
        run<8, 2, 1>(src, rows, ld_words, out, src_mb);
        run<8, 2, 1, 1>(src, rows, ld_words, out, src_mb);
        run<8, 2, 1, 2>(src, rows, ld_words, out, src_mb);
        run<8, 2, 1, 3>(src, rows, ld_words, out, src_mb);
        run<8, 2, 1, 4>(src, rows, ld_words, out, src_mb);
        run<8, 2, 1, 7>(src, rows, ld_words, out, src_mb);
        run<8, 2, 1, 8>(src, rows, ld_words, out, src_mb);
        run<8, 2, 1, 16>(src, rows, ld_words, out, src_mb);
        run<8, 2, 1, 24>(src, rows, ld_words, out, src_mb);
        run<8, 2, 1, 32>(src, rows, ld_words, out, src_mb);
        (void)hipFree(src); (void)hipFree(out);
    }
    return 0;
}
