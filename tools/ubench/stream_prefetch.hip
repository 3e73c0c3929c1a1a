// Microbenchmark of the wide engine's memory behaviour with the REAL sharing pattern of cross_wide_kernel:
// per XCD 32 workgroups = 8 P blocks (256 rows x 1 KB, re-read every tile) x 4 Q streams (a new 256-row tile every 8
// stages, shared by the 8 workgroups of the stream), operands 2 x 100 MB (beyond L2, as the f16 copies of 2 x 100k x 512).
//   schedule: V2 of stage_sched.hip (DMA pieces between the MFMAs, last chunk carried across the barrier)
//   pf = 0   no prefetch: the first workgroup to touch a Q line takes the L2 miss inside its pipeline
//   pf = L   every workgroup touches its 1/8 slice of the Q slab L stages ahead with scalar loads (s_load_dword, one
//            per 128-B line: lgkmcnt, not vmcnt - the vector pipeline never waits for them)
// Build: hipcc --offload-arch=gfx950 -O3 stream_prefetch.hip -o stream_prefetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int WROW = 32, WTB = 256, TILE_WORDS = WTB * WROW, STAGE_WORDS = 2 * TILE_WORDS;
constexpr int LDW = 256;                                   // row stride in words (1 KB rows)
struct Frags { f32x4 q[4], p[2]; };

template <int SCHED>
__global__ void __launch_bounds__(512, 1) stream(const float* __restrict__ Pm, const float* __restrict__ Qm, int64_t nblk, int stages,
                                                 int pf, int tiles_per_stream, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, chunk = (tid & 7) ^ ((srow >> 1) & 7);
    const int sw = (r >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = ((2 * c + h) ^ sw) * 4;
    const int qrow = (wm * 128 + r) * WROW, prow = TILE_WORDS + (wn * 64 + r) * WROW;
    auto frags = [&](const float* st, int c) {
        Frags f;
#pragma unroll
        for (int m = 0; m < 4; ++m) f.q[m] = *reinterpret_cast<const f32x4*>(st + qrow + m * 32 * WROW + coff[c]);
#pragma unroll
        for (int n = 0; n < 2; ++n) f.p[n] = *reinterpret_cast<const f32x4*>(st + prow + n * 32 * WROW + coff[c]);
        return f;
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    auto mma = [&](const Frags& f, int m, int n) {
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.q[m]), __builtin_bit_cast(f16x8, f.p[n]), acc[m][n], 0, 0, 0);
    };
    // work mapping of cross_wide_kernel
    const int xcd = blockIdx.x & 7, within = (blockIdx.x >> 3) & 31;
    const int pslot = within >> 2, qstream = within & 3;
    const int64_t pblk = (xcd * 8 + pslot) % nblk;
    const int64_t q0 = ((int64_t)(xcd * 4 + qstream) * tiles_per_stream) % nblk;
    const float* pbase = Pm + pblk * WTB * (int64_t)LDW;
    auto qbase = [&](int tile) { return Qm + ((q0 + tile) % nblk) * WTB * (int64_t)LDW; };
    __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)pbase, 0, WTB * LDW * 4, 0x00020000), qrs;
    unsigned vo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vo[j] = (unsigned)(((j * 64 + srow) * LDW + chunk * 4) * 4);
    int ftile = -1;
    auto piece = [&](int g, int j) {                       // j < 4: Q rows j*64.., j >= 4: P rows
        float* st = lds + (g & 1) * STAGE_WORDS;
        const int so = (g & 7) * 128;
        if (j < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (__attribute__((address_space(3))) void*)(st + (j * 64 + wave * 8) * 32), 16, (int)vo[j], so, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (__attribute__((address_space(3))) void*)(st + TILE_WORDS + ((j - 4) * 64 + wave * 8) * 32), 16, (int)vo[j - 4], so, 0, 0);
    };
    auto set_q = [&](int g) {
        if ((g >> 3) != ftile) {
            ftile = g >> 3;
            qrs = __builtin_amdgcn_make_buffer_rsrc((void*)qbase(ftile), 0, WTB * LDW * 4, 0x00020000);
        }
    };
    unsigned sink_s = 0;
    // this workgroup's slice of the Q slab of stage g: rows pslot*32 .. +31, one 128-B line each; wave w touches 4 lines
    auto prefetch = [&](int g) {
        const float* base = qbase(g >> 3) + (int64_t)(pslot * 32 + wave * 4) * LDW + (g & 7) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned tmp;
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(tmp) : "s"(base + (int64_t)i * LDW) : "memory");
            sink_s ^= tmp;                                  // keeps the destination allocated; never waited for explicitly
        }
    };
    auto mm_dma = [&](const Frags& f, int g, int j0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            mma(f, m, 0);
            mma(f, m, 1);
            piece(g, j0 + m);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto mm = [&](const Frags& f) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { mma(f, m, 0); mma(f, m, 1); }
    };
    set_q(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) piece(0, j);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    Frags f0, f1;
    f1 = frags(lds, 3);
    for (int g = 0; g < stages; ++g) {
        const float* st = lds + (g & 1) * STAGE_WORDS;
        set_q(g + 1);
        if (pf > 0) prefetch(g + pf);
        if (SCHED == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) piece(g + 1, j);
            f0 = frags(st, 0);
            f1 = frags(st, 1);
            mm(f0);
            f0 = frags(st, 2);
            mm(f1);
            f1 = frags(st, 3);
            mm(f0);
            mm(f1);
        } else {
            f0 = frags(st, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm_dma(f1, g + 1, 0);
            f1 = frags(st, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm_dma(f0, g + 1, 4);
            f0 = frags(st, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(f1);
            f1 = frags(st, 3);
            __builtin_amdgcn_sched_barrier(0);
            mm(f0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
    float sink = (float)sink_s;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) sink += acc[a][b][5];
    sink += f1.q[0].x;
    if (sink == 12345.678f) out[0] = sink;
}

template <int SCHED>
static void run(const float* P, const float* Q, int64_t nblk, int pf, float* out) {
    const int stages = 8 * 64;                             // 64 tiles per workgroup
    const size_t lds_bytes = 2 * STAGE_WORDS * 4;
    (void)hipFuncSetAttribute((const void*)stream<SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(stream<SCHED>, dim3(256), dim3(512), lds_bytes, 0, P, Q, nblk, stages, pf, 12, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double us = best * 1e3 / stages;
    printf("schedule %d prefetch lead %d stages: %.3f us per stage -> %.0f TFLOP/s of 2500 (%.2f)\n", SCHED, pf, us,
           2.0 * 256 * 256 * 64 * 256 / us * 1e-6, 2.0 * 256 * 256 * 64 * 256 / us * 1e-6 / 2500.0);
}

int main() {
    const int64_t nblk = 390;                              // 390 blocks x 256 KB = 100 MB per operand
    float *P, *Q, *out;
    (void)hipMalloc(&P, nblk * WTB * LDW * 4);
    (void)hipMalloc(&Q, nblk * WTB * LDW * 4);
    (void)hipMalloc(&out, 4);
    {   // random f16 operands (values in [-2, 2)): constant data would let the chip clock ~15 % higher than real inputs do
        const size_t words = (size_t)nblk * WTB * LDW;
        unsigned* h = (unsigned*)malloc(words * 4);
        unsigned long long s = 88172645463325252ull;
        for (int pass = 0; pass < 2; ++pass) {
            for (size_t i = 0; i < words; ++i) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                const unsigned lo = 0x3800u + (unsigned)(s & 0x7ff) + ((unsigned)(s >> 20) & 0x8000u);
                const unsigned hi = 0x3800u + (unsigned)((s >> 32) & 0x7ff) + ((unsigned)(s >> 50) & 0x8000u);
                h[i] = lo | (hi << 16);
            }
            (void)hipMemcpy(pass ? (void*)Q : (void*)P, h, words * 4, hipMemcpyHostToDevice);
        }
        free(h);
    }
    run<0>(P, Q, nblk, 0, out);
    run<1>(P, Q, nblk, 0, out);
    run<1>(P, Q, nblk, 8, out);
    return 0;
}
