// Microbenchmark: producer / consumer wave specialisation for the f16 filter engine.
//   workgroup = 8 waves: waves 0-3 ONLY issue LDS-DMA (a stage = 64 f16 of 128 Q rows + 256 P rows = 48 KB, three stages in
//   LDS, counted vmcnt: the stage issued in interval i is waited for at the end of interval i+1), waves 4-7 ONLY multiply
//   (each 128 Q rows x 64 P rows = 4 x 2 MFMA tiles, 32 MFMAs per stage) - one loader and one consumer per SIMD.
//   The DMA instructions block their wave until the texture path has taken them; here that wave has nothing else to do.
// Tile per workgroup: 128 x 256 (half of the production engine's 256 x 256: 48 KB of fill per 4.2 MFLOP instead of 64 KB per 8.4).
// Build: hipcc --offload-arch=gfx950 -O3 loader.hip -o loader
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int WROW = 32;                                  // LDS row: 32 words = 64 f16
constexpr int QROWS = 128, PROWS = 256;
constexpr int STAGE_WORDS = (QROWS + PROWS) * WROW;       // 48 KB
#ifndef RING
#define RING 3
#endif

__global__ void __launch_bounds__(512, 1) run(const float* __restrict__ X, int64_t n_rows, int ld /* words */, int nk, int ntiles,
                                             float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nq_tiles = n_rows / QROWS, np_blocks = n_rows / PROWS;
    const int64_t pb = (blockIdx.x * 5) % np_blocks;
    const int G = ntiles * nk;
    if (wave < 4) {
        // ---- loader: 4 Q instructions (8 rows each: rows 32 j + 8 wave) + 8 P instructions per stage
        const int srow = wave * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((srow >> 1) & 7);
        const unsigned vo = (unsigned)((srow * ld + chunk * 4) * 4);
        const unsigned grp = (unsigned)(32 * ld * 4);
        __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + pb * PROWS * (int64_t)ld), 0, PROWS * ld * 4, 0x00020000);
        auto issue = [&](int g) {
            const int t = g / nk, kt = g % nk;
            const int64_t qt = (blockIdx.x * 3 + t) % nq_tiles;
            __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + qt * QROWS * (int64_t)ld), 0, g < G ? QROWS * ld * 4 : 0, 0x00020000);
            float* s = lds + (g % RING) * STAGE_WORDS + wave * 8 * WROW;
            const unsigned so = (unsigned)(kt * WROW * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (__attribute__((address_space(3))) void*)(s + j * 32 * WROW), 16, (int)vo, (int)(so + j * grp), 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (__attribute__((address_space(3))) void*)(s + QROWS * WROW + j * 32 * WROW), 16, (int)vo, (int)(so + j * grp), 0, 0);
        };
        for (int g = 0; g < RING - 1; ++g) issue(g);
        __builtin_amdgcn_s_waitcnt(0x0F70 | (12 * (RING - 2) > 15 ? 15 : 12 * (RING - 2)));   // stage 0 landed
        if (12 * (RING - 2) > 15) __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        for (int i = 0; i < G; ++i) {
            issue(i + RING - 1);
#if RING == 3
            __builtin_amdgcn_s_waitcnt(0x0F70 | 12);          // everything but the 12 just issued: stage i+1 landed
#else
            __builtin_amdgcn_s_waitcnt(0x4F70 | 8);           // RING == 4: vmcnt(24) = hi bits 01, low 1000
#endif
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
    } else {
        // ---- consumer
        const int wn = wave - 4, r = lane & 31, h = lane >> 5;
        const int sw = (r >> 1) & 7;
        int coff[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) coff[c] = ((2 * c + h) ^ sw) * 4;
        const int qrow = r * WROW, prow = QROWS * WROW + (wn * 64 + r) * WROW;
        f32x16 acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
        struct Frags { f32x4 q[4], p[2]; };
        auto frags = [&](const float* st, int c) {
            Frags f;
#pragma unroll
            for (int n = 0; n < 2; ++n) f.p[n] = *reinterpret_cast<const f32x4*>(st + prow + n * 32 * WROW + coff[c]);
#pragma unroll
            for (int m = 0; m < 4; ++m) f.q[m] = *reinterpret_cast<const f32x4*>(st + qrow + m * 32 * WROW + coff[c]);
            return f;
        };
        auto mm = [&](const Frags& f) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.q[m]), __builtin_bit_cast(f16x8, f.p[n]), acc[m][n], 0, 0, 0);
        };
        __builtin_amdgcn_s_barrier();
        for (int i = 0; i < G; ++i) {
            const float* st = lds + (i % RING) * STAGE_WORDS;
            Frags f0 = frags(st, 0), f1 = frags(st, 1);
            mm(f0);
            f0 = frags(st, 2);
            mm(f1);
            f1 = frags(st, 3);
            mm(f0);
            mm(f1);
            if ((i % nk) == nk - 1) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) asm volatile("" ::"v"(acc[a][b]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (acc[0][0][0] == 12345.f) out[0] = acc[1][1][3];
    }
}

int main() {
    const int D = 512, ld = D / 2, nk = D / 64;
    for (int rows : {1024, 102400}) {
        std::vector<uint16_t> h((size_t)rows * D);
        unsigned s = 12345u;
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u;
            const float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f;
            _Float16 q = (_Float16)f;
            v = *reinterpret_cast<uint16_t*>(&q);
        }
        float *x, *out;
        (void)hipMalloc(&x, h.size() * 2);
        (void)hipMalloc(&out, 4);
        (void)hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        const size_t lds_bytes = (size_t)RING * STAGE_WORDS * 4;
        (void)hipFuncSetAttribute((const void*)run, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const int ntiles = 128;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(run, dim3(256), dim3(512), lds_bytes, 0, x, (int64_t)rows, ld, nk, 4, out);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(run, dim3(256), dim3(512), lds_bytes, 0, x, (int64_t)rows, ld, nk, ntiles, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * QROWS * PROWS * D * ntiles * 256;
            printf("ring %d rows %6d: %.3f ms  %.0f TFLOP/s (%.3f of 2500)  %.3f us per 48 KB stage (%s)\n", RING, rows, ms, flop / ms * 1e-9,
                   flop / ms * 1e-9 / 2500.0, ms * 1e3 / (ntiles * nk), hipGetErrorString(hipGetLastError()));
        }
        (void)hipFree(x);
        (void)hipFree(out);
    }
    return 0;
}
