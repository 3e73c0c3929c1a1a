// Microbenchmark: the f16 filter engine's pipeline with 32-deep stages in a FOUR-stage ring (4 x 32 KB), the DMA of
// stage s+3 spread evenly over the MFMAs of stage s (one piece per four MFMAs), counted vmcnt (two stages stay in flight
// across every barrier).  256 x 256 tile, 8 waves (2 x 4), every wave loads and multiplies - as the production engine,
// which uses 64-deep stages in a two-stage ring, issues a stage's DMA during the first half of the previous stage and waits
// vmcnt(0) before every barrier.  tools/ubench/dma_depth.hip: the LDS-DMA path of a CU delivers 57 GB/s (64 KB per 1.13 us)
// whatever the depth - the question here is whether keeping it busy ALL the time beats twice the barriers.
// LDS image: rows of 64 B (32 f16), four 16-B slots, slot = chunk ^ ((row >> 2) & 3): conflict-free ds_read_b128.
// Build: hipcc --offload-arch=gfx950 -O3 ring32.hip -o ring32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int RW = 16;                                    // LDS row: 16 words = 32 f16
constexpr int TB = 256;
constexpr int PART = TB * RW;                             // one operand's sub-slab: 16 KB
constexpr int SUB = 2 * PART;                             // 32 KB
constexpr int RING = 4;

__global__ void __launch_bounds__(512, 1) run(const float* __restrict__ X, int64_t n_rows, int ld /* words */, int nsub /* D / 32 */,
                                             int ntiles, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
    const int64_t T = n_rows / TB;
    const int64_t pb = (blockIdx.x * 5) % T;
    const int S = ntiles * nsub;                          // sub-stages
    // ---- DMA: lane l -> row (l >> 2) of a 16-row piece, slot l & 3 holding logical chunk (l & 3) ^ (l >> 4)
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned vo = (unsigned)(((lane >> 2) * ld + chunk * 4) * 4);
    __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + pb * TB * (int64_t)ld), 0, TB * ld * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t qrs = prs;
    int fs = 0;                                           // sub-stage being fetched
    auto set_q = [&](int s) {
        const int t = s / nsub;
        const int64_t qt = (blockIdx.x * 3 + t) % T;
        qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + qt * TB * (int64_t)ld), 0, s < S ? TB * ld * 4 : 0, 0x00020000);
    };
    set_q(0);
    // piece i (0..3) of sub-stage fs: i < 2 -> Q rows (i * 8 + wave) * 16 .., else P rows
    auto piece = [&](int i) {
        const int pq = (i & 1) * 8 + wave;
        const unsigned so = (unsigned)((pq * 16 * ld + (fs % nsub) * RW) * 4);
        float* dst = lds + (fs % RING) * SUB + (i >> 1) * PART + pq * 16 * RW;
        __builtin_amdgcn_raw_ptr_buffer_load_lds((i >> 1) ? prs : qrs, (__attribute__((address_space(3))) void*)dst, 16, (int)vo, (int)so, 0, 0);
    };
    auto next_fetch = [&]() {
        ++fs;
        if (fs % nsub == 0) set_q(fs);
    };
    // ---- fragments: chunk 2c + h of row r sits in slot (2c + h) ^ ((r >> 2) & 3)
    const int key = (r >> 2) & 3;
    int coff[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) coff[c] = ((2 * c + h) ^ key) * 4;
    const int qrow = (wm * 128 + r) * RW, prow = PART + (wn * 64 + r) * RW;
    struct Frags { f32x4 q[4], p[2]; };
    auto frags = [&](const float* st, int c) {
        Frags f;
#pragma unroll
        for (int n = 0; n < 2; ++n) f.p[n] = *reinterpret_cast<const f32x4*>(st + prow + n * 32 * RW + coff[c]);
#pragma unroll
        for (int m = 0; m < 4; ++m) f.q[m] = *reinterpret_cast<const f32x4*>(st + qrow + m * 32 * RW + coff[c]);
        return f;
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // eight MFMAs of one chunk with DMA pieces j0, j0 + 1 behind the fourth and the eighth
    auto mm = [&](const Frags& f, int j0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.q[m]), __builtin_bit_cast(f16x8, f.p[n]), acc[m][n], 0, 0, 0);
            if (m & 1) piece(j0 + (m >> 1));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // prologue: sub-stages 0, 1, 2
    for (int s = 0; s < RING - 1; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) piece(i);
        next_fetch();
    }
    __builtin_amdgcn_s_waitcnt(0x0F70 | 8);               // sub-stage 0 landed (8 pieces of 1 and 2 may be in flight)
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < S; ++s) {
        const float* st = lds + (s % RING) * SUB;
        Frags f0 = frags(st, 0);
        Frags f1 = frags(st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(f0, 0);
        mm(f1, 2);
        next_fetch();
        if ((s % nsub) == nsub - 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) asm volatile("" ::"v"(acc[a][b]));
        }
        __builtin_amdgcn_s_waitcnt(0x0F70 | 8);           // all but the last eight pieces: sub-stage s + 1 has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (acc[0][0][0] == 12345.f) out[0] = acc[1][1][3];
}

int main() {
    const int D = 512, ld = D / 2, nsub = D / 32;
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
        const size_t lds_bytes = (size_t)RING * SUB * 4;
        (void)hipFuncSetAttribute((const void*)run, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const int ntiles = 64;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(run, dim3(256), dim3(512), lds_bytes, 0, x, (int64_t)rows, ld, nsub, 4, out);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(run, dim3(256), dim3(512), lds_bytes, 0, x, (int64_t)rows, ld, nsub, ntiles, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * TB * TB * D * ntiles * 256;
            printf("ring32 rows %6d: %.3f ms  %.0f TFLOP/s (%.3f of 2500)  %.3f us per 64-deep slab (%s)\n", rows, ms, flop / ms * 1e-9,
                   flop / ms * 1e-9 / 2500.0, ms * 1e3 / (ntiles * (D / 64)), hipGetErrorString(hipGetLastError()));
        }
        (void)hipFree(x);
        (void)hipFree(out);
    }
    return 0;
}
