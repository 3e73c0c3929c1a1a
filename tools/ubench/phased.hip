// Microbenchmark of the wide engine's pipeline alone (csrc/wide_engine.h -> wide_phased.h, or the one-barrier schedule
// with -DAM_WIDE_ONE_BARRIER): 256 workgroups, each multiplies its own 256-row P block with `ntiles` Q tiles of a random
// f16 matrix; the epilogue only keeps the accumulators alive.  Prints TFLOP/s for an L2-resident and an L2-missing operand.
// Build (from the repository root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iaudio-metrics_amd/csrc [-D...] tools/ubench/phased.hip -o tools/ubench/phased
#include "pairwise_common.h"
#include "wide_engine.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>

namespace am {
struct KeepEpilogue {
    float sink = 0.f;
    __device__ __forceinline__ void aux_issue(int, int64_t) {}
    __device__ __forceinline__ void aux_commit(int) {}
    __device__ __forceinline__ void aux_dma(int, int64_t, int) {}
    __device__ __forceinline__ void aux_cook(int, int64_t) {}
    __device__ __forceinline__ void finish(int, int64_t, f32x16 (&acc)[4][2]) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) asm volatile("" ::"v"(acc[m][n]));
    }
};
struct SeqTiles {
    int64_t q0, total;
    __device__ __forceinline__ int64_t operator()(int t) const { return (q0 + t) % total; }
};
__global__ void __launch_bounds__(WTHREADS, 1) run(const float* X, int64_t n, int64_t ld, int Dh, int ntiles, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const WLane L;
    KeepEpilogue epi;
    const int64_t T = n / WTB;
    const int64_t pb = (blockIdx.x * 5) % T;
    wide_pipeline(X, n, ld, SeqTiles{(int64_t)blockIdx.x * 3, T}, X, n, ld, pb * WTB, ntiles, Dh, lds, L, epi);
    if (epi.sink == 12345.f) out[0] = epi.sink;
}
}  // namespace am

int main() {
    for (int D : {512, 64})                                   // D = 64: rows of 128 B, a DMA instruction reads 1 KB contiguous
    for (int rows : {1024 * (512 / D), 102400 * (512 / D)}) {
        const int ldh = D / 2;
        std::vector<uint16_t> h((size_t)rows * D);
        unsigned s = 12345u;
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u;
            const float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f;
#ifdef AM_WIDE_BF16
            unsigned u;
            memcpy(&u, &f, 4);
            v = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);          // bf16, round to nearest even
#else
            _Float16 q = (_Float16)f;
            v = *reinterpret_cast<uint16_t*>(&q);
#ifdef UB_DROP_BITS
            v &= (uint16_t)~((1u << UB_DROP_BITS) - 1u);
#endif
#endif
        }
        float *x, *out;
        (void)hipMalloc(&x, h.size() * 2);
        (void)hipMalloc(&out, 4);
        (void)hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        const size_t lds_bytes = 163840;
        (void)hipFuncSetAttribute((const void*)am::run, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const int ntiles = 64 * (512 / D);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(am::run, dim3(256), dim3(am::WTHREADS), lds_bytes, 0, x, (int64_t)rows, (int64_t)ldh, ldh, 4, out);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(am::run, dim3(256), dim3(am::WTHREADS), lds_bytes, 0, x, (int64_t)rows, (int64_t)ldh, ldh, ntiles, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * 256 * 256 * D * ntiles * 256;
            printf("%s D %3d rows %6d: %.3f ms  %.0f TFLOP/s (%.3f of 2500)  %.3f us per k-slab\n", VARIANT, D, rows, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0,
                   ms * 1e3 / (ntiles * (D / 64)));
        }
        (void)hipFree(x);
        (void)hipFree(out);
    }
    return 0;
}
