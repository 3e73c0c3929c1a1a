// Microbenchmark: do v_mfma_f32_32x32x2_f32 and f32 VALU FMAs overlap on gfx950 SIMDs?
//   mode 0: every wave issues only MFMAs            mode 1: every wave issues only v_fma_f32
//   mode 2: 8 waves per workgroup, waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD)
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float mfma_loop(int iters, float seed) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = seed, y = seed * 0.5f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    return a0[0] + a1[1] + a2[2] + a3[3];
}

__device__ __forceinline__ float valu_loop(int iters, float seed) {
    float r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = seed + j;
    const float m = 1.0000001f, c = 1e-7f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)          // 128 independent-ish FMAs per iteration
#pragma unroll
            for (int j = 0; j < 16; ++j) r[j] = fmaf(r[j], m, c);
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += r[j];
    return s;
}

__global__ void __launch_bounds__(512) bench(int mode, int mfma_iters, int valu_iters, float* out) {
    const int wave = threadIdx.x >> 6;
    float v;
    if (mode == 0) v = mfma_loop(mfma_iters, threadIdx.x * 1e-3f);
    else if (mode == 1) v = valu_loop(valu_iters, threadIdx.x * 1e-3f);
    else v = (wave < 4) ? mfma_loop(mfma_iters, threadIdx.x * 1e-3f) : valu_loop(valu_iters, threadIdx.x * 1e-3f);
    if (v == 123.456f) out[0] = v;
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int mi = 20000;            // 4 MFMAs per iteration: 80000 MFMAs x 64 cycles = 5.12M cycles per wave
    const int vi = 20000;            // 128 FMAs per iteration = 2.56M v_fma per wave x 2 cycles = 5.12M cycles
    for (int mode = 0; mode < 3; ++mode) {
        for (int threads : {256, 512}) {
            if (mode == 2 && threads == 256) continue;
            hipLaunchKernelGGL(bench, dim3(256), dim3(threads), 0, 0, mode, mi, vi, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(bench, dim3(256), dim3(threads), 0, 0, mode, mi, vi, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double waves = 256.0 * threads / 64;
            double mf = 0, vf = 0;
            if (mode == 0) mf = waves * mi * 4.0 * 2 * 32 * 32 * 2;
            if (mode == 1) vf = waves * vi * 128.0 * 64 * 2;
            if (mode == 2) { mf = waves / 2 * mi * 4.0 * 2 * 32 * 32 * 2; vf = waves / 2 * vi * 128.0 * 64 * 2; }
            printf("mode %d threads %d: %.3f ms  mfma %.1f TF  valu %.1f TF  total %.1f TF\n", mode, threads, ms,
                   mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
        }
    }
    return 0;
}
