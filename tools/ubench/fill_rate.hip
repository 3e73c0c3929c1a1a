// Microbenchmark: how fast can one CU's 8 waves fill a 64 KB LDS stage (512 rows x 128 B, the wide engine's k-slab)?
//   mode 0  LDS-DMA (buffer_load_dwordx4 ... lds), 8 pieces per wave per stage, vmcnt(0) + barrier per stage
//   mode 1  global_load_dwordx4 -> VGPR -> ds_write_b128, same bytes
//   mode 2  global_load_dwordx4 -> VGPR only (no LDS write)
//   +8      32 MFMAs (v_mfma_f32_32x32x16_f16, register operands) per wave per stage beside the fill
//   src_mb  size of the source region the row blocks are drawn from (2 = L2 resident per XCD, 200 = L2-missing)
// Build: hipcc --offload-arch=gfx950 -O3 fill_rate.hip -o fill_rate ; run: ./fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWS = 512, ROWB = 128;                 // stage = 512 rows x 128 B = 64 KB
constexpr int STAGE_WORDS = ROWS * ROWB / 4;

__global__ void __launch_bounds__(512, 1) fill(const float* __restrict__ src, int64_t src_rows, int ld_words, int stages, int mode,
                                               float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mfma = mode & 8;
    const int m = mode & 7;
    // row block of this workgroup: 512 consecutive rows somewhere in the source, moving every 8 stages (a new tile)
    const int srow = tid >> 3, chunk = tid & 7;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    f16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.01f + j); b[j] = (_Float16)(j * 0.5f - lane * 0.02f); }
    float sink = 0.f;
    const int64_t nblk = src_rows / ROWS;
    for (int g = 0; g < stages; ++g) {
        const int64_t blk = ((int64_t)blockIdx.x * 7 + (g >> 3) * 13) % nblk;
        const int kslab = g & 7;
        const float* base = src + (blk * ROWS) * (int64_t)ld_words + kslab * 32;
        float* st = lds + (g & 1) * STAGE_WORDS;
        if (m == 0) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(ROWS * ld_words * 4 - kslab * 128), 0x00020000);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned vo = (unsigned)(((j * 64 + srow) * ld_words + chunk * 4) * 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(st + (j * 64 + wave * 8) * 32), 16, (int)vo, 0, 0, 0);
            }
        } else {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(base + (int64_t)(j * 64 + srow) * ld_words + chunk * 4);
            if (m == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(st + (j * 64 + srow) * 32 + chunk * 4) = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) sink += v[j].x + v[j].w;
            }
        }
        if (mfma) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0)
        __syncthreads();
        sink += st[(tid * 37) & (STAGE_WORDS - 1)];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sink += acc[i][3];
    if (sink == 12345.678f) out[0] = sink;
}

int main(int argc, char** argv) {
    const int stages = 4096;
    for (int src_mb : {16, 200}) {
        const int ld_words = 256;                                     // 1 KB rows (D = 512 f16)
        const int64_t rows = (int64_t)src_mb * 1024 * 1024 / 1024;
        float *src, *out;
        hipMalloc(&src, rows * 1024);
        hipMalloc(&out, 4);
        hipMemset(src, 0x11, rows * 1024);
        hipFuncSetAttribute((const void*)fill, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_WORDS * 4);
        for (int mode : {0, 1, 2, 8, 9, 10, 15}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(fill, dim3(256), dim3(512), 2 * STAGE_WORDS * 4, 0, src, rows, ld_words, 64, mode, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(fill, dim3(256), dim3(512), 2 * STAGE_WORDS * 4, 0, src, rows, ld_words, stages, mode, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double us_stage = ms * 1e3 / stages;
            printf("src %3d MB mode %2d: %.3f us per 64 KB stage  = %.1f GB/s per CU, %.2f TB/s chip%s\n", src_mb, mode, us_stage,
                   65536.0 / us_stage * 1e-3, 65536.0 * 256 / us_stage * 1e-6, (mode & 8) ? "  (+32 MFMA/wave: 1.0 us at 2.05 GHz if alone)" : "");
        }
        hipFree(src); hipFree(out);
    }
    return 0;
}
