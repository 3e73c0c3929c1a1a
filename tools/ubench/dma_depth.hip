// Microbenchmark: LDS-DMA throughput of one CU as a function of the bytes it keeps in flight.
//   8 waves per workgroup (one workgroup per CU), each issues buffer_load_dwordx4 ... lds instructions (1 KB each: eight
//   128-B row segments, the engine's access pattern) and waits with a counted vmcnt so that DEPTH instructions per wave
//   stay outstanding: in flight per CU = 8 * DEPTH KB.  No MFMA, no LDS reads, no barriers.
// Answers: is the fill latency-bound (throughput ~ bytes in flight) or throughput-bound (flat)?
// Build: hipcc --offload-arch=gfx950 -O3 dma_depth.hip -o dma_depth
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int DEPTH>
__global__ void __launch_bounds__(512, 1) fill(const float* __restrict__ src, int64_t src_rows, int ld_words, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nblk = src_rows / 512;
    const unsigned vo = (unsigned)(((lane >> 3) * ld_words + (lane & 7) * 4) * 4);
    // each wave owns a 16 KB region of LDS (16 slots of 1 KB) and cycles through it
    float* base = lds + wave * 16 * 256;
    for (int it = 0; it < iters; ++it) {
        const int64_t blk = ((int64_t)blockIdx.x * 7 + (it >> 3) * 13 + wave) % nblk;
        const float* p = src + (blk * 512 + (it & 7) * 64 + wave * 8) * (int64_t)ld_words + ((it * 5) & 7) * 32;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 8 * ld_words * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(base + (it & 15) * 256), 16, (int)vo, 0, 0, 0);
        if (DEPTH < 16) __builtin_amdgcn_s_waitcnt(0x0F70 | (DEPTH - 1));       // at most DEPTH outstanding (DEPTH <= 16)
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (lds[(tid * 37) & 32767] == 12345.678f) out[0] = 1.f;
}

template <int DEPTH>
static void run(const float* src, int64_t rows, int ld_words, float* out, int src_mb) {
    const int iters = 8192;
    const size_t lds_bytes = 128 * 1024;
    (void)hipFuncSetAttribute((const void*)fill<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fill<DEPTH>, dim3(256), dim3(512), lds_bytes, 0, src, rows, ld_words, 256, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fill<DEPTH>, dim3(256), dim3(512), lds_bytes, 0, src, rows, ld_words, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes_cu = 8.0 * iters * 1024;
    printf("src %4d MB  depth %2d (%3d KB in flight per CU): %.1f GB/s per CU = %.2f TB/s chip; 64 KB per %.3f us\n", src_mb, DEPTH, 8 * DEPTH,
           bytes_cu / ms * 1e-6, bytes_cu * 256 / ms * 1e-9, 65536.0 / (bytes_cu / ms * 1e-3) );
}

int main() {
    for (int src_mb : {2, 16, 200}) {
        const int ld_words = 256;
        const int64_t rows = (int64_t)src_mb * 1024;
        float *src, *out;
        (void)hipMalloc(&src, rows * 1024);
        (void)hipMalloc(&out, 4);
        (void)hipMemset(src, 0x11, rows * 1024);
        run<1>(src, rows, ld_words, out, src_mb);
        run<2>(src, rows, ld_words, out, src_mb);
        run<4>(src, rows, ld_words, out, src_mb);
        run<8>(src, rows, ld_words, out, src_mb);
        run<12>(src, rows, ld_words, out, src_mb);
        run<16>(src, rows, ld_words, out, src_mb);
        (void)hipFree(src);
        (void)hipFree(out);
    }
    return 0;
}
