// Microbenchmark (round 3): per-CU throughput of the L2 -> LDS path on gfx950.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), 1 KiB per wave-instruction, `depth` instructions in flight per wave
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staged), same pieces
//   mode 2: global_load_dwordx4 -> VGPR only (no LDS write): the load path alone
// Every workgroup streams `iters` x (waves x depth) KiB from a window of `win_kb` KiB (per workgroup, so the working set
// stays L2 resident when win_kb * workgroups-per-XCD < 4 MiB).  Build: hipcc --offload-arch=gfx950 -O3 -o dma_mb dma_microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const char* __restrict__ src, long long win_bytes, int iters,
                                                      float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = src + (long long)blockIdx.x * win_bytes;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    float acc = 0.f;
    long long off = (long long)wave * DEPTH * 1024;
    const long long stride = (long long)nw * DEPTH * 1024;
    for (int it = 0; it < iters; ++it) {
        if (off + DEPTH * 1024 > win_bytes) off = (long long)wave * DEPTH * 1024;
        if (MODE == 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                __builtin_amdgcn_global_load_lds(base + off + d * 1024 + lane * 16,
                                                 (lds_ptr_t)(uintptr_t)(lds0 + (unsigned)((wave * DEPTH + d) * 1024)), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const uint4*>(base + off + d * 1024 + lane * 16);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (MODE == 1) *reinterpret_cast<uint4*>(smem + (wave * DEPTH + d) * 1024 + lane * 16) = v[d];
                else acc += __uint_as_float(v[d].x ^ v[d].w);
            }
        }
        off += stride;
    }
    if (MODE == 0 || MODE == 1) acc += reinterpret_cast<float*>(smem)[threadIdx.x];
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int DEPTH>
static void run(const char* src, int wgs, int waves, long long win_bytes, int iters, float* sink, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t smem = (size_t)waves * DEPTH * 1024;
    auto k = stream_kernel<MODE, DEPTH>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(waves * 64), smem, 0, src, win_bytes, 8, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(waves * 64), smem, 0, src, win_bytes, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * waves * DEPTH * 1024.0 * iters;
    printf("%-22s wgs=%4d waves=%2d depth=%2d win=%5lld KiB: %8.3f ms  %7.1f GB/s chip  %6.1f GB/s per CU (%.1f B/clk @2.1GHz)\n",
           name, wgs, waves, DEPTH, win_bytes / 1024, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.1);
}

int main() {
    const long long total = 1ll << 30;
    char* src;
    float* sink;
    hipMalloc(&src, total);
    hipMemset(src, 1, total);
    hipMalloc(&sink, 4);
    for (long long win_kb : {64ll, 2048ll}) {          // L2-resident per-WG window / streaming from MALL+HBM
        const long long wb = win_kb * 1024;
        const int wgs = 256;
        const int iters = 2000;
        printf("--- window %lld KiB per workgroup, %d workgroups (1 per CU)\n", win_kb, wgs);
        run<0, 4>(src, wgs, 4, wb, iters, sink, "lds-dma");
        run<0, 8>(src, wgs, 4, wb, iters, sink, "lds-dma");
        run<0, 4>(src, wgs, 8, wb, iters, sink, "lds-dma");
        run<0, 8>(src, wgs, 8, wb, iters, sink, "lds-dma");
        run<0, 8>(src, wgs, 16, wb, iters, sink, "lds-dma");
        run<1, 4>(src, wgs, 4, wb, iters, sink, "regs+ds_write");
        run<1, 8>(src, wgs, 4, wb, iters, sink, "regs+ds_write");
        run<1, 4>(src, wgs, 8, wb, iters, sink, "regs+ds_write");
        run<1, 8>(src, wgs, 8, wb, iters, sink, "regs+ds_write");
        run<1, 8>(src, wgs, 16, wb, iters, sink, "regs+ds_write");
        run<2, 8>(src, wgs, 4, wb, iters, sink, "regs only");
        run<2, 8>(src, wgs, 8, wb, iters, sink, "regs only");
        run<2, 8>(src, wgs, 16, wb, iters, sink, "regs only");
    }
    return 0;
}
