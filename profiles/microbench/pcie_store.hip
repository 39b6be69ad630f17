// Micro-benchmark: kernel stores from HBM to pinned host memory vs hipMemcpyAsync (DMA), 8 MB and 1 MB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_copy_nt(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(src[i], dst + i);
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (size_t bytes : {(size_t)1 << 20, (size_t)8300000 / 16 * 16, (size_t)32 << 20}) {
        void *d, *h;
        hipMalloc(&d, bytes); hipHostMalloc(&h, bytes, hipHostMallocDefault);
        hipMemset(d, 1, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 4; ++mode) {
            for (int blocks : {64, 256, 1024}) {
                if (mode == 0 && blocks != 64) continue;
                float best = 1e9;
                for (int it = 0; it < 6; ++it) {
                    hipEventRecord(e0, st);
                    if (mode == 0) hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st);
                    else if (mode == 1) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, (const f4*)d, (f4*)h, bytes / 16);
                    else if (mode == 2) hipLaunchKernelGGL(k_copy_nt, dim3(blocks), dim3(256), 0, st, (const f4*)d, (f4*)h, bytes / 16);
                    else hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(64), 0, st, (const f4*)d, (f4*)h, bytes / 16);
                    hipEventRecord(e1, st);
                    hipStreamSynchronize(st);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (it > 0 && ms < best) best = ms;
                }
                printf("%8zu bytes mode %d (%s) blocks %4d: %.1f us  %.1f GB/s\n", bytes, mode,
                       mode == 0 ? "hipMemcpyAsync" : mode == 1 ? "kernel stores" : mode == 2 ? "kernel nt stores" : "kernel 64thr", blocks, best * 1e3,
                       bytes / (best * 1e-3) / 1e9);
            }
        }
        hipFree(d); hipHostFree(h);
    }
    return 0;
}
