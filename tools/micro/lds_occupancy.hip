// How many 256-thread workgroups share a CU as a function of their dynamic LDS size (MI355X: 160 KiB per CU).
// Each workgroup waits a fixed time; 2048 of them on 256 CUs take 8 / k rounds when k fit a CU.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lds_occupancy.hip -o lds_occupancy && ./lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_wait(unsigned long long ticks, int *sink)
{
    extern __shared__ int lds[];
    const unsigned long long t0 = wall_clock64();
    lds[threadIdx.x] = (int)t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x] == 12345) *sink = 1;
}
int main()
{
    int *sink; hipMalloc(&sink, 4);
    const unsigned long long ticks = 5000;                    // 50 us at 100 MHz
    for (int bytes = 24 * 1024; bytes <= 40 * 1024; bytes += 512) {
        hipFuncSetAttribute((const void *)k_wait, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_wait, dim3(256), dim3(256), bytes, 0, 10ull, sink);      // warm-up
        hipEventRecord(a);
        hipLaunchKernelGGL(k_wait, dim3(256 * 16), dim3(256), bytes, 0, ticks, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("LDS %6d B: %7.1f us = %.2f rounds of 50 us -> %.2f workgroups per CU\n", bytes, ms * 1e3, ms * 1e3 / 50.0, 16.0 / (ms * 1e3 / 50.0));
    }
    return 0;
}
