// clock_rates.hip -- calibrates the two counters the clock probes and the kernel stamps read:
//   s_memrealtime against HIP-event (wall) time: one wave spins until N ticks have passed -> ticks per second (100 MHz expected)
//   s_memtime against s_memrealtime in the same spin -> what s_memtime counts per second on an otherwise idle card
// build: hipcc -O2 --offload-arch=gfx950 -o clock_rates clock_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void spin(uint64_t ticks, uint64_t* out) {
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    uint64_t r = r0;
    while (r - r0 < ticks) { __builtin_amdgcn_s_sleep(8); r = __builtin_amdgcn_s_memrealtime(); }
    if (threadIdx.x == 0) { out[0] = r - r0; out[1] = __builtin_amdgcn_s_memtime() - c0; }
}
int main() {
    uint64_t* d; uint64_t h[2];
    if (hipMalloc(&d, 16) != hipSuccess) return 1;
    int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    int ckhz = 0; (void)hipDeviceGetAttribute(&ckhz, hipDeviceAttributeClockRate, 0);
    printf("hipDeviceAttributeWallClockRate %d kHz, hipDeviceAttributeClockRate %d kHz\n", khz, ckhz);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (uint64_t ticks : {10000000ull, 50000000ull, 100000000ull}) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, ticks, d);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("spin %llu realtime ticks: %.3f ms by HIP events -> s_memrealtime %.3f MHz, s_memtime %.3f MHz (ratio %.4f)\n",
               (unsigned long long)h[0], ms, h[0] / (ms * 1e3), h[1] / (ms * 1e3), (double)h[1] / (double)h[0]);
    }
    return 0;
}
