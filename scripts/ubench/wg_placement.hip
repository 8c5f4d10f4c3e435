// wg_placement.hip -- where does the dispatcher put the workgroups of a 2-per-CU kernel, and in which order?
// Every workgroup (512 threads, 72 KiB of LDS: two fit a CU) records HW_ID, XCC_ID and its start time, then spins ~50 us.
// usage: wg_placement [n_workgroups]     prints blockIdx -> (xcc, se, sh, cu), and for each CU the co-resident pairs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(512, 4) probe(unsigned* out) {
    extern __shared__ unsigned smem[];
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
        unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
        unsigned long long t = __builtin_amdgcn_s_memtime();
        out[4 * blockIdx.x + 0] = hw;
        out[4 * blockIdx.x + 1] = xcc;
        out[4 * blockIdx.x + 2] = (unsigned)t;
        out[4 * blockIdx.x + 3] = (unsigned)(t >> 32);
        smem[0] = hw;
    }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 5000ull) {}   // memtime ticks at 100 MHz: 50 us
    if (smem[0] == 0xFFFFFFFFu) out[0] = 0;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2048;
    unsigned* d;
    hipMalloc(&d, n * 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 73856);
    hipLaunchKernelGGL(probe, dim3(n), dim3(512), 73856, 0, d);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n * 4);
    hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> per_cu;
    for (int b = 0; b < n; b++) {
        const unsigned hw = h[4 * b], xcc = h[4 * b + 1] & 0xF;
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned long long t = ((unsigned long long)h[4 * b + 3] << 32) | h[4 * b + 2];
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back({t, b});
        if (b < 48) printf("wg %4d -> xcc %u se %u sh %u cu %2u\n", b, xcc, se, sh, cu);
    }
    printf("distinct CUs used: %zu\n", per_cu.size());
    int shown = 0;
    for (auto& kv : per_cu) {
        std::sort(kv.second.begin(), kv.second.end());
        if (shown++ < 6) {
            printf("CU %05x:", kv.first);
            for (auto& p : kv.second) printf(" wg %d @%llu", p.second, p.first - kv.second[0].first);
            printf("\n");
        }
    }
    // difference between the blockIdx of the first two workgroups of every CU
    std::map<int, int> hist;
    for (auto& kv : per_cu) if (kv.second.size() >= 2) hist[abs(kv.second[1].second - kv.second[0].second)]++;
    for (auto& kv : hist) printf("first-pair blockIdx distance %d: %d CUs\n", kv.first, kv.second);
    return 0;
}
