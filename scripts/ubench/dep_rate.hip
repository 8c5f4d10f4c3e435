// dep_rate.hip -- does a DEPENDENT packed instruction issue as fast as an independent one?  v_pk_add_u16 / v_pk_min_i16 /
// v_perm_b32 chains with dependency distance D = 1, 2, 4, 8 (D independent accumulators, round robin), at 1, 2, 3, 4 waves per
// SIMD.  Answers whether the update kernels' two-waves-per-SIMD shortfall is dependent-issue latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4000;

#define OP_ADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define OP_MIN(i) "v_pk_min_i16 %" #i ", %" #i ", %8\n"
#define OP_PRM(i) "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define OP_A32(i) "v_add_u32 %" #i ", %" #i ", %8\n"
// 8 instructions per asm block; D accumulators used round robin
#define SEQ1(OP) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0)
#define SEQ2(OP) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1)
#define SEQ4(OP) OP(0) OP(1) OP(2) OP(3) OP(0) OP(1) OP(2) OP(3)
#define SEQ8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define KERNEL(NAME, SEQ)                                                                                          \
__global__ void NAME(uint32_t* out, uint32_t seed) {                                                              \
    uint32_t v0 = threadIdx.x + seed, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19; \
    uint32_t c = seed | 0x00010001u;                                                                               \
    for (int it = 0; it < ITER; ++it) {                                                                            \
        _Pragma("unroll") for (int k = 0; k < 8; ++k)                                                              \
            asm volatile(SEQ : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c)); \
    }                                                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;                            \
}
KERNEL(add1, SEQ1(OP_ADD)) KERNEL(add2, SEQ2(OP_ADD)) KERNEL(add4, SEQ4(OP_ADD)) KERNEL(add8, SEQ8(OP_ADD))
KERNEL(min1, SEQ1(OP_MIN)) KERNEL(min2, SEQ2(OP_MIN)) KERNEL(min4, SEQ4(OP_MIN)) KERNEL(min8, SEQ8(OP_MIN))
KERNEL(prm1, SEQ1(OP_PRM)) KERNEL(prm2, SEQ2(OP_PRM)) KERNEL(prm4, SEQ4(OP_PRM)) KERNEL(prm8, SEQ8(OP_PRM))
KERNEL(a321, SEQ1(OP_A32)) KERNEL(a322, SEQ2(OP_A32)) KERNEL(a324, SEQ4(OP_A32)) KERNEL(a328, SEQ8(OP_A32))
typedef void (*kern_t)(uint32_t*, uint32_t);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* d;
    CHECK(hipMalloc(&d, (size_t)cus * 4 * 8 * 64 * 4));
    struct K { const char* name; kern_t k; } ks[] = {
        {"v_pk_add_u16 D=1", add1}, {"v_pk_add_u16 D=2", add2}, {"v_pk_add_u16 D=4", add4}, {"v_pk_add_u16 D=8", add8},
        {"v_pk_min_i16 D=1", min1}, {"v_pk_min_i16 D=2", min2}, {"v_pk_min_i16 D=4", min4}, {"v_pk_min_i16 D=8", min8},
        {"v_perm_b32   D=1", prm1}, {"v_perm_b32   D=2", prm2}, {"v_perm_b32   D=4", prm4}, {"v_perm_b32   D=8", prm8},
        {"v_add_u32    D=1", a321}, {"v_add_u32    D=2", a322}, {"v_add_u32    D=4", a324}, {"v_add_u32    D=8", a328}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("device %s, %d CUs; cycles per wave-instruction per SIMD at 2.4 GHz nominal\n%-20s %9s %9s %9s %9s\n", prop.name, cus,
           "chain", "1 w/SIMD", "2 w/SIMD", "3 w/SIMD", "4 w/SIMD");
    for (auto& k : ks) {
        printf("%-20s", k.name);
        for (int w : {1, 2, 3, 4}) {
            const int blocks = cus * 4 * w;
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, d, 1u);       // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, d, 1u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_simd = (double)ITER * 64.0 * w;             // wave-instructions issued on one SIMD
            printf(" %9.2f", ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
