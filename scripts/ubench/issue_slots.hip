// issue_slots.hip -- what one instruction costs a SIMD, by kind and by how many waves share the SIMD, with the clocks READ
// rather than assumed: every wave brackets its loop with s_memtime (shader clocks) and s_memrealtime (constant reference clock,
// hipDeviceAttributeWallClockRate), and the launch is timed with HIP events on top.  Printed per stream and occupancy:
//   ns / instr / SIMD  from the event time (no clock assumption at all)
//   clk / instr / SIMD from s_memtime, and the effective shader clock = s_memtime ticks / s_memrealtime ticks x reference rate
// Streams: pure packed adds, VOP2 adds, fp32 FMA in both encodings (v_fmac_f32 = VOP2, v_fma_f32 = VOP3), and packed adds
// interleaved 1:1 with instructions that do not use the vector ALU at all (s_nop, s_add_u32, s_waitcnt on an idle counter): if
// a non-VALU instruction were free, "per pair" would cost what one packed add costs.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 3000;

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define S_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %[c]\n"
#define S_ADD32(i) "v_add_u32 %" #i ", %" #i ", %[c]\n"
#define S_FMAC(i)  "v_fmac_f32 %" #i ", %[c], %[c]\n"
#define S_FMA(i)   "v_fma_f32 %" #i ", %" #i ", %[c], %[c]\n"
#define S_PKSUBC(i) "v_pk_sub_i16 %" #i ", %" #i ", %[c] clamp\n"
#define S_PERM(i)  "v_perm_b32 %" #i ", %" #i ", %[c], %[c]\n"
#define S_BFI(i)   "v_bfi_b32 %" #i ", %[c], %" #i ", %[c]\n"
#define S_PK_NOP(i)  "v_pk_add_u16 %" #i ", %" #i ", %[c]\ns_nop 0\n"
#define S_PK_SADD(i) "v_pk_add_u16 %" #i ", %" #i ", %[c]\ns_add_u32 %[sc], %[sc], 1\n"
#define S_PK_WAIT(i) "v_pk_add_u16 %" #i ", %" #i ", %[c]\ns_waitcnt lgkmcnt(0)\n"
#define S_PK_2NOP(i) "v_pk_add_u16 %" #i ", %" #i ", %[c]\ns_nop 0\ns_nop 0\n"

#define KERNEL(NAME, SEQ)                                                                                               \
__global__ void NAME(uint64_t* out, uint32_t seed) {                                                                    \
    uint32_t v0 = threadIdx.x + seed, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19; \
    uint32_t c = seed | 0x00010001u, sc = seed;                                                                         \
    const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();                            \
    for (int it = 0; it < ITER; ++it) {                                                                                 \
        _Pragma("unroll") for (int k = 0; k < 8; ++k)                                                                   \
            asm volatile(R8(SEQ) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) , [sc] "+s"(sc) : [c] "v"(c) : "memory", "scc"); \
    }                                                                                                                   \
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();                            \
    const uint32_t x = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7 ^ sc;                                                         \
    if ((threadIdx.x & 63) == 0) {                                                                                      \
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                          \
        out[2 * w] = (c1 - c0) + (x == 0x1234567u ? 1 : 0);                                                             \
        out[2 * w + 1] = r1 - r0;                                                                                       \
    }                                                                                                                   \
}
KERNEL(k_pkadd, S_PKADD) KERNEL(k_add32, S_ADD32) KERNEL(k_fmac, S_FMAC) KERNEL(k_fma, S_FMA)
KERNEL(k_pk_nop, S_PK_NOP) KERNEL(k_pk_sadd, S_PK_SADD) KERNEL(k_pk_wait, S_PK_WAIT) KERNEL(k_pk_2nop, S_PK_2NOP)
KERNEL(k_pksub, S_PKSUBC) KERNEL(k_perm, S_PERM) KERNEL(k_bfi, S_BFI)
typedef void (*kern_t)(uint64_t*, uint32_t);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int wall_khz = 0;
    if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess || wall_khz <= 0) wall_khz = 100000;
    printf("device %s, %d CUs, nominal clock %d kHz, s_memrealtime rate %d kHz\n", prop.name, cus, prop.clockRate, wall_khz);
    uint64_t* d;
    const size_t max_waves = (size_t)cus * 4 * 4;
    CHECK(hipMalloc((void**)&d, max_waves * 2 * sizeof(uint64_t)));
    std::vector<uint64_t> h(max_waves * 2);
    struct K { const char* name; kern_t k; int per_slot; } ks[] = {
        {"v_pk_add_u16 (VOP3P)", k_pkadd, 1}, {"v_pk_sub_i16 clamp (VOP3P)", k_pksub, 1}, {"v_perm_b32 (VOP3)", k_perm, 1},
        {"v_bfi_b32 (VOP3)", k_bfi, 1}, {"v_add_u32 (VOP2)", k_add32, 1}, {"v_fmac_f32 (VOP2)", k_fmac, 1}, {"v_fma_f32 (VOP3)", k_fma, 1},
        {"pk_add + s_nop       per pair", k_pk_nop, 1}, {"pk_add + s_add_u32   per pair", k_pk_sadd, 1},
        {"pk_add + s_waitcnt   per pair", k_pk_wait, 1}, {"pk_add + 2 x s_nop   per triple", k_pk_2nop, 1}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-34s | %-29s | %-29s | %s\n", "stream (64 slots x 3000 per wave)", "ns per slot per SIMD (events)", "shader clk per slot (s_memtime)", "effective shader clock MHz");
    printf("%-34s | %9s %9s %9s | %9s %9s %9s | %9s %9s %9s\n", "waves per SIMD", "1", "2", "4", "1", "2", "4", "1", "2", "4");
    for (auto& k : ks) {
        double ns[3], clk[3], mhz[3];
        int col = 0;
        for (int w : {1, 2, 4}) {
            const int blocks = cus * w;                       // 256 threads = 4 waves = one per SIMD; w blocks per CU
            for (int warm = 0; warm < 3; ++warm) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const size_t waves = (size_t)blocks * 4;
            CHECK(hipMemcpy(h.data(), d, waves * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
            std::vector<double> c, f;
            for (size_t i = 0; i < waves; ++i) if (h[2 * i + 1]) { c.push_back((double)h[2 * i]); f.push_back((double)h[2 * i] / (double)h[2 * i + 1]); }
            std::sort(c.begin(), c.end()); std::sort(f.begin(), f.end());
            const double slots_per_simd = (double)ITER * 64.0 * w;                      // one launch
            ns[col] = ms * 1e6 / 3.0 / slots_per_simd;
            clk[col] = c[c.size() / 2] / ((double)ITER * 64.0) / w;                     // a wave's clocks per slot / waves sharing the SIMD
            mhz[col] = f[f.size() / 2] * wall_khz / 1000.0;
            ++col;
        }
        printf("%-34s | %9.3f %9.3f %9.3f | %9.2f %9.2f %9.2f | %9.0f %9.0f %9.0f\n", k.name, ns[0], ns[1], ns[2], clk[0], clk[1], clk[2], mhz[0], mhz[1], mhz[2]);
    }
    return 0;
}
