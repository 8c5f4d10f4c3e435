// op_rates.hip -- issue cost of one wave64 instruction by OPCODE, at 1 / 2 / 3 / 4 waves per SIMD, wall-clock timed (HIP events, no
// clock assumption) with the effective shader clock read beside it (s_memtime / s_memrealtime).  issue_slots.hip found ONE
// instruction (v_add_u32, VOP2) at 2.45 clocks where every packed / VOP3 form costs 4.3 - 4.5: this table asks which other opcodes
// share that rate (encoding? integer class? operand count?), i.e. what an update kernel could be rebuilt from.
// Every stream is 8 independent chains x 8 repetitions per loop trip, each instruction reading and writing its own register.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 2000;
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define Q8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, SEQ)                                                                                               \
__global__ void NAME(uint64_t* out, uint32_t seed) {                                                                    \
    uint32_t v0 = threadIdx.x + seed, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19; \
    uint32_t c = seed | 0x00010001u, d = seed * 3u + 1u;                                                                \
    const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();                            \
    for (int it = 0; it < ITER; ++it) {                                                                                 \
        _Pragma("unroll") for (int k = 0; k < 8; ++k)                                                                   \
            asm volatile(R8(SEQ) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : [c] "v"(c), [d] "v"(d), [s] "s"(seed) : "memory", "vcc", "scc"); \
    }                                                                                                                   \
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();                            \
    const uint32_t x = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;                                                          \
    if ((threadIdx.x & 63) == 0) {                                                                                      \
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                          \
        out[2 * w] = (c1 - c0) + (x == 0x1234567u ? 1 : 0);                                                             \
        out[2 * w + 1] = r1 - r0;                                                                                       \
    }                                                                                                                   \
}
typedef void (*kern_t)(uint64_t*, uint32_t);
// one line per opcode: M(i) expands to the instruction on chain register i
#define M_add_u32(i)   "v_add_u32 %" #i ", %" #i ", %[c]\n"
#define M_add_u32_s(i) "v_add_u32 %" #i ", %[s], %" #i "\n"
#define M_add_u32_e64(i) "v_add_u32_e64 %" #i ", %" #i ", %[c]\n"
#define M_sub_u32(i)   "v_sub_u32 %" #i ", %" #i ", %[c]\n"
#define M_subrev_u32(i) "v_subrev_u32 %" #i ", %" #i ", %[c]\n"
#define M_and_b32(i)   "v_and_b32 %" #i ", %" #i ", %[c]\n"
#define M_or_b32(i)    "v_or_b32 %" #i ", %" #i ", %[c]\n"
#define M_xor_b32(i)   "v_xor_b32 %" #i ", %" #i ", %[c]\n"
#define M_lshlrev(i)   "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define M_lshrrev(i)   "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define M_ashrrev(i)   "v_ashrrev_i32 %" #i ", 1, %" #i "\n"
#define M_min_u32(i)   "v_min_u32 %" #i ", %" #i ", %[c]\n"
#define M_max_i32(i)   "v_max_i32 %" #i ", %" #i ", %[c]\n"
#define M_cndmask(i)   "v_cndmask_b32 %" #i ", %" #i ", %[c], vcc\n"
#define M_mul_u24(i)   "v_mul_u32_u24 %" #i ", %" #i ", %[c]\n"
#define M_add_u16(i)   "v_add_u16 %" #i ", %" #i ", %[c]\n"
#define M_sub_u16(i)   "v_sub_u16 %" #i ", %" #i ", %[c]\n"
#define M_min_u16(i)   "v_min_u16 %" #i ", %" #i ", %[c]\n"
#define M_max_i16(i)   "v_max_i16 %" #i ", %" #i ", %[c]\n"
#define M_lshl_b16(i)  "v_lshlrev_b16 %" #i ", 1, %" #i "\n"
#define M_add_co(i)    "v_add_co_u32 %" #i ", vcc, %" #i ", %[c]\n"
#define M_addc_co(i)   "v_addc_co_u32 %" #i ", vcc, %" #i ", %[c], vcc\n"
#define M_mov(i)       "v_mov_b32 %" #i ", %[c]\n"
#define M_not(i)       "v_not_b32 %" #i ", %" #i "\n"
#define M_bfrev(i)     "v_bfrev_b32 %" #i ", %" #i "\n"
#define M_add3(i)      "v_add3_u32 %" #i ", %" #i ", %[c], %[d]\n"
#define M_and_or(i)    "v_and_or_b32 %" #i ", %" #i ", %[c], %[d]\n"
#define M_or3(i)       "v_or3_b32 %" #i ", %" #i ", %[c], %[d]\n"
#define M_xad(i)       "v_xad_u32 %" #i ", %" #i ", %[c], %[d]\n"
#define M_lshl_add(i)  "v_lshl_add_u32 %" #i ", %" #i ", 1, %[c]\n"
#define M_lshl_or(i)   "v_lshl_or_b32 %" #i ", %" #i ", 1, %[c]\n"
#define M_add_lshl(i)  "v_add_lshl_u32 %" #i ", %" #i ", %[c], 1\n"
#define M_bfe(i)       "v_bfe_u32 %" #i ", %" #i ", 1, 31\n"
#define M_bfi(i)       "v_bfi_b32 %" #i ", %[c], %" #i ", %[d]\n"
#define M_alignbit(i)  "v_alignbit_b32 %" #i ", %" #i ", %[c], 1\n"
#define M_alignbyte(i) "v_alignbyte_b32 %" #i ", %" #i ", %[c], 1\n"
#define M_perm(i)      "v_perm_b32 %" #i ", %" #i ", %[c], %[d]\n"
#define M_min3(i)      "v_min3_u32 %" #i ", %" #i ", %[c], %[d]\n"
#define M_med3(i)      "v_med3_i32 %" #i ", %" #i ", %[c], %[d]\n"
#define M_mad_u24(i)   "v_mad_u32_u24 %" #i ", %" #i ", %[c], %[d]\n"
#define M_mad_u16(i)   "v_mad_u16 %" #i ", %" #i ", %[c], %[d]\n"
#define M_sad_u16(i)   "v_sad_u16 %" #i ", %" #i ", %[c], %[d]\n"
#define M_sad_u32(i)   "v_sad_u32 %" #i ", %" #i ", %[c], %[d]\n"
#define M_bitop3(i)    "v_bitop3_b32 %" #i ", %" #i ", %[c], %[d] bitop3:0xe4\n"
#define M_pk_add_u16(i) "v_pk_add_u16 %" #i ", %" #i ", %[c]\n"
#define M_pk_sub_u16(i) "v_pk_sub_u16 %" #i ", %" #i ", %[c]\n"
#define M_pk_min_u16(i) "v_pk_min_u16 %" #i ", %" #i ", %[c]\n"
#define M_pk_max_i16(i) "v_pk_max_i16 %" #i ", %" #i ", %[c]\n"
#define M_pk_sub_clamp(i) "v_pk_sub_i16 %" #i ", %" #i ", %[c] clamp\n"
#define M_pk_lshl(i)   "v_pk_lshlrev_b16 %" #i ", 1, %" #i "\n"
#define M_pk_mad_u16(i) "v_pk_mad_u16 %" #i ", %" #i ", %[c], %[d]\n"
#define M_pk_mul_u16(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %[c]\n"
#define M_cmp_u32(i)   "v_cmp_gt_u32 vcc, %" #i ", %[c]\n"
#define M_cmp_u16(i)   "v_cmp_gt_u16 vcc, %" #i ", %[c]\n"
#define M_cmp_e64(i)   "v_cmp_gt_u32_e64 s[20:21], %" #i ", %[c]\n"
#define M_add_dpp(i)   "v_add_u32_dpp %" #i ", %" #i ", %[c] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define M_mov_dpp(i)   "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define M_add_sdwa(i)  "v_add_u32_sdwa %" #i ", %" #i ", %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define M_fma_f32(i)   "v_fma_f32 %" #i ", %" #i ", %[c], %[d]\n"
#define M_add_f32(i)   "v_add_f32 %" #i ", %" #i ", %[c]\n"
#define M_mul_lo_u32(i) "v_mul_lo_u32 %" #i ", %" #i ", %[c]\n"
#define M_cvt_pk(i)    "v_cvt_pk_u16_u32 %" #i ", %" #i ", %[c]\n"
#define M_bitop3_s(i)  "v_bitop3_b32 %" #i ", %" #i ", %[c], %[s] bitop3:0xe4\n"
#define M_bitop3_lit(i) "v_bitop3_b32 %" #i ", %" #i ", %[c], 0x55555555 bitop3:0xe4\n"
#define M_and_lit(i)   "v_and_b32 %" #i ", 0x55555555, %" #i "\n"
#define M_and_inl(i)   "v_and_b32 %" #i ", 15, %" #i "\n"
#define M_add_inl(i)   "v_add_u32 %" #i ", 1, %" #i "\n"
#define M_lshl_v(i)    "v_lshlrev_b32 %" #i ", %[c], %" #i "\n"
#define M_lshr_v(i)    "v_lshrrev_b32 %" #i ", %[c], %" #i "\n"
#define M_pk_add_s(i)  "v_pk_add_u16 %" #i ", %" #i ", %[s]\n"
#define M_perm_s(i)    "v_perm_b32 %" #i ", %" #i ", %[c], %[s]\n"
#define A_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %[c]\n"
#define A_ADD32(i) "v_add_u32 %" #i ", %" #i ", %[c]\n"
#define A_MIN(i)   "v_pk_min_i16 %" #i ", %" #i ", %[d]\n"
#define A_SUB(i)   "v_pk_sub_i16 %" #i ", %" #i ", %[d] clamp\n"
#define A_PERM(i)  "v_perm_b32 %" #i ", %" #i ", %[c], %[d]\n"
#define A_BFI(i)   "v_bfi_b32 %" #i ", %[c], %" #i ", %[d]\n"
#define A_BIT(i)   "v_bitop3_b32 %" #i ", %" #i ", %[c], %[d] bitop3:0xe4\n"
// the update step's mix with every instruction 8 slots from the one it depends on (as in the kernel): chain i only via M_x(0)
#define M_acs_pk(i)    Q8(A_PKADD) Q8(A_PKADD) Q8(A_MIN) Q8(A_SUB)
#define M_acs_a32(i)   Q8(A_ADD32) Q8(A_ADD32) Q8(A_MIN) Q8(A_SUB)
#define M_gat_bfi(i)   Q8(A_PERM) Q8(A_BFI)
#define M_gat_bit(i)   Q8(A_PERM) Q8(A_BIT)
#define M_cnd_e64(i)   "v_cndmask_b32_e64 %" #i ", %" #i ", %[c], s[20:21]\n"
#define M_cmp_cnd(i)   "v_cmp_gt_u32 vcc, %" #i ", %[c]\nv_cndmask_b32 %" #i ", %" #i ", %[d], vcc\n"
#define M_cmp_cnd_far(i) "v_cndmask_b32 %" #i ", %" #i ", %[d], vcc\n"
#define M_bfe_i32(i)   "v_bfe_i32 %" #i ", %" #i ", 3, 1\n"
#define M_pk_ashr(i)   "v_pk_ashrrev_i16 %" #i ", 15, %" #i "\n"
#define M_swap32(i)    "v_permlane32_swap_b32 %" #i ", %[dd" #i "]\n"
// mixes: the update step's add / min / sub pattern with the adds in either encoding
#define M_mix_pk(i)    "v_pk_add_u16 %" #i ", %" #i ", %[c]\nv_pk_add_u16 %" #i ", %" #i ", %[d]\nv_pk_min_i16 %" #i ", %" #i ", %[c]\nv_pk_sub_i16 %" #i ", %" #i ", %[d] clamp\n"
#define M_mix_a32(i)   "v_add_u32 %" #i ", %" #i ", %[c]\nv_add_u32 %" #i ", %" #i ", %[d]\nv_pk_min_i16 %" #i ", %" #i ", %[c]\nv_pk_sub_i16 %" #i ", %" #i ", %[d] clamp\n"
#define LIST(X) X(add_u32,"VOP2",1) X(add_u32_s,"VOP2 sgpr src",1) X(add_u32_e64,"VOP3",1) X(sub_u32,"VOP2",1) X(subrev_u32,"VOP2",1) X(and_b32,"VOP2",1) X(or_b32,"VOP2",1) \
  X(xor_b32,"VOP2",1) X(lshlrev,"VOP2",1) X(lshrrev,"VOP2",1) X(ashrrev,"VOP2",1) X(min_u32,"VOP2",1) X(max_i32,"VOP2",1) X(cndmask,"VOP2",1) X(mul_u24,"VOP2",1) \
  X(add_u16,"VOP2",1) X(sub_u16,"VOP2",1) X(min_u16,"VOP2",1) X(max_i16,"VOP2",1) X(lshl_b16,"VOP2",1) X(add_co,"VOP2",1) X(addc_co,"VOP2",1) \
  X(mov,"VOP1",1) X(not,"VOP1",1) X(bfrev,"VOP1",1) \
  X(add3,"VOP3",1) X(and_or,"VOP3",1) X(or3,"VOP3",1) X(xad,"VOP3",1) X(lshl_add,"VOP3",1) X(lshl_or,"VOP3",1) X(add_lshl,"VOP3",1) X(bfe,"VOP3",1) X(bfi,"VOP3",1) \
  X(alignbit,"VOP3",1) X(alignbyte,"VOP3",1) X(perm,"VOP3",1) X(min3,"VOP3",1) X(med3,"VOP3",1) X(mad_u24,"VOP3",1) X(mad_u16,"VOP3",1) X(sad_u16,"VOP3",1) X(sad_u32,"VOP3",1) \
  X(bitop3,"VOP3",1) X(mul_lo_u32,"VOP3",1) X(cvt_pk,"VOP3",1) X(fma_f32,"VOP3",1) X(add_f32,"VOP2",1) \
  X(pk_add_u16,"VOP3P",1) X(pk_sub_u16,"VOP3P",1) X(pk_min_u16,"VOP3P",1) X(pk_max_i16,"VOP3P",1) X(pk_sub_clamp,"VOP3P",1) X(pk_lshl,"VOP3P",1) X(pk_mad_u16,"VOP3P",1) X(pk_mul_u16,"VOP3P",1) \
  X(cmp_u32,"VOPC",1) X(cmp_u16,"VOPC",1) X(cmp_e64,"VOP3 cmp",1) X(add_dpp,"VOP2 DPP",1) X(mov_dpp,"VOP1 DPP",1) X(add_sdwa,"VOP2 SDWA",1) \
  X(bitop3_s,"VOP3 sgpr src",1) X(and_lit,"VOP2 literal src",1) X(and_inl,"VOP2 inline const",1) X(add_inl,"VOP2 inline const",1) \
  X(lshl_v,"VOP2 vgpr shift",1) X(lshr_v,"VOP2 vgpr shift",1) X(pk_add_s,"VOP3P sgpr src",1) X(perm_s,"VOP3 sgpr src",1) \
  X(cnd_e64,"VOP3 cndmask, sgpr-pair condition",1) X(cmp_cnd,"v_cmp + v_cndmask pairs",2) X(bfe_i32,"VOP3",1) X(pk_ashr,"VOP3P",1) \
  X(mix_pk,"2 pk_add + min + sub, back to back dependent",4) X(mix_a32,"2 add_u32 + min + sub, back to back dependent",4) \
  X(acs_pk,"2 pk_add + min + sub, 8 apart",32) X(acs_a32,"2 add_u32 + min + sub, 8 apart",32) X(gat_bfi,"perm + bfi, 8 apart",16) X(gat_bit,"perm + bitop3, 8 apart",16)
#define MK(name, cls, n) KERNEL(k_##name, M_##name)
LIST(MK)
#define ROW(name, cls, n) {#name, cls, k_##name, n},

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int wall_khz = 0;
    if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess || wall_khz <= 0) wall_khz = 100000;
    printf("device %s, %d CUs, s_memrealtime rate %d kHz; %d x 64 slots per wave\n", prop.name, cus, wall_khz, ITER);
    uint64_t* d;
    const size_t max_waves = (size_t)cus * 4 * 4;
    CHECK(hipMalloc((void**)&d, max_waves * 2 * sizeof(uint64_t)));
    std::vector<uint64_t> h(max_waves * 2);
    struct K { const char* name; const char* cls; kern_t k; int per_slot; } ks[] = { LIST(ROW) };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-14s %-32s | ns per instruction per SIMD          | clk per instruction (ns x measured MHz)\n", "opcode", "class");
    printf("%-14s %-32s | %8s %8s %8s %8s | %8s %8s %8s %8s\n", "waves / SIMD", "", "1", "2", "3", "4", "1", "2", "3", "4");
    for (auto& k : ks) {
        double ns[4], clk[4];
        int col = 0;
        for (int w : {1, 2, 3, 4}) {
            const int blocks = cus * w;                       // 256 threads = 4 waves = one per SIMD; w blocks per CU
            for (int warm = 0; warm < 2; ++warm) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const size_t waves = (size_t)blocks * 4;
            CHECK(hipMemcpy(h.data(), d, waves * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
            std::vector<double> f;
            for (size_t i = 0; i < waves; ++i) if (h[2 * i + 1]) f.push_back((double)h[2 * i] / (double)h[2 * i + 1]);
            std::sort(f.begin(), f.end());
            const double mhz = f[f.size() / 2] * wall_khz / 1000.0;
            const double instr_per_simd = (double)ITER * 64.0 * w * k.per_slot;
            ns[col] = ms * 1e6 / 3.0 / instr_per_simd;
            clk[col] = ns[col] * mhz / 1000.0;
            ++col;
        }
        printf("%-14s %-32s | %8.3f %8.3f %8.3f %8.3f | %8.2f %8.2f %8.2f %8.2f\n", k.name, k.cls, ns[0], ns[1], ns[2], ns[3], clk[0], clk[1], clk[2], clk[3]);
        fflush(stdout);
    }
    return 0;
}
