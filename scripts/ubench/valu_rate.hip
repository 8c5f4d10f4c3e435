// valu_rate.hip -- issue rate of the VALU instructions the Viterbi ACS is built from, on gfx950.
// Each kernel runs ITER x 64 independent-ish instructions per wave; we time with waves/SIMD = 1, 2, 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 2000;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// 8 accumulators, 8 instructions per inner block, 8 blocks = 64 instr per iteration
#define KERNEL(NAME, ASMSTR)                                                                       \
__global__ void NAME(uint32_t* out, uint32_t seed) {                                               \
    uint32_t v0 = threadIdx.x + seed, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19; \
    uint32_t c = seed | 0x00010001u;                                                               \
    for (int it = 0; it < ITER; ++it) {                                                            \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                            \
            asm volatile(ASMSTR(0) ASMSTR(1) ASMSTR(2) ASMSTR(3) ASMSTR(4) ASMSTR(5) ASMSTR(6) ASMSTR(7) \
                : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c)); \
        }                                                                                          \
    }                                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;            \
}

#define A_PK_ADD(i)   "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_PK_SUBC(i)  "v_pk_sub_u16 %" #i ", %" #i ", %8 clamp\n"
#define A_PK_MIN(i)   "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define A_PK_MAXI(i)  "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define A_ADD32(i)    "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_ADD16(i)    "v_add_u16 %" #i ", %" #i ", %8\n"
#define A_MIN16(i)    "v_min_u16 %" #i ", %" #i ", %8\n"
#define A_MIN32(i)    "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_LSHLOR(i)   "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define A_PERM(i)     "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define A_CNDM(i)     "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMPADDC(i)  "v_cmp_gt_u16 vcc, %" #i ", %8\nv_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define A_SUBCO(i)    "v_sub_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define A_ADD3(i)     "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define A_SAD16(i)    "v_sad_u16 %" #i ", %" #i ", %8, %8\n"
#define A_MAD16(i)    "v_mad_u16 %" #i ", %" #i ", %8, %8\n"
#define A_PKMAD(i)    "v_pk_mad_u16 %" #i ", %" #i ", %8, %8\n"
#define A_PKFMA16(i)  "v_pk_fma_f16 %" #i ", %" #i ", %8, %8\n"
#define A_FMA32(i)    "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define A_PKADD32(i)  "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define A_SDWA(i)     "v_add_u16_sdwa %" #i ", %" #i ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define A_MIN3(i)     "v_min3_u16 %" #i ", %" #i ", %8, %8\n"
#define A_DOT2(i)     "v_dot2_u32_u16 %" #i ", %" #i ", %8, %" #i "\n"

#define A_CNDS(i)     "v_cndmask_b32 %" #i ", %" #i ", %8, s[10:11]\n"
#define A_BFI(i)      "v_bfi_b32 %" #i ", %8, %" #i ", %8\n"
#define A_ANDOR(i)    "v_and_or_b32 %" #i ", %" #i ", %8, %8\n"
#define A_PKSUBI(i)   "v_pk_sub_i16 %" #i ", %" #i ", %8 clamp\n"
#define A_PKMINI(i)   "v_pk_min_i16 %" #i ", %" #i ", %8\n"
#define A_SHR32(i)    "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define A_AND32(i)    "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_XOR32(i)    "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_SUB16(i)    "v_sub_u16 %" #i ", %" #i ", %8\n"
#define A_MAXI16(i)   "v_max_i16 %" #i ", %" #i ", %8\n"
#define A_MOV(i)      "v_mov_b32 %" #i ", %8\n"
#define A_CMP16(i)    "v_cmp_gt_u16 vcc, %" #i ", %8\n"
#define A_ADDC(i)     "v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %8, %" #i ", 1\n"
#define A_MOVDPP(i)   "v_mov_b32_dpp %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_ADDDPP(i)   "v_add_u16_dpp %" #i ", %8, %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_CNDV(i)     "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_PKADD_S(i)  "v_pk_add_u16 %" #i ", %" #i ", s12\n"
#define A_ADD16_S(i)  "v_add_u16 %" #i ", s12, %" #i "\n"

#define A_PERMSGN(i)  "v_perm_b32 %" #i ", %" #i ", %8, s13\n"
#define A_PERMV(i)    "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define A_ANDOR_S(i)  "v_and_or_b32 %" #i ", %8, s13, %" #i "\n"
#define A_BFI_S(i)    "v_bfi_b32 %" #i ", s13, %8, %" #i "\n"
#define A_OR3(i)      "v_or3_b32 %" #i ", %" #i ", %8, %8\n"
#define A_LSHLADD(i)  "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define A_CMP32(i)    "v_cmp_gt_u32 vcc, %" #i ", %8\n"
#define A_CMPSDWA(i)  "v_cmp_gt_i16_sdwa vcc, %" #i ", %8 src0_sel:WORD_1 src1_sel:WORD_1\n"
#define A_CMPE64(i)   "v_cmp_gt_i16 s[14:15], %" #i ", %8\n"
#define A_PKADD_OPSEL(i) "v_pk_add_u16 %" #i ", %" #i ", %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
#define A_PKASHR(i)   "v_pk_ashrrev_i16 %" #i ", 15, %" #i "\n"
#define A_PKLSHR(i)   "v_pk_lshrrev_b16 %" #i ", 15, %" #i "\n"
#define A_BPERM(i)    "ds_bpermute_b32 %" #i ", %8, %" #i "\ns_waitcnt lgkmcnt(0)\n"
#define A_ADDLIT(i)   "v_add_u32 %" #i ", 0x12345, %" #i "\n"
#define A_ANDLIT(i)   "v_and_b32 %" #i ", 0x80808080, %" #i "\n"
#define A_OR32(i)     "v_or_b32 %" #i ", %" #i ", %8\n"
#define A_SUB32(i)    "v_sub_u32 %" #i ", %" #i ", %8\n"
#define A_MAX32(i)    "v_max_i32 %" #i ", %" #i ", %8\n"
#define A_ASHR32(i)   "v_ashrrev_i32 %" #i ", 31, %" #i "\n"
#define A_BFE(i)      "v_bfe_u32 %" #i ", %" #i ", 15, 1\n"
#define A_MBCNT(i)    "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define A_XAD(i)      "v_xad_u32 %" #i ", %" #i ", %8, %8\n"
#define A_ADDF32(i)   "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_MULF32(i)   "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_ADDF16(i)   "v_add_f16 %" #i ", %" #i ", %8\n"
#define A_PKADDF16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define A_PKMINF16(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
#define A_MED3(i)     "v_med3_i32 %" #i ", %" #i ", %8, %8\n"
#define A_MIN3_32(i)  "v_min3_u32 %" #i ", %" #i ", %8, %8\n"
#define A_SAD8(i)     "v_sad_u8 %" #i ", %" #i ", %8, %8\n"
#define A_DOT4(i)     "v_dot4_u32_u8 %" #i ", %" #i ", %8, %" #i "\n"
#define A_ADDCO(i)    "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"

KERNEL(k_n_permsgn, A_PERMSGN)
KERNEL(k_n_permv, A_PERMV)
KERNEL(k_n_andor_s, A_ANDOR_S)
KERNEL(k_n_bfi_s, A_BFI_S)
KERNEL(k_n_or3, A_OR3)
KERNEL(k_n_lshladd, A_LSHLADD)
KERNEL(k_n_cmp32, A_CMP32)
KERNEL(k_n_cmpsdwa, A_CMPSDWA)
KERNEL(k_n_cmpe64, A_CMPE64)
KERNEL(k_n_pkadd_opsel, A_PKADD_OPSEL)
KERNEL(k_n_pkashr, A_PKASHR)
KERNEL(k_n_pklshr, A_PKLSHR)
KERNEL(k_n_bperm, A_BPERM)
KERNEL(k_n_addlit, A_ADDLIT)
KERNEL(k_n_andlit, A_ANDLIT)
KERNEL(k_n_or32, A_OR32)
KERNEL(k_n_sub32, A_SUB32)
KERNEL(k_n_max32, A_MAX32)
KERNEL(k_n_ashr32, A_ASHR32)
KERNEL(k_n_bfe, A_BFE)
KERNEL(k_n_mbcnt, A_MBCNT)
KERNEL(k_n_xad, A_XAD)
KERNEL(k_n_addf32, A_ADDF32)
KERNEL(k_n_mulf32, A_MULF32)
KERNEL(k_n_addf16, A_ADDF16)
KERNEL(k_n_pkaddf16, A_PKADDF16)
KERNEL(k_n_pkminf16, A_PKMINF16)
KERNEL(k_n_med3, A_MED3)
KERNEL(k_n_min3_32, A_MIN3_32)
KERNEL(k_n_sad8, A_SAD8)
KERNEL(k_n_dot4, A_DOT4)
KERNEL(k_n_addco, A_ADDCO)

#define A_MIX_PK_AND(i)  "v_pk_add_u16 %" #i ", %" #i ", %8\nv_and_b32 %" #i ", %8, %" #i "\n"
#define A_MIX_PK_PK(i)   "v_pk_add_u16 %" #i ", %" #i ", %8\nv_pk_min_i16 %" #i ", %" #i ", %8\n"
#define A_MIX_AND_AND(i) "v_and_b32 %" #i ", %8, %" #i "\nv_or_b32 %" #i ", %8, %" #i "\n"
#define A_MIX_PK_AND2(i) "v_pk_add_u16 %" #i ", %" #i ", %8\nv_and_b32 %" #i ", %8, %" #i "\nv_or_b32 %" #i ", %8, %" #i "\n"
KERNEL(k_mix_pk_and, A_MIX_PK_AND)
KERNEL(k_mix_pk_pk, A_MIX_PK_PK)
KERNEL(k_mix_and_and, A_MIX_AND_AND)
KERNEL(k_mix_pk_and2, A_MIX_PK_AND2)

#define KERNEL_RAW(NAME, BODY)                                                                     \
__global__ void NAME(uint32_t* out, uint32_t seed) {                                               \
    uint32_t v0 = threadIdx.x + seed, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19; \
    uint32_t c = seed | 0x00010001u;                                                               \
    for (int it = 0; it < ITER; ++it) {                                                            \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                            \
            asm volatile(BODY                                                                      \
                : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c)); \
        }                                                                                          \
    }                                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;            \
}
KERNEL_RAW(k_imix_pk_and,
        "v_pk_add_u16 %0, %0, %8\n"
        "v_and_b32 %4, %8, %4\n"
        "v_pk_add_u16 %1, %1, %8\n"
        "v_and_b32 %5, %8, %5\n"
        "v_pk_add_u16 %2, %2, %8\n"
        "v_and_b32 %6, %8, %6\n"
        "v_pk_add_u16 %3, %3, %8\n"
        "v_and_b32 %7, %8, %7\n"
        "v_pk_add_u16 %4, %4, %8\n"
        "v_and_b32 %0, %8, %0\n"
        "v_pk_add_u16 %5, %5, %8\n"
        "v_and_b32 %1, %8, %1\n"
        "v_pk_add_u16 %6, %6, %8\n"
        "v_and_b32 %2, %8, %2\n"
        "v_pk_add_u16 %7, %7, %8\n"
        "v_and_b32 %3, %8, %3\n")
KERNEL_RAW(k_imix_pk_and_or,
        "v_pk_add_u16 %0, %0, %8\n"
        "v_and_b32 %4, %8, %4\n"
        "v_or_b32 %6, %8, %6\n"
        "v_pk_add_u16 %1, %1, %8\n"
        "v_and_b32 %5, %8, %5\n"
        "v_or_b32 %7, %8, %7\n"
        "v_pk_add_u16 %2, %2, %8\n"
        "v_and_b32 %6, %8, %6\n"
        "v_or_b32 %0, %8, %0\n"
        "v_pk_add_u16 %3, %3, %8\n"
        "v_and_b32 %7, %8, %7\n"
        "v_or_b32 %1, %8, %1\n"
        "v_pk_add_u16 %4, %4, %8\n"
        "v_and_b32 %0, %8, %0\n"
        "v_or_b32 %2, %8, %2\n"
        "v_pk_add_u16 %5, %5, %8\n"
        "v_and_b32 %1, %8, %1\n"
        "v_or_b32 %3, %8, %3\n"
        "v_pk_add_u16 %6, %6, %8\n"
        "v_and_b32 %2, %8, %2\n"
        "v_or_b32 %4, %8, %4\n"
        "v_pk_add_u16 %7, %7, %8\n"
        "v_and_b32 %3, %8, %3\n"
        "v_or_b32 %5, %8, %5\n")
KERNEL(k_cnds, A_CNDS)
KERNEL(k_bfi, A_BFI)
KERNEL(k_andor, A_ANDOR)
KERNEL(k_pksubi, A_PKSUBI)
KERNEL(k_pkmini, A_PKMINI)
KERNEL(k_shr32, A_SHR32)
KERNEL(k_and32, A_AND32)
KERNEL(k_xor32, A_XOR32)
KERNEL(k_sub16, A_SUB16)
KERNEL(k_maxi16, A_MAXI16)
KERNEL(k_mov, A_MOV)
KERNEL(k_cmp16, A_CMP16)
KERNEL(k_addc, A_ADDC)
KERNEL(k_alignbit, A_ALIGNBIT)
KERNEL(k_movdpp, A_MOVDPP)
KERNEL(k_adddpp, A_ADDDPP)
KERNEL(k_pkadds, A_PKADD_S)
KERNEL(k_add16s, A_ADD16_S)
KERNEL(k_pk_add, A_PK_ADD)
KERNEL(k_pk_subc, A_PK_SUBC)
KERNEL(k_pk_min, A_PK_MIN)
KERNEL(k_pk_maxi, A_PK_MAXI)
KERNEL(k_add32, A_ADD32)
KERNEL(k_add16, A_ADD16)
KERNEL(k_min16, A_MIN16)
KERNEL(k_min32, A_MIN32)
KERNEL(k_lshlor, A_LSHLOR)
KERNEL(k_perm, A_PERM)
KERNEL(k_cndm, A_CNDM)
KERNEL(k_cmpaddc, A_CMPADDC)
KERNEL(k_subco, A_SUBCO)
KERNEL(k_add3, A_ADD3)
KERNEL(k_sad16, A_SAD16)
KERNEL(k_mad16, A_MAD16)
KERNEL(k_pkmad, A_PKMAD)
KERNEL(k_pkfma16, A_PKFMA16)
KERNEL(k_fma32, A_FMA32)
KERNEL(k_sdwa, A_SDWA)
KERNEL(k_min3, A_MIN3)

// v_permlane32_swap needs two distinct registers
__global__ void k_plswap(uint32_t* out, uint32_t seed) {
    uint32_t v0 = threadIdx.x + seed, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            asm volatile("v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\nv_permlane32_swap_b32 %6, %7\n"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
}

typedef void (*kern_t)(uint32_t*, uint32_t);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t* d;
    CHECK(hipMalloc(&d, (size_t)cus * 4 * 8 * 64 * 4));
    struct K { const char* name; kern_t k; int instr_per_slot; };
    std::vector<K> ks = {{"v_pk_add_u16", k_pk_add, 1}, {"v_pk_sub_u16 clamp", k_pk_subc, 1}, {"v_pk_min_u16", k_pk_min, 1},
                         {"v_pk_max_i16", k_pk_maxi, 1}, {"v_add_u32", k_add32, 1}, {"v_add_u16", k_add16, 1}, {"v_min_u16", k_min16, 1},
                         {"v_min_u32", k_min32, 1}, {"v_lshl_or_b32", k_lshlor, 1}, {"v_perm_b32", k_perm, 1}, {"v_cndmask_b32", k_cndm, 1},
                         {"v_cmp_gt_u16+v_addc_co", k_cmpaddc, 2}, {"v_sub_co_u32", k_subco, 1}, {"v_add3_u32", k_add3, 1},
                         {"v_sad_u16", k_sad16, 1}, {"v_mad_u16", k_mad16, 1}, {"v_pk_mad_u16", k_pkmad, 1}, {"v_pk_fma_f16", k_pkfma16, 1},
                         {"v_fma_f32", k_fma32, 1}, {"v_add_u16_sdwa", k_sdwa, 1}, {"v_min3_u16", k_min3, 1}, {"v_permlane32_swap", k_plswap, 1},
        {"v_cndmask_b32 sgpr-mask", k_cnds, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_and_or_b32", k_andor, 1}, {"v_pk_sub_i16 clamp", k_pksubi, 1},
        {"v_pk_min_i16", k_pkmini, 1}, {"v_lshrrev_b32 (vop2)", k_shr32, 1}, {"v_and_b32", k_and32, 1}, {"v_xor_b32", k_xor32, 1},
        {"v_sub_u16", k_sub16, 1}, {"v_max_i16", k_maxi16, 1}, {"v_mov_b32", k_mov, 1}, {"v_cmp_gt_u16 (vopc)", k_cmp16, 1},
        {"v_addc_co_u32", k_addc, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_mov_b32_dpp", k_movdpp, 1}, {"v_add_u16_dpp", k_adddpp, 1},
        {"v_pk_add_u16 sgpr", k_pkadds, 1}, {"v_add_u16 sgpr", k_add16s, 1},
        {"permsgn", k_n_permsgn, 1},
        {"permv", k_n_permv, 1},
        {"andor_s", k_n_andor_s, 1},
        {"bfi_s", k_n_bfi_s, 1},
        {"or3", k_n_or3, 1},
        {"lshladd", k_n_lshladd, 1},
        {"cmp32", k_n_cmp32, 1},
        {"cmpsdwa", k_n_cmpsdwa, 1},
        {"cmpe64", k_n_cmpe64, 1},
        {"pkadd_opsel", k_n_pkadd_opsel, 1},
        {"pkashr", k_n_pkashr, 1},
        {"pklshr", k_n_pklshr, 1},
        {"bperm", k_n_bperm, 1},
        {"addlit", k_n_addlit, 1},
        {"andlit", k_n_andlit, 1},
        {"or32", k_n_or32, 1},
        {"sub32", k_n_sub32, 1},
        {"max32", k_n_max32, 1},
        {"ashr32", k_n_ashr32, 1},
        {"bfe", k_n_bfe, 1},
        {"mbcnt", k_n_mbcnt, 1},
        {"xad", k_n_xad, 1},
        {"addf32", k_n_addf32, 1},
        {"mulf32", k_n_mulf32, 1},
        {"addf16", k_n_addf16, 1},
        {"pkaddf16", k_n_pkaddf16, 1},
        {"pkminf16", k_n_pkminf16, 1},
        {"med3", k_n_med3, 1},
        {"min3_32", k_n_min3_32, 1},
        {"sad8", k_n_sad8, 1},
        {"dot4", k_n_dot4, 1},
        {"addco", k_n_addco, 1},
        {"mix pk_add+and (per pair)", k_mix_pk_and, 1}, {"mix pk_add+pk_min (per pair)", k_mix_pk_pk, 1},
        {"mix and+or (per pair)", k_mix_and_and, 1}, {"mix pk_add+and+or (per triple)", k_mix_pk_and2, 1},
        {"indep mix pk_add+and (per pair)", k_imix_pk_and, 1}, {"indep mix pk+and+or (per triple)", k_imix_pk_and_or, 1},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-26s %10s %10s %10s   (cycles per wave-instruction per SIMD at 2.4 GHz nominal)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (auto& k : ks) {
        printf("%-26s", k.name);
        for (int w : {1, 2, 4}) {
            // blocks of 256 threads = 4 waves = one per SIMD; w blocks per CU
            const int blocks = cus * w;
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_simd = (double)ITER * 64 * k.instr_per_slot * w * 3;   // per SIMD over 3 launches
            const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
            printf(" %10.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
