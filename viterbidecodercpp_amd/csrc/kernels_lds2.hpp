// kernels_lds2.hpp -- PLAN_LDS2: large constraint lengths (K = 10..16, any polynomials): one workgroup per frame PAIR,
// state metrics of both frames packed in one u32 per state in ONE LDS buffer updated in place, FOUR trellis steps per barrier.
//
// Device implementation of the reference's scalar strategy
//   ViterbiDecoder_Scalar::update / bfly / renormalise   include/viterbi/viterbi_decoder_scalar.h:29-153
//   ViterbiDecoder_Core::chainback                       include/viterbi/viterbi_decoder_core.h:214-236
// bit-identical to PLAN_LDS; this is the plan that makes K = 15 (Cassini, 16384 states) worth running on the GPU.
//
// Mapping (gfx950, wave64), N = 2^(K-1) states in G = N/16 radix-16 groups, one or two groups per thread (Lds2Geom: 512 threads at
// K = 15), JB = K-5 bits of group index j:
//   * LDS: ONE metric buffer of N u32 = (frame A metric | frame B metric << 16), both biased by 0x8000 so that the
//     reference's unsigned compare is a SIGNED compare here (64 KiB at K = 15; updated in place, see lds2_update_body), plus two
//     sets of four 64-entry branch-metric tables {E[p], max_error - E[p]}.
//   * radix-16 block: thread j loads the 16 states [r | j] (r = 0..15 in the TOP four index bits: lane-consecutive dwords,
//     conflict free) and runs FOUR trellis steps on them in registers.  Stage c pairs the registers that differ in bit 3-c
//     (their states differ in the top bit: a butterfly) and writes the two results back in place, so after stage c
//     register r = (r3 r2 r1 r0) holds state [r(2-c)..r0 | j | r3..r(3-c)]; after stage 3 that is 16 j + r: the thread
//     stores its 16 new metrics as four ds_write_b128.  One workgroup barrier and one pass over the metric buffers per
//     FOUR steps (the one-step-per-barrier version spent 57 % of its cycles waiting).
//   * the 4 x 8 branch patterns of a thread are loop invariant: LDS byte offsets into the step's table, one ds_read_b64 per
//     butterfly.  The four tables of block b+1 are built during block b (wavefront c builds step c, lane p makes entry p; the
//     one- and two-wavefront workgroups of K = 10, 11, 12 deal their lanes out over (step, pattern) instead: PASSES).
//   * add-compare-select in packed 16-bit, exact for wrapping metrics: min = v_pk_min_i16, decision = sign of the signed
//     SATURATING difference (strict '>' of the reference: a tie keeps predecessor 0).
//   * decisions: the sign bits of the thread's 16 registers are byte-gathered into one dword per step (frame A in bytes
//     0/2, frame B in bytes 1/3) and stored ws[pair][block of 4 steps][j][step % 4]: the four dwords of a block that group j
//     writes sit side by side -- 16 bytes -- because the chainback needs exactly those four per memory round trip (the group
//     index it reads is the SAME for the four steps of a block, lds2_chainback_kernel), and HBM hands out 64-byte sectors: with
//     one step per row (round 4) every round trip fetched four sectors for 16 bytes of use, 15.5 x the algorithmic bytes at K = 15.
//     lds2_locate() maps (step, state) -> (thread, register), lds2_ws_index() (step, thread) -> dword;
//     vit_hip_export_decisions() converts to the reference bit order.
//   * renormalisation (new_metric[0] >= threshold, scalar.h:48) after the LAST step of a block is the block-uniform rare
//     branch it always was (min over registers -> wave scan -> LDS exchange).  After one of the first three steps only
//     thread 0 can see it (state 0 sits in its register 0): the update is in place, so that case is PREDICTED from a bound on
//     metric[0] and the block then runs stage by stage with the reduction in between (slow_block; about once per 1000 steps).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.hpp"
#include "kernel_desc.hpp"

namespace vit {

typedef uint32_t u32;
typedef unsigned short l2_u16x2 __attribute__((ext_vector_type(2)));
typedef short l2_s16x2 __attribute__((ext_vector_type(2)));

#define VIT_L2 __device__ __forceinline__
VIT_L2 u32 l2_add(u32 a, u32 b) { return __builtin_bit_cast(u32, (l2_u16x2)(__builtin_bit_cast(l2_u16x2, a) + __builtin_bit_cast(l2_u16x2, b))); }
VIT_L2 u32 l2_sub(u32 a, u32 b) { return __builtin_bit_cast(u32, (l2_u16x2)(__builtin_bit_cast(l2_u16x2, a) - __builtin_bit_cast(l2_u16x2, b))); }
VIT_L2 u32 l2_min_s(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(l2_s16x2, a), __builtin_bit_cast(l2_s16x2, b))); }
VIT_L2 u32 l2_max_s(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(l2_s16x2, a), __builtin_bit_cast(l2_s16x2, b))); }
// plain expressions, not inline asm: the hazard recogniser pads every use of an asm statement's result with an s_nop
VIT_L2 u32 l2_sub_sat_s(u32 a, u32 b) {
    return __builtin_bit_cast(u32, __builtin_elementwise_sub_sat(__builtin_bit_cast(l2_s16x2, a), __builtin_bit_cast(l2_s16x2, b)));
}
// (a & mask) | (b & ~mask)
VIT_L2 u32 l2_bfi(u32 mask, u32 a, u32 b) { return __builtin_amdgcn_bitop3_b32(a, b, mask, 0xE4); }   // truth table of mask ? a : b
// "this wavefront's LDS reads of the block's inputs (and its table writes) are done": a RELEASE on the LDS address space in
// front of the counter's add, so that neither the compiler nor the memory model may sink those accesses below it (a relaxed
// add left that to the scheduler's good will).  The fence is restricted to LDS: a generic release would also wait for the
// decision stores to HBM that are still in flight from the previous block (s_waitcnt vmcnt(0)).
#ifndef VIT_L2_ASM_ARRIVE
#define VIT_L2_ASM_ARRIVE 1
#endif
VIT_L2 void lds2_arrive(u32* counter, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
#if VIT_L2_ASM_ARRIVE
    // lane 0 alone, by exec mask (every caller runs whole wavefronts): left to the compiler, `if (lane == 0) atomic add` becomes its
    // wave-wide reduction -- v_mbcnt x 2, v_cmp, a population count, a branch -- for a constant 1
    (void)lane;
    u32 addr = (u32)(uintptr_t)(__attribute__((address_space(3))) u32*)counter;
    u32 one = 1u;
    uint64_t save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(save) : "v"(addr), "v"(one) : "memory");
#else
    if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
// e += d on the lanes of `lanes` only, d wave-uniform: the packed add runs under a temporary exec mask (scalar moves around one
// VALU instruction; the caller's exec mask is all ones or the code is not reached)
VIT_L2 u32 l2_add_where(u32 e, u32 d_uniform, uint64_t lanes) {
    uint64_t save;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %3\n\tv_pk_add_u16 %0, %0, %2\n\ts_mov_b64 exec, %1"
                 : "+v"(e), "=&s"(save)
                 : "s"(d_uniform), "s"(lanes));
    return e;
}
VIT_L2 void l2_opaque(u32& x) { asm volatile("" : "+v"(x)); }   // the compiler may not assume anything about x across this point
// LDS accesses spelled out (VIT_L2_ASM_LS): address register + instruction offset and nothing else.  hipcc's own loads and stores
// want every `base + constant` as a value of its own, hoists those out of the block loop, and the in-loop opaque copies that stop
// it cost ten v_mov per block (the loop-carried addresses are copied into the block and back).  The compiler does not count these
// in lgkmcnt: l2_lds_landed() is the wait, and ties the destination registers to it.
typedef u32 l2_u32x2 __attribute__((ext_vector_type(2)));
typedef u32 l2_u32x4 __attribute__((ext_vector_type(4)));
template <int O0, int O1>                                      // offsets in units of 64 dwords
VIT_L2 l2_u32x2 l2_ds_read2st64(u32 lds_addr) {
    l2_u32x2 v;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(lds_addr), "n"(O0), "n"(O1) : "memory");
    return v;
}
template <int O0, int O1>                                      // offsets in dwords
VIT_L2 l2_u32x2 l2_ds_read2_b32(u32 lds_addr) {
    l2_u32x2 v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(lds_addr), "n"(O0), "n"(O1) : "memory");
    return v;
}
template <int OFF>
VIT_L2 void l2_ds_write_b128(u32 lds_addr, l2_u32x4 v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(lds_addr), "v"(v), "n"(OFF) : "memory");
}
VIT_L2 void l2_lds_landed(l2_u32x2 (&p)[8]) {                   // s_waitcnt lgkmcnt(0); nothing below may read p before it
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : : "memory");
}
#ifndef VIT_L2_ASM_LS
#define VIT_L2_ASM_LS 1
#endif
template <class F, int... Is>
VIT_L2 void l2_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
VIT_L2 void l2_static_for(F&& f) { l2_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct Lds2UpdateArgs {
    const uint8_t* symbols;          // [F][n_steps][R] soft_t
    size_t sym_frame_stride_bytes;
    u32* ws;                         // [pairs][blocks of 4 steps][G groups][4] decision dwords (lds2_ws_index)
    size_t ws_pair_stride;           // ceil4(rows) * G dwords
    void* metrics_out;               // [F][N] error_t or null
    uint64_t* renorm_sum;            // [F] or null
    const u32* start_state;          // [F] or null
    const uint16_t* pattern;         // [H] bit i = (branch_table[i][j] == high)
    const void* metrics_in;          // [F][N] error_t: resume from these metrics (no reset); null => reset(start_state)
    u32 frames;
    u32 t_begin, t_end;              // trellis steps [t_begin, t_end) of every frame; the symbol chunk starts at step t_begin
    int32_t R;
    uint8_t idx_f[4], idx_t[4];      // per block stage: the bank-conflict-free table index map (lds2_tab_index)
    DevConfig cfg;
};

// position of the decision bit of the thread's register r (0..15), frame half h, inside the decision dword of a step that
// ran as stage c of its block: the two registers of butterfly b (r without its bit 3-c) are gathered together as bytes
// {A r0, B r0, A r1, B r1}, bit b of each
__host__ __device__ constexpr u32 lds2_dec_bit(u32 r, u32 h, u32 c) {
    const u32 pb = 3u - c;
    const u32 b = ((r >> (pb + 1u)) << pb) | (r & ((1u << pb) - 1u));
    return b + 8u * h + 16u * ((r >> pb) & 1u);
}

// state held by register r of thread j after stage c of a block (c = -1: as loaded), sbits = K-1
__host__ __device__ constexpr u32 lds2_state_of(int c, u32 r, u32 j, int sbits) {
    const int nlow = c + 1, jb = sbits - 4;
    return ((r & ((1u << (4 - nlow)) - 1u)) << (jb + nlow)) | (j << nlow) | (r >> (4 - nlow));
}
// inverse: the thread and register that hold the decision of NEXT-state s at trellis step t (stage t % 4)
__host__ __device__ inline void lds2_locate(u32 s, u32 t, int sbits, u32& j, u32& r) {
    const int nlow = (int)(t & 3u) + 1, jb = sbits - 4;
    j = (s >> nlow) & ((1u << jb) - 1u);
    r = ((s & ((1u << nlow) - 1u)) << (4 - nlow)) | (s >> (jb + nlow));
}

// dword of (trellis step t, group j) inside a frame pair's workspace: blocks of four steps, [block][j][t % 4]; g = groups per step
// (VIT_L2_WS_GROUPED=0 builds round 4's [step][j] rows for same-box comparisons, scripts/gpu_ab_l2.sh)
#ifndef VIT_L2_WS_GROUPED
#define VIT_L2_WS_GROUPED 1
#endif
__host__ __device__ inline size_t lds2_ws_index(size_t t, size_t j, size_t g) {
    return VIT_L2_WS_GROUPED ? (t >> 2) * (4 * g) + j * 4 + (t & 3) : t * g + j;
}

// LDS index of natural state s inside a metric buffer of n states.  The natural order makes the four ds_write_b128 of a
// thread's 16 new states (64-byte lane stride) 4-way bank conflicts; this order keeps BOTH access patterns of a block
// conflict free: write view s = 16 J + 4 q + i  ->  piece q of thread J is 16 contiguous bytes at q*n/4 + 4*(J ^ 2q)
// (8 consecutive J cover all 32 banks), read view s = r*T + j -> for 32 consecutive j the low bits (i, q, J0) still map to
// 32 different banks because q is folded into bits 1-2 of J.  sw(0) = 0.
__host__ __device__ constexpr u32 lds2_sw(u32 s, u32 n) {
    const u32 q = (s >> 2) & 3u;
    return q * (n / 4u) + (((s >> 4) ^ (q << 1)) << 2) + (s & 3u);
}

// Position of pattern p inside a step's 64-entry branch-metric table.  A ds_read_b64 serves lanes 0-31 and 32-63 in one LDS
// cycle each if no two lanes of a group need DIFFERENT entries that share banks, i.e. entries whose positions differ only
// in bit 5.  The patterns the 32 lanes of a group ask for in stage c form a coset of the linear space V_c spanned by the
// patterns of five state bits, so a position map that is linear in p and sends V_c into {bit 5 = 0} makes every table read
// conflict free: bit 5 = parity(p & f) for an f orthogonal to V_c (one exists: dim V_c <= 5 < 6), bits 0-4 = p without its
// bit t, where t is a set bit of f (which makes the map a bijection).  In natural order a third of the read cycles were
// bank conflicts (SQ_LDS_BANK_CONFLICT 11 % of all LDS cycles).
__host__ __device__ constexpr u32 lds2_tab_index(u32 p, u32 f, u32 t) {
    const u32 low = (p & ((1u << t) - 1u)) | ((p >> (t + 1u)) << t);
    u32 x = p & f;
    x ^= x >> 4; x ^= x >> 2; x ^= x >> 1;
    return low | ((x & 1u) << 5);
}

template <int K>
struct Lds2Geom {
    static constexpr int SBITS = K - 1, N = 1 << SBITS, H = N / 2;
    static constexpr int G = N / 16;                       // radix-16 groups (= decision dwords per trellis step and frame pair)
    static constexpr int JB = SBITS - 4;                   // bits of a group index
    static constexpr int GPT = G >= 256 ? 2 : 1;           // groups per thread (K >= 13: two, A = tid and B = tid + T)
    // thread slots that own groups.  K = 10 has 32 of them: the workgroup is still one full wavefront (the table build wants 64
    // lanes, one per entry) and lanes 32 - 63 DUPLICATE lanes 0 - 31 -- same loads, same arithmetic, same stores of the same values
    // to the same addresses -- so that no path of the kernel needs a lane predicate; half the vector unit idles through the
    // add-compare-select, which still leaves K = 10 at half the per-state rate of its neighbours instead of a twentieth (PLAN_LDS)
    static constexpr int TG = G / GPT;
    static constexpr int T = TG < 64 ? 64 : TG, NW = T / 64;
    static constexpr int BLK = 4;                          // trellis steps per block
    static constexpr int CPW = (BLK + NW - 1) / NW;        // tables a wavefront builds per block
    // K = 15: 512 threads and 72 KiB of LDS per workgroup, so that TWO workgroups share a CU (4 waves per SIMD, 128 VGPRs);
    // K = 16: 1024 threads and 136 KiB, one workgroup per CU -- what the in-place update makes possible at all
    static constexpr int MINW = K >= 15 ? 4 : 2;
    // frame pairs per workgroup.  K = 10 has 32 group slots: its one wavefront decodes TWO frame pairs, lanes 0 - 31 the first and lanes
    // 32 - 63 the second, each half with its own tables, control words and metric buffer (round 5: the upper half mirrored the lower one
    // through every load, butterfly and store -- half the vector unit idle, half of K = 11's rate per state update)
    static constexpr int PPW = TG < 64 ? 2 : 1;
    static constexpr size_t tab_bytes = (size_t)2 * BLK * GPT * 64 * 8;
    static constexpr size_t smem_bytes = (size_t)PPW * (tab_bytes + 32 * 4 + (size_t)N * 4);
    static_assert(TG >= 32 && T <= 1024, "PLAN_LDS2 serves K = 10..16");
};

// One workgroup per frame pair.  The N packed metrics live in ONE LDS buffer that a block of four trellis steps updates IN
// PLACE: every thread loads the 16 states of each of its radix-16 groups, all threads pass barrier B1 (nobody reads the old
// values after it), four stages run in registers, the 16 new states of each group are stored, barrier B2.  Half the LDS of
// a double-buffered block, so that two workgroups fit a CU and one computes while the other waits at its barriers (the
// one-workgroup-per-CU build spent 16 % of its time in the barrier: 58.0 ms with, 49.0 ms without, 4096 x 8192 at K = 15).
// In place means a block cannot be re-run: the rare mid-block renormalisation (state 0 reaching the threshold after one of
// the first three steps) is therefore PREDICTED, not discovered -- new[0] <= old[0] + E[pattern 0], so thread 0 bounds
// metric[0] over the next block's first three steps from that block's (already built) tables and, if the bound reaches the
// threshold, the whole workgroup takes the careful stage-by-stage routine for that block.
// RT: the code rate as a compile-time constant (the Cassini instantiation: the six per-symbol loops lose their run-time bound
// checks and the phi copies around them), or 0 = taken from the arguments (any R <= 6).
template <int K, int SHIFT, int RT = 0>
__device__ __forceinline__ void lds2_update_body(Lds2UpdateArgs a) {
    using GM = Lds2Geom<K>;
    constexpr int N = GM::N, T = GM::T, G = GM::G, GPT = GM::GPT, NW = GM::NW, BLK = GM::BLK, SBITS = GM::SBITS;
    constexpr u32 BIAS2 = 0x80008000u;
    constexpr u32 STEP_TAB = (u32)GPT * 512u;              // bytes of one step's tables (group-A table, group-B table)
    constexpr u32 SET_TAB = (u32)BLK * STEP_TAB;           // bytes of one table set (the four steps of a block)
    extern __shared__ __attribute__((aligned(16))) u32 lds2_smem[];
    constexpr int PPW = GM::PPW;                               // frame pairs per workgroup (2 at K = 10: one per half of the wavefront)
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 half = PPW == 2 ? (u32)lane >> 5 : 0u;           // which of the workgroup's pairs this lane works for
    const int ls = PPW == 2 ? (lane & 31) : tid;               // the lane's index among its pair's thread slots ...
    constexpr int LT = PPW == 2 ? 32 : T;                      // ... of which there are this many
    // layout: [tables of pair 0 | tables of pair 1][control words 0 | 1][metrics 0 | 1] -- everything per pair at `half` strides
    // tables first: their byte offsets (< 8 KiB) fit instruction offsets and 16-bit halves of a register
    uint2* const etab = (uint2*)((char*)lds2_smem + half * (u32)GM::tab_bytes);   // [2 sets][BLK][GPT][64] {E, max_error - E}
    u32* const wmin = (u32*)((char*)lds2_smem + (size_t)PPW * GM::tab_bytes + half * 128u); // [16]
    u32* const flag = wmin + 16;                               // [0] careful routine wanted for the next block; [2] scratch; [4] arrivals
    u32* const arrive = (u32*)((char*)lds2_smem + (size_t)PPW * GM::tab_bytes) + 16 + 4;   // wavefronts that have loaded their block's metrics, running total (one counter per workgroup)
    uint64_t* const rs_acc = (uint64_t*)(flag + 8);            // [2] sum of subtracted minima, frame A / B (thread 0 only)
    u32* const sig = flag + 12;                                // [4] careful blocks: thread 0's per-stage "renormalise" word (slow_block)
    u32 careful_seq = 0;                                       // careful stages run so far: the same count in every wavefront
    u32* const met = (u32*)((char*)lds2_smem + (size_t)PPW * (GM::tab_bytes + 128u) + half * (u32)(N * 4));   // [N], 16-byte aligned

    const int gid = tid & (GM::TG - 1);          // the thread's group slot (== tid except at K = 10, where lanes 32 - 63 mirror 0 - 31)
    // the wavefront's index, pinned uniform: everything derived from it (which table it builds, whether it builds one at all)
    // is then scalar control flow and scalar data, not exec-mask branches over vector compares
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the step range, pinned uniform up front: read where it is needed, the compiler re-loads it on one side of a per-thread
    // branch (tid == 0 ...) and merges the copies in a phi that counts as divergent -- and with it the block loop's counter,
    // its table-set bit and every branch on them (vector compares, exec-mask branches, loop counters in VGPRs)
    const u32 t_begin = (u32)__builtin_amdgcn_readfirstlane((int)a.t_begin), t_end = (u32)__builtin_amdgcn_readfirstlane((int)a.t_end);
    // (the second pair of the last workgroup may not exist: its lanes then decode the last pair again -- the same values to the same
    // addresses, like the mirrored half of round 5 -- so that no path needs a lane predicate)
    const u32 npairs = (a.frames + 1u) / 2u;
    const u32 pair = PPW == 2 ? (blockIdx.x * 2u + half < npairs ? blockIdx.x * 2u + half : npairs - 1u) : blockIdx.x;
    const u32 fA = 2 * pair;
    const bool validB = fA + 1 < a.frames;
    const u32 fB = validB ? fA + 1 : fA;
    const int R = RT ? RT : a.R;

    const u32 HIGH2 = (u32)(uint16_t)a.cfg.high * 0x10001u, LOW2 = (u32)(uint16_t)a.cfg.low * 0x10001u;
    const u32 MAXE2 = (u32)a.cfg.max_error * 0x10001u;
    const u32 THRM1B2 = ((u32)(uint16_t)(a.cfg.threshold - 1) * 0x10001u) ^ BIAS2;   // metric >= thr  <=>  biased > this
    const u32 FORCE = a.cfg.threshold == 0 ? BIAS2 : 0u;

    // ---- loop-invariant per-thread constants: table byte offset of the branch pattern of each of my 4 x 8 group-A
    // butterflies (two 16-bit LDS byte offsets per register).  Group B = group A + T differs in ONE bit of the butterfly
    // index, and the code is linear (lds2_supported()), so its pattern is the group-A pattern ^ pattern[that bit]: the
    // builder writes a second table per step with the entries permuted by that XOR, and group B reads it through the same
    // offsets (+512: an instruction offset).
    u32 prow2[BLK][4];
    l2_static_for<BLK>([&](auto cc) __attribute__((always_inline)) {
        constexpr int C = decltype(cc)::value, PB = 3 - C;
        l2_static_for<4>([&](auto hc) __attribute__((always_inline)) {
            constexpr int h2 = decltype(hc)::value;
            u32 pk = 0;
            l2_static_for<2>([&](auto ec) __attribute__((always_inline)) {
                constexpr int h = 2 * h2 + decltype(ec)::value;
                constexpr u32 r0 = ((h >> PB) << (PB + 1)) | (h & ((1 << PB) - 1));
                const u32 lower = lds2_state_of(C - 1, r0, (u32)gid, SBITS);   // top bit clear: butterfly index < H
                // byte offset from lds2_smem of this butterfly's entry in table set 0 (set 1: + SET_TAB, an instruction offset)
                const u32 off = half * (u32)GM::tab_bytes + (u32)C * STEP_TAB + lds2_tab_index((u32)a.pattern[lower] & 63u, a.idx_f[C], a.idx_t[C]) * 8u;
                pk |= off << (16 * decltype(ec)::value);
            });
            prow2[C][h2] = pk;
        });
    });

    // ---- reset (viterbi_decoder_core.h:202-211), or resume from the metrics an earlier call left ----
    if (a.metrics_in) {
        for (int s = ls; s < N; s += LT) {
            u32 lo, hi;
            if (SHIFT) {
                lo = (u32)((const uint8_t*)a.metrics_in)[(size_t)fA * N + s] << 8;
                hi = (u32)((const uint8_t*)a.metrics_in)[(size_t)fB * N + s] << 8;
            } else {
                lo = ((const uint16_t*)a.metrics_in)[(size_t)fA * N + s];
                hi = ((const uint16_t*)a.metrics_in)[(size_t)fB * N + s];
            }
            met[lds2_sw((u32)s, (u32)N)] = (lo | (hi << 16)) ^ BIAS2;
        }
    } else {
        const u32 sA = a.start_state ? (a.start_state[fA] & (u32)(N - 1)) : 0u;
        const u32 sB = a.start_state ? (a.start_state[fB] & (u32)(N - 1)) : 0u;
        for (int s = ls; s < N; s += LT) {
            const u32 lo = ((u32)s == sA) ? a.cfg.init_start : a.cfg.init_non_start;
            const u32 hi = ((u32)s == sB) ? a.cfg.init_start : a.cfg.init_non_start;
            met[lds2_sw((u32)s, (u32)N)] = (lo | (hi << 16)) ^ BIAS2;
        }
    }
    if (ls < 16) flag[ls] = 0;                // flags and the two 64-bit renormalisation sums

    // ---- branch-metric table builder: lane p makes entry p of the tables of one step ----
    const uint8_t* symA = a.symbols + (size_t)fA * a.sym_frame_stride_bytes;
    const uint8_t* symB = a.symbols + (size_t)fB * a.sym_frame_stride_bytes;
    auto load_syms = [&](u32 abs_step, u32 (&y)[6]) __attribute__((always_inline)) {
        // packed (frame A | frame B << 16) symbols of trellis step `abs_step` in the device's 16-bit domain
        const u32 step = abs_step - t_begin;      // the chunk starts at step t_begin
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < R) {
                u32 ya, yb;
                if (SHIFT) {
                    ya = (u32)symA[(size_t)step * R + i] << 8;
                    yb = (u32)symB[(size_t)step * R + i] << 8;
                } else {
                    ya = ((const uint16_t*)symA)[(size_t)step * R + i];
                    yb = ((const uint16_t*)symB)[(size_t)step * R + i];
                }
                y[i] = ya | (yb << 16);
            }
        }
    };
    // pattern of the butterfly-index bit that separates group B from group A in stage c: bit JB-1+c
    u32 xorB[BLK] = {0, 0, 0, 0};
    if constexpr (GPT == 2) {
#pragma unroll
        for (int c = 0; c < BLK; ++c) xorB[c] = (u32)a.pattern[(size_t)1 << (GM::JB - 1 + c)] & 63u;
    }
    // (An all-scalar build -- the symbols are wave-uniform -- was measured SLOWER in round 2: 14 VALU per table pair instead of 70, but
    // 250 dependent scalar instructions in front of the building wavefronts' add-compare-select; they reached the block's barrier late.)
    // lane p: E[p] = sum_i |bt_i - y_i| with bt_i = high where bit i of p is set  (scalar.h:66-73); EB = max_error - E (:107),
    // stored at position lds2_tab_index(p): 70 VALU per table pair, on four of the eight wavefronts
    // `high` and `low` live in VECTOR registers for the build: the symbols are in SGPRs and a packed subtract takes only one
    // scalar operand -- with both in SGPRs every symbol cost a v_mov first
    // (only the compile-time-rate instantiation has the two registers to spare)
    u32 HIGH2v = HIGH2, LOW2v = LOW2;
    if constexpr (false) {      // (round 3: RT != 0; the two registers now hold the decision gather's masks, see M55 / M33 below)
        l2_opaque(HIGH2v);
        l2_opaque(LOW2v);
    }
    // Table B (read by the threads' second groups) holds E[p ^ xb] where table A holds E[p] -- a PERMUTATION of the same 64 values:
    // lane p computes E[p] once and stores it twice, at its place in table A and at the place of entry p ^ xb in table B (round 3
    // summed both tables' entries per lane: 6 selects and 6 adds more per table pair)
#ifndef VIT_L2_SCALAR_BUILD
#define VIT_L2_SCALAR_BUILD 0
#endif
    auto build_table = [&](uint2* tab, const u32 (&y)[6], u32 pos, u32 posB, bool valid = true, u32 pat = 0u) __attribute__((always_inline)) {
        u32 e = 0;
        if constexpr (VIT_L2_SCALAR_BUILD != 0) {
            // EXPERIMENT (-DVIT_L2_SCALAR_BUILD=1): the symbols are wave-uniform, so what does not depend on the lane runs on the scalar
            // unit -- per symbol and frame b = |low - y|, a = |high - y| (wrapping in 16 bits like soft_t), the base sum of the b and the
            // deltas a - b -- and lane p adds the deltas of its set bits: ONE packed add per symbol under an exec mask, 8 VALU per
            // table where the per-lane form takes 34, against ~20 scalar instructions per symbol in front of them
            const int hi16 = (int)a.cfg.high, lo16 = (int)a.cfg.low;
            auto absd = [](int expected, int sym) __attribute__((always_inline)) -> u32 {
                const int d = (int)(int16_t)(uint16_t)(u32)(expected - sym);           // soft_t(expected - sym), wrapping
                return (u32)(d < 0 ? -d : d);                                          // |-32768| = 0x8000 in the low 16 bits
            };
            u32 sA = 0, sB = 0, dl[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (i < R) {
                    const int yA = (int)(int16_t)(uint16_t)(y[i] & 0xFFFFu), yB = (int)y[i] >> 16;
                    const u32 aA = absd(hi16, yA), bA = absd(lo16, yA), aB = absd(hi16, yB), bB = absd(lo16, yB);
                    sA += bA;
                    sB += bB;
                    dl[i] = ((aA - bA) & 0xFFFFu) | ((aB - bB) << 16);
                }
            }
            const u32 base = (sA & 0xFFFFu) | (sB << 16);
            uint64_t save;
            // lanes with bit i of the lane number set: 0xAAAA.., 0xCCCC.., 0xF0F0.., 0xFF00.., 0xFFFF0000.., the upper half
            asm volatile(
                "s_mov_b64 %[sv], exec\n\t"
                "v_mov_b32 %[e], %[b]\n\t"
                "s_mov_b32 exec_lo, 0xaaaaaaaa\n\ts_mov_b32 exec_hi, 0xaaaaaaaa\n\tv_pk_add_u16 %[e], %[e], %[d0]\n\t"
                "s_mov_b32 exec_lo, 0xcccccccc\n\ts_mov_b32 exec_hi, 0xcccccccc\n\tv_pk_add_u16 %[e], %[e], %[d1]\n\t"
                "s_mov_b32 exec_lo, 0xf0f0f0f0\n\ts_mov_b32 exec_hi, 0xf0f0f0f0\n\tv_pk_add_u16 %[e], %[e], %[d2]\n\t"
                "s_mov_b32 exec_lo, 0xff00ff00\n\ts_mov_b32 exec_hi, 0xff00ff00\n\tv_pk_add_u16 %[e], %[e], %[d3]\n\t"
                "s_mov_b32 exec_lo, 0xffff0000\n\ts_mov_b32 exec_hi, 0xffff0000\n\tv_pk_add_u16 %[e], %[e], %[d4]\n\t"
                "s_mov_b32 exec_lo, 0\n\ts_mov_b32 exec_hi, -1\n\tv_pk_add_u16 %[e], %[e], %[d5]\n\t"
                "s_mov_b64 exec, %[sv]"
                : [e] "=&v"(e), [sv] "=&s"(save)
                : [b] "s"(base), [d0] "s"(dl[0]), [d1] "s"(dl[1]), [d2] "s"(dl[2]), [d3] "s"(dl[3]), [d4] "s"(dl[4]), [d5] "s"(dl[5]));
            const uint2 v = make_uint2(e, l2_sub(MAXE2, e));
            tab[pos] = v;
            if constexpr (GPT == 2) tab[64 + posB] = v;
            return;
        }
        // the lane number is formed again here (two instructions, pinned): kept across the block it is one of the two values the
        // 120-register cap sends to scratch, reloaded behind a full vmcnt(0) right in front of the build
        u32 ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        l2_opaque(ln);
        if constexpr (PPW == 2) ln = pat;       // the pass form with 32 lanes per pair: the lane's pattern is not its lane number's low bits (R = 6)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < R) {
                // (the two-groups-per-thread capped instantiations other than Cassini's 16-bit one -- K = 14, K = 15 at other rates or 8-bit --
                // keep the nine-instruction form: the select-first form holds one more value across the symbol loop and they would
                // spill 12 - 16 bytes under their 120-register cap)
                constexpr bool SELECT_FIRST = !(K == 14 || K == 15) || (RT != 0 && SHIFT == 0);
                if constexpr (!SELECT_FIRST) {
                const u32 d1 = l2_sub(HIGH2v, y[i]), d0 = l2_sub(LOW2v, y[i]);
                const u32 a1 = l2_max_s(d1, l2_sub(0u, d1)), a0 = l2_max_s(d0, l2_sub(0u, d0));
                // select by an all-ones / all-zeros mask of the lane's pattern bit (v_bfe_i32 + v_bitop3) instead of v_cmp + v_cndmask
                const u32 mk = (u32)__builtin_amdgcn_sbfe((int)ln, (u32)i, 1u);
                e = l2_add(e, __builtin_amdgcn_bitop3_b32(a1, a0, mk, 0xE4));
                } else {
                // the lane's EXPECTED symbol first (high where bit i of its pattern is set: an all-ones / all-zeros mask of the bit
                // picks it, v_bfe_i32 + v_bitop3), then ONE absolute difference: six instructions per symbol where forming both
                // |high - y| and |low - y| and selecting afterwards took nine (round 4)
                const u32 mk = (u32)__builtin_amdgcn_sbfe((int)ln, (u32)i, 1u);
                const u32 c = __builtin_amdgcn_bitop3_b32(HIGH2v, LOW2v, mk, 0xE4);
                const u32 d = l2_sub(c, y[i]);
                e = l2_add(e, l2_max_s(d, l2_sub(0u, d)));
                }
            }
        }
        const uint2 v = make_uint2(e, l2_sub(MAXE2, e));
        if (valid) {                             // (always true but in the pass form below, where a lane's step may lie outside the chunk)
            tab[pos] = v;
            if constexpr (GPT == 2) tab[64 + posB] = v;
        }
    };
    // wavefront w builds the tables of block steps c = w, w + NW, ... (< BLK); symbols are fetched one block ahead
    constexpr int CPW = GM::CPW;
    // The symbols of a step are the same for every lane: they land in VGPRs (vector loads, issued just before a block's
    // closing barrier) and are moved to SGPRs right behind it (tables_commit), so that no vector register carries them through
    // the block's compute phase -- at 128 registers per thread every one of them was a spill.
    u32 yland[(NW <= 2 && RT == 0) ? 1 : CPW][6];   // lds2_supported(): R <= 6 (one-wavefront workgroups: pass 0 only, see PASSES)
    u32 ysym[CPW][6];                        // wave-uniform
#pragma unroll
    for (int i = 0; i < CPW; ++i)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (i < (int)(sizeof(yland) / sizeof(yland[0]))) yland[i][k] = 0;
            ysym[i][k] = 0;
        }
    u32 tabpos[CPW];                         // where this lane's entry goes in the tables this wavefront builds (lds2_tab_index)
    u32 tabxorB[CPW];                        // ... and, XORed onto it, where it goes in table B: the map is linear, index(p ^ xb) = index(p) ^ index(xb); wave-uniform
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = (wave + NW * i) & (BLK - 1);
        tabpos[i] = lds2_tab_index((u32)lane, a.idx_f[c], a.idx_t[c]);
        tabxorB[i] = (u32)__builtin_amdgcn_readfirstlane((int)lds2_tab_index(xorB[c], a.idx_f[c], a.idx_t[c]));
    }
    // Wavefronts beyond the block's four steps (K = 15, 16: eight / sixteen wavefronts) build no table.  Their symbol LOADS stay
    // (a load under a condition turns the landing registers into phis of old and new values, live across the block: the 120-
    // register cap then spills), but with a compile-time even rate and 16-bit symbols (RAW: the Cassini instantiation) everything
    // behind the loads -- broadcasting the six raw dwords to SGPRs and pairing frame A's and frame B's symbols there, on the SCALAR
    // unit (s_pack_ll / s_pack_hh) -- happens on the building wavefronts only; round 3 packed in vector registers (14 instructions)
    // and broadcast (6) on all eight.
    constexpr bool RAW = RT != 0 && SHIFT == 0 && RT % 2 == 0;
    // ONE-WAVEFRONT workgroups (K = 10, 11) build all four tables of a block themselves -- a 64-entry table per step, of which a
    // rate-1/2 code uses four entries.  There the lanes are dealt out over (step, pattern) instead: with E = max(16, 2^R) entries per
    // table a pass of the build serves 64 / E steps at once -- all four at R <= 4, two at R = 5 -- and the block needs 1, 2 or 4 passes
    // (R = 6: the old one table per pass).  A lane's step differs from its neighbours' now, so the symbols stay in VECTOR registers
    // (each lane loads its own step's: no broadcast to SGPRs) and a lane whose step lies outside the chunk skips its store.
    // (K = 12, two wavefronts of two steps each, does the same with E >= 32: one pass up to R = 5)
    constexpr bool PASSES = NW <= 2 && RT == 0;                     // K = 10, 11, 12 and (two groups per thread: a table B per step) 13
    constexpr int SPW = BLK / (NW <= 2 ? NW : 1);                   // steps a wavefront serves: 4, 2
    constexpr int LGE_MIN = PPW == 2 ? 3 : SPW == 4 ? 4 : 5;        // 64 / SPW lanes per step at most (two pairs per wavefront: 32 / SPW)
    const int lgE = R <= LGE_MIN ? LGE_MIN : R;                     // wave-uniform
    constexpr int LGLW = PPW == 2 ? 5 : 6;                          // log2 of the lanes that build one pair's tables
    const int n_pass = PASSES ? (lgE <= LGLW ? SPW >> (LGLW - lgE) : SPW << (lgE - LGLW)) : 0;   // 1, 2, 4 (two pairs per wavefront: 2, 4, 8)
    // the block step this lane serves in pass j is pass_step(j) = j * (64 / E) + lane / E; where its entry goes: pass_pos(j) -- kept in
    // a register for pass 0 (the only pass of every code with R <= 4), formed again in the later passes of an R = 5, 6 code (four
    // more loop-invariant registers send the kernel to scratch, with a reload behind a full vmcnt(0) in every block)
    // entry e = j * (lanes per pair) + lane index of the wavefront's SPW * E entries: step e / E, pattern e % E
    u32 pass_c0 = (u32)lane >> lgE;
    auto pass_step = [&](int j) __attribute__((always_inline)) -> u32 {
        if constexpr (PPW == 2) return ((((u32)j << LGLW) + (u32)ls) >> lgE) & (u32)(BLK - 1);
        return ((u32)(wave * SPW) + ((u32)j << (6 - lgE)) + pass_c0) & (u32)(BLK - 1);
    };
    auto pass_pat = [&](int j) __attribute__((always_inline)) -> u32 {
        if constexpr (PPW == 2) return (((u32)j << LGLW) + (u32)ls) & ((1u << lgE) - 1u);
        return (u32)lane & ((1u << lgE) - 1u);
    };
    auto pass_pos = [&](int j) __attribute__((always_inline)) -> u32 {
        const u32 cc = pass_step(j);
        const u32 f = cc == 0 ? a.idx_f[0] : cc == 1 ? a.idx_f[1] : cc == 2 ? a.idx_f[2] : a.idx_f[3];
        const u32 t = cc == 0 ? a.idx_t[0] : cc == 1 ? a.idx_t[1] : cc == 2 ? a.idx_t[2] : a.idx_t[3];
        // entry of the lane's pattern inside the table of ITS step (tables of a set: BLK x GPT x 64 entries)
        return cc * (u32)(GPT * 64) + lds2_tab_index(pass_pat(j), f, t);
    };
    // ... and, XORed onto it, where the same sum goes in the step's table B (two groups per thread): index(p ^ xb) = index(p) ^ index(xb)
    auto pass_xorB = [&](int j) __attribute__((always_inline)) -> u32 {
        const u32 cc = pass_step(j);
        const u32 f = cc == 0 ? a.idx_f[0] : cc == 1 ? a.idx_f[1] : cc == 2 ? a.idx_f[2] : a.idx_f[3];
        const u32 t = cc == 0 ? a.idx_t[0] : cc == 1 ? a.idx_t[1] : cc == 2 ? a.idx_t[2] : a.idx_t[3];
        const u32 xb = cc == 0 ? xorB[0] : cc == 1 ? xorB[1] : cc == 2 ? xorB[2] : xorB[3];
        return lds2_tab_index(xb, f, t);
    };
    if constexpr (PASSES) {
        tabpos[0] = pass_pos(0);
        if constexpr (GPT == 2) tabxorB[0] = pass_xorB(0);          // per lane here (the other form keeps it wave-uniform)
    }
    const bool builder = wave < BLK;                 // wave-uniform (wave comes from v_readfirstlane)
    auto tables_load = [&](u32 t0) __attribute__((always_inline)) {     // symbols of block starting at t0
        if constexpr (PASSES) {                                     // pass 0 only: the later passes of an R = 5, 6 code fetch when they run
            u32 ts = t0 + pass_step(0);
            ts = ts < t_begin ? t_begin : ts;
            ts = ts >= t_end ? t_end - 1u : ts;
            load_syms(ts, yland[0]);
            return;
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            // unconditional, from a step clamped into the chunk [t_begin, t_end): what a clamped step fetches is never used
            // (tables_build skips the steps outside the range)
            u32 ts = t0 + (u32)((wave + NW * i) & (BLK - 1));
            ts = ts < t_begin ? t_begin : ts;
            ts = ts >= t_end ? t_end - 1u : ts;
            if constexpr (RAW) {
                const size_t off = (size_t)(ts - t_begin) * (size_t)(RT * 2);
#pragma unroll
                for (int d = 0; d < RT / 2; ++d) {
                    __builtin_memcpy(&yland[i][d], symA + off + 4 * d, 4);          // two 16-bit symbols of frame A (2-byte aligned)
                    __builtin_memcpy(&yland[i][3 + d], symB + off + 4 * d, 4);      // ... and of frame B
                }
            } else {
                load_syms(ts, yland[i]);
            }
        }
    };
    auto tables_commit = [&]() __attribute__((always_inline)) {
        if constexpr (PASSES) return;                               // the symbols are per lane: nothing to broadcast
        if constexpr (RAW) {
            if (NW > BLK && !builder) return;        // only SGPRs are written here
#pragma unroll
            for (int i = 0; i < CPW; ++i)
#pragma unroll
                for (int d = 0; d < RT / 2; ++d) {
                    const u32 ra = (u32)__builtin_amdgcn_readfirstlane((int)yland[i][d]);
                    const u32 rb = (u32)__builtin_amdgcn_readfirstlane((int)yland[i][3 + d]);
                    ysym[i][2 * d] = (ra & 0xFFFFu) | (rb << 16);                    // (frame A | frame B << 16), symbol 2d
                    ysym[i][2 * d + 1] = (ra >> 16) | (rb & 0xFFFF0000u);            // symbol 2d + 1
                }
        } else {
#pragma unroll
            for (int i = 0; i < CPW; ++i)
#pragma unroll
                for (int k = 0; k < 6; ++k) ysym[i][k] = (u32)__builtin_amdgcn_readfirstlane((int)yland[i][k]);
        }
    };
    auto tables_build = [&](u32 t0, int set) __attribute__((always_inline)) {   // from the symbols loaded by tables_load(t0)
        if constexpr (PASSES) {
            {
                const u32 ts = t0 + pass_step(0);
                build_table(etab + (size_t)(set * BLK) * GPT * 64, yland[0], tabpos[0], tabpos[0] ^ tabxorB[0], ts < t_end && ts >= t_begin, pass_pat(0));
            }
            // R = 5, 6: one or three more passes, each fetching its symbols on the spot (the other wavefronts of the SIMD cover the
            // wait; symbols of four passes held a block ahead are 18 registers the 120-register kernel does not have) -- a real loop,
            // so that nothing of a later pass becomes a loop-invariant register of the block loop
#pragma nounroll
            for (int j = 1; j < n_pass; ++j) {
                const u32 ts = t0 + pass_step(j);
                const bool valid = ts < t_end && ts >= t_begin;
                u32 tc = ts < t_begin ? t_begin : ts;
                tc = tc >= t_end ? t_end - 1u : tc;
                u32 y[6] = {0, 0, 0, 0, 0, 0};
                load_syms(tc, y);
                const u32 pj = pass_pos(j);
                build_table(etab + (size_t)(set * BLK) * GPT * 64, y, pj, GPT == 2 ? pj ^ pass_xorB(j) : 0u, valid, pass_pat(j));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int c = wave + NW * i;
            if (c < BLK && t0 + (u32)c < t_end && t0 + (u32)c >= t_begin)
                build_table(etab + (size_t)(set * BLK + c) * GPT * 64, ysym[i], tabpos[i], tabpos[i] ^ tabxorB[i]);
        }
    };
    // thread 0: could state 0 reach the threshold after one of the first three steps of the block at t_next, whose tables
    // are in `set`?  new[0] <= old[0] + E[pattern of butterfly 0 = 0] (wrapping or not), so old[0] + E0 + E1 + E2 bounds it.
    auto predict = [&](u32 m0_biased, u32 t_next, int set) __attribute__((always_inline)) -> u32 {
        const u32 m0 = m0_biased ^ BIAS2;
        u32 lo = m0 & 0xFFFFu, hi = m0 >> 16;
#pragma unroll
        for (int c = 0; c < BLK - 1; ++c) {
            if (t_next + (u32)c < t_end) {
                const u32 e = etab[(size_t)(set * BLK + c) * GPT * 64].x;
                lo += e & 0xFFFFu;
                hi += e >> 16;
            }
        }
        const u32 thr = a.cfg.threshold;
        return (lo >= thr || hi >= thr) ? 1u : 0u;      // threshold 0: always
    };
    const u32 tb0 = t_begin & ~(u32)(BLK - 1);   // the block that holds step t_begin (0 for a fresh decode)
    tables_load(tb0);
    tables_commit();
    tables_build(tb0, 0);
    tables_load(tb0 + BLK);
    tables_commit();
    __syncthreads();
    if (ls == 0) flag[0] = predict(met[0], tb0, 0);
    __syncthreads();
    // "the next block runs the careful routine": block-uniform; with two pairs per wavefront, if EITHER pair's prediction says so
    auto read_careful = [&]() __attribute__((always_inline)) -> u32 {
        const u32 v = flag[0];
        if constexpr (PPW == 2) return (u32)__builtin_amdgcn_readlane((int)v, 0) | (u32)__builtin_amdgcn_readlane((int)v, 32);
        return (u32)__builtin_amdgcn_readfirstlane((int)v);
    };

    u32* const ws_pair = a.ws + (size_t)pair * a.ws_pair_stride;

    u32 mA[16], mB[GPT == 2 ? 16 : 1];
    // the bit-field-insert masks of the decision gather in VECTOR registers: v_bitop3_b32 with three VGPR operands issues in 2.7
    // clocks, with a scalar operand in 4.7 (profiles/r4_op_rates.txt); opaque, or the optimiser folds them back into SGPR literals
    u32 M55 = 0x55555555u, M33 = 0x33333333u, M0F = 0x0F0F0F0Fu;
    // how many of them: all three where registers allow (K = 10 ... 13, 16); under the 120-register cap with two groups per thread
    // (K = 14, 15) the Cassini instantiation carries 0x55.. and 0x33.. (six of a gather's seven inserts) in the two registers its
    // table build gives up (`high` / `low` stay scalar there: a v_mov per symbol on the four building wavefronts) and the others
    // none -- a register more spills (12 - 28 bytes of scratch, reloads inside the block)
#ifndef VIT_L2_VECTOR_MASKS
    constexpr bool CAPPED = K == 10 || K == 11 || K == 14 || K == 15;          // lds2_update_is_capped()
    constexpr int VECTOR_MASKS = CAPPED && GPT == 2 ? (RT != 0 ? 2 : 0) : 3;      // (K = 10, 11: one group per thread, registers to spare)
#else
    constexpr int VECTOR_MASKS = VIT_L2_VECTOR_MASKS;
#endif
    if constexpr (VECTOR_MASKS >= 2) asm volatile("" : "+v"(M55), "+v"(M33));
    if constexpr (VECTOR_MASKS >= 3) asm volatile("" : "+v"(M0F));
    // one trellis step on the 16 registers of one group: stage C of a block, table entries at addr[h] + tab_off,
    // decisions of the step -> one dword  (scalar.h:113-134)
    auto stage = [&](auto cc, u32 (&m)[16], const u32 (&addr)[8], u32 tab_off) __attribute__((always_inline)) -> u32 {
        constexpr int C = decltype(cc)::value, PB = 3 - C;
        // per butterfly: add-compare-select, then its four sign bits straight into the step's decision dword.  v_perm selectors
        // 8..11 replicate a 16-bit half's sign over a byte (clean 0x00 / 0xFF): {A r0, B r0, A r1, B r1}, butterfly h -> bit h
        constexpr u32 SIGN_BYTES = 0x0b0a0908u;
        u32 lo4 = 0, sg4[4] = {0u, 0u, 0u, 0u};
        l2_static_for<8>([&](auto hc) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value;
            constexpr int r0 = ((h >> PB) << (PB + 1)) | (h & ((1 << PB) - 1)), r1 = r0 | (1 << PB);
            const uint2 ee = *(const uint2*)((const char*)lds2_smem + addr[h] + tab_off);
            const u32 ma = m[r0], mb = m[r1];
            const u32 x0 = l2_add(ma, ee.x), y0 = l2_add(mb, ee.y);   // -> next state 2a
            const u32 x1 = l2_add(ma, ee.y), y1 = l2_add(mb, ee.x);   // -> next state 2a+1
            m[r0] = l2_min_s(x0, y0);
            m[r1] = l2_min_s(x1, y1);
            const u32 d0 = l2_sub_sat_s(y0, x0);       // sign set <=> x0 > y0 (strict: tie keeps predecessor 0)
            const u32 d1 = l2_sub_sat_s(y1, x1);
            const u32 sg = __builtin_amdgcn_perm(d1, d0, SIGN_BYTES);
            // a tree of bit-field inserts over the clean bytes, four butterflies at a time: 3 + 3 + 1 inserts for the eight
            // perms (the accumulate-by-and_or chain took 8): bit h of every byte <- butterfly h
            sg4[h & 3] = sg;
            if constexpr ((h & 3) == 3) {
                const u32 nib = l2_bfi(M33, l2_bfi(M55, sg4[0], sg4[1]), l2_bfi(M55, sg4[2], sg4[3]));
                if constexpr (h == 3) lo4 = nib;
                else lo4 = l2_bfi(M0F, lo4, nib);
                // four butterflies in flight at a time: the temporaries of eight do not fit beside two groups of metrics
                if constexpr (GPT == 2) __builtin_amdgcn_sched_barrier(0);
            }
        });
        return lo4;
    };
    // swizzled metric buffer (lds2_sw): read view, register r = state r*G + g; write view, piece q = states 16 g + 4 q ...
    constexpr bool SEP = (G / 16) % 8 == 0;     // r*G/16 does not reach the three bits the swizzle touches: base + r*G/4
    // Addresses: ONE byte offset per thread for the 16 / 32 metric reads of a block (register r of group A at +r*G bytes, group
    // B another T bytes up: instruction offsets) and FOUR for its stores (piece q of group A; group B 16 T bytes up), all
    // relative to lds2_smem and formed once, in front of the block loop.  Each use goes through an OPAQUE copy (an empty asm: no
    // instruction): left to itself hipcc materialises the 32 + 8 derived addresses as loop-invariant registers and spills them
    // (236 bytes of scratch, ~50 reloads per block); re-forming them from the thread index in every block, as round 2 did,
    // cost 27 VALU per block.
    const u32 MET_OFF = (u32)PPW * ((u32)GM::tab_bytes + 32u * 4u) + half * (u32)(N * 4);   // byte offset of met[] inside lds2_smem (a constant but with two pairs per wavefront)
    const u32 ld_off = MET_OFF + 4u * lds2_sw((u32)gid, (u32)N);     // relative to lds2_smem
    // the compile-time-rate instantiation has the registers for all four store offsets; the others carry one and form the rest
    // with a v_xor each
    constexpr int NST = RT ? 4 : 1;
    constexpr bool ASM_LS = VIT_L2_ASM_LS && SEP && NST == 4 && G % 256 == 0 && (GPT == 1 || T % 256 == 0);   // see load_metrics
    // K = 10, 11 (G = 32, 64 groups): register r's G bytes reach into the bits the swizzle folds q into, so lds2_sw(r G + g) is not
    // "base + r G" -- but it is for the registers that agree in r mod NB (NB = 4, 2): NB per-thread bases, the rest of r in the
    // offset field, and the 16 loads of a block are eight ds_read2_b32 with no address arithmetic (the compiler's form recomputed
    // the swizzle per register and block: 99 of the 683 vector instructions a K = 11 block took)
    constexpr bool ASM_LD_SMALL = VIT_L2_ASM_LS && !SEP && GPT == 1 && (G == 32 || G == 64);
    constexpr int NB = G == 32 ? 4 : 2;
    u32 ld_base[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) ld_base[k] = MET_OFF + 4u * lds2_sw((u32)(k * G) + (u32)gid, (u32)N);
    u32 st_off[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) st_off[q] = MET_OFF + 4u * (u32)(q * (N / 4)) + 16u * ((u32)gid ^ (u32)(q << 1));
    // the offsets are turned into ABSOLUTE LDS addresses once, here (the address of lds2_smem inside the workgroup's LDS is added
    // now, not in every block), and used through explicit address-space-3 pointers; the loop-carried values themselves are made
    // opaque in place (no copy) at each use
    typedef __attribute__((address_space(3))) char lds_char_t;
    const u32 lds_base = (u32)(uintptr_t)(lds_char_t*)lds2_smem;
    u32 ld_addr = lds_base + ld_off;
#pragma unroll
    for (int q = 0; q < NST; ++q) st_off[q] += lds_base;
    auto at = [&](u32 lds_addr) __attribute__((always_inline)) -> lds_char_t* { return (lds_char_t*)(uintptr_t)lds_addr; };
    // scalar byte base of the 4 G decision dwords of the block that starts at step t0 (a multiple of 4), less what st_off[0]
    // carries beside 16 g
    auto ws_base_of = [&](u32 t0) __attribute__((always_inline)) -> char* {
        return (char*)(ws_pair + (size_t)t0 * G) - (size_t)(MET_OFF + lds_base);
    };
    // stage C of every group of this thread.  The eight table offsets of the stage are unpacked HERE from their packed
    // registers, once for both groups: the packed words change with every block (the table-set bit is toggled in them), so
    // the unpacked values cannot be hoisted out of the block loop -- hoisted, they were 32 loop-invariant registers that
    // the 128-register budget does not have (hipcc spilled 9-22 of them, and every reload's s_waitcnt vmcnt also waited for
    // the decision stores in flight)
    u32 tab_set_base = 0;                       // byte offset of the table set the current block reads (0 or SET_TAB)
    auto stage_all = [&](auto cc, char* wsp_b) __attribute__((always_inline)) {   // wsp_b: ws_base_of(first step of the block)
        constexpr int C = decltype(cc)::value;
        u32 addr[8];
        l2_static_for<8>([&](auto hc) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value;
            addr[h] = tab_set_base + ((h & 1) ? (prow2[C][h >> 1] >> 16) : (prow2[C][h >> 1] & 0xFFFFu));
        });
        // this stage's dword of the block: [group][stage] -- group g's four dwords of a block are 16 contiguous bytes (what the
        // chainback fetches per round trip).  The thread's byte offset 16 g is the loop-carried LDS store address st_off[0] less a
        // constant, and the constant sits in the block's scalar base (wsp_b): no vector register beyond those the metric stores
        // hold anyway (a separate 16 g, and 16 (g + T) for the second group, cost the 120-register instantiations 8 - 20 bytes of
        // scratch).  The stage is an instruction offset; the second group's 16 T bytes go into a scalar base of their own.
#if VIT_L2_WS_GROUPED
        u32 offB = (u32)(16 * T);
        asm volatile("" : "+s"(offB));             // an SGPR whose value the compiler does not know
        if constexpr (ASM_LS) {
            // spelled out like the metric accesses (the address register is loop invariant there: left to the compiler, its
            // zero-extension becomes a 64-bit value of its own and every store a 64-bit vector add): scalar base, 32-bit vector
            // offset, the stage in the offset field.  Not in the compiler's vmcnt: nothing in the kernel waits for a store.
            const u32 dA = stage(cc, mA, addr, 0u);
            asm volatile("global_store_dword %0, %1, %2 offset:%3" : : "v"(st_off[0]), "v"(dA), "s"(wsp_b), "n"(4 * C) : "memory");
            if constexpr (GPT == 2) {
                const u32 dB = stage(cc, mB, addr, 512u);
                asm volatile("global_store_dword %0, %1, %2 offset:%3" : : "v"(st_off[0]), "v"(dB), "s"(wsp_b + offB), "n"(4 * C) : "memory");
            }
        } else {
            l2_opaque(st_off[0]);
            *((u32*)(wsp_b + st_off[0]) + C) = stage(cc, mA, addr, 0u);
            if constexpr (GPT == 2) *((u32*)((wsp_b + offB) + st_off[0]) + C) = stage(cc, mB, addr, 512u);
        }
#else
        u32 row_off = (u32)(C * G);
        asm volatile("" : "+s"(row_off));
        u32* const row = (u32*)(wsp_b + (size_t)(MET_OFF + lds_base)) + row_off;
        row[gid] = stage(cc, mA, addr, 0u);
        if constexpr (GPT == 2) row[T + gid] = stage(cc, mB, addr, 512u);
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    // the compile-time-rate instantiation (Cassini: K = 15, both offsets multiples of 256 bytes) spells its metric loads and stores
    // out; its five addresses are made opaque ONCE, in front of the block loop
    if constexpr (ASM_LS) {
        l2_opaque(ld_addr);
#pragma unroll
        for (int q = 0; q < NST; ++q) l2_opaque(st_off[q]);
    }
    if constexpr (ASM_LD_SMALL) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            ld_base[k] += lds_base;
            l2_opaque(ld_base[k]);
        }
    }
    l2_u32x2 ldA[8], ldB[8];                   // landing pairs of the spelled-out loads: registers 2k, 2k + 1 of a group
    auto load_group = [&](u32 (&m)[16], u32 g, u32 off) __attribute__((always_inline)) {
        l2_static_for<16>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            if constexpr (SEP) m[r] = *(const __attribute__((address_space(3))) u32*)at(off + (u32)(r * (G / 4)) * 4u);
            else m[r] = met[lds2_sw((u32)(r * G) + g, (u32)N)];
        });
    };
    auto load_metrics = [&]() __attribute__((always_inline)) {
        if constexpr (ASM_LS) {
            // register r of group A at ld_addr + r G bytes, of group B another T bytes up: r G / 256 and T / 256 in the offset fields
            l2_static_for<8>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                ldA[k] = l2_ds_read2st64<(2 * k) * G / 256, (2 * k + 1) * G / 256>(ld_addr);
                if constexpr (GPT == 2) ldB[k] = l2_ds_read2st64<(2 * k) * G / 256 + T / 256, (2 * k + 1) * G / 256 + T / 256>(ld_addr);
            });
            return;
        }
        if constexpr (ASM_LD_SMALL) {
            l2_static_for<8>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                constexpr int r_lo = (k / NB) * 2 * NB + (k % NB), r_hi = r_lo + NB;      // the same base: r_lo = r_hi (mod NB)
                ldA[k] = l2_ds_read2_b32<(r_lo - r_lo % NB) * G / 4, (r_hi - r_hi % NB) * G / 4>(ld_base[r_lo % NB]);
            });
            return;
        }
        u32 t = (u32)gid;
        l2_opaque(ld_addr);
        if constexpr (!SEP) l2_opaque(t);
        load_group(mA, t, ld_addr);
        if constexpr (GPT == 2) load_group(mB, t + (u32)T, ld_addr + (u32)T);      // lds2_sw(g + T) = lds2_sw(g) + T / 4 dwords
    };
    // "the metric loads (and this wavefront's table writes) have landed": lgkmcnt(0)
    auto await_metrics = [&]() __attribute__((always_inline)) {
        if constexpr (ASM_LS) {
            l2_lds_landed(ldA);
            if constexpr (GPT == 2) l2_lds_landed(ldB);
            l2_static_for<8>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                mA[2 * k] = ldA[k].x;
                mA[2 * k + 1] = ldA[k].y;
                if constexpr (GPT == 2) { mB[2 * k] = ldB[k].x; mB[2 * k + 1] = ldB[k].y; }
            });
        } else if constexpr (ASM_LD_SMALL) {
            l2_lds_landed(ldA);
            l2_static_for<8>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                constexpr int r_lo = (k / NB) * 2 * NB + (k % NB), r_hi = r_lo + NB;
                mA[r_lo] = ldA[k].x;
                mA[r_hi] = ldA[k].y;
            });
        } else {
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
    };
    auto store_metrics = [&]() __attribute__((always_inline)) {                 // after stage 3: register r holds state 16 g + r
        l2_static_for<4>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) u32x4_t lds_uint4_t;
            if constexpr (ASM_LS) {
                l2_ds_write_b128<0>(st_off[q], l2_u32x4{mA[4 * q], mA[4 * q + 1], mA[4 * q + 2], mA[4 * q + 3]});
                if constexpr (GPT == 2) l2_ds_write_b128<16 * T>(st_off[q], l2_u32x4{mB[4 * q], mB[4 * q + 1], mB[4 * q + 2], mB[4 * q + 3]});
                return;
            }
            u32 o;
            if constexpr (NST == 4) {
                l2_opaque(st_off[q]);
                o = st_off[q];
            } else {
                if constexpr (q == 0) l2_opaque(st_off[0]);
                // 16 (tid ^ 2q) = 16 tid ^ 32 q, on the tid part only: the constant parts (and the LDS base) are taken off first
                o = ((st_off[0] - MET_OFF - lds_base) ^ (u32)(q << 5)) + MET_OFF + lds_base + 4u * (u32)(q * (N / 4));
            }
            *(lds_uint4_t*)at(o) = u32x4_t{mA[4 * q], mA[4 * q + 1], mA[4 * q + 2], mA[4 * q + 3]};
            if constexpr (GPT == 2) *(lds_uint4_t*)at(o + 16u * (u32)T) = u32x4_t{mB[4 * q], mB[4 * q + 1], mB[4 * q + 2], mB[4 * q + 3]};
        });
        // the spelled-out stores are not in the compiler's count: the barrier behind them needs the wait written down
        if constexpr (ASM_LS) __builtin_amdgcn_s_waitcnt(0xC07F);
    };
    // block-wide renormalisation of the registers (scalar.h:139-153) for the frames whose sign bit is set in `need`
    u32 renorm_count = 0;                      // reductions run so far (block-uniform): picks the set of per-wavefront minima
    auto renormalise = [&](u32 need) __attribute__((always_inline)) {
        const u32 msk = ((need & 0x8000u) ? 0x0000FFFFu : 0u) | ((need & 0x80000000u) ? 0xFFFF0000u : 0u);
        u32 mn = mA[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mn = l2_min_s(mn, mA[i]);
        if constexpr (GPT == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) mn = l2_min_s(mn, mB[i]);
        }
#pragma unroll
        for (int off = PPW == 2 ? 16 : 32; off >= 1; off >>= 1) mn = l2_min_s(mn, (u32)__shfl_xor((int)mn, off));   // (two pairs per wavefront: each half on its own)
        // two sets of per-wavefront minima, used in turn (up to eight wavefronts: K <= 15): the barrier below is then the only one of a
        // reduction -- the next reduction writes the other set, and the one after it cannot start before everybody has passed
        // the next one's barrier, i.e. has long read this set
        constexpr bool WMIN2 = NW <= 8;
        u32* const wm = wmin + (WMIN2 ? (renorm_count & 1u) * 8u : 0u);
        renorm_count += 1u;
        if ((PPW == 2 ? ls : lane) == 0) wm[wave] = mn;
        __syncthreads();
        for (int w = 0; w < NW; ++w) mn = l2_min_s(mn, wm[w]);
        const u32 sub = (mn ^ BIAS2) & msk;   // true (unbiased) minimum of each frame that renormalises
#pragma unroll
        for (int i = 0; i < 16; ++i) mA[i] = l2_sub(mA[i], sub);
        if constexpr (GPT == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) mB[i] = l2_sub(mB[i], sub);
        }
        if (ls == 0) {                        // update()'s return value: kept in LDS, not in four VGPRs of every thread
            rs_acc[0] += (uint64_t)((sub & 0xFFFFu) >> SHIFT);
            rs_acc[1] += (uint64_t)((sub >> 16) >> SHIFT);
        }
        if constexpr (!WMIN2) __syncthreads();   // one set only (K = 16): it may be rewritten by the next reduction
    };
    // the careful version of a block: stages [c_first, nst) with the threshold test and the reduction after EVERY stage; used
    // when the prediction says state 0 may cross the threshold inside the block, for the entry block of a resumed call
    // (c_first > 0) and for the last partial block of a frame (nst < BLK).  Returns the prediction for the next block.
    auto slow_block = [&](u32 t0, int set, int c_first, int nst) __attribute__((always_inline)) -> u32 {
        tables_build(t0 + BLK, set ^ 1);       // nobody has built the next block's tables yet
        tables_load(t0 + 2 * BLK);
        tables_commit();
        // an opaque copy of the thread index: the scatter / gather addresses below are loop invariant, and hoisted out of the
        // main loop they would cost the FAST path its registers for a once-per-frame use
        u32 tid_o = (u32)gid;
        asm volatile("" : "+v"(tid_o));
        if (c_first == 0) {
            load_metrics();
            await_metrics();
        } else {                               // mid-block entry: the registers as stage c_first - 1 would have left them
            l2_static_for<16>([&](auto rc) __attribute__((always_inline)) {
                constexpr u32 r = decltype(rc)::value;
                mA[r] = met[lds2_sw(lds2_state_of(c_first - 1, r, tid_o, SBITS), (u32)N)];
                if constexpr (GPT == 2) mB[r] = met[lds2_sw(lds2_state_of(c_first - 1, r, tid_o + (u32)T, SBITS), (u32)N)];
            });
        }
        lds2_arrive(arrive, lane);             // keeps the fast path's count in step
        __syncthreads();                       // B1: all inputs are in registers, the buffer may be overwritten
        l2_static_for<BLK>([&](auto cc) __attribute__((always_inline)) {
            constexpr int C = decltype(cc)::value;
            if (C >= c_first && C < nst) {
                stage_all(cc, ws_base_of(t0));
                // state 0 is register 0 of thread 0's first group after every stage: thread 0 tells the other wavefronts whether a frame
                // has to renormalise.  A ONE-WAY signal, not a barrier: nothing but this word passes between the stages of a careful
                // block (the metrics stay in registers), so nobody waits for anybody but wavefront 0.  Slot C of the block holds
                // (sequence number of the careful stage << 2 | the two frames' bits); the block's closing barrier separates this
                // block's writes of a slot from the next block's.  (Round 5: two workgroup barriers per stage -- with the reference's
                // 8-bit soft configuration 62 % of Cassini's blocks are careful ones: a frame renormalises every 8.5 steps.  Cassini
                // SOFT8 4096 x 8192: 61.3 -> 59.0 ms.  Building the tables on the LAST four wavefronts, so that wavefront 0 reaches
                // its word sooner, changed nothing -- 59.4: what is left of SOFT8's distance to HARD8's 50.4 ms is the reduction and
                // the 32 subtractions themselves, about once per block.)
                careful_seq += 1u;
                if (ls == 0) {
                    const u32 nb2 = (l2_sub_sat_s(THRM1B2, mA[0]) | FORCE) & BIAS2;
                    __hip_atomic_store(&sig[C], (careful_seq << 2) | ((nb2 >> 15) & 1u) | ((nb2 >> 30) & 2u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                u32 sv;
                while (((sv = __hip_atomic_load(&sig[C], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 2) != careful_seq) __builtin_amdgcn_s_sleep(1);
                if constexpr (PPW == 1) sv = (u32)__builtin_amdgcn_readfirstlane((int)sv);   // block-uniform: scalar branch
                const u32 need = ((sv & 1u) << 15) | ((sv & 2u) << 30);               // (two pairs per wavefront: per half)
                if (PPW == 2 ? __builtin_amdgcn_ballot_w64(need != 0) != 0 : need != 0) renormalise(need);
                if (C == nst - 1) {
                    if (C == BLK - 1) {
                        store_metrics();
                    } else {
                        l2_static_for<16>([&](auto rc) __attribute__((always_inline)) {
                            constexpr u32 r = decltype(rc)::value;
                            met[lds2_sw(lds2_state_of(C, r, tid_o, SBITS), (u32)N)] = mA[r];
                            if constexpr (GPT == 2) met[lds2_sw(lds2_state_of(C, r, tid_o + (u32)T, SBITS), (u32)N)] = mB[r];
                        });
                    }
                }
            }
        });
        if (ls == 0) flag[0] = nst == BLK ? predict(mA[0], t0 + BLK, set ^ 1) : 1u;
        __syncthreads();                       // B2
        return read_careful();
    };
    // the fast version: four stages back to back between the two barriers.  Returns the prediction for the next block.
    auto fast_block = [&](int set, u32 t0, u32 arrive_target) __attribute__((always_inline)) -> u32 {
        // the metric loads go out first; behind them, while they are in flight: the tables of the NEXT block from the symbols
        // fetched during the previous one
        load_metrics();
        tables_build(t0 + BLK, set ^ 1);
        // B1, split: "my loads have returned" is announced here (one LDS add per wavefront) ...
        await_metrics();                             // lgkmcnt(0): the 16 / 32 metric loads (and the table writes) are complete
        lds2_arrive(arrive, lane);
        char* const wsp = ws_base_of(t0);           // uniform: the 4 G decision dwords of this block (t0 is a multiple of 4)

        l2_static_for<BLK>([&](auto cc) __attribute__((always_inline)) { stage_all(cc, wsp); });
        // ... and awaited only here, four trellis steps later, before the first store: by now every wavefront has long arrived
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < arrive_target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        tables_load(t0 + 2 * BLK);             // the symbols of the block after next: in flight across the barrier
        store_metrics();
        if (ls == 0) flag[0] = predict(mA[0], t0 + BLK, set ^ 1);
        __syncthreads();                       // B2
        tables_commit();
        const u32 pred = read_careful();                                       // uniform by construction, and carried as a dword (a
                                                                              // loop-carried bool becomes a lane mask): the block loop's control flow stays scalar
        // renormalise when new_metric[0] >= threshold after the block's last step: block-uniform
        u32 need = (l2_sub_sat_s(THRM1B2, met[0]) | FORCE) & BIAS2;            // sign bits: frame A / frame B
        if constexpr (PPW == 1) need = (u32)__builtin_amdgcn_readfirstlane((int)need);
        if (PPW == 2 ? __builtin_amdgcn_ballot_w64(need != 0) != 0 : need != 0) {
            __syncthreads();                   // everybody has read met[0] and flag[0] before they can change
            renormalise(need);
            store_metrics();
            __syncthreads();
        }
        return pred;
    };

    u32 t0 = tb0;
    u32 arrivals_wanted = (u32)NW;            // value of *arrive once every wavefront has loaded the current block's metrics
    int set = 0;                              // table set that holds block t0
    u32 careful = read_careful();
    while (t0 < t_end) {
        const u32 left = t_end - t0;
        const int nst = left < (u32)BLK ? (int)left : BLK;
        // > 0 only in the entry block of a resumed call.  Written so that it stays on the scalar unit: t0 >= t_begin & ~3, hence
        // t_begin - t0 is 1..3 there and wraps to a huge value everywhere else (as `t0 < t_begin ? ... : 0` hipcc computed it
        // with a saturating VECTOR subtract and a v_readfirstlane, four VALU per block)
        const u32 c_gap = t_begin - t0;
        const int c_first = c_gap < (u32)BLK ? (int)c_gap : 0;
        if (nst < BLK || c_first > 0 || careful != 0) careful = slow_block(t0, set, c_first, nst);
        else careful = fast_block(set, t0, arrivals_wanted);
        // the next block reads the other table set: one base offset flips; the packed offsets are made opaque once per block (no
        // instruction), so that their unpacked values cannot be hoisted out of the loop as 32 loop-invariant registers
        tab_set_base ^= SET_TAB;
#pragma unroll
        for (int c = 0; c < BLK; ++c)
#pragma unroll
            for (int h2 = 0; h2 < 4; ++h2) l2_opaque(prow2[c][h2]);
        arrivals_wanted += (u32)NW;
        t0 += (u32)BLK;
        set ^= 1;
    }

    if (a.metrics_out) {
        // an opaque copy of the thread index: the addresses of this once-per-call write-out are functions of it, and formed in
        // front of the block loop (where hipcc hoists them to) they sit in registers the loop needs -- or in scratch
        int s_first = ls;
        asm volatile("" : "+v"(s_first));
        for (int s = s_first; s < N; s += LT) {
            const u32 v = met[lds2_sw((u32)s, (u32)N)] ^ BIAS2;
            if (SHIFT) {
                ((uint8_t*)a.metrics_out)[(size_t)fA * N + s] = (uint8_t)((v & 0xFFFFu) >> 8);
                if (validB) ((uint8_t*)a.metrics_out)[(size_t)fB * N + s] = (uint8_t)(v >> 24);
            } else {
                ((uint16_t*)a.metrics_out)[(size_t)fA * N + s] = (uint16_t)(v & 0xFFFFu);
                if (validB) ((uint16_t*)a.metrics_out)[(size_t)fB * N + s] = (uint16_t)(v >> 16);
            }
        }
    }
    if (a.renorm_sum && ls == 0) {
        a.renorm_sum[fA] = rs_acc[0];
        if (validB) a.renorm_sum[fB] = rs_acc[1];
    }
}

template <int K, int SHIFT, int RT = 0>
__global__ void __launch_bounds__(Lds2Geom<K>::T, Lds2Geom<K>::MINW) lds2_update_kernel(Lds2UpdateArgs a) { lds2_update_body<K, SHIFT, RT>(a); }
// The same body capped at 120 registers (the attribute counts in halves of the unified file), used for K = 10, 11, 14, 15: four
// 125..128-register waves fill a SIMD's 512, four 120-register ones leave 32 -- the chainback kernel's 24, which then runs
// beside the next batch's update instead of behind it.  All of them fit the cap without scratch (tests/test_codegen.py); round 3:
// alone the capped Cassini kernel was 0.6 % slower (49.2 -> 49.5 ms), the overlap hid 2.3 ms of chainback: 51.6 -> 50.1 - 50.3 ms per
// 4096 x 8192 batch.  (K = 10, 11 use 96 - 110 registers since their symbols stay per lane; K = 12 needs 96 anyway; K = 13 and 16
// would spill.)
template <int K, int SHIFT, int RT = 0>
__global__ void __launch_bounds__(Lds2Geom<K>::T, Lds2Geom<K>::MINW) __attribute__((amdgpu_num_vgpr(60)))
lds2_update_kernel_c120(Lds2UpdateArgs a) { lds2_update_body<K, SHIFT, RT>(a); }
// the instantiation lds2_launch_update picks for (K, R): capped kernel or not, compile-time rate or not
inline bool lds2_update_is_capped(int K) { return K == 11 || K == 14 || K == 15; }   // (K = 10: two pairs per wavefront, 12.4 KiB of LDS: three waves per SIMD, no cap needed)
inline int lds2_update_rt(int K, int R) { return (K == 15 && R == 6) ? 6 : 0; }
inline size_t lds2_smem_bytes(int K) {
    const size_t N = (size_t)1 << (K - 1), G = N / 16, GPT = G >= 256 ? 2 : 1;
    return (size_t)(K == 10 ? 2 : 1) * ((size_t)2 * 4 * GPT * 64 * 8 + 32 * 4 + N * 4);                 // Lds2Geom<K>::smem_bytes
}
// what one wave of the update / chainback kernel of (K, R) allocates: from the kernel descriptors (kernel_desc.hpp)
inline bool lds2_kernel_resources(int K, int R, int shift, bool update, kd::KernelResources* out, unsigned* dyn_lds_bytes = nullptr) {
    if (dyn_lds_bytes) *dyn_lds_bytes = update ? (unsigned)lds2_smem_bytes(K) : 0u;
    const kd::KernelResources* r = nullptr;
    if (update) {
        char inst[64];
        snprintf(inst, sizeof(inst), "ILi%dELi%dELi%dEEEvNS_14Lds2UpdateArgsE", K, shift ? 8 : 0, lds2_update_rt(K, R));
        r = kd::find(kd::own_library(), {lds2_update_is_capped(K) ? "23lds2_update_kernel_c120I" : "18lds2_update_kernelI", inst});
    } else {
        r = kd::find(kd::own_library(), {"21lds2_chainback_kernelE"});
    }
    if (!r) return false;
    *out = *r;
    return true;
}
// codes whose update kernel leaves room for a chainback wave on every SIMD (the pipeline overlaps their chainback): as many
// update workgroups as a CU's LDS holds, their waves spread over four SIMDs, plus one chainback wave -- by descriptor allocation.
// (K = 11, 12, 14, 15 on an MI355X; unreadable descriptors answer "no": back to back)
inline bool lds2_chainback_fits_beside_update(int K, int R, int shift) {
    kd::KernelResources u, c;
    if (!lds2_kernel_resources(K, R, shift, true, &u) || !lds2_kernel_resources(K, R, shift, false, &c)) return false;
    size_t threads = (((size_t)1 << (K - 1)) / 16) / ((((size_t)1 << (K - 1)) / 16) >= 256 ? 2 : 1);   // Lds2Geom<K>::T
    if (threads < 64) threads = 64;
    const size_t wg_per_cu = (160 * 1024) / lds2_smem_bytes(K);
    size_t waves_per_simd = (wg_per_cu * (threads / 64) + 3) / 4;          // what the LDS admits ...
    if (waves_per_simd > 512 / u.vgpr_alloc) waves_per_simd = 512 / u.vgpr_alloc;   // ... and what the register file does
    return waves_per_simd * u.vgpr_alloc + c.vgpr_alloc <= 512;
}

// ---- chainback / export on the PLAN_LDS2 layout -------------------------------------------------------------------
struct Lds2ChainbackArgs {
    const u32* ws;
    size_t ws_pair_stride;
    uint8_t* out;
    const u32* end_state;
    u32 frames, L;
    int32_t K;
};

// one lane per frame; the decision dword of (step t, next-state s): ws[pair][t][j], bit of register r -- lds2_locate().
// The chase is a chain of dependent loads (the dword index j is a slice of the state), but only partly: thread index j of
// step t uses state bits [nlow, nlow + K-5) with nlow = t%4 + 1, and going k steps back leaves the low K-1-k bits of the
// state known.  Starting from a step with t%4 == 3 the addresses of steps t, t-1, t-2, t-3 all follow from state_t alone
// (their slices end below the bits that are still to be decoded), so the four loads are issued together and the four bits
// are resolved from registers: ONE memory round trip per four steps instead of four.
__global__ void lds2_chainback_kernel(Lds2ChainbackArgs a) {
    const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.frames) return;
    const int TSB = a.K - 1;
    const u32 T = (1u << TSB) / 16u;
    const int ignore = TSB < 8 ? TSB : 8;           // ViterbiTracebackBuffer::get_layout (core.h:129-149)
    const int shift_state = 8 - ignore, shift_tail = TSB - ignore, total_bits = TSB + shift_state;
    const u32* ws = a.ws + (f >> 1) * a.ws_pair_stride;   // [block][group][step % 4]
    const u32 half = (u32)(f & 1);
    uint8_t* out = a.out + f * (((size_t)a.L + 7) / 8);
    u32 reg = (a.end_state ? (a.end_state[f] & ((1u << TSB) - 1u)) : 0u) << shift_state;
    const u32 jmask = T - 1u;
    auto step = [&](size_t j, u32 w) {               // resolve decoded bit j from the already loaded dword of step j + TSB
        const u32 state = reg >> shift_state;
        const u32 t = (u32)(j + (size_t)TSB);
        u32 tj, r;
        lds2_locate(state, t, TSB, tj, r);
        const u32 bit = (w >> lds2_dec_bit(r, half, t & 3u)) & 1u;
        reg = (reg >> 1) | (bit << (total_bits - 1));
        if ((j & 7) == 0) out[j >> 3] = (uint8_t)((reg >> shift_tail) & 0xFFu);
    };
    auto load = [&](size_t j, u32 state_j) -> u32 {  // dword of step j + TSB for a survivor whose (known part of the) state is state_j
        const u32 t = (u32)(j + (size_t)TSB);
        const u32 nlow = (t & 3u) + 1u;
        return ws[lds2_ws_index(t, (state_j >> nlow) & jmask, T)];
    };
    size_t j = a.L;
    // ragged top: single dependent steps down to a step with t % 4 == 3
    while (j > 0 && (((j - 1 + (size_t)TSB) & 3) != 3)) {
        --j;
        step(j, load(j, reg >> shift_state));
    }
    // blocks of four steps: t = j-1+TSB has t % 4 == 3
    while (j >= 4) {
        const u32 s0 = reg >> shift_state;
        // state k steps back = (s0 >> k) | (newest k bits on top): the slice of step t-k ends at bit (4-k) + K-6 < K-1-k -- and it
        // is the SAME slice of s0 for the four steps, bits [4, K-1): one group index, whose four dwords of the block are one
        // 16-byte load out of one sector
        const u32 tt = (u32)(j - 1 + (size_t)TSB);               // t % 4 == 3: the block's last step
#if VIT_L2_WS_GROUPED
        const uint4 w = *(const uint4*)(ws + lds2_ws_index(tt & ~3u, (s0 >> 4) & jmask, T));
#else
        const u32 jj = (s0 >> 4) & jmask;
        const uint4 w = make_uint4(ws[lds2_ws_index(tt - 3, jj, T)], ws[lds2_ws_index(tt - 2, jj, T)], ws[lds2_ws_index(tt - 1, jj, T)],
                                   ws[lds2_ws_index(tt, jj, T)]);
#endif
        step(j - 1, w.w);
        step(j - 2, w.z);
        step(j - 3, w.y);
        step(j - 4, w.x);
        j -= 4;
    }
    while (j > 0) {
        --j;
        step(j, load(j, reg >> shift_state));
    }
}

struct Lds2ExportArgs {
    const u32* ws;
    size_t ws_pair_stride;
    uint64_t* out;          // [F][n_steps][W]
    u32 frames, n_steps;
    int32_t K;
};

__global__ void lds2_export_kernel(Lds2ExportArgs a) {
    const int TSB = a.K - 1;
    const u32 W = 1u << (TSB - 6);
    const u32 T = (1u << TSB) / 16u;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (frame, step, word)
    const size_t total = (size_t)a.frames * a.n_steps * W;
    if (idx >= total) return;
    const u32 w = (u32)(idx % W);
    const u32 t = (u32)((idx / W) % a.n_steps);
    const u32 f = (u32)(idx / ((size_t)W * a.n_steps));
    const u32* ws = a.ws + (size_t)(f >> 1) * a.ws_pair_stride;
    const u32 half = f & 1u;
    uint64_t word = 0;
    for (u32 b = 0; b < 64; ++b) {
        u32 tj, r;
        lds2_locate(w * 64 + b, t, TSB, tj, r);
        word |= (uint64_t)((ws[lds2_ws_index(t, tj, T)] >> lds2_dec_bit(r, half, t & 3u)) & 1u) << b;
    }
    a.out[idx] = word;
}

// ---- host side ----------------------------------------------------------------------------------------------------
inline bool lds2_supported(int K, int R) { return K >= 10 && K <= 16 && R >= 1 && R <= 6; }   // 64-entry branch-metric table
inline size_t lds2_threads(int K) { return ((size_t)1 << (K - 1)) / 16; }
inline size_t lds2_rows4(int K, size_t L) { return (L + (size_t)K - 1 + 3) / 4 * 4; }          // whole blocks of four steps
inline size_t lds2_workspace_bytes(int K, size_t frames, size_t L) {
    return ((frames + 1) / 2) * lds2_rows4(K, L) * lds2_threads(K) * 4;
}

template <int K, int SHIFT>
int lds2_launch_update(const Lds2UpdateArgs& a, unsigned pairs, hipStream_t st) {
    using GM = Lds2Geom<K>;
    void (*kern)(Lds2UpdateArgs) = nullptr;
    if constexpr (K == 15) kern = (a.R == 6) ? lds2_update_kernel_c120<K, SHIFT, 6> : lds2_update_kernel_c120<K, SHIFT, 0>;
    else if constexpr (K == 11 || K == 14) kern = lds2_update_kernel_c120<K, SHIFT, 0>;
    else kern = lds2_update_kernel<K, SHIFT, 0>;
    if (GM::smem_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)GM::smem_bytes) != hipSuccess)
            return -1;
    }
#ifdef VIT_HIP_EXPERIMENTS
    if (getenv("VIT_HIP_DEBUG")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), GM::T, GM::smem_bytes);
        fprintf(stderr, "lds2_update_kernel<%d>: %d threads, %zu B LDS per workgroup, %d workgroup(s) per CU\n", K, GM::T, GM::smem_bytes, nb);
    }
#endif
    hipLaunchKernelGGL(kern, dim3((pairs + (unsigned)GM::PPW - 1u) / (unsigned)GM::PPW), dim3(GM::T), GM::smem_bytes, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int SHIFT>
int lds2_launch_update_k(int K, const Lds2UpdateArgs& a, unsigned pairs, hipStream_t st) {
    switch (K) {
        case 10: return lds2_launch_update<10, SHIFT>(a, pairs, st);
        case 11: return lds2_launch_update<11, SHIFT>(a, pairs, st);
        case 12: return lds2_launch_update<12, SHIFT>(a, pairs, st);
        case 13: return lds2_launch_update<13, SHIFT>(a, pairs, st);
        case 14: return lds2_launch_update<14, SHIFT>(a, pairs, st);
        case 15: return lds2_launch_update<15, SHIFT>(a, pairs, st);
        case 16: return lds2_launch_update<16, SHIFT>(a, pairs, st);
        default: return -1;
    }
}

// the per-stage table index maps (lds2_tab_index) from the host copy of the branch patterns
inline void lds2_index_maps(int K, const uint16_t* pattern, uint8_t (&f_out)[4], uint8_t (&t_out)[4]) {
    for (int c = 0; c < 4; ++c) {
        // lanes 0-31 of a wavefront differ in thread-index bits 0-4 = butterfly-index bits c .. c+4 in stage c
        uint32_t basis[5];
        for (int i = 0; i < 5; ++i) basis[i] = pattern[(size_t)1 << (c + i)] & 63u;
        uint32_t f = 0;
        for (uint32_t cand = 1; cand < 64 && !f; ++cand) {
            bool ok = true;
            for (int i = 0; i < 5; ++i) ok = ok && ((__builtin_popcount(cand & basis[i]) & 1) == 0);
            if (ok) f = cand;
        }
        if (!f) f = 32;     // cannot happen (five vectors never span six dimensions); natural order
        f_out[c] = (uint8_t)f;
        t_out[c] = (uint8_t)(31 - __builtin_clz(f));
    }
    (void)K;
}

inline int lds2_update(int K, int R, const DevConfig& cfg, int shift, const uint16_t* d_pattern, const uint16_t* h_pattern, const void* d_symbols,
                       size_t sym_stride, size_t frames, size_t first_step, size_t n_steps, size_t L, void* d_ws,
                       const void* d_metrics_in, void* d_metrics, uint64_t* d_renorm, const uint32_t* d_start, hipStream_t st) {
    if (frames == 0 || n_steps == 0) return 0;
    Lds2UpdateArgs a{};
    a.symbols = (const uint8_t*)d_symbols;
    a.sym_frame_stride_bytes = sym_stride * (shift ? 1 : 2);
    a.ws = (u32*)d_ws;
    a.ws_pair_stride = lds2_rows4(K, L) * lds2_threads(K);
    a.metrics_out = d_metrics;
    a.renorm_sum = d_renorm;
    a.start_state = d_start;
    a.pattern = d_pattern;
    lds2_index_maps(K, h_pattern, a.idx_f, a.idx_t);
    a.metrics_in = d_metrics_in;
    a.frames = (u32)frames;
    a.t_begin = (u32)first_step;
    a.t_end = (u32)(first_step + n_steps);
    a.R = R;
    a.cfg = cfg;
    const unsigned pairs = (unsigned)((frames + 1) / 2);
    return shift ? lds2_launch_update_k<8>(K, a, pairs, st) : lds2_launch_update_k<0>(K, a, pairs, st);
}

inline int lds2_chainback(int K, const void* d_ws, size_t frames, size_t L, uint8_t* d_out, const uint32_t* d_end,
                          hipStream_t st) {
    if (frames == 0 || L == 0) return 0;
    Lds2ChainbackArgs a{};
    a.ws = (const u32*)d_ws;
    a.ws_pair_stride = lds2_rows4(K, L) * lds2_threads(K);
    a.out = d_out;
    a.end_state = d_end;
    a.frames = (u32)frames;
    a.L = (u32)L;
    a.K = K;
    hipLaunchKernelGGL(lds2_chainback_kernel, dim3((unsigned)((frames + 63) / 64)), dim3(64), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

inline int lds2_export(int K, const void* d_ws, size_t frames, size_t n_steps, size_t L, uint64_t* d_out, hipStream_t st) {
    if (frames == 0 || n_steps == 0) return 0;
    Lds2ExportArgs a{};
    a.ws = (const u32*)d_ws;
    a.ws_pair_stride = lds2_rows4(K, L) * lds2_threads(K);
    a.out = d_out;
    a.frames = (u32)frames;
    a.n_steps = (u32)n_steps;
    a.K = K;
    const size_t total = frames * n_steps * ((size_t)1 << (K - 7));
    hipLaunchKernelGGL(lds2_export_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vit
