// kernels_reg.hpp -- PLAN_REG: state metrics resident in VGPRs, packed 2x16-bit arithmetic, frames in pairs.
//
// Device implementation of the reference's scalar strategy for the K = 3, 5, 7, 9 stock codes
//   ViterbiDecoder_Scalar::update / bfly / renormalise   include/viterbi/viterbi_decoder_scalar.h:29-153
//   ViterbiDecoder_Core::chainback                       include/viterbi/viterbi_decoder_core.h:214-236
// Results are bit-identical to PLAN_LDS and to the reference; only the mapping onto the machine differs.
//
// Mapping (gfx950, wave64), LANE_BITS = 2 (K = 7, 9).  One wavefront decodes a TILE of 32 frames.  Lane l = 16*q + g
// (q = 0..3, g = 0..15) works on the frame pair (A = 32*tile + g, B = A + 16): every VGPR holds one state metric of frame A
// in its low half and the same state's metric of frame B in its high half, so one v_pk_* instruction advances two frames.
// The 2^(K-1) states of a frame are spread over the 4 q-lanes x NREG registers: "slot" x = (q << REG_BITS) | r.
// LANE_BITS = 0 (K = 3, 5): all states of a pair sit in one lane (64 pairs = 128 frames per wave, no cross-lane traffic).
//
//   * in-place butterflies by index rotation: at trellis step t (phase ph = t mod (K-1)) state s lives in slot
//     rotr^ph(s).  The butterfly {(0|X),(1|X)} -> {(X|0),(X|1)} then reads and writes the SAME two slots, which differ in
//     slot bit p = K-2-ph.  When p is a register bit the step needs no data movement at all; when p is one of the two
//     lane bits, one v_permlane32_swap / v_permlane16_swap per register pair exchanges that lane bit with the top
//     register bit before the butterflies and again after them (2 of every K-1 steps).
//   * branch metrics: the convolutional code is linear, so the R-bit branch pattern of butterfly (q, r) is
//     pat(r) ^ pat(q); pat(r) is a compile-time constant per register.  Only 2^R packed error sums E[p] (and
//     max_error - E[p]) exist per frame pair and step.  K = 7, 9: the four q-lanes of a pair SHARE that work through an LDS
//     ring -- in every group of 4 steps lane (q, g) computes the sums of step 4J+q once, and every step each lane reads
//     E[p ^ pat(q)] through per-lane addresses (LDSBM in reg_update_body).  K = 3, 5: computed in the lane, one step ahead.
//   * add-compare-select per register pair, 8 packed instructions, exact for wrapping u16 metrics: metrics are kept
//     biased (m ^ 0x8000) so the reference's unsigned compare is a signed one; new = v_pk_min_i16(x, y); the decision is
//     the SIGN of v_pk_sub_i16(y, x) with clamp -- saturation keeps the sign exact at any distance, a tie gives 0 (the
//     reference's strict '>' keeps predecessor 0).
//   * u8 metrics / s8 symbols are carried in the HIGH byte of each 16-bit half, which makes 16-bit wrapping arithmetic
//     reproduce mod-256 arithmetic.
//   * decision bits: the sign bits of a lane's registers are byte-gathered (v_perm_b32 over the pair (r, r+8)) into one
//     dword per 16 registers: byte 0/1 = frame A/B registers 0-7, byte 2/3 = frame A/B registers 8-15.  Stored as coalesced
//     16-byte-per-lane rows ws[tile][step group][lane] (1 KiB per wave store).  Bit order is the slot order of the step
//     (a rotation of the state index) -- vit_hip_export_decisions() converts to the reference's bit order.
//   * chainback: K = 7 uses one lane per frame PAIR and K <= 5 one lane per frame, with everything on the dependent chain
//     lane-local (reg_chainback16_body / reg_chainback0_body); K = 9 keeps the update kernel's lane roles, each q-lane
//     extracts the candidate bit of its slice and a ds_bpermute fetches the survivor's (reg_chainback_coop_body).  The
//     row rings of K = 7, 9 keep counted vmcnt waits because no store is pending inside their loops: output bytes are
//     parked in LDS and flushed every 1024 steps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "common.hpp"

namespace vit {

typedef uint32_t u32;
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));

#define VIT_DEV __device__ __forceinline__

VIT_DEV u32 pk_add(u32 a, u32 b) {
    return __builtin_bit_cast(u32, (u16x2_t)(__builtin_bit_cast(u16x2_t, a) + __builtin_bit_cast(u16x2_t, b)));
}
VIT_DEV u32 pk_sub(u32 a, u32 b) {
    return __builtin_bit_cast(u32, (u16x2_t)(__builtin_bit_cast(u16x2_t, a) - __builtin_bit_cast(u16x2_t, b)));
}
VIT_DEV u32 pk_max_s(u32 a, u32 b) {
    return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, a), __builtin_bit_cast(s16x2_t, b)));
}
// (a & mask) | (b & ~mask).  Inline asm on purpose, like pk_sub_sat_s below: written as a plain expression hipcc re-associates
// the insert tree of the decision gather into v_and + v_or3 (11 instructions instead of 7); through
// __builtin_amdgcn_bitop3_b32 it keeps the 7, but without the asm statements as anchors the machine scheduler stretches a step
// over 175 registers (146 with them; held to 168 it spills), which costs the two-update-waves-plus-chainback residency.  The
// price of the asm: the hazard recogniser pads some of its uses with s_nop (2 per step at K = 7)
VIT_DEV u32 bfi_s(u32 mask, u32 a, u32 b) {
    u32 d;
#ifdef VIT_REG_BFI_VOP3
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b));
#else
    // v_bitop3_b32 with all three operands in VGPRs issues in 2.7 clocks where v_bfi_b32 (and v_bitop3 with a scalar or literal
    // operand) takes 4.6 (profiles/r4_op_rates.txt): the masks are pinned in vector registers (truth table 0xE4: s2 ? s0 : s1)
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(d) : "v"(a), "v"(b), "v"(mask));
#endif
    return d;
}
VIT_DEV u32 pk_min_s(u32 a, u32 b) {
    return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(s16x2_t, a), __builtin_bit_cast(s16x2_t, b)));
}
// per-half SIGNED saturating subtract: the sign bit of each half is exactly (a < b) as signed 16-bit, whatever the distance.
// Kept as inline asm on purpose: __builtin_elementwise_sub_sat selects the same instruction, but the sixteen asm statements of
// a step are what keeps the machine scheduler from interleaving whole trellis steps (223 registers instead of 146 at K = 7;
// held to 168 it spills 228 - 832 bytes per lane)
VIT_DEV u32 pk_sub_sat_s(u32 a, u32 b) {
    u32 d;
    asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// the same with a wave-uniform first operand taken straight from an SGPR (no v_mov)
VIT_DEV u32 pk_sub_sat_s_uniform(u32 a_uniform, u32 b) {
    u32 d;
    asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "s"(a_uniform), "v"(b));
    return d;
}

template <class F, int... Is>
VIT_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
VIT_DEV void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr int clcm(int a, int b) { return a / cgcd(a, b) * b; }

// ---- compile-time description of one code -------------------------------------------------------------------------
template <int K_, int R_, u32 G0, u32 G1, u32 G2, u32 G3, int LANE_BITS_ = 2, u32 G4 = 0, u32 G5 = 0>
struct RegSpec {
    static constexpr int K = K_, R = R_;
    // R = 5, 6 (K = 7, 9): 2^R sums per step would be 4 - 8 KiB of LDS ring per step and wave.  The pattern is SPLIT instead: the
    // low RL polynomials and the high R - RL each get their own table (8 + 4 or 8 + 8 entries: the ring of an R = 4 code), and
    // the consumer forms E[p] = Elo[p_lo] + Ehi[p_hi], max_error - E[p] = (max_error - Elo[p_lo]) - Ehi[p_hi] with two packed
    // instructions per butterfly.  Everything below that speaks of "a part" means the bit range [off, off + n) of the pattern.
    static constexpr bool SPLIT = R_ > 4 && LANE_BITS_ == 2;     // (K < 7: every sum is formed in the lane, no table to split)
    static constexpr int RL = SPLIT ? 3 : R_, RH = R_ - RL;      // polynomials of the low / high part
    static constexpr int NPL = 1 << RL, NPH = SPLIT ? 1 << RH : 0;
    // table entries per step and frame pair: a power of two (R = 5: 8 + 4 padded to 16), because the producer forms its write
    // addresses by XOR on top of the row's base (q x row bytes must not share bits with the index)
    static constexpr int NPROW = (NPL + NPH) <= 8 ? (NPL + NPH <= 4 ? (NPL + NPH <= 2 ? NPL + NPH : 4) : 8) : 16;
    static_assert(!SPLIT || NPL + NPH <= 16, "split tables: at most 8 + 8 entries");
    static constexpr int SB = K - 1;              // state bits
    // GENERIC (all polynomials zero in the template arguments): ONE code object for EVERY polynomial set of this (K, R).  The
    // reference takes G at run time (viterbi_branch_table.h:34-55); the specialised kernels bake pat(r) into register choices and
    // ds_read offsets, which needs a compiler per set.  Here each butterfly h of a step fetches ITS OWN {E, max_error - E} pair from
    // the ring through a run-time address: the kernel's prologue forms, from the polynomials in its arguments, the table
    // [layout step][lane group q][butterfly h] of ring byte offsets ((pat_reg ^ pat_lane) << 7) in LDS; a step reads its row two
    // steps ahead (two ds_read_b128 at K = 7), ORs the lane's pair offset in (NREG / 2 two-pass-free v_or_b32) and issues NREG / 2
    // ds_read_b64 instead of 2^R.  The producer writes entry P at index P (bm_index is the identity for zero polynomials).
    static constexpr bool GENERIC = G0 == 0 && G1 == 0 && G2 == 0 && G3 == 0 && G4 == 0 && G5 == 0;
    static_assert(!GENERIC || R_ <= 4, "generic kernels: whole patterns (R <= 4)");
    // LANE_BITS = 2: two state-slot bits live in the lane index (lane bits 4 and 5), 16 frame pairs per wave;
    // LANE_BITS = 0 (small K): every state of a frame pair lives in ONE lane's registers, 64 frame pairs per wave
    static constexpr int LANE_BITS = LANE_BITS_;
    static constexpr int PAIRS = 64 >> LANE_BITS;              // frame pairs per wave
    static constexpr int TILE = 2 * PAIRS;                     // frames per wave
    static constexpr int REG_BITS = SB - LANE_BITS;
    static constexpr int NREG = 1 << REG_BITS;    // packed metric registers per lane
    static constexpr int DW = NREG >= 16 ? NREG / 16 : 1;  // decision dwords per lane per step
    static constexpr int SPS = 4 / DW;            // steps per 16-byte decision row
    static constexpr u32 SMASK = (1u << SB) - 1u;
    static constexpr u32 G(int i) { return i == 0 ? G0 : i == 1 ? G1 : i == 2 ? G2 : i == 3 ? G3 : i == 4 ? G4 : G5; }

    static constexpr u32 rotl(u32 x, int n) {
        n %= SB;
        return n == 0 ? (x & SMASK) : (((x << n) | (x >> (SB - n))) & SMASK);
    }
    static constexpr u32 rotr(u32 x, int n) { return rotl(x, (SB - (n % SB)) % SB); }
    static constexpr u32 parity(u32 v) {
        v ^= v >> 16; v ^= v >> 8; v ^= v >> 4; v ^= v >> 2; v ^= v >> 1;
        return v & 1u;
    }
    // R-bit branch pattern of the half-state-like value v: bit i = parity((v << 1) & G[i])   (viterbi_branch_table.h:48-51)
    static constexpr u32 pat(u32 v) {
        u32 p = 0;
        for (int i = 0; i < R; ++i) p |= parity((v << 1) & G(i)) << i;
        return p;
    }
    // does any butterfly carry pattern p?  pat() is linear, its image a subspace: a polynomial set that repeats a polynomial (DAB:
    // 109, 79, 83, 109) reaches only half of the 2^R patterns, and the producer makes no entry for the others
    static constexpr bool pat_used(u32 p) {
        if (GENERIC) return true;                 // the polynomials arrive at run time: every pattern may occur
        for (u32 v = 0; v <= SMASK; ++v)
            if (pat(v) == p) return true;
        return false;
    }
    // position of the decision bit of slot register r (frame half h) inside its decision dword r/16: the sign bits of
    // register pairs (r, r+8) are byte-gathered: byte 0/1 = registers 0-7 of frame A/B, byte 2/3 = registers 8-15 of A/B
    static constexpr u32 dec_bit(u32 r, u32 h) { return (r & 7u) + 8u * h + 16u * ((r >> 3) & 1u); }
    // slot bit exchanged by the butterflies of phase ph, and whether that is a lane bit
    static constexpr int pbit(int ph) { return SB - 1 - ph; }
    static constexpr bool lane_phase(int ph) { return pbit(ph) >= REG_BITS; }
    static constexpr int T = REG_BITS > 0 ? REG_BITS - 1 : 0;  // register bit a lane bit is exchanged with

    // pattern contributed by the register index r0 (bit of the butterfly cleared) in phase ph
    static constexpr u32 pat_reg(int ph, u32 r0) { return pat(rotl(r0, ph)); }
    // pattern contributed by lane group q in phase ph
    static constexpr u32 pat_lane(int ph, u32 q) {
        if (LANE_BITS == 0) return 0;
        if (!lane_phase(ph)) return pat(rotl(q << REG_BITS, ph));
        // after the lane<->register exchange: this lane's bit `lb` stands for slot register bit T, the butterfly bit
        // itself (slot lane bit lb) is 0 for the r0 member; the other lane bit is unchanged
        const int lb = pbit(ph) - REG_BITS;
        const u32 u = (q >> lb) & 1u, o = (q >> (1 - lb)) & 1u;
        u32 x = 0;
        if (u) x |= 1u << T;
        if (o) x |= 1u << (REG_BITS + (1 - lb));
        return pat(rotl(x, ph));
    }
    // ---- three-exchange lane phases (16-register codes, K = 7) ----------------------------------------------------------
    // The two lane phases of a period are adjacent (phases 0 and 1), so the exchange that would bring the first lane bit back
    // can instead fetch the second one: X(l1,rT) B X(l0,rT) B X(l1,rT) B serves phases 0, 1, 2 with three exchanges instead
    // of four (8 of 32 v_permlane*_swap per period less) and leaves the two lane bits swapped; the next period runs the
    // mirror image X(l0,rT) B X(l1,rT) B X(l0,rT) B and restores them.  The layout therefore has period 2 (K-1): lay(u, w) is
    // the LOGICAL slot bit the physical position w (0: lane bit 5 "l1", 1: lane bit 4 "l0", 2: register bit T) holds during
    // and after step u of a double period; register bits below T never move.  State 0 is slot 0 in every layout.
    static constexpr bool X3 = LANE_BITS == 2 && NREG == 16;
    static constexpr int PER = X3 ? 2 * SB : SB;              // period of the lane layout, in steps
    static constexpr int lay(int u, int w) {
        constexpr int HI = SB - 1, MID = SB - 2;
        const int v = ((u % PER) + PER) % PER;
        const int a = v == 0 ? T : v == 1 ? T : v <= SB - 1 ? MID : v == SB ? MID : HI;               // l1
        const int b = v == 0 ? MID : v == 1 ? HI : v <= SB - 1 ? HI : v == SB ? T : v == SB + 1 ? T : MID;   // l0
        const int c = v == 0 ? HI : v == 1 ? MID : v <= SB - 1 ? T : v == SB ? HI : v == SB + 1 ? MID : T;   // rT
        return w == 0 ? a : w == 1 ? b : c;
    }
    // the lane bit exchanged with register bit T in front of step u's butterflies: 1 = l1 (v_permlane32_swap), 0 = l0
    // (v_permlane16_swap), -1 = none
    static constexpr int xch(int u) {
        const int v = ((u % PER) + PER) % PER;
        return (v == 0 || v == 2 || v == SB + 1) ? 1 : (v == 1 || v == SB || v == SB + 2) ? 0 : -1;
    }
    // pattern contributed by lane group q (bit 1 = l1, bit 0 = l0) in step u of a double period
    static constexpr u32 pat_lane3(int u, u32 q) {
        u32 x = 0;
        if (q & 2u) x |= 1u << lay(u, 0);
        if (q & 1u) x |= 1u << lay(u, 1);
        return pat(rotl(x, u % SB));
    }
    // logical slot held by physical position (q, r) in the layout of step u
    static constexpr u32 slot3(int u, u32 q, u32 r) {
        u32 x = r & ((1u << T) - 1u);
        if ((r >> T) & 1u) x |= 1u << lay(u, 2);
        if (q & 2u) x |= 1u << lay(u, 0);
        if (q & 1u) x |= 1u << lay(u, 1);
        return x;
    }
    // what brings the decision dword of layout step v back to the canonical slot order: 0 nothing, 1 / 2 the half-dword
    // exchange with lane bit 5 / 4 (register bit T holds that lane bit's logical bit), 3 the swap of the two lane bits, 4 / 5 both
    static constexpr int tail_kind(int v) {
        if (!X3) return 0;
        const int A = lay(v, 0), B = lay(v, 1), C = lay(v, 2);
        const bool ex1 = A == T, ex0 = B == T;
        const int A2 = ex1 ? C : A;                 // what l1 holds after the exchange
        const bool lsw = A2 == SB - 2;
        return ex1 ? (lsw ? 4 : 1) : ex0 ? (lsw ? 5 : 2) : (lsw ? 3 : 0);
    }
    // ---- where a step's branch metrics sit in the LDS ring: the LANE BASIS of the step ----
    // Lane group q of a pair needs E[p ^ x_q] where x_q = q0 a ^ q1 b is linear in the two lane bits (a, b: the patterns the lane
    // bits contribute in the step's layout).  Written to the ring in pattern order that is a per-lane xor on every read address --
    // a table of (layout steps x patterns) address registers (48 at K = 7, R = 2; hipcc spilled the 128 of K = 9, R = 4) or one
    // v_xor in front of every ds_read (CDMA 2000: 28 per step, each stalling its read: 11 % of the kernel).  Instead the PRODUCER
    // lane writes entry P at index T P, T the change of basis that sends a -> 1 and b -> 2 (bm_basis), so that the readers'
    // index is T p ^ q: the lane part is q in EVERY step (or one bit of it where a and b are dependent: bm_form), i.e. four
    // per-lane base addresses for the whole kernel, and T p goes into the instruction's offset field.  The producer pays one
    // v_xor per pattern and group of four steps (the columns T e_j of its step are per-lane constants).
    static constexpr u32 lane_pat(int u, u32 q) { return LANE_BITS == 0 ? 0u : X3 ? pat_lane3(u % PER, q) : pat_lane(u % SB, q); }
    // ... restricted to the n pattern bits from bit `off` (a part of a split pattern; the whole pattern: off = 0, n = R)
    static constexpr u32 lane_pat_part(int u, u32 q, int off, int n) { return (lane_pat(u, q) >> off) & ((1u << n) - 1u); }
    // how the lane bits enter: 0: a, b independent (index ^ q); 1: a == b != 0 (^ q0^q1); 2: b == 0 (^ q0); 3: a == 0 (^ q1); 4: neither
    static constexpr int bm_form(int u, int off = 0, int n = R) {
        const u32 a = lane_pat_part(u, 1, off, n), b = lane_pat_part(u, 2, off, n);
        return (a != 0 && b != 0 && a != b) ? 0 : (a != 0 && a == b) ? 1 : (a != 0) ? 2 : (b != 0) ? 3 : 4;
    }
    static constexpr int bm_form_bits(int f) { return f == 0 ? 2 : f == 4 ? 0 : 1; }      // low index bits the lane part touches
    struct BmBasis { u32 v[8]; };
    static constexpr bool bm_in_span(const BmBasis& B, int n, u32 x) {
        for (u32 m = 0; m < (1u << n); ++m) {
            u32 sum = 0;
            for (int k = 0; k < n; ++k) if ((m >> k) & 1u) sum ^= B.v[k];
            if (sum == x) return true;
        }
        return false;
    }
    static constexpr BmBasis bm_basis(int u, int off = 0, int nb = R) {
        BmBasis B{};
        int n = 0;
        const u32 a = lane_pat_part(u, 1, off, nb), b = lane_pat_part(u, 2, off, nb);
        const int f = bm_form(u, off, nb);
        if (f == 0) { B.v[n++] = a; B.v[n++] = b; }
        else if (f == 1 || f == 2) B.v[n++] = a;
        else if (f == 3) B.v[n++] = b;
        for (int j = 0; j < nb && n < nb; ++j)
            if (!bm_in_span(B, n, 1u << j)) B.v[n++] = 1u << j;
        return B;
    }
    // T p: the coordinates of (the part of) pattern p in the basis of layout step u (bit k = coefficient of basis vector k)
    static constexpr u32 bm_index(int u, u32 p, int off = 0, int nb = R) {
        const BmBasis B = bm_basis(u, off, nb);
        for (u32 c = 0; c < (1u << nb); ++c) {
            u32 sum = 0;
            for (int k = 0; k < nb; ++k) if ((c >> k) & 1u) sum ^= B.v[k];
            if (sum == p) return c;
        }
        return 0;
    }
    // 64-bit lane mask: lanes whose group q has pattern bit i set in phase ph
    static constexpr uint64_t lane_mask(int ph, int i) {
        uint64_t m = 0;
        for (u32 q = 0; q < 4; ++q)
            if ((pat_lane(ph, q) >> i) & 1u) m |= 0xFFFFull << (16 * q);
        return m;
    }
};

// ---- branch metrics fetched in SUB-CHUNKS (K = 9 with sixteen patterns: CDMA 2000) --------------------------------------
// A step's 2^R sums {E[p], max_error - E[p]} are 32 registers at R = 4, 64 with the one-step look-ahead double buffer -- on
// top of 64 metrics and 32 live decision values that is 368 allocated registers and ONE wave per SIMD.  Here the butterflies of
// a step are taken in sub-chunks of CS = 4 and each sub-chunk fetches the (at most four) patterns the NEXT one needs: 16
// registers instead of 64, the price being 32 instead of 16 ds_read_b64 per step (the LDS has the room: a quarter of its
// cycles).  All compile time: which butterfly of a sub-chunk is the first to use a pattern, and the register slot it lands in.
template <class SP>
struct RegChunk {
#ifndef VIT_REG_CHUNK_CS
#define VIT_REG_CHUNK_CS 4
#endif
    static constexpr int CS = VIT_REG_CHUNK_CS;
    static constexpr int NSUB = SP::NREG / 2 / CS;
    // (total functions: the generic lambdas that call them are instantiated for every code, also where their results are unused)
    static constexpr int pb(int PH) { return SP::lane_phase(PH % SP::SB) ? SP::T : SP::pbit(PH % SP::SB); }   // classic lane phases only (no X3)
    static constexpr int r0(int PH, int h) {
        const int PB = pb(PH);
        return ((h >> PB) << (PB + 1)) | (h & ((1 << PB) - 1));
    }
    static constexpr u32 pat(int PH, int h) { return SP::pat_reg(PH % SP::SB, (u32)r0(PH, h)); }
    // `part`: 0 the whole pattern, 1 / 2 the low / high part of a split pattern (RegSpec::SPLIT): a sub-chunk fetches every
    // DISTINCT value of the part once
    static constexpr u32 pat_part(int PH, int h, int part) {
        const u32 p = pat(PH, h);
        return part == 0 ? p : part == 1 ? (p & (u32)(SP::NPL - 1)) : (p >> SP::RL);
    }
    static constexpr int first(int PH, int h, int part = 0) {
        if (SP::GENERIC) return h;          // polynomials at run time: nothing is known to repeat, every butterfly fetches its own pair
        for (int k = h / CS * CS; k < h; ++k)
            if (pat_part(PH, k, part) == pat_part(PH, h, part)) return k;
        return h;
    }
    static constexpr int slot(int PH, int h, int part = 0) {
        if (SP::GENERIC) return h % CS;
        const int f = first(PH, h, part);
        int n = 0;
        for (int k = h / CS * CS; k < f; ++k)
            if (first(PH, k, part) == k) ++n;
        return n;
    }
};

#ifdef VIT_HIP_CLOCK_STAMPS
// measurement build only (make EXTRA=-DVIT_HIP_CLOCK_STAMPS; include/vit_hip_experiments.h): lane 0 of every update wave stores
// s_memtime / s_memrealtime at entry and exit and where it ran -- the shader clock THE UPDATE WAVES saw, per XCD
inline uint64_t* g_clock_stamps = nullptr;      // [tiles][6] uint64 or null
#endif
struct RegUpdateArgs {
    const uint8_t* symbols;
    size_t sym_frame_stride_bytes;
    size_t sym_total_bytes;
    uint4* ws;
    size_t ws_tile_stride;  // in uint4 units
    void* metrics_out;      // [F][N] error_t or null
    uint64_t* renorm_sum;   // [F] or null
    const u32* start_state; // [F] or null
    const void* metrics_in; // [F][N] error_t: resume from these metrics (no reset); null => reset(start_state)
    u32 frames;
    u32 t_begin, t_end;     // trellis steps [t_begin, t_end) of every frame; the symbol chunk starts at step t_begin
    DevConfig cfg;
    u32 gen_G[6];           // the code's polynomials: read by the GENERIC kernels only (RegSpec::GENERIC)
#ifdef VIT_HIP_CLOCK_STAMPS
    uint64_t* stamps;
#endif
};

typedef int v4i32_t __attribute__((ext_vector_type(4)));
typedef int v2i32_t __attribute__((ext_vector_type(2)));
typedef int v3i32_t __attribute__((ext_vector_type(3)));
// 16 bytes of a frame's symbol stream through a bounds-checked buffer descriptor: bytes at or beyond the end of the
// caller's buffer read as zero in hardware (no branches, no over-read).  `voff` is dword aligned; when a frame does not
// start on a dword (odd strides: R = 3 with 8-bit symbols) `fix` is set wave-wide and the 16 bytes are funnel-shifted
// out of 20 with v_alignbyte.
VIT_DEV uint4 load_chunk(__amdgpu_buffer_rsrc_t rsrc, u32 voff, u32 mis, bool fix) {
    const v4i32_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
    uint4 r = make_uint4((u32)v.x, (u32)v.y, (u32)v.z, (u32)v.w);
    if (fix) {
        const u32 e = (u32)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff + 16, 0, 0);
        r.x = __builtin_amdgcn_alignbyte(r.y, r.x, mis);
        r.y = __builtin_amdgcn_alignbyte(r.z, r.y, mis);
        r.z = __builtin_amdgcn_alignbyte(r.w, r.z, mis);
        r.w = __builtin_amdgcn_alignbyte(e, r.w, mis);
    }
    return r;
}

// load_chunk for the entry block of a RESUMED call, whose first bytes may lie below the frame's chunk (steps before
// t_begin): an offset that wrapped below zero makes the hardware range check fail for the WHOLE access, including the valid
// dwords above zero, so every dword is fetched on its own and the ones below zero are skipped (per-tile offsets stay below
// 2^31: reg_update())
VIT_DEV uint4 load_chunk_entry(__amdgpu_buffer_rsrc_t rsrc, u32 voff, u32 mis, bool fix) {
    u32 w[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        const u32 o = voff + 4u * (u32)d;
        w[d] = ((int)o >= 0 && (d < 4 || fix)) ? (u32)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o, 0, 0) : 0u;
    }
    uint4 r = make_uint4(w[0], w[1], w[2], w[3]);
    if (fix) {
        r.x = __builtin_amdgcn_alignbyte(w[1], w[0], mis);
        r.y = __builtin_amdgcn_alignbyte(w[2], w[1], mis);
        r.z = __builtin_amdgcn_alignbyte(w[3], w[2], mis);
        r.w = __builtin_amdgcn_alignbyte(w[4], w[3], mis);
    }
    return r;
}

// RESUME = false: reset + update from step 0 (the throughput kernel; every t_begin / metrics_in branch below folds away).
// RESUME = true : the same steps for decoders that already hold state (vit_hip_update_batch_resume): metrics come from
//                 metrics_in, the first unrolled block is entered in the middle.
template <class SP, int SHIFT, bool RESUME = false>
VIT_DEV void reg_update_body(const RegUpdateArgs& a) {
    constexpr int SB = SP::SB, R = SP::R, NREG = SP::NREG, REG_BITS = SP::REG_BITS, DW = SP::DW, SPS = SP::SPS, T = SP::T;
    constexpr int SBY = SHIFT ? 1 : 2;   // sizeof(soft_t)
    constexpr int BPS = R * SBY;         // symbol bytes per trellis step per frame
    constexpr int NP = 1 << R;
    // GENERIC (RegSpec): polynomials at run time -- every butterfly of a step fetches its own pair: NH "patterns" per step
    constexpr bool GENERIC = SP::GENERIC;
    constexpr int NH = NREG / 2;
    constexpr int NPF = GENERIC ? NH : NP;         // {E, max_error - E} pairs a lane holds per step
    // LDSBM: the 4 q-lanes of a frame pair share the branch-metric work through LDS.  In every group of 4 trellis steps
    // lane (q, g) loads only the symbols of step 4J+q of its pair, computes that step's 2^R sums E[p] / max_error - E[p]
    // ONCE (with the unswapped high / low) and parks them in an LDS ring; every step each lane then fetches
    // E[p ^ pat_lane(phase, q)] with one ds_read_b64 per pattern through per-lane addresses.  22 VALU per lane and step
    // become 6 (+ 4 LDS reads that issue beside the other wave's VALU).
    constexpr bool LDSBM = SP::LANE_BITS == 2;
    constexpr int GROUP = 4;
    // unrolled block: whole phases, whole decision rows, and whole 16-byte symbol chunks / whole 4-step groups
    constexpr bool X3 = SP::X3;          // three-exchange lane phases: the lane layout has period PER = 2 (K-1)
    constexpr int PER = SP::PER;
    constexpr int U0 = clcm(clcm(PER, GROUP), SPS);
    // LDS ring, in steps: the slot of step t is t % RING, a compile-time constant because RING divides the block length; 16 KiB
    // per wave at most, so that eight waves still fit a CU (R = 4 with 6 state bits: 8 steps, block of 24)
    // K = 9, R = 3, 4: branch metrics fetched per sub-chunk of four butterflies instead of per step (RegChunk).  Their LDS ring
    // holds ONE group of four steps (8 KiB per wave at R = 4, so that eight update waves and two 40 KiB chainback workgroups share
    // a CU): the next group is produced INSIDE a group's last step, right behind the last fetch of the old one
    constexpr bool BMCHUNK = LDSBM && !SP::X3 && ((NREG >= 64 && NP >= 8) || (NREG >= 32 && NP >= 16) || (SP::GENERIC && NREG >= 32));
    using RC = RegChunk<SP>;
    // R = 5, 6: the pattern is split into a low and a high part with a table each (RegSpec::SPLIT): NPROW entries per step
    constexpr bool SPLIT = SP::SPLIT;
    constexpr int RL = SP::RL, RH = SP::RH, NPL = SP::NPL, NPH = SP::NPH, NPROW = SP::NPROW;
    static_assert(!SPLIT || LDSBM, "split patterns are an LDS-ring feature (K = 7, 9)");
    constexpr int RING = !LDSBM ? 1 : BMCHUNK ? GROUP : (U0 * NPROW * 128 <= 16384 ? U0 : 8);
    // the 16-register codes (K = 7) unroll two periods (24 steps) so that the symbol ring below can run six groups ahead
    constexpr int U = LDSBM ? clcm(clcm(U0, RING), NREG == 16 ? 24 : 1) : clcm(clcm(SB, 16 / cgcd(16, BPS)), SPS);
    constexpr int NCH = LDSBM ? 0 : U * BPS / 16;
    // depth of the symbol register ring, in groups of 4 steps (a divisor of the block): 6 groups = 24 steps of load latency
    // hidden for K = 7 (+1 % alone and overlapped over 3 groups), 2 for K = 9 (a step is four times longer there)
    constexpr int NG = !LDSBM ? 1 : ((U / GROUP) % 6 == 0 ? 6 : (U / GROUP) % 3 == 0 ? 3 : (U / GROUP) % 2 == 0 ? 2 : (U / GROUP) % 7 == 0 ? 7 : 1);
    constexpr int NDW = (BPS + 3) / 4 + 1;        // dwords that cover BPS bytes at any byte phase: 2 (BPS <= 4), 3 (BPS <= 8) or 4 (BPS <= 12)
    static_assert(!LDSBM || (BPS <= 12 && (U / GROUP) % NG == 0 && U % RING == 0), "LDSBM geometry");
    constexpr int ROW = NPROW * 16;               // uint2 {E, EB} entries per step: [pattern (SPLIT: low part, then high part)][pair g]
    __shared__ uint2 bm_ring[LDSBM ? RING * ROW : 1];
    static_assert(!GENERIC || (!SPLIT && NH >= 2 && (!BMCHUNK || RC::CS == 4)), "generic kernels: whole patterns");
    // GENERIC below K = 7 (all states of a pair in one lane, no ring): the lane parks the 2^R pairs it has just formed in LDS,
    // [step parity][pattern][lane], and every butterfly reads its own back through the same run-time offsets (pattern x 512 bytes)
    __shared__ uint2 gen_bm[GENERIC && !LDSBM ? 2 * NP * 64 : 1];
    constexpr int GEN_SHIFT = LDSBM ? 7 : 9;       // a table entry: pattern x (pairs of a row x 8 bytes)
    // GENERIC: ring byte offset of the entry butterfly h of lane group q reads in layout step UP: [UP][q][h], filled below
    __shared__ __attribute__((aligned(16))) u32 gen_tab[GENERIC ? PER * 4 * NH : 4];
    constexpr u32 BIAS2 = 0x80008000u;   // metrics are kept as (m ^ 0x8000): unsigned order == signed order of the biased value

    // this kernel is VALU-issue bound; the chainback of the previous batch may be co-resident on a second stream: let
    // these waves win the issue arbitration, the (latency/HBM-bound) bit chase fills the gaps (+4 % on the overlapped step)
    __builtin_amdgcn_s_setprio(2);
    const u32 T_BEGIN = RESUME ? a.t_begin : 0u;
    constexpr int PAIRS = SP::PAIRS, TILE = SP::TILE;
    const int lane = threadIdx.x & 63;
    const u32 g = lane & (PAIRS - 1), q = SP::LANE_BITS ? lane >> 4 : 0;
    const size_t tile = blockIdx.x;
    const u32 fA_raw = (u32)tile * TILE + g, fB_raw = fA_raw + PAIRS;
    const bool validA = fA_raw < a.frames, validB = fB_raw < a.frames;
    const u32 fA = validA ? fA_raw : a.frames - 1, fB = validB ? fB_raw : a.frames - 1;

    // descriptor over [first frame of this tile (rounded down to a dword), end of the symbol buffer): wave-uniform base,
    // 32-bit per-lane offsets
    const size_t tile_off = tile * TILE * a.sym_frame_stride_bytes;
    const u32 bmis = (u32)(((uintptr_t)a.symbols + tile_off) & 3u);
    // rounded up to whole dwords: the range check is per dword, and the dword holding the buffer's last bytes must not
    // be dropped (it cannot straddle an allocation granule)
    const size_t remain = (a.sym_total_bytes - tile_off + bmis + 3) & ~(size_t)3;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.symbols + tile_off - bmis), 0, remain > 0xFFFF0000ull ? 0xFFFF0000u : (u32)remain, 0x00020000);
    // byte offset of ABSOLUTE step 0 (mod 2^32): the chunk holds steps t_begin.. ; the look-ahead of a resumed call may
    // touch steps below t_begin -- those offsets land in the neighbouring frame's chunk or wrap beyond num_records (reads 0),
    // and nothing computed from them is used
    const u32 rawA = (fA - (u32)tile * TILE) * (u32)a.sym_frame_stride_bytes + bmis - T_BEGIN * (u32)BPS;
    const u32 rawB = (fB - (u32)tile * TILE) * (u32)a.sym_frame_stride_bytes + bmis - T_BEGIN * (u32)BPS;
    const u32 offA = rawA & ~3u, offB = rawB & ~3u, misA = rawA & 3u, misB = rawB & 3u;
    const bool fix = __builtin_amdgcn_ballot_w64((misA | misB) != 0) != 0;   // wave-uniform

    const u32 HIGH2 = (u32)(uint16_t)a.cfg.high * 0x10001u, LOW2 = (u32)(uint16_t)a.cfg.low * 0x10001u;
    const u32 MAXE2 = (u32)a.cfg.max_error * 0x10001u;
    // new_metric[0] >= threshold  <=>  biased metric > biased (threshold - 1); threshold == 0 means "always"
    const u32 THRM1B2 = ((u32)(uint16_t)(a.cfg.threshold - 1) * 0x10001u) ^ BIAS2;
    const u32 FORCE = a.cfg.threshold == 0 ? BIAS2 : 0u;
    // state 0 is slot 0: lane group q == 0, register 0
    // the per-step test "does any frame of this wave renormalise" in two instructions: max(m[0], TLANE) != TLANE  <=>  some
    // half of m[0] exceeds the threshold; lanes that do not hold state 0 compare against the largest value (never exceeded).
    // Threshold 0 ("always"): the maximum is pinned to the largest value and compared with a constant it can never equal.
    // The sign bits the rare branch needs are formed inside it.
    const u32 TLANE = (q == 0 && a.cfg.threshold != 0) ? THRM1B2 : 0x7FFF7FFFu;
    const u32 TCMP = (q == 0 && a.cfg.threshold == 0) ? 0u : TLANE;
    // v_perm selectors {own = bytes 4-7, partner = bytes 0-3} for the half-dword exchange after a lane-bit phase
    const u32 SELX32 = (lane & 32) ? 0x07060302u : 0x01000504u;
    const u32 SELX16 = (lane & 16) ? 0x07060302u : 0x01000504u;
    // ds_bpermute addresses: the lane with the two lane-group bits swapped (q = 1 <-> q = 2), and the exchange partners
    const u32 LSWADDR = 4u * (u32)((lane & 15) | ((lane & 16) << 1) | ((lane & 32) >> 1));
    const u32 XADDR32 = 4u * (u32)(lane ^ 32), XADDR16 = 4u * (u32)(lane ^ 16);
    // the bit-field-insert masks of the decision gather, pinned in VECTOR registers (bfi_s): opaque to the optimiser, which would
    // otherwise fold them back into literals
    u32 M55 = 0x55555555u, M33 = 0x33333333u, M0F = 0x0F0F0F0Fu;
#ifndef VIT_REG_BFI_VOP3
    asm volatile("" : "+v"(M55), "+v"(M33), "+v"(M0F));
#endif

    // the branch pattern contributed by this lane's group q is folded in by exchanging `high` and `low` per polynomial:
    // per-lane constants, one pair per (phase, polynomial)
    u32 HIc[LDSBM ? 1 : SB][R], LOc[LDSBM ? 1 : SB][R];
    if constexpr (!LDSBM) {
        static_for<SB>([&](auto pc) __attribute__((always_inline)) {
            constexpr int ph = decltype(pc)::value;
            static_for<R>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr uint64_t LM = SP::lane_mask(ph, i);
                const bool swapped = (LM >> lane) & 1ull;
                HIc[ph][i] = swapped ? LOW2 : HIGH2;
                LOc[ph][i] = swapped ? HIGH2 : LOW2;
            });
        });
    }
    // LDSBM: byte addresses inside the ring (RegSpec: lane basis).  READ: bm_rd[form][l] = entry (l ^ lane part of the form) of pair
    // g, the rest of the index is an instruction offset; at most 4 + 2 + 2 + 2 + 1 registers, the forms a code never uses fold
    // away.  WRITE: its own step of a group (step 4J+q: row q of the group), entry 0 of pair g
    u32 bm_rd[5][4];
    const u32 bm_wr = (q * ROW + g) * 8u;
    // the producer's columns T e_j (x 128 bytes) of ITS step of a group: per-lane constants of the group's place in the layout
    // period, NCLS places.  Picked out of a packed compile-time constant by shift: written as selects hipcc turned them into
    // exec-mask branches inside the block loop
    constexpr int NCLS = LDSBM ? clcm(PER, GROUP) / GROUP : 1;
    u32 bm_col[NCLS][R];
    if constexpr (LDSBM) {
        static_for<NCLS>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            static_for<R>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                // (SPLIT: columns 0 .. RL-1 belong to the low part's basis, the others to the high part's)
                constexpr int off = j < RL ? 0 : RL, nb = j < RL ? RL : RH;
                constexpr u32 PACK = SP::bm_index((GROUP * c + 0) % PER, 1u << (j - off), off, nb) | (SP::bm_index((GROUP * c + 1) % PER, 1u << (j - off), off, nb) << 4) |
                                     (SP::bm_index((GROUP * c + 2) % PER, 1u << (j - off), off, nb) << 8) | (SP::bm_index((GROUP * c + 3) % PER, 1u << (j - off), off, nb) << 12);
                bm_col[c][j] = ((PACK >> (4u * q)) & 15u) << 7;
            });
        });
        const u32 lane_part[5] = {q, (q ^ (q >> 1)) & 1u, q & 1u, (q >> 1) & 1u, 0u};
        static_for<5>([&](auto fc) __attribute__((always_inline)) {
            constexpr int f = decltype(fc)::value;
            static_for<4>([&](auto lc) __attribute__((always_inline)) {
                constexpr u32 l = decltype(lc)::value;
                bm_rd[f][l] = (((l ^ lane_part[f]) << 4) + g) * 8u;
            });
        });
    }
    // LDSBM symbol fetch: the BPS bytes of step 4J+q sit at byte rawX + BPS*q + 4*BPS*J of the tile: NDW dwords from the
    // dword below and per-lane v_alignbyte (4*BPS is a multiple of 4, so the shift is the same for every group)
    const u32 gvA = (rawA + BPS * q) & ~3u, gsA = (rawA + BPS * q) & 3u;
    const u32 gvB = (rawB + BPS * q) & ~3u, gsB = (rawB + BPS * q) & 3u;

    // ---- reset (viterbi_decoder_core.h:202-211): phase 0, slot == state; or resume: slot x holds state rotl^ph(x) ----
    u32 m[NREG];
    if (RESUME) {
        const int ph = (int)(T_BEGIN % SB);
        const int lu = (int)((T_BEGIN + (u32)PER - 1u) % (u32)PER);   // the layout step T_BEGIN - 1 left behind (canonical before step 0)
        constexpr size_t N = (size_t)1 << SB;
        static_for<NREG>([&](auto rc) __attribute__((always_inline)) {
            constexpr u32 r = decltype(rc)::value;
            const u32 x = X3 ? SP::slot3(lu, q, r) : ((q << REG_BITS) | r);
            const u32 s = ((x << ph) | (x >> (SB - ph))) & SP::SMASK;
            u32 lo, hi;
            if (SHIFT) {
                lo = (u32)((const uint8_t*)a.metrics_in)[(size_t)fA * N + s] << 8;
                hi = (u32)((const uint8_t*)a.metrics_in)[(size_t)fB * N + s] << 8;
            } else {
                lo = ((const uint16_t*)a.metrics_in)[(size_t)fA * N + s];
                hi = ((const uint16_t*)a.metrics_in)[(size_t)fB * N + s];
            }
            m[r] = (lo | (hi << 16)) ^ BIAS2;
        });
    } else {
        const u32 sA = a.start_state ? (a.start_state[fA] & SP::SMASK) : 0u;
        const u32 sB = a.start_state ? (a.start_state[fB] & SP::SMASK) : 0u;
        static_for<NREG>([&](auto rc) __attribute__((always_inline)) {
            constexpr u32 r = decltype(rc)::value;
            const u32 x = (q << REG_BITS) | r;
            const u32 lo = (x == sA) ? a.cfg.init_start : a.cfg.init_non_start;
            const u32 hi = (x == sB) ? a.cfg.init_start : a.cfg.init_non_start;
            m[r] = (lo | (hi << 16)) ^ BIAS2;
        });
    }

    // the unrolled block that holds step t_begin starts at step tb0 (0 for a fresh decode)
    const u32 tb0 = T_BEGIN - T_BEGIN % (u32)U;
    uint4 cA[NCH ? NCH : 1], cB[NCH ? NCH : 1];
    static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        if constexpr (RESUME) {
            cA[c] = load_chunk_entry(rsrc, offA + tb0 * BPS + 16 * c, misA, fix);
            cB[c] = load_chunk_entry(rsrc, offB + tb0 * BPS + 16 * c, misB, fix);
        } else {
            cA[c] = load_chunk(rsrc, offA + 16 * c, misA, fix);
            cB[c] = load_chunk(rsrc, offB + 16 * c, misB, fix);
        }
    });
    // LDSBM: symbols of this lane's step of the next NG groups (register ring, slot = group % NG)
    u32 gA[NG][NDW], gB[NG][NDW];
    auto load_group = [&](u32 first_step, auto slotc) __attribute__((always_inline)) {
        constexpr int sl = decltype(slotc)::value;
        if constexpr (NDW == 2) {
            const v2i32_t va = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(gvA + first_step * BPS), 0, 0);
            const v2i32_t vb = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(gvB + first_step * BPS), 0, 0);
            gA[sl][0] = (u32)va.x; gA[sl][1] = (u32)va.y;
            gB[sl][0] = (u32)vb.x; gB[sl][1] = (u32)vb.y;
        } else if constexpr (NDW == 3) {
            const v3i32_t va = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(gvA + first_step * BPS), 0, 0);
            const v3i32_t vb = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(gvB + first_step * BPS), 0, 0);
            gA[sl][0] = (u32)va.x; gA[sl][1] = (u32)va.y; gA[sl][2] = (u32)va.z;
            gB[sl][0] = (u32)vb.x; gB[sl][1] = (u32)vb.y; gB[sl][2] = (u32)vb.z;
        } else {
            const v4i32_t va = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(gvA + first_step * BPS), 0, 0);
            const v4i32_t vb = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(gvB + first_step * BPS), 0, 0);
            gA[sl][0] = (u32)va.x; gA[sl][1] = (u32)va.y; gA[sl][2] = (u32)va.z; gA[sl][3] = (u32)va.w;
            gB[sl][0] = (u32)vb.x; gB[sl][1] = (u32)vb.y; gB[sl][2] = (u32)vb.z; gB[sl][3] = (u32)vb.w;
        }
    };

    uint64_t rsA = 0, rsB = 0;
    u32 dq[4] = {0, 0, 0, 0};
    u32 pend_acc = 0, pend_f1 = 0, pend_f2 = 0;   // three-exchange layouts: a step's decision dword and its cross-lane fetches, consumed one step later
    static_assert(!X3 || SP::tail_kind((U - 1) % PER) == 0, "the last step of an unrolled block leaves no decision fetch pending");
    uint4* ws_tile = a.ws + tile * a.ws_tile_stride;

    // E[p] = sum_i |bt_i - y_i| for every branch pattern p and EB[p] = max_error - E[p]  (scalar.h:66-73,107) of block
    // step `un` (un == U: step 0 of the NEXT block, whose chunks are already refilled).  Depends on symbols only, so it is
    // computed one step AHEAD, inside the basic block of the previous step's add-compare-select: its dependent chain
    // (perm -> sub -> neg -> max -> add -> sub) then overlaps ACS work instead of stalling the start of every step.
    u32 E[2][BMCHUNK || SPLIT ? 1 : NPF], EB[2][BMCHUNK || SPLIT ? 1 : NPF];
    // SPLIT, whole-step look-ahead (K = 7): the two parts as they come out of the ring -- {Elo, max_error - Elo} and Ehi
    u32 El[2][SPLIT && !BMCHUNK ? NPL : 1], EBl[2][SPLIT && !BMCHUNK ? NPL : 1], Eh[2][SPLIT && !BMCHUNK ? NPH : 1];
#ifndef VIT_REG_CHUNK_AHEAD
#define VIT_REG_CHUNK_AHEAD 1
#endif
    constexpr int AHEAD = VIT_REG_CHUNK_AHEAD;   // sub-chunks between a fetch and its use
    constexpr int NEB = 2 * AHEAD;               // ... and the buffers that takes (a power of two dividing the sub-chunks of a step)
    static_assert(RC::NSUB % NEB == 0 || !BMCHUNK, "buffer index = sub-chunk % NEB in every step");
    u32 Ec[NEB][RC::CS], EBc[NEB][RC::CS];       // BMCHUNK: [sub-chunk % NEB][slot]
    u32 Hc[NEB][SPLIT ? RC::CS : 1];             // ... SPLIT: the high parts (Ec / EBc then hold the low parts)
    // GENERIC: the offset table and the two rows in flight
    u32 TG[2][!GENERIC ? 1 : BMCHUNK ? RC::CS : NH];
    const u32 g8 = g * 8u;
    const u32* gen_mine = gen_tab + q * NH;        // this lane group's rows
    // BMCHUNK: the four offsets of sub-chunk s of block step un, into TG[s & 1] (sub-chunks are fetched in order, two rows in flight)
    auto gen_chunk_row = [&](auto unc, auto sc) __attribute__((always_inline)) {
        constexpr int un = decltype(unc)::value, s = decltype(sc)::value, UPn = (un % U) % PER;
        const uint4 v = *(const uint4*)(gen_mine + UPn * 4 * NH + 4 * s);
        TG[s & 1][0] = v.x; TG[s & 1][1] = v.y; TG[s & 1][2] = v.z; TG[s & 1][3] = v.w;
    };
    auto gen_row = [&](auto unc) __attribute__((always_inline)) {
        constexpr int un = decltype(unc)::value, UPn = (un % U) % PER;
        if constexpr (NH % 4 == 0) {
            static_for<NH / 4>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                const uint4 v = *(const uint4*)(gen_mine + UPn * 4 * NH + 4 * k);
                TG[un & 1][4 * k] = v.x; TG[un & 1][4 * k + 1] = v.y; TG[un & 1][4 * k + 2] = v.z; TG[un & 1][4 * k + 3] = v.w;
            });
        } else {
            static_for<NH>([&](auto kc) __attribute__((always_inline)) { TG[un & 1][decltype(kc)::value] = gen_mine[UPn * 4 * NH + decltype(kc)::value]; });   // K = 3: two butterflies
        }
    };
    auto branch_metrics = [&](auto unc) __attribute__((always_inline)) {
        if constexpr (!LDSBM) {
        constexpr int un = decltype(unc)::value;
        constexpr int us = un % U;            // position inside the (current or next) block
        constexpr int PHn = us % SB;
        constexpr int buf = un & 1;
        u32 A1[R], A0[R];
        static_for<R>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int o = us * BPS + i * SBY;
            constexpr int c = o / 16, d = (o % 16) / 4, sub = o % 4;
            const u32 wa = d == 0 ? cA[c].x : d == 1 ? cA[c].y : d == 2 ? cA[c].z : cA[c].w;
            const u32 wb = d == 0 ? cB[c].x : d == 1 ? cB[c].y : d == 2 ? cB[c].z : cB[c].w;
            constexpr u32 sel = SHIFT ? (0x0cu | ((u32)sub << 8) | (0x0cu << 16) | ((u32)(4 + sub) << 24))
                                      : ((u32)sub | ((u32)(sub + 1) << 8) | ((u32)(4 + sub) << 16) | ((u32)(5 + sub) << 24));
            const u32 Y = __builtin_amdgcn_perm(wb, wa, sel);   // (frame A | frame B << 16) in the device's 16-bit domain
            // error_t(get_abs(soft_t(expected - sym)))  (scalar.h:68-71) for expected = high and = low
            const u32 d1 = pk_sub(HIc[PHn][i], Y), d0 = pk_sub(LOc[PHn][i], Y);
            A1[i] = pk_max_s(d1, pk_sub(0u, d1));
            A0[i] = pk_max_s(d0, pk_sub(0u, d0));
        });
        static_for<NP>([&](auto pc) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            u32 e = (p & 1) ? A1[0] : A0[0];
            static_for<R - 1>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value + 1;
                e = pk_add(e, ((p >> i) & 1) ? A1[i] : A0[i]);
            });
            if constexpr (GENERIC) gen_bm[(buf * NP + p) * 64 + lane] = make_uint2(e, pk_sub(MAXE2, e));
            else {
                E[buf][p] = e;
                EB[buf][p] = pk_sub(MAXE2, e);
            }
        });
        if constexpr (GENERIC) {
            // one wavefront per workgroup, LDS operations complete in order: the reads below see the pairs just written
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            static_for<NH>([&](auto hc) __attribute__((always_inline)) {
                constexpr int h = decltype(hc)::value;
                const uint2 v = *(const uint2*)((const char*)gen_bm + (TG[buf][h] | g8) + buf * NP * 512);
                E[buf][h] = v.x;
                EB[buf][h] = v.y;
            });
            gen_row(std::integral_constant<int, un + 2>{});
        }
        }
    };
    static_assert(U % 2 == 0, "E double buffer alternates per step");
    // LDSBM producer: this lane's step of the group whose first step has ring slot `slot0`, from symbol ring slot `sl`
    // (`gs`: the group's first step as a step of the unrolled block; lane group q makes step gs + q)
    auto bm_produce = [&](auto gsc, auto slc) __attribute__((always_inline)) {
        constexpr int gs = decltype(gsc)::value % U, slot0 = gs % RING, sl = decltype(slc)::value;
        u32 wa[NDW - 1], wb[NDW - 1];             // the step's bytes 4k .. 4k+3 of frame A / B
        static_for<NDW - 1>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            wa[k] = __builtin_amdgcn_alignbyte(gA[sl][k + 1], gA[sl][k], gsA);
            wb[k] = __builtin_amdgcn_alignbyte(gB[sl][k + 1], gB[sl][k], gsB);
        });
        u32 A1[R], A0[R];
        static_for<R>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int k = (i * SBY) / 4, sub = (i * SBY) % 4;
            constexpr u32 sel = SHIFT ? (0x0cu | ((u32)sub << 8) | (0x0cu << 16) | ((u32)(4 + sub) << 24))
                                      : ((u32)sub | ((u32)(sub + 1) << 8) | ((u32)(4 + sub) << 16) | ((u32)(5 + sub) << 24));
            const u32 Y = __builtin_amdgcn_perm(wb[k], wa[k], sel);
            const u32 d1 = pk_sub(HIGH2, Y), d0 = pk_sub(LOW2, Y);
            A1[i] = pk_max_s(d1, pk_sub(0u, d1));
            A0[i] = pk_max_s(d0, pk_sub(0u, d0));
        });
        // entry P goes to index T P of the lane's step: the address of P is the xor of the columns bm_col[][j] = T e_j of its set
        // bits, built from the address of P without its lowest set bit.  K = 9 (64 metric registers per lane): waddr[0] is laundered
        // so that hipcc does not hoist the 2^R - 1 addresses of every group of the period out of the block loop, only the columns
        // stay resident and the xors run once per group
        constexpr int cls = (gs % (NCLS * GROUP)) / GROUP;
        // one part of the pattern (the whole of it unless SPLIT): polynomials off .. off + nb - 1, entries from `base` of the step's row
        auto emit_part = [&](auto offc, auto nbc, auto basec) __attribute__((always_inline)) {
            constexpr int off = decltype(offc)::value, nb = decltype(nbc)::value, base = decltype(basec)::value, NPP = 1 << nb;
            u32 waddr[NPP];
            waddr[0] = bm_wr;
            if constexpr (BMCHUNK) asm volatile("" : "+v"(waddr[0]));    // (K = 7: registers to spare -- the addresses stay resident, no xor in the loop)
            static_for<NPP>([&](auto pc) __attribute__((always_inline)) {
                constexpr int p = decltype(pc)::value;
                if constexpr (p != 0) {
                    constexpr int j = __builtin_ctz((unsigned)p);
                    waddr[p] = waddr[p & (p - 1)] ^ bm_col[cls][off + j];
                }
                if constexpr (SPLIT || SP::pat_used((u32)p)) {      // (whole patterns only: every part of a split pattern occurs)
                u32 e = (p & 1) ? A1[off] : A0[off];
                static_for<nb - 1>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value + 1;
                    e = pk_add(e, ((p >> i) & 1) ? A1[off + i] : A0[off + i]);
                });
                char* dst = (char*)bm_ring + waddr[p] + slot0 * ROW * 8 + base * 128;
                if constexpr (base == 0) *(uint2*)dst = make_uint2(e, pk_sub(MAXE2, e));   // {E, max_error - E} (SPLIT: of the low part)
                else *(u32*)dst = e;                                                       // the high part's sum alone
                }
            });
        };
        emit_part(std::integral_constant<int, 0>{}, std::integral_constant<int, RL>{}, std::integral_constant<int, 0>{});
        if constexpr (SPLIT) emit_part(std::integral_constant<int, RL>{}, std::integral_constant<int, RH>{}, std::integral_constant<int, NPL>{});
        // one wavefront per workgroup: LDS operations of a wave complete in order, so the other lanes' reads that follow
        // in program order see these rows; the fence only keeps the compiler from moving them across
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // where (part of) pattern `pv` of block step `us` sits in the ring for THIS lane: per-lane base of the step's lane form + the
    // rest of the index as an instruction offset (part 0: the whole pattern, 1 / 2: low / high part of a split pattern)
    auto bm_entry = [&](auto usc, auto pvc, auto partc) __attribute__((always_inline)) -> const char* {
        constexpr int us = decltype(usc)::value, part = decltype(partc)::value, PHn = us % PER;
        constexpr u32 pv = decltype(pvc)::value;
        constexpr int off = part == 2 ? RL : 0, nb = part == 0 ? R : part == 1 ? RL : RH, base = part == 2 ? NPL : 0;
        constexpr int f = SP::bm_form(PHn, off, nb), w = SP::bm_form_bits(f);
        constexpr u32 tp = SP::bm_index(PHn, pv, off, nb);
        return (const char*)bm_ring + bm_rd[f][tp & ((1u << w) - 1u)] + (((tp >> w) << w) << 7) + base * 128 + (us % RING) * ROW * 8;
    };
    if constexpr (GENERIC) {
        // pat(v): bit i = parity((v << 1) & G[i])  (viterbi_branch_table.h:48-51), with the polynomials of the kernel arguments
        auto pat_rt = [&](u32 v) __attribute__((always_inline)) -> u32 {
            u32 p = 0;
#pragma unroll
            for (int i = 0; i < R; ++i) p |= ((u32)__builtin_popcount((v << 1) & a.gen_G[i]) & 1u) << i;
            return p;
        };
        for (u32 e = (u32)lane; e < (u32)(PER * 4 * NH); e += 64u) {
            const u32 h = e % (u32)NH, qq = (e / (u32)NH) & 3u;
            const int up = (int)(e / (u32)(4 * NH)), ph = up % SB;
            // the register bit the butterflies of this step pair on (the step body's PB) and butterfly h's lower register
            const bool lp = !X3 && SP::lane_phase(ph);
            const int pb = (lp || (X3 && ph < SB - REG_BITS + 1)) ? T : SP::pbit(ph);
            const u32 r0 = ((h >> pb) << (pb + 1)) | (h & ((1u << pb) - 1u));
            // what the lane group contributes: RegSpec::pat_lane3 / pat_lane with the run-time pat
            u32 x = 0;
            if (X3) {
                if (qq & 2u) x |= 1u << SP::lay(up, 0);
                if (qq & 1u) x |= 1u << SP::lay(up, 1);
            } else if (!lp) {
                x = qq << REG_BITS;
            } else {
                const int lb = SP::pbit(ph) - REG_BITS;
                if ((qq >> lb) & 1u) x |= 1u << T;
                if ((qq >> (1 - lb)) & 1u) x |= 1u << (REG_BITS + (1 - lb));
            }
            gen_tab[e] = (pat_rt(SP::rotl(r0, ph)) ^ pat_rt(SP::rotl(x, ph))) << GEN_SHIFT;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (BMCHUNK) {
            gen_chunk_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            gen_chunk_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        } else {
            gen_row(std::integral_constant<int, 0>{});
            gen_row(std::integral_constant<int, 1>{});
        }
    }
    // LDSBM consumer: this lane's view of block step `un` (un == U: step 0 of the next block)
    auto bm_fetch = [&](auto unc) __attribute__((always_inline)) {
        constexpr int un = decltype(unc)::value;
        constexpr int us = un % U, buf = un & 1;
        if constexpr (GENERIC) {
            // this step's row of the offset table was read two steps ago (TG[un & 1]); the row of step un + 2 takes its place
            static_for<NH>([&](auto hc) __attribute__((always_inline)) {
                constexpr int h = decltype(hc)::value;
                const uint2 v = *(const uint2*)((const char*)bm_ring + (TG[buf][h] | g8) + (us % RING) * ROW * 8);
                E[buf][h] = v.x;
                EB[buf][h] = v.y;
            });
            gen_row(std::integral_constant<int, un + 2>{});
        } else if constexpr (!SPLIT) {
            static_for<NP>([&](auto pc) __attribute__((always_inline)) {
                constexpr int p = decltype(pc)::value;
                const uint2 v = *(const uint2*)bm_entry(std::integral_constant<int, us>{}, std::integral_constant<u32, (u32)p>{}, std::integral_constant<int, 0>{});
                E[buf][p] = v.x;
                EB[buf][p] = v.y;
            });
        } else {
            static_for<NPL>([&](auto pc) __attribute__((always_inline)) {
                constexpr int p = decltype(pc)::value;
                const uint2 v = *(const uint2*)bm_entry(std::integral_constant<int, us>{}, std::integral_constant<u32, (u32)p>{}, std::integral_constant<int, 1>{});
                El[buf][p] = v.x;
                EBl[buf][p] = v.y;
            });
            static_for<NPH>([&](auto pc) __attribute__((always_inline)) {
                constexpr int p = decltype(pc)::value;
                Eh[buf][p] = *(const u32*)bm_entry(std::integral_constant<int, us>{}, std::integral_constant<u32, (u32)p>{}, std::integral_constant<int, 2>{});
            });
        }
    };
    // BMCHUNK consumer: the patterns sub-chunk `s` of block step `un` needs (un == U: step 0 of the next block)
    auto bm_fetch_chunk = [&](auto unc, auto sc) __attribute__((always_inline)) {
        constexpr int un = decltype(unc)::value, s = decltype(sc)::value;
        constexpr int us = un % U, PHn = us % PER, buf = s % NEB;
        if constexpr (GENERIC) {
            // the sub-chunk's row of the offset table was read two sub-chunks ago (TG[s & 1]); the row two sub-chunks on takes its place
            static_assert(AHEAD == 1 && RC::NSUB % 2 == 0, "generic sub-chunk fetch: rows alternate between two buffers");
            static_for<RC::CS>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                const uint2 v = *(const uint2*)((const char*)bm_ring + (TG[s & 1][k] | g8) + (us % RING) * ROW * 8);
                Ec[buf][k] = v.x;
                EBc[buf][k] = v.y;
            });
            if constexpr (s + 2 < RC::NSUB) gen_chunk_row(unc, std::integral_constant<int, s + 2>{});
            else gen_chunk_row(std::integral_constant<int, un + 1>{}, std::integral_constant<int, s + 2 - RC::NSUB>{});
        } else
        static_for<RC::CS>([&](auto kc) __attribute__((always_inline)) {
            constexpr int h = s * RC::CS + decltype(kc)::value;
            if constexpr (!SPLIT) {
                if constexpr (RC::first(PHn, h) == h) {
                    constexpr u32 p = RC::pat(PHn, h);
                    constexpr int sl = RC::slot(PHn, h);
                    const uint2 v = *(const uint2*)bm_entry(std::integral_constant<int, us>{}, std::integral_constant<u32, p>{}, std::integral_constant<int, 0>{});
                    Ec[buf][sl] = v.x;
                    EBc[buf][sl] = v.y;
                }
            } else {
                // every distinct low part and every distinct high part of the sub-chunk's patterns, once each
                if constexpr (RC::first(PHn, h, 1) == h) {
                    constexpr int sl = RC::slot(PHn, h, 1);
                    const uint2 v = *(const uint2*)bm_entry(std::integral_constant<int, us>{}, std::integral_constant<u32, RC::pat_part(PHn, h, 1)>{}, std::integral_constant<int, 1>{});
                    Ec[buf][sl] = v.x;
                    EBc[buf][sl] = v.y;
                }
                if constexpr (RC::first(PHn, h, 2) == h) {
                    constexpr int sl = RC::slot(PHn, h, 2);
                    Hc[buf][sl] = *(const u32*)bm_entry(std::integral_constant<int, us>{}, std::integral_constant<u32, RC::pat_part(PHn, h, 2)>{}, std::integral_constant<int, 2>{});
                }
            }
        });
    };
    if constexpr (LDSBM) {
        static_for<NG>([&](auto sc) __attribute__((always_inline)) { load_group(tb0 + (u32)(decltype(sc)::value * GROUP), sc); });
        bm_produce(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        load_group(tb0 + (u32)(NG * GROUP), std::integral_constant<int, 0>{});
        if constexpr (BMCHUNK) static_for<AHEAD>([&](auto ac) __attribute__((always_inline)) { bm_fetch_chunk(std::integral_constant<int, 0>{}, ac); });
        else bm_fetch(std::integral_constant<int, 0>{});
    } else {
        branch_metrics(std::integral_constant<int, 0>{});
    }
    if constexpr (DW != 4 && RESUME) {
        // resuming inside a 16-byte decision row: keep the steps an earlier call already wrote
        if (T_BEGIN % SPS != 0) {
            const uint4 v = ws_tile[(size_t)(T_BEGIN / SPS) * 64 + lane];
            dq[0] = v.x; dq[1] = v.y; dq[2] = v.z; dq[3] = v.w;
        }
    }

    u32 t0 = tb0;
    uint4* ws_blk = ws_tile + (size_t)(tb0 / SPS) * 64;   // wave-uniform: decision rows of the current block (U is a multiple of SPS)
    // one unrolled block of U trellis steps; `guarded` adds the per-step bound check needed only by the last, partial block
    auto block = [&](auto guarded_c) __attribute__((always_inline)) {
        constexpr bool GUARDED = decltype(guarded_c)::value;
        static_for<U>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            constexpr int PH = u % SB;
            constexpr int cur = u & 1;
            if (!GUARDED || t0 + u < a.t_end) {
                if constexpr (LDSBM) {
                    if constexpr (!BMCHUNK && u % GROUP == 0) {
                        // first step of block group Jb: produce group Jb + 1, then refill its symbol slot with group Jb + 1 + NG
                        constexpr int Jb = u / GROUP;
                        constexpr int sl = (Jb + 1) % NG;
                        bm_produce(std::integral_constant<int, GROUP * (Jb + 1)>{}, std::integral_constant<int, sl>{});
                        load_group(t0 + (u32)(GROUP * (Jb + 1 + NG)), std::integral_constant<int, sl>{});
                    }
                    if constexpr (!BMCHUNK) bm_fetch(std::integral_constant<int, u + 1>{});
                } else {
                    branch_metrics(std::integral_constant<int, u + 1>{});
                }
                // ---- refill the symbol chunks whose last reader was that look-ahead (next needed ~U steps from now) ----
                static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    constexpr int last_use = (16 * c + 15) / BPS;          // step of this block that reads byte 16c+15
                    constexpr int first_use = (16 * c) / BPS;
                    static_assert(last_use >= 1, "a 16-byte chunk spans at least two steps");
                    if constexpr (last_use - 1 == u) {
                        if (t0 + U + first_use < a.t_end) {
                            const u32 o = (t0 + U) * BPS + 16 * c;
                            cA[c] = load_chunk(rsrc, offA + o, misA, fix);
                            cB[c] = load_chunk(rsrc, offB + o, misB, fix);
                        }
                    }
                });

                // a resumed call enters its first block in the middle: the steps below t_begin only keep the look-ahead
                // pipeline above running
                if (!GUARDED || !RESUME || t0 + u >= T_BEGIN) {
                constexpr int UP = u % PER;                        // step of the lane layout's (double) period
                constexpr bool LP = !X3 && SP::lane_phase(PH);     // classic scheme: exchange, butterflies, exchange back
                constexpr int XCH = X3 ? SP::xch(UP) : -1;         // three-exchange scheme: ONE exchange in front of the butterflies
                constexpr int PB = (LP || (X3 && PH < SB - REG_BITS + 1)) ? T : SP::pbit(PH);   // register bit the butterflies pair on
                constexpr int LB = LP ? SP::pbit(PH) - REG_BITS : (XCH >= 0 ? XCH : 0);
                auto lane_swap = [&](u32& x0, u32& x1) __attribute__((always_inline)) {
                    if constexpr (LB == 1) {
                        auto r2 = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
                        x0 = r2[0]; x1 = r2[1];
                    } else {
                        auto r2 = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
                        x0 = r2[0]; x1 = r2[1];
                    }
                };
                if constexpr (LP || XCH >= 0) {
                    static_for<NREG / 2>([&](auto hc) __attribute__((always_inline)) {
                        constexpr int r0 = decltype(hc)::value;   // bit T clear because r0 < NREG/2
                        lane_swap(m[r0], m[r0 | (1 << T)]);
                    });
                }
                // ---- add-compare-select, in place  (scalar.h:113-134) ----
                // biased metrics: x > y (unsigned, reference)  <=>  sign of sat_i16(y' - x'); min is the signed min
                u32 D[NREG];
                constexpr bool EARLY_TEST = !LP;            // register 0 is final as soon as its butterfly is done
                uint64_t early_any = 0;
                // ---- ... and, chunk by chunk, the gather of the 2 x NREG sign bits.  v_perm_b32 selectors 8..11 replicate the
                //      sign of a 16-bit half over a whole byte, so one perm of the register pair (r, r+8) yields four CLEAN bytes
                //      (0x00 / 0xFF): {A r, B r, A r+8, B r+8}; pair j then drops into bit j of every byte with a single v_and_or
                //      (no shift chain).  One dword per 16 registers: byte 0/1 = frame A/B registers 0-7, byte 2/3 = frame A/B
                //      registers 8-15, register k at bit k%8 ----
                constexpr u32 SIGN_BYTES = 0x0b0a0908u;
                u32 acc[DW];
                auto gather = [&](auto dc) __attribute__((always_inline)) {
                    constexpr int d = decltype(dc)::value;
                    auto dreg = [&](auto rc) __attribute__((always_inline)) -> u32 {   // fewer than 16 registers: K < 7
                        constexpr int rr = decltype(rc)::value;
                        if constexpr (rr < NREG) return D[rr]; else return 0u;
                    };
                    // a tree of bit-field inserts over the CLEAN bytes: 8 perms + 4 + 2 + 1 (one fewer than the chain, depth 4)
                    u32 P[8];
                    static_for<4>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int k = decltype(kc)::value;
                        constexpr int r = 16 * d + k;
                        P[k] = __builtin_amdgcn_perm(dreg(std::integral_constant<int, r + 8>{}), dreg(std::integral_constant<int, r>{}), SIGN_BYTES);
                        P[k + 4] = __builtin_amdgcn_perm(dreg(std::integral_constant<int, r + 12>{}), dreg(std::integral_constant<int, r + 4>{}), SIGN_BYTES);
                    });
                    const u32 q0 = bfi_s(M55, P[0], P[1]), q1 = bfi_s(M55, P[2], P[3]);
                    const u32 q2 = bfi_s(M55, P[4], P[5]), q3 = bfi_s(M55, P[6], P[7]);
                    acc[d] = bfi_s(M0F, bfi_s(M33, q0, q1), bfi_s(M33, q2, q3));
                };
                // 64-register codes (K = 9): a 16-register chunk is gathered as soon as its butterflies are done and the
                // scheduler may not move work across that point -- 64 metrics + 64 live decision values + branch metrics + index
                // registers do not fit 256 registers (hipcc spilled 24 bytes per lane and reloaded them twice per step)
                constexpr bool CHUNKED = NREG >= 64;
                static_for<NREG / 2>([&](auto hc) __attribute__((always_inline)) {
                    constexpr int h = decltype(hc)::value;
                    // h enumerates the NREG/2 register indices with bit PB clear
                    constexpr int r0 = ((h >> PB) << (PB + 1)) | (h & ((1 << PB) - 1));
                    constexpr int r1 = r0 | (1 << PB);
                    constexpr u32 p = SP::pat_reg(PH, (u32)r0);
                    if constexpr (BMCHUNK && h % RC::CS == 0) {
                        // first butterfly of a sub-chunk: ask for the next sub-chunk's patterns (the last one: for the first of step u + 1)
                        // (the first of a NEW group comes from the produce at the end of this step)
                        if constexpr (h / RC::CS + AHEAD < RC::NSUB) bm_fetch_chunk(uc, std::integral_constant<int, h / RC::CS + AHEAD>{});
                        else bm_fetch_chunk(std::integral_constant<int, u + 1>{}, std::integral_constant<int, h / RC::CS + AHEAD - RC::NSUB>{});
                        if constexpr (u % GROUP == GROUP - 1 && h / RC::CS + AHEAD == RC::NSUB - 1) {
                            // last step of block group Jb and the group's LAST fetch has just been issued (LDS operations of a wave
                            // complete in order): produce group Jb + 1 over it HERE, in the middle of the step, so that the new
                            // group's first sub-chunk is fetched a whole sub-chunk ahead like every other (produced at the end of the
                            // step its fetch sat right in front of its use: an exposed LDS round trip per group)
                            constexpr int Jb = u / GROUP;
                            constexpr int sl = (Jb + 1) % NG;
                            bm_produce(std::integral_constant<int, GROUP * (Jb + 1)>{}, std::integral_constant<int, sl>{});
                            load_group(t0 + (u32)(GROUP * (Jb + 1 + NG)), std::integral_constant<int, sl>{});
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    u32 e_p, eb_p;
                    if constexpr (BMCHUNK && !SPLIT) {
                        static_assert(RC::r0(PH, h) == r0 && RC::pat(PH, h) == p, "RegChunk mirrors the butterfly enumeration");
                        e_p = Ec[(h / RC::CS) % NEB][RC::slot(PH, h)];
                        eb_p = EBc[(h / RC::CS) % NEB][RC::slot(PH, h)];
                    } else if constexpr (BMCHUNK) {
                        static_assert(RC::r0(PH, h) == r0 && RC::pat(PH, h) == p, "RegChunk mirrors the butterfly enumeration");
                        // E[p] = Elo + Ehi, max_error - E[p] = (max_error - Elo) - Ehi: two packed instructions per butterfly
                        const u32 eh = Hc[(h / RC::CS) % NEB][RC::slot(PH, h, 2)];
                        e_p = pk_add(Ec[(h / RC::CS) % NEB][RC::slot(PH, h, 1)], eh);
                        eb_p = pk_sub(EBc[(h / RC::CS) % NEB][RC::slot(PH, h, 1)], eh);
                    } else if constexpr (SPLIT) {
                        const u32 eh = Eh[cur][p >> RL];
                        e_p = pk_add(El[cur][p & (NPL - 1)], eh);
                        eb_p = pk_sub(EBl[cur][p & (NPL - 1)], eh);
                    } else if constexpr (GENERIC) {
                        e_p = E[cur][h];
                        eb_p = EB[cur][h];
                    } else {
                        e_p = E[cur][p];
                        eb_p = EB[cur][p];
                    }
                    const u32 ma = m[r0], mb = m[r1];
#ifdef VIT_EXPERIMENT_ADD32
                    // TIMING EXPERIMENT ONLY (results wrong whenever a low half carries): the ceiling of VOP2 adds
                    const u32 x0 = ma + e_p, y0 = mb + eb_p;
                    const u32 x1 = ma + eb_p, y1 = mb + e_p;
#else
                    const u32 x0 = pk_add(ma, e_p), y0 = pk_add(mb, eb_p);   // -> next state (X|0)
                    const u32 x1 = pk_add(ma, eb_p), y1 = pk_add(mb, e_p);   // -> next state (X|1)
#endif
                    m[r0] = pk_min_s(x0, y0);
                    m[r1] = pk_min_s(x1, y1);
                    D[r0] = pk_sub_sat_s(y0, x0);   // sign set <=> x0 > y0 (strict: tie keeps predecessor 0)
                    D[r1] = pk_sub_sat_s(y1, x1);
                    // the threshold compare right behind the butterfly that makes state 0 (slot 0 = register 0 of lane group 0), into
                    // an SGPR pair: the branch at the end of the step then finds its condition long since ready instead of waiting
                    // for a v_cmp issued in front of it (K7 update alone -1.3 %, hard8 -2.5 %; not in the classic lane phases, whose
                    // exchange after the butterflies still moves register 0)
                    if constexpr (EARLY_TEST && r0 == 0) {
                        if constexpr (NREG >= 64) {
                            const u32 maskq0 = lane < 16 ? BIAS2 : 0u;
                            const u32 c = (pk_sub_sat_s_uniform(THRM1B2, m[0]) | FORCE) & maskq0;
                            asm volatile("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(early_any) : "v"(c));
                        } else {
                            const u32 pm = pk_max_s(m[0], TLANE);
                            asm volatile("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(early_any) : "v"(pm), "v"(TCMP));
                        }
                    }
                    if constexpr (CHUNKED) {
                        if constexpr (PB <= 3) {
                            // the pair bit lies inside a chunk: butterflies 8k .. 8k+7 complete registers 16k .. 16k+15
                            if constexpr ((h & 7) == 7) { gather(std::integral_constant<int, h / 8>{}); __builtin_amdgcn_sched_barrier(0); }
                        } else if constexpr ((h & 15) == 15) {
                            // pairs (r, r + 16) or (r, r + 32): sixteen butterflies complete the two chunks of their r0 and r1
                            constexpr int c0 = (((h - 15) >> PB) << (PB + 1) | ((h - 15) & ((1 << PB) - 1))) / 16;
                            gather(std::integral_constant<int, c0>{});
                            gather(std::integral_constant<int, c0 + (1 << (PB - 4))>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                });
                if constexpr (!CHUNKED) static_for<DW>([&](auto dc) __attribute__((always_inline)) { gather(dc); });
                if constexpr (LP && NREG > 16) {
                    static_for<DW / 2>([&](auto dc) __attribute__((always_inline)) {
                        constexpr int d = decltype(dc)::value;
                        lane_swap(acc[d], acc[d + DW / 2]);
                    });
                }
                if constexpr (LP) {
                    static_for<NREG / 2>([&](auto hc) __attribute__((always_inline)) {
                        constexpr int r0 = decltype(hc)::value;
                        lane_swap(m[r0], m[r0 | (1 << T)]);
                    });
                }
                // ---- decision rows: 16 bytes per lane, 1 KiB per wave, coalesced ----
                if constexpr (DW == 4) {
                    ws_blk[u * 64 + lane] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
                } else if constexpr (!X3) {
                    static_for<DW>([&](auto dc) __attribute__((always_inline)) {
                        constexpr int d = decltype(dc)::value;
                        dq[(u % SPS) * DW + d] = acc[d];
                    });
                    if constexpr (u % SPS == SPS - 1)
                        ws_blk[(u / SPS) * 64 + lane] = make_uint4(dq[0], dq[1], dq[2], dq[3]);
                } else {
                    // Three-exchange layouts: the decision dword goes back to the canonical slot order (rows stay what chainback
                    // and export read) through cross-lane fetches -- the half-dword exchange with the lane whose register bit T
                    // holds one of this lane's logical bits (EX: one ds_bpermute + v_perm: lane bit 0 keeps its low half and
                    // takes the partner's low half as its high half, lane bit 1 takes the partner's high half as its low half and
                    // keeps its high half), the lane permutation that swaps the two lane bits (LSW: one ds_bpermute), or both
                    // (two independent ds_bpermute + v_perm).  A ds_bpermute takes a hundred cycles and more to come back and a
                    // wave waits for it in order (s_waitcnt lgkmcnt): the fetches of step u are therefore CONSUMED in step u + 1,
                    // behind that step's add-compare-select (unguarded blocks; the last step of a block needs none, so nothing is
                    // pending across blocks).
                    auto finish_tail = [&](auto vc) __attribute__((always_inline)) {
                        constexpr int v = decltype(vc)::value;
                        constexpr int KV = SP::tail_kind(v % PER);
                        u32 fin;
                        if constexpr (KV == 1) fin = __builtin_amdgcn_perm(pend_acc, pend_f1, SELX32);
                        else if constexpr (KV == 2) fin = __builtin_amdgcn_perm(pend_acc, pend_f1, SELX16);
                        else if constexpr (KV == 3) fin = pend_f1;
                        else if constexpr (KV == 4) fin = __builtin_amdgcn_perm(pend_f1, pend_f2, SELX16);   // EX1 seen from lane LSW(l): its bit 5 is my bit 4
                        else fin = __builtin_amdgcn_perm(pend_f1, pend_f2, SELX32);
                        dq[v % SPS] = fin;
                        if constexpr (v % SPS == SPS - 1)
                            ws_blk[(v / SPS) * 64 + lane] = make_uint4(dq[0], dq[1], dq[2], dq[3]);
                    };
                    constexpr int KU = SP::tail_kind(UP);
                    constexpr bool DEFER = !GUARDED;
                    if constexpr (DEFER && u > 0) {
                        if constexpr (SP::tail_kind((u - 1) % PER) != 0) finish_tail(std::integral_constant<int, u - 1>{});
                    }
                    if constexpr (KU == 0) {
                        dq[u % SPS] = acc[0];
                        if constexpr (u % SPS == SPS - 1)
                            ws_blk[(u / SPS) * 64 + lane] = make_uint4(dq[0], dq[1], dq[2], dq[3]);
                    } else {
                        pend_acc = acc[0];
                        if constexpr (KU == 1) pend_f1 = (u32)__builtin_amdgcn_ds_bpermute((int)XADDR32, (int)acc[0]);
                        else if constexpr (KU == 2) pend_f1 = (u32)__builtin_amdgcn_ds_bpermute((int)XADDR16, (int)acc[0]);
                        else pend_f1 = (u32)__builtin_amdgcn_ds_bpermute((int)LSWADDR, (int)acc[0]);
                        if constexpr (KU == 4) pend_f2 = (u32)__builtin_amdgcn_ds_bpermute((int)(LSWADDR ^ 128u), (int)acc[0]);
                        if constexpr (KU == 5) pend_f2 = (u32)__builtin_amdgcn_ds_bpermute((int)(LSWADDR ^ 64u), (int)acc[0]);
                        if constexpr (!DEFER) finish_tail(uc);
                    }
                }
                // ---- renormalise when new_metric[0] >= threshold  (scalar.h:48-50, :139-153); state 0 is slot 0 (of lane group
                //      0, register 0) in every layout ----
                {
                    bool any;
                    if constexpr (EARLY_TEST) {
                        any = early_any != 0;
                    } else if constexpr (NREG >= 64) {
                        // 64-register codes (K = 9) have no register to spare for a second per-lane constant: three instructions of 356
                        const u32 maskq0 = lane < 16 ? BIAS2 : 0u;
                        any = __builtin_amdgcn_ballot_w64(((pk_sub_sat_s_uniform(THRM1B2, m[0]) | FORCE) & maskq0) != 0) != 0;
                    } else {
                        any = __builtin_amdgcn_ballot_w64(pk_max_s(m[0], TLANE) != TCMP) != 0;
                    }
                    if constexpr (NREG >= 64) {
                    // K = 9 (64 metric registers, 232 of the 240 the cap leaves): round 5's body, kept as it was.  The LDS-free body
                    // below and its folded subtraction take 40 % of this body's instructions away, and the 8-bit soft rate did rise
                    // by 1.6 % -- but hipcc then schedules the HOT path of this register-bound kernel 1 - 3 % slower for every decode
                    // type (same-box A/B, 65536 x 8192: SOFT16 10.96 -> 11.02 - 11.12 ms without the fold, 11.25 - 11.32 with it;
                    // HARD8 11.30 -> 11.42; SOFT8 12.40 -> 12.20), and two of the reference's three decode types hardly ever take the body
                    if (__builtin_expect(any, 0)) {
                        const u32 maskq = (SP::LANE_BITS == 0 || lane < 16) ? BIAS2 : 0u;              // formed here, not carried through the hot path
                        const u32 need2 = (pk_sub_sat_s_uniform(THRM1B2, m[0]) | FORCE) & maskq;   // sign bits: frame A / frame B
                        const u32 nq = SP::LANE_BITS ? (u32)__shfl((int)need2, (int)g) : need2;
                        const u32 msk = ((nq & 0x8000u) ? 0x0000FFFFu : 0u) | ((nq & 0x80000000u) ? 0xFFFF0000u : 0u);
                        u32 mn = m[0];
                        static_for<NREG - 1>([&](auto rc) __attribute__((always_inline)) {
                            mn = pk_min_s(mn, m[decltype(rc)::value + 1]);
                        });
                        if constexpr (SP::LANE_BITS) {
                            mn = pk_min_s(mn, (u32)__shfl_xor((int)mn, 16));
                            mn = pk_min_s(mn, (u32)__shfl_xor((int)mn, 32));
                        }
                        const u32 sub = (mn ^ BIAS2) & msk;   // the true (unbiased) minimum of each frame that renormalises
                        static_for<NREG>([&](auto rc) __attribute__((always_inline)) {
                            constexpr int r = decltype(rc)::value;
                            m[r] = pk_sub(m[r], sub);
                        });
                        rsA += (uint64_t)((sub & 0xFFFFu) >> SHIFT);
                        rsB += (uint64_t)((sub >> 16) >> SHIFT);
                    }
                    } else
                    if (__builtin_expect(any, 0)) {   // rare: the body goes out of line, the hot path falls through
                        u32 mn = m[0];
                        static_for<NREG - 1>([&](auto rc) __attribute__((always_inline)) {
                            mn = pk_min_s(mn, m[decltype(rc)::value + 1]);
                        });
                        u32 sub;
                        if constexpr (SP::LANE_BITS) {
                            // "rare" is every second to fifth step with the reference's 8-bit soft configuration (a frame renormalises
                            // every 50 - 150 steps, a wave holds 32): the body stays off the LDS.  Round 5 fetched the tripped-half mask
                            // of lane group 0 and the partial minima of the other three groups with three ds_bpermute, two of them
                            // awaited back to back (SOFT8 5 - 15 % behind SOFT16 in every code).  Now:
                            //  * which halves tripped: two 16-bit compares on state 0's register into scalar lane masks; the sixteen
                            //    bits of lane group 0 are spread over the other three groups on the SCALAR unit (shifts and ors that
                            //    issue beside the vector instructions) and come back as v_cndmask selectors;
                            //  * the minimum over the four lane groups: the two swap instructions the butterflies use -- rows (1, 0) and
                            //    (3, 2) exchanged, then the wave's halves -- with a packed minimum behind each.
                            const int16_t thr = (int16_t)(uint16_t)(THRM1B2 & 0xFFFFu);
                            uint64_t tA = __builtin_amdgcn_ballot_w64((int16_t)(uint16_t)(m[0] & 0xFFFFu) > thr);
                            uint64_t tB = __builtin_amdgcn_ballot_w64((int16_t)(uint16_t)(m[0] >> 16) > thr);
                            if (a.cfg.threshold == 0) tA = tB = 0xFFFFull;          // "always" (scalar.h:48: metric >= 0)
                            tA &= 0xFFFFull; tB &= 0xFFFFull;                       // lane group 0 holds state 0
                            tA |= tA << 16; tA |= tA << 32;
                            tB |= tB << 16; tB |= tB << 32;
                            auto s16 = __builtin_amdgcn_permlane16_swap(mn, mn, false, false);
                            mn = pk_min_s(s16[0], s16[1]);
                            auto s32 = __builtin_amdgcn_permlane32_swap(mn, mn, false, false);
                            mn = pk_min_s(s32[0], s32[1]);
                            const u32 unb = mn ^ BIAS2;           // the true (unbiased) minimum of both frames
                            u32 sa, sb;
                            asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(sa) : "v"(unb), "s"(tA));
                            asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(sb) : "v"(unb), "s"(tB));
                            sub = (sa & 0x0000FFFFu) | (sb & 0xFFFF0000u);
                        } else {
                            const u32 need2 = pk_sub_sat_s_uniform(THRM1B2, m[0]) | FORCE;   // sign bits: frame A / frame B
                            const u32 msk = ((need2 & 0x8000u) ? 0x0000FFFFu : 0u) | ((need2 & 0x80000000u) ? 0xFFFF0000u : 0u);
                            sub = (mn ^ BIAS2) & msk;             // the true (unbiased) minimum of each frame that renormalises
                        }
                        // Where the step's branch metrics are fewer than the state metrics (rate 1/2 at K = 5, 7, 8, 9: 8 registers
                        // against 16 - 64) the subtraction goes into the NEXT step's branch metrics, which the look-ahead has already
                        // put into registers: (m - c) + e == m + (e - c) in the wrapping arithmetic every candidate is formed in, so
                        // the next step's candidates, decisions and minima are bit for bit those of renormalised metrics -- and are
                        // themselves the renormalised values the steps behind it start from.  The last step of a call has no next
                        // step: there the metrics themselves are corrected (they are the call's result).
#ifndef VIT_REG_FOLD
#define VIT_REG_FOLD 1
#endif
                        constexpr bool FOLD = VIT_REG_FOLD && SP::LANE_BITS != 0 && !BMCHUNK && !SPLIT && 2 * NPF < NREG;
                        bool folded = false;
                        if constexpr (FOLD) {
                            if ((!GUARDED && u + 1 < U) || t0 + (u32)u + 1u < a.t_end) {
                                static_for<NPF>([&](auto pc) __attribute__((always_inline)) {
                                    constexpr int pp = decltype(pc)::value;
                                    E[(u + 1) & 1][pp] = pk_sub(E[(u + 1) & 1][pp], sub);
                                    EB[(u + 1) & 1][pp] = pk_sub(EB[(u + 1) & 1][pp], sub);
                                });
                                folded = true;
                            }
                        }
                        if (!folded) {
                            static_for<NREG>([&](auto rc) __attribute__((always_inline)) {
                                constexpr int r = decltype(rc)::value;
                                m[r] = pk_sub(m[r], sub);
                            });
                        }
                        rsA += (uint64_t)((sub & 0xFFFFu) >> SHIFT);
                        rsB += (uint64_t)((sub >> 16) >> SHIFT);
                    }
                }
                } else if constexpr (BMCHUNK) {
                    // a step below t_begin of a resumed call: only the look-ahead (and, in a group's last step, the produce) the
                    // skipped butterflies would have issued
                    if constexpr (u % GROUP == GROUP - 1) {
                        constexpr int Jb = u / GROUP;
                        constexpr int sl = (Jb + 1) % NG;
                        bm_produce(std::integral_constant<int, GROUP * (Jb + 1)>{}, std::integral_constant<int, sl>{});
                        load_group(t0 + (u32)(GROUP * (Jb + 1 + NG)), std::integral_constant<int, sl>{});
                    }
                    if constexpr (GENERIC) {
                        // the skipped butterflies' fetches would have moved the offset rows on: put the next step's first two in place
                        gen_chunk_row(std::integral_constant<int, u + 1>{}, std::integral_constant<int, 0>{});
                        gen_chunk_row(std::integral_constant<int, u + 1>{}, std::integral_constant<int, 1>{});
                    }
                    static_for<AHEAD>([&](auto ac) __attribute__((always_inline)) { bm_fetch_chunk(std::integral_constant<int, u + 1>{}, ac); });
                }
            }
        });
    };
    // whole blocks run unguarded; the (at most two) partial ones -- the entry block of a resumed call, the last block of any
    // call -- share ONE guarded copy of the unrolled body
    if constexpr (RESUME) {
        for (; t0 < a.t_end; t0 += U, ws_blk += (U / SPS) * 64) {
            if (t0 >= T_BEGIN && t0 + U <= a.t_end) block(std::false_type{});
            else block(std::true_type{});
        }
    } else {
        // Enter the loop with NO memory operation pending.  hipcc places one s_waitcnt vmcnt(N) per use for both ways into the
        // loop header and must take the stricter: coming from the prologue, the symbols the first block needs are the youngest
        // loads, so the steady state inherited vmcnt(2) where ten younger stores and loads may stay in flight -- every block
        // waited for loads issued four steps earlier and for all but the last two decision stores (K = 9; K = 7: vmcnt(7..15)
        // instead of 16 / 17).  With the drain the waits are what the ring depth allows; the kernels' times did not move (alone
        // or beside a chainback: the loads were back in time anyway), the margin against memory latency did
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (; t0 + U <= a.t_end; t0 += U, ws_blk += (U / SPS) * 64) block(std::false_type{});
        if (t0 < a.t_end) block(std::true_type{});
    }

    if constexpr (DW != 4) {
        if (a.t_end % SPS != 0) ws_tile[(size_t)(a.t_end / SPS) * 64 + lane] = make_uint4(dq[0], dq[1], dq[2], dq[3]);
    }

    // ---- final metrics in state order (get_error / m_metrics "old" buffer) ----
    if (a.metrics_out) {
        const int ph = (int)(a.t_end % SB);
        const int lu = (int)((a.t_end + (u32)PER - 1u) % (u32)PER);      // the lane layout step t_end - 1 left behind
        constexpr size_t N = (size_t)1 << SB;
        static_for<NREG>([&](auto rc) __attribute__((always_inline)) {
            constexpr u32 r = decltype(rc)::value;
            const u32 x = X3 ? SP::slot3(lu, q, r) : ((q << REG_BITS) | r);
            const u32 s = ((x << ph) | (x >> (SB - ph))) & SP::SMASK;   // state held by slot x after step t_end - 1
            const u32 v = m[r] ^ BIAS2;
            if (SHIFT) {
                if (validA) ((uint8_t*)a.metrics_out)[(size_t)fA * N + s] = (uint8_t)((v & 0xFFFFu) >> 8);
                if (validB) ((uint8_t*)a.metrics_out)[(size_t)fB * N + s] = (uint8_t)(v >> 24);
            } else {
                if (validA) ((uint16_t*)a.metrics_out)[(size_t)fA * N + s] = (uint16_t)(v & 0xFFFFu);
                if (validB) ((uint16_t*)a.metrics_out)[(size_t)fB * N + s] = (uint16_t)(v >> 16);
            }
        });
    }
    if (a.renorm_sum && q == 0) {
        if (validA) a.renorm_sum[fA] = rsA;
        if (validB) a.renorm_sum[fB] = rsB;
    }
}

// ---- chainback on the PLAN_REG layout -----------------------------------------------------------------------------
struct RegChainbackArgs {
    const uint4* ws;
    size_t ws_tile_stride;
    uint8_t* out;             // [F][ceil(L/8)]
    const u32* end_state;     // [F] or null
    u32 frames, L;
    u32 wave_priority;        // != 0: raise the waves' issue priority above the update kernel's (s_setprio 3)
};

template <class SP>
VIT_DEV void reg_chainback_coop_body(const RegChainbackArgs& a) {
    // K = 9 (64 registers: one 16-byte row per lane per step) and K = 8 (32 registers: a row holds two steps): the update
    // kernel's lane roles.  Every q-lane extracts the candidate bit of its slice, a ds_bpermute fetches the one of the slice
    // that owns the survivor's slot.  Rows are fetched NBUF rows ahead into a register ring; the main loop is branch-free so
    // the ring keeps counted vmcnt waits.
    static_assert(SP::LANE_BITS == 2 && SP::DW * SP::SPS == 4 && SP::NREG >= 32, "K = 8, 9 layout");
    constexpr int SB = SP::SB, NREG = SP::NREG, REG_BITS = SP::REG_BITS, DW = SP::DW, SPS = SP::SPS;
    constexpr int IGN = SB < 8 ? SB : 8;                       // ViterbiTracebackBuffer::get_layout (core.h:129-149)
    constexpr int SHIFT_STATE = 8 - IGN, SHIFT_TAIL = SB - IGN, TOTAL_BITS = SB + SHIFT_STATE;
    constexpr int PASS = 16;                                   // trellis steps per pass over the ring = 2 output bytes
    constexpr int NBUF = PASS / SPS;                           // rows in the ring
    // pass step k (0..15, descending t = t_top - k with t_top = 7 mod 8) completes byte (t - SB) / 8 iff (7 - k - SB) % 8 == 0
    constexpr int K0 = (((7 - SB) % 8) + 8) % 8;               // the first such step of a pass; the second is K0 + 8

    const int lane = threadIdx.x & 63;
    const u32 g = lane & 15, q = lane >> 4;
    const size_t tile = blockIdx.x;
    const u32 fA_raw = (u32)tile * 32 + g, fB_raw = fA_raw + 16;
    const bool validA = fA_raw < a.frames, validB = fB_raw < a.frames;
    const u32 fA = validA ? fA_raw : a.frames - 1, fB = validB ? fB_raw : a.frames - 1;   // surplus lanes: identical stores
    const uint4* rows = a.ws + tile * a.ws_tile_stride + lane;      // row of step t = rows[(t / SPS) * 64]
    const size_t out_stride = ((size_t)a.L + 7) / 8;
    uint8_t* outA = a.out + (size_t)fA * out_stride;
    uint8_t* outB = a.out + (size_t)fB * out_stride;

    u32 regA = (a.end_state ? (a.end_state[fA] & SP::SMASK) : 0u) << SHIFT_STATE;
    u32 regB = (a.end_state ? (a.end_state[fB] & SP::SMASK) : 0u) << SHIFT_STATE;

    // `sub`: which of the row's SPS steps (compile time in the ring, run time in the ragged ends)
    auto trace = [&](u32& reg, const uint4& v, u32 half, u32 ph1, u32 sub) __attribute__((always_inline)) {
        const u32 state = reg >> SHIFT_STATE;
        const u32 x = ((state >> ph1) | (state << (SB - ph1))) & SP::SMASK;   // slot of `state` after step t
        const u32 qs = x >> REG_BITS, rs = x & (NREG - 1);
        // the candidate bit of this lane's slice, branch-free: inside the lane's 128-bit row the decision of slot register rs
        // (frame half h, step `sub` of the row) sits at bit 32 DW sub + [rs >> 3 | h | rs & 7] = 32 DW sub + rs + (rs & ~7) + 8 h
        // (dword rs / 16 of the step, SP::dec_bit inside it): pick the 64-bit half, shift once.  (As a four-way select of the
        // dword hipcc built exec-mask branches: 78 instructions per step, 43 of them VALU.)
        const u32 pos = 32u * (u32)DW * sub + rs + (rs & ~7u) + 8u * half;
        const uint64_t lo64 = ((uint64_t)v.y << 32) | v.x, hi64 = ((uint64_t)v.w << 32) | v.z;
        const uint64_t w64 = (pos & 64u) ? hi64 : lo64;
        const u32 mine = (u32)(w64 >> (pos & 63u)) & 1u;
        const u32 bit = (u32)__shfl((int)mine, (int)(qs * 16 + g));          // the slice that owns slot x
        reg = (reg >> 1) | (bit << (TOTAL_BITS - 1));
    };
    auto emit = [&](u32 jb) __attribute__((always_inline)) {
        if (q == 0) {
            outA[jb] = (uint8_t)((regA >> SHIFT_TAIL) & 0xFFu);
            outB[jb] = (uint8_t)((regB >> SHIFT_TAIL) & 0xFFu);
        }
    };
    auto slow_step = [&](int t) __attribute__((always_inline)) {
        const uint4 v = rows[(size_t)(t / SPS) * 64];
        const u32 ph1 = (u32)((t + 1) % SB);
        trace(regA, v, 0u, ph1, (u32)(t % SPS));
        trace(regB, v, 1u, ph1, (u32)(t % SPS));
        const int j = t - SB;
        if ((j & 7) == 0) emit((u32)j >> 3);
    };

    int t = (int)a.L - 1 + SB;                                  // rows are consumed from t = L-1+SB down to SB
    while (t >= SB && (t & 7) != 7) slow_step(t--);             // down to a multiple-of-8 boundary (the last step of a row)
    if (t >= SB && t - (PASS - 1) >= SB) {
        uint4 buf[NBUF];
#pragma unroll
        for (int b = 0; b < NBUF; ++b) buf[b] = rows[(size_t)(t / SPS - b) * 64];
        // One pass over the ring = 16 steps = 2 output bytes per frame.  Inside the inner loop ONLY loads are outstanding
        // (see reg_chainback16_body: a pending store makes hipcc drain the ring with vmcnt(0) at every loop top): the bytes
        // are parked in LDS, [dword][lane][byte] so that neither the byte writes nor the dword reads of the flush conflict,
        // and flushed as (possibly unaligned) dword stores every KI passes.
        constexpr int BPP = PASS / 8;                            // bytes per frame and pass
        constexpr int KI = 128 / BPP;                            // 128 bytes per frame between flushes
        __shared__ u32 obufA[32 * 64], obufB[32 * 64];
        typedef u32 u32_unaligned __attribute__((aligned(1)));
        u32 ph_run = (u32)((t + 1) % SB);                        // (t + 1) % SB of the ring's top step; only used when SB != 8
        while (t - (PASS - 1) >= SB) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): retire the previous flush (and the ring)
            int it = 0;
            for (; it < KI && t - (PASS - 1) >= SB; ++it, t -= PASS) {
#pragma unroll
                for (int b = 0; b < NBUF; ++b) {
#pragma unroll
                    for (int sidx = SPS - 1; sidx >= 0; --sidx) {
                        const int k = b * SPS + (SPS - 1 - sidx);
                        // step t - k with t % 8 == 7: for SB == 8 the phase (t - k + 1) % 8 is a compile-time constant
                        const u32 ph1 = SB == 8 ? (u32)((8 - (k % 8)) % 8) : ph_run;
                        trace(regA, buf[b], 0u, ph1, (u32)sidx);
                        trace(regB, buf[b], 1u, ph1, (u32)sidx);
                        if (SB != 8) ph_run = ph_run == 0 ? SB - 1 : ph_run - 1;
                        if (k % 8 == K0) {
                            // byte (t - k - SB) / 8 is complete; slot counts down from the top of the flush
                            const int slot = 127 - (it * BPP + k / 8);
                            ((uint8_t*)obufA)[(slot >> 2) * 256 + lane * 4 + (slot & 3)] = (uint8_t)((regA >> SHIFT_TAIL) & 0xFFu);
                            ((uint8_t*)obufB)[(slot >> 2) * 256 + lane * 4 + (slot & 3)] = (uint8_t)((regB >> SHIFT_TAIL) & 0xFFu);
                        }
                    }
                    asm volatile("" : "+v"(regA), "+v"(regB) : : "memory");   // pin the chase in front of the refill
                    const int nxt = t / SPS - b - NBUF;
                    buf[b] = rows[(size_t)(nxt < 0 ? 0 : nxt) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the passes completed slots 127 down to 128 - it*BPP; the lowest of them is the byte of step t + PASS - (K0 + 8)
            if (q == 0) {
                const int nbytes = it * BPP;
                const u32 jb_lo = (u32)(t + PASS - (K0 + 8) - SB) >> 3;    // byte index of slot 128 - nbytes
                int sl = 128 - nbytes;
                // leading bytes up to a dword boundary of the slot space, then whole dwords
                for (; (sl & 3) != 0 && sl < 128; ++sl) {
                    outA[jb_lo + (u32)(sl - (128 - nbytes))] = ((const uint8_t*)obufA)[(sl >> 2) * 256 + lane * 4 + (sl & 3)];
                    outB[jb_lo + (u32)(sl - (128 - nbytes))] = ((const uint8_t*)obufB)[(sl >> 2) * 256 + lane * 4 + (sl & 3)];
                }
                for (; sl < 128; sl += 4) {
                    *(u32_unaligned*)(outA + jb_lo + (u32)(sl - (128 - nbytes))) = obufA[(sl >> 2) * 64 + lane];
                    *(u32_unaligned*)(outB + jb_lo + (u32)(sl - (128 - nbytes))) = obufB[(sl >> 2) * 64 + lane];
                }
            }
        }
    }
    while (t >= SB) slow_step(t--);
}

#ifndef VIT_REG_CB_RING_DEPTH_K7
#define VIT_REG_CB_RING_DEPTH_K7 4
#endif
// ---- lane-local chainback through an LDS ring (K = 7, K = 9): one lane per frame PAIR, 128 frames per wave ----
// ring slots of four 1 KiB rows + parked output dwords (512 B per iteration between flushes): 8 slots + 16 iterations at K = 9
// (40 KiB), 4 + 16 at K = 7 (24 KiB: three of these workgroups and the twelve update waves of a three-waves-per-SIMD batch share a
// CU's 160 KiB), 2 + 8 at K = 7 with sixteen branch patterns (12 KiB: the R = 4 update waves hold 16 KiB of branch-metric ring
// each, and eight of them plus two of these workgroups must fit a CU -- DAB Radio on the overlapped schedule)
constexpr int reg_cb_ring_depth(int nreg, int R) { return nreg == 16 ? (R >= 4 ? 2 : VIT_REG_CB_RING_DEPTH_K7) : 8; }
constexpr int reg_cb_ring_flush_iters(int nreg, int R) { return (nreg == 16 && R >= 4) ? 8 : 16; }
// the four tiles of a wave's ring slot sit VIT_REG_CB_STAGGER bytes further apart than their 1 KiB rows: with 0 the lanes of the four
// tiles read the same banks (a lane's byte sits at tile * 1024 + pair * 16 + ...: eight bank groups for 64 lanes; SQ_LDS_BANK_CONFLICT
// 65 - 74 % of the kernel's LDS-active cycles), with 4 the tiles fall on the banks between
#ifndef VIT_REG_CB_STAGGER
#define VIT_REG_CB_STAGGER 0
#endif
constexpr unsigned reg_cb_ring_slot_bytes() { return 4096u + 4u * (unsigned)VIT_REG_CB_STAGGER; }
constexpr unsigned reg_cb_ring_lds_bytes(int nreg, int R) {
    return (unsigned)reg_cb_ring_depth(nreg, R) * reg_cb_ring_slot_bytes() + (unsigned)reg_cb_ring_flush_iters(nreg, R) * 512u;
}
// A step's decisions of a tile are 64 lanes x 16 B / SPS and the chase needs ONE bit per frame of them.  Here the rows never
// touch a register: the wave streams the 1 KiB rows of FOUR tiles into an LDS ring with direct-to-LDS loads
// (global_load_lds_dwordx4, each row in its own order; counted vmcnt waits), and each lane reads the one BYTE that holds its
// survivor's bit.  The chase keeps the survivor's SLOT, not its state: the update's in-place layout moves no state between
// slots, so going back one step replaces exactly one bit of the slot index (position (SB - (t+1) % SB) % SB) by the decision
// bit -- a shift and a bit-field insert, no rotation.  9 vector instructions and one ds_read_u8 per frame and step, two dozen
// registers: the kernel fits beside whatever the update kernels leave of a SIMD (K = 9: two 240-register waves; K = 7: three
// 152-register ones), and its issue slots are a hundredth of theirs (K = 9, cooperative body: a tenth).
template <class SP>
VIT_DEV void reg_chainback_ring_body(const RegChainbackArgs& a) {
    static_assert(SP::LANE_BITS == 2 && (SP::NREG == 16 || SP::NREG == 64) && SP::DW * SP::SPS == 4 && SP::SB <= 8, "K = 7, 9 layouts");
    constexpr int SB = SP::SB, REG_BITS = SP::REG_BITS, DW = SP::DW, SPS = SP::SPS;
    constexpr int IGN = SB < 8 ? SB : 8;                       // ViterbiTracebackBuffer::get_layout (core.h:129-149)
    constexpr int SHIFT_STATE = 8 - IGN;                       // the 8-bit register holds the state in its top SB bits ...
    static_assert(SB - IGN == 0, "... and its low byte is the output byte (shift_tail = 0)");
    constexpr int ITER = 32;                                   // steps per iteration = one output dword per frame
    constexpr int ROWS_IT = ITER / SPS;                        // rows an iteration retires
    // ring depth in rows (divides ROWS_IT).  K = 9, beside an update kernel: 4, 8 and 16 rows move the same bytes per second
    // (the chase waits for issue slots, not for rows) but 16 -- 72 KiB of LDS per wave -- keep update waves off the CU (12.6 ->
    // 14.1 ms per batch)
    constexpr int D = reg_cb_ring_depth(SP::NREG, SP::R);
    static_assert(ROWS_IT % D == 0, "ring slots are compile-time constants");
    constexpr int KI = reg_cb_ring_flush_iters(SP::NREG, SP::R);   // iterations between flushes: 4 output bytes per frame each
    // the top step of every iteration: t = CT mod 32 (its last decoded bit j = t - SB is a multiple of 32)
    constexpr int CT = (ITER - 1 + SB) % ITER;
    // DYNAMIC LDS (reg_cb_ring_lds_bytes() at launch), on purpose: hipcc pads the register allocation of a kernel whose STATIC LDS
    // limits its occupancy up to the count that enforces that occupancy (40 KiB per one-wave workgroup = one wave per SIMD =
    // .amdhsa_next_free_vgpr 257 for the 22 registers this body uses), and a 264-register wave does not fit beside two update
    // waves.  With the size unknown at compile time the descriptor says what the body needs (tests/test_codegen.py reads it).
    static_assert(reg_cb_ring_lds_bytes(SP::NREG, SP::R) == D * reg_cb_ring_slot_bytes() + 2 * KI * 64 * 4, "launch size of the ring kernels");
    extern __shared__ uint4 reg_cb_dyn_lds[];
    uint4* const ring = reg_cb_dyn_lds;                        // [slot][tile of the wave][lane of the row]
    u32* const obuf = (u32*)((char*)reg_cb_dyn_lds + D * reg_cb_ring_slot_bytes());     // [frame half][iteration][lane]
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef __attribute__((address_space(1))) const void glb_void_t;
    typedef u32 u32_unaligned __attribute__((aligned(1)));

    const int lane = threadIdx.x & 63;
    const u32 n_tiles = (a.frames + 31u) >> 5;
    const u32 jt = (u32)lane >> 4, g = (u32)lane & 15u;
    // surplus tiles / frames redo the last one: the update kernel filled their slots with that frame's decisions, so the
    // stores are identical
    const u32 tl_raw = blockIdx.x * 4u + jt;
    const u32 tl = tl_raw < n_tiles ? tl_raw : n_tiles - 1;
    const u32 fA = tl * 32u + g < a.frames ? tl * 32u + g : a.frames - 1;
    const u32 fB = tl * 32u + g + 16u < a.frames ? tl * 32u + g + 16u : a.frames - 1;
    const size_t out_stride = ((size_t)a.L + 7) / 8;
    uint8_t* outA = a.out + (size_t)fA * out_stride;
    uint8_t* outB = a.out + (size_t)fB * out_stride;
    const uint8_t* my_rows = (const uint8_t*)(a.ws + (size_t)tl * a.ws_tile_stride);   // row of step t at + 1024 (t / SPS)
    const u32 src_off = (u32)lane * 16u;                                                // the row piece this lane moves
    const u32 pair_base = jt * (1024u + (u32)VIT_REG_CB_STAGGER) + g * 16u;               // the pair's q = 0 piece inside a ring slot

    int t = (int)a.L - 1 + SB;                                  // rows are consumed from t = L-1+SB down to SB
    // the output shift register, 32 bits wide: after 32 steps its four bytes are output bytes (reference: the top byte of an
    // 8-bit register that starts as end_state << shift_state) ...
    const u32 esA = a.end_state ? (a.end_state[fA] & SP::SMASK) : 0u, esB = a.end_state ? (a.end_state[fB] & SP::SMASK) : 0u;
    u32 RA = esA << (24 + SHIFT_STATE), RB = esB << (24 + SHIFT_STATE);
    // ... and the survivor's slot after step t: rotr_SB(state, (t + 1) % SB)
    u32 ph = (u32)(t + 1) % (u32)SB;
    u32 xA = ((esA >> ph) | (esA << (SB - ph))) & SP::SMASK, xB = ((esB >> ph) | (esB << (SB - ph))) & SP::SMASK;
    u32 pos = (SB - ph) % SB;                                   // the slot bit the decision of step t replaces (wave-uniform)

    // slot x = (q << REG_BITS) | r: q-piece x >> REG_BITS of the row, and inside its 16 bytes dword sidx * DW + (r >> 4), byte
    // 2 ((r >> 3) & 1) + half, bit r & 7 (SP::dec_bit)  =>  byte 4 DW sidx + 2 (r >> 3) + half
    auto byte_of = [&](u32 x, u32 base) __attribute__((always_inline)) -> u32 {
        return base + 256u * __builtin_amdgcn_ubfe(x, REG_BITS, 2) + 2u * __builtin_amdgcn_ubfe(x, 3, REG_BITS - 3);
    };
    auto step_back = [&](u32& R, u32& x, u32 byte, u32 p) __attribute__((always_inline)) {
        const u32 bit = __builtin_amdgcn_ubfe(byte, x & 7u, 1);
        R = (R >> 1) | (bit << 31);
        x = (x & ~(1u << p)) | (bit << p);
    };
    auto slow_step = [&](int tt) __attribute__((always_inline)) {
        const uint8_t* row = my_rows + (size_t)(tt / SPS) * 1024u + (u32)(tt % SPS) * (4u * DW);   // straight from the workspace
        const u32 bA = row[byte_of(xA, g * 16u)], bB = row[byte_of(xB, g * 16u) + 1u];
        step_back(RA, xA, bA, pos);
        step_back(RB, xB, bB, pos);
        pos = pos == SB - 1 ? 0u : pos + 1u;
        const int j = tt - SB;
        if ((j & 7) == 0) { outA[j >> 3] = (uint8_t)(RA >> 24); outB[j >> 3] = (uint8_t)(RB >> 24); }
    };

    while (t >= SB && (t & (ITER - 1)) != CT) slow_step(t--);   // down to the top of an iteration
    if (t - (ITER - 1) >= SB) {
        // uniform row bases of the wave's four tiles
        const uint8_t* tb[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const u32 tr = blockIdx.x * 4u + (u32)jj;
            tb[jj] = (const uint8_t*)(a.ws + (size_t)(tr < n_tiles ? tr : n_tiles - 1) * a.ws_tile_stride);
        }
        // the ring's LDS address, opaque to hipcc: a direct-to-LDS load whose destination it can name (the array's base: slot 0)
        // makes it guard every later read of the array that it cannot tell apart with s_waitcnt vmcnt(0) -- one full drain of
        // the ring per pass, in front of the read that follows the refill of slot 0
        typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
        u32 ring_lds = (u32)(uintptr_t)(lds_u8_t*)ring;
        asm volatile("" : "+s"(ring_lds));
        auto fill = [&](int slot, int row) __attribute__((always_inline)) {
            // scalar tile base + ONE 32-bit per-lane offset for the four loads (a tile's rows stay below 4 GiB)
            const u32 vo = src_off + (u32)(row < 0 ? 0 : row) * 1024u;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(tb[jj] + vo), (lds_void_t*)(lds_u8_t*)(uintptr_t)(ring_lds + (u32)slot * reg_cb_ring_slot_bytes() + (u32)jj * (1024u + (u32)VIT_REG_CB_STAGGER)), 16, 0, 0);
        };
        // t = CT mod 32: row t / SPS sits in slot (CT / SPS) % D in every iteration (an iteration retires ROWS_IT rows, a
        // multiple of D)
        constexpr int S0 = (CT / SPS) % D;
#pragma unroll
        for (int k = 0; k < D; ++k) fill((S0 - k + D) % D, t / SPS - k);
        const uint8_t* ring_b = (const uint8_t*)ring;
        while (t - (ITER - 1) >= SB) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): retire the previous flush (and the ring)
            const int t0 = t;
            int it = 0;
            for (; it < KI && t - (ITER - 1) >= SB; ++it, t -= ITER) {
#pragma unroll
                for (int pk = 0; pk < ITER; ++pk) {
                    // step t - pk = CT - pk mod 32: its row is rk rows below the iteration's top row, step sidx of it
                    const int cs = CT - pk;                      // may go negative inside an iteration
                    const int rk = CT / SPS - (cs >= 0 ? cs / SPS : -((-cs + SPS - 1) / SPS));
                    const int sidx = ((cs % SPS) + SPS) % SPS;
                    const int slot = ((S0 - rk) % D + D) % D;
                    // the row's four loads are the oldest of the 4 D in flight
                    constexpr int NV = 4 * (D - 1);
                    __builtin_amdgcn_s_waitcnt(0x0F70 | (NV & 15) | ((NV >> 4) << 14));   // vmcnt(4 D - 4)
                    const u32 wA = ring_b[slot * (int)reg_cb_ring_slot_bytes() + sidx * 4 * DW + byte_of(xA, pair_base)];
                    const u32 wB = ring_b[slot * (int)reg_cb_ring_slot_bytes() + sidx * 4 * DW + 1 + byte_of(xB, pair_base)];
                    if constexpr (ITER % SB == 0) {
                        // the replaced bit's position is a compile-time constant: (SB - (t - pk + 1) % SB) % SB with
                        // t = CT mod SB (SB divides 32)
                        const u32 p = (u32)((SB - (((CT - pk + 1) % SB) + SB) % SB) % SB);
                        step_back(RA, xA, wA, p);
                        step_back(RB, xB, wB, p);
                    } else {
                        step_back(RA, xA, wA, pos);
                        step_back(RB, xB, wB, pos);
                        pos = pos == SB - 1 ? 0u : pos + 1u;
                    }
                    if (sidx == 0) {
                        // the row is done and its reads have returned (their bits are in RA / RB): refill the slot
                        asm volatile("" : "+v"(RA), "+v"(RB), "+v"(xA), "+v"(xB) : : "memory");
                        fill(slot, (t - pk) / SPS - D);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // bytes j/8 of the 32 steps, first decoded bit of each byte on top: RA's top byte is the LOWEST byte index
                obuf[it * 64 + lane] = __builtin_amdgcn_perm(RA, RA, 0x00010203u);
                obuf[(KI + it) * 64 + lane] = __builtin_amdgcn_perm(RB, RB, 0x00010203u);
            }
            // iteration i covered steps t0 - 32 i - 31 ... t0 - 32 i: bytes (t0 - 32 i - 31 - SB) / 8 ... + 3
            for (int i = 0; i < it; ++i) {
                const u32 jb = (u32)(t0 - ITER * i - (ITER - 1) - SB) >> 3;
                *(u32_unaligned*)(outA + jb) = obuf[i * 64 + lane];
                *(u32_unaligned*)(outB + jb) = obuf[(KI + i) * 64 + lane];
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // the ring's last (surplus) loads land before LDS is released
        if constexpr (ITER % SB == 0) pos = (SB - (u32)(t + 1) % (u32)SB) % SB;
    }
    while (t >= SB) slow_step(t--);
}

// ---- lane-local chainback for the 16-register codes (K = 7): one lane per frame PAIR, 64 pairs (four tiles) per wave ----
// Nothing on the dependent bit-chase leaves the lane: the four q-rows of a step (16 B each, 4 steps per row) are all loaded
// by the pair's lane, and every loaded byte is used (byte 0/2 of a dword = frame A, byte 1/3 = frame B; a lane per FRAME
// loaded each row twice, and the kernel ran at the L1 delivery rate of ~12 B/clk/CU, not at HBM's).  Per frame two v_perm
// pick its bytes into a 64-bit word whose bit index IS the slot index, and the survivor's bit is one shift away; the two
// chases of a lane are independent, which hides their dependent latency.  Rows are fetched NBUF groups (4 steps each) ahead
// into a register ring (28 x 1 KiB loads in flight per wave); the main loop is branch-free so that hipcc can retire the ring
// with counted s_waitcnt vmcnt(N) instead of draining it (a branch around a load costs a vmcnt(0)).
template <class SP>
VIT_DEV void reg_chainback16_body(const RegChainbackArgs& a) {
    static_assert(SP::NREG == 16 && SP::DW == 1 && SP::SPS == 4 && SP::SB == 6, "K = 7 layout");
    constexpr int SB = SP::SB;
    constexpr int IGN = SB < 8 ? SB : 8;                       // ViterbiTracebackBuffer::get_layout (core.h:129-149)
    constexpr int SHIFT_STATE = 8 - IGN;
    static_assert(SB - IGN == 0, "shift_tail = 0: the output byte is the 8-bit register itself");
    constexpr int NBUF = 8;                                    // even: keeps the byte-output positions compile-time

    const int lane = threadIdx.x & 63;
    const u32 n_tiles = (a.frames + 31u) >> 5;
    const u32 p_raw = blockIdx.x * 64 + lane;                  // frame pair (tile, g): frames 32*tile + g and + 16
    const u32 tile = (p_raw >> 4) < n_tiles ? (p_raw >> 4) : n_tiles - 1, g = p_raw & 15u;
    // surplus lanes / frames redo the last frame: the update kernel filled their half with that frame's decisions, so the
    // stores are identical
    const u32 fA = tile * 32u + g < a.frames ? tile * 32u + g : a.frames - 1;
    const u32 fB = tile * 32u + g + 16u < a.frames ? tile * 32u + g + 16u : a.frames - 1;
    const uint4* rows = a.ws + (size_t)tile * a.ws_tile_stride + g;   // row (group, q) = rows[group*64 + q*16]
    const size_t out_stride = ((size_t)a.L + 7) / 8;
    uint8_t* outA = a.out + (size_t)fA * out_stride;
    uint8_t* outB = a.out + (size_t)fB * out_stride;
    // {row q.byte(h), row q.byte(2+h), row q+1.byte(h), row q+1.byte(2+h)}: frame h's 16+16 bits of two q-rows
    constexpr u32 SELA = 0x06040200u, SELB = 0x07050301u;

    // The reference's 8-bit register (state in its top SB bits, core.h:110-113) is the TOP BYTE of a 32-bit shift register here,
    // and beside it the chase keeps the survivor's SLOT: the update's in-place layout moves no state between slots, so going
    // back one step replaces exactly one bit of the slot index -- position (SB - (t+1) % SB) % SB -- by the decision bit (a
    // shift and a bit-field insert where forming the slot from the state took a rotation: 8 instead of 15 instructions per frame
    // and step; they are what the chainback takes from the update kernel it runs beside)
    const u32 esA = a.end_state ? (a.end_state[fA] & SP::SMASK) : 0u, esB = a.end_state ? (a.end_state[fB] & SP::SMASK) : 0u;
    u32 regA = esA << (24 + SHIFT_STATE), regB = esB << (24 + SHIFT_STATE);
    const u32 ph_top = ((u32)a.L + (u32)SB) % (u32)SB;         // (t + 1) % SB of the first step, t = L - 1 + SB
    u32 xA = ((esA >> ph_top) | (esA << (SB - ph_top))) & SP::SMASK, xB = ((esB >> ph_top) | (esB << (SB - ph_top))) & SP::SMASK;

    // one traceback step of one frame on the four dwords (one per q-row) that hold step t; steps come in descending order
    // `p`: the slot bit this step's decision replaces, (SB - (t + 1) % SB) % SB -- wave-uniform, kept on the scalar unit
    auto chase1 = [&](u32& reg, u32& x, u32 lo, u32 hi, u32 p) __attribute__((always_inline)) {
        const u32 w = x > 31u ? hi : lo;
        const u32 bit = __builtin_amdgcn_ubfe(w, x, 1);        // v_bfe_u32 takes the offset's low five bits: x & 31
        reg = (reg >> 1) | (bit << 31);
        x = (x & ~(1u << p)) | (bit << p);
    };
    auto chase = [&](u32 d0, u32 d1, u32 d2, u32 d3, u32 p) __attribute__((always_inline)) {
        chase1(regA, xA, __builtin_amdgcn_perm(d1, d0, SELA), __builtin_amdgcn_perm(d3, d2, SELA), p);
        chase1(regB, xB, __builtin_amdgcn_perm(d1, d0, SELB), __builtin_amdgcn_perm(d3, d2, SELB), p);
    };
    auto pos_of = [](int t) __attribute__((always_inline)) -> u32 { return (u32)((SB - (t + 1) % SB) % SB); };
    auto emit = [&](u32 jb) __attribute__((always_inline)) {    // byte jb is complete
        outA[jb] = (uint8_t)(regA >> 24);
        outB[jb] = (uint8_t)(regB >> 24);
    };
    // ragged ends: load the step's dwords directly (dependent latency, only a handful of steps)
    auto slow_step = [&](int t) __attribute__((always_inline)) {
        const u32* r32 = (const u32*)(rows + (size_t)(t >> 2) * 64) + (t & 3);
        chase(r32[0], r32[16 * 4], r32[32 * 4], r32[48 * 4], pos_of(t));
        const int j = t - SB;
        if ((j & 7) == 0) emit((u32)j >> 3);
    };

    int t = (int)a.L - 1 + SB;                                  // rows are consumed from t = L-1+SB down to SB
    // top: down to a group boundary, and to an EVEN group on top of the ring
    while (t >= SB && (((t & 3) != 3) || (((t >> 2) & 1) != 0))) slow_step(t--);
    const int g_top = t >> 2;                                   // even (or the loop above ran out of steps)
    const int g_min = (SB + 3) / 4;                             // lowest group whose 4 steps are all >= SB
    if (t >= SB && g_top - (NBUF - 1) >= g_min) {
        uint4 buf[NBUF][4];
#pragma unroll
        for (int b = 0; b < NBUF; ++b)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) buf[b][qq] = rows[(size_t)(g_top - b) * 64 + qq * 16];
        int gb = g_top;
        u32 pos = (u32)__builtin_amdgcn_readfirstlane((int)pos_of(4 * gb + 3));   // of the first step of the ring; +1 (mod SB) per step
        // One inner iteration = NBUF groups = 32 steps = 4 output bytes per frame.  Inside the inner loop ONLY loads are
        // outstanding: with a store pending next to them hipcc must assume out-of-order completion and drains the ring with
        // s_waitcnt vmcnt(0) at every loop top (one full memory round trip per 32 steps -- that, not bandwidth, was the
        // kernel's 1.16 ms).  The 4 bytes are assembled in a register, parked in LDS (lgkmcnt) and flushed as one
        // (possibly unaligned) dword store per frame every KI iterations.
        constexpr int KI = 32;
        __shared__ uint2 obuf[KI * 64];
        typedef u32 u32_unaligned __attribute__((aligned(1)));
        while (gb - (NBUF - 1) >= g_min) {
            // retire the previous flush (and the ring): the inner loop then starts with nothing but its own loads in flight
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            const int gb0 = gb;
            int it = 0;
            for (; it < KI && gb - (NBUF - 1) >= g_min; ++it, gb -= NBUF) {
                u32 accA = 0, accB = 0;
#pragma unroll
                for (int b = 0; b < NBUF; ++b) {
                    const int grp = gb - b;                     // parity == parity of b (gb is even)
#pragma unroll
                    for (int sidx = 3; sidx >= 0; --sidx) {
                        const u32 d0 = sidx == 0 ? buf[b][0].x : sidx == 1 ? buf[b][0].y : sidx == 2 ? buf[b][0].z : buf[b][0].w;
                        const u32 d1 = sidx == 0 ? buf[b][1].x : sidx == 1 ? buf[b][1].y : sidx == 2 ? buf[b][1].z : buf[b][1].w;
                        const u32 d2 = sidx == 0 ? buf[b][2].x : sidx == 1 ? buf[b][2].y : sidx == 2 ? buf[b][2].z : buf[b][2].w;
                        const u32 d3 = sidx == 0 ? buf[b][3].x : sidx == 1 ? buf[b][3].y : sidx == 2 ? buf[b][3].z : buf[b][3].w;
                        chase(d0, d1, d2, d3, pos);
                        pos = pos + 1 == (u32)SB ? 0u : pos + 1;
                        // j = 4*grp + sidx - 6 is a multiple of 8  <=>  grp odd and sidx == 2: byte (4*grp - 4) / 8 is
                        // complete; bytes come out in descending order, so the first one ends up in the top byte
                        if ((b & 1) == 1 && sidx == 2) {
                            accA = (accA << 8) | (regA >> 24);
                            accB = (accB << 8) | (regB >> 24);
                        }
                    }
                    // pin the chase of this group in front of its refill: without a side effect per group the optimiser sinks
                    // the whole dependent chain below all eight refills and parks 128 permuted words in AGPRs / scratch
                    asm volatile("" : "+v"(regA), "+v"(regB), "+v"(xA), "+v"(xB) : : "memory");
                    const int nxt = grp - NBUF;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) { typedef u32 u32x4_t __attribute__((ext_vector_type(4))); const u32x4_t v4 = __builtin_nontemporal_load((const u32x4_t*)&rows[(size_t)(nxt < 0 ? 0 : nxt) * 64 + qq * 16]); buf[b][qq] = make_uint4(v4.x, v4.y, v4.z, v4.w); }
                    // keep the refill where it is written: hipcc otherwise sinks/merges the loads and the ring loses its depth
                    __builtin_amdgcn_sched_barrier(0);
                }
                obuf[it * 64 + lane] = make_uint2(accA, accB);
            }
            // iteration i started at group gb0 - NBUF*i (even) and completed bytes (gb0 - NBUF*i)/2 - 4 ... + 3
            for (int i = 0; i < it; ++i) {
                const uint2 v = obuf[i * 64 + lane];
                const u32 jb = (u32)((gb0 - NBUF * i) / 2 - 4);
                *(u32_unaligned*)(outA + jb) = v.x;
                *(u32_unaligned*)(outB + jb) = v.y;
            }
        }
        t = 4 * gb + 3;
    }
    // bottom: what the ring did not cover
    while (t >= SB) slow_step(t--);
}

template <class SP>
VIT_DEV void reg_chainback0_body(const RegChainbackArgs& a) {
    // LANE_BITS = 0 (K <= 6): every state of a frame sits in ONE lane's registers, so one lane per frame chases its survivor
    // through rows it loads itself.  A 16-byte row holds SPS = 4 / DW trellis steps of DW decision dwords (DW = 1: up to 16
    // registers, K <= 5; DW = 2: 32 registers, K = 6).
    static_assert(SP::LANE_BITS == 0 && SP::NREG <= 32 && (SP::DW == 1 || SP::DW == 2), "small-K layout");
    constexpr int SB = SP::SB, DW = SP::DW, SPS = SP::SPS;
    constexpr int IGN = SB < 8 ? SB : 8;                       // ViterbiTracebackBuffer::get_layout (core.h:129-149)
    constexpr int SHIFT_STATE = 8 - IGN, SHIFT_TAIL = SB - IGN, TOTAL_BITS = SB + SHIFT_STATE;
    constexpr int NBUF = 32 / SPS;                             // a window of the ring = 32 trellis steps = 4 output bytes

    const int lane = threadIdx.x & 63;
    const u32 f_raw = blockIdx.x * 64 + lane;
    const bool valid = f_raw < a.frames;
    const u32 f = valid ? f_raw : a.frames - 1;                // surplus lanes redo the last frame (identical stores)
    const u32 tile = f / SP::TILE, g = f % SP::PAIRS, half = (f % SP::TILE) / SP::PAIRS;
    const uint4* rows = a.ws + (size_t)tile * a.ws_tile_stride + g;   // row of step group grp = rows[grp*64]
    const size_t out_stride = ((size_t)a.L + 7) / 8;
    uint8_t* out = a.out + (size_t)f * out_stride;
    const u32 hshift = 8u * half;

    u32 reg = (a.end_state ? (a.end_state[f] & SP::SMASK) : 0u) << SHIFT_STATE;
    auto chase = [&](u32 w0, u32 w1, u32 ph1) __attribute__((always_inline)) {
        const u32 state = reg >> SHIFT_STATE;
        const u32 x = ((state >> ph1) | (state << (SB - ph1))) & SP::SMASK;   // slot (= register) of `state` after step t
        const u32 w = (DW == 2 && (x & 16u)) ? w1 : w0;                       // decision dword = register / 16
        const u32 bit = (w >> ((x & 7u) + hshift + 16u * ((x >> 3) & 1u))) & 1u;  // SP::dec_bit(x & 15, half)
        reg = (reg >> 1) | (bit << (TOTAL_BITS - 1));
    };
    auto slow_step = [&](int t) __attribute__((always_inline)) {
        const u32* r32 = (const u32*)(rows + (size_t)(t / SPS) * 64) + (t % SPS) * DW;
        chase(r32[0], DW == 2 ? r32[DW - 1] : 0u, (u32)((t + 1) % SB));
        const int j = t - SB;
        if ((j & 7) == 0) out[(u32)j >> 3] = (uint8_t)((reg >> SHIFT_TAIL) & 0xFFu);
    };

    int t = (int)a.L - 1 + SB;
    while (t >= SB && (t & 7) != 3) slow_step(t--);             // windows start at t = 3 (mod 8): the last step of a row
    const int g_top = t / SPS;
    const int g_min = (SB + SPS - 1) / SPS;                      // lowest row all of whose steps are >= SB
    if (t >= SB && g_top - (NBUF - 1) >= g_min) {
        uint4 buf[NBUF];
#pragma unroll
        for (int b = 0; b < NBUF; ++b) buf[b] = rows[(size_t)(g_top - b) * 64];
        int gb = g_top;
        u32 ph = (u32)((SPS * gb + SPS - 1 + 1) % SB);
        // as in reg_chainback16_body: no store inside the ring loop (it would force vmcnt(0) at every loop top); the 4 bytes
        // of a window are assembled in a register, parked in LDS and flushed as one dword store every KI windows
        constexpr int KI = 32;
        __shared__ u32 obuf[KI * 64];
        typedef u32 u32_unaligned __attribute__((aligned(1)));
        // window step k (0..31, descending t = t_top - k, t_top = 3 mod 8) completes byte (t - SB) / 8 iff (3 - k - SB) % 8 == 0
        constexpr int K_LAST = 24 + (((3 - SB) % 8) + 8) % 8;   // the last of the window's four such steps
        while (gb - (NBUF - 1) >= g_min) {
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): retire the previous flush (and the ring)
            const int gb0 = gb;
            int it = 0;
            for (; it < KI && gb - (NBUF - 1) >= g_min; ++it, gb -= NBUF) {
                u32 acc = 0;
#pragma unroll
                for (int b = 0; b < NBUF; ++b) {
#pragma unroll
                    for (int sidx = SPS - 1; sidx >= 0; --sidx) {
                        const int k = b * SPS + (SPS - 1 - sidx);
                        const int c0 = sidx * DW, c1 = sidx * DW + DW - 1;
                        const u32 w0 = c0 == 0 ? buf[b].x : c0 == 1 ? buf[b].y : c0 == 2 ? buf[b].z : buf[b].w;
                        const u32 w1 = c1 == 0 ? buf[b].x : c1 == 1 ? buf[b].y : c1 == 2 ? buf[b].z : buf[b].w;
                        chase(w0, w1, ph);
                        ph = ph == 0 ? SB - 1 : ph - 1;
                        // bytes come out in descending order
                        if ((((3 - k - SB) % 8) + 8) % 8 == 0) acc = (acc << 8) | ((reg >> SHIFT_TAIL) & 0xFFu);
                    }
                    asm volatile("" : "+v"(reg) : : "memory");   // pin the chase of this row in front of its refill
                    const int nxt = gb - b - NBUF;
                    buf[b] = rows[(size_t)(nxt < 0 ? 0 : nxt) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                }
                obuf[it * 64 + lane] = acc;
            }
            for (int i = 0; i < it; ++i) {
                const int t_top = SPS * (gb0 - NBUF * i) + SPS - 1;
                const u32 jb = (u32)(t_top - K_LAST - SB) >> 3;                 // lowest of the window's 4 bytes
                *(u32_unaligned*)(out + jb) = obuf[i * 64 + lane];
            }
        }
        t = SPS * gb + SPS - 1;
    }
    while (t >= SB) slow_step(t--);
}

// ---- export to the reference layout [F][n_steps][W] ----------------------------------------------------------------
struct RegExportArgs {
    const u32* ws32;
    size_t ws_tile_stride;  // uint4 units
    uint64_t* out;
    u32 frames, n_steps;
};

template <class SP>
VIT_DEV void reg_export_body(const RegExportArgs& a) {
    constexpr int SB = SP::SB, NREG = SP::NREG, REG_BITS = SP::REG_BITS, DW = SP::DW, SPS = SP::SPS;
    constexpr int W = SB >= 6 ? 1 << (SB - 6) : 1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (frame, step, word)
    const size_t total = (size_t)a.frames * a.n_steps * W;
    if (idx >= total) return;
    const u32 w = (u32)(idx % W);
    const u32 t = (u32)((idx / W) % a.n_steps);
    const u32 f = (u32)(idx / ((size_t)W * a.n_steps));
    const u32 tile = f / SP::TILE, g = f % SP::PAIRS, half = (f % SP::TILE) / SP::PAIRS;
    const int ph1 = (int)((t + 1) % SB);
    uint64_t word = 0;
    for (u32 b = 0; b < 64; ++b) {
        const u32 s = w * 64 + b;
        if (s > SP::SMASK) break;
        const u32 x = ((s >> ph1) | (s << (SB - ph1))) & SP::SMASK;
        const u32 qs = x >> REG_BITS, rs = x & (NREG - 1);
        const size_t row = (size_t)tile * a.ws_tile_stride + (size_t)(t / SPS) * 64 + qs * SP::PAIRS + g;
        const u32 dwi = (DW == 4) ? (rs >> 4) : (t % SPS) * DW + (rs >> 4);
        const u32 v = a.ws32[row * 4 + dwi];
        word |= (uint64_t)((v >> SP::dec_bit(rs, half)) & 1u) << b;
    }
    a.out[idx] = word;
}

// ---- kernels: thin __global__ wrappers (the bodies above are shared with the run-time compiled instantiations) --------
// two waves per SIMD (<= 256 VGPRs) for every code: the tiles of a 65536-frame batch outnumber the SIMDs two to one, and a lone
// wave issues at 5.4 - 5.9 clocks per instruction where two reach 4.5.
template <class SP>
constexpr int reg_update_min_waves() { return 2; }
// Capped at 240 registers (the attribute counts in halves of the unified file: 120) where hipcc would take more: two such waves
// leave 32 registers per SIMD -- the allocation of the LDS-ring chainback kernel (24 at K = 9, 32 at K = 7), which then runs
// BESIDE the next batch's update (vit_hip_pipeline_create, rule 1).
//   K = 9, R = 2 (IS-95A): hipcc's schedule for 256 holds no more live values than fit 240 (no scratch either way), the capped
//          kernel runs 2.7 % FASTER (10.99 vs 11.29 ms, 65536 x 8192);
//   K = 9, R = 4 (CDMA 2000): 217 used with the sub-chunk branch-metric fetch (RegChunk; 368 allocated -- ONE wave per SIMD --
//          with the whole-step double buffer);
//   K = 7, R = 3 (LTE): 248 -> 240, no scratch.
// (the attribute takes a literal, not a template-dependent value: the translation unit of the code sets it -- reg_inst.hip
// with -DVIT_REG_ID=1, 3, 4, reg_jit.hpp for the run-time compiled codes of those geometries)
#if !defined(VIT_REG_UPDATE_VGPR_CAP) && defined(VIT_REG_ID)
#if VIT_REG_ID == 1 || VIT_REG_ID == 3 || VIT_REG_ID == 4
#define VIT_REG_UPDATE_VGPR_CAP __attribute__((amdgpu_num_vgpr(120)))
#endif
#endif
#ifndef VIT_REG_UPDATE_VGPR_CAP
#define VIT_REG_UPDATE_VGPR_CAP
#endif
template <class SP, int SHIFT>
__global__ void __launch_bounds__(64, reg_update_min_waves<SP>()) VIT_REG_UPDATE_VGPR_CAP reg_update_kernel(RegUpdateArgs a) {
#ifdef VIT_HIP_CLOCK_STAMPS
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    reg_update_body<SP, SHIFT, false>(a);
#ifdef VIT_HIP_CLOCK_STAMPS
    if (a.stamps && (threadIdx.x & 63) == 0) {
        uint64_t* st = a.stamps + (size_t)blockIdx.x * 6;
        st[0] = c0; st[1] = r0; st[2] = __builtin_amdgcn_s_memtime(); st[3] = __builtin_amdgcn_s_memrealtime();
        st[4] = (uint64_t)__builtin_amdgcn_s_getreg((3 << 11) | 20);      // hwreg(HW_REG_XCC_ID, 0, 4)
        st[5] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // hwreg(HW_REG_HW_ID): wave, SIMD, CU, SH, SE
    }
#endif
}
// the resumed update (batched streaming) carries the mid-block entry on top of the throughput kernel's registers: one wave per SIMD
// is promised, so that no instantiation spills (at two, K = 7 R = 3, 4 took 56 - 168 bytes of scratch per lane and K = 9 R = 2 eight)
template <class SP, int SHIFT>
__global__ void __launch_bounds__(64, 1) reg_resume_kernel(RegUpdateArgs a) { reg_update_body<SP, SHIFT, true>(a); }

// chainback: one kernel name per code, the body is picked by the code's geometry
template <class SP>
VIT_DEV void reg_chainback_body(const RegChainbackArgs& a) {
    if constexpr (SP::LANE_BITS == 0) reg_chainback0_body<SP>(a);
    else if constexpr (SP::NREG == 16) reg_chainback16_body<SP>(a);
    else if constexpr (SP::NREG == 64) reg_chainback_ring_body<SP>(a);
    else reg_chainback_coop_body<SP>(a);
}
// dynamic LDS of a code's chainback kernel (`alt`: of its alternative kernel): the LDS-ring body is K = 9's kernel and K = 7's alternative
constexpr unsigned reg_chainback_dyn_lds_bytes(int K, int R, bool alt) {
    return (K == 9 && !alt) ? reg_cb_ring_lds_bytes(64, R) : (K == 7 && alt) ? reg_cb_ring_lds_bytes(16, R) : 0u;
}
template <class SP>
constexpr unsigned reg_chainback_frames_per_block() { return (SP::NREG == 16 || SP::NREG == 64) && SP::LANE_BITS == 2 ? 128u : SP::LANE_BITS == 0 ? 64u : 32u; }
// (the K = 9 body's LDS ring -- 40 KiB per one-wave workgroup -- allows four of them per CU)
template <class SP>
constexpr int reg_chainback_min_waves() { return SP::NREG == 64 ? 1 : 2; }
template <class SP>
__global__ void __launch_bounds__(64, reg_chainback_min_waves<SP>()) reg_chainback_kernel(RegChainbackArgs a) {
    // beside TWO update kernels (the small-batch pipeline schedule) the bit chase would otherwise get the issue slots both leave
    // over and become the pipeline's bottleneck: there it runs at the higher wave priority
    if (a.wave_priority) __builtin_amdgcn_s_setprio(3);
    reg_chainback_body<SP>(a);
}
// The body reg_chainback_kernel does NOT run, as a kernel of its own (128 / 32 frames per block as the bodies say).
//  K = 7: the LDS-ring body (32 registers, 24 KiB of dynamic LDS).  Alone it is slower than the register ring (an LDS round trip on
//         the dependent chain: 0.86 / 0.71 / 0.68 ms against 0.80 / 0.52 / 0.49 at 65536 / 32768 / 8192 frames), but it is the
//         better neighbour of update waves: the pipeline launches it for chainbacks that run beside the next update of the
//         one-update schedule (65536 x 8192: 157 -> 160 Gbit/s; three update waves + this kernel fit a SIMD, 3 x 152 + 32) -- NOT
//         for the two-update schedule, whose chainback runs at the higher wave priority and has to be fast (hard8 32768 x 8192: 145
//         against 163 Gbit/s);
//  K = 9: the cooperative body (one wave per 32 frames; every q-lane repeats the chase and a ds_bpermute picks the owner's bit:
//         34 vector instructions per step for 32 frames, 136 registers allocated).  Level with the ring body alone on the device,
//         no use beside two update waves; tests only (vit_hip_chainback_batch_ex).
template <class SP>
VIT_DEV void reg_chainback_alt_body(const RegChainbackArgs& a) {
    if constexpr (SP::NREG == 64) reg_chainback_coop_body<SP>(a);                              // 32 frames per wave
    else if constexpr (SP::NREG == 16 && SP::LANE_BITS == 2) reg_chainback_ring_body<SP>(a);   // 128 frames per wave
}
template <class SP>
constexpr int reg_chainback_alt_min_waves() { return SP::NREG == 16 ? 1 : 2; }
template <class SP>
__global__ void __launch_bounds__(64, reg_chainback_alt_min_waves<SP>()) reg_chainback_alt_kernel(RegChainbackArgs a) { reg_chainback_alt_body<SP>(a); }
template <class SP>
__global__ void reg_export_kernel(RegExportArgs a) { reg_export_body<SP>(a); }

#ifndef VIT_REG_JIT_TU
}  // namespace vit
#include "kernel_desc.hpp"
namespace vit {
// ---- host side ----------------------------------------------------------------------------------------------------
using Spec_K7R2 = RegSpec<7, 2, 109, 79, 0, 0>;             // Voyager          (common_codes.h:23)
using Spec_K7R3 = RegSpec<7, 3, 91, 117, 121, 0>;           // LTE              (:24)
using Spec_K7R4 = RegSpec<7, 4, 109, 79, 83, 109>;          // DAB Radio        (:25)
using Spec_K9R2 = RegSpec<9, 2, 491, 369, 0, 0>;            // CDMA IS-95A      (:26)
using Spec_K9R4 = RegSpec<9, 4, 501, 441, 331, 315>;        // CDMA 2000        (:27)
using Spec_K3R2 = RegSpec<3, 2, 7, 5, 0, 0, 0>;             // Basic K=3        (:21)  all 4 states in one lane
using Spec_K5R2 = RegSpec<5, 2, 23, 25, 0, 0, 0>;           // Basic K=5        (:22)  all 16 states in one lane

// a run-time compiled instantiation (reg_jit.hpp): the same four kernels for polynomials that are not in the table above
struct RegJitModule {
    hipModule_t module = nullptr;
    hipFunction_t update[2] = {nullptr, nullptr};   // [0] 16-bit, [1] 8-bit metrics/symbols
    hipFunction_t resume[2] = {nullptr, nullptr};
    hipFunction_t chainback = nullptr, export_ = nullptr;
    hipFunction_t chainback_alt = nullptr;         // K = 7, 9: the alternative body (reg_chainback_alt_body)
    unsigned chainback_frames_per_block = 32;
    kd::Table kernels;                              // kernel descriptors of the module's code object (kernel_desc.hpp)
};

struct RegCode {
    int id = -1;   // 0..6 in the order above; -1 with jit != nullptr for a run-time compiled code
    int K = 0, R = 0;
    int tile = 32; // frames per wavefront
    uint32_t G[6] = {0, 0, 0, 0, 0, 0};
    const RegJitModule* jit = nullptr;
    bool generic = false;   // jit is the GENERIC code object of (K, R): the kernels read G from their arguments (RegSpec::GENERIC)
};

// (K, R) with a generic register-plan kernel: the LDS-ring geometries (K = 7: whole-step fetch, K = 8, 9: per sub-chunk) and, below K = 7,
// the one-lane geometry with the pairs parked in LDS; whole patterns (R <= 4)
inline bool reg_generic_supported(int K, int R) {
    // (K = 6 at an odd rate unrolls an 80- to 240-step block: minutes of hipcc per object; K = 2: one butterfly, nothing to look up)
    return K >= 3 && K <= 9 && R >= 1 && R <= 4 && !(K == 6 && (R & 1));
}

inline bool reg_code_supported(int K, int R) {
    return (K == 7 && R >= 2 && R <= 4) || (K == 9 && (R == 2 || R == 4)) || ((K == 3 || K == 5) && R == 2);
}

inline bool reg_code_init(RegCode* rc, int K, int R, const uint32_t* G, const DevConfig&) {
    struct Entry { int K, R; uint32_t G[4]; };
    static const Entry table[7] = {{7, 2, {109, 79, 0, 0}}, {7, 3, {91, 117, 121, 0}}, {7, 4, {109, 79, 83, 109}},
                                   {9, 2, {491, 369, 0, 0}}, {9, 4, {501, 441, 331, 315}},
                                   {3, 2, {7, 5, 0, 0}}, {5, 2, {23, 25, 0, 0}}};
    for (int id = 0; id < 7; ++id) {
        if (table[id].K != K || table[id].R != R) continue;
        bool same = true;
        for (int i = 0; i < R; ++i) same = same && (table[id].G[i] == G[i]);
        if (same) {
            rc->id = id; rc->K = K; rc->R = R; rc->tile = K < 7 ? 128 : 32;
            for (int i = 0; i < 4; ++i) rc->G[i] = table[id].G[i];
            rc->G[4] = rc->G[5] = 0;
            return true;
        }
    }
    return false;
}

// host mirror of RegSpec's geometry: lane bits, registers per lane, decision dwords per step, steps per 16-byte row
inline int reg_lane_bits(int K) { return K >= 7 ? 2 : 0; }
inline size_t reg_steps_per_row(int K) {
    const size_t nreg = (size_t)1 << (K - 1 - reg_lane_bits(K));
    const size_t dw = nreg >= 16 ? nreg / 16 : 1;
    return 4 / dw;
}
inline size_t reg_groups(const RegCode& rc, size_t L) {
    const size_t S = L + (size_t)rc.K - 1;
    const size_t sps = reg_steps_per_row(rc.K);
    return (S + sps - 1) / sps;
}
inline size_t reg_tiles(const RegCode& rc, size_t frames) { return (frames + (size_t)rc.tile - 1) / (size_t)rc.tile; }
inline size_t reg_workspace_bytes(const RegCode& rc, size_t frames, size_t L) {
    return reg_tiles(rc, frames) * reg_groups(rc, L) * 1024;
}

// The kernels are instantiated one code per translation unit (reg_inst.hip, compiled with -DVIT_REG_ID=0..4) so that the
// heavy unrolled bodies build in parallel; these are the per-code launchers those units define.
template <int ID> struct RegSpecOf;
template <> struct RegSpecOf<0> { using type = Spec_K7R2; };
template <> struct RegSpecOf<1> { using type = Spec_K7R3; };
template <> struct RegSpecOf<2> { using type = Spec_K7R4; };
template <> struct RegSpecOf<3> { using type = Spec_K9R2; };
template <> struct RegSpecOf<4> { using type = Spec_K9R4; };
template <> struct RegSpecOf<5> { using type = Spec_K3R2; };
template <> struct RegSpecOf<6> { using type = Spec_K5R2; };

template <int ID> int reg_launch_update(int shift, const RegUpdateArgs& a, unsigned tiles, hipStream_t st);
template <int ID> int reg_launch_chainback(const RegChainbackArgs& a, unsigned tiles, hipStream_t st, bool coop);
template <int ID> int reg_launch_export(const RegExportArgs& a, unsigned blocks, hipStream_t st);

#ifdef VIT_REG_ID
template <> int reg_launch_update<VIT_REG_ID>(int shift, const RegUpdateArgs& a, unsigned tiles, hipStream_t st) {
    using SP = RegSpecOf<VIT_REG_ID>::type;
    if (a.metrics_in) {
        if (shift) hipLaunchKernelGGL((reg_resume_kernel<SP, 8>), dim3(tiles), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((reg_resume_kernel<SP, 0>), dim3(tiles), dim3(64), 0, st, a);
    } else {
        if (shift) hipLaunchKernelGGL((reg_update_kernel<SP, 8>), dim3(tiles), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((reg_update_kernel<SP, 0>), dim3(tiles), dim3(64), 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <> int reg_launch_chainback<VIT_REG_ID>(const RegChainbackArgs& a, unsigned tiles, hipStream_t st, bool coop) {
    using SP = RegSpecOf<VIT_REG_ID>::type;
    constexpr unsigned FPB = reg_chainback_frames_per_block<SP>();
    if (coop && SP::NREG == 64) hipLaunchKernelGGL(reg_chainback_alt_kernel<SP>, dim3(tiles), dim3(64), 0, st, a);
    else if (coop && SP::NREG == 16 && SP::LANE_BITS == 2) hipLaunchKernelGGL(reg_chainback_alt_kernel<SP>, dim3((a.frames + 127) / 128), dim3(64), reg_chainback_dyn_lds_bytes(SP::K, SP::R, true), st, a);
    else hipLaunchKernelGGL(reg_chainback_kernel<SP>, dim3((a.frames + FPB - 1) / FPB), dim3(64), reg_chainback_dyn_lds_bytes(SP::K, SP::R, false), st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <> int reg_launch_export<VIT_REG_ID>(const RegExportArgs& a, unsigned blocks, hipStream_t st) {
    using SP = RegSpecOf<VIT_REG_ID>::type;
    hipLaunchKernelGGL(reg_export_kernel<SP>, dim3(blocks), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#else
template <> int reg_launch_update<0>(int, const RegUpdateArgs&, unsigned, hipStream_t);
template <> int reg_launch_update<1>(int, const RegUpdateArgs&, unsigned, hipStream_t);
template <> int reg_launch_update<2>(int, const RegUpdateArgs&, unsigned, hipStream_t);
template <> int reg_launch_update<3>(int, const RegUpdateArgs&, unsigned, hipStream_t);
template <> int reg_launch_update<4>(int, const RegUpdateArgs&, unsigned, hipStream_t);
template <> int reg_launch_update<5>(int, const RegUpdateArgs&, unsigned, hipStream_t);
template <> int reg_launch_update<6>(int, const RegUpdateArgs&, unsigned, hipStream_t);
template <> int reg_launch_chainback<0>(const RegChainbackArgs&, unsigned, hipStream_t, bool);
template <> int reg_launch_chainback<1>(const RegChainbackArgs&, unsigned, hipStream_t, bool);
template <> int reg_launch_chainback<2>(const RegChainbackArgs&, unsigned, hipStream_t, bool);
template <> int reg_launch_chainback<3>(const RegChainbackArgs&, unsigned, hipStream_t, bool);
template <> int reg_launch_chainback<4>(const RegChainbackArgs&, unsigned, hipStream_t, bool);
template <> int reg_launch_chainback<5>(const RegChainbackArgs&, unsigned, hipStream_t, bool);
template <> int reg_launch_chainback<6>(const RegChainbackArgs&, unsigned, hipStream_t, bool);
template <> int reg_launch_export<0>(const RegExportArgs&, unsigned, hipStream_t);
template <> int reg_launch_export<1>(const RegExportArgs&, unsigned, hipStream_t);
template <> int reg_launch_export<2>(const RegExportArgs&, unsigned, hipStream_t);
template <> int reg_launch_export<3>(const RegExportArgs&, unsigned, hipStream_t);
template <> int reg_launch_export<4>(const RegExportArgs&, unsigned, hipStream_t);
template <> int reg_launch_export<5>(const RegExportArgs&, unsigned, hipStream_t);
template <> int reg_launch_export<6>(const RegExportArgs&, unsigned, hipStream_t);

// ---- which kernels can share a SIMD: read from the kernel DESCRIPTORS (kernel_desc.hpp), the numbers the wave launcher uses ----
// (hipFuncGetAttributes().numRegs is the count the code USES; hipcc pads the allocation of kernels whose static LDS limits their
// occupancy -- round 3's K = 9 chainback used 22 registers and allocated 264)
enum RegKernelKind { REG_KERNEL_UPDATE = 0, REG_KERNEL_CHAINBACK = 1, REG_KERNEL_CHAINBACK_ALT = 2, REG_KERNEL_RESUME = 3 };
inline bool reg_kernel_resources(const RegCode& rc, int shift, int kind, kd::KernelResources* out, unsigned* dyn_lds_bytes = nullptr) {
    if (dyn_lds_bytes) *dyn_lds_bytes = kind == REG_KERNEL_CHAINBACK ? reg_chainback_dyn_lds_bytes(rc.K, rc.R, false)
                                      : kind == REG_KERNEL_CHAINBACK_ALT ? reg_chainback_dyn_lds_bytes(rc.K, rc.R, true) : 0u;
    const kd::KernelResources* r = nullptr;
    if (rc.jit) {
        const char* name = kind == REG_KERNEL_UPDATE ? (shift ? "vit_jit_update_8" : "vit_jit_update_16")
                         : kind == REG_KERNEL_RESUME ? (shift ? "vit_jit_resume_8" : "vit_jit_resume_16")
                         : kind == REG_KERNEL_CHAINBACK ? "vit_jit_chainback" : "vit_jit_chainback_alt";
        for (const auto& e : rc.jit->kernels)
            if (e.first == name) r = &e.second;
    } else {
        // Itanium mangling of vit::<kernel><RegSpec<K, R, G0, G1, G2, G3, LANE_BITS, G4, G5>[, SHIFT]>(Args)
        char spec[128], tail[48];
        snprintf(spec, sizeof(spec), "7RegSpecILi%dELi%dELj%uELj%uELj%uELj%uELi%dELj%uELj%uEEE", rc.K, rc.R, rc.G[0], rc.G[1], rc.G[2], rc.G[3],
                 reg_lane_bits(rc.K), rc.G[4], rc.G[5]);
        snprintf(tail, sizeof(tail), "ELi%dEEEvNS_13RegUpdateArgsE", shift ? 8 : 0);
        std::vector<std::string> frag;
        if (kind == REG_KERNEL_UPDATE) frag = {"17reg_update_kernelI", spec, tail};
        else if (kind == REG_KERNEL_RESUME) frag = {"17reg_resume_kernelI", spec, tail};
        else if (kind == REG_KERNEL_CHAINBACK) frag = {"20reg_chainback_kernelI", spec};
        else frag = {"24reg_chainback_alt_kernelI", spec};
        r = kd::find(kd::own_library(), frag);
    }
    if (!r) return false;
    *out = *r;
    return true;
}

constexpr unsigned SIMD_VGPRS = 512, CU_LDS_BYTES = 160 * 1024;
// can `n_update` update waves and one chainback wave of this code share a SIMD: registers (the descriptors' allocations), and per
// CU the LDS of 4 n_update update waves plus two chainback workgroups (a 65536-frame batch is 512 of them on 256 CUs)?  When they
// cannot (K = 9, R = 4: one update wave allocates 360 registers), the chainback of a batch only runs in the gaps between update
// kernels, and the pipeline does better feeding the SIMDs half-size sub-batches from two streams.  Descriptors that cannot be
// read (the library file moved away under the process) answer "no": the conservative schedule.
inline bool reg_chainback_fits_beside_updates(const RegCode& rc, int shift, int n_update, bool alt_chainback = false) {
    kd::KernelResources u, c;
    unsigned dyn = 0;
    if (!reg_kernel_resources(rc, shift, REG_KERNEL_UPDATE, &u) ||
        !reg_kernel_resources(rc, shift, alt_chainback ? REG_KERNEL_CHAINBACK_ALT : REG_KERNEL_CHAINBACK, &c, &dyn))
        return false;
    if ((unsigned)n_update * u.vgpr_alloc + c.vgpr_alloc > SIMD_VGPRS) return false;
    return 4u * (unsigned)n_update * u.lds_static_bytes + 2u * (c.lds_static_bytes + dyn) <= CU_LDS_BYTES;
}
inline bool reg_chainback_fits_beside_two_updates(const RegCode& rc, int shift) { return reg_chainback_fits_beside_updates(rc, shift, 2); }

inline int reg_jit_launch(hipFunction_t fn, const void* args, size_t args_bytes, unsigned grid, unsigned block, hipStream_t st,
                          unsigned dyn_lds_bytes = 0) {
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void*>(args), HIP_LAUNCH_PARAM_BUFFER_SIZE, &args_bytes,
                      HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, dyn_lds_bytes, st, nullptr, config) == hipSuccess ? 0 : -1;
}

// steps [first_step, first_step + n_steps) of every frame.  d_metrics_in == null: reset(start_state) (first_step must be 0);
// else resume from those metrics.  sym_stride: soft_t elements between the chunks of consecutive frames.
inline int reg_update(const RegCode& rc, const DevConfig& cfg, int shift, const void* d_symbols, size_t sym_stride, size_t frames,
                      size_t first_step, size_t n_steps, size_t L, void* d_ws, const void* d_metrics_in, void* d_metrics,
                      uint64_t* d_renorm, const uint32_t* d_start, hipStream_t st) {
    if (frames == 0 || n_steps == 0) return 0;
    RegUpdateArgs a{};
    a.symbols = (const uint8_t*)d_symbols;
    a.sym_frame_stride_bytes = sym_stride * (shift ? 1 : 2);
    // per-lane 32-bit buffer offsets; a resumed call's look-ahead may form slightly negative offsets (see reg_update_body):
    // they must stay distinguishable from valid ones (sign bit) and out of the descriptor's range
    if (a.sym_frame_stride_bytes * (size_t)rc.tile + 65536 >= (d_metrics_in ? 0x7FFF0000ull : 0xFFFF0000ull)) return -2;
    // the last frame's chunk ends n_steps into its stride
    a.sym_total_bytes = (frames - 1) * a.sym_frame_stride_bytes + n_steps * (size_t)rc.R * (shift ? 1 : 2);
    a.ws = (uint4*)d_ws;
    a.ws_tile_stride = reg_groups(rc, L) * 64;
    a.metrics_out = d_metrics;
    a.renorm_sum = d_renorm;
    a.start_state = d_start;
    a.metrics_in = d_metrics_in;
    a.frames = (u32)frames;
    a.t_begin = (u32)first_step;
    a.t_end = (u32)(first_step + n_steps);
    a.cfg = cfg;
    for (int i = 0; i < 6; ++i) a.gen_G[i] = rc.G[i];
#ifdef VIT_HIP_CLOCK_STAMPS
    a.stamps = g_clock_stamps;
#endif
    const unsigned tiles = (unsigned)reg_tiles(rc, frames);
    if (rc.jit) return reg_jit_launch((d_metrics_in ? rc.jit->resume : rc.jit->update)[shift ? 1 : 0], &a, sizeof(a), tiles, 64, st);
    switch (rc.id) {
        case 0: return reg_launch_update<0>(shift, a, tiles, st);
        case 1: return reg_launch_update<1>(shift, a, tiles, st);
        case 2: return reg_launch_update<2>(shift, a, tiles, st);
        case 3: return reg_launch_update<3>(shift, a, tiles, st);
        case 4: return reg_launch_update<4>(shift, a, tiles, st);
        case 5: return reg_launch_update<5>(shift, a, tiles, st);
        case 6: return reg_launch_update<6>(shift, a, tiles, st);
        default: return -1;
    }
}

// `prefer_alt`: launch the code's OTHER chainback kernel (K = 7: the LDS-ring body, 32 registers -- what the pipeline asks for
// when the chainback shares SIMDs with update waves; K = 9: the cooperative body): vit_hip_chainback_batch_ex picks it.
inline int reg_chainback(const RegCode& rc, const void* d_ws, size_t frames, size_t L, uint8_t* d_out, const uint32_t* d_end,
                         hipStream_t st, unsigned wave_priority = 0, bool prefer_alt = false) {
    if (frames == 0 || L == 0) return 0;
    RegChainbackArgs a{};
    a.ws = (const uint4*)d_ws;
    a.ws_tile_stride = reg_groups(rc, L) * 64;
    a.out = d_out;
    a.end_state = d_end;
    a.frames = (u32)frames;
    a.L = (u32)L;
    a.wave_priority = wave_priority;
    const unsigned tiles = (unsigned)reg_tiles(rc, frames);
    // K = 7, 9: the other chainback kernel of the code (reg_chainback_alt_body)
    bool coop = prefer_alt && (rc.K == 9 || rc.K == 7);
#ifdef VIT_HIP_EXPERIMENTS
    if (const char* e = getenv("VIT_HIP_CHAINBACK_ALT")) coop = (rc.K == 9 || rc.K == 7) && *e == '1';   // A/B builds only
#endif
    if (rc.jit) {
        if (coop && rc.jit->chainback_alt)
            return reg_jit_launch(rc.jit->chainback_alt, &a, sizeof(a), rc.K == 9 ? tiles : (unsigned)((frames + 127) / 128), 64, st,
                                  reg_chainback_dyn_lds_bytes(rc.K, rc.R, true));
        const unsigned fpb = rc.jit->chainback_frames_per_block;
        return reg_jit_launch(rc.jit->chainback, &a, sizeof(a), (unsigned)((frames + fpb - 1) / fpb), 64, st,
                              reg_chainback_dyn_lds_bytes(rc.K, rc.R, false));
    }
    switch (rc.id) {
        case 0: return reg_launch_chainback<0>(a, tiles, st, coop);
        case 1: return reg_launch_chainback<1>(a, tiles, st, coop);
        case 2: return reg_launch_chainback<2>(a, tiles, st, coop);
        case 3: return reg_launch_chainback<3>(a, tiles, st, coop);
        case 4: return reg_launch_chainback<4>(a, tiles, st, coop);
        case 5: return reg_launch_chainback<5>(a, tiles, st, coop);
        case 6: return reg_launch_chainback<6>(a, tiles, st, coop);
        default: return -1;
    }
}

inline int reg_export(const RegCode& rc, const void* d_ws, size_t frames, size_t n_steps, size_t L, uint64_t* d_out,
                      hipStream_t st) {
    if (frames == 0 || n_steps == 0) return 0;
    RegExportArgs a{};
    a.ws32 = (const u32*)d_ws;
    a.ws_tile_stride = reg_groups(rc, L) * 64;
    a.out = d_out;
    a.frames = (u32)frames;
    a.n_steps = (u32)n_steps;
    const size_t W = rc.K >= 7 ? (size_t)1 << (rc.K - 7) : 1;
    const size_t total = frames * n_steps * W;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (rc.jit) return reg_jit_launch(rc.jit->export_, &a, sizeof(a), blocks, 256, st);
    switch (rc.K) {
        case 9: return reg_launch_export<3>(a, blocks, st);
        case 3: return reg_launch_export<5>(a, blocks, st);
        case 5: return reg_launch_export<6>(a, blocks, st);
        default: return reg_launch_export<0>(a, blocks, st);
    }
}
#endif  // VIT_REG_ID
#endif  // VIT_REG_JIT_TU

}  // namespace vit
