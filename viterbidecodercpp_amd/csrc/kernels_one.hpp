// kernels_one.hpp -- the LATENCY route: ONE frame, ONE wavefront, K <= 7 (at most 64 states), behind vit_hip_update_host /
// vit_hip_chainback_host -- i.e. behind the header-level drop-in ViterbiDecoder_HIP::update() + ViterbiDecoder_Core::chainback()
// of a single decoder object (examples/run_simple.cpp:67-80; BASELINE configs[0]).
//
// Device implementation of the reference's scalar strategy for one frame
//   ViterbiDecoder_Scalar::update / bfly / renormalise   include/viterbi/viterbi_decoder_scalar.h:29-153
//   ViterbiDecoder_Core::chainback                       include/viterbi/viterbi_decoder_core.h:214-236
// A single frame is a chain of L + K - 1 dependent trellis steps: nothing but the length of one step's dependent chain
// matters.  PLAN_LDS (the general kernel this route used before) goes through LDS and a workgroup barrier twice per step --
// about 400 ns per step, 3.3 ms for an 8192-bit frame.  Here:
//   * lane == state (the reference's order, so __ballot(decision) IS the reference's decision word); the 64 path metrics live in
//     ONE register, in the device's 16-bit domain (u8 metrics in the high byte: 16-bit wrapping arithmetic == mod-256
//     arithmetic);
//   * the two predecessor metrics old[j], old[j + H] (j = s >> 1) come through the LDS crossbar without touching memory
//     (ds_bpermute_b32 x 2, issued together): the only cross-lane traffic of a step;
//   * the dependent chain of a step is bpermute -> add -> min (the decision hangs off the side): the branch metrics of the step
//     depend on the symbols only and are formed while the bpermutes travel -- per group of 64 steps lane l precomputes
//     (|low - y| , |high - y|) of step l as one packed register per polynomial, the step loop reads step k's pair with one
//     v_readlane and each lane rotates its choice of the pair into place (v_alignbit by 0 or 16);
//   * decision words are parked one per lane (v_writelane) and stored as one coalesced 512-byte row group every 64 steps;
//   * renormalisation (new[0] >= threshold: lane 0's register, v_readfirstlane + scalar compare) is a rare wave-uniform branch
//     with a cross-lane minimum (__shfl_xor).
// The chainback kernel copies the frame's decision rows into LDS with all 64 lanes and chases 64 SEGMENTS of the frame at once,
// each from a speculated top state that a sequential pass then verifies (one_chainback_kernel below): exact, and microseconds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "common.hpp"
#include "kernels_lds.hpp"

namespace vit {

template <class F, int... Is>
__device__ __forceinline__ void one_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void one_static_for(F&& f) { one_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct OneUpdateArgs {
    const uint8_t* symbols;      // device: n_steps * R soft_t of ONE frame
    size_t sym_total_bytes;
    uint64_t* decisions;         // [n_steps] 64-bit words (W = 1: K <= 7), rows of THIS call
    void* metrics_io;            // [N] error_t: read (no reset) and written
    uint64_t* renorm_sum;        // [1]
    const uint16_t* pattern;     // [H] bit i = (branch_table[i][j] == high)
    int32_t K, n_steps;
    DevConfig cfg;
    // frame route (one_frame_kernel): the metrics to start from travel IN THE KERNEL ARGUMENTS (128 bytes, device domain: u8 metrics
    // already in the high byte) -- no load from host memory in front of the first step; metrics_io is then only written
    int32_t metrics_in_args;
    uint16_t metrics_in[64];
};

template <int R, int SHIFT>
__device__ __forceinline__ void one_update_body(const OneUpdateArgs& a) {
    const int N = 1 << (a.K - 1);
    const int H = N >> 1;
    const int lane = threadIdx.x & 63;
    // lanes beyond the N states mirror the lower ones (same predecessors, same results): no lane predicate on the chain; their
    // decision bits are masked out of the ballot
    const int s = lane & (N - 1);
    const int j = (s >> 1) & (H > 0 ? H - 1 : 0);
    const uint64_t live = N >= 64 ? ~0ull : ((1ull << N) - 1ull);
    const int addr0 = 4 * j, addr1 = 4 * (j + H);               // ds_bpermute byte addresses of the predecessors' lanes

    // per polynomial: 16 where this lane's butterfly expects `high` (branch_table[i][j] == high), else 0 -- the rotate count that
    // brings |high - y| (upper half of the packed pair below) or |low - y| (lower half) into the low 16 bits
    const uint32_t p = a.pattern[j];
    uint32_t rot[R];
#pragma unroll
    for (int i = 0; i < R; ++i) rot[i] = ((p >> i) & 1u) ? 16u : 0u;

    uint32_t m;                                                 // the path metric of state `s`, low 16 bits
    if (a.metrics_in_args) m = a.metrics_in[s];
    else if (SHIFT) m = (uint32_t)(((const uint8_t*)a.metrics_io)[s]) << 8;
    else m = ((const uint16_t*)a.metrics_io)[s];

    // The symbols of a step are the same for every lane.  Per group of 64 steps lane l forms (|low - y_i| | |high - y_i| << 16) of
    // step t0 + l for the R symbols (scalar.h:66-73, in soft_t then error_t width: abs_soft) -- R packed registers -- and the step
    // loop picks step k's values out of lane k with one v_readlane each: nothing of the symbol path is on a step's chain.
    auto symbol_pairs = [&](int t0, uint32_t (&ap)[R]) __attribute__((always_inline)) {
        const int t = t0 + lane;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            int32_t y = 0;
            if (t < a.n_steps) {
                if (SHIFT) y = (int32_t)(int16_t)(uint16_t)((uint32_t)((const uint8_t*)a.symbols)[(size_t)t * R + i] << 8);
                else y = (int32_t)((const int16_t*)a.symbols)[(size_t)t * R + i];
            }
            ap[i] = (uint32_t)abs_soft(a.cfg.low, y) | ((uint32_t)abs_soft(a.cfg.high, y) << 16);
        }
    };

    uint64_t acc = 0;        // sum of subtracted minima (update()'s return value)
    uint32_t w_lo = 0, w_hi = 0;     // the decision word this lane parks: step (64 k + lane) of the call
    const uint32_t max_error = a.cfg.max_error, threshold = a.cfg.threshold;
    const bool b_in = (s & 1) != 0;                             // input bit of this lane's next-state
    const uint32_t XA = b_in ? 0xFFFFu : 0u, CA = b_in ? max_error + 1u : 0u;
    const uint32_t XB = b_in ? 0u : 0xFFFFu, CB = b_in ? 0u : max_error + 1u;

    // state 0's metric after the last step run, as a signed value: -1 = "no step yet" (below every threshold, 0 included: the
    // reference tests after a step, never the metrics a call starts from)
    uint32_t m_state0 = 0xFFFFFFFFu;
    auto renormalise = [&]() __attribute__((always_inline)) {
        uint32_t mn = m;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)mn, off);
            mn = o < mn ? o : mn;
        }
        m = (m - mn) & 0xFFFFu;
        acc += (uint64_t)(mn >> SHIFT);
    };

    uint32_t ap_next[R];
    symbol_pairs(0, ap_next);
    for (int t0 = 0; t0 < a.n_steps; t0 += 64) {
        uint32_t ap[R];
#pragma unroll
        for (int i = 0; i < R; ++i) ap[i] = ap_next[i];
        symbol_pairs(t0 + 64, ap_next);                         // in flight across this group's 64 steps
        const int nb = a.n_steps - t0 < 64 ? a.n_steps - t0 : 64;
        // the 64 steps of a group are UNROLLED: the lane a step reads its symbols from and parks its decision word in is then an
        // inline constant of v_readlane / v_writelane (with a run-time lane both take it from an SGPR, and v_writelane has room
        // for one scalar operand only: the word); a partial last group leaves through the scalar test in front of each step
        auto group = [&](auto partial_c) __attribute__((always_inline)) {
        one_static_for<64>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if (!decltype(partial_c)::value || k < nb) {
            // ---- the dependent chain of the step starts here: the predecessors' metrics ----
            uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(addr0, (int)m);
            uint32_t pb = (uint32_t)__builtin_amdgcn_ds_bpermute(addr1, (int)m);
            // renormalise when new_metric[0] >= threshold  (scalar.h:48-50, :139-153): state 0 is lane 0.  The test of the PREVIOUS
            // step's result sits HERE, behind the issue of this step's fetches: v_readfirstlane -> s_cmp -> s_cbranch used to end every
            // step's chain (the next fetch could not issue before the branch had resolved: ~30 of a step's ~165 clocks); now they
            // resolve while the fetches travel.  The subtraction commutes with nothing it skips -- the decisions of the step that
            // tripped are final, and the only readers of the renormalised metrics are this step's fetches, which the rare branch
            // simply issues again from the corrected register
            if (__builtin_expect((int32_t)m_state0 >= (int32_t)threshold, 0)) {
                renormalise();
                pa = (uint32_t)__builtin_amdgcn_ds_bpermute(addr0, (int)m);
                pb = (uint32_t)__builtin_amdgcn_ds_bpermute(addr1, (int)m);
            }
            // ... while they travel: e = sum_i |expected_i - y_i|  (scalar.h:66-73) from lane k's packed pairs
            uint32_t e = 0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const uint32_t pair = (uint32_t)__builtin_amdgcn_readlane((int)ap[i], k);
                e += __builtin_amdgcn_alignbit(pair, pair, rot[i]);            // low half: the lane's choice of the pair
            }
            // xa = b ? max_error - e : e  (via predecessor (0|X), scalar.h:107,:113,:115), xb the other way round (:114,:116).  Mod
            // 2^16, max_error - e == ~e + (max_error + 1): xa = (e ^ XA) + CA with per-lane constants, so that each candidate is ONE
            // xor (formed while the bpermutes travel) and ONE three-operand add behind them
            const uint32_t m0 = (pa + CA + (e ^ XA)) & 0xFFFFu, m1 = (pb + CB + (e ^ XB)) & 0xFFFFu;
            const bool d = m0 > m1;                                            // strict: tie -> 0        :123-124
            m = m0 < m1 ? m0 : m1;                                             // d ? m1 : m0             :127-128
            const uint64_t word = __ballot(d) & live;                          // :131-134
            asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(w_lo) : "s"((uint32_t)word), "n"(k));   // (s_nop: see one_update7_body)
            asm("v_writelane_b32 %0, %1, %2" : "+v"(w_hi) : "s"((uint32_t)(word >> 32)), "n"(k));
            m_state0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);      // tested behind the NEXT step's fetches (or after the last step)
            }
        });
        };
        if (nb == 64) group(std::false_type{});       // whole groups: no per-step test
        else group(std::true_type{});
        // the group's rows: one coalesced store (512 bytes when the group is whole)
        if (lane < nb) a.decisions[(size_t)t0 + (size_t)lane] = ((uint64_t)w_hi << 32) | w_lo;
    }
    if ((int32_t)m_state0 >= (int32_t)threshold) renormalise();      // the last step's test

    if (lane < N) {
        if (SHIFT) ((uint8_t*)a.metrics_io)[lane] = (uint8_t)(m >> 8);
        else ((uint16_t*)a.metrics_io)[lane] = (uint16_t)m;
    }
    if (lane == 0) a.renorm_sum[0] = acc;
}

// ---- K = 7 (64 states: a full wavefront), R <= 4: the IN-PLACE trellis with two helper wavefronts (round 6, second session) ----
// The body above pays two LDS round trips and ~22 instructions per step (70 ns).  What a step of a lone wavefront costs is the length
// of its DEPENDENT chain and the instructions on it; everything that is not the add-compare-select itself moves to other wavefronts of
// the workgroup, which has three:
//   * wave 0 walks the trellis with the index rotation of the batch kernels (kernels_reg.hpp): at step t (phase ph = t mod 6) lane l
//     holds state rotl^ph(l), the butterfly {(0|X),(1|X)} -> {(X|0),(X|1)} reads and writes the two lanes that differ in lane bit
//     p = 5 - ph, and the lane whose bit p is b makes state (X|b).  Both members' metrics reach both lanes with one swap
//     (v_permlane32_swap / v_permlane16_swap of the register with its copy: every lane then holds (lo member, hi member)) or two
//     bank-masked DPP moves (row_shr / row_shl by 8 or 4, quad_perm for the two low bits) -- no LDS crossbar on the chain;
//     u = lo + eu, v = hi + ev, new = min(u, v) (written twice: the next exchange consumes two copies), decision = u > v (strict,
//     scalar.h:123-124) pushed into a per-lane history word by ONE v_addc (carry-in = the compare's VCC).  The lane's
//     (eu, ev) = (b ? max_error - e : e, b ? e : max_error - e) of its butterfly's branch pattern is ONE ds_read_b64, four steps
//     ahead, out of a table wave 1 has made.  Chain: exchange -> add -> min; 10 instructions per step.
//     The threshold test (scalar.h:48-50) would add a v_cmp and a scalar branch that waits for it: 9 of 36 ns.  A whole block of 32
//     steps CANNOT renormalise when state 0's metric plus 32 times the block's largest branch metric stays below the threshold
//     (new[0] <= old[0] + e, :113,:127; wave 1 publishes the maximum with the table): such blocks run a copy of the loop without the
//     test, the others test every step -- one step LATE, behind the next step's exchange and adds, on which the subtraction of the
//     minimum is made up (it commutes with both; the decisions of the step that tripped are final).
//   * wave 1, lane == step: per block of 32 steps it forms {e, max_error - e} of all 2^R patterns for both b (scalar.h:66-73,:107)
//     from symbols it fetched a whole period of six blocks earlier (host-mapped memory on the frame route: the PCIe round trip must
//     not become the step time -- and it did, twice: behind __syncthreads(), whose fence drains vmcnt, and behind the single
//     vmcnt(0) hipcc puts at the top of a loop that joins paths with different loads in flight: the six loads of the NEXT period
//     are issued right behind that wait, so that everything it waits for is a period old).  Loads only in this wavefront: with the
//     row stores beside them every use of a symbol waited for vmcnt(0) (loads and stores retire out of order against each other).
//   * wave 2, lane == next-state: it turns the block of history words wave 0 parked in LDS into the reference's decision rows: lane
//     s' reads the history word of the lane that held s' (a rotation of its own index by the row's phase), one bit per row -> v_cmp ->
//     the 64-bit row, parked with v_writelane and stored coalesced, 256 bytes per block.  Rows in HBM are the reference's (chainback,
//     export and the host's m_decisions never see the rotation).
// One hardware barrier per block of 32 steps hands the table and the history over (LDS only: s_waitcnt lgkmcnt(0) + s_barrier).
// Results are bit for bit the body's above; 30 ns per step (scripts/host_route_latency.py; profiles/r6_frame_route.txt).
template <int R, int SHIFT>
__device__ __forceinline__ void one_update7_body(const OneUpdateArgs& a) {
    static_assert(R >= 1 && R <= 4, "the per-step table holds 2^(R+1) entries");
    constexpr int NPAT = 1 << R;
    constexpr int ROWB = NPAT * 16;            // bytes per step: [pattern]{b = 0: (e, M - e), b = 1: (M - e, e)} as four dwords
    constexpr int BLK = 32, SUP = 96;          // steps per block / per unrolled super-block (whole phases: 96 = 16 x 6)
    __shared__ __attribute__((aligned(16))) uint32_t one_bm[SUP * NPAT * 4];
    __shared__ uint32_t one_hist[2][64];
    __shared__ uint32_t one_emax[3];           // per ring third: the largest branch metric of the block (what a step can add to a path at most)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_steps = a.n_steps;
    const int nblocks = (n_steps + BLK - 1) / BLK;
    // every wavefront meets at the barrier nbar times inside its loop: the blocks, rounded up to the producer's unrolled period -- its
    // loop then has no conditional in it, and hipcc can count its loads (a join of paths with different numbers of loads in flight
    // makes it wait for vmcnt(0), i.e. for the PCIe round trip of the newest)
    constexpr int NS = 6;                              // blocks between the fetch of a step's symbols and their use
    const int nbar = (nblocks + NS - 1) / NS * NS;
    const uint32_t max_error = a.cfg.max_error, threshold = a.cfg.threshold;
    // The per-block hand-over concerns LDS only (table rows one way, history words the other): wait for this wave's LDS operations and
    // meet at the hardware barrier.  NOT __syncthreads(): its workgroup-scope fence also drains vmcnt, i.e. the helper would sit out
    // the PCIe round trip of the symbols it has just asked for at EVERY block (first build: 1.65 us per block whatever wave 0 did)
    // (sched_barrier: hipcc otherwise hoists the arithmetic on every symbol slot to the top of the producer's unrolled loop, and with it
    // one wait for ALL the loads in flight)
    auto lds_barrier = []() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto rotl6 = [](uint32_t x, int n) __attribute__((always_inline)) -> uint32_t { return n == 0 ? x : (((x << n) | (x >> (6 - n))) & 63u); };
    auto rotr6 = [&](uint32_t x, int n) __attribute__((always_inline)) -> uint32_t { return rotl6(x, (6 - n) % 6); };

    if (wave == 0) {
        // ---------------- the trellis ----------------
        uint32_t m = a.metrics_in_args ? (uint32_t)a.metrics_in[lane]
                   : SHIFT ? (uint32_t)(((const uint8_t*)a.metrics_io)[lane]) << 8 : (uint32_t)((const uint16_t*)a.metrics_io)[lane];
        // where this lane's (eu, ev) sits inside a step's table row, per phase: pattern of butterfly X = rotl^ph(lane without bit p)
        uint32_t tab[6];
#pragma unroll
        for (int ph = 0; ph < 6; ++ph) {
            const int pbit = 5 - ph;
            const uint32_t b = ((uint32_t)lane >> pbit) & 1u;
            const uint32_t X = rotl6((uint32_t)lane & ~(1u << pbit), ph);          // (0|X): below 32
            tab[ph] = ((uint32_t)(a.pattern[X] & (NPAT - 1)) * 2u + b) * 8u;
        }
        uint64_t acc = 0;
        uint32_t hist = 0;
        auto renormalise = [&]() __attribute__((always_inline)) {
            uint32_t mn = m;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)mn, off);
                mn = o < mn ? o : mn;
            }
            m = (m - mn) & 0xFFFFu;
            acc += (uint64_t)(mn >> SHIFT);
        };
        const uint32_t thr_lane = lane == 0 ? threshold : 0xFFFFFFFFu;
        lds_barrier();                                 // the helper's table of block 0
        // the lane's (eu, ev) of super-block step ts: fetched several steps ahead of its use (an LDS round trip is longer than a step)
        auto fetch = [&](auto tc) __attribute__((always_inline)) -> uint2 {
            constexpr int ts = decltype(tc)::value;
            return *(const uint2*)((const char*)one_bm + tab[ts % 6] + ts * ROWB);
        };
        // The dependent chain of a step is exchange -> add -> min; everything else hangs off its side.  Two things keep it that short:
        // the minimum is written TWICE (m and m2), so that the exchange -- which consumes two copies of the register -- needs no move
        // in front of it; and the threshold test of step t (a v_cmp into a scalar pair) is only BRANCHED on behind the issue of step
        // t + 1's exchange, which ran on the untested metrics: the rare branch renormalises and exchanges again (the decisions of the
        // step that tripped are final, and the exchange is the only reader of the metrics in front of the test) -- on the exchanged pair.
        uint32_t m2 = m;
        uint64_t pend = 0;                             // lanes whose new metric reached the threshold in the previous step (lane 0 or none)
        auto exchange = [&](auto phc, uint32_t& lo, uint32_t& hi) __attribute__((always_inline)) {
            constexpr int PH = decltype(phc)::value;
            if constexpr (PH == 0) {
                auto r2 = __builtin_amdgcn_permlane32_swap(m, m2, false, false);
                lo = r2[0]; hi = r2[1];
            } else if constexpr (PH == 1) {
                auto r2 = __builtin_amdgcn_permlane16_swap(m, m2, false, false);
                lo = r2[0]; hi = r2[1];
            } else if constexpr (PH == 2) {        // lane bit 3: row_shr:8 into banks 2, 3; row_shl:8 into banks 0, 1
                lo = (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m2, 0x118, 0xF, 0xC, false);
                hi = (uint32_t)__builtin_amdgcn_update_dpp((int)m2, (int)m2, 0x108, 0xF, 0x3, false);
            } else if constexpr (PH == 3) {        // lane bit 2: row_shr:4 into banks 1, 3; row_shl:4 into banks 0, 2
                lo = (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m2, 0x114, 0xF, 0xA, false);
                hi = (uint32_t)__builtin_amdgcn_update_dpp((int)m2, (int)m2, 0x104, 0xF, 0x5, false);
            } else if constexpr (PH == 4) {        // lane bit 1: quad_perm [0,1,0,1] / [2,3,2,3]
                lo = (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m2, 0x44, 0xF, 0xF, false);
                hi = (uint32_t)__builtin_amdgcn_update_dpp((int)m2, (int)m2, 0xEE, 0xF, 0xF, false);
            } else {                               // lane bit 0: quad_perm [0,0,2,2] / [1,1,3,3]
                lo = (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m2, 0xA0, 0xF, 0xF, false);
                hi = (uint32_t)__builtin_amdgcn_update_dpp((int)m2, (int)m2, 0xF5, 0xF, 0xF, false);
            }
        };
        // TEST = false: a block that CANNOT renormalise (state 0's metric plus 32 times the block's largest branch metric stays below
        // the threshold: new[0] <= old[0] + e, scalar.h:113,:127) runs without the per-step test -- a v_cmp whose result a scalar branch
        // waits for costs a lone wavefront 9 of a step's 36 ns
        auto step = [&](auto tc, auto testc, const uint2 bm) __attribute__((always_inline)) {
            constexpr int ts = decltype(tc)::value;        // step inside the super-block: phase and table slot are compile time
            constexpr int PH = ts % 6;
            constexpr bool TEST = decltype(testc)::value;
            uint32_t lo, hi;
            exchange(std::integral_constant<int, PH>{}, lo, hi);
            // candidates in the low 16 bits (whatever the adds carry above them is ignored by the 16-bit compare and minimum, and the
            // minimum writes zeros there)
            uint32_t u = lo + bm.x, v = hi + bm.y;
            if constexpr (TEST) asm volatile("" : "+v"(u), "+v"(v), "+v"(lo), "+v"(hi));        // issued in front of the branch, not sunk behind it
            // renormalise when new_metric[0] >= threshold (:48-50), the PREVIOUS step's: state 0 is lane 0 in every phase.  On the
            // exchanged pair (every state's metric is some lane's lo or hi), and behind the adds: subtracting the minimum commutes with
            // both, and the scalar branch has the exchange AND the adds between it and the v_cmp it waits for
            if (TEST && __builtin_expect(pend != 0, 0)) {
                uint32_t mn = (lo & 0xFFFFu) < (hi & 0xFFFFu) ? lo : hi;
                mn &= 0xFFFFu;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const uint32_t o = (uint32_t)__shfl_xor((int)mn, off);
                    mn = o < mn ? o : mn;
                }
                u -= mn; v -= mn;                      // (mod 2^16: only the low halves are read from here on)
                acc += (uint64_t)(mn >> SHIFT);
            }
            // decision = u > v (tie -> 0, :123-124) straight into the history word (hist = 2 hist + carry), new metric = the smaller (:127-128)
            uint32_t &hs = hist, &mm = m, &mm2 = m2;
            // (order inside the asm: hipcc cannot see into it, and on gfx940+ a vector instruction that reads VCC wants two
            // instructions between itself and the one that wrote it -- the two minima are those; what FOLLOWS the block hipcc pads
            // itself: it puts s_nop 1 in front of a DPP move or lane swap that reads the block's outputs)
            asm volatile("v_cmp_gt_u16 vcc, %3, %4\n\tv_min_u16 %1, %3, %4\n\tv_min_u16 %2, %3, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                         : "+v"(hs), "=&v"(mm), "=&v"(mm2) : "v"(u), "v"(v) : "vcc");
            // the other lanes compare against a value no metric reaches: one v_cmp into a scalar pair, branched on one step later
            if constexpr (TEST) pend = __builtin_amdgcn_ballot_w64(m >= thr_lane);
        };
        for (int b0 = 0; b0 < nblocks; b0 += 3) {
            one_static_for<3>([&](auto bbc) __attribute__((always_inline)) {
                constexpr int bb = decltype(bbc)::value;
                if (b0 + bb < nblocks) {
                    const int nb = n_steps - (b0 + bb) * BLK < BLK ? n_steps - (b0 + bb) * BLK : BLK;
                    constexpr int QD = 4;                        // table rows in flight (2, 4, 8: the same step time)
                    uint2 q[QD];
                    one_static_for<QD>([&](auto dc) __attribute__((always_inline)) { q[decltype(dc)::value] = fetch(std::integral_constant<int, bb * BLK + decltype(dc)::value>{}); });
                    auto one = [&](auto kc, auto testc) __attribute__((always_inline)) {
                        constexpr int k = decltype(kc)::value;
                        const uint2 bm = q[k % QD];
                        if constexpr (k + QD < BLK) q[k % QD] = fetch(std::integral_constant<int, bb * BLK + k + QD>{});   // (rows of THIS block only: the next one's appear behind the barrier)
                        step(std::integral_constant<int, bb * BLK + k>{}, testc, bm);
                    };
                    if (nb == BLK) {
                        // can state 0 reach the threshold inside the next GRP steps?  (m[0] + GRP x the block's largest branch metric against
                        // the threshold; a threshold of 0 -- "always" -- never passes.)  Decided once per block: groups of 16 or 8 pass the
                        // test more often -- with the reference's 8-bit soft configuration 32 steps' worth of the bound never fits under
                        // the threshold -- but the decision itself (v_readfirstlane -> scalar compare -> branch, and settling the last
                        // test at the end of every tested group) costs what they save: 8192-bit frames, same box, GRP = 32 / 16 / 8:
                        // Voyager SOFT16 273 / 278 / 288 us, SOFT8 318 / 326 / 326, DAB SOFT8 367 / 382 / 404
                        constexpr int GRP = 32;
                        const uint32_t em = (uint32_t)__builtin_amdgcn_readfirstlane((int)one_emax[bb]);
                        one_static_for<BLK / GRP>([&](auto gc) __attribute__((always_inline)) {
                            constexpr int g0 = decltype(gc)::value * GRP;
                            const uint32_t m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
                            if (m0 + (uint32_t)GRP * em < threshold) {
                                one_static_for<GRP>([&](auto kc) __attribute__((always_inline)) { one(std::integral_constant<int, g0 + decltype(kc)::value>{}, std::false_type{}); });
                            } else {
                                one_static_for<GRP>([&](auto kc) __attribute__((always_inline)) { one(std::integral_constant<int, g0 + decltype(kc)::value>{}, std::true_type{}); });
                                // the group's last test is settled here (every group starts with nothing pending)
                                if (pend != 0) { renormalise(); m2 = m; pend = 0; }
                            }
                        });
                    } else {
                        one_static_for<BLK>([&](auto kc) __attribute__((always_inline)) { if (decltype(kc)::value < nb) one(kc, std::true_type{}); });
                        hist <<= (BLK - nb);           // the block's first step at bit 31
                        if (pend != 0) { renormalise(); m2 = m; pend = 0; }
                    }
                    one_hist[(b0 + bb) & 1][lane] = hist;
                    lds_barrier();
                }
            });
        }
        for (int b = nblocks; b < nbar; ++b) lds_barrier();
        __syncthreads();                               // (the helper's last conversion)
        const uint32_t st = rotl6((uint32_t)lane, n_steps % 6);     // the state this lane holds after the last step
        if (SHIFT) ((uint8_t*)a.metrics_io)[st] = (uint8_t)(m >> 8);
        else ((uint16_t*)a.metrics_io)[st] = (uint16_t)m;
        if (lane == 0) a.renorm_sum[0] = acc;
    } else if (wave == 1) {
        // ---------------- helper 1: branch-metric tables ahead of the trellis ----------------
        // (loads only: with the row stores in the same wavefront hipcc must wait for vmcnt(0) in front of every use of a symbol --
        // loads and stores retire out of order against each other -- i.e. for the PCIe round trip of the loads it has JUST issued)
        const int16_t* sym16 = (const int16_t*)a.symbols;
        const int8_t* sym8 = (const int8_t*)a.symbols;
        // symbols of the lane's step of blocks x (mod NS): fetched NS blocks ahead of their use -- on the frame route they come from
        // host memory, and the round trip over PCIe (several microseconds) must not become the step time.  Unconditional loads (steps
        // beyond the call's last repeat it: their table rows are never read)
        int32_t ys[NS][R], yn[NS][R];                  // the set in use (blocks 6j + 1 .. 6j + 6) and the set in flight (the next six)
        auto load_syms = [&](int x, int32_t (&dst)[R]) __attribute__((always_inline)) {
            int t = x * BLK + (lane & (BLK - 1));
            t = t < n_steps ? t : n_steps - 1;
#pragma unroll
            for (int i = 0; i < R; ++i) dst[i] = SHIFT ? (int32_t)sym8[(size_t)t * R + i] : (int32_t)sym16[(size_t)t * R + i];   // (8-bit: shifted in produce)
        };
        // the table rows of block x (slot = step mod 96; the two halves of the wavefront write the same rows)
        auto produce = [&](auto bbc, const int32_t (&yy)[R]) __attribute__((always_inline)) {
            constexpr int bb = decltype(bbc)::value % 3;   // x mod 3: ring third
            uint32_t a0[R], a1[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int32_t y = SHIFT ? (int32_t)(int16_t)(uint16_t)((uint32_t)yy[i] << 8) : yy[i];      // the device's 16-bit domain
                a0[i] = abs_soft(a.cfg.low, y); a1[i] = abs_soft(a.cfg.high, y);
            }
            uint4* row = (uint4*)((char*)one_bm + (bb * BLK + (lane & (BLK - 1))) * ROWB);
            uint32_t emax = 0;
#pragma unroll
            for (int p = 0; p < NPAT; ++p) {
                uint32_t e = 0;
#pragma unroll
                for (int i = 0; i < R; ++i) e += ((p >> i) & 1) ? a1[i] : a0[i];
                e &= 0xFFFFu;
                const uint32_t eb = (max_error - e) & 0xFFFFu;
                row[p] = make_uint4(e, eb, eb, e);
                emax = e > emax ? e : emax;
                emax = eb > emax ? eb : emax;
            }
            // the block's largest (the two halves of the wavefront hold the same 32 steps)
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)emax, off);
                emax = o > emax ? o : emax;
            }
            if (lane == 0) one_emax[bb] = emax;
        };
        static_assert(NS % 3 == 0, "the unrolled period names the ring thirds");
        // hipcc joins the paths into the loop with ONE wait for everything in flight at its top: so everything in flight there is a
        // whole period (six blocks, ~5 us) old -- the six loads of the NEXT period are issued right behind that wait
        {
            int32_t y0[R];
            load_syms(0, y0);
            one_static_for<NS>([&](auto sc) __attribute__((always_inline)) { load_syms(1 + decltype(sc)::value, yn[decltype(sc)::value]); });
            produce(std::integral_constant<int, 0>{}, y0);
        }
        lds_barrier();                                 // table of block 0 in place
        for (int b0 = 0; b0 < nbar; b0 += NS) {
            one_static_for<NS>([&](auto sc) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < R; ++i) ys[decltype(sc)::value][i] = yn[decltype(sc)::value][i];
            });
            one_static_for<NS>([&](auto sc) __attribute__((always_inline)) { load_syms(b0 + NS + 1 + decltype(sc)::value, yn[decltype(sc)::value]); });
            one_static_for<NS>([&](auto bbc) __attribute__((always_inline)) {
                constexpr int bb = decltype(bbc)::value;
                produce(std::integral_constant<int, (bb + 1) % NS>{}, ys[bb]);    // block b0 + bb + 1 (beyond the last: rows nobody reads)
                lds_barrier();
            });
        }
        __syncthreads();
    } else {
        // ---------------- helper 2: the reference's decision rows behind the trellis ----------------
        // history words of block xb (one_hist[xb & 1]) -> the reference's rows of its nb steps
        uint32_t src[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) src[c] = rotr6((uint32_t)lane, c);         // the lane that held next-state `lane` in a row of phase c
        uint32_t w_lo = 0, w_hi = 0;                   // the rows this lane parks: row `lane` of the block being converted
        auto convert = [&](int xb, auto bbc) __attribute__((always_inline)) {
            constexpr int bb = decltype(bbc)::value;   // xb mod 3
            const int nb = n_steps - xb * BLK < BLK ? n_steps - xb * BLK : BLK;
            uint32_t h[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) h[c] = one_hist[xb & 1][src[c]];
            one_static_for<BLK>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                constexpr int c = (bb * BLK + k + 1) % 6;           // row t = 32 xb + k holds the states AFTER step t
                const uint64_t word = __ballot((h[c] >> (31 - k)) & 1u);
                uint32_t &wl = w_lo, &wh = w_hi;           // (named here: an asm operand alone does not make the lambda capture them)
                // s_nop: the row comes straight out of a v_cmp (often into VCC), and a v_writelane that reads a scalar register the
                // instruction in front of it has written gets the OLD value -- hipcc cannot see into the asm to pad it (first build:
                // the low word of every row whose compare had nothing between it and the v_writelane was stale)
                asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(wl) : "s"((uint32_t)word), "n"(k));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(wh) : "s"((uint32_t)(word >> 32)), "n"(k));
            });
            if (lane < nb) a.decisions[(size_t)xb * BLK + (size_t)lane] = ((uint64_t)w_hi << 32) | w_lo;
        };
        lds_barrier();
        for (int b0 = 0; b0 < nblocks; b0 += 3) {
            one_static_for<3>([&](auto bbc) __attribute__((always_inline)) {
                constexpr int bb = decltype(bbc)::value;
                const int b = b0 + bb;
                if (b < nblocks) {
                    if (b > 0) convert(b - 1, std::integral_constant<int, (bb + 2) % 3>{});
                    lds_barrier();
                }
            });
        }
        for (int b = nblocks; b < nbar; ++b) lds_barrier();
        if (nblocks > 0) {
            const int xb = nblocks - 1;
            if (xb % 3 == 0) convert(xb, std::integral_constant<int, 0>{});
            else if (xb % 3 == 1) convert(xb, std::integral_constant<int, 1>{});
            else convert(xb, std::integral_constant<int, 2>{});
        }
        __syncthreads();
    }
}
inline bool one_update7_supported(int K, int R) { return K == 7 && R >= 1 && R <= 4; }

template <int R, int SHIFT>
__global__ void __launch_bounds__(64) one_update_kernel(OneUpdateArgs a) { one_update_body<R, SHIFT>(a); }
template <int R, int SHIFT>
__global__ void __launch_bounds__(192) one_update7_kernel(OneUpdateArgs a) { one_update7_body<R, SHIFT>(a); }

struct OneChainbackArgs {
    const uint64_t* decisions;   // [L + K-1] words
    uint8_t* out;                // [ceil(L/8)]
    uint32_t end_state;
    uint32_t L;
    int32_t K;
};

// Chainback of ONE frame, in parallel, exact.  A traceback is a chain of L dependent steps -- but survivor paths merge: chased from
// ANY state, a path joins the true survivor after a few constraint lengths.  So the rows of a chunk (up to 8192 steps, staged in
// LDS by the whole wavefront) are cut into 64 segments of whole bytes, one per lane; lane i first chases WARM steps above its
// segment from state 0 to guess the state at its segment's top, then chases its segment from that guess, writing its bytes.  A
// short sequential pass then VERIFIES the guesses top-down -- segment i's true top state is segment i-1's bottom state -- and any
// segment whose guess was wrong is chased again from the true state (never on real data; random decision rows do it, and the
// tests feed some).  Results are those of the sequential chase in every case; the kernel takes microseconds instead of the third
// of a millisecond a 8192-step dependent chain costs one wavefront.
constexpr int ONE_CB_CHUNK = 8192;       // rows staged per pass (64 KiB of LDS)
constexpr int ONE_CB_WARM = 64;          // speculative steps in front of a segment
__device__ __forceinline__ void one_chainback_body(const OneChainbackArgs& a, uint64_t* one_rows) {
    // one_rows: [ONE_CB_CHUNK + 64 + 8] rows (one pad row per segment), then the chunk's output bytes
    uint8_t* const obytes = (uint8_t*)(one_rows + ONE_CB_CHUNK + 64 + 8);
    const int lane = threadIdx.x & 63;
    const int TSB = a.K - 1;
    const int ignore = TSB < 8 ? TSB : 8;                   // ViterbiTracebackBuffer::get_layout (core.h:129-149)
    const int shift_state = 8 - ignore, shift_tail = TSB - ignore, total_bits = TSB + shift_state;
    const uint32_t smask = (1u << TSB) - 1u;
    uint32_t state_top = a.end_state & smask;               // true state at the top of the current chunk (wave-uniform)
    size_t j = a.L;                                         // decoded bits still to come: j-1 .. 0
    while (j > 0) {
        // the chunk covers decoded bits [j0, j), j0 a multiple of 8.  Inside it rows are addressed top-down: q = n8 - 1 - (bit - j0)
        // with n8 = the chunk's bits rounded up to whole bytes (the frame's top byte may be partial: q then starts above 0).
        const size_t j0 = j > (size_t)ONE_CB_CHUNK ? ((j - ONE_CB_CHUNK + 7) & ~(size_t)7) : 0;
        const uint32_t n = (uint32_t)(j - j0), n8 = (n + 7u) & ~7u, q_first = n8 - n;
        // 64 segments of `seg` rows (whole bytes), lane i: q in [i seg, (i+1) seg); LDS slot of row q = q + q / seg -- one pad row
        // per segment, so that the lanes' row streams (seg rows apart) start two banks apart instead of in the same bank
        const uint32_t seg = ((n8 / 8u + 63u) / 64u) * 8u;
        const uint32_t nseg = (n8 + seg - 1u) / seg;
        for (uint32_t q = q_first + (uint32_t)lane; q < n8; q += 64) one_rows[q + q / seg] = a.decisions[j0 + (size_t)(n8 - 1u - q) + (size_t)TSB];
        __syncthreads();
        auto step = [&](uint32_t& reg, uint32_t slot) __attribute__((always_inline)) {
            const uint64_t w = one_rows[slot];
            const uint32_t state = reg >> shift_state;
            const uint32_t bit = (uint32_t)(w >> state) & 1u;
            reg = (reg >> 1) | (bit << (total_bits - 1));
        };
        // chase segment `sidx` from `top_state`, writing its bytes where `writer`; returns the state below it.  Bit (relative to j0)
        // of row q: n8 - 1 - q; a byte is complete when that is a multiple of 8
        auto run_segment = [&](uint32_t sidx, uint32_t top_state, bool writer) __attribute__((always_inline)) -> uint32_t {
            const uint32_t qa = sidx * seg > q_first ? sidx * seg : q_first;
            const uint32_t qb = (sidx + 1u) * seg < n8 ? (sidx + 1u) * seg : n8;
            uint32_t reg = top_state << shift_state;
            for (uint32_t q = qa; q < qb; ++q) {
                step(reg, q + sidx);
                const uint32_t rel = n8 - 1u - q;
                if ((rel & 7u) == 0 && writer) obytes[rel >> 3] = (uint8_t)((reg >> shift_tail) & 0xFFu);
            }
            return (reg >> shift_state) & smask;
        };
        uint32_t guess = state_top, bottom = 0;
        if ((uint32_t)lane < nseg) {
            if (lane > 0) {
                // warm-up: from state 0, up to WARM rows above the segment's top (all inside the chunk)
                const uint32_t qt = (uint32_t)lane * seg;
                const uint32_t warm = qt - q_first < (uint32_t)ONE_CB_WARM ? qt - q_first : (uint32_t)ONE_CB_WARM;
                uint32_t reg = 0;
                for (uint32_t q = qt - warm; q < qt; ++q) step(reg, q + q / seg);
                guess = (reg >> shift_state) & smask;
            }
            bottom = run_segment((uint32_t)lane, guess, true);
        }
        __syncthreads();
        // verification, top-down (wave-uniform): the true top state of segment i is the bottom state of segment i - 1
        uint32_t truth = state_top;
        for (uint32_t i = 0; i < nseg; ++i) {
            const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)guess, (int)i);
            uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)bottom, (int)i);
            // a wrong guess: the segment again, from the true state (every lane follows the same uniform chase, lane 0 writes)
            if (g != truth) b = run_segment(i, truth, lane == 0);
            truth = b;
        }
        state_top = truth;
        __syncthreads();
        // the chunk's bytes, coalesced
        const uint32_t nbytes = n8 / 8u;
        for (uint32_t k = (uint32_t)lane; k < nbytes; k += 64) a.out[(j0 >> 3) + k] = obytes[k];
        __syncthreads();
        j = j0;
    }
}

__global__ void __launch_bounds__(64) one_chainback_kernel(OneChainbackArgs a) {
    extern __shared__ uint64_t one_rows_smem[];
    one_chainback_body(a, one_rows_smem);
}

// The FRAME route of the header-level drop-in (vit_hip_update_host_lazy): update() and the chainback() that will follow it in ONE
// launch -- a single decoder object's reset -> update -> chainback (examples/run_simple.cpp:76-80) then costs one kernel start
// instead of two launches, two synchronisations and four staging copies.  The symbols are read straight from pinned host-mapped
// memory (lane l fetches step l's, a 64-step group ahead of their use: the PCIe round trip hides behind the 4.6 us a group takes),
// the decision rows stay in HBM (the host fetches them only if somebody reads m_decisions), metrics, renormalisation sum and the
// decoded bytes are written to host-mapped memory, and the last thing the wavefront does is a system-scope release store of `seq`
// into the word the host is polling.
struct OneFrameArgs {
    OneUpdateArgs u;
    OneChainbackArgs c;
    int32_t do_chainback;
    uint32_t seq;
    uint32_t* done;              // host-mapped
};
template <int R, int SHIFT>
__global__ void __launch_bounds__(64) one_frame_kernel(OneFrameArgs a) {
    extern __shared__ uint64_t one_rows_smem[];
    one_update_body<R, SHIFT>(a.u);
    if (a.do_chainback) {
        // the rows this wavefront stored are read back by the same wavefront: stores complete and visible, no stale lines
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        __syncthreads();
        one_chainback_body(a.c, one_rows_smem);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");           // system scope: everything above is visible to the host ...
    __syncthreads();
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(a.done, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before this
}

template <int R, int SHIFT>
__global__ void __launch_bounds__(192) one_frame7_kernel(OneFrameArgs a) {
    extern __shared__ uint64_t one_rows_smem[];
    one_update7_body<R, SHIFT>(a.u);
    if (threadIdx.x >= 64) return;                          // the helper wavefronts are done (a wave that has ended leaves the barriers below)
    if (a.do_chainback) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // the helper's row stores: complete (its last barrier is behind them) and visible
        __syncthreads();
        one_chainback_body(a.c, one_rows_smem);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(a.done, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

constexpr size_t one_chainback_lds_bytes() { return ((size_t)ONE_CB_CHUNK + 64 + 8) * 8 + ONE_CB_CHUNK / 8 + 8; }
inline bool one_supported(int K, int R) { return K >= 2 && K <= 7 && R >= 1 && R <= 8; }

template <int SHIFT>
inline int one_launch_update(int R, const OneUpdateArgs& a, hipStream_t st) {
    if (one_update7_supported(a.K, R)) {
        switch (R) {
            case 1: hipLaunchKernelGGL((one_update7_kernel<1, SHIFT>), dim3(1), dim3(192), 0, st, a); break;
            case 2: hipLaunchKernelGGL((one_update7_kernel<2, SHIFT>), dim3(1), dim3(192), 0, st, a); break;
            case 3: hipLaunchKernelGGL((one_update7_kernel<3, SHIFT>), dim3(1), dim3(192), 0, st, a); break;
            default: hipLaunchKernelGGL((one_update7_kernel<4, SHIFT>), dim3(1), dim3(192), 0, st, a); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    switch (R) {
        case 1: hipLaunchKernelGGL((one_update_kernel<1, SHIFT>), dim3(1), dim3(64), 0, st, a); break;
        case 2: hipLaunchKernelGGL((one_update_kernel<2, SHIFT>), dim3(1), dim3(64), 0, st, a); break;
        case 3: hipLaunchKernelGGL((one_update_kernel<3, SHIFT>), dim3(1), dim3(64), 0, st, a); break;
        case 4: hipLaunchKernelGGL((one_update_kernel<4, SHIFT>), dim3(1), dim3(64), 0, st, a); break;
        case 5: hipLaunchKernelGGL((one_update_kernel<5, SHIFT>), dim3(1), dim3(64), 0, st, a); break;
        case 6: hipLaunchKernelGGL((one_update_kernel<6, SHIFT>), dim3(1), dim3(64), 0, st, a); break;
        case 7: hipLaunchKernelGGL((one_update_kernel<7, SHIFT>), dim3(1), dim3(64), 0, st, a); break;
        case 8: hipLaunchKernelGGL((one_update_kernel<8, SHIFT>), dim3(1), dim3(64), 0, st, a); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int SHIFT>
inline int one_launch_frame(int R, const OneFrameArgs& a, hipStream_t st) {
    const size_t smem = a.do_chainback ? one_chainback_lds_bytes() : 0;
#define VIT_ONE_FRAME_CASE(r)                                                                                                              \
    case r:                                                                                                                                \
        if (smem > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(one_frame_kernel<r, SHIFT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1; \
        hipLaunchKernelGGL((one_frame_kernel<r, SHIFT>), dim3(1), dim3(64), smem, st, a);                                                  \
        break;
#define VIT_ONE_FRAME7_CASE(r)                                                                                                             \
    case r:                                                                                                                                \
        if (smem > 32 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(one_frame7_kernel<r, SHIFT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1; \
        hipLaunchKernelGGL((one_frame7_kernel<r, SHIFT>), dim3(1), dim3(192), smem, st, a);                                                \
        break;
    if (one_update7_supported(a.u.K, R)) {
        switch (R) {
            VIT_ONE_FRAME7_CASE(1) VIT_ONE_FRAME7_CASE(2) VIT_ONE_FRAME7_CASE(3) VIT_ONE_FRAME7_CASE(4)
            default: return -1;
        }
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
#undef VIT_ONE_FRAME7_CASE
    switch (R) {
        VIT_ONE_FRAME_CASE(1) VIT_ONE_FRAME_CASE(2) VIT_ONE_FRAME_CASE(3) VIT_ONE_FRAME_CASE(4)
        VIT_ONE_FRAME_CASE(5) VIT_ONE_FRAME_CASE(6) VIT_ONE_FRAME_CASE(7) VIT_ONE_FRAME_CASE(8)
        default: return -1;
    }
#undef VIT_ONE_FRAME_CASE
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vit
