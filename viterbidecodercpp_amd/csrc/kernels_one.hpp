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
            asm("v_writelane_b32 %0, %1, %2" : "+v"(w_lo) : "s"((uint32_t)word), "n"(k));
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

template <int R, int SHIFT>
__global__ void __launch_bounds__(64) one_update_kernel(OneUpdateArgs a) { one_update_body<R, SHIFT>(a); }

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

constexpr size_t one_chainback_lds_bytes() { return ((size_t)ONE_CB_CHUNK + 64 + 8) * 8 + ONE_CB_CHUNK / 8 + 8; }
inline bool one_supported(int K, int R) { return K >= 2 && K <= 7 && R >= 1 && R <= 8; }

template <int SHIFT>
inline int one_launch_update(int R, const OneUpdateArgs& a, hipStream_t st) {
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
    switch (R) {
        VIT_ONE_FRAME_CASE(1) VIT_ONE_FRAME_CASE(2) VIT_ONE_FRAME_CASE(3) VIT_ONE_FRAME_CASE(4)
        VIT_ONE_FRAME_CASE(5) VIT_ONE_FRAME_CASE(6) VIT_ONE_FRAME_CASE(7) VIT_ONE_FRAME_CASE(8)
        default: return -1;
    }
#undef VIT_ONE_FRAME_CASE
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vit
