// kernels_lds.hpp -- PLAN_LDS: state metrics staged in LDS, one wavefront per contiguous slab of next-states.
//
// The general-purpose device implementation of the reference's scalar strategy
//   ViterbiDecoder_Scalar::update / bfly / renormalise   include/viterbi/viterbi_decoder_scalar.h:29-153
//   ViterbiDecoder_Core::chainback                       include/viterbi/viterbi_decoder_core.h:214-236
// for any K <= 16 and R <= 8, including incremental (streaming) update calls.
//
// Mapping (gfx950, wave64):
//   * one workgroup per frame, `waves` wavefronts; wave w owns next-states [w*N/waves, (w+1)*N/waves) and walks them in
//     chunks of 64 with lane == next-state, so `__ballot(decision)` IS the reference's 64-bit decision word
//     (bit s%64 of word s/64) -- no bit shuffling;
//   * both metric buffers (old/new) live in LDS as u16; u8 error metrics are carried in the high byte of the u16
//     (value << 8) so that 16-bit wrapping add/sub/compare reproduce the reference's mod-256 arithmetic exactly;
//   * the R symbols of a step are wave-uniform: a 64-dword window of the frame's symbol stream is held one dword per lane
//     and indexed with v_readlane, so |high-y|, |low-y| are computed once per step on the scalar unit;
//   * decision words are parked one per lane (lane n%64 keeps the wave's n-th word) and flushed as one coalesced
//     512-byte store every 64 words;
//   * renormalisation (rare, wave-uniform branch on new[0] >= threshold) is a __shfl_xor min-scan per wave plus an
//     LDS exchange between waves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"

namespace vit {

struct LdsUpdateArgs {
    const uint8_t* symbols;          // device, raw bytes of [F][n_steps][R] soft_t
    size_t sym_frame_stride_bytes;
    size_t sym_total_bytes;
    uint64_t* decisions;             // [F][rows][W]
    size_t dec_frame_stride_words;
    uint32_t dec_row0;               // first row written by this call
    void* metrics_io;                // [F][N] error_t: read when !reset, written always when non-null
    uint64_t* renorm_sum;            // [F] or null
    const uint32_t* start_state;     // [F] or null
    const uint16_t* pattern;         // [H] bit i = (branch_table[i][j] == high)
    int32_t K, n_steps, reset;
    DevConfig cfg;
};

__device__ __forceinline__ uint16_t abs_soft(int32_t expected, int32_t y) {
    // const soft_t error = expected - sym; error_t(get_abs(error))   (viterbi_decoder_scalar.h:68-71, :155-159)
    const int16_t d = (int16_t)(expected - y);
    const int16_t n = (int16_t)(-(int32_t)d);
    return (uint16_t)(d > 0 ? d : n);
}

template <int R, int SHIFT>
__global__ void lds_update_kernel(LdsUpdateArgs a) {
    extern __shared__ uint16_t smem[];
    constexpr int SB = SHIFT ? 1 : 2;  // sizeof(soft_t)
    const int N = 1 << (a.K - 1);
    const int H = N >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const size_t f = blockIdx.x;

    // K = 16: two 64 KiB metric buffers leave no room for the 32 KiB pattern table: it is then read from global memory (L2)
    const bool pat_in_lds = N <= 16384;
    uint16_t* met_old = smem;
    uint16_t* met_new = smem + N;
    uint16_t* pat_lds = smem + 2 * N;             // [max(H,1)] when pat_in_lds
    uint16_t* wmin = pat_in_lds ? pat_lds + (H > 0 ? H : 1) : pat_lds;   // [nwaves]
    const uint16_t* pat = pat_in_lds ? pat_lds : a.pattern;

    const int npw = (N >= 64 * nwaves) ? N / nwaves : N;  // next-states per wave
    const int ch = npw >= 64 ? npw / 64 : 1;              // 64-state chunks per wave per step
    const int slab0 = (N >= 64 * nwaves) ? wave * npw : 0;
    const int W = N >= 64 ? N / 64 : 1;
    int chs = 0;
    while ((1 << chs) < ch) chs++;

    // ---- prime LDS: branch patterns + starting metrics (ViterbiDecoder_Core::reset, core.h:202-211) ----
    if (pat_in_lds)
        for (int j = tid; j < (H > 0 ? H : 1); j += blockDim.x) pat_lds[j] = a.pattern[j];
    if (a.reset) {
        const uint32_t start = a.start_state ? (a.start_state[f] & (uint32_t)(N - 1)) : 0u;
        for (int s = tid; s < N; s += blockDim.x) met_old[s] = (s == (int)start) ? a.cfg.init_start : a.cfg.init_non_start;
    } else {
        for (int s = tid; s < N; s += blockDim.x) {
            if (SHIFT) met_old[s] = (uint16_t)(((const uint8_t*)a.metrics_io)[f * N + s]) << 8;
            else met_old[s] = ((const uint16_t*)a.metrics_io)[f * N + s];
        }
    }
    __syncthreads();

    // ---- symbol window: 2 x 64 dwords of this frame's stream, one dword per lane ----
    // The window is aligned on the ABSOLUTE address (the caller's pointer need only be soft_t aligned), and a dword that
    // straddles the end of the buffer is assembled from single bytes: nothing outside [symbols, symbols + total) is read.
    const size_t frame_b = f * a.sym_frame_stride_bytes;
    const size_t base_mis = (size_t)((uintptr_t)a.symbols & 3u);
    size_t wbase = (frame_b + base_mis) & ~(size_t)3;          // offset from the dword-aligned address below `symbols`
    int ob = (int)((frame_b + base_mis) & 3);   // byte offset of the current step inside the window
    const uint8_t* const sym_al = a.symbols - base_mis;         // dword aligned; bytes [0, base_mis) are not ours
    const size_t end_al = a.sym_total_bytes + base_mis;
    auto load_win = [&](size_t base) -> uint32_t {
        const size_t addr = base + 4 * (size_t)lane;
        if (addr >= base_mis && addr + 4 <= end_al) return *(const uint32_t*)(sym_al + addr);
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (addr + k >= base_mis && addr + k < end_al) v |= (uint32_t)sym_al[addr + k] << (8 * k);
        return v;
    };
    uint32_t win0 = load_win(wbase), win1 = load_win(wbase + 256);

    uint64_t acc = 0;       // sum of subtracted minima (update()'s return value)
    uint32_t cnt = 0;       // decision words produced by this wave so far
    uint32_t flushed = 0;   // ... of which already stored
    uint64_t myword = 0;
    uint64_t* dec_frame = a.decisions + f * a.dec_frame_stride_words;

    auto flush = [&](uint32_t upto) {
        const uint32_t n = flushed + (uint32_t)lane;
        if (n < upto) {
            const uint32_t t_n = n >> chs, q_n = n & (uint32_t)(ch - 1);
            const size_t row = (size_t)a.dec_row0 + t_n;
            const size_t widx = (N >= 64 * nwaves) ? (size_t)wave * ch + q_n : 0;
            dec_frame[row * W + widx] = myword;
        }
    };

    const bool wave_has_work = (N >= 64 * nwaves) || wave == 0;

    for (int t = 0; t < a.n_steps; ++t) {
        // |high - y_i|, |low - y_i| for the R symbols of this step (wave-uniform -> scalar unit)
        uint16_t A1[R], A0[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int o = ob + i * SB;
            const uint32_t dw = (o < 256) ? __builtin_amdgcn_readlane(win0, o >> 2)
                                          : __builtin_amdgcn_readlane(win1, (o - 256) >> 2);
            const uint32_t v = dw >> ((o & 3) * 8);
            const int32_t y = SHIFT ? (int32_t)(int16_t)(uint16_t)((v & 0xFFu) << 8) : (int32_t)(int16_t)(uint16_t)(v & 0xFFFFu);
            A1[i] = abs_soft(a.cfg.high, y);
            A0[i] = abs_soft(a.cfg.low, y);
        }
        ob += R * SB;
        if (ob >= 256) {
            ob -= 256;
            wbase += 256;
            win0 = win1;
            win1 = load_win(wbase + 256);
        }

        if (wave_has_work) {
            for (int q = 0; q < ch; ++q) {
                const int s_raw = slab0 + q * 64 + lane;
                const bool active = s_raw < N;
                const int s = s_raw & (N - 1);
                const int j = (s >> 1) & (H > 0 ? H - 1 : 0);
                const uint32_t p = pat[j];
                uint16_t e = 0;  // error_t total_error  (scalar.h:66-73)
#pragma unroll
                for (int i = 0; i < R; ++i) e = (uint16_t)(e + (((p >> i) & 1u) ? A1[i] : A0[i]));
                const uint16_t ebar = (uint16_t)(a.cfg.max_error - e);        // scalar.h:107
                const bool b = s & 1;                                         // input bit of this next-state
                const uint16_t m0 = (uint16_t)(met_old[j] + (b ? ebar : e));      // via predecessor (0|X)  :113,:115
                const uint16_t m1 = (uint16_t)(met_old[j + H] + (b ? e : ebar));  // via predecessor (1|X)  :114,:116
                const bool d = m0 > m1;                                       // strict: tie -> 0        :123-124
                if (active) met_new[s] = d ? m1 : m0;                         // :127-128
                const uint64_t mask = __ballot(active && d);                  // :131-134
                if ((uint32_t)lane == (cnt & 63u)) myword = mask;
                cnt++;
                if ((cnt & 63u) == 0) {
                    flush(cnt);
                    flushed = cnt;
                }
            }
        }
        __syncthreads();

        if (__builtin_expect(met_new[0] >= a.cfg.threshold, 0)) {  // scalar.h:48 -- state 0 only; block-uniform; rare: out of line
            uint16_t mn = 0xFFFF;             // renormalise, scalar.h:139-153
            if (wave_has_work)
                for (int q = 0; q < ch; ++q) {
                    const int s_raw = slab0 + q * 64 + lane;
                    if (s_raw < N) {
                        const uint16_t v = met_new[s_raw];
                        mn = v < mn ? v : mn;
                    }
                }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const uint16_t o = (uint16_t)__shfl_xor((int)mn, off);
                mn = o < mn ? o : mn;
            }
            if (nwaves > 1) {
                if (lane == 0) wmin[wave] = mn;
                __syncthreads();
                for (int w = 0; w < nwaves; ++w) {
                    const uint16_t o = wmin[w];
                    mn = o < mn ? o : mn;
                }
            }
            if (wave_has_work)
                for (int q = 0; q < ch; ++q) {
                    const int s_raw = slab0 + q * 64 + lane;
                    if (s_raw < N) met_new[s_raw] = (uint16_t)(met_new[s_raw] - mn);
                }
            acc += (uint64_t)(mn >> SHIFT);
            __syncthreads();
        }
        uint16_t* tmp = met_old;  // m_metrics.swap()  (scalar.h:51)
        met_old = met_new;
        met_new = tmp;
    }

    if (wave_has_work) flush(cnt);

    if (a.metrics_io) {
        for (int s = tid; s < N; s += blockDim.x) {
            if (SHIFT) ((uint8_t*)a.metrics_io)[f * N + s] = (uint8_t)(met_old[s] >> 8);
            else ((uint16_t*)a.metrics_io)[f * N + s] = met_old[s];
        }
    }
    if (a.renorm_sum && tid == 0) a.renorm_sum[f] = acc;
}

// ---- chainback on the reference layout [F][S][W] ----------------------------------------------------------------
struct LdsChainbackArgs {
    const uint64_t* decisions;
    size_t dec_frame_stride_words;
    uint8_t* out;              // [F][ceil(L/8)]
    const uint32_t* end_state; // [F] or null
    size_t frames, L;
    int32_t K;
};

// One lane per frame.  The traceback shift register of ViterbiTracebackBuffer (core.h:87-153) is modelled exactly:
// K-1 state bits above `shift_state` padding bits, decision bit pushed in at the top, top 8 bits = output byte.
__global__ void lds_chainback_kernel(LdsChainbackArgs a) {
    const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.frames) return;
    const int TSB = a.K - 1;
    const int W = TSB >= 6 ? 1 << (TSB - 6) : 1;
    const int ignore = TSB < 8 ? TSB : 8;
    const int shift_state = 8 - ignore, shift_tail = TSB - ignore, total_bits = TSB + shift_state;
    const uint64_t* dec = a.decisions + f * a.dec_frame_stride_words;
    uint8_t* out = a.out + f * ((a.L + 7) / 8);
    uint32_t reg = (a.end_state ? (a.end_state[f] & ((1u << TSB) - 1u)) : 0u) << shift_state;

    size_t j = a.L;
    if (W == 1) {
        // the word address does not depend on the state: fetch 8 rows ahead of the dependent bit-chase
        while (j > 0) {
            const int n = j >= 8 ? 8 : (int)j;
            uint64_t w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = (k < n) ? dec[(j - 1 - k) + TSB] : 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < n) {
                    const size_t jj = j - 1 - k;
                    const uint32_t state = reg >> shift_state;
                    const uint32_t bit = (uint32_t)(w[k] >> state) & 1u;
                    reg = (reg >> 1) | (bit << (total_bits - 1));
                    if ((jj & 7) == 0) out[jj >> 3] = (uint8_t)((reg >> shift_tail) & 0xFFu);
                }
            }
            j -= n;
        }
    } else {
        while (j > 0) {
            const size_t jj = j - 1;
            const uint32_t state = reg >> shift_state;
            const uint64_t w = dec[(jj + TSB) * W + (state >> 6)];
            const uint32_t bit = (uint32_t)(w >> (state & 63u)) & 1u;
            reg = (reg >> 1) | (bit << (total_bits - 1));
            if ((jj & 7) == 0) out[jj >> 3] = (uint8_t)((reg >> shift_tail) & 0xFFu);
            j = jj;
        }
    }
}

}  // namespace vit
