// kernels_lds2.hpp -- PLAN_LDS2: large constraint lengths (K = 11..15, any polynomials): one workgroup per frame PAIR,
// state metrics of both frames packed in one u32 per state and double-buffered in LDS, one thread per 8 butterflies.
//
// Device implementation of the reference's scalar strategy
//   ViterbiDecoder_Scalar::update / bfly / renormalise   include/viterbi/viterbi_decoder_scalar.h:29-153
//   ViterbiDecoder_Core::chainback                       include/viterbi/viterbi_decoder_core.h:214-236
// bit-identical to PLAN_LDS; this is the plan that makes K = 15 (Cassini, 16384 states) worth running on the GPU.
//
// Mapping (gfx950, wave64), N = 2^(K-1) states, H = N/2 butterflies, T = H/8 threads (1024 at K = 15):
//   * LDS: old[N] and new[N] as u32 = (frame A metric | frame B metric << 16), both biased by 0x8000 so that the
//     reference's unsigned compare is a SIGNED compare here (2 x 64 KiB at K = 15), plus a ring of per-step branch-metric
//     tables.  Natural state order: butterfly j reads old[j], old[j+H] (lane-consecutive dwords: conflict free) and writes
//     new[2j], new[2j+1] as ONE 8-byte store (lane-consecutive: conflict free).
//   * thread `tid` owns butterflies j = tid + T*k, k = 0..7, for the whole frame, so its 8 branch patterns are loop
//     invariant registers; per step it needs only {E[p], max_error - E[p]} for those patterns: one ds_read_b64 each from the
//     step's 64-entry table.
//   * the table of step t+NW is built during step t by wavefront (t mod NW) -- lane p computes entry p from the 2R symbols
//     of the two frames -- so every wavefront pays for one table per NW steps and the table is ready long before it is read.
//   * add-compare-select in packed 16-bit, exact for wrapping metrics: min = v_pk_min_i16, decision = sign of the signed
//     SATURATING difference (strict '>' of the reference: a tie keeps predecessor 0).
//   * decisions: sign bits of the thread's 16 new states are byte-gathered into one dword (frame A in bytes 0/2, frame B in
//     bytes 1/3) and stored ws[pair][step][tid]: 4 KiB contiguous per step.  vit_hip_export_decisions() converts to the
//     reference bit order; lds2_chainback_kernel walks this layout directly.
//   * one workgroup barrier per trellis step (metrics double buffer); renormalisation is a block-uniform rare branch:
//     packed min over registers -> wave __shfl_xor scan -> LDS exchange between wavefronts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"

namespace vit {

typedef uint32_t u32;
typedef unsigned short l2_u16x2 __attribute__((ext_vector_type(2)));
typedef short l2_s16x2 __attribute__((ext_vector_type(2)));

#define VIT_L2 __device__ __forceinline__
VIT_L2 u32 l2_add(u32 a, u32 b) { return __builtin_bit_cast(u32, (l2_u16x2)(__builtin_bit_cast(l2_u16x2, a) + __builtin_bit_cast(l2_u16x2, b))); }
VIT_L2 u32 l2_sub(u32 a, u32 b) { return __builtin_bit_cast(u32, (l2_u16x2)(__builtin_bit_cast(l2_u16x2, a) - __builtin_bit_cast(l2_u16x2, b))); }
VIT_L2 u32 l2_min_s(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(l2_s16x2, a), __builtin_bit_cast(l2_s16x2, b))); }
VIT_L2 u32 l2_max_s(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(l2_s16x2, a), __builtin_bit_cast(l2_s16x2, b))); }
VIT_L2 u32 l2_sub_sat_s(u32 a, u32 b) {
    u32 d;
    asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

struct Lds2UpdateArgs {
    const uint8_t* symbols;          // [F][n_steps][R] soft_t
    size_t sym_frame_stride_bytes;
    u32* ws;                         // [pairs][rows][T] decision dwords
    size_t ws_pair_stride;           // rows * T
    void* metrics_out;               // [F][N] error_t or null
    uint64_t* renorm_sum;            // [F] or null
    const u32* start_state;          // [F] or null
    const uint16_t* pattern;         // [H] bit i = (branch_table[i][j] == high)
    u32 frames, n_steps;
    int32_t R;
    DevConfig cfg;
};

// position of the decision bit of the thread's new-state register r = 2k+b (butterfly k, input bit b), frame half h
__host__ __device__ constexpr u32 lds2_dec_bit(u32 r, u32 h) { return (r & 7u) + 8u * h + 16u * ((r >> 3) & 1u); }

template <int K>
struct Lds2Geom {
    static constexpr int N = 1 << (K - 1), H = N / 2, BPT = 8, T = H / BPT, NW = T / 64;
    static constexpr int RING = 2 * NW;                // branch-metric tables in flight
    static constexpr size_t smem_bytes = (size_t)2 * N * 4 + (size_t)RING * 64 * 8 + (size_t)NW * 4 + 16;
    static_assert(T >= 64 && T <= 1024, "PLAN_LDS2 serves K = 11..15");
};

template <int K, int SHIFT>
__global__ void __launch_bounds__(Lds2Geom<K>::T) lds2_update_kernel(Lds2UpdateArgs a) {
    using GM = Lds2Geom<K>;
    constexpr int N = GM::N, H = GM::H, BPT = GM::BPT, T = GM::T, NW = GM::NW, RING = GM::RING;
    constexpr u32 BIAS2 = 0x80008000u;
    extern __shared__ __attribute__((aligned(16))) u32 lds2_smem[];
    u32* met_old = lds2_smem;
    u32* met_new = lds2_smem + N;
    uint2* etab = (uint2*)(lds2_smem + 2 * N);             // [RING][64] {E, max_error - E}
    u32* wmin = (u32*)(etab + RING * 64);             // [NW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 pair = blockIdx.x;
    const u32 fA = 2 * pair;
    const bool validB = fA + 1 < a.frames;
    const u32 fB = validB ? fA + 1 : fA;
    const int R = a.R;
    constexpr int SBY = SHIFT ? 1 : 2;

    const u32 HIGH2 = (u32)(uint16_t)a.cfg.high * 0x10001u, LOW2 = (u32)(uint16_t)a.cfg.low * 0x10001u;
    const u32 MAXE2 = (u32)a.cfg.max_error * 0x10001u;
    const u32 THRM1B2 = ((u32)(uint16_t)(a.cfg.threshold - 1) * 0x10001u) ^ BIAS2;   // metric >= thr  <=>  biased > this
    const u32 FORCE = a.cfg.threshold == 0 ? BIAS2 : 0u;

    // ---- loop-invariant per-thread constants: branch pattern of each of my 8 butterflies (table row offset) ----
    u32 prow[BPT];
#pragma unroll
    for (int k = 0; k < BPT; ++k) prow[k] = (u32)a.pattern[tid + T * k] & 63u;

    // ---- reset (viterbi_decoder_core.h:202-211) ----
    {
        const u32 sA = a.start_state ? (a.start_state[fA] & (u32)(N - 1)) : 0u;
        const u32 sB = a.start_state ? (a.start_state[fB] & (u32)(N - 1)) : 0u;
        for (int s = tid; s < N; s += T) {
            const u32 lo = ((u32)s == sA) ? a.cfg.init_start : a.cfg.init_non_start;
            const u32 hi = ((u32)s == sB) ? a.cfg.init_start : a.cfg.init_non_start;
            met_old[s] = (lo | (hi << 16)) ^ BIAS2;
        }
    }

    // ---- branch-metric table builder: lane p makes entry p of the table of one step ----
    const uint8_t* symA = a.symbols + (size_t)fA * a.sym_frame_stride_bytes;
    const uint8_t* symB = a.symbols + (size_t)fB * a.sym_frame_stride_bytes;
    auto load_syms = [&](u32 step, u32 (&y)[8]) __attribute__((always_inline)) {
        // packed (frame A | frame B << 16) symbols of `step` in the device's 16-bit domain
#pragma unroll
        for (int i = 0; i < 8; ++i) {
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
    auto build_table = [&](u32 step, const u32 (&y)[8]) __attribute__((always_inline)) {
        // E[p] = sum_i |bt_i - y_i| with bt_i = high where bit i of p is set  (scalar.h:66-73); EB = max_error - E (:107)
        u32 e = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < R) {
                const u32 bt = ((lane >> i) & 1) ? HIGH2 : LOW2;
                const u32 d = l2_sub(bt, y[i]);
                e = l2_add(e, l2_max_s(d, l2_sub(0u, d)));   // error_t(get_abs(soft_t(expected - sym)))  (scalar.h:68-71)
            }
        }
        etab[(step % RING) * 64 + lane] = make_uint2(e, l2_sub(MAXE2, e));
    };
    u32 ysym[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // prologue: wave w builds the tables of steps w, and prefetches the symbols of step w + NW (its first in-loop build)
    if ((u32)wave < a.n_steps) {
        load_syms((u32)wave, ysym);
        build_table((u32)wave, ysym);
    }
    if ((u32)wave + NW < a.n_steps) load_syms((u32)wave + NW, ysym);
    __syncthreads();

    uint64_t rsA = 0, rsB = 0;
    u32* ws_pair = a.ws + (size_t)pair * a.ws_pair_stride;

    for (u32 t = 0; t < a.n_steps; ++t) {
        // ---- my turn to build: table of step t + NW (symbols were loaded NW steps ago), then prefetch t + 2 NW ----
        if ((int)(t % NW) == wave) {
            if (t + NW < a.n_steps) build_table(t + NW, ysym);
            if (t + 2 * NW < a.n_steps) load_syms(t + 2 * NW, ysym);
        }
        const uint2* tab = etab + (t % RING) * 64;
        // ---- add-compare-select for my 8 butterflies  (scalar.h:113-134) ----
        u32 nm[2 * BPT], D[2 * BPT];
#pragma unroll
        for (int k = 0; k < BPT; ++k) {
            const int j = tid + T * k;
            const u32 ma = met_old[j], mb = met_old[j + H];
            const uint2 ee = tab[prow[k]];
            const u32 x0 = l2_add(ma, ee.x), y0 = l2_add(mb, ee.y);   // -> next state 2j
            const u32 x1 = l2_add(ma, ee.y), y1 = l2_add(mb, ee.x);   // -> next state 2j+1
            nm[2 * k] = l2_min_s(x0, y0);
            nm[2 * k + 1] = l2_min_s(x1, y1);
            D[2 * k] = l2_sub_sat_s(y0, x0);       // sign set <=> x0 > y0 (strict: tie keeps predecessor 0)
            D[2 * k + 1] = l2_sub_sat_s(y1, x1);
            *(uint2*)(met_new + 2 * j) = make_uint2(nm[2 * k], nm[2 * k + 1]);
        }
        // ---- gather the 2 x 16 sign bits: bytes 1 and 3 of the register pair (r, r+8) ----
        {
            constexpr u32 SIGNS = 0x80808080u, HI_BYTES = 0x07050301u;
            u32 lo4 = 0, hi4 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo4 = (__builtin_amdgcn_perm(D[k + 8], D[k], HI_BYTES) & SIGNS) | (lo4 >> 1);
                hi4 = (__builtin_amdgcn_perm(D[k + 12], D[k + 4], HI_BYTES) & SIGNS) | (hi4 >> 1);
            }
            ws_pair[(size_t)t * T + tid] = (lo4 >> 4) | hi4;
        }
        __syncthreads();
        // ---- renormalise when new_metric[0] >= threshold  (scalar.h:48-50, :139-153): block-uniform ----
        const u32 need = (l2_sub_sat_s(THRM1B2, met_new[0]) | FORCE) & BIAS2;   // sign bits: frame A / frame B
        if (need != 0) {
            const u32 msk = ((need & 0x8000u) ? 0x0000FFFFu : 0u) | ((need & 0x80000000u) ? 0xFFFF0000u : 0u);
            u32 mn = nm[0];
#pragma unroll
            for (int i = 1; i < 2 * BPT; ++i) mn = l2_min_s(mn, nm[i]);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) mn = l2_min_s(mn, (u32)__shfl_xor((int)mn, off));
            if (lane == 0) wmin[wave] = mn;
            __syncthreads();
            for (int w = 0; w < NW; ++w) mn = l2_min_s(mn, wmin[w]);
            const u32 sub = (mn ^ BIAS2) & msk;   // true (unbiased) minimum of each frame that renormalises
#pragma unroll
            for (int k = 0; k < BPT; ++k) {
                const int j = tid + T * k;
                *(uint2*)(met_new + 2 * j) = make_uint2(l2_sub(nm[2 * k], sub), l2_sub(nm[2 * k + 1], sub));
            }
            rsA += (uint64_t)((sub & 0xFFFFu) >> SHIFT);
            rsB += (uint64_t)((sub >> 16) >> SHIFT);
            __syncthreads();
        }
        u32* tmp = met_old;   // m_metrics.swap()  (scalar.h:51)
        met_old = met_new;
        met_new = tmp;
    }

    if (a.metrics_out) {
        for (int s = tid; s < N; s += T) {
            const u32 v = met_old[s] ^ BIAS2;
            if (SHIFT) {
                ((uint8_t*)a.metrics_out)[(size_t)fA * N + s] = (uint8_t)((v & 0xFFFFu) >> 8);
                if (validB) ((uint8_t*)a.metrics_out)[(size_t)fB * N + s] = (uint8_t)(v >> 24);
            } else {
                ((uint16_t*)a.metrics_out)[(size_t)fA * N + s] = (uint16_t)(v & 0xFFFFu);
                if (validB) ((uint16_t*)a.metrics_out)[(size_t)fB * N + s] = (uint16_t)(v >> 16);
            }
        }
    }
    if (a.renorm_sum && tid == 0) {
        a.renorm_sum[fA] = rsA;
        if (validB) a.renorm_sum[fB] = rsB;
    }
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

// one lane per frame; the decision dword of (step t, state s): ws[pair][t][(s>>1) % T], bit of register 2*((s>>1)/T) + (s&1)
__global__ void lds2_chainback_kernel(Lds2ChainbackArgs a) {
    const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.frames) return;
    const int TSB = a.K - 1;
    const u32 T = (1u << (TSB - 1)) / 8u;
    const u32 tshift = (u32)(TSB - 1 - 3);           // log2(T)
    const int ignore = TSB < 8 ? TSB : 8;           // ViterbiTracebackBuffer::get_layout (core.h:129-149)
    const int shift_state = 8 - ignore, shift_tail = TSB - ignore, total_bits = TSB + shift_state;
    const u32* ws = a.ws + (f >> 1) * a.ws_pair_stride;
    const u32 half = (u32)(f & 1);
    uint8_t* out = a.out + f * (((size_t)a.L + 7) / 8);
    u32 reg = (a.end_state ? (a.end_state[f] & ((1u << TSB) - 1u)) : 0u) << shift_state;
    for (size_t j = a.L; j-- > 0;) {
        const u32 state = reg >> shift_state;
        const u32 jj = state >> 1;
        const u32 r = 2u * (jj >> tshift) + (state & 1u);
        const u32 w = ws[(j + (size_t)TSB) * T + (jj & (T - 1u))];
        const u32 bit = (w >> lds2_dec_bit(r, half)) & 1u;
        reg = (reg >> 1) | (bit << (total_bits - 1));
        if ((j & 7) == 0) out[j >> 3] = (uint8_t)((reg >> shift_tail) & 0xFFu);
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
    const u32 T = (1u << (TSB - 1)) / 8u;
    const u32 tshift = (u32)(TSB - 1 - 3);
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (frame, step, word)
    const size_t total = (size_t)a.frames * a.n_steps * W;
    if (idx >= total) return;
    const u32 w = (u32)(idx % W);
    const u32 t = (u32)((idx / W) % a.n_steps);
    const u32 f = (u32)(idx / ((size_t)W * a.n_steps));
    const u32* ws = a.ws + (size_t)(f >> 1) * a.ws_pair_stride + (size_t)t * T;
    const u32 half = f & 1u;
    uint64_t word = 0;
    for (u32 b = 0; b < 64; ++b) {
        const u32 s = w * 64 + b;
        const u32 jj = s >> 1;
        const u32 r = 2u * (jj >> tshift) + (s & 1u);
        word |= (uint64_t)((ws[jj & (T - 1u)] >> lds2_dec_bit(r, half)) & 1u) << b;
    }
    a.out[idx] = word;
}

// ---- host side ----------------------------------------------------------------------------------------------------
inline bool lds2_supported(int K, int R) { return K >= 11 && K <= 15 && R >= 1 && R <= 6; }   // 64-entry branch-metric table
inline size_t lds2_threads(int K) { return ((size_t)1 << (K - 2)) / 8; }
inline size_t lds2_workspace_bytes(int K, size_t frames, size_t L) {
    return ((frames + 1) / 2) * (L + (size_t)K - 1) * lds2_threads(K) * 4;
}

template <int K, int SHIFT>
int lds2_launch_update(const Lds2UpdateArgs& a, unsigned pairs, hipStream_t st) {
    using GM = Lds2Geom<K>;
    auto kern = lds2_update_kernel<K, SHIFT>;
    if (GM::smem_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)GM::smem_bytes) != hipSuccess)
            return -1;
    }
    hipLaunchKernelGGL(kern, dim3(pairs), dim3(GM::T), GM::smem_bytes, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int SHIFT>
int lds2_launch_update_k(int K, const Lds2UpdateArgs& a, unsigned pairs, hipStream_t st) {
    switch (K) {
        case 11: return lds2_launch_update<11, SHIFT>(a, pairs, st);
        case 12: return lds2_launch_update<12, SHIFT>(a, pairs, st);
        case 13: return lds2_launch_update<13, SHIFT>(a, pairs, st);
        case 14: return lds2_launch_update<14, SHIFT>(a, pairs, st);
        case 15: return lds2_launch_update<15, SHIFT>(a, pairs, st);
        default: return -1;
    }
}

inline int lds2_update(int K, int R, const DevConfig& cfg, int shift, const uint16_t* d_pattern, const void* d_symbols,
                       size_t frames, size_t n_steps, size_t L, void* d_ws, void* d_metrics, uint64_t* d_renorm,
                       const uint32_t* d_start, hipStream_t st) {
    if (frames == 0 || n_steps == 0) return 0;
    Lds2UpdateArgs a{};
    a.symbols = (const uint8_t*)d_symbols;
    a.sym_frame_stride_bytes = n_steps * (size_t)R * (shift ? 1 : 2);
    a.ws = (u32*)d_ws;
    a.ws_pair_stride = (L + (size_t)K - 1) * lds2_threads(K);
    a.metrics_out = d_metrics;
    a.renorm_sum = d_renorm;
    a.start_state = d_start;
    a.pattern = d_pattern;
    a.frames = (u32)frames;
    a.n_steps = (u32)n_steps;
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
    a.ws_pair_stride = (L + (size_t)K - 1) * lds2_threads(K);
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
    a.ws_pair_stride = (L + (size_t)K - 1) * lds2_threads(K);
    a.out = d_out;
    a.frames = (u32)frames;
    a.n_steps = (u32)n_steps;
    a.K = K;
    const size_t total = frames * n_steps * ((size_t)1 << (K - 7));
    hipLaunchKernelGGL(lds2_export_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vit
