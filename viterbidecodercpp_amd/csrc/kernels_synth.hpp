// kernels_synth.hpp -- synthetic test frames generated in HBM, and the bit-error counter of the BER harness.
//
// What the reference's harness does on the host for one block at a time, for a whole batch in one kernel:
//   random info bytes                                       examples/run_snr_ber.cpp:338-341
//   convolutional encoder, shift-register form              include/viterbi/convolutional_encoder_shift_register.h:42-62
//   K-1 zero tail bits, symbols step-major/polynomial-minor examples/helpers/test_helpers.h:17-64 (encode_data)
//   +-1.0 BPSK + N(0, sigma^2)                              examples/run_snr_ber.cpp:342-350
//   quantiser round / clamp to [low, high]                  examples/run_snr_ber.cpp:352-359
//   bit errors = popcount(tx ^ rx)                          examples/helpers/test_helpers.h:95-104 (get_total_bit_errors)
// This is data synthesis for tests and measurements, not part of update()/chainback().
//
// Reproducible by construction: every random number is a pure function of (seed, global frame index, position) through the
// counter-based generator Philox4x32-10 (Salmon et al., SC'11), so the batch does not depend on how it is split over
// launches, ranks or threads, and the host mirror in viterbidecodercpp_amd/synth.py regenerates it:
//   info byte i of frame f    = byte (i & 15) (little endian) of Philox(ctr = {i >> 4, f_lo, 0, f_hi}, key = seed)
//   normal deviate n of frame f (n = step*R + polynomial): block = Philox(ctr = {n >> 2, f_lo, 1, f_hi}, key = seed),
//       u1 = ((x >> 8) + 0.5) * 2^-24 in (0,1), u2 = (y >> 8) * 2^-24 in [0,1), r = sqrt(-2 ln u1),
//       pair (x0, x1) -> z0 = r cos(2 pi u2), z1 = r sin(2 pi u2); pair (x2, x3) -> z2, z3; n & 3 picks.
// One thread makes 8 trellis steps (one info byte) of one frame: 8*R symbols = 32 contiguous bytes at R = 2 / int16, so a
// wavefront writes 2 KiB contiguous.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vit {

struct SynthArgs {
    uint8_t* tx;            // [F][L/8] or null
    void* symbols;          // [F][S][R] soft_t
    uint64_t seed, first_frame;
    uint32_t frames, L, S, K, R;
    uint32_t G[8];
    int32_t high, low;
    float sigma;            // sqrt(noise_variance)
    float scale;            // (high - low)/2 / sqrt(1 + noise_variance)
    float mean;             // (high + low)/2
    int32_t noise_free;
};

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename soft_t, int R>
__global__ void __launch_bounds__(256) synth_kernel(SynthArgs a) {
    const uint32_t nchunks = (a.S + 7u) / 8u;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)a.frames * nchunks) return;
    const uint32_t f = (uint32_t)(gid / nchunks), b = (uint32_t)(gid % nchunks);
    const uint64_t frame = a.first_frame + f;
    const uint32_t nbytes = a.L / 8u;

    // info bytes b-2, b-1, b (zero before the frame and in the tail): 24 input bits, newest = bit 0
    uint32_t word = 0;
    {
        uint32_t x[4] = {0u, 0u, 0u, 0u};
        uint32_t have_blk = 0xFFFFFFFFu;   // bytes of one 16-byte Philox block share a call
#pragma unroll
        for (int d = 2; d >= 0; --d) {
            uint32_t v = 0;
            if (b >= (uint32_t)d && b - (uint32_t)d < nbytes) {
                const uint32_t i = b - (uint32_t)d;
                if ((i >> 4) != have_blk) {
                    have_blk = i >> 4;
                    philox4x32_10(have_blk, (uint32_t)frame, 0u, (uint32_t)(frame >> 32), (uint32_t)a.seed, (uint32_t)(a.seed >> 32), x);
                }
                const uint32_t w = (i >> 2) & 3u;
                v = ((w == 0 ? x[0] : w == 1 ? x[1] : w == 2 ? x[2] : x[3]) >> (8u * (i & 3u))) & 0xFFu;
            }
            word = (word << 8) | v;
        }
    }
    if (a.tx && b < nbytes) a.tx[(size_t)f * nbytes + b] = (uint8_t)(word & 0xFFu);

    const uint32_t kmask = (1u << a.K) - 1u;
    const uint32_t t0 = 8u * b;
    const uint32_t nsteps = a.S - t0 < 8u ? a.S - t0 : 8u;
    const uint32_t n0 = t0 * (uint32_t)R;      // first deviate index of this thread: a multiple of 8
    constexpr int NV = 8 * R;
    soft_t val[NV];
    float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const uint32_t s = (uint32_t)(k / R), i = (uint32_t)(k % R);
        const uint32_t reg = (word >> (7u - s)) & kmask;       // bit j = input bit of step t - j  (shift_register.h:47-58)
        const uint32_t bit = (uint32_t)__builtin_popcount(reg & a.G[i]) & 1u;
        int32_t q;
        if (a.noise_free) {
            q = bit ? a.high : a.low;
        } else {
            if ((k & 3) == 0) {
                uint32_t x[4];
                philox4x32_10((n0 + (uint32_t)k) >> 2, (uint32_t)frame, 1u, (uint32_t)(frame >> 32), (uint32_t)a.seed,
                              (uint32_t)(a.seed >> 32), x);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float u1 = __fmul_rn((float)(x[2 * h] >> 8) + 0.5f, 5.9604644775390625e-08f);
                    const float u2 = __fmul_rn((float)(x[2 * h + 1] >> 8), 5.9604644775390625e-08f);
                    const float r = sqrtf(__fmul_rn(-2.0f, logf(u1)));
                    float sn, cs;
                    sincospif(__fmul_rn(2.0f, u2), &sn, &cs);
                    z[2 * h] = __fmul_rn(r, cs);
                    z[2 * h + 1] = __fmul_rn(r, sn);
                }
            }
            // run_snr_ber.cpp:348-358 (no fused multiply-add: the host mirror rounds after every operation)
            const float noisy = __fadd_rn(bit ? 1.0f : -1.0f, __fmul_rn(a.sigma, z[k & 3]));
            const float v = __fadd_rn(__fmul_rn(noisy, a.scale), a.mean);
            const float rv = roundf(v);
            q = rv > (float)a.high ? a.high : rv < (float)a.low ? a.low : (int32_t)rv;
        }
        val[k] = (soft_t)q;
    }
    soft_t* dst = (soft_t*)a.symbols + ((size_t)f * a.S + t0) * R;
    constexpr int BYTES = NV * (int)sizeof(soft_t);
    if (nsteps == 8u && ((uintptr_t)dst & 3u) == 0) {
        uint32_t pk[BYTES / 4];
        __builtin_memcpy(pk, val, BYTES);
        if (BYTES % 16 == 0 && ((uintptr_t)dst & 15u) == 0) {
#pragma unroll
            for (int w = 0; w < BYTES / 16; ++w) ((uint4*)dst)[w] = make_uint4(pk[4 * w], pk[4 * w + 1], pk[4 * w + 2], pk[4 * w + 3]);
        } else if (((uintptr_t)dst & 7u) == 0) {
#pragma unroll
            for (int w = 0; w < BYTES / 8; ++w) ((uint2*)dst)[w] = make_uint2(pk[2 * w], pk[2 * w + 1]);
        } else {
#pragma unroll
            for (int w = 0; w < BYTES / 4; ++w) ((uint32_t*)dst)[w] = pk[w];
        }
    } else {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if ((uint32_t)(k / R) < nsteps) dst[k] = val[k];
    }
}

struct BitErrArgs {
    const uint8_t* a;
    const uint8_t* b;
    size_t n_bytes;
    unsigned long long* count;
};

// popcount(a ^ b) summed into *count (test_helpers.h:95-104); 16 bytes per lane per iteration, one atomic per wavefront
__global__ void __launch_bounds__(256) bit_errors_kernel(BitErrArgs a) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nvec = (((uintptr_t)a.a | (uintptr_t)a.b) & 15u) == 0 ? a.n_bytes / 16 : 0;
    unsigned long long acc = 0;
    for (size_t v = i; v < nvec; v += stride) {
        const uint4 x = ((const uint4*)a.a)[v], y = ((const uint4*)a.b)[v];
        acc += __builtin_popcount(x.x ^ y.x) + __builtin_popcount(x.y ^ y.y) + __builtin_popcount(x.z ^ y.z) +
               __builtin_popcount(x.w ^ y.w);
    }
    for (size_t k = nvec * 16 + i; k < a.n_bytes; k += stride) acc += __builtin_popcount((uint32_t)(a.a[k] ^ a.b[k]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(a.count, acc);
}

// ---- the shader clock under a packed-integer load (measurement harness) -----------------------------------------------
// Every wave runs `iters` rounds of 64 independent v_pk_add_u16 (the update kernels' instruction class) and brackets them with
// s_memtime (counts shader-core clocks) and s_memrealtime (counts a constant reference clock, hipDeviceAttributeWallClockRate):
// their ratio is the clock the SIMDs actually ran at while loaded the way the update kernels load them -- the number a
// cycles-per-instruction ceiling has to be multiplied with.  out[wave] = {shader clocks, reference clocks}.
__global__ void __launch_bounds__(256) shader_clock_kernel(uint64_t* out, uint32_t iters, uint32_t seed) {
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x * 16u + (uint32_t)i;
    const uint32_t inc = seed | 0x00010001u;
    const uint64_t c0 = __builtin_readcyclecounter();          // s_memtime
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t k = 0; k < iters; ++k) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[i]) : "v"(inc));
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= v[i];
    if ((threadIdx.x & 63u) == 0) {
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[2 * w] = (c1 - c0) + (x == 0x12345u ? 1u : 0u);   // keeps the adds alive
        out[2 * w + 1] = r1 - r0;
    }
}

}  // namespace vit
