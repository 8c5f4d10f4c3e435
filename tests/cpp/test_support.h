// test_support.h -- helpers for the C++ test programs: stock codes / decode configurations (same constants as the
// reference harness: examples/helpers/common_codes.h:20-30, decode_type.h:21-64), a shift-register encoder for test data,
// deterministic noise.  Test infrastructure only.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <limits>
#include <vector>

#include "viterbi_hip/viterbi_decoder_config.h"

template <typename soft_t, typename error_t>
struct DecodeSetup {
    soft_t high, low;
    ViterbiDecoder_Config<error_t> config;
};

template <typename soft_t, typename error_t>
DecodeSetup<soft_t, error_t> make_setup(int high, int low, unsigned margin_mult, size_t R) {
    const error_t max_error = error_t(error_t(high - low) * error_t(R));
    const error_t margin = error_t(max_error * error_t(margin_mult));
    DecodeSetup<soft_t, error_t> s;
    s.high = soft_t(high);
    s.low = soft_t(low);
    s.config.soft_decision_max_error = max_error;
    s.config.initial_start_error = 0;
    s.config.initial_non_start_error = margin;
    s.config.renormalisation_threshold = error_t(std::numeric_limits<error_t>::max() - margin);
    return s;
}
inline DecodeSetup<int16_t, uint16_t> soft16_setup(size_t R) { return make_setup<int16_t, uint16_t>(127, -127, 5, R); }
inline DecodeSetup<int8_t, uint8_t> soft8_setup(size_t R) { return make_setup<int8_t, uint8_t>(3, -3, 2, R); }
inline DecodeSetup<int8_t, uint8_t> hard8_setup(size_t R) { return make_setup<int8_t, uint8_t>(1, -1, 3, R); }

// MSB-first bits in, K-1 zero tail bits, one soft symbol per coded bit, step-major / polynomial-minor
template <typename soft_t, typename code_t>
std::vector<soft_t> encode_frame(size_t K, size_t R, const code_t* G, const std::vector<uint8_t>& bytes, soft_t high, soft_t low) {
    std::vector<soft_t> out;
    out.reserve((bytes.size() * 8 + K - 1) * R);
    uint32_t reg = 0;
    const size_t total = bytes.size() * 8 + K - 1;
    for (size_t b = 0; b < total; b++) {
        const uint32_t in = b < bytes.size() * 8 ? (bytes[b / 8] >> (7 - b % 8)) & 1u : 0u;
        reg = (reg << 1) | in;
        for (size_t i = 0; i < R; i++) out.push_back((__builtin_popcount(reg & uint32_t(G[i]) & ((1u << K) - 1u)) & 1) ? high : low);
    }
    return out;
}

struct XorShift {
    uint64_t s;
    explicit XorShift(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    uint32_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return uint32_t(s >> 16); }
};

inline size_t count_bit_errors(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b) {
    size_t n = 0;
    for (size_t i = 0; i < a.size(); i++) n += size_t(__builtin_popcount(unsigned(a[i] ^ b[i])));
    return n;
}
