// viterbi_hip/viterbi_decoder_core.h -- decoder state object with the reference's public surface
// (include/viterbi/viterbi_decoder_core.h:157-243): same template parameters in the same order, same member functions,
// same public data members (m_branch_table, m_config, m_metrics, m_decisions, m_current_decoded_bit) with the same
// accessors (m_metrics.get_old(), m_decisions[t]) -- so code written against the reference compiles against this header.
// The state is host-visible; update() (viterbi_decoder_hip.h) and chainback() execute on the GPU through the C ABI.
// There is no CPU decode path: a failing GPU call aborts loudly, like the reference's asserts.
#pragma once
#include <assert.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../vit_hip.h"
#include "viterbi_branch_table.h"
#include "viterbi_decoder_config.h"

namespace viterbi_hip_detail {
inline void require_ok(int rc, const char* what) {
    if (rc != VIT_HIP_OK) {
        fprintf(stderr, "viterbi_hip: %s failed (%d): %s\n", what, rc, vit_hip_last_error());
        abort();
    }
}
}  // namespace viterbi_hip_detail

// double-buffered path metrics; after every update() the "old" buffer holds the live metrics
template <size_t constraint_length, typename error_t>
class ViterbiErrorMetrics {
public:
    static constexpr size_t K = constraint_length;
    static constexpr size_t TOTAL_STATE_BITS = K - 1;
    static constexpr size_t NUMSTATES = size_t(1) << TOTAL_STATE_BITS;
    error_t* get_old() { return m_buf[m_index]; }
    error_t* get_new() { return m_buf[1 - m_index]; }
    void swap() { m_index = 1 - m_index; }
private:
    alignas(64) error_t m_buf[2][NUMSTATES];
    size_t m_index = 0;
};

// decision history: row t = max(NUMSTATES/64, 1) 64-bit words, bit s%64 of word s/64 = decision for next-state s
template <size_t constraint_length, typename decision_bits_t>
class ViterbiDecisionBits {
public:
    using format_t = decision_bits_t;
    static constexpr size_t K = constraint_length;
    static constexpr size_t NUMSTATES = size_t(1) << (K - 1);
    static constexpr size_t TOTAL_BITS_PER_BLOCK = sizeof(format_t) * 8;
    static constexpr size_t TOTAL_BLOCKS = (NUMSTATES / TOTAL_BITS_PER_BLOCK) ? (NUMSTATES / TOTAL_BITS_PER_BLOCK) : 1;
    void resize(size_t rows) { m_words.resize(rows * TOTAL_BLOCKS); m_rows = rows; }
    size_t size() const { return m_rows; }
    format_t* operator[](size_t row) { return m_words.data() + row * TOTAL_BLOCKS; }
    const format_t* operator[](size_t row) const { return m_words.data() + row * TOTAL_BLOCKS; }
private:
    std::vector<format_t> m_words;
    size_t m_rows = 0;
};

template <size_t constraint_length, size_t code_rate, typename error_t, typename soft_t>
class ViterbiDecoder_Core {
public:
    static constexpr size_t K = constraint_length;
    static constexpr size_t R = code_rate;
    static constexpr size_t TOTAL_STATE_BITS = K - 1;
    static constexpr size_t NUMSTATES = size_t(1) << TOTAL_STATE_BITS;
    using BranchTable = ViterbiBranchTable<K, R, soft_t>;
    using Config = ViterbiDecoder_Config<error_t>;
    using Metrics = ViterbiErrorMetrics<K, error_t>;
    using Decisions = ViterbiDecisionBits<K, uint64_t>;
    static_assert(sizeof(uintptr_t) == 8, "decision words are 64-bit, as in the reference on 64-bit hosts");

    ViterbiDecoder_Core(const BranchTable& branch_table, const Config& config, int device = 0)
        : m_branch_table(branch_table), m_config(config) {
        static_assert(K >= 2 && R >= 1, "need K >= 2 and R >= 1");
        viterbi_hip_detail::require_ok(
            vit_hip_create(int(K), int(R), int(sizeof(soft_t)), int(sizeof(error_t)), branch_table.data(), &m_config, device,
                           &m_hip),
            "vit_hip_create");
        reset();
        set_traceback_length(0);
    }
    ~ViterbiDecoder_Core() { vit_hip_destroy(m_hip); }
    ViterbiDecoder_Core(const ViterbiDecoder_Core&) = delete;
    ViterbiDecoder_Core& operator=(const ViterbiDecoder_Core&) = delete;

    // number of decoded (information) bits kept for traceback; the K-1 tail steps are added on top
    void set_traceback_length(size_t traceback_length) {
        const size_t rows = traceback_length + TOTAL_STATE_BITS;
        m_decisions.resize(rows);
        if (m_current_decoded_bit > rows) m_current_decoded_bit = rows;
    }
    size_t get_traceback_length() const { return m_decisions.size() - TOTAL_STATE_BITS; }

    error_t get_error(size_t end_state = 0) {
        assert(end_state < NUMSTATES);
        return m_metrics.get_old()[end_state];
    }

    void reset(size_t starting_state = 0) {
        m_current_decoded_bit = 0;
        error_t* m = m_metrics.get_old();
        for (size_t s = 0; s < NUMSTATES; s++) m[s] = m_config.initial_non_start_error;
        m[starting_state & (NUMSTATES - 1)] = m_config.initial_start_error;
    }

    // traceback on the GPU (vit_hip_chainback_host): bytes_out receives ceil(total_bits/8) bytes, MSB-first
    void chainback(uint8_t* bytes_out, size_t total_bits, size_t end_state = 0) {
        assert(get_traceback_length() >= total_bits);
        assert(m_current_decoded_bit >= total_bits + TOTAL_STATE_BITS);
        assert(end_state < NUMSTATES);
        viterbi_hip_detail::require_ok(vit_hip_chainback_host(m_hip, m_decisions[0], total_bits, end_state, bytes_out),
                                       "vit_hip_chainback_host");
    }

    vit_hip_handle hip_handle() const { return m_hip; }

public:
    const BranchTable& m_branch_table;
    const Config m_config;
    Metrics m_metrics;
    Decisions m_decisions;
    size_t m_current_decoded_bit = 0;

private:
    vit_hip_handle m_hip = nullptr;
};
