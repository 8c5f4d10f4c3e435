// viterbi_hip/viterbi_branch_table.h -- host-side branch table with the reference's interface
// (include/viterbi/viterbi_branch_table.h:20-74): ctor(G, high, low), operator[](rate index), data().
// The table is plain host memory; vit_hip_create() copies it to the GPU, and one table can serve many decoders.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <array>

template <size_t constraint_length, size_t code_rate, typename soft_t>
class ViterbiBranchTable {
public:
    static constexpr size_t K = constraint_length;
    static constexpr size_t R = code_rate;
    static constexpr size_t TOTAL_STATE_BITS = K - 1;
    // butterfly symmetry: only the half-states (0|X|0) are tabulated
    static constexpr size_t NUMSTATES = (size_t(1) << TOTAL_STATE_BITS) / 2;
    static constexpr size_t SIZE_IN_BYTES = sizeof(soft_t) * NUMSTATES;

    // G: R polynomials, least significant bit = newest (input) bit
    template <typename code_t>
    ViterbiBranchTable(const code_t* G, soft_t high, soft_t low) : m_high(high), m_low(low) {
        static_assert(K > 1 && R > 1, "need K >= 2 and R >= 2");
        for (size_t i = 0; i < R; i++) {
            for (size_t s = 0; s < NUMSTATES; s++) {
                const uint64_t taps = (uint64_t(s) << 1) & uint64_t(G[i]);
                m_rows[i][s] = (__builtin_popcountll(taps) & 1) ? high : low;
            }
        }
    }

    const soft_t* operator[](size_t rate_index) const { return m_rows[rate_index].data(); }
    const soft_t* data() const { return m_rows[0].data(); }   // [R][NUMSTATES], rows contiguous
    soft_t* data() { return m_rows[0].data(); }               // writable: the receiving side of vit_hip_broadcast_table
    soft_t soft_decision_high() const { return m_high; }
    soft_t soft_decision_low() const { return m_low; }
    // after the rows were overwritten through data() (the receiving side of vit_hip_broadcast_table): recover high / low from
    // them -- row entry 0 is always `low` (parity of half-state 0 is 0), any other value in the table is `high`
    void refresh_levels() {
        m_low = m_rows[0][0];
        m_high = m_low;
        for (size_t i = 0; i < R; i++)
            for (size_t s = 0; s < NUMSTATES; s++)
                if (m_rows[i][s] != m_low) { m_high = m_rows[i][s]; return; }
    }

private:
    soft_t m_high, m_low;
    alignas(64) std::array<std::array<soft_t, NUMSTATES>, R> m_rows;
};
