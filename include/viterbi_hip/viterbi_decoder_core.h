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

namespace viterbi_hip_detail {
// Streamed update() calls are queued on the host and run on the GPU in one launch (ViterbiDecoder_Core::flush_pending).
// The public data members callers may read directly -- m_metrics.get_old(), m_decisions[row] -- bring the decoder up to
// date first through this hook (a null hook: nothing is ever pending).
struct FlushHook {
    void (*fn)(void*) = nullptr;
    void* owner = nullptr;
    void operator()() const { if (fn) fn(owner); }
};
}  // namespace viterbi_hip_detail

// double-buffered path metrics; after every update() the "old" buffer holds the live metrics
template <size_t constraint_length, typename error_t>
class ViterbiErrorMetrics {
public:
    static constexpr size_t K = constraint_length;
    static constexpr size_t TOTAL_STATE_BITS = K - 1;
    static constexpr size_t NUMSTATES = size_t(1) << TOTAL_STATE_BITS;
    error_t* get_old() { m_hook(); return m_buf[m_index]; }
    error_t* get_new() { m_hook(); return m_buf[1 - m_index]; }
    void swap() { m_index = 1 - m_index; }
    error_t* raw_old() { return m_buf[m_index]; }                       // no flush: the decoder's own accesses
    void set_flush_hook(viterbi_hip_detail::FlushHook h) { m_hook = h; }
private:
    alignas(64) error_t m_buf[2][NUMSTATES];
    size_t m_index = 0;
    viterbi_hip_detail::FlushHook m_hook;
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
    void resize(size_t rows) { m_write_hook(); m_words.resize(rows * TOTAL_BLOCKS); m_rows = rows; }
    size_t size() const { return m_rows; }
    // a mutable row may be written through: the host copy becomes the only authority (m_write_hook); a const row is only brought up to date
    format_t* operator[](size_t row) { m_write_hook(); return m_words.data() + row * TOTAL_BLOCKS; }
    const format_t* operator[](size_t row) const { m_hook(); return m_words.data() + row * TOTAL_BLOCKS; }
    format_t* raw_row(size_t row) { return m_words.data() + row * TOTAL_BLOCKS; }   // no flush: the decoder's own accesses
    void set_flush_hook(viterbi_hip_detail::FlushHook h) { m_hook = h; if (!m_write_hook.fn) m_write_hook = h; }
    void set_write_hook(viterbi_hip_detail::FlushHook h) { m_write_hook = h; }
private:
    std::vector<format_t> m_words;
    size_t m_rows = 0;
    viterbi_hip_detail::FlushHook m_hook, m_write_hook;
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
        install_hooks();
        reset();
        set_traceback_length(0);
    }
    ~ViterbiDecoder_Core() { vit_hip_destroy(m_hip); }
    // Copyable like the reference's class (implicit copy constructor there; assignment is ill-formed there too: a reference and a
    // const member).  The copy gets a device handle of its own on the same GPU and the source's whole state -- metrics, decision
    // history, cursor, what is still owed to update()'s return value -- after the source has run whatever it had queued.
    ViterbiDecoder_Core(const ViterbiDecoder_Core& other) : m_branch_table(other.m_branch_table), m_config(other.m_config) {
        const_cast<ViterbiDecoder_Core&>(other).flush_pending();
        const_cast<ViterbiDecoder_Core&>(other).fetch_device_rows();      // the copy starts from host rows that are complete
        vit_hip_info info;
        viterbi_hip_detail::require_ok(vit_hip_get_info(other.m_hip, &info), "vit_hip_get_info");
        viterbi_hip_detail::require_ok(
            vit_hip_create(int(K), int(R), int(sizeof(soft_t)), int(sizeof(error_t)), m_branch_table.data(), &m_config, info.device,
                           &m_hip),
            "vit_hip_create");
        m_metrics = other.m_metrics;
        m_decisions = other.m_decisions;
        install_hooks();
        m_current_decoded_bit = other.m_current_decoded_bit;
        m_unreported_renormalisation = other.m_unreported_renormalisation;
        m_exact_update_return = other.m_exact_update_return;
    }
    ViterbiDecoder_Core& operator=(const ViterbiDecoder_Core&) = delete;

    // number of decoded (information) bits kept for traceback; the K-1 tail steps are added on top
    void set_traceback_length(size_t traceback_length) {
        flush_pending();
        fetch_device_rows();
        const size_t rows = traceback_length + TOTAL_STATE_BITS;
        m_decisions.resize(rows);
        if (m_current_decoded_bit > rows) m_current_decoded_bit = rows;
    }
    size_t get_traceback_length() const { return m_decisions.size() - TOTAL_STATE_BITS; }

    error_t get_error(size_t end_state = 0) {
        assert(end_state < NUMSTATES);
        return m_metrics.get_old()[end_state];      // get_old() runs whatever update() calls are still queued
    }

    void reset(size_t starting_state = 0) {
        // A frame's renormalisation never leaks into the next frame's total: whatever is still queued is run (the reference has run
        // every step update() was given, and the decision rows / metrics stay readable until overwritten), and a sum no update()
        // call has returned yet is DROPPED here.  That can only be the tail of a frame shorter than the traceback buffer streamed in
        // short calls (deferred mode; its last calls were flushed by get_error() / chainback() / this reset): a caller who wants
        // that frame's exact total collects it with take_unreported_renormalisation() BEFORE reset(), or streams in exact mode
        // (set_exact_update_return), where nothing is ever owed.
        flush_pending();
        m_dropped_renormalisation = m_unreported_renormalisation;      // readable afterwards: last_dropped_renormalisation()
        m_unreported_renormalisation = 0;
        m_current_decoded_bit = 0;
        error_t* m = m_metrics.raw_old();
        for (size_t s = 0; s < NUMSTATES; s++) m[s] = m_config.initial_non_start_error;
        m[starting_state & (NUMSTATES - 1)] = m_config.initial_start_error;
    }

    // traceback on the GPU (vit_hip_chainback_host): bytes_out receives ceil(total_bits/8) bytes, MSB-first
    void chainback(uint8_t* bytes_out, size_t total_bits, size_t end_state = 0) {
        assert(get_traceback_length() >= total_bits);
        assert(m_current_decoded_bit >= total_bits + TOTAL_STATE_BITS);
        assert(end_state < NUMSTATES);
        flush_pending();
        if (m_dev_begin == 0 && m_dev_end >= total_bits + TOTAL_STATE_BITS) {
            // every row the traceback reads was made on the device and is still there: no row travels (and if the update that
            // completed the frame already chained back exactly these bits, no kernel runs either)
            viterbi_hip_detail::require_ok(vit_hip_chainback_host_lazy(m_hip, total_bits, end_state, bytes_out), "vit_hip_chainback_host_lazy");
            return;
        }
        fetch_device_rows();
        viterbi_hip_detail::require_ok(vit_hip_chainback_host(m_hip, m_decisions.raw_row(0), total_bits, end_state, bytes_out),
                                       "vit_hip_chainback_host");
    }

    // ---- where the decision rows live ------------------------------------------------------------------------------------------
    // update() leaves the rows it computes on the DEVICE (vit_hip_update_host_lazy): rows [m_dev_begin, m_dev_end) of m_decisions are
    // stale on the host and authoritative in the handle's row store; every other row is authoritative on the host.  Reading
    // m_decisions[row] fetches the stale rows first; a MUTABLE m_decisions[row] (the caller may write through it), resize and copy
    // also hand authority back to the host, after which chainback() uploads the host rows as the reference's does.
    void run_update(const soft_t* symbols, size_t steps, size_t first_row, uint64_t* renorm) {
        if (m_dev_end > m_dev_begin && (first_row > m_dev_end || first_row + steps < m_dev_begin)) fetch_device_rows();   // not adjacent: one range only
        viterbi_hip_detail::require_ok(
            vit_hip_update_host_lazy(m_hip, m_metrics.raw_old(), symbols, steps, first_row, get_traceback_length(), 0, renorm),
            "vit_hip_update_host_lazy");
        if (m_dev_end == m_dev_begin) { m_dev_begin = first_row; m_dev_end = first_row + steps; }
        else {
            if (first_row < m_dev_begin) m_dev_begin = first_row;
            if (first_row + steps > m_dev_end) m_dev_end = first_row + steps;
        }
    }
    void fetch_device_rows() {
        if (m_dev_end == m_dev_begin) return;
        const size_t b = m_dev_begin, e = m_dev_end < m_decisions.size() ? m_dev_end : m_decisions.size();
        m_dev_begin = m_dev_end = 0;                            // first: raw accesses below must not re-enter
        if (e > b)
            viterbi_hip_detail::require_ok(vit_hip_fetch_decisions_host(m_hip, b, e - b, m_decisions.raw_row(b)), "vit_hip_fetch_decisions_host");
    }
    size_t rows_on_device() const { return m_dev_end - m_dev_begin; }

    vit_hip_handle hip_handle() const { return m_hip; }
    // the kernel plan behind this decoder and, where it is the slow compatibility plan, how to get a faster one (vit_hip_plan_note)
    const char* plan_note() const { return vit_hip_plan_note(m_hip); }

    // ---- deferred streaming (used by ViterbiDecoder_HIP::update) -------------------------------------------------------------
    // The reference's streaming callers hand update() the R symbols of ONE trellis step at a time
    // (examples/helpers/puncture_code_helpers.h:51).  One GPU launch per step would cost 25 us a call; instead short calls are
    // queued here (a memcpy) and run as ONE vit_hip_update_host when the queue is full, when the cursor reaches the end of the
    // traceback buffer, or when anything reads the decoder's state (get_error, chainback, m_metrics.get_old(), m_decisions[row]).
    // m_current_decoded_bit counts queued steps at once.  update()'s return value is the renormalisation sum of every step
    // COMPUTED since the last value it returned: zero while steps are queued, their whole sum from the call that runs them, so
    // the caller's running total (accumulated_error += update(...)) is the reference's whenever the queue is empty -- in
    // particular after the call that completes a frame.
    //
    // EXACT mode (set_exact_update_return(true), or compile with -DVITERBI_HIP_EXACT_UPDATE_RETURN): nothing is deferred -- every
    // update() call runs its steps at once and returns exactly what ViterbiDecoder_Scalar::update returns for that call
    // (scalar.h:29-56), at one GPU launch (about 25 us) per call.  For callers that use the per-call value itself rather than
    // its running total.
    static constexpr size_t MAX_PENDING_STEPS = 2048;
    static constexpr size_t MAX_DEFERRED_CALL_STEPS = 64;     // calls longer than this flush the queue and run directly

    void set_exact_update_return(bool exact) {
        flush_pending();
        m_exact_update_return = exact;
    }
    bool get_exact_update_return() const { return m_exact_update_return; }

    void enqueue_steps(const soft_t* symbols, size_t steps) {
        m_pending_symbols.insert(m_pending_symbols.end(), symbols, symbols + steps * R);
        m_pending_steps += steps;
        m_current_decoded_bit += steps;
        if (m_exact_update_return || m_pending_steps >= MAX_PENDING_STEPS || m_current_decoded_bit >= m_decisions.size()) flush_pending();
    }
    // runs the queued steps; their renormalisation sum is added to what the next update() call returns
    void flush_pending() {
        if (m_pending_steps == 0) return;
        const size_t steps = m_pending_steps, first_row = m_current_decoded_bit - steps;
        m_pending_steps = 0;                                   // first: the accessors below must not re-enter
        uint64_t renorm = 0;
        run_update(m_pending_symbols.data(), steps, first_row, &renorm);
        m_pending_symbols.clear();
        m_unreported_renormalisation += renorm;
    }
    // the renormalisation sum of steps that have been computed but whose sum no update() call has returned yet (deferred mode:
    // flushes triggered by get_error(), chainback() or a direct read of the state); collecting it here settles the debt.  The
    // next update() of the SAME frame returns it otherwise; reset() drops it
    uint64_t take_unreported_renormalisation() {
        const uint64_t v = m_unreported_renormalisation;
        m_unreported_renormalisation = 0;
        return v;
    }
    // what the last reset() dropped (see reset()): non-zero only when a frame shorter than the traceback buffer was streamed in short
    // calls and nobody collected its tail -- a caller that logs per-frame totals can assert on it or add it to the frame it belongs to
    uint64_t last_dropped_renormalisation() const { return m_dropped_renormalisation; }
    size_t pending_steps() const { return m_pending_steps; }

public:
    const BranchTable& m_branch_table;
    const Config m_config;
    Metrics m_metrics;
    Decisions m_decisions;
    size_t m_current_decoded_bit = 0;

private:
    static void flush_hook(void* self) { static_cast<ViterbiDecoder_Core*>(self)->flush_pending(); }
    static void rows_hook(void* self) {
        static_cast<ViterbiDecoder_Core*>(self)->flush_pending();
        static_cast<ViterbiDecoder_Core*>(self)->fetch_device_rows();
    }
    void install_hooks() {
        m_metrics.set_flush_hook(viterbi_hip_detail::FlushHook{&ViterbiDecoder_Core::flush_hook, this});
        // reading or writing a row through m_decisions: queued steps run, device-resident rows come home
        m_decisions.set_flush_hook(viterbi_hip_detail::FlushHook{&ViterbiDecoder_Core::rows_hook, this});
        m_decisions.set_write_hook(viterbi_hip_detail::FlushHook{&ViterbiDecoder_Core::rows_hook, this});
    }
    vit_hip_handle m_hip = nullptr;
    size_t m_dev_begin = 0, m_dev_end = 0;       // rows [begin, end) of m_decisions are stale here and live in the handle's device row store
    std::vector<soft_t> m_pending_symbols;
    size_t m_pending_steps = 0;
    uint64_t m_unreported_renormalisation = 0;
    uint64_t m_dropped_renormalisation = 0;
#ifdef VITERBI_HIP_EXACT_UPDATE_RETURN
    bool m_exact_update_return = true;
#else
    bool m_exact_update_return = false;
#endif
};
