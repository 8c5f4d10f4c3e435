// run_stream_hip.cpp -- the reference's STREAMING call pattern against the HIP strategy: update() with the R symbols of one
// trellis step per call (examples/helpers/puncture_code_helpers.h:51, examples/run_punctured_decoder.cpp:165-176), a punctured
// K=7 R=1/4 stream with erasure symbols.  Checks, against one update() call over the whole stream on a second decoder:
// the running total of the return values, get_error(), every metric, every decision word, the chainback bytes -- and, mid-stream,
// that reading the decoder's public state (m_metrics.get_old(), m_decisions[row]) shows the up-to-date values.  Then the EXACT
// mode (set_exact_update_return: per-call return values are the reference's) and a frame shorter than the traceback buffer.
// Prints the amortised cost of one N = R call.   usage: run_stream_hip [info bits]
#include <inttypes.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "viterbi_hip/viterbi_decoder_core.h"
#include "viterbi_hip/viterbi_decoder_hip.h"
#include "test_support.h"

int main(int argc, char** argv) {
    constexpr size_t K = 7, R = 4;
    const uint8_t G[R] = {109, 79, 83, 109};
    const auto setup = soft16_setup(R);
    const size_t total_input_bits = argc > 1 ? size_t(atol(argv[1])) : 8192, total_input_bytes = total_input_bits / 8;
    const size_t S = total_input_bits + K - 1;
    XorShift rng(7);
    std::vector<uint8_t> tx(total_input_bytes), rx_stream(total_input_bytes), rx_once(total_input_bytes);
    for (auto& b : tx) b = uint8_t(rng.next());
    std::vector<int16_t> symbols = encode_frame<int16_t>(K, R, G, tx, setup.high, setup.low);
    // noise, and every fourth symbol punctured (erasure value 0)
    for (size_t i = 0; i < symbols.size(); i++) {
        const int n = int(rng.next() % 161) - 80;
        int v = int(symbols[i]) + n;
        symbols[i] = int16_t(v > setup.high ? setup.high : v < setup.low ? setup.low : v);
        if (i % 4 == 3) symbols[i] = 0;
    }
    auto branch_table = ViterbiBranchTable<K, R, int16_t>(G, setup.high, setup.low);
    using Core = ViterbiDecoder_Core<K, R, uint16_t, int16_t>;
    using Decoder = ViterbiDecoder_HIP<K, R, uint16_t, int16_t>;

    Core once(branch_table, setup.config);
    once.set_traceback_length(total_input_bits);
    once.reset();
    const uint64_t acc_once = Decoder::template update<uint64_t>(once, symbols.data(), symbols.size());
    const uint64_t err_once = acc_once + uint64_t(once.get_error());
    if (acc_once == 0) { printf("FAIL: the test frame never renormalises\n"); return 1; }
    once.chainback(rx_once.data(), total_input_bits);

    Core vitdec(branch_table, setup.config);
    vitdec.set_traceback_length(total_input_bits);
    bool ok = true;
    double per_call_us = 0;
    for (int rep = 0; rep < 3; rep++) {                      // a Core is reused across frames after reset()
        vitdec.reset();
        uint64_t acc = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t t = 0; t < S; t++) acc += Decoder::template update<uint64_t>(vitdec, symbols.data() + t * R, R);
        const auto t1 = std::chrono::steady_clock::now();
        per_call_us = std::chrono::duration<double>(t1 - t0).count() * 1e6 / double(S);
        ok = ok && vitdec.m_current_decoded_bit == S && vitdec.pending_steps() == 0;   // the last call completes the frame: nothing is left queued
        ok = ok && acc == acc_once;
        ok = ok && acc + uint64_t(vitdec.get_error()) == err_once;
        ok = ok && memcmp(vitdec.m_metrics.get_old(), once.m_metrics.get_old(), Core::NUMSTATES * sizeof(uint16_t)) == 0;
        ok = ok && memcmp(vitdec.m_decisions[0], once.m_decisions[0], S * sizeof(uint64_t)) == 0;
        vitdec.chainback(rx_stream.data(), total_input_bits);
        ok = ok && rx_stream == rx_once;
    }
    // mid-stream reads: stop after 1000 single-step calls and look at the public state
    {
        Core ref(branch_table, setup.config);
        ref.set_traceback_length(total_input_bits);
        ref.reset();
        const size_t n = 1000 < S ? 1000 : S / 2;
        const uint64_t acc_ref = Decoder::template update<uint64_t>(ref, symbols.data(), n * R);   // one call of n steps: runs directly
        vitdec.reset();
        uint64_t acc = 0;
        for (size_t t = 0; t < n; t++) acc += Decoder::template update<uint64_t>(vitdec, symbols.data() + t * R, R);
        ok = ok && vitdec.m_current_decoded_bit == n;
        ok = ok && memcmp(vitdec.m_decisions[n - 1], ref.m_decisions[n - 1], sizeof(uint64_t)) == 0;        // flushes
        ok = ok && vitdec.pending_steps() == 0;
        ok = ok && memcmp(vitdec.m_metrics.get_old(), ref.m_metrics.get_old(), Core::NUMSTATES * sizeof(uint16_t)) == 0;
        acc += Decoder::template update<uint64_t>(vitdec, symbols.data() + n * R, R);                         // reports what the flush computed
        ok = ok && acc >= acc_ref;
        acc += Decoder::template update<uint64_t>(vitdec, symbols.data() + (n + 1) * R, (S - n - 1) * R);     // the rest in one call
        ok = ok && acc == acc_once && acc + uint64_t(vitdec.get_error()) == err_once;
    }
    // EXACT mode: every call returns the reference's per-call value -- its running total after n calls is what ONE call over the
    // first n steps returns (checked at 16 prefixes), nothing is ever queued
    {
        Core exact(branch_table, setup.config);
        exact.set_traceback_length(total_input_bits);
        exact.set_exact_update_return(true);
        exact.reset();
        std::vector<uint64_t> prefix(S + 1, 0);
        for (size_t t = 0; t < S; t++) {
            prefix[t + 1] = prefix[t] + Decoder::template update<uint64_t>(exact, symbols.data() + t * R, R);
            ok = ok && exact.pending_steps() == 0;
        }
        ok = ok && prefix[S] == acc_once && prefix[S] + uint64_t(exact.get_error()) == err_once;
        ok = ok && memcmp(exact.m_decisions[0], once.m_decisions[0], S * sizeof(uint64_t)) == 0;
        Core ref(branch_table, setup.config);
        ref.set_traceback_length(total_input_bits);
        for (int k = 1; k <= 16; k++) {
            const size_t n = 65 + (S - 66) * size_t(k) / 16;                        // > 64 steps: one direct launch
            ref.reset();
            ok = ok && Decoder::template update<uint64_t>(ref, symbols.data(), n * R) == prefix[n];
        }
        if (!ok) printf("exact-mode check failed\n");
    }
    // a frame SHORTER than the traceback buffer, streamed at N = R in the default (deferred) mode: the last calls stay queued, the
    // flush comes from get_error() -- its sum is owed to the SAME frame: take_unreported_renormalisation() collects it; reset()
    // drops it, so that the next frame's total starts from zero (the reference's `reset(); err = 0; err += update(...)` pattern)
    {
        Core shorty(branch_table, setup.config);
        shorty.set_traceback_length(total_input_bits + 777);
        for (int collect = 0; collect < 2; collect++) {
            shorty.reset();
            uint64_t acc = 0;
            for (size_t t = 0; t < S; t++) acc += Decoder::template update<uint64_t>(shorty, symbols.data() + t * R, R);
            const uint64_t e = uint64_t(shorty.get_error());                        // runs what is still queued
            ok = ok && shorty.pending_steps() == 0 && e == uint64_t(once.get_error());
            if (collect) {
                acc += shorty.take_unreported_renormalisation();
                ok = ok && acc == acc_once && acc + e == err_once;
            } else {
                shorty.reset();                                                     // the debt does NOT survive the reset ...
                ok = ok && acc + shorty.last_dropped_renormalisation() == acc_once;   // ... but what was dropped stays readable
                const uint64_t first = Decoder::template update<uint64_t>(shorty, symbols.data(), R);
                (void)shorty.get_error();                                           // (run the queued step)
                ok = ok && first + shorty.take_unreported_renormalisation() == 0;   // one step from reset renormalises nothing
                ok = ok && acc <= acc_once;                                         // what the dropped tail held is simply not reported
            }
        }
        if (!ok) printf("short-frame check failed\n");
    }
    const size_t errors = count_bit_errors(tx, rx_stream);
    printf("streaming update(N=R): %zu calls, %.3f us per call amortised (%.1f Mbit/s), %zu/%zu incorrect bits, error_metric=%" PRIu64 "\n",
           S, per_call_us, 1.0 / per_call_us, errors, total_input_bits, err_once);
    printf("%s\n", ok ? "PASS" : "FAIL");
    return ok ? 0 : 1;
}
