// run_simple_hip.cpp -- the reference's minimal usage example (examples/run_simple.cpp:19-95) against the HIP strategy:
// K=7 R=1/4 (DAB), 16-bit soft decisions, 1024 random bytes, noise-free; exit status != 0 on any bit error.
// Only the include lines and the `using Decoder = ...` line differ from a program written for the reference.
#include <inttypes.h>
#include <stdio.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "viterbi_hip/viterbi_decoder_core.h"
#include "viterbi_hip/viterbi_decoder_hip.h"
#include "test_support.h"

int main() {
    constexpr size_t K = 7, R = 4;
    const uint8_t G[R] = {109, 79, 83, 109};
    const auto setup = soft16_setup(R);

    const size_t total_input_bytes = 1024, total_input_bits = total_input_bytes * 8;
    XorShift rng(1);
    std::vector<uint8_t> tx(total_input_bytes), rx(total_input_bytes);
    for (auto& b : tx) b = uint8_t(rng.next());
    const std::vector<int16_t> symbols = encode_frame<int16_t>(K, R, G, tx, setup.high, setup.low);

    auto branch_table = ViterbiBranchTable<K, R, int16_t>(G, setup.high, setup.low);
    ViterbiDecoder_Core<K, R, uint16_t, int16_t> vitdec(branch_table, setup.config);
    using Decoder = ViterbiDecoder_HIP<K, R, uint16_t, int16_t>;

    vitdec.set_traceback_length(total_input_bits);
    vitdec.reset();
    const uint64_t accumulated_error = Decoder::template update<uint64_t>(vitdec, symbols.data(), symbols.size());
    const uint64_t error = accumulated_error + uint64_t(vitdec.get_error());
    vitdec.chainback(rx.data(), total_input_bits);
    printf("error_metric=%" PRIu64 "\n", error);

    // the latency of the route: reset -> update -> chainback of the same 8192-bit frame, median of 21 (BASELINE configs[0]'s call
    // pattern; the single-frame kernels of csrc/kernels_one.hpp serve every K <= 7)
    {
        std::vector<double> us;
        for (int rep = 0; rep < 21; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            vitdec.reset();
            (void)Decoder::template update<uint64_t>(vitdec, symbols.data(), symbols.size());
            vitdec.chainback(rx.data(), total_input_bits);
            us.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6);
        }
        std::sort(us.begin(), us.end());
        printf("latency: update + chainback of one %zu-bit frame: median %.1f us, min %.1f us (%.1f Mbit/s)\n", total_input_bits, us[10], us[0],
               double(total_input_bits) / us[10]);
    }

    // a decoder is copyable, as in the reference: a copy taken in mid-frame carries the whole state and finishes the frame on its own
    {
        const size_t half = (total_input_bits / 2) * R;
        vitdec.reset();
        uint64_t err_a = Decoder::template update<uint64_t>(vitdec, symbols.data(), half);
        auto twin = vitdec;
        uint64_t err_b = err_a;
        err_a += Decoder::template update<uint64_t>(vitdec, symbols.data() + half, symbols.size() - half);
        err_b += Decoder::template update<uint64_t>(twin, symbols.data() + half, symbols.size() - half);
        std::vector<uint8_t> rx_a(total_input_bytes), rx_b(total_input_bytes);
        vitdec.chainback(rx_a.data(), total_input_bits);
        twin.chainback(rx_b.data(), total_input_bits);
        const bool same = rx_a == rx && rx_b == rx && err_a == err_b && vitdec.get_error() == twin.get_error();
        printf("copy in mid-frame: %s\n", same ? "identical" : "DIFFERENT");
        if (!same) return 1;
    }

    const size_t total_errors = count_bit_errors(tx, rx);
    printf("%zu/%zu incorrect bits\n", total_errors, total_input_bits);
    if (total_errors > 0 || error != 0) {
        printf("ERROR: simple example had decoding errors\n");
        return 1;
    }
    return 0;
}
