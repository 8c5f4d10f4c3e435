// viterbi_hip/viterbi_decoder_hip.h -- ViterbiDecoder_HIP<K,R,error_t,soft_t>: the MI355X decoder STRATEGY.
// It satisfies the reference's decoder concept (include/viterbi/viterbi_decoder_scalar.h:16-29): `is_valid` and a static
// `update<sum_error_t>(Base&, const soft_t* symbols, size_t N)` that leaves Base::m_metrics, the decision rows
// [cursor, cursor + N/R) and m_current_decoded_bit exactly as ViterbiDecoder_Scalar would -- bit for bit, including the
// strict-'>' tie rule and wrapping error_t arithmetic -- so it slots into the reference's factory lists
// (examples/helpers/simd_type.h:50-112) next to SCALAR / SIMD_SSE / SIMD_AVX.  May be called repeatedly with any
// multiple of R symbols (streaming, examples/helpers/puncture_code_helpers.h:51); calls of up to 64 trellis steps are queued
// on the host and run on the GPU together (viterbi_decoder_core.h, "deferred streaming"): the state every accessor shows is
// the reference's, the return value is the renormalisation sum of the steps computed since the last call returned one.
#pragma once
#include <type_traits>

#include "viterbi_decoder_core.h"

template <size_t constraint_length, size_t code_rate, typename error_t, typename soft_t>
class ViterbiDecoder_HIP {
public:
    using Base = ViterbiDecoder_Core<constraint_length, code_rate, error_t, soft_t>;
    // what libvit_hip.so serves: 2 <= K <= 16 (the state metrics of a frame live in the 160 KiB LDS of one CU), R <= 8,
    // (int16_t,uint16_t) or (int8_t,uint8_t)
    static constexpr bool is_valid =
        Base::K >= 2 && Base::K <= 16 && Base::R >= 1 && Base::R <= 8 &&
        ((std::is_same<soft_t, int16_t>::value && std::is_same<error_t, uint16_t>::value) ||
         (std::is_same<soft_t, int8_t>::value && std::is_same<error_t, uint8_t>::value));

    template <typename sum_error_t>
    static sum_error_t update(Base& base, const soft_t* symbols, const size_t N) {
        static_assert(is_valid, "ViterbiDecoder_HIP: unsupported K, R or metric types");
        assert(N % Base::R == 0);
        const size_t total_decoded_bits = N / Base::R;
        assert(total_decoded_bits + base.m_current_decoded_bit <= base.get_traceback_length() + Base::TOTAL_STATE_BITS);
        if (total_decoded_bits == 0) return sum_error_t(0);
        if (total_decoded_bits <= Base::MAX_DEFERRED_CALL_STEPS) {
            // streaming: queue the steps (ViterbiDecoder_Core::enqueue_steps); they run in one launch with their neighbours
            base.enqueue_steps(symbols, total_decoded_bits);
            return sum_error_t(base.take_unreported_renormalisation());
        }
        base.flush_pending();
        uint64_t renorm = 0;
        // one launch; the rows stay on the device until somebody reads m_decisions (viterbi_decoder_core.h: run_update), and a call
        // that completes the frame also decodes it (chainback(traceback length, end state 0) then costs a memcpy)
        base.run_update(symbols, total_decoded_bits, base.m_current_decoded_bit, &renorm);
        base.m_current_decoded_bit += total_decoded_bits;
        return sum_error_t(renorm + base.take_unreported_renormalisation());
    }
};
