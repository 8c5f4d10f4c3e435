// viterbi_hip/viterbi_decoder_config.h -- decoder constants, same fields and order as the reference's
// ViterbiDecoder_Config<error_t> (include/viterbi/viterbi_decoder_config.h:11-18) so aggregate initialisation and the
// C ABI's `config` argument (4 x error_t) both work on it unchanged.
#pragma once

template <typename error_t>
struct ViterbiDecoder_Config {
    error_t soft_decision_max_error;    // (high - low) * R: largest branch error of one trellis step
    error_t initial_start_error;        // metric of the starting state after reset()
    error_t initial_non_start_error;    // metric of every other state after reset()
    error_t renormalisation_threshold;  // renormalise when metric[0] reaches this
};
