/*
 * viterbi_oracle.h -- CPU restatement of the reference's update()+chainback() hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path (libvit_hip.so) never links,
 * loads or calls anything in oracle/.
 *
 * Parity status: PINNED.  oracle/ref_shim.cpp builds the real reference
 * (/root/reference/include/viterbi/ *.h) into oracle/_ref/libvitref.so and
 * tests/golden/make_golden.py dumps its outputs to tests/golden/ *.npz; tests/test_oracle.py checks this
 * restatement against both, bit for bit (decision words, metrics, renormalisation sum, chainback bytes).
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 */
#ifndef VITERBI_ORACLE_H
#define VITERBI_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Arithmetic description of one decoder instantiation: ViterbiDecoder_Core<K,R,error_t,soft_t>
 * (include/viterbi/viterbi_decoder_core.h:157-158) + ViterbiDecoder_Config<error_t>
 * (include/viterbi/viterbi_decoder_config.h:11-18). */
typedef struct vo_params {
    int32_t K;            /* constraint length, 2..25                                  */
    int32_t R;            /* code rate denominator, >= 1                               */
    int32_t soft_bytes;   /* sizeof(soft_t): 1 (int8_t) or 2 (int16_t)                 */
    int32_t error_bytes;  /* sizeof(error_t): 1 (uint8_t) or 2 (uint16_t)              */
    uint32_t soft_decision_max_error;
    uint32_t initial_start_error;
    uint32_t initial_non_start_error;
    uint32_t renormalisation_threshold;
} vo_params;

/* number of states, half states and 64-bit decision words per trellis step */
size_t vo_num_states(int K);
size_t vo_decision_words(int K);

/* ViterbiBranchTable ctor (viterbi_branch_table.h:34-55): table[i*H + s], values widened to int16. */
int vo_branch_table(int K, int R, const uint32_t* G, int soft_high, int soft_low, int16_t* table);

/* ViterbiDecoder_Core::reset (viterbi_decoder_core.h:202-211): metrics widened to uint32 (values < 2^(8*error_bytes)). */
void vo_reset(const vo_params* p, uint32_t* metrics, size_t starting_state);

/* ViterbiDecoder_Scalar::update (viterbi_decoder_scalar.h:29-55) incl. bfly (:58-136) and renormalise (:139-153).
 *   symbols   : n_steps*R values of soft_t (int8_t or int16_t according to p->soft_bytes)
 *   metrics   : [N] in/out ("old" metrics before/after)
 *   decisions : [n_steps][W] out, rows for THIS call only
 * returns the sum of the subtracted minima (update()'s return value). */
uint64_t vo_update(const vo_params* p, const int16_t* table, uint32_t* metrics,
                   const void* symbols, size_t n_steps, uint64_t* decisions);

/* ViterbiDecoder_Core::chainback (viterbi_decoder_core.h:214-236) with ViterbiTracebackBuffer (:87-153).
 *   decisions : [L+K-1][W], row 0 = first trellis step.  out: ceil(L/8) bytes. */
void vo_chainback(int K, const uint64_t* decisions, size_t L, size_t end_state, uint8_t* out);

/* Test-data generator: ConvolutionalEncoder_ShiftRegister::consume_byte
 * (convolutional_encoder_shift_register.h:42-62) driven as encode_data does (examples/helpers/test_helpers.h:17-64):
 * MSB-first input bits, K-1 zero tail bits, output bits step-major / polynomial-minor, one byte (0/1) per symbol.
 * out_bits must hold (8*n_bytes + K-1)*R entries. */
void vo_encode(int K, int R, const uint32_t* G, const uint8_t* bytes, size_t n_bytes, uint8_t* out_bits);

/* Whole-frame convenience used by bench.py's cpu_baseline ("port") and the at-scale parity checks:
 * reset -> update -> chainback for `frames` frames laid out [F][S][R]; threads = worker threads (>=1).
 * bytes_out [F][L/8]; final_metrics [F][N] (may be NULL); renorm_sum [F] (may be NULL). */
int vo_decode_frames(const vo_params* p, const int16_t* table, const void* symbols, size_t frames, size_t L,
                     uint8_t* bytes_out, uint32_t* final_metrics, uint64_t* renorm_sum, int threads);

/* The same, plus a 64-bit digest of every decision word of each frame (decision_hash [F], may be NULL), so that a whole
 * full-size batch can be compared word for word without keeping 4-70 GB of decision rows on the host:
 *   hash = sum_i words[i] * ((2 i + 1) * VO_HASH_MUL)  mod 2^64,  i = t * W + w over the frame's [S][W] history.
 * Every multiplier is odd, so a change of any single word changes the digest. */
#define VO_HASH_MUL 0x9E3779B97F4A7C15ull
uint64_t vo_hash_decisions(const uint64_t* words, size_t n);
int vo_decode_frames_hashed(const vo_params* p, const int16_t* table, const void* symbols, size_t frames, size_t L,
                            uint8_t* bytes_out, uint32_t* final_metrics, uint64_t* renorm_sum, uint64_t* decision_hash,
                            int threads);

#ifdef __cplusplus
}
#endif
#endif
