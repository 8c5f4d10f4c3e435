/*
 * viterbi_oracle.c -- plain-C restatement of the reference scalar decoder (see viterbi_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into, loaded by or called from the product path.
 *
 * The reference is a C++ template over <K, R, error_t, soft_t>; here the widths are run-time parameters and the
 * narrowing conversions the C++ types perform implicitly are written out:
 *   to_soft()  == conversion of an int expression to soft_t  (two's-complement truncation)
 *   to_err()   == conversion to error_t                      (reduction mod 2^(8*error_bytes))
 */
#include "viterbi_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static inline int32_t to_soft(int32_t v, int soft_bytes) {
    return soft_bytes == 1 ? (int32_t)(int8_t)(uint8_t)v : (int32_t)(int16_t)(uint16_t)v;
}

static inline uint32_t err_mask(int error_bytes) { return error_bytes == 1 ? 0xFFu : 0xFFFFu; }

static inline unsigned parity32(uint32_t x) {
    /* ParityTable::parse (parity_table.h:42-55): XOR-fold down to one bit */
    x ^= x >> 16;
    x ^= x >> 8;
    x ^= x >> 4;
    x ^= x >> 2;
    x ^= x >> 1;
    return x & 1u;
}

size_t vo_num_states(int K) { return (size_t)1 << (K - 1); }

size_t vo_decision_words(int K) {
    /* ViterbiDecisionBits::TOTAL_BLOCKS (viterbi_decoder_core.h:64-66) with uintptr_t == 64 bit (:168) */
    size_t n = vo_num_states(K) / 64;
    return n ? n : 1;
}

int vo_branch_table(int K, int R, const uint32_t* G, int soft_high, int soft_low, int16_t* table) {
    /* viterbi_branch_table.h:44-54: y = P{(0|X|0) & G[i]}, only the K-2 middle bits are enumerated */
    if (K < 2 || R < 1 || soft_high <= soft_low) return -1;
    const size_t H = vo_num_states(K) / 2;
    for (size_t s = 0; s < H; s++) {
        for (int i = 0; i < R; i++) {
            const uint32_t v = (uint32_t)(s << 1) & G[i];
            table[(size_t)i * H + s] = (int16_t)(parity32(v) ? soft_high : soft_low);
        }
    }
    return 0;
}

void vo_reset(const vo_params* p, uint32_t* metrics, size_t starting_state) {
    /* viterbi_decoder_core.h:202-211 */
    const size_t N = vo_num_states(p->K);
    for (size_t i = 0; i < N; i++) metrics[i] = p->initial_non_start_error;
    metrics[starting_state & (N - 1)] = p->initial_start_error;
}

static inline int32_t load_sym(const void* symbols, size_t idx, int soft_bytes) {
    return soft_bytes == 1 ? (int32_t)((const int8_t*)symbols)[idx] : (int32_t)((const int16_t*)symbols)[idx];
}

/* one trellis step: viterbi_decoder_scalar.h:58-136 */
static void vo_bfly(const vo_params* p, const int16_t* table, const void* symbols, size_t sym_base,
                    uint64_t* decision, const uint32_t* oldm, uint32_t* newm) {
    const size_t N = vo_num_states(p->K);
    const size_t H = N / 2;
    const size_t W = vo_decision_words(p->K);
    const uint32_t M = err_mask(p->error_bytes);
    const int sb = p->soft_bytes;

    for (size_t w = 0; w < W; w++) decision[w] = 0; /* :60-62 */

    for (size_t j = 0; j < H; j++) {
        uint32_t e = 0; /* error_t total_error (:66) */
        for (int i = 0; i < p->R; i++) {
            const int32_t sym = load_sym(symbols, sym_base + (size_t)i, sb);
            const int32_t expected = (int32_t)table[(size_t)i * H + j];
            const int32_t diff = to_soft(expected - sym, sb);          /* const soft_t error (:70) */
            const int32_t absd = to_soft(diff > 0 ? diff : -diff, sb); /* get_abs returns T (:155-159) */
            e = (e + ((uint32_t)absd & M)) & M;                        /* error_t(abs) ; += (:71-72) */
        }
        const uint32_t ebar = (p->soft_decision_max_error - e) & M; /* :107 */

        const uint32_t m00 = (oldm[j] + e) & M;        /* :113 */
        const uint32_t m10 = (oldm[j + H] + ebar) & M; /* :114 */
        const uint32_t m01 = (oldm[j] + ebar) & M;     /* :115 */
        const uint32_t m11 = (oldm[j + H] + e) & M;    /* :116 */

        const uint64_t d0 = m00 > m10; /* strict: tie keeps the r=0 predecessor (:123) */
        const uint64_t d1 = m01 > m11; /* :124 */

        newm[2 * j] = d0 ? m10 : m00;     /* :127 */
        newm[2 * j + 1] = d1 ? m11 : m01; /* :128 */

        const size_t s0 = 2 * j; /* :131-134 */
        decision[s0 / 64] |= (d0 | (d1 << 1)) << (s0 % 64);
    }
}

/* viterbi_decoder_scalar.h:139-153 */
static uint32_t vo_renormalise(uint32_t* m, size_t N) {
    uint32_t mn = m[0];
    for (size_t s = 1; s < N; s++)
        if (m[s] < mn) mn = m[s];
    for (size_t s = 0; s < N; s++) m[s] -= mn;
    return mn;
}

uint64_t vo_update(const vo_params* p, const int16_t* table, uint32_t* metrics, const void* symbols,
                   size_t n_steps, uint64_t* decisions) {
    /* viterbi_decoder_scalar.h:42-54: the double buffer + swap is modelled by ping-ponging two arrays */
    const size_t N = vo_num_states(p->K);
    const size_t W = vo_decision_words(p->K);
    uint32_t* scratch = (uint32_t*)malloc(N * sizeof(uint32_t));
    uint32_t* oldm = metrics;
    uint32_t* newm = scratch;
    uint64_t total = 0;
    for (size_t t = 0; t < n_steps; t++) {
        vo_bfly(p, table, symbols, t * (size_t)p->R, decisions + t * W, oldm, newm);
        if (newm[0] >= p->renormalisation_threshold) total += vo_renormalise(newm, N); /* :48-50 */
        uint32_t* tmp = oldm; /* swap (:51) */
        oldm = newm;
        newm = tmp;
    }
    if (oldm != metrics) memcpy(metrics, oldm, N * sizeof(uint32_t));
    free(scratch);
    return total;
}

void vo_chainback(int K, const uint64_t* decisions, size_t L, size_t end_state, uint8_t* out) {
    /* viterbi_decoder_core.h:214-236 with the shift register of ViterbiTracebackBuffer (:87-153) modelled exactly:
     * the register holds the K-1 state bits above `shift_state` padding bits; a decision bit is pushed in at the top
     * (push_bit_in :110-113), which is the predecessor map state' = (state >> 1) | (bit << (K-2)); get_data() (:105-107)
     * always exposes the register's top 8 bits, so the LAST write to out[j/8] (at j%8 == 0) holds bits j..j+7
     * MSB-first.  A trailing partial byte (L%8 != 0) therefore carries end-state bits in its low positions. */
    const size_t W = vo_decision_words(K);
    const size_t TSB = (size_t)K - 1;
    const size_t ignore = TSB < 8 ? TSB : 8;      /* get_layout :129-149 */
    const size_t shift_state = 8 - ignore;
    const size_t shift_tail = TSB - ignore;
    const size_t total_bits = TSB + shift_state;
    uint64_t reg = (uint64_t)end_state << shift_state; /* set_state :100-103 */
    for (size_t i = 0; i < L; i++) {
        const size_t j = L - 1 - i;
        const uint64_t* row = decisions + (j + TSB) * W;
        const uint64_t state = reg >> shift_state; /* get_state :95-97 */
        const uint64_t bit = (row[state / 64] >> (state % 64)) & 1u;
        reg = (reg >> 1) | (bit << (total_bits - 1));
        out[j / 8] = (uint8_t)((reg >> shift_tail) & 0xFFu);
    }
}

void vo_encode(int K, int R, const uint32_t* G, const uint8_t* bytes, size_t n_bytes, uint8_t* out_bits) {
    /* convolutional_encoder_shift_register.h:42-62 + test_helpers.h:46-60 */
    const uint32_t kmask = (K >= 32) ? 0xFFFFFFFFu : ((1u << K) - 1u);
    uint32_t reg = 0;
    size_t o = 0;
    const size_t total_bits = n_bytes * 8 + (size_t)(K - 1);
    for (size_t b = 0; b < total_bits; b++) {
        unsigned in = 0;
        if (b < n_bytes * 8) in = (bytes[b / 8] >> (7 - (b % 8))) & 1u;
        reg = (reg << 1) | in;
        for (int i = 0; i < R; i++) out_bits[o++] = (uint8_t)parity32((G[i] & kmask) & reg);
    }
}

/* ---- whole-frame driver ------------------------------------------------------------------------------------- */

typedef struct {
    const vo_params* p;
    const int16_t* table;
    const uint8_t* symbols;
    size_t f0, f1, L;
    uint8_t* bytes_out;
    uint32_t* final_metrics;
    uint64_t* renorm_sum;
    uint64_t* decision_hash;
} vo_job;

uint64_t vo_hash_decisions(const uint64_t* words, size_t n) {
    uint64_t h = 0;
    for (size_t i = 0; i < n; i++) h += words[i] * ((2u * (uint64_t)i + 1u) * VO_HASH_MUL);
    return h;
}

static void* vo_worker(void* arg) {
    vo_job* j = (vo_job*)arg;
    const vo_params* p = j->p;
    const size_t N = vo_num_states(p->K), W = vo_decision_words(p->K);
    const size_t S = j->L + (size_t)p->K - 1;
    const size_t frame_sym_bytes = S * (size_t)p->R * (size_t)p->soft_bytes;
    uint32_t* m = (uint32_t*)malloc(N * sizeof(uint32_t));
    uint64_t* dec = (uint64_t*)malloc(S * W * sizeof(uint64_t));
    for (size_t f = j->f0; f < j->f1; f++) {
        vo_reset(p, m, 0);
        const uint64_t rs = vo_update(p, j->table, m, j->symbols + f * frame_sym_bytes, S, dec);
        vo_chainback(p->K, dec, j->L, 0, j->bytes_out + f * ((j->L + 7) / 8));
        if (j->final_metrics) memcpy(j->final_metrics + f * N, m, N * sizeof(uint32_t));
        if (j->renorm_sum) j->renorm_sum[f] = rs;
        if (j->decision_hash) j->decision_hash[f] = vo_hash_decisions(dec, S * W);
    }
    free(dec);
    free(m);
    return NULL;
}

int vo_decode_frames(const vo_params* p, const int16_t* table, const void* symbols, size_t frames, size_t L,
                     uint8_t* bytes_out, uint32_t* final_metrics, uint64_t* renorm_sum, int threads) {
    return vo_decode_frames_hashed(p, table, symbols, frames, L, bytes_out, final_metrics, renorm_sum, NULL, threads);
}

int vo_decode_frames_hashed(const vo_params* p, const int16_t* table, const void* symbols, size_t frames, size_t L,
                            uint8_t* bytes_out, uint32_t* final_metrics, uint64_t* renorm_sum, uint64_t* decision_hash,
                            int threads) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > frames) threads = frames ? (int)frames : 1;
    pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
    vo_job* jobs = (vo_job*)malloc((size_t)threads * sizeof(vo_job));
    for (int t = 0; t < threads; t++) {
        jobs[t].p = p;
        jobs[t].table = table;
        jobs[t].symbols = (const uint8_t*)symbols;
        jobs[t].f0 = frames * (size_t)t / (size_t)threads;
        jobs[t].f1 = frames * (size_t)(t + 1) / (size_t)threads;
        jobs[t].L = L;
        jobs[t].bytes_out = bytes_out;
        jobs[t].final_metrics = final_metrics;
        jobs[t].renorm_sum = renorm_sum;
        jobs[t].decision_hash = decision_hash;
        if (threads == 1)
            vo_worker(&jobs[t]);
        else
            pthread_create(&th[t], NULL, vo_worker, &jobs[t]);
    }
    if (threads > 1)
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
    return 0;
}
