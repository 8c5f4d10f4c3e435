/* oracle_sanitize.c -- the C restatement (oracle/viterbi_oracle.c) under AddressSanitizer + UBSan on the CPU build
 * (GPU sanitizers are not available on the pool).  Built and run by tests/test_oracle.py; exit code 0 = clean.
 * Exercises every entry point at the extremes the GPU tests use: K = 2 (two states, one word with 62 spare bits), K = 7
 * (exactly one word), K = 9 / 11 (several words), both widths, thresholds 0 and type-max, ragged L, chunked update,
 * non-zero start and end states, and the threaded whole-frame driver with its decision digests. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "viterbi_oracle.h"

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static int run_case(int K, int R, int bytes, uint32_t thr, size_t L, int full_range) {
    uint32_t G[8];
    for (int i = 0; i < R; i++) G[i] = (rnd() | 1u | (1u << (K - 1))) & ((1u << K) - 1u);
    const int high = bytes == 2 ? 127 : (full_range ? 127 : 1), low = -high;
    vo_params p = {K, R, bytes, bytes, (uint32_t)((high - low) * R), 0u, (uint32_t)((high - low) * R * 3), thr};
    const size_t N = vo_num_states(K), W = vo_decision_words(K), S = L + (size_t)K - 1, H = N / 2 ? N / 2 : 1;
    int16_t* table = (int16_t*)malloc((size_t)R * H * sizeof(int16_t));
    vo_branch_table(K, R, G, high, low, table);
    const size_t frames = 5;
    void* sym = malloc(frames * S * (size_t)R * (size_t)bytes);
    for (size_t i = 0; i < frames * S * (size_t)R; i++) {
        int v = full_range ? (int)(rnd() % (bytes == 2 ? 65536u : 256u)) - (bytes == 2 ? 32768 : 128)
                           : low + (int)(rnd() % (unsigned)(high - low + 1));
        if (bytes == 2) ((int16_t*)sym)[i] = (int16_t)v; else ((int8_t*)sym)[i] = (int8_t)v;
    }
    /* one shot vs three chunks, random start / end state */
    uint32_t* m1 = (uint32_t*)malloc(N * sizeof(uint32_t));
    uint32_t* m2 = (uint32_t*)malloc(N * sizeof(uint32_t));
    uint64_t* d1 = (uint64_t*)malloc(S * W * sizeof(uint64_t));
    uint64_t* d2 = (uint64_t*)malloc(S * W * sizeof(uint64_t));
    const size_t start = rnd() % N, end = rnd() % N;
    vo_reset(&p, m1, start);
    vo_reset(&p, m2, start);
    const uint64_t r1 = vo_update(&p, table, m1, sym, S, d1);
    const size_t c0 = S / 3, c1 = S / 2;
    const size_t step_bytes = (size_t)R * (size_t)bytes;
    uint64_t r2 = vo_update(&p, table, m2, sym, c0, d2);
    r2 += vo_update(&p, table, m2, (const char*)sym + c0 * step_bytes, c1 - c0, d2 + c0 * W);
    r2 += vo_update(&p, table, m2, (const char*)sym + c1 * step_bytes, S - c1, d2 + c1 * W);
    int bad = r1 != r2 || memcmp(m1, m2, N * sizeof(uint32_t)) || memcmp(d1, d2, S * W * sizeof(uint64_t));
    uint8_t* o1 = (uint8_t*)malloc((L + 7) / 8 + 1);
    o1[(L + 7) / 8] = 0xA5;                                   /* canary right behind the output */
    vo_chainback(K, d1, L, end, o1);
    bad |= o1[(L + 7) / 8] != 0xA5;
    /* threaded driver + digests */
    uint8_t* out = (uint8_t*)malloc(frames * ((L + 7) / 8));
    uint32_t* fm = (uint32_t*)malloc(frames * N * sizeof(uint32_t));
    uint64_t rs[5], hs[5];
    vo_decode_frames_hashed(&p, table, sym, frames, L, out, fm, rs, hs, 3);
    if (start == 0) bad |= rs[0] != r1 || hs[0] != vo_hash_decisions(d1, S * W);
    free(out); free(fm); free(o1); free(d2); free(d1); free(m2); free(m1); free(sym); free(table);
    return bad;
}

int main(void) {
    int bad = 0, n = 0;
    const int Ks[] = {2, 3, 5, 7, 8, 9, 11};
    for (size_t k = 0; k < sizeof(Ks) / sizeof(Ks[0]); k++)
        for (int R = 1; R <= 6; R += (Ks[k] > 8 ? 5 : 1))
            for (int bytes = 1; bytes <= 2; bytes++)
                for (int t = 0; t < 3; t++) {
                    const uint32_t tmax = bytes == 2 ? 65535u : 255u;
                    const uint32_t thr = t == 0 ? 0u : t == 1 ? tmax : tmax - (uint32_t)(bytes == 2 ? 254 : 2) * (uint32_t)R * 3u;
                    const size_t L = (size_t)(1 + rnd() % 300);
                    bad |= run_case(Ks[k], R, bytes, thr, L, t == 1);
                    n++;
                }
    printf("oracle_sanitize: %d cases, %s\n", n, bad ? "MISMATCH" : "ok");
    return bad ? 1 : 0;
}
