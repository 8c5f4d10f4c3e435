/*
 * ref_shim.cpp -- C-ABI doorway onto the REAL reference, compiled from the sources where they lie under
 * /root/reference (never copied into this repo).  Output: oracle/_ref/libvitref.so (git-ignored, travels to the GPU box).
 *
 * TEST INFRASTRUCTURE ONLY.  Used (a) to pin oracle/viterbi_oracle.c and to generate tests/golden/ fixtures in the
 * build container, (b) as bench.py's cpu_baseline of kind "reference" (the reference's own scalar / SSE / AVX2
 * strategies timed on the GPU box's host cores).  The product path never loads it.
 *
 * Everything here is glue: it instantiates the reference's templates
 *   ViterbiBranchTable<K,R,soft_t>            include/viterbi/viterbi_branch_table.h:20
 *   ViterbiDecoder_Core<K,R,error_t,soft_t>   include/viterbi/viterbi_decoder_core.h:157
 *   ViterbiDecoder_Scalar / _SSE_u16/_u8 / _AVX_u16/_u8 (via examples/helpers/simd_type.h:50-86)
 * for the eight stock codes (examples/helpers/common_codes.h:20-30) and drives them in the call pattern of
 * examples/run_simple.cpp:76-80 / examples/run_benchmark.cpp:268-281.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <chrono>
#include <pthread.h>
#include <sched.h>
#include <thread>
#include <vector>
#include <memory>
#include <algorithm>

#include "viterbi/convolutional_encoder.h"
#include "viterbi/convolutional_encoder_shift_register.h"
#include "viterbi/convolutional_encoder_lookup.h"
#include "viterbi/viterbi_decoder_core.h"
#include "helpers/common_codes.h"
#include "helpers/simd_type.h"
#include "helpers/decode_type.h"

namespace {

struct RunArgs {
    int simd;
    int high, low;
    const uint32_t* cfg;
    const void* symbols;
    size_t n_steps, L, start_state, end_state, chunk_steps;
    uint64_t* decisions;
    uint32_t* metrics;
    uint64_t* renorm_sum;
    uint32_t* error_at_end;
    uint8_t* bytes_out;
};

struct BenchArgs {
    int simd;
    int high, low;
    const uint32_t* cfg;
    const void* symbols;
    size_t frames, L;
    int threads, passes, sweeps;
    const int* cpus;         // [threads] logical CPU each worker is pinned to, or null (unpinned)
    uint8_t* bytes_out;
    double* pass_seconds;    // [passes]
};

template <typename error_t>
ViterbiDecoder_Config<error_t> make_config(const uint32_t* cfg) {
    ViterbiDecoder_Config<error_t> c;
    c.soft_decision_max_error = error_t(cfg[0]);
    c.initial_start_error = error_t(cfg[1]);
    c.initial_non_start_error = error_t(cfg[2]);
    c.renormalisation_threshold = error_t(cfg[3]);
    return c;
}

template <class decoder_t, size_t K, size_t R, typename error_t, typename soft_t>
int run_one(ViterbiDecoder_Core<K,R,error_t,soft_t>& dec, const RunArgs& a) {
    using Core = ViterbiDecoder_Core<K,R,error_t,soft_t>;
    const soft_t* sym = reinterpret_cast<const soft_t*>(a.symbols);
    dec.set_traceback_length(a.L);
    dec.reset(a.start_state);
    uint64_t acc = 0;
    const size_t chunk = a.chunk_steps ? a.chunk_steps : a.n_steps;
    for (size_t t = 0; t < a.n_steps; t += chunk) {
        const size_t n = std::min(chunk, a.n_steps - t);
        acc += decoder_t::template update<uint64_t>(dec, sym + t*R, n*R);
    }
    if (a.renorm_sum) *a.renorm_sum = acc;
    if (a.error_at_end) *a.error_at_end = uint32_t(dec.get_error(a.end_state));
    if (a.metrics) {
        auto* m = dec.m_metrics.get_old();
        for (size_t s = 0; s < Core::NUMSTATES; s++) a.metrics[s] = uint32_t(m[s]);
    }
    if (a.decisions) {
        constexpr size_t W = Core::Decisions::TOTAL_BLOCKS;
        for (size_t t = 0; t < a.n_steps; t++) {
            auto* row = dec.m_decisions[t];
            for (size_t w = 0; w < W; w++) a.decisions[t*W + w] = uint64_t(row[w]);
        }
    }
    if (a.bytes_out) dec.chainback(a.bytes_out, a.L, a.end_state);
    return 0;
}

// Timing protocol of bench.py's cpu_baseline: the T workers are created ONCE, pinned (one per physical core of one
// socket, chosen by the caller) and parked on a barrier; a pass is the interval between the release of the start barrier
// and the arrival of the last worker at the end barrier, during which every worker decodes its own contiguous share of the
// frames `sweeps` times (reset -> update -> chainback per frame, one decoder per thread, one shared branch table:
// examples/run_benchmark.cpp:193-197, :268-281).  Pass 0 is an untimed warm-up.  No thread is created or joined inside
// a timed pass.
template <class decoder_t, size_t K, size_t R, typename error_t, typename soft_t>
void bench_one(const ViterbiBranchTable<K,R,soft_t>& table, const ViterbiDecoder_Config<error_t>& cfg, BenchArgs& a) {
    using Core = ViterbiDecoder_Core<K,R,error_t,soft_t>;
    const size_t S = a.L + K - 1;
    const size_t frame_syms = S*R;
    const size_t out_bytes = (a.L + 7)/8;
    const int T = std::max(1, a.threads);
    const int passes = std::max(1, a.passes), sweeps = std::max(1, a.sweeps);
    std::vector<std::unique_ptr<Core>> cores;
    for (int t = 0; t < T; t++) {
        cores.emplace_back(new Core(table, cfg));
        cores.back()->set_traceback_length(a.L);
    }
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, unsigned(T) + 1u);
    auto worker = [&](int t) {
        if (a.cpus) {
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(a.cpus[t], &set);
            (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
        }
        Core& dec = *cores[t];
        const soft_t* sym = reinterpret_cast<const soft_t*>(a.symbols);
        const size_t f0 = a.frames*size_t(t)/size_t(T), f1 = a.frames*size_t(t+1)/size_t(T);
        for (int p = 0; p <= passes; p++) {
            pthread_barrier_wait(&bar);
            for (int s = 0; s < sweeps; s++)
                for (size_t f = f0; f < f1; f++) {
                    dec.reset();
                    (void)decoder_t::template update<uint64_t>(dec, sym + f*frame_syms, frame_syms);
                    dec.chainback(a.bytes_out + f*out_bytes, a.L);
                }
            pthread_barrier_wait(&bar);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
    for (int p = 0; p <= passes; p++) {
        pthread_barrier_wait(&bar);
        const auto t0 = std::chrono::steady_clock::now();
        pthread_barrier_wait(&bar);
        const auto t1 = std::chrono::steady_clock::now();
        if (p > 0) a.pass_seconds[p - 1] = std::chrono::duration<double>(t1 - t0).count();
    }
    for (auto& x: th) x.join();
    pthread_barrier_destroy(&bar);
}

enum Op { OP_RUN, OP_BENCH, OP_VALID, OP_TABLE };

struct Call {
    Op op;
    int bytes;     // 2: (uint16_t, int16_t)   1: (uint8_t, int8_t)
    RunArgs* run;
    BenchArgs* bench;
    int16_t* table_out;
    int result;
};

template <typename E, typename S> struct TypeTag { using error_t = E; using soft_t = S; };
struct CfgArgs { int high, low; const uint32_t* cfgv; };

// only `decoder_t` is spelled explicitly at the call site (the reference's SELECT_FACTORY_ITEM macro cannot carry
// template-argument commas); everything else is deduced.
template <class decoder_t, size_t K, size_t R, typename code_t, typename error_t, typename soft_t>
void dispatch_decoder(const Code<K,R,code_t>& code, Call& c, TypeTag<error_t,soft_t>, CfgArgs ca) {
    const int high = ca.high, low = ca.low; const uint32_t* cfgv = ca.cfgv;
    if constexpr (decoder_t::is_valid) {
        if (c.op == OP_VALID) { c.result = 1; return; }
        auto table = ViterbiBranchTable<K,R,soft_t>(code.G.data(), soft_t(high), soft_t(low));
        const auto cfg = make_config<error_t>(cfgv);
        if (c.op == OP_RUN) {
            using Core = ViterbiDecoder_Core<K,R,error_t,soft_t>;
            auto dec = std::make_unique<Core>(table, cfg);
            c.result = run_one<decoder_t>(*dec, *c.run);
        } else if (c.op == OP_BENCH) {
            bench_one<decoder_t>(table, cfg, *c.bench);
            c.result = 0;
        }
    } else {
        c.result = (c.op == OP_VALID) ? 0 : -2;
    }
}

template <class factory_t, size_t K, size_t R, typename code_t, typename error_t, typename soft_t>
void dispatch_simd(const Code<K,R,code_t>& code, Call& c, int simd, int high, int low, const uint32_t* cfgv) {
    c.result = -2;   // "strategy not compiled in"
    const TypeTag<error_t,soft_t> tag;
    const CfgArgs ca{high, low, cfgv};
    SELECT_FACTORY_ITEM(factory_t, SIMD_Type(simd), K, R, {
        dispatch_decoder<it>(code, c, tag, ca);
    });
}

template <size_t K, size_t R, typename code_t>
void dispatch_types(const Code<K,R,code_t>& code, Call& c) {
    int simd = 0, high = 1, low = -1;
    const uint32_t* cfgv = nullptr;
    if (c.op == OP_RUN)   { simd = c.run->simd;   high = c.run->high;   low = c.run->low;   cfgv = c.run->cfg; }
    if (c.op == OP_BENCH) { simd = c.bench->simd; high = c.bench->high; low = c.bench->low; cfgv = c.bench->cfg; }
    if (c.op == OP_VALID) { simd = c.run->simd; }
    if (c.op == OP_TABLE) {
        high = c.run->high; low = c.run->low;
        constexpr size_t H = (size_t(1) << (K-1))/2;
        if (c.bytes == 2) {
            auto table = ViterbiBranchTable<K,R,int16_t>(code.G.data(), int16_t(high), int16_t(low));
            for (size_t i = 0; i < R; i++) for (size_t s = 0; s < H; s++) c.table_out[i*H+s] = int16_t(table[i][s]);
        } else {
            auto table = ViterbiBranchTable<K,R,int8_t>(code.G.data(), int8_t(high), int8_t(low));
            for (size_t i = 0; i < R; i++) for (size_t s = 0; s < H; s++) c.table_out[i*H+s] = int16_t(table[i][s]);
        }
        c.result = 0;
        return;
    }
    if (c.bytes == 2) dispatch_simd<ViterbiDecoder_Factory_u16, K, R, code_t, uint16_t, int16_t>(code, c, simd, high, low, cfgv);
    else              dispatch_simd<ViterbiDecoder_Factory_u8,  K, R, code_t, uint8_t,  int8_t >(code, c, simd, high, low, cfgv);
}

void dispatch_code(int code_id, Call& c) {
    c.result = -1;
    SELECT_COMMON_CODES(code_id, { dispatch_types(it, c); });
}

} // namespace

extern "C" {

int vitref_num_codes() { return int(COMMON_CODES.N); }

int vitref_stock_code(int code_id, int* K, int* R, uint32_t* G, char* name, size_t name_cap) {
    int ok = -1;
    SELECT_COMMON_CODES(code_id, {
        *K = int(it.K); *R = int(it.R);
        for (size_t i = 0; i < it.R; i++) G[i] = uint32_t(it.G[i]);
        if (name && name_cap) { strncpy(name, it.name, name_cap-1); name[name_cap-1] = 0; }
        ok = 0;
    });
    return ok;
}

/* decode_type: 0 SOFT16, 1 SOFT8, 2 HARD8 (examples/helpers/decode_type.h:17-64) */
int vitref_stock_config(int decode_type, int R, int* high, int* low, uint32_t* cfg) {
    if (decode_type == 0) {
        const auto c = get_soft16_decoding_config(size_t(R));
        *high = c.soft_decision_high; *low = c.soft_decision_low;
        cfg[0] = c.decoder_config.soft_decision_max_error; cfg[1] = c.decoder_config.initial_start_error;
        cfg[2] = c.decoder_config.initial_non_start_error; cfg[3] = c.decoder_config.renormalisation_threshold;
        return 2;
    }
    const auto c = (decode_type == 1) ? get_soft8_decoding_config(size_t(R)) : get_hard8_decoding_config(size_t(R));
    *high = c.soft_decision_high; *low = c.soft_decision_low;
    cfg[0] = c.decoder_config.soft_decision_max_error; cfg[1] = c.decoder_config.initial_start_error;
    cfg[2] = c.decoder_config.initial_non_start_error; cfg[3] = c.decoder_config.renormalisation_threshold;
    return 1;
}

/* simd: 0 SCALAR, 1 SIMD_SSE, 2 SIMD_AVX.  returns 1 valid, 0 not valid for this K, <0 unknown */
int vitref_is_valid(int code_id, int bytes, int simd) {
    RunArgs ra{}; ra.simd = simd;
    Call c{OP_VALID, bytes, &ra, nullptr, nullptr, -1};
    dispatch_code(code_id, c);
    return c.result == -2 ? 0 : c.result;
}

int vitref_branch_table(int code_id, int bytes, int high, int low, int16_t* out) {
    RunArgs ra{}; ra.high = high; ra.low = low;
    Call c{OP_TABLE, bytes, &ra, nullptr, out, -1};
    dispatch_code(code_id, c);
    return c.result;
}

int vitref_run(int code_id, int bytes, int simd, int high, int low, const uint32_t* cfg,
               const void* symbols, size_t n_steps, size_t L, size_t start_state, size_t end_state, size_t chunk_steps,
               uint64_t* decisions, uint32_t* metrics, uint64_t* renorm_sum, uint32_t* error_at_end, uint8_t* bytes_out) {
    RunArgs ra{simd, high, low, cfg, symbols, n_steps, L, start_state, end_state, chunk_steps,
               decisions, metrics, renorm_sum, error_at_end, bytes_out};
    Call c{OP_RUN, bytes, &ra, nullptr, nullptr, -1};
    dispatch_code(code_id, c);
    return c.result;
}

/* which: 0 ConvolutionalEncoder_ShiftRegister, 1 ConvolutionalEncoder_Lookup; output one byte (0/1) per symbol,
 * (8*n_bytes + K-1)*R of them, in the order examples/helpers/test_helpers.h:17-64 (encode_data) produces. */
int vitref_encode(int code_id, int which, const uint8_t* bytes, size_t n_bytes, uint8_t* out_bits) {
    int ok = -1;
    SELECT_COMMON_CODES(code_id, {
        std::unique_ptr<ConvolutionalEncoder> enc;
        if (which == 0) enc.reset(new ConvolutionalEncoder_ShiftRegister<uint32_t>(it.K, it.R, it.G.data()));
        else            enc.reset(new ConvolutionalEncoder_Lookup(it.K, it.R, it.G.data()));
        enc->reset();
        const size_t K = it.K; const size_t R = it.R;
        std::vector<uint8_t> y(R);
        size_t o = 0;
        auto push = [&](size_t total_bits) {
            for (size_t i = 0; i < total_bits; i++) out_bits[o++] = (y[i/8] >> (i%8)) & 1u;
        };
        for (size_t i = 0; i < n_bytes; i++) { enc->consume_byte(bytes[i], y.data()); push(8*R); }
        for (size_t i = 0; i < K-1; ) {
            const size_t n = std::min(K-1-i, size_t(8));
            enc->consume_byte(0x00, y.data());
            push(n*R);
            i += n;
        }
        ok = 0;
    });
    return ok;
}

/* cpu_baseline timing (see bench_one): `passes` timed passes (after one warm-up pass) over all frames on `threads` persistent
 * host threads, each frame decoded `sweeps` times per pass; pass_seconds[passes] receives the wall time of every pass.
 * cpus: [threads] logical CPU per worker or NULL.  symbols: [frames][L+K-1][R] soft_t.  returns 0, <0 on error. */
int vitref_bench(int code_id, int bytes, int simd, int high, int low, const uint32_t* cfg,
                 const void* symbols, size_t frames, size_t L, int threads, int passes, int sweeps, const int* cpus,
                 uint8_t* bytes_out, double* pass_seconds) {
    BenchArgs ba{simd, high, low, cfg, symbols, frames, L, threads, passes, sweeps, cpus, bytes_out, pass_seconds};
    Call c{OP_BENCH, bytes, nullptr, &ba, nullptr, -1};
    dispatch_code(code_id, c);
    return c.result;
}

} // extern "C"
