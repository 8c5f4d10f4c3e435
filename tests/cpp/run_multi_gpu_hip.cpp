// run_multi_gpu_hip.cpp -- the multi-GPU path from C++, no Python anywhere: one host thread per GPU of the node (the
// reference runs one decoder per thread over a shared branch table: examples/run_benchmark.cpp:193-197), an RCCL communicator
// over all of them, ONE collective -- rank 0's branch table + config broadcast over xGMI (vit_hip_broadcast_table) -- and then
// every GPU decodes its own contiguous shard of one global batch of frames with no further exchange.
//   usage: run_multi_gpu_hip [frames_per_gpu] [bits] [steps]     (uses every visible GPU; on a one-GPU box the communicator
//   has one rank and the same code runs).  The timed part is `steps` batches through the shipped pipeline API
//   (ViterbiDecoder_HIP_Pipeline = vit_hip_pipeline_*) per rank; the last line is ONE JSON record in bench.py's schema (value =
//   aggregate decoded Mbit/s over all GPUs, per-rank rates beside it), so that a node run yields a scaling point from C++ too.
// Checks: every rank's received table equals rank 0's; noise-free shards decode to exactly what was sent; a noisy shard
// decodes with a BER in range; shards are disjoint parts of ONE batch (the generator is keyed by the global frame index).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

#include "viterbi_hip/host_affinity.h"
#include "viterbi_hip/viterbi_decoder_hip_batch.h"
#include "test_support.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr size_t K = 7, R = 2;
using Batch = ViterbiDecoder_HIP_Batch<K, R, uint16_t, int16_t>;

struct RankResult {
    int rc = 1;
    bool table_ok = false, clean_ok = false;
    uint64_t noisy_errors = 0;
    double decode_ms = 0;          // wall time of the timed pipeline region (all `steps` batches)
    uint8_t first_tx_byte = 0;
    viterbi_hip::AffinityResult affinity;   // where the rank's host thread was pinned (the CPUs local to its GPU)
};

static int rank_main(int rank, int nranks, ncclComm_t comm, size_t frames, size_t L, int steps, RankResult* res) {
    res->affinity = viterbi_hip::pin_thread_to_gpu_cpus(rank);      // before this thread's first GPU call, like bench.py's ranks
    HIP_OK(hipSetDevice(rank));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    const auto setup = soft16_setup(R);
    const uint8_t G[R] = {109, 79};
    const uint8_t G_other[R] = {0x5B, 0x79};   // what a non-root rank starts with: overwritten by the broadcast
    auto table = ViterbiBranchTable<K, R, int16_t>(rank == 0 ? G : G_other, rank == 0 ? setup.high : int16_t(1),
                                                   rank == 0 ? setup.low : int16_t(-1));
    ViterbiDecoder_Config<uint16_t> config = setup.config;
    if (rank != 0) memset(&config, 0, sizeof(config));
    Batch::broadcast_table(comm, 0, rank, table, config, rank, st);
    const auto want = ViterbiBranchTable<K, R, int16_t>(G, setup.high, setup.low);
    res->table_ok = memcmp(table.data(), want.data(), R * 32 * sizeof(int16_t)) == 0 &&
                    memcmp(&config, &setup.config, sizeof(config)) == 0 &&
                    table.soft_decision_high() == setup.high && table.soft_decision_low() == setup.low;

    Batch dec(table, config, rank);
    const size_t S = L + K - 1, out_bytes = L / 8, ws_bytes = dec.workspace_bytes(frames, L);
    int16_t* d_sym; void* d_ws; uint8_t *d_tx, *d_out; uint64_t* d_cnt;
    HIP_OK(hipMalloc((void**)&d_sym, frames * S * R * sizeof(int16_t)));
    HIP_OK(hipMalloc(&d_ws, ws_bytes));
    HIP_OK(hipMalloc((void**)&d_tx, frames * out_bytes));
    HIP_OK(hipMalloc((void**)&d_out, frames * out_bytes));
    HIP_OK(hipMalloc((void**)&d_cnt, 2 * sizeof(uint64_t)));
    HIP_OK(hipMemsetAsync(d_cnt, 0, 2 * sizeof(uint64_t), st));
    const uint64_t first_frame = uint64_t(rank) * frames;        // this rank's shard of the global batch
    // noise-free shard: must decode exactly
    dec.synth(frames, L, 42, first_frame, 0.f, true, d_tx, d_sym, st);
    dec.decode(d_sym, frames, L, d_ws, ws_bytes, d_out, nullptr, nullptr, nullptr, st);
    dec.count_bit_errors(d_out, d_tx, frames * out_bytes, d_cnt, st);
    // noisy shard: `steps` batches through the pipeline, timed from the first submit to the last completion (2 warm-up batches)
    dec.synth(frames, L, 43, first_frame, 3.0f, false, d_tx, d_sym, st);
    HIP_OK(hipStreamSynchronize(st));
    {
        ViterbiDecoder_HIP_Pipeline<K, R, uint16_t, int16_t> pipe(dec, frames, L);
        for (int k = 0; k < 2; k++) pipe.submit(d_sym, frames, d_out);
        pipe.sync();
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < steps; k++) pipe.submit(d_sym, frames, d_out);
        pipe.sync();
        res->decode_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
    }
    dec.count_bit_errors(d_out, d_tx, frames * out_bytes, d_cnt + 1, st);
    uint64_t cnt[2];
    HIP_OK(hipMemcpyAsync(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&res->first_tx_byte, d_tx, 1, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    res->clean_ok = cnt[0] == 0;
    res->noisy_errors = cnt[1];
    hipFree(d_sym); hipFree(d_ws); hipFree(d_tx); hipFree(d_out); hipFree(d_cnt);
    hipStreamDestroy(st);
    (void)nranks;
    res->rc = 0;
    return 0;
}

int main(int argc, char** argv) {
    const size_t frames = argc > 1 ? size_t(atol(argv[1])) : 8192, L = argc > 2 ? size_t(atol(argv[2])) : 2048;
    const int steps = argc > 3 ? atoi(argv[3]) : 10;
    int n = 0;
    HIP_OK(hipGetDeviceCount(&n));
    if (n < 1) { printf("no GPU\n"); return 1; }
    std::vector<ncclComm_t> comms(n);
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) devs[i] = i;
    if (ncclCommInitAll(comms.data(), n, devs.data()) != ncclSuccess) { printf("ncclCommInitAll failed\n"); return 1; }
    std::vector<RankResult> res(n);
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++) th.emplace_back(rank_main, r, n, comms[r], frames, L, steps, &res[r]);
    for (auto& t : th) t.join();
    for (int r = 0; r < n; r++) ncclCommDestroy(comms[r]);
    bool ok = true;
    double max_ms = 0;
    for (int r = 0; r < n; r++) {
        const double ber = double(res[r].noisy_errors) / double(frames * L);
        printf("rank %d: table %s, noise-free shard %s, noisy shard BER %.2e, %d batches in %.3f ms = %.1f Mbit/s; host thread %s to %d CPUs of NUMA node %d%s%s\n", r,
               res[r].table_ok ? "ok" : "BAD", res[r].clean_ok ? "exact" : "WRONG", ber, steps, res[r].decode_ms,
               double(frames * L) * steps / res[r].decode_ms / 1e3, res[r].affinity.pinned ? "pinned" : "NOT pinned", res[r].affinity.cpus_used,
               res[r].affinity.numa_node, res[r].affinity.error.empty() ? "" : ": ", res[r].affinity.error.c_str());
        ok = ok && res[r].rc == 0 && res[r].table_ok && res[r].clean_ok && ber > 0 && ber < 2e-3;
        max_ms = res[r].decode_ms > max_ms ? res[r].decode_ms : max_ms;
    }
    // one record in bench.py's schema: whole-job throughput over the slowest rank's wall time (ranks start together: one thread each)
    const double value = double(frames * L) * steps * n / max_ms / 1e3;
    printf("{\"metric\": \"decoded Mbit/s (= ACS trellis steps/s), update()+chainback()\", \"value\": %.1f, \"unit\": \"Mbit/s\", "
           "\"n_gpus\": %d, \"steps\": %d, \"warmup\": 2, \"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": \"weak\", "
           "\"vs_baseline\": null, \"dtype\": \"u16\", \"data\": \"synthetic\", \"config\": {\"workload\": \"Voyager K=7 R=1/2 SOFT16, "
           "%zu frames x %zu info bits per GPU, AWGN Eb/N0=3 dB\", \"host\": \"tests/cpp/run_multi_gpu_hip (C++, one thread per GPU, "
           "RCCL broadcast of the table, vit_hip_pipeline_*)\"}, \"per_rank_Mbit_s\": [", value, n, steps, max_ms / steps, frames, L);
    for (int r = 0; r < n; r++) printf("%s%.1f", r ? ", " : "", double(frames * L) * steps / res[r].decode_ms / 1e3);
    printf("]}\n");
    printf("%s\n", ok ? "PASS" : "FAIL");
    return ok ? 0 : 1;
}
