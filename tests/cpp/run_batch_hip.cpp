// run_batch_hip.cpp -- the batched route from C++: ViterbiDecoder_HIP_Batch (include/viterbi_hip/viterbi_decoder_hip_batch.h)
// decodes a few thousand noisy frames resident in device memory in one call and must agree, frame by frame, with the
// single-frame drop-in (ViterbiDecoder_Core + ViterbiDecoder_HIP) and with the oracle.  Also prints the batch throughput.
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "viterbi_hip/viterbi_decoder_core.h"
#include "viterbi_hip/viterbi_decoder_hip.h"
#include "viterbi_hip/viterbi_decoder_hip_batch.h"
#include "test_support.h"
#include "../../oracle/viterbi_oracle.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    constexpr size_t K = 7, R = 2;
    const uint8_t G[R] = {109, 79};
    const auto setup = soft16_setup(R);
    const size_t frames = 4099, L = 2048, S = L + K - 1, out_bytes = L / 8;   // odd frame count: ragged last tile

    XorShift rng(7);
    std::vector<int16_t> symbols(frames * S * R);
    std::vector<uint8_t> tx(frames * out_bytes);
    for (size_t f = 0; f < frames; f++) {
        std::vector<uint8_t> bytes(out_bytes);
        for (auto& b : bytes) b = uint8_t(rng.next());
        memcpy(&tx[f * out_bytes], bytes.data(), out_bytes);
        std::vector<int16_t> sym = encode_frame<int16_t>(K, R, G, bytes, setup.high, setup.low);
        for (auto& v : sym) {                      // ~10 % of the symbols weakened or flipped
            const uint32_t r = rng.next() % 100;
            if (r < 4) v = int16_t(-v); else if (r < 10) v = int16_t(v / 4);
        }
        memcpy(&symbols[f * S * R], sym.data(), S * R * sizeof(int16_t));
    }

    auto table = ViterbiBranchTable<K, R, int16_t>(G, setup.high, setup.low);
    ViterbiDecoder_HIP_Batch<K, R, uint16_t, int16_t> batch(table, setup.config);
    const size_t ws_bytes = batch.workspace_bytes(frames, L);
    int16_t* d_sym; void* d_ws; uint8_t* d_out; uint16_t* d_met; uint64_t* d_rs;
    HIP_OK(hipMalloc((void**)&d_sym, symbols.size() * sizeof(int16_t)));
    HIP_OK(hipMalloc(&d_ws, ws_bytes));
    HIP_OK(hipMalloc((void**)&d_out, frames * out_bytes));
    HIP_OK(hipMalloc((void**)&d_met, frames * 64 * sizeof(uint16_t)));
    HIP_OK(hipMalloc((void**)&d_rs, frames * sizeof(uint64_t)));
    HIP_OK(hipMemcpy(d_sym, symbols.data(), symbols.size() * sizeof(int16_t), hipMemcpyHostToDevice));

    batch.decode(d_sym, frames, L, d_ws, ws_bytes, d_out, d_met, d_rs);     // warm-up + the result we check
    HIP_OK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    const int reps = 20;
    for (int r = 0; r < reps; r++) batch.decode(d_sym, frames, L, d_ws, ws_bytes, d_out);
    HIP_OK(hipDeviceSynchronize());
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;

    // the double-buffered pipeline: same batches, chainback of one beside the update of the next; same bytes
    double sec_pipe = 0;
    {
        uint8_t* d_out2;
        HIP_OK(hipMalloc((void**)&d_out2, frames * out_bytes));
        ViterbiDecoder_HIP_Pipeline<K, R, uint16_t, int16_t> pipe(batch, frames, L);
        for (int r = 0; r < 3; r++) pipe.submit(d_sym, frames, d_out2);
        pipe.sync();
        const auto p0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) pipe.submit(d_sym, frames, d_out2);
        pipe.sync();
        sec_pipe = std::chrono::duration<double>(std::chrono::steady_clock::now() - p0).count() / reps;
        std::vector<uint8_t> a(frames * out_bytes), b(frames * out_bytes);
        HIP_OK(hipMemcpy(a.data(), d_out, a.size(), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(b.data(), d_out2, b.size(), hipMemcpyDeviceToHost));
        if (memcmp(a.data(), b.data(), a.size()) != 0) { printf("pipeline bytes differ from the serial decode\nFAIL\n"); return 1; }
        // producer stream -> pipeline -> consumer stream with no host synchronisation in between: the symbols are still being
        // copied into a fresh buffer when submit() is called (wait_event orders the update behind the copy), and the consumer's
        // copy of the bytes waits on done_event
        int16_t* d_sym2; uint8_t* d_out3;
        hipStream_t producer, consumer; hipEvent_t ready, done;
        HIP_OK(hipMalloc((void**)&d_sym2, frames * S * R * sizeof(int16_t)));
        HIP_OK(hipMalloc((void**)&d_out3, frames * out_bytes));
        HIP_OK(hipStreamCreateWithFlags(&producer, hipStreamNonBlocking)); HIP_OK(hipStreamCreateWithFlags(&consumer, hipStreamNonBlocking));
        HIP_OK(hipEventCreate(&ready)); HIP_OK(hipEventCreate(&done));
        HIP_OK(hipMemsetAsync(d_sym2, 0x55, frames * S * R * sizeof(int16_t), producer));
        HIP_OK(hipMemsetAsync(d_out2, 0xA5, frames * out_bytes, producer));
        HIP_OK(hipMemcpyAsync(d_sym2, d_sym, frames * S * R * sizeof(int16_t), hipMemcpyDeviceToDevice, producer));
        HIP_OK(hipEventRecord(ready, producer));
        pipe.wait_event(ready);
        pipe.submit(d_sym2, frames, d_out2, nullptr, done);
        HIP_OK(hipStreamWaitEvent(consumer, done, 0));
        HIP_OK(hipMemcpyAsync(d_out3, d_out2, frames * out_bytes, hipMemcpyDeviceToDevice, consumer));
        HIP_OK(hipStreamSynchronize(consumer));
        HIP_OK(hipMemcpy(b.data(), d_out3, b.size(), hipMemcpyDeviceToHost));
        if (memcmp(a.data(), b.data(), a.size()) != 0) { printf("pipeline ordered by events: bytes differ from the serial decode\nFAIL\n"); return 1; }
        pipe.sync();
        (void)hipEventDestroy(ready); (void)hipEventDestroy(done); (void)hipStreamDestroy(producer); (void)hipStreamDestroy(consumer);
        (void)hipFree(d_sym2); (void)hipFree(d_out3);
        (void)hipFree(d_out2);
    }

    // one decoder shared by two host threads, as the reference shares one branch table between the Cores of its worker
    // threads (run_benchmark.cpp:193-197): each thread decodes its half of the batch with its own workspace and stream, five
    // times over; the bytes must be those of the single call above (vit_hip.h, "threads")
    {
        const size_t half = frames / 2;
        std::atomic<int> fails{0};
        uint8_t* d_out3;
        if (hipMalloc((void**)&d_out3, frames * out_bytes) != hipSuccess) { printf("hipMalloc failed\nFAIL\n"); return 1; }
        auto worker = [&](size_t f0, size_t nf) {
            void* ws = nullptr; hipStream_t st = nullptr;
            const size_t wb = batch.workspace_bytes(nf, L);
            if (hipMalloc(&ws, wb) != hipSuccess || hipStreamCreate(&st) != hipSuccess) { fails++; return; }
            for (int r = 0; r < 5; r++) batch.decode(d_sym + f0 * S * R, nf, L, ws, wb, d_out3 + f0 * out_bytes, nullptr, nullptr, nullptr, st);
            if (hipStreamSynchronize(st) != hipSuccess) fails++;
            hipStreamDestroy(st); hipFree(ws);
        };
        std::thread ta(worker, size_t(0), half), tb(worker, half, frames - half);
        ta.join(); tb.join();
        std::vector<uint8_t> a(frames * out_bytes), b(frames * out_bytes);
        HIP_OK(hipMemcpy(a.data(), d_out, a.size(), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(b.data(), d_out3, b.size(), hipMemcpyDeviceToHost));
        if (fails.load() || memcmp(a.data(), b.data(), a.size()) != 0) { printf("two threads on one decoder: bytes differ from the single call\nFAIL\n"); return 1; }
        hipFree(d_out3);
    }

    std::vector<uint8_t> out(frames * out_bytes);
    std::vector<uint16_t> met(frames * 64);
    std::vector<uint64_t> rs(frames);
    HIP_OK(hipMemcpy(out.data(), d_out, out.size(), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(met.data(), d_met, met.size() * 2, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(rs.data(), d_rs, rs.size() * 8, hipMemcpyDeviceToHost));

    // checker 1: the oracle on every frame
    vo_params p = {int32_t(K), int32_t(R), 2, 2, setup.config.soft_decision_max_error, setup.config.initial_start_error,
                   setup.config.initial_non_start_error, setup.config.renormalisation_threshold};
    const uint32_t G32[R] = {109, 79};
    std::vector<int16_t> otable(R * 32);
    vo_branch_table(int(K), int(R), G32, setup.high, setup.low, otable.data());
    std::vector<uint8_t> want(frames * out_bytes);
    std::vector<uint32_t> want_met(frames * 64);
    std::vector<uint64_t> want_rs(frames);
    vo_decode_frames(&p, otable.data(), symbols.data(), frames, L, want.data(), want_met.data(), want_rs.data(), 8);
    size_t bad = 0;
    for (size_t f = 0; f < frames; f++) {
        bool ok = memcmp(&out[f * out_bytes], &want[f * out_bytes], out_bytes) == 0 && rs[f] == want_rs[f];
        for (size_t s = 0; s < 64 && ok; s++) ok = met[f * 64 + s] == want_met[f * 64 + s];
        bad += !ok;
    }
    // checker 2: the single-frame drop-in on a few frames
    ViterbiDecoder_Core<K, R, uint16_t, int16_t> vitdec(table, setup.config);
    vitdec.set_traceback_length(L);
    std::vector<uint8_t> rx(out_bytes);
    for (size_t f : {size_t(0), size_t(31), size_t(32), frames - 1}) {
        vitdec.reset();
        const uint64_t acc = ViterbiDecoder_HIP<K, R, uint16_t, int16_t>::update<uint64_t>(vitdec, &symbols[f * S * R], S * R);
        vitdec.chainback(rx.data(), L);
        bad += !(acc == rs[f] && memcmp(rx.data(), &out[f * out_bytes], out_bytes) == 0 && vitdec.get_error() == met[f * 64]);
    }
    size_t errs = 0;
    for (size_t i = 0; i < out.size(); i++) errs += size_t(__builtin_popcount(unsigned(out[i] ^ tx[i])));
    printf("frames=%zu bits/frame=%zu mismatching frames=%zu bit errors vs transmitted=%zu  batch decode %.3f ms = %.1f Gbit/s, pipelined %.3f ms = %.1f Gbit/s\n",
           frames, L, bad, errs, sec * 1e3, double(frames * L) / sec / 1e9, sec_pipe * 1e3, double(frames * L) / sec_pipe / 1e9);
    printf("%s\n", bad == 0 ? "PASS" : "FAIL");
    return bad == 0 ? 0 : 1;
}
