// viterbi_hip/viterbi_decoder_hip_batch.h -- RAII wrapper over the batched C ABI (include/vit_hip.h): many independent
// frames, buffers resident in HBM, explicit stream.  Built from the same host types as the single-frame drop-in
// (ViterbiBranchTable, ViterbiDecoder_Config), so a program moves from
//     for each frame: vitdec.reset(); Decoder::update(vitdec, ...); vitdec.chainback(...)      (examples/run_benchmark.cpp:268-281)
// to one call per batch without touching how the code or the configuration is described.  No HIP header is needed here:
// device pointers and the stream are passed as plain pointers; allocation stays with the caller.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../vit_hip.h"
#include "viterbi_branch_table.h"
#include "viterbi_decoder_config.h"

template <size_t constraint_length, size_t code_rate, typename error_t, typename soft_t>
class ViterbiDecoder_HIP_Batch {
public:
    static constexpr size_t K = constraint_length, R = code_rate;
    static constexpr size_t NUMSTATES = size_t(1) << (K - 1);
    using BranchTable = ViterbiBranchTable<K, R, soft_t>;
    using Config = ViterbiDecoder_Config<error_t>;

    ViterbiDecoder_HIP_Batch(const BranchTable& table, const Config& config, int device = 0) {
        check(vit_hip_create(int(K), int(R), int(sizeof(soft_t)), int(sizeof(error_t)), table.data(), &config, device, &m_hip),
              "vit_hip_create");
    }
    ~ViterbiDecoder_HIP_Batch() { vit_hip_destroy(m_hip); }
    ViterbiDecoder_HIP_Batch(const ViterbiDecoder_HIP_Batch&) = delete;
    ViterbiDecoder_HIP_Batch& operator=(const ViterbiDecoder_HIP_Batch&) = delete;

    // bytes of device workspace (256-byte aligned) the decision history of `frames` frames of `total_bits` bits needs
    size_t workspace_bytes(size_t frames, size_t total_bits) const { return vit_hip_workspace_bytes(m_hip, frames, total_bits); }
    static size_t symbols_per_frame(size_t total_bits) { return (total_bits + K - 1) * R; }

    // reset -> update -> chainback for every frame; everything is enqueued on `stream` (a hipStream_t), nothing synchronises
    void decode(const soft_t* d_symbols, size_t frames, size_t total_bits, void* d_workspace, size_t workspace_size,
                uint8_t* d_bytes_out, error_t* d_final_metrics = nullptr, uint64_t* d_renorm_sum = nullptr,
                const uint32_t* d_end_state = nullptr, void* stream = nullptr) {
        check(vit_hip_decode_batch(m_hip, d_symbols, frames, total_bits, d_workspace, workspace_size, d_bytes_out,
                                   d_final_metrics, d_renorm_sum, d_end_state, stream), "vit_hip_decode_batch");
    }
    // the two phases separately, as the reference times them (run_benchmark.cpp:272-281)
    void update(const soft_t* d_symbols, size_t frames, size_t total_bits, void* d_workspace, size_t workspace_size,
                error_t* d_final_metrics = nullptr, uint64_t* d_renorm_sum = nullptr, const uint32_t* d_start_state = nullptr,
                void* stream = nullptr) {
        check(vit_hip_update_batch(m_hip, d_symbols, frames, total_bits + K - 1, total_bits, d_workspace, workspace_size,
                                   d_final_metrics, d_renorm_sum, d_start_state, stream), "vit_hip_update_batch");
    }
    void chainback(const void* d_workspace, size_t frames, size_t total_bits, uint8_t* d_bytes_out,
                   const uint32_t* d_end_state = nullptr, void* stream = nullptr) {
        check(vit_hip_chainback_batch(m_hip, d_workspace, frames, total_bits, d_bytes_out, d_end_state, stream),
              "vit_hip_chainback_batch");
    }
    // depuncturing front-end (examples/helpers/puncture_code_helpers.h:17-55 for a batch): d_source_index[k] is the position of
    // mother-code symbol k among the `punctured_per_frame` transmitted ones, or < 0 for a punctured symbol (erasure value 0)
    void depuncture(const soft_t* d_punctured, size_t punctured_per_frame, const int32_t* d_source_index, size_t frames,
                    size_t total_bits, soft_t* d_symbols_out, void* stream = nullptr) {
        check(vit_hip_depuncture_batch(m_hip, d_punctured, punctured_per_frame, d_source_index, symbols_per_frame(total_bits),
                                       frames, d_symbols_out, stream), "vit_hip_depuncture_batch");
    }
    // batched streaming: decoders that keep their state on the device between calls (the reference's update() cursor,
    // viterbi_decoder_scalar.h:37-54): reset() once, then resume() over consecutive step ranges, then chainback()
    void reset(size_t frames, error_t* d_metrics, const uint32_t* d_start_state = nullptr, void* stream = nullptr) {
        check(vit_hip_reset_batch(m_hip, frames, d_start_state, d_metrics, stream), "vit_hip_reset_batch");
    }
    void resume(const soft_t* d_symbols, size_t symbol_frame_stride, size_t frames, size_t first_step, size_t n_steps,
                size_t total_bits, void* d_workspace, size_t workspace_size, error_t* d_metrics_inout,
                uint64_t* d_renorm_sum = nullptr, void* stream = nullptr) {
        check(vit_hip_update_batch_resume(m_hip, d_symbols, symbol_frame_stride, frames, first_step, n_steps, total_bits,
                                          d_workspace, workspace_size, d_metrics_inout, d_renorm_sum, stream),
              "vit_hip_update_batch_resume");
    }
    // test / measurement harness on the device: the BER harness's frame generator (examples/run_snr_ber.cpp:311-359) and
    // get_total_bit_errors (examples/helpers/test_helpers.h:95-104)
    void synth(size_t frames, size_t total_bits, uint64_t seed, uint64_t first_frame, float ebn0_db, bool noise_free,
               uint8_t* d_tx_bytes, soft_t* d_symbols, void* stream = nullptr) {
        check(vit_hip_synth_batch(m_hip, frames, total_bits, seed, first_frame, ebn0_db, noise_free ? 1 : 0, d_tx_bytes,
                                  d_symbols, stream), "vit_hip_synth_batch");
    }
    void count_bit_errors(const uint8_t* d_a, const uint8_t* d_b, size_t n_bytes, uint64_t* d_count, void* stream = nullptr) {
        check(vit_hip_count_bit_errors(m_hip, d_a, d_b, n_bytes, d_count, stream), "vit_hip_count_bit_errors");
    }
    // multi-GPU set-up: the shared branch table and config travel once from rank `root` to every rank of an RCCL
    // communicator (ncclComm_t); each rank then constructs its own decoder from its copy.  The other ranks pass a table
    // built from any polynomials (it is overwritten) -- the reference shares one table between decoders (README.md:14)
    static void broadcast_table(void* nccl_comm, int root, int rank, BranchTable& table, Config& config, int device,
                                void* stream = nullptr) {
        check(vit_hip_broadcast_table(nccl_comm, root, rank, int(K), int(R), int(sizeof(soft_t)), int(sizeof(error_t)),
                                      table.data(), &config, device, stream), "vit_hip_broadcast_table");
        if (rank != root) table.refresh_levels();   // soft_decision_high()/low() must describe the rows that were received
    }
    vit_hip_handle hip_handle() const { return m_hip; }
    // which plan serves this code and whether a faster one exists (vit_hip_plan_note): one line of text
    const char* plan_note() const { return vit_hip_plan_note(m_hip); }

private:
    static void check(int rc, const char* what) {
        if (rc != VIT_HIP_OK) {
            fprintf(stderr, "viterbi_hip: %s failed (%d): %s\n", what, rc, vit_hip_last_error());
            abort();
        }
    }
    vit_hip_handle m_hip = nullptr;
};

// Double-buffered pipeline over a batch decoder (vit_hip_pipeline_*): chainback of batch i runs beside the update of batch
// i+1 on a second stream.  submit() only enqueues; results of a batch are valid after a later sync().
template <size_t constraint_length, size_t code_rate, typename error_t, typename soft_t>
class ViterbiDecoder_HIP_Pipeline {
public:
    using Batch = ViterbiDecoder_HIP_Batch<constraint_length, code_rate, error_t, soft_t>;
    // `want`: schedule overrides (vit_hip_pipeline_options, struct_size set by the caller); nullptr: the library's rules
    ViterbiDecoder_HIP_Pipeline(Batch& decoder, size_t max_frames, size_t total_bits, const vit_hip_pipeline_options* want = nullptr) {
        if (vit_hip_pipeline_create_ex(decoder.hip_handle(), max_frames, total_bits, want, &m_pipe) != VIT_HIP_OK) die("vit_hip_pipeline_create_ex");
    }
    ~ViterbiDecoder_HIP_Pipeline() { vit_hip_pipeline_destroy(m_pipe); }
    ViterbiDecoder_HIP_Pipeline(const ViterbiDecoder_HIP_Pipeline&) = delete;
    ViterbiDecoder_HIP_Pipeline& operator=(const ViterbiDecoder_HIP_Pipeline&) = delete;
    void submit(const soft_t* d_symbols, size_t frames, uint8_t* d_bytes_out, const uint32_t* d_end_state = nullptr,
                void* done_event = nullptr) {
        if (vit_hip_pipeline_submit(m_pipe, d_symbols, frames, d_bytes_out, d_end_state, done_event) != VIT_HIP_OK) die("vit_hip_pipeline_submit");
    }
    void sync() {
        if (vit_hip_pipeline_sync(m_pipe) != VIT_HIP_OK) die("vit_hip_pipeline_sync");
    }
    // the pipeline runs on private non-blocking streams: order the NEXT submitted batch behind a hipEvent_t the producer of its
    // symbols recorded on its own stream (vit_hip_pipeline_wait_event); the other direction is submit()'s done_event
    void wait_event(void* producer_event) {
        if (vit_hip_pipeline_wait_event(m_pipe, producer_event) != VIT_HIP_OK) die("vit_hip_pipeline_wait_event");
    }
    // which schedule the library chose for max_frames: workspaces, update kernels in flight, chainback overlap
    vit_hip_pipeline_schedule schedule() const {
        vit_hip_pipeline_schedule s;
        if (vit_hip_pipeline_get_schedule_v2(m_pipe, &s, sizeof(s)) != VIT_HIP_OK) die("vit_hip_pipeline_get_schedule_v2");
        return s;
    }
    // per-batch kernel durations (HIP events on the kernels' own streams), the way the reference times update and chainback
    // separately (examples/run_benchmark.cpp:272-281); see vit_hip_pipeline_get_timing
    void set_timing(bool enable) {
        if (vit_hip_pipeline_set_timing(m_pipe, enable ? 1 : 0) != VIT_HIP_OK) die("vit_hip_pipeline_set_timing");
    }
    size_t timing(size_t capacity, float* update_ms, float* chainback_ms, float* complete_ms) const {
        size_t n = 0;
        if (vit_hip_pipeline_get_timing(m_pipe, capacity, update_ms, chainback_ms, complete_ms, &n) != VIT_HIP_OK) die("vit_hip_pipeline_get_timing");
        return n;
    }

private:
    static void die(const char* what) {
        fprintf(stderr, "viterbi_hip: %s failed: %s\n", what, vit_hip_last_error());
        abort();
    }
    vit_hip_pipeline_t m_pipe = nullptr;
};
