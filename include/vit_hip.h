/*
 * vit_hip.h -- C ABI of the MI355X (gfx950) Viterbi hot path: update() + chainback().
 *
 * This is the drop-in boundary.  The reference (williamyang98/ViterbiDecoderCpp) has no FFI layer: its hot path sits
 * behind a C++ template concept (static update()) plus a state object with public fields.  The C++ host layer in
 * include/viterbi_hip/ keeps that surface and forwards to the entry points below; bindings in any other language bind
 * these symbols directly.  Plain pointers and sizes only: no HIP, torch or C++ types.
 *
 * Reference interface each entry point replaces (paths relative to the reference tree):
 *   vit_hip_create            ViterbiDecoder_Core ctor            include/viterbi/viterbi_decoder_core.h:170-177
 *                             (+ ViterbiBranchTable::data()       include/viterbi/viterbi_branch_table.h:64-66,
 *                                ViterbiDecoder_Config            include/viterbi/viterbi_decoder_config.h:11-18)
 *   vit_hip_update_batch      reset() + Decoder::update<sum_t>()  viterbi_decoder_core.h:202-211, viterbi_decoder_scalar.h:29-55
 *   vit_hip_chainback_batch   chainback()                         viterbi_decoder_core.h:214-236
 *   vit_hip_decode_batch      the call pattern reset->update->chainback of examples/run_simple.cpp:76-80
 *   vit_hip_chainback_batch_ex  the same, naming which of a code's chainback kernels runs (no reference counterpart)
 *   vit_hip_pipeline_*        the same pattern over a stream of batches (the benchmark's loop over frames,
 *                             examples/run_benchmark.cpp:266-282, with its two separately timed phases :272-281) scheduled on
 *                             two or three HIP streams; _set_timing/_get_timing report the two phases per batch
 *   vit_hip_update_host       update() on a host-resident Core (streaming, N = R allowed:
 *                                                                 examples/helpers/puncture_code_helpers.h:51)
 *   vit_hip_chainback_host    chainback() on host-resident decision rows
 *   vit_hip_export_decisions  ViterbiDecisionBits rows            viterbi_decoder_core.h:49-83 (m_decisions[t][w])
 *   vit_hip_depuncture_batch  decode_punctured_symbols()          examples/helpers/puncture_code_helpers.h:17-55 (the
 *                             re-insertion of punctured symbols as the erasure value 0, for a whole batch; the update
 *                             itself then runs through vit_hip_update_batch instead of one update() per R symbols)
 *   vit_hip_reset_batch       reset(starting_state)               viterbi_decoder_core.h:202-211, for a batch of decoders
 *   vit_hip_update_batch_resume  update() called again on a decoder that already holds state: the cursor
 *                             m_current_decoded_bit and the metrics carry over
 *                                                                 viterbi_decoder_scalar.h:29-55 (:37-54 the cursor)
 *   vit_hip_broadcast_table   "Branch table can be shared between multiple decoders"   README.md:14,
 *                             viterbi_branch_table.h:17-18, one decoder per worker examples/run_benchmark.cpp:193-197:
 *                             here the workers are GPUs and the table travels once over RCCL/xGMI
 *   vit_hip_synth_batch       the BER harness's frame generator   examples/run_snr_ber.cpp:311-359,
 *                             examples/helpers/test_helpers.h:17-64, convolutional_encoder_shift_register.h:42-62
 *   vit_hip_count_bit_errors  get_total_bit_errors()              examples/helpers/test_helpers.h:95-104
 *
 * Semantics are those of the reference SCALAR strategy (strict '>' decision, wrapping error_t arithmetic,
 * renormalise only when new_metric[0] >= threshold): SURVEY.md section 8(a').  All results are bit-exact.
 *
 * Conventions
 *   - every function returns VIT_HIP_OK (0) or a negative VIT_HIP_ERR_*; no exception crosses the ABI;
 *     vit_hip_last_error() returns a thread-local description of the last failure.
 *   - pointers named d_* are DEVICE pointers on the handle's device; others are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Batch calls only enqueue work: no
 *     allocation, no synchronisation, safe to capture into a hipGraph.
 *   - N = 2^(K-1) states, H = N/2, W = max(N/64, 1) 64-bit decision words per step, S = L + K-1 steps per frame.
 *   - threads: the reference keeps one Core per thread and shares the const branch table (examples/run_benchmark.cpp:193-197,
 *     viterbi_branch_table.h:17-18).  Here a handle plays the table's part for the BATCH calls: they read the handle and
 *     write only caller-owned buffers (workspace, outputs), so several threads / streams may issue them on one handle at
 *     once as long as each call in flight has its own workspace.  The HOST-route calls (vit_hip_update_host,
 *     vit_hip_chainback_host) stage through buffers the handle owns, and vit_hip_set_plan / vit_hip_destroy modify it: one
 *     thread at a time per handle for those (one handle per thread, as one Core per thread in the reference).
 */
#ifndef VIT_HIP_H
#define VIT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIT_HIP_OK 0
#define VIT_HIP_ERR_INVALID_ARG (-1)
#define VIT_HIP_ERR_UNSUPPORTED (-2)
#define VIT_HIP_ERR_RUNTIME (-3)     /* a HIP runtime call failed              */
#define VIT_HIP_ERR_NO_DEVICE (-4)   /* no usable GPU: the product path never falls back to the CPU */
#define VIT_HIP_ERR_WORKSPACE (-5)   /* workspace too small / misaligned       */

/* kernel plans (vit_hip_set_plan): which device implementation serves update()/chainback() */
#define VIT_HIP_PLAN_AUTO 0
#define VIT_HIP_PLAN_LDS 1  /* state metrics staged in LDS, one wavefront per contiguous state slab, ballot decisions */
#define VIT_HIP_PLAN_REG 2  /* state metrics resident in VGPRs, frames in pairs, packed 16-bit ACS (K = 2..9, R <= 6).
                               The stock codes are built in; other polynomials: precompiled at install time
                               (vit_hip_precompile: no compiler on the serving host) or vit_hip_set_plan(h, PLAN_REG) compiles
                               an instantiation with hipcc on first use (10-40 s, cached on disk: VIT_HIP_CACHE_DIR)   */
#define VIT_HIP_PLAN_LDS2 3 /* packed frame pair per workgroup, 16 states per thread, four trellis steps per barrier, u32
                               metrics updated in place in LDS (K = 10..16, R <= 6, any polynomials; K = 10: two frame
                               pairs per wavefront)                                                              */

typedef struct vit_hip_decoder* vit_hip_handle;
typedef void* vit_hip_stream_t;

typedef struct vit_hip_info {
    int32_t K, R, soft_bytes, error_bytes;
    int32_t num_states;      /* N */
    int32_t decision_words;  /* W */
    int32_t device;
    int32_t plan;            /* resolved plan (VIT_HIP_PLAN_LDS / _REG / _LDS2) */
    int32_t soft_decision_high, soft_decision_low; /* recovered from the branch table */
    uint32_t polynomials[16];                      /* recovered G[i] (bit 0 and bit K-1 forced to 1), 0 if not linear */
    int32_t table_is_linear;
    int32_t workspace_tile_frames; /* the decision workspace is an array of independent slabs of this many frames: frame f
                                      lives in slab f / tile, which starts vit_hip_workspace_slab_bytes(h, L) * (f / tile)
                                      bytes into the workspace -- a slab-aligned sub-range of a batch can be chained back or
                                      exported on its own by passing that address */
} vit_hip_info;

const char* vit_hip_last_error(void);
int vit_hip_device_count(void);

/* branch_table: host, [R][H] soft_t exactly as ViterbiBranchTable::data() lays it out (rows contiguous, no padding).
 * config: host, 4 x error_t in the order of ViterbiDecoder_Config (max_error, initial_start_error,
 * initial_non_start_error, renormalisation_threshold).  (soft_bytes, error_bytes) must be (2,2) or (1,1).
 * The table must be two-valued (high/low); it is copied, the caller's table need not outlive the handle. */
int vit_hip_create(int K, int R, int soft_bytes, int error_bytes, const void* branch_table, const void* config,
                   int device, vit_hip_handle* out);
int vit_hip_destroy(vit_hip_handle h);
int vit_hip_get_info(vit_hip_handle h, vit_hip_info* info);
/* PLAN_AUTO never compiles anything: it resolves to a built-in plan, or to PLAN_REG for a code whose kernels were precompiled at
 * install time (vit_hip_precompile below: kernels specialised for its polynomials, else the GENERIC kernels of its (K, R)) -- set
 * VIT_HIP_JIT=1 before vit_hip_create to let it compile.  vit_hip_set_plan(h, PLAN_REG) on a handle that runs the generic kernels looks
 * for specialised ones (package cache, user cache, compiler) and keeps the generic ones when there are none. */
int vit_hip_set_plan(vit_hip_handle h, int plan);
/* Install-time instantiation of the register plan for one polynomial set and symbol width (soft_bytes 1 or 2), for hosts that run
 * WITHOUT a compiler.  In the reference the polynomials are a run-time constructor argument of the branch table
 * (include/viterbi/viterbi_branch_table.h:34-55) and any set runs at the compiled <K,R>'s full speed; here the register plan's
 * kernels are specialised per set, so sets outside the eight stock codes are compiled ahead of use: this call writes the code
 * object into `directory` (NULL: the package cache `<directory of libvit_hip.so>/precompiled`, which vit_hip_create consults for
 * every code it has no built-in kernels for -- such a code then runs PLAN_REG from vit_hip_create on, PLAN_AUTO included).
 * Needs hipcc ($VIT_HIP_HIPCC, else /opt/rocm/bin/hipcc) and the kernel sources beside the library, but NO GPU: a build host
 * without a card can run it (python -m viterbidecodercpp_amd.tools.precompile does, for a list of common sets, from build()).
 * An object that already exists is kept.  path_out (optional) receives the file's path.  K = 2..9, R <= 6.
 * ALL polynomials zero names the GENERIC kernels of (K, R) (K = 3 .. 9 with R = 1 .. 4; not K = 6 at an odd rate): one code object that reads the
 * polynomials from its arguments at run time, as the reference's branch table does -- 0.79 - 0.98 of the specialised kernels' rate.  build()
 * installs them, and vit_hip_create falls back to them (PLAN_AUTO included) for every set of those (K, R) that has neither built-in nor
 * precompiled kernels: no such code drops to the compatibility plan on a host without a compiler.  vit_hip_plan_note says "GENERIC". */
int vit_hip_precompile(int K, int R, const uint32_t* polynomials, int soft_bytes, const char* directory, char* path_out,
                       size_t path_capacity);
/* One line of text about the plan the handle runs and -- where that is PLAN_LDS, the compatibility plan -- whether a faster one
 * exists for this (K, R) and how to get it: no combination the header-level is_valid admits lands on the slow plan silently.
 * Thread-local storage, valid until the thread's next call.  (The reference has no counterpart: its strategies are chosen at
 * compile time, include/viterbi/viterbi_decoder_scalar.h:25.) */
const char* vit_hip_plan_note(vit_hip_handle h);

/* Opaque blob carrying everything vit_hip_create needs (header + table + config): the payload broadcast to the other
 * ranks of a node (RCCL over xGMI via torch.distributed) so that every GPU builds an identical decoder. */
size_t vit_hip_blob_bytes(int K, int R, int soft_bytes, int error_bytes);
int vit_hip_pack_blob(int K, int R, int soft_bytes, int error_bytes, const void* branch_table, const void* config,
                      void* blob, size_t blob_bytes);
int vit_hip_create_from_blob(const void* blob, size_t blob_bytes, int device, vit_hip_handle* out);

/* ---- batched, device-resident (the throughput route) --------------------------------------------------------- */

/* bytes of decision workspace needed for `frames` frames of L info bits (layout is plan-specific and opaque). */
size_t vit_hip_workspace_bytes(vit_hip_handle h, size_t frames, size_t L);
/* bytes between consecutive slabs of vit_hip_info.workspace_tile_frames frames inside a workspace.  Not the same as
 * vit_hip_workspace_bytes(h, tile, L): that is rounded up to 256 bytes, the slab stride of PLAN_LDS (the dense reference
 * layout [F][S][W]) is not.  vit_hip_chainback_batch and vit_hip_export_decisions accept any slab address. */
size_t vit_hip_workspace_slab_bytes(vit_hip_handle h, size_t L);

/* reset(start_state) + update() over n_steps trellis steps (n_steps <= L + K-1) for every frame.
 *   d_symbols       [frames][n_steps][R] soft_t, frame-major, step-major, polynomial-minor
 *   d_workspace     >= vit_hip_workspace_bytes(h, frames, L), 256-byte aligned; receives the decision history
 *   d_final_metrics [frames][N] error_t or NULL  (Core::m_metrics "old" buffer after the last swap)
 *   d_renorm_sum    [frames] uint64 or NULL      (update()'s return value)
 *   d_start_state   [frames] uint32 or NULL (=> 0) */
int vit_hip_update_batch(vit_hip_handle h, const void* d_symbols, size_t frames, size_t n_steps, size_t L,
                         void* d_workspace, size_t workspace_bytes, void* d_final_metrics, uint64_t* d_renorm_sum,
                         const uint32_t* d_start_state, vit_hip_stream_t stream);

/* chainback(bytes_out, L, end_state) for every frame from the workspace a full update (n_steps = L+K-1) filled.
 *   d_bytes_out [frames][ceil(L/8)] ; d_end_state [frames] uint32 or NULL (=> 0) */
int vit_hip_chainback_batch(vit_hip_handle h, const void* d_workspace, size_t frames, size_t L, uint8_t* d_bytes_out,
                            const uint32_t* d_end_state, vit_hip_stream_t stream);
/* the same with the kernel named: VIT_HIP_KERNEL_CHAINBACK (what vit_hip_chainback_batch launches) or
 * VIT_HIP_KERNEL_CHAINBACK_ALT, the other chainback kernel of the K = 7 / K = 9 register-plan codes (K = 7: the LDS-ring kernel,
 * 32 registers -- what a pipeline runs BESIDE update waves; K = 9: the cooperative kernel).  Codes with one chainback kernel
 * ignore the choice.  Results are identical; tests and schedulers use it (there is no environment switch). */
int vit_hip_chainback_batch_ex(vit_hip_handle h, const void* d_workspace, size_t frames, size_t L, uint8_t* d_bytes_out,
                               const uint32_t* d_end_state, vit_hip_stream_t stream, int kernel);

/* update over S = L+K-1 steps from state 0, then chainback: both phases enqueued on `stream`. */
int vit_hip_decode_batch(vit_hip_handle h, const void* d_symbols, size_t frames, size_t L, void* d_workspace,
                         size_t workspace_bytes, uint8_t* d_bytes_out, void* d_final_metrics, uint64_t* d_renorm_sum,
                         const uint32_t* d_end_state, vit_hip_stream_t stream);

/* decision history in the reference's layout: d_decisions [frames][n_steps][W] uint64, bit s%64 of word s/64 of row t
 * = decision for next-state s at step t. */
int vit_hip_export_decisions(vit_hip_handle h, const void* d_workspace, size_t frames, size_t n_steps, size_t L,
                             uint64_t* d_decisions, vit_hip_stream_t stream);

/* Depuncturing front-end for punctured codes (DAB, LTE rate matching ...): builds the [frames][symbols_per_frame] soft_t
 * stream vit_hip_update_batch consumes from the symbols that were actually transmitted.
 *   d_punctured     [frames][punctured_per_frame] soft_t, frame-major
 *   d_source_index  [symbols_per_frame] int32: index into the frame's punctured symbols, or < 0 for a punctured position,
 *                   which reads as the erasure value 0 (examples/run_punctured_decoder.cpp:165).  The same map serves every
 *                   frame; for a puncturing mask it is the exclusive prefix sum of the mask where the mask is set.
 *   d_symbols_out   [frames][symbols_per_frame] soft_t */
int vit_hip_depuncture_batch(vit_hip_handle h, const void* d_punctured, size_t punctured_per_frame,
                             const int32_t* d_source_index, size_t symbols_per_frame, size_t frames, void* d_symbols_out,
                             vit_hip_stream_t stream);

/* ---- double-buffered decode pipeline --------------------------------------------------------------------------------- */

/* update() is bound by integer issue, chainback() by memory latency: run back to back they leave each other's resource
 * idle.  A pipeline owns the decision workspaces and HIP streams of a stream of batches and schedules them (this is what
 * bench.py times).  submit() only enqueues and returns; batches complete in submit order; the caller's symbol and output
 * buffers of a batch must stay untouched until a later sync() (or until `done_event`, an optional hipEvent_t passed as
 * void*, has fired).  The schedule is fixed at create time from max_frames (vit_hip_pipeline_get_schedule reports it).  Which
 * kernels may share a SIMD is decided from their kernel DESCRIPTORS (vit_hip_get_kernel_resources: the register allocation the
 * wave launcher uses, not the count the code touches), 512 registers and 160 KiB of LDS per SIMD / CU:
 *   - register plan, more than one and at most two update waves per SIMD (4 x CUs x workspace_tile_frames < max_frames <=
 *     2 x that: 65536 frames at K = 7, 9 on an MI355X) whose chainback kernel fits beside two update waves: two workspaces,
 *     the chainback of batch i on a second, high-priority stream beside the update of batch i+1.
 *       K = 7 (2 x 152 registers): the chainback kernel of this schedule is the LDS-ring one (32 registers, 24 KiB of LDS; the
 *         register-ring kernel, 160 registers, stays the kernel of a chainback that runs alone): 65536 x 8192 back to back
 *         4.15 ms per batch -> 3.30 - 3.45 beside the register-ring kernel -> 3.27 - 3.40;
 *         up to THREE update waves per SIMD (3 x 152 + 32): a 98304-frame batch 5.8 -> 5.0 ms;
 *       K = 9, R = 2 (update capped at 240 registers, LDS-streaming chainback 24 registers with its ring in DYNAMIC LDS -- as
 *         static LDS hipcc padded its allocation to 264 and it ran alone on its SIMD): 65536 x 8192 12.8 -> 12.1 ms;
 *   - register plan, at most ONE update wave per SIMD (max_frames <= 4 x CUs x workspace_tile_frames: the 32768-frame share
 *     of an 8-GPU run): THREE workspaces and TWO update streams, so that two update kernels share the SIMDs (a lone wave
 *     issues a packed instruction every 5.3 cycles, two every 4.5) with the chainbacks beside them on the third stream, at a
 *     higher wave priority than the updates (between two staggered update kernels the chainback would otherwise get the issue
 *     slots both leave over and become the bottleneck; here the fast register-ring kernel serves K = 7 too);
 *       K = 7, R = 3 (LTE: update capped at 240, 2 x 240 + 32), K = 7, R = 4 (DAB: 2 x 224 + 32; eight update waves hold 16 KiB
 *         of LDS each, the chainback ring of these codes is 12 KiB) and K = 9, R = 4 (CDMA 2000: 224 registers with the
 *         sub-chunk branch-metric fetch, 8 KiB of LDS per wave) take the same schedule;
 *   - register plan whose update waves leave no registers (or LDS) for a chainback wave beside two of them (no built-in code any
 *     more; a run-time compiled code whose kernels spill past the caps may): ANY batch of more than one wave per SIMD is fed to
 *     the kernels as SUB-BATCHES of one wave per SIMD through the same three-workspace, two-update-stream schedule (timing
 *     records and vit_hip_pipeline_last_workspace are per sub-batch; an error in the middle of a submit() leaves the sub-batches
 *     already enqueued in flight: sync() or destroy the pipeline before touching the buffers);
 *   - PLAN_LDS2 where the update waves a CU's LDS admits leave the chainback kernel's 24 registers on every SIMD (K = 11, 12,
 *     14, 15: four waves of at most 120; K = 13: three of 144): two workspaces, chainback beside the next update (K = 15,
 *     4096 frames: 51.6 -> 50.2 ms per batch; K = 13, 8192 x 4096: 16.4 -> 16.0);
 *   - larger register-plan batches, K = 16 (4 x 128 registers fill the SIMDs) and PLAN_LDS: a chainback in the update's way
 *     costs more than it hides: one stream, update and chainback back to back. */
typedef struct vit_hip_pipeline* vit_hip_pipeline_t;
typedef struct vit_hip_pipeline_schedule {
    int32_t workspaces;             /* decision workspaces owned (2 or 3) */
    int32_t update_streams;         /* update kernels that may be in flight at once (1 or 2) */
    int32_t chainback_overlapped;   /* 1: chainbacks run on their own stream beside the next update; 0: back to back */
    int32_t chainback_wave_priority;/* 1: the chainback kernel runs at a higher wave priority than the update kernels */
    size_t overlap_max_frames;      /* largest batch whose chainback is overlapped */
    size_t two_updates_max_frames;  /* largest max_frames that gets the two-update schedule without sub-batches */
    size_t workspace_bytes_each;
    size_t sub_batch_frames;        /* a submitted batch reaches the kernels in sub-batches of at most this many frames
                                       (= max_frames unless the schedule splits); timing records are per sub-batch */
    int32_t chainback_small_kernel; /* 1: overlapped chainbacks run the small-footprint kernel of the code (K = 7: LDS ring, 32 registers) */
    int32_t reserved;
} vit_hip_pipeline_schedule;
int vit_hip_pipeline_create(vit_hip_handle h, size_t max_frames, size_t L, vit_hip_pipeline_t* out);
/* The supported way to pick a schedule other than the library's (A/B measurements, a host that knows what else shares the GPU):
 * every field left at its "rule" value keeps what vit_hip_pipeline_create would choose; the result is what
 * vit_hip_pipeline_get_schedule_v2 reports.  A request the hardware cannot serve (sub-batches for a batch of one wave per SIMD,
 * three update streams without sub-batches) is ignored field by field, never an error.  `want` == NULL: the rules.
 * The library reads NO environment variable for its schedules (a build with -DVIT_HIP_EXPERIMENTS does, for scripts/gpu_ab.sh). */
typedef struct vit_hip_pipeline_options {
    uint32_t struct_size;            /* sizeof(vit_hip_pipeline_options) in the caller's build */
    int32_t chainback_overlap;       /* -1 rule; 0 back to back on the update's stream; 1 overlapped whatever the batch size */
    int32_t update_streams;          /*  0 rule; 1, 2 or 3 update kernels in flight (3 only with sub-batches of one wave per SIMD) */
    int32_t sub_batches;             /* -1 rule; 0 never split; 1 split batches of more than one wave per SIMD */
    int32_t workspaces;              /*  0 rule (update_streams + 1); 2..4 */
    int32_t chainback_small_kernel;  /* -1 rule; 0 the code's stand-alone chainback kernel; 1 its small-footprint kernel (K = 7) */
    int32_t chainback_wave_priority; /* -1 rule; 0 / 1 */
} vit_hip_pipeline_options;
int vit_hip_pipeline_create_ex(vit_hip_handle h, size_t max_frames, size_t L, const vit_hip_pipeline_options* want,
                               vit_hip_pipeline_t* out);
int vit_hip_pipeline_submit(vit_hip_pipeline_t p, const void* d_symbols, size_t frames, uint8_t* d_bytes_out,
                            const uint32_t* d_end_state, void* done_event);
int vit_hip_pipeline_sync(vit_hip_pipeline_t p);
int vit_hip_pipeline_destroy(vit_hip_pipeline_t p);
/* schedule_bytes = sizeof(vit_hip_pipeline_schedule) in the CALLER's build: at most that many bytes are written, so a binary
 * built against an older, shorter struct keeps working when the struct grows. */
int vit_hip_pipeline_get_schedule_v2(vit_hip_pipeline_t p, vit_hip_pipeline_schedule* schedule, size_t schedule_bytes);
/* legacy entry point: writes the struct as it was when the symbol was introduced (up to and including sub_batch_frames) */
int vit_hip_pipeline_get_schedule(vit_hip_pipeline_t p, vit_hip_pipeline_schedule* schedule);
/* the decision workspace of the most recently submitted (sub-)batch and the frame range of the submitted batch it holds
 * (vit_hip_export_decisions reads the history from it; first_frame / frames may be NULL): valid after a sync() and until the
 * next submit(); owned by the pipeline. */
int vit_hip_pipeline_last_workspace(vit_hip_pipeline_t p, void** d_workspace, size_t* first_frame, size_t* frames);
/* Per-batch timing with HIP events on the streams the kernels run on (the reference times its two phases separately:
 * examples/run_benchmark.cpp:272-281).  set_timing(p, 1) synchronises, clears the records and starts recording every
 * submitted batch; set_timing(p, 0) stops.  get_timing() returns, for the batches completed by the last sync(), in submit
 * order (one record per SUB-batch: vit_hip_pipeline_schedule.sub_batch_frames): the update kernel's duration, the chainback kernel's duration, and the time at which the batch's chainback finished
 * measured from the start of the first recorded update (all in ms; consecutive differences of complete_ms are the
 * pipeline's per-batch step times).  Arrays may be NULL; at most `capacity` entries are written, *n_batches is the count. */
int vit_hip_pipeline_set_timing(vit_hip_pipeline_t p, int enable);
int vit_hip_pipeline_get_timing(vit_hip_pipeline_t p, size_t capacity, float* update_ms, float* chainback_ms, float* complete_ms,
                                size_t* n_batches);

/* Order the NEXT submitted batch behind work of the caller's own stream: its update kernel will not start before `event` (a
 * hipEvent_t passed as void*, recorded by the caller on the stream that produces d_symbols / last used d_bytes_out) has fired.
 * The pipeline runs on private non-blocking streams, so without this (or a device / stream synchronisation) a submit() right
 * after an asynchronous producer of the symbols races with it.  The other direction is submit()'s `done_event`: make the
 * consumer's stream wait on it.  No reference counterpart (the reference is synchronous host code). */
int vit_hip_pipeline_wait_event(vit_hip_pipeline_t p, void* event);

/* ---- what the hardware allocates per wave of a kernel (diagnostics; the pipeline's residency rules read the same table) ---- */

/* From the kernel DESCRIPTOR in the code object (compute_pgm_rsrc1/3, group_segment_fixed_size): the numbers the wave launcher
 * uses to decide which waves share a SIMD / CU.  NOT hipFuncGetAttributes().numRegs, which is the count the code uses: hipcc
 * pads the allocation of kernels whose static LDS limits their occupancy.  No reference counterpart. */
typedef struct vit_hip_kernel_resources {
    uint32_t vgpr_alloc;        /* entries of the SIMD's 512-entry unified register file one wave occupies */
    uint32_t accum_offset;      /* of which architectural VGPRs (the rest are accumulation registers) */
    uint32_t lds_static_bytes;  /* per workgroup, fixed at compile time */
    uint32_t lds_dynamic_bytes; /* per workgroup, added at launch (vit_hip_get_kernel_resources only; 0 from vit_hip_list_kernels) */
    uint32_t scratch_bytes;     /* per lane */
} vit_hip_kernel_resources;
#define VIT_HIP_KERNEL_UPDATE 0
#define VIT_HIP_KERNEL_CHAINBACK 1
#define VIT_HIP_KERNEL_CHAINBACK_ALT 2   /* the other chainback kernel of K = 7 / 9 (vit_hip_chainback_batch_ex) */
#define VIT_HIP_KERNEL_RESUME 3
/* the kernel a handle launches for `kernel` (register plan and PLAN_LDS2; VIT_HIP_ERR_UNSUPPORTED for PLAN_LDS) */
int vit_hip_get_kernel_resources(vit_hip_handle h, int kernel, vit_hip_kernel_resources* out);
/* every kernel of the library's embedded gfx950 code objects, by index (needs no GPU): VIT_HIP_ERR_INVALID_ARG past the end.
 * `name` receives the (mangled) kernel name, truncated to name_capacity - 1 characters. */
int vit_hip_list_kernels(size_t index, char* name, size_t name_capacity, vit_hip_kernel_resources* out);

/* ---- batched streaming: a batch of decoders fed in chunks, state resident on the device ------------------------- */

/* reset(start_state) for every frame: d_metrics [frames][N] error_t <- initial_non_start_error, initial_start_error at the
 * start state (d_start_state [frames] uint32 or NULL => 0). */
int vit_hip_reset_batch(vit_hip_handle h, size_t frames, const uint32_t* d_start_state, void* d_metrics,
                        vit_hip_stream_t stream);

/* update() on decoders that already hold state: NO reset.  Every frame's cursor stands at trellis step `first_step`
 * (= m_current_decoded_bit) and its metrics are in d_metrics_inout [frames][N] error_t; the call consumes n_steps more
 * steps (first_step + n_steps <= L + K-1), writes decision rows [first_step, first_step + n_steps) of the workspace and
 * leaves the new metrics in d_metrics_inout.  A sequence of calls over consecutive step ranges is bit-identical to one
 * vit_hip_update_batch over the whole range (decision rows, metrics, and the renormalisation sums added up).
 *   d_symbols            frame f's chunk starts at d_symbols + f * symbol_frame_stride (in soft_t elements) and holds
 *                        [n_steps][R]; symbol_frame_stride = 0 means n_steps * R (chunks packed back to back)
 *   d_renorm_sum         [frames] uint64 or NULL: update()'s return value for THIS call (not accumulated) */
int vit_hip_update_batch_resume(vit_hip_handle h, const void* d_symbols, size_t symbol_frame_stride, size_t frames,
                                size_t first_step, size_t n_steps, size_t L, void* d_workspace, size_t workspace_bytes,
                                void* d_metrics_inout, uint64_t* d_renorm_sum, vit_hip_stream_t stream);

/* ---- multi-GPU set-up for C/C++ hosts ----------------------------------------------------------------------------------- */

/* Broadcast the shared branch table + config from rank `root` to every rank of an RCCL communicator (ncclComm_t passed as
 * void*), over xGMI: the one collective of the multi-GPU path (frames shard with no data-path exchange).  On `root`
 * branch_table / config are inputs, on the other ranks they are outputs ([R][H] soft_t and 4 x error_t); every rank passes
 * the same K, R and widths (template parameters in the reference).  Collective and synchronous on `stream`.  librccl.so is
 * dlopen()ed on first use: the library carries no link-time dependency on RCCL.
 * Failure contract: argument errors are detected alike on every rank before the collective.  A rank whose DATA is bad still
 * enters it (a root that cannot pack or upload the table broadcasts a poisoned header; every rank then returns an error).  The
 * 264 KiB device staging buffer is allocated once per device at the process's first call and kept, so a later call cannot run
 * out of memory in front of the collective.  A rank that has no usable device (hipSetDevice fails, or that one-off allocation
 * fails at the first call) returns VIT_HIP_ERR_NO_DEVICE WITHOUT entering: the caller must then abort the communicator
 * (ncclCommAbort) -- the other ranks are waiting in ncclBroadcast. */
int vit_hip_broadcast_table(void* nccl_comm, int root, int rank, int K, int R, int soft_bytes, int error_bytes,
                            void* branch_table, void* config, int device, vit_hip_stream_t stream);

/* ---- synthetic frames and error counting on the device (test / measurement harness) ----------------------------------- */

/* The reference BER harness's generator for a whole batch, in one kernel: random info bytes -> convolutional encoder
 * (MSB-first bits, K-1 zero tail bits) -> +-1 BPSK + N(0, sigma^2) -> round / clamp to [low, high]
 * (sigma^2 = 10^(-(ebn0_db - 10 log10 R + 3)/10), scale (high-low)/2 / sqrt(1 + sigma^2): run_snr_ber.cpp:319-359).
 * Every value is a pure function of (seed, first_frame + f, position) through Philox4x32-10: splitting a batch over calls
 * or ranks does not change it.  noise_free != 0: symbols are exactly high / low (ebn0_db ignored).
 *   d_tx_bytes [frames][L/8] or NULL (L % 8 == 0) ; d_symbols [frames][L+K-1][R] soft_t */
int vit_hip_synth_batch(vit_hip_handle h, size_t frames, size_t L, uint64_t seed, uint64_t first_frame, float ebn0_db,
                        int noise_free, uint8_t* d_tx_bytes, void* d_symbols, vit_hip_stream_t stream);

/* *d_count (uint64, device) += number of differing bits between two device byte arrays. */
int vit_hip_count_bit_errors(vit_hip_handle h, const uint8_t* d_a, const uint8_t* d_b, size_t n_bytes, uint64_t* d_count,
                             vit_hip_stream_t stream);

/* The clock the SIMDs sustain under the update kernels' instruction class, measured on the device: every SIMD runs four waves
 * of independent v_pk_add_u16 for about 2 ms between readings of s_memtime (shader clocks) and s_memrealtime (constant
 * reference clock); *mhz_out = their median ratio x the reference rate.  *cycles_per_pk_instr_out (may be NULL) = shader
 * clocks one SIMD needed per wave64 packed instruction in that loop.  With cycles_per_pk_instr_out == NULL the probe is ONE
 * wave per CU: light enough to read the clock WHILE other kernels run (launched on the null stream; the decode pipeline's
 * streams are non-blocking) without adding a chip full of vector work to their power draw.  Synchronous; measurement harness
 * (bench.py quotes its VALU ceiling at this clock, not at a nominal one). */
int vit_hip_shader_clock_mhz(int device, double* mhz_out, double* cycles_per_pk_instr_out);

/* ---- host-pointer compatibility route (one decoder object, streaming) ------------------------------------------ */

/* update() on a host-resident decoder state: metrics_inout [N] error_t ("old" metrics), symbols n_steps*R soft_t,
 * decisions_out [n_steps][W] (rows for THIS call), *renorm_sum_out = update()'s return value.  Synchronous. */
int vit_hip_update_host(vit_hip_handle h, void* metrics_inout, const void* symbols, size_t n_steps,
                        uint64_t* decisions_out, uint64_t* renorm_sum_out);

/* chainback() on host-resident rows: decisions [L+K-1][W].  Synchronous. */
int vit_hip_chainback_host(vit_hip_handle h, const uint64_t* decisions, size_t L, size_t end_state, uint8_t* bytes_out);

/* ---- the FRAME route of the header-level drop-in: update() and the chainback() behind it in ONE launch ---------------------------
 * vit_hip_update_host / vit_hip_chainback_host above keep the reference's division of labour -- every call hands the decision rows
 * back to the host and takes them in again (core.h:214-236 reads m_decisions on the host) -- which for one frame means two launches,
 * two synchronisations and four staging copies around 0.3 ms of trellis.  The lazy pair leaves the rows where they were made:
 *   vit_hip_update_host_lazy    runs n_steps steps like vit_hip_update_host, but stores their decision rows as rows [first_row,
 *       first_row + n_steps) of the handle's DEVICE row store instead of returning them (the store keeps earlier rows; a handle has
 *       one frame's rows, as a ViterbiDecoder_Core has one m_decisions).  K <= 7: symbols are read straight from pinned host-mapped
 *       memory, metrics / renormalisation sum come back through it, and the host polls a completion word instead of synchronising
 *       the stream.  When the call COMPLETES a frame of `speculate_bits` decoded bits (first_row + n_steps == speculate_bits + K-1;
 *       0 = never) the same launch also chains back those bits from `speculate_end_state` and keeps the bytes;
 *   vit_hip_chainback_host_lazy chainback(L, end_state) over the device row store: the bytes kept by the completing update if it was
 *       asked for exactly this (L, end_state) and no other lazy update ran since -- no GPU work at all -- else one kernel;
 *   vit_hip_fetch_decisions_host  copies rows of the device row store to the host (what reading m_decisions[row] triggers).
 * The caller tracks which rows are authoritative where (include/viterbi_hip/viterbi_decoder_core.h does): rows written by
 * vit_hip_update_host_lazy live on the device until fetched; rows the caller changed on the host make it use the eager pair again. */
int vit_hip_update_host_lazy(vit_hip_handle h, void* metrics_inout, const void* symbols, size_t n_steps, size_t first_row,
                             size_t speculate_bits, size_t speculate_end_state, uint64_t* renorm_sum_out);
int vit_hip_chainback_host_lazy(vit_hip_handle h, size_t L, size_t end_state, uint8_t* bytes_out);
int vit_hip_fetch_decisions_host(vit_hip_handle h, size_t first_row, size_t n_rows, uint64_t* decisions_out);

#ifdef __cplusplus
}
#endif
#endif /* VIT_HIP_H */
