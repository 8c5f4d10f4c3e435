// vit_hip.hip -- C ABI (include/vit_hip.h) over the gfx950 kernels.  Host-side logic only: argument checking, plan
// selection, workspace layout, launches.  There is no CPU decode path in this library: without a usable GPU every entry
// point fails with VIT_HIP_ERR_NO_DEVICE / VIT_HIP_ERR_RUNTIME.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/vit_hip.h"
#include "../../include/vit_hip_experiments.h"
#include "kernels_lds.hpp"
#include "kernels_lds2.hpp"
#include "kernels_one.hpp"
#include "kernels_reg.hpp"
#include "kernels_synth.hpp"
#include "reg_jit.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define VIT_HIP_CHECK(expr)                                                                             \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(VIT_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
        if (prev == dev) prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

constexpr uint32_t BLOB_MAGIC = 0x56495442u;  // "VITB"

struct BlobHeader {
    uint32_t magic;
    int32_t K, R, soft_bytes, error_bytes;
};

}  // namespace

struct vit_hip_decoder {
    int K = 0, R = 0, soft_bytes = 0, error_bytes = 0, device = 0;
    int N = 0, H = 0, W = 0, shift = 0;
    int plan = VIT_HIP_PLAN_LDS;
    int high = 0, low = 0;
    bool linear = false;
    uint32_t G[16] = {0};
    uint32_t cfg_raw[4] = {0, 0, 0, 0};
    vit::DevConfig cfg{};
    std::vector<uint16_t> pattern;  // [H] host copy
    uint16_t* d_pattern = nullptr;
    vit::RegCode reg_code{};        // PLAN_REG description (valid when reg_ok)
    bool reg_ok = false;
    bool lds2_ok = false;
    std::string reg_origin;         // path of the code object a run-time / install-time compiled PLAN_REG was loaded from
    // host-route scratch
    hipStream_t stream = nullptr;
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    void* h_stage = nullptr;        // pinned host staging: one H2D and one D2H per host-route call
    size_t stage_bytes = 0;
    // frame route (vit_hip_update_host_lazy / vit_hip_chainback_host_lazy / vit_hip_fetch_decisions_host)
    uint64_t* d_rows = nullptr;     // decision rows of the handle's ONE host-route frame, [row][W], kept on the device
    size_t rows_cap = 0;            // rows allocated
    void* h_map = nullptr;          // pinned AND host-mapped: [64 B control | 192 B | metrics | symbols | decoded bytes]
    size_t map_bytes = 0;
    uint32_t seq = 0;               // the value the next frame kernel stores into the control word when it is done
    bool spec_valid = false;        // the decoded bytes in h_map are those of chainback(spec_bits, spec_end) over the rows in d_rows
    size_t spec_bits = 0, spec_end = 0, spec_off = 0;
};

namespace {

int lds_waves(int N) {
    if (N <= 256) return 1;
    int w = N / 256;
    return w > 16 ? 16 : w;
}

size_t lds_smem_bytes(int N, int waves) {
    const int H = N / 2;
    const int pat = N <= 16384 ? (H > 0 ? H : 1) : 0;     // K = 16 reads the patterns from global memory (kernels_lds.hpp)
    return (size_t)(2 * N + pat + waves) * sizeof(uint16_t);
}

template <int R, int SHIFT>
int launch_lds_update_t(const vit::LdsUpdateArgs& a, size_t frames, int N, hipStream_t st) {
    const int waves = lds_waves(N);
    const size_t smem = lds_smem_bytes(N, waves);
    auto kern = vit::lds_update_kernel<R, SHIFT>;
    if (smem > 64 * 1024) {
        VIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)smem));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)frames), dim3(64 * waves), smem, st, a);
    VIT_HIP_CHECK(hipGetLastError());
    return VIT_HIP_OK;
}

template <int SHIFT>
int launch_lds_update_r(int R, const vit::LdsUpdateArgs& a, size_t frames, int N, hipStream_t st) {
    switch (R) {
        case 1: return launch_lds_update_t<1, SHIFT>(a, frames, N, st);
        case 2: return launch_lds_update_t<2, SHIFT>(a, frames, N, st);
        case 3: return launch_lds_update_t<3, SHIFT>(a, frames, N, st);
        case 4: return launch_lds_update_t<4, SHIFT>(a, frames, N, st);
        case 5: return launch_lds_update_t<5, SHIFT>(a, frames, N, st);
        case 6: return launch_lds_update_t<6, SHIFT>(a, frames, N, st);
        case 7: return launch_lds_update_t<7, SHIFT>(a, frames, N, st);
        case 8: return launch_lds_update_t<8, SHIFT>(a, frames, N, st);
        default: return fail(VIT_HIP_ERR_UNSUPPORTED, "code rate R must be 1..8");
    }
}

// decisions in the reference layout [F][rows][W]; rows = L + K-1
int lds_update(vit_hip_handle h, const void* d_symbols, size_t sym_stride, size_t frames, size_t n_steps, size_t rows,
               uint32_t row0, uint64_t* d_decisions, void* d_metrics, bool reset, uint64_t* d_renorm, const uint32_t* d_start,
               hipStream_t st) {
    if (frames == 0 || n_steps == 0) return VIT_HIP_OK;
    vit::LdsUpdateArgs a{};
    a.symbols = (const uint8_t*)d_symbols;
    a.sym_frame_stride_bytes = sym_stride * (size_t)h->soft_bytes;
    a.sym_total_bytes = (frames - 1) * a.sym_frame_stride_bytes + n_steps * (size_t)h->R * (size_t)h->soft_bytes;
    a.decisions = d_decisions;
    a.dec_frame_stride_words = rows * (size_t)h->W;
    a.dec_row0 = row0;
    a.metrics_io = d_metrics;
    a.renorm_sum = d_renorm;
    a.start_state = d_start;
    a.pattern = h->d_pattern;
    a.K = h->K;
    a.n_steps = (int)n_steps;
    a.reset = reset ? 1 : 0;
    a.cfg = h->cfg;
    return h->shift ? launch_lds_update_r<8>(h->R, a, frames, h->N, st) : launch_lds_update_r<0>(h->R, a, frames, h->N, st);
}

int lds_chainback(vit_hip_handle h, const uint64_t* d_decisions, size_t frames, size_t L, uint8_t* d_out,
                  const uint32_t* d_end, hipStream_t st) {
    if (frames == 0 || L == 0) return VIT_HIP_OK;
    vit::LdsChainbackArgs a{};
    a.decisions = d_decisions;
    a.dec_frame_stride_words = (L + (size_t)h->K - 1) * (size_t)h->W;
    a.out = d_out;
    a.end_state = d_end;
    a.frames = frames;
    a.L = L;
    a.K = h->K;
    const int block = 64;
    hipLaunchKernelGGL(vit::lds_chainback_kernel, dim3((unsigned)((frames + block - 1) / block)), dim3(block), 0, st, a);
    VIT_HIP_CHECK(hipGetLastError());
    return VIT_HIP_OK;
}

// run-time compiled PLAN_REG for polynomials outside the ahead-of-time table (reg_jit.hpp)
// package_only: load an install-time precompiled code object if the package cache holds one; compile nothing
bool try_reg_jit(vit_hip_handle h, bool package_only) {
    // (a handle on the GENERIC kernels asks again when the caller allows compiling: kernels specialised for its polynomials are faster)
    if (h->reg_ok && !(h->reg_code.generic && !package_only)) return true;
    if (!h->linear || !vit::reg_jit_supported(h->K, h->R)) return false;
    std::string err;
    const vit::RegJitModule* m = vit::reg_jit_get(h->K, h->R, h->G, h->shift, h->device, package_only, err, &h->reg_origin);
    bool generic = false;
    if (!m && vit::reg_generic_supported(h->K, h->R)) {
        // no kernels specialised for these polynomials (and, unless package_only, no compiler to make them): the GENERIC code object
        // of this (K, R) -- polynomials read from the kernel arguments (RegSpec::GENERIC) -- if the package cache holds it
        const uint32_t zero[6] = {0, 0, 0, 0, 0, 0};
        std::string err2;
        m = vit::reg_jit_get(h->K, h->R, zero, h->shift, h->device, true, err2, &h->reg_origin);
        generic = m != nullptr;
    }
    if (!m) {
        if (h->reg_ok) return true;             // the upgrade failed: the generic kernels stay
        if (!package_only) g_last_error = err;
        return false;
    }
    h->reg_code.id = -1;
    h->reg_code.K = h->K;
    h->reg_code.R = h->R;
    h->reg_code.tile = h->K < 7 ? 128 : 32;
    h->reg_code.jit = m;
    h->reg_code.generic = generic;
    for (int i = 0; i < 6; ++i) h->reg_code.G[i] = i < h->R ? h->G[i] : 0u;
    h->reg_ok = true;
    return true;
}

int ensure_scratch(vit_hip_handle h, size_t bytes) {
    if (bytes <= h->scratch_bytes) return VIT_HIP_OK;
    if (h->d_scratch) VIT_HIP_CHECK(hipFree(h->d_scratch));
    h->d_scratch = nullptr;
    h->scratch_bytes = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    VIT_HIP_CHECK(hipMalloc(&h->d_scratch, want));
    h->scratch_bytes = want;
    return VIT_HIP_OK;
}

int ensure_stage(vit_hip_handle h, size_t bytes) {
    if (bytes <= h->stage_bytes) return VIT_HIP_OK;
    if (h->h_stage) VIT_HIP_CHECK(hipHostFree(h->h_stage));
    h->h_stage = nullptr;
    h->stage_bytes = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    VIT_HIP_CHECK(hipHostMalloc(&h->h_stage, want, hipHostMallocDefault));
    h->stage_bytes = want;
    return VIT_HIP_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// frame route: the device row store holds rows [0, rows) of the frame (contents kept when it grows)
int ensure_rows(vit_hip_handle h, size_t rows) {
    if (rows <= h->rows_cap) return VIT_HIP_OK;
    const size_t want = rows + rows / 2 + 64;
    uint64_t* p = nullptr;
    VIT_HIP_CHECK(hipMalloc((void**)&p, want * (size_t)h->W * 8));
    if (h->d_rows) {
        VIT_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (hipMemcpy(p, h->d_rows, h->rows_cap * (size_t)h->W * 8, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(p); return fail(VIT_HIP_ERR_RUNTIME, "row store copy failed"); }
        (void)hipFree(h->d_rows);
    }
    h->d_rows = p;
    h->rows_cap = want;
    return VIT_HIP_OK;
}

int ensure_map(vit_hip_handle h, size_t bytes) {
    if (bytes <= h->map_bytes) return VIT_HIP_OK;
    VIT_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->h_map) VIT_HIP_CHECK(hipHostFree(h->h_map));
    h->h_map = nullptr;
    h->map_bytes = 0;
    h->spec_valid = false;
    const size_t want = bytes + bytes / 2 + 4096;
    VIT_HIP_CHECK(hipHostMalloc(&h->h_map, want, hipHostMallocMapped | hipHostMallocCoherent));
    memset(h->h_map, 0, 256);
    h->map_bytes = want;
    return VIT_HIP_OK;
}

int read_soft(const void* p, size_t idx, int soft_bytes) {
    return soft_bytes == 1 ? (int)((const int8_t*)p)[idx] : (int)((const int16_t*)p)[idx];
}

unsigned parity_u32(uint32_t x) { return (unsigned)__builtin_popcount(x) & 1u; }

}  // namespace

namespace vit {
// depuncturing gather (examples/helpers/puncture_code_helpers.h:17-55 for a batch): out[f][k] = in[f][idx[k]] or 0.  One
// thread makes 8 consecutive output symbols of one frame (16-byte / 8-byte store); the index map is read once per 8 symbols
// as two 16-byte loads and stays in L2; source reads are consecutive because the map is monotonic.
template <typename T>
__global__ void depuncture_kernel(const T* __restrict__ in, size_t in_stride, const int32_t* __restrict__ idx, size_t n_out,
                                  size_t frames, T* __restrict__ out) {
    const size_t chunks = (n_out + 7) / 8;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= chunks * frames) return;
    const size_t f = gid / chunks, k0 = (gid % chunks) * 8;
    const T* src = in + f * in_stride;
    T* dst = out + f * n_out + k0;
    T v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const size_t k = k0 + i;
        const int32_t j = k < n_out ? idx[k] : -1;
        v[i] = j >= 0 ? src[j] : (T)0;
    }
    if (k0 + 8 <= n_out && ((uintptr_t)dst % (8 * sizeof(T))) == 0) {
        typedef T vec8 __attribute__((ext_vector_type(8)));
        vec8 w;
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = v[i];
        *(vec8*)dst = w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (k0 + i < n_out) dst[i] = v[i];
    }
}
// ViterbiDecoder_Core::reset (core.h:202-211) for a batch: metrics [F][N] error_t
template <typename error_t>
__global__ void reset_kernel(error_t* met, const uint32_t* start, size_t frames, uint32_t N, uint32_t init_start, uint32_t init_non_start) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= frames * N) return;
    const size_t f = idx / N;
    const uint32_t s = (uint32_t)(idx % N);
    const uint32_t st = start ? (start[f] & (N - 1u)) : 0u;
    met[idx] = (error_t)(s == st ? init_start : init_non_start);
}

template <typename soft_t>
int launch_synth(int R, const SynthArgs& a, unsigned blocks, hipStream_t st) {
    switch (R) {
        case 1: hipLaunchKernelGGL((synth_kernel<soft_t, 1>), dim3(blocks), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((synth_kernel<soft_t, 2>), dim3(blocks), dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((synth_kernel<soft_t, 3>), dim3(blocks), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((synth_kernel<soft_t, 4>), dim3(blocks), dim3(256), 0, st, a); break;
        case 5: hipLaunchKernelGGL((synth_kernel<soft_t, 5>), dim3(blocks), dim3(256), 0, st, a); break;
        case 6: hipLaunchKernelGGL((synth_kernel<soft_t, 6>), dim3(blocks), dim3(256), 0, st, a); break;
        case 7: hipLaunchKernelGGL((synth_kernel<soft_t, 7>), dim3(blocks), dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL((synth_kernel<soft_t, 8>), dim3(blocks), dim3(256), 0, st, a); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
}  // namespace vit

extern "C" {

const char* vit_hip_last_error(void) { return g_last_error.c_str(); }

int vit_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int vit_hip_create_impl(int K, int R, int soft_bytes, int error_bytes, const void* branch_table, const void* config,
                   int device, vit_hip_handle* out) {
    if (!out) return fail(VIT_HIP_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!branch_table || !config) return fail(VIT_HIP_ERR_INVALID_ARG, "branch_table/config is NULL");
    if (K < 2 || K > 16) return fail(VIT_HIP_ERR_UNSUPPORTED, "constraint length K must be 2..16 (the 2^(K-1) state metrics of a frame pair live in one CU's LDS)");
    if (R < 1 || R > 8) return fail(VIT_HIP_ERR_UNSUPPORTED, "code rate R must be 1..8");
    if (!((soft_bytes == 2 && error_bytes == 2) || (soft_bytes == 1 && error_bytes == 1)))
        return fail(VIT_HIP_ERR_UNSUPPORTED, "(soft_t,error_t) must be (int16_t,uint16_t) or (int8_t,uint8_t)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(VIT_HIP_ERR_NO_DEVICE, "no HIP device available (this library has no CPU decode path)");
    if (device < 0 || device >= ndev) return fail(VIT_HIP_ERR_INVALID_ARG, "device index out of range");

    vit_hip_decoder* h = new vit_hip_decoder();
    h->K = K; h->R = R; h->soft_bytes = soft_bytes; h->error_bytes = error_bytes; h->device = device;
    h->N = 1 << (K - 1); h->H = h->N / 2; h->W = h->N >= 64 ? h->N / 64 : 1;
    h->shift = soft_bytes == 1 ? 8 : 0;
    const int Ht = h->H > 0 ? h->H : 1;   // K=2 still stores one half-state

    // the table is two-valued: low == value at half-state 0 (parity of 0 is 0), high == the other value
    const int low = read_soft(branch_table, 0, soft_bytes);
    int high = low;
    bool have_high = false;
    for (int i = 0; i < R; ++i)
        for (int s = 0; s < Ht; ++s) {
            const int v = read_soft(branch_table, (size_t)i * Ht + s, soft_bytes);
            if (v != low) {
                if (!have_high) { high = v; have_high = true; }
                else if (v != high) { delete h; return fail(VIT_HIP_ERR_INVALID_ARG, "branch table holds more than two distinct values"); }
            }
        }
    for (int i = 0; i < R; ++i)
        if (read_soft(branch_table, (size_t)i * Ht, soft_bytes) != low) {
            delete h;
            return fail(VIT_HIP_ERR_INVALID_ARG, "branch table row does not start with the low value");
        }
    if (!have_high) high = low;  // degenerate code: every expected symbol is `low`
    h->high = high; h->low = low;
    h->pattern.assign(Ht, 0);
    for (int i = 0; i < R; ++i)
        for (int s = 0; s < Ht; ++s)
            if (have_high && read_soft(branch_table, (size_t)i * Ht + s, soft_bytes) == high) h->pattern[s] |= (uint16_t)(1u << i);

    // recover the polynomials (middle bits from the unit half-states; bit 0 and bit K-1 are implied by the butterfly
    // identity the reference relies on, viterbi_decoder_scalar.h:85-95) and check that the table is linear
    for (int i = 0; i < R; ++i) {
        uint32_t g = 1u | (1u << (K - 1));
        for (int k = 0; k + 2 < K; ++k)
            if ((h->pattern[(size_t)1 << k] >> i) & 1u) g |= 1u << (k + 1);
        h->G[i] = g;
    }
    h->linear = true;
    for (int s = 0; s < Ht && h->linear; ++s)
        for (int i = 0; i < R; ++i)
            if (parity_u32(((uint32_t)s << 1) & h->G[i] & ~(1u | (1u << (K - 1)))) != ((h->pattern[s] >> i) & 1u)) { h->linear = false; break; }

    for (int k = 0; k < 4; ++k)
        h->cfg_raw[k] = error_bytes == 1 ? (uint32_t)((const uint8_t*)config)[k] : (uint32_t)((const uint16_t*)config)[k];
    h->cfg.max_error = (uint16_t)(h->cfg_raw[0] << h->shift);
    h->cfg.init_start = (uint16_t)(h->cfg_raw[1] << h->shift);
    h->cfg.init_non_start = (uint16_t)(h->cfg_raw[2] << h->shift);
    h->cfg.threshold = (uint16_t)(h->cfg_raw[3] << h->shift);
    h->cfg.high = (int16_t)(uint16_t)((uint32_t)high << h->shift);
    h->cfg.low = (int16_t)(uint16_t)((uint32_t)low << h->shift);

    DeviceGuard guard(device);
    if (!guard.ok) { delete h; return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed"); }
    if (hipMalloc((void**)&h->d_pattern, Ht * sizeof(uint16_t)) != hipSuccess ||
        hipMemcpy(h->d_pattern, h->pattern.data(), Ht * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        if (h->d_pattern) (void)hipFree(h->d_pattern);
        delete h;
        return fail(VIT_HIP_ERR_RUNTIME, "device allocation failed in vit_hip_create");
    }
    h->reg_ok = h->linear && vit::reg_code_supported(K, R) && vit::reg_code_init(&h->reg_code, K, R, h->G, h->cfg);
    if (!h->reg_ok && h->linear && vit::reg_jit_supported(K, R)) {
        // a code object precompiled at install time (package cache, reg_jit.hpp) is as good as a built-in one: no compiler runs
        const char* e = getenv("VIT_HIP_JIT");
        // compiling: opt-in at create time (vit_hip_set_plan(PLAN_REG) always tries); either way the search ends at the package's GENERIC kernels
        if (e && *e == '1') (void)try_reg_jit(h, false);
        else (void)try_reg_jit(h, true);
    }
    h->lds2_ok = h->linear && vit::lds2_supported(K, R);   // the group-B tables rely on the code being linear
    h->plan = h->reg_ok ? VIT_HIP_PLAN_REG : h->lds2_ok ? VIT_HIP_PLAN_LDS2 : VIT_HIP_PLAN_LDS;
    *out = h;
    return VIT_HIP_OK;
}

int vit_hip_destroy(vit_hip_handle h) {
    if (!h) return VIT_HIP_OK;
    DeviceGuard guard(h->device);
    if (h->d_pattern) (void)hipFree(h->d_pattern);
    if (h->d_scratch) (void)hipFree(h->d_scratch);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->h_map) (void)hipHostFree(h->h_map);
    if (h->d_rows) (void)hipFree(h->d_rows);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return VIT_HIP_OK;
}

const char* vit_hip_plan_note(vit_hip_handle h) {
    if (!h) return "NULL handle";
    thread_local std::string note;
    char head[160];
    snprintf(head, sizeof(head), "K=%d R=%d: %s", h->K, h->R,
             h->plan == VIT_HIP_PLAN_REG ? "PLAN_REG (state metrics in registers, 32-128 frames per wavefront)"
             : h->plan == VIT_HIP_PLAN_LDS2 ? "PLAN_LDS2 (frame pair per workgroup, four trellis steps per barrier)"
                                            : "PLAN_LDS, the COMPATIBILITY plan (one frame per workgroup: ~20x slower per state update than the other plans)");
    note = head;
    if (h->plan == VIT_HIP_PLAN_REG && h->reg_code.jit) {
        const bool pkg = h->reg_origin.compare(0, vit::package_cache_dir().size(), vit::package_cache_dir()) == 0;
        note += std::string(pkg ? "; kernels precompiled at install time, loaded from the package cache " : "; kernels compiled at run time, loaded from the user cache ") + h->reg_origin;
        if (h->reg_code.generic)
            note += "; these are the GENERIC kernels of this (K, R) -- polynomials read at run time, 2 - 20 % behind kernels specialised for them: "
                    "python -m viterbidecodercpp_amd.tools.precompile K R G... at install time, or VIT_HIP_JIT=1 before vit_hip_create (hipcc), gets those";
    }
    if (h->plan == VIT_HIP_PLAN_LDS) {
        if (h->reg_ok || (h->linear && vit::reg_jit_supported(h->K, h->R)))
            note += h->reg_ok ? "; the register plan is available: vit_hip_set_plan(h, VIT_HIP_PLAN_REG)"
                              : "; a register-plan instantiation for these polynomials can be compiled at run time: vit_hip_set_plan(h, VIT_HIP_PLAN_REG) "
                                "(hipcc, 30-60 s once, cached on disk), or VIT_HIP_JIT=1 before vit_hip_create; or at install time, for hosts without a "
                                "compiler: python -m viterbidecodercpp_amd.tools.precompile K R G... (vit_hip_precompile)";
        else if (h->lds2_ok)
            note += "; PLAN_LDS2 is available: vit_hip_set_plan(h, VIT_HIP_PLAN_LDS2)";
        else if (!h->linear)
            note += "; no faster plan: the branch table is not that of a linear convolutional code";
        else
            note += "; no faster plan exists for this (K, R): the register plan serves K = 2..9 with R <= 6, PLAN_LDS2 K = 10..16 with R <= 6";
    }
    return note.c_str();
}

int vit_hip_get_info(vit_hip_handle h, vit_hip_info* info) {
    if (!h || !info) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL argument");
    memset(info, 0, sizeof(*info));
    info->K = h->K; info->R = h->R; info->soft_bytes = h->soft_bytes; info->error_bytes = h->error_bytes;
    info->num_states = h->N; info->decision_words = h->W; info->device = h->device; info->plan = h->plan;
    info->soft_decision_high = h->high; info->soft_decision_low = h->low;
    for (int i = 0; i < h->R && i < 16; ++i) info->polynomials[i] = h->linear ? h->G[i] : 0u;
    info->table_is_linear = h->linear ? 1 : 0;
    info->workspace_tile_frames = h->plan == VIT_HIP_PLAN_REG ? h->reg_code.tile : h->plan == VIT_HIP_PLAN_LDS2 ? 2 : 1;
    return VIT_HIP_OK;
}

static int vit_hip_set_plan_impl(vit_hip_handle h, int plan) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (plan == VIT_HIP_PLAN_AUTO) plan = h->reg_ok ? VIT_HIP_PLAN_REG : h->lds2_ok ? VIT_HIP_PLAN_LDS2 : VIT_HIP_PLAN_LDS;
    // (a handle on the package's GENERIC kernels: asking for the register plan by name looks for kernels specialised for its polynomials --
    // package cache, user cache, compiler -- and keeps the generic ones when there are none)
    if (plan == VIT_HIP_PLAN_REG && (!h->reg_ok || (h->reg_code.jit && h->reg_code.generic))) {
        DeviceGuard guard(h->device);
        g_last_error.clear();
        if (!guard.ok || !try_reg_jit(h, false))
            return fail(VIT_HIP_ERR_UNSUPPORTED, "PLAN_REG not available for this code: " +
                        (g_last_error.empty() ? std::string("K must be 2..9 and R <= 6, linear branch table") : g_last_error));
    }
    if (plan == VIT_HIP_PLAN_LDS2 && !h->lds2_ok)
        return fail(VIT_HIP_ERR_UNSUPPORTED, "PLAN_LDS2 serves K = 10..16 with R <= 6 and a linear branch table (see kernels_lds2.hpp)");
    if (plan != VIT_HIP_PLAN_LDS && plan != VIT_HIP_PLAN_REG && plan != VIT_HIP_PLAN_LDS2)
        return fail(VIT_HIP_ERR_INVALID_ARG, "unknown plan");
    h->plan = plan;
    return VIT_HIP_OK;
}

size_t vit_hip_blob_bytes(int K, int R, int soft_bytes, int error_bytes) {
    if (K < 2 || K > 30 || R < 1 || R > 64 || soft_bytes < 1 || soft_bytes > 8 || error_bytes < 1 || error_bytes > 8) return 0;
    const size_t Ht = K > 2 ? ((size_t)1 << (K - 2)) : 1;
    return sizeof(BlobHeader) + (size_t)R * Ht * (size_t)soft_bytes + 4 * (size_t)error_bytes;
}

int vit_hip_pack_blob(int K, int R, int soft_bytes, int error_bytes, const void* branch_table, const void* config,
                      void* blob, size_t blob_bytes) {
    const size_t need = vit_hip_blob_bytes(K, R, soft_bytes, error_bytes);
    if (!need || !branch_table || !config || !blob) return fail(VIT_HIP_ERR_INVALID_ARG, "bad blob arguments");
    if (blob_bytes < need) return fail(VIT_HIP_ERR_INVALID_ARG, "blob buffer too small");
    BlobHeader hd{BLOB_MAGIC, K, R, soft_bytes, error_bytes};
    uint8_t* p = (uint8_t*)blob;
    memcpy(p, &hd, sizeof(hd));
    const size_t tb = need - sizeof(hd) - 4 * (size_t)error_bytes;
    memcpy(p + sizeof(hd), branch_table, tb);
    memcpy(p + sizeof(hd) + tb, config, 4 * (size_t)error_bytes);
    return VIT_HIP_OK;
}

static int vit_hip_create_from_blob_impl(const void* blob, size_t blob_bytes, int device, vit_hip_handle* out) {
    if (!blob || blob_bytes < sizeof(BlobHeader)) return fail(VIT_HIP_ERR_INVALID_ARG, "blob too small");
    BlobHeader hd;
    memcpy(&hd, blob, sizeof(hd));
    if (hd.magic != BLOB_MAGIC) return fail(VIT_HIP_ERR_INVALID_ARG, "bad blob magic");
    // the header is untrusted: validate before any size arithmetic
    if (hd.K < 2 || hd.K > 16 || hd.R < 1 || hd.R > 8 ||
        !((hd.soft_bytes == 2 && hd.error_bytes == 2) || (hd.soft_bytes == 1 && hd.error_bytes == 1)))
        return fail(VIT_HIP_ERR_INVALID_ARG, "corrupt blob header");
    const size_t need = vit_hip_blob_bytes(hd.K, hd.R, hd.soft_bytes, hd.error_bytes);
    if (!need || blob_bytes < need) return fail(VIT_HIP_ERR_INVALID_ARG, "blob truncated");
    const uint8_t* p = (const uint8_t*)blob + sizeof(hd);
    const size_t tb = need - sizeof(hd) - 4 * (size_t)hd.error_bytes;
    // copies keep the caller free of alignment requirements
    std::vector<uint16_t> table((tb + 1) / 2), cfg(4);
    memcpy(table.data(), p, tb);
    memcpy(cfg.data(), p + tb, 4 * (size_t)hd.error_bytes);
    return vit_hip_create(hd.K, hd.R, hd.soft_bytes, hd.error_bytes, table.data(), cfg.data(), device, out);
}

size_t vit_hip_workspace_bytes(vit_hip_handle h, size_t frames, size_t L) {
    if (!h) return 0;
    if (h->plan == VIT_HIP_PLAN_REG) return vit::reg_workspace_bytes(h->reg_code, frames, L);
    if (h->plan == VIT_HIP_PLAN_LDS2) return align_up(vit::lds2_workspace_bytes(h->K, frames, L), 256);
    return align_up(frames * (L + (size_t)h->K - 1) * (size_t)h->W * 8, 256);
}

size_t vit_hip_workspace_slab_bytes(vit_hip_handle h, size_t L) {
    if (!h) return 0;
    // PLAN_REG: one tile of frames; PLAN_LDS2: one frame pair (rows x T dwords, T >= 64: a multiple of 256 bytes); PLAN_LDS:
    // the reference layout [F][S][W], dense -- one frame's rows, NOT rounded up to the 256 bytes the whole workspace is
    if (h->plan == VIT_HIP_PLAN_REG) return vit::reg_workspace_bytes(h->reg_code, (size_t)h->reg_code.tile, L);
    if (h->plan == VIT_HIP_PLAN_LDS2) return vit::lds2_workspace_bytes(h->K, 2, L);
    return (L + (size_t)h->K - 1) * (size_t)h->W * 8;
}

namespace {
// reset + update (d_metrics_in == null, first_step == 0) or resumed update: one body behind both entry points
int update_batch_impl(vit_hip_handle h, const void* d_symbols, size_t sym_stride, size_t frames, size_t first_step, size_t n_steps,
                      size_t L, void* d_workspace, size_t workspace_bytes, const void* d_metrics_in, void* d_metrics_out,
                      uint64_t* d_renorm_sum, const uint32_t* d_start_state, vit_hip_stream_t stream) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (frames == 0) return VIT_HIP_OK;
    if (!d_symbols || !d_workspace) return fail(VIT_HIP_ERR_INVALID_ARG, "d_symbols/d_workspace is NULL");
    if (first_step + n_steps > L + (size_t)h->K - 1) return fail(VIT_HIP_ERR_INVALID_ARG, "steps exceed traceback length + K-1");
    if (n_steps > 0x7FFFFFF0u || frames > 0x7FFFFFF0u) return fail(VIT_HIP_ERR_INVALID_ARG, "batch too large");
    if (sym_stride == 0) sym_stride = n_steps * (size_t)h->R;
    if (sym_stride < n_steps * (size_t)h->R) return fail(VIT_HIP_ERR_INVALID_ARG, "symbol_frame_stride shorter than one chunk");
    if (workspace_bytes < vit_hip_workspace_bytes(h, frames, L)) return fail(VIT_HIP_ERR_WORKSPACE, "workspace too small");
    if (((uintptr_t)d_workspace & 255u) != 0) return fail(VIT_HIP_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    if (h->soft_bytes == 2 && ((uintptr_t)d_symbols & 1u)) return fail(VIT_HIP_ERR_INVALID_ARG, "int16 symbols must be 2-byte aligned");
    if (n_steps == 0) return VIT_HIP_OK;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    if (h->plan == VIT_HIP_PLAN_REG) {
        const int rc = vit::reg_update(h->reg_code, h->cfg, h->shift, d_symbols, sym_stride, frames, first_step, n_steps, L,
                                       d_workspace, d_metrics_in, d_metrics_out, d_renorm_sum, d_start_state, st);
        if (rc != 0) return fail(VIT_HIP_ERR_RUNTIME, rc == -2 ? "register-plan update: a tile's symbols exceed 32-bit offsets"
                                                                : "register-plan update launch failed");
        return VIT_HIP_OK;
    }
    if (h->plan == VIT_HIP_PLAN_LDS2) {
        const int rc = vit::lds2_update(h->K, h->R, h->cfg, h->shift, h->d_pattern, h->pattern.data(), d_symbols, sym_stride, frames, first_step,
                                        n_steps, L, d_workspace, d_metrics_in, d_metrics_out, d_renorm_sum, d_start_state, st);
        if (rc != 0) return fail(VIT_HIP_ERR_RUNTIME, "PLAN_LDS2 update launch failed");
        return VIT_HIP_OK;
    }
    // PLAN_LDS reads and writes its metrics through one buffer
    if (d_metrics_in && d_metrics_in != d_metrics_out) return fail(VIT_HIP_ERR_INVALID_ARG, "PLAN_LDS resumes in place");
    return lds_update(h, d_symbols, sym_stride, frames, n_steps, L + (size_t)h->K - 1, (uint32_t)first_step, (uint64_t*)d_workspace,
                      d_metrics_out, d_metrics_in == nullptr, d_renorm_sum, d_start_state, st);
}
}  // namespace

int vit_hip_update_batch(vit_hip_handle h, const void* d_symbols, size_t frames, size_t n_steps, size_t L,
                         void* d_workspace, size_t workspace_bytes, void* d_final_metrics, uint64_t* d_renorm_sum,
                         const uint32_t* d_start_state, vit_hip_stream_t stream) {
    return update_batch_impl(h, d_symbols, 0, frames, 0, n_steps, L, d_workspace, workspace_bytes, nullptr, d_final_metrics,
                             d_renorm_sum, d_start_state, stream);
}

int vit_hip_update_batch_resume(vit_hip_handle h, const void* d_symbols, size_t symbol_frame_stride, size_t frames,
                                size_t first_step, size_t n_steps, size_t L, void* d_workspace, size_t workspace_bytes,
                                void* d_metrics_inout, uint64_t* d_renorm_sum, vit_hip_stream_t stream) {
    if (!d_metrics_inout) return fail(VIT_HIP_ERR_INVALID_ARG, "d_metrics_inout is NULL (vit_hip_reset_batch fills it for step 0)");
    return update_batch_impl(h, d_symbols, symbol_frame_stride, frames, first_step, n_steps, L, d_workspace, workspace_bytes,
                             d_metrics_inout, d_metrics_inout, d_renorm_sum, nullptr, stream);
}

// alt_kernel: the code's other chainback kernel (K = 7: the LDS-ring body, K = 9: the cooperative one); ignored by codes with one
static int chainback_batch_impl(vit_hip_handle h, const void* d_workspace, size_t frames, size_t L, uint8_t* d_bytes_out,
                                const uint32_t* d_end_state, vit_hip_stream_t stream, unsigned wave_priority, bool alt_kernel = false) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (frames == 0 || L == 0) return VIT_HIP_OK;
    if (!d_workspace || !d_bytes_out) return fail(VIT_HIP_ERR_INVALID_ARG, "d_workspace/d_bytes_out is NULL");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    if (h->plan == VIT_HIP_PLAN_REG) {
        const int rc = vit::reg_chainback(h->reg_code, d_workspace, frames, L, d_bytes_out, d_end_state, st, wave_priority, alt_kernel);
        if (rc != 0) return fail(VIT_HIP_ERR_RUNTIME, "register-plan chainback launch failed");
        return VIT_HIP_OK;
    }
    if (h->plan == VIT_HIP_PLAN_LDS2) {
        const int rc = vit::lds2_chainback(h->K, d_workspace, frames, L, d_bytes_out, d_end_state, st);
        if (rc != 0) return fail(VIT_HIP_ERR_RUNTIME, "PLAN_LDS2 chainback launch failed");
        return VIT_HIP_OK;
    }
    return lds_chainback(h, (const uint64_t*)d_workspace, frames, L, d_bytes_out, d_end_state, st);
}

int vit_hip_chainback_batch(vit_hip_handle h, const void* d_workspace, size_t frames, size_t L, uint8_t* d_bytes_out,
                            const uint32_t* d_end_state, vit_hip_stream_t stream) {
    return chainback_batch_impl(h, d_workspace, frames, L, d_bytes_out, d_end_state, stream, 0);
}

int vit_hip_chainback_batch_ex(vit_hip_handle h, const void* d_workspace, size_t frames, size_t L, uint8_t* d_bytes_out,
                               const uint32_t* d_end_state, vit_hip_stream_t stream, int kernel) {
    if (kernel != VIT_HIP_KERNEL_CHAINBACK && kernel != VIT_HIP_KERNEL_CHAINBACK_ALT)
        return fail(VIT_HIP_ERR_INVALID_ARG, "kernel must be VIT_HIP_KERNEL_CHAINBACK or VIT_HIP_KERNEL_CHAINBACK_ALT");
    return chainback_batch_impl(h, d_workspace, frames, L, d_bytes_out, d_end_state, stream, 0, kernel == VIT_HIP_KERNEL_CHAINBACK_ALT);
}

int vit_hip_decode_batch(vit_hip_handle h, const void* d_symbols, size_t frames, size_t L, void* d_workspace,
                         size_t workspace_bytes, uint8_t* d_bytes_out, void* d_final_metrics, uint64_t* d_renorm_sum,
                         const uint32_t* d_end_state, vit_hip_stream_t stream) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    const int rc = vit_hip_update_batch(h, d_symbols, frames, L + (size_t)h->K - 1, L, d_workspace, workspace_bytes,
                                        d_final_metrics, d_renorm_sum, nullptr, stream);
    if (rc != VIT_HIP_OK) return rc;
    return vit_hip_chainback_batch(h, d_workspace, frames, L, d_bytes_out, d_end_state, stream);
}

int vit_hip_export_decisions(vit_hip_handle h, const void* d_workspace, size_t frames, size_t n_steps, size_t L,
                             uint64_t* d_decisions, vit_hip_stream_t stream) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (frames == 0 || n_steps == 0) return VIT_HIP_OK;
    if (!d_workspace || !d_decisions) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL buffer");
    if (n_steps > L + (size_t)h->K - 1) return fail(VIT_HIP_ERR_INVALID_ARG, "n_steps exceeds traceback length + K-1");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    if (h->plan == VIT_HIP_PLAN_REG) {
        const int rc = vit::reg_export(h->reg_code, d_workspace, frames, n_steps, L, d_decisions, st);
        if (rc != 0) return fail(VIT_HIP_ERR_RUNTIME, "register-plan export launch failed");
        return VIT_HIP_OK;
    }
    if (h->plan == VIT_HIP_PLAN_LDS2) {
        const int rc = vit::lds2_export(h->K, d_workspace, frames, n_steps, L, d_decisions, st);
        if (rc != 0) return fail(VIT_HIP_ERR_RUNTIME, "PLAN_LDS2 export launch failed");
        return VIT_HIP_OK;
    }
    const size_t rows = L + (size_t)h->K - 1;
    const size_t W8 = (size_t)h->W * 8;
    VIT_HIP_CHECK(hipMemcpy2DAsync(d_decisions, n_steps * W8, d_workspace, rows * W8, n_steps * W8, frames,
                                   hipMemcpyDeviceToDevice, st));
    return VIT_HIP_OK;
}

int vit_hip_depuncture_batch(vit_hip_handle h, const void* d_punctured, size_t punctured_per_frame,
                             const int32_t* d_source_index, size_t symbols_per_frame, size_t frames, void* d_symbols_out,
                             vit_hip_stream_t stream) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (frames == 0 || symbols_per_frame == 0) return VIT_HIP_OK;
    if (!d_punctured || !d_source_index || !d_symbols_out) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL buffer");
    if (symbols_per_frame % (size_t)h->R != 0)
        return fail(VIT_HIP_ERR_INVALID_ARG, "symbols_per_frame must be a multiple of the code rate R");
    const size_t chunks = (symbols_per_frame + 7) / 8;
    if (chunks * frames > 0x7FFFFFFFull * 256ull) return fail(VIT_HIP_ERR_INVALID_ARG, "batch too large for one launch");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((chunks * frames + 255) / 256);
    if (h->soft_bytes == 2)
        hipLaunchKernelGGL(vit::depuncture_kernel<int16_t>, dim3(blocks), dim3(256), 0, st, (const int16_t*)d_punctured,
                           punctured_per_frame, d_source_index, symbols_per_frame, frames, (int16_t*)d_symbols_out);
    else
        hipLaunchKernelGGL(vit::depuncture_kernel<int8_t>, dim3(blocks), dim3(256), 0, st, (const int8_t*)d_punctured,
                           punctured_per_frame, d_source_index, symbols_per_frame, frames, (int8_t*)d_symbols_out);
    VIT_HIP_CHECK(hipGetLastError());
    return VIT_HIP_OK;
}

}  // extern "C"

struct vit_hip_pipeline {
    vit_hip_handle h = nullptr;
    size_t max_frames = 0, L = 0, ws_bytes = 0;
    // schedule (fixed at create time from max_frames): n_ws decision workspaces used round robin, n_upd update streams
    int n_ws = 2, n_upd = 1;
    static constexpr int MAX_UPD = 3, MAX_WS = 4;
    void* ws[MAX_WS] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t s_upd[MAX_UPD] = {nullptr, nullptr, nullptr}, s_cb = nullptr;
    hipEvent_t upd_done[MAX_WS] = {nullptr, nullptr, nullptr, nullptr}, cb_done[MAX_WS] = {nullptr, nullptr, nullptr, nullptr};
    bool cb_pending[MAX_WS] = {false, false, false, false};
    bool cb_small = false;              // K = 7 beside update waves: the 32-register LDS-ring chainback kernel
    unsigned long long n = 0;
    size_t overlap_max_frames = 0;      // largest batch whose chainback is worth running beside the next update
    size_t two_updates_max_frames = 0;  // largest batch that leaves room for a second update kernel beside the first
    unsigned cb_wave_priority = 0;      // two-update schedule: the chainback kernel outranks the update waves
    size_t sub_frames = 0;              // a submitted batch is fed to the kernels in sub-batches of at most this many frames
    size_t last_first_frame = 0, last_frames = 0;   // frame range of the most recent sub-batch (the one ws[(n-1) % n_ws] holds)
    size_t sym_frame_bytes = 0, out_frame_bytes = 0;
    // optional per-batch timing (vit_hip_pipeline_set_timing): four events per submitted batch, resolved by sync()
    bool timing = false;
    struct Rec { hipEvent_t u0, u1, c0, c1; };
    std::vector<Rec> pending_recs;
    std::vector<hipEvent_t> event_pool;
    std::vector<float> t_update, t_chainback, t_complete;   // ms; t_complete: end of the batch's chainback since epoch
    hipEvent_t epoch = nullptr;          // start of the first timed batch's update
};

namespace {
hipEvent_t pipe_event(vit_hip_pipeline* p) {
    if (!p->event_pool.empty()) {
        hipEvent_t e = p->event_pool.back();
        p->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

extern "C" {

#ifdef VIT_HIP_EXPERIMENTS
// A/B builds only (make EXPERIMENTS=1; scripts/gpu_ab.sh): the environment fills whatever the caller's options left to the rules.
// The shipped library has no such switch: vit_hip_pipeline_create_ex is the supported override.
static void pipeline_options_from_env(vit_hip_pipeline_options* o) {
    auto flag = [](const char* name, int32_t* v, int32_t unset) {
        const char* e = getenv(name);
        if (e && *v == unset && (*e == '0' || *e == '1')) *v = *e - '0';
    };
    flag("VIT_HIP_PIPELINE_OVERLAP", &o->chainback_overlap, -1);
    flag("VIT_HIP_PIPELINE_SPLIT", &o->sub_batches, -1);
    flag("VIT_HIP_PIPELINE_CB_SMALL", &o->chainback_small_kernel, -1);
    flag("VIT_HIP_PIPELINE_CB_PRIO", &o->chainback_wave_priority, -1);
    if (const char* e = getenv("VIT_HIP_PIPELINE_UPDATES")) if (o->update_streams == 0 && *e >= '1' && *e <= '3') o->update_streams = *e - '0';
    if (const char* e = getenv("VIT_HIP_PIPELINE_WS")) if (o->workspaces == 0 && *e >= '2' && *e <= '4') o->workspaces = *e - '0';
}
#endif

static int vit_hip_pipeline_create_impl(vit_hip_handle h, size_t max_frames, size_t L, const vit_hip_pipeline_options* want,
                                        vit_hip_pipeline_t* out) {
    if (!h || !out || max_frames == 0) return fail(VIT_HIP_ERR_INVALID_ARG, "bad pipeline arguments");
    *out = nullptr;
    // the caller's options, read up to the size its build knows; anything beyond stays at "rule"
    vit_hip_pipeline_options opt{(uint32_t)sizeof(vit_hip_pipeline_options), -1, 0, -1, 0, -1, -1};
    if (want) {
        if (want->struct_size < 2 * sizeof(uint32_t)) return fail(VIT_HIP_ERR_INVALID_ARG, "vit_hip_pipeline_options.struct_size is not set");
        memcpy(&opt, want, want->struct_size < sizeof(opt) ? want->struct_size : sizeof(opt));
        if (opt.chainback_overlap < -1 || opt.chainback_overlap > 1 || opt.update_streams < 0 || opt.update_streams > 3 ||
            opt.sub_batches < -1 || opt.sub_batches > 1 || (opt.workspaces != 0 && (opt.workspaces < 2 || opt.workspaces > 4)) ||
            opt.chainback_small_kernel < -1 || opt.chainback_small_kernel > 1 || opt.chainback_wave_priority < -1 || opt.chainback_wave_priority > 1)
            return fail(VIT_HIP_ERR_INVALID_ARG, "vit_hip_pipeline_options: field out of range");
    }
#ifdef VIT_HIP_EXPERIMENTS
    pipeline_options_from_env(&opt);
#endif
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    vit_hip_pipeline* p = new (std::nothrow) vit_hip_pipeline();
    if (!p) return fail(VIT_HIP_ERR_RUNTIME, "out of host memory");
    p->h = h; p->max_frames = max_frames; p->L = L;
    // Rule 1 -- chainback beside the next update.  The overlap pays while the update waves leave a chainback wave its registers
    // (by the kernels' DESCRIPTORS, kernel_desc.hpp: 512 per SIMD) and LDS: PLAN_REG with at most two update waves per SIMD (three
    // at K = 7 with the 32-register LDS-ring chainback).  A larger batch fills the SIMDs by itself (K7 131072 frames: 8.70 ms
    // overlapped vs 8.10 ms back to back), and the PLAN_LDS update takes whole CUs: those batches run back to back on one stream.
    // PLAN_LDS2: wherever the update waves a CU's LDS admits leave the chainback's 24 registers on every SIMD (K = 11, 12, 14, 15:
    // four waves of at most 120 -- K15 4096 x 8192: 51.6 -> 50.3 ms per batch; K = 13: three of 144).
    // Rule 2 -- two updates in flight.  A batch of at most ONE update wave per SIMD (frames <= 4 x CUs x tile: the 32768-frame
    // share of BASELINE configs[3]) issues at the one-wave rate (5.27 cycles per packed instruction against 4.52 with two
    // waves, profiles/r2_dep_rate.txt): a second update stream and a third workspace put the next batch's update beside it
    // (hard8 32768 x 8192: 2.01 -> 1.87 ms per batch; HISTORY.md, round 3).
    {
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        const size_t per_wave = (h->plan == VIT_HIP_PLAN_REG && cus > 0) ? (size_t)4 * (size_t)cus * (size_t)h->reg_code.tile : 0;
        p->overlap_max_frames = 2 * per_wave;
        p->two_updates_max_frames = per_wave;
        // THREE update waves per SIMD and a chainback wave beside them: only with the K = 7 LDS-ring chainback kernel (3 x 152 + 32
        // of 512 registers by the kernel descriptors; 98304 x 8192: 161 Gbit/s overlapped against 139 back to back)
        if (per_wave > 0 && h->K == 7 && vit::reg_chainback_fits_beside_updates(h->reg_code, h->shift, 3, true)) p->overlap_max_frames = 3 * per_wave;
        // PLAN_LDS2 codes whose update kernel is capped at 120 registers: the 24-register chainback fits beside four of its waves
        if (h->plan == VIT_HIP_PLAN_LDS2 && vit::lds2_chainback_fits_beside_update(h->K, h->R, h->shift)) p->overlap_max_frames = (size_t)-1;
        if (opt.chainback_overlap >= 0) p->overlap_max_frames = opt.chainback_overlap ? (size_t)-1 : 0;
        if (opt.update_streams == 1) p->two_updates_max_frames = 0;                        // never two updates in flight
        if (opt.update_streams >= 2) p->two_updates_max_frames = p->overlap_max_frames;    // wherever the chainback is overlapped
    }
    // Rule 3 -- sub-batches.  Where two update waves leave no registers (or LDS) for a chainback wave (no built-in code since
    // round 5: LTE is capped at 240 registers, DAB's chainback ring is 12 KiB, CDMA 2000 fetches its branch metrics in sub-chunks;
    // a run-time compiled code may still land here), the chainback of a two-waves-per-SIMD batch cannot run beside the next update
    // at all: any batch of more than one wave per SIMD is fed to the kernels as sub-batches of one update wave per SIMD from the
    // two update streams (LTE 65536 x 8192, round 3: 118 - 120 -> 133 - 145 Gbit/s).
    p->sub_frames = max_frames;
    p->n_upd = max_frames <= p->two_updates_max_frames ? 2 : 1;
    if (h->plan == VIT_HIP_PLAN_REG && p->two_updates_max_frames > 0 && max_frames > p->two_updates_max_frames &&
        !vit::reg_chainback_fits_beside_updates(h->reg_code, h->shift, 2, /* K = 7: the LDS-ring kernel is the one that runs there */ h->K == 7)) {
        p->sub_frames = p->two_updates_max_frames;
        p->n_upd = 2;
    }
    {
        // one update wave per SIMD, whatever opt.update_streams did to two_updates_max_frames above
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        const size_t per_wave = (h->plan == VIT_HIP_PLAN_REG && cus > 0) ? (size_t)4 * (size_t)cus * (size_t)h->reg_code.tile : 0;
        if (opt.sub_batches == 1 && per_wave > 0 && max_frames > per_wave) { p->sub_frames = per_wave; p->n_upd = 2; }
        if (opt.sub_batches == 0 && p->sub_frames < max_frames) { p->sub_frames = max_frames; p->n_upd = 1; }
        if (opt.update_streams == 3 && per_wave > 0 && p->sub_frames <= per_wave) p->n_upd = 3;   // three update kernels in flight
    }
    p->n_ws = opt.workspaces ? opt.workspaces : p->n_upd + 1;
    // K = 7, chainback beside the update waves of ONE update kernel: the LDS-ring kernel (32 registers, 24 KiB of LDS) leaves
    // the update waves their SIMDs -- 65536 x 8192: 157 -> 160 Gbit/s over the register-ring kernel (160 registers), which stays
    // the kernel of a chainback that runs alone (7 % faster there) and of the two-update schedule (there it runs at the higher wave
    // priority and has to be FAST, not small: hard8 32768 x 8192 163 against 145 Gbit/s)
    p->cb_small = h->plan == VIT_HIP_PLAN_REG && h->K == 7 && p->n_upd == 1;
    if (opt.chainback_small_kernel >= 0) p->cb_small = opt.chainback_small_kernel == 1 && h->plan == VIT_HIP_PLAN_REG && h->K == 7;
    p->ws_bytes = vit_hip_workspace_bytes(h, p->sub_frames, L);
    p->sym_frame_bytes = (L + (size_t)h->K - 1) * (size_t)h->R * (size_t)h->soft_bytes;
    p->out_frame_bytes = (L + 7) / 8;
    p->cb_wave_priority = p->n_upd > 1 ? 1u : 0u;
    if (opt.chainback_wave_priority >= 0) p->cb_wave_priority = (unsigned)opt.chainback_wave_priority;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);           // hi = numerically lowest = highest priority
    bool ok = hipStreamCreateWithPriority(&p->s_cb, hipStreamNonBlocking, hi) == hipSuccess;   // the short bit chase gets out of the update's way
    for (int k = 0; k < p->n_upd && ok; ++k) ok = hipStreamCreateWithFlags(&p->s_upd[k], hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < p->n_ws && ok; ++k)
        ok = hipMalloc(&p->ws[k], p->ws_bytes) == hipSuccess &&
             hipEventCreateWithFlags(&p->upd_done[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&p->cb_done[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)vit_hip_pipeline_destroy(p);
        return fail(VIT_HIP_ERR_RUNTIME, "pipeline allocation failed (two or three decision workspaces of vit_hip_workspace_bytes each)");
    }
    *out = p;
    return VIT_HIP_OK;
}

static int vit_hip_pipeline_submit_impl(vit_hip_pipeline_t p, const void* d_symbols, size_t frames, uint8_t* d_bytes_out,
                            const uint32_t* d_end_state, void* done_event) {
    if (!p) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL pipeline");
    if (frames > p->max_frames) return fail(VIT_HIP_ERR_INVALID_ARG, "batch larger than the pipeline was created for");
    if (frames == 0) return VIT_HIP_OK;
    DeviceGuard guard(p->h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    // a batch goes to the kernels in sub-batches of at most sub_frames frames (one, unless pipeline_create chose to split)
    for (size_t f0 = 0; f0 < frames; f0 += p->sub_frames) {
        const size_t nf = frames - f0 < p->sub_frames ? frames - f0 : p->sub_frames;
        const bool last = f0 + nf >= frames;
        const uint8_t* sym = (const uint8_t*)d_symbols + f0 * p->sym_frame_bytes;
        uint8_t* out = d_bytes_out + f0 * p->out_frame_bytes;
        const uint32_t* es = d_end_state ? d_end_state + f0 : nullptr;
        const int k = (int)(p->n % (unsigned long long)p->n_ws);
        hipStream_t s_upd = p->s_upd[(int)(p->n % (unsigned long long)p->n_upd)];
        vit_hip_pipeline::Rec rec{nullptr, nullptr, nullptr, nullptr};
        // the four timing events go back to the pool on every early exit (an error below leaves the sub-batches already
        // enqueued in flight: the caller syncs -- or destroys -- the pipeline before touching the buffers again)
        struct RecGuard {
            vit_hip_pipeline* p; vit_hip_pipeline::Rec* r; bool armed = true;
            ~RecGuard() {
                if (!armed) return;
                for (hipEvent_t e : {r->u0, r->u1, r->c0, r->c1})
                    if (e && e != p->epoch) p->event_pool.push_back(e);
            }
        } rec_guard{p, &rec};
        if (p->timing) {
            rec.u0 = pipe_event(p); rec.u1 = pipe_event(p); rec.c0 = pipe_event(p); rec.c1 = pipe_event(p);
            if (!rec.u0 || !rec.u1 || !rec.c0 || !rec.c1) return fail(VIT_HIP_ERR_RUNTIME, "hipEventCreate failed");
        }
        // the chainback that last read this workspace must have finished before the update overwrites it
        if (p->cb_pending[k]) VIT_HIP_CHECK(hipStreamWaitEvent(s_upd, p->cb_done[k], 0));
        if (p->timing) {
            VIT_HIP_CHECK(hipEventRecord(rec.u0, s_upd));
            if (!p->epoch) p->epoch = rec.u0;
        }
        int rc = vit_hip_update_batch(p->h, sym, nf, p->L + (size_t)p->h->K - 1, p->L, p->ws[k], p->ws_bytes, nullptr, nullptr, nullptr, s_upd);
        if (rc != VIT_HIP_OK) return rc;
        if (p->timing) VIT_HIP_CHECK(hipEventRecord(rec.u1, s_upd));
        hipStream_t s_cb = s_upd;                                   // back to back unless the overlap pays (pipeline_create)
        if (frames <= p->overlap_max_frames || p->n_upd > 1) {
            // all chainbacks go through ONE stream: batches complete in submit order whichever update stream fed them
            s_cb = p->s_cb;
            VIT_HIP_CHECK(hipEventRecord(p->upd_done[k], s_upd));
            VIT_HIP_CHECK(hipStreamWaitEvent(s_cb, p->upd_done[k], 0));
        }
        if (p->timing) VIT_HIP_CHECK(hipEventRecord(rec.c0, s_cb));
        rc = chainback_batch_impl(p->h, p->ws[k], nf, p->L, out, es, s_cb, p->cb_wave_priority, p->cb_small && s_cb != s_upd);
        if (rc != VIT_HIP_OK) return rc;
        if (p->timing) {
            VIT_HIP_CHECK(hipEventRecord(rec.c1, s_cb));
            p->pending_recs.push_back(rec);
        }
        rec_guard.armed = false;
        VIT_HIP_CHECK(hipEventRecord(p->cb_done[k], s_cb));
        if (done_event && last) VIT_HIP_CHECK(hipEventRecord((hipEvent_t)done_event, s_cb));
        p->cb_pending[k] = true;
        p->last_first_frame = f0;
        p->last_frames = nf;
        p->n++;
    }
    return VIT_HIP_OK;
}

static int vit_hip_pipeline_sync_impl(vit_hip_pipeline_t p) {
    if (!p) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL pipeline");
    DeviceGuard guard(p->h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    for (int k = 0; k < p->n_upd; ++k) VIT_HIP_CHECK(hipStreamSynchronize(p->s_upd[k]));
    VIT_HIP_CHECK(hipStreamSynchronize(p->s_cb));
    // resolve the timing records of the batches that have now completed
    for (const auto& r : p->pending_recs) {
        float u = 0.f, c = 0.f, d = 0.f;
        VIT_HIP_CHECK(hipEventElapsedTime(&u, r.u0, r.u1));
        VIT_HIP_CHECK(hipEventElapsedTime(&c, r.c0, r.c1));
        VIT_HIP_CHECK(hipEventElapsedTime(&d, p->epoch, r.c1));
        p->t_update.push_back(u); p->t_chainback.push_back(c); p->t_complete.push_back(d);
    }
    for (const auto& r : p->pending_recs) {
        if (r.u0 != p->epoch) p->event_pool.push_back(r.u0);
        p->event_pool.push_back(r.u1); p->event_pool.push_back(r.c0); p->event_pool.push_back(r.c1);
    }
    p->pending_recs.clear();
    return VIT_HIP_OK;
}

static int vit_hip_pipeline_set_timing_impl(vit_hip_pipeline_t p, int enable) {
    if (!p) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL pipeline");
    const int rc = vit_hip_pipeline_sync_impl(p);               // nothing in flight while the records are reset
    if (rc != VIT_HIP_OK) return rc;
    p->t_update.clear(); p->t_chainback.clear(); p->t_complete.clear();
    if (p->epoch) { p->event_pool.push_back(p->epoch); p->epoch = nullptr; }
    p->timing = enable != 0;
    return VIT_HIP_OK;
}

int vit_hip_pipeline_get_timing(vit_hip_pipeline_t p, size_t capacity, float* update_ms, float* chainback_ms, float* complete_ms,
                                size_t* n_batches) {
    if (!p || !n_batches) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL argument");
    const size_t n = p->t_update.size();
    *n_batches = n;
    for (size_t i = 0; i < n && i < capacity; ++i) {
        if (update_ms) update_ms[i] = p->t_update[i];
        if (chainback_ms) chainback_ms[i] = p->t_chainback[i];
        if (complete_ms) complete_ms[i] = p->t_complete[i];
    }
    return VIT_HIP_OK;
}

int vit_hip_pipeline_last_workspace(vit_hip_pipeline_t p, void** d_workspace, size_t* first_frame, size_t* frames) {
    if (!p || !d_workspace) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL argument");
    if (p->n == 0) return fail(VIT_HIP_ERR_INVALID_ARG, "no batch has been submitted");
    *d_workspace = p->ws[(int)((p->n - 1) % (unsigned long long)p->n_ws)];
    if (first_frame) *first_frame = p->last_first_frame;
    if (frames) *frames = p->last_frames;
    return VIT_HIP_OK;
}

int vit_hip_pipeline_get_schedule_v2(vit_hip_pipeline_t p, vit_hip_pipeline_schedule* out, size_t schedule_bytes) {
    if (!p || !out) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL argument");
    vit_hip_pipeline_schedule full;
    vit_hip_pipeline_schedule* s = &full;
    memset(s, 0, sizeof(*s));
    s->workspaces = p->n_ws;
    s->update_streams = p->n_upd;
    s->chainback_overlapped = (p->n_upd > 1 || p->max_frames <= p->overlap_max_frames) ? 1 : 0;
    s->overlap_max_frames = p->overlap_max_frames;
    s->two_updates_max_frames = p->two_updates_max_frames;
    s->workspace_bytes_each = p->ws_bytes;
    s->sub_batch_frames = p->sub_frames;
    s->chainback_wave_priority = (int32_t)p->cb_wave_priority;
    s->chainback_small_kernel = (p->cb_small && s->chainback_overlapped) ? 1 : 0;
    memcpy(out, s, schedule_bytes < sizeof(full) ? schedule_bytes : sizeof(full));
    return VIT_HIP_OK;
}

int vit_hip_pipeline_get_schedule(vit_hip_pipeline_t p, vit_hip_pipeline_schedule* s) {
    // the struct as binaries built against the header that introduced this symbol know it: it already ended in
    // chainback_small_kernel + reserved and this entry point filled them.  Fields added later are reached through _v2 only.
    return vit_hip_pipeline_get_schedule_v2(p, s, offsetof(vit_hip_pipeline_schedule, reserved) + sizeof(int32_t));
}

int vit_hip_pipeline_wait_event(vit_hip_pipeline_t p, void* event) {
    if (!p || !event) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL argument");
    DeviceGuard guard(p->h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    // the next batch may go to any update stream (two-update schedules alternate): all of them wait
    for (int k = 0; k < p->n_upd; ++k) VIT_HIP_CHECK(hipStreamWaitEvent(p->s_upd[k], (hipEvent_t)event, 0));
    return VIT_HIP_OK;
}

static int vit_hip_get_kernel_resources_impl(vit_hip_handle h, int kernel, vit_hip_kernel_resources* out) {
    if (!h || !out) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL argument");
    memset(out, 0, sizeof(*out));
    vit::kd::KernelResources r;
    unsigned dyn = 0;
    if (h->plan == VIT_HIP_PLAN_REG) {
        if (kernel < VIT_HIP_KERNEL_UPDATE || kernel > VIT_HIP_KERNEL_RESUME) return fail(VIT_HIP_ERR_INVALID_ARG, "unknown kernel");
        if (!vit::reg_kernel_resources(h->reg_code, h->shift, kernel, &r, &dyn))
            return fail(VIT_HIP_ERR_RUNTIME, "kernel descriptor not found in the library's code objects");
    } else if (h->plan == VIT_HIP_PLAN_LDS2) {
        if (kernel != VIT_HIP_KERNEL_UPDATE && kernel != VIT_HIP_KERNEL_CHAINBACK) return fail(VIT_HIP_ERR_INVALID_ARG, "unknown kernel");
        if (!vit::lds2_kernel_resources(h->K, h->R, h->shift, kernel == VIT_HIP_KERNEL_UPDATE, &r, &dyn))
            return fail(VIT_HIP_ERR_RUNTIME, "kernel descriptor not found in the library's code objects");
    } else {
        return fail(VIT_HIP_ERR_UNSUPPORTED, "kernel resources are reported for the register plan and PLAN_LDS2");
    }
    out->vgpr_alloc = r.vgpr_alloc; out->accum_offset = r.accum_offset; out->lds_static_bytes = r.lds_static_bytes;
    out->lds_dynamic_bytes = dyn; out->scratch_bytes = r.scratch_bytes;
    return VIT_HIP_OK;
}

static int vit_hip_list_kernels_impl(size_t index, char* name, size_t name_capacity, vit_hip_kernel_resources* out) {
    const vit::kd::Table& t = vit::kd::own_library();
    if (t.empty()) return fail(VIT_HIP_ERR_RUNTIME, "the library's own file could not be read for its kernel descriptors");
    if (index >= t.size()) return fail(VIT_HIP_ERR_INVALID_ARG, "index past the last kernel");
    if (name && name_capacity > 0) {
        const size_t n = t[index].first.size() < name_capacity - 1 ? t[index].first.size() : name_capacity - 1;
        memcpy(name, t[index].first.data(), n);
        name[n] = 0;
    }
    if (out) {
        memset(out, 0, sizeof(*out));
        const vit::kd::KernelResources& r = t[index].second;
        out->vgpr_alloc = r.vgpr_alloc; out->accum_offset = r.accum_offset; out->lds_static_bytes = r.lds_static_bytes;
        out->scratch_bytes = r.scratch_bytes;
    }
    return VIT_HIP_OK;
}

int vit_hip_pipeline_destroy(vit_hip_pipeline_t p) {
    if (!p) return VIT_HIP_OK;
    DeviceGuard guard(p->h->device);
    for (int k = 0; k < vit_hip_pipeline::MAX_UPD; ++k)
        if (p->s_upd[k]) (void)hipStreamSynchronize(p->s_upd[k]);
    if (p->s_cb) (void)hipStreamSynchronize(p->s_cb);
    for (int k = 0; k < vit_hip_pipeline::MAX_WS; ++k) {
        if (p->ws[k]) (void)hipFree(p->ws[k]);
        if (p->upd_done[k]) (void)hipEventDestroy(p->upd_done[k]);
        if (p->cb_done[k]) (void)hipEventDestroy(p->cb_done[k]);
    }
    for (const auto& r : p->pending_recs) {
        if (r.u0 && r.u0 != p->epoch) (void)hipEventDestroy(r.u0);
        if (r.u1) (void)hipEventDestroy(r.u1);
        if (r.c0) (void)hipEventDestroy(r.c0);
        if (r.c1) (void)hipEventDestroy(r.c1);
    }
    if (p->epoch) (void)hipEventDestroy(p->epoch);
    for (hipEvent_t e : p->event_pool) (void)hipEventDestroy(e);
    for (int k = 0; k < vit_hip_pipeline::MAX_UPD; ++k)
        if (p->s_upd[k]) (void)hipStreamDestroy(p->s_upd[k]);
    if (p->s_cb) (void)hipStreamDestroy(p->s_cb);
    delete p;
    return VIT_HIP_OK;
}

// ---- RCCL broadcast of the shared table (the one collective of the multi-GPU path) ----
namespace {
typedef int (*nccl_broadcast_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, int, void* /*ncclComm_t*/, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
struct RcclApi {
    nccl_broadcast_fn broadcast = nullptr;
    nccl_errstr_fn errstr = nullptr;
};
// The communicator belongs to the RCCL the host program uses: take the symbol from the process first (a C/C++ host that
// links -lrccl), and load librccl.so only when the process does not export it.
const RcclApi* rccl_api() {
    // resolved exactly once, by whichever thread gets here first (function-local static: the others wait for the
    // initialiser to finish and then see the filled table -- one host thread per GPU calls this at the same moment)
    static const RcclApi api = [] {
        RcclApi a;
        void* sym = dlsym(RTLD_DEFAULT, "ncclBroadcast");
        void* lib = nullptr;
        if (!sym) {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : names) {
                lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (lib) break;
            }
            if (lib) sym = dlsym(lib, "ncclBroadcast");
        }
        a.broadcast = (nccl_broadcast_fn)sym;
        a.errstr = (nccl_errstr_fn)(lib ? dlsym(lib, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
        return a;
    }();
    return api.broadcast ? &api : nullptr;
}
}  // namespace

namespace {
constexpr size_t BROADCAST_STAGING_BYTES = 264 * 1024;      // >= vit_hip_blob_bytes(16, 8, 2, 2)
struct BroadcastStaging { void* buf[64] = {nullptr}; std::mutex alloc; std::mutex use[64]; };
BroadcastStaging* broadcast_staging_table() { static BroadcastStaging t; return &t; }
// the calling thread has `device` current (DeviceGuard); nullptr if the device index is out of the table or the one-off hipMalloc fails
void* broadcast_staging(int device) {
    if (device < 0 || device >= 64) return nullptr;
    BroadcastStaging& t = *broadcast_staging_table();
    std::lock_guard<std::mutex> lock(t.alloc);
    if (!t.buf[device] && hipMalloc(&t.buf[device], BROADCAST_STAGING_BYTES) != hipSuccess) { t.buf[device] = nullptr; (void)hipGetLastError(); }
    return t.buf[device];
}
std::mutex* broadcast_staging_mutex(int device) { return &broadcast_staging_table()->use[device]; }
}  // namespace

static int vit_hip_broadcast_table_impl(void* nccl_comm, int root, int rank, int K, int R, int soft_bytes, int error_bytes,
                            void* branch_table, void* config, int device, vit_hip_stream_t stream) {
    // Argument checks depend only on what every rank passes alike (K, R, widths, pointers being non-NULL): a bad call fails on
    // all ranks the same way and nobody is left waiting in ncclBroadcast.  Past them, a rank enters the collective exactly once
    // whatever happens to its DATA: a root that cannot pack or upload its table broadcasts a poisoned header, which the other
    // ranks report as an error.  What a rank cannot do is take part without a device: hipSetDevice failing on ONE rank (or, at
    // the process's FIRST broadcast on that device only, the one-off hipMalloc of the 264 KiB staging buffer) returns
    // VIT_HIP_ERR_NO_DEVICE from that rank BEFORE the collective, and the other ranks wait in ncclBroadcast until the caller aborts
    // the communicator (ncclCommAbort) -- the usual contract of a rank that dies in front of a collective; include/vit_hip.h says so.
    if (!nccl_comm || !branch_table || !config) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL argument");
    if (K < 2 || K > 16 || R < 1 || R > 8 || !((soft_bytes == 2 && error_bytes == 2) || (soft_bytes == 1 && error_bytes == 1)))
        return fail(VIT_HIP_ERR_UNSUPPORTED, "unsupported (K, R, soft_t, error_t)");
    const size_t need = vit_hip_blob_bytes(K, R, soft_bytes, error_bytes);
    if (need > BROADCAST_STAGING_BYTES) return fail(VIT_HIP_ERR_UNSUPPORTED, "blob larger than the staging buffer");
    const RcclApi* api = rccl_api();
    if (!api) return fail(VIT_HIP_ERR_RUNTIME, "RCCL not available: ncclBroadcast is neither in the process nor in librccl.so");
    std::vector<uint8_t> blob(need, 0);
    DeviceGuard guard(device);
    if (!guard.ok)
        return fail(VIT_HIP_ERR_NO_DEVICE, "hipSetDevice failed on this rank BEFORE the broadcast: abort the communicator, the other ranks are waiting in it");
    hipStream_t st = (hipStream_t)stream;
    // the staging buffer: allocated ONCE per device, at the first call, for the largest blob the library accepts (K = 16, R = 8,
    // 16-bit: 256 KiB + header) and kept for the life of the process -- a later broadcast cannot fail in front of the collective
    // for want of memory; the only pre-collective failure left is a rank without a usable device
    void* d_buf = broadcast_staging(device);
    if (!d_buf)
        return fail(VIT_HIP_ERR_NO_DEVICE, "no staging buffer on this rank's device BEFORE the broadcast (first call: hipMalloc of 264 KiB failed): abort the communicator, the other ranks are waiting in it");
    // one broadcast at a time per device buffer
    std::lock_guard<std::mutex> staging_lock(*broadcast_staging_mutex(device));
    std::string root_error;
    if (rank == root) {
        if (vit_hip_pack_blob(K, R, soft_bytes, error_bytes, branch_table, config, blob.data(), need) != VIT_HIP_OK) {
            root_error = g_last_error;
            memset(blob.data(), 0, sizeof(BlobHeader));          // poisoned: magic 0
        }
        if (hipMemcpyAsync(d_buf, blob.data(), need, hipMemcpyHostToDevice, st) != hipSuccess) {
            root_error = "hipMemcpyAsync (blob to device) failed";
            (void)hipGetLastError();
            (void)hipMemset(d_buf, 0, sizeof(BlobHeader));       // poison through the null stream, not the one that just failed
        }
    }
    int result = VIT_HIP_OK;
    const int nrc = api->broadcast(d_buf, d_buf, need, 1 /* ncclUint8 */, root, nccl_comm, st);
    if (nrc != 0)
        result = fail(VIT_HIP_ERR_RUNTIME, std::string("ncclBroadcast: ") + (api->errstr ? api->errstr(nrc) : "error"));
    else if (hipMemcpyAsync(blob.data(), d_buf, need, hipMemcpyDeviceToHost, st) != hipSuccess ||
             hipStreamSynchronize(st) != hipSuccess)
        result = fail(VIT_HIP_ERR_RUNTIME, "copying the broadcast blob back failed");
    if (result != VIT_HIP_OK) return result;
    if (!root_error.empty()) return fail(VIT_HIP_ERR_RUNTIME, "root rank could not pack the table: " + root_error);
    BlobHeader hd;
    memcpy(&hd, blob.data(), sizeof(hd));
    if (hd.magic != BLOB_MAGIC) return fail(VIT_HIP_ERR_RUNTIME, "the root rank failed to pack its table (poisoned header received)");
    if (hd.K != K || hd.R != R || hd.soft_bytes != soft_bytes || hd.error_bytes != error_bytes)
        return fail(VIT_HIP_ERR_INVALID_ARG, "the root rank broadcast a table for a different (K, R, soft_t, error_t)");
    if (rank != root) {
        const size_t tb = need - sizeof(hd) - 4 * (size_t)error_bytes;
        memcpy(branch_table, blob.data() + sizeof(hd), tb);
        memcpy(config, blob.data() + sizeof(hd) + tb, 4 * (size_t)error_bytes);
    }
    return VIT_HIP_OK;
}

int vit_hip_reset_batch(vit_hip_handle h, size_t frames, const uint32_t* d_start_state, void* d_metrics,
                        vit_hip_stream_t stream) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (frames == 0) return VIT_HIP_OK;
    if (!d_metrics) return fail(VIT_HIP_ERR_INVALID_ARG, "d_metrics is NULL");
    const size_t total = frames * (size_t)h->N;
    if (total > 0x7FFFFFFFull * 256ull) return fail(VIT_HIP_ERR_INVALID_ARG, "batch too large for one launch");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (h->error_bytes == 2)
        hipLaunchKernelGGL(vit::reset_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, (uint16_t*)d_metrics, d_start_state,
                           frames, (uint32_t)h->N, h->cfg_raw[1], h->cfg_raw[2]);
    else
        hipLaunchKernelGGL(vit::reset_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, (uint8_t*)d_metrics, d_start_state,
                           frames, (uint32_t)h->N, h->cfg_raw[1], h->cfg_raw[2]);
    VIT_HIP_CHECK(hipGetLastError());
    return VIT_HIP_OK;
}

int vit_hip_synth_batch(vit_hip_handle h, size_t frames, size_t L, uint64_t seed, uint64_t first_frame, float ebn0_db,
                        int noise_free, uint8_t* d_tx_bytes, void* d_symbols, vit_hip_stream_t stream) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (frames == 0) return VIT_HIP_OK;
    if (!d_symbols) return fail(VIT_HIP_ERR_INVALID_ARG, "d_symbols is NULL");
    if (L % 8 != 0) return fail(VIT_HIP_ERR_INVALID_ARG, "L must be a multiple of 8 (whole info bytes)");
    if (!h->linear) return fail(VIT_HIP_ERR_UNSUPPORTED, "the branch table is not that of a convolutional code (no polynomials)");
    if (h->soft_bytes == 2 && ((uintptr_t)d_symbols & 1u)) return fail(VIT_HIP_ERR_INVALID_ARG, "int16 symbols must be 2-byte aligned");
    const size_t S = L + (size_t)h->K - 1;
    const size_t chunks = (S + 7) / 8;
    if (frames > 0xFFFFFFFFull || S > 0x0FFFFFFFull || chunks * frames > 0x7FFFFFFFull * 256ull)
        return fail(VIT_HIP_ERR_INVALID_ARG, "batch too large for one launch");
    vit::SynthArgs a{};
    a.tx = d_tx_bytes;
    a.symbols = d_symbols;
    a.seed = seed;
    a.first_frame = first_frame;
    a.frames = (uint32_t)frames; a.L = (uint32_t)L; a.S = (uint32_t)S; a.K = (uint32_t)h->K; a.R = (uint32_t)h->R;
    for (int i = 0; i < h->R && i < 8; ++i) a.G[i] = h->G[i];
    a.high = h->high; a.low = h->low;
    a.noise_free = noise_free ? 1 : 0;
    // run_snr_ber.cpp:311-330, in float like the reference
    const float EsNo_dB = ebn0_db - 10.0f * log10f((float)h->R);
    const float noise_variance = powf(10.0f, -(EsNo_dB + 3.0f) / 10.0f);
    a.sigma = sqrtf(noise_variance);
    a.mean = ((float)h->high + (float)h->low) / 2.0f;
    a.scale = (((float)h->high - (float)h->low) / 2.0f) * (1.0f / sqrtf(1.0f + noise_variance));
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    const unsigned blocks = (unsigned)((chunks * frames + 255) / 256);
    const int rc = h->soft_bytes == 2 ? vit::launch_synth<int16_t>(h->R, a, blocks, (hipStream_t)stream)
                                      : vit::launch_synth<int8_t>(h->R, a, blocks, (hipStream_t)stream);
    if (rc != 0) return fail(VIT_HIP_ERR_RUNTIME, "synth kernel launch failed");
    return VIT_HIP_OK;
}

int vit_hip_count_bit_errors(vit_hip_handle h, const uint8_t* d_a, const uint8_t* d_b, size_t n_bytes, uint64_t* d_count,
                             vit_hip_stream_t stream) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (n_bytes == 0) return VIT_HIP_OK;
    if (!d_a || !d_b || !d_count) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL buffer");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    vit::BitErrArgs a{d_a, d_b, n_bytes, (unsigned long long*)d_count};
    size_t blocks = (n_bytes / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vit::bit_errors_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    VIT_HIP_CHECK(hipGetLastError());
    return VIT_HIP_OK;
}

static int vit_hip_shader_clock_mhz_impl(int device, double* mhz_out, double* cycles_per_pk_instr_out) {
    if (!mhz_out) return fail(VIT_HIP_ERR_INVALID_ARG, "mhz_out is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VIT_HIP_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(VIT_HIP_ERR_INVALID_ARG, "device index out of range");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    int cus = 0, wall_khz = 0;
    VIT_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || wall_khz <= 0) wall_khz = 100000;
    // with the issue rate asked for: four waves on every SIMD of the chip (one 256-thread workgroup per SIMD), about 2 ms of
    // packed adds at 2.4 GHz.  Clock only: ONE wave per CU -- a probe that can run beside other kernels without adding a chip
    // full of vector work to their power draw (the card lowers its clock under it: 1.96 GHz read by the heavy probe beside the
    // K7 pipeline against 2.3 GHz by this one)
    const bool light = cycles_per_pk_instr_out == nullptr;
    const unsigned threads = light ? 64u : 256u;
    const unsigned blocks = (unsigned)(cus > 0 ? cus : 256) * (light ? 1u : 4u), iters = 4000;
    const size_t waves = (size_t)blocks * (threads / 64u);
    uint64_t* d_out = nullptr;
    VIT_HIP_CHECK(hipMalloc((void**)&d_out, waves * 2 * sizeof(uint64_t)));
    std::vector<uint64_t> host(waves * 2);
    hipLaunchKernelGGL(vit::shader_clock_kernel, dim3(blocks), dim3(threads), 0, nullptr, d_out, iters, 3u);
    const hipError_t e1 = hipGetLastError();
    const hipError_t e2 = hipMemcpy(host.data(), d_out, waves * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(VIT_HIP_ERR_RUNTIME, "shader clock kernel failed");
    std::vector<double> ratio, cyc;
    for (size_t w = 0; w < waves; ++w)
        if (host[2 * w + 1] > 0) {
            ratio.push_back((double)host[2 * w] / (double)host[2 * w + 1]);
            cyc.push_back((double)host[2 * w] / ((double)iters * 64.0));
        }
    if (ratio.empty()) return fail(VIT_HIP_ERR_RUNTIME, "shader clock kernel returned no samples");
    std::sort(ratio.begin(), ratio.end());
    std::sort(cyc.begin(), cyc.end());
    *mhz_out = ratio[ratio.size() / 2] * (double)wall_khz / 1000.0;
    // four waves share a SIMD: the SIMD issues one of these instructions every (wave cycles per instruction) / 4
    if (cycles_per_pk_instr_out) *cycles_per_pk_instr_out = cyc[cyc.size() / 2] / 4.0;
    return VIT_HIP_OK;
}

int vit_hip_update_host(vit_hip_handle h, void* metrics_inout, const void* symbols, size_t n_steps,
                        uint64_t* decisions_out, uint64_t* renorm_sum_out) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (!metrics_inout) return fail(VIT_HIP_ERR_INVALID_ARG, "metrics_inout is NULL");
    if (renorm_sum_out) *renorm_sum_out = 0;
    if (n_steps == 0) return VIT_HIP_OK;
    if (!symbols || !decisions_out) return fail(VIT_HIP_ERR_INVALID_ARG, "symbols/decisions_out is NULL");
    if (n_steps > 0x7FFFFFF0u) return fail(VIT_HIP_ERR_INVALID_ARG, "n_steps too large");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    // device scratch and pinned staging share one layout: [metrics | renorm sum | symbols] in, [metrics | renorm sum | decision
    // rows] out -- ONE copy each way per call (five pageable copies before: this route is called once per trellis step by
    // streaming callers, examples/helpers/puncture_code_helpers.h:51)
    const size_t sym_bytes = n_steps * (size_t)h->R * (size_t)h->soft_bytes;
    const size_t dec_bytes = n_steps * (size_t)h->W * 8;
    const size_t met_bytes = (size_t)h->N * (size_t)h->error_bytes;
    const size_t met_b = align_up(met_bytes, 256), rs_b = 256;
    const size_t sym_b = align_up(sym_bytes, 256), dec_b = align_up(dec_bytes, 256);
    const size_t in_b = met_b + rs_b + sym_b;
    int rc = ensure_scratch(h, in_b + dec_b);
    if (rc != VIT_HIP_OK) return rc;
    rc = ensure_stage(h, in_b + dec_b);
    if (rc != VIT_HIP_OK) return rc;
    uint8_t* base = (uint8_t*)h->d_scratch;
    uint8_t* d_met = base;
    uint64_t* d_rs = (uint64_t*)(base + met_b);
    uint8_t* d_sym = base + met_b + rs_b;
    uint64_t* d_dec = (uint64_t*)(base + in_b);
    uint8_t* hs = (uint8_t*)h->h_stage;
    memcpy(hs, metrics_inout, met_bytes);
    memcpy(hs + met_b + rs_b, symbols, sym_bytes);
    VIT_HIP_CHECK(hipMemcpyAsync(base, hs, met_b + rs_b + sym_bytes, hipMemcpyHostToDevice, h->stream));
    // streaming state lives on the host between calls: one frame.  K <= 7 (at most 64 states): the one-wavefront latency kernel
    // (kernels_one.hpp: lane == state, metrics in a register, ~100 clocks per step); larger codes: the LDS plan on one frame
    if (vit::one_supported(h->K, h->R)) {
        vit::OneUpdateArgs oa{};
        oa.symbols = d_sym;
        oa.sym_total_bytes = sym_bytes;
        oa.decisions = d_dec;
        oa.metrics_io = d_met;
        oa.renorm_sum = d_rs;
        oa.pattern = h->d_pattern;
        oa.K = h->K;
        oa.n_steps = (int)n_steps;
        oa.cfg = h->cfg;
        if ((h->shift ? vit::one_launch_update<8>(h->R, oa, h->stream) : vit::one_launch_update<0>(h->R, oa, h->stream)) != 0)
            return fail(VIT_HIP_ERR_RUNTIME, "single-frame update launch failed");
    } else {
        rc = lds_update(h, d_sym, n_steps * (size_t)h->R, 1, n_steps, n_steps, 0, d_dec, d_met, false, d_rs, nullptr, h->stream);
        if (rc != VIT_HIP_OK) return rc;
    }
    // out: metrics and renorm sum sit in front of the symbols, the decision rows behind them: copy [metrics | rs] and the rows
    // as one contiguous range when the symbols are short (the common streaming case), else as two
    if (sym_b <= 4096) {
        VIT_HIP_CHECK(hipMemcpyAsync(hs, base, in_b + dec_bytes, hipMemcpyDeviceToHost, h->stream));
        VIT_HIP_CHECK(hipStreamSynchronize(h->stream));
        memcpy(decisions_out, hs + in_b, dec_bytes);
    } else {
        VIT_HIP_CHECK(hipMemcpyAsync(hs, base, met_b + rs_b, hipMemcpyDeviceToHost, h->stream));
        VIT_HIP_CHECK(hipMemcpyAsync(hs + met_b + rs_b, d_dec, dec_bytes, hipMemcpyDeviceToHost, h->stream));
        VIT_HIP_CHECK(hipStreamSynchronize(h->stream));
        memcpy(decisions_out, hs + met_b + rs_b, dec_bytes);
    }
    memcpy(metrics_inout, hs, met_bytes);
    uint64_t rs = 0;
    memcpy(&rs, hs + met_b, 8);
    if (renorm_sum_out) *renorm_sum_out = rs;
    return VIT_HIP_OK;
}

int vit_hip_chainback_host(vit_hip_handle h, const uint64_t* decisions, size_t L, size_t end_state, uint8_t* bytes_out) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (L == 0) return VIT_HIP_OK;
    if (!decisions || !bytes_out) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL buffer");
    if (end_state >= (size_t)h->N) return fail(VIT_HIP_ERR_INVALID_ARG, "end_state out of range");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    const size_t rows = L + (size_t)h->K - 1;
    const size_t dec_b = align_up(rows * (size_t)h->W * 8, 256);
    const size_t out_b = align_up((L + 7) / 8, 256);
    int rc = ensure_scratch(h, dec_b + out_b + 256);
    if (rc != VIT_HIP_OK) return rc;
    uint8_t* base = (uint8_t*)h->d_scratch;
    uint32_t es = (uint32_t)end_state;
    uint32_t* d_es = (uint32_t*)(base + dec_b + out_b);
    // one pinned staging copy each way (pageable copies go through the runtime's own staging and synchronise more often)
    const size_t dec_bytes = rows * (size_t)h->W * 8, out_bytes = (L + 7) / 8;
    rc = ensure_stage(h, dec_b + out_b);
    if (rc != VIT_HIP_OK) return rc;
    uint8_t* hs = (uint8_t*)h->h_stage;
    memcpy(hs, decisions, dec_bytes);
    VIT_HIP_CHECK(hipMemcpyAsync(base, hs, dec_bytes, hipMemcpyHostToDevice, h->stream));
    if (vit::one_supported(h->K, h->R)) {
        // K <= 7: rows staged through LDS by the whole wavefront, one lane chases (kernels_one.hpp); the end state is a kernel argument
        vit::OneChainbackArgs ca{};
        ca.decisions = (const uint64_t*)base;
        ca.out = base + dec_b;
        ca.end_state = es;
        ca.L = (uint32_t)L;
        ca.K = h->K;
        if (L > 0xFFFFFFF0ull) return fail(VIT_HIP_ERR_INVALID_ARG, "L too large");
        // per launch, like every other > 64 KiB launcher of the library: the attribute belongs to the CURRENT device's function object
        // (a process-wide `static` would opt in only the device of the first caller: a decoder on device 1..7 launched without it)
        VIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vit::one_chainback_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)vit::one_chainback_lds_bytes()));
        hipLaunchKernelGGL(vit::one_chainback_kernel, dim3(1), dim3(64), vit::one_chainback_lds_bytes(), h->stream, ca);
        VIT_HIP_CHECK(hipGetLastError());
    } else {
        VIT_HIP_CHECK(hipMemcpyAsync(d_es, &es, 4, hipMemcpyHostToDevice, h->stream));
        rc = lds_chainback(h, (const uint64_t*)base, 1, L, base + dec_b, d_es, h->stream);
        if (rc != VIT_HIP_OK) return rc;
    }
    VIT_HIP_CHECK(hipMemcpyAsync(hs + dec_b, base + dec_b, out_bytes, hipMemcpyDeviceToHost, h->stream));
    VIT_HIP_CHECK(hipStreamSynchronize(h->stream));
    memcpy(bytes_out, hs + dec_b, out_bytes);
    return VIT_HIP_OK;
}

// ---- frame route: ONE launch for update() + the chainback() that follows, rows kept on the device -----------------------------
int vit_hip_update_host_lazy(vit_hip_handle h, void* metrics_inout, const void* symbols, size_t n_steps, size_t first_row,
                             size_t speculate_bits, size_t speculate_end_state, uint64_t* renorm_sum_out) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (!metrics_inout) return fail(VIT_HIP_ERR_INVALID_ARG, "metrics_inout is NULL");
    if (renorm_sum_out) *renorm_sum_out = 0;
    if (n_steps == 0) return VIT_HIP_OK;
    if (!symbols) return fail(VIT_HIP_ERR_INVALID_ARG, "symbols is NULL");
    if (n_steps > 0x7FFFFFF0u || first_row > 0x7FFFFFF0u) return fail(VIT_HIP_ERR_INVALID_ARG, "n_steps / first_row too large");
    if (speculate_bits > 0 && speculate_end_state >= (size_t)h->N) return fail(VIT_HIP_ERR_INVALID_ARG, "speculate_end_state out of range");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    h->spec_valid = false;                                   // whatever was decoded ahead belonged to the rows as they were
    int rc = ensure_rows(h, first_row + n_steps);
    if (rc != VIT_HIP_OK) return rc;
    const size_t sym_bytes = n_steps * (size_t)h->R * (size_t)h->soft_bytes;
    const size_t met_bytes = (size_t)h->N * (size_t)h->error_bytes;
    if (!vit::one_supported(h->K, h->R)) {
        // K > 7: the LDS plan on one frame, as vit_hip_update_host runs it -- but the rows go straight into the device row store
        const size_t met_b = align_up(met_bytes, 256), rs_b = 256, sym_b = align_up(sym_bytes, 256);
        rc = ensure_scratch(h, met_b + rs_b + sym_b);
        if (rc != VIT_HIP_OK) return rc;
        rc = ensure_stage(h, met_b + rs_b + sym_b);
        if (rc != VIT_HIP_OK) return rc;
        uint8_t* base = (uint8_t*)h->d_scratch;
        uint8_t* hs = (uint8_t*)h->h_stage;
        memcpy(hs, metrics_inout, met_bytes);
        memcpy(hs + met_b + rs_b, symbols, sym_bytes);
        VIT_HIP_CHECK(hipMemcpyAsync(base, hs, met_b + rs_b + sym_bytes, hipMemcpyHostToDevice, h->stream));
        rc = lds_update(h, base + met_b + rs_b, n_steps * (size_t)h->R, 1, n_steps, n_steps, 0, h->d_rows + first_row * (size_t)h->W, base, false,
                        (uint64_t*)(base + met_b), nullptr, h->stream);
        if (rc != VIT_HIP_OK) return rc;
        VIT_HIP_CHECK(hipMemcpyAsync(hs, base, met_b + rs_b, hipMemcpyDeviceToHost, h->stream));
        VIT_HIP_CHECK(hipStreamSynchronize(h->stream));
        memcpy(metrics_inout, hs, met_bytes);
        if (renorm_sum_out) memcpy(renorm_sum_out, hs + met_b, 8);
        return VIT_HIP_OK;
    }
    const bool spec = speculate_bits > 0 && first_row + n_steps == speculate_bits + (size_t)h->K - 1 && speculate_bits <= 0xFFFFFFF0ull;
    const size_t met_off = 256, sym_off = met_off + align_up(met_bytes, 256), out_off = sym_off + align_up(sym_bytes, 256);
    const size_t out_bytes = spec ? (speculate_bits + 7) / 8 : 0;
    rc = ensure_map(h, out_off + align_up(out_bytes, 256));
    if (rc != VIT_HIP_OK) return rc;
    uint8_t* hm = (uint8_t*)h->h_map;
    void* dm_v = nullptr;
    VIT_HIP_CHECK(hipHostGetDevicePointer(&dm_v, h->h_map, 0));
    uint8_t* dm = (uint8_t*)dm_v;
    memcpy(hm + sym_off, symbols, sym_bytes);
    vit::OneFrameArgs fa{};
    fa.u.symbols = dm + sym_off;
    fa.u.sym_total_bytes = sym_bytes;
    fa.u.decisions = h->d_rows + first_row;                  // W == 1 here
    fa.u.metrics_io = dm + met_off;
    fa.u.renorm_sum = (uint64_t*)(dm + 64);
    fa.u.pattern = h->d_pattern;
    fa.u.K = h->K;
    fa.u.n_steps = (int)n_steps;
    fa.u.cfg = h->cfg;
    fa.u.metrics_in_args = 1;
    for (int s = 0; s < 64; ++s) {
        const int t = s & (h->N - 1);
        fa.u.metrics_in[s] = h->error_bytes == 1 ? (uint16_t)((uint32_t)((const uint8_t*)metrics_inout)[t] << 8) : ((const uint16_t*)metrics_inout)[t];
    }
    fa.do_chainback = spec ? 1 : 0;
    fa.c.decisions = h->d_rows;
    fa.c.out = dm + out_off;
    fa.c.end_state = (uint32_t)speculate_end_state;
    fa.c.L = (uint32_t)speculate_bits;
    fa.c.K = h->K;
    fa.seq = ++h->seq ? h->seq : ++h->seq;                   // never 0
    fa.done = (uint32_t*)dm;
    volatile uint32_t* done = (volatile uint32_t*)hm;
    if ((h->shift ? vit::one_launch_frame<8>(h->R, fa, h->stream) : vit::one_launch_frame<0>(h->R, fa, h->stream)) != 0)
        return fail(VIT_HIP_ERR_RUNTIME, "frame kernel launch failed");
    // the kernel's last act is a system-scope release store of seq into the control word: poll it (a stream synchronisation costs more
    // than the whole chainback); a kernel that never gets there shows up in the stream's status
    const uint32_t seq = fa.seq;
    for (uint64_t spins = 0; __atomic_load_n(done, __ATOMIC_ACQUIRE) != seq; ++spins) {
        if ((spins & 0xFFFFu) == 0xFFFFu) {
            const hipError_t q = hipStreamQuery(h->stream);
            if (q != hipErrorNotReady) {
                if (q != hipSuccess) return fail(VIT_HIP_ERR_RUNTIME, std::string("frame kernel: ") + hipGetErrorString(q));
                if (__atomic_load_n(done, __ATOMIC_ACQUIRE) != seq) return fail(VIT_HIP_ERR_RUNTIME, "frame kernel finished without reporting");
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    // let the runtime see that the launch has retired: polled completions never pass through it, and a queue it believes full of
    // pending kernels is drained the hard way once in ~200 launches (two 38 ms calls in 400)
    (void)hipStreamQuery(h->stream);
    memcpy(metrics_inout, hm + met_off, met_bytes);
    if (renorm_sum_out) memcpy(renorm_sum_out, hm + 64, 8);
    if (spec) { h->spec_valid = true; h->spec_bits = speculate_bits; h->spec_end = speculate_end_state; h->spec_off = out_off; }
    return VIT_HIP_OK;
}

int vit_hip_fetch_decisions_host(vit_hip_handle h, size_t first_row, size_t n_rows, uint64_t* decisions_out) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (n_rows == 0) return VIT_HIP_OK;
    if (!decisions_out) return fail(VIT_HIP_ERR_INVALID_ARG, "decisions_out is NULL");
    if (first_row + n_rows > h->rows_cap || !h->d_rows) return fail(VIT_HIP_ERR_INVALID_ARG, "rows outside the device row store");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    VIT_HIP_CHECK(hipMemcpyAsync(decisions_out, h->d_rows + first_row * (size_t)h->W, n_rows * (size_t)h->W * 8, hipMemcpyDeviceToHost, h->stream));
    VIT_HIP_CHECK(hipStreamSynchronize(h->stream));
    return VIT_HIP_OK;
}

int vit_hip_chainback_host_lazy(vit_hip_handle h, size_t L, size_t end_state, uint8_t* bytes_out) {
    if (!h) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL handle");
    if (L == 0) return VIT_HIP_OK;
    if (!bytes_out) return fail(VIT_HIP_ERR_INVALID_ARG, "NULL buffer");
    if (end_state >= (size_t)h->N) return fail(VIT_HIP_ERR_INVALID_ARG, "end_state out of range");
    const size_t rows = L + (size_t)h->K - 1, out_bytes = (L + 7) / 8;
    if (rows > h->rows_cap || !h->d_rows) return fail(VIT_HIP_ERR_INVALID_ARG, "the device row store does not hold L + K - 1 rows");
    if (h->spec_valid && h->spec_bits == L && h->spec_end == end_state) {      // decoded by the launch that completed the frame
        memcpy(bytes_out, (const uint8_t*)h->h_map + h->spec_off, out_bytes);
        return VIT_HIP_OK;
    }
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(VIT_HIP_ERR_RUNTIME, "hipSetDevice failed");
    const size_t out_b = align_up(out_bytes, 256);
    int rc = ensure_scratch(h, out_b + 256);
    if (rc != VIT_HIP_OK) return rc;
    rc = ensure_stage(h, out_b);
    if (rc != VIT_HIP_OK) return rc;
    uint8_t* base = (uint8_t*)h->d_scratch;
    if (vit::one_supported(h->K, h->R)) {
        if (L > 0xFFFFFFF0ull) return fail(VIT_HIP_ERR_INVALID_ARG, "L too large");
        vit::OneChainbackArgs ca{};
        ca.decisions = h->d_rows;
        ca.out = base;
        ca.end_state = (uint32_t)end_state;
        ca.L = (uint32_t)L;
        ca.K = h->K;
        VIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vit::one_chainback_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)vit::one_chainback_lds_bytes()));
        hipLaunchKernelGGL(vit::one_chainback_kernel, dim3(1), dim3(64), vit::one_chainback_lds_bytes(), h->stream, ca);
        VIT_HIP_CHECK(hipGetLastError());
    } else {
        uint32_t es = (uint32_t)end_state;
        uint32_t* d_es = (uint32_t*)(base + out_b);
        VIT_HIP_CHECK(hipMemcpyAsync(d_es, &es, 4, hipMemcpyHostToDevice, h->stream));
        rc = lds_chainback(h, h->d_rows, 1, L, base, d_es, h->stream);
        if (rc != VIT_HIP_OK) return rc;
    }
    VIT_HIP_CHECK(hipMemcpyAsync(h->h_stage, base, out_bytes, hipMemcpyDeviceToHost, h->stream));
    VIT_HIP_CHECK(hipStreamSynchronize(h->stream));
    memcpy(bytes_out, h->h_stage, out_bytes);
    return VIT_HIP_OK;
}

#ifdef VIT_HIP_CLOCK_STAMPS
// measurement build only (include/vit_hip_experiments.h): every register-plan update launched from now on stamps into d_stamps
// ([tiles][6] uint64: shader clock and constant clock at entry, at exit, XCC id, HW id); NULL switches it off
int vit_hip_experiment_clock_stamps(void* d_stamps) {
    vit::g_clock_stamps = (uint64_t*)d_stamps;
    return VIT_HIP_OK;
}
#endif

// entry points that allocate on the host: no exception crosses the ABI
#define VIT_HIP_NOTHROW(stmt)                                                                  \
    try {                                                                                      \
        stmt;                                                                                  \
    } catch (const std::exception& e) {                                                        \
        try { g_last_error = std::string("host exception: ") + e.what(); } catch (...) {}      \
        return VIT_HIP_ERR_RUNTIME;                                                            \
    } catch (...) {                                                                            \
        return VIT_HIP_ERR_RUNTIME;                                                            \
    }
int vit_hip_create(int K, int R, int soft_bytes, int error_bytes, const void* branch_table, const void* config,
                   int device, vit_hip_handle* out) {
    VIT_HIP_NOTHROW(return vit_hip_create_impl(K, R, soft_bytes, error_bytes, branch_table, config, device, out));
}

int vit_hip_create_from_blob(const void* blob, size_t blob_bytes, int device, vit_hip_handle* out) {
    VIT_HIP_NOTHROW(return vit_hip_create_from_blob_impl(blob, blob_bytes, device, out));
}

int vit_hip_set_plan(vit_hip_handle h, int plan) {
    VIT_HIP_NOTHROW(return vit_hip_set_plan_impl(h, plan));
}

static int vit_hip_precompile_impl(int K, int R, const uint32_t* polynomials, int soft_bytes, const char* directory, char* path_out,
                                   size_t path_capacity) {
    if (!polynomials) return fail(VIT_HIP_ERR_INVALID_ARG, "polynomials is NULL");
    if (soft_bytes != 1 && soft_bytes != 2) return fail(VIT_HIP_ERR_UNSUPPORTED, "soft_bytes must be 1 or 2");
    if (!vit::reg_jit_supported(K, R)) return fail(VIT_HIP_ERR_UNSUPPORTED, "the register plan serves K = 2..9 with R <= 6");
    // the same normal form vit_hip_create recovers from a branch table: bit 0 and bit K-1 of every polynomial set
    uint32_t G[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < R; ++i) G[i] = (polynomials[i] & ((1u << K) - 1u)) | 1u | (1u << (K - 1));
    // all polynomials zero: the GENERIC kernels of (K, R), which read the polynomials from their arguments (RegSpec::GENERIC)
    bool generic = true;
    for (int i = 0; i < R; ++i) generic = generic && polynomials[i] == 0;
    if (generic) {
        if (!vit::reg_generic_supported(K, R)) return fail(VIT_HIP_ERR_UNSUPPORTED, "generic register-plan kernels exist for K = 3..9 with R = 1..4 (not K = 6 at an odd R)");
        for (int i = 0; i < R; ++i) G[i] = 0;
    }
    std::string err;
    const int shift = soft_bytes == 1 ? 8 : 0;
    const std::string name = vit::reg_jit_object_name(K, R, G, shift, err);
    if (name.empty()) return fail(VIT_HIP_ERR_RUNTIME, err);
    const std::string dir = directory && *directory ? std::string(directory) : vit::package_cache_dir();
    (void)mkdir(dir.c_str(), 0755);
    const std::string path = dir + "/" + name;
    struct stat st;
    if (!(stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) && !vit::reg_jit_compile(K, R, G, shift, path, err))
        return fail(VIT_HIP_ERR_RUNTIME, err);
    if (path_out && path_capacity) snprintf(path_out, path_capacity, "%s", path.c_str());
    return VIT_HIP_OK;
}

int vit_hip_precompile(int K, int R, const uint32_t* polynomials, int soft_bytes, const char* directory, char* path_out, size_t path_capacity) {
    VIT_HIP_NOTHROW(return vit_hip_precompile_impl(K, R, polynomials, soft_bytes, directory, path_out, path_capacity));
}

int vit_hip_shader_clock_mhz(int device, double* mhz_out, double* cycles_per_pk_instr_out) {
    VIT_HIP_NOTHROW(return vit_hip_shader_clock_mhz_impl(device, mhz_out, cycles_per_pk_instr_out));
}

int vit_hip_pipeline_create(vit_hip_handle h, size_t max_frames, size_t L, vit_hip_pipeline_t* out) {
    VIT_HIP_NOTHROW(return vit_hip_pipeline_create_impl(h, max_frames, L, nullptr, out));
}

int vit_hip_pipeline_create_ex(vit_hip_handle h, size_t max_frames, size_t L, const vit_hip_pipeline_options* want,
                               vit_hip_pipeline_t* out) {
    VIT_HIP_NOTHROW(return vit_hip_pipeline_create_impl(h, max_frames, L, want, out));
}

int vit_hip_pipeline_submit(vit_hip_pipeline_t p, const void* d_symbols, size_t frames, uint8_t* d_bytes_out,
                            const uint32_t* d_end_state, void* done_event) {
    VIT_HIP_NOTHROW(return vit_hip_pipeline_submit_impl(p, d_symbols, frames, d_bytes_out, d_end_state, done_event));
}

int vit_hip_pipeline_sync(vit_hip_pipeline_t p) {
    VIT_HIP_NOTHROW(return vit_hip_pipeline_sync_impl(p));
}

int vit_hip_pipeline_set_timing(vit_hip_pipeline_t p, int enable) {
    VIT_HIP_NOTHROW(return vit_hip_pipeline_set_timing_impl(p, enable));
}

int vit_hip_get_kernel_resources(vit_hip_handle h, int kernel, vit_hip_kernel_resources* out) {
    VIT_HIP_NOTHROW(return vit_hip_get_kernel_resources_impl(h, kernel, out));
}

int vit_hip_list_kernels(size_t index, char* name, size_t name_capacity, vit_hip_kernel_resources* out) {
    VIT_HIP_NOTHROW(return vit_hip_list_kernels_impl(index, name, name_capacity, out));
}

int vit_hip_broadcast_table(void* nccl_comm, int root, int rank, int K, int R, int soft_bytes, int error_bytes,
                            void* branch_table, void* config, int device, vit_hip_stream_t stream) {
    VIT_HIP_NOTHROW(return vit_hip_broadcast_table_impl(nccl_comm, root, rank, K, R, soft_bytes, error_bytes, branch_table, config, device, stream));
}

}  // extern "C"
