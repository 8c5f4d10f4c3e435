// residency_probe.hip -- test infrastructure: proves EMPIRICALLY that a kernel's waves become resident beside N waves per SIMD of
// another kernel.  A "spinner" kernel with a chosen register allocation and LDS footprint (the update kernel's, read from its
// kernel descriptor) fills every SIMD with N waves that spin on a host-visible flag; the kernel under test (the real chainback,
// launched through the C ABI on another stream) can only run -- and complete -- while the spinners still hold the SIMDs if its
// waves fit beside them.  Every spinner leaves by itself after a few seconds (s_memrealtime), whatever the host does.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" {
struct ProbeShared {             // host memory the device can see (hipHostMalloc)
    volatile uint32_t release;   // host -> device: 1 = spinners may leave
    uint32_t pad;
    unsigned long long started;  // device -> host: spinner waves that reached their loop
};
}

// VG = registers of the unified file the wave must allocate (granules of 8): touching v[VG-1] makes next_free_vgpr = VG
#define SPINNER(VG, LAST)                                                                                                   \
    __global__ void __launch_bounds__(64) spinner_##VG(ProbeShared* sh, unsigned long long timeout_ticks) {                  \
        extern __shared__ uint32_t lds[];                                                                                    \
        asm volatile("v_mov_b32 " LAST ", 0" ::: LAST);                                                                      \
        lds[threadIdx.x] = threadIdx.x;                                                                                      \
        if (threadIdx.x == 0) atomicAdd_system(&sh->started, 1ull);                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                                      \
        while (__hip_atomic_load(&sh->release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u &&                        \
               __builtin_amdgcn_s_memrealtime() - t0 < timeout_ticks)                                                        \
            __builtin_amdgcn_s_sleep(32);                                                                                    \
        if (lds[threadIdx.x] == 0xFFFFFFFFu) sh->pad = 1;                                                                    \
    }
SPINNER(120, "v119") SPINNER(128, "v127") SPINNER(136, "v135") SPINNER(144, "v143") SPINNER(152, "v151") SPINNER(160, "v159") SPINNER(168, "v167") SPINNER(176, "v175") SPINNER(184, "v183") SPINNER(192, "v191") SPINNER(200, "v199") SPINNER(208, "v207")
SPINNER(216, "v215") SPINNER(224, "v223") SPINNER(232, "v231") SPINNER(240, "v239") SPINNER(248, "v247") SPINNER(256, "v255")

extern "C" {

int probe_alloc(ProbeShared** out) {
    void* p = nullptr;
    if (hipHostMalloc(&p, sizeof(ProbeShared), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return -1;
    *out = (ProbeShared*)p;
    (*out)->release = 0; (*out)->pad = 0; (*out)->started = 0;
    return 0;
}
int probe_free(ProbeShared* p) { return hipHostFree(p) == hipSuccess ? 0 : -1; }

// launch `blocks` one-wave spinners allocating `vgprs` registers and `lds_bytes` of LDS each on `stream`; -2: no such register class
int probe_spin(void* stream, unsigned blocks, int vgprs, unsigned lds_bytes, ProbeShared* sh, double timeout_s) {
    void (*k)(ProbeShared*, unsigned long long) = nullptr;
    switch (vgprs) {
        case 120: k = spinner_120; break; case 128: k = spinner_128; break; case 136: k = spinner_136; break; case 144: k = spinner_144; break; case 152: k = spinner_152; break;
        case 160: k = spinner_160; break; case 168: k = spinner_168; break; case 176: k = spinner_176; break;
        case 184: k = spinner_184; break; case 192: k = spinner_192; break; case 200: k = spinner_200; break; case 208: k = spinner_208; break;
        case 216: k = spinner_216; break; case 224: k = spinner_224; break; case 232: k = spinner_232; break;
        case 240: k = spinner_240; break; case 248: k = spinner_248; break; case 256: k = spinner_256; break;
        default: return -2;
    }
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess || khz <= 0) khz = 100000;
    const unsigned long long ticks = (unsigned long long)(timeout_s * 1000.0 * (double)khz);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), lds_bytes < 256 ? 256 : lds_bytes, (hipStream_t)stream, sh, ticks);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// the allocation the descriptor of the spinner for `vgprs` really has would be checked by the caller through hipcc -S; here:
unsigned long long probe_started(ProbeShared* sh) { return __atomic_load_n(&sh->started, __ATOMIC_ACQUIRE); }
void probe_release(ProbeShared* sh) { __atomic_store_n((uint32_t*)&sh->release, 1u, __ATOMIC_RELEASE); }
void probe_reset(ProbeShared* sh) { sh->release = 0; __atomic_store_n(&sh->started, 0ull, __ATOMIC_RELEASE); }

}  // extern "C"
