// reg_jit.hpp -- run-time instantiation of PLAN_REG for polynomials outside the ahead-of-time table.
//
// PLAN_REG's branch-pattern bookkeeping is compile-time (kernels_reg.hpp: RegSpec<K,R,G...>), so a new code needs a new
// instantiation.  On first use we write a four-kernel translation unit that includes kernels_reg.hpp with the caller's
// polynomials, compile it with `hipcc --genco` for gfx950 (about as long as one of the ahead-of-time units: 10-40 s),
// keep the code object in a disk cache and load it with hipModuleLoad.  Later handles -- and later processes -- reuse it.
//   cache directory : $VIT_HIP_CACHE_DIR, else $HOME/.cache/vit_hip, else /tmp/vit_hip_cache
//   compiler        : $VIT_HIP_HIPCC, else /opt/rocm/bin/hipcc
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <unistd.h>

#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>

#include "kernels_reg.hpp"

namespace vit {

inline bool reg_jit_supported(int K, int R) { return (K == 3 || K == 4 || K == 5 || K == 7 || K == 9) && R >= 1 && R <= 4; }

namespace jit_detail {

inline std::string this_library_dir() {
    Dl_info info;
    if (dladdr((const void*)&this_library_dir, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t slash = p.find_last_of('/');
        return slash == std::string::npos ? std::string(".") : p.substr(0, slash);
    }
    return ".";
}

inline std::string cache_dir() {
    const char* e = getenv("VIT_HIP_CACHE_DIR");
    std::string d;
    if (e && *e) d = e;
    else if ((e = getenv("HOME")) && *e) d = std::string(e) + "/.cache/vit_hip";
    else d = "/tmp/vit_hip_cache";
    // mkdir -p (two levels are enough for the defaults)
    const size_t slash = d.find_last_of('/');
    if (slash != std::string::npos && slash > 0) (void)mkdir(d.substr(0, slash).c_str(), 0755);
    (void)mkdir(d.c_str(), 0755);
    if (access(d.c_str(), W_OK) != 0) {
        d = "/tmp/vit_hip_cache";
        (void)mkdir(d.c_str(), 0755);
    }
    return d;
}

inline uint64_t fnv1a_file(const std::string& path, uint64_t h) {
    std::ifstream f(path, std::ios::binary);
    char buf[4096];
    while (f) {
        f.read(buf, sizeof(buf));
        for (std::streamsize i = 0; i < f.gcount(); ++i) { h ^= (uint8_t)buf[i]; h *= 1099511628211ull; }
    }
    return h;
}

inline std::mutex& mutex() { static std::mutex m; return m; }
inline std::map<std::string, RegJitModule*>& modules() { static std::map<std::string, RegJitModule*> m; return m; }

}  // namespace jit_detail

// returns nullptr and fills `err` on failure.  The module belongs to the current device.
inline const RegJitModule* reg_jit_get(int K, int R, const uint32_t* G, int device, std::string& err) {
    using namespace jit_detail;
    if (!reg_jit_supported(K, R)) { err = "PLAN_REG run-time instantiation serves K in {3,4,5,7,9}, R <= 4"; return nullptr; }
    const int lane_bits = K >= 7 ? 2 : 0;
    const std::string src_dir = this_library_dir() + "/csrc";
    uint64_t h = 1469598103934665603ull;
    h = fnv1a_file(src_dir + "/kernels_reg.hpp", h);
    h = fnv1a_file(src_dir + "/common.hpp", h);
    if (h == 1469598103934665603ull) { err = "kernel sources not found next to the library (" + src_dir + ")"; return nullptr; }
    std::ostringstream key;
    key << "reg_K" << K << "R" << R;
    for (int i = 0; i < 4; ++i) key << "_" << (i < R ? G[i] : 0u);
    key << "_" << std::hex << h;
    std::lock_guard<std::mutex> lock(mutex());
    const std::string mkey = key.str() + "@" + std::to_string(device);
    auto it = modules().find(mkey);
    if (it != modules().end()) return it->second;

    const std::string dir = cache_dir();
    const std::string base = dir + "/" + key.str();
    const std::string hsaco = base + ".hsaco";
    if (access(hsaco.c_str(), R_OK) != 0) {
        const std::string src = base + "." + std::to_string((long)getpid()) + ".hip";
        {
            std::ofstream f(src);
            f << "#define VIT_REG_JIT_TU 1\n#include \"" << src_dir << "/kernels_reg.hpp\"\n"
              << "using SP = vit::RegSpec<" << K << ", " << R;
            for (int i = 0; i < 4; ++i) f << ", " << (i < R ? G[i] : 0u) << "u";
            f << ", " << lane_bits << ">;\n"
              << "extern \"C\" __global__ void __launch_bounds__(64, vit::reg_update_min_waves<SP>()) vit_jit_update_16(vit::RegUpdateArgs a) { vit::reg_update_body<SP, 0>(a); }\n"
              << "extern \"C\" __global__ void __launch_bounds__(64, vit::reg_update_min_waves<SP>()) vit_jit_update_8(vit::RegUpdateArgs a) { vit::reg_update_body<SP, 8>(a); }\n"
              << "extern \"C\" __global__ void __launch_bounds__(64, 2) vit_jit_chainback(vit::RegChainbackArgs a) { vit::reg_chainback_body<SP>(a); }\n"
              << "extern \"C\" __global__ void vit_jit_export(vit::RegExportArgs a) { vit::reg_export_body<SP>(a); }\n";
        }
        const char* cc = getenv("VIT_HIP_HIPCC");
        const std::string tmp = hsaco + "." + std::to_string((long)getpid()) + ".tmp";
        const std::string cmd = std::string(cc && *cc ? cc : "/opt/rocm/bin/hipcc") +
                                " -O3 -std=c++17 --offload-arch=gfx950 --genco -o '" + tmp + "' '" + src + "' > '" + base + ".log' 2>&1";
        const int rc = system(cmd.c_str());
        if (rc != 0 || access(tmp.c_str(), R_OK) != 0) {
            err = "hipcc failed for the run-time PLAN_REG instantiation (see " + base + ".log)";
            return nullptr;
        }
        (void)rename(tmp.c_str(), hsaco.c_str());
        (void)unlink(src.c_str());
    }
    RegJitModule* m = new RegJitModule();
    if (hipModuleLoad(&m->module, hsaco.c_str()) != hipSuccess ||
        hipModuleGetFunction(&m->update[0], m->module, "vit_jit_update_16") != hipSuccess ||
        hipModuleGetFunction(&m->update[1], m->module, "vit_jit_update_8") != hipSuccess ||
        hipModuleGetFunction(&m->chainback, m->module, "vit_jit_chainback") != hipSuccess ||
        hipModuleGetFunction(&m->export_, m->module, "vit_jit_export") != hipSuccess) {
        err = "could not load " + hsaco;
        delete m;
        return nullptr;
    }
    m->chainback_frames_per_block = K == 7 ? 128u : lane_bits == 0 ? 64u : 32u;
    modules()[mkey] = m;
    return m;
}

}  // namespace vit
