// reg_jit.hpp -- run-time instantiation of PLAN_REG for polynomials outside the ahead-of-time table.
//
// PLAN_REG's branch-pattern bookkeeping is compile-time (kernels_reg.hpp: RegSpec<K,R,G...>), so a new code needs a new
// instantiation.  On first use we write a four-kernel translation unit that includes kernels_reg.hpp with the caller's
// polynomials, compile it with `hipcc --genco` for gfx950 (about as long as one of the ahead-of-time units: 10-40 s),
// keep the code object in a disk cache and load it with hipModuleLoad.  Later handles -- and later processes -- reuse it.
//   cache directory : $VIT_HIP_CACHE_DIR, else $XDG_CACHE_HOME/vit_hip, else $HOME/.cache/vit_hip, else /tmp/vit_hip_cache-<uid>;
//                     created 0700 and used only if it belongs to this user and nobody else can write to it -- a code
//                     object found there is loaded into the caller's GPU context, so it must not be plantable by others
//   compiler        : $VIT_HIP_HIPCC, else /opt/rocm/bin/hipcc; started with posix_spawn and an argv array (a fresh child
//                     process, no shell).  Needed only to COMPILE: a code object that is already in the package cache
//                     (precompiled at install time, see below) or in the user cache loads on a host without any compiler
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <spawn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "kernels_reg.hpp"

extern char** environ;

namespace vit {

// K = 6 keeps all 32 states of a frame pair in one lane; its unrolled block is lcm(5, 16 / gcd(16, R * sizeof(soft_t))) steps,
// which only stays reasonable (<= 40) for R = 2 and R = 4
inline bool reg_jit_supported(int K, int R) {
    // R = 5, 6: split pattern tables (RegSpec::SPLIT), a feature of the LDS branch-metric ring (four lanes per frame pair: K >= 7).
    // K = 8 with eight patterns and more fetches its branch metrics per sub-chunk like K = 9 (one group of four steps in the ring)
    if (K >= 7 && K <= 9) return R >= 1 && R <= 6;
    // K < 7: the whole state of a frame pair sits in one lane and every branch metric is summed there (no table to split); K = 6 at an
    // odd rate unrolls a 240-step block, K = 6 at R = 6 runs out of registers (644 bytes of scratch) and is still far ahead of PLAN_LDS
    return K >= 2 && K <= 6 && R >= 1 && R <= 6;
}

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

// owned by this user, not writable by group or others, and of the expected kind
inline bool private_to_user(const std::string& path, bool want_dir) {
    struct stat st;
    if (lstat(path.c_str(), &st) != 0) return false;
    if (want_dir ? !S_ISDIR(st.st_mode) : !S_ISREG(st.st_mode)) return false;
    return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}

inline bool make_private_dir(const std::string& d) {
    const size_t slash = d.find_last_of('/');
    if (slash != std::string::npos && slash > 0) (void)mkdir(d.substr(0, slash).c_str(), 0700);   // e.g. ~/.cache
    (void)mkdir(d.c_str(), 0700);
    return private_to_user(d, true) && access(d.c_str(), W_OK) == 0;
}

// empty string: no usable private directory (the caller then refuses to JIT rather than trust a shared one)
inline std::string cache_dir() {
    const char* e = getenv("VIT_HIP_CACHE_DIR");
    if (e && *e) return make_private_dir(e) ? std::string(e) : std::string();
    std::string d;
    if ((e = getenv("XDG_CACHE_HOME")) && *e) { d = std::string(e) + "/vit_hip"; if (make_private_dir(d)) return d; }
    if ((e = getenv("HOME")) && *e) { d = std::string(e) + "/.cache/vit_hip"; if (make_private_dir(d)) return d; }
    d = "/tmp/vit_hip_cache-" + std::to_string((unsigned long)geteuid());
    return make_private_dir(d) ? d : std::string();
}

// run argv[0] with argv (no shell), stdout+stderr to `log_path` (or captured into *out when non-null); returns exit status
inline int run_child(const std::vector<std::string>& argv, const std::string& log_path, std::string* out) {
    std::vector<char*> av;
    for (const std::string& a : argv) av.push_back(const_cast<char*>(a.c_str()));
    av.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    int pipefd[2] = {-1, -1};
    if (out) {
        if (pipe(pipefd) != 0) { posix_spawn_file_actions_destroy(&fa); return -1; }
        posix_spawn_file_actions_adddup2(&fa, pipefd[1], 1);
        posix_spawn_file_actions_adddup2(&fa, pipefd[1], 2);
        posix_spawn_file_actions_addclose(&fa, pipefd[0]);
    } else {
        posix_spawn_file_actions_addopen(&fa, 1, log_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
        posix_spawn_file_actions_adddup2(&fa, 1, 2);
    }
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, av[0], &fa, nullptr, av.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (out) close(pipefd[1]);
    if (rc != 0) { if (out) close(pipefd[0]); return -1; }
    if (out) {
        char buf[512];
        ssize_t n;
        while ((n = read(pipefd[0], buf, sizeof(buf))) > 0) out->append(buf, (size_t)n);
        close(pipefd[0]);
    }
    int status = 0;
    if (waitpid(pid, &status, 0) < 0) return -1;
    return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

inline uint64_t fnv1a_str(const std::string& s, uint64_t h) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
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

// The code objects have two homes:
//   * the PACKAGE cache `<directory of libvit_hip.so>/precompiled/` -- written at install / build time by
//     `python -m viterbidecodercpp_amd.tools.precompile` (vit_hip_precompile), read-only at run time and as trustworthy as the
//     library beside it.  vit_hip_create consults it for every code outside the ahead-of-time table: a code found there runs
//     PLAN_REG from the first call on a host that has no compiler at all;
//   * the USER cache (cache_dir()): filled by run-time compilation (vit_hip_set_plan(PLAN_REG) / VIT_HIP_JIT=1), consulted only then.
// A file's name carries the code, the symbol width, the target and a hash of the kernel sources it was compiled from (NOT the
// compiler's version: any hipcc's gfx950 code object of these sources is the same kernel), so a stale object is never found.
inline std::string package_cache_dir() { return jit_detail::this_library_dir() + "/precompiled"; }

// file name (without directory) of the code object of one (code, symbol width); empty + err when the kernel sources are not
// beside the library
inline std::string reg_jit_object_name(int K, int R, const uint32_t* G, int shift, std::string& err) {
    using namespace jit_detail;
    const std::string src_dir = this_library_dir() + "/csrc";
    uint64_t h = 1469598103934665603ull;
    h = fnv1a_file(src_dir + "/kernels_reg.hpp", h);
    h = fnv1a_file(src_dir + "/common.hpp", h);
    if (h == 1469598103934665603ull) { err = "kernel sources not found next to the library (" + src_dir + ")"; return std::string(); }
    h = fnv1a_file(src_dir + "/kernel_desc.hpp", h);
    std::ostringstream key;
    key << "reg_K" << K << "R" << R;
    for (int i = 0; i < 6; ++i) key << "_" << (i < R ? G[i] : 0u);
    key << (shift ? "_s8" : "_s16") << "_gfx950_" << std::hex << h << ".hsaco";
    return key.str();
}

// Compile the four (five) kernels of one code and symbol width into `hsaco` (atomic rename; the log stays beside it on failure).
// No GPU call: this also runs on a build host without a card.
inline bool reg_jit_compile(int K, int R, const uint32_t* G, int shift, const std::string& hsaco, std::string& err) {
    using namespace jit_detail;
    if (!reg_jit_supported(K, R)) { err = "PLAN_REG run-time instantiation serves K = 2..9 with R <= 6"; return false; }
    const int lane_bits = K >= 7 ? 2 : 0;
    const std::string src_dir = this_library_dir() + "/csrc";
    const char* cc_env = getenv("VIT_HIP_HIPCC");
    const std::string hipcc = cc_env && *cc_env ? cc_env : "/opt/rocm/bin/hipcc";
    if (access(hipcc.c_str(), X_OK) != 0) { err = "cannot run " + hipcc + " (hipcc is needed to compile a register-plan instantiation; set VIT_HIP_HIPCC, or precompile on a build host: python -m viterbidecodercpp_amd.tools.precompile)"; return false; }
    const std::string base = hsaco.substr(0, hsaco.size() > 6 ? hsaco.size() - 6 : hsaco.size());   // strip ".hsaco"
    const std::string src = base + "." + std::to_string((long)getpid()) + ".hip";
    {
        std::ofstream f(src);
        f << "#define VIT_REG_JIT_TU 1\n";
        if (K == 9 || (K == 7 && R == 3)) f << "#define VIT_REG_UPDATE_VGPR_CAP __attribute__((amdgpu_num_vgpr(120)))\n";   // kernels_reg.hpp, reg_update_kernel
        f << "#include \"" << src_dir << "/kernels_reg.hpp\"\n"
          << "using SP = vit::RegSpec<" << K << ", " << R;
        for (int i = 0; i < 4; ++i) f << ", " << (i < R ? G[i] : 0u) << "u";
        f << ", " << lane_bits << ", " << (4 < R ? G[4] : 0u) << "u, " << (5 < R ? G[5] : 0u) << "u>;\n"
          << "extern \"C\" __global__ void __launch_bounds__(64, vit::reg_update_min_waves<SP>()) VIT_REG_UPDATE_VGPR_CAP vit_jit_update_" << (shift ? 8 : 16)
          << "(vit::RegUpdateArgs a) { vit::reg_update_body<SP, " << (shift ? 8 : 0) << ", false>(a); }\n"
          << "extern \"C\" __global__ void __launch_bounds__(64, 1) vit_jit_resume_" << (shift ? 8 : 16) << "(vit::RegUpdateArgs a) { vit::reg_update_body<SP, "
          << (shift ? 8 : 0) << ", true>(a); }\n"
          << "extern \"C\" __global__ void __launch_bounds__(64, vit::reg_chainback_min_waves<SP>()) vit_jit_chainback(vit::RegChainbackArgs a) { if (a.wave_priority) __builtin_amdgcn_s_setprio(3); vit::reg_chainback_body<SP>(a); }\n"
          << "extern \"C\" __global__ void vit_jit_export(vit::RegExportArgs a) { vit::reg_export_body<SP>(a); }\n";
        if (K == 9 || K == 7) f << "extern \"C\" __global__ void __launch_bounds__(64, vit::reg_chainback_alt_min_waves<SP>()) vit_jit_chainback_alt(vit::RegChainbackArgs a) { vit::reg_chainback_alt_body<SP>(a); }\n";
        if (!f) { err = "cannot write " + src; return false; }
    }
    const std::string tmp = hsaco + "." + std::to_string((long)getpid()) + ".tmp";
    const int rc = run_child({hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--genco", "-o", tmp, src}, base + ".log", nullptr);
    if (rc != 0 || access(tmp.c_str(), R_OK) != 0) {
        (void)unlink(tmp.c_str());
        err = "hipcc failed for the run-time PLAN_REG instantiation (see " + base + ".log)";
        return false;
    }
    (void)chmod(tmp.c_str(), 0644);
    if (rename(tmp.c_str(), hsaco.c_str()) != 0) { (void)unlink(tmp.c_str()); err = "cannot move the code object to " + hsaco; return false; }
    (void)unlink(src.c_str());
    (void)unlink((base + ".log").c_str());
    return true;
}

// returns nullptr and fills `err` on failure.  The module belongs to the current device.
// one code object per (code, symbol width): a decoder handle has ONE width, and the update / resume kernels of the other one are
// half of the compile time (1 - 4 minutes for the K = 8, 9 codes)
// `package_only`: look in the package cache and nowhere else, compile nothing (what vit_hip_create does for PLAN_AUTO).
// `origin` (optional) receives the path of the code object the module was loaded from.
inline const RegJitModule* reg_jit_get(int K, int R, const uint32_t* G, int shift, int device, bool package_only, std::string& err,
                                       std::string* origin = nullptr) {
    using namespace jit_detail;
    if (!reg_jit_supported(K, R)) { err = "PLAN_REG run-time instantiation serves K = 2..9 with R <= 6"; return nullptr; }
    const int lane_bits = K >= 7 ? 2 : 0;
    const std::string name = reg_jit_object_name(K, R, G, shift, err);
    if (name.empty()) return nullptr;
    std::lock_guard<std::mutex> lock(mutex());
    static std::map<std::string, std::string> origins;
    const std::string mkey = name + "@" + std::to_string(device);
    std::string hsaco = package_cache_dir() + "/" + name;
    auto it = modules().find(mkey);
    // a module this process loaded from the USER cache does not make the code "installed": package_only keeps ignoring it
    if (it != modules().end() && (!package_only || origins[mkey] == hsaco)) {
        if (origin) *origin = origins[mkey];
        return it->second;
    }
    if (it != modules().end()) { err = "no precompiled code object for this code in " + package_cache_dir(); return nullptr; }
    struct stat st;
    const bool in_package = stat(hsaco.c_str(), &st) == 0 && S_ISREG(st.st_mode);
    if (!in_package) {
        if (package_only) { err = "no precompiled code object for this code in " + package_cache_dir(); return nullptr; }
        const std::string dir = cache_dir();
        if (dir.empty()) { err = "no private cache directory for run-time compiled kernels (set VIT_HIP_CACHE_DIR to a directory only you can write)"; return nullptr; }
        hsaco = dir + "/" + name;
        if (!private_to_user(hsaco, false)) {
            (void)unlink(hsaco.c_str());                 // not ours or writable by others: never load it
            if (!reg_jit_compile(K, R, G, shift, hsaco, err)) return nullptr;
            (void)chmod(hsaco.c_str(), 0600);
        }
        if (!private_to_user(hsaco, false)) { err = "could not load " + hsaco; return nullptr; }
    }
    RegJitModule* m = new RegJitModule();
    if (hipModuleLoad(&m->module, hsaco.c_str()) != hipSuccess ||
        hipModuleGetFunction(&m->update[shift ? 1 : 0], m->module, shift ? "vit_jit_update_8" : "vit_jit_update_16") != hipSuccess ||
        hipModuleGetFunction(&m->resume[shift ? 1 : 0], m->module, shift ? "vit_jit_resume_8" : "vit_jit_resume_16") != hipSuccess ||
        hipModuleGetFunction(&m->chainback, m->module, "vit_jit_chainback") != hipSuccess ||
        hipModuleGetFunction(&m->export_, m->module, "vit_jit_export") != hipSuccess) {
        (void)hipGetLastError();
        err = "could not load " + hsaco;
        delete m;
        return nullptr;
    }
    m->chainback_frames_per_block = (K == 7 || K == 9) ? 128u : lane_bits == 0 ? 64u : 32u;
    (void)kd::parse_file(hsaco, m->kernels);           // an unreadable table only makes the pipeline pick its conservative schedule
    if ((K == 9 || K == 7) && hipModuleGetFunction(&m->chainback_alt, m->module, "vit_jit_chainback_alt") != hipSuccess) m->chainback_alt = nullptr;
    modules()[mkey] = m;
    origins[mkey] = hsaco;
    if (origin) *origin = hsaco;
    return m;
}

}  // namespace vit
