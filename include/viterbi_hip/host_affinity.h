// viterbi_hip/host_affinity.h -- pin the calling host thread to the CPUs local to ITS GPU (the NUMA node the card hangs off).
//
// The reference runs one decoder per worker thread over a shared branch table (examples/run_benchmark.cpp:193-197); here a worker
// drives one GPU, and eight workers of one node should not all queue their launches from socket 0.  GPU `index` is the index-th
// GPU node of the KFD topology (HIP's device order; an integer HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES list is honoured), its
// PCI function's local_cpulist comes from the DRM render node.  Best effort, Linux only, no HIP call: call it BEFORE the thread's
// first GPU call.  The same logic as bench.py's pin_to_gpu_numa_node().
#pragma once
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace viterbi_hip {

struct AffinityResult {
    int gpu = -1, numa_node = -1, cpus_local = 0, cpus_used = 0;
    bool pinned = false;
    std::string error;
};

inline AffinityResult pin_thread_to_gpu_cpus(int index) {
    AffinityResult res;
    res.gpu = index;
    std::vector<int> render_minor;                                   // of every GPU node this process may open, in KFD order
    for (int n = 0; n < 256; ++n) {
        std::ifstream f("/sys/class/kfd/kfd/topology/nodes/" + std::to_string(n) + "/properties");
        if (!f) { if (n > 64) break; else continue; }
        std::string key; long long val; long long simd = 0, minor = -1;
        while (f >> key >> val) { if (key == "simd_count") simd = val; else if (key == "drm_render_minor") minor = val; }
        if (simd > 0) render_minor.push_back((int)minor);
    }
    const char* vis = getenv("HIP_VISIBLE_DEVICES");
    if (!vis || !*vis) vis = getenv("ROCR_VISIBLE_DEVICES");
    if (vis && *vis) {
        std::vector<int> pick; bool ints = true; std::stringstream ss(vis); std::string tok;
        while (std::getline(ss, tok, ',')) {
            char* end = nullptr; const long v = strtol(tok.c_str(), &end, 10);
            if (end == tok.c_str() || *end) { ints = false; break; }
            if (v >= 0 && (size_t)v < render_minor.size()) pick.push_back(render_minor[(size_t)v]);
        }
        if (ints) render_minor = pick;
    }
    if (index < 0 || (size_t)index >= render_minor.size() || render_minor[(size_t)index] < 0) { res.error = "GPU not found in the KFD topology"; return res; }
    const std::string dev = "/sys/class/drm/renderD" + std::to_string(render_minor[(size_t)index]) + "/device/";
    { std::ifstream f(dev + "numa_node"); if (f) f >> res.numa_node; }
    std::ifstream f(dev + "local_cpulist");
    std::string list;
    if (!f || !std::getline(f, list)) { res.error = "no local_cpulist for the GPU's PCI function"; return res; }
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (pthread_getaffinity_np(pthread_self(), sizeof(allowed), &allowed) != 0) { res.error = "pthread_getaffinity_np failed"; return res; }
    std::stringstream ss(list); std::string part;
    while (std::getline(ss, part, ',')) {
        if (part.empty()) continue;
        int a = 0, b = 0;
        if (sscanf(part.c_str(), "%d-%d", &a, &b) == 1) b = a;
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) { ++res.cpus_local; if (CPU_ISSET(c, &allowed)) CPU_SET(c, &want); }
    }
    if (CPU_COUNT(&want) > 0 && pthread_setaffinity_np(pthread_self(), sizeof(want), &want) == 0) res.pinned = true;
    if (pthread_getaffinity_np(pthread_self(), sizeof(allowed), &allowed) == 0) res.cpus_used = CPU_COUNT(&allowed);
    return res;
}

}  // namespace viterbi_hip
