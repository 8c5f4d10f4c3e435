// kernel_desc.hpp -- what the wave launcher (SPI) sees of a kernel: its KERNEL DESCRIPTOR, read from the code object.
//
// Which kernels can share a SIMD is decided by the register ALLOCATION in the descriptor (compute_pgm_rsrc1
// .granulated_workitem_vgpr_count), not by the count of registers the code uses: hipcc pads the allocation of a kernel whose
// static LDS limits its occupancy up to the count that enforces that occupancy, and hipFuncGetAttributes().numRegs /
// the code-object metadata `.vgpr_count` / the -Rpass-analysis remark all report the USED count (round 3 shipped a schedule
// built on 22 "used" registers of a kernel that allocated 264).  So the pipeline's residency rules read the descriptors:
//   * of the ahead-of-time kernels from this library's own file (section .hip_fatbin: one clang offload bundle per
//     translation unit, each holding a gfx950 code object), and
//   * of a run-time compiled code from its .hsaco file (reg_jit.hpp).
// Pure host code, no GPU and no ROCm call needed (tests/test_host.py checks the table against hipcc's -S output).
#pragma once
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace vit {
namespace kd {

struct KernelResources {
    uint32_t vgpr_alloc = 0;        // entries of the SIMD's 512-entry unified register file one wave occupies (arch + accumulation)
    uint32_t accum_offset = 0;      // first accumulation register (the arch VGPR share, in registers)
    uint32_t lds_static_bytes = 0;  // group_segment_fixed_size (dynamic LDS comes on top at launch)
    uint32_t scratch_bytes = 0;     // private_segment_fixed_size per lane
};
typedef std::vector<std::pair<std::string, KernelResources>> Table;

namespace detail {
inline uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

struct Section { uint32_t name, type; uint64_t addr, off, size; uint32_t link; uint64_t entsize; };

// the section headers of a little-endian ELF64 image; false if `p` is not one
inline bool elf_sections(const uint8_t* p, size_t n, std::vector<Section>& out, uint16_t* shstrndx) {
    if (n < 64 || memcmp(p, "\x7f" "ELF", 4) != 0 || p[4] != 2 /* ELFCLASS64 */ || p[5] != 1 /* little endian */) return false;
    const uint64_t shoff = rd64(p + 0x28);
    const uint16_t shentsize = rd16(p + 0x3A), shnum = rd16(p + 0x3C);
    *shstrndx = rd16(p + 0x3E);
    if (shentsize < 64 || shoff > n || (uint64_t)shnum * shentsize > n - shoff) return false;
    for (uint16_t i = 0; i < shnum; ++i) {
        const uint8_t* s = p + shoff + (uint64_t)i * shentsize;
        Section x;
        x.name = rd32(s); x.type = rd32(s + 4); x.addr = rd64(s + 16); x.off = rd64(s + 24); x.size = rd64(s + 32);
        x.link = rd32(s + 40); x.entsize = rd64(s + 56);
        if (x.type != 8 /* SHT_NOBITS */ && (x.off > n || x.size > n - x.off)) return false;
        out.push_back(x);
    }
    return *shstrndx < out.size();
}
}  // namespace detail

// every `<kernel>.kd` symbol of one AMDGPU code object (an ELF64 image): appended to `out` under the kernel's name
inline bool parse_code_object(const uint8_t* p, size_t n, Table& out) {
    using namespace detail;
    std::vector<Section> sec;
    uint16_t shstrndx = 0;
    if (!elf_sections(p, n, sec, &shstrndx)) return false;
    if (rd16(p + 0x12) != 224 /* EM_AMDGPU */) return false;
    for (const Section& st : sec) {
        if (st.type != 2 /* SHT_SYMTAB */) continue;           // .symtab holds every kernel's descriptor symbol (.dynsym only the exported ones)
        if (st.link >= sec.size() || st.entsize < 24) continue;
        const Section& str = sec[st.link];
        if (str.type == 8 /* SHT_NOBITS: no bytes in the file, its offset / size were never range-checked */) continue;
        for (uint64_t o = 0; o + st.entsize <= st.size; o += st.entsize) {
            const uint8_t* s = p + st.off + o;
            const uint32_t name_off = rd32(s);
            const uint16_t shndx = rd16(s + 6);
            const uint64_t value = rd64(s + 8), size = rd64(s + 16);
            if (name_off >= str.size || shndx == 0 || shndx >= sec.size() || size != 64) continue;
            const char* name = (const char*)p + str.off + name_off;
            const size_t len = strnlen(name, (size_t)(str.size - name_off));
            if (len < 4 || memcmp(name + len - 3, ".kd", 3) != 0) continue;
            const Section& home = sec[shndx];
            if (home.type == 8 /* SHT_NOBITS */) continue;
            if (value < home.addr || value - home.addr + 64 > home.size) continue;
            const uint8_t* d = p + home.off + (value - home.addr);   // amd_kernel_descriptor_t, 64 bytes
            KernelResources r;
            r.lds_static_bytes = rd32(d + 0);
            r.scratch_bytes = rd32(d + 4);
            const uint32_t rsrc3 = rd32(d + 44), rsrc1 = rd32(d + 48);
            r.vgpr_alloc = ((rsrc1 & 63u) + 1u) * 8u;          // gfx90a and later: granules of 8 of the unified file
            r.accum_offset = ((rsrc3 & 63u) + 1u) * 4u;
            out.emplace_back(std::string(name, len - 3), r);
        }
    }
    return true;
}

// every AMDGPU code object inside the clang offload bundles that start on 4 KiB boundaries of [p, p + n): magic, u64 entry count,
// then per entry {u64 offset, u64 size, u64 triple length, triple}
inline bool parse_bundles(const uint8_t* p, size_t n, Table& out) {
    using namespace detail;
    static const char MAGIC[] = "__CLANG_OFFLOAD_BUNDLE__";
    bool any = false;
    for (uint64_t b = 0; b + 32 <= n; b += 4096) {
        const uint8_t* h = p + b;
        if (memcmp(h, MAGIC, 24) != 0) continue;
        const uint64_t count = rd64(h + 24);
        uint64_t q = 32;
        for (uint64_t i = 0; i < count && b + q + 24 <= n; ++i) {
            const uint64_t off = rd64(h + q), size = rd64(h + q + 8), tl = rd64(h + q + 16);
            q += 24;
            if (tl > n - b - q) break;
            const std::string triple((const char*)h + q, (size_t)tl);
            q += tl;
            if (size == 0 || triple.find("amdgcn") == std::string::npos) continue;
            if (off > n - b || size > n - b - off) continue;
            any = parse_code_object(h + off, (size_t)size, out) || any;
        }
    }
    return any;
}

// a file that is a code object itself, an offload bundle of code objects (what `hipcc --genco` writes: .hsaco), or a host ELF
// whose .hip_fatbin section holds one bundle per translation unit
inline bool parse_file(const std::string& path, Table& out) {
    using namespace detail;
    std::vector<uint8_t> buf;
    {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) return false;
        if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return false; }
        const long sz = ftell(f);
        if (sz <= 0 || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return false; }
        buf.resize((size_t)sz);
        const size_t got = fread(buf.data(), 1, buf.size(), f);
        fclose(f);
        if (got != buf.size()) return false;
    }
    const uint8_t* p = buf.data();
    const size_t n = buf.size();
    if (n >= 24 && memcmp(p, "__CLANG_OFFLOAD_BUNDLE__", 24) == 0) return parse_bundles(p, n, out);
    if (n >= 64 && memcmp(p, "\x7f" "ELF", 4) == 0 && rd16(p + 0x12) == 224) return parse_code_object(p, n, out);
    std::vector<Section> sec;
    uint16_t shstrndx = 0;
    if (!elf_sections(p, n, sec, &shstrndx)) return false;
    const Section& names = sec[shstrndx];
    if (names.type == 8 /* SHT_NOBITS */) return false;
    bool any = false;
    for (const Section& s : sec) {
        if (s.type == 8 /* SHT_NOBITS */) continue;
        if (s.name >= names.size || strncmp((const char*)p + names.off + s.name, ".hip_fatbin", (size_t)(names.size - s.name)) != 0) continue;
        any = parse_bundles(p + s.off, (size_t)s.size, out) || any;
    }
    return any;
}

// the kernels of THIS shared library (the file the calling code was loaded from), read once
inline const Table& own_library() {
    static Table table;
    static std::once_flag once;
    std::call_once(once, [] {
        Dl_info info;
        if (dladdr((const void*)&own_library, &info) && info.dli_fname) (void)parse_file(info.dli_fname, table);
    });
    return table;
}

// the one kernel whose name contains every fragment (nullptr if none or more than one does)
inline const KernelResources* find(const Table& t, const std::vector<std::string>& fragments, std::string* name_out = nullptr) {
    const KernelResources* hit = nullptr;
    for (const auto& e : t) {
        bool all = true;
        for (const std::string& f : fragments) all = all && e.first.find(f) != std::string::npos;
        if (!all) continue;
        if (hit) return nullptr;
        hit = &e.second;
        if (name_out) *name_out = e.first;
    }
    return hit;
}

}  // namespace kd
}  // namespace vit
