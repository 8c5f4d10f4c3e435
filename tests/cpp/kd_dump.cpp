// kd_dump.cpp -- prints the kernel-descriptor table csrc/kernel_desc.hpp reads out of a file (a library with a .hip_fatbin section, a
// `hipcc --genco` offload bundle, or a bare code object): "<vgpr_alloc> <accum_offset> <lds_static> <scratch> <kernel>" per line.
// Host-only; tests/test_host.py feeds it a freshly compiled .hsaco and compares with hipcc's -S output.
#include <stdio.h>

#include "../../viterbidecodercpp_amd/csrc/kernel_desc.hpp"

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: kd_dump <file>\n"); return 2; }
    vit::kd::Table t;
    if (!vit::kd::parse_file(argv[1], t)) { fprintf(stderr, "kd_dump: no kernel descriptors found in %s\n", argv[1]); return 1; }
    for (const auto& e : t)
        printf("%u %u %u %u %s\n", e.second.vgpr_alloc, e.second.accum_offset, e.second.lds_static_bytes, e.second.scratch_bytes, e.first.c_str());
    return 0;
}
