// kernels_reg.hpp -- PLAN_REG placeholder (register-resident kernels land in the next commit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_lds.hpp"
namespace vit {
struct RegCode { int K = 0, R = 0; };
inline bool reg_code_supported(int, int) { return false; }
inline bool reg_code_init(RegCode*, int, int, const uint32_t*, const DevConfig&) { return false; }
inline size_t reg_workspace_bytes(const RegCode&, size_t, size_t) { return 0; }
inline int reg_update(const RegCode&, const DevConfig&, int, const void*, size_t, size_t, size_t, void*, void*, uint64_t*, const uint32_t*, hipStream_t) { return -1; }
inline int reg_chainback(const RegCode&, const void*, size_t, size_t, uint8_t*, const uint32_t*, hipStream_t) { return -1; }
inline int reg_export(const RegCode&, const void*, size_t, size_t, size_t, uint64_t*, hipStream_t) { return -1; }
}
