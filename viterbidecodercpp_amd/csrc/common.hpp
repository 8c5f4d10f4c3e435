// common.hpp -- types shared by the kernel plans.
#pragma once
#include <stdint.h>

namespace vit {

// Decoder constants in the device's 16-bit domain: every value is (reference value << shift), shift = 0 for
// (int16_t,uint16_t) and 8 for (int8_t,uint8_t).
struct DevConfig {
    uint16_t max_error, init_start, init_non_start, threshold;
    int16_t high, low;
};

}  // namespace vit
