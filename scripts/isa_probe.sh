#!/bin/bash
# Which packed / 8-bit integer arithmetic does the gfx950 assembler know?  (evidence for DESIGN.md section 3: u8 metrics ride in the
# high byte of 16-bit halves because there is no packed-u8 add / min / sub to run four frames per instruction)
MC=/opt/rocm/lib/llvm/bin/llvm-mc
for ins in "v_pk_add_u8 v0, v1, v2" "v_pk_min_u8 v0, v1, v2" "v_pk_sub_u8 v0, v1, v2" "v_pk_max_u8 v0, v1, v2" "v_add_u8 v0, v1, v2" \
           "v_min_u8 v0, v1, v2" "v_sad_u8 v0, v1, v2, v3" "v_msad_u8 v0, v1, v2, v3" "v_qsad_pk_u16_u8 v[0:1], v[2:3], v4, v[6:7]" \
           "v_dot4_u32_u8 v0, v1, v2, v3" "v_pk_add_u16 v0, v1, v2" "v_pk_min_i16 v0, v1, v2" "v_pk_sub_i16 v0, v1, v2 clamp" \
           "v_pk_add_min_u16 v0, v1, v2, v3" "v_pk_min3_u16 v0, v1, v2, v3" "v_min3_u16 v0, v1, v2, v3" "v_add_min_u32 v0, v1, v2, v3" \
           "v_bitop3_b32 v0, v1, v2, v3 bitop3:0xe4" "v_permlane16_swap_b32 v0, v1" "v_permlane32_swap_b32 v0, v1" "v_permlane8_swap_b32 v0, v1"; do
    printf "%-52s " "$ins"
    if echo "$ins" | $MC -arch=amdgcn -mcpu=gfx950 -show-encoding 2>&1 | grep -q "error"; then echo "not an instruction of gfx950"; else echo "assembles"; fi
done
