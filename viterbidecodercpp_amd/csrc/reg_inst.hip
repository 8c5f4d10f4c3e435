// reg_inst.hip -- one PLAN_REG code per translation unit: compile with -DVIT_REG_ID=<0..4> (see Makefile).
#include "kernels_reg.hpp"
