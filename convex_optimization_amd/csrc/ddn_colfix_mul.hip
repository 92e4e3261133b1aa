// Multiplier-mode column kernels of common image heights (see ddn_colfix.hpp).
#define DDN_COLFIX_MODE COL_MUL
#define DDN_COLFIX_NAME col_fixed_mul_f32
#define DDN_COLFIX_STAGE_MODE COL_FWD
#define DDN_COLFIX_STAGE_NAME col_fixed_fwd_f32
#include "ddn_colfix.hpp"
