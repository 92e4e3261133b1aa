// Resolvent-mode column kernels of common image heights (see ddn_colfix.hpp).
#define DDN_COLFIX_MODE COL_RES2
#define DDN_COLFIX_NAME col_fixed_res2_f32
#define DDN_COLFIX_STAGE_MODE COL_FWD2
#define DDN_COLFIX_STAGE_NAME col_fixed_fwd2_f32
#include "ddn_colfix.hpp"
