"""MI355X-native implementation of the deblur_denoise iteration hot path of
rahul3/convex_optimization: same public Python API, hand-written gfx950 HIP kernels underneath
(libddn, include/ddn.h).  There is no CPU path: every compute entry point needs the compiled
library and a HIP device."""
from . import core, op_math, utils
from . import admm, chambolle_pock, primal_dr, primal_dual_dr
from .solvers import admm_solver, chambolle_pock as chambolle_pock_solver, primal_dr_splitting, primal_dual_dr_splitting
from .op_math import DeblurDenoiseOperators, apply_periodic_conv_2d, eig_vals_for_periodic_conv_op
from . import blur_deblur, distributed, slab
from .blur_deblur import blur_image, deblur_image, blur_and_deblur_image

__version__ = "0.1.0"
