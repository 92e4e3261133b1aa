// Explicit instantiation of the fused pddr row kernels for float.
#define DDN_SW_ONE_TERM 1   // one row per workgroup in the LDS: the one-term index swizzle (ddn_fft.hpp)
#include "ddn_launch.hpp"
template int row_launch<float, PddrP<float>>(hipStream_t, int, int, size_t, int, const RowCommon<float>&, const PddrP<float>::Args&);
