// Explicit instantiation of the fused pddr row kernels for float.
#include "ddn_launch.hpp"
template int row_launch<float, PddrP<float>>(hipStream_t, int, int, size_t, int, const RowCommon<float>&, const PddrP<float>::Args&);
