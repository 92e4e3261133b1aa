// Explicit instantiation of the fused pddr row kernels for double.
#include "ddn_launch.hpp"
template int row_launch<double, PddrP<double>>(hipStream_t, int, int, size_t, int, const RowCommon<double>&, const PddrP<double>::Args&);
