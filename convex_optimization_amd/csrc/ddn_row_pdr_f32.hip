// Explicit instantiation of the fused pdr row kernels for float.
#include "ddn_launch.hpp"
template int row_launch<float, PdrP<float>>(hipStream_t, int, int, size_t, int, const RowCommon<float>&, const PdrP<float>::Args&);
