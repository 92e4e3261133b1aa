// Explicit instantiation of the fused pdr row kernels for double.
#include "ddn_launch.hpp"
template int row_launch<double, PdrP<double>>(hipStream_t, int, int, size_t, int, const RowCommon<double>&, const PdrP<double>::Args&);
