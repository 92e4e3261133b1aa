// Explicit instantiation of the fused admm row kernels for double.
#include "ddn_launch.hpp"
template int row_launch<double, AdmmP<double>>(hipStream_t, int, int, size_t, int, const RowCommon<double>&, const AdmmP<double>::Args&);
