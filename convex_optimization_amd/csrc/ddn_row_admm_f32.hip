// Explicit instantiation of the fused admm row kernels for float.
#include "ddn_launch.hpp"
template int row_launch<float, AdmmP<float>>(hipStream_t, int, int, size_t, int, const RowCommon<float>&, const AdmmP<float>::Args&);
