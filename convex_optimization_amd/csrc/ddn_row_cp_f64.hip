// Explicit instantiation of the Chambolle-Pock row kernels for double.
#include "ddn_launch.hpp"
template int cp_launch<double>(hipStream_t, int, int, size_t, int, const RowCommon<double>&, const CpArgs<double>&);
