// Explicit instantiation of the Chambolle-Pock row kernels for float.
#include "ddn_launch.hpp"
template int cp_launch<float>(hipStream_t, int, int, size_t, int, const RowCommon<float>&, const CpArgs<float>&);
