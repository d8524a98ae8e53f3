// K1/K2 instantiations for value type double (index types int32 / int64).
#include "spmm_impl.h"

namespace tsgu {
int spmm_dispatch_f64(int itype, const SpmmParams& P, int64_t batch, hipStream_t stream) {
    if (itype == TSGU_I32) return spmm_launch<double, int32_t>(P, batch, stream);
    if (itype == TSGU_I64) return spmm_launch<double, int64_t>(P, batch, stream);
    return TSGU_ERR_BAD_DTYPE;
}
}  // namespace tsgu
