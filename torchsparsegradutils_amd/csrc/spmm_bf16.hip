// K1/K2 instantiations for value type bf16_t (index types int32 / int64).
#include "spmm_impl.h"

namespace tsgu {
int spmm_dispatch_bf16(int itype, const SpmmParams& P, int64_t batch, hipStream_t stream) {
    if (itype == TSGU_I32) return spmm_launch<bf16_t, int32_t>(P, batch, stream);
    if (itype == TSGU_I64) return spmm_launch<bf16_t, int64_t>(P, batch, stream);
    return TSGU_ERR_BAD_DTYPE;
}
}  // namespace tsgu
