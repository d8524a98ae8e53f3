// K3 instantiations for value type double (index types int32 / int64).
#include "sddmm_impl.h"

namespace tsgu {
int sddmm_dispatch_f64(int itype, const SddmmParams& P, int64_t batch, hipStream_t stream) {
    if (itype == TSGU_I32) return sddmm_launch<double, int32_t>(P, batch, stream);
    if (itype == TSGU_I64) return sddmm_launch<double, int64_t>(P, batch, stream);
    return TSGU_ERR_BAD_DTYPE;
}
int coo_sddmm_dispatch_f64(int itype, const CooSddmmParams& P, hipStream_t stream) {
    if (itype == TSGU_I32) return coo_sddmm_launch<double, int32_t>(P, stream);
    if (itype == TSGU_I64) return coo_sddmm_launch<double, int64_t>(P, stream);
    return TSGU_ERR_BAD_DTYPE;
}
}  // namespace tsgu
