// Wave-pipelined LDS-tiled K1/K2/K3 instantiations for value type float.
#include "wavetile_impl.h"

namespace tsgu {
int wavetile_dispatch_f32(int itype, int mode, const WtParams& P, bool can_wide, int n_cu, hipStream_t stream) {
#define GO(I)                                                                                    \
    switch (mode) {                                                                              \
        case kWtSpmm: return wavetile_launch<float, I, kWtSpmm>(P, can_wide, n_cu, stream);         \
        case kWtSpmmPerm: return wavetile_launch<float, I, kWtSpmmPerm>(P, can_wide, n_cu, stream); \
        case kWtSddmm: return wavetile_launch<float, I, kWtSddmm>(P, can_wide, n_cu, stream);       \
        case kWtBwd: return wavetile_launch<float, I, kWtBwd>(P, can_wide, n_cu, stream);           \
    }                                                                                            \
    return TSGU_ERR_BAD_ARG;
    if (itype == TSGU_I32) { GO(int32_t) }
    if (itype == TSGU_I64) { GO(int64_t) }
#undef GO
    return TSGU_ERR_BAD_DTYPE;
}
}  // namespace tsgu
