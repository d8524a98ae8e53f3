// Row-pair gather kernels: fp32 instantiations (one translation unit per value type for build parallelism).
#include "rowpack_impl.h"

namespace tsgu {

int rp_dispatch_f32(int mode, bool perm, int itype, const RpParams& P, hipStream_t s) {
#define TSGU_RP_I(IT)                                                                       \
    switch (mode) {                                                                         \
        case kRpSpmm: return perm ? rp_launch<float, IT, kRpSpmm, true>(P, s) : rp_launch<float, IT, kRpSpmm, false>(P, s); \
        case kRpBwd: return rp_launch<float, IT, kRpBwd, true>(P, s);                       \
        case kRpSddmm: return rp_launch<float, IT, kRpSddmm, false>(P, s);                  \
    }                                                                                       \
    return TSGU_ERR_BAD_ARG;
    if (itype == TSGU_I32) { TSGU_RP_I(int32_t) }
    if (itype == TSGU_I64) { TSGU_RP_I(int64_t) }
#undef TSGU_RP_I
    return TSGU_ERR_BAD_DTYPE;
}

}  // namespace tsgu
