// Plane-march SDDMM for the lower triangular halves (by displacement) of truncated 27- and 7-point stencils: march_sets.h.
#include "march_sets.h"

namespace tsgu {
int march_run_lower(int cl, const MarchParams& P, hipStream_t s) {
    switch (P.mask) {
        case kMarchLowerIncl: return march_subset_sddmm<kMarchLowerIncl>(cl, P, s);
        case kMarchLowerStrict: return march_subset_sddmm<kMarchLowerStrict>(cl, P, s);
        case kMarchLowerIncl & kMarchCross: return march_subset_sddmm<kMarchLowerIncl & kMarchCross>(cl, P, s);
        case kMarchLowerStrict & kMarchCross: return march_subset_sddmm<kMarchLowerStrict & kMarchCross>(cl, P, s);
    }
    return kMarchNotMine;
}
}  // namespace tsgu
