// Plane-march SDDMM for the upper triangular halves (by displacement) of truncated 27- and 7-point stencils: march_sets.h.
#include "march_sets.h"

namespace tsgu {
int march_run_upper(int cl, const MarchParams& P, hipStream_t s) {
    switch (P.mask) {
        case kMarchUpperIncl: return march_subset_sddmm<kMarchUpperIncl>(cl, P, s);
        case kMarchUpperStrict: return march_subset_sddmm<kMarchUpperStrict>(cl, P, s);
        case kMarchUpperIncl & kMarchCross: return march_subset_sddmm<kMarchUpperIncl & kMarchCross>(cl, P, s);
        case kMarchUpperStrict & kMarchCross: return march_subset_sddmm<kMarchUpperStrict & kMarchCross>(cl, P, s);
    }
    return kMarchNotMine;
}
}  // namespace tsgu
