// Plane-march kernels for the lower triangular halves (by displacement) of truncated 27- and 7-point stencils: march_sets.h.
#include "march_sets.h"

namespace tsgu {
int march_run_lower(int mode, int cl, const MarchParams& P, hipStream_t s) {
    switch (P.mask) {
        case kMarchLowerIncl: return march_subset<kMarchLowerIncl, kRowsPointer>(mode, cl, P, s);
        case kMarchLowerStrict: return march_subset<kMarchLowerStrict, kRowsPointer>(mode, cl, P, s);
        case kMarchLowerIncl & kMarchCross: return march_subset<kMarchLowerIncl & kMarchCross, kRowsPointer>(mode, cl, P, s);
        case kMarchLowerStrict & kMarchCross: return march_subset<kMarchLowerStrict & kMarchCross, kRowsPointer>(mode, cl, P, s);
    }
    return kMarchNotMine;
}
}  // namespace tsgu
