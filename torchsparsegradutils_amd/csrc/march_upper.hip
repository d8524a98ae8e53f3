// Plane-march kernels for the upper triangular halves (by displacement) of truncated 27- and 7-point stencils: march_sets.h.
#include "march_sets.h"

namespace tsgu {
int march_run_upper(int mode, int cl, const MarchParams& P, hipStream_t s) {
    switch (P.mask) {
        case kMarchUpperIncl: return march_subset<kMarchUpperIncl, kRowsPointer>(mode, cl, P, s);
        case kMarchUpperStrict: return march_subset<kMarchUpperStrict, kRowsPointer>(mode, cl, P, s);
        case kMarchUpperIncl & kMarchCross: return march_subset<kMarchUpperIncl & kMarchCross, kRowsPointer>(mode, cl, P, s);
        case kMarchUpperStrict & kMarchCross: return march_subset<kMarchUpperStrict & kMarchCross, kRowsPointer>(mode, cl, P, s);
    }
    return kMarchNotMine;
}
}  // namespace tsgu
