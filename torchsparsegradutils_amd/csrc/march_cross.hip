// Plane-march kernels for the 7-point cross (periodic or truncated): march_sets.h.
#include "march_sets.h"

namespace tsgu {
int march_run_cross(int mode, int cl, bool uni, const MarchParams& P, hipStream_t s) {
    return uni ? march_subset<kMarchCross, kRowsUniform>(mode, cl, P, s) : march_subset<kMarchCross, kRowsPointer>(mode, cl, P, s);
}
}  // namespace tsgu
