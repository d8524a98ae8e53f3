// Plane-march kernels for displacement sets given at run time (any subset of the box without compiled kernels): march_sets.h.
#include "march_sets.h"

namespace tsgu {
int march_run_any(int mode, int cl, int threads, bool uni, const MarchParams& P, hipStream_t s) {
    return march_both_sizes<0u>(mode, cl, threads, uni, P, s);
}
}  // namespace tsgu
