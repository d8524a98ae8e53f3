// Plane-march kernels for the whole 3 x 3 x 3 box (27-point stencils, periodic or truncated): march_sets.h.
#include "march_sets.h"

namespace tsgu {
int march_run_box(int mode, int cl, int threads, bool uni, const MarchParams& P, hipStream_t s) {
    return march_both_sizes<kBoxAll>(mode, cl, threads, uni, P, s);
}
}  // namespace tsgu
