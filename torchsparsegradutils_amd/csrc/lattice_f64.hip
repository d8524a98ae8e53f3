// fp64 instantiations of the lattice plane-sweep kernels (own translation unit: compiles in parallel with the others).
// Dense rows of p doubles = p/2 chunks of 16 bytes: p in {4, 8, 16, 32}.
#include "lattice_impl.h"

namespace tsgu {

template <int MODE, int NT>
static int lat_go_f64(int cl, const LatParams& P, hipStream_t s) {
    switch (cl) {
        case 2: return lat_launch_one<double, 2, MODE, NT>(P, s);
        case 4: return lat_launch_one<double, 4, MODE, NT>(P, s);
        case 8: return lat_launch_one<double, 8, MODE, NT>(P, s);
        case 16: return lat_launch_one<double, 16, MODE, NT>(P, s);
    }
    return TSGU_ERR_BAD_ARG;
}

template <int NT>
static int lat_mode_f64(int mode, int cl, const LatParams& P, hipStream_t s) {
    switch (mode) {
        case kLatSpmm: return lat_go_f64<kLatSpmm, NT>(cl, P, s);
        case kLatSddmm: return lat_go_f64<kLatSddmm, NT>(cl, P, s);
        case kLatSpmmT: return lat_go_f64<kLatSpmmT, NT>(cl, P, s);
    }
    return TSGU_ERR_BAD_ARG;
}

int lat_dispatch_f64(int mode, int cl, int threads, const LatParams& P, hipStream_t s) {
    if (P.cpl != 1) return TSGU_ERR_BAD_ARG;
    switch (threads) {
        case 256: return lat_mode_f64<256>(mode, cl, P, s);
        case 512: return lat_mode_f64<512>(mode, cl, P, s);
        case 1024: return lat_mode_f64<1024>(mode, cl, P, s);
    }
    return TSGU_ERR_BAD_ARG;
}

}  // namespace tsgu
