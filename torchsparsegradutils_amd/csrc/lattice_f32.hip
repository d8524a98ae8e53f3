// fp32 instantiations of the lattice plane-sweep kernels (own translation unit: compiles in parallel with the others).
#include "lattice_impl.h"

namespace tsgu {

template <int MODE, int NT>
static int lat_go_f32(int cl, const LatParams& P, hipStream_t s) {
    switch (cl) {
        case 1:   // 16-byte dense rows (4 fp32 columns: the Krylov loops' right-hand sides): the stored-order product only
            if constexpr (MODE == kLatSpmm) return lat_launch_one<float, 1, MODE, NT>(P, s);
            else return TSGU_ERR_BAD_ARG;
        case 2: return lat_launch_one<float, 2, MODE, NT>(P, s);
        case 4: return lat_launch_one<float, 4, MODE, NT>(P, s);
        case 8: return lat_launch_one<float, 8, MODE, NT>(P, s);
        case 16: return lat_launch_one<float, 16, MODE, NT>(P, s);
    }
    return TSGU_ERR_BAD_ARG;
}

template <int NT>
static int lat_mode_f32(int mode, int cl, const LatParams& P, hipStream_t s) {
    switch (mode) {
        case kLatSpmm: return lat_go_f32<kLatSpmm, NT>(cl, P, s);
        case kLatSddmm: return lat_go_f32<kLatSddmm, NT>(cl, P, s);
        case kLatSpmmT: return lat_go_f32<kLatSpmmT, NT>(cl, P, s);
    }
    return TSGU_ERR_BAD_ARG;
}

int lat_dispatch_f32(int mode, int cl, int threads, const LatParams& P, hipStream_t s) {
    switch (threads) {
        case 256: return lat_mode_f32<256>(mode, cl, P, s);
        case 512: return lat_mode_f32<512>(mode, cl, P, s);
        case 1024: return lat_mode_f32<1024>(mode, cl, P, s);
    }
    return TSGU_ERR_BAD_ARG;
}

}  // namespace tsgu
