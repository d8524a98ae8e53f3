// bf16 instantiations of the lattice plane-sweep kernels (fp32 accumulation).
#include "lattice_impl.h"

namespace tsgu {

template <int MODE, int NT>
static int lat_go_bf16(int cl, const LatParams& P, hipStream_t s) {
    switch (cl) {
        case 2: return lat_launch_one<bf16_t, 2, MODE, NT>(P, s);
        case 4: return lat_launch_one<bf16_t, 4, MODE, NT>(P, s);
        case 8: return lat_launch_one<bf16_t, 8, MODE, NT>(P, s);
        case 16: return lat_launch_one<bf16_t, 16, MODE, NT>(P, s);
    }
    return TSGU_ERR_BAD_ARG;
}

template <int NT>
static int lat_mode_bf16(int mode, int cl, const LatParams& P, hipStream_t s) {
    switch (mode) {
        case kLatSpmm: return lat_go_bf16<kLatSpmm, NT>(cl, P, s);
        case kLatSddmm: return lat_go_bf16<kLatSddmm, NT>(cl, P, s);
        case kLatSpmmT: return lat_go_bf16<kLatSpmmT, NT>(cl, P, s);
    }
    return TSGU_ERR_BAD_DTYPE;
}

int lat_dispatch_bf16(int mode, int cl, int threads, const LatParams& P, hipStream_t s) {
    switch (threads) {
        case 256: return lat_mode_bf16<256>(mode, cl, P, s);
        case 512: return lat_mode_bf16<512>(mode, cl, P, s);
        case 1024: return lat_mode_bf16<1024>(mode, cl, P, s);
    }
    return TSGU_ERR_BAD_ARG;
}

}  // namespace tsgu
