// Plane-march kernels by displacement set: one translation unit per group of sets (the kernels are large; the build compiles
// the groups in parallel).  Every function returns a tsgu status, or kMarchNotMine for a set it has no kernels for.
#pragma once

#include "march_impl.h"

namespace tsgu {

// Displacement sets (bit (dx+1)·9 + (dy+1)·3 + dz+1): the whole box, the 7-point cross, and the triangular halves of both by
// displacement — with / without the centre.
constexpr int kMarchNotMine = 1;
constexpr uint32_t kMarchCross = (1u << 4) | (1u << 10) | (1u << 12) | (1u << 13) | (1u << 14) | (1u << 16) | (1u << 22);
constexpr uint32_t kMarchLowerIncl = (1u << 14) - 1u, kMarchLowerStrict = (1u << 13) - 1u;
constexpr uint32_t kMarchUpperIncl = kBoxAll & ~kMarchLowerStrict, kMarchUpperStrict = kBoxAll & ~kMarchLowerIncl;

// What is compiled (measured at C2's lattice, 32 columns, us per launch — general plane sweep / plane march):
//   the whole box      all three products                          (periodic 91 / 113 / 108 against 81 / 103 / 92)
//   triangular halves  the SDDMM only, 256 threads                 (lower half of the truncated box: 89 against 78; its stored-order
//                      and transposed products are faster on the sweep: 71 / 72 against 87 / 106)
//   the 7-point cross  nothing: the sweep wins all three (59 / 59 / 50 against 73 / 74 / 78), as it does for sets given at run time
constexpr int kMarchSubsetThreads = 256;

int march_run_box(int mode, int cl, int threads, bool uni, const MarchParams& P, hipStream_t s);    // march_box.hip
int march_run_lower(int cl, const MarchParams& P, hipStream_t s);                                   // march_lower.hip (SDDMM)
int march_run_upper(int cl, const MarchParams& P, hipStream_t s);                                   // march_upper.hip (SDDMM)

// is there a kernel for (product, displacement set, rows of one length)?
constexpr bool march_is_triangular(uint32_t m) {
    return m == kMarchLowerIncl || m == kMarchLowerStrict || m == kMarchUpperIncl || m == kMarchUpperStrict ||
           m == (kMarchLowerIncl & kMarchCross) || m == (kMarchLowerStrict & kMarchCross) || m == (kMarchUpperIncl & kMarchCross) ||
           m == (kMarchUpperStrict & kMarchCross);
}
inline bool march_supported(int mode, uint32_t mask, bool uni, int threads) {
    if (mask == kBoxAll) return threads == 256 || threads == 512;
    return mode == kLatSddmm && !uni && threads == kMarchSubsetThreads && march_is_triangular(mask);
}

template <int MODE, int NT, uint32_t MASK, int ROWS>
int march_by_lanes(int cl, const MarchParams& P, hipStream_t s) {
    switch (cl) {
        case 4: return march_launch<float, 4, MODE, NT, MASK, ROWS>(P, s);
        case 8: return march_launch<float, 8, MODE, NT, MASK, ROWS>(P, s);
        case 16: return march_launch<float, 16, MODE, NT, MASK, ROWS>(P, s);
    }
    return TSGU_ERR_BAD_ARG;
}

// rows that are not of one length: the whole box knows their starts by arithmetic, every other set reads the row pointer
constexpr int march_ragged_rows(uint32_t mask) { return mask == kBoxAll ? kRowsBox : kRowsPointer; }

// the SDDMM of one displacement set at the workgroup size of the compiled subsets (rows by row pointer)
template <uint32_t MASK>
int march_subset_sddmm(int cl, const MarchParams& P, hipStream_t s) {
    return march_by_lanes<kLatSddmm, kMarchSubsetThreads, MASK, kRowsPointer>(cl, P, s);
}

// one displacement set at both workgroup sizes
template <uint32_t MASK>
int march_both_sizes(int mode, int cl, int threads, bool uni, const MarchParams& P, hipStream_t s) {
#define TSGU_MARCH_CASE(M, T)                                                                   \
    if (mode == M && threads == T) {                                                            \
        if constexpr (MASK == kBoxAll && M != kLatSddmm) {                                      \
            if (uni && P.raw) return march_by_lanes<M, T, MASK, kRowsRaw>(cl, P, s);            \
        }                                                                                       \
        return uni ? march_by_lanes<M, T, MASK, kRowsUniform>(cl, P, s) : march_by_lanes<M, T, MASK, march_ragged_rows(MASK)>(cl, P, s); \
    }
    TSGU_MARCH_CASE(kLatSpmm, 256)
    TSGU_MARCH_CASE(kLatSpmm, 512)
    TSGU_MARCH_CASE(kLatSddmm, 256)
    TSGU_MARCH_CASE(kLatSddmm, 512)
    TSGU_MARCH_CASE(kLatSpmmT, 256)
    TSGU_MARCH_CASE(kLatSpmmT, 512)
#undef TSGU_MARCH_CASE
    return TSGU_ERR_BAD_ARG;
}

}  // namespace tsgu
