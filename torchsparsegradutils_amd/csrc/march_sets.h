// Plane-march kernels by displacement set: one translation unit per group of sets (the kernels are large; the build compiles
// the groups in parallel).  Every function returns a tsgu status, or kMarchNotMine for a set it has no kernels for.
#pragma once

#include "march_impl.h"

namespace tsgu {

// Displacement sets the kernels are compiled for (bit (dx+1)·9 + (dy+1)·3 + dz+1): the whole box, the 7-point cross, and the
// triangular halves of both by displacement — with / without the centre.  Any other subset runs the run-time-mask kernels.
constexpr int kMarchNotMine = 1;
constexpr uint32_t kMarchCross = (1u << 4) | (1u << 10) | (1u << 12) | (1u << 13) | (1u << 14) | (1u << 16) | (1u << 22);
constexpr uint32_t kMarchLowerIncl = (1u << 14) - 1u, kMarchLowerStrict = (1u << 13) - 1u;
constexpr uint32_t kMarchUpperIncl = kBoxAll & ~kMarchLowerStrict, kMarchUpperStrict = kBoxAll & ~kMarchLowerIncl;

// workgroup size the compiled subsets exist for (the whole box and the run-time-mask kernels: 256 and 512)
constexpr int march_subset_threads(int mode) { return mode == kLatSpmm ? 256 : 512; }

int march_run_box(int mode, int cl, int threads, bool uni, const MarchParams& P, hipStream_t s);    // march_box.hip
int march_run_cross(int mode, int cl, bool uni, const MarchParams& P, hipStream_t s);               // march_cross.hip
int march_run_lower(int mode, int cl, const MarchParams& P, hipStream_t s);                         // march_lower.hip
int march_run_upper(int mode, int cl, const MarchParams& P, hipStream_t s);                         // march_upper.hip
int march_run_any(int mode, int cl, int threads, bool uni, const MarchParams& P, hipStream_t s);    // march_any.hip

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

// one displacement set at the workgroup size of the compiled subsets, all three products
template <uint32_t MASK, int ROWS>
int march_subset(int mode, int cl, const MarchParams& P, hipStream_t s) {
    switch (mode) {
        case kLatSpmm: return march_by_lanes<kLatSpmm, march_subset_threads(kLatSpmm), MASK, ROWS>(cl, P, s);
        case kLatSddmm: return march_by_lanes<kLatSddmm, march_subset_threads(kLatSddmm), MASK, ROWS>(cl, P, s);
        case kLatSpmmT: return march_by_lanes<kLatSpmmT, march_subset_threads(kLatSpmmT), MASK, ROWS>(cl, P, s);
    }
    return TSGU_ERR_BAD_ARG;
}

// one displacement set at both workgroup sizes
template <uint32_t MASK>
int march_both_sizes(int mode, int cl, int threads, bool uni, const MarchParams& P, hipStream_t s) {
#define TSGU_MARCH_CASE(M, T)                                                                   \
    if (mode == M && threads == T)                                                              \
        return uni ? march_by_lanes<M, T, MASK, kRowsUniform>(cl, P, s) : march_by_lanes<M, T, MASK, march_ragged_rows(MASK)>(cl, P, s);
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
