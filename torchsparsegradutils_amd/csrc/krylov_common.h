// Shared pieces of the Krylov kernels (CG in krylov.hip, BiCGSTAB in bicgstab.hip): lane layout for
// contiguous [n][p] arrays, deterministic block-level column sums, the partial-row fold kernel.
#pragma once

#include "tsgu_common.h"

namespace tsgu {

constexpr int kPasses = 4;  // row passes per block in the elementwise kernels

// Lane layout for a contiguous [n][p] array: lpr lanes per row, each VEC columns wide.
struct VecGeom {
    int vec;
    int lpr;  // lanes per row = ceil(p / vec)
    int rpp;  // rows per pass = 256 / lpr
    int64_t blocks;
};

inline bool vec_geom(int wide, bool can_wide, int64_t n, int64_t p, VecGeom& g) {
    g.vec = can_wide ? wide : 1;
    const int64_t lpr = (p + g.vec - 1) / g.vec;
    if (lpr > kBlock) return false;
    g.lpr = (int)lpr;
    g.rpp = kBlock / g.lpr;
    const int64_t rows_per_block = (int64_t)g.rpp * kPasses;
    g.blocks = (n + rows_per_block - 1) / rows_per_block;
    return true;
}

// Sum partial[b][c] over b for the calling block's column window [c0, c0+w): result valid
// in threads t < w (thread t owns column c0+t).  red: LDS scratch of kBlock Acc.
template <typename Acc>
__device__ __forceinline__ Acc block_colsum(const Acc* __restrict__ partial, int64_t n_partial, int64_t p,
                                            int64_t c0, int w, Acc* red) {
    const int t = threadIdx.x;
    const int subs = kBlock / w;
    const int sub = t / w, lc = t % w;
    Acc s = 0;
    if (sub < subs) {
        // four independent chains keep four loads in flight per thread (fixed order => deterministic)
        Acc s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int64_t b = sub;
        for (; b + 3 * (int64_t)subs < n_partial; b += 4 * (int64_t)subs) {
            s0 += partial[b * p + c0 + lc];
            s1 += partial[(b + subs) * p + c0 + lc];
            s2 += partial[(b + 2 * (int64_t)subs) * p + c0 + lc];
            s3 += partial[(b + 3 * (int64_t)subs) * p + c0 + lc];
        }
        for (; b < n_partial; b += subs) s0 += partial[b * p + c0 + lc];
        s = (s0 + s1) + (s2 + s3);
    }
    red[t] = s;
    __syncthreads();
    Acc tot = 0;
    if (t < w) {
        for (int k = 0; k < subs; ++k) tot += red[k * w + t];
    }
    __syncthreads();
    return tot;
}

// First level of a three-stage column sum: `rows` partial rows -> kFoldRows rows (block b sums the
// contiguous slice of rows [b*chunk, (b+1)*chunk) in row order).  Keeps the single-block finalisers
// (which need all columns at once) short when a kernel produced tens of thousands of partials.
constexpr int kFoldRows = 256;

template <typename V>
__global__ __launch_bounds__(kBlock) void colsum_fold_kernel(const V* __restrict__ partial, int64_t rows, int64_t p,
                                                             int64_t chunk, V* __restrict__ out, const int* __restrict__ flags) {
    __shared__ V red[kBlock];
    if (flags && flags[0] != 0) return;
    const int64_t r0 = (int64_t)blockIdx.x * chunk;
    int64_t r1 = r0 + chunk;
    r1 = r1 < rows ? r1 : rows;
    for (int64_t c0 = 0; c0 < p; c0 += 64) {
        const int w = (int)(p - c0 < 64 ? p - c0 : 64);
        const V tot = r0 < r1 ? block_colsum<V>(partial + r0 * p, r1 - r0, p, c0, w, red) : (V)0;
        if ((int)threadIdx.x < w) out[(int64_t)blockIdx.x * p + c0 + threadIdx.x] = tot;
    }
}

template <typename V>
inline bool geom_for(int64_t n, int64_t p, bool aligned, VecGeom& g) {
    constexpr int wide = VT<V>::kWide;
    return vec_geom(wide, aligned && (p % wide == 0), n, p, g);
}

}  // namespace tsgu

#define TSGU_VSWITCH(vtype, CALL_F32, CALL_F64) \
    do {                                        \
        if ((vtype) == TSGU_F32) {              \
            using V = float;                    \
            CALL_F32;                           \
        } else if ((vtype) == TSGU_F64) {       \
            using V = double;                   \
            CALL_F64;                           \
        } else {                                \
            return TSGU_ERR_BAD_DTYPE;          \
        }                                       \
    } while (0)

