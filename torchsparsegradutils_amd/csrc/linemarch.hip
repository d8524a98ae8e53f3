// Whole-line plane march (linemarch_impl.h): bf16 instantiations and the launcher the march entry points (march.hip) call.
#include "linemarch_impl.h"

namespace tsgu {

// fills P from the plan (geometry, tile) and validates it; returns the dynamic LDS bytes or a negative status
int linemarch_fill(LineParams& P, const tsgu_march_plan* pl, int mode, int64_t p, int64_t n_rows, int64_t nnz) {
    if (!pl || p != 16) return TSGU_ERR_BAD_DTYPE;
    if (pl->ntap != 9 || pl->ry != 1 || pl->rz != 1 || pl->mask != (1u << 27) - 1u || (pl->periodic & 7) != 7 || pl->uniform_len != 27) return TSGU_ERR_BAD_ARG;
    if (pl->nb <= 0 || pl->nx < 3 || pl->ny < 3 || pl->nz < 3 || pl->nseg <= 0 || pl->nseg > pl->nx || pl->tz != pl->nz) return TSGU_ERR_BAD_ARG;
    if (n_rows >= 0 && ((int64_t)pl->nb * pl->nx * pl->ny * pl->nz != n_rows || 27 * n_rows != nnz)) return TSGU_ERR_BAD_ARG;
    if (nnz > 0x7fffffffLL || nnz * 2 + 16 > 0xffffffffLL) return TSGU_ERR_TOO_LARGE;
    P.nb = pl->nb, P.nx = pl->nx, P.ny = pl->ny, P.nz = pl->nz;
    P.ty = pl->ty;
    P.tiles_y = pl->ny / (pl->ty > 0 ? pl->ty : 1);
    P.nseg = pl->nseg;
    P.seg_len = (pl->nx + pl->nseg - 1) / pl->nseg;
    if ((int64_t)(P.nseg - 1) * P.seg_len >= pl->nx) return TSGU_ERR_BAD_ARG;
    const int lds = linemarch_layout(P, pl->threads, mode);
    if (lds < 0) return lds;
    P.nblocks = (int64_t)P.nb * P.nseg * P.tiles_y;
    if (P.nblocks > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    return lds;
}

template <int NT, int MODE>
static int run_nz(const LineParams& P, hipStream_t stream) {
    switch (P.nz) {
        case 8: return linemarch_launch_t<NT, 8, MODE>(P, stream);
        case 16: return linemarch_launch_t<NT, 16, MODE>(P, stream);
        case 32: return linemarch_launch_t<NT, 32, MODE>(P, stream);
        case 64: return linemarch_launch_t<NT, 64, MODE>(P, stream);
    }
    return TSGU_ERR_BAD_ARG;
}

template <int MODE>
static int run_nt(const LineParams& P, int threads, hipStream_t stream) {
    switch (threads) {
        case 256: return run_nz<256, MODE>(P, stream);
        case 512: return run_nz<512, MODE>(P, stream);
        case 1024: return run_nz<1024, MODE>(P, stream);
    }
    return TSGU_ERR_BAD_ARG;
}

int linemarch_run(int mode, const LineParams& P, int threads, hipStream_t stream) {
    if (mode == kLatSpmmT) return run_nt<kLatSpmmT>(P, threads, stream);
    if (mode == kLatSpmm) return run_nt<kLatSpmm>(P, threads, stream);
    if (mode == kLatSddmm) return run_nt<kLatSddmm>(P, threads, stream);
    return TSGU_ERR_BAD_ARG;
}

}  // namespace tsgu
