// Plane-march kernels (march_impl.h): extern "C" entry points (declared in include/tsgu_hip.h) and the fp32 instantiations.
#include "linemarch_impl.h"
#include "march_sets.h"

using namespace tsgu;

namespace tsgu {
int linemarch_fill(LineParams& P, const tsgu_march_plan* pl, int mode, int64_t p, int64_t n_rows, int64_t nnz);
int linemarch_run(int mode, const LineParams& P, int threads, hipStream_t stream);
}  // namespace tsgu

namespace {

int march_lanes(int vtype, int64_t p) {
    if (vtype != TSGU_F32 || p <= 0 || (p * 4) % 16) return 0;
    const int64_t cl = p * 4 / 16;
    return (cl == 4 || cl == 8 || cl == 16) ? (int)cl : 0;   // p = 16, 32 or 64
}

int fill(MarchParams& P, const tsgu_march_plan* pl, int mode, int vtype, int64_t p, int64_t n_rows, int64_t nnz, int& cl) {
    if (!pl || n_rows < 0 || nnz < 0) return TSGU_ERR_BAD_ARG;
    cl = march_lanes(vtype, p);
    if (cl == 0) return TSGU_ERR_BAD_DTYPE;
    if (pl->ntap != 9 || pl->ry != 1 || pl->rz != 1) return TSGU_ERR_BAD_ARG;
    if (pl->nb <= 0 || pl->nx < 3 || pl->ny < 3 || pl->nz < 3 || pl->nseg <= 0 || pl->nseg > pl->nx) return TSGU_ERR_BAD_ARG;
    if ((int64_t)pl->nb * pl->nx * pl->ny * pl->nz != n_rows) return TSGU_ERR_BAD_ARG;
    if (pl->mask == 0 || pl->mask >= (1u << 27) || pl->uniform_len < 0 || pl->uniform_len > 27) return TSGU_ERR_BAD_ARG;
    if (pl->uniform_len > 0 ? (int64_t)pl->uniform_len * n_rows != nnz : (pl->rstart == nullptr && pl->mask != kBoxAll)) return TSGU_ERR_BAD_ARG;
    if (pl->mask == kBoxAll && pl->uniform_len == 0) {
        // the whole box on a lattice truncated somewhere: row starts by arithmetic — 3 entries per point and dimension, 2 at a face
        auto line = [](int n, int per) -> int64_t { return per ? 3 * (int64_t)n : 3 * (int64_t)n - 2; };
        if (pl->nb * line(pl->nx, pl->periodic & 1) * line(pl->ny, pl->periodic & 2) * line(pl->nz, pl->periodic & 4) != nnz) return TSGU_ERR_BAD_ARG;
        if (36 * (int64_t)pl->ny * pl->nz >= (1 << 24)) return TSGU_ERR_TOO_LARGE;    // the per-row constants go through 24-bit multiplies
    }
    if (pl->threads != 256 && pl->threads != 512) return TSGU_ERR_BAD_ARG;
    if (!pl->kidx || !pl->rcls || pl->ncls <= 0 || pl->ncls > kMarchMaxCls || pl->ident < 0 || pl->ident >= pl->ncls) return TSGU_ERR_BAD_ARG;
    if (nnz > 0x7fffffffLL || n_rows > 0x7fffffffLL || nnz * 4 + 16 > 0xffffffffLL) return TSGU_ERR_TOO_LARGE;
    P.nb = pl->nb, P.nx = pl->nx, P.ny = pl->ny, P.nz = pl->nz;
    P.ty = pl->ty, P.tz = pl->tz, P.ry = pl->ry, P.rz = pl->rz;
    P.tiles_y = (pl->ny + pl->ty - 1) / pl->ty;
    P.tiles_z = (pl->nz + pl->tz - 1) / pl->tz;
    P.nseg = pl->nseg;
    P.seg_len = (pl->nx + pl->nseg - 1) / pl->nseg;
    if ((int64_t)(P.nseg - 1) * P.seg_len >= pl->nx) return TSGU_ERR_BAD_ARG;
    P.ncls = pl->ncls, P.ident = pl->ident;
    P.mask = pl->mask;
    P.per_x = pl->periodic & 1, P.per_y = pl->periodic >> 1 & 1, P.per_z = pl->periodic >> 2 & 1;
    P.uniform = pl->uniform_len;
    // bit 3 of `periodic`: the caller has checked that rows store (dx, dy, dz) at 9·rank_x + 3·rank_y + rank_z — raw value rows
    P.raw = (pl->periodic & 8) != 0 && (pl->periodic & 7) == 7 && pl->mask == kBoxAll && pl->uniform_len == 27;
    P.rstart = static_cast<const int*>(pl->rstart);
    const int hz = pl->tz + 2 * pl->rz;
    for (int i = 0; i < 9; ++i) {
        if (pl->tap_dy[i] < -1 || pl->tap_dy[i] > 1 || pl->tap_dz[i] < -1 || pl->tap_dz[i] > 1) return TSGU_ERR_BAD_ARG;
        // the transposed product relies on the symmetry of the ascending tap list
        if (pl->tap_dy[i] != -pl->tap_dy[8 - i] || pl->tap_dz[i] != -pl->tap_dz[8 - i]) return TSGU_ERR_BAD_ARG;
        P.tap_row[i] = pl->tap_dy[i] * hz + pl->tap_dz[i];
    }
    P.kidx = static_cast<const unsigned char*>(pl->kidx);
    P.rcls = static_cast<const unsigned char*>(pl->rcls);
    P.nnz = nnz;
    const int rc = march_layout(P, mode, cl, pl->threads, pl->ntap);
    if (rc < 0) return rc;
    const int64_t nblocks = (int64_t)P.nb * P.nseg * P.tiles_y * P.tiles_z;
    if (nblocks > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    P.nblocks = nblocks;
    return TSGU_OK;
}

template <int MODE>
int dispatch(int cl, int threads, const MarchParams& P, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool uni = P.uniform > 0;                   // rows of one length (periodic lattices): no row-pointer reads
    if (!march_supported(MODE, P.mask, uni, threads)) return TSGU_ERR_BAD_ARG;
    if (P.mask == kBoxAll) return march_run_box(MODE, cl, threads, uni, P, s);
    const int rc = (P.mask & ~kMarchLowerIncl) == 0 ? march_run_lower(cl, P, s) : march_run_upper(cl, P, s);
    return rc == kMarchNotMine ? TSGU_ERR_BAD_ARG : rc;
}

}  // namespace

extern "C" {

int tsgu_march_supported(int mode, int mask, int uniform_len, int threads) {
    if (mode < 0 || mode > kLatSpmmT || mask <= 0 || mask > (int)kBoxAll) return 0;
    return march_supported(mode, (uint32_t)mask, uniform_len > 0, threads) ? 1 : 0;
}

int tsgu_march_lds_bytes(int mode, int vtype, int64_t p, int ty, int tz, int ry, int rz, int ncls, int threads) {
    if (vtype == TSGU_BF16) {   // whole-line march (linemarch_impl.h): the transposed product at 16 columns; tz = the lattice's nz
        if (mode < 0 || mode > kLatSpmmT || p != 16 || ry != 1 || rz != 1) return TSGU_ERR_BAD_DTYPE;
        LineParams L{};
        L.nz = tz, L.ty = ty, L.ny = ty;
        return linemarch_layout(L, threads, mode);
    }
    const int cl = march_lanes(vtype, p);
    if (cl == 0 || mode < 0 || mode > kLatSpmmT) return TSGU_ERR_BAD_DTYPE;
    if (threads != 256 && threads != 512) return TSGU_ERR_BAD_ARG;
    MarchParams P{};
    P.ty = ty, P.tz = tz, P.ry = ry, P.rz = rz, P.ncls = ncls;
    return march_layout(P, mode, cl, threads, 9);
}

int tsgu_csr_spmm_march(int vtype, const tsgu_march_plan* plan, int transposed, int64_t n_rows, int64_t nnz, const void* val,
                        const void* B, int64_t ldb, void* C, int64_t ldc, int64_t p, int device, void* stream) {
    if (vtype == TSGU_BF16) {   // whole-line march: periodic 27-point box, 16 columns, Aᵀ·G
        LineParams L{};
        const int lds = linemarch_fill(L, plan, transposed ? kLatSpmmT : kLatSpmm, p, n_rows, nnz);
        if (lds < 0) return lds;
        if (n_rows == 0) return TSGU_OK;
        if (!B || !C || !val || ldb < p || ldc < p || ldb % 8 || ldc % 8 || !aligned16(B) || !aligned16(C) || !aligned16(val)) return TSGU_ERR_BAD_ARG;
        if ((int64_t)plan->ny * plan->nz * ldb * 2 > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
        if (const int rc = set_device(device)) return rc;
        L.val = val, L.S = B, L.lds_ = ldb, L.out = C, L.ldo = ldc;
        return linemarch_run(transposed ? kLatSpmmT : kLatSpmm, L, plan->threads, static_cast<hipStream_t>(stream));
    }
    MarchParams P{};
    int cl = 0;
    const int mode = transposed ? kLatSpmmT : kLatSpmm;
    if (const int rc = fill(P, plan, mode, vtype, p, n_rows, nnz, cl)) return rc;
    if (n_rows == 0) return TSGU_OK;
    if (!B || !C || !val || ldb < p || ldc < p) return TSGU_ERR_BAD_ARG;
    if (ldb % 4 || ldc % 4 || !aligned16(B) || !aligned16(C)) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    const int64_t plane = (int64_t)plan->ny * plan->nz * 4;
    if (plane * ldb > 0x7fffffffLL || plane * ldc > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    P.val = val;
    P.S = B;
    P.lds_ = ldb;
    P.out = C;
    P.ldo = ldc;
    return transposed ? dispatch<kLatSpmmT>(cl, plan->threads, P, stream) : dispatch<kLatSpmm>(cl, plan->threads, P, stream);
}

int tsgu_csr_sddmm_march(int vtype, const tsgu_march_plan* plan, int64_t n_rows, int64_t nnz, const void* R, int64_t ldr, const void* Cm,
                         int64_t ldc, void* out_vals, double alpha, int accumulate, int64_t p, int device, void* stream) {
    if (vtype == TSGU_BF16) {   // whole-line march: periodic 27-point box, 16 columns
        if (accumulate) return TSGU_ERR_BAD_ARG;
        LineParams L{};
        const int lds = linemarch_fill(L, plan, kLatSddmm, p, n_rows, nnz);
        if (lds < 0) return lds;
        if (n_rows == 0) return TSGU_OK;
        if (!R || !Cm || !out_vals || ldr < p || ldc < p || ldr % 8 || ldc % 8 || !aligned16(R) || !aligned16(Cm) || !aligned16(out_vals)) return TSGU_ERR_BAD_ARG;
        if ((int64_t)plan->ny * plan->nz * ldr * 2 > 0x7fffffffLL || (int64_t)plan->ny * plan->nz * ldc * 2 > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
        if (const int rc = set_device(device)) return rc;
        L.Own = R, L.ldown = ldr, L.S = Cm, L.lds_ = ldc, L.gvals = out_vals, L.alpha = (float)alpha;
        return linemarch_run(kLatSddmm, L, plan->threads, static_cast<hipStream_t>(stream));
    }
    MarchParams P{};
    int cl = 0;
    if (const int rc = fill(P, plan, kLatSddmm, vtype, p, n_rows, nnz, cl)) return rc;
    if (n_rows == 0) return TSGU_OK;
    if (!R || !Cm || !out_vals || ldr < p || ldc < p) return TSGU_ERR_BAD_ARG;
    if (ldr % 4 || ldc % 4 || !aligned16(R) || !aligned16(Cm)) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    const int64_t plane = (int64_t)plan->ny * plan->nz * 4;
    if (plane * ldr > 0x7fffffffLL || plane * ldc > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    P.Own = R;
    P.ldown = ldr;
    P.S = Cm;
    P.lds_ = ldc;
    P.gvals = out_vals;
    P.alpha = (float)alpha;
    P.accumulate = accumulate != 0;
    return dispatch<kLatSddmm>(cl, plan->threads, P, stream);
}

}  // extern "C"
