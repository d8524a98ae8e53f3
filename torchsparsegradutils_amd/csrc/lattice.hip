// Lattice plane-sweep kernels: extern "C" entry points (declared in include/tsgu_hip.h).
#include "lattice_impl.h"

namespace tsgu {
int lat_dispatch_f32(int mode, int cl, int threads, const LatParams& P, hipStream_t s);
int lat_dispatch_bf16(int mode, int cl, int threads, const LatParams& P, hipStream_t s);
int lat_dispatch_f64(int mode, int cl, int threads, const LatParams& P, hipStream_t s);
}  // namespace tsgu

using namespace tsgu;

namespace {

int vbytes_of(int vtype) { return vtype == TSGU_F32 ? 4 : (vtype == TSGU_BF16 ? 2 : (vtype == TSGU_F64 ? 8 : 0)); }

// column lanes of 16 bytes per dense row (one lane: fp32 SpMM only)
int lanes_of(int vtype, int64_t p, int mode) {
    const int vb = vbytes_of(vtype);
    if (vb == 0 || p <= 0 || (p * vb) % 16) return 0;
    const int64_t cl = p * vb / 16;
    if (cl == 1) return vtype == TSGU_F32 && mode == kLatSpmm ? 1 : 0;
    return (cl == 2 || cl == 4 || cl == 8 || cl == 16) ? (int)cl : 0;
}

int fill(LatParams& P, const tsgu_lattice_plan* pl, int mode, int vtype, int64_t p, int64_t n_rows, int64_t nnz, int& cl) {
    if (!pl || n_rows < 0 || nnz < 0) return TSGU_ERR_BAD_ARG;
    cl = lanes_of(vtype, p, mode);
    if (cl == 0) return TSGU_ERR_BAD_DTYPE;
    if ((pl->kind != 0) != (mode == kLatSpmmT)) return TSGU_ERR_BAD_ARG;
    if (pl->nb <= 0 || pl->nx <= 0 || pl->ny <= 0 || pl->nz <= 0 || pl->nseg <= 0 || pl->nseg > pl->nx) return TSGU_ERR_BAD_ARG;
    if ((int64_t)pl->nb * pl->nx * pl->ny * pl->nz != n_rows) return TSGU_ERR_BAD_ARG;
    if (pl->ry >= pl->ny + (pl->ny == 1) || pl->rz >= pl->nz + (pl->nz == 1)) return TSGU_ERR_BAD_ARG;
    if (pl->threads != 256 && pl->threads != 512 && pl->threads != 1024) return TSGU_ERR_BAD_ARG;
    if (!pl->rec || !pl->lens || !pl->rcls || !pl->wlist || (pl->uniform_len <= 0 && !pl->rstart)) return TSGU_ERR_BAD_ARG;
    if (nnz > 0x7fffffffLL || n_rows > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    // the kernels address a plane / the value array with 32-bit byte offsets from wave-uniform 64-bit bases
    if (nnz * vbytes_of(vtype) + 16 > 0xffffffffLL) return TSGU_ERR_TOO_LARGE;
    if (pl->uniform_len > 0 && (int64_t)pl->uniform_len * n_rows != nnz) return TSGU_ERR_BAD_ARG;
    P.nb = pl->nb, P.nx = pl->nx, P.ny = pl->ny, P.nz = pl->nz;
    P.ty = pl->ty, P.tz = pl->tz, P.ry = pl->ry, P.rz = pl->rz;
    P.ring = pl->ring;
    P.cpl = pl->chunks_per_lane == 0 ? 1 : pl->chunks_per_lane;
    if (P.cpl == 2 && (cl < 8 || pl->recw != 28)) return TSGU_ERR_BAD_ARG;
    P.tiles_y = (pl->ny + pl->ty - 1) / pl->ty;
    P.tiles_z = (pl->nz + pl->tz - 1) / pl->tz;
    P.nseg = pl->nseg;
    P.seg_len = (pl->nx + pl->nseg - 1) / pl->nseg;
    if ((int64_t)(P.nseg - 1) * P.seg_len >= pl->nx) return TSGU_ERR_BAD_ARG;   // every segment must own at least one plane
    P.ncls = pl->ncls, P.nloc = pl->nloc, P.recw = pl->recw, P.uniform_len = pl->uniform_len;
    P.wlist = static_cast<const unsigned char*>(pl->wlist);
    P.rec = pl->rec;
    P.lens = static_cast<const unsigned char*>(pl->lens);
    P.rcls = static_cast<const unsigned char*>(pl->rcls);
    P.rstart = static_cast<const int*>(pl->rstart);
    P.nnz = nnz;
    const int rc = lat_layout(P, mode, cl, vbytes_of(vtype), pl->threads);
    if (rc < 0) return rc;
    const int64_t nblocks = (int64_t)P.nb * P.nseg * P.tiles_y * P.tiles_z;
    if (nblocks > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    P.nblocks = nblocks;
    return TSGU_OK;
}

int dispatch(int vtype, int mode, int cl, int threads, const LatParams& P, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vtype == TSGU_F32) return lat_dispatch_f32(mode, cl, threads, P, s);
    if (vtype == TSGU_BF16) return lat_dispatch_bf16(mode, cl, threads, P, s);
    if (vtype == TSGU_F64) return lat_dispatch_f64(mode, cl, threads, P, s);
    return TSGU_ERR_BAD_DTYPE;
}

}  // namespace

extern "C" {

int tsgu_lattice_lds_bytes(int mode, int vtype, int64_t p, int ty, int tz, int ry, int rz, int nloc, int recw, int threads, int ring,
                           int chunks_per_lane) {
    const int cl = lanes_of(vtype, p, mode);
    if (cl == 0 || mode < 0 || mode > 2) return TSGU_ERR_BAD_DTYPE;
    if (threads != 256 && threads != 512 && threads != 1024) return TSGU_ERR_BAD_ARG;
    LatParams P{};
    P.ty = ty, P.tz = tz, P.ry = ry, P.rz = rz, P.ncls = nloc, P.nloc = nloc, P.recw = recw, P.ring = ring;
    P.cpl = chunks_per_lane == 0 ? 1 : chunks_per_lane;
    return lat_layout(P, mode, cl, vbytes_of(vtype), threads);
}

static int spmm_lattice(int vtype, const tsgu_lattice_plan* plan, int64_t n_rows, int64_t nnz, const void* val, const void* B, int64_t ldb,
                        void* C, int64_t ldc, int64_t p, void* dot_partial, int64_t dot_rows, const int* skip, const void* dot_w, int device,
                        void* stream) {
    LatParams P{};
    int cl = 0;
    const int mode = plan && plan->kind != 0 ? kLatSpmmT : kLatSpmm;
    if (const int rc = fill(P, plan, mode, vtype, p, n_rows, nnz, cl)) return rc;
    if (dot_partial && (mode != kLatSpmm || (vtype != TSGU_F32 && vtype != TSGU_F64) || P.cpl != 1 || dot_rows != P.nblocks)) return TSGU_ERR_BAD_ARG;
    if (n_rows == 0) return TSGU_OK;
    if (!B || !C || (nnz > 0 && !val) || ldb < p || ldc < p) return TSGU_ERR_BAD_ARG;
    const int vec = 16 / vbytes_of(vtype);
    if (ldb % vec || ldc % vec || !aligned16(B) || !aligned16(C)) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    const int64_t plane = (int64_t)plan->ny * plan->nz * vbytes_of(vtype);
    if (plane * ldb > 0x7fffffffLL || plane * ldc > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    P.val = val;
    P.S = B;
    P.lds_ = ldb;
    P.out = C;
    P.ldo = ldc;
    P.dot_partial = dot_partial;
    P.skip = skip;
    P.dot_w = dot_w;
    if (dot_w && (!dot_partial || !aligned16(dot_w))) return TSGU_ERR_BAD_ARG;
    return dispatch(vtype, mode, cl, plan->threads, P, stream);
}

int tsgu_csr_spmm_lattice(int vtype, const tsgu_lattice_plan* plan, int64_t n_rows, int64_t nnz, const void* val, const void* B,
                          int64_t ldb, void* C, int64_t ldc, int64_t p, int device, void* stream) {
    return spmm_lattice(vtype, plan, n_rows, nnz, val, B, ldb, C, ldc, p, nullptr, 0, nullptr, nullptr, device, stream);
}

int tsgu_csr_spmm_lattice_dot(int vtype, const tsgu_lattice_plan* plan, int64_t n_rows, int64_t nnz, const void* val, const void* B,
                              int64_t ldb, void* C, int64_t ldc, int64_t p, void* dot_partial, int64_t dot_rows, const int* skip,
                              const void* dot_w, int device, void* stream) {
    if (!dot_partial) return TSGU_ERR_BAD_ARG;
    return spmm_lattice(vtype, plan, n_rows, nnz, val, B, ldb, C, ldc, p, dot_partial, dot_rows, skip, dot_w, device, stream);
}

int tsgu_csr_sddmm_lattice(int vtype, const tsgu_lattice_plan* plan, int64_t n_rows, int64_t nnz, const void* R, int64_t ldr,
                           const void* Cm, int64_t ldc, void* out_vals, double alpha, int64_t p, int device, void* stream) {
    LatParams P{};
    int cl = 0;
    if (const int rc = fill(P, plan, kLatSddmm, vtype, p, n_rows, nnz, cl)) return rc;
    if (n_rows == 0 || nnz == 0) return TSGU_OK;
    if (!R || !Cm || !out_vals || ldr < p || ldc < p) return TSGU_ERR_BAD_ARG;
    const int vec = 16 / vbytes_of(vtype);
    if (ldr % vec || ldc % vec || !aligned16(R) || !aligned16(Cm)) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    const int64_t plane = (int64_t)plan->ny * plan->nz * vbytes_of(vtype);
    if (plane * ldr > 0x7fffffffLL || plane * ldc > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    P.Own = R;
    P.ldown = ldr;
    P.S = Cm;
    P.lds_ = ldc;
    P.gvals = out_vals;
    P.alpha = (float)alpha;
    return dispatch(vtype, kLatSddmm, cl, plan->threads, P, stream);
}

}  // extern "C"
