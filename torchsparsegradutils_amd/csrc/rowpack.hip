// Row-pair gather kernels: extern "C" entry points (declared in include/tsgu_hip.h).
#include "rowpack_impl.h"

namespace tsgu {
int rp_dispatch_f32(int mode, bool perm, int itype, const RpParams& P, hipStream_t s);
int rp_dispatch_bf16(int mode, bool perm, int itype, const RpParams& P, hipStream_t s);
}  // namespace tsgu

using namespace tsgu;

namespace {

int fill(RpParams& P, int64_t n_rows, int64_t n_src, int64_t nnz, int64_t p, const void* ptr, const tsgu_rowpack_plan* pl) {
    if (!pl || n_rows < 0 || n_src < 0 || nnz < 0 || p <= 0) return TSGU_ERR_BAD_ARG;
    if (!ptr || !pl->uptr || (nnz > 0 && !pl->ucol) || (nnz > 0 && pl->sperm && !pl->upos)) return TSGU_ERR_BAD_ARG;
    if (n_rows > 0x7fffffffLL || nnz > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    if ((pl->nclasses > 0) != (pl->wcls != nullptr)) return TSGU_ERR_BAD_ARG;
    P.n_rows = n_rows;
    P.n_src = n_src;
    P.nnz = nnz;
    P.p = p;
    P.ptr = ptr;
    P.uptr = static_cast<const int*>(pl->uptr);
    P.ucol = static_cast<const int*>(pl->ucol);
    P.upos = static_cast<const uint32_t*>(pl->upos);
    P.sperm = static_cast<const int*>(pl->sperm);
    P.order = static_cast<const int*>(pl->order);
    P.vpair = static_cast<const int*>(pl->vpair);
    P.eptr = static_cast<const int*>(pl->eptr);
    P.wcls = static_cast<const int*>(pl->wcls);
    P.wbase = static_cast<const int*>(pl->wbase);
    P.cne = static_cast<const int*>(pl->cne);
    P.nblocks = pl->nblocks;
    P.ecap = pl->ecap;
    P.ucap = pl->ucap;
    P.rgroup = pl->rows_per_group;
    P.srcstart = static_cast<const int*>(pl->srcstart);
    if (pl->srcstart && (!pl->wcls || !pl->sperm)) return TSGU_ERR_BAD_ARG;
    return TSGU_OK;
}

int dispatch(int vtype, int mode, bool perm, int itype, const RpParams& P, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vtype == TSGU_F32) return rp_dispatch_f32(mode, perm, itype, P, s);
    if (vtype == TSGU_BF16) return rp_dispatch_bf16(mode, perm, itype, P, s);
    return TSGU_ERR_BAD_DTYPE;
}

}  // namespace

extern "C" {

int tsgu_rowpack_geometry(int vtype, int64_t p, int* rows_per_block, int* entry_lanes, int* max_entries, int* max_union,
                          int* lds_budget_bytes) {
    int cl = 0, ep = 0;
    if (vtype == TSGU_F32) {
        if (!rp_geom<float>(p, cl, ep)) return TSGU_ERR_BAD_DTYPE;
    } else if (vtype == TSGU_BF16) {
        if (!rp_geom<bf16_t>(p, cl, ep)) return TSGU_ERR_BAD_DTYPE;
    } else {
        return TSGU_ERR_BAD_DTYPE;
    }
    if (rows_per_block) *rows_per_block = 2 * (kBlock / (cl * ep));
    if (entry_lanes) *entry_lanes = ep;
    if (max_entries) *max_entries = kRpMaxQ * kBlock;
    if (max_union) *max_union = kRpMaxU * kBlock;
    if (lds_budget_bytes) *lds_budget_bytes = 64 * 1024;
    return TSGU_OK;
}

int tsgu_csr_spmm_rowpack(int vtype, int itype, int64_t n_rows, int64_t n_cols, int64_t nnz, const void* ptr,
                          const tsgu_rowpack_plan* plan, const void* val, const void* B, int64_t ldb, void* C, int64_t ldc,
                          int64_t p, int device, void* stream) {
    RpParams P{};
    if (const int rc = fill(P, n_rows, n_cols, nnz, p, ptr, plan)) return rc;
    if (n_rows == 0) return TSGU_OK;
    if (!B || !C || (nnz > 0 && !val) || ldb < p || ldc < p) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    P.val = val;
    P.S = B;
    P.lds_ = ldb;
    P.out = C;
    P.ldo = ldc;
    return dispatch(vtype, kRpSpmm, plan->sperm != nullptr, itype, P, stream);
}

int tsgu_csr_mm_backward_rowpack(int vtype, int itype, int64_t n_rows_t, int64_t n_cols_t, int64_t nnz, const void* t_ptr,
                                 const tsgu_rowpack_plan* plan, const void* val, const void* G, int64_t ldg, const void* B,
                                 int64_t ldb, void* gradA_vals, void* gradB, int64_t ldgb, int64_t p, int device, void* stream) {
    RpParams P{};
    if (const int rc = fill(P, n_rows_t, n_cols_t, nnz, p, t_ptr, plan)) return rc;
    if (n_rows_t == 0) return TSGU_OK;
    if (!plan->sperm || !B || !gradB || (nnz > 0 && (!G || !gradA_vals || !val)) || ldg < p || ldb < p || ldgb < p)
        return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    P.val = val;
    P.S = G;
    P.lds_ = ldg;
    P.Own = B;
    P.ldown = ldb;
    P.out = gradB;
    P.ldo = ldgb;
    P.gradA = gradA_vals;
    return dispatch(vtype, kRpBwd, true, itype, P, stream);
}

int tsgu_csr_sddmm_rowpack(int vtype, int itype, int64_t n_rows, int64_t n_cols, int64_t nnz, const void* ptr,
                           const tsgu_rowpack_plan* plan, const void* R, int64_t ldr, const void* Cm, int64_t ldc,
                           void* out_vals, double alpha, int64_t p, int device, void* stream) {
    RpParams P{};
    if (const int rc = fill(P, n_rows, n_cols, nnz, p, ptr, plan)) return rc;
    if (n_rows == 0 || nnz == 0) return TSGU_OK;
    if (plan->sperm || plan->upos || !R || !Cm || !out_vals || ldr < p || ldc < p) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    P.Own = R;
    P.ldown = ldr;
    P.S = Cm;
    P.lds_ = ldc;
    P.gradA = out_vals;
    P.alpha = (float)alpha;
    return dispatch(vtype, kRpSddmm, false, itype, P, stream);
}

}  // extern "C"
