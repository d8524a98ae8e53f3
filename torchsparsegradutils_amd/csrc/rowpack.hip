// Row-pair gather kernels: instantiations (fp32) and the extern "C" entry points.
#include "rowpack_impl.h"

using namespace tsgu;

namespace {

int fill(RpParams& P, int64_t n_rows, int64_t n_src, int64_t nnz, int64_t p, const void* ptr, const void* uptr, const void* ucol,
         const void* upos, int ecap, int ucap, const void* sperm, const void* order, const void* vpair, const void* eptr,
         int64_t nblocks, const void* val) {
    if (n_rows < 0 || n_src < 0 || nnz < 0 || p <= 0) return TSGU_ERR_BAD_ARG;
    if (!ptr || !uptr || (nnz > 0 && (!ucol || !val)) || (nnz > 0 && sperm && !upos)) return TSGU_ERR_BAD_ARG;
    if (n_rows > 0x7fffffffLL || nnz > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    P.n_rows = n_rows;
    P.n_src = n_src;
    P.nnz = nnz;
    P.p = p;
    P.ptr = ptr;
    P.uptr = static_cast<const int*>(uptr);
    P.ucol = static_cast<const int*>(ucol);
    P.upos = static_cast<const uint32_t*>(upos);
    P.sperm = static_cast<const int*>(sperm);
    P.order = static_cast<const int*>(order);
    P.vpair = static_cast<const int*>(vpair);
    P.eptr = static_cast<const int*>(eptr);
    P.nblocks = nblocks;
    if ((vpair != nullptr) != (eptr != nullptr)) return TSGU_ERR_BAD_ARG;
    P.val = static_cast<const float*>(val);
    P.ecap = ecap;
    P.ucap = ucap;
    return TSGU_OK;
}

}  // namespace

extern "C" {

int tsgu_rowpack_limits(int vtype, int64_t p, int* rows_per_block, int* max_entries, int* max_union, int* lds_budget_bytes) {
    if (vtype != TSGU_F32 || (p != 16 && p != 32 && p != 64)) return TSGU_ERR_BAD_DTYPE;
    if (rows_per_block) *rows_per_block = 2 * (kBlock / (int)(p / 4));
    if (max_entries) *max_entries = kRpMaxQ * kBlock;
    if (max_union) *max_union = kRpMaxU * kBlock;
    if (lds_budget_bytes) *lds_budget_bytes = 64 * 1024;
    return TSGU_OK;
}

int tsgu_csr_spmm_rowpack(int vtype, int itype, int64_t n_rows, int64_t n_cols, int64_t nnz, const void* ptr,
                          const void* uptr, const void* ucol, const void* upos, int ecap, int ucap,
                          const void* sperm, const void* order, const void* vpair, const void* eptr, int64_t nblocks,
                          const void* val, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t p, int device, void* stream) {
    if (vtype != TSGU_F32) return TSGU_ERR_BAD_DTYPE;
    RpParams P{};
    if (const int rc = fill(P, n_rows, n_cols, nnz, p, ptr, uptr, ucol, upos, ecap, ucap, sperm, order, vpair, eptr, nblocks, val)) return rc;
    if (n_rows == 0) return TSGU_OK;
    if (!B || !C || ldb < p || ldc < p) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    P.S = static_cast<const float*>(B);
    P.lds_ = ldb;
    P.out = static_cast<float*>(C);
    P.ldo = ldc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (itype == TSGU_I32) return sperm ? rp_launch<int32_t, kRpSpmm, true>(P, s) : rp_launch<int32_t, kRpSpmm, false>(P, s);
    if (itype == TSGU_I64) return sperm ? rp_launch<int64_t, kRpSpmm, true>(P, s) : rp_launch<int64_t, kRpSpmm, false>(P, s);
    return TSGU_ERR_BAD_DTYPE;
}

int tsgu_csr_mm_backward_rowpack(int vtype, int itype, int64_t n_rows_t, int64_t n_cols_t, int64_t nnz, const void* t_ptr,
                                 const void* uptr, const void* ucol, const void* upos, int ecap, int ucap,
                                 const void* sperm, const void* order, const void* vpair, const void* eptr, int64_t nblocks,
                                 const void* val, const void* G, int64_t ldg, const void* B, int64_t ldb,
                                 void* gradA_vals, void* gradB, int64_t ldgb, int64_t p, int device, void* stream) {
    if (vtype != TSGU_F32) return TSGU_ERR_BAD_DTYPE;
    RpParams P{};
    if (const int rc = fill(P, n_rows_t, n_cols_t, nnz, p, t_ptr, uptr, ucol, upos, ecap, ucap, sperm, order, vpair, eptr, nblocks, val)) return rc;
    if (n_rows_t == 0) return TSGU_OK;
    if (!sperm || !B || !gradB || (nnz > 0 && (!G || !gradA_vals)) || ldg < p || ldb < p || ldgb < p) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    P.S = static_cast<const float*>(G);
    P.lds_ = ldg;
    P.Own = static_cast<const float*>(B);
    P.ldown = ldb;
    P.out = static_cast<float*>(gradB);
    P.ldo = ldgb;
    P.gradA = static_cast<float*>(gradA_vals);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (itype == TSGU_I32) return rp_launch<int32_t, kRpBwd, true>(P, s);
    if (itype == TSGU_I64) return rp_launch<int64_t, kRpBwd, true>(P, s);
    return TSGU_ERR_BAD_DTYPE;
}

int tsgu_csr_sddmm_rowpack(int vtype, int itype, int64_t n_rows, int64_t n_cols, int64_t nnz, const void* ptr,
                           const void* uptr, const void* ucol, int ecap, int ucap, const void* order,
                           const void* R, int64_t ldr, const void* Cm, int64_t ldc, void* out_vals, double alpha,
                           int64_t p, int device, void* stream) {
    if (vtype != TSGU_F32) return TSGU_ERR_BAD_DTYPE;
    RpParams P{};
    // the value array is not read: pass the output as a placeholder for the non-null check
    if (const int rc = fill(P, n_rows, n_cols, nnz, p, ptr, uptr, ucol, nullptr, ecap, ucap, nullptr, order, nullptr, nullptr, 0,
                            out_vals))
        return rc;
    if (n_rows == 0 || nnz == 0) return TSGU_OK;
    if (!R || !Cm || !out_vals || ldr < p || ldc < p) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    P.val = nullptr;
    P.Own = static_cast<const float*>(R);
    P.ldown = ldr;
    P.S = static_cast<const float*>(Cm);
    P.lds_ = ldc;
    P.gradA = static_cast<float*>(out_vals);
    P.alpha = (float)alpha;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (itype == TSGU_I32) return rp_launch<int32_t, kRpSddmm, false>(P, s);
    if (itype == TSGU_I64) return rp_launch<int64_t, kRpSddmm, false>(P, s);
    return TSGU_ERR_BAD_DTYPE;
}

}  // extern "C"
