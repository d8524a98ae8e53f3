// Workgroup-tiled SpMM / fused backward: instantiations (fp32) and the extern "C" entry points.
#include "blocktile_impl.h"

using namespace tsgu;

namespace {

int fill(BtParams& P, int64_t n_rows, int64_t nnz, int64_t p, const void* ptr, const void* ndist, const void* trow,
         int capd, int ecap, const void* ent, const void* sperm, const void* val) {
    if (n_rows < 0 || nnz < 0 || p <= 0 || capd <= 0 || ecap <= 0) return TSGU_ERR_BAD_ARG;
    if (!ptr || !ndist || !trow || (nnz > 0 && (!ent || !val))) return TSGU_ERR_BAD_ARG;
    if (n_rows > 0x7fffffffLL || nnz > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    P.n_rows = n_rows;
    P.nnz = nnz;
    P.p = p;
    P.ptr = ptr;
    P.ndist = static_cast<const int*>(ndist);
    P.trow = static_cast<const int*>(trow);
    P.ent = static_cast<const uint32_t*>(ent);
    P.sperm = static_cast<const int*>(sperm);
    P.val = static_cast<const float*>(val);
    P.capd = capd;
    P.ecap = ecap;
    return TSGU_OK;
}

}  // namespace

extern "C" {

int tsgu_blocktile_limits(int vtype, int64_t p, int tile, int* rows_per_block, int* distinct_multiple, int* max_distinct,
                          int* max_entries, int* lds_budget_bytes) {
    if (vtype != TSGU_F32 || (p != 16 && p != 32 && p != 64)) return TSGU_ERR_BAD_DTYPE;
    const int cl = (int)(p / 4), ri = kWave / cl;
    if (rows_per_block) *rows_per_block = kBlock / cl;  // one row per CL-lane group, EP = 1
    if (distinct_multiple) *distinct_multiple = tile ? ri : 4;
    if (max_distinct) *max_distinct = tile ? kBtMaxT * 4 * ri : kBtMaxD * kBlock;
    if (max_entries) *max_entries = kBtMaxQ * kBlock;
    if (lds_budget_bytes) *lds_budget_bytes = 64 * 1024;
    return TSGU_OK;
}

int tsgu_csr_spmm_blocktile(int vtype, int itype, int64_t n_rows, int64_t nnz, const void* ptr,
                            const void* ndist, const void* trow, int capd, int ecap, int rpb, int tile,
                            const void* ent, const void* sperm, const void* val,
                            const void* B, int64_t ldb, void* C, int64_t ldc, int64_t p,
                            int device, void* stream) {
    if (vtype != TSGU_F32) return TSGU_ERR_BAD_DTYPE;
    BtParams P{};
    if (const int rc = fill(P, n_rows, nnz, p, ptr, ndist, trow, capd, ecap, ent, sperm, val)) return rc;
    if (n_rows == 0) return TSGU_OK;
    if (!B || !C || ldb < p || ldc < p) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    P.S = static_cast<const float*>(B);
    P.lds_ = ldb;
    P.out = static_cast<float*>(C);
    P.ldo = ldc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (itype == TSGU_I32)
        return sperm ? bt_launch<int32_t, kBtSpmm, true>(P, rpb, tile, s) : bt_launch<int32_t, kBtSpmm, false>(P, rpb, tile, s);
    if (itype == TSGU_I64)
        return sperm ? bt_launch<int64_t, kBtSpmm, true>(P, rpb, tile, s) : bt_launch<int64_t, kBtSpmm, false>(P, rpb, tile, s);
    return TSGU_ERR_BAD_DTYPE;
}

int tsgu_csr_mm_backward_blocktile(int vtype, int itype, int64_t n_rows_t, int64_t nnz, const void* t_ptr,
                                   const void* ndist, const void* trow, int capd, int ecap, int rpb, int tile,
                                   const void* ent, const void* sperm, const void* val,
                                   const void* G, int64_t ldg, const void* B, int64_t ldb,
                                   void* gradA_vals, void* gradB, int64_t ldgb, int64_t p,
                                   int device, void* stream) {
    if (vtype != TSGU_F32) return TSGU_ERR_BAD_DTYPE;
    BtParams P{};
    if (const int rc = fill(P, n_rows_t, nnz, p, t_ptr, ndist, trow, capd, ecap, ent, sperm, val)) return rc;
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
    if (itype == TSGU_I32) return bt_launch<int32_t, kBtBwd, true>(P, rpb, tile, s);
    if (itype == TSGU_I64) return bt_launch<int64_t, kBtBwd, true>(P, rpb, tile, s);
    return TSGU_ERR_BAD_DTYPE;
}

}  // extern "C"
