// Fused sparse_mm backward: instantiations (fp32, bf16) and the extern "C" entry point.
#include "bwd_impl.h"

using namespace tsgu;

extern "C" {

int tsgu_csr_mm_backward(int vtype, int itype, int64_t n_rows, int64_t n_cols, int64_t nnz_per_item,
                         const void* t_ptr, const void* t_idx, const void* t_perm, const void* val,
                         const void* G, int64_t ldg, int64_t g_batch_stride,
                         const void* B, int64_t ldb, int64_t b_batch_stride,
                         void* gradA_vals, void* gradB, int64_t ldgb, int64_t gb_batch_stride,
                         int64_t p, int64_t batch, int device, void* stream) {
    if (n_rows < 0 || n_cols < 0 || nnz_per_item < 0 || p <= 0 || batch < 0) return TSGU_ERR_BAD_ARG;
    if (n_cols == 0 || batch == 0) return TSGU_OK;
    if (!t_ptr || !gradB || !B || (nnz_per_item > 0 && (!t_idx || !t_perm || !val || !G || !gradA_vals)))
        return TSGU_ERR_BAD_ARG;
    if (ldg < p || ldb < p || ldgb < p || n_rows > 0x7fffffffLL) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    BwdParams P{};
    P.n_rows_t = n_cols;
    P.nnz_per_item = nnz_per_item;
    P.p = p;
    P.tptr = t_ptr;
    P.tidx = t_idx;
    P.tperm = t_perm;
    P.val = val;
    P.G = G;
    P.ldg = ldg;
    P.g_bs = g_batch_stride;
    P.B = B;
    P.ldb = ldb;
    P.b_bs = b_batch_stride;
    P.gradA = gradA_vals;
    P.gradB = gradB;
    P.ldo = ldgb;
    P.o_bs = gb_batch_stride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vtype == TSGU_F32) {
        if (itype == TSGU_I32) return bwd_launch<float, int32_t>(P, batch, s);
        if (itype == TSGU_I64) return bwd_launch<float, int64_t>(P, batch, s);
    } else if (vtype == TSGU_BF16) {
        if (itype == TSGU_I32) return bwd_launch<bf16_t, int32_t>(P, batch, s);
        if (itype == TSGU_I64) return bwd_launch<bf16_t, int64_t>(P, batch, s);
    }
    return TSGU_ERR_BAD_DTYPE;
}

}  // extern "C"
