// extern "C" entry points of the wave-pipelined LDS-tiled kernels (declared in include/tsgu_hip.h).
#include "wavetile_impl.h"

namespace tsgu {
int wavetile_dispatch_f32(int, int, const WtParams&, bool, int, hipStream_t);
int wavetile_dispatch_bf16(int, int, const WtParams&, bool, int, hipStream_t);

static int wt_go(int vtype, int itype, int mode, const WtParams& P, bool can_wide, int device, hipStream_t s) {
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return TSGU_ERR_RUNTIME;
    if (vtype == TSGU_F32) return wavetile_dispatch_f32(itype, mode, P, can_wide, n_cu, s);
    if (vtype == TSGU_BF16) return wavetile_dispatch_bf16(itype, mode, P, can_wide, n_cu, s);
    return TSGU_ERR_BAD_DTYPE;
}
static int wide_of(int vtype) { return vtype == TSGU_F32 ? 4 : vtype == TSGU_F64 ? 2 : 8; }
}  // namespace tsgu

using namespace tsgu;

extern "C" {

int tsgu_wavetile_geometry(int vtype, int64_t p, int* rows_per_task, int* max_distinct, int* max_entries) {
    if (p <= 0 || (vtype != TSGU_F32 && vtype != TSGU_BF16)) return TSGU_ERR_BAD_DTYPE;
    const int wide = wide_of(vtype);
    const RowGeom g = pick_geom(wide, p % wide == 0, p);
    if (g.col_tiles != 1) return TSGU_ERR_TOO_LARGE;
    const int ng = kWave / g.cl;
    if (rows_per_task) *rows_per_task = kWave / (g.cl * g.ep);
    if (max_distinct) *max_distinct = (kWtLT * ng) < 256 ? (kWtLT * ng) : 256;
    if (max_entries) *max_entries = kWtEnt * kWave;
    return TSGU_OK;
}

int tsgu_csr_spmm_wavetile(int vtype, int itype, int64_t n_rows, int64_t n_cols, int64_t nnz,
                           const void* crow, const void* val, const void* perm,
                           const void* tmeta, const void* tcols, const void* lidx,
                           const void* B, int64_t ldb, void* C, int64_t ldc, int64_t p,
                           int device, void* stream) {
    if (n_rows <= 0 || n_cols <= 0 || nnz <= 0 || p <= 0) return TSGU_ERR_BAD_ARG;
    if (!crow || !tmeta || !C || !B || !val || !tcols || !lidx) return TSGU_ERR_BAD_ARG;
    if (ldb < p || ldc < p || n_cols > 0x7fffffffLL) return TSGU_ERR_BAD_ARG;
    const int wide = wide_of(vtype);
    const bool wide_p = p % wide == 0;
    const bool can = wide_p && ldb % wide == 0 && ldc % wide == 0 && aligned16(B) && aligned16(C);
    if (wide_p && !can) return TSGU_ERR_BAD_ARG;  // the plan was built for the 16-byte geometry
    if (const int rc = set_device(device)) return rc;
    WtParams P{};
    P.n_rows = n_rows;
    P.nnz = nnz;
    P.p = p;
    P.crow = crow;
    P.val = val;
    P.perm = perm;
    P.tmeta = static_cast<const int2*>(tmeta);
    P.tcols = static_cast<const int*>(tcols);
    P.lidx = static_cast<const unsigned char*>(lidx);
    P.X = B;
    P.ldx = ldb;
    P.out = C;
    P.ldo = ldc;
    return wt_go(vtype, itype, perm ? kWtSpmmPerm : kWtSpmm, P, can, device, static_cast<hipStream_t>(stream));
}

int tsgu_csr_sddmm_wavetile(int vtype, int itype, int64_t n_rows, int64_t n_cols, int64_t nnz,
                            const void* crow, const void* tmeta, const void* tcols, const void* lidx,
                            const void* G, int64_t ldg, const void* B, int64_t ldb,
                            void* out, double alpha, int swap_roles, int64_t p,
                            int device, void* stream) {
    if (n_rows <= 0 || n_cols <= 0 || nnz <= 0 || p <= 0) return TSGU_ERR_BAD_ARG;
    if (!crow || !tmeta || !tcols || !lidx || !out || !G || !B) return TSGU_ERR_BAD_ARG;
    if (ldg < p || ldb < p || n_cols > 0x7fffffffLL) return TSGU_ERR_BAD_ARG;
    const int wide = wide_of(vtype);
    const bool wide_p = p % wide == 0;
    const bool can = wide_p && ldb % wide == 0 && ldg % wide == 0 && aligned16(B) && aligned16(G);
    if (wide_p && !can) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    WtParams P{};
    P.n_rows = n_rows;
    P.nnz = nnz;
    P.p = p;
    P.crow = crow;
    P.tmeta = static_cast<const int2*>(tmeta);
    P.tcols = static_cast<const int*>(tcols);
    P.lidx = static_cast<const unsigned char*>(lidx);
    if (!swap_roles) {
        P.R = G;
        P.ldr = ldg;
        P.X = B;
        P.ldx = ldb;
    } else {
        P.R = B;
        P.ldr = ldb;
        P.X = G;
        P.ldx = ldg;
    }
    P.out = out;
    P.alpha = alpha;
    return wt_go(vtype, itype, kWtSddmm, P, can, device, static_cast<hipStream_t>(stream));
}

int tsgu_csr_mm_backward_wavetile(int vtype, int itype, int64_t n_rows, int64_t n_cols, int64_t nnz,
                                  const void* t_ptr, const void* t_perm, const void* val,
                                  const void* tmeta, const void* tcols, const void* lidx,
                                  const void* G, int64_t ldg, const void* B, int64_t ldb,
                                  void* gradA_vals, void* gradB, int64_t ldgb, int64_t p,
                                  int device, void* stream) {
    if (n_rows <= 0 || n_cols <= 0 || nnz <= 0 || p <= 0) return TSGU_ERR_BAD_ARG;
    if (!t_ptr || !t_perm || !val || !tmeta || !tcols || !lidx || !G || !B || !gradA_vals || !gradB) return TSGU_ERR_BAD_ARG;
    if (ldg < p || ldb < p || ldgb < p || n_rows > 0x7fffffffLL) return TSGU_ERR_BAD_ARG;
    const int wide = wide_of(vtype);
    const bool wide_p = p % wide == 0;
    const bool can = wide_p && ldb % wide == 0 && ldg % wide == 0 && ldgb % wide == 0 && aligned16(B) && aligned16(G) &&
                     aligned16(gradB);
    if (wide_p && !can) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    WtParams P{};
    P.n_rows = n_cols;  // rows of Aᵀ
    P.nnz = nnz;
    P.p = p;
    P.crow = t_ptr;
    P.val = val;
    P.perm = t_perm;
    P.tmeta = static_cast<const int2*>(tmeta);
    P.tcols = static_cast<const int*>(tcols);
    P.lidx = static_cast<const unsigned char*>(lidx);
    P.X = G;
    P.ldx = ldg;
    P.R = B;
    P.ldr = ldb;
    P.out = gradB;
    P.ldo = ldgb;
    P.out2 = gradA_vals;
    return wt_go(vtype, itype, kWtBwd, P, can, device, static_cast<hipStream_t>(stream));
}

}  // extern "C"
