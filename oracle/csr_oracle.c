/*
 * csr_oracle.c — TEST INFRASTRUCTURE ONLY.  Plain-C, single-threaded restatement of the
 * arithmetic on the reference's hot path, used as the checker for the HIP kernels.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; the product package (torchsparsegradutils_amd/) never does.
 *
 * The reference (cai4cai/torchsparsegradutils) is pure Python; its arithmetic is executed by
 * PyTorch ATen ops (third-party, pinned only as torch>=2.5 in the reference's pyproject.toml:23;
 * CPU backends: MKL sparse addmm / MKL sparse trsm).  Each function restates the mathematical
 * definition of the ATen call at the cited reference line; the restatement is pinned against
 * outputs of the real reference (tests/golden/make_golden.py → tests/golden/ *.npz).
 *
 * All index arrays are int64, dense operands row-major with leading dimensions in elements.
 * Arithmetic type == storage type (float or double), products accumulated in entry order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)

#define DEFINE_ALL(T, SFX)                                                                        \
    /* C = A·B  — torch.sparse.mm(A, B), reference sparse_matmul.py:155 */                     \
    void CAT(oracle_csr_spmm, SFX)(int64_t n_rows, const int64_t* crow, const int64_t* col,       \
                                   const T* val, const T* B, int64_t ldb, T* C, int64_t ldc,      \
                                   int64_t p) {                                                   \
        for (int64_t i = 0; i < n_rows; ++i) {                                                    \
            T* c = C + i * ldc;                                                                   \
            for (int64_t q = 0; q < p; ++q) c[q] = (T)0;                                          \
            for (int64_t k = crow[i]; k < crow[i + 1]; ++k) {                                     \
                const T a = val[k];                                                               \
                const T* b = B + col[k] * ldb;                                                    \
                for (int64_t q = 0; q < p; ++q) c[q] += a * b[q];                                 \
            }                                                                                     \
        }                                                                                         \
    }                                                                                             \
    /* D = Aᵀ·G — torch.sparse.mm(A.t(), grad), reference sparse_matmul.py:229 */               \
    void CAT(oracle_csr_spmm_t, SFX)(int64_t n_rows, int64_t n_cols, const int64_t* crow,         \
                                     const int64_t* col, const T* val, const T* G, int64_t ldg,   \
                                     T* D, int64_t ldd, int64_t p) {                              \
        for (int64_t j = 0; j < n_cols; ++j)                                                      \
            for (int64_t q = 0; q < p; ++q) D[j * ldd + q] = (T)0;                                \
        for (int64_t i = 0; i < n_rows; ++i) {                                                    \
            const T* g = G + i * ldg;                                                             \
            for (int64_t k = crow[i]; k < crow[i + 1]; ++k) {                                     \
                const T a = val[k];                                                               \
                T* d = D + col[k] * ldd;                                                          \
                for (int64_t q = 0; q < p; ++q) d[q] += a * g[q];                                 \
            }                                                                                     \
        }                                                                                         \
    }                                                                                             \
    /* out[k] = alpha·<R[row k,:], Cm[col k,:]> — index_select ×2, mul, sum(dim=1);             \
       reference sparse_matmul.py:186-205 (alpha=+1), sparse_solve.py:216-235 and :487-504       \
       (alpha=-1; the transpose role swap of sparse_solve.py:223-225 = caller swaps R and Cm) */ \
    void CAT(oracle_csr_sddmm, SFX)(int64_t n_rows, const int64_t* crow, const int64_t* col,      \
                                    const T* R, int64_t ldr, const T* Cm, int64_t ldc, T alpha,   \
                                    T* out, int64_t p) {                                          \
        for (int64_t i = 0; i < n_rows; ++i) {                                                    \
            const T* r = R + i * ldr;                                                             \
            for (int64_t k = crow[i]; k < crow[i + 1]; ++k) {                                     \
                const T* c = Cm + col[k] * ldc;                                                   \
                T s = (T)0;                                                                       \
                for (int64_t q = 0; q < p; ++q) s += r[q] * c[q];                                 \
                out[k] = alpha * s;                                                               \
            }                                                                                     \
        }                                                                                         \
    }                                                                                             \
    /* COO flavour (explicit rows) — reference sparse_matmul.py:185,201-205 */                   \
    void CAT(oracle_coo_sddmm, SFX)(int64_t nnz, const int64_t* row, const int64_t* col,          \
                                    const T* R, int64_t ldr, const T* Cm, int64_t ldc, T alpha,   \
                                    T* out, int64_t p) {                                          \
        for (int64_t k = 0; k < nnz; ++k) {                                                       \
            const T* r = R + row[k] * ldr;                                                        \
            const T* c = Cm + col[k] * ldc;                                                       \
            T s = (T)0;                                                                           \
            for (int64_t q = 0; q < p; ++q) s += r[q] * c[q];                                     \
            out[k] = alpha * s;                                                                   \
        }                                                                                         \
    }                                                                                             \
    /* X = op(A)^{-1} B — torch.triangular_solve(B, A, upper, transpose, unitriangular),         \
       reference _compat.py:42-48.  Semantics observed on the reference's CPU backend: entries   \
       on the other side of the diagonal are ignored; with unitriangular stored diagonals are    \
       ignored; transpose solves with the transpose of the selected triangle.  Returns 0.        \
       Substitution in plain row (or, for transpose, column-push) order. */                      \
    int CAT(oracle_csr_sptrsm, SFX)(int64_t n, const int64_t* crow, const int64_t* col,           \
                                    const T* val, int upper, int unit, int transpose, const T* B, \
                                    int64_t ldb, T* X, int64_t ldx, int64_t p) {                  \
        for (int64_t i = 0; i < n; ++i)                                                           \
            for (int64_t q = 0; q < p; ++q) X[i * ldx + q] = B[i * ldb + q];                      \
        T* diag = (T*)malloc((size_t)(n > 0 ? n : 1) * sizeof(T));                                \
        if (!diag) return -1;                                                                     \
        for (int64_t i = 0; i < n; ++i) {                                                         \
            T d = (T)0;                                                                           \
            for (int64_t k = crow[i]; k < crow[i + 1]; ++k)                                       \
                if (col[k] == i) d += val[k];                                                     \
            diag[i] = unit ? (T)1 : d;                                                            \
        }                                                                                         \
        if (!transpose) {                                                                         \
            /* row sweep: forward for lower, backward for upper */                                \
            for (int64_t s = 0; s < n; ++s) {                                                     \
                const int64_t i = upper ? n - 1 - s : s;                                          \
                T* x = X + i * ldx;                                                               \
                for (int64_t k = crow[i]; k < crow[i + 1]; ++k) {                                 \
                    const int64_t j = col[k];                                                     \
                    if (upper ? j > i : j < i) {                                                  \
                        const T a = val[k];                                                       \
                        const T* xj = X + j * ldx;                                                \
                        for (int64_t q = 0; q < p; ++q) x[q] -= a * xj[q];                        \
                    }                                                                             \
                }                                                                                 \
                if (!unit)                                                                        \
                    for (int64_t q = 0; q < p; ++q) x[q] /= diag[i];                              \
            }                                                                                     \
        } else {                                                                                  \
            /* Aᵀ x = b: Aᵀ of a lower factor is upper → backward; column-push form */           \
            for (int64_t s = 0; s < n; ++s) {                                                     \
                const int64_t i = upper ? s : n - 1 - s;                                          \
                T* x = X + i * ldx;                                                               \
                if (!unit)                                                                        \
                    for (int64_t q = 0; q < p; ++q) x[q] /= diag[i];                              \
                for (int64_t k = crow[i]; k < crow[i + 1]; ++k) {                                 \
                    const int64_t j = col[k];                                                     \
                    if (upper ? j > i : j < i) {                                                  \
                        const T a = val[k];                                                       \
                        T* xj = X + j * ldx;                                                      \
                        for (int64_t q = 0; q < p; ++q) xj[q] -= a * x[q];                        \
                    }                                                                             \
                }                                                                                 \
            }                                                                                     \
        }                                                                                         \
        free(diag);                                                                               \
        return 0;                                                                                 \
    }

DEFINE_ALL(float, f32)
DEFINE_ALL(double, f64)

/* Number of dependency levels of a lower (upper=0) or upper (upper=1) triangular CSR pattern:
   level(i) = 1 + max level of the rows it depends on.  Used to characterise K4 test/bench matrices
   (a level-scheduled solver would need this many grid barriers). */
int64_t oracle_csr_levels(int64_t n, const int64_t* crow, const int64_t* col, int upper) {
    int64_t* lvl = (int64_t*)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
    int64_t top = 0;
    if (!lvl) return -1;
    for (int64_t s = 0; s < n; ++s) {
        const int64_t i = upper ? n - 1 - s : s;
        int64_t l = 0;
        for (int64_t k = crow[i]; k < crow[i + 1]; ++k) {
            const int64_t j = col[k];
            if ((upper ? j > i : j < i) && lvl[j] > l) l = lvl[j];
        }
        lvl[i] = l + 1;
        if (lvl[i] > top) top = lvl[i];
    }
    free(lvl);
    return top;
}

int oracle_abi_version(void) { return 1; }
