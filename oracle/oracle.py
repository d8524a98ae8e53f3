"""CPU oracle for the sparse hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package never does.  It restates, in numpy on top of the plain-C
loops of ``csr_oracle.c``, the algorithm of the reference's hot path:

* ``sparse_mm`` forward/backward            reference sparse_matmul.py:141-234
* ``sparse_triangular_solve`` fwd/bwd       reference sparse_solve.py:161-252, _compat.py:42-48
* ``sparse_generic_solve`` backward rule    reference sparse_solve.py:455-519
* ``linear_cg`` (+ Jacobi, Lanczos output)   reference utils/linear_cg.py:213-430
* ``bicgstab`` (+ diagonal preconditioner)   reference utils/bicgstab.py:112-247
* ``minres`` (shifts, value, preconditioner) reference utils/minres.py:140-311

The arithmetic itself lives in PyTorch ATen (third-party; the reference pins ``torch>=2.5``,
pyproject.toml:23), so each kernel-level function restates the mathematical definition of the
ATen call at the cited line.  PINNING: ``tests/test_oracle_golden.py`` checks every function
here against outputs of the real reference captured by ``tests/golden/make_golden.py`` (run in
the build container, where the reference is importable) — parity is pinned, not assumed.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "csr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", _SO, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


I64 = ctypes.c_int64


# ---- kernel-level restatements ---------------------------------------------------------------

def csr_spmm(crow, col, val, B):
    """C = A·B (reference sparse_matmul.py:155)."""
    dt = val.dtype
    crow, col, val, B = _i64(crow), _i64(col), _c(val, dt), _c(B, dt)
    n, p = crow.size - 1, B.shape[1]
    C = np.empty((n, p), dtype=dt)
    fn = getattr(_load(), "oracle_csr_spmm_" + _sfx(dt))
    fn(I64(n), _ptr(crow), _ptr(col), _ptr(val), _ptr(B), I64(p), _ptr(C), I64(p), I64(p))
    return C


def csr_spmm_t(crow, col, val, G, n_cols):
    """D = Aᵀ·G (reference sparse_matmul.py:229)."""
    dt = val.dtype
    crow, col, val, G = _i64(crow), _i64(col), _c(val, dt), _c(G, dt)
    n, p = crow.size - 1, G.shape[1]
    D = np.empty((n_cols, p), dtype=dt)
    fn = getattr(_load(), "oracle_csr_spmm_t_" + _sfx(dt))
    fn(I64(n), I64(n_cols), _ptr(crow), _ptr(col), _ptr(val), _ptr(G), I64(p), _ptr(D), I64(p), I64(p))
    return D


def csr_sddmm(crow, col, R, Cm, alpha=1.0):
    """out[k] = alpha·<R[row k], Cm[col k]> (reference sparse_matmul.py:186-205)."""
    dt = R.dtype
    crow, col, R, Cm = _i64(crow), _i64(col), _c(R, dt), _c(Cm, dt)
    n, p = crow.size - 1, R.shape[1]
    out = np.empty(col.size, dtype=dt)
    ct = ctypes.c_float if dt == np.float32 else ctypes.c_double
    fn = getattr(_load(), "oracle_csr_sddmm_" + _sfx(dt))
    fn(I64(n), _ptr(crow), _ptr(col), _ptr(R), I64(p), _ptr(Cm), I64(p), ct(alpha), _ptr(out), I64(p))
    return out


def coo_sddmm(row, col, R, Cm, alpha=1.0):
    """COO branch of the same rule (reference sparse_matmul.py:185,201-205)."""
    dt = R.dtype
    row, col, R, Cm = _i64(row), _i64(col), _c(R, dt), _c(Cm, dt)
    out = np.empty(row.size, dtype=dt)
    ct = ctypes.c_float if dt == np.float32 else ctypes.c_double
    fn = getattr(_load(), "oracle_coo_sddmm_" + _sfx(dt))
    fn(I64(row.size), _ptr(row), _ptr(col), _ptr(R), I64(R.shape[1]), _ptr(Cm), I64(R.shape[1]), ct(alpha),
       _ptr(out), I64(R.shape[1]))
    return out


def csr_sptrsm(crow, col, val, B, upper, unit=False, transpose=False):
    """X = op(A)^{-1}B (reference _compat.py:42-48)."""
    dt = val.dtype
    crow, col, val, B = _i64(crow), _i64(col), _c(val, dt), _c(B, dt)
    n, p = crow.size - 1, B.shape[1]
    X = np.empty((n, p), dtype=dt)
    fn = getattr(_load(), "oracle_csr_sptrsm_" + _sfx(dt))
    rc = fn(I64(n), _ptr(crow), _ptr(col), _ptr(val), int(upper), int(unit), int(transpose), _ptr(B), I64(p),
            _ptr(X), I64(p), I64(p))
    assert rc == 0
    return X


def csr_levels(crow, col, upper=False):
    """Dependency levels of a triangular pattern (what a level-scheduled solve would barrier on)."""
    crow, col = _i64(crow), _i64(col)
    fn = _load().oracle_csr_levels
    fn.restype = ctypes.c_int64
    return int(fn(I64(crow.size - 1), _ptr(crow), _ptr(col), int(upper)))


def expand_rows(crow):
    """repeat_interleave(arange(n), diff(crow)) (reference sparse_matmul.py:190-192)."""
    crow = _i64(crow)
    return np.repeat(np.arange(crow.size - 1, dtype=np.int64), np.diff(crow))


def block_diag_csr(crows, cols, vals, m):
    """Batched CSR (b, ·) → one block-diagonal CSR (reference utils/utils.py:615-645)."""
    b, nnz = cols.shape
    crow = np.concatenate([[0]] + [crows[i, 1:].astype(np.int64) + i * nnz for i in range(b)])
    col = np.concatenate([cols[i].astype(np.int64) + i * m for i in range(b)])
    return crow, col, vals.reshape(-1)


# ---- autograd-rule restatements ----------------------------------------------------------------

def sparse_mm_fwd_bwd(crow, col, val, B, G, n_cols):
    """forward C and backward (gradA values at A's pattern, gradB) — sparse_matmul.py:141-234."""
    C = csr_spmm(crow, col, val, B)
    gradA = csr_sddmm(crow, col, G, B, 1.0)
    gradB = csr_spmm_t(crow, col, val, G, n_cols)
    return C, gradA, gradB


def triangular_solve_fwd_bwd(crow, col, val, B, G, upper, unit, transpose):
    """x, gradA values, gradB — sparse_solve.py:161-252."""
    x = csr_sptrsm(crow, col, val, B, upper, unit, transpose)
    gradB = csr_sptrsm(crow, col, val, G, upper, unit, not transpose)  # :202-204
    if transpose:  # :223-225  -gradB[col]·x[row]
        gradA = csr_sddmm(crow, col, x, gradB, -1.0)
    else:  # :226-228  -gradB[row]·x[col]
        gradA = csr_sddmm(crow, col, gradB, x, -1.0)
    return x, gradA, gradB


def linear_cg(crow, col, val, rhs, tolerance, max_iter=1000, eps=1e-10, stop_updating_after=1e-10,
              record_iters=(), n_tridiag=0, max_tridiag_iter=20, precond_diag=None):
    """Multi-RHS CG (reference utils/linear_cg.py:213-430), optionally with a diagonal (Jacobi) preconditioner
    `precond_diag` (z = r * precond_diag) and with the Lanczos tridiagonal matrices of the first `n_tridiag` columns
    (reference :303-310, :385-427).

    Returns (x, iterations, snapshots) where snapshots[k] is the un-normalised iterate after k iterations for k in
    record_iters — and, for n_tridiag > 0, (x, iterations, snapshots, T) with T of shape (n_tridiag, r, r)."""
    dt = rhs.dtype
    e = dt.type(eps)
    n = rhs.shape[0]
    n_tridiag_iter = min(max_tridiag_iter, n)  # :251
    rhs_norm = np.sqrt((rhs * rhs).sum(0, keepdims=True, dtype=dt))  # :257
    rhs_is_zero = rhs_norm < e
    rhs_norm = np.where(rhs_is_zero, dt.type(1), rhs_norm)
    rhs = rhs / rhs_norm  # :262
    x = np.zeros_like(rhs)
    r = rhs - csr_spmm(crow, col, val, x)  # :266
    rnorm = np.sqrt((r * r).sum(0, keepdims=True, dtype=dt))
    has_conv = rnorm < dt.type(stop_updating_after)
    snaps = {}
    t_mat = np.zeros((n_tridiag_iter, n_tridiag_iter, n_tridiag), dtype=dt) if n_tridiag else None

    def done(k_done, last):
        if n_tridiag:
            return x * rhs_norm, k_done, snaps, np.ascontiguousarray(t_mat[: last + 1, : last + 1].transpose(2, 0, 1))  # :426-427
        return x * rhs_norm, k_done, snaps

    if has_conv.all() and not n_tridiag:  # :286
        return done(0, 0)
    prec = (lambda v: v.copy()) if precond_diag is None else (lambda v: v * precond_diag.reshape(-1, 1).astype(dt))
    z = prec(r)
    pvec = z.copy()
    rz = (z * r).sum(0, keepdims=True, dtype=dt)  # :294
    k_done = 0
    update_tridiag, last_tridiag_iter = True, 0
    prev_alpha_recip = prev_beta = None
    for k in range(max_iter):
        Ap = csr_spmm(crow, col, val, pvec)  # :322
        pAp = (pvec * Ap).sum(0, keepdims=True, dtype=dt)  # :64-65
        zero = pAp < e
        alpha = np.where(zero, dt.type(0), rz / np.where(zero, dt.type(1), pAp))  # :68-71
        alpha = np.where(has_conv, dt.type(0), alpha)  # :74
        r = r - alpha * Ap  # :78
        z = prec(r)
        x = x + alpha * pvec  # :32
        rz_old = rz
        rz = (z * r).sum(0, keepdims=True, dtype=dt)  # :36-37
        zero = rz_old < e
        beta = np.where(zero, dt.type(0), rz / np.where(zero, dt.type(1), rz_old))  # :40-43
        pvec = pvec * beta + z  # :47
        rnorm = np.sqrt((r * r).sum(0, keepdims=True, dtype=dt))  # :372
        rnorm = np.where(rhs_is_zero, dt.type(0), rnorm)
        has_conv = rnorm < dt.type(stop_updating_after)  # :374
        k_done = k + 1
        if k_done in record_iters:
            snaps[k_done] = (x * rhs_norm).copy()
        if (k >= min(10, max_iter - 1) and rnorm.mean() < tolerance
                and not (n_tridiag and k < min(n_tridiag_iter, max_iter - 1))):  # :376-382
            break
        if n_tridiag and k < n_tridiag_iter and update_tridiag:  # :385-406
            a_t = alpha[0, :n_tridiag]
            b_t = beta[0, :n_tridiag]
            a_zero = a_t == 0
            a_recip = dt.type(1) / np.where(a_zero, dt.type(1), a_t)
            if k == 0:
                t_mat[k, k] = a_recip
            else:
                t_mat[k, k] = a_recip + prev_beta * prev_alpha_recip
                t_mat[k, k - 1] = np.sqrt(prev_beta) * prev_alpha_recip
                t_mat[k - 1, k] = t_mat[k, k - 1]
                if t_mat[k - 1, k].max() < 1e-6:
                    update_tridiag = False
            last_tridiag_iter = k
            prev_alpha_recip = a_recip
            prev_beta = b_t.copy()
    return done(k_done, last_tridiag_iter)


def bicgstab(crow, col, val, b, matvec_max=None, abstol=1e-8, reltol=1e-6, precond_diag=None, x0=None):
    """Single-vector BiCGSTAB (reference utils/bicgstab.py:126-247), optionally right-preconditioned by a diagonal
    (`precond_diag`: q = M p, z = M s, :191-194, :216-219) and with an initial guess `x0` (the reference then starts from the
    UNCORRECTED residual r0 = b and spends no matvec on it, :158-161)."""
    dt = b.dtype
    n = b.shape[0]
    mv = lambda v: csr_spmm(crow, col, val, v.reshape(-1, 1)).reshape(-1)
    pre = (lambda v: v) if precond_diag is None else (lambda v: np.asarray(precond_diag, dtype=dt).reshape(-1) * v)
    matvec_max = 2 * n if matvec_max is None else matvec_max
    if x0 is None:
        x = np.zeros(n, dtype=dt)
        r0 = b - mv(x)
        n_mv = 1
    else:
        x = np.array(x0, dtype=dt)
        r0 = b.copy()
        n_mv = 0
    rho = alpha = omega = dt.type(1)
    rho_next = np.dot(r0, r0)
    resid = resid0 = np.abs(np.sqrt(rho_next))
    thresh = max(abstol, reltol * resid0)
    finished = resid <= thresh or n_mv >= matvec_max
    if not finished:
        r = r0.copy()
        p = np.zeros(n, dtype=dt)
        v = np.zeros(n, dtype=dt)
    while not finished:
        beta = rho_next / rho * alpha / omega
        rho = rho_next
        p = p * beta - beta * omega * v + r
        q = pre(p)
        v = mv(q)
        n_mv += 1
        alpha = rho / np.dot(r0, v)
        s = r - alpha * v
        resid = np.linalg.norm(s)
        if resid <= thresh:
            x = x + alpha * q
            break
        if n_mv >= matvec_max:
            break
        z = pre(s)
        t = mv(z)
        n_mv += 1
        omega = np.dot(t, s) / np.dot(t, t)
        rho_next = -omega * np.dot(r0, t)
        r = s - omega * t
        x = x + omega * z + alpha * q
        resid = np.linalg.norm(r)
        if resid <= thresh or n_mv >= matvec_max:
            break
    return x, n_mv


def minres(crow, col, val, rhs, shifts=(0.0,), value=None, precond_diag=None, max_iter=1000, tolerance=1e-4, eps=1e-25):
    """MINRES for (value·A + shift_s·I) x_s = b, all right-hand sides and shifts at once (reference utils/minres.py:140-311):
    right-hand sides normalised (:221-233), max_iter capped at n + 1 (:236), Lanczos with an optional diagonal preconditioner
    (:241-271), one Givens QR per shift (:273-289), solution update (:290-296), relative-update stopping test on every tenth
    iteration, averaged over shifts and columns (:299-305).  Returns [n_shift][n][p]."""
    dt = rhs.dtype
    B = rhs.reshape(rhs.shape[0], -1).astype(dt)
    n, p = B.shape
    mm = lambda V: csr_spmm(crow, col, val, np.ascontiguousarray(V))
    apply = (lambda V: mm(V)) if value is None else (lambda V: mm(V) * dt.type(value))
    pre = (lambda V: V.copy()) if precond_diag is None else (lambda V: V * np.asarray(precond_diag, dtype=dt).reshape(-1, 1))
    sh = np.asarray(shifts, dtype=dt).reshape(-1, 1, 1)
    S = sh.shape[0]
    nrm = np.linalg.norm(B, axis=0, keepdims=True)
    zero = nrm < 1e-10
    nrm = np.where(zero, dt.type(1), nrm)
    B = B / nrm
    max_iter = min(max_iter, n + 1)
    sol = np.zeros((S, n, p), dtype=dt)
    z_pp = np.zeros_like(B)
    z_p = B.copy()
    q_p = pre(z_p)
    with np.errstate(invalid="ignore", divide="ignore"):
        beta_p = np.sqrt((z_p * q_p).sum(axis=0, keepdims=True))
        z_p = z_p / beta_p
        q_p = q_p / beta_p
        c_pp = np.ones((S, 1, p), dtype=dt)
        s_pp = np.zeros_like(c_pp)
        c_p = np.ones_like(c_pp)
        s_p = np.zeros_like(c_pp)
        w_pp = np.zeros_like(sol)
        w_p = np.zeros_like(sol)
        scale_p = np.repeat(beta_p[None], S, axis=0)
        for i in range(max_iter + 2):
            prod = apply(q_p)
            alpha = (prod * q_p).sum(axis=0, keepdims=True)
            z_c = prod - alpha * z_p - beta_p * z_pp
            q_c = pre(z_c)
            beta_c = np.maximum(np.sqrt((z_c * q_c).sum(axis=0, keepdims=True)), dt.type(eps))
            z_c = z_c / beta_c
            q_c = q_c / beta_c
            subsub = s_pp * beta_p
            sub = c_pp * beta_p
            alpha_s = alpha + sh
            diag = alpha_s * c_p - s_p * sub
            sub = sub * c_p + s_p * alpha_s
            radius = np.sqrt(diag * diag + beta_c * beta_c)
            c_c = diag / radius
            s_c = beta_c / radius
            diag = diag * c_c + s_c * beta_c
            scale_c = -(scale_p * s_c)
            scale_p = scale_p * c_c
            w_c = (q_p - sub * w_p - subsub * w_pp) / diag
            update = w_c * scale_p
            sol = sol + update
            if (i + 1) % 10 == 0:
                ratio = np.linalg.norm(update, axis=-2) / np.linalg.norm(sol, axis=-2)
                if ratio.mean() < tolerance:
                    break
            z_pp, z_p = z_p, z_c
            q_p = q_c
            beta_p = beta_c
            c_pp, c_p = c_p, c_c
            s_pp, s_p = s_p, s_c
            w_pp, w_p = w_p, w_c
            scale_p = scale_c
    sol = np.where(zero[None], dt.type(0), sol)
    return sol * nrm[None]
