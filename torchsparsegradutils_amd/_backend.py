"""ctypes binding of ``libtsgu_hip.so`` (C ABI in ``include/tsgu_hip.h``).

PyTorch is used for device memory and streams only: every wrapper below takes
tensors, checks them, and hands raw device pointers + the current HIP stream to
the hand-written gfx950 kernels.  There is NO fallback: if the shared library is
missing, or a tensor does not live on a HIP device, the call raises.
"""

from __future__ import annotations

import ctypes
import functools
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSGU_LIB_PATH") or os.path.join(_HERE, "csrc", "libtsgu_hip.so")  # env override: kernel A/B builds

TSGU_F32, TSGU_F64, TSGU_BF16 = 0, 1, 2
TSGU_I32, TSGU_I64 = 0, 1

_VTYPE = {torch.float32: TSGU_F32, torch.float64: TSGU_F64, torch.bfloat16: TSGU_BF16}
_ITYPE = {torch.int32: TSGU_I32, torch.int64: TSGU_I64}

_lib = None
_lib_lock = threading.Lock()

_i64 = ctypes.c_int64
_int = ctypes.c_int
_ptr = ctypes.c_void_p
_dbl = ctypes.c_double

# name -> (restype, argtypes); must list every symbol declared in include/tsgu_hip.h
ABI_VERSION = 7          # TSGU_ABI_VERSION of include/tsgu_hip.h this binding was written against

SIGNATURES = {
    "tsgu_abi_version": (_int, []),
    "tsgu_status_string": (ctypes.c_char_p, [_int]),
    "tsgu_device_info": (_int, [_int, ctypes.c_char_p, _int, ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "tsgu_device_copy": (_int, [_ptr, _ptr, _i64, _int, _ptr]),
    "tsgu_device_cu_count": (_int, [_int, ctypes.POINTER(_int)]),
    "tsgu_index_fingerprint": (_int, [_int, _i64, _ptr, _ptr, _int, _int, _ptr]),
    "tsgu_index_fingerprint_match": (_int, [_int, _i64, _ptr, _ptr, _ptr, _ptr, _int, _int, _ptr]),
    "tsgu_tile_geometry": (_int, [_int, _i64, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "tsgu_csr_spmm_tile": (_int, [_int, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _int, _ptr]),
    "tsgu_csr_sddmm_tile": (_int, [_int, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _dbl, _i64, _int, _ptr]),
    "tsgu_csr_spmm": (
        _int,
        [_int, _int, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _i64,
         _ptr, _i64, _ptr, _int, _ptr],
    ),
    "tsgu_spmm_num_blocks": (_i64, [_int, _i64, _i64, _i64, _i64]),
    "tsgu_csr_sddmm": (
        _int,
        [_int, _int, _i64, _i64, _i64, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _i64, _i64, _ptr, _dbl, _int, _i64, _i64,
         _int, _ptr],
    ),
    "tsgu_coo_sddmm": (_int, [_int, _int, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _dbl, _i64, _int, _ptr]),
    "tsgu_csr_mm_backward": (
        _int,
        [_int, _int, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _i64, _i64, _ptr, _ptr, _i64,
         _i64, _i64, _i64, _int, _ptr],
    ),
    "tsgu_minres_scalar": (
        _int,
        [_int, _int, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _dbl, _dbl, _dbl, _i64, _int, _ptr],
    ),
    "tsgu_minres_vector": (
        _int,
        [_int, _int, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _int, _int, _ptr],
    ),
    "tsgu_minres_scalar_ms": (
        _int,
        [_int, _int, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _dbl, _dbl, _ptr, _int, _dbl, _i64, _int, _ptr],
    ),
    "tsgu_minres_vector_ms": (
        _int,
        [_int, _int, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _int, _int, _i64, _dbl, _int, _ptr],
    ),
    "tsgu_bicg_update_x_precond": (_int, [_int, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _int, _ptr]),
    "tsgu_rowpack_geometry": (
        _int,
        [_int, _i64, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int),
         ctypes.POINTER(_int)],
    ),
    "tsgu_csr_spmm_rowpack": (_int, [_int, _int, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _int, _ptr]),
    "tsgu_csr_mm_backward_rowpack": (
        _int, [_int, _int, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _int, _ptr]),
    "tsgu_csr_sddmm_rowpack": (
        _int, [_int, _int, _i64, _i64, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _dbl, _i64, _int, _ptr]),
    "tsgu_lattice_lds_bytes": (_int, [_int, _int, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _int]),
    "tsgu_csr_spmm_lattice": (_int, [_int, _ptr, _i64, _i64, _ptr, _ptr, _i64, _ptr, _i64, _i64, _int, _ptr]),
    "tsgu_csr_spmm_lattice_dot": (_int, [_int, _ptr, _i64, _i64, _ptr, _ptr, _i64, _ptr, _i64, _i64, _ptr, _i64, _ptr, _ptr, _int, _ptr]),
    "tsgu_csr_sddmm_lattice": (_int, [_int, _ptr, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _dbl, _i64, _int, _ptr]),
    "tsgu_march_supported": (_int, [_int, _int, _int, _int]),
    "tsgu_march_lds_bytes": (_int, [_int, _int, _i64, _int, _int, _int, _int, _int, _int]),
    "tsgu_csr_spmm_march": (_int, [_int, _ptr, _int, _i64, _i64, _ptr, _ptr, _i64, _ptr, _i64, _i64, _int, _ptr]),
    "tsgu_csr_sddmm_march": (_int, [_int, _ptr, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _dbl, _int, _i64, _int, _ptr]),
    "tsgu_lattice_slots": (_int, []),
    "tsgu_lattice_rows": (_int, [_int, _i64, _ptr, _ptr, _int, _int, _int, _int, _ptr, _int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                 _int, _int, _int, _ptr]),
    "tsgu_lattice_row_codes": (_int, [_int, _i64, _ptr, _ptr, _int, _int, _int, _int, _ptr, _int, _ptr, _int, _ptr, _int, _ptr]),
    "tsgu_lattice_block_classes": (_int, [_i64, _ptr, _int, _int, _int, _int, _int, _int, _int, _ptr, _int, _ptr]),
    "tsgu_csr_sptrsm": (
        _int,
        [_int, _int, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _int, _int, _ptr, _i64, _i64, _ptr, _i64, _i64, _ptr, _int, _int, _ptr],
    ),
    "tsgu_sptrsm_work_bytes": (_i64, [_i64, _i64]),
    "tsgu_cg_fold_rows": (_i64, []),
    "tsgu_cg_alpha": (_int, [_int, _ptr, _i64, _ptr, _ptr, _ptr, _dbl, _i64, _int, _ptr]),
    "tsgu_cg_update1": (_int, [_int, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _int, _ptr]),
    "tsgu_cg_num_blocks": (_i64, [_int, _i64, _i64]),
    "tsgu_cg_beta": (_int, [_int, _ptr, _i64, _ptr, _ptr, _dbl, _dbl, _dbl, _int, _int, _i64, _int, _ptr]),
    "tsgu_cg_beta_precond": (_int, [_int, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _dbl, _dbl, _dbl, _int, _int, _i64, _int, _ptr]),
    "tsgu_cg_update2": (_int, [_int, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _int, _ptr]),
    "tsgu_cg2_num_blocks": (_i64, [_int, _i64, _i64]),
    "tsgu_cg2_residual": (_int, [_int, _i64, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _int, _dbl, _ptr, _int, _ptr]),
    "tsgu_cg2_direction": (_int, [_int, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _int, _dbl, _dbl, _dbl, _int, _ptr, _int, _int, _ptr]),
    "tsgu_cg_update1_alpha": (_int, [_int, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _dbl, _ptr, _int, _ptr]),
    "tsgu_bicg_scalar": (_int, [_int, _int, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _dbl, _dbl, _int, _int, _i64, _int, _ptr]),
    "tsgu_bicg_vector": (_int, [_int, _int, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _int, _ptr]),
    "tsgu_coldot_max_blocks": (_i64, [_i64, _i64]),
    "tsgu_coldot": (_int, [_int, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _int, _ptr]),
}


class HipExtensionMissing(RuntimeError):
    pass


def load_library():
    """Load (once) and return the ctypes handle; raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                f"{LIB_PATH} not found: build the gfx950 extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C torchsparsegradutils_amd/csrc). "
                "torchsparsegradutils_amd has no CPU or eager fallback."
            )
        # torch has already loaded its libamdhip64.so (same SONAME), so the kernels register
        # with the runtime that owns torch's streams and allocations.
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError => header/library mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.tsgu_abi_version() != ABI_VERSION:
            raise HipExtensionMissing("libtsgu_hip.so ABI version mismatch; rebuild the extension")
        _lib = lib
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load_library().tsgu_status_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (tsgu status {status})")


def vtype_of(t: torch.Tensor) -> int:
    try:
        return _VTYPE[t.dtype]
    except KeyError:
        raise RuntimeError(f"torchsparsegradutils_amd: unsupported value dtype {t.dtype}") from None


def itype_of(t: torch.Tensor) -> int:
    try:
        return _ITYPE[t.dtype]
    except KeyError:
        raise RuntimeError(f"torchsparsegradutils_amd: unsupported index dtype {t.dtype}") from None


def operand_device(*tensors: torch.Tensor) -> torch.device:
    """The one device all operands of a call live on (CPU included: CPU operands are computed by the torch-op path, _cpu.py;
    the HIP bindings below still refuse them through `require_device`)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"all operands must be on the same device, got {dev} and {t.device}")
    return dev


def require_device(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                f"the gfx950 kernels of torchsparsegradutils_amd were handed a tensor on '{t.device}': CPU operands are served by "
                "the torch-op path (_cpu.py) only when ALL operands of a call live on the CPU"
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"all operands must be on the same device, got {dev} and {t.device}")
    return dev


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def strided2d(t: torch.Tensor):
    """(tensor, row stride, column stride) for a 2-D dense operand WITHOUT copying when it is row-major or a
    transposed view (unit row stride — what ``bvec.t()`` in the reference's sparse multivariate normal hands over,
    distributions/sparse_multivariate_normal.py:96); anything else is made contiguous."""
    if t.dim() == 2 and t.size(0) > 1 and t.size(1) > 1 and t.stride(0) == 1 and t.stride(1) >= t.size(0):
        return t, 1, t.stride(1)
    t = rowmajor(t)
    return t, _ld(t), 1


def is_transposed_view(t: torch.Tensor) -> bool:
    """2-D operand with unit ROW stride (``x.t()`` of a contiguous matrix): consumed in place by K1 / K4."""
    return t.dim() == 2 and t.size(0) > 1 and t.size(1) > 1 and t.stride(0) == 1 and t.stride(1) >= t.size(0)


def rowmajor(t: torch.Tensor) -> torch.Tensor:
    """Return `t` (2-D or 3-D) with unit stride in the last dim and a sane leading dimension."""
    if t.is_contiguous():
        return t
    if t.stride(-1) != 1 and t.size(-1) != 1:
        return t.contiguous()
    if t.dim() >= 2 and t.size(-2) > 1 and t.stride(-2) < t.size(-1):
        return t.contiguous()
    if t.dim() == 3 and t.size(0) > 1 and t.stride(0) < t.size(1) * t.stride(1):
        return t.contiguous()
    if t.size(-1) == 1 and t.stride(-1) != 1:
        return t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(-2) if t.size(-2) > 1 else max(t.size(-1), 1)


def _bs(t: torch.Tensor) -> int:
    return t.stride(0) if (t.dim() == 3 and t.size(0) > 1) else 0


def csr_spmm(crow, col, val, B, n_rows: int, n_cols: int, perm=None, out=None, dot_w=None, max_row_nnz: int = 0):
    """C = A·B for (batched) CSR arrays.  B: (m, p) or (b, m, p).  Returns C or (C, dot_partial).
    A 2-D B that is a transposed view is consumed in place and C comes back in the same (transposed) layout."""
    lib = load_library()
    dev = require_device(crow, col, val, B, perm, out, dot_w)
    if val.dtype != B.dtype:
        raise RuntimeError(f"expected A and B to have the same dtype, got {val.dtype} and {B.dtype}")
    batched = B.dim() == 3
    batch = B.size(0) if batched else 1
    p = B.size(-1)
    b_cs = 1
    if not batched and dot_w is None and out is None:
        B, ldb, b_cs = strided2d(B)
    else:
        B = rowmajor(B)
        ldb = _ld(B)
    nnz = col.size(-1)
    crow, col, val = crow.contiguous(), col.contiguous(), val.contiguous()
    if perm is not None:
        perm = perm.contiguous()
    shape = (batch, n_rows, p) if batched else (n_rows, p)
    c_cs = 1
    if out is None:
        if b_cs != 1:
            out = torch.empty((p, n_rows), dtype=B.dtype, device=dev).t()   # same layout as the operand: coalesced stores
            ldc, c_cs = 1, n_rows
        else:
            out = torch.empty(shape, dtype=B.dtype, device=dev)
            ldc = _ld(out)
    else:
        ldc = _ld(out)
    partial = None
    vt = vtype_of(val)
    if dot_w is not None:
        nblk = lib.tsgu_spmm_num_blocks(vt, n_rows, nnz, p, max_row_nnz)
        partial = torch.empty((batch * nblk, p), dtype=B.dtype, device=dev)
        dot_w = rowmajor(dot_w)
    with torch.cuda.device(dev):
        check(
            lib.tsgu_csr_spmm(
                vt, itype_of(crow), n_rows, n_cols, nnz, _p(crow), _p(col), _p(val), _p(perm),
                _p(B), ldb, b_cs, _bs(B), _p(out), ldc, c_cs, _bs(out), p, batch, max_row_nnz,
                _p(dot_w), _ld(dot_w) if dot_w is not None else 0, _p(partial), dev.index, _stream(dev),
            ),
            "tsgu_csr_spmm",
        )
    return out if dot_w is None else (out, partial)


def csr_sddmm(crow, col, G, B, n_rows: int, n_cols: int, alpha: float = 1.0, swap_roles: bool = False):
    """out[k] = alpha·<G[row k], B[col k]> (or roles swapped) for (batched) CSR patterns."""
    lib = load_library()
    dev = require_device(crow, col, G, B)
    if G.dtype != B.dtype:
        raise RuntimeError(f"expected both dense operands to have the same dtype, got {G.dtype} and {B.dtype}")
    batched = G.dim() == 3
    batch = G.size(0) if batched else 1
    p = G.size(-1)
    G, B = rowmajor(G), rowmajor(B)
    crow, col = crow.contiguous(), col.contiguous()
    nnz = col.size(-1)
    out = torch.empty(col.shape, dtype=G.dtype, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.tsgu_csr_sddmm(
                vtype_of(G), itype_of(crow), n_rows, n_cols, nnz, _p(crow), _p(col),
                _p(G), _ld(G), _bs(G), _p(B), _ld(B), _bs(B), _p(out), float(alpha), int(bool(swap_roles)),
                p, batch, dev.index, _stream(dev),
            ),
            "tsgu_csr_sddmm",
        )
    return out


def csr_mm_backward(tplan, val, G, B, n_rows: int, n_cols: int):
    """(gradA values in A's order, gradB) in one pass over the transposed plan `tplan` of A."""
    lib = load_library()
    dev = require_device(tplan.crow, val, G, B)
    if not (val.dtype == G.dtype == B.dtype):
        raise RuntimeError("expected A, B and the upstream gradient to have the same dtype")
    batched = G.dim() == 3
    batch = G.size(0) if batched else 1
    p = G.size(-1)
    G, B = rowmajor(G), rowmajor(B)
    val = val.contiguous()
    gradA = torch.empty(val.shape, dtype=val.dtype, device=dev)
    gradB = torch.empty(B.shape, dtype=B.dtype, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.tsgu_csr_mm_backward(
                vtype_of(val), itype_of(tplan.crow), n_rows, n_cols, tplan.col.size(-1),
                _p(tplan.crow), _p(tplan.col), _p(tplan.perm), _p(val),
                _p(G), _ld(G), _bs(G), _p(B), _ld(B), _bs(B), _p(gradA), _p(gradB), _ld(gradB), _bs(gradB),
                p, batch, dev.index, _stream(dev),
            ),
            "tsgu_csr_mm_backward",
        )
    return gradA, gradB


def fused_backward_supported(dtype: torch.dtype, p: int) -> bool:
    wide = {torch.float32: 4, torch.bfloat16: 8}.get(dtype)
    if wide is None or p <= 0:
        return False
    vec = wide if p % wide == 0 else 1
    return (p + vec - 1) // vec <= 64


class _RowpackPlanStruct(ctypes.Structure):
    """``tsgu_rowpack_plan`` of include/tsgu_hip.h."""

    _fields_ = [("nblocks", _i64), ("ecap", ctypes.c_int32), ("ucap", ctypes.c_int32), ("nclasses", ctypes.c_int32),
                ("rows_per_group", ctypes.c_int32)] + [(k, _ptr) for k in ("uptr", "ucol", "upos", "sperm", "order", "vpair", "eptr",
                                                                     "wcls", "wbase", "cne", "srcstart")]


def _plan_struct(rp):
    """ctypes image of a _pattern.RowPackPlan (cached on the plan; the plan keeps the tensors alive)."""
    st = rp._cstruct
    if st is None:
        st = _RowpackPlanStruct(rp.nblocks, rp.ecap, rp.ucap, rp.nclasses, rp.group, _p(rp.uptr), _p(rp.ucol), _p(rp.upos), _p(rp.sperm),
                                _p(rp.order), _p(rp.vpair), _p(rp.eptr), _p(rp.wcls), _p(rp.wbase), _p(rp.cne), _p(rp.srcstart))
        rp._cstruct = st
    return ctypes.addressof(st)


@functools.lru_cache(maxsize=None)
def rowpack_geometry(dtype: torch.dtype, p: int):
    """(rows_per_block, (max_entries, max_union, lds_budget_bytes), entry_lanes) of the row-pair gather kernels, or None."""
    if dtype not in (torch.float32, torch.bfloat16) or p <= 0:
        return None
    lib = load_library()
    r, e, a, b, c = _int(0), _int(0), _int(0), _int(0), _int(0)
    if lib.tsgu_rowpack_geometry(_VTYPE[dtype], p, ctypes.byref(r), ctypes.byref(e), ctypes.byref(a), ctypes.byref(b),
                                 ctypes.byref(c)) != 0:
        return None
    return r.value, (a.value, b.value, c.value), e.value


def csr_spmm_rowpack(crow, val, rp, B, n_rows: int):
    """C = A·B through the row-pair union walk; `rp` is a _pattern.RowPackPlan of the walked pattern."""
    lib = load_library()
    dev = require_device(crow, val, B)
    B = rowmajor(B)
    p = B.size(-1)
    out = torch.empty((n_rows, p), dtype=B.dtype, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.tsgu_csr_spmm_rowpack(
                vtype_of(val), itype_of(crow), n_rows, B.size(0), rp.nnz, _p(crow), _plan_struct(rp), _p(val.contiguous()),
                _p(B), _ld(B), _p(out), _ld(out), p, dev.index, _stream(dev),
            ),
            "tsgu_csr_spmm_rowpack",
        )
    return out


def csr_sddmm_rowpack(crow, rp, R, Cm, n_rows: int, alpha: float = 1.0):
    """out[k] = alpha·<R[row k], Cm[col k]> in stored order through the row-pair union walk (plan of a stored-order pattern)."""
    lib = load_library()
    dev = require_device(crow, R, Cm)
    R, Cm = rowmajor(R), rowmajor(Cm)
    p = R.size(-1)
    out = torch.empty((rp.nnz,), dtype=R.dtype, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.tsgu_csr_sddmm_rowpack(
                vtype_of(R), itype_of(crow), n_rows, Cm.size(0), rp.nnz, _p(crow), _plan_struct(rp), _p(R), _ld(R), _p(Cm),
                _ld(Cm), _p(out), float(alpha), p, dev.index, _stream(dev),
            ),
            "tsgu_csr_sddmm_rowpack",
        )
    return out


# ---- row-block tile kernels (csrc/tile_impl.h) --------------------------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def tile_geometry(dtype: torch.dtype, p: int):
    """(rows_per_block, max_union, max_entries) of the tile kernels for (dtype, p), or None when they are not compiled for it."""
    if dtype != torch.float32 or p <= 0:
        return None
    lib = load_library()
    r, u, e = _int(0), _int(0), _int(0)
    if lib.tsgu_tile_geometry(_VTYPE[dtype], p, ctypes.byref(r), ctypes.byref(u), ctypes.byref(e)) != 0:
        return None
    return r.value, u.value, e.value


def _tile_struct(tp):
    st = tp._cstruct
    if st is None:
        from ._tile import TilePlanStruct

        st = TilePlanStruct(tp.n_rows, tp.n_cols, tp.nnz, tp.n_blocks, tp.rows_per_block, tp.max_union, tp.max_entries, 0, _p(tp.desc),
                            _p(tp.ucol), _p(tp.lidx), _p(tp.rptr), _p(tp.cpos), _p(tp.cslot), _p(tp.ent), _p(tp.xrow))
        tp._cstruct = st
    return ctypes.addressof(st)


def csr_spmm_tile(tp, val, B):
    """C = A·B (a plan with value chunks `cpos` / `cslot`: Aᵀ·G through A's own values) by the row-block tile walk."""
    lib = load_library()
    dev = require_device(val, B)
    B = rowmajor(B)
    p = B.size(-1)
    out = torch.empty((tp.n_rows, p), dtype=B.dtype, device=dev)
    with torch.cuda.device(dev):
        check(lib.tsgu_csr_spmm_tile(vtype_of(val), _tile_struct(tp), _p(val.contiguous()), _p(B), _ld(B), _p(out), _ld(out), p, dev.index,
                                     _stream(dev)), "tsgu_csr_spmm_tile")
    return out


def csr_sddmm_tile(tp, R, Cm, alpha: float = 1.0):
    """out[k] = alpha·<R[row k], Cm[col k]> in stored order by the row-block tile walk (plan of a stored-order pattern)."""
    lib = load_library()
    dev = require_device(R, Cm)
    R, Cm = rowmajor(R), rowmajor(Cm)
    p = R.size(-1)
    out = torch.empty((tp.nnz,), dtype=R.dtype, device=dev)
    with torch.cuda.device(dev):
        check(lib.tsgu_csr_sddmm_tile(vtype_of(R), _tile_struct(tp), _p(R), _ld(R), _p(Cm), _ld(Cm), _p(out), float(alpha), p, dev.index,
                                      _stream(dev)), "tsgu_csr_sddmm_tile")
    return out


def csr_mm_backward_rowpack(tcrow, rp, val, G, B, n_rows_t: int):
    """(gradA values in A's order, gradB) in one pass over the transposed pattern's RowPackPlan."""
    lib = load_library()
    dev = require_device(tcrow, val, G, B)
    G, B = rowmajor(G), rowmajor(B)
    p = G.size(-1)
    val = val.contiguous()
    grad_a = torch.empty_like(val)
    grad_b = torch.empty((n_rows_t, p), dtype=G.dtype, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.tsgu_csr_mm_backward_rowpack(
                vtype_of(val), itype_of(tcrow), n_rows_t, G.size(0), rp.nnz, _p(tcrow), _plan_struct(rp), _p(val), _p(G), _ld(G),
                _p(B), _ld(B), _p(grad_a), _p(grad_b), _ld(grad_b), p, dev.index, _stream(dev),
            ),
            "tsgu_csr_mm_backward_rowpack",
        )
    return grad_a, grad_b


# ---- lattice plane-sweep kernels (csrc/lattice_impl.h; plans from _lattice.py) -------------------------------
LAT_SPMM, LAT_SDDMM, LAT_SPMMT = 0, 1, 2


def lattice_lds_bytes(mode: int, vtype: int, p: int, ty: int, tz: int, ry: int, rz: int, nloc: int, recw: int, threads: int,
                      ring: int = 4, cpl: int = 1) -> int:
    """Dynamic LDS bytes of a lattice launch configuration, or a negative tsgu status when it does not fit."""
    return int(load_library().tsgu_lattice_lds_bytes(mode, vtype, p, ty, tz, ry, rz, nloc, recw, threads, ring, cpl))


def lattice_config(lp, mode: int, dtype: torch.dtype, p: int):
    """Launch configuration (tile, segments, record tables) of the _lattice.LatticePlan `lp` for these operands, or None."""
    from . import _lattice

    if dtype not in (torch.float32, torch.bfloat16, torch.float64):
        return None
    es = {torch.float32: 4, torch.bfloat16: 2, torch.float64: 8}[dtype]
    lanes = (p * es) // 16
    if (p * es) % 16 or lanes not in (1, 2, 4, 8, 16) or (lanes == 1 and not (mode == LAT_SPMM and dtype == torch.float32 and lp.kind == 0)):
        return None      # (one lane per row — 4 fp32 columns — is compiled for the stored-order product only)
    import sys

    return _lattice.config_for(lp, mode, _VTYPE[dtype], p, es, lattice_lds_bytes, be=sys.modules[__name__])


def lattice_tune(lp, mode: int, dtype: torch.dtype, p: int, time_ms):
    """Measured choice among the best-ranked launch configurations of `lp` for these operands (`_lattice.tune_config`)."""
    import sys

    from . import _lattice

    es = {torch.float32: 4, torch.bfloat16: 2, torch.float64: 8}[dtype]
    return _lattice.tune_config(lp, mode, _VTYPE[dtype], p, es, lattice_lds_bytes, sys.modules[__name__], time_ms)


def lattice_rows(crow, col, dims, status, slot, thash=None, trep=None, remap=None, ctable=None, lens=None, rcls=None, disp=None,
                 box_mask: int = 0, periodic: int = 0):
    """Row analysis kernels of csrc/lattice_plan.hip: pass 1 (hash -> slot table) when `ctable` is None, pass 2 (class
    assignment + exact check) otherwise; the rows of the transposed pattern when `disp` is given.  `box_mask` / `periodic`
    (pass 2 of the stored-order walk): also the plane-march condition, status[4]."""
    lib = _lib or load_library()
    dev = require_device(crow, col, status)
    nb, nx, ny, nz = dims
    with _on_device(dev):
        rc = lib.tsgu_lattice_rows(itype_of(crow), crow.numel() - 1, _p(crow), _p(col), nb, nx, ny, nz, _p(disp),
                                   0 if disp is None else disp.numel(), _p(slot), _p(thash), _p(trep), _p(remap), _p(ctable), _p(lens),
                                   _p(rcls), _p(status), int(box_mask), int(periodic), dev.index, _stream(dev))
    check(rc, "tsgu_lattice_rows")


def lattice_block_classes(rcls, n_rows, dims, ty, tz, nseg, mask):
    lib = _lib or load_library()
    dev = require_device(rcls, mask)
    nb, nx, ny, nz = dims
    with _on_device(dev):
        rc = lib.tsgu_lattice_block_classes(n_rows, _p(rcls), nb, nx, ny, nz, ty, tz, nseg, _p(mask), dev.index, _stream(dev))
    check(rc, "tsgu_lattice_block_classes")


def lattice_row_codes(crow, col, dims, rows, out, disp=None):
    lib = _lib or load_library()
    dev = require_device(crow, col, rows, out)
    nb, nx, ny, nz = dims
    with _on_device(dev):
        rc = lib.tsgu_lattice_row_codes(itype_of(crow), crow.numel() - 1, _p(crow), _p(col), nb, nx, ny, nz, _p(disp),
                                        0 if disp is None else disp.numel(), _p(rows), rows.numel(), _p(out), dev.index, _stream(dev))
    check(rc, "tsgu_lattice_row_codes")


def march_lds_bytes(mode: int, vtype: int, p: int, ty: int, tz: int, ry: int, rz: int, ncls: int, threads: int) -> int:
    """Dynamic LDS bytes of a plane-march launch configuration (csrc/march_impl.h), or a negative tsgu status."""
    return int(load_library().tsgu_march_lds_bytes(mode, vtype, p, ty, tz, ry, rz, ncls, threads))


def march_supported(mode: int, mask: int, uniform_len: int, threads: int) -> bool:
    """Is there a plane-march kernel for this product / displacement set / row form / workgroup size (csrc/march_sets.h)?"""
    return bool(load_library().tsgu_march_supported(mode, mask, uniform_len, threads))


def march_config(lp, mode: int, dtype: torch.dtype, p: int):
    """Launch configuration of the plane-march kernels for the stored-order _lattice.LatticePlan `lp`, or None when the pattern
    is not a full periodic box stencil / the operands are not covered (fp32, 32 or 64 columns)."""
    from . import _lattice

    if lp is None or lp.kind != 0:
        return None
    if dtype == torch.bfloat16:      # whole-line march (csrc/linemarch_impl.h): Aᵀ·G of a periodic 27-point stencil at 16 columns
        return _lattice.linemarch_config_for(lp, mode, _VTYPE[dtype], p, march_lds_bytes)
    if dtype != torch.float32:
        return None
    return _lattice.march_config_for(lp, mode, _VTYPE[dtype], p, march_lds_bytes, march_supported)


# Per-kernel timing hook (bench.py): a list to which the lattice launchers append (name, start event, end event) recorded on the
# launch stream around the C call.  None = off (the product never pays for it).
KERNEL_EVENTS = None


def _timed(name: str, dev: torch.device):
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record(torch.cuda.current_stream(dev))
    return name, ev


def _timed_end(tok, dev: torch.device) -> None:
    tok[1][1].record(torch.cuda.current_stream(dev))
    KERNEL_EVENTS.append((tok[0], tok[1][0], tok[1][1]))


class _on_device:
    """`with torch.cuda.device(dev)` only when `dev` is not already current (the context manager costs ~8 us per launch)."""

    __slots__ = ("ctx",)

    def __init__(self, dev: torch.device):
        self.ctx = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def _raw_stream(dev: torch.device) -> int:
    """Current HIP stream of `dev` as an integer handle (the accessor torch's own generated code uses: 0.1 us instead of 2)."""
    return torch._C._cuda_getCurrentRawStream(dev.index)


def csr_spmm_lattice(lp, cfg, val, B, dot: bool = False, skip: int = 0, dot_w=None, out=None):
    """C = A·B (plan kind 0) or Aᵀ·B for the transposed plan (kind 1; `val` in A's own order) by the plane sweep / plane march.
    `dot` (plane sweep, fp32, stored order): also the per-workgroup partial sums of <C[row], B[row]> per column — returns
    (C, partial [workgroups][p]), the Krylov loops' fused dot epilogue; `skip` (with `dot`): address of a device int32 — the launch
    does nothing when it is non-zero (iterations queued past the end of a solve); `dot_w` (with `dot`): the partial sums are of
    <C[row], dot_w[row]> instead; `out` (with `dot`): a contiguous (n_rows, p) tensor that receives C."""
    lib = _lib or load_library()
    dev = B.device
    if not B.is_cuda or val.device != dev:
        require_device(val, B)
        raise RuntimeError(f"all operands must be on the same device, got {val.device} and {dev}")
    if not B.is_contiguous():
        B = rowmajor(B)
    p = B.size(-1)
    n_rows = lp.n_rows
    if out is None or not dot:
        out = torch.empty((n_rows, p), dtype=B.dtype, device=dev)
    if not val.is_contiguous():
        val = val.contiguous()
    march = getattr(cfg, "march", False)
    transposed = cfg.mode == LAT_SPMMT if march else bool(lp.kind)
    if dot:
        if march or transposed or B.dtype not in (torch.float32, torch.float64):
            raise RuntimeError("csr_spmm_lattice: the dot epilogue exists for the fp32 / fp64 stored-order plane sweep only")
        nwg = lp.nb * cfg.nseg * -(-lp.ny // cfg.ty) * -(-lp.nz // cfg.tz)
        partial = torch.empty((nwg, p), dtype=B.dtype, device=dev)
        with _on_device(dev):
            rc = lib.tsgu_csr_spmm_lattice_dot(_VTYPE[val.dtype], cfg.struct_addr, n_rows, lp.nnz, val.data_ptr(), B.data_ptr(), _ld(B),
                                               out.data_ptr(), p, p, partial.data_ptr(), nwg, skip or None,
                                               None if dot_w is None else dot_w.data_ptr(), dev.index, _raw_stream(dev))
        if rc:
            check(rc, "tsgu_csr_spmm_lattice_dot")
        return out, partial
    tok = _timed("lattice_spmm_t" if transposed else "lattice_spmm", dev) if KERNEL_EVENTS is not None else None
    index = dev.index
    ctx = None if torch.cuda.current_device() == index else torch.cuda.device(dev)      # (the context manager costs ~8 us: only when needed)
    if ctx is not None:
        ctx.__enter__()
    try:
        ldb = B.stride(0) if B.size(0) > 1 else max(p, 1)
        stream = torch._C._cuda_getCurrentRawStream(index)
        if march:
            ct = cfg.col_tile          # operands wider than 64 columns: one launch per tile of 64 columns
            vt, sa, nnz, vp, bp, op = _VTYPE[val.dtype], cfg.struct_addr, lp.nnz, val.data_ptr(), B.data_ptr(), out.data_ptr()
            fn = lib.tsgu_csr_spmm_march
            tr = int(transposed)
            rc = fn(vt, sa, tr, n_rows, nnz, vp, bp, ldb, op, p, ct, index, stream)
            for j in range(ct, p, ct):
                if rc:
                    break
                rc = fn(vt, sa, tr, n_rows, nnz, vp, bp + j * 4, ldb, op + j * 4, p, ct, index, stream)
        else:
            rc = lib.tsgu_csr_spmm_lattice(_VTYPE[val.dtype], cfg.struct_addr, n_rows, lp.nnz, val.data_ptr(), B.data_ptr(), ldb,
                                           out.data_ptr(), p, p, index, stream)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    if tok is not None:
        _timed_end(tok, dev)
    if rc:
        check(rc, "tsgu_csr_spmm_march" if march else "tsgu_csr_spmm_lattice")
    return out


def csr_sddmm_lattice(lp, cfg, R, Cm, alpha: float = 1.0):
    """out[k] = alpha·<R[row k], Cm[col k]> in stored order by the plane sweep (plan kind 0)."""
    lib = _lib or load_library()
    dev = R.device
    if not R.is_cuda or Cm.device != dev:
        require_device(R, Cm)
        raise RuntimeError(f"all operands must be on the same device, got {Cm.device} and {dev}")
    if not R.is_contiguous():
        R = rowmajor(R)
    if not Cm.is_contiguous():
        Cm = rowmajor(Cm)
    p = R.size(-1)
    n_rows = lp.n_rows
    out = torch.empty((lp.nnz,), dtype=R.dtype, device=dev)
    tok = _timed("lattice_sddmm", dev) if KERNEL_EVENTS is not None else None
    index = dev.index
    ctx = None if torch.cuda.current_device() == index else torch.cuda.device(dev)
    if ctx is not None:
        ctx.__enter__()
    try:
        ldr = R.stride(0) if R.size(0) > 1 else max(p, 1)
        ldc = Cm.stride(0) if Cm.size(0) > 1 else max(p, 1)
        stream = torch._C._cuda_getCurrentRawStream(index)
        if getattr(cfg, "march", False):
            ct = cfg.col_tile          # operands wider than 64 columns: the dots of the later column tiles are added to the first
            for j in range(0, p, ct):
                rc = lib.tsgu_csr_sddmm_march(_VTYPE[R.dtype], cfg.struct_addr, n_rows, lp.nnz, R.data_ptr() + j * 4, ldr,
                                              Cm.data_ptr() + j * 4, ldc, out.data_ptr(), float(alpha), int(j > 0), ct, index, stream)
                if rc:
                    break
        else:
            rc = lib.tsgu_csr_sddmm_lattice(_VTYPE[R.dtype], cfg.struct_addr, n_rows, lp.nnz, R.data_ptr(), ldr, Cm.data_ptr(), ldc,
                                            out.data_ptr(), float(alpha), p, index, stream)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    if tok is not None:
        _timed_end(tok, dev)
    if rc:
        check(rc, "tsgu_csr_sddmm_lattice")
    return out


def _tiled_ok(*dense) -> bool:
    return all(t.dim() == 2 and t.data_ptr() % 16 == 0 and (_ld(t) * t.element_size()) % 16 == 0 for t in dense)


def coo_sddmm(row, col, G, B, alpha: float = 1.0):
    if not G.is_cuda and operand_device(row, col, G, B) is not None:
        from . import _cpu

        out = _cpu.coo_sddmm(row, col, G, B)
        return out if alpha == 1.0 else out.mul_(alpha)
    lib = load_library()
    dev = require_device(row, col, G, B)
    if G.dtype != B.dtype:
        raise RuntimeError(f"expected both dense operands to have the same dtype, got {G.dtype} and {B.dtype}")
    G, B = rowmajor(G), rowmajor(B)
    row, col = row.contiguous(), col.contiguous()
    nnz = row.numel()
    out = torch.empty((nnz,), dtype=G.dtype, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.tsgu_coo_sddmm(
                vtype_of(G), itype_of(row), nnz, _p(row), _p(col), _p(G), _ld(G), _p(B), _ld(B), _p(out),
                float(alpha), G.size(-1), dev.index, _stream(dev),
            ),
            "tsgu_coo_sddmm",
        )
    return out


def csr_sptrsm(ptr, idx, val, B, n: int, lower: bool, unit: bool, perm=None, wg_per_cu: int = 1):
    """X = M^{-1} B for the row-gather structure (ptr, idx, [perm], val) of a triangular M.  `wg_per_cu`: persistent workgroups per
    compute unit (1 … 8; speed only, see include/tsgu_hip.h)."""
    lib = load_library()
    dev = require_device(ptr, idx, val, B, perm)
    if val.dtype != B.dtype:
        raise RuntimeError(f"expected A and B to have the same dtype, got {val.dtype} and {B.dtype}")
    if B.dim() != 2 or B.size(0) != n or ptr.numel() != n + 1:
        raise RuntimeError(f"tsgu_csr_sptrsm: a system of {n} rows needs a right-hand side of {n} rows and a row pointer of {n + 1} "
                           f"words, got {tuple(B.shape)} and {ptr.numel()}")
    B, ldb, b_cs = strided2d(B)
    p = B.size(-1)
    ptr, idx, val = ptr.contiguous(), idx.contiguous(), val.contiguous()
    if perm is not None:
        perm = perm.contiguous()
    X = torch.empty((n, p), dtype=B.dtype, device=dev)
    if n == 0 or p == 0:
        return X  # nothing to solve (the kernel would not even initialise its error word)
    work = torch.empty((lib.tsgu_sptrsm_work_bytes(n, p),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.tsgu_csr_sptrsm(
                vtype_of(val), itype_of(ptr), n, idx.numel(), _p(ptr), _p(idx), _p(perm), _p(val),
                int(bool(lower)), int(bool(unit)), _p(B), ldb, b_cs, _p(X), _ld(X), p, _p(work), int(wg_per_cu), dev.index, _stream(dev),
            ),
            "tsgu_csr_sptrsm",
        )
    # error word sits behind the 64 ticket counters (struct TrsmWork in csrc/sptrsm.hip).  It is only ever set by the
    # 4 s device-side wait bound (a dependency that never arrives).  Default: read back before X is handed out (one host
    # sync per solve); TSGU_SPTRSM_CHECK=lazy — and any solve inside a stream capture, where a host read is not allowed —
    # copies it asynchronously to pinned memory and examines it at the next solve / `poll_errors()`.
    _defer_error_check(work[512:516].view(torch.int32), dev)
    return X


_PENDING = []            # (event, pinned int32 slot, what)
_PENDING_LOCK = threading.Lock()
# Default: the error word of a solve is read back before X is handed out (one host sync per solve; a C3-sized solve takes
# milliseconds).  TSGU_SPTRSM_CHECK=lazy defers the check (X is then UNVERIFIED until `poll_errors()` — exported by the
# package — has looked at it: the next solve, wait_for_plans() and interpreter exit call it).
_SYNC_CHECK = os.environ.get("TSGU_SPTRSM_CHECK", "sync") != "lazy"


def _defer_error_check(word: torch.Tensor, dev: torch.device, what: str = "tsgu_csr_sptrsm (dependency wait)") -> None:
    capturing = torch.cuda.is_current_stream_capturing()
    if _SYNC_CHECK and not capturing:
        if int(word.item()) != 0:
            check(-7, what)
        return
    if capturing:
        return      # (a graph replay cannot report through the host; the 4 s device-side bound still ends the wait)
    poll_errors()
    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
    host.copy_(word, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    with _PENDING_LOCK:
        _PENDING.append((ev, host, what))


def _poll_at_exit() -> None:
    try:
        poll_errors(block=True)
    except Exception as exc:  # noqa: BLE001
        import sys

        print(f"torchsparsegradutils_amd: {exc}", file=sys.stderr)


if not _SYNC_CHECK:
    import atexit

    atexit.register(_poll_at_exit)


def poll_errors(block: bool = False) -> None:
    """Raise if a device-side error word of an earlier launch is set.  Non-blocking by default (only completed
    launches are examined); ``block=True`` waits for all of them (tests, end of a run)."""
    with _PENDING_LOCK:
        items = list(_PENDING)
        _PENDING.clear()
    keep, failed = [], None
    for ev, host, what in items:
        if block:
            ev.synchronize()
        if ev.query():
            if int(host.item()) != 0 and failed is None:
                failed = what
        else:
            keep.append((ev, host, what))
    if keep:
        with _PENDING_LOCK:
            _PENDING[:0] = keep
    if failed is not None:
        check(-7, failed)


def index_fingerprint(*tensors: torch.Tensor) -> torch.Tensor:
    """[len(tensors)][2] int64 device tensor: the 128-bit content fingerprint of each (contiguous) index tensor; queued on the
    current stream, nothing is read back here."""
    lib = load_library()
    dev = require_device(*tensors)
    out = torch.zeros((len(tensors), 2), dtype=torch.int64, device=dev)      # (one fill for all the words; the launches accumulate)
    with torch.cuda.device(dev):
        for i, t in enumerate(tensors):
            t = t.contiguous()
            check(lib.tsgu_index_fingerprint(itype_of(t), t.numel(), _p(t), out[i].data_ptr(), 1, dev.index, _stream(dev)),
                  "tsgu_index_fingerprint")
    return out


def index_fingerprint_match(tensors, refs=None, copy: bool = False, hash: bool = True):
    """One pass per index tensor (see tsgu_index_fingerprint_match): returns (words, copies) — `words` a [len(tensors)][3] int64
    device tensor {fingerprint word 0, word 1, non-zero iff the tensor differs from its `refs` entry}, `copies` fresh contiguous
    copies of the tensors (None unless `copy`).  `hash=False` (with `refs`, without `copy`): compare only — the fingerprint words stay 0
    (equal tensors have their reference's fingerprint).  Queued on the current stream, nothing is read back here."""
    lib = load_library()
    dev = require_device(*tensors)
    out = torch.zeros((len(tensors), 3), dtype=torch.int64, device=dev)
    copies = [] if copy else None
    with torch.cuda.device(dev):
        for i, t in enumerate(tensors):
            t = t.contiguous()
            r = None
            if refs is not None:
                r = refs[i]
                if r.dtype != t.dtype or r.numel() != t.numel() or r.device != t.device or not r.is_contiguous():
                    raise ValueError("index_fingerprint_match: a reference tensor does not have the geometry of its index tensor")
            c = torch.empty_like(t) if copy else None
            check(lib.tsgu_index_fingerprint_match(itype_of(t), t.numel(), _p(t), _p(r) if r is not None else None,
                                                   _p(c) if c is not None else None, out[i].data_ptr(), 1 | (0 if hash or r is None or copy else 2), dev.index, _stream(dev)),
                  "tsgu_index_fingerprint_match")
            if copy:
                copies.append(c)
    return out, copies


def coldot(X, Y):
    """Column-wise dot products of two (n, p) arrays -> (p,) tensor (deterministic)."""
    if not X.is_cuda and not Y.is_cuda:
        from . import _cpu

        return _cpu.coldot(X, Y)
    lib = load_library()
    dev = require_device(X, Y)
    if X.dtype != Y.dtype:
        raise RuntimeError(f"tsgu_coldot: expected both operands to have the same dtype, got {X.dtype} and {Y.dtype}")
    X, Y = rowmajor(X), rowmajor(Y)
    n, p = X.shape
    nb = lib.tsgu_coldot_max_blocks(n, p)
    if nb < 0:
        raise RuntimeError("tsgu_coldot: more than 256 right-hand sides are not supported by the fused path")
    partial = torch.empty((nb, p), dtype=X.dtype, device=dev)
    out = torch.empty((p,), dtype=X.dtype, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.tsgu_coldot(vtype_of(X), n, p, _p(X), _ld(X), _p(Y), _ld(Y), _p(partial), _p(out), dev.index,
                            _stream(dev)),
            "tsgu_coldot",
        )
    return out


def device_copy(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst <- src by the library's own 16-bytes-per-lane streaming kernel (the measured HBM ceiling of bench.py)."""
    lib = load_library()
    dev = require_device(src, dst)
    nbytes = src.numel() * src.element_size()
    if dst.numel() * dst.element_size() != nbytes or not (src.is_contiguous() and dst.is_contiguous()):
        raise RuntimeError("device_copy: contiguous tensors of equal byte size expected")
    with torch.cuda.device(dev):
        check(lib.tsgu_device_copy(_p(src), _p(dst), nbytes, dev.index, _stream(dev)), "tsgu_device_copy")


def device_cu_count(index: int = 0) -> int:
    lib = load_library()
    n = _int(0)
    check(lib.tsgu_device_cu_count(index, ctypes.byref(n)), "tsgu_device_cu_count")
    return n.value


def device_info(index: int = 0):
    lib = load_library()
    name = ctypes.create_string_buffer(128)
    ncu, wave = _int(0), _int(0)
    check(lib.tsgu_device_info(index, name, 128, ctypes.byref(ncu), ctypes.byref(wave)), "tsgu_device_info")
    return name.value.decode(), ncu.value, wave.value
