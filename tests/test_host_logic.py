"""CPU-side checks: index helpers bit-exact vs the reference's golden outputs, validation messages,
the C-ABI library exports every symbol the header declares, and the product refuses CPU tensors."""

import ctypes
import os
import re

import numpy as np
import pytest
import torch

import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same_sparse(S, z, prefix):
    if S.layout == torch.sparse_csr:
        assert np.array_equal(S.crow_indices().numpy(), z[prefix + "crow"])
        assert np.array_equal(S.col_indices().numpy(), z[prefix + "col"])
        assert np.array_equal(S.values().numpy(), z[prefix + "val"])
    else:
        assert np.array_equal(S._indices().numpy(), z[prefix + "idx"])
        assert np.array_equal(S._values().numpy(), z[prefix + "val"])


def test_convert_coo_to_csr_bit_exact():
    from torchsparsegradutils_amd.utils import convert_coo_to_csr

    z = G.load("utils_index.npz")
    A = G.sparse_from(z, "c2c_in_", (9, 7))
    _same_sparse(convert_coo_to_csr(A), z, "c2c_out_")
    Ab = G.sparse_from(z, "c2cb_in_", (3, 5, 4))
    _same_sparse(convert_coo_to_csr(Ab), z, "c2cb_out_")
    with pytest.raises(ValueError, match="Unsupported layout"):
        convert_coo_to_csr(torch.eye(3).to_sparse_csr())


@pytest.mark.parametrize("layout", ["coo", "csr"])
def test_block_diag_and_split_bit_exact(layout):
    from torchsparsegradutils_amd.utils import sparse_block_diag, sparse_block_diag_split

    z = G.load("utils_index.npz")
    shapes = [(3, 4), (2, 2), (4, 3)]
    blocks = [G.sparse_from(z, f"bd_{layout}_in{i}_", s) for i, s in enumerate(shapes)]
    D = sparse_block_diag(*blocks)
    assert D.shape == (9, 9)
    _same_sparse(D, z, f"bd_{layout}_out_")
    parts = sparse_block_diag_split(D, *shapes)
    for i, part in enumerate(parts):
        assert part.shape == shapes[i]
        _same_sparse(part, z, f"bd_{layout}_split{i}_")
    assert sparse_block_diag(blocks[0]) is blocks[0]  # single input returned unchanged (utils.py:566-567)
    with pytest.raises(ValueError, match="At least one sparse tensor must be provided."):
        sparse_block_diag()
    with pytest.raises(TypeError):
        sparse_block_diag(blocks[0], 3)
    with pytest.raises(ValueError, match="does not match"):
        sparse_block_diag_split(D, (3, 4), (2, 2))


def test_stack_csr_sparse_eye_and_row_helpers():
    from torchsparsegradutils_amd.utils import sparse_eye, stack_csr
    from torchsparsegradutils_amd.utils.utils import _compress_row_indices, _demcompress_crow_indices, _sort_coo_indices

    a = torch.tensor([[0.0, 1], [2, 0]]).to_sparse_csr()
    b = torch.tensor([[3.0, 0], [0, 4]]).to_sparse_csr()
    s = stack_csr([a, b])
    assert s.shape == (2, 2, 2) and torch.equal(s.to_dense(), torch.stack([a.to_dense(), b.to_dense()]))
    with pytest.raises(ValueError, match="Cannot stack empty list of tensors."):
        stack_csr([])
    with pytest.raises(TypeError):
        stack_csr(a)
    for layout in (torch.sparse_coo, torch.sparse_csr):
        for idt in (torch.int32, torch.int64):
            E = sparse_eye((3, 4, 4), layout=layout, values_dtype=torch.float32, indices_dtype=idt)
            assert torch.equal(E.to_dense(), torch.eye(4).expand(3, 4, 4))
    with pytest.raises(ValueError, match="square"):
        sparse_eye((3, 4))
    crow = torch.tensor([0, 2, 2, 5], dtype=torch.int32)
    rows = _demcompress_crow_indices(crow, 3)
    assert rows.dtype == torch.int32 and rows.tolist() == [0, 0, 2, 2, 2]
    assert torch.equal(_compress_row_indices(rows, 3), crow)
    idx = torch.tensor([[2, 0, 1, 0], [1, 3, 0, 1]])
    srt, perm = _sort_coo_indices(idx)
    ref = torch.sparse_coo_tensor(idx, torch.arange(4.0), (3, 4)).coalesce()
    assert torch.equal(srt, ref.indices()) and torch.equal(perm.float(), ref.values())


def test_validation_messages_match_reference():
    import torchsparsegradutils_amd as m

    E = G.errors()
    A = torch.eye(3).to_sparse_coo()
    B = torch.ones(3, 2)
    cases = {
        "mm_not_tensor": lambda: m.sparse_mm(A, 3),
        "mm_low_dim": lambda: m.sparse_mm(A, torch.ones(3)),
        "mm_dim_mismatch": lambda: m.sparse_mm(A, torch.ones(1, 3, 2)),
        "mm_csc": lambda: m.sparse_mm(torch.eye(3).to_sparse_csc(), B),
        "mm_dense_A": lambda: m.sparse_mm(torch.eye(3), B),
        "mm_sparse_B": lambda: m.sparse_mm(A, B.to_sparse_coo()),
        "mm_batch": lambda: m.sparse_mm(torch.stack([A, A]), torch.ones(3, 3, 2)),
        "mm_inner": lambda: m.sparse_mm(A, torch.ones(4, 2)),
        "tri_not_tensor": lambda: m.sparse_triangular_solve(A, None),
        "tri_low_dim": lambda: m.sparse_triangular_solve(A, torch.ones(3)),
        "tri_dim_mismatch": lambda: m.sparse_triangular_solve(A, torch.ones(1, 3, 2)),
        "tri_csc": lambda: m.sparse_triangular_solve(torch.eye(3).to_sparse_csc(), B),
        "tri_sparse_B": lambda: m.sparse_triangular_solve(A, B.to_sparse_coo()),
        "tri_not_square": lambda: m.sparse_triangular_solve(torch.ones(3, 4).to_sparse_coo(), B),
        "tri_inner": lambda: m.sparse_triangular_solve(A, torch.ones(4, 2)),
        "tri_batch": lambda: m.sparse_triangular_solve(torch.stack([A, A]), torch.ones(3, 3, 2)),
        "gs_not_tensor": lambda: m.sparse_generic_solve(A, 1.0),
        "gs_layout": lambda: m.sparse_generic_solve(torch.eye(3), B),
        "gs_dim": lambda: m.sparse_generic_solve(torch.stack([A, A]), B),
        "gs_square": lambda: m.sparse_generic_solve(torch.ones(3, 4).to_sparse_coo(), B),
        "gs_B_dim": lambda: m.sparse_generic_solve(A, torch.ones(3, 2, 2)),
        "gs_incompatible": lambda: m.sparse_generic_solve(A, torch.ones(4, 2)),
        "gs_B_sparse": lambda: m.sparse_generic_solve(A, B.to_sparse_coo()),
    }
    for name, fn in cases.items():
        want = E[name]
        assert want["type"] in ("ValueError", "TypeError"), name
        with pytest.raises(Exception) as e:
            fn()
        assert type(e.value).__name__ == want["type"], name
        assert str(e.value) == want["msg"], name
    # dtype mismatch only warns (reference sparse_solve.py:398-403); the solver's own product then raises torch's dtype error
    with pytest.warns(UserWarning) as w:
        with pytest.raises(RuntimeError, match="expected scalar type Float but found Double"):
            m.sparse_generic_solve(A, B.double())
    assert E["gs_dtype_warning"]["warnings"] == [str(w[0].message)]


def test_cpu_solvers_run_on_tensor_ops():
    """CPU operands of the Krylov entry points: the recurrences as tensor ops around the ATen product (the fused step kernels are
    the MI355X path and are never entered with a CPU tensor)."""
    from torchsparsegradutils_amd.utils import BICGSTABSettings, LinearCGSettings, MINRESSettings, bicgstab, linear_cg, minres

    A = (2 * torch.eye(3)).to_sparse_csr()
    B = torch.ones(3, 2)
    for fn in (lambda: linear_cg(A, B, settings=LinearCGSettings(cg_tolerance=1e-6)),
               lambda: bicgstab(A, B, settings=BICGSTABSettings()), lambda: minres(A, B, settings=MINRESSettings())):
        x = fn()
        assert not x.is_cuda and torch.allclose(x, torch.full((3, 2), 0.5))


def test_abi_library_exports_every_declared_symbol():
    from torchsparsegradutils_amd import _backend

    header = open(os.path.join(ROOT, "include", "tsgu_hip.h")).read()
    declared = set(re.findall(r"\b(tsgu_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 15
    lib = _backend.load_library()  # loads without a GPU; no compute call is made here
    raw = ctypes.CDLL(_backend.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/tsgu_hip.h but not exported"
        assert name in _backend.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_backend.SIGNATURES) <= declared
    assert lib.tsgu_abi_version() == 7
    # row-pair geometry: (rows per workgroup, entry lanes) per (value type, p); unsupported shapes are refused
    r, e = ctypes.c_int(0), ctypes.c_int(0)
    for vt, p_, want in ((_backend.TSGU_F32, 32, (64, 1)), (_backend.TSGU_F32, 64, (32, 1)), (_backend.TSGU_F32, 16, (64, 2)),
                         (_backend.TSGU_F32, 8, (64, 4)), (_backend.TSGU_BF16, 16, (64, 4)), (_backend.TSGU_BF16, 128, (32, 1))):
        assert lib.tsgu_rowpack_geometry(vt, p_, ctypes.byref(r), ctypes.byref(e), None, None, None) == 0
        assert (r.value, e.value) == want, (vt, p_, r.value, e.value)
    assert lib.tsgu_rowpack_geometry(_backend.TSGU_F64, 32, None, None, None, None, None) != 0
    assert lib.tsgu_rowpack_geometry(_backend.TSGU_F32, 12, None, None, None, None, None) != 0
    assert lib.tsgu_status_string(-2).decode().startswith("bad argument")
    # pure host-side helpers of the ABI
    assert lib.tsgu_spmm_num_blocks(_backend.TSGU_F32, 10 ** 6, 27 * 10 ** 6, 32, 27) == 31250
    # 16-byte dense rows + short sparse rows: one lane per row, 256 rows per workgroup
    assert lib.tsgu_spmm_num_blocks(_backend.TSGU_F32, 2000376, 13907376, 4, 7) == 7814
    # the same operand with long rows keeps 8 entry lanes per row (32 rows per workgroup)
    assert lib.tsgu_spmm_num_blocks(_backend.TSGU_F32, 2000376, 40 * 2000376, 4, 0) == 62512
    # ragged: the same short-row operand with one very long row keeps 8 entry lanes per row as well
    assert lib.tsgu_spmm_num_blocks(_backend.TSGU_F32, 2000376, 13907376, 4, 5000) == 10419   # 32 rows x 6 runs per workgroup
    assert lib.tsgu_sptrsm_work_bytes(10, 1) >= 516


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "torchsparsegradutils_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "liboracle" not in src, f


def test_ctypes_signatures_agree_with_the_header_prototypes():
    """Every prototype of include/tsgu_hip.h against `_backend.SIGNATURES`: the same number of parameters, and for each one the
    same class (pointer / 64-bit integer / int / double) — a mismatch would only show as stack garbage on the GPU box."""
    import ctypes as C

    from torchsparsegradutils_amd import _backend

    header = open(os.path.join(ROOT, "include", "tsgu_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = re.findall(r"\b(?:int|int64_t|const char\s*\*|size_t)\s+(tsgu_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)
    assert len(protos) >= 40

    def klass_of_decl(decl):
        decl = " ".join(decl.split())
        if decl == "void":
            return None
        if "*" in decl:
            return "ptr"
        if re.search(r"\bint64_t\b", decl):
            return "i64"
        if re.search(r"\bdouble\b", decl):
            return "dbl"
        if re.search(r"\b(int|tsgu_vtype|tsgu_itype)\b", decl):
            return "int"
        raise AssertionError(f"unclassified parameter {decl!r}")

    def klass_of_ctype(t):
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "_type_") and not isinstance(t._type_, str):
            return "ptr"
        return {C.c_int64: "i64", C.c_int: "int", C.c_double: "dbl"}[t]

    seen = set()
    for name, params in protos:
        want = [k for k in (klass_of_decl(d) for d in params.split(",")) if k is not None]
        _, argtypes = _backend.SIGNATURES[name]
        got = [klass_of_ctype(t) for t in argtypes]
        assert got == want, (name, got, want)
        seen.add(name)
    assert seen == set(_backend.SIGNATURES)


def test_cpu_tensors_take_the_torch_op_path_and_mixed_devices_raise():
    """INTEGRATION.md, "CPU tensors": operands that live on the CPU are computed by the package's torch-op path (_cpu.py: the ATen
    calls the reference makes; tests/test_cpu_path.py pins it to the reference's golden vectors).  The switch is the operands'
    device only: the HIP bindings themselves refuse a CPU tensor, so nothing computed for a GPU caller can come from the CPU
    (test_product_never_imports_the_oracle keeps the oracle out of the package)."""
    import warnings

    import torchsparsegradutils_amd as m
    from torchsparsegradutils_amd import _backend

    A = (2 * torch.eye(4)).to_sparse_csr()
    B = torch.ones(4, 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for call in (lambda: m.sparse_mm(A, B),
                     lambda: m.sparse_mm((2 * torch.eye(4)).to_sparse_coo(), B),
                     lambda: 4 * m.sparse_triangular_solve(A, B, upper=False),
                     lambda: 4 * m.sparse_generic_solve(A, B, solve=m.utils.linear_cg),
                     lambda: 4 * m.linalg_solve_triangular_compat(A, B, upper=False)):
            out = call()
            assert not out.is_cuda and torch.allclose(out, torch.full((4, 2), 2.0))
    for fn in (_backend.csr_spmm, _backend.csr_sddmm):
        with pytest.raises(RuntimeError, match="gfx950 kernels"):
            fn(A.crow_indices(), A.col_indices(), A.values() if fn is _backend.csr_spmm else B, B, 4, 4)


def test_compat_dense_branch_is_the_reference_dispatch():
    """reference _compat.py:17-34: dense operands go to torch.linalg.solve_triangular, `transpose` as a transposed view with
    `upper` flipped — all eight flag combinations, and the keyword-only signature."""
    import inspect

    from torchsparsegradutils_amd import linalg_solve_triangular_compat

    sig = inspect.signature(linalg_solve_triangular_compat)
    assert [p.kind for p in sig.parameters.values()][2:] == [inspect.Parameter.KEYWORD_ONLY] * 3
    assert sig.parameters["unitriangular"].default is False and sig.parameters["transpose"].default is False
    g = torch.Generator().manual_seed(0)
    T = torch.randn(5, 5, generator=g, dtype=torch.float64) + 5 * torch.eye(5, dtype=torch.float64)
    B = torch.randn(5, 2, generator=g, dtype=torch.float64)
    for upper in (False, True):
        for unit in (False, True):
            for tr in (False, True):
                x = linalg_solve_triangular_compat(T, B, upper=upper, unitriangular=unit, transpose=tr)
                want = torch.linalg.solve_triangular(T.transpose(-2, -1) if tr else T, B, upper=(not upper) if tr else upper, unitriangular=unit)
                assert torch.equal(x, want)


def test_cpp_host_module_of_the_step_is_built_against_the_abi_library():
    """csrc/host/step.cpp (the steady-state sparse_mm step's host path as a torch C++ autograd function) is built by the same
    Makefile, links the C ABI library it launches through, and sparse_matmul picks it up; without a GPU nothing is launched."""
    import torchsparsegradutils_amd.sparse_matmul as sm
    from torchsparsegradutils_amd import _backend

    lib = _backend.load_library()
    assert sm._host is not None, "torchsparsegradutils_amd/_tsgu_host.so is missing: make -C torchsparsegradutils_amd/csrc"
    assert sm._host.abi_version() == lib.tsgu_abi_version()
    assert hasattr(sm._host, "StepPlan") and hasattr(sm._host, "step")
    # CPU tensors never reach it (the documented error of the package comes from the Python path)
    A = torch.eye(4).to_sparse_csr()
    assert sm._step_plan(A, torch.zeros(4, 16)) is None if torch.zeros(1).is_cuda else True
