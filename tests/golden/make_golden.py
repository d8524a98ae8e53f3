"""Generate the golden vectors in this directory by RUNNING THE REAL REFERENCE on CPU.

Run in the build container only (the reference checkout lives at /root/reference there and never
travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Every ``*.npz`` holds the inputs and the reference's outputs of one case family; ``errors.json``
holds the reference's exception types/messages for every validation branch of the three entry
points.  The files are data (inputs + expected outputs), never reference source.
"""

import json
import os
import sys
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))

import torchsparsegradutils as ref  # noqa: E402  (the reference)
from torchsparsegradutils.utils import (  # noqa: E402
    BICGSTABSettings,
    LinearCGSettings,
    MINRESSettings,
    bicgstab,
    convert_coo_to_csr,
    linear_cg,
    minres,
    sparse_block_diag,
    sparse_block_diag_split,
    stack_csr,
)

from torchsparsegradutils_amd.utils import synthetic  # noqa: E402  (our generators: inputs only)

OUT = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore")


def npy(t):
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().copy()
    return t.detach().cpu().numpy().copy()


def sparse_parts(prefix, S, d):
    """store index/value arrays of a sparse tensor (COO or CSR, 2-D or batched)."""
    if S.layout == torch.sparse_csr:
        d[prefix + "crow"] = npy(S.crow_indices())
        d[prefix + "col"] = npy(S.col_indices())
        d[prefix + "val"] = npy(S.values())
    else:
        d[prefix + "idx"] = npy(S._indices())
        d[prefix + "val"] = npy(S._values())
        d[prefix + "coalesced"] = np.array(S.is_coalesced())


def run_mm(A, B, G):
    A = A.detach().clone().requires_grad_(True)
    B = B.detach().clone().requires_grad_(True)
    C = ref.sparse_mm(A, B)
    C.backward(G)
    return C.detach(), A.grad, B.grad


def rand_coo(n, m, nnz, dtype, gen):
    flat = torch.randperm(n * m, generator=gen)[:nnz]
    idx = torch.stack((flat // m, flat % m))
    val = torch.randn(nnz, dtype=dtype, generator=gen)
    return torch.sparse_coo_tensor(idx, val, (n, m)).coalesce()


# ---------------------------------------------------------------------------------------------
def case_c1():
    """G1: BASELINE config C1 exactly (COO 4096², 1 % density, 16 RHS, fp32)."""
    g = torch.Generator().manual_seed(0)
    A = rand_coo(4096, 4096, 167772, torch.float32, g)
    B = torch.randn(4096, 16, generator=g)
    G = torch.rand(4096, 16, generator=g)
    C, gA, gB = run_mm(A, B, G)
    assert gA.layout == torch.sparse_coo and gA._nnz() == 167772
    d = {
        "rows": npy(A.indices()[0]).astype(np.uint16),
        "cols": npy(A.indices()[1]).astype(np.uint16),
        "val": npy(A.values()),
        "B": npy(B),
        "G": npy(G),
        "C": npy(C),
        "gradA_val": npy(gA._values()),
        "gradA_idx_equal_input": np.array(torch.equal(gA._indices(), A.indices())),
        "gradB": npy(gB),
    }
    np.savez_compressed(os.path.join(OUT, "mm_c1_coo.npz"), **d)


def case_mm_small():
    """G2/G3: layouts × index dtypes × value dtypes, rectangular, batched test shapes."""
    d = {}
    g = torch.Generator().manual_seed(1)
    names = []
    for vd, vn in ((torch.float32, "f32"), (torch.float64, "f64")):
        # 2-D rectangular, p not a multiple of the vector width
        A = rand_coo(48, 40, 300, vd, g)
        B = torch.randn(40, 10, dtype=vd, generator=g)
        G = torch.randn(48, 10, dtype=vd, generator=g)
        for layout in ("coo", "csr"):
            for idt, iname in ((torch.int32, "i32"), (torch.int64, "i64")):
                if layout == "coo":
                    if idt == torch.int32:
                        continue  # COO indices are always int64 in torch
                    Ain = A
                else:
                    Ac = A.to_sparse_csr()
                    Ain = torch.sparse_csr_tensor(
                        Ac.crow_indices().to(idt), Ac.col_indices().to(idt), Ac.values(), Ac.shape
                    )
                name = f"r2_{layout}_{iname}_{vn}_"
                C, gA, gB = run_mm(Ain, B, G)
                sparse_parts(name + "A_", Ain, d)
                d[name + "B"], d[name + "G"], d[name + "C"], d[name + "gradB"] = npy(B), npy(G), npy(C), npy(gB)
                sparse_parts(name + "gradA_", gA, d)
                names.append(name)
        # batched shapes used by the reference tests (test_sparse_matmul.py:16-24)
        for b, n, m, p in ((4, 8, 16, 10), (11, 7, 4, 9)):
            nnz = max(4, n * m // 4)
            items = [rand_coo(n, m, nnz, vd, g) for _ in range(b)]
            B = torch.randn(b, m, p, dtype=vd, generator=g)
            G = torch.randn(b, n, p, dtype=vd, generator=g)
            Acoo = torch.stack(items)
            Acsr = stack_csr([t.to_sparse_csr() for t in items])
            for layout, Ain in (("coo", Acoo), ("csr", Acsr)):
                name = f"b{b}_{layout}_{vn}_"
                C, gA, gB = run_mm(Ain, B, G)
                sparse_parts(name + "A_", Ain, d)
                d[name + "B"], d[name + "G"], d[name + "C"], d[name + "gradB"] = npy(B), npy(G), npy(C), npy(gB)
                sparse_parts(name + "gradA_", gA, d)
                names.append(name)
    # un-coalesced 2-D COO: duplicates keep their own gradient entry (sparse_matmul.py:185,209)
    idx = torch.tensor([[0, 2, 0, 1, 2, 0], [1, 0, 1, 2, 0, 3]])
    val = torch.randn(6, generator=g)
    Au = torch.sparse_coo_tensor(idx, val, (3, 4))
    B = torch.randn(4, 5, generator=g)
    G = torch.randn(3, 5, generator=g)
    C, gA, gB = run_mm(Au, B, G)
    name = "uncoal_"
    sparse_parts(name + "A_", Au, d)
    d[name + "B"], d[name + "G"], d[name + "C"], d[name + "gradB"] = npy(B), npy(G), npy(C), npy(gB)
    sparse_parts(name + "gradA_", gA, d)
    names.append(name)
    d["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "mm_small.npz"), **d)


def case_mm_stencil():
    """G4: scaled-down C2 — periodic 27-point stencil on 12³ (N=1728), 32 RHS, fp32 + fp64 twin.
    Inputs are stored once in fp64; the fp32 case uses their fp32 roundings."""
    d = {}
    n = 12 ** 3
    crow, col = synthetic.stencil27_periodic(12, 12, 12, torch.int32)
    g = torch.Generator().manual_seed(2)
    val64 = torch.randn(col.numel(), dtype=torch.float64, generator=g)
    B64 = torch.randn(n, 32, dtype=torch.float64, generator=g)
    G64 = torch.randn(n, 32, dtype=torch.float64, generator=g)
    d["val64"], d["B64"], d["G64"] = npy(val64), npy(B64), npy(G64)
    for vd, vn in ((torch.float32, "f32"), (torch.float64, "f64")):
        A = torch.sparse_csr_tensor(crow, col, val64.to(vd), (n, n))
        C, gA, gB = run_mm(A, B64.to(vd), G64.to(vd))
        assert gA.crow_indices().dtype == torch.int32
        assert torch.equal(gA.crow_indices(), crow) and torch.equal(gA.col_indices(), col)
        d[vn + "_C"], d[vn + "_gradA_val"], d[vn + "_gradB"] = npy(C), npy(gA.values()), npy(gB)
    np.savez_compressed(os.path.join(OUT, "mm_stencil27_12.npz"), **d)


def tri_matrix(n, upper, unit, gen, dtype, density=0.25):
    M = torch.randn(n, n, dtype=dtype, generator=gen) * 0.3
    M = M * (torch.rand(n, n, generator=gen) < density)
    M = torch.triu(M, 1) if upper else torch.tril(M, -1)
    if not unit:
        M = M + torch.diag(1.0 + torch.rand(n, dtype=dtype, generator=gen))
    return M


def case_triangular():
    """G5: all 2³ flag combinations × {COO, CSR} × {2-D, batched}; plus a structured lower factor."""
    d = {}
    names = []
    g = torch.Generator().manual_seed(3)
    for vd, vn in ((torch.float32, "f32"), (torch.float64, "f64")):
        for upper in (False, True):
            for unit in (False, True):
                for transpose in (False, True):
                    for batched in (False, True):
                        n, p = (12, 6) if batched else (64, 6)
                        if batched:
                            Ms = [tri_matrix(n, upper, unit, g, vd, 0.4) for _ in range(4)]
                            B = torch.randn(4, n, p, dtype=vd, generator=g)
                            G = torch.randn(4, n, p, dtype=vd, generator=g)
                            # equal nnz per item is required for batched CSR: reuse one pattern
                            mask = Ms[0] != 0
                            Ms = [torch.where(mask, torch.where(M != 0, M, torch.full_like(M, 0.05)), torch.zeros_like(M)) for M in Ms]
                        else:
                            M = tri_matrix(n, upper, unit, g, vd)
                            B = torch.randn(n, p, dtype=vd, generator=g)
                            G = torch.randn(n, p, dtype=vd, generator=g)
                        for layout in ("coo", "csr"):
                            if batched:
                                if layout == "coo":
                                    A = torch.stack([M.to_sparse_coo() for M in Ms])
                                else:
                                    A = stack_csr([M.to_sparse_csr() for M in Ms])
                            else:
                                A = M.to_sparse_coo() if layout == "coo" else M.to_sparse_csr()
                            A = A.detach().requires_grad_(True)
                            Bq = B.clone().requires_grad_(True)
                            x = ref.sparse_triangular_solve(A, Bq, upper=upper, unitriangular=unit, transpose=transpose)
                            x.backward(G)
                            name = f"{vn}_{'b' if batched else 's'}_{layout}_u{int(upper)}_d{int(unit)}_t{int(transpose)}_"
                            sparse_parts(name + "A_", A.detach(), d)
                            d[name + "B"], d[name + "G"], d[name + "x"], d[name + "gradB"] = npy(B), npy(G), npy(x), npy(Bq.grad)
                            sparse_parts(name + "gradA_", A.grad, d)
                            names.append(name)
    d["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "tri_flags.npz"), **d)

    # structured: lower triangle of the periodic 27-pt stencil on 8³ (N=512), fp32/int32, 8 RHS
    d = {}
    crow, col = synthetic.stencil27_periodic(8, 8, 8, torch.int32)
    g = torch.Generator().manual_seed(4)
    val = torch.rand(col.numel(), generator=g) * 0.1
    crow, col, val = synthetic.lower_of(crow, col, val)
    rows = torch.repeat_interleave(torch.arange(512, dtype=torch.int32), crow[1:] - crow[:-1])
    val = torch.where(rows == col, 1.0 + torch.rand(col.numel(), generator=g), val)
    B = torch.randn(512, 8, generator=g)
    G = torch.randn(512, 8, generator=g)
    for transpose in (False, True):
        A = torch.sparse_csr_tensor(crow, col, val, (512, 512)).requires_grad_(True)
        Bq = B.clone().requires_grad_(True)
        x = ref.sparse_triangular_solve(A, Bq, upper=False, transpose=transpose)
        x.backward(G)
        t = f"t{int(transpose)}_"
        d[t + "x"], d[t + "gradA_val"], d[t + "gradB"] = npy(x), npy(A.grad.values()), npy(Bq.grad)
        assert A.grad.crow_indices().dtype == torch.int32
    d["crow"], d["col"], d["val"], d["B"], d["G"] = npy(crow), npy(col), npy(val), npy(B), npy(G)
    np.savez_compressed(os.path.join(OUT, "tri_stencil_lower.npz"), **d)


def case_cg():
    """G7: CG on the 7-point Laplacian 16³, 4 RHS — iterates after k iterations, final solution,
    iteration count, and gradients through sparse_generic_solve."""
    d = {}
    crow, col, val = synthetic.laplacian7(16, 16, 16, torch.int32, torch.float32)
    n = 4096
    g = torch.Generator().manual_seed(5)
    B = torch.randn(n, 4, generator=g)
    G = torch.randn(n, 4, generator=g)
    A = torch.sparse_csr_tensor(crow, col, val, (n, n))
    d["crow"], d["col"], d["val"], d["B"], d["G"] = npy(crow), npy(col), npy(val), npy(B), npy(G)
    for vd, vn in ((torch.float32, "f32"), (torch.float64, "f64")):
        Av = torch.sparse_csr_tensor(crow, col, val.to(vd), (n, n))
        for k in (1, 5, 11, 20):
            x = linear_cg(Av, B.to(vd), max_tridiag_iter=min(k, 20),
                          settings=LinearCGSettings(max_cg_iterations=k, cg_tolerance=1e-30))
            d[f"{vn}_iter{k}"] = npy(x)
        calls = [0]

        def mv(v, Av=Av):
            calls[0] += 1
            return Av.matmul(v)

        x = linear_cg(mv, B.to(vd), settings=LinearCGSettings(max_cg_iterations=1000, cg_tolerance=1e-6))
        d[f"{vn}_final"] = npy(x)
        d[f"{vn}_iters"] = np.array(calls[0] - 1)  # one SpMM is the initial residual (linear_cg.py:266)
    # gradients through sparse_generic_solve with CG (fp64 so that the comparison is tight)
    Ag = torch.sparse_csr_tensor(crow, col, val.double(), (n, n)).requires_grad_(True)
    Bq = B.double().clone().requires_grad_(True)
    st = LinearCGSettings(max_cg_iterations=2000, cg_tolerance=1e-12)
    x = ref.sparse_generic_solve(Ag, Bq, solve=linear_cg, settings=st)
    x.backward(G.double())
    d["gs_x"], d["gs_gradA_val"], d["gs_gradB"] = npy(x), npy(Ag.grad.values()), npy(Bq.grad)
    np.savez_compressed(os.path.join(OUT, "cg_lap16.npz"), **d)


def case_generic_small():
    """Small SPD / non-symmetric systems through every solver (reference test_sparse_solve.py shapes)."""
    d = {}
    g = torch.Generator().manual_seed(6)
    n = 12
    M = torch.randn(n, n, dtype=torch.float64, generator=g) * (torch.rand(n, n, generator=g) < 0.3)
    S = M @ M.t() + n * torch.eye(n, dtype=torch.float64)
    S = S * (S.abs() > 1e-12)
    names = []
    for layout in ("coo", "csr"):
        for bshape in ((n,), (n, 1), (n, 6)):
            B = torch.randn(*bshape, dtype=torch.float64, generator=g)
            G = torch.randn(*bshape, dtype=torch.float64, generator=g)
            for sname, solver, kw in (
                ("cg", linear_cg, {"settings": LinearCGSettings(cg_tolerance=1e-12)}),
                ("bicgstab", bicgstab, {"settings": BICGSTABSettings(reltol=1e-12, abstol=1e-14)}),
                ("minres", minres, {"settings": MINRESSettings(minres_tolerance=1e-12)}),
                ("default", None, {}),
            ):
                A = (S.to_sparse_coo() if layout == "coo" else S.to_sparse_csr()).requires_grad_(True)
                Bq = B.clone().requires_grad_(True)
                x = ref.sparse_generic_solve(A, Bq, solve=solver, **kw)
                x.backward(G)
                name = f"{layout}_{len(bshape)}d{bshape[-1]}_{sname}_"
                d[name + "B"], d[name + "G"], d[name + "x"], d[name + "gradB"] = npy(B), npy(G), npy(x), npy(Bq.grad)
                sparse_parts(name + "gradA_", A.grad, d)
                names.append(name)
    d["S"] = npy(S)
    # G8: non-symmetric tridiagonal + explicit transpose solver (test_sparse_solve.py:75-121)
    n = 16
    T = torch.diag(torch.full((n,), 4.0, dtype=torch.float64)) + torch.diag(torch.full((n - 1,), -1.0, dtype=torch.float64), 1) \
        + torch.diag(torch.full((n - 1,), -2.0, dtype=torch.float64), -1)
    B = torch.randn(n, 3, dtype=torch.float64, generator=g)
    G = torch.randn(n, 3, dtype=torch.float64, generator=g)
    st = BICGSTABSettings(reltol=1e-13, abstol=1e-15)

    def bic(A, b, **kw):
        return bicgstab(A, b, settings=st)

    def bic_t(A, b, **kw):
        return bicgstab(A.t().to_sparse_csr() if A.layout == torch.sparse_csr else A.t(), b, settings=st)

    A = T.to_sparse_csr().requires_grad_(True)
    Bq = B.clone().requires_grad_(True)
    x = ref.sparse_generic_solve(A, Bq, solve=bic, transpose_solve=bic_t)
    x.backward(G)
    d["nonsym_T"], d["nonsym_B"], d["nonsym_G"] = npy(T), npy(B), npy(G)
    d["nonsym_x"], d["nonsym_gradB"], d["nonsym_gradA_val"] = npy(x), npy(Bq.grad), npy(A.grad.values())
    # plain bicgstab iterates on a CSR operator, fp32, default settings
    A32 = T.float().to_sparse_csr()
    b32 = B[:, 0].float()
    d["bicg32_x"] = npy(bicgstab(A32, b32))
    d["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "generic_small.npz"), **d)


def case_bf16():
    """G9: bf16.  The reference's CSR path cannot run bf16 on CPU (addmm_out_sparse_csr_impl_mkl not
    implemented for BFloat16), so the CSR oracle is the reference in fp32 on bf16-rounded inputs; the
    reference's own COO-bf16 output is stored next to it."""
    d = {}
    crow, col = synthetic.stencil27_periodic(8, 8, 8, torch.int32)
    g = torch.Generator().manual_seed(7)
    val = torch.randn(col.numel(), generator=g).bfloat16()
    B = torch.randn(512, 16, generator=g).bfloat16()
    G = torch.randn(512, 16, generator=g).bfloat16()
    A32 = torch.sparse_csr_tensor(crow, col, val.float(), (512, 512))
    C, gA, gB = run_mm(A32, B.float(), G.float())
    d["crow"], d["col"], d["val_bf16"], d["B_bf16"], d["G_bf16"] = npy(crow), npy(col), npy(val), npy(B), npy(G)
    d["C_f32"], d["gradA_f32"], d["gradB_f32"] = npy(C), npy(gA.values()), npy(gB)
    try:
        Acoo = torch.sparse_csr_tensor(crow, col, val.float(), (512, 512)).to_sparse_coo().to(torch.bfloat16)
        Cb = ref.sparse_mm(Acoo, B)
        d["C_coo_bf16"] = npy(Cb)
    except Exception as e:  # pragma: no cover
        print("COO bf16 not runnable:", e)
    try:
        ref.sparse_mm(torch.sparse_csr_tensor(crow, col, val, (512, 512)), B)
        d["csr_bf16_runs"] = np.array(True)
    except Exception:
        d["csr_bf16_runs"] = np.array(False)
    np.savez_compressed(os.path.join(OUT, "bf16_stencil.npz"), **d)


def case_utils():
    """Index helpers: bit-exact outputs of the reference's CSR/COO plumbing."""
    d = {}
    g = torch.Generator().manual_seed(8)
    A = rand_coo(9, 7, 20, torch.float64, g)
    perm = torch.randperm(20, generator=g)
    Au = torch.sparse_coo_tensor(A.indices()[:, perm], A.values()[perm], A.shape)  # unsorted, no duplicates
    C = convert_coo_to_csr(Au)
    sparse_parts("c2c_in_", Au, d)
    sparse_parts("c2c_out_", C, d)
    items = [rand_coo(5, 4, 8, torch.float64, g) for _ in range(3)]
    Ab = torch.stack(items)
    Cb = convert_coo_to_csr(Ab)
    sparse_parts("c2cb_in_", Ab, d)
    sparse_parts("c2cb_out_", Cb, d)
    blocks = [rand_coo(3, 4, 5, torch.float64, g), rand_coo(2, 2, 3, torch.float64, g), rand_coo(4, 3, 6, torch.float64, g)]
    for layout in ("coo", "csr"):
        bl = blocks if layout == "coo" else [b.to_sparse_csr() for b in blocks]
        D = sparse_block_diag(*bl)
        for i, b in enumerate(bl):
            sparse_parts(f"bd_{layout}_in{i}_", b, d)
        sparse_parts(f"bd_{layout}_out_", D, d)
        parts = sparse_block_diag_split(D, (3, 4), (2, 2), (4, 3))
        for i, b in enumerate(parts):
            sparse_parts(f"bd_{layout}_split{i}_", b, d)
    np.savez_compressed(os.path.join(OUT, "utils_index.npz"), **d)


def case_errors():
    """G6/G10: exception type + message of every validation branch."""
    out = {}
    A = torch.eye(3).to_sparse_coo()
    B = torch.ones(3, 2)

    def rec(name, fn):
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            try:
                fn()
                out[name] = {"type": "None", "msg": ""}
            except Exception as e:  # noqa: BLE001
                out[name] = {"type": type(e).__name__, "msg": str(e)}
            out[name]["warnings"] = [str(x.message) for x in w if issubclass(x.category, UserWarning)
                                     and "Sparse CSR tensor support" not in str(x.message)]

    rec("mm_not_tensor", lambda: ref.sparse_mm(A, 3))
    rec("mm_low_dim", lambda: ref.sparse_mm(A, torch.ones(3)))
    rec("mm_dim_mismatch", lambda: ref.sparse_mm(A, torch.ones(1, 3, 2)))
    rec("mm_csc", lambda: ref.sparse_mm(torch.eye(3).to_sparse_csc(), B))
    rec("mm_dense_A", lambda: ref.sparse_mm(torch.eye(3), B))
    rec("mm_sparse_B", lambda: ref.sparse_mm(A, B.to_sparse_coo()))
    rec("mm_batch", lambda: ref.sparse_mm(torch.stack([A, A]), torch.ones(3, 3, 2)))
    rec("mm_inner", lambda: ref.sparse_mm(A, torch.ones(4, 2)))
    rec("tri_not_tensor", lambda: ref.sparse_triangular_solve(A, None))
    rec("tri_low_dim", lambda: ref.sparse_triangular_solve(A, torch.ones(3)))
    rec("tri_dim_mismatch", lambda: ref.sparse_triangular_solve(A, torch.ones(1, 3, 2)))
    rec("tri_csc", lambda: ref.sparse_triangular_solve(torch.eye(3).to_sparse_csc(), B))
    rec("tri_sparse_B", lambda: ref.sparse_triangular_solve(A, B.to_sparse_coo()))
    rec("tri_not_square", lambda: ref.sparse_triangular_solve(torch.ones(3, 4).to_sparse_coo(), B))
    rec("tri_inner", lambda: ref.sparse_triangular_solve(A, torch.ones(4, 2)))
    rec("tri_batch", lambda: ref.sparse_triangular_solve(torch.stack([A, A]), torch.ones(3, 3, 2)))

    def unit_with_diag():
        L = (torch.tril(torch.ones(3, 3))).to_sparse_csr().requires_grad_(True)
        x = ref.sparse_triangular_solve(L, B.clone().requires_grad_(True), upper=False, unitriangular=True)
        x.sum().backward()

    rec("tri_unit_with_diag_backward", unit_with_diag)
    rec("gs_not_tensor", lambda: ref.sparse_generic_solve(A, 1.0))
    rec("gs_layout", lambda: ref.sparse_generic_solve(torch.eye(3), B))
    rec("gs_dim", lambda: ref.sparse_generic_solve(torch.stack([A, A]), B))
    rec("gs_square", lambda: ref.sparse_generic_solve(torch.ones(3, 4).to_sparse_coo(), B))
    rec("gs_B_dim", lambda: ref.sparse_generic_solve(A, torch.ones(3, 2, 2)))
    rec("gs_incompatible", lambda: ref.sparse_generic_solve(A, torch.ones(4, 2)))
    rec("gs_B_sparse", lambda: ref.sparse_generic_solve(A, B.to_sparse_coo()))
    rec("gs_dtype_warning", lambda: ref.sparse_generic_solve(A, B.double()))
    with open(os.path.join(OUT, "errors.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    for fn in (case_c1, case_mm_small, case_mm_stencil, case_triangular, case_cg, case_generic_small, case_bf16,
               case_utils, case_errors):
        fn()
        print("wrote", fn.__name__)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
