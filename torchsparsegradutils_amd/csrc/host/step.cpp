// Host side of the steady-state sparse_mm step in C++: `SparseMatMul.forward / backward` (torchsparsegradutils_amd/sparse_matmul.py,
// reference sparse_matmul.py:132-234) for a CSR pattern whose launches are FINAL (plane march / plane sweep / row-block tiles /
// plan-free kernels; 2-D, coalesced COO, or batched CSR).
//
// Why: a forward + backward step is three kernel launches (~0.23 ms of GPU time at C2).  The Python host path — autograd.Function,
// plan look-ups, ctypes marshalling, the engine's hand-over to a thread that must take the GIL — costs 0.08 ms per step on a fast
// host and 0.25 ms on a slow one, where the step then waits for Python, not for HBM.  This file is the same sequence of calls
// (allocate the result, launch through the C ABI of include/tsgu_hip.h on torch's current stream, rebuild the sparse gradient),
// as a torch::autograd::Function whose backward runs on the engine thread without the interpreter.  No kernel lives here and
// nothing is decided here: Python builds a StepPlan once the pattern's configurations are settled and keeps every table alive.
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/tsgu_hip.h"

namespace {

struct Product {          // one of the three products of a step (structured patterns; plan-free steps carry their arrays in StepPlan)
    int kind = 0;         // 0 plane march, 1 plane sweep, 2 plan-free gather kernels, 3 row-block tiles
    std::string blob;     // a copy of the tsgu_march_plan / tsgu_lattice_plan / tsgu_tile_plan the Python object built (plain struct of
                          // sizes + device pointers; the tables it points to are the tensors the StepPlan holds)
    int transposed = 0;   // march: walk the transposed pattern (gradB)
    const void* plan() const { return blob.data(); }
};

struct StepPlan {
    at::Tensor crow, col;     // A's index tensors (the gradient's)
    int64_t n_rows = 0, n_cols = 0, nnz = 0, p = 0;
    int vtype = 0, device = 0;
    Product fwd, sddmm, spmm_t;
    // plan-free steps (kind 2 in all three): the transposed pattern (row pointer over A's columns, A's row of each entry, position in
    // A's value array), the longest row (a hint of tsgu_csr_spmm) and whether both gradients can come from ONE walk
    at::Tensor t_ptr, t_idx, t_perm;
    int64_t max_row_nnz = 0, t_max_row_nnz = 0;
    int itype = 0, fused_backward = 0;
    // layout of A: CSR (crow / col above) or 2-D coalesced COO (`indices`; crow / col are then the CSR arrays derived from it)
    int coo = 0;
    at::Tensor indices;
    // batched CSR (torch layout: crow [b][n+1], col / values [b][nnz]; equal nnz per item): the launches see the block-diagonal 2-D
    // problem (n_rows = b·n …, what the Python path hands the structured kernels: _pattern.flat_of), the tensors keep their batch shape
    int64_t batch = 0, item_rows = 0, item_cols = 0, item_nnz = 0;
    std::vector<at::Tensor> tables;   // every device table the three plan structs point into: no Python object is owned here, so the
                                      // last reference may go away on the autograd engine's thread (with a graph node) without the GIL
};
using StepPlanPtr = std::shared_ptr<StepPlan>;

void check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + " failed: " + tsgu_status_string(rc));
}

void* stream_of(int device) { return static_cast<void*>(c10::hip::getCurrentHIPStream(static_cast<c10::DeviceIndex>(device)).stream()); }

void spmm(const StepPlan& s, const Product& pr, int64_t rows_out, const at::Tensor& val, const at::Tensor& dense, at::Tensor& out) {
    const int64_t ld = s.p;      // (contiguous operands only: checked by step() / made so in backward)
    if (pr.kind == 3)      // (a plan with `perm` walks the transposed pattern through A's own values)
        check(tsgu_csr_spmm_tile(s.vtype, static_cast<const tsgu_tile_plan*>(pr.plan()), val.data_ptr(), dense.data_ptr(), ld, out.data_ptr(), s.p,
                                 s.p, s.device, stream_of(s.device)),
              "tsgu_csr_spmm_tile");
    else if (pr.kind == 0)
        check(tsgu_csr_spmm_march(s.vtype, static_cast<const tsgu_march_plan*>(pr.plan()), pr.transposed, rows_out, s.nnz, val.data_ptr(),
                                  dense.data_ptr(), ld, out.data_ptr(), s.p, s.p, s.device, stream_of(s.device)),
              "tsgu_csr_spmm_march");
    else
        check(tsgu_csr_spmm_lattice(s.vtype, static_cast<const tsgu_lattice_plan*>(pr.plan()), rows_out, s.nnz, val.data_ptr(),
                                    dense.data_ptr(), ld, out.data_ptr(), s.p, s.p, s.device, stream_of(s.device)),
              "tsgu_csr_spmm_lattice");
}

bool plain(const at::Tensor& t, int64_t rows, int64_t p) {
    return t.defined() && t.dim() == 2 && t.size(0) == rows && t.size(1) == p && t.is_contiguous() &&
           reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0;
}

// a dense operand of a batched step: (b, rows, p), contiguous, 16-byte aligned — the same memory as the (b·rows, p) operand of the flat problem
bool plain3(const at::Tensor& t, int64_t b, int64_t rows, int64_t p) {
    return t.defined() && t.dim() == 3 && t.size(0) == b && t.size(1) == rows && t.size(2) == p && t.is_contiguous() &&
           reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0;
}

class StepFunction : public torch::autograd::Function<StepFunction> {
   public:
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& A, const at::Tensor& B, StepPlanPtr handle) {
        const StepPlan& s = *handle;
        const c10::hip::HIPGuard on_device(static_cast<c10::DeviceIndex>(s.device));   // (the launchers select the device: restore the caller's)
        // torch keeps the value tensor a sparse tensor was built from as given, strided views included (a column of a 2-D parameter):
        // the launches below take a raw pointer to nnz consecutive values (the Python path: `val.contiguous()` in _backend.py)
        at::Tensor val = s.coo ? A._values() : A.values();
        if (!val.is_contiguous()) val = val.contiguous();
        TORCH_CHECK(val.numel() == s.nnz && val.dim() == (s.batch ? 2 : 1), "step plan of another matrix (number of stored values)");
        at::Tensor C = s.batch ? at::empty({s.batch, s.item_rows, s.p}, B.options()) : at::empty({s.n_rows, s.p}, B.options());
        if (s.fwd.kind == 2 && s.batch)
            // batched operands off a lattice: the plan-free kernels take the batch as it is (item strides; crow [b][n+1], col / val [b][nnz])
            check(tsgu_csr_spmm(s.vtype, s.itype, s.item_rows, s.item_cols, s.item_nnz, s.crow.data_ptr(), s.col.data_ptr(), val.data_ptr(), nullptr,
                                B.data_ptr(), s.p, 1, s.item_cols * s.p, C.data_ptr(), s.p, 1, s.item_rows * s.p, s.p, s.batch, s.max_row_nnz, nullptr, 0,
                                nullptr, s.device, stream_of(s.device)),
                  "tsgu_csr_spmm");
        else if (s.fwd.kind == 2)
            check(tsgu_csr_spmm(s.vtype, s.itype, s.n_rows, s.n_cols, s.nnz, s.crow.data_ptr(), s.col.data_ptr(), val.data_ptr(), nullptr,
                                B.data_ptr(), B.size(0) > 1 ? B.stride(0) : s.p, 1, 0, C.data_ptr(), s.p, 1, 0, s.p, 1, s.max_row_nnz, nullptr, 0,
                                nullptr, s.device, stream_of(s.device)),
                  "tsgu_csr_spmm");
        else
            spmm(s, s.fwd, s.n_rows, val, B, C);
        ctx->save_for_backward({val, B});
        // the gradient carries THIS operand's index tensors (reference sparse_matmul.py:208-219) — the plan may have been adopted from
        // an earlier tensor with the same content (pattern cache: content fingerprint), whose tensors the kernels may read instead
        if (s.coo) ctx->saved_data["idx"] = A._indices();
        else ctx->saved_data["crow"] = A.crow_indices(), ctx->saved_data["col"] = A.col_indices();
        ctx->saved_data["plan"] = c10::IValue(reinterpret_cast<int64_t>(&s));
        // the plan (and with it every table the launches read) lives as long as the graph node: an empty tensor whose deleter owns
        // a reference travels with the node's saved data
        auto* owner = new StepPlanPtr(std::move(handle));
        ctx->saved_data["keep"] = c10::IValue(at::from_blob(owner, {0}, [owner](void*) { delete owner; }, at::TensorOptions().dtype(at::kByte)));
        return C;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const StepPlan& s = *reinterpret_cast<const StepPlan*>(ctx->saved_data["plan"].toInt());
        const c10::hip::HIPGuard on_device(static_cast<c10::DeviceIndex>(s.device));
        const auto saved = ctx->get_saved_variables();
        const at::Tensor& val = saved[0];
        const at::Tensor& B = saved[1];
        at::Tensor G = grads[0];
        at::Tensor gradA, gradB;
        if (!G.defined()) return {gradA, gradB, at::Tensor()};
        if (s.batch ? !plain3(G, s.batch, s.item_rows, s.p) : !plain(G, s.n_rows, s.p)) G = G.contiguous();
        if (reinterpret_cast<uintptr_t>(G.data_ptr()) % 16 != 0) G = G.clone();
        const bool need_a = ctx->needs_input_grad(0), need_b = ctx->needs_input_grad(1);
        const int64_t ldb = s.batch ? s.p : (B.size(0) > 1 ? B.stride(0) : s.p);
        at::Tensor gv;
        if (s.fwd.kind == 2 && s.batch) {
            // the same launches with item sizes and batch strides (the transposed pattern is per item: t_ptr [b][m+1], t_idx / t_perm [b][nnz])
            const int64_t gs = s.item_rows * s.p, bs = s.item_cols * s.p;
            if (need_a && need_b && s.fused_backward) {
                gv = at::empty({s.batch, s.item_nnz}, val.options());
                gradB = at::empty({s.batch, s.item_cols, s.p}, G.options());
                check(tsgu_csr_mm_backward(s.vtype, s.itype, s.item_rows, s.item_cols, s.item_nnz, s.t_ptr.data_ptr(), s.t_idx.data_ptr(),
                                           s.t_perm.data_ptr(), val.data_ptr(), G.data_ptr(), s.p, gs, B.data_ptr(), s.p, bs, gv.data_ptr(),
                                           gradB.data_ptr(), s.p, bs, s.p, s.batch, s.device, stream_of(s.device)),
                      "tsgu_csr_mm_backward");
            } else {
                if (need_a) {
                    gv = at::empty({s.batch, s.item_nnz}, val.options());
                    check(tsgu_csr_sddmm(s.vtype, s.itype, s.item_rows, s.item_cols, s.item_nnz, s.crow.data_ptr(), s.col.data_ptr(), G.data_ptr(), s.p,
                                         gs, B.data_ptr(), s.p, bs, gv.data_ptr(), 1.0, 0, s.p, s.batch, s.device, stream_of(s.device)),
                          "tsgu_csr_sddmm");
                }
                if (need_b) {
                    gradB = at::empty({s.batch, s.item_cols, s.p}, G.options());
                    check(tsgu_csr_spmm(s.vtype, s.itype, s.item_cols, s.item_rows, s.item_nnz, s.t_ptr.data_ptr(), s.t_idx.data_ptr(), val.data_ptr(),
                                        s.t_perm.data_ptr(), G.data_ptr(), s.p, 1, gs, gradB.data_ptr(), s.p, 1, bs, s.p, s.batch, s.t_max_row_nnz,
                                        nullptr, 0, nullptr, s.device, stream_of(s.device)),
                          "tsgu_csr_spmm");
                }
            }
        } else if (s.fwd.kind == 2) {
            if (need_a && need_b && s.fused_backward) {
                // both gradients in one pass over the transposed pattern: every upstream row is gathered once (reference :172-229)
                gv = at::empty({s.nnz}, val.options());
                gradB = at::empty({s.n_cols, s.p}, G.options());
                check(tsgu_csr_mm_backward(s.vtype, s.itype, s.n_rows, s.n_cols, s.nnz, s.t_ptr.data_ptr(), s.t_idx.data_ptr(), s.t_perm.data_ptr(),
                                           val.data_ptr(), G.data_ptr(), s.p, 0, B.data_ptr(), ldb, 0, gv.data_ptr(), gradB.data_ptr(), s.p, 0, s.p, 1,
                                           s.device, stream_of(s.device)),
                      "tsgu_csr_mm_backward");
            } else {
                if (need_a) {
                    gv = at::empty({s.nnz}, val.options());
                    check(tsgu_csr_sddmm(s.vtype, s.itype, s.n_rows, s.n_cols, s.nnz, s.crow.data_ptr(), s.col.data_ptr(), G.data_ptr(), s.p, 0,
                                         B.data_ptr(), ldb, 0, gv.data_ptr(), 1.0, 0, s.p, 1, s.device, stream_of(s.device)),
                          "tsgu_csr_sddmm");
                }
                if (need_b) {
                    gradB = at::empty({s.n_cols, s.p}, G.options());
                    check(tsgu_csr_spmm(s.vtype, s.itype, s.n_cols, s.n_rows, s.nnz, s.t_ptr.data_ptr(), s.t_idx.data_ptr(), val.data_ptr(),
                                        s.t_perm.data_ptr(), G.data_ptr(), s.p, 1, 0, gradB.data_ptr(), s.p, 1, 0, s.p, 1, s.t_max_row_nnz, nullptr, 0,
                                        nullptr, s.device, stream_of(s.device)),
                          "tsgu_csr_spmm");
                }
            }
        } else {
            if (need_a) {
                // gradA[k] = <G[row k,:], B[col k,:]> at A's stored entries only (reference sparse_matmul.py:172-205)
                gv = s.batch ? at::empty({s.batch, s.item_nnz}, val.options()) : at::empty({s.nnz}, val.options());
                if (s.sddmm.kind == 3)
                    check(tsgu_csr_sddmm_tile(s.vtype, static_cast<const tsgu_tile_plan*>(s.sddmm.plan()), G.data_ptr(), s.p, B.data_ptr(), ldb,
                                              gv.data_ptr(), 1.0, s.p, s.device, stream_of(s.device)),
                          "tsgu_csr_sddmm_tile");
                else if (s.sddmm.kind == 0)
                    check(tsgu_csr_sddmm_march(s.vtype, static_cast<const tsgu_march_plan*>(s.sddmm.plan()), s.n_rows, s.nnz, G.data_ptr(), s.p,
                                               B.data_ptr(), ldb, gv.data_ptr(), 1.0, 0, s.p, s.device, stream_of(s.device)),
                          "tsgu_csr_sddmm_march");
                else
                    check(tsgu_csr_sddmm_lattice(s.vtype, static_cast<const tsgu_lattice_plan*>(s.sddmm.plan()), s.n_rows, s.nnz, G.data_ptr(), s.p,
                                                 B.data_ptr(), ldb, gv.data_ptr(), 1.0, s.p, s.device, stream_of(s.device)),
                          "tsgu_csr_sddmm_lattice");
            }
            if (need_b) {
                // gradB = Aᵀ·G (reference sparse_matmul.py:229), through A's own arrays
                gradB = s.batch ? at::empty({s.batch, s.item_cols, s.p}, G.options()) : at::empty({s.n_cols, s.p}, G.options());
                spmm(s, s.spmm_t, s.n_cols, val, G, gradB);
            }
        }
        if (need_a) {
            // the sparse gradient in A's own layout, with A's index tensors (reference sparse_matmul.py:208-219)
            if (s.coo)
                gradA = at::sparse_coo_tensor(ctx->saved_data["idx"].toTensor(), gv, {s.n_rows, s.n_cols}, gv.options().layout(at::kSparse));
            else if (s.batch)
                gradA = at::sparse_csr_tensor(ctx->saved_data["crow"].toTensor(), ctx->saved_data["col"].toTensor(), gv,
                                              {s.batch, s.item_rows, s.item_cols}, gv.options().layout(at::kSparseCsr));
            else
                gradA = at::sparse_csr_tensor(ctx->saved_data["crow"].toTensor(), ctx->saved_data["col"].toTensor(), gv, {s.n_rows, s.n_cols},
                                              gv.options().layout(at::kSparseCsr));
        }
        return {gradA, gradB, at::Tensor()};
    }

};

at::Tensor step(const at::Tensor& A, const at::Tensor& B, StepPlanPtr handle) {
    TORCH_CHECK(handle != nullptr, "no step plan");
    const StepPlan& s = *handle;
    // (Python has validated layout / dims / dtypes / device; these are the conditions of the raw-pointer launches)
    if (s.batch) {
        TORCH_CHECK(A.layout() == at::kSparseCsr && A.dim() == 3 && A.size(0) == s.batch && A.size(1) == s.item_rows && A.size(2) == s.item_cols,
                    "step plan of another (batched) matrix");
        TORCH_CHECK(plain3(B, s.batch, s.item_cols, s.p), "the fast step takes a contiguous, 16-byte aligned (batch, n_cols, p) operand");
        return StepFunction::apply(A, B, std::move(handle));
    }
    TORCH_CHECK(A.layout() == (s.coo ? at::kSparse : at::kSparseCsr) && A.dim() == 2 && A.size(0) == s.n_rows && A.size(1) == s.n_cols,
                "step plan of another matrix");
    TORCH_CHECK(plain(B, s.n_cols, s.p), "the fast step takes a contiguous, 16-byte aligned (n_cols, p) operand");
    return StepFunction::apply(A, B, std::move(handle));
}

}  // namespace

PYBIND11_MODULE(_tsgu_host, m) {
    m.doc() = "C++ host path of the steady-state sparse_mm step (see csrc/host/step.cpp)";
    py::class_<StepPlan, StepPlanPtr>(m, "StepPlan")
        .def(py::init([](at::Tensor crow, at::Tensor col, int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t p, int vtype, int device,
                         std::tuple<int, py::bytes, int> fwd, std::tuple<int, py::bytes, int> sddmm, std::tuple<int, py::bytes, int> spmm_t,
                         std::vector<at::Tensor> tables) {
            auto s = std::make_shared<StepPlan>();
            s->crow = std::move(crow);
            s->col = std::move(col);
            s->n_rows = n_rows, s->n_cols = n_cols, s->nnz = nnz, s->p = p, s->vtype = vtype, s->device = device;
            auto prod = [](const std::tuple<int, py::bytes, int>& t) {
                Product q;
                q.kind = std::get<0>(t), q.blob = std::string(std::get<1>(t)), q.transposed = std::get<2>(t);
                const size_t want = q.kind == 0 ? sizeof(tsgu_march_plan)
                                    : (q.kind == 1 ? sizeof(tsgu_lattice_plan) : (q.kind == 3 ? sizeof(tsgu_tile_plan) : 0));
                if (q.blob.size() != want) throw std::invalid_argument("plan struct of the wrong size");
                q.blob.reserve(64);      // (heap storage: 16-byte aligned, never moved again)
                return q;
            };
            s->fwd = prod(fwd), s->sddmm = prod(sddmm), s->spmm_t = prod(spmm_t);
            s->tables = std::move(tables);
            return s;
        }))
        .def("set_plan_free", [](StepPlan& s, int itype, at::Tensor t_ptr, at::Tensor t_idx, at::Tensor t_perm, int64_t max_row_nnz,
                                 int64_t t_max_row_nnz, bool fused_backward) {
            s.fwd.kind = s.sddmm.kind = s.spmm_t.kind = 2;
            s.itype = itype;
            s.t_ptr = std::move(t_ptr), s.t_idx = std::move(t_idx), s.t_perm = std::move(t_perm);
            s.max_row_nnz = max_row_nnz, s.t_max_row_nnz = t_max_row_nnz, s.fused_backward = fused_backward ? 1 : 0;
        })
        .def("set_batch", [](StepPlan& s, int64_t batch, int64_t item_rows, int64_t item_cols, int64_t item_nnz) {
            if (batch < 1 || batch * item_rows != s.n_rows || batch * item_cols != s.n_cols || batch * item_nnz != s.nnz)
                throw std::invalid_argument("batch geometry does not match the flat problem");
            s.batch = batch, s.item_rows = item_rows, s.item_cols = item_cols, s.item_nnz = item_nnz;
        })
        .def("set_coo", [](StepPlan& s, at::Tensor indices) {
            s.coo = 1;
            s.indices = std::move(indices);
        })
        .def_readonly("p", &StepPlan::p)
        .def_readonly("n_rows", &StepPlan::n_rows);
    m.def("step", &step, "C = A @ B with the sparsity-preserving backward, host path in C++");
    m.def("abi_version", []() { return tsgu_abi_version(); });
}
