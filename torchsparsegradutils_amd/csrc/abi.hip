// extern "C" entry points for K1/K2/K3 (declared in include/tsgu_hip.h).
// Argument validation lives here; the typed launchers are in spmm_*.hip / sddmm_*.hip.
#include "sddmm_impl.h"
#include "spmm_impl.h"

#include <cstring>

namespace tsgu {
int spmm_dispatch_f32(int, const SpmmParams&, int64_t, hipStream_t);
int spmm_dispatch_f64(int, const SpmmParams&, int64_t, hipStream_t);
int spmm_dispatch_bf16(int, const SpmmParams&, int64_t, hipStream_t);
int sddmm_dispatch_f32(int, const SddmmParams&, int64_t, hipStream_t);
int sddmm_dispatch_f64(int, const SddmmParams&, int64_t, hipStream_t);
int sddmm_dispatch_bf16(int, const SddmmParams&, int64_t, hipStream_t);
int coo_sddmm_dispatch_f32(int, const CooSddmmParams&, hipStream_t);
int coo_sddmm_dispatch_f64(int, const CooSddmmParams&, hipStream_t);
int coo_sddmm_dispatch_bf16(int, const CooSddmmParams&, hipStream_t);
}  // namespace tsgu

using namespace tsgu;

// 128-bit content fingerprint of an index array (see include/tsgu_hip.h): position-weighted sums in wrapping 64-bit integer
// arithmetic — integer addition commutes, so the atomics of different workgroups give the same words in any order.  The fingerprint
// only SELECTS a candidate pattern; what decides an adoption is the exact comparison (CMP: the number of elements that differ from
// `ref` is added to out[2]) the same pass makes.  COPY: the pass also writes the array to `copy` (the cache's own copy, what later
// comparisons read).
template <typename I, bool CMP, bool COPY, bool HASH = true>
__global__ __launch_bounds__(256) void tsgu_fingerprint_kernel(const I* __restrict__ x, const I* __restrict__ ref, I* __restrict__ copy,
                                                                int64_t n, unsigned long long* __restrict__ out) {
    constexpr int V = 16 / (int)sizeof(I);               // indices per 16-byte load
    struct alignas(16) Vec {
        I v[V];
    };
    unsigned long long h1 = 0, h2 = 0;
    unsigned int diff = 0;
    // position weights from 32-bit multiplicative hashes of the index (a 64-bit `%` costs ~100 instructions per element: the
    // first version of this kernel took 0.3 ms for C2's 27 M column indices)
    auto add = [&](int64_t k, I xv) {
        if constexpr (!HASH) return;      // (compare only: the caller takes the fingerprint of the reference when the tensors are equal)
        const unsigned long long a = (unsigned long long)(long long)xv + 0x9e3779b97f4a7c15ull;
        const unsigned int kl = (unsigned int)k;
        const unsigned int w1 = (kl * 2654435761u) | 1u, w2 = ((kl ^ (kl >> 15)) * 2246822519u) | 1u;
        h1 += a * (unsigned long long)w1;
        h2 += (a ^ (a >> 29)) * (unsigned long long)w2;
    };
    // 16-byte loads, four of them in flight per thread: with one 4-byte load per thread and trip the second version was bound by
    // memory latency (0.27 ms for 108 MB = 0.4 TB/s; the whole step of a caller with fresh index tensors waits for this kernel)
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    bool vec = reinterpret_cast<uintptr_t>(x) % 16 == 0;
    if (CMP) vec = vec && reinterpret_cast<uintptr_t>(ref) % 16 == 0;
    if (COPY) vec = vec && reinterpret_cast<uintptr_t>(copy) % 16 == 0;
    const int64_t nv = vec ? n / V : 0;
    const Vec* const xv = reinterpret_cast<const Vec*>(x);
    const Vec* const rv = reinterpret_cast<const Vec*>(ref);
    Vec* const cv = reinterpret_cast<Vec*>(copy);
    int64_t c = gtid;
    for (; c + 3 * stride < nv; c += 4 * stride) {
        Vec a[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = xv[c + u * stride];
        if (CMP) {
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = rv[c + u * stride];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                add((c + u * stride) * V + j, a[u].v[j]);
                if (CMP) diff += a[u].v[j] != r[u].v[j];
            }
            if (COPY) cv[c + u * stride] = a[u];
        }
    }
    for (; c < nv; c += stride) {
        const Vec a = xv[c];
#pragma unroll
        for (int j = 0; j < V; ++j) add(c * V + j, a.v[j]);
        if (CMP) {
            const Vec r = rv[c];
#pragma unroll
            for (int j = 0; j < V; ++j) diff += a.v[j] != r.v[j];
        }
        if (COPY) cv[c] = a;
    }
    for (int64_t k = nv * V + gtid; k < n; k += stride) {      // the tail (everything, for pointers that are not 16-byte aligned)
        const I a = x[k];
        add(k, a);
        if (CMP) diff += a != ref[k];
        if (COPY) copy[k] = a;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        h1 += __shfl_xor(h1, m, 64);
        h2 += __shfl_xor(h2, m, 64);
    }
    const unsigned long long wave_diff = CMP ? __popcll(__ballot(diff != 0)) : 0;      // (a count of lanes: non-zero iff anything differs)
    // ONE set of atomics per workgroup: same-address 64-bit atomics retire at ~80 M/s, and with a pair per wave of 2048 workgroups
    // the 16 384 atomics (0.2 ms) — not the 108 MB — were the kernel's duration
    __shared__ unsigned long long part[3][4];
    if ((threadIdx.x & 63) == 0) part[0][threadIdx.x >> 6] = h1, part[1][threadIdx.x >> 6] = h2, part[2][threadIdx.x >> 6] = wave_diff;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (HASH) {
            atomicAdd(out, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
            atomicAdd(out + 1, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
        }
        if (CMP) {
            const unsigned long long d = part[2][0] + part[2][1] + part[2][2] + part[2][3];
            if (d) atomicAdd(out + 2, d);
        }
    }
}

template <typename I>
static void fingerprint_launch(unsigned blocks, hipStream_t s, const void* x, const void* ref, void* copy, int64_t n, void* out, bool hash) {
    const I* const xi = static_cast<const I*>(x);
    const I* const ri = static_cast<const I*>(ref);
    I* const ci = static_cast<I*>(copy);
    unsigned long long* const o = static_cast<unsigned long long*>(out);
    if (ref && !copy && !hash)
        hipLaunchKernelGGL((tsgu_fingerprint_kernel<I, true, false, false>), dim3(blocks), dim3(256), 0, s, xi, ri, ci, n, o);
    else if (ref && copy)
        hipLaunchKernelGGL((tsgu_fingerprint_kernel<I, true, true>), dim3(blocks), dim3(256), 0, s, xi, ri, ci, n, o);
    else if (ref)
        hipLaunchKernelGGL((tsgu_fingerprint_kernel<I, true, false>), dim3(blocks), dim3(256), 0, s, xi, ri, ci, n, o);
    else if (copy)
        hipLaunchKernelGGL((tsgu_fingerprint_kernel<I, false, true>), dim3(blocks), dim3(256), 0, s, xi, ri, ci, n, o);
    else
        hipLaunchKernelGGL((tsgu_fingerprint_kernel<I, false, false>), dim3(blocks), dim3(256), 0, s, xi, ri, ci, n, o);
}

extern "C" {

int tsgu_abi_version(void) { return TSGU_ABI_VERSION; }

const char* tsgu_status_string(int status) {
    switch (status) {
        case TSGU_OK: return "ok";
        case TSGU_ERR_BAD_DTYPE: return "unsupported value/index dtype";
        case TSGU_ERR_BAD_ARG: return "bad argument (null pointer, negative size or leading dimension too small)";
        case TSGU_ERR_TOO_LARGE: return "problem exceeds a kernel limit (n_cols >= 2^31, batch > 65535 or grid too large)";
        case TSGU_ERR_LAUNCH: return "HIP kernel launch failed";
        case TSGU_ERR_RUNTIME: return "HIP runtime call failed";
        case TSGU_ERR_TIMEOUT: return "device-side dependency wait timed out";
    }
    return "unknown status";
}

int tsgu_device_info(int device, char* name, int cap, int* n_cu, int* wave_size) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return TSGU_ERR_RUNTIME;
    if (name && cap > 0) {
        std::strncpy(name, prop.gcnArchName, (size_t)cap - 1);
        name[cap - 1] = 0;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    return TSGU_OK;
}

int tsgu_device_cu_count(int device, int* n_cu) {
    // (hipDeviceGetAttribute: microseconds; hipGetDeviceProperties — what tsgu_device_info and torch's get_device_properties call —
    // took 117 ms of a process' first sparse_mm step)
    int n = 0;
    if (!n_cu || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return TSGU_ERR_RUNTIME;
    *n_cu = n;
    return TSGU_OK;
}

// Streaming copy, 16 bytes per lane, grid-stride: the measured HBM ceiling a kernel of this library can be compared with
// (bench.py reports it next to torch's copy_; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy kernel).
__global__ __launch_bounds__(256) void tsgu_copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(src + i));
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(dst + i));
    }
}

int tsgu_device_copy(const void* src, void* dst, int64_t bytes, int device, void* stream) {
    if (!src || !dst || bytes < 0 || bytes % 16 || !aligned16(src) || !aligned16(dst)) return TSGU_ERR_BAD_ARG;
    if (bytes == 0) return TSGU_OK;
    if (const int rc = set_device(device)) return rc;
    const int64_t n16 = bytes / 16;
    const int64_t want = (n16 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 256 * 32 ? want : 256 * 32);   // 32 workgroups per CU, grid-stride beyond that
    hipLaunchKernelGGL(tsgu_copy16_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint4*>(src),
                       static_cast<uint4*>(dst), n16);
    return check_launch();
}

int tsgu_index_fingerprint_match(int itype, int64_t n, const void* x, const void* ref, void* copy, void* out3, int accumulate,
                                 int device, void* stream) {
    if (n < 0 || !out3 || (n > 0 && !x)) return TSGU_ERR_BAD_ARG;
    if (itype != TSGU_I32 && itype != TSGU_I64) return TSGU_ERR_BAD_DTYPE;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool hash = (accumulate & 2) == 0;      // bit 1: compare only (ref given, no copy) — out3[0..1] are left alone
    if (!(accumulate & 1) && hipMemsetAsync(out3, 0, 24, s) != hipSuccess) return TSGU_ERR_RUNTIME;
    if (n == 0) return TSGU_OK;
    const int64_t want = (n + 256 * 16 - 1) / (256 * 16);
    const unsigned blocks = (unsigned)(want < 512 ? want : 512);      // (two workgroups per CU: 64 bytes x 512 threads in flight on each)
    if (itype == TSGU_I32)
        fingerprint_launch<int32_t>(blocks, s, x, ref, copy, n, out3, hash);
    else
        fingerprint_launch<int64_t>(blocks, s, x, ref, copy, n, out3, hash);
    return check_launch();
}

int tsgu_index_fingerprint(int itype, int64_t n, const void* x, void* out2, int accumulate, int device, void* stream) {
    if (n < 0 || !out2 || (n > 0 && !x)) return TSGU_ERR_BAD_ARG;
    if (itype != TSGU_I32 && itype != TSGU_I64) return TSGU_ERR_BAD_DTYPE;
    if (!accumulate) {
        if (const int rc = set_device(device)) return rc;
        if (hipMemsetAsync(out2, 0, 16, static_cast<hipStream_t>(stream)) != hipSuccess) return TSGU_ERR_RUNTIME;
    }
    return tsgu_index_fingerprint_match(itype, n, x, nullptr, nullptr, out2, 1, device, stream);      // (no ref: out2[2] is never touched)
}

int64_t tsgu_spmm_num_blocks(int vtype, int64_t n_rows, int64_t nnz_per_item, int64_t p, int64_t max_row_nnz) {
    // mirrors spmm_geom(): the fused-dot path is only used with contiguous, 16-byte
    // aligned operands, so "wide" depends on p alone.
    const int wide = vtype == TSGU_F32 ? 4 : vtype == TSGU_F64 ? 2 : 8;
    RowGeom g = pick_geom(wide, p % wide == 0, p);
    prefer_row_per_lane(g, n_rows, nnz_per_item, max_row_nnz);  // the fused-dot path never walks a permutation
    const int64_t rpb = kBlock / (g.cl * g.ep);
    const int64_t rows = rpb * spmm_row_mult(n_rows, nnz_per_item, rpb);
    return (n_rows + rows - 1) / rows;
}

int tsgu_csr_spmm(int vtype, int itype, int64_t n_rows, int64_t n_cols, int64_t nnz_per_item,
                  const void* crow, const void* col, const void* val, const void* perm,
                  const void* B, int64_t ldb, int64_t b_col_stride, int64_t b_batch_stride,
                  void* C, int64_t ldc, int64_t c_col_stride, int64_t c_batch_stride,
                  int64_t p, int64_t batch, int64_t max_row_nnz,
                  const void* dot_w, int64_t ldw, void* dot_partial,
                  int device, void* stream) {
    if (n_rows < 0 || n_cols < 0 || nnz_per_item < 0 || p < 0 || batch < 0) return TSGU_ERR_BAD_ARG;
    if (n_rows == 0 || p == 0 || batch == 0) return TSGU_OK;
    if (!crow || !C || (nnz_per_item > 0 && (!col || !val || !B))) return TSGU_ERR_BAD_ARG;
    if (b_col_stride < 1 || c_col_stride < 1) return TSGU_ERR_BAD_ARG;
    if ((b_col_stride == 1 && ldb < p) || (c_col_stride == 1 && ldc < p) || ldb < 1 || ldc < 1) return TSGU_ERR_BAD_ARG;
    if (n_cols > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    if ((dot_partial != nullptr) != (dot_w != nullptr)) return TSGU_ERR_BAD_ARG;
    if (dot_partial && (b_col_stride != 1 || c_col_stride != 1)) return TSGU_ERR_BAD_ARG;
    if (dot_partial && (ldw < p || vtype == TSGU_BF16)) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    SpmmParams P{};
    P.n_rows = n_rows;
    P.nnz_per_item = nnz_per_item;
    P.p = p;
    P.crow = crow;
    P.col = col;
    P.val = val;
    P.perm = perm;
    P.B = B;
    P.ldb = ldb;
    P.b_bs = b_batch_stride;
    P.max_row_nnz = max_row_nnz;
    P.b_cs = b_col_stride;
    P.c_cs = c_col_stride;
    P.C = C;
    P.ldc = ldc;
    P.c_bs = c_batch_stride;
    P.W = dot_w;
    P.ldw = ldw;
    P.dot_partial = dot_partial;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (vtype) {
        case TSGU_F32: return spmm_dispatch_f32(itype, P, batch, s);
        case TSGU_F64: return spmm_dispatch_f64(itype, P, batch, s);
        case TSGU_BF16: return spmm_dispatch_bf16(itype, P, batch, s);
    }
    return TSGU_ERR_BAD_DTYPE;
}

int tsgu_csr_sddmm(int vtype, int itype, int64_t n_rows, int64_t n_cols, int64_t nnz_per_item,
                   const void* crow, const void* col,
                   const void* G, int64_t ldg, int64_t g_batch_stride,
                   const void* B, int64_t ldb, int64_t b_batch_stride,
                   void* out, double alpha, int swap_roles,
                   int64_t p, int64_t batch, int device, void* stream) {
    if (n_rows < 0 || n_cols < 0 || nnz_per_item < 0 || p < 0 || batch < 0) return TSGU_ERR_BAD_ARG;
    if (n_rows == 0 || nnz_per_item == 0 || batch == 0) return TSGU_OK;
    if (!crow || !col || !out || (p > 0 && (!G || !B))) return TSGU_ERR_BAD_ARG;
    if (ldg < p || ldb < p) return TSGU_ERR_BAD_ARG;
    if (n_cols > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    if (const int rc = set_device(device)) return rc;
    SddmmParams P{};
    P.n_rows = n_rows;
    P.nnz_per_item = nnz_per_item;
    P.p = p;
    P.crow = crow;
    P.col = col;
    if (!swap_roles) {
        P.R = G;
        P.ldr = ldg;
        P.r_bs = g_batch_stride;
        P.Cm = B;
        P.ldc = ldb;
        P.c_bs = b_batch_stride;
    } else {
        P.R = B;
        P.ldr = ldb;
        P.r_bs = b_batch_stride;
        P.Cm = G;
        P.ldc = ldg;
        P.c_bs = g_batch_stride;
    }
    P.out = out;
    P.alpha = alpha;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (vtype) {
        case TSGU_F32: return sddmm_dispatch_f32(itype, P, batch, s);
        case TSGU_F64: return sddmm_dispatch_f64(itype, P, batch, s);
        case TSGU_BF16: return sddmm_dispatch_bf16(itype, P, batch, s);
    }
    return TSGU_ERR_BAD_DTYPE;
}

int tsgu_coo_sddmm(int vtype, int itype, int64_t nnz, const void* row, const void* col,
                   const void* G, int64_t ldg, const void* B, int64_t ldb,
                   void* out, double alpha, int64_t p, int device, void* stream) {
    if (nnz < 0 || p < 0) return TSGU_ERR_BAD_ARG;
    if (nnz == 0) return TSGU_OK;
    if (!row || !col || !out || (p > 0 && (!G || !B))) return TSGU_ERR_BAD_ARG;
    if (ldg < p || ldb < p) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    CooSddmmParams P{};
    P.nnz = nnz;
    P.p = p;
    P.row = row;
    P.col = col;
    P.R = G;
    P.ldr = ldg;
    P.Cm = B;
    P.ldc = ldb;
    P.out = out;
    P.alpha = alpha;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (vtype) {
        case TSGU_F32: return coo_sddmm_dispatch_f32(itype, P, s);
        case TSGU_F64: return coo_sddmm_dispatch_f64(itype, P, s);
        case TSGU_BF16: return coo_sddmm_dispatch_bf16(itype, P, s);
    }
    return TSGU_ERR_BAD_DTYPE;
}

}  // extern "C"
