// K1/K2: CSR × dense on gfx950.  One group of CL×EP lanes owns one row; a 256-thread
// workgroup owns 256/(CL·EP) consecutive rows.  The workgroup's slice of the (col, val)
// stream is contiguous in memory, so it is read once with fully coalesced loads into LDS
// and then broadcast to the row groups from there; the dense RHS rows are gathered with
// 16-byte loads (one 128-B line per fp32 row at p = 32) and accumulated in registers.
// No atomics, fixed summation order (entry order inside a row) => run-to-run deterministic.
#pragma once

#include "tsgu_common.h"

namespace tsgu {

constexpr int kStageCap = 2048;  // staged entries per pass (16 KiB LDS for 4-byte values)
constexpr int kMaxRowMult = 8;   // runs of RPB rows one workgroup may own

template <typename V>
struct StageBytes {
    static constexpr int value = kStageCap * (int)(sizeof(int) + (sizeof(V) < 4 ? 4 : sizeof(V)));
};

struct SpmmParams {
    int64_t n_rows, nnz_per_item, p;
    const void* crow;
    const void* col;
    const void* val;
    const void* perm;
    const void* B;
    int64_t ldb, b_bs;
    int64_t b_cs, c_cs;  // column strides (elements) of B and C: 1 = row-major; != 1 (transposed views) => one column per grid.z slice
    void* C;
    int64_t ldc, c_bs;
    const void* W;
    int64_t ldw;
    void* dot_partial;
    int64_t max_row_nnz;  // longest row of the walked pattern (0 = unknown): see prefer_row_per_lane
    int64_t nblocks;  // row blocks per batch item
    int rmul;         // runs of RPB rows per workgroup
};

// LDS image of the staged entries.  4-byte-or-narrower values are packed with their
// column into one 8-byte slot (one ds_read_b64 per entry); doubles use two arrays.
template <typename V, bool PACK = (sizeof(V) <= 4)>
struct Staged;

template <typename V>
struct Staged<V, true> {
    using Acc = typename VT<V>::Acc;
    uint2* slot;
    __device__ __forceinline__ explicit Staged(unsigned char* base) : slot(reinterpret_cast<uint2*>(base)) {}
    __device__ __forceinline__ void put(int i, int c, V v) const {
        slot[i] = make_uint2((unsigned)c, __float_as_uint(VT<V>::up(v)));
    }
    __device__ __forceinline__ void get(int i, int& c, Acc& a) const {
        const uint2 e = slot[i];
        c = (int)e.x;
        a = __uint_as_float(e.y);
    }
};

template <typename V>
struct Staged<V, false> {
    using Acc = typename VT<V>::Acc;
    double* vals;
    int* cols;
    __device__ __forceinline__ explicit Staged(unsigned char* base)
        : vals(reinterpret_cast<double*>(base)), cols(reinterpret_cast<int*>(base + kStageCap * sizeof(double))) {}
    __device__ __forceinline__ void put(int i, int c, V v) const {
        cols[i] = c;
        vals[i] = v;
    }
    __device__ __forceinline__ void get(int i, int& c, Acc& a) const {
        c = cols[i];
        a = vals[i];
    }
};

template <typename V, typename I, int VEC, int CL, int EP, bool DOT, bool PERM, bool MULTI>
__global__ __launch_bounds__(kBlock) void csr_spmm_kernel(const SpmmParams P) {
    using Acc = typename VT<V>::Acc;
    constexpr int GROUP = CL * EP;
    constexpr int RPB = kBlock / GROUP;
#ifndef TSGU_SPMM_U
#define TSGU_SPMM_U 4
#endif
    constexpr int U = TSGU_SPMM_U;  // gathers issued back to back per lane
    constexpr int SU = 4;  // staging loads issued back to back per thread

    __shared__ __attribute__((aligned(16))) unsigned char smem[StageBytes<V>::value];
    const Staged<V> stage(smem);

    const int tid = threadIdx.x;
    const int grp = tid / GROUP;
    const int gl = tid % GROUP;
    const int cl = gl % CL;
    const int ep = gl / CL;

    const int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    const int64_t item = blockIdx.y;
    const int64_t cbase = ((int64_t)blockIdx.z * CL + cl) * VEC;  // first column of this lane
    const bool col_ok = cbase < P.p;

    const I* __restrict__ crow = static_cast<const I*>(P.crow) + item * (P.n_rows + 1);
    const I* __restrict__ col = static_cast<const I*>(P.col) + item * P.nnz_per_item;
    const V* __restrict__ val = static_cast<const V*>(P.val) + item * P.nnz_per_item;
    const I* __restrict__ perm = static_cast<const I*>(P.perm);
    const V* __restrict__ B = static_cast<const V*>(P.B) + item * P.b_bs + cbase * P.b_cs;
    const uint32_t ldb = (uint32_t)P.ldb;

    // A workgroup owns RPB·rmul consecutive rows (rmul > 1 for short rows, so that the dependent
    // crow -> (col,val) -> gather round trips are amortised over more work); row group `grp` handles
    // rows row0 + grp + m·RPB, m < rmul.
    const int rmul = MULTI ? P.rmul : 1;
    constexpr int MM = MULTI ? kMaxRowMult : 1;  // compile-time bound of the row-run loops
    const int64_t row0 = vb * RPB * rmul;
    const int64_t row1 = row0 + (int64_t)RPB * rmul < P.n_rows ? row0 + (int64_t)RPB * rmul : P.n_rows;
    const int64_t blk_begin = (int64_t)crow[row0];
    const int64_t blk_end = (int64_t)crow[row1];

    // stage entries [cs, ce) of the (col, val) stream: all loads of a pass are issued before the first
    // LDS write, so a pass costs one memory round trip (two with the value indirection).
    auto stage_pass = [&](int64_t cs, int64_t ce) {
        for (int64_t base = cs + tid; base < ce; base += (int64_t)kBlock * SU) {
            I cj[SU];
            V vv[SU];
            int64_t q[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int64_t k = base + (int64_t)u * kBlock;
                const bool ok = k < ce;
                cj[u] = ok ? stream_load(col + k) : (I)0;
                if constexpr (PERM) q[u] = ok ? (int64_t)stream_load(perm + item * P.nnz_per_item + k) : 0;
                else q[u] = ok ? k : 0;
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                if constexpr (PERM) vv[u] = val[q[u]];  // gathered: keep cacheable
                else vv[u] = stream_load(val + q[u]);
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int64_t k = base + (int64_t)u * kBlock;
                if (k < ce) stage.put((int)(k - cs), (int)cj[u], vv[u]);
            }
        }
    };

    // accumulate the staged entries [lo, hi) (absolute positions inside the pass starting at cs)
    auto consume = [&](int64_t cs, int64_t lo, int64_t hi, Acc(&acc)[VEC]) {
        int i = (int)(lo - cs) + ep;
        const int iend = (int)(hi - cs);
        if (col_ok) {
            for (; i + (U - 1) * EP < iend; i += U * EP) {
                int j[U];
                Acc a[U];
                Acc b[U][VEC];
#pragma unroll
                for (int u = 0; u < U; ++u) stage.get(i + u * EP, j[u], a[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) load_vec<V, VEC>(B + row_off(j[u], ldb), b[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = fma(a[u], b[u][v], acc[v]);
                }
            }
            for (; i < iend; i += EP) {
                int j;
                Acc a;
                Acc b[VEC];
                stage.get(i, j, a);
                load_vec<V, VEC>(B + row_off(j, ldb), b);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = fma(a, b[v], acc[v]);
            }
        }
    };

    Acc dotp[VEC];  // DOT: Σ over this lane's rows of C[row,c]·W[row,c]
#pragma unroll
    for (int v = 0; v < VEC; ++v) dotp[v] = 0;

    auto finish_row = [&](int64_t row, bool row_ok, Acc(&acc)[VEC]) {
        if constexpr (EP > 1) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = ep_sum<Acc, CL, EP>(acc[v]);
        }
        if (row_ok && col_ok && ep == 0) {
            V* __restrict__ C = static_cast<V*>(P.C) + item * P.c_bs + row * P.ldc + cbase * P.c_cs;
            store_vec<V, VEC, true>(C, acc);
            if constexpr (DOT) {
                Acc w[VEC];
                load_vec<V, VEC>(static_cast<const V*>(P.W) + row * P.ldw + cbase, w);
#pragma unroll
                for (int v = 0; v < VEC; ++v) dotp[v] = fma(acc[v], w[v], dotp[v]);
            }
        }
    };

    const bool fits = blk_end - blk_begin <= kStageCap;
    if (fits) {
        // common case: the whole workgroup slice is staged once, then every row is consumed from LDS.
        // The row bounds are fetched before the staging pass so that they share its round trip.
        I rb[MM], re_[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            const int64_t row = row0 + grp + (int64_t)m * RPB;
            const bool ok = m < rmul && row < row1;
            rb[m] = ok ? crow[row] : (I)0;
            re_[m] = ok ? crow[row + 1] : (I)0;
        }
        stage_pass(blk_begin, blk_end);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            if (m >= rmul) break;
            const int64_t row = row0 + grp + (int64_t)m * RPB;
            const bool row_ok = row < row1;
            const int64_t start = (int64_t)rb[m];
            const int64_t end = (int64_t)re_[m];
            Acc acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0;
            consume(blk_begin, start, end, acc);
            finish_row(row, row_ok, acc);
        }
    } else {
        // long rows: each run of RPB rows streams its entries through the staging window in passes
        for (int m = 0; m < rmul; ++m) {
            const int64_t s0 = row0 + (int64_t)m * RPB;
            if (s0 >= row1) break;
            const int64_t s1 = s0 + RPB < row1 ? s0 + RPB : row1;
            const int64_t sb = (int64_t)crow[s0], se = (int64_t)crow[s1];
            const int64_t row = s0 + grp;
            const bool row_ok = row < s1;
            const int64_t start = row_ok ? (int64_t)crow[row] : 0;
            const int64_t end = row_ok ? (int64_t)crow[row + 1] : 0;
            Acc acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0;
            for (int64_t cs = sb; cs < se; cs += kStageCap) {
                const int64_t ce = cs + kStageCap < se ? cs + kStageCap : se;
                __syncthreads();
                stage_pass(cs, ce);
                __syncthreads();
                consume(cs, start > cs ? start : cs, end < ce ? end : ce, acc);
            }
            finish_row(row, row_ok, acc);
        }
    }

    if constexpr (DOT) {
        // partial[block][c] = Σ_rows C[row,c]·W[row,c]: the lanes of a wave that own the same columns are summed with
        // xor-shuffles (fixed tree), then the four waves through LDS — a serial sum over the row groups costs
        // 256 dependent LDS reads per workgroup with one lane per row (10 us at C4).
        constexpr int TW = CL * VEC;  // columns covered by one column tile
        const int lane = tid & (kWave - 1), wave = tid / kWave;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            Acc x = ep == 0 ? dotp[v] : (Acc)0;
#pragma unroll
            for (int m = CL; m < kWave; m <<= 1) x += shfl_xor_acc(x, m);
            dotp[v] = x;
        }
        __syncthreads();
        Acc* red = reinterpret_cast<Acc*>(smem);
        constexpr int NW = kBlock / kWave;
        if (lane < CL) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) red[wave * TW + lane * VEC + v] = dotp[v];
        }
        __syncthreads();
        if (tid < TW) {
            Acc s = 0;
            for (int r = 0; r < NW; ++r) s += red[r * TW + tid];
            const int64_t c = (int64_t)blockIdx.z * TW + tid;
            if (c < P.p) {
                static_cast<Acc*>(P.dot_partial)[(item * P.nblocks + vb) * P.p + c] = s;
            }
        }
    }
}

// rows-per-workgroup multiplier: aim at ~1.5k staged entries per workgroup, at most 8 runs of RPB rows
inline int spmm_row_mult(int64_t n_rows, int64_t nnz, int64_t rpb) {
    if (n_rows <= 0) return 1;
    const double per_run = (double)nnz / (double)n_rows * (double)rpb;
    int k = per_run > 0 ? (int)(1536.0 / per_run) : 8;
    if (k < 1) k = 1;
    if (k > kMaxRowMult) k = kMaxRowMult;
    return k;
}

inline int64_t spmm_rows_per_block(const RowGeom& g) { return kBlock / (g.cl * g.ep); }

template <typename V>
inline RowGeom spmm_geom(const SpmmParams& P, int64_t batch) {
    constexpr int wide = VT<V>::kWide;
    bool can = (P.p % wide == 0) && (P.ldb % wide == 0) && (P.ldc % wide == 0) && aligned16(P.B) && aligned16(P.C);
    if (batch > 1) can = can && (P.b_bs % wide == 0) && (P.c_bs % wide == 0);
    if (P.W) can = can && (P.ldw % wide == 0) && aligned16(P.W);
    RowGeom g = pick_geom(wide, can, P.p);
    if (P.b_cs != 1 || P.c_cs != 1) {
        // column-strided operands (e.g. the .t() views the sparse multivariate normal passes): one column per grid.z
        // slice and lanes along ROWS — with unit row stride consecutive lanes then touch consecutive addresses
        g.vec = 1;
        g.cl = 1;
        g.ep = (P.n_rows > 0 && P.nnz_per_item <= 16 * P.n_rows) ? 1 : 8;
        g.col_tiles = P.p;
        return g;
    }
    prefer_row_per_lane(g, P.n_rows, P.nnz_per_item, P.max_row_nnz);
    return g;
}

template <typename V, typename I>
int spmm_launch(SpmmParams P, int64_t batch, hipStream_t stream) {
    const RowGeom g = spmm_geom<V>(P, batch);
    const int64_t rpb = spmm_rows_per_block(g);
    P.rmul = spmm_row_mult(P.n_rows, P.nnz_per_item, rpb);
    P.nblocks = (P.n_rows + rpb * P.rmul - 1) / (rpb * P.rmul);
    if (P.nblocks > 0x7fffffffLL || batch > 65535 || g.col_tiles > 65535) return TSGU_ERR_TOO_LARGE;
    const dim3 grid((unsigned)P.nblocks, (unsigned)batch, (unsigned)g.col_tiles);
    const bool dot = P.dot_partial != nullptr;
    // the partial buffer is sized by tsgu_spmm_num_blocks(), which assumes the 16-byte geometry
    // whenever p allows it: refuse operands that would silently fall back to scalar lanes.
    if (dot && (P.p % VT<V>::kWide == 0) && g.vec == 1) return TSGU_ERR_BAD_ARG;
    const bool has_perm = P.perm != nullptr;
    if (dot && has_perm) return TSGU_ERR_BAD_ARG;
    if (P.ldb > 0xffffffffLL) return TSGU_ERR_TOO_LARGE;
    return dispatch_geom(g, [&](auto cl, auto ep) -> int {
        constexpr int CL = decltype(cl)::value, EP = decltype(ep)::value;
        constexpr int W = VT<V>::kWide;
#define TSGU_SPMM_GO(VECW, DOTF, PERMF)                                                                              \
    do {                                                                                                             \
        if (P.rmul > 1)                                                                                              \
            hipLaunchKernelGGL((csr_spmm_kernel<V, I, VECW, CL, EP, DOTF, PERMF, true>), grid, dim3(kBlock), 0, stream, P);  \
        else                                                                                                         \
            hipLaunchKernelGGL((csr_spmm_kernel<V, I, VECW, CL, EP, DOTF, PERMF, false>), grid, dim3(kBlock), 0, stream, P); \
    } while (0)
        if (g.vec == 1) {
            if (dot) TSGU_SPMM_GO(1, true, false);
            else if (has_perm) TSGU_SPMM_GO(1, false, true);
            else TSGU_SPMM_GO(1, false, false);
        } else {
            if (dot) TSGU_SPMM_GO(W, true, false);
            else if (has_perm) TSGU_SPMM_GO(W, false, true);
            else TSGU_SPMM_GO(W, false, false);
        }
#undef TSGU_SPMM_GO
        return check_launch();
    });
}

}  // namespace tsgu
