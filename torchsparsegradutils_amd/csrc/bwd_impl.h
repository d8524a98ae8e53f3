// Fused backward of C = A·B:  gradB = Aᵀ·G  and  gradA[k] = <G[row k,:], B[col k,:]>  in ONE pass.
//
// Both gradients need, for every stored entry (i, j), the upstream row G[i,:].  Run separately
// (K2 + K3) the backward gathers 2·nnz dense rows through L1 — and the L1/TA path is what bounds
// these kernels.  Walking the cached transposed pattern (rows of Aᵀ = columns j of A) the G rows of
// column j are gathered once and used twice: accumulated into gradB[j,:] (weighted by A's value,
// fetched through `perm`) and dotted with B[j,:] (held in registers) for gradA at position perm[k].
// Same lane geometry / LDS staging / determinism as K1; gradA leaves through LDS as a scatter of
// 4-byte stores addressed by perm.
#pragma once

#include "tsgu_common.h"

namespace tsgu {

constexpr int kBwdCap = 1024;  // staged entries per pass: {row idx, value, perm} + dot results = 16 KiB

struct BwdParams {
    int64_t n_rows_t;  // rows of Aᵀ = columns of A
    int64_t nnz_per_item, p;
    const void* tptr;   // [n_rows_t+1]
    const void* tidx;   // [nnz] row index in A of each transposed entry
    const void* tperm;  // [nnz] position of the entry in A's value array
    const void* val;    // A values
    const void* G;      // [n_rows_A][ldg]
    int64_t ldg, g_bs;
    const void* B;      // [n_cols_A][ldb]
    int64_t ldb, b_bs;
    void* gradA;        // [nnz] (A's order)
    void* gradB;        // [n_cols_A][ldo]
    int64_t ldo, o_bs;
    int64_t nblocks;
};

template <typename V, typename I, int VEC, int CL, int EP>
__global__ __launch_bounds__(kBlock) void csr_mm_backward_kernel(const BwdParams P) {
    using Acc = typename VT<V>::Acc;
    constexpr int GROUP = CL * EP;
    constexpr int RPB = kBlock / GROUP;
#ifndef TSGU_BWD_U
#define TSGU_BWD_U 4
#endif
    constexpr int U = TSGU_BWD_U;      // gathers issued back to back per lane
    static_assert(sizeof(V) <= 4, "fused backward is instantiated for 4-byte-or-narrower values");

    __shared__ uint2 s_ia[kBwdCap];    // {row index in A, value bits}
    __shared__ int s_q[kBwdCap];       // local position in A's value array (fits 31 bits per item)
    __shared__ float s_dot[kBwdCap];

    const int tid = threadIdx.x;
    const int grp = tid / GROUP;
    const int gl = tid % GROUP;
    const int cl = gl % CL;
    const int ep = gl / CL;
    const int64_t c0 = (int64_t)cl * VEC;
    const bool col_ok = c0 < P.p;
    const int64_t cc = col_ok ? c0 : 0;

    const int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    const int64_t item = blockIdx.y;
    const I* __restrict__ tptr = static_cast<const I*>(P.tptr) + item * (P.n_rows_t + 1);
    const I* __restrict__ tidx = static_cast<const I*>(P.tidx) + item * P.nnz_per_item;
    const I* __restrict__ tperm = static_cast<const I*>(P.tperm) + item * P.nnz_per_item;
    const V* __restrict__ val = static_cast<const V*>(P.val) + item * P.nnz_per_item;
    const V* __restrict__ G = static_cast<const V*>(P.G) + item * P.g_bs + cc;
    const V* __restrict__ B = static_cast<const V*>(P.B) + item * P.b_bs;
    V* __restrict__ gradA = static_cast<V*>(P.gradA) + item * P.nnz_per_item;
    const uint32_t ldg = (uint32_t)P.ldg;

    const int64_t row0 = vb * RPB;
    const int64_t row1 = row0 + RPB < P.n_rows_t ? row0 + RPB : P.n_rows_t;
    const int64_t row = row0 + grp;
    const bool row_ok = row < P.n_rows_t;

    const int64_t blk_begin = (int64_t)tptr[row0];
    const int64_t blk_end = (int64_t)tptr[row1];
    const int64_t start = row_ok ? (int64_t)tptr[row] : 0;
    const int64_t end = row_ok ? (int64_t)tptr[row + 1] : 0;

    Acc own[VEC], acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) own[v] = acc[v] = 0;
    if (row_ok && col_ok) load_vec<V, VEC>(B + row * P.ldb + c0, own);

    for (int64_t cs = blk_begin; cs < blk_end; cs += kBwdCap) {
        const int64_t ce = cs + kBwdCap < blk_end ? cs + kBwdCap : blk_end;
        if (cs != blk_begin) __syncthreads();
        // phase A: coalesced loads of (row index, perm) -> LDS
        for (int64_t base = cs + tid; base < ce; base += (int64_t)kBlock * 4) {
            I ri[4], q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t k = base + (int64_t)u * kBlock;
                const bool ok = k < ce;
                ri[u] = ok ? stream_load(tidx + k) : (I)0;
                q[u] = ok ? stream_load(tperm + k) : (I)0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t k = base + (int64_t)u * kBlock;
                if (k < ce) {
                    s_ia[k - cs].x = (unsigned)ri[u];
                    s_q[k - cs] = (int)q[u];
                }
            }
        }
        __syncthreads();
        // phase B: A's values through perm, "row-transposed" lane order: consecutive lanes take the same
        // local entry index t of consecutive rows.  Neighbouring rows of Aᵀ reference neighbouring entries
        // of the same row of A (adjacent positions of val), so the TA merges them into one L1 access instead
        // of one 64-byte access per 4-byte value.
        {
            const int xr = tid % RPB, xt = tid / RPB;
            const int64_t xrow = row0 + xr;
            int xlo = 0, xhi = 0;
            if (xrow < row1) {
                const int64_t a = (int64_t)tptr[xrow], b = (int64_t)tptr[xrow + 1];
                xlo = (int)((a > cs ? a : cs) - cs);
                xhi = (int)((b < ce ? b : ce) - cs);
            }
            constexpr int XT = kBlock / RPB;
            for (int kk = xlo + xt; kk < xhi; kk += XT * 2) {
                const int k2 = kk + XT;
                const int q0 = s_q[kk];
                const int q1 = k2 < xhi ? s_q[k2] : q0;
                const V v0 = val[q0];
                const V v1 = val[q1];
                s_ia[kk].y = __float_as_uint(VT<V>::up(v0));
                if (k2 < xhi) s_ia[k2].y = __float_as_uint(VT<V>::up(v1));
            }
        }
        __syncthreads();

        const int64_t lo = start > cs ? start : cs;
        const int64_t hi = end < ce ? end : ce;
        int i = (int)(lo - cs) + ep;
        const int iend = (int)(hi - cs);
        for (; i + (U - 1) * EP < iend; i += U * EP) {
            uint2 e[U];
            Acc g[U][VEC], d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = s_ia[i + u * EP];
#pragma unroll
            for (int u = 0; u < U; ++u) load_vec<V, VEC>(G + row_off((int)e[u].x, ldg), g[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const Acc a = __uint_as_float(e[u].y);
                d[u] = 0;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    acc[v] = fma(a, g[u][v], acc[v]);
                    d[u] = fma(own[v], g[u][v], d[u]);
                }
                d[u] = col_ok ? d[u] : (Acc)0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = group_total<Acc, CL>(d[u]);      // (total in the lanes cl >= group_total_lane: no LDS round trip)
            if (cl == group_total_lane<CL>()) {
#pragma unroll
                for (int u = 0; u < U; ++u) s_dot[i + u * EP] = d[u];
            }
        }
        for (; i < iend; i += EP) {
            const uint2 e = s_ia[i];
            Acc g[VEC];
            load_vec<V, VEC>(G + row_off((int)e.x, ldg), g);
            const Acc a = __uint_as_float(e.y);
            Acc d = 0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[v] = fma(a, g[v], acc[v]);
                d = fma(own[v], g[v], d);
            }
            d = col_ok ? d : (Acc)0;
            d = group_total<Acc, CL>(d);
            if (cl == group_total_lane<CL>()) s_dot[i] = d;
        }
        __syncthreads();
        // gradA[perm[k]] = <G[i,:], B[j,:]> : 4-byte scatter in the same row-transposed lane order
        {
            const int xr = tid % RPB, xt = tid / RPB;
            const int64_t xrow = row0 + xr;
            if (xrow < row1) {
                const int64_t a = (int64_t)tptr[xrow], b = (int64_t)tptr[xrow + 1];
                const int xlo = (int)((a > cs ? a : cs) - cs);
                const int xhi = (int)((b < ce ? b : ce) - cs);
                for (int kk = xlo + xt; kk < xhi; kk += kBlock / RPB) gradA[s_q[kk]] = VT<V>::down(s_dot[kk]);
            }
        }
    }

    if constexpr (EP > 1) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = ep_sum<Acc, CL, EP>(acc[v]);
    }
    if (row_ok && col_ok && ep == 0) {
        V* __restrict__ O = static_cast<V*>(P.gradB) + item * P.o_bs + row * P.ldo + c0;
        store_vec<V, VEC, true>(O, acc);
    }
}

template <typename V, typename I>
int bwd_launch(BwdParams P, int64_t batch, hipStream_t stream) {
    constexpr int wide = VT<V>::kWide;
    bool can = (P.p % wide == 0) && (P.ldg % wide == 0) && (P.ldb % wide == 0) && (P.ldo % wide == 0) &&
               aligned16(P.G) && aligned16(P.B) && aligned16(P.gradB);
    if (batch > 1) can = can && (P.g_bs % wide == 0) && (P.b_bs % wide == 0) && (P.o_bs % wide == 0);
    const RowGeom g = pick_geom(wide, can, P.p);
    if (g.col_tiles != 1) return TSGU_ERR_BAD_ARG;  // p > CL·VEC: caller uses K2 + K3
    const int64_t rpb = kBlock / (g.cl * g.ep);
    P.nblocks = (P.n_rows_t + rpb - 1) / rpb;
    if (P.nblocks > 0x7fffffffLL || batch > 65535 || P.ldg > 0xffffffffLL || P.nnz_per_item > 0x7fffffffLL)
        return TSGU_ERR_TOO_LARGE;
    const dim3 grid((unsigned)P.nblocks, (unsigned)batch, 1);
    return dispatch_geom(g, [&](auto cl, auto ep) -> int {
        constexpr int CL = decltype(cl)::value, EP = decltype(ep)::value;
        if (g.vec == 1)
            hipLaunchKernelGGL((csr_mm_backward_kernel<V, I, 1, CL, EP>), grid, dim3(kBlock), 0, stream, P);
        else
            hipLaunchKernelGGL((csr_mm_backward_kernel<V, I, wide, CL, EP>), grid, dim3(kBlock), 0, stream, P);
        return check_launch();
    });
}

}  // namespace tsgu
