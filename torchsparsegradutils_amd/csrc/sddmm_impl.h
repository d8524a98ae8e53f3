// K3: sparsity-masked SDDMM on gfx950.  out[k] = alpha · <R[row(k),:], Cm[col(k),:]> for
// every stored entry k.  Same ownership as K1 (a CL×EP lane group per row, a 256-thread
// workgroup per run of consecutive rows): the row operand is read once per row into
// registers, the column indices are staged through LDS with coalesced loads, the column
// operand rows are gathered with 16-byte loads, the per-entry dot is reduced across the
// CL column lanes with DPP/xor shuffles, and the results leave through LDS so that the
// nnz-long output is written with full-width coalesced stores.  No nnz×p temporaries.
#pragma once

#include "tsgu_common.h"

namespace tsgu {

constexpr int kSddmmCap = 2048;

struct SddmmParams {
    int64_t n_rows, nnz_per_item, p;
    const void* crow;
    const void* col;
    const void* R;  // row operand  [n_rows][ldr]
    int64_t ldr, r_bs;
    const void* Cm;  // column operand [n_cols][ldc]
    int64_t ldc, c_bs;
    void* out;
    double alpha;
    int64_t nblocks;
    int64_t col_tiles;
};

// (the dot of an entry: its total is needed by the ONE lane that stores it — cl == group_total_lane<CL>())
template <typename Acc, int CL>
__device__ __forceinline__ Acc reduce_cl(Acc x) {
    return group_total<Acc, CL>(x);
}

template <typename V, typename I, int VEC, int CL, int EP>
__global__ __launch_bounds__(kBlock) void csr_sddmm_kernel(const SddmmParams P) {
    using Acc = typename VT<V>::Acc;
    constexpr int GROUP = CL * EP;
    constexpr int RPB = kBlock / GROUP;
    constexpr int TW = CL * VEC;
    constexpr int U = 4;

    __shared__ int s_col[kSddmmCap];
    __shared__ Acc s_out[kSddmmCap];

    const int tid = threadIdx.x;
    const int grp = tid / GROUP;
    const int gl = tid % GROUP;
    const int cl = gl % CL;
    const int ep = gl / CL;

    const int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    const int64_t item = blockIdx.y;

    const I* __restrict__ crow = static_cast<const I*>(P.crow) + item * (P.n_rows + 1);
    const I* __restrict__ col = static_cast<const I*>(P.col) + item * P.nnz_per_item;
    const V* __restrict__ R = static_cast<const V*>(P.R) + item * P.r_bs;
    const V* __restrict__ Cm = static_cast<const V*>(P.Cm) + item * P.c_bs;
    V* __restrict__ out = static_cast<V*>(P.out) + item * P.nnz_per_item;

    const int64_t row0 = vb * RPB;
    const int64_t row1 = row0 + RPB < P.n_rows ? row0 + RPB : P.n_rows;
    const int64_t row = row0 + grp;
    const bool row_ok = row < P.n_rows;

    const int64_t blk_begin = (int64_t)crow[row0];
    const int64_t blk_end = (int64_t)crow[row1];
    const int64_t start = row_ok ? (int64_t)crow[row] : 0;
    const int64_t end = row_ok ? (int64_t)crow[row + 1] : 0;

    const int64_t c0 = (int64_t)cl * VEC;
    const bool single_tile = P.col_tiles == 1;
    const bool lane_ok0 = c0 < P.p;

    Acc r0[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) r0[v] = 0;
    if (row_ok && lane_ok0) load_vec<V, VEC>(R + row * P.ldr + c0, r0);

    const Acc alpha = (Acc)P.alpha;
    const uint32_t ldc = (uint32_t)P.ldc;

    for (int64_t cs = blk_begin; cs < blk_end; cs += kSddmmCap) {
        const int64_t ce = cs + kSddmmCap < blk_end ? cs + kSddmmCap : blk_end;
        if (cs != blk_begin) __syncthreads();
        for (int64_t base = cs + tid; base < ce; base += (int64_t)kBlock * 4) {
            I cj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t k = base + (int64_t)u * kBlock;
                cj[u] = k < ce ? col[k] : (I)0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t k = base + (int64_t)u * kBlock;
                if (k < ce) s_col[k - cs] = (int)cj[u];
            }
        }
        __syncthreads();

        const int64_t lo = start > cs ? start : cs;
        const int64_t hi = end < ce ? end : ce;
        int i = (int)(lo - cs) + ep;
        const int iend = (int)(hi - cs);

        if (single_tile) {
            // Lanes whose columns lie beyond p (p not a multiple of CL·VEC) read columns 0.. instead
            // and their partial is discarded: every load stays unconditional, so the U gathers of a
            // pass are all in flight together.
            const int64_t cc = lane_ok0 ? c0 : 0;
            for (; i + (U - 1) * EP < iend; i += U * EP) {
                Acc b[U][VEC];
                Acc d[U];
                int j[U];
#pragma unroll
                for (int u = 0; u < U; ++u) j[u] = s_col[i + u * EP];
#pragma unroll
                for (int u = 0; u < U; ++u) load_vec<V, VEC>(Cm + row_off(j[u], ldc) + cc, b[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    d[u] = 0;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) d[u] = fma(r0[v], b[u][v], d[u]);
                    d[u] = lane_ok0 ? d[u] : (Acc)0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) d[u] = reduce_cl<Acc, CL>(d[u]);
                if (cl == group_total_lane<CL>()) {
#pragma unroll
                    for (int u = 0; u < U; ++u) s_out[i + u * EP] = d[u];
                }
            }
            for (; i < iend; i += EP) {
                Acc b[VEC];
                const int j = s_col[i];
                load_vec<V, VEC>(Cm + row_off(j, ldc) + cc, b);
                Acc d = 0;
#pragma unroll
                for (int v = 0; v < VEC; ++v) d = fma(r0[v], b[v], d);
                d = lane_ok0 ? d : (Acc)0;
                d = reduce_cl<Acc, CL>(d);
                if (cl == group_total_lane<CL>()) s_out[i] = d;
            }
        } else {
            // wide RHS (p > CL·VEC): walk the column tiles per entry; the row operand is
            // re-read from L1 instead of being held in registers.
            for (; i < iend; i += EP) {
                const int j = s_col[i];
                Acc d = 0;
                for (int64_t t = 0; t < P.col_tiles; ++t) {
                    const int64_t c = t * TW + c0;
                    if (c < P.p) {
                        Acc rr[VEC], b[VEC];
                        load_vec<V, VEC>(R + row * P.ldr + c, rr);
                        load_vec<V, VEC>(Cm + row_off(j, ldc) + c, b);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) d = fma(rr[v], b[v], d);
                    }
                }
                d = reduce_cl<Acc, CL>(d);
                if (cl == group_total_lane<CL>()) s_out[i] = d;
            }
        }
        __syncthreads();
        for (int64_t k = cs + tid; k < ce; k += kBlock) out[k] = VT<V>::down(alpha * s_out[k - cs]);
    }
}

// COO flavour: explicit row indices in any order, one CL-lane group per entry.
struct CooSddmmParams {
    int64_t nnz, p;
    const void* row;
    const void* col;
    const void* R;
    int64_t ldr;
    const void* Cm;
    int64_t ldc;
    void* out;
    double alpha;
    int64_t col_tiles;
};

template <typename V, typename I, int VEC, int CL>
__global__ __launch_bounds__(kBlock) void coo_sddmm_kernel(const CooSddmmParams P) {
    using Acc = typename VT<V>::Acc;
    constexpr int EPB = kBlock / CL;  // entries per block pass
    constexpr int TW = CL * VEC;
    const int tid = threadIdx.x;
    const int cl = tid % CL;
    const int64_t c0 = (int64_t)cl * VEC;
    const I* __restrict__ row = static_cast<const I*>(P.row);
    const I* __restrict__ col = static_cast<const I*>(P.col);
    const V* __restrict__ R = static_cast<const V*>(P.R);
    const V* __restrict__ Cm = static_cast<const V*>(P.Cm);
    V* __restrict__ out = static_cast<V*>(P.out);
    const Acc alpha = (Acc)P.alpha;
    for (int64_t e = (int64_t)blockIdx.x * EPB + tid / CL; e < ((P.nnz + EPB - 1) / EPB) * EPB;
         e += (int64_t)gridDim.x * EPB) {
        const bool ok = e < P.nnz;
        const int64_t i = ok ? (int64_t)row[e] : 0;
        const int64_t j = ok ? (int64_t)col[e] : 0;
        Acc d = 0;
        for (int64_t t = 0; t < P.col_tiles; ++t) {
            const int64_t c = t * TW + c0;
            if (ok && c < P.p) {
                Acc rr[VEC], b[VEC];
                load_vec<V, VEC>(R + i * P.ldr + c, rr);
                load_vec<V, VEC>(Cm + j * P.ldc + c, b);
#pragma unroll
                for (int v = 0; v < VEC; ++v) d = fma(rr[v], b[v], d);
            }
        }
        d = reduce_cl<Acc, CL>(d);
        if (ok && cl == group_total_lane<CL>()) out[e] = VT<V>::down(alpha * d);
    }
}

template <typename V, typename I>
int sddmm_launch(SddmmParams P, int64_t batch, hipStream_t stream) {
    constexpr int wide = VT<V>::kWide;
    bool can = (P.p % wide == 0) && (P.ldr % wide == 0) && (P.ldc % wide == 0) && aligned16(P.R) && aligned16(P.Cm);
    if (batch > 1) can = can && (P.r_bs % wide == 0) && (P.c_bs % wide == 0);
    const RowGeom g = pick_geom(wide, can, P.p);
    const int64_t rpb = kBlock / (g.cl * g.ep);
    P.nblocks = (P.n_rows + rpb - 1) / rpb;
    P.col_tiles = g.col_tiles;
    if (P.nblocks > 0x7fffffffLL || batch > 65535 || P.ldc > 0xffffffffLL) return TSGU_ERR_TOO_LARGE;
    const dim3 grid((unsigned)P.nblocks, (unsigned)batch, 1);
    return dispatch_geom(g, [&](auto cl, auto ep) -> int {
        constexpr int CL = decltype(cl)::value, EP = decltype(ep)::value;
        if (g.vec == 1)
            hipLaunchKernelGGL((csr_sddmm_kernel<V, I, 1, CL, EP>), grid, dim3(kBlock), 0, stream, P);
        else
            hipLaunchKernelGGL((csr_sddmm_kernel<V, I, wide, CL, EP>), grid, dim3(kBlock), 0, stream, P);
        return check_launch();
    });
}

template <typename V, typename I>
int coo_sddmm_launch(CooSddmmParams P, hipStream_t stream) {
    constexpr int wide = VT<V>::kWide;
    const bool can = (P.p % wide == 0) && (P.ldr % wide == 0) && (P.ldc % wide == 0) && aligned16(P.R) && aligned16(P.Cm);
    const RowGeom g = pick_geom(wide, can, P.p);
    P.col_tiles = g.col_tiles;
    const int64_t epb = kBlock / g.cl;
    int64_t nb = (P.nnz + epb - 1) / epb;
    if (nb > 1 << 20) nb = 1 << 20;  // grid-stride beyond ~1M blocks
    const dim3 grid((unsigned)nb, 1, 1);
    return dispatch_geom(g, [&](auto cl, auto) -> int {
        constexpr int CL = decltype(cl)::value;
        if (g.vec == 1)
            hipLaunchKernelGGL((coo_sddmm_kernel<V, I, 1, CL>), grid, dim3(kBlock), 0, stream, P);
        else
            hipLaunchKernelGGL((coo_sddmm_kernel<V, I, wide, CL>), grid, dim3(kBlock), 0, stream, P);
        return check_launch();
    });
}

}  // namespace tsgu
