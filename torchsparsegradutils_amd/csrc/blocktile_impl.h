// Workgroup-tiled gather kernels: the dense rows a block of sparse rows needs are brought into LDS ONCE
// (LDS-DMA, `global_load_lds_dwordx4`) and every stored entry then reads its row from LDS.
//
// Why: the row-gather kernels (spmm_impl.h, bwd_impl.h) are bounded by the L1/TA path, not by HBM — at C2
// every one of the 27e6 entries pulls a 128-byte dense row through L1 (3.5 GB per pass) although only
// 1e6 distinct rows exist.  A block of RPB consecutive sparse rows of a banded / stencil / mesh matrix
// references few distinct dense rows (C2, 32 rows: 864 entries, 306 distinct), so a per-block dictionary
// cuts the L1 traffic of the gather ~2.8x and replaces the 4-byte column index by a 16-bit local index.
//
// The fused backward additionally needs A's values and gradA addressed through the transposition
// permutation (4-byte scattered accesses).  The plan stores that permutation SORTED inside each block
// (`sperm`), so neighbouring lanes touch neighbouring words (entries (i, j), (i, j+1), ... of one row of A
// are adjacent) and each entry carries `spos`, the slot of its value in that sorted order.
//
// Plan layout (built once per sparsity pattern by _pattern.build_block_plan, all int32 / uint32):
//   ndist[nblocks]           distinct dense rows of block b
//   trow [nblocks][capd]     their indices (padded by repeating the last one)
//   ent  [nnz]               per stored entry, in the walked pattern's order: lidx | spos << 16
//   sperm[nnz]               Bwd / transposed SpMM: positions in A's value array, ascending inside each block
// fp32 values, p = CL*4 columns exactly covered by CL lanes of 16 bytes; everything else uses the gather kernels.
#pragma once

#include "tsgu_common.h"

namespace tsgu {

enum BtMode { kBtSpmm = 0, kBtBwd = 1 };

struct BtParams {
    int64_t n_rows, nnz, p;
    const void* ptr;       // [n_rows+1] entry offsets of the walked pattern (A for SpMM, Aᵀ for transposed SpMM / backward)
    const int* ndist;      // [nblocks]
    const int* trow;       // [nblocks][capd]
    const uint32_t* ent;   // [nnz]
    const int* sperm;      // [nnz] or null (values are then read in walked order)
    const float* val;      // A's values
    const float* S;        // gathered dense operand (B for SpMM, G for the backward)
    int64_t lds_;          // its leading dimension (elements)
    const float* Own;      // backward: B (row j is dotted with every gathered row)
    int64_t ldown;
    float* out;            // C / gradB
    int64_t ldo;
    float* gradA;          // backward: [nnz] in A's order
    int capd, ecap;
    int64_t nblocks;
};

typedef __attribute__((address_space(3))) void* bt_lds_ptr;
typedef const __attribute__((address_space(1))) void* bt_glb_ptr;

#ifndef TSGU_BT_NT
#define TSGU_BT_NT 1
#endif
constexpr int kBtStreamAux = TSGU_BT_NT ? 2 : 0;  // `nt`: single-use streams must not evict the dense rows from L2
constexpr int kBtMaxQ = 8;   // staged entries per block <= 8 * 256
constexpr int kBtMaxT = 16;  // tile DMA rounds per wave
constexpr int kBtMaxD = 4;   // dictionary entries per block <= 4 * 256 when the dense rows are gathered from global memory

template <typename I, int CL, int EP, int MODE, bool PERM, bool TILE>
__global__ __launch_bounds__(kBlock) void csr_blocktile_kernel(const BtParams P) {
    constexpr int VEC = 4;
    constexpr int GROUP = CL * EP;
    constexpr int RPB = kBlock / GROUP;
    constexpr int ROWF = CL * VEC;     // floats per tile row
    constexpr int RI = kWave / CL;     // tile rows per DMA wave-instruction
    constexpr int U = 4;
    static_assert(MODE == kBtSpmm || PERM, "the backward always walks the transposed pattern");

    // LDS: [tile: capd dense rows | or the capd row indices][entry words: ecap][values / dot results: ecap]
    extern __shared__ uint4 bt_smem[];
    float* tile = reinterpret_cast<float*>(bt_smem);
    int* s_trow = reinterpret_cast<int*>(bt_smem);
    uint32_t* s_ent = reinterpret_cast<uint32_t*>(bt_smem) + (TILE ? (size_t)P.capd * ROWF : (size_t)P.capd);
    float* s_val = reinterpret_cast<float*>(s_ent + P.ecap);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const int grp = tid / GROUP;
    const int gl = tid % GROUP;
    const int cl = gl % CL;
    const int ep = gl / CL;

    const int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    const I* __restrict__ ptr = static_cast<const I*>(P.ptr);
    const int64_t row0 = vb * RPB;
    const int64_t row1 = row0 + RPB < P.n_rows ? row0 + RPB : P.n_rows;
    const int64_t row = row0 + grp;
    const bool row_ok = row < P.n_rows;
    const int64_t e0 = (int64_t)ptr[row0];
    const int ne = (int)((int64_t)ptr[row1] - e0);
    const int nd = P.ndist[vb];
    const int lo = row_ok ? (int)((int64_t)ptr[row] - e0) : 0;
    const int hi = row_ok ? (int)((int64_t)ptr[row + 1] - e0) : 0;
    const int* __restrict__ trow = P.trow + vb * P.capd;

    // ---- phase A: everything the block needs goes to LDS by DMA --------------------------------
    // (1) TILE: indices of the distinct dense rows (RI per wave-instruction, the CL lanes of a row load the same word)
    int src[kBtMaxT];
    if constexpr (TILE) {
#pragma unroll
        for (int t = 0; t < kBtMaxT; ++t) {
            const int d0 = (t * 4 + wave) * RI;
            src[t] = 0;
            if (d0 < nd) {
                const int d = d0 + lane / CL;
                src[t] = stream_load(trow + (d < nd ? d : nd - 1));
            }
        }
    }
    // (2) permutation words for the value staging
    int qv[kBtMaxQ];
    if constexpr (PERM) {
#pragma unroll
        for (int q = 0; q < kBtMaxQ; ++q) {
            const int t = q * kBlock + tid;
            qv[q] = 0;
            if (q * kBlock < ne) qv[q] = t < ne ? stream_load(P.sperm + e0 + t) : 0;
        }
    }
    // (3) entry words: contiguous -> LDS
#pragma unroll
    for (int q = 0; q < kBtMaxQ; ++q) {
        const int t = q * kBlock + tid;
        if (q * kBlock < ne) {
            if (t < ne)
                __builtin_amdgcn_global_load_lds((bt_glb_ptr)(P.ent + e0 + t), (bt_lds_ptr)(s_ent + q * kBlock + wave * kWave), 4, 0, kBtStreamAux);
        }
    }
    // (4) the tile (TILE) or the block's row dictionary
    const uint32_t ld = (uint32_t)P.lds_;
    if constexpr (TILE) {
        const float* __restrict__ S = P.S + (lane % CL) * VEC;
#pragma unroll
        for (int t = 0; t < kBtMaxT; ++t) {
            const int d0 = (t * 4 + wave) * RI;
            if (d0 < nd)
                __builtin_amdgcn_global_load_lds((bt_glb_ptr)(S + row_off(src[t], ld)), (bt_lds_ptr)(tile + d0 * ROWF), 16, 0, 0);
        }
    } else {
#pragma unroll
        for (int q = 0; q < kBtMaxD; ++q) {
            const int t = q * kBlock + tid;
            if (q * kBlock < nd) {
                if (t < nd)
                    __builtin_amdgcn_global_load_lds((bt_glb_ptr)(trow + t), (bt_lds_ptr)(s_trow + q * kBlock + wave * kWave), 4, 0, kBtStreamAux);
            }
        }
    }
    // (5) values (through the sorted permutation, or contiguous)
#pragma unroll
    for (int q = 0; q < kBtMaxQ; ++q) {
        const int t = q * kBlock + tid;
        if (q * kBlock < ne) {
            if (t < ne) {
                const float* vsrc = PERM ? P.val + qv[q] : P.val + e0 + t;
                __builtin_amdgcn_global_load_lds((bt_glb_ptr)vsrc, (bt_lds_ptr)(s_val + q * kBlock + wave * kWave), 4, 0, PERM ? 0 : kBtStreamAux);
            }
        }
    }
    float own[VEC], acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) own[v] = acc[v] = 0.f;
    if constexpr (MODE == kBtBwd) {
        if (row_ok) load_vec<float, VEC>(P.Own + row * P.ldown + cl * VEC, own);
    }
    __syncthreads();

    // ---- phase B: rows of the block; dense rows come from the LDS tile (TILE) or from global memory ------
    const float* trd = tile + cl * VEC;
    const float* __restrict__ Sg = P.S + cl * VEC;
    int i = lo + ep;
    for (; i + (U - 1) * EP < hi; i += U * EP) {
        uint32_t w[U];
        float a[U], d[U];
        float g[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = s_ent[i + u * EP];
        if constexpr (TILE) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 t4 = *reinterpret_cast<const float4*>(trd + (w[u] & 0xffffu) * ROWF);
                g[u][0] = t4.x, g[u][1] = t4.y, g[u][2] = t4.z, g[u][3] = t4.w;
            }
        } else {
            int r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = s_trow[w[u] & 0xffffu];
#pragma unroll
            for (int u = 0; u < U; ++u) load_vec<float, VEC>(Sg + row_off(r[u], ld), g[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = s_val[PERM ? (int)(w[u] >> 16) : i + u * EP];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = fma(a[u], g[u][v], acc[v]);
            if constexpr (MODE == kBtBwd) {
                d[u] = own[0] * g[u][0];
#pragma unroll
                for (int v = 1; v < VEC; ++v) d[u] = fma(own[v], g[u][v], d[u]);
            }
        }
        if constexpr (MODE == kBtBwd) {
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = group_sum<float, CL>(d[u]);
            if (cl == 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) s_val[w[u] >> 16] = d[u];
            }
        }
    }
    for (; i < hi; i += EP) {
        const uint32_t w = s_ent[i];
        float g[VEC];
        if constexpr (TILE) {
            const float4 t4 = *reinterpret_cast<const float4*>(trd + (w & 0xffffu) * ROWF);
            g[0] = t4.x, g[1] = t4.y, g[2] = t4.z, g[3] = t4.w;
        } else {
            load_vec<float, VEC>(Sg + row_off(s_trow[w & 0xffffu], ld), g);
        }
        const float a = s_val[PERM ? (int)(w >> 16) : i];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fma(a, g[v], acc[v]);
        if constexpr (MODE == kBtBwd) {
            float d = own[0] * g[0];
#pragma unroll
            for (int v = 1; v < VEC; ++v) d = fma(own[v], g[v], d);
            d = group_sum<float, CL>(d);
            if (cl == 0) s_val[w >> 16] = d;
        }
    }

    if constexpr (EP > 1) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = ep_sum<float, CL, EP>(acc[v]);
    }
    if (row_ok && ep == 0) store_vec<float, VEC, true>(P.out + row * P.ldo + cl * VEC, acc);

    if constexpr (MODE == kBtBwd) {
        // gradA leaves in the sorted-permutation order: neighbouring lanes write neighbouring words
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kBtMaxQ; ++q) {
            const int t = q * kBlock + tid;
            if (q * kBlock < ne) {
                if (t < ne) P.gradA[qv[q]] = s_val[t];
            }
        }
    }
}

template <typename I, int CL, int MODE, bool PERM, bool TILE>
int bt_launch_cl(const BtParams& P, int rpb, hipStream_t stream) {
    const size_t lds = (TILE ? (size_t)P.capd * CL * 16 : (size_t)P.capd * 4) + (size_t)P.ecap * 8;
    const dim3 grid((unsigned)P.nblocks), block(kBlock);
#define TSGU_BT_CASE(EPV)                                                                                        \
    if (rpb * CL * EPV == kBlock) {                                                                              \
        hipLaunchKernelGGL((csr_blocktile_kernel<I, CL, EPV, MODE, PERM, TILE>), grid, block, lds, stream, P);   \
        return check_launch();                                                                                   \
    }
    TSGU_BT_CASE(1)
    TSGU_BT_CASE(2)
    TSGU_BT_CASE(4)
#undef TSGU_BT_CASE
    return TSGU_ERR_BAD_ARG;
}

template <typename I, int MODE, bool PERM, bool TILE>
int bt_launch_t(BtParams P, int rpb, hipStream_t stream) {
    if (P.p % 4 != 0 || P.lds_ % 4 != 0 || P.ldo % 4 != 0 || !aligned16(P.S) || !aligned16(P.out)) return TSGU_ERR_BAD_ARG;
    if (MODE == kBtBwd && (P.ldown % 4 != 0 || !aligned16(P.Own))) return TSGU_ERR_BAD_ARG;
    const int64_t cl = P.p / 4;
    if (rpb <= 0 || P.ecap > kBtMaxQ * kBlock || P.ecap % kBlock != 0 || P.lds_ > 0xffffffffLL) return TSGU_ERR_BAD_ARG;
    P.nblocks = (P.n_rows + rpb - 1) / rpb;
    if (P.nblocks > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    if (P.nblocks == 0) return TSGU_OK;
    const int64_t ri = cl > 0 ? kWave / cl : 0;
    if (ri == 0 || P.capd % 4 != 0) return TSGU_ERR_BAD_ARG;
    if (TILE && (P.capd % ri != 0 || P.capd > kBtMaxT * 4 * ri)) return TSGU_ERR_BAD_ARG;
    if (!TILE && P.capd > kBtMaxD * kBlock) return TSGU_ERR_BAD_ARG;
    if ((TILE ? (size_t)P.capd * cl * 16 : (size_t)P.capd * 4) + (size_t)P.ecap * 8 > 64 * 1024) return TSGU_ERR_TOO_LARGE;
    switch (cl) {
        case 4: return bt_launch_cl<I, 4, MODE, PERM, TILE>(P, rpb, stream);
        case 8: return bt_launch_cl<I, 8, MODE, PERM, TILE>(P, rpb, stream);
        case 16: return bt_launch_cl<I, 16, MODE, PERM, TILE>(P, rpb, stream);
    }
    return TSGU_ERR_BAD_ARG;
}

template <typename I, int MODE, bool PERM>
int bt_launch(const BtParams& P, int rpb, int tile, hipStream_t stream) {
    return tile ? bt_launch_t<I, MODE, PERM, true>(P, rpb, stream) : bt_launch_t<I, MODE, PERM, false>(P, rpb, stream);
}

}  // namespace tsgu
