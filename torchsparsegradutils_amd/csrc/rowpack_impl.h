// Row-pair gather kernels ("rowpack"): one lane group owns TWO consecutive sparse rows and walks the UNION of
// their column sets, so a dense row that both sparse rows reference is gathered once and used twice.
//
// Why: the gather kernels are bounded by the L1/TA path — every stored entry pulls a 128-byte dense row
// through L1 (C2: 27e6 x 128 B per pass) — while neighbouring rows of stencil / banded / mesh matrices share
// most of their columns (27-point stencil: rows j and j+1 share 18 of 27).  Walking the union (36 gathers per
// row pair instead of 54) removes a third of that traffic without any LDS tile, i.e. at full occupancy.
//
// Plan (built once per sparsity pattern by _pattern.build_rowpack_plan):
//   uptr [npairs+1]  int32   union-entry offsets per row pair, npairs = ceil(n_rows / 2)
//   ucol [nu]        int32   dense-row index of each union entry (ascending inside a pair)
//   upos [nu]        uint32  (permuted walks) two 16-bit halves, one per row of the pair: slot of that row's value inside the
//                            workgroup's staged value slice, bit 15 set = the row has no entry in this column
//                            Walks in stored order carry no upos: bits 30 / 31 of ucol say which rows own the column
//                            and the slots are consecutive per row.
//   sperm[nnz]       int32   (patterns walked through a permutation) positions in the value array, ascending
//                            inside each workgroup's entry range; the slots of `upos` index this order.  NULL
//                            when the values are in the walked order (slot = entry offset inside the workgroup).
// A workgroup of 256 threads = 256/CL lane groups covers 2*256/CL consecutive rows; its entries (ptr range) are
// staged once: union records and values by LDS-DMA.  Each row's sum still runs over its own entries in ascending
// column order, so results are bit-identical to the plain gather kernels with one lane group per row.
// Non-finite inputs: a row never touches a dense row it does not reference (the second row's update is
// predicated, not multiplied by zero).
#pragma once

#include "tsgu_common.h"

namespace tsgu {

enum RpMode { kRpSpmm = 0, kRpBwd = 1, kRpSddmm = 2 };

struct RpParams {
    int64_t n_rows, nnz, p;
    int64_t n_src;          // rows of the gathered dense operand (= columns of the walked pattern)
    const void* ptr;        // [n_rows+1] entry offsets of the walked pattern
    const int* uptr;        // [npairs+1]
    const int* ucol;        // [nu]
    const uint32_t* upos;   // [nu]
    const int* sperm;       // [nnz] or null
    const int* order;       // [nblocks] or null: workgroup b processes row block order[b] (a permutation; speed only)
    const int* vpair;       // [nblocks*GPB] or null: row pair owned by each lane-group slot (-1 = none): lets a workgroup own
                            // any set of pairs (e.g. a 3-D brick of a lattice) instead of consecutive ones; needs eptr + sperm
    const int* eptr;        // [nblocks+1] with vpair: start of each workgroup's entries in sperm
    const float* val;
    const float* S;         // gathered dense operand (B for SpMM, G for the backward)
    int64_t lds_;
    const float* Own;       // backward: B
    int64_t ldown;
    float* out;             // C / gradB
    int64_t ldo;
    float* gradA;           // backward: [nnz] in A's order; SDDMM: [nnz] output in walked order
    float alpha;            // SDDMM scale
    int ecap, ucap;         // LDS capacities: staged values / union records per workgroup
    int64_t nblocks;
};

typedef __attribute__((address_space(3))) void* rp_lds_ptr;
typedef const __attribute__((address_space(1))) void* rp_glb_ptr;

constexpr int kRpMaxQ = 8;    // staged values per workgroup <= 8 * 256
constexpr int kRpMaxU = 12;   // union records per workgroup <= 12 * 256
constexpr int kRpAbsent = 0x8000;

#ifndef TSGU_RP_WAVES
#define TSGU_RP_WAVES 0
#endif
#if TSGU_RP_WAVES > 0
#define TSGU_RP_OCC __attribute__((amdgpu_waves_per_eu(TSGU_RP_WAVES, TSGU_RP_WAVES)))
#else
#define TSGU_RP_OCC
#endif

template <typename I, int CL, int MODE, bool PERM, bool SMALL>
__global__ __launch_bounds__(kBlock) TSGU_RP_OCC void csr_rowpack_kernel(const RpParams P) {
    constexpr int VEC = 4;
    constexpr int GPB = kBlock / CL;  // lane groups (row pairs) per workgroup
    constexpr int RPB = 2 * GPB;      // rows per workgroup
#ifndef TSGU_RP_U
#define TSGU_RP_U 4
#endif
#ifndef TSGU_RP_UB
#define TSGU_RP_UB 0
#endif


    constexpr int U = (MODE == kRpBwd && TSGU_RP_UB > 0) ? TSGU_RP_UB : TSGU_RP_U;  // gathers in flight per lane
    static_assert(MODE != kRpBwd || PERM, "the backward always walks the transposed pattern");
    static_assert(MODE != kRpSddmm || !PERM, "SDDMM walks the pattern in stored order");

    extern __shared__ uint4 rp_smem[];
    int* s_ucol = reinterpret_cast<int*>(rp_smem);
    uint32_t* s_upos = reinterpret_cast<uint32_t*>(s_ucol + P.ucap);           // permuted walks only
    float* s_val = reinterpret_cast<float*>(s_ucol + (PERM ? 2 : 1) * (size_t)P.ucap);

    const int tid = threadIdx.x;
    const int wave = tid / kWave;
    const int grp = tid / CL;
    const int cl = tid % CL;

    int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    if (P.order) vb = P.order[vb];
    const I* __restrict__ ptr = static_cast<const I*>(P.ptr);
    const int64_t npairs = (P.n_rows + 1) / 2;
    const int64_t nslots = P.nblocks * GPB;  // uptr is indexed by lane-group slot (= pair index when vpair is null)
    const int64_t pair0 = vb * GPB;
    const int64_t pair1 = P.vpair ? pair0 + GPB : (pair0 + GPB < npairs ? pair0 + GPB : npairs);
    const int64_t slot_id = pair0 + grp;
    int64_t pair = slot_id;
    if (P.vpair) pair = slot_id < nslots ? (int64_t)P.vpair[slot_id] : -1;
    const bool pair_ok = pair >= 0 && pair < npairs;
    const int64_t ra = 2 * pair, rb = 2 * pair + 1;
    const bool b_ok = rb < P.n_rows;

    int64_t e0;
    int ne;
    if (P.vpair) {
        e0 = (int64_t)P.eptr[vb];
        ne = (int)((int64_t)P.eptr[vb + 1] - e0);
    } else {
        const int64_t row0 = vb * RPB;
        const int64_t row1 = row0 + RPB < P.n_rows ? row0 + RPB : P.n_rows;
        e0 = (int64_t)ptr[row0];
        ne = (int)((int64_t)ptr[row1] - e0);
    }
    const int64_t u0 = (int64_t)P.uptr[pair0];
    const int nu = (int)((int64_t)P.uptr[pair1] - u0);
    const int lo = pair_ok ? (int)((int64_t)P.uptr[slot_id] - u0) : 0;
    const int hi = pair_ok ? (int)((int64_t)P.uptr[slot_id + 1] - u0) : 0;

    // ---- phase A: union records and values of the workgroup's rows -> LDS (DMA, no VGPR round trip) ----
    int qv[kRpMaxQ];
    if constexpr (PERM) {
#pragma unroll
        for (int q = 0; q < kRpMaxQ; ++q) {
            const int t = q * kBlock + tid;
            qv[q] = 0;
            if (q * kBlock < ne) qv[q] = t < ne ? stream_load(P.sperm + e0 + t) : 0;
        }
    }
#pragma unroll
    for (int q = 0; q < kRpMaxU; ++q) {
        const int t = q * kBlock + tid;
        if (q * kBlock < nu) {
            if (t < nu) {
                __builtin_amdgcn_global_load_lds((rp_glb_ptr)(P.ucol + u0 + t), (rp_lds_ptr)(s_ucol + q * kBlock + wave * kWave), 4, 0, 2);
                if constexpr (PERM)
                    __builtin_amdgcn_global_load_lds((rp_glb_ptr)(P.upos + u0 + t), (rp_lds_ptr)(s_upos + q * kBlock + wave * kWave), 4, 0, 2);
            }
        }
    }
    if constexpr (MODE != kRpSddmm) {  // SDDMM reads no values
#pragma unroll
    for (int q = 0; q < kRpMaxQ; ++q) {
        const int t = q * kBlock + tid;
        if (q * kBlock < ne) {
            if (t < ne) {
                const float* vsrc = PERM ? P.val + qv[q] : P.val + e0 + t;
                __builtin_amdgcn_global_load_lds((rp_glb_ptr)vsrc, (rp_lds_ptr)(s_val + q * kBlock + wave * kWave), 4, 0, PERM ? 0 : 2);
            }
        }
    }
    }
    float own_a[VEC], own_b[VEC], acc_a[VEC], acc_b[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) own_a[v] = own_b[v] = acc_a[v] = acc_b[v] = 0.f;
    if constexpr (MODE != kRpSpmm) {
        if (pair_ok) load_vec<float, VEC>(P.Own + ra * P.ldown + cl * VEC, own_a);
        if (pair_ok && b_ok) load_vec<float, VEC>(P.Own + rb * P.ldown + cl * VEC, own_b);
    }
    __syncthreads();

    // ---- phase B: walk the union of the pair's columns; one gather serves both rows ----
    // dense row c starts at byte c·ld·4 of a wave-uniform base: when the operand is smaller than 4 GiB and the factors
    // fit 24 bits (launcher checks) the offset is ONE full-rate v_mad_u32_u24 and the load uses the SGPR-base form,
    // instead of a quarter-rate 64-bit multiply-add + 64-bit shift-add per gather (the phase is VALU-issue-bound)
    const char* __restrict__ Sbase = reinterpret_cast<const char*>(P.S);
    const uint32_t ldb4 = (uint32_t)P.lds_ * 4u, cl16 = (uint32_t)cl * 16u;
    auto gather = [&](int c, float (&g)[VEC]) {
        if constexpr (SMALL) {
            const uint32_t boff = __umul24((uint32_t)c, ldb4) + cl16;
            load_vec<float, VEC>(reinterpret_cast<const float*>(Sbase + boff), g);
        } else {
            load_vec<float, VEC>(P.S + cl * VEC + row_off(c, (uint32_t)P.lds_), g);
        }
    };

    auto use = [&](const float (&g)[VEC], uint32_t half, float (&acc)[VEC], const float (&own)[VEC]) {
        if (!(half & kRpAbsent)) {  // uniform inside the lane group, divergent across the wave: exec-masked
            const float a = s_val[half];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = fma(a, g[v], acc[v]);
            if constexpr (MODE == kRpBwd) {
                float d = own[0] * g[0];
#pragma unroll
                for (int v = 1; v < VEC; ++v) d = fma(own[v], g[v], d);
                d = group_sum<float, CL>(d);
                if (cl == 0) s_val[half] = d;
            }
        }
    };

    int i = lo;
    if constexpr (PERM) {
        for (; i + U <= hi; i += U) {
            int c[U];
            uint32_t w[U];
            float g[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = s_ucol[i + u];
                w[u] = s_upos[i + u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) gather(c[u], g[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                use(g[u], w[u] & 0xffffu, acc_a, own_a);
                use(g[u], w[u] >> 16, acc_b, own_b);
            }
        }
        for (; i < hi; ++i) {
            const int c = s_ucol[i];
            const uint32_t w = s_upos[i];
            float g[VEC];
            gather(c, g);
            use(g, w & 0xffffu, acc_a, own_a);
            use(g, w >> 16, acc_b, own_b);
        }
    } else {
        // values in walked order: the slots of a row are consecutive, so the record only says WHICH rows own the column
        // (bits 30 / 31 of ucol) and two running counters replace the slot words (no upos stream, half the record LDS)
        int ka = pair_ok ? (int)((int64_t)ptr[ra] - e0) : 0;
        int kb = (pair_ok && b_ok) ? (int)((int64_t)ptr[rb] - e0) : 0;
        auto use_seq = [&](const float (&g)[VEC], bool present, int& k, float (&acc)[VEC], const float (&own)[VEC]) {
            if (present) {
                if constexpr (MODE == kRpSddmm) {
                    // gradient of the stored entry: <row operand, gathered column operand>, into the entry's slot
                    float d = own[0] * g[0];
#pragma unroll
                    for (int v = 1; v < VEC; ++v) d = fma(own[v], g[v], d);
                    d = group_sum<float, CL>(d);
                    if (cl == 0) s_val[k] = d;
                } else {
                    const float a = s_val[k];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = fma(a, g[v], acc[v]);
                }
                ++k;
            }
        };
        for (; i + U <= hi; i += U) {
            uint32_t w[U];
            float g[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = (uint32_t)s_ucol[i + u];
#pragma unroll
            for (int u = 0; u < U; ++u) gather((int)(w[u] & 0x3fffffffu), g[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                use_seq(g[u], (w[u] >> 30) & 1u, ka, acc_a, own_a);
                use_seq(g[u], w[u] >> 31, kb, acc_b, own_b);
            }
        }
        for (; i < hi; ++i) {
            const uint32_t w = (uint32_t)s_ucol[i];
            float g[VEC];
            gather((int)(w & 0x3fffffffu), g);
            use_seq(g, (w >> 30) & 1u, ka, acc_a, own_a);
            use_seq(g, w >> 31, kb, acc_b, own_b);
        }
    }

    if constexpr (MODE != kRpSddmm) {
        if (pair_ok) {
            store_vec<float, VEC, true>(P.out + ra * P.ldo + cl * VEC, acc_a);
            if (b_ok) store_vec<float, VEC, true>(P.out + rb * P.ldo + cl * VEC, acc_b);
        }
    } else {
        // the block's gradients sit in stored order in LDS: one coalesced, streaming write
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kRpMaxQ; ++q) {
            const int t = q * kBlock + tid;
            if (q * kBlock < ne) {
                if (t < ne) __builtin_nontemporal_store(P.alpha * s_val[t], P.gradA + e0 + t);
            }
        }
    }

    if constexpr (MODE == kRpBwd) {
        // gradA leaves in the sorted-permutation order: neighbouring lanes write neighbouring words
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kRpMaxQ; ++q) {
            const int t = q * kBlock + tid;
            if (q * kBlock < ne) {
                if (t < ne) P.gradA[qv[q]] = s_val[t];
            }
        }
    }
}

template <typename I, int MODE, bool PERM>
int rp_launch(RpParams P, hipStream_t stream) {
    if (P.p % 4 != 0 || P.lds_ % 4 != 0 || !aligned16(P.S)) return TSGU_ERR_BAD_ARG;
    if (MODE != kRpSddmm && (P.ldo % 4 != 0 || !aligned16(P.out))) return TSGU_ERR_BAD_ARG;
    if (MODE != kRpSpmm && (P.ldown % 4 != 0 || !aligned16(P.Own))) return TSGU_ERR_BAD_ARG;
    const int64_t cl = P.p / 4;
    if (cl != 4 && cl != 8 && cl != 16) return TSGU_ERR_BAD_ARG;
    if (P.ecap <= 0 || P.ucap <= 0 || P.ecap > kRpMaxQ * kBlock || P.ucap > kRpMaxU * kBlock || P.ecap >= kRpAbsent ||
        P.ucap % 4 != 0 || P.lds_ > 0xffffffffLL)
        return TSGU_ERR_BAD_ARG;
    const int64_t rpb = 2 * (kBlock / cl);
    if (P.vpair) {
        if (!PERM || !P.eptr || P.nblocks <= 0) return TSGU_ERR_BAD_ARG;  // nblocks comes with the plan
    } else {
        P.nblocks = (P.n_rows + rpb - 1) / rpb;
    }
    if (P.nblocks > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    if (P.nblocks == 0) return TSGU_OK;
    const size_t lds = (size_t)P.ucap * (PERM ? 8 : 4) + (size_t)P.ecap * 4;
    if (!PERM && P.n_src >= (1ll << 30)) return TSGU_ERR_TOO_LARGE;  // ownership bits live in bits 30 / 31 of ucol
    if (lds > 64 * 1024) return TSGU_ERR_TOO_LARGE;
    const dim3 grid((unsigned)P.nblocks), block(kBlock);
    // 32-bit byte offsets into the gathered operand: rows < 2^24, row pitch < 2^24 bytes, whole operand < 4 GiB
    const bool small = P.n_src < (1ll << 24) && P.lds_ * 4 < (1ll << 24) && P.n_src * P.lds_ * 4 < (1ll << 32);
#define TSGU_RP_GO(CLV)                                                                                           \
    if (small) hipLaunchKernelGGL((csr_rowpack_kernel<I, CLV, MODE, PERM, true>), grid, block, lds, stream, P);   \
    else hipLaunchKernelGGL((csr_rowpack_kernel<I, CLV, MODE, PERM, false>), grid, block, lds, stream, P);
    switch (cl) {
        case 4: TSGU_RP_GO(4) break;
        case 8: TSGU_RP_GO(8) break;
        case 16: TSGU_RP_GO(16) break;
    }
#undef TSGU_RP_GO
    return check_launch();
}

}  // namespace tsgu
