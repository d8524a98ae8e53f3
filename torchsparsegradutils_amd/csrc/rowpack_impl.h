// Row-pair gather kernels ("rowpack"): one lane group owns TWO consecutive sparse rows and walks the UNION of
// their column sets, so a dense row that both sparse rows reference is gathered once and used twice.
//
// Why: the gather kernels are bounded by the L1/TA path — every stored entry pulls a dense row through L1
// (C2: 27e6 x 128 B per pass) — while neighbouring rows of stencil / banded / mesh matrices share most of their
// columns (27-point stencil: rows j and j+1 share 18 of 27).  Walking the union (36 gathers per row pair instead
// of 54) removes a third of that traffic without any LDS tile, i.e. at full occupancy.
//
// Plan (built once per sparsity pattern by _pattern.build_rowpack_plan; layout in include/tsgu_hip.h).  Two forms:
//   stream form      uptr/ucol/upos/sperm hold one record per union entry / stored entry of the whole matrix;
//   dictionary form  workgroups whose records are translations of each other share one copy ("class"): the
//                    records are stored relative to the workgroup's base pair / base column / base value position,
//                    wcls[b] and wbase[b] = {pair, column, position} select and place them.  On lattice stencils the
//                    index streams (a third of the HBM traffic of the stream form) become a few L2-resident KB.
// Geometry: CL column lanes x EP entry lanes per pair (CL*VEC = p; EP > 1 only for narrow dense rows, so that a
// lane group is never smaller than 8 lanes and a 256-thread workgroup never owns more than 64 rows).  With EP == 1
// each row's sum runs over its own entries in ascending stored order: bit-identical to the plain gather kernels
// with one lane group per row.  With EP > 1 the entries of a pair are dealt round-robin to the entry lanes and the
// partial sums are combined with a fixed xor tree (deterministic, not bit-identical to a serial sum).
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
    const int* uptr;        // stream: [nblocks*GPB+1]; dictionary: [nclasses][GPB+1], relative
    const int* ucol;        // stream: [nu]; dictionary: [nclasses][ucap], relative to the workgroup's base column
    const uint32_t* upos;   // like ucol (plans with explicit slots) or null
    const int* sperm;       // stream: [nnz]; dictionary: [nclasses][ecap], relative to the base position; or null
    const int* order;       // [nblocks] or null: workgroup b processes row block order[b] (a permutation; speed only)
    const int* vpair;       // stream: [nblocks*GPB] row pair of each lane-group slot (-1 = none); dictionary:
                            // [nclasses][GPB] relative to the base pair; null = consecutive pairs
    const int* eptr;        // stream form with vpair: [nblocks+1] start of each workgroup's entries in sperm
    const int* wcls;        // dictionary form: [nblocks] class of each workgroup (null = stream form)
    const int* wbase;       // dictionary form: [nblocks][3] = {base pair, base column, base value position}
    const int* cne;         // dictionary form, permuted plans: [nclasses] stored entries per workgroup of the class
    const int* srcstart;    // dictionary form, permuted plans, optional: first value position of every source row — sperm records are
                            // then (source row - wbase[b][2]) << 8 | offset inside the row
    const void* val;
    const void* S;          // gathered dense operand (B for SpMM, G for the backward)
    int64_t lds_;
    const void* Own;        // backward: B; SDDMM: the row operand
    int64_t ldown;
    void* out;              // C / gradB
    int64_t ldo;
    void* gradA;            // backward: [nnz] in A's order; SDDMM: [nnz] output in walked order
    float alpha;            // SDDMM scale
    int ecap, ucap;         // LDS capacities: staged values / union records per workgroup
    int rgroup;             // rows per lane group: 2 (pairs) or 4 (quads; stored-order walks without slots only)
    int64_t nblocks;
};

typedef float rp_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* rp_lds_ptr;
typedef const __attribute__((address_space(1))) void* rp_glb_ptr;

#ifndef TSGU_RP_MINGROUP
#define TSGU_RP_MINGROUP 8    // smallest lane group (column lanes x entry lanes) that owns a row pair
#endif
constexpr int kRpMaxQ = 8 * (8 / TSGU_RP_MINGROUP);    // staged values per workgroup <= kRpMaxQ * 256
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
#ifndef TSGU_RP_U
#define TSGU_RP_U 4
#endif
// PERM : the values are addressed through the (workgroup-sorted) permutation `sperm`
// SLOTS: union records carry explicit value slots (`upos`); otherwise ownership bits + running counters
// R    : rows per lane group — 2 (pairs: every mode) or 4 (quads: stored-order walks without slots; the union of four
//        consecutive stencil rows has 54 columns instead of 2 x 36 = fewer gathers through L1 per output row)
template <typename V, typename I, int CL, int EP, int MODE, bool PERM, bool SLOTS, bool SMALL, int R = 2>
__global__ __launch_bounds__(kBlock) TSGU_RP_OCC void csr_rowpack_kernel(const RpParams P) {
    using T = VT<V>;
    constexpr int VEC = T::kWide;
    constexpr int GROUP = CL * EP;
    constexpr int GPB = kBlock / GROUP;  // lane groups per workgroup
    constexpr int RPB = R * GPB;         // rows per workgroup
    constexpr int MAXQ = kRpMaxQ * R / 2;  // staged values per workgroup <= MAXQ * 256
    // gathers in flight per lane; the bf16 backward keeps 8 floats per gathered row: 3 fit 5 waves per SIMD better (C5: -11 %)
    constexpr int U = (MODE == kRpBwd && VEC == 8) ? 3 : TSGU_RP_U;
    constexpr bool kDma = std::is_same<V, float>::value;  // 4-byte values go to LDS by DMA
    static_assert(SLOTS || !PERM, "permuted plans carry explicit slots");
    static_assert(EP == 1 || SLOTS, "several entry lanes per pair need explicit slots");
    static_assert(MODE != kRpBwd || PERM, "the backward always walks the transposed pattern");
    static_assert(MODE != kRpSddmm || (!PERM && !SLOTS), "SDDMM walks the pattern in stored order");
    static_assert(R == 2 || (R == 4 && !SLOTS), "quads walk ownership-bit records");
    constexpr int kOwnShift = 32 - R;                      // ownership bits live in the top R bits of the column word
    constexpr uint32_t kColMask = (1u << kOwnShift) - 1u;

    extern __shared__ uint4 rp_smem[];
    int* s_ucol = reinterpret_cast<int*>(rp_smem);
    uint32_t* s_upos = reinterpret_cast<uint32_t*>(s_ucol + P.ucap);           // SLOTS only
    float* s_val = reinterpret_cast<float*>(s_ucol + (SLOTS ? 2 : 1) * (size_t)P.ucap);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);  // provably wave-uniform: no waterfall loop around the LDS-DMA
    const int grp = tid / GROUP;
    const int gl = tid % GROUP;
    const int cl = gl % CL;
    const int ep = gl / CL;

    int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    if (P.order) vb = P.order[vb];
    const I* __restrict__ ptr = static_cast<const I*>(P.ptr);
    const V* __restrict__ val = static_cast<const V*>(P.val);
    const int64_t ngroups = (P.n_rows + R - 1) / R;
    const bool dict = P.wcls != nullptr;

    // ---- where this workgroup's records live -------------------------------------------------------------
    const int* up;         // GPB+1 union offsets of the lane-group slots
    int64_t urec;          // first union record of the workgroup inside ucol / upos
    const int* sp = nullptr;  // the workgroup's slice of sperm
    int64_t pair;          // index of the row group (pair / quad) this lane group owns
    int64_t colbase = 0;   // added to every (relative) union column
    int permbase = 0;      // added to every (relative) value position
    int64_t e0 = 0;        // first entry of the workgroup (plans whose values are read in stored order)
    int ne = 0;
    if (dict) {
        const int cls = P.wcls[vb];
        const int* wb = P.wbase + 3 * vb;
        const int64_t bpair = wb[0];
        colbase = wb[1];
        permbase = wb[2];
        up = P.uptr + (int64_t)cls * (GPB + 1);
        urec = (int64_t)cls * P.ucap;
        const int v = P.vpair ? P.vpair[(int64_t)cls * GPB + grp] : grp;
        pair = v >= 0 ? bpair + v : -1;
        if constexpr (PERM) {
            ne = P.cne[cls];
            sp = P.sperm + (int64_t)cls * P.ecap;
        }
    } else {
        const int64_t slot0 = vb * GPB;
        up = P.uptr + slot0;
        urec = up[0];
        pair = P.vpair ? (int64_t)P.vpair[slot0 + grp] : slot0 + grp;
        if (PERM && P.vpair) {
            e0 = (int64_t)P.eptr[vb];
            ne = (int)((int64_t)P.eptr[vb + 1] - e0);
            sp = P.sperm + e0;
        }
    }
    if (!(PERM && (dict || P.vpair))) {
        // consecutive rows: the workgroup's entries are one contiguous range of the walked pattern
        const int64_t row0 = vb * RPB;
        const int64_t row1 = row0 + RPB < P.n_rows ? row0 + RPB : P.n_rows;
        e0 = (int64_t)ptr[row0];
        ne = (int)((int64_t)ptr[row1] - e0);
        if constexpr (PERM) {
            if (!dict) sp = P.sperm + e0;
        }
    }
    const bool pair_ok = pair >= 0 && pair < ngroups;
    const int64_t row_first = R * pair;
    bool row_ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) row_ok[r] = pair_ok && row_first + r < P.n_rows;
    const int u0 = up[0];
    const int nu = up[GPB] - u0;
    const int lo = pair_ok ? up[grp] - u0 : 0;
    const int hi = pair_ok ? up[grp + 1] - u0 : 0;

    // ---- phase A: union records and values of the workgroup's rows -> LDS ----
    int qv[PERM ? MAXQ : 1];
    if constexpr (PERM) {
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int t = q * kBlock + tid;
            qv[q] = 0;
            if (q * kBlock < ne) {
                // dictionary tables are shared by many workgroups: keep them cacheable; streams are single-use
                if (t < ne) {
                    if (dict && P.srcstart) {
                        const int w = sp[t];
                        qv[q] = P.srcstart[permbase + (w >> 8)] + (w & 0xff);
                    } else {
                        qv[q] = (dict ? sp[t] : stream_load(sp + t)) + permbase;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kRpMaxU; ++q) {
        const int t = q * kBlock + tid;
        if (q * kBlock < nu) {
            if (t < nu) {
                if (dict) {
                    __builtin_amdgcn_global_load_lds((rp_glb_ptr)(P.ucol + urec + t), (rp_lds_ptr)(s_ucol + q * kBlock + wave * kWave), 4, 0, 0);
                    if constexpr (SLOTS)
                        __builtin_amdgcn_global_load_lds((rp_glb_ptr)(P.upos + urec + t), (rp_lds_ptr)(s_upos + q * kBlock + wave * kWave), 4, 0, 0);
                } else {
                    __builtin_amdgcn_global_load_lds((rp_glb_ptr)(P.ucol + urec + t), (rp_lds_ptr)(s_ucol + q * kBlock + wave * kWave), 4, 0, 2);
                    if constexpr (SLOTS)
                        __builtin_amdgcn_global_load_lds((rp_glb_ptr)(P.upos + urec + t), (rp_lds_ptr)(s_upos + q * kBlock + wave * kWave), 4, 0, 2);
                }
            }
        }
    }
    if constexpr (MODE != kRpSddmm) {  // SDDMM reads no values
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int t = q * kBlock + tid;
            if (q * kBlock < ne) {
                if (t < ne) {
                    const V* vsrc = PERM ? val + qv[PERM ? q : 0] : val + e0 + t;
                    if constexpr (kDma) {
                        __builtin_amdgcn_global_load_lds((rp_glb_ptr)vsrc, (rp_lds_ptr)(s_val + q * kBlock + wave * kWave), 4, 0, PERM ? 0 : 2);
                    } else {
                        s_val[t] = T::up(PERM ? *vsrc : stream_load(vsrc));  // narrow values: widened on the way in
                    }
                }
            }
        }
    }
    // The accumulators are explicit PAIRS (elements 2h, 2h+1 = the two halves of a loaded 8-byte word): left to itself
    // the vectoriser pairs elements (1,2) and (0,3) of a row in the slotted kernels and pays three register moves per
    // update to feed its packed FMAs (13 of the 58 VALU instructions per 4 union entries of the transposed walk).
    float own[MODE != kRpSpmm ? R : 1][VEC];
    rp_f2 acc2[R][VEC / 2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int h = 0; h < VEC / 2; ++h) acc2[r][h] = rp_f2{0.f, 0.f};
    }
    auto axpy = [](float a, const float (&g)[VEC], rp_f2 (&a2)[VEC / 2]) {
        const rp_f2 av = {a, a};
#pragma unroll
        for (int h = 0; h < VEC / 2; ++h) a2[h] = __builtin_elementwise_fma(av, rp_f2{g[2 * h], g[2 * h + 1]}, a2[h]);
    };
    if constexpr (MODE != kRpSpmm) {
        const V* __restrict__ Own = static_cast<const V*>(P.Own);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) own[r][v] = 0.f;
            if (row_ok[r]) load_vec<V, VEC>(Own + (row_first + r) * P.ldown + cl * VEC, own[r]);
        }
    }
    __syncthreads();

    // ---- phase B: walk the union of the group's columns; one gather serves every row that owns the column ----
    // dense row c starts at byte c·ld·sizeof(V) of a wave-uniform base: when the operand is smaller than 4 GiB and the
    // factors fit 24 bits (launcher checks) the offset is ONE full-rate v_mad_u32_u24 and the load uses the SGPR-base
    // form, instead of a quarter-rate 64-bit multiply-add + 64-bit shift-add per gather (the phase is VALU-issue-bound)
    const V* __restrict__ Sv = static_cast<const V*>(P.S) + colbase * P.lds_;
    const char* __restrict__ Sbase = reinterpret_cast<const char*>(Sv);
    const uint32_t ldbb = (uint32_t)P.lds_ * (uint32_t)sizeof(V), cl16 = (uint32_t)cl * 16u;
    auto gather = [&](int c, float (&g)[VEC]) {
        if constexpr (SMALL) {
            const uint32_t boff = __umul24((uint32_t)c, ldbb) + cl16;
            load_vec<V, VEC>(reinterpret_cast<const V*>(Sbase + boff), g);
        } else {
            load_vec<V, VEC>(Sv + cl * VEC + row_off(c, (uint32_t)P.lds_), g);
        }
    };

    if constexpr (SLOTS) {
        auto use = [&](const float (&g)[VEC], uint32_t half, rp_f2 (&a_)[VEC / 2], const float (&own_)[VEC]) {
            if (!(half & kRpAbsent)) {  // uniform inside the CL lanes of an entry lane, divergent across the wave: exec-masked
                const float a = s_val[half];
                axpy(a, g, a_);
                if constexpr (MODE == kRpBwd) {
                    float d = own_[0] * g[0];
#pragma unroll
                    for (int v = 1; v < VEC; ++v) d = fma(own_[v], g[v], d);
                    d = group_sum<float, CL>(d);
                    if (cl == 0) s_val[half] = d;
                }
            }
        };
        constexpr int OB = MODE != kRpSpmm ? 1 : 0;  // own[] has one row only for SpMM (unused)
        int i = lo + ep;
        for (; i + (U - 1) * EP < hi; i += U * EP) {
            int c[U];
            uint32_t w[U];
            float g[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = s_ucol[i + u * EP];
                w[u] = s_upos[i + u * EP];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) gather(c[u], g[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                use(g[u], w[u] & 0xffffu, acc2[0], own[0]);
                use(g[u], w[u] >> 16, acc2[1], own[OB]);
            }
        }
        for (; i < hi; i += EP) {
            const int c = s_ucol[i];
            const uint32_t w = s_upos[i];
            float g[VEC];
            gather(c, g);
            use(g, w & 0xffffu, acc2[0], own[0]);
            use(g, w >> 16, acc2[1], own[OB]);
        }
        if constexpr (EP > 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int h = 0; h < VEC / 2; ++h) {
                    acc2[r][h].x = ep_sum<float, CL, EP>(acc2[r][h].x);
                    acc2[r][h].y = ep_sum<float, CL, EP>(acc2[r][h].y);
                }
            }
        }
    } else {
        // values in walked order: the slots of a row are consecutive, so the record only says WHICH rows own the column
        // (top R bits of ucol) and R running counters replace the slot words (no upos stream, half the record LDS)
        int i = lo;
        int k[R];
#pragma unroll
        for (int r = 0; r < R; ++r) k[r] = row_ok[r] ? (int)((int64_t)ptr[row_first + r] - e0) : 0;
        auto use_seq = [&](const float (&g)[VEC], uint32_t w) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if ((w >> (kOwnShift + r)) & 1u) {
                    if constexpr (MODE == kRpSddmm) {
                        // gradient of the stored entry: <row operand, gathered column operand>, into the entry's slot
                        float d = own[r][0] * g[0];
#pragma unroll
                        for (int v = 1; v < VEC; ++v) d = fma(own[r][v], g[v], d);
                        d = group_sum<float, CL>(d);
                        if (cl == 0) s_val[k[r]] = d;
                    } else {
                        const float a = s_val[k[r]];
                        axpy(a, g, acc2[r]);
                    }
                    ++k[r];
                }
            }
        };
        for (; i + U <= hi; i += U) {
            uint32_t w[U];
            float g[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = (uint32_t)s_ucol[i + u];
#pragma unroll
            for (int u = 0; u < U; ++u) gather((int)(w[u] & kColMask), g[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) use_seq(g[u], w[u]);
        }
        for (; i < hi; ++i) {
            const uint32_t w = (uint32_t)s_ucol[i];
            float g[VEC];
            gather((int)(w & kColMask), g);
            use_seq(g, w);
        }
    }

    if constexpr (MODE != kRpSddmm) {
        V* __restrict__ out = static_cast<V*>(P.out);
        if (ep == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (row_ok[r]) {
                    float res[VEC];
#pragma unroll
                    for (int h = 0; h < VEC / 2; ++h) {
                        res[2 * h] = acc2[r][h].x;
                        res[2 * h + 1] = acc2[r][h].y;
                    }
                    store_vec<V, VEC, true>(out + (row_first + r) * P.ldo + cl * VEC, res);
                }
            }
        }
    } else {
        // the block's gradients sit in stored order in LDS: one coalesced, streaming write
        __syncthreads();
        V* __restrict__ gout = static_cast<V*>(P.gradA);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int t = q * kBlock + tid;
            if (q * kBlock < ne) {
                if (t < ne) {
                    if constexpr (kDma) __builtin_nontemporal_store(P.alpha * s_val[t], gout + e0 + t);
                    else gout[e0 + t] = T::down(P.alpha * s_val[t]);
                }
            }
        }
    }

    if constexpr (MODE == kRpBwd) {
        // gradA leaves in the sorted-permutation order: neighbouring lanes write neighbouring words
        __syncthreads();
        V* __restrict__ gout = static_cast<V*>(P.gradA);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int t = q * kBlock + tid;
            if (q * kBlock < ne) {
                if (t < ne) gout[qv[q]] = T::down(s_val[t]);
            }
        }
    }
}

// lane geometry for (value type, p): CL column lanes of 16 bytes, EP entry lanes so that a pair has >= 8 lanes
template <typename V>
inline bool rp_geom(int64_t p, int& cl, int& ep) {
    constexpr int vec = VT<V>::kWide;
    if (p <= 0 || p % vec != 0) return false;
    const int64_t ncl = p / vec;
    if (ncl != 2 && ncl != 4 && ncl != 8 && ncl != 16) return false;
    cl = (int)ncl;
    ep = cl >= TSGU_RP_MINGROUP ? 1 : TSGU_RP_MINGROUP / cl;
    return true;
}

template <typename V, typename I, int MODE, bool PERM>
int rp_launch(RpParams P, hipStream_t stream) {
    constexpr int vec = VT<V>::kWide;
    int cl = 0, ep = 0;
    if (!rp_geom<V>(P.p, cl, ep)) return TSGU_ERR_BAD_ARG;
    if (P.lds_ % vec != 0 || !aligned16(P.S)) return TSGU_ERR_BAD_ARG;
    if (MODE != kRpSddmm && (P.ldo % vec != 0 || !aligned16(P.out))) return TSGU_ERR_BAD_ARG;
    if (MODE != kRpSpmm && (P.ldown % vec != 0 || !aligned16(P.Own))) return TSGU_ERR_BAD_ARG;
    const int rg = P.rgroup == 0 ? 2 : P.rgroup;
    const bool slots = PERM || P.upos != nullptr;
    if (rg != 2 && (rg != 4 || slots || ep != 1)) return TSGU_ERR_BAD_ARG;   // quads: ownership-bit records, one entry lane
    if (P.ecap <= 0 || P.ucap <= 0 || P.ecap > kRpMaxQ * (rg / 2) * kBlock || P.ucap > kRpMaxU * kBlock || P.ecap >= kRpAbsent ||
        P.ucap % 4 != 0 || P.lds_ > 0xffffffffLL)
        return TSGU_ERR_BAD_ARG;
    if (!slots && ep != 1) return TSGU_ERR_BAD_ARG;      // several entry lanes per pair need explicit slots
    if (MODE == kRpSddmm && slots) return TSGU_ERR_BAD_ARG;
    const int64_t gpb = kBlock / (cl * ep);
    if (P.wcls) {
        if (!P.wbase || (PERM && !P.cne) || P.eptr) return TSGU_ERR_BAD_ARG;
    } else if (P.vpair) {
        if (!PERM || !P.eptr) return TSGU_ERR_BAD_ARG;
    }
    // without a permutation (or without vpair / classes) a workgroup owns rg·gpb consecutive rows
    if ((!PERM || !(P.vpair || P.wcls)) && P.nblocks != (P.n_rows + rg * gpb - 1) / (rg * gpb)) return TSGU_ERR_BAD_ARG;
    if (P.nblocks > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    if (P.nblocks <= 0) return P.n_rows == 0 ? TSGU_OK : TSGU_ERR_BAD_ARG;
    const size_t lds = (size_t)P.ucap * (slots ? 8 : 4) + (size_t)P.ecap * 4;
    if (!slots && P.n_src >= (1ll << (32 - rg))) return TSGU_ERR_TOO_LARGE;  // ownership bits live in the top bits of ucol
    if (lds > 64 * 1024) return TSGU_ERR_TOO_LARGE;
    const dim3 grid((unsigned)P.nblocks), block(kBlock);
    // 32-bit byte offsets into the gathered operand: rows < 2^24, row pitch < 2^24 bytes, whole operand < 4 GiB
    const int64_t pitch = P.lds_ * (int64_t)sizeof(V);
    const bool small = P.n_src < (1ll << 24) && pitch < (1ll << 24) && P.n_src * pitch < (1ll << 32);
#define TSGU_RP_GO(CLV, EPV, SL, RG)                                                                                          \
    do {                                                                                                                     \
        if (small) hipLaunchKernelGGL((csr_rowpack_kernel<V, I, CLV, EPV, MODE, PERM, SL, true, RG>), grid, block, lds, stream, P);  \
        else hipLaunchKernelGGL((csr_rowpack_kernel<V, I, CLV, EPV, MODE, PERM, SL, false, RG>), grid, block, lds, stream, P);       \
    } while (0)
    constexpr int MG = TSGU_RP_MINGROUP;
#define TSGU_RP_EPOF(CLV) ((CLV) >= MG ? 1 : MG / (CLV))
    if constexpr (PERM) {
        switch (cl) {
            case 2: TSGU_RP_GO(2, TSGU_RP_EPOF(2), true, 2); break;
            case 4: TSGU_RP_GO(4, TSGU_RP_EPOF(4), true, 2); break;
            case 8: TSGU_RP_GO(8, 1, true, 2); break;
            case 16: TSGU_RP_GO(16, 1, true, 2); break;
        }
    } else {
        if (slots) {
            if constexpr (MODE == kRpSddmm) return TSGU_ERR_BAD_ARG;
            else {
                switch (cl) {
                    case 2: TSGU_RP_GO(2, TSGU_RP_EPOF(2), true, 2); break;
                    case 4: TSGU_RP_GO(4, TSGU_RP_EPOF(4), true, 2); break;
                    case 8: TSGU_RP_GO(8, 1, true, 2); break;
                    case 16: TSGU_RP_GO(16, 1, true, 2); break;
                }
            }
        } else if (rg == 4) {
            // row quads were parity-green but never faster than pairs where it mattered: not compiled any more
            return TSGU_ERR_BAD_ARG;
        } else {
            switch (cl) {
                case 8: TSGU_RP_GO(8, 1, false, 2); break;
                case 16: TSGU_RP_GO(16, 1, false, 2); break;
                default: return TSGU_ERR_BAD_ARG;
            }
        }
    }
#undef TSGU_RP_EPOF
#undef TSGU_RP_GO
    return check_launch();
}

}  // namespace tsgu
