// Row-block tile kernels ("tile"): general CSR patterns whose neighbouring rows share columns, WITHOUT a lattice — mesh orderings,
// banded factors, FEM matrices (the matrices the reference itself benchmarks: benchmarks/results/sparse_mm_suite_results.csv).
//
// Why: the gather kernels (spmm_impl.h, rowpack_impl.h) pull every referenced dense row through the vector L1 — 18-27 requests of
// 128 B per sparse row — and the row-pair walk spends ~25 VALU + 4 LDS instructions per union entry (EXPERIMENTS.md §9): on a
// brick-numbered mesh they reach 29 % of the roofline where the plane sweep reaches 60 %.  This family gives such patterns the
// sweep's STRUCTURE: a persistent workgroup walks a run of row blocks (R = 64 consecutive rows); the DISTINCT dense rows a block
// references (its "tile": 216 rows for a 4^3 brick of a 27-point mesh, listed once per pattern in the plan) arrive in LDS by 16-byte
// LDS-DMA one block AHEAD of the walk (double buffer), next to the block's slice of the value array and one byte per entry that
// names the entry's dense row inside the tile.  The walk is then: one byte + one value + one 16-byte LDS read + the FMAs per entry,
// accumulators in registers, no per-entry records, no ownership tests.
//
// Plan (built once per pattern by _tile.py, layout in include/tsgu_hip.h):
//   desc[b] = {u0, U, e0, E}   block b: its tile is ucol[u0 .. u0+U) (ascending distinct columns, padded to a multiple of 8 with
//                              repeats of the last one), its entries are e0 .. e0+E of the walked pattern
//   lidx[k]                    position of entry k's column inside its block's tile (one byte: U <= 256)
//   rptr                       int32 row pointer of the walked pattern
//   perm, slot (optional)      the transposed pattern walks A's own values (Aᵀ·G): per block, perm = the positions in the value array in
//                              ASCENDING order, slot = the entry of the block each belongs to.  Their values travel through registers
//                              (ordinary loads in source order, neighbouring lanes on neighbouring values; written to the LDS value
//                              buffer between the step's `s_waitcnt vmcnt(0)` and its barrier) — in walk order every lane of a 4-byte
//                              gather touched its own cache line and the staging, not the walk, bounded the kernel
//
// Modes: kTileSpmm  C = A·B (perm: Aᵀ·G on the transposed pattern); kTileSddmm  out[k] = alpha·<R[row k], Cm[col k]> in stored order.
// (Round 5 also built BOTH gradients in one walk of the transposed pattern's plan — G staged once, B[j] in registers, dots back through
// LDS to the threads that hold the value positions: bit-identical gradA, 1052 MB instead of 1263 MB of traffic, but 300 us against
// 115 + 173 us: the gradients of A leave as 26 M scattered 4-byte stores and the store path, not HBM, is what the walk then waits for.
// Removed; EXPERIMENTS.md §10.)
// kTileSpmm sums run in ascending entry order of the walked pattern: the same order — and the same bits — as the plan-free kernels.
//
// Synchronisation: the bulk streams (tile rows, values, entry bytes, row pointer slice) are LDS-DMA issued from inline asm — invisible
// to hipcc's s_waitcnt bookkeeping, which would otherwise drain them at the first LDS read — and a step ends with `s_waitcnt
// vmcnt(0)` + `s_barrier`.  The few words a thread needs to ISSUE the next DMAs (its tile rows' column numbers, its value
// positions, the SDDMM's own rows) are ordinary loads issued right after the DMAs of a step and first used at the top of the next
// one, i.e. behind that wait: hipcc's own wait for them never drains a DMA.
#pragma once

#include "march_impl.h"      // lat_dma16 / lat_dma4 / lat_step_sync / lat_lds_addr

namespace tsgu {

enum TileMode { kTileSpmm = 0, kTileSddmm = 1 };

constexpr int kTileRows = 64;       // rows per block
constexpr int kTileUMax = 224;      // distinct dense rows per block (multiple of 8)
constexpr int kTileEMax = 2048;     // entries per block
constexpr int kTileThreads = 512;   // eight waves: with 8 lanes per row every wave owns 8 of the block's 64 rows
constexpr int kTileEP = kTileEMax / kTileThreads;          // value dwords per thread

struct TileDesc {
    int u0, U, e0, E;
};

struct TileParams {
    int64_t n_rows, n_cols, nnz, n_blocks;
    const TileDesc* desc;        // [n_blocks + 4] (four trailing empty blocks: the pipeline reads ahead)
    const int* ucol;
    const unsigned char* lidx;   // [nnz + 16]
    const int* rptr;             // [n_rows + 1]
    const int* perm;             // optional [nnz]: per block, positions in the value array in ascending order …
    const unsigned short* slot;  // … and the entry of the block each of them belongs to
    const void* val;             // SpMM: values
    const void* S;               // gathered dense operand (B; Cm for the SDDMM)
    int64_t lds_;
    const void* Own;             // SDDMM: row operand R
    int64_t ldown;
    void* out;                   // C [n_rows][p]
    int64_t ldo;
    void* gvals;                 // SDDMM output [nnz]
    float alpha;
    int blocks_per_wg;
    int accumulate;              // SDDMM: add to gvals (the later column tiles of a wide operand)
    int cyclic;                  // block of (workgroup w, step k): 0: w·blocks_per_wg + k (a run per workgroup); 1: k·workgroups + w
};

// LDS layout (bytes): two buffers of {tile | values | entry bytes | row pointer slice} + one zero row behind them
template <int RB>
struct TileLds {
    static constexpr int kTile = kTileUMax * RB;
    static constexpr int kVals = kTileEMax * 4;
    static constexpr int kLidx = kTileEMax + 16;
    static constexpr int kRs = (kTileRows + 8) * 4;
    static constexpr int kBuf = kTile + kVals + kLidx + kRs;
    static constexpr int oVals = kTile, oLidx = kTile + kVals, oRs = kTile + kVals + kLidx;
    static constexpr int oZero = 2 * kBuf;
    static constexpr int kTotal = 2 * kBuf + RB;
};

// wave-uniform copy of a descriptor (scalar registers: its fields go into lane predicates and M0-relative addresses)
__device__ __forceinline__ TileDesc tile_uniform(const TileDesc d) {
    return TileDesc{__builtin_amdgcn_readfirstlane(d.u0), __builtin_amdgcn_readfirstlane(d.U), __builtin_amdgcn_readfirstlane(d.e0),
                    __builtin_amdgcn_readfirstlane(d.E)};
}

// smallest and largest value over the wave of a per-row length (all lanes of a row's group hold the same value): one v_readlane per
// row and scalar min / max — three dependent ds_bpermute round trips (what __shfl_xor compiles to) at the head of every walk cost
// more than the whole scalar chain
template <int CL>
__device__ __forceinline__ void tile_wave_minmax(int x, int& lo, int& hi) {
    lo = hi = __builtin_amdgcn_readlane(x, 0);
#pragma unroll
    for (int g = 1; g < kWave / CL; ++g) {
        const int y = __builtin_amdgcn_readlane(x, g * CL);
        lo = y < lo ? y : lo;
        hi = y > hi ? y : hi;
    }
}

template <typename V, int CL, int MODE, bool PERM>
__global__ __launch_bounds__(kTileThreads, 4) void tile_kernel(const TileParams P) {
    static_assert(std::is_same<V, float>::value && CL == 8, "fp32 operands of 32 columns (so far)");
    constexpr int RB = CL * 16;                    // bytes of a dense row
    constexpr int RPW = kWave / CL;                // rows per wave
    constexpr int NW = kTileThreads / kWave;
    static_assert(RPW * NW == kTileRows, "one pass: every wave owns RPW rows of the block");
    constexpr int UP = (kTileUMax * CL + kTileThreads - 1) / kTileThreads;      // 16-byte tile pieces per thread
    using L = TileLds<RB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_lds[];
    const unsigned lds0 = lat_lds_addr(tile_lds);

    const int t = threadIdx.x, lane = t & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(t / kWave);      // (a scalar: it goes into M0 for the DMA destinations)
    const int sub = lane % CL, grp = lane / CL;
    // Which blocks a workgroup walks.  Neighbouring blocks share dense rows; an XCD's L2 (4 MB) only catches that when they are
    // worked on at about the same time BY THE SAME XCD.  `cyclic`: at step k the whole chip works on blocks k·G … k·G + G - 1 and —
    // virtual workgroup ids are XCD-contiguous (xcd_chunked_block) — every XCD on a run of G / 8 consecutive blocks.  Otherwise a
    // workgroup owns a run of consecutive blocks (its own successive tiles overlap, the tiles of its XCD's other workgroups do not).
    const int64_t vb = xcd_chunked_block(blockIdx.x, gridDim.x);
    const int64_t b_first = P.cyclic ? vb : vb * P.blocks_per_wg;
    const int64_t b_step = P.cyclic ? (int64_t)gridDim.x : 1;
    int nloc;
    if (P.cyclic) nloc = b_first < P.n_blocks ? (int)((P.n_blocks - b_first + b_step - 1) / b_step) : 0;
    else nloc = (int)((P.n_blocks - b_first) < P.blocks_per_wg ? (P.n_blocks - b_first) : P.blocks_per_wg);
    if (nloc <= 0) return;
    // descriptor of local step k (steps beyond the last block read the trailing empty descriptors)
    const TileDesc* __restrict__ desc_all = P.desc;
    auto desc_at = [&](int k) -> TileDesc {
        int64_t b = b_first + (int64_t)k * b_step;
        if (b > P.n_blocks) b = P.n_blocks;
        return desc_all[b];
    };
    const float* __restrict__ S = static_cast<const float*>(P.S);
    const uint32_t ld_bytes = (uint32_t)P.lds_ * 4u;
    const unsigned wave_piece = (unsigned)(wave * kWave);
    if (t < RB / 4) reinterpret_cast<float*>(tile_lds + L::oZero)[t] = 0.f;       // the zero row (visible after the first step sync)

    // Every compiler-visible global load of the loop below is FIRST USED behind the `s_waitcnt vmcnt(0)` that ends the step it was
    // issued in (hipcc's own wait for it is then free); a load consumed inside the same step would make hipcc drain the DMAs.
    int ucolr[UP];                                       // column numbers of this thread's tile pieces of the block staged next
    unsigned toff[UP];                                   // … and their byte offsets in the gathered operand
    int permr[PERM ? kTileEP : 1];                       // value positions this thread fetches for the block staged next …
    unsigned slotr[PERM ? kTileEP : 1];                  // … the entries of that block they belong to …
    unsigned slotw[PERM ? kTileEP : 1];                  // … (of the block whose values are in flight)
    float valr[PERM ? kTileEP : 1];                      // values in flight: loaded while block k is walked, written to LDS at the end of the step
    uint4 own_cur = {0, 0, 0, 0}, own_nxt = {0, 0, 0, 0};      // SDDMM: this lane's 16 bytes of its row of R, block walked / next block
#pragma unroll
    for (int i = 0; i < UP; ++i) ucolr[i] = 0;

    // Staging predicates are WAVE-UNIFORM (scalar branches, no exec masks): a wave instruction of tile pieces covers 8 whole rows
    // and U is a multiple of 8; value / byte instructions run when their first lane has work, lanes beyond the end repeat the last
    // element (into slots nobody reads).
    const int wave_row = wave * RPW;                     // first tile row of this wave's piece 0
    const int wave_e = wave * kWave;                     // first entry / dword of this wave's instruction 0
    auto load_words = [&](const TileDesc d) {            // for the block staged one step later: byte offsets of its tile rows, value positions
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            if (wave_row + i * (kTileThreads / CL) < d.U) {
                ucolr[i] = P.ucol[d.u0 + (t / CL) + i * (kTileThreads / CL)];      // (multiplied by the row bytes when it is pinned: NOT here)
            }
        }
        if constexpr (PERM) {
#pragma unroll
            for (int i = 0; i < kTileEP; ++i) {
                if (wave_e + i * kTileThreads < d.E) {
                    int e = t + i * kTileThreads;
                    e = e < d.E ? e : d.E - 1;
#ifndef TSGU_TILE_NT_WORDS
#define TSGU_TILE_NT_WORDS 0
#endif
#if TSGU_TILE_NT_WORDS
                    permr[i] = __builtin_nontemporal_load(P.perm + ((int64_t)d.e0 + e));      // (a single-use stream)
#else
                    permr[i] = P.perm[(int64_t)d.e0 + e];
#endif
                    slotr[i] = P.slot[(int64_t)d.e0 + e];
                }
            }
        }
    };
    auto pin_words = [&]() {
        // pinned in straight-line code before the first DMA of a step: hipcc otherwise places its wait for ucolr[i] inside the
        // predicated block of piece i — behind the DMA of piece i - 1
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            lat_pin(ucolr[i]);
            toff[i] = __umul24((unsigned)ucolr[i], ld_bytes) + (unsigned)sub * 16u;      // (launcher: columns and row bytes below 2^24)
        }
        if constexpr (PERM) {
#pragma unroll
            for (int i = 0; i < kTileEP; ++i) {
                lat_pin(permr[i]);
                lat_pin(slotr[i]);
            }
        }
    };

    auto stage = [&](int k, const TileDesc d) {          // issue the DMAs of block k into buffer k & 1
        const unsigned buf = lds0 + (unsigned)(k & 1) * L::kBuf;
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            if (wave_row + i * (kTileThreads / CL) < d.U)
                lat_dma16<false>(S, toff[i], buf + (wave_piece + (unsigned)i * kTileThreads) * 16u);
        }
        if constexpr (MODE == kTileSpmm) {
#pragma unroll
            for (int i = 0; i < kTileEP; ++i) {
                if (wave_e + i * kTileThreads < d.E) {
                    if constexpr (PERM) {
                        // (an ordinary load, issued behind the step's DMAs and first used behind its `s_waitcnt vmcnt(0)`: put_values)
                        valr[i] = static_cast<const float*>(P.val)[(uint32_t)permr[i]];
                        slotw[i] = slotr[i];
                    } else {
                        int e = t + i * kTileThreads;
                        e = e < d.E ? e : d.E - 1;
                        lat_dma4<true>(P.val, (uint32_t)(d.e0 + e) * 4u, buf + L::oVals + (wave_piece + (unsigned)i * kTileThreads) * 4u);
                    }
                }
            }
        }
        {   // entry bytes: dwords from the 4-byte aligned address below e0 (the walk adds e0 & 3)
            const int a0 = d.e0 & ~3, nd = (d.e0 + d.E - a0 + 3) >> 2;
#pragma unroll
            for (int i = 0; i < (kTileEMax / 4 + 1 + kTileThreads - 1) / kTileThreads; ++i) {
                if (wave_e + i * kTileThreads < nd) {       // (lane predicate: a repeated dword would land beyond the byte region)
                    const int q = t + i * kTileThreads;
                    if (q < nd) lat_dma4<true>(P.lidx, (uint32_t)(a0 + q * 4), buf + L::oLidx + (wave_piece + (unsigned)i * kTileThreads) * 4u);
                }
            }
        }
        {   // row pointer slice: rptr[r0 .. r0 + R] (waves 0 and 1)
            const int64_t r0 = (b_first + (int64_t)k * b_step) * kTileRows;
            const int nr = (int)((P.n_rows - r0) < kTileRows ? (P.n_rows - r0) : kTileRows) + 1;
            if (wave_e < nr) {                              // (waves 0 and 1; lane predicate: the slice is 65 words, the region 72)
                if (t < nr) lat_dma4<true>(P.rptr, (uint32_t)(r0 + t) * 4u, buf + L::oRs + wave_piece * 4u);
            }
        }
    };

    auto put_values = [&](int k, const TileDesc d) {     // the fetched values of block k into its value buffer (behind the step's vmcnt(0))
        if constexpr (PERM) {
            float* const vb = reinterpret_cast<float*>(tile_lds + (k & 1) * L::kBuf + L::oVals);
#pragma unroll
            for (int i = 0; i < kTileEP; ++i) {
                if (wave_e + i * kTileThreads < d.E) vb[slotw[i]] = valr[i];      // (lanes beyond the end repeat the last pair)
            }
        }
    };

    auto load_own = [&](int k) {                         // SDDMM: this lane's part of its row of R in block k
        if constexpr (MODE == kTileSddmm) {
            const float* Own = static_cast<const float*>(P.Own);
            const int64_t r = (b_first + (int64_t)k * b_step) * kTileRows + wave * RPW + grp;
            own_nxt = r < P.n_rows ? *reinterpret_cast<const uint4*>(Own + r * P.ldown + sub * 4) : uint4{0, 0, 0, 0};
        }
    };

    auto walk = [&](int k, const TileDesc d) {
        const unsigned char* buf = tile_lds + (k & 1) * L::kBuf;
        const float* vals = reinterpret_cast<const float*>(buf + L::oVals);
        // entry bytes: entry k of the block sits at byte (e0 & 3) + k of the byte region.  They are read as ALIGNED dwords and shifted
        // into place (v_alignbyte): a byte-misaligned 8- or 16-byte LDS read is slow (the SDDMM with one misaligned 8-byte read per round
        // ran 141 us against 120 us with eight byte reads)
        const unsigned* lidw = reinterpret_cast<const unsigned*>(buf + L::oLidx);
        auto bytes8 = [&](int at, unsigned& lo, unsigned& hi) {      // the 8 entry bytes of the round that starts at entry `at`
            const int b = (d.e0 & 3) + at;
            const unsigned* w = lidw + (b >> 2);
            const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
            lo = __builtin_amdgcn_alignbyte(w1, w0, (unsigned)b & 3u);
            hi = __builtin_amdgcn_alignbyte(w2, w1, (unsigned)b & 3u);
        };
        const int* rs = reinterpret_cast<const int*>(buf + L::oRs);
        const unsigned char* trow = buf + sub * 16;
        const unsigned char* zrow = tile_lds + L::oZero + sub * 16;
        const int rl = wave * RPW + grp;
        const int64_t r = (b_first + (int64_t)k * b_step) * kTileRows + rl;
        const bool live = r < P.n_rows;
        const int s = live ? rs[rl] - d.e0 : 0, e = live ? rs[rl + 1] - d.e0 : 0;
        // eight entries per round; the rounds every row of the wave has in full run without predicates (a scalar trip count)
        int len_lo, len_hi;
        tile_wave_minmax<CL>(e - s, len_lo, len_hi);
        const int nfull = len_lo >> 3;
        int kk = s;
        if constexpr (MODE == kTileSpmm) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            unsigned li[8];
            float v[8];
            auto fetch = [&](int at) {                   // entry bytes and values of the round that starts at `at` (reads beyond a row's
                unsigned lo, hi;                         // end stay inside the LDS buffer: harmless)
                bytes8(at, lo, hi);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    li[j] = ((j < 4 ? lo : hi) >> (8 * (j & 3))) & 0xffu;
                    v[j] = vals[at + j];
                }
            };
            fetch(kk);
            for (int it = 0; it < nfull; ++it) {
                float4 bj[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bj[j] = *reinterpret_cast<const float4*>(trow + li[j] * RB);
                float vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) vv[j] = v[j];
                kk += 8;
                fetch(kk);                               // the next round's bytes and values travel while this round is summed
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[0] = fmaf(vv[j], bj[j].x, acc[0]);
                    acc[1] = fmaf(vv[j], bj[j].y, acc[1]);
                    acc[2] = fmaf(vv[j], bj[j].z, acc[2]);
                    acc[3] = fmaf(vv[j], bj[j].w, acc[3]);
                }
            }
            // what is left of the rows.  Rows of ONE length (the common case): fewer than eight entries, a scalar count, no predicates …
            if (len_lo == len_hi) {
                const int rem = len_lo & 7;
                float4 bj[7];
#pragma unroll
                for (int j = 0; j < 7; ++j)
                    if (j < rem) bj[j] = *reinterpret_cast<const float4*>(trow + li[j] * RB);
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    if (j < rem) {
                        acc[0] = fmaf(v[j], bj[j].x, acc[0]);
                        acc[1] = fmaf(v[j], bj[j].y, acc[1]);
                        acc[2] = fmaf(v[j], bj[j].z, acc[2]);
                        acc[3] = fmaf(v[j], bj[j].w, acc[3]);
                    }
                }
                kk = e;
            }
            // … otherwise whole rounds in which a missing entry reads the zero row with a zero value: a row never touches a dense row
            // it does not reference, 0·0 adds nothing
            while (__any(kk < e)) {
                float4 bj[8];
                float vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = kk + j < e;
                    bj[j] = *reinterpret_cast<const float4*>(ok ? trow + li[j] * RB : zrow);
                    vv[j] = ok ? v[j] : 0.f;
                }
                kk += 8;
                if (__any(kk < e)) fetch(kk);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[0] = fmaf(vv[j], bj[j].x, acc[0]);
                    acc[1] = fmaf(vv[j], bj[j].y, acc[1]);
                    acc[2] = fmaf(vv[j], bj[j].z, acc[2]);
                    acc[3] = fmaf(vv[j], bj[j].w, acc[3]);
                }
            }
            if (live) store_vec<float, 4, true>(static_cast<float*>(P.out) + r * P.ldo + sub * 4, acc);
        } else {
            const float g0 = __uint_as_float(own_cur.x), g1 = __uint_as_float(own_cur.y), g2 = __uint_as_float(own_cur.z),
                        g3 = __uint_as_float(own_cur.w);
            float* gv = static_cast<float*>(P.gvals) + d.e0;
            // Eight entries per round, slot j of lane `sub` holds entry kk + (j ^ sub): the 8 lanes of a row then read 8 different
            // dense rows at once (each its own 16-byte column chunk, conflict-free) and the cross-lane sums form a TRANSPOSED tree with
            // fixed slots — after the step over lane bit m a lane keeps the entries whose bit m equals its own:
            //   partner sub ^ 7 (half mirror):  h4[j] = part[j] + partner's part[7 - j]      (both are entry j ^ sub)
            //   partner sub ^ 2:                h2[j] = h4[j]   + partner's h4[j + 2]
            //   partner sub ^ 1:                h     = h2[0]   + partner's h2[1]            = the dot of entry kk + sub
            // 7 DPP additions per 8 entries, no selects.
            unsigned sel[8];                             // v_perm selectors: byte (j ^ sub) of the round's 8 entry bytes into bits 0..7, zeros above
#pragma unroll
            for (int j = 0; j < 8; ++j) sel[j] = 0x0c0c0c00u | (unsigned)(j ^ sub);
            auto round = [&]() -> float {
                unsigned lo8, hi8;
                bytes8(kk, lo8, hi8);
                unsigned li[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) li[j] = __builtin_amdgcn_perm(hi8, lo8, sel[j]);
                float4 bj[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bj[j] = *reinterpret_cast<const float4*>(trow + li[j] * RB);
#ifndef TSGU_TILE_SDDMM_GROUPED
#define TSGU_TILE_SDDMM_GROUPED 1
#endif
#if TSGU_TILE_SDDMM_GROUPED
                asm volatile("" ::: "memory");           // (all eight LDS requests leave before the first dot waits for one)
#endif
                float part[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) part[j] = fmaf(g3, bj[j].w, fmaf(g2, bj[j].z, fmaf(g1, bj[j].y, g0 * bj[j].x)));
                float h4[4], h2[2];
#pragma unroll
                for (int j = 0; j < 4; ++j) h4[j] = part[j] + dpp_move<0x141>(part[7 - j]);
#pragma unroll
                for (int j = 0; j < 2; ++j) h2[j] = h4[j] + dpp_move<0x4E>(h4[j + 2]);
                return h2[0] + dpp_move<0xB1>(h2[1]);
            };
            for (int it = 0; it < nfull; ++it) {
                const float h = P.alpha * round();
                gv[(unsigned)(kk + sub)] = P.accumulate ? gv[(unsigned)(kk + sub)] + h : h;
                kk += 8;
            }
            // the rest under a store predicate only: the slots of entries beyond a row's end hold whatever dense row the byte behind the
            // row names (inside the LDS buffer); their sums stay in their own slots of the tree and are never stored
            while (__any(kk < e)) {
                const float h = round();
                if (kk + sub < e) gv[kk + sub] = P.accumulate ? gv[kk + sub] + P.alpha * h : P.alpha * h;
                kk += 8;
            }
        }
    };

    // ---- pipeline: descriptors three blocks ahead, a thread's words two, DMA one, walk ---------------------------------------
    TileDesc d0 = tile_uniform(desc_at(0)), d1 = tile_uniform(desc_at(1)), d2 = tile_uniform(desc_at(2));
    TileDesc raw = desc_at(3);
    load_words(d0);
    pin_words();
    stage(0, d0);
    load_own(0);
    load_words(d1);
    if constexpr (PERM) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        put_values(0, d0);
    }
    lat_step_sync();
    for (int k = 0; k < nloc; ++k) {
        const TileDesc d3 = tile_uniform(raw);          // (loaded during the previous step)
        pin_words();
        own_cur = own_nxt;
        if (k + 1 < nloc) stage(k + 1, d1);             // (uses the words loaded during step k - 1)
        if (k + 2 < nloc) load_words(d2);
        if (k + 1 < nloc) load_own(k + 1);
        raw = desc_at(k + 4);
        walk(k, d0);
        if constexpr (PERM) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (k + 1 < nloc) put_values(k + 1, d1);
        }
        lat_step_sync();
        d0 = d1, d1 = d2, d2 = d3;
    }
}

}  // namespace tsgu
