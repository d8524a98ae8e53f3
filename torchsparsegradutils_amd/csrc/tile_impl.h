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
//   cpos, cslot (optional)     the transposed pattern walks A's own values (Aᵀ·G): a block's values are runs of consecutive positions in
//                              the value array (one run per source row); they are fetched as 16-BYTE CHUNKS in ascending source order —
//                              chunk q of the plan: cpos[q] = position of its first value, cslot[q][0..3] = the entry of the block each
//                              of its four values belongs to (0xffff: not this block's).  5-6 bytes per entry of round 5's perm + slot
//                              become ~3.6.  The values travel through registers (one 16-byte load per lane, neighbouring lanes on
//                              neighbouring chunks; written to the LDS value buffer between the step's `s_waitcnt vmcnt(0)` and its
//                              barrier) — in walk order every lane of a 4-byte gather touched its own cache line
//
// Operands wider than one column tile (32 fp32 columns): ONE launch.  A block's entry bytes, values and row pointer slice are staged
// once; the pipeline's steps are (block, column tile) pairs and only the dense tile is staged per step; the SDDMM keeps the dots of
// the earlier column tiles in the (otherwise unused) value region of the block's buffer.
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
constexpr int kTileEMax = 1792;     // entries per block (28 per row on average; round 5: 2048 — the 16-bit entry records below take their LDS)
constexpr int kTileThreads = 512;   // eight waves: with 8 lanes per row every wave owns 8 of the block's 64 rows
constexpr int kTileCP = 2;                                 // value chunks (16 bytes) per thread: at most 1024 chunks per block
constexpr int kTileCMax = kTileCP * kTileThreads;

constexpr int kTileXMax = kTileEMax / 8 + kTileRows * 7 / 8;      // entry records (8 entries = 16 bytes) per block: every row padded to whole
                                                                    // records (sum of ceil(len / 8) <= E / 8 + 7/8 per row)

struct TileDesc {
    int u0, U, e0, E;
    int c0, NC;                  // value chunks of the block (plans with cpos / cslot): cpos[c0 .. c0 + NC)
    int x0, X;                   // entry records of the block: ent[x0 .. x0 + X) (16 bytes each)
};

struct TileParams {
    int64_t n_rows, n_cols, nnz, n_blocks;
    const TileDesc* desc;        // [n_blocks + 4] (four trailing empty blocks: the pipeline reads ahead)
    const int* ucol;
    const unsigned char* lidx;   // [nnz + 16] (the SDDMM's walk)
    const uint4* ent;            // entry records (forward / Aᵀ·G walk): per row whole ROUNDS of eight 16-bit tile offsets (tile row · row bytes),
                                 // padded with the offset of the zero row
    const unsigned short* xrow;  // [n_rows rounded up to whole blocks] first record of every row inside its block
    const int* rptr;             // [n_rows + 1]
    const int* cpos;             // optional: per value chunk, the position of its first value in the value array …
    const uint2* cslot;          // … and the entries of the block its four values belong to (4 x uint16, 0xffff: none)
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
    int ncol;                    // column tiles of 32 (p / 32)
    int cyclic;                  // block of (workgroup w, step k): 0: w·blocks_per_wg + k (a run per workgroup); 1: k·workgroups + w
};

// LDS layout (bytes): two tile buffers (filled per pipeline step = per (block, column tile); each followed by a ZERO ROW, the tile row
// the padding of the entry records names) and two plan buffers {values | entry records or entry bytes | row pointer slice | first record
// of every row} (filled per block)
template <int RB>
struct TileLds {
    static constexpr int kTile = (kTileUMax + 1) * RB;
    static constexpr int oZeroRow = kTileUMax * RB;           // inside a tile buffer (no DMA ever writes it: U <= kTileUMax)
    static constexpr int kVals = kTileEMax * 4 + 16;
    static constexpr int kEnt = kTileXMax * 16;               // (>= the kTileEMax + 16 entry bytes of the SDDMM)
    static constexpr int kRs = (kTileRows + 8) * 4;
    static constexpr int kXr = kTileRows * 2;
    static constexpr int kPlan = kVals + kEnt + kRs + kXr;
    static constexpr int kTileStride = kTile, kPlanStride = kPlan;
    static constexpr int oPlan = 2 * kTile;
    static constexpr int oVals = 0, oLidx = kVals, oEnt = kVals, oRs = kVals + kEnt, oXr = kVals + kEnt + kRs;      // inside a plan buffer
    static constexpr int kTotal = 2 * kTile + 2 * kPlan;
    static_assert(kTotal <= 80 * 1024, "two workgroups per CU");
};

// wave-uniform copy of a descriptor (scalar registers: its fields go into lane predicates and M0-relative addresses)
template <bool CHUNKS, bool RECORDS>
__device__ __forceinline__ TileDesc tile_uniform(const TileDesc d) {
    TileDesc u{__builtin_amdgcn_readfirstlane(d.u0), __builtin_amdgcn_readfirstlane(d.U), __builtin_amdgcn_readfirstlane(d.e0),
               __builtin_amdgcn_readfirstlane(d.E), 0, 0, 0, 0};
    if constexpr (CHUNKS) {
        u.c0 = __builtin_amdgcn_readfirstlane(d.c0);
        u.NC = __builtin_amdgcn_readfirstlane(d.NC);
    }
    if constexpr (RECORDS) {
        u.x0 = __builtin_amdgcn_readfirstlane(d.x0);
        u.X = __builtin_amdgcn_readfirstlane(d.X);
    }
    return u;
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

// WIDE = false: one column tile (ncol == 1, known at compile time: every step is a crossing — the loop of round 5, and its speed:
// with a run-time ncol the 32-column SDDMM ran 122 us against 110)
template <typename V, int CL, int MODE, bool PERM, bool WIDE>
__global__ __launch_bounds__(kTileThreads, 4) void tile_kernel(const TileParams P) {
    static_assert(std::is_same<V, float>::value && CL == 8, "fp32 operands, column tiles of 32 (so far)");
    constexpr int RB = CL * 16;                    // bytes of a dense row of one column tile
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
    const int ncol = WIDE ? P.ncol : 1;            // column tiles per block (p / 32)
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
    // descriptor of local block k (blocks beyond the last read the trailing empty descriptors)
    const TileDesc* __restrict__ desc_all = P.desc;
    auto desc_at = [&](int k) -> TileDesc {
        int64_t b = b_first + (int64_t)k * b_step;
        if (b > P.n_blocks) b = P.n_blocks;
        return desc_all[b];
    };
    auto uni = [](const TileDesc d) { return tile_uniform<PERM, MODE == kTileSpmm>(d); };
    const float* __restrict__ S = static_cast<const float*>(P.S);
    const uint32_t ld_bytes = (uint32_t)P.lds_ * 4u;
    const unsigned wave_piece = (unsigned)(wave * kWave);
    if (t < RB / 2) reinterpret_cast<float*>(tile_lds + (t / (RB / 4)) * L::kTileStride + L::oZeroRow)[t % (RB / 4)] = 0.f;      // the zero rows (visible after the first step sync)

    // Every compiler-visible global load of the loop below is FIRST USED behind the `s_waitcnt vmcnt(0)` that ends the step it was
    // issued in (hipcc's own wait for it is then free); a load consumed inside the same step would make hipcc drain the DMAs.
    int ucolr[UP];                                       // column numbers of this thread's tile pieces of the block staged next
    unsigned toff[UP];                                   // byte offsets in the gathered operand of the tile pieces of the block being staged
    int cposr[PERM ? kTileCP : 1];                       // value chunks this thread fetches for the block staged next: first position …
    uint2 cslotr[PERM ? kTileCP : 1];                    // … and the entries of that block their four values belong to
    uint2 cslotw[PERM ? kTileCP : 1];                    // … (of the block whose values are in flight)
    u32x4_t valr[PERM ? kTileCP : 1];                    // values in flight: loaded while a block is walked, written to LDS at the end of the step
    uint4 own_cur = {0, 0, 0, 0}, own_nxt = {0, 0, 0, 0};      // SDDMM: this lane's 16 bytes of its row of R, step walked / next step
#pragma unroll
    for (int i = 0; i < UP; ++i) ucolr[i] = 0;

    // Staging predicates are WAVE-UNIFORM (scalar branches, no exec masks): a wave instruction of tile pieces covers 8 whole rows
    // and U is a multiple of 8; value / byte instructions run when their first lane has work, lanes beyond the end repeat the last
    // element (into slots nobody reads).
    const int wave_row = wave * RPW;                     // first tile row of this wave's piece 0
    const int wave_e = wave * kWave;                     // first entry / dword / chunk of this wave's instruction 0
    auto load_words = [&](const TileDesc d) {            // for the block staged one crossing later: its tile rows' columns, its value chunks
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            if (wave_row + i * (kTileThreads / CL) < d.U) {
                ucolr[i] = P.ucol[d.u0 + (t / CL) + i * (kTileThreads / CL)];      // (multiplied by the row bytes when it is pinned: NOT here)
            }
        }
        if constexpr (PERM) {
#pragma unroll
            for (int i = 0; i < kTileCP; ++i) {
                if (wave_e + i * kTileThreads < d.NC) {
                    int q = t + i * kTileThreads;
                    q = q < d.NC ? q : d.NC - 1;
                    cposr[i] = P.cpos[(int64_t)d.c0 + q];
                    cslotr[i] = P.cslot[(int64_t)d.c0 + q];
                }
            }
        }
    };
    auto pin_loaded = [&](TileDesc& rawd) {
        // Everything a step loaded for later steps is pinned at the TOP of the next step, in straight-line code before its first DMA —
        // on crossing and non-crossing steps alike: hipcc's wait for a load sits where it first sees the value used, and a use it
        // meets on only one path of a branch, or behind a DMA, becomes `s_waitcnt vmcnt(0)` in the middle of the step (draining the
        // DMAs: the walk then starts with its tile still on the way).  After the pins its scoreboard is clean.
#pragma unroll
        for (int i = 0; i < UP; ++i) lat_pin(ucolr[i]);
        if constexpr (PERM) {
#pragma unroll
            for (int i = 0; i < kTileCP; ++i) {
                lat_pin(cposr[i]);
                lat_pin(cslotr[i].x);
                lat_pin(cslotr[i].y);
            }
        }
        lat_pin(rawd.u0), lat_pin(rawd.U), lat_pin(rawd.e0), lat_pin(rawd.E);
        if constexpr (PERM) lat_pin(rawd.c0), lat_pin(rawd.NC);
        if constexpr (MODE == kTileSpmm) lat_pin(rawd.x0), lat_pin(rawd.X);
        if constexpr (MODE == kTileSddmm) lat_pin(own_cur.x), lat_pin(own_cur.y), lat_pin(own_cur.z), lat_pin(own_cur.w);
    };
    auto set_offsets = [&]() {                           // `toff` of the block whose columns `ucolr` holds
#pragma unroll
        for (int i = 0; i < UP; ++i) toff[i] = __umul24((unsigned)ucolr[i], ld_bytes) + (unsigned)sub * 16u;      // (launcher: columns and row bytes below 2^24)
    };

    auto stage_tile = [&](int tb, int c, const TileDesc d) {      // the DMAs of column tile c of the block `toff` belongs to, into tile buffer tb
        const unsigned buf = lds0 + (unsigned)tb * L::kTileStride;
        const float* const Sc = S + c * (RB / 4);
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            if (wave_row + i * (kTileThreads / CL) < d.U)
                lat_dma16<false>(Sc, toff[i], buf + (wave_piece + (unsigned)i * kTileThreads) * 16u);
        }
    };
    auto stage_plan = [&](int pb, int k, const TileDesc d) {      // values, entry bytes and row pointer slice of local block k into plan buffer pb
        const unsigned buf = lds0 + L::oPlan + (unsigned)pb * L::kPlanStride;
        if constexpr (MODE == kTileSpmm && !PERM) {
            // values in stored order: 16-byte pieces (four values per lane; the source is only 4-byte aligned, which the LDS-DMA takes at
            // full speed).  A piece that would read beyond the end of the value array (the last block only) is copied value by value.
            const int nq = (d.E + 3) >> 2;
            if (wave_e < nq) {
                const int q = t < nq ? t : nq - 1;
                const uint32_t g = (uint32_t)d.e0 + 4u * (uint32_t)q;
                if (__builtin_expect((int64_t)g + 4 <= P.nnz, 1)) {
                    lat_dma16<true>(P.val, g * 4u, buf + L::oVals + wave_piece * 16u);
                } else {
                    float* dst = reinterpret_cast<float*>(tile_lds + L::oPlan + pb * L::kPlanStride + L::oVals) + 4 * q;
                    for (int e = 0; e < 4; ++e) dst[e] = (int64_t)g + e < P.nnz ? static_cast<const float*>(P.val)[g + e] : 0.f;
                }
            }
        } else if constexpr (MODE == kTileSpmm) {
#pragma unroll
            for (int i = 0; i < kTileCP; ++i) {
                if (wave_e + i * kTileThreads < d.NC) {
                    // (an ordinary 16-byte load, issued behind the step's DMAs and first used behind its `s_waitcnt vmcnt(0)`: put_values;
                    // the plan keeps every chunk inside the value array)
                    struct __attribute__((packed, aligned(4))) Chunk {
                        u32x4_t v;
                    };
                    valr[i] = reinterpret_cast<const Chunk*>(static_cast<const float*>(P.val) + (uint32_t)cposr[i])->v;
                    cslotw[i] = cslotr[i];
                }
            }
        }
        if constexpr (MODE == kTileSpmm) {
            // entry records: 16-byte pieces (the block's records start on a 16-byte boundary), and the first record of every row
            if (wave_e < d.X) {
                const int q = t < d.X ? t : d.X - 1;
                lat_dma16<true>(P.ent, (uint32_t)(d.x0 + q) * 16u, buf + L::oEnt + wave_piece * 16u);
            }
            if (wave == 2) {                                // (waves 0 and 1 issue the row pointer slice below)
                const int64_t r0 = (b_first + (int64_t)k * b_step) * kTileRows;
                if (lane < kTileRows / 2) lat_dma4<true>(P.xrow, (uint32_t)(r0 * 2 + lane * 4), buf + L::oXr);
            }
        } else {   // entry bytes: dwords from the 4-byte aligned address below e0 (the walk adds e0 & 3)
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

    auto put_values = [&](int pb, const TileDesc d) {    // the fetched values of a block into its value buffer (behind the step's vmcnt(0))
        if constexpr (PERM) {
            unsigned* const vbuf = reinterpret_cast<unsigned*>(tile_lds + L::oPlan + pb * L::kPlanStride + L::oVals);
#pragma unroll
            for (int i = 0; i < kTileCP; ++i) {
                if (wave_e + i * kTileThreads < d.NC) {     // (lanes beyond the end repeat the last chunk: the same values into the same slots)
                    const unsigned s0 = cslotw[i].x & 0xffffu, s1 = cslotw[i].x >> 16, s2 = cslotw[i].y & 0xffffu, s3 = cslotw[i].y >> 16;
                    if (s0 != 0xffffu) vbuf[s0] = valr[i][0];
                    if (s1 != 0xffffu) vbuf[s1] = valr[i][1];
                    if (s2 != 0xffffu) vbuf[s2] = valr[i][2];
                    if (s3 != 0xffffu) vbuf[s3] = valr[i][3];
                }
            }
        }
    };

    auto load_own = [&](int k, int c) {                  // SDDMM: this lane's part of its row of R in column tile c of local block k
        if constexpr (MODE == kTileSddmm) {
            const float* Own = static_cast<const float*>(P.Own) + c * (RB / 4);
            const int64_t r = (b_first + (int64_t)k * b_step) * kTileRows + wave * RPW + grp;
            own_nxt = r < P.n_rows ? *reinterpret_cast<const uint4*>(Own + r * P.ldown + sub * 4) : uint4{0, 0, 0, 0};
        }
    };

    auto walk = [&](int tb, int pb, int k, int c, const TileDesc d) {
        const unsigned char* tbuf = tile_lds + tb * L::kTileStride;
        unsigned char* pbuf = tile_lds + L::oPlan + pb * L::kPlanStride;
        const float* vals = reinterpret_cast<const float*>(pbuf + L::oVals);
        // entry bytes: entry k of the block sits at byte (e0 & 3) + k of the byte region.  They are read as ALIGNED dwords and shifted
        // into place (v_alignbyte): a byte-misaligned 8- or 16-byte LDS read is slow (the SDDMM with one misaligned 8-byte read per round
        // ran 141 us against 120 us with eight byte reads)
        const unsigned* lidw = reinterpret_cast<const unsigned*>(pbuf + L::oLidx);
        auto bytes8 = [&](int at, unsigned& lo, unsigned& hi) {      // the 8 entry bytes of the round that starts at entry `at`
            const int b = (d.e0 & 3) + at;
            const unsigned* w = lidw + (b >> 2);
            const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
            lo = __builtin_amdgcn_alignbyte(w1, w0, (unsigned)b & 3u);
            hi = __builtin_amdgcn_alignbyte(w2, w1, (unsigned)b & 3u);
        };
        const int* rs = reinterpret_cast<const int*>(pbuf + L::oRs);
        const unsigned char* trow = tbuf + sub * 16;
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
            // Entry records: per row whole rounds of eight 16-bit tile offsets (tile row · row bytes, ready to be added to the lane's
            // address: `v_add_u32_sdwa` takes the half-word — one instruction per entry where a byte stream cost an extraction, a
            // shift-add and a share of two `v_alignbyte`), 16-byte aligned: ONE `ds_read_b128` per round.  A row's last record is padded
            // with the offset of the zero row.
            const uint4* recs = reinterpret_cast<const uint4*>(pbuf + L::oEnt) + (live ? reinterpret_cast<const unsigned short*>(pbuf + L::oXr)[rl] : 0);
            const unsigned zoff = (unsigned)L::oZeroRow;
            const int len = e - s;
            const int nr = (len + 7) >> 3;                 // this row's records
            const int nr_hi = (len_hi + 7) >> 3;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            // A round: its values (requested FIRST: the LDS returns in order, so they are there when the rows are), its eight tile
            // rows, then the NEXT round's record, then the sums.  (One set of value registers: with the values fetched a round ahead
            // hipcc kept two, and the chunk variant spilled.)
            auto values_of = [&](int t, float (&v)[8]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = vals[s + 8 * t + j];      // (reads beyond a row's end stay inside the LDS buffer: masked below)
            };
            auto rows_of = [&](const uint4 rec, float4 (&bj)[8], int n) {
                const unsigned ww[4] = {rec.x, rec.y, rec.z, rec.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < n) {
                        const unsigned off = (j & 1) ? ww[j >> 1] >> 16 : ww[j >> 1] & 0xffffu;
                        bj[j] = *reinterpret_cast<const float4*>(trow + off);
                    }
                }
            };
            auto add = [&](const float (&v)[8], const float4 (&bj)[8], int n) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < n) {
                        acc[0] = fmaf(v[j], bj[j].x, acc[0]);
                        acc[1] = fmaf(v[j], bj[j].y, acc[1]);
                        acc[2] = fmaf(v[j], bj[j].z, acc[2]);
                        acc[3] = fmaf(v[j], bj[j].w, acc[3]);
                    }
                }
            };
            uint4 w = recs[0];
            int tr = 0;
#pragma nounroll
            for (; tr < nfull; ++tr) {                   // the rounds every row of the wave has in full: no predicates
                float v[8];
                float4 bj[8];
                values_of(tr, v);
                rows_of(w, bj, 8);
                w = recs[tr + 1];
                add(v, bj, 8);
            }
            if (len_lo == len_hi) {
                // rows of ONE length (the common case): fewer than eight entries are left, a scalar count, no predicates
                const int rem = len_lo & 7;
                if (rem) {
                    float v[8];
                    float4 bj[8];
                    values_of(tr, v);
                    rows_of(w, bj, rem);
                    add(v, bj, rem);
                }
            } else {
                // … otherwise whole rounds: a row that has ended reads the zero row with zero values (its record is replaced: what lies
                // behind its last record belongs to the next row), the padding of a last record names the zero row itself and only its
                // values are cleared.  A row never touches a dense row it does not reference, 0·0 adds nothing
#pragma nounroll
                for (; tr < nr_hi; ++tr) {
                    const bool in = tr < nr;
                    const unsigned z2 = zoff | (zoff << 16);
                    const uint4 rec = {in ? w.x : z2, in ? w.y : z2, in ? w.z : z2, in ? w.w : z2};
                    float v[8];
                    float4 bj[8];
                    values_of(tr, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = 8 * tr + j < len ? v[j] : 0.f;
                    rows_of(rec, bj, 8);
                    w = recs[tr + 1];
                    add(v, bj, 8);
                }
            }
            if (live) store_vec<float, 4, true>(static_cast<float*>(P.out) + r * P.ldo + c * (RB / 4) + sub * 4, acc);
        } else {
            const float g0 = __uint_as_float(own_cur.x), g1 = __uint_as_float(own_cur.y), g2 = __uint_as_float(own_cur.z),
                        g3 = __uint_as_float(own_cur.w);
            float* gv = static_cast<float*>(P.gvals) + d.e0;
            // wide operands: the dots of the column tiles are added up in the block's (otherwise unused) value region — every entry
            // belongs to one lane, the same lane in every column tile — and leave for HBM with the last tile
            float* gacc = reinterpret_cast<float*>(pbuf + L::oVals);
            const bool first_tile = c == 0, last_tile = c + 1 == ncol;
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
                asm volatile("" ::: "memory");           // (all eight LDS requests leave before the first dot waits for one)
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
            // (one branch-free loop per case: a scalar branch inside the round loop kept hipcc from overlapping the rounds — 112 -> 140 us)
            auto rounds = [&](auto emit) {
                for (int it = 0; it < nfull; ++it) {
                    emit(kk + sub, P.alpha * round());
                    kk += 8;
                }
                // the rest under a store predicate only: the slots of entries beyond a row's end hold whatever dense row the byte behind
                // the row names (inside the LDS buffer); their sums stay in their own slots of the tree and are never stored
                while (__any(kk < e)) {
                    const float h = P.alpha * round();
                    if (kk + sub < e) emit(kk + sub, h);
                    kk += 8;
                }
            };
            if (first_tile && last_tile) rounds([&](int idx, float a) { gv[(unsigned)idx] = a; });
            else if (first_tile) rounds([&](int idx, float a) { gacc[idx] = a; });
            else if (!last_tile) rounds([&](int idx, float a) { gacc[idx] = gacc[idx] + a; });
            else rounds([&](int idx, float a) { gv[(unsigned)idx] = gacc[idx] + a; });
        }
    };

    // ---- pipeline.  A STEP is one (block, column tile) pair: the dense tile of the next step arrives while this one is walked.  The
    // last step of a block ("crossing") also stages the next block's values, entry bytes and row pointer slice; descriptors run three
    // blocks ahead, a thread's words (tile columns, value chunks) two ------------------------------------------------------------
    TileDesc d0 = uni(desc_at(0)), d1 = uni(desc_at(1)), d2 = uni(desc_at(2));
    TileDesc raw = desc_at(3);
    load_words(d0);
    pin_loaded(raw);
    set_offsets();
    stage_tile(0, 0, d0);
    stage_plan(0, 0, d0);
    load_own(0, 0);
    load_words(d1);
    if constexpr (PERM) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        put_values(0, d0);
    }
    lat_step_sync();
    // (ONE flat loop over the steps — with the column tiles as an inner loop hipcc hoisted the walk's per-block address arithmetic out of
    // it and paid 25 more registers, a spill in the chunk variant)
    const int nsteps = nloc * ncol;
    int k = 0, c = 0;
#pragma nounroll
    for (int step = 0; step < nsteps; ++step) {
        const bool crossing = c + 1 == ncol;
        own_cur = own_nxt;
        pin_loaded(raw);
        TileDesc d3 = d2;
        // what the NEXT step walks: the next column tile of this block, or (crossing) the first one of the next block.  ONE own-row
        // load and ONE set of tile DMAs in common code: with a copy in each branch hipcc gave the two loads different registers and
        // resolved them with a move behind `s_waitcnt vmcnt(0)` — between the DMAs and the walk, draining them
        const bool has_next = crossing ? k + 1 < nloc : true;
        const int kn = crossing ? k + 1 : k, cn = crossing ? 0 : c + 1;
        if (crossing) {
            d3 = uni(raw);                              // (loaded during the previous crossing step)
            set_offsets();                              // (the words loaded one crossing ago: `toff` now belongs to block k + 1)
        }
        if (has_next) stage_tile((step + 1) & 1, cn, crossing ? d1 : d0);      // (`toff` belongs to block k until its crossing)
        if (crossing) {
            if (k + 1 < nloc) stage_plan((k + 1) & 1, k + 1, d1);
            if (k + 2 < nloc) load_words(d2);
        }
        // (BEHIND the DMAs: the row of R misses to HBM, the tile rows mostly hit in L2, and VMEM returns in order — issued first it held
        // the tile back: SDDMM 111 -> 122 us)
        if (has_next) load_own(kn, cn);
        if (crossing) raw = desc_at(k + 4);
        walk(step & 1, k & 1, k, c, d0);
        if constexpr (PERM) {
            if (crossing) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (k + 1 < nloc) put_values((k + 1) & 1, d1);
            }
        }
        lat_step_sync();
        if (crossing) {
            d0 = d1, d1 = d2, d2 = d3;
            c = 0;
            ++k;
        } else {
            ++c;
        }
    }
}

}  // namespace tsgu
