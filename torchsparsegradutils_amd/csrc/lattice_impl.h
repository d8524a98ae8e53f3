// Lattice plane-sweep kernels ("lattice"): stencil-like patterns on a row-major 3-D (or batched 3-D, or 2-D) lattice.
//
// Why: the gather kernels (spmm_impl.h, rowpack_impl.h) pull every referenced dense row through the vector L1 — 18-27
// requests of 128 B per sparse row — and are bounded by the number of L1 misses a CU can keep in flight.
// On a lattice the rows a tile of points references are the tile plus a halo, so here a workgroup owns a TY x TZ tile of
// the (y, z) plane and MARCHES along x: it keeps a rolling window of four halo planes of the gathered dense operand in
// LDS (three in use + ring-3 being filled), filled by 16-byte LDS-DMA (`global_load_lds_dwordx4`) issued ring-3 planes ahead,
// so every gather is a `ds_read_b128` and the dense operand crosses L2 -> CU ~1.5x instead of 18-27x.  The DMA of plane
// x+2 and of the values of plane x+1 is in flight during the whole computation of plane x: latency is hidden by the
// ring, not by occupancy (one or two workgroups per CU, up to the full 160 KiB of LDS).
//
// Pattern description (built once per sparsity pattern by _lattice.build_lattice_plan, layout in include/tsgu_hip.h):
//   row = ((item*nx + x)*ny + y)*nz + z;  every stored entry (row, col) has col = the lattice point at a displacement
//   (dx, dy, dz), |dx| <= 1, |dy| <= ry, |dz| <= rz, periodic wrap allowed (wrapped halo rows are simply loaded from the
//   wrapped address).  Rows with the same displacement sequence form a CLASS (27 for a periodic or a Dirichlet 27-point
//   stencil); `rcls[row]` names the class and a per-class record table turns entry k into an LDS offset relative to the
//   row's own position in the halo tile.  There is one copy of the table per ring phase (x & 3), so that an address is
//   ONE shift-add.  Tables are padded to `recw` (multiple of 4) entries with records that point beyond the LDS
//   allocation: such reads return zero on gfx950 (tools/probe_lds.hip, tests/test_gpu_lattice.py), and the padded value
//   slots are zeroed, so no predicate is needed in the inner loop and a row never touches a dense row it does not reference.
//
// Modes (one template):
//   kLatSpmm   C = A·B                  ring = B halo planes, values of the tile's own rows staged per plane (stored order)
//   kLatSddmm  out[k] = alpha·<R[row], Cm[col(k)]>   ring = Cm halo planes, R rows in registers, result in stored order
//   kLatSpmmT  gradB = Aᵀ·G on the transposed pattern: ring = G halo planes + a second ring with the VALUES of the halo
//              rows (entry k of source row i sits at slot k of i's value row), records carry both offsets
// Summation runs in ascending entry order of the walked pattern: the same order as the plan-free kernels.
//
// Synchronisation: the DMA is issued by inline asm (invisible to hipcc's s_waitcnt bookkeeping, which would otherwise
// drain it at the first LDS read); every plane step ends with `s_waitcnt vmcnt(0) lgkmcnt(0)` + `s_barrier`.  Ordinary
// global loads inside the loop (class bytes, row starts, the SDDMM's own rows) are issued at the END of a step for the next
// one and pinned at the top of the next step before any DMA is issued, so that hipcc's wait for them never drains the DMA.
#pragma once

#include <atomic>

#include "tsgu_common.h"

namespace tsgu {

enum LatMode { kLatSpmm = 0, kLatSddmm = 1, kLatSpmmT = 2 };

constexpr int kLatND = 3;    // ring DMA pieces per thread and plane   (halo rows x chunks <= kLatND * NT)
// value DMA pieces per thread and plane: as few as the supported geometries need (each costs registers for the whole march)
constexpr int lat_nvd(int mode, int vbytes) { return vbytes == 4 ? 2 : (vbytes == 8 || mode == 2 ? 4 : 3); }
constexpr int kLatNP = 1;    // row passes per plane                   (tile rows <= kLatNP * NT / CL)
constexpr int kLatMaxLds = 160 * 1024;
#ifndef TSGU_LAT_DOT2
#define TSGU_LAT_DOT2 1     // 0: bf16 products by widening + fp32 FMAs (one entry at a time), the round-3 form before the pairs
#endif
constexpr int kLatPadRec = 0x7ff00;  // record of a padded table entry: this many bytes beyond the row's own LDS position

struct LatParams {
    int nb, nx, ny, nz;      // items, planes per item, lines per plane, points per line
    int ty, tz, ry, rz;      // tile (lines x points) and halo radii
    int cpl;                 // 16-byte chunks of a dense row per lane (1 or 2)
    int ring;                // halo planes resident in LDS: 3 in use + (ring - 3) being filled ahead (4..8)
    int tiles_y, tiles_z;    // tiles per plane
    int nseg, seg_len;       // x segments per item, planes per segment (the last one may be shorter)
    int ncls, recw;          // row classes of the whole pattern, record width (multiple of 4, <= 32)
    int nloc;                // classes per workgroup list (the records of a workgroup's own classes are what it keeps in LDS)
    const unsigned char* wlist;  // [nblocks][nloc] class ids of each workgroup's rows, 0xff = unused
    int uniform_len;         // > 0: every row of the value-owning pattern has this many entries (`rstart` is not read)
    int slot;                // bytes of one staged value row (recw values rounded up to 16 bytes)
    const void* rec;         // [ring][ncls][recw] int32 byte offsets (SpMM / SDDMM) or pairs of them (SpMMT) — see tsgu_hip.h
    const unsigned char* lens;  // [ncls] entries per row of the class
    const unsigned char* rcls;  // [rows] class of each row of the walked pattern
    const int* rstart;       // [rows+1] first value position of each row of the VALUE-OWNING pattern (A's crow)
    const void* val;
    int64_t nnz;
    const void* S;           // gathered dense operand
    int64_t lds_;
    const void* Own;         // SDDMM: row operand
    int64_t ldown;
    void* out;               // C / gradB
    int64_t ldo;
    void* gvals;             // SDDMM output [nnz]
    float alpha;
    void* dot_partial;       // SpMM, fp32 / fp64: [workgroups][p] partial sums of <out[row,:], S[row,:]> per column (Krylov loops), or null
    const int* skip;         // when given and *skip != 0 the launch does nothing (a solver loop that has finished on the device)
    const void* dot_w;       // dot epilogue: the second operand W [rows][p] with the leading dimension of `out` (null: the gathered operand itself)
    int64_t nblocks;
    // LDS layout (bytes from the start of the dynamic region; filled by lat_layout)
    int o_vals, o_zero, o_tab, o_len, o_map, lds_bytes;
};

typedef __attribute__((address_space(3))) void* lat_lds_ptr;

__device__ __forceinline__ unsigned lat_lds_addr(const void* p) { return (unsigned)(size_t)(lat_lds_ptr)p; }

// 16-byte LDS-DMA: lane l's 16 bytes land at `lds_wave_base + 16*l` (wave-uniform base in M0); the source is a
// wave-uniform 64-bit base (SGPR pair) + a per-lane 32-bit byte offset.
template <bool NT_POLICY>
__device__ __forceinline__ void lat_dma16(const void* sbase64, uint32_t voff, unsigned lds_wave_base) {
    unsigned keep;
    if constexpr (NT_POLICY)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase64), "s"(lds_wave_base) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase64), "s"(lds_wave_base) : "memory");
}

__device__ __forceinline__ void lat_step_sync() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <typename X>
__device__ __forceinline__ void lat_pin(X& x) { asm volatile("" : "+v"(x)); }

__device__ __forceinline__ int lat_mod(int v, int n) {
    v %= n;
    return v < 0 ? v + n : v;
}

constexpr uint32_t kLatNone = 0xffffffffu;

// NCH > 0: the record width is 4·NCH, known at compile time (the entry loop is unrolled and scheduled as one block)
// CPL = 16-byte chunks of a dense row per lane: CL / CPL lanes own a row.  CPL = 2 halves the per-entry bookkeeping (records,
// values, addresses are per lane GROUP) and the cross-lane reduction of the SDDMM; a wave then covers twice the rows.
template <typename V, int CL, int CPL, int MODE, int NT, int NCH>
__global__ __launch_bounds__(NT, 4) void lattice_kernel(const LatParams P) {   // 4 waves per SIMD: at most 128 VGPRs
    using T = VT<V>;
    using A = typename T::Acc;           // accumulation type: fp32 (fp32 / bf16 operands) or fp64
    constexpr int VEC = T::kWide;
    constexpr int RB = CL * 16;          // bytes of a dense row
    constexpr int LPR = CL / CPL;        // lanes per row
    constexpr int RPP = NT / LPR;        // rows per pass
    constexpr int kVB = (int)sizeof(V);
    // transposed walk with dense rows of a multiple of 128 bytes: the value ring has the pitch of the dense ring and ONE word
    // carries both offsets (dense-row offset = multiple of 128, + 4·k' in its low 7 bits); otherwise two words per entry
    // (measured: slower — with a 128-byte value pitch the broadcast value reads of the eight rows of a wave fall on ONE bank,
    // C2 transposed product 107 -> 136 us; the 112-byte pitch of the two-word form spreads them.  Kept switched off.)
    constexpr bool kPacked = false && MODE == kLatSpmmT && RB % 128 == 0;
    constexpr bool kDot2 = TSGU_LAT_DOT2 && kVB == 2 && MODE != kLatSddmm;   // bf16 products: two entries per v_dot2_f32_bf16
    constexpr int kNVD = lat_nvd(MODE, kVB);
    constexpr int kRecB = (MODE == kLatSpmmT && !kPacked) ? 8 : 4;   // bytes of a record
    constexpr int kUnroll = NCH > 0 ? NCH : 2;
    static_assert(NT % kWave == 0 && NT % CL == 0 && RB % 16 == 0 && (CPL == 1 || CPL == 2) && CL % CPL == 0, "geometry");

    extern __shared__ uint4 lat_smem[];
    char* const sm = reinterpret_cast<char*>(lat_smem);
    const unsigned sbase = lat_lds_addr(lat_smem);
    if (P.skip != nullptr && *P.skip != 0) return;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int cd = tid % CL;             // chunk of a dense row this lane moves in the ring DMA
    const int c = tid % LPR;             // lane inside the group that owns a row
    const int g = tid / LPR;

    const int HZ = P.tz + 2 * P.rz, HY = P.ty + 2 * P.ry, HR = HY * HZ, NR = P.ty * P.tz;
    const int PB = HR * RB;
    const int VL = P.slot / 16;
    const int plane_rows = P.ny * P.nz;
    const int R = P.ring, K = P.ring - 3, VB = P.ring - 2;   // ring slots, planes in flight ahead, value buffers (SpMM)
    const bool uniform = P.uniform_len > 0;
    auto wrap = [](int v, int m) { return v >= m ? v - m : v; };

    // ---- the tile and x segment of this workgroup -------------------------------------------------------
    const int64_t vblock = xcd_chunked_block(blockIdx.x, P.nblocks);
    int64_t vb = vblock;
    const int tzi = (int)(vb % P.tiles_z);
    vb /= P.tiles_z;
    const int tyi = (int)(vb % P.tiles_y);
    vb /= P.tiles_y;
    const int seg = (int)(vb % P.nseg);
    const int item = (int)(vb / P.nseg);
    const int xs = seg * P.seg_len;
    const int L = P.seg_len < P.nx - xs ? P.seg_len : P.nx - xs;   // output planes of this segment
    const int y0 = tyi * P.ty, z0 = tzi * P.tz;
    const int item_row0 = item * P.nx * plane_rows;
    // ring plane xrel (xrel = 1 is the first output plane) is lattice plane (xs - 1 + xrel) mod nx of the item
    auto row_of_x = [&](int x) -> int { return item_row0 + x * plane_rows; };   // first row of lattice plane x

    // ---- records of this workgroup's classes -> LDS (ordinary loads: before anything asynchronous exists) ----------
    // local class li = wlist[block][li]; cmap[global class] = li | len << 8.  A pattern has up to 255 classes (125 for the
    // transposed walk of a periodic 27-point stencil), a workgroup's tile column meets a handful of them.
    {
        const unsigned char* wl = P.wlist + vblock * P.nloc;
        unsigned short* cmap = reinterpret_cast<unsigned short*>(sm + P.o_map);
        for (int li = tid; li < P.nloc; li += NT) {
            const int gc = wl[li];
            if (gc != 0xff) cmap[gc] = (unsigned short)(li | (P.lens[gc] << 8));
        }
        const int cw = P.recw * kRecB / 4;            // words per class row
        const int tab_words = P.ring * P.nloc * cw;
        const int* src = static_cast<const int*>(P.rec);
        int* dst = reinterpret_cast<int*>(sm + P.o_tab);
        for (int i = tid; i < tab_words; i += NT) {
            const int ph = i / (P.nloc * cw), r = i - ph * (P.nloc * cw);
            const int li = r / cw, w = r - li * cw;
            const int gc = wl[li];
            dst[i] = gc != 0xff ? src[(ph * P.ncls + gc) * cw + w] : 0;
        }
        if (MODE == kLatSpmmT && tid < 4) reinterpret_cast<int*>(sm + P.o_zero)[tid] = 0;
    }

    // ---- per-thread descriptors: 32-bit byte offsets inside a plane, constant for the whole march -------------------
    const uint32_t ldsb = (uint32_t)P.lds_ * kVB;
    // ring pieces: piece e = d*NT + tid is chunk (e % CL) = c of halo row e / CL
    uint32_t roff[kLatND];
    const int ring_pieces = HR * CL;
#pragma unroll
    for (int d = 0; d < kLatND; ++d) {
        const int e = d * NT + tid;
        const int hr = e / CL;
        const int hy = hr / HZ, hz = hr - hy * HZ;
        roff[d] = e < ring_pieces ? (uint32_t)(lat_mod(y0 - P.ry + hy, P.ny) * P.nz + lat_mod(z0 - P.rz + hz, P.nz)) * ldsb + (uint32_t)cd * 16u : kLatNone;
    }
    // value pieces: SpMM: piece e is chunk e % VL of tile row e / VL;  SpMMT: of halo row e / VL.
    // vrow = row of the piece inside its plane (-1: none); vuo = its byte offset from the plane's first value when every row
    // has uniform_len entries (then no row start is ever loaded)
    int vrow[kNVD];
    uint32_t vch16[kNVD], vuo[kNVD];
    const int val_pieces = MODE == kLatSpmm ? NR * VL : (MODE == kLatSpmmT ? HR * VL : 0);
    if constexpr (MODE != kLatSddmm) {
#pragma unroll
        for (int d = 0; d < kNVD; ++d) {
            const int e = d * NT + tid;
            const int r = e / VL;
            vch16[d] = (uint32_t)(e - r * VL) * 16u;
            vrow[d] = -1;
            if (e < val_pieces) {
                if constexpr (MODE == kLatSpmm) {
                    const int ly = r / P.tz, lz = r - ly * P.tz;
                    if (y0 + ly < P.ny && z0 + lz < P.nz) vrow[d] = (y0 + ly) * P.nz + z0 + lz;
                } else {
                    const int hy = r / HZ, hz = r - hy * HZ;
                    vrow[d] = lat_mod(y0 - P.ry + hy, P.ny) * P.nz + lat_mod(z0 - P.rz + hz, P.nz);
                }
            }
            vuo[d] = (uint32_t)(vrow[d] < 0 ? 0 : vrow[d]) * (uint32_t)(P.uniform_len * kVB) + vch16[d];
        }
    }
    // compute rows: pass q handles tile row q*RPP + g
    // With two chunks per lane, lane c of a row owns chunks c and c + LPR and reads them in an order that depends on the
    // row (bit 1 of its index): the four rows a 16-lane LDS access covers then fall on four different bank quarters.
    int crow[kLatNP];                    // row inside its plane (-1: none)
    uint32_t coo[kLatNP][CPL];           // byte offsets of the row's 16-byte pieces inside a plane of the output
    uint32_t cown[kLatNP][CPL];          // ... of the row operand (SDDMM)
    int cen[kLatNP][CPL], csl[kLatNP];   // LDS: own position in a halo plane (+ chunk), own value / stage row
    const uint32_t ldob = (uint32_t)P.ldo * kVB;
#pragma unroll
    for (int q = 0; q < kLatNP; ++q) {
        const int r = q * RPP + g;
        const int ly = r / P.tz, lz = r - ly * P.tz;
        const bool ok = r < NR && y0 + ly < P.ny && z0 + lz < P.nz;
        crow[q] = ok ? (y0 + ly) * P.nz + z0 + lz : -1;
        const int hrow = (ly + P.ry) * HZ + lz + P.rz;
        const int swap = CPL == 2 ? ((r >> 1) & 1) : 0;
#pragma unroll
        for (int cp = 0; cp < CPL; ++cp) {
            const int chunk = c + LPR * (cp ^ swap);
            coo[q][cp] = (uint32_t)(ok ? crow[q] : 0) * ldob + (uint32_t)chunk * 16u;
            cown[q][cp] = (uint32_t)(ok ? crow[q] : 0) * ((uint32_t)P.ldown * kVB) + (uint32_t)chunk * 16u;
            cen[q][cp] = hrow * RB + chunk * 16;
        }
        csl[q] = MODE == kLatSpmmT ? hrow * P.slot : r * P.slot;
    }

    const char* const Sb = static_cast<const char*>(P.S);
    const char* const valb = static_cast<const char*>(P.val);
    const uint32_t val_bytes = (uint32_t)(P.nnz * kVB);

    // ring plane with first row `prow` into ring slot `slot`
    // (`prow` is wave-uniform; the readfirstlane says so to the compiler, which must keep the DMA's base in scalar registers)
    auto dma_ring = [&](int prow, int slot) {
        prow = __builtin_amdgcn_readfirstlane(prow);
        slot = __builtin_amdgcn_readfirstlane(slot);
        const char* const pbase = Sb + (int64_t)prow * ldsb;      // wave-uniform
        const unsigned base = sbase + (unsigned)(slot * PB) + (unsigned)(wave * kWave * 16);
#pragma unroll
        for (int d = 0; d < kLatND; ++d) {
            if (d * NT < ring_pieces) {
                if (roff[d] != kLatNone) lat_dma16<false>(pbase, roff[d], base + (unsigned)(d * NT * 16));
            }
        }
    };
    // values of the plane with first row `prow` into value buffer `buf` (SpMM: of R - 2; SpMMT: ring slot); `st` = row
    // starts (elements) of the pieces when the rows are not all of one length
    auto dma_vals = [&](int prow, int buf, const int (&st)[kNVD]) {
        if constexpr (MODE != kLatSddmm) {
            prow = __builtin_amdgcn_readfirstlane(prow);
            buf = __builtin_amdgcn_readfirstlane(buf);
            const unsigned region = MODE == kLatSpmm ? (unsigned)(P.o_vals + buf * NR * P.slot) : (unsigned)(P.o_vals + buf * HR * P.slot);
            const unsigned base = sbase + region + (unsigned)(wave * kWave * 16);
            const uint32_t plane0 = uniform ? (uint32_t)prow * (uint32_t)(P.uniform_len * kVB) : 0u;   // bytes before the plane's values
            const char* const pbase = valb + plane0;
#pragma unroll
            for (int d = 0; d < kNVD; ++d) {
                if (d * NT < val_pieces) {
                    if (vrow[d] >= 0) {
                        // (2-byte values: rows of odd length start on 2-byte boundaries; the 16-byte LDS-DMA takes such sources at
                        // full speed — measured faster than staging from the 4-byte boundary below and shifting in the reader)
                        const uint32_t off = uniform ? vuo[d] : (uint32_t)st[d] * kVB + vch16[d];
                        if (__builtin_expect(plane0 + off + 16u <= val_bytes, 1)) {
                            lat_dma16<MODE == kLatSpmm>(pbase, off, base + (unsigned)(d * NT * 16));   // (transposed walk: the neighbouring tiles read these halo value rows again — no streaming policy)
                        } else {
                            // the last 16 bytes of the value array: element-wise, never reading beyond the array
                            V* dst = reinterpret_cast<V*>(sm + region + (d * NT + tid) * 16);
#pragma nounroll
                            for (int e = 0; e < VEC; ++e) {
                                V z;
                                __builtin_memset(&z, 0, sizeof(V));
                                dst[e] = plane0 + off + (e + 1) * kVB <= val_bytes ? *reinterpret_cast<const V*>(pbase + off + e * kVB) : z;
                            }
                        }
                    }
                }
            }
        }
    };
    auto load_vst = [&](int prow, int (&st)[kNVD]) {   // only for rows of different lengths
        if constexpr (MODE != kLatSddmm) {
            if (!uniform) {
                const int* const rs = P.rstart + prow;
#pragma unroll
                for (int d = 0; d < kNVD; ++d) {
                    st[d] = 0;
                    if (d * NT < val_pieces) {
                        if (vrow[d] >= 0) st[d] = rs[(uint32_t)vrow[d]];
                    }
                }
            }
        }
    };
    // what a compute row needs from memory besides the rings: class byte; for the SDDMM its value start and its own dense row
    struct RowRegs {
        int cls[kLatNP];
        int rst[MODE == kLatSddmm ? kLatNP : 1];
        uint4 own[MODE == kLatSddmm ? kLatNP : 1][CPL];    // the row operand's 16-byte pieces, as loaded
    };
    auto load_rows = [&](int prow, RowRegs& rr) {
        const unsigned char* const cbase = P.rcls + prow;                                              // wave-uniform bases,
        const char* const obase = static_cast<const char*>(P.Own) + (int64_t)prow * P.ldown * kVB;     // 32-bit lane offsets
#pragma unroll
        for (int q = 0; q < kLatNP; ++q) {
            rr.cls[q] = 0;
            if (q * RPP < NR) {
                if (crow[q] >= 0) {
                    rr.cls[q] = cbase[(uint32_t)crow[q]];
                    if constexpr (MODE == kLatSddmm) rr.rst[q] = uniform ? (prow + crow[q]) * P.uniform_len : P.rstart[prow + crow[q]];
                    if constexpr (MODE == kLatSddmm) {
#pragma unroll
                        for (int cp = 0; cp < CPL; ++cp) rr.own[q][cp] = *reinterpret_cast<const uint4*>(obase + cown[q][cp]);
                    }
                }
            }
        }
    };
    auto pin_rows = [&](RowRegs& rr) {
#pragma unroll
        for (int q = 0; q < kLatNP; ++q) {
            lat_pin(rr.cls[q]);
            if constexpr (MODE == kLatSddmm) {
                lat_pin(rr.rst[q]);
#pragma unroll
                for (int cp = 0; cp < CPL; ++cp) {
                    lat_pin(rr.own[q][cp].x);
                    lat_pin(rr.own[q][cp].y);
                    lat_pin(rr.own[q][cp].z);
                    lat_pin(rr.own[q][cp].w);
                }
            }
        }
    };
    auto pin_vst = [&](int (&st)[kNVD]) {
        if constexpr (MODE != kLatSddmm) {
            if (!uniform) {
#pragma unroll
                for (int d = 0; d < kNVD; ++d) lat_pin(st[d]);
            }
        }
    };

    // ---- prologue: ring planes 0 .. K+1, values of planes 1 .. K (SpMM) / 0 .. K+1 (SpMMT) --------------------------
    int vst[kNVD];      // row starts for the NEXT value DMA, loaded one step ahead (rows of different lengths only)
#pragma unroll
    for (int d = 0; d < kNVD; ++d) vst[d] = 0;
    RowRegs nxt;           // rows of the NEXT output plane, loaded one step ahead
    int x_run = lat_mod(xs - 1, P.nx);      // lattice plane of ring index 0
    for (int xr = 0; xr <= K + 1; ++xr) {
        const int prow = row_of_x(x_run);
        dma_ring(prow, xr);
        if ((MODE == kLatSpmmT) || (MODE == kLatSpmm && xr >= 1 && xr <= K)) {
            load_vst(prow, vst);
            pin_vst(vst);
            dma_vals(prow, MODE == kLatSpmm ? wrap(xr, VB) : xr, vst);
            if (!uniform) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // vst is reused by the next plane
        }
        x_run = wrap(x_run + 1, P.nx);
    }
    // x_run = lattice plane of ring index K + 2: the next plane the march fetches
    int x_out = xs;                           // ring index 1
    int x_nxt = wrap(xs + 1, P.nx);           // ring index 2
    if constexpr (MODE == kLatSpmm) load_vst(row_of_x(lat_mod(xs + K, P.nx)), vst);          // plane K + 1: its values go out in step 1
    if constexpr (MODE == kLatSpmmT) load_vst(row_of_x(x_run), vst);                         // plane K + 2
    load_rows(row_of_x(x_out), nxt);
    lat_step_sync();

    const unsigned short* const cmap_s = reinterpret_cast<const unsigned short*>(sm + P.o_map);
    const int vbuf = NR * P.slot;

    // results of the previous plane leave at the START of the next step, so that the wait for the DMA at the end of a step
    // never waits for a store acknowledgement (and, for the SDDMM, the stage rows are read after the barrier)
    A acc[MODE == kLatSddmm ? 1 : kLatNP][CPL][VEC];
    int plen[MODE == kLatSddmm ? kLatNP : 1], prst[MODE == kLatSddmm ? kLatNP : 1];
    // fused dot epilogue (SpMM, fp32): Σ over this lane's rows of out[row, c]·S[row, c] — the own row of S is the centre of the
    // halo plane in LDS
    constexpr bool kCanDot = MODE == kLatSpmm && (kVB == 4 || kVB == 8) && CPL == 1;
    A dotp[kCanDot ? CPL : 1][VEC];
#pragma unroll
    for (int cp = 0; cp < (kCanDot ? CPL : 1); ++cp) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) dotp[cp][v] = (A)0;
    }
    const bool want_dot = kCanDot && P.dot_partial != nullptr;
    auto flush = [&](int prow) {
        if constexpr (MODE != kLatSddmm) {
            char* const obase = static_cast<char*>(P.out) + (int64_t)prow * ldob;
#pragma unroll
            for (int q = 0; q < kLatNP; ++q) {
                if (q * RPP < NR) {
                    if (crow[q] >= 0) {
#pragma unroll
                        for (int cp = 0; cp < CPL; ++cp) store_vec<V, VEC, true>(reinterpret_cast<V*>(obase + coo[q][cp]), acc[q][cp]);
                    }
                }
            }
        } else {
            if constexpr (kVB == 2 && kLatNP == 1 && CPL == 1) {
                // bf16, rows of one length, a wave covering whole z-lines of the tile: the gradients of a line are ONE contiguous
                // run of the output (consecutive rows), written as 16-byte aligned pieces of eight across the row boundaries
                // (+ 2-byte stores for the unaligned ends).  Row by row a 54-byte row takes 14 four-byte stores, and at C5 the
                // store path was what the kernel waited for (117 M vector-L1 accesses per launch).
                constexpr int RPW = kWave / LPR;
                if (uniform && RPW % P.tz == 0) {
                    const int ulen = P.uniform_len, lane = tid % kWave;
                    const int sl4 = P.slot / 4;                       // floats per stage row
                    for (int li = 0; li < RPW / P.tz; ++li) {
                        const int rl = wave * RPW + li * P.tz;          // first tile row of the line
                        const int lyy = rl / P.tz;
                        if (rl >= NR || y0 + lyy >= P.ny) break;
                        const int nrows = P.nz - z0 < P.tz ? P.nz - z0 : P.tz;
                        const int64_t elem0 = ((int64_t)prow + (int64_t)(y0 + lyy) * P.nz + z0) * ulen;
                        const int nel = nrows * ulen;
                        const float* const st0 = reinterpret_cast<const float*>(sm + P.o_vals) + (int64_t)rl * sl4;
                        unsigned short* const go = reinterpret_cast<unsigned short*>(P.gvals) + elem0;
                        auto at = [&](int e) -> float {               // element e of the line (row e / ulen, entry e % ulen)
                            const int rr = e / ulen;
                            return st0[rr * sl4 + (e - rr * ulen)];
                        };
                        const int head = (int)((8 - (elem0 & 7)) & 7) < nel ? (int)((8 - (elem0 & 7)) & 7) : nel;
                        const int chunks = (nel - head) / 8;
                        const int tail0 = head + chunks * 8;
                        if (lane < head) go[lane] = T::down(at(lane)).bits;
                        if (lane < nel - tail0) go[tail0 + lane] = T::down(at(tail0 + lane)).bits;
                        for (int ch = lane; ch < chunks; ch += kWave) {
                            const int e = head + ch * 8;
                            int rr = e / ulen, kk = e - rr * ulen;
                            float f[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                f[j] = st0[rr * sl4 + kk];
                                if (++kk == ulen) kk = 0, ++rr;
                            }
                            uint32_t w[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[j]) : "v"(f[2 * j]), "v"(f[2 * j + 1]));
                            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                            const u4v o = {w[0], w[1], w[2], w[3]};
                            __builtin_nontemporal_store(o, reinterpret_cast<u4v*>(go + e));
                        }
                    }
                    return;
                }
            }
            // the row's gradients in stored order: 16-byte pieces where four fit, single elements at the end
#pragma unroll
            for (int q = 0; q < kLatNP; ++q) {
                if (q * RPP < NR) {
                    if (crow[q] >= 0) {
                        V* const go = static_cast<V*>(P.gvals) + prst[q];
                        const int len = plen[q];
                        if constexpr (kVB == 8) {
                            // fp64: 16-byte pieces of two (a row starts on an 8-byte boundary), a single element at the end
                            const double* const st = reinterpret_cast<const double*>(sm + P.o_vals + csl[q]);
#pragma nounroll
                            for (int k0 = c * 2; k0 < len; k0 += LPR * 2) {
                                if (k0 + 2 <= len) {
                                    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
                                    const d2u o = {st[k0], st[k0 + 1]};
                                    __builtin_nontemporal_store(o, reinterpret_cast<d2u*>(reinterpret_cast<double*>(go) + k0));
                                } else {
                                    reinterpret_cast<double*>(go)[k0] = st[k0];
                                }
                            }
                            continue;
                        }
                        const float* const st = reinterpret_cast<const float*>(sm + P.o_vals + csl[q]);
#pragma nounroll
                        for (int k0 = c * 4; k0 < len; k0 += LPR * 4) {
                            const float4 w = *reinterpret_cast<const float4*>(st + k0);
                            const float wv[4] = {w.x, w.y, w.z, w.w};
                            if (k0 + 4 <= len) {
                                if constexpr (kVB == 4) {
                                    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                                    const f4u o = {w.x, w.y, w.z, w.w};
                                    __builtin_nontemporal_store(o, reinterpret_cast<f4u*>(go + k0));
                                } else {
                                    // four bf16 gradients: packed conversions, 4-byte aligned pair stores (a row of odd length
                                    // starts on a 2-byte boundary: then the pairs straddle the piece and its ends go out alone;
                                    // 8-byte stores on 2-byte boundaries were measured 40 % slower than this)
                                    uint32_t p01, p23, p12;
                                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p01) : "v"(wv[0]), "v"(wv[1]));
                                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p23) : "v"(wv[2]), "v"(wv[3]));
                                    unsigned short* const g16 = reinterpret_cast<unsigned short*>(go + k0);
                                    if (((prst[q] + k0) & 1) == 0) {
                                        uint32_t* const g32 = reinterpret_cast<uint32_t*>(g16);
                                        __builtin_nontemporal_store(p01, g32);
                                        __builtin_nontemporal_store(p23, g32 + 1);
                                    } else {
                                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p12) : "v"(wv[1]), "v"(wv[2]));
                                        g16[0] = (unsigned short)p01;
                                        __builtin_nontemporal_store(p12, reinterpret_cast<uint32_t*>(g16 + 1));
                                        g16[3] = (unsigned short)(p23 >> 16);
                                    }
                                }
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (k0 + j < len) go[k0 + j] = T::down((A)wv[j]);
                            }
                        }
                    }
                }
            }
        }
    };

    // ---- the march ------------------------------------------------------------------------------------------
    int ph = 1 % R;          // ring slot of the output plane = phase of the record tables
    int vbi = 1 % VB;        // value buffer of the output plane (SpMM)
    int prow_prev = 0;
    for (int xo = 1; xo <= L; ++xo) {
        // 1. what was loaded during the previous step is complete (the step ended with vmcnt(0)): take it over
        pin_rows(nxt);
        pin_vst(vst);
        RowRegs cur = nxt;
        int cl_li[kLatNP];   // local class | length << 8 of the rows of this plane (one LDS read, waited for in the compute part)
#pragma unroll
        for (int q = 0; q < kLatNP; ++q) cl_li[q] = (q * RPP < NR) ? cmap_s[cur.cls[q]] : 0;
        if (xo > 1) flush(prow_prev);
        // 2. asynchronous fetches for later planes
        if (xo + K <= L) {   // output plane xo + K exists: its last halo plane (ring index xo + K + 1) and its values are fetched now
            const int tgt = wrap(ph + R - 2, R);                     // slot of ring index xo + K + 1 (free since the last barrier)
            const int prow_dma = row_of_x(x_run);
            dma_ring(prow_dma, tgt);
            if constexpr (MODE == kLatSpmm) dma_vals(row_of_x(x_run == 0 ? P.nx - 1 : x_run - 1), wrap(vbi + VB - 1, VB), vst);   // ring index xo + K
            if constexpr (MODE == kLatSpmmT) dma_vals(prow_dma, tgt, vst);
        }
        // 3. small ordinary loads for the next step: they complete behind the computation below
        if (xo < L) {
            load_rows(row_of_x(x_nxt), nxt);
            if constexpr (MODE == kLatSpmm) load_vst(row_of_x(x_run), vst);                       // ring index xo + K + 1
            if constexpr (MODE == kLatSpmmT) load_vst(row_of_x(wrap(x_run + 1, P.nx)), vst);      // ring index xo + K + 2
        }
        // 4. the output plane
        prow_prev = row_of_x(x_out);
        const char* const tabs = sm + P.o_tab + ph * P.nloc * P.recw * kRecB;
#pragma unroll
        for (int q = 0; q < kLatNP; ++q) {
            if (q * RPP < NR) {
                if (crow[q] >= 0) {
                    const int len = cl_li[q] >> 8;
                    const char* const tb = tabs + (cl_li[q] & 0xff) * (P.recw * kRecB);
                    const char* const cb = sm + cen[q][0];
                    const char* const cb1 = sm + cen[q][CPL - 1];   // second chunk (= cb with one chunk per lane)
                    // the dense row at byte offset `rec` from the own position: this lane's chunk(s)
                    // (kept as the 16 raw bytes until it is used: a staged bf16 row widened to eight floats would double the registers
                    // of the pipeline's buffers)
                    auto load_b = [&](int rec, uint4 (&bb)[CPL]) {
                        bb[0] = *reinterpret_cast<const uint4*>(cb + rec);
                        if constexpr (CPL == 2) bb[1] = *reinterpret_cast<const uint4*>(cb1 + rec);
                    };
                    auto widen = [](const uint4& raw, A (&f)[VEC]) {
                        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
                        if constexpr (kVB == 8) {
                            f[0] = __hiloint2double((int)w[1], (int)w[0]);
                            f[1] = __hiloint2double((int)w[3], (int)w[2]);
                        } else if constexpr (kVB == 4) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(w[i]);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                f[2 * i] = __uint_as_float(w[i] << 16);
                                f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
                            }
                        }
                    };
                    auto axpy = [&](A a, const uint4 (&bb)[CPL]) {
#pragma unroll
                        for (int cp = 0; cp < CPL; ++cp) {
                            A f[VEC];
                            widen(bb[cp], f);
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[MODE == kLatSddmm ? 0 : q][cp][v] = fma(a, f[v], acc[MODE == kLatSddmm ? 0 : q][cp][v]);
                        }
                    };
                    // bf16 products: two entries per instruction.  acc[col] += a0·b0[col] + a1·b1[col] is one v_dot2_f32_bf16 on
                    // the packed pair (a0, a1) and the pair (b0[col], b1[col]) that one v_perm_b32 gathers from the two raw rows —
                    // 2 instructions per column and entry pair, against 2 widenings + 1 packed FMA per column pair and entry
                    // (the widening was a third of the kernel's VALU work).  Entries pair up by their position in the row
                    // (2k, 2k+1): the same pairs in every launch configuration.
                    auto axpy2 = [&](uint32_t ap, const uint4 (&b0)[CPL], const uint4 (&b1)[CPL]) {
                        if constexpr (kDot2) {
                            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                            const bf2 av = __builtin_bit_cast(bf2, ap);
#pragma unroll
                            for (int cp = 0; cp < CPL; ++cp) {
                                const uint32_t x[4] = {b0[cp].x, b0[cp].y, b0[cp].z, b0[cp].w};
                                const uint32_t y[4] = {b1[cp].x, b1[cp].y, b1[cp].z, b1[cp].w};
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const uint32_t lo = __builtin_amdgcn_perm(y[i], x[i], 0x05040100u);   // (b0[2i],   b1[2i])
                                    const uint32_t hi = __builtin_amdgcn_perm(y[i], x[i], 0x07060302u);   // (b0[2i+1], b1[2i+1])
                                    acc[q][cp][2 * i] = __builtin_amdgcn_fdot2_f32_bf16(av, __builtin_bit_cast(bf2, lo), acc[q][cp][2 * i], false);
                                    acc[q][cp][2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(av, __builtin_bit_cast(bf2, hi), acc[q][cp][2 * i + 1], false);
                                }
                            }
                        }
                    };
                    // four entries: a[] holds four values, or — kDot2 — the two raw bf16 pairs as bit patterns in a[0], a[1]
                    auto axpy4 = [&](const A (&a)[4], const uint4 (&b)[4][CPL]) {
                        if constexpr (kDot2) {
                            axpy2(__float_as_uint(a[0]), b[0], b[1]);
                            axpy2(__float_as_uint(a[1]), b[2], b[3]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) axpy(a[j], b[j]);
                        }
                    };
                    const int recw = NCH > 0 ? 4 * NCH : P.recw;
                    if constexpr (MODE == kLatSpmm) {
                        char* const vs = sm + P.o_vals + vbi * vbuf + csl[q];
                        if (len < recw) {   // padded slots hold whatever follows the row in the value array: zero them
#pragma nounroll
                            for (int t = len + c; t < recw; t += LPR) __builtin_memset(vs + t * kVB, 0, kVB);
                        }
#pragma unroll
                        for (int cp = 0; cp < CPL; ++cp) {
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[q][cp][v] = (A)0;
                        }
                        auto chunk_vals = [&](int k0, A (&a)[4]) {
                            if constexpr (kVB == 8) {
                                const uint4 w0 = *reinterpret_cast<const uint4*>(vs + k0 * 8), w1 = *reinterpret_cast<const uint4*>(vs + k0 * 8 + 16);
                                a[0] = __hiloint2double((int)w0.y, (int)w0.x), a[1] = __hiloint2double((int)w0.w, (int)w0.z);
                                a[2] = __hiloint2double((int)w1.y, (int)w1.x), a[3] = __hiloint2double((int)w1.w, (int)w1.z);
                            } else if constexpr (kVB == 4) {
                                const uint4 w = *reinterpret_cast<const uint4*>(vs + k0 * 4);
                                a[0] = __uint_as_float(w.x), a[1] = __uint_as_float(w.y), a[2] = __uint_as_float(w.z), a[3] = __uint_as_float(w.w);
                            } else {
                                const uint2 w = *reinterpret_cast<const uint2*>(vs + k0 * 2);
                                if constexpr (kDot2) {   // the packed pairs (entries k0, k0+1) and (k0+2, k0+3), untouched
                                    a[0] = __uint_as_float(w.x), a[1] = __uint_as_float(w.y), a[2] = a[3] = 0.f;
                                } else {
                                    a[0] = __uint_as_float(w.x << 16), a[1] = __uint_as_float(w.x & 0xffff0000u);
                                    a[2] = __uint_as_float(w.y << 16), a[3] = __uint_as_float(w.y & 0xffff0000u);
                                }
                            }
                        };
                        if constexpr (NCH > 0) {
                            // Software pipeline over chunks of four entries: the dense rows of chunk i+1 are requested before
                            // chunk i is consumed, so that a wave issues LDS reads at the rate it retires them instead of
                            // queueing a whole row's reads behind every other wave's (LDS and VALU then run side by side).
                            // The empty asm statements are compiler fences: they pin the order of the LDS requests.
                            // Three stages: values + records of chunk i+2, dense rows of chunk i+1, FMAs of chunk i.
                            A a[3][4];
                            int4 ro[3];
                            uint4 b[2][4][CPL];
                            auto stage_a = [&](int i) {
                                chunk_vals(4 * i, a[i % 3]);
                                ro[i % 3] = *reinterpret_cast<const int4*>(tb + 16 * i);
                            };
                            auto stage_b = [&](int i) {
                                const int rv[4] = {ro[i % 3].x, ro[i % 3].y, ro[i % 3].z, ro[i % 3].w};
#pragma unroll
                                for (int j = 0; j < 4; ++j) load_b(rv[j], b[i & 1][j]);
                            };
                            stage_a(0);
                            if (NCH > 1) stage_a(1);
                            asm volatile("" ::: "memory");
                            stage_b(0);
#pragma unroll
                            for (int i = 0; i < NCH; ++i) {
                                if (i + 2 < NCH) stage_a(i + 2);
                                asm volatile("" ::: "memory");
                                if (i + 1 < NCH) stage_b(i + 1);
                                asm volatile("" ::: "memory");
                                axpy4(a[i % 3], b[i & 1]);
                            }
                        } else {
#pragma unroll 2
                            for (int k0 = 0; k0 < recw; k0 += 4) {
                                A a[4];
                                chunk_vals(k0, a);
                                const int4 ro = *reinterpret_cast<const int4*>(tb + k0 * 4);
                                const int rv[4] = {ro.x, ro.y, ro.z, ro.w};
                                uint4 b[4][CPL];
#pragma unroll
                                for (int j = 0; j < 4; ++j) load_b(rv[j], b[j]);
                                axpy4(a, b);
                            }
                        }
                        if constexpr (kCanDot) {
                            if (want_dot) {
#pragma unroll
                                for (int cp = 0; cp < CPL; ++cp) {
                                    // <C[row,:], W[row,:]>: W = the gathered operand (its own row is the centre of the halo plane in LDS) or a
                                    // second operand read from memory (BiCGSTAB's <r0, A q>)
                                    const uint4 wr = P.dot_w == nullptr
                                                         ? *reinterpret_cast<const uint4*>(sm + ph * PB + cen[q][cp])
                                                         : *reinterpret_cast<const uint4*>(static_cast<const char*>(P.dot_w) + (int64_t)prow_prev * ldob + coo[q][cp]);
                                    A f[VEC];
                                    widen(wr, f);
#pragma unroll
                                    for (int v = 0; v < VEC; ++v) dotp[cp][v] = fma(acc[q][cp][v], f[v], dotp[cp][v]);
                                }
                            }
                        }
                    } else if constexpr (MODE == kLatSddmm) {
                        A* const st = reinterpret_cast<A*>(sm + P.o_vals + csl[q]);   // staging row in the accumulation type (slot = recw*sizeof(A) bytes in this mode)
                        plen[q] = len;
                        prst[q] = cur.rst[q];
                        auto consume = [&](int k0, const uint4 (&bb)[4][CPL]) {
                            A dsum[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                A d;
                                if constexpr (kVB == 2) {
                                    // packed bf16 pairs straight into the dot instruction (fp32 accumulation, no widening)
                                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                                    d = 0.f;
#pragma unroll
                                    for (int cp = 0; cp < CPL; ++cp) {
                                        const uint32_t ow[4] = {cur.own[q][cp].x, cur.own[q][cp].y, cur.own[q][cp].z, cur.own[q][cp].w};
                                        const uint32_t bw[4] = {bb[j][cp].x, bb[j][cp].y, bb[j][cp].z, bb[j][cp].w};
#pragma unroll
                                        for (int i = 0; i < 4; ++i)
                                            d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ow[i]), __builtin_bit_cast(bf2, bw[i]), d, false);
                                    }
                                } else {
                                    A f[VEC], o[VEC];
                                    widen(bb[j][0], f);
                                    widen(cur.own[q][0], o);
                                    d = o[0] * f[0];
#pragma unroll
                                    for (int v = 1; v < VEC; ++v) d = fma(o[v], f[v], d);
                                    if constexpr (CPL == 2) {
                                        widen(bb[j][1], f);
                                        widen(cur.own[q][1], o);
#pragma unroll
                                        for (int v = 0; v < VEC; ++v) d = fma(o[v], f[v], d);
                                    }
                                }
                                dsum[j] = group_sum<A, LPR>(d);
                            }
                            if constexpr (LPR >= 4) {
                                if (c < 4) {
                                    const A mine = c == 0 ? dsum[0] : (c == 1 ? dsum[1] : (c == 2 ? dsum[2] : dsum[3]));
                                    st[k0 + c] = (A)P.alpha * mine;
                                }
                            } else {   // two lanes per row: each stages two of the four sums
                                st[k0 + c] = (A)P.alpha * (c == 0 ? dsum[0] : dsum[1]);
                                st[k0 + 2 + c] = (A)P.alpha * (c == 0 ? dsum[2] : dsum[3]);
                            }
                        };
                        if constexpr (NCH > 0) {   // three stages, as in the SpMM above
                            int4 ro[3];
                            uint4 b[2][4][CPL];
                            auto stage_b = [&](int i) {
                                const int rv[4] = {ro[i % 3].x, ro[i % 3].y, ro[i % 3].z, ro[i % 3].w};
#pragma unroll
                                for (int j = 0; j < 4; ++j) load_b(rv[j], b[i & 1][j]);
                            };
                            ro[0] = *reinterpret_cast<const int4*>(tb);
                            if (NCH > 1) ro[1] = *reinterpret_cast<const int4*>(tb + 16);
                            asm volatile("" ::: "memory");
                            stage_b(0);
#pragma unroll
                            for (int i = 0; i < NCH; ++i) {
                                if (i + 2 < NCH) ro[(i + 2) % 3] = *reinterpret_cast<const int4*>(tb + 16 * (i + 2));
                                asm volatile("" ::: "memory");
                                if (i + 1 < NCH) stage_b(i + 1);
                                asm volatile("" ::: "memory");
                                consume(4 * i, b[i & 1]);
                            }
                        } else {
#pragma unroll 2
                            for (int k0 = 0; k0 < recw; k0 += 4) {
                                const int4 ro = *reinterpret_cast<const int4*>(tb + k0 * 4);
                                const int rv[4] = {ro.x, ro.y, ro.z, ro.w};
                                uint4 b[4][CPL];
#pragma unroll
                                for (int j = 0; j < 4; ++j) load_b(rv[j], b[j]);
                                consume(k0, b);
                            }
                        }
                    } else {
                        // transposed walk: value of entry k of halo row i sits at slot k of i's staged value row
                        // (padded entries point both reads beyond the LDS allocation: 0 · 0)
                        const char* const vcb = sm + P.o_vals + csl[q];
                        auto load_val = [](const char* at) -> A {
                            if constexpr (kVB == 8) return *reinterpret_cast<const double*>(at);
                            else if constexpr (kVB == 4) return *reinterpret_cast<const float*>(at);
                            else if constexpr (kDot2) return __uint_as_float((uint32_t)*reinterpret_cast<const unsigned short*>(at));   // raw bits
                            else return __uint_as_float((uint32_t)*reinterpret_cast<const unsigned short*>(at) << 16);
                        };
                        // kDot2: the raw bf16 values of four entries -> two packed pairs in p2[0], p2[1] (axpy4's form)
                        auto pair_up = [](const A (&a)[4], A (&p2)[4]) {
                            if constexpr (kDot2) {
                                p2[0] = __uint_as_float(__float_as_uint(a[0]) | (__float_as_uint(a[1]) << 16));
                                p2[1] = __uint_as_float(__float_as_uint(a[2]) | (__float_as_uint(a[3]) << 16));
                                p2[2] = p2[3] = 0.f;
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) p2[j] = a[j];
                            }
                        };
#pragma unroll
                        for (int cp = 0; cp < CPL; ++cp) {
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[q][cp][v] = (A)0;
                        }
                        if constexpr (NCH > 0) {   // three stages: records of chunk i+2, dense rows + values of chunk i+1, FMAs of chunk i
                            int4 r01[3], r23[3];
                            uint4 b[2][4][CPL];
                            A a[2][4];
                            auto stage_a = [&](int i) {
                                if constexpr (kPacked) {
                                    r01[i % 3] = *reinterpret_cast<const int4*>(tb + 16 * i);
                                } else {
                                    r01[i % 3] = *reinterpret_cast<const int4*>(tb + 32 * i);
                                    r23[i % 3] = *reinterpret_cast<const int4*>(tb + 32 * i + 16);
                                }
                            };
                            auto stage_b = [&](int i) {
                                int go[4], vo[4];
                                if constexpr (kPacked) {
                                    const int w[4] = {r01[i % 3].x, r01[i % 3].y, r01[i % 3].z, r01[i % 3].w};
#pragma unroll
                                    for (int j = 0; j < 4; ++j) go[j] = w[j] & ~127, vo[j] = w[j];
                                } else {
                                    go[0] = r01[i % 3].x, go[1] = r01[i % 3].z, go[2] = r23[i % 3].x, go[3] = r23[i % 3].z;
                                    vo[0] = r01[i % 3].y, vo[1] = r01[i % 3].w, vo[2] = r23[i % 3].y, vo[3] = r23[i % 3].w;
                                }
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    load_b(go[j], b[i & 1][j]);
                                    a[i & 1][j] = load_val(vcb + vo[j]);
                                }
                            };
                            stage_a(0);
                            if (NCH > 1) stage_a(1);
                            asm volatile("" ::: "memory");
                            stage_b(0);
#pragma unroll
                            for (int i = 0; i < NCH; ++i) {
                                if (i + 2 < NCH) stage_a(i + 2);
                                asm volatile("" ::: "memory");
                                if (i + 1 < NCH) stage_b(i + 1);
                                asm volatile("" ::: "memory");
                                A p2[4];
                                pair_up(a[i & 1], p2);
                                axpy4(p2, b[i & 1]);
                            }
                        } else {
#pragma unroll 2
                            for (int k0 = 0; k0 < recw; k0 += 4) {
                                int go[4], vo[4];
                                if constexpr (kPacked) {
                                    const int4 r = *reinterpret_cast<const int4*>(tb + k0 * 4);
                                    const int w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                                    for (int j = 0; j < 4; ++j) go[j] = w[j] & ~127, vo[j] = w[j];
                                } else {
                                    const int4 r01 = *reinterpret_cast<const int4*>(tb + k0 * 8);
                                    const int4 r23 = *reinterpret_cast<const int4*>(tb + k0 * 8 + 16);
                                    go[0] = r01.x, go[1] = r01.z, go[2] = r23.x, go[3] = r23.z;
                                    vo[0] = r01.y, vo[1] = r01.w, vo[2] = r23.y, vo[3] = r23.w;
                                }
                                uint4 b[4][CPL];
                                A a[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    load_b(go[j], b[j]);
                                    a[j] = load_val(vcb + vo[j]);
                                }
                                A p2[4];
                                pair_up(a, p2);
                                axpy4(p2, b);
                            }
                        }
                    }
                }
            }
        }
        ph = wrap(ph + 1, R);
        vbi = wrap(vbi + 1, VB);
        x_out = x_nxt;
        x_nxt = wrap(x_nxt + 1, P.nx);
        x_run = wrap(x_run + 1, P.nx);
        lat_step_sync();
    }
    flush(prow_prev);
    if constexpr (kCanDot) {
        if (want_dot) {
            // the workgroup's partial row: column cc = chunk·VEC + v summed over the lanes that own that chunk, in lane order
            // (the last step ended with a barrier: the ring is free)
            A* const red = reinterpret_cast<A*>(sm);
#pragma unroll
            for (int cp = 0; cp < CPL; ++cp) {
                const int chunk = c + LPR * cp;
#pragma unroll
                for (int v = 0; v < VEC; ++v) red[(g * CL + chunk) * VEC + v] = dotp[cp][v];
            }
            __syncthreads();
            if (tid < CL * VEC) {
                A s = (A)0;
                for (int r = 0; r < NT / LPR; ++r) s += red[r * CL * VEC + tid];
                static_cast<A*>(P.dot_partial)[vblock * (CL * VEC) + tid] = s;
            }
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------------------

inline int lat_round16(int v) { return (v + 15) / 16 * 16; }

// Fills the LDS layout of P for (mode, CL, sizeof(V)); returns the dynamic LDS bytes or a negative status.
inline int lat_layout(LatParams& P, int mode, int cl, int vbytes, int nt) {
    if (P.ring < 4 || P.ring > 8) return TSGU_ERR_BAD_ARG;
    const int64_t R = P.ring;
    if (P.ty <= 0 || P.tz <= 0 || P.ry < 0 || P.rz < 0 || P.ncls <= 0 || P.ncls > 255 || P.nloc <= 0 || P.nloc > P.ncls || P.recw <= 0 ||
        P.recw % 4 || P.recw > 32)
        return TSGU_ERR_BAD_ARG;
    const int HR = (P.ty + 2 * P.ry) * (P.tz + 2 * P.rz), NR = P.ty * P.tz, RB = cl * 16;
    const bool packed = false && mode == kLatSpmmT && RB % 128 == 0;   // see kPacked in lattice_kernel
    P.slot = packed ? RB : lat_round16(P.recw * (mode == kLatSddmm ? (vbytes == 8 ? 8 : 4) : vbytes));   // SDDMM: a row of the accumulation type
    // a pitch of a multiple of 64 bytes would put the value rows of a wave on four banks (bf16, 28 entries: 64 -> 80 bytes)
    if (!packed && P.slot % 64 == 0) P.slot += 16;
    if (packed && P.recw * vbytes > RB) return TSGU_ERR_TOO_LARGE;
    const int VL = P.slot / 16;
    if (P.cpl != 1 && P.cpl != 2) return TSGU_ERR_BAD_ARG;
    if (cl % P.cpl || (P.cpl == 2 && cl < 4)) return TSGU_ERR_BAD_ARG;
    if ((int64_t)HR * cl > (int64_t)kLatND * nt || NR > kLatNP * (nt / (cl / P.cpl))) return TSGU_ERR_TOO_LARGE;
    if (mode == kLatSpmm && (int64_t)NR * VL > (int64_t)lat_nvd(mode, vbytes) * nt) return TSGU_ERR_TOO_LARGE;
    if (mode == kLatSpmmT && (int64_t)HR * VL > (int64_t)lat_nvd(mode, vbytes) * nt) return TSGU_ERR_TOO_LARGE;
    int64_t o = R * HR * RB;
    P.o_vals = (int)o;
    if (mode == kLatSpmm) o += (R - 2) * NR * P.slot;
    else if (mode == kLatSddmm) o += (int64_t)NR * P.slot;
    else o += R * HR * P.slot;
    P.o_zero = (int)o;
    o += 16;
    P.o_tab = (int)o;
    o += lat_round16(P.ring * P.nloc * P.recw * (mode == kLatSpmmT && !packed ? 8 : 4));
    P.o_len = (int)o;
    P.o_map = (int)o;
    o += 512;     // uint16 per class of the pattern: local class | length << 8
    if (o > kLatMaxLds) return TSGU_ERR_TOO_LARGE;
    P.lds_bytes = (int)o;
    return (int)o;
}

template <typename V, int CL, int CPL, int MODE, int NT, int NCH>
int lat_launch_nch(const LatParams& P, hipStream_t stream) {
    // more than 64 KiB of dynamic LDS has to be allowed once per kernel and device
    static std::atomic<uint64_t> allowed{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TSGU_ERR_RUNTIME;
    if (!(allowed.load(std::memory_order_acquire) >> dev & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lattice_kernel<V, CL, CPL, MODE, NT, NCH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLatMaxLds) != hipSuccess)
            return TSGU_ERR_RUNTIME;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL((lattice_kernel<V, CL, CPL, MODE, NT, NCH>), dim3((unsigned)P.nblocks), dim3(NT), (size_t)P.lds_bytes, stream, P);
    return check_launch();
}

// record widths with an unrolled entry loop: 28 (27-point stencils) and 8 (7-point); anything else loops at run time
template <typename V, int CL, int MODE, int NT>
int lat_launch_one(const LatParams& P, hipStream_t stream) {
    if (P.cpl == 2) {
        // two chunks per lane: compiled for the 27-point record width and the stored-order walks only (measured at C2: no
        // faster than one chunk per lane — half the waves per CU cancel the saved instructions; kept for wider rows)
        if constexpr (CL >= 8 && MODE != kLatSpmmT) {
            if (P.recw == 28) return lat_launch_nch<V, CL, 2, MODE, NT, 7>(P, stream);
        }
        return TSGU_ERR_BAD_ARG;
    }
    // (the unrolled, three-stage products of 2-byte values need more than the 128 registers of four waves per SIMD: they take
    // the run-time loop)
    if (sizeof(V) == 8) return lat_launch_nch<V, CL, 1, MODE, NT, 0>(P, stream);   // fp64: the run-time loop (the unrolled pipeline needs twice the registers)
    if (P.recw == 28 && !(MODE != kLatSddmm && sizeof(V) == 2)) return lat_launch_nch<V, CL, 1, MODE, NT, 7>(P, stream);
    if (P.recw == 8) return lat_launch_nch<V, CL, 1, MODE, NT, 2>(P, stream);
    return lat_launch_nch<V, CL, 1, MODE, NT, 0>(P, stream);
}

}  // namespace tsgu
