// Plane-march kernels: the lattice plane sweep (lattice_impl.h) for FULL periodic box stencils, organised by SOURCE plane.
//
// lattice_impl.h walks an output plane's rows entry by entry: 27 `ds_read_b128` of dense rows per row, and the LDS pipe is
// what its kernels wait for (a build that skips two thirds of those reads is 17-24 us faster per kernel at C2).  When every
// row holds the whole (2·1+1) x NTAP box — a periodic stencil: all rows have 3·NTAP entries, and every row's entries are a
// permutation of the same displacement set — the sum can be taken source plane by source plane instead:
//
//     at step s the dense rows of halo plane s are read ONCE per in-plane displacement ("tap", NTAP = 9) and used for the
//     three output planes s+1 (its dx = -1 part), s (dx = 0) and s-1 (dx = +1), whose accumulators live in registers and
//     rotate; an output plane is complete after three steps.
//
// So a row costs NTAP dense-row reads instead of 3·NTAP, the tap offsets are the same for all rows (nine wave-uniform
// constants: no record tables, no per-entry address arithmetic beyond one add per tap), and only TWO halo planes are
// resident (the one in use and the one being filled) — the ring that limited the residency of lattice_impl.h is halved.
//
// Canonical order.  The class of the interior rows ("ident") stores its entries in ascending (dx, dy, dz): slot
// p·NTAP + i = part p (dx = p - 1), tap i.  Rows of the other classes (wrap-around at a lattice face: 6 % of C2) hold the same
// displacements in another order; `kidx[class][slot]` is the stored position of canonical slot `slot`.  Values are staged in LDS in
// CANONICAL order: waves whose rows are all `ident` copy them with the 16-byte LDS-DMA, the others gather them with the 4-byte
// LDS-DMA (`global_load_lds_dword`, one lane per value, source offset through `kidx`) — both asynchronous, same LDS layout.
// The SDDMM scatters through the same table when it writes a row's gradients to its stage row, so gradA leaves in A's stored
// order as before.  Summation order = canonical order: bit-identical to the plan-free kernels for `ident` rows, equal to
// rounding for the wrapped ones.
//
// Modes:  kLatSpmm   C = A·B      values of the tile's own rows, four plane buffers (targets s-1, s, s+1 + the one being filled)
//         kLatSddmm  gradA        own rows of the three targets in registers; a lane keeps the dots of the slots = its lane (mod CL)
//         kLatSpmmT  gradB = Aᵀ·G  a second two-plane ring holds the canonical VALUE rows of the halo rows: entry (i -> j) sits in
//                                 source row i at slot (dx+1)·NTAP + tap(dy, dz), and the source of target j through tap i' is the
//                                 row at -tap: slot (dx+1)·NTAP + NTAP-1-i' of the row at own + tap i' (the tap list is symmetric).
//                                 No transposed pattern, no transposed plan.
#pragma once

#include <type_traits>

#include "lattice_impl.h"

namespace tsgu {

constexpr int kMarchMaxCls = 64;     // classes of a pattern (27 for a periodic 27-point stencil)
constexpr int kMarchND = 3;          // ring DMA pieces per thread and plane

struct MarchParams {
    int nb, nx, ny, nz;
    int ty, tz, ry, rz;
    int tiles_y, tiles_z;
    int nseg, seg_len;
    int ncls, ident;
    int tap_row[9];              // halo-row displacement dy·HZ + dz of tap i, ascending
    unsigned mask;               // bit p·NTAP + i: displacement (dx = p - 1, tap i) occurs in the pattern (all 27: a full box)
    int per_x, per_y, per_z;     // the lattice is periodic in x / y / z (else: truncated — neighbours beyond a face do not exist)
    int uniform;                 // > 0: every row stores this many entries (`rstart` is not read); 0: rows start at rstart[row]
    const int* rstart;           // [rows + 1] first value position of each row (A's row pointer as int32)
    int accumulate;              // SDDMM: add to gvals instead of overwriting (column tiles of wide dense operands)
    int raw;                     // periodic whole box, rows with sorted columns (the stored position of (dx, dy, dz) is 9·rank_x + 3·rank_y +
                                 // rank_z: checked by the caller): SpMM / SpMMT stage the value rows of interior planes RAW (plain 16-byte copies
                                 // for every wave, no gathers for the rows that wrap in y / z) and resolve the (y, z) order where a value is read
    const unsigned char* kidx;   // [ncls][32] stored position of canonical slot s (0xff: the row has no such entry); byte 31: row length
    const unsigned char* rcls;   // [rows] class of each row
    const void* val;
    int64_t nnz;
    const void* S;               // gathered dense operand
    int64_t lds_;
    const void* Own;             // SDDMM: row operand
    int64_t ldown;
    void* out;                   // C / gradB
    int64_t ldo;
    void* gvals;                 // SDDMM output [nnz]
    float alpha;
    int64_t nblocks;
    int o_vals, o_stage, o_tab, o_rows, lds_bytes;   // LDS layout (bytes), filled by march_layout / march_bwd_layout
};

// 4-byte LDS-DMA: lane l's dword lands at `lds_wave_base + 4*l`.  NT_POLICY: streaming (the values of a tile's own rows are read
// once); off for the transposed product, whose halo value rows the neighbouring tiles read again (they should stay in L2)
template <bool NT_POLICY>
__device__ __forceinline__ void lat_dma4(const void* sbase64, uint32_t voff, unsigned lds_wave_base) {
    unsigned keep;
    if constexpr (NT_POLICY)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase64), "s"(lds_wave_base) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase64), "s"(lds_wave_base) : "memory");
}

template <int I, int N, typename F>
__device__ __forceinline__ void lat_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lat_static_for<I + 1, N>(f);
    }
}

// N <= 8 partial sums per lane -> lane c < N of an 8-lane group gets the group total of d[c] (the other lanes: anything).
// 21 instructions for eight sums (eight butterflies cost 40), and the same summation tree as group_sum<float, 8>:
// ((0+1)+(2+3)) + ((4+5)+(6+7)), bit for bit.
// (scalars, not an array: a select between two array elements becomes an indexed access and the array then lives in a
// ladder of v_cndmask)
template <int CTRL>
__device__ __forceinline__ float lat_pair_sum(bool odd, float a0, float a1) {   // "even" lanes keep a0, "odd" lanes a1
    const float u = odd ? a1 : a0, w = odd ? a0 : a1;
    return u + dpp_move<CTRL>(w);
}
template <int N>
__device__ __forceinline__ float group_sum_t8(float d0, float d1, float d2, float d3, float d4, float d5, float d6, float d7, int c) {
    static_assert(N >= 1 && N <= 8, "one value per lane of the group");
    const bool b0 = (c & 1) != 0, b1 = (c & 2) != 0, b2 = (c & 4) != 0;
    // lanes c^1
    const float t0 = N > 1 ? lat_pair_sum<0xB1>(b0, d0, d1) : d0 + dpp_move<0xB1>(d0);
    const float t1 = N > 3 ? lat_pair_sum<0xB1>(b0, d2, d3) : (N > 2 ? d2 + dpp_move<0xB1>(d2) : 0.f);
    const float t2 = N > 5 ? lat_pair_sum<0xB1>(b0, d4, d5) : (N > 4 ? d4 + dpp_move<0xB1>(d4) : 0.f);
    const float t3 = N > 7 ? lat_pair_sum<0xB1>(b0, d6, d7) : (N > 6 ? d6 + dpp_move<0xB1>(d6) : 0.f);
    // lanes c^2
    const float r0 = N > 2 ? lat_pair_sum<0x4E>(b1, t0, t1) : t0 + dpp_move<0x4E>(t0);
    const float r1 = N > 6 ? lat_pair_sum<0x4E>(b1, t2, t3) : (N > 4 ? t2 + dpp_move<0x4E>(t2) : 0.f);
    // lanes c^4: lanes 0-3 of a group add lane + 4 (row_shl:4), lanes 4-7 add lane - 4 (row_shr:4)
    const float lo = r0 + dpp_move<0x104>(r0);
    if constexpr (N <= 4) return lo;
    const float hi = r1 + dpp_move<0x114>(r1);
    return b2 ? hi : lo;
}

// MASK: the pattern's displacement set — one of the sets the kernels are compiled for (march_sets.h: the whole box; the SDDMM
// of the triangular halves): straight-line code, the tests on MASK fold at compile time.  The whole box (FULL) also copies
// the value rows of waves whose rows are all of the canonical class with the 16-byte DMA; subsets gather every value row
// through kidx (waves of canonical rows: without a look-up).
//
// ROWS — where a row's values start:
//   kRowsUniform  every row stores the same number of entries (P.uniform; periodic lattices): row r starts at r·P.uniform
//   kRowsBox      the whole box on a lattice truncated in some dimension: a row at (x, y, z) stores cx(x)·cy(y)·cz(z) entries, c = 2
//                 at a face of a truncated dimension, else 3 — so the start of a row is arithmetic: a per-row constant of the march
//                 times cx(x) plus a per-plane scalar.  No row pointer is read
//   kRowsPointer  P.rstart[row] — loaded a step ahead next to the class bytes (a load waited for where it is issued would also
//                 wait for the DMAs in flight)
//
// Truncated lattices (P.per_* == 0).  A row at a face simply has no entry towards the neighbours beyond it; its canonical
// slots for them hold 0.  The halo rows beyond a face are ZERO in LDS (never fetched, cleared once), halo planes beyond an x
// face are skipped — so the only products that involve a staged zero value are 0 · 0, and no row touches a dense row it does
// not reference (non-finite operands behave as in the reference).
constexpr uint32_t kBoxAll = (1u << 27) - 1u;      // every displacement of the 3 x 3 x 3 box

enum MarchRows { kRowsPointer = 0, kRowsUniform = 1, kRowsBox = 2, kRowsRaw = 3 };     // kRowsRaw: kRowsUniform + raw value rows (SpMM / SpMMT of the periodic whole box)

template <typename V, int CL, int MODE, int NT, int NTAP, uint32_t MASK, int ROWS>
__global__ __launch_bounds__(NT, 4) void march_kernel(const MarchParams P) {     // 4 waves per SIMD: at most 128 VGPRs
    constexpr bool PTR = ROWS == kRowsPointer, RAWR = ROWS == kRowsRaw, UNIF = ROWS == kRowsUniform || RAWR, BOXA = ROWS == kRowsBox;
    static_assert(!RAWR || (MASK == kBoxAll && MODE != kLatSddmm), "raw value rows: SpMM / SpMMT of the whole box");
    static_assert(!BOXA || MASK == kBoxAll, "row starts by box arithmetic: the whole box only");
    constexpr bool FULL = MASK == kBoxAll;     // the whole box
    static_assert(MASK != 0 && MASK <= kBoxAll, "a displacement set of the 3 x 3 x 3 box");
    static_assert(sizeof(V) == 4, "fp32 values and operands");
    constexpr int RB = CL * 16;                 // bytes of a dense row
    constexpr int NG = NT / CL;                 // row groups of the workgroup
    constexpr int RPW = kWave / CL;             // rows per wave
    constexpr int NS = 3 * NTAP;                // canonical slots per row
    constexpr int SLOTS = (NS + 3) / 4 * 4;     // slots of a staged value row
    constexpr int VP = SLOTS * 4;               // its pitch in bytes (112: the rows of a wave fall on different banks)
    constexpr int VL = SLOTS / 4;               // its 16-byte pieces
    constexpr int NGI = (RPW * SLOTS + kWave - 1) / kWave;   // 4-byte DMA instructions that gather the value rows of a wave
    constexpr int NPASS = MODE == kLatSpmmT ? 2 : 1;        // staged rows: tile rows (one per group) or halo rows (up to two)
    constexpr int RJ = (NS + CL - 1) / CL;      // SDDMM: dots a lane keeps per target
    constexpr int NF = (RPW * VL + kWave - 1) / kWave;       // 16-byte DMA instructions that copy the value rows of a wave (plain path)
    static_assert(NT % kWave == 0 && kWave % CL == 0 && VP % 64 != 0 && NS < 31, "geometry");

    extern __shared__ uint4 lat_smem[];
    char* const sm = reinterpret_cast<char*>(lat_smem);
    const unsigned sbase = lat_lds_addr(lat_smem);

    const int tid = threadIdx.x;
    const int lane = tid % kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int c = tid % CL;
    const int g = tid / CL;

    const int HZ = P.tz + 2 * P.rz, HY = P.ty + 2 * P.ry, HR = HY * HZ, NR = P.ty * P.tz;
    const int PB = HR * RB;
    const int plane_rows = P.ny * P.nz;
    auto wrap = [](int v, int m) { return v >= m ? v - m : v; };
    // displacement `bit` occurs in the pattern: a compile-time fact wherever `bit` is one (the unrolled tap loops)
    auto has = [](int bit) -> bool { return ((MASK >> bit) & 1u) != 0; };
    constexpr uint32_t mask = MASK;

    // ---- the tile and x segment of this workgroup (as lattice_kernel) ---------------------------------------------
    const int64_t vblock = xcd_chunked_block(blockIdx.x, P.nblocks);
    int64_t vb = vblock;
    const int tzi = (int)(vb % P.tiles_z);
    vb /= P.tiles_z;
    const int tyi = (int)(vb % P.tiles_y);
    vb /= P.tiles_y;
    const int seg = (int)(vb % P.nseg);
    const int item = (int)(vb / P.nseg);
    const int xs = seg * P.seg_len;
    const int L = P.seg_len < P.nx - xs ? P.seg_len : P.nx - xs;   // output planes: ring indices 1 .. L; halo planes 0 and L+1
    const int y0 = tyi * P.ty, z0 = tzi * P.tz;
    const int item_row0 = item * P.nx * plane_rows;
    auto row_of_x = [&](int x) -> int { return item_row0 + x * plane_rows; };
    // ring index s of this segment is lattice plane xs - 1 + s of the item: beyond an x face of a truncated lattice there is none
    auto x_ok = [&](int s) -> bool { return P.per_x || (unsigned)(xs - 1 + s) < (unsigned)P.nx; };
    // halo row (hy, hz) of the tile: its row inside a plane, or -1 beyond a y / z face of a truncated lattice
    auto halo_row = [&](int hy, int hz) -> int {
        const int yy = y0 - P.ry + hy, zz = z0 - P.rz + hz;
        const bool in = (P.per_y || (unsigned)yy < (unsigned)P.ny) && (P.per_z || (unsigned)zz < (unsigned)P.nz);
        return in ? lat_mod(yy, P.ny) * P.nz + lat_mod(zz, P.nz) : -1;
    };

    // kRowsBox: entries per point along a dimension (3, or 2 at a face of a truncated dimension) and their prefix sums
    auto cnt1 = [](int t, int n, int per) -> int { return per ? 3 : 3 - (t == 0) - (t == n - 1); };
    auto pre1 = [](int t, int per) -> int { return per ? 3 * t : 3 * t - (t > 0); };
    const int boxLz = pre1(P.nz, P.per_z) - (P.per_z ? 0 : 1), boxLy = pre1(P.ny, P.per_y) - (P.per_y ? 0 : 1);
    const int boxLyz = boxLy * boxLz, boxLtot = (pre1(P.nx, P.per_x) - (P.per_x ? 0 : 1)) * boxLyz;
    // the per-row constant of row r (inside its plane): its start is plane_base(x) + plane_cx(x)·row_const(r)
    auto row_const = [&](int r) -> int {
        if constexpr (BOXA) {
            const int y = r / P.nz, z = r - y * P.nz;
            return boxLz * pre1(y, P.per_y) + cnt1(y, P.ny, P.per_y) * pre1(z, P.per_z);
        } else {
            return r;
        }
    };
    auto plane_cx = [&](int x) -> int {
        if constexpr (BOXA) return cnt1(x, P.nx, P.per_x);
        else return FULL ? NS : P.uniform;
    };
    auto plane_base = [&](int x) -> int {
        if constexpr (BOXA) return item * boxLtot + boxLyz * pre1(x, P.per_x);
        else return row_of_x(x) * (FULL ? NS : P.uniform);
    };

    // ---- tables -> LDS: kidx, and the row-in-plane index of every staged row (-1: outside the lattice) ----------------
    const int staged_rows = MODE == kLatSpmmT ? HR : NR;
    {
        const int* src = reinterpret_cast<const int*>(P.kidx);
        int* dst = reinterpret_cast<int*>(sm + P.o_tab);
        for (int i = tid; i < P.ncls * 8; i += NT) dst[i] = src[i];
        int* rows = reinterpret_cast<int*>(sm + P.o_rows);
        for (int r = tid; r < staged_rows; r += NT) {
            int v;
            if constexpr (MODE == kLatSpmmT) {
                const int hy = r / HZ, hz = r - hy * HZ;
                v = halo_row(hy, hz);
            } else {
                const int ly = r / P.tz, lz = r - ly * P.tz;
                v = (y0 + ly < P.ny && z0 + lz < P.nz) ? (y0 + ly) * P.nz + z0 + lz : -1;
            }
            rows[r] = v;
        }
    }
    __syncthreads();
    const unsigned char* const kidx_s = reinterpret_cast<const unsigned char*>(sm + P.o_tab);
    const int* const rows_s = reinterpret_cast<const int*>(sm + P.o_rows);
    // RAW value rows (periodic whole box, MarchParams::raw).  A row with sorted columns stores (dx, dy, dz) at position 9·t + q: t = the rank
    // of its x-neighbour plane (= dx + 1 in interior planes, rotated at the two x faces: the same for all rows of a plane), q = 3·rank_y +
    // rank_z (depends on the row's (y, z) only: a per-lane constant of the march).  Rows are staged AS STORED, by plain 16-byte copies in
    // EVERY wave (with canonical rows the waves that hold a row wrapping in y / z — the tiles at the y / z faces, at every step — gather
    // 28 values per row with 4-byte requests, and a launch takes as long as its slowest workgroup: forward 73 -> 70 us, transposed
    // product 80 -> 70 us at C2 with every wave on the plain copy).  The readers resolve q (tapl / tvl below).  The x-face planes (t
    // rotated; two planes of nx): the forward has their parts put in order while they are staged (arithmetic 4-byte requests: one tap
    // loop — a second copy with run-time offsets measured 4 us slower), the transposed product copies them plainly too and reads them
    // through a second copy of its tap loop with run-time value offsets (3 us faster than gathering them).
    constexpr bool rawv = RAWR;
    auto xrank = [&](int x, int d) -> int { return x == 0 ? (d < 0 ? 2 : d) : (x == P.nx - 1 ? (d > 0 ? 0 : d + 2) : d + 1); };

    // ---- per-thread descriptors ---------------------------------------------------------------------------------
    const uint32_t ldsb = (uint32_t)P.lds_ * 4u;
    uint32_t roff[kMarchND];
    const int ring_pieces = HR * CL;
#pragma unroll
    for (int d = 0; d < kMarchND; ++d) {
        const int e = d * NT + tid;
        const int hr = e / CL;
        const int hy = hr / HZ, hz = hr - hy * HZ;
        const int hrow_in_plane = e < ring_pieces ? halo_row(hy, hz) : -1;
        roff[d] = hrow_in_plane >= 0 ? (uint32_t)hrow_in_plane * ldsb + (uint32_t)c * 16u : kLatNone;
    }
    if (!(P.per_y && P.per_z)) {
        // truncated in y or z: the halo rows beyond a face are zero in both ring slots (no DMA ever writes them); so are their
        // staged value rows in the transposed product
#pragma unroll
        for (int d = 0; d < kMarchND; ++d) {
            const int e = d * NT + tid;
            if (e < ring_pieces && roff[d] == kLatNone) {
                lat_smem[e] = make_uint4(0, 0, 0, 0);
                lat_smem[HR * CL + e] = make_uint4(0, 0, 0, 0);
            }
        }
        if constexpr (MODE == kLatSpmmT) {
            for (int e = tid; e < HR * VL; e += NT) {
                if (rows_s[e / VL] < 0) {
                    *reinterpret_cast<uint4*>(sm + P.o_vals + e * 16) = make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4*>(sm + P.o_vals + HR * VP + e * 16) = make_uint4(0, 0, 0, 0);
                }
            }
            // (a staged slot of an in-lattice halo row that holds no entry is never read: its reader would be a target beyond a face)
        }
        if constexpr (MODE == kLatSpmm) {
            // The product multiplies the canonical slots of a row at a y / z face with the (zero) halo rows beyond that face: those
            // slots hold 0 — cleared ONCE: a tile row keeps its (y, z) for the whole march, so no request ever writes them.  (Slots
            // that are absent only at an x face are never read: the source plane beyond an x face is skipped.)
            for (int e = tid; e < NR * SLOTS; e += NT) {
                const int r = e / SLOTS, slot = e - r * SLOTS;
                const int ry_ = r / P.tz, rz_ = r - ry_ * P.tz;
                const int yy = y0 + ry_ + (slot / 3) % 3 - 1, zz = z0 + rz_ + slot % 3 - 1;
                const bool beyond = (!P.per_y && (unsigned)yy >= (unsigned)P.ny) || (!P.per_z && (unsigned)zz >= (unsigned)P.nz);
                if (slot < NS && beyond) {
#pragma unroll
                    for (int vbi = 0; vbi < 4; ++vbi) *reinterpret_cast<float*>(sm + P.o_vals + vbi * NR * VP + e * 4) = 0.f;
                }
            }
        }
    }
    // the compute row of this lane group
    const int ly = g / P.tz, lz = g - ly * P.tz;
    const bool ok = g < NR && y0 + ly < P.ny && z0 + lz < P.nz;
    const int crow = ok ? (y0 + ly) * P.nz + z0 + lz : -1;          // row inside its plane
    const int hrow = (ly + P.ry) * HZ + lz + P.rz;                  // its position in a halo plane
    const uint32_t coo = (uint32_t)(ok ? crow : 0) * ((uint32_t)P.ldo * 4u) + (uint32_t)c * 16u;
    const uint32_t cown = (uint32_t)(ok ? crow : 0) * ((uint32_t)P.ldown * 4u) + (uint32_t)c * 16u;
    const int cen = hrow * RB + c * 16;

    // staged value rows: pass q, wave w stages rows q·NG + w·RPW .. + RPW - 1 (tile rows, or halo rows for the transposed product)
    //   srow[q]  the row whose class / first value position this lane loads (its group's row)
    //   foff[q][f]  plain path: this lane's f-th 16-byte piece — kLatNone: nothing to copy; else (rows of uniform length) its byte
    //               offset inside the plane's values
    int srow[NPASS];
    uint32_t foff[NPASS][NF];
    if constexpr (MODE != kLatSddmm) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int r = q * NG + g;
            srow[q] = r < staged_rows ? rows_s[r] : -1;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int piece = f * kWave + lane;
                const int fr = q * NG + wave * RPW + piece / VL;
                const int frow = (piece < RPW * VL && fr < staged_rows) ? rows_s[fr] : -1;
                if constexpr (BOXA) foff[q][f] = frow >= 0 ? (uint32_t)row_const(frow) * 4u : kLatNone;
                else foff[q][f] = frow >= 0 ? (uint32_t)frow * (uint32_t)(NS * 4) + (uint32_t)(piece % VL) * 16u : kLatNone;
            }
        }
    }

    // kRowsBox: the per-row constant of the row each gather lane serves (its start is plane_base + plane_cx · constant), once for
    // the whole march — the face tiles gather at every step, and a launch takes as long as its slowest workgroup
    uint32_t fpo[NF];          // kRowsBox: byte offset of this lane's f-th 16-byte piece inside its row (row constants < 2^24)
#pragma unroll
    for (int f = 0; f < NF; ++f) fpo[f] = (uint32_t)((f * kWave + lane) % VL) * 16u;
    int gconst[BOXA ? NPASS : 1][BOXA ? NGI : 1];
    if constexpr (BOXA && MODE != kLatSddmm) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
#pragma unroll
            for (int n = 0; n < NGI; ++n) {
                const int rw = (n * kWave + lane) / SLOTS;
                const int fr = q * NG + wave * RPW + rw;
                const int rr = (rw < RPW && fr < staged_rows) ? rows_s[fr] : -1;
                gconst[q][n] = row_const(rr > 0 ? rr : 0);
            }
        }
    }

    // INTERIOR planes (1 <= x <= nx - 2): the class of the row at (x, y, z) depends on (y, z) only — whether it wraps / sits at a
    // face in y or z — so everything a wave looks up to stage its value rows is the same at every step: found ONCE here.
    //   mid_plain  bit q: the wave's rows of pass q are all of the canonical class (plain 16-byte copies)
    //   gk[q][n]   the gather lane's source, relative to the plane: stored position of its slot (+ NS · row for rows of one
    //              length), or -1: nothing to request
    // The tiles at a y / z face gather at EVERY step, and a launch takes as long as its slowest workgroup: with the look-ups
    // (two dependent LDS round trips per slot group) inside the march the C2 forward took 83 us, the transposed product 98 us;
    // with every wave on the plain copy (wrong results) 69 / 78 us.  Only the two x-face planes still look up per step.
    int mid_plain = 0;
    int gk[PTR ? 1 : NPASS][PTR ? 1 : NGI];
    if constexpr (MODE != kLatSddmm && !PTR && !RAWR) {
        const int prow_mid = row_of_x(1);
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int first = q * NG + wave * RPW;
            const int cm = srow[q] >= 0 ? (int)P.rcls[prow_mid + srow[q]] : P.ident;
            if (FULL && __builtin_amdgcn_ballot_w64(srow[q] >= 0 && cm != P.ident) == 0) mid_plain |= 1 << q;
            const bool allid = !FULL && __builtin_amdgcn_ballot_w64(srow[q] >= 0 && cm != P.ident) == 0;
#pragma unroll
            for (int n = 0; n < NGI; ++n) {
                const int e = n * kWave + lane;
                const int rw = e / SLOTS, slot = e - rw * SLOTS;
                const int rc = __builtin_amdgcn_ds_bpermute((rw < RPW ? rw * CL : 0) * 4, cm);
                const int rr = (rw < RPW && first + rw < staged_rows) ? rows_s[first + rw] : -1;
                const int k = allid ? __builtin_popcount(mask & ((1u << (slot & 31)) - 1u)) : (int)kidx_s[rc * 32 + (slot & 31)];
                const bool want = first < staged_rows && rr >= 0 && slot < NS && has(slot) && k != 0xFF;
                gk[q][n] = !want ? -1 : (UNIF ? (FULL ? NS : P.uniform) * rr + k : k);
            }
        }
    }

    const char* const Sb = static_cast<const char*>(P.S);
    const char* const valb = static_cast<const char*>(P.val);
    const uint32_t val_bytes = (uint32_t)(P.nnz * 4);

    auto dma_ring = [&](int prow, int slot) {
        const char* const pbase = Sb + (int64_t)prow * ldsb;
        const unsigned base = sbase + (unsigned)(slot * PB) + (unsigned)(wave * kWave * 16);
#pragma unroll
        for (int d = 0; d < kMarchND; ++d) {
            if (d * NT < ring_pieces) {
                if (roff[d] != kLatNone) lat_dma16<false>(pbase, roff[d], base + (unsigned)(d * NT * 16));
            }
        }
    };
    // canonical value rows of a lattice plane of the item into the buffer at byte `region`; cls[q] / stt[q] = class and (row
    // pointers) first value position of srow[q]
    // (pbase, pcx: plane_base / plane_cx of the plane — kept up to date by the march, below; unused with row pointers)
    // mid: an interior plane — no look-ups (above)
    auto stage_vals = [&](int pbase, int pcx, bool mid, int xv, unsigned region, const int (&cls)[NPASS], const int (&stt)[NPASS]) {
        if constexpr (MODE != kLatSddmm) {
#pragma unroll
            for (int q = 0; q < NPASS; ++q) {
                const int first = q * NG + wave * RPW;                       // wave-uniform
                if (first < staged_rows) {
                    const unsigned wbase = sbase + region + (unsigned)(first * VP);
                    // (raw rows: plain copies in every wave — the forward puts the parts of the two x-face planes in order while staging
                    // them (arithmetic 4-byte requests below), the transposed product reads them rotated)
                    const bool plain = FULL && (rawv ? (MODE == kLatSpmmT || mid)
                                                     : (!PTR && mid ? (mid_plain >> q & 1) != 0
                                                                    : __builtin_amdgcn_ballot_w64(srow[q] >= 0 && cls[q] != P.ident) == 0));
                    if (plain) {
                        // rows of the canonical class hold all NS values in canonical order: a plain copy
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            // foff = 4 · (the per-row constant of the piece's row) (+ the piece's offset inside the row: uniform rows)
                            uint32_t fo = foff[q][f];
                            if constexpr (UNIF) {
                                if (fo != kLatNone) fo += (uint32_t)pbase * 4u;
                            } else if constexpr (BOXA) {
                                if (fo != kLatNone) fo = __umul24((uint32_t)pcx, fo) + ((uint32_t)pbase * 4u + fpo[f]);   // (one v_mad_u32_u24)
                            } else {
                                const int piece = f * kWave + lane;
                                const int rw = piece / VL;
                                const int rs = __builtin_amdgcn_ds_bpermute((rw < RPW ? rw * CL : 0) * 4, stt[q]);
                                if (fo != kLatNone) fo = (uint32_t)rs * 4u + (uint32_t)(piece % VL) * 16u;
                            }
                            if (fo != kLatNone) {
                                if (__builtin_expect(fo + 16u <= val_bytes, 1)) {
                                    lat_dma16<MODE == kLatSpmm>(valb, fo, wbase + (unsigned)(f * kWave * 16));
                                } else {   // the last 16 bytes of the value array: element-wise, never reading beyond the array
                                    float* dst = reinterpret_cast<float*>(sm + region + first * VP + (f * kWave + lane) * 16);
#pragma nounroll
                                    for (int e = 0; e < 4; ++e)
                                        dst[e] = fo + (e + 1) * 4 <= val_bytes ? *reinterpret_cast<const float*>(valb + fo + e * 4) : 0.f;
                                }
                            }
                        }
                    } else if (!PTR && !rawv && mid) {
                        // an interior plane: every lane knows its source
#pragma unroll
                        for (int n = 0; n < NGI; ++n) {
                            const int gkn = gk[PTR ? 0 : q][PTR ? 0 : n];
                            uint32_t src;
                            if constexpr (BOXA) src = (uint32_t)(pbase + gkn) * 4u + __umul24((uint32_t)pcx, (uint32_t)gconst[q][n]) * 4u;
                            else src = (uint32_t)(pbase + gkn) * 4u;
                            if (gkn >= 0) lat_dma4<MODE == kLatSpmm>(valb, src, wbase + (unsigned)(n * kWave * 4));
                        }
                    } else {
                        // (the planes at an x face) gathered through kidx, one lane per canonical slot, one slot group after the other (finding all sources
                        // first and then issuing the requests back to back measured 10-20 us SLOWER per launch at C2).  A wave
                        // whose rows are all of the canonical class (the bulk of a pattern that is a SUBSET of the box) needs no
                        // look-up: the stored position of a slot is the number of the pattern's displacements below it.
                        const bool allid = !FULL && __builtin_amdgcn_ballot_w64(srow[q] >= 0 && cls[q] != P.ident) == 0;
#pragma unroll
                        for (int n = 0; n < NGI; ++n) {
                            const int e = n * kWave + lane;
                            const int rw = e / SLOTS, slot = e - rw * SLOTS;
                            const int src_lane = (rw < RPW ? rw * CL : 0) * 4;
                            const int rc = __builtin_amdgcn_ds_bpermute(src_lane, cls[q]);
                            const int rr = (rw < RPW && first + rw < staged_rows) ? rows_s[first + rw] : -1;
                            int rs;
                            if constexpr (PTR) rs = __builtin_amdgcn_ds_bpermute(src_lane, stt[q]);
                            else if constexpr (BOXA) rs = pbase + (int)__umul24((uint32_t)pcx, (uint32_t)gconst[q][n]);
                            else rs = pbase + pcx * (rr > 0 ? rr : 0);
                            // (a row at a face of a truncated lattice has no entry towards the neighbours beyond it: k = 0xff,
                            // nothing is requested — its slot was cleared once, below, or is never read)
                            int k = allid ? __builtin_popcount(mask & ((1u << (slot & 31)) - 1u)) : (int)kidx_s[rc * 32 + (slot & 31)];
                            if (rawv) k = NTAP * xrank(xv, slot / NTAP - 1) + slot % NTAP;      // (raw rows, an x-face plane: only the parts are put in order)
                            if (rr >= 0 && slot < NS && has(slot) && (UNIF || k != 0xFF))
                                lat_dma4<MODE == kLatSpmm>(valb, (uint32_t)rs * 4u + (uint32_t)k * 4u, wbase + (unsigned)(n * kWave * 4));
                        }
                    }
                }
            }
        }
    };
    auto load_cls = [&](int prow, int (&cls)[NPASS], int (&stt)[NPASS]) {
        if constexpr (MODE != kLatSddmm) {
            const unsigned char* const cbase = P.rcls + prow;
#pragma unroll
            for (int q = 0; q < NPASS; ++q) {
                cls[q] = P.ident;
                if constexpr (PTR) stt[q] = 0;
                if (srow[q] >= 0) {
                    cls[q] = cbase[(uint32_t)srow[q]];
                    if constexpr (PTR) stt[q] = P.rstart[prow + srow[q]];
                }
            }
        }
    };
    auto pin_cls = [&](int (&cls)[NPASS], int (&stt)[NPASS]) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            lat_pin(cls[q]);
            if constexpr (PTR) lat_pin(stt[q]);
        }
    };

    int tapb[NTAP];   // byte offset of tap i from the row's own position in a halo plane (wave-uniform)
#pragma unroll
    for (int i = 0; i < NTAP; ++i) tapb[i] = P.tap_row[i] * RB;
    // raw value rows — forward: register i of a part holds the value at STORED position i (the same vector reads as with canonical rows)
    // and is multiplied with the dense row of the tap stored there (tapl[i]; canonical rows: tapb[i]);  transposed product: the value
    // triple of tap i sits at position q of its SOURCE row's parts (tvq[i] = 4·q bytes; canonical rows: 4·(8 - i))
    int tapl[NTAP], tvq[NTAP];
#pragma unroll
    for (int i = 0; i < NTAP; ++i) tapl[i] = tapb[i], tvq[i] = 4 * (NTAP - 1 - i);
    if constexpr (RAWR) {
        {
            const int prow_mid = row_of_x(1);
            if constexpr (MODE == kLatSpmm) {
                const int cm = crow >= 0 ? (int)P.rcls[prow_mid + crow] : P.ident;
#pragma unroll
                for (int i = 0; i < NTAP; ++i) {
                    const int q = (int)kidx_s[cm * 32 + NTAP + i] - NTAP;      // stored position of tap i inside the dx = 0 part
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) tapl[t] = q == t ? tapb[i] : tapl[t];
                }
            } else {
#pragma unroll
                for (int i = 0; i < NTAP; ++i) {
                    const int sr = crow >= 0 ? rows_s[hrow + P.tap_row[i]] : -1;          // the source of tap i (a halo row)
                    const int cs = sr >= 0 ? (int)P.rcls[prow_mid + sr] : P.ident;
                    tvq[i] = 4 * ((int)kidx_s[cs * 32 + NTAP + (NTAP - 1 - i)] - NTAP);   // its entry towards this row: displacement -tap i
                }
            }
        }
    }
    // SDDMM of the whole box at eight lanes per row: tap i ^ c for register i (see XR below)
    int tapx[8];
    if constexpr (MODE == kLatSddmm && MASK == kBoxAll && CL == 8 && NTAP == 9) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            tapx[i] = tapb[0];
#pragma unroll
            for (int t = 1; t < 8; ++t) tapx[i] = (i ^ c) == t ? tapb[t] : tapx[i];
        }
    }

    auto as4 = [](const uint4& raw, float (&f)[4]) {
        f[0] = __uint_as_float(raw.x), f[1] = __uint_as_float(raw.y), f[2] = __uint_as_float(raw.z), f[3] = __uint_as_float(raw.w);
    };

    // lattice planes: ring index s of this segment is plane (xs - 1 + s) mod nx of the item
    int x_ring = lat_mod(xs - 1, P.nx);

    if constexpr (MODE == kLatSpmm || MODE == kLatSpmmT) {
        constexpr int VB = MODE == kLatSpmm ? 4 : 2;                        // value buffers
        const int vbuf = (MODE == kLatSpmm ? NR : HR) * VP;
        // SpMM stages the values of target plane s+2 at step s; SpMMT those of halo plane s+1 (with the dense plane)
        int x_val = MODE == kLatSpmm ? xs : x_ring;                         // lattice plane of the next value plane (ring 1 / ring 0)
        // its plane_base / plane_cx, stepped with it: base(x + 1) = base(x) + (entries of plane x) — a handful of scalar
        // instructions per step (the closed forms cost ~25, and these kernels are bound by instruction issue as much as by HBM)
        int vbase = PTR ? 0 : plane_base(x_val), vcx = PTR ? 0 : plane_cx(x_val);
        auto is_mid = [&](int x) -> bool { return !PTR && x >= 1 && x <= P.nx - 2; };     // (wave-uniform)
        auto next_val_plane = [&]() {
            if constexpr (BOXA) vbase += boxLyz * vcx;
            else if constexpr (UNIF) vbase += plane_rows * (FULL ? NS : P.uniform);
            x_val = wrap(x_val + 1, P.nx);
            if constexpr (!PTR) {
                if (x_val == 0) vbase = plane_base(0);
                if constexpr (BOXA) vcx = cnt1(x_val, P.nx, P.per_x);
            }
        };
        int cls[NPASS], cld[NPASS], stt[NPASS], std_[NPASS];
        const int first_val = MODE == kLatSpmm ? 1 : 0;                     // ring index of the first value plane
        const int last_val = MODE == kLatSpmm ? L : L + 1;
#pragma unroll
        for (int q = 0; q < NPASS; ++q) cls[q] = cld[q] = P.ident, stt[q] = std_[q] = 0;
        if (x_ok(first_val) && !is_mid(x_val)) {
            load_cls(row_of_x(x_val), cls, stt);
            pin_cls(cls, stt);
        }
        if (x_ok(0)) dma_ring(row_of_x(x_ring), 0);
        if (x_ok(first_val)) stage_vals(vbase, vcx, is_mid(x_val), x_val, (unsigned)(P.o_vals + (MODE == kLatSpmm ? 1 : 0) * vbuf), cls, stt);
        x_ring = wrap(x_ring + 1, P.nx);
        next_val_plane();
        const int first_next = first_val + 1;                               // ring index of the next value plane
        if (first_next <= last_val && x_ok(first_next) && !is_mid(x_val)) load_cls(row_of_x(x_val), cld, std_);
        lat_step_sync();

        float accP[4], accC[4], accN[4], done[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) accP[v] = accC[v] = accN[v] = done[v] = 0.f;
        int x_out = xs;          // lattice plane of the next result to leave (ring index 1)
        const uint32_t ldob = (uint32_t)P.ldo * 4u;
        auto flush = [&]() {
            if (crow >= 0) {
                char* const obase = static_cast<char*>(P.out) + (int64_t)row_of_x(x_out) * ldob;
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v o = {done[0], done[1], done[2], done[3]};
                __builtin_nontemporal_store(o, reinterpret_cast<f4v*>(obase + coo));
            }
            x_out = wrap(x_out + 1, P.nx);
        };
        for (int s = 0; s <= L + 1; ++s) {
            // 1. the class bytes loaded during the previous step; the results that were completed by it
            pin_cls(cld, std_);
#pragma unroll
            for (int q = 0; q < NPASS; ++q) cls[q] = cld[q], stt[q] = std_[q];
            if (s >= 3) flush();
            // 2. asynchronous fetches: the next halo plane, the next value plane
            if (s + 1 <= L + 1 && x_ok(s + 1)) dma_ring(row_of_x(x_ring), (s + 1) & 1);
            const int vnext = MODE == kLatSpmm ? s + 2 : s + 1;
            if (vnext <= last_val && x_ok(vnext)) stage_vals(vbase, vcx, is_mid(x_val), x_val, (unsigned)(P.o_vals + (vnext & (VB - 1)) * vbuf), cls, stt);
            x_ring = wrap(x_ring + 1, P.nx);
            next_val_plane();
            // 3. class bytes for the next step
            if (vnext + 1 <= last_val && x_ok(vnext + 1) && !is_mid(x_val)) load_cls(row_of_x(x_val), cld, std_);
            // 4. source plane s: targets s+1 (N, part 0), s (C, part 1), s-1 (P, part 2).  A target outside 1..L accumulates
            // whatever its value buffer holds: it is never stored
#pragma unroll
            for (int v = 0; v < 4; ++v) accN[v] = 0.f;
            if (crow >= 0 && x_ok(s)) {
                const char* const bb = sm + (s & 1) * PB + cen;
                if constexpr (MODE == kLatSpmm) {
                    // values: part p of a row = slots p·NTAP .. p·NTAP + NTAP - 1 of its canonical row
                    const char* const vrow = sm + P.o_vals + g * VP;
                    const char* const vp[3] = {vrow + ((s + 1) & 3) * vbuf, vrow + (s & 3) * vbuf, vrow + ((s - 1) & 3) * vbuf};
                    constexpr int kF0 = 0, kF1 = NTAP / 4, kF2 = 2 * NTAP / 4;
                    constexpr int kN0 = (NTAP - 1) / 4 - kF0 + 1, kN1 = (2 * NTAP - 1) / 4 - kF1 + 1, kN2 = (3 * NTAP - 1) / 4 - kF2 + 1;
                    constexpr int kNA = kN0 > kN1 ? (kN0 > kN2 ? kN0 : kN2) : (kN1 > kN2 ? kN1 : kN2);
                    float a[3][kNA * 4];
                    constexpr int first[3] = {kF0, kF1, kF2}, count[3] = {kN0, kN1, kN2};
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
#pragma unroll
                        for (int m = 0; m < count[p]; ++m) {
                            const uint4 w = *reinterpret_cast<const uint4*>(vp[p] + (first[p] + m) * 16);
                            a[p][4 * m] = __uint_as_float(w.x), a[p][4 * m + 1] = __uint_as_float(w.y);
                            a[p][4 * m + 2] = __uint_as_float(w.z), a[p][4 * m + 3] = __uint_as_float(w.w);
                        }
                    }
                    if constexpr (FULL) {
                        uint4 b[NTAP];
                        constexpr int kAhead = 3;
                        // (raw rows: register i = stored position i of each part, multiplied with the dense row of the tap stored there)
#pragma unroll
                        for (int i = 0; i < kAhead && i < NTAP; ++i) b[i] = *reinterpret_cast<const uint4*>(bb + tapl[i]);
#pragma unroll
                        for (int i = 0; i < NTAP; ++i) {
                            if (i + kAhead < NTAP) b[i + kAhead] = *reinterpret_cast<const uint4*>(bb + tapl[i + kAhead]);
                            asm volatile("" ::: "memory");
                            float f[4];
                            as4(b[i], f);
                            const float a0 = a[0][i - 4 * first[0]], a1 = a[1][NTAP + i - 4 * first[1]], a2 = a[2][2 * NTAP + i - 4 * first[2]];
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                accN[v] = fmaf(a0, f[v], accN[v]);
                                accC[v] = fmaf(a1, f[v], accC[v]);
                                accP[v] = fmaf(a2, f[v], accP[v]);
                            }
                        }
                    } else {
                        // a subset of the box: the dense row of a tap is read when any of its three parts occurs — all reads
                        // first (one wait), then the products (wave-uniform branches on the pattern's displacement set)
                        uint4 b[NTAP];
#pragma unroll
                        for (int i = 0; i < NTAP; ++i) {
                            b[i] = make_uint4(0, 0, 0, 0);
                            if (has(i) || has(NTAP + i) || has(2 * NTAP + i)) {
                                b[i] = *reinterpret_cast<const uint4*>(bb + tapb[i]);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < NTAP; ++i) {
                            float f[4];
                            as4(b[i], f);
                            const float a0 = a[0][i - 4 * first[0]], a1 = a[1][NTAP + i - 4 * first[1]], a2 = a[2][2 * NTAP + i - 4 * first[2]];
                            if (has(i)) {
#pragma unroll
                                for (int v = 0; v < 4; ++v) accN[v] = fmaf(a0, f[v], accN[v]);
                            }
                            if (has(NTAP + i)) {
#pragma unroll
                                for (int v = 0; v < 4; ++v) accC[v] = fmaf(a1, f[v], accC[v]);
                            }
                            if (has(2 * NTAP + i)) {
#pragma unroll
                                for (int v = 0; v < 4; ++v) accP[v] = fmaf(a2, f[v], accP[v]);
                            }
                        }
                    }
                } else {
                    // transposed walk: the source through tap i is the halo row at own + tap i; its entry towards a target
                    // at dx sits at canonical slot (dx+1)·NTAP + (NTAP-1-i) of ITS value row
                    const char* const vb0 = sm + P.o_vals + (s & 1) * vbuf + hrow * VP;
                    int tapv[NTAP], tvl[NTAP];      // value row of tap i's source (wave-uniform) / + the position of its triple (raw rows: per lane)
#pragma unroll
                    for (int i = 0; i < NTAP; ++i) tapv[i] = P.tap_row[i] * VP, tvl[i] = tapv[i] + tvq[i];
                    if constexpr (FULL) {
                        // the part of target dx sits at x-position rank(dx) of the SOURCE plane: 2 / 1 / 0 for N / C / P — immediate offsets;
                        // raw rows at an x-face source plane: rotated, run-time offsets in a second copy of the loop (ROT)
                        auto taps = [&](auto rot, int oN, int oC, int oP) {
                            constexpr bool ROT = decltype(rot)::value;
                            uint4 b[NTAP];
                            float a[NTAP][3];
                            constexpr int kAhead = 2;
                            auto fetch = [&](int i) {
                                b[i] = *reinterpret_cast<const uint4*>(bb + tapb[i]);
                                const char* const vr = vb0 + tvl[i];
                                if constexpr (ROT) {
                                    a[i][0] = *reinterpret_cast<const float*>(vr + oN), a[i][1] = *reinterpret_cast<const float*>(vr + oC);
                                    a[i][2] = *reinterpret_cast<const float*>(vr + oP);
                                } else {
#pragma unroll
                                    for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const float*>(vr + (2 - p) * (NTAP * 4));
                                }
                            };
#pragma unroll
                            for (int i = 0; i < kAhead && i < NTAP; ++i) fetch(i);
#pragma unroll
                            for (int i = 0; i < NTAP; ++i) {
                                if (i + kAhead < NTAP) fetch(i + kAhead);
                                asm volatile("" ::: "memory");
                                float f[4];
                                as4(b[i], f);
                                // part index here counts the TARGET: N = s+1 (dx = +1), C = s (dx = 0), P = s-1 (dx = -1)
                                const float aN = a[i][0], aC = a[i][1], aP = a[i][2];
#pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    accN[v] = fmaf(aN, f[v], accN[v]);
                                    accC[v] = fmaf(aC, f[v], accC[v]);
                                    accP[v] = fmaf(aP, f[v], accP[v]);
                                }
                            }
                        };
                        bool face = false;
                        if constexpr (RAWR) {
                            int xsrc = xs - 1 + s;
                            xsrc = xsrc < 0 ? xsrc + P.nx : (xsrc >= P.nx ? xsrc - P.nx : xsrc);
                            face = xsrc == 0 || xsrc == P.nx - 1;
                            if (face) taps(std::true_type{}, xrank(xsrc, 1) * (NTAP * 4), xrank(xsrc, 0) * (NTAP * 4), xrank(xsrc, -1) * (NTAP * 4));
                        }
                        if (!face) taps(std::false_type{}, 0, 0, 0);
                    } else {
                        uint4 b[NTAP];
                        float a[NTAP][3];
#pragma unroll
                        for (int i = 0; i < NTAP; ++i) {
                            const int sN = 2 * NTAP + NTAP - 1 - i, sC = NTAP + NTAP - 1 - i, sP = NTAP - 1 - i;
                            const char* const vr = vb0 + tapv[i];
                            b[i] = make_uint4(0, 0, 0, 0);
                            a[i][0] = a[i][1] = a[i][2] = 0.f;
                            if (has(sN) || has(sC) || has(sP)) {
                                b[i] = *reinterpret_cast<const uint4*>(bb + tapb[i]);
                                if (has(sN)) a[i][0] = *reinterpret_cast<const float*>(vr + sN * 4);
                                if (has(sC)) a[i][1] = *reinterpret_cast<const float*>(vr + sC * 4);
                                if (has(sP)) a[i][2] = *reinterpret_cast<const float*>(vr + sP * 4);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < NTAP; ++i) {
                            const int sN = 2 * NTAP + NTAP - 1 - i, sC = NTAP + NTAP - 1 - i, sP = NTAP - 1 - i;
                            float f[4];
                            as4(b[i], f);
                            if (has(sN)) {
#pragma unroll
                                for (int v = 0; v < 4; ++v) accN[v] = fmaf(a[i][0], f[v], accN[v]);
                            }
                            if (has(sC)) {
#pragma unroll
                                for (int v = 0; v < 4; ++v) accC[v] = fmaf(a[i][1], f[v], accC[v]);
                            }
                            if (has(sP)) {
#pragma unroll
                                for (int v = 0; v < 4; ++v) accP[v] = fmaf(a[i][2], f[v], accP[v]);
                            }
                        }
                    }
                }
            }
            // 5. target s-1 is complete; rotate
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                done[v] = accP[v];
                accP[v] = accC[v];
                accC[v] = accN[v];
            }
            lat_step_sync();
        }
        flush();   // target L (target L-1 left at the top of step L+1; with L = 1 nothing left earlier)
    } else {
        // ---- SDDMM -----------------------------------------------------------------------------------------------
        // Whole box at eight lanes per row (XR): lane c walks the taps 0..7 in the order r ^ c (r = 0..7; per-lane tap offsets, no cost
        // in the loop), so the partial dot of tap t sits in register t ^ c — then the eight-lane sums of eight taps need NO selects:
        // recursive halving with fixed register pairs (R[2r] + xor1(R[2r+1]), …) leaves the total of tap c in lane c: 9 instructions
        // per part instead of 21 (same summation tree as group_sum<float, 8>, operands commuted: the same bits).  A lane then keeps,
        // per target, the dots of canonical slots 9·j + c (register j = part j) and — lanes 0..2, register 3 — the ninth tap's 9·c + 8.
        constexpr bool XR = FULL && CL == 8 && NTAP == 9;
        auto slot_of = [&](int j) -> int { return XR ? (j < 3 ? 9 * j + c : (c < 3 ? 9 * c + 8 : NS)) : j * CL + c; };
        struct Own {
            uint4 row;
            int cls, start;
        };
        // interior planes (1 <= x <= nx - 2): the row's class — and with it the stored positions of the dots this lane writes — is
        // the same at every step (see the value staging above): looked up once, no class byte is loaded for those planes
        auto is_mid = [&](int x) -> bool { return x >= 1 && x <= P.nx - 2; };
        int kpk[(RJ + 3) / 4];
        int len_mid;
        {
            const int cm = crow >= 0 ? (int)P.rcls[row_of_x(1) + crow] : P.ident;
#pragma unroll
            for (int w = 0; w < (RJ + 3) / 4; ++w) kpk[w] = -1;
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                const int slot = slot_of(j);
                const int k = (slot < NS && has(slot)) ? (int)kidx_s[cm * 32 + slot] : 0xFF;
                kpk[j / 4] = (kpk[j / 4] & ~(0xFF << (8 * (j % 4)))) | (k << (8 * (j % 4)));
            }
            len_mid = kidx_s[cm * 32 + 31];
        }
        auto load_own = [&](int x, Own& o) {
            const int prow = row_of_x(x);
            o.cls = P.ident;
            o.start = 0;
            o.row = make_uint4(0, 0, 0, 0);
            if (crow >= 0) {
                if (!is_mid(x)) o.cls = P.rcls[prow + crow];
                if constexpr (PTR) o.start = P.rstart[prow + crow];
                o.row = *reinterpret_cast<const uint4*>(static_cast<const char*>(P.Own) + (int64_t)prow * P.ldown * 4 + cown);
            }
        };
        auto pin_own = [&](Own& o) {
            lat_pin(o.cls);
            if constexpr (PTR) lat_pin(o.start);
            lat_pin(o.row.x);
            lat_pin(o.row.y);
            lat_pin(o.row.z);
            lat_pin(o.row.w);
        };
        Own oP, oC, oN, old;
        oP.row = oC.row = make_uint4(0, 0, 0, 0);
        oP.cls = oC.cls = P.ident;
        oP.start = oC.start = 0;
        int x_own = xs;                                       // lattice plane of ring index 1
        load_own(x_own, oN);
        pin_own(oN);
        x_own = wrap(x_own + 1, P.nx);
        old = oN;
        if (x_ok(0)) dma_ring(row_of_x(x_ring), 0);
        x_ring = wrap(x_ring + 1, P.nx);
        lat_step_sync();

        float rP[RJ], rC[RJ], rN[RJ];
#pragma unroll
        for (int j = 0; j < RJ; ++j) rP[j] = rC[j] = rN[j] = 0.f;
        // Stage rows (wave-private).  Rows of ONE length on the whole box (PACKED): the pitch is the row itself (NS values), so the
        // RPW rows of a wave — consecutive rows of one z-line, i.e. one contiguous run of gradA — are one contiguous run in LDS too and
        // leave as 16-byte pieces of the RUN: RPW·NS / 4 lanes, one store instruction, every piece 16-byte aligned when the run is
        // (round 5: six 16-byte pieces at 4-byte alignment + three single words per ROW — four store instructions per wave).
        constexpr bool PACKED = FULL && UNIF && (RPW * NS) % 4 == 0;
        constexpr int SP = PACKED ? NS * 4 : VP;
        const bool wave_packed = PACKED && P.tz % RPW == 0 && __all(crow >= 0);      // (wave-uniform: all RPW rows of the wave exist)
        float* const st = reinterpret_cast<float*>(sm + P.o_vals + g * SP);   // this row's stage row
        const int own_const = PTR ? 0 : row_const(crow > 0 ? crow : 0);   // start of this lane's row in plane x: plane_base(x) + plane_cx(x)·own_const
        bool staged = false;
        int fl_start = 0, fl_len = 0;                          // first value position and length of the staged row
        int x_out = xs;                                        // lattice plane of the next target to be staged (ring index 1)
        auto flush = [&]() {
            if constexpr (PACKED) {
                if (wave_packed) {
                    // (all lanes of the wave are here: `staged` depends on the step only once every row exists)
                    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                    const int first = __builtin_amdgcn_readfirstlane(fl_start);          // lane 0: the wave's first row
                    if (lane < RPW * NS / 4) {
                        const float4 w4 = *reinterpret_cast<const float4*>(sm + P.o_vals + (wave * RPW) * SP + lane * 16);
                        f4u* const dst = reinterpret_cast<f4u*>(static_cast<float*>(P.gvals) + first) + lane;
                        f4u o = {w4.x, w4.y, w4.z, w4.w};
                        if (P.accumulate) {
                            const f4u prev = *dst;
                            o += prev;
                        }
                        __builtin_nontemporal_store(o, dst);
                    }
                    return;
                }
            }
            // the row's gradients in stored order: 16-byte pieces, single elements at the end
            if (crow >= 0) {
                float* const go = static_cast<float*>(P.gvals) + fl_start;
                auto piece = [&](int k0) {
                    float4 w = *reinterpret_cast<const float4*>(st + k0);
                    if (k0 + 4 <= fl_len) {
                        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                        if (P.accumulate) {
                            const f4u prev = *reinterpret_cast<const f4u*>(go + k0);
                            w.x += prev.x, w.y += prev.y, w.z += prev.z, w.w += prev.w;
                        }
                        const f4u o = {w.x, w.y, w.z, w.w};
                        __builtin_nontemporal_store(o, reinterpret_cast<f4u*>(go + k0));
                    } else {
                        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (k0 + j < fl_len) go[k0 + j] = P.accumulate ? go[k0 + j] + wv[j] : wv[j];
                    }
                };
                if constexpr (UNIF) {
                    // (rows of one length: the rolled loop costs one register less — 80, six waves per SIMD)
#pragma nounroll
                    for (int k0 = c * 4; k0 < fl_len; k0 += CL * 4) piece(k0);
                } else {
                    // a lane holds one 16-byte piece of a row (8 / 16 lanes) or two (4 lanes): unrolled, 17 registers less than the
                    // loop with a run-time bound
                    constexpr int kPieces = (SLOTS + CL * 4 - 1) / (CL * 4);
#pragma unroll
                    for (int it = 0; it < kPieces; ++it) {
                        const int k0 = c * 4 + it * CL * 4;
                        if (k0 < fl_len) piece(k0);
                    }
                }
            }
        };
        for (int s = 0; s <= L + 1; ++s) {
            // 1. take over the rows loaded during the previous step: they are target s+1 now
            if (s >= 1) {
                pin_own(old);
                oP = oC, oC = oN, oN = old;
            }
            if (staged) flush();
            staged = false;
            // 2. the next halo plane; 3. own rows of target s+2
            if (s + 1 <= L + 1 && x_ok(s + 1)) dma_ring(row_of_x(x_ring), (s + 1) & 1);
            x_ring = wrap(x_ring + 1, P.nx);
            if (s + 2 <= L) load_own(x_own, old);
            x_own = wrap(x_own + 1, P.nx);
            // 4. source plane s
            if (crow >= 0) {
              if (x_ok(s)) {
                const char* const bb = sm + (s & 1) * PB + cen;
                // The dots of targets s+1 and s share every dense row: they run as the two halves of packed instructions (own
                // columns paired {N, C}, the dense column broadcast by op_sel — each half is the same mul + fma chain as the
                // scalar form, bit for bit); target s-1 runs scalar.
                typedef float f2v __attribute__((ext_vector_type(2)));
                f2v nc[4];
                float op[4];
                {
                    float on[4], oc[4];
                    as4(oN.row, on);
                    as4(oC.row, oc);
                    as4(oP.row, op);
#pragma unroll
                    for (int v = 0; v < 4; ++v) nc[v] = f2v{on[v], oc[v]};
                }
                float pd[3][NTAP];
                if constexpr (FULL) {
                    uint4 b[NTAP];
                    constexpr int kAhead = 3;
                    // (XR: register i < 8 takes tap i ^ c; the ninth tap is the same for all lanes)
                    auto tap_at = [&](int i) -> int { return XR && i < 8 ? tapx[i] : tapb[i]; };
#pragma unroll
                    for (int i = 0; i < kAhead && i < NTAP; ++i) b[i] = *reinterpret_cast<const uint4*>(bb + tap_at(i));
#pragma unroll
                    for (int i = 0; i < NTAP; ++i) {
                        if (i + kAhead < NTAP) b[i + kAhead] = *reinterpret_cast<const uint4*>(bb + tap_at(i + kAhead));
                        asm volatile("" ::: "memory");
                        float f[4];
                        as4(b[i], f);
                        const f2v f01 = {f[0], f[1]}, f23 = {f[2], f[3]};
                        f2v d2;
                        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d2) : "v"(nc[0]), "v"(f01));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d2) : "v"(nc[1]), "v"(f01));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d2) : "v"(nc[2]), "v"(f23));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d2) : "v"(nc[3]), "v"(f23));
                        // (as instructions: the vectoriser would pair the dots of neighbouring taps instead and pay two register
                        // moves per pair to line their operands up)
                        float d;
                        asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(op[0]), "v"(f[0]));
#pragma unroll
                        for (int v = 1; v < 4; ++v) asm("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(op[v]), "v"(f[v]));
                        pd[0][i] = d2.x, pd[1][i] = d2.y, pd[2][i] = d;
                    }
                } else {
                    // a subset of the box: the same mul + fma chain per dot, only for the displacements that occur (all dense
                    // rows are read first)
                    uint4 b[NTAP];
#pragma unroll
                    for (int i = 0; i < NTAP; ++i) {
                        b[i] = make_uint4(0, 0, 0, 0);
                        if (has(i) || has(NTAP + i) || has(2 * NTAP + i)) {
                            b[i] = *reinterpret_cast<const uint4*>(bb + tapb[i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NTAP; ++i) {
                        float f[4];
                        as4(b[i], f);
                        pd[0][i] = pd[1][i] = pd[2][i] = 0.f;
                        if (has(i)) {
                            float d;
                            asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(nc[0].x), "v"(f[0]));
#pragma unroll
                            for (int v = 1; v < 4; ++v) asm("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(nc[v].x), "v"(f[v]));
                            pd[0][i] = d;
                        }
                        if (has(NTAP + i)) {
                            float d;
                            asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(nc[0].y), "v"(f[0]));
#pragma unroll
                            for (int v = 1; v < 4; ++v) asm("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(nc[v].y), "v"(f[v]));
                            pd[1][i] = d;
                        }
                        if (has(2 * NTAP + i)) {
                            float d;
                            asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(op[0]), "v"(f[0]));
#pragma unroll
                            for (int v = 1; v < 4; ++v) asm("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(op[v]), "v"(f[v]));
                            pd[2][i] = d;
                        }
                    }
                }
                // the 3·NTAP partial dots of this step, by canonical slot p·NTAP + i; slot j·CL + c belongs to lane c, register j
                if constexpr (XR) {
                    auto halve8 = [&](const float (&R)[NTAP]) -> float {       // registers 0..7: tap r ^ c;  -> the eight-lane total of tap c
                        const float u0 = R[0] + dpp_move<0xB1>(R[1]), u1 = R[2] + dpp_move<0xB1>(R[3]);
                        const float u2 = R[4] + dpp_move<0xB1>(R[5]), u3 = R[6] + dpp_move<0xB1>(R[7]);
                        const float v0 = u0 + dpp_move<0x4E>(u1), v1 = u2 + dpp_move<0x4E>(u3);
                        // lane c ^ 4: lanes 0..3 take lane + 4 (row_shl:4, banks 0 and 2), lanes 4..7 lane - 4 (row_shr:4, banks 1 and 3)
                        int y = __builtin_amdgcn_update_dpp(0, __float_as_int(v1), 0x104, 0xF, 0x5, false);
                        y = __builtin_amdgcn_update_dpp(y, __float_as_int(v1), 0x114, 0xF, 0xA, false);
                        return v0 + __int_as_float(y);
                    };
                    rN[0] = halve8(pd[0]);
                    rC[1] = halve8(pd[1]);
                    rP[2] = halve8(pd[2]);
                    // the ninth tap: lane 0 / 1 / 2 gets the total of part 0 / 1 / 2
                    const float t8 = group_sum_t8<3>(pd[0][8], pd[1][8], pd[2][8], 0.f, 0.f, 0.f, 0.f, 0.f, c);
                    rN[3] = c == 0 ? t8 : rN[3];
                    rC[3] = c == 1 ? t8 : rC[3];
                    rP[3] = c == 2 ? t8 : rP[3];
                } else if constexpr (CL == 8) {
                    // eight slots at a time: the transposed reduction leaves the total of slot 8j + c in lane c
                    lat_static_for<0, RJ>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        if (!FULL && ((mask >> (8 * j)) & 0xffu) == 0) return;   // none of these eight displacements occurs
                        auto sl = [&](auto E) -> float {      // partial dot of canonical slot 8j + e
                            constexpr int slot = 8 * j + decltype(E)::value;
                            if constexpr (slot < NS) return pd[slot / NTAP][slot % NTAP];
                            else return 0.f;
                        };
                        using std::integral_constant;
                        const float tot = group_sum_t8<(NS - 8 * j < 8 ? NS - 8 * j : 8)>(
                            sl(integral_constant<int, 0>{}), sl(integral_constant<int, 1>{}), sl(integral_constant<int, 2>{}), sl(integral_constant<int, 3>{}),
                            sl(integral_constant<int, 4>{}), sl(integral_constant<int, 5>{}), sl(integral_constant<int, 6>{}), sl(integral_constant<int, 7>{}), c);
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            const int lo = (p * NTAP > 8 * j ? p * NTAP : 8 * j) - 8 * j, hi = ((p + 1) * NTAP < 8 * j + 8 ? (p + 1) * NTAP : 8 * j + 8) - 8 * j;
                            if (lo < hi) {
                                float& r = p == 0 ? rN[j] : (p == 1 ? rC[j] : rP[j]);
                                const bool whole = lo == 0 && (hi == 8 || 8 * j + hi == NS);   // (lanes beyond the last slot are never read)
                                r = whole || (c >= lo && c < hi) ? tot : r;
                            }
                        }
                    });
                } else {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
#pragma unroll
                        for (int i = 0; i < NTAP; ++i) {
                            if (has(p * NTAP + i)) {
                                const float tot = group_sum<float, CL>(pd[p][i]);
                                const int slot = p * NTAP + i;
                                float& r = p == 0 ? rN[slot / CL] : (p == 1 ? rC[slot / CL] : rP[slot / CL]);
                                r = c == slot % CL ? tot : r;
                            }
                        }
                    }
                }
              }
                // 5. target s-1 is complete: its dots go to the stage row at their STORED positions
                if (s >= 2) {
                    const bool mid = is_mid(x_out);                    // (wave-uniform)
                    const bool plain = FULL && oP.cls == P.ident;
#pragma unroll
                    for (int j = 0; j < RJ; ++j) {
                        const int slot = slot_of(j);
                        if (slot < NS && has(slot)) {
                            int k;
                            if (mid) k = (kpk[j / 4] >> (8 * (j % 4))) & 0xFF;
                            else k = plain ? slot : (int)kidx_s[oP.cls * 32 + slot];
                            if (UNIF || k != 0xFF) st[k] = P.alpha * rP[j];
                        }
                    }
                    if constexpr (PTR) fl_start = oP.start;
                    else fl_start = plane_base(x_out) + plane_cx(x_out) * own_const;
                    fl_len = (FULL && UNIF) ? NS : (mid ? len_mid : (int)kidx_s[oP.cls * 32 + 31]);
                    staged = true;
                }
            }
            if (s >= 2) x_out = wrap(x_out + 1, P.nx);
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                rP[j] = rC[j];
                rC[j] = rN[j];
            }
            lat_step_sync();
        }
        if (staged) flush();
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------

// Fills the LDS layout of P for (mode, CL); returns the dynamic LDS bytes or a negative status.
inline int march_layout(MarchParams& P, int mode, int cl, int nt, int ntap) {
    if (P.ty <= 0 || P.tz <= 0 || P.ry < 0 || P.rz < 0 || P.ncls <= 0 || P.ncls > kMarchMaxCls || ntap != 9) return TSGU_ERR_BAD_ARG;
    const int HR = (P.ty + 2 * P.ry) * (P.tz + 2 * P.rz), NR = P.ty * P.tz, RB = cl * 16;
    const int VP = (3 * ntap + 3) / 4 * 4 * 4;
    const int NG = nt / cl;
    if ((int64_t)HR * cl > (int64_t)kMarchND * nt || NR > NG) return TSGU_ERR_TOO_LARGE;
    if (mode == kLatSpmmT && HR > 2 * NG) return TSGU_ERR_TOO_LARGE;
    int64_t o = 2 * (int64_t)HR * RB;
    P.o_vals = (int)o;
    if (mode == kLatSpmm) o += 4 * (int64_t)NR * VP;
    else if (mode == kLatSddmm) o += (int64_t)NR * VP;
    else o += 2 * (int64_t)HR * VP;
    P.o_tab = (int)o;
    o += P.ncls * 32;
    P.o_rows = (int)o;
    o += lat_round16((mode == kLatSpmmT ? HR : NR) * 4);
    if (o > kLatMaxLds) return TSGU_ERR_TOO_LARGE;
    P.lds_bytes = (int)o;
    return (int)o;
}

template <typename V, int CL, int MODE, int NT, uint32_t MASK, int ROWS>
int march_launch(const MarchParams& P, hipStream_t stream) {
    static std::atomic<uint64_t> allowed{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TSGU_ERR_RUNTIME;
    if (!(allowed.load(std::memory_order_acquire) >> dev & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&march_kernel<V, CL, MODE, NT, 9, MASK, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLatMaxLds) != hipSuccess)
            return TSGU_ERR_RUNTIME;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL((march_kernel<V, CL, MODE, NT, 9, MASK, ROWS>), dim3((unsigned)P.nblocks), dim3(NT), (size_t)P.lds_bytes, stream, P);
    return check_launch();
}

}  // namespace tsgu
