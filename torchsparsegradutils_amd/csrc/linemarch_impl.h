// Whole-line plane march (bf16, 16 dense columns): the plane march of march_impl.h for periodic 27-point box stencils whose
// value rows are 54 bytes — BASELINE configs[4] (C5: batched, 64 x 64 x 32 lattices, 16 RHS, bf16).
//
// Why another kernel.  The fp32 march stages the value rows of a plane in CANONICAL order (slot = (dx+1)·9 + tap): waves of
// interior rows copy them with the 16-byte LDS-DMA, rows that wrap around a face gather them value by value with the 4-byte
// DMA.  A 2-byte value cannot be placed by a DMA of its own size, a dense row of 32 bytes is owned by TWO lanes (so a wave
// covers a whole z-line of 32 points and ALWAYS contains the two rows that wrap in z), and round 3's bf16 march — every value
// gathered as a dword — spent more instructions on the gather than on the walk (0.81-0.95 ms against 0.55 ms for the general
// sweep).  The general sweep in turn keeps four planes of halo value rows + record tables in LDS (121 KB at C5): one workgroup
// per CU with four computing waves, 418 us per transposed product for 1.0 GB of traffic.
//
// Here the value rows are staged RAW (stored order, whole z-lines = nz·54 contiguous bytes, 16-byte DMA) and the permutation
// is resolved where a value is READ.  Columns of a CSR row are sorted, so the stored position of displacement (dx, dy, dz) in
// the row at (x, y, z) is 9·rank_x + 3·rank_y + rank_z, rank_d = rank of the (wrapped) neighbour coordinate among the three
// of its dimension (checked against the lattice plan's class tables by the caller, `_lattice.linemarch_ok`).  A lane keeps
// (y, z) for the whole march, so 3·rank_y + rank_z of its nine source rows are nine per-lane constants folded into the LDS
// addresses; rank_x is the same for all rows of a source plane: wave-uniform per step.
// A tile is TY whole z-lines (the z wrap is inside the line: no z halo, folded into the per-lane addresses as well), two
// planes of (TY + 2) lines are resident: 55 KB at TY = 8 — two workgroups of eight waves per CU, ALL of them computing.
//
// Launched back to back these kernels are bound by instruction issue rather than by memory, so:
// the step loop is unrolled six times (TSGU_LINE_SIX_STEPS: register roles and plane buffer at compile time), the forward
// and the SDDMM walk the nine STORED POSITIONS of an x-part (value offsets are immediates; per position the lane keeps the
// dense-row address of the tap stored there), the kernels are compiled per line length (NZ: every LDS offset an immediate),
// the schedule is pinned pair by pair (an empty asm on the accumulators), and accumulators written by v_dot2c are never
// moved or converted by instructions the compiler cannot see (the dot pipeline needs wait states before an ordinary VALU read).
//
// Sum per target row: source planes x-1, x, x+1 in that order (plane by plane as in march_impl.h), inside a plane by pairs —
// of taps in ascending (dy, dz) (transposed product) or of stored positions (forward) — two entries per v_dot2_f32_bf16, fp32
// accumulation, one rounding at the end: every element within one bf16 ulp of the exact result, like the sweep's (same
// tests), not bit-identical to it.
#pragma once

#include "lattice_impl.h"

namespace tsgu {

constexpr int kLineKG = 2;     // 16-byte DMA pieces per thread and plane: dense rows ((ty + 2)·nz·2 = NT + 4·nz <= 2·NT)
constexpr int kLineKV = 4;     //   value rows with halo lines (transposed product);  own lines only (forward): NT·27/16 <= 2·NT
constexpr int kLineKVOwn = 2;
constexpr int kLineRowB = 32;  // bytes of a dense row (16 bf16 columns)
constexpr int kLineValB = 54;  // bytes of a value row (27 bf16 values)

// two floats -> packed bf16 pair (round to nearest even): v_cvt_pk_bf16_f32 through the compiler, NOT inline asm — the results of
// v_dot2c need wait states before another VALU reads them (they run in the matrix pipeline on gfx950) and hipcc inserts those only
// for consumers it can see: an asm conversion right behind the last dot read a stale accumulator (column 7 of every row)
__device__ __forceinline__ uint32_t line_cvt2(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
}

// one bf16 value at an LDS byte address, as a 16-bit read that stays one: two values at neighbouring positions merged into one
// ds_read_b32 / b64 sit on the 2-byte boundary of an odd row — unaligned LDS accesses (the forward kernel ran 315 instead of 190 us);
// volatile keeps them apart, the explicit address space keeps them DS instructions (a volatile generic pointer becomes flat_load)
// (the position is an index of the pointer, not part of the integer address: it then folds into the instruction's offset field)
__device__ __forceinline__ uint32_t line_lds_u16(unsigned lds_addr, int position) {
    typedef const volatile __attribute__((address_space(3))) unsigned short* lds_u16p;
    return reinterpret_cast<lds_u16p>(static_cast<size_t>(lds_addr))[position];
}

struct LineParams {
    int nb, nx, ny, nz;
    int ty, tiles_y;
    int nseg, seg_len;
    const void* val;         // [rows][27] bf16, A's own order
    const void* S;           // gathered dense operand (G for the transposed product)
    int64_t lds_;
    void* out;
    int64_t ldo;
    int64_t nblocks;
    int g_bytes;             // dense rows of one resident plane: (ty + 2)·nz·32
    int buf_stride;          // distance of the two plane buffers: a power of two >= the bytes of one (transposed product), their sum (SDDMM)
    int lds_bytes;
    // SDDMM
    const void* Own;         // row operand [rows][ldown]
    int64_t ldown;
    void* gvals;             // [nnz] bf16, A's stored order
    float alpha;
    int o_stage, stage_bytes;   // three stage planes of ty·nz·54 bytes behind the two plane buffers
};

// Shared by the three kernels: the rank of the neighbour at displacement d of coordinate s among its three neighbours on a periodic
// lattice of n >= 3 points (0 / 1 / 2) — what decides where a row with sorted columns stores a displacement
__device__ __forceinline__ int line_rank(int s, int d, int n) {
    // lower face: the neighbours are n-1, 0, 1 -> ranks 2, 0, 1;  upper face: n-2, n-1, 0 -> 1, 2, 0;  inside: 0, 1, 2
    return s == 0 ? (d < 0 ? 2 : d) : (s == n - 1 ? (d > 0 ? 0 : d + 2) : d + 1);
}

// The step loop of all three kernels is unrolled six times: which of the three register sets holds the target that is completed /
// continued / started in a step (period 3) and which of the two plane buffers is read (period 2) are then compile-time — no
// register rotation (sixteen v_mov per step that each waited for the dot pipeline: an accumulator written by v_dot2c needs wait
// states before an ordinary VALU instruction reads it) and no address toggling.  Measured: forward 190 -> 165 us, SDDMM 198 -> 170 us,
// transposed product 215 -> 192 us at C5 together with the stored-position walk and the per-line-length instantiation.
#define TSGU_LINE_SIX_STEPS(run, last)                                                   \
    for (int j_ = 0;;) {                                                                 \
        run(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, j_);     \
        if (++j_ > (last)) break;                                                        \
        run(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, j_);     \
        if (++j_ > (last)) break;                                                        \
        run(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, j_);     \
        if (++j_ > (last)) break;                                                        \
        run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, j_);     \
        if (++j_ > (last)) break;                                                        \
        run(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, j_);     \
        if (++j_ > (last)) break;                                                        \
        run(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, j_);     \
        if (++j_ > (last)) break;                                                        \
    }

// Aᵀ·G.  NT = TY·NZ·2 threads: lane pair (c = 0, 1: columns 8c .. 8c+7) per row of the tile.
template <int NT, int NZ>
__global__ __launch_bounds__(NT, 4) void linemarch_spmmt_kernel(const LineParams P) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    constexpr int TY = NT / (2 * NZ), G_BYTES = (TY + 2) * NZ * kLineRowB, REGION = G_BYTES + (TY + 2) * NZ * kLineValB;
    extern __shared__ uint4 line_smem[];
    char* const sm = reinterpret_cast<char*>(line_smem);
    const unsigned sbase = lat_lds_addr(line_smem);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int c = tid & 1, r = tid >> 1;
    const int ly = r / NZ, z = r - ly * NZ;
    auto wrapn = [](int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); };

    int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    const int tyi = (int)(vb % P.tiles_y);
    vb /= P.tiles_y;
    const int seg = (int)(vb % P.nseg);
    const int item = (int)(vb / P.nseg);
    const int x0 = seg * P.seg_len;
    const int L = P.seg_len < P.nx - x0 ? P.seg_len : P.nx - x0;
    const int y0 = tyi * TY, y = y0 + ly;
    const int plane_rows = P.ny * NZ;
    const int64_t item_row0 = (int64_t)item * P.nx * plane_rows;

    // ---- per-lane constants of the march: LDS byte addresses in buffer 0 of the dense row and of the x-triple of each tap ----------
    unsigned gaddr[9], vaddr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int ty_ = k / 3 - 1, tz_ = k % 3 - 1;
        const int ys = wrapn(y + ty_, P.ny), zs = wrapn(z + tz_, NZ);
        const int hrow = (ly + 1 + ty_) * NZ + zs;
        // the source row at (ys, zs) holds the entry towards (y, z) = its displacement (-ty_, -tz_)
        const int q = 3 * line_rank(ys, -ty_, P.ny) + line_rank(zs, -tz_, NZ);
        gaddr[k] = (unsigned)(hrow * kLineRowB + c * 16);
        vaddr[k] = sbase + (unsigned)(G_BYTES + hrow * kLineValB + 2 * q);
    }
    // DMA pieces: piece i of a plane lands at byte 16·i of its region; dense rows [(TY + 2)·NZ][2], value lines [(TY + 2)][NZ·54 / 16]
    constexpr int NG = (TY + 2) * NZ * 2, vpl = NZ * kLineValB / 16, NV = (TY + 2) * vpl;
    uint32_t goff[kLineKG], voff[kLineKV];
#pragma unroll
    for (int k = 0; k < kLineKG; ++k) {
        const int i = tid + k * NT;
        const int hl = i / (2 * NZ), rem = i - hl * (2 * NZ);
        const int yl = wrapn(y0 - 1 + hl, P.ny);
        goff[k] = i < NG ? (uint32_t)(((int64_t)(yl * NZ + (rem >> 1)) * P.lds_) * 2 + (rem & 1) * 16) : kLatNone;
    }
#pragma unroll
    for (int k = 0; k < kLineKV; ++k) {
        const int i = tid + k * NT;
        const int hl = i / vpl, w = i - hl * vpl;
        const int yl = wrapn(y0 - 1 + hl, P.ny);
        voff[k] = i < NV ? (uint32_t)(yl * NZ * kLineValB + w * 16) : kLatNone;
    }
    auto dma_plane = [&](int xsrc, int buf) {
        const int64_t prow = item_row0 + (int64_t)xsrc * plane_rows;
        const char* const gsrc = static_cast<const char*>(P.S) + prow * P.lds_ * 2;
        const char* const vsrc = static_cast<const char*>(P.val) + prow * kLineValB;
        const unsigned dst = sbase + (unsigned)buf * (unsigned)REGION + (unsigned)wave * (kWave * 16);
#pragma unroll
        for (int k = 0; k < kLineKG; ++k)
            if (goff[k] != kLatNone) lat_dma16<false>(gsrc, goff[k], __builtin_amdgcn_readfirstlane(dst + k * (NT * 16)));
#pragma unroll
        for (int k = 0; k < kLineKV; ++k)
            if (voff[k] != kLatNone) lat_dma16<false>(vsrc, voff[k], __builtin_amdgcn_readfirstlane(dst + G_BYTES + k * (NT * 16)));
    };
    const uint32_t ooff = (uint32_t)(((int64_t)r * P.ldo + c * 8) * 2);     // this lane's 16 bytes inside a tile plane of the result

    float acc[3][8];     // targets x-1 (completed in this step), x, x+1 (started in this step) of source plane x: sets RA, RB, RC of a step
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = acc[2][i] = 0.f;

    // ---- prologue: source plane x0 - 1 -> buffer 0 --------------------------------------------------------------------------
    dma_plane(wrapn(x0 - 1, P.nx), 0);
    lat_step_sync();

    // step j: source plane xs = x0 - 1 + j; its dx = +1 part starts target x0 + j (set RC), its dx = -1 part completes x0 + j - 2 (set RA).
    auto run = [&](auto rc, auto bc, int j) {
        constexpr int RA = decltype(rc)::value, RB = (RA + 1) % 3, RC = (RA + 2) % 3, BUF = decltype(bc)::value;
        const int xs = wrapn(x0 - 1 + j, P.nx);
        if (j <= L) dma_plane(wrapn(xs + 1, P.nx), BUF ^ 1);
        // which position of a source row's x-triple belongs to the target of set RA / RB / RC: 0 / 1 / 2 in interior planes (compile-time
        // offsets: the fast body); rotated at the two x faces, where the offsets are run-time (one add per value read, two planes of nx)
        const int tA = xs == 0 ? 2 : (xs == P.nx - 1 ? 1 : 0), tB = xs == 0 ? 0 : (xs == P.nx - 1 ? 2 : 1), tC = xs == 0 ? 1 : (xs == P.nx - 1 ? 0 : 2);
        auto body = [&](auto at_face) {
        constexpr bool FACE = decltype(at_face)::value;
        auto val16 = [&](int k, int t) -> uint32_t {
            if constexpr (FACE) return line_lds_u16(vaddr[k] + 18u * (unsigned)(t == 0 ? tA : (t == 1 ? tB : tC)), BUF * (REGION / 2));
            else return line_lds_u16(vaddr[k], 9 * t + BUF * (REGION / 2));
        };
        // (the target started in this step takes its first sums with a zero addend — interior planes, where it is the set RC at
        // compile time; at a face any of the three positions may be the new target's: its set is cleared first)
        if constexpr (FACE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[RC][i] = 0.f;
        }
        auto dots = [&](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t lo, uint32_t hi, int i, bool first) {
            const bf2 l = __builtin_bit_cast(bf2, lo), h = __builtin_bit_cast(bf2, hi);
            const bf2 v0 = __builtin_bit_cast(bf2, a0), v1 = __builtin_bit_cast(bf2, a1), v2 = __builtin_bit_cast(bf2, a2);
            acc[RA][2 * i] = __builtin_amdgcn_fdot2_f32_bf16(v0, l, acc[RA][2 * i], false);
            acc[RA][2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(v0, h, acc[RA][2 * i + 1], false);
            acc[RB][2 * i] = __builtin_amdgcn_fdot2_f32_bf16(v1, l, acc[RB][2 * i], false);
            acc[RB][2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(v1, h, acc[RB][2 * i + 1], false);
            acc[RC][2 * i] = __builtin_amdgcn_fdot2_f32_bf16(v2, l, first && !FACE ? 0.f : acc[RC][2 * i], false);
            acc[RC][2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(v2, h, first && !FACE ? 0.f : acc[RC][2 * i + 1], false);
        };
        // tap pairs one after the other (see the forward kernel for the empty asm)
        uint4 xr[2], yr[2];
        uint32_t a[2][3];
        auto fetch = [&](int pp) {
            const int k0 = 2 * pp, k1 = 2 * pp + 1;
            xr[pp & 1] = *reinterpret_cast<const uint4*>(sm + gaddr[k0] + BUF * REGION);
#pragma unroll
            for (int t = 0; t < 3; ++t) a[pp & 1][t] = val16(k0, t);
            if (pp < 4) {
                yr[pp & 1] = *reinterpret_cast<const uint4*>(sm + gaddr[k1] + BUF * REGION);
#pragma unroll
                for (int t = 0; t < 3; ++t) a[pp & 1][t] |= val16(k1, t) << 16;
            }
        };
        fetch(0);
#pragma unroll
        for (int pp = 0; pp < 5; ++pp) {
            if (pp < 4) fetch(pp + 1);
            const uint32_t xw[4] = {xr[pp & 1].x, xr[pp & 1].y, xr[pp & 1].z, xr[pp & 1].w};
            const uint32_t yw[4] = {yr[pp & 1].x, yr[pp & 1].y, yr[pp & 1].z, yr[pp & 1].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // pairs: (x[2i], y[2i]) and (x[2i+1], y[2i+1]);  the ninth tap alone: (value, 0) · (element, 0)
                const uint32_t lo = pp < 4 ? __builtin_amdgcn_perm(yw[i], xw[i], 0x05040100u) : (xw[i] & 0xffffu);
                const uint32_t hi = pp < 4 ? __builtin_amdgcn_perm(yw[i], xw[i], 0x07060302u) : (xw[i] >> 16);
                dots(a[pp & 1][0], a[pp & 1][1], a[pp & 1][2], lo, hi, i, pp == 0);
            }
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[0][4]), "+v"(acc[0][5]), "+v"(acc[0][6]), "+v"(acc[0][7]),
                              "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[1][4]), "+v"(acc[1][5]), "+v"(acc[1][6]), "+v"(acc[1][7]),
                              "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]), "+v"(acc[2][4]), "+v"(acc[2][5]), "+v"(acc[2][6]), "+v"(acc[2][7])
                         :: "memory");
        }
        };
        if (xs == 0 || xs == P.nx - 1) body(std::true_type{});
        else body(std::false_type{});
        if (j >= 2) {
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = line_cvt2(acc[RA][2 * i], acc[RA][2 * i + 1]);
            char* const tile = static_cast<char*>(P.out) + (item_row0 + (int64_t)(x0 + j - 2) * plane_rows + y0 * NZ) * P.ldo * 2;   // wave-uniform
            stream_store16(tile + ooff, make_uint4(o[0], o[1], o[2], o[3]));
        }
        lat_step_sync();
    };
    TSGU_LINE_SIX_STEPS(run, L + 1)
}

// C = A·B.  The planes of B march through LDS (two buffers); the raw value lines of the tile's OWN rows sit in a ring of four
// planes (the three live targets + the one being filled).  The stored position of (dx, dy, dz) in the own row is 9·rank_x +
// (3·rank_y + rank_z): the lane walks the nine STORED POSITIONS of an x-part in order (value at byte 2·s of the part: an immediate
// offset) and keeps, per position, the address of the dense row it multiplies (per-lane constants; face rows walk their taps in
// another order than interior rows).  Ring slot and 9·rank_x are wave-uniform per step and target: three adds per step.
template <int NT, int NZ>
__global__ __launch_bounds__(NT, 4) void linemarch_spmm_kernel(const LineParams P) {
    constexpr int TY = NT / (2 * NZ), G_BYTES = (TY + 2) * NZ * kLineRowB, STAGE = TY * NZ * kLineValB, O_STAGE = 2 * G_BYTES;
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    extern __shared__ uint4 line_smem[];
    char* const sm = reinterpret_cast<char*>(line_smem);
    const unsigned sbase = lat_lds_addr(line_smem);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int c = tid & 1, r = tid >> 1;
    const int ly = r / NZ, z = r - ly * NZ;
    auto wrapn = [](int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); };

    int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    const int tyi = (int)(vb % P.tiles_y);
    vb /= P.tiles_y;
    const int seg = (int)(vb % P.nseg);
    const int item = (int)(vb / P.nseg);
    const int x0 = seg * P.seg_len;
    const int L = P.seg_len < P.nx - x0 ? P.seg_len : P.nx - x0;
    const int y0 = tyi * TY, y = y0 + ly;
    const int plane_rows = P.ny * NZ;
    const int64_t item_row0 = (int64_t)item * P.nx * plane_rows;

    // dense-row address (buffer 0 / 1) of the tap stored at position s of an x-part of this lane's row
    unsigned g0[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = k / 3 - 1, dz = k % 3 - 1;
        const unsigned a = (unsigned)(((ly + 1 + dy) * NZ + wrapn(z + dz, NZ)) * kLineRowB + c * 16);
        const int s = 3 * line_rank(y, dy, P.ny) + line_rank(z, dz, NZ);
#pragma unroll
        for (int q = 0; q < 9; ++q)
            if (q == s) g0[q] = a;
    }
    const unsigned vrow = (unsigned)(O_STAGE + r * kLineValB);
    const uint32_t ooff = (uint32_t)(((int64_t)r * P.ldo + c * 8) * 2);     // this lane's 16 bytes inside a tile plane of the result
    constexpr int NG = (TY + 2) * NZ * 2;
    constexpr int NV = TY * (NZ * kLineValB / 16);      // the tile's own value lines: one contiguous run of the plane
    uint32_t goff[kLineKG], voff[kLineKVOwn];
#pragma unroll
    for (int k = 0; k < kLineKG; ++k) {
        const int i = tid + k * NT;
        const int hl = i / (2 * NZ), rem = i - hl * (2 * NZ);
        const int yl = wrapn(y0 - 1 + hl, P.ny);
        goff[k] = i < NG ? (uint32_t)(((int64_t)(yl * NZ + (rem >> 1)) * P.lds_) * 2 + (rem & 1) * 16) : kLatNone;
    }
#pragma unroll
    for (int k = 0; k < kLineKVOwn; ++k) {
        const int i = tid + k * NT;
        voff[k] = i < NV ? (uint32_t)(y0 * NZ * kLineValB + i * 16) : kLatNone;
    }
    // buffer `buf` <- plane xg of B;  value slot `slot` <- the own lines of plane xv
    auto dma_plane = [&](int xg, int buf, int xv, int slot) {
        const char* const gsrc = static_cast<const char*>(P.S) + (item_row0 + (int64_t)xg * plane_rows) * P.lds_ * 2;
        const char* const vsrc = static_cast<const char*>(P.val) + (item_row0 + (int64_t)xv * plane_rows) * kLineValB;
        const unsigned lane0 = sbase + (unsigned)wave * (kWave * 16);
#pragma unroll
        for (int k = 0; k < kLineKG; ++k)
            if (goff[k] != kLatNone) lat_dma16<false>(gsrc, goff[k], __builtin_amdgcn_readfirstlane(lane0 + (unsigned)buf * (unsigned)G_BYTES + k * (NT * 16)));
#pragma unroll
        for (int k = 0; k < kLineKVOwn; ++k)
            if (voff[k] != kLatNone) lat_dma16<true>(vsrc, voff[k], __builtin_amdgcn_readfirstlane(lane0 + (unsigned)(O_STAGE + slot * STAGE) + k * (NT * 16)));
    };

    float acc[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = acc[2][i] = 0.f;

    dma_plane(wrapn(x0 - 1, P.nx), 0, x0, 0);
    lat_step_sync();

    // step j: B's plane xb = x0 - 1 + j; targets j - 2 (dx = +1, completed: set RC), j - 1 (dx = 0), j (dx = -1, started)
    auto run = [&](auto rc, auto bc, int j) {
        constexpr int RA = decltype(rc)::value, RB = (RA + 1) % 3, RC = (RA + 2) % 3, BUF = decltype(bc)::value;
        const int xb = wrapn(x0 - 1 + j, P.nx);
        if (j <= L) dma_plane(wrapn(xb + 1, P.nx), BUF ^ 1, wrapn(xb + 2, P.nx), (j + 1) & 3);
        const unsigned vA = vrow + (unsigned)(((j + 2) & 3) * STAGE + 18 * line_rank(wrapn(xb - 1, P.nx), 1, P.nx));
        const unsigned vB = vrow + (unsigned)(((j + 3) & 3) * STAGE + 18 * line_rank(xb, 0, P.nx));
        const unsigned vC = vrow + (unsigned)((j & 3) * STAGE + 18 * line_rank(wrapn(xb + 1, P.nx), -1, P.nx));
        auto val16 = [&](unsigned v, int s) -> uint32_t { return line_lds_u16(sbase + v, s); };
        auto dots = [&](uint32_t aA, uint32_t aB, uint32_t aC, uint32_t lo, uint32_t hi, int i, bool first) {
            const bf2 l = __builtin_bit_cast(bf2, lo), h = __builtin_bit_cast(bf2, hi);
            const bf2 wA = __builtin_bit_cast(bf2, aA), wB = __builtin_bit_cast(bf2, aB), wC = __builtin_bit_cast(bf2, aC);
            acc[RA][2 * i] = __builtin_amdgcn_fdot2_f32_bf16(wA, l, acc[RA][2 * i], false);
            acc[RA][2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(wA, h, acc[RA][2 * i + 1], false);
            acc[RB][2 * i] = __builtin_amdgcn_fdot2_f32_bf16(wB, l, acc[RB][2 * i], false);
            acc[RB][2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(wB, h, acc[RB][2 * i + 1], false);
            acc[RC][2 * i] = __builtin_amdgcn_fdot2_f32_bf16(wC, l, first ? 0.f : acc[RC][2 * i], false);
            acc[RC][2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(wC, h, first ? 0.f : acc[RC][2 * i + 1], false);
        };
        // Position pairs one after the other, the raw reads of pair pp + 1 requested before pair pp is consumed.  The empty asm ties
        // the 24 accumulators (and memory) to the end of a pair: left alone, hipcc requests all nine dense rows, builds all 40 operand
        // pairs and only then starts on the dots — 40 live registers more than the 128 that two workgroups per CU leave a wave.
        uint4 xr[2], yr[2];
        uint32_t aA[2], aB[2], aC[2];
        auto fetch = [&](int pp) {
            const int s0 = 2 * pp, s1 = 2 * pp + 1;
            xr[pp & 1] = *reinterpret_cast<const uint4*>(sm + g0[s0] + BUF * G_BYTES);
            aA[pp & 1] = val16(vA, s0), aB[pp & 1] = val16(vB, s0), aC[pp & 1] = val16(vC, s0);
            if (pp < 4) {
                yr[pp & 1] = *reinterpret_cast<const uint4*>(sm + g0[s1] + BUF * G_BYTES);
                aA[pp & 1] |= val16(vA, s1) << 16, aB[pp & 1] |= val16(vB, s1) << 16, aC[pp & 1] |= val16(vC, s1) << 16;
            }
        };
        fetch(0);
#pragma unroll
        for (int pp = 0; pp < 5; ++pp) {
            if (pp < 4) fetch(pp + 1);
            const uint32_t xw[4] = {xr[pp & 1].x, xr[pp & 1].y, xr[pp & 1].z, xr[pp & 1].w};
            const uint32_t yw[4] = {yr[pp & 1].x, yr[pp & 1].y, yr[pp & 1].z, yr[pp & 1].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // pairs: (x[2i], y[2i]) and (x[2i+1], y[2i+1]);  the ninth position alone: (value, 0) · (element, 0)
                const uint32_t lo = pp < 4 ? __builtin_amdgcn_perm(yw[i], xw[i], 0x05040100u) : (xw[i] & 0xffffu);
                const uint32_t hi = pp < 4 ? __builtin_amdgcn_perm(yw[i], xw[i], 0x07060302u) : (xw[i] >> 16);
                dots(aA[pp & 1], aB[pp & 1], aC[pp & 1], lo, hi, i, pp == 0);
            }
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[0][4]), "+v"(acc[0][5]), "+v"(acc[0][6]), "+v"(acc[0][7]),
                              "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[1][4]), "+v"(acc[1][5]), "+v"(acc[1][6]), "+v"(acc[1][7]),
                              "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]), "+v"(acc[2][4]), "+v"(acc[2][5]), "+v"(acc[2][6]), "+v"(acc[2][7])
                         :: "memory");
        }
        if (j >= 2) {
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = line_cvt2(acc[RA][2 * i], acc[RA][2 * i + 1]);
            char* const tile = static_cast<char*>(P.out) + (item_row0 + (int64_t)(x0 + j - 2) * plane_rows + y0 * NZ) * P.ldo * 2;   // wave-uniform
            stream_store16(tile + ooff, make_uint4(o[0], o[1], o[2], o[3]));
        }
        lat_step_sync();
    };
    TSGU_LINE_SIX_STEPS(run, L + 1)
}

// gradA = alpha · <R[row], Cm[col]> in A's stored order.  The gathered operand's planes march through LDS as above; a lane pair
// owns a row, its own rows R of the three live targets sit in registers (both halves: a lane takes every second tap with all 16
// columns — no cross-lane sum — and both lanes the ninth), a dot is eight v_dot2_f32_bf16 on the packed pairs.  The stored
// position of (dx, dy, dz) in the OWN row: 9·rank_x (wave-uniform per step and target) + 3·rank_y + rank_z (per-lane constants).
// Results are staged as bf16 in the row's 54-byte slot of its target plane's stage (three planes: a target lives for three
// steps); the rows of a tile plane are consecutive in memory (whole z-lines), so a wave's 32 rows are 1728 contiguous bytes that
// start on a 16-byte boundary: the wave itself flushes what it staged for a finished target as aligned 16-byte pieces — no barrier
// between the last write and the flush, gradA leaves fully coalesced.
template <int NT, int NZ>
__global__ __launch_bounds__(NT, 4) void linemarch_sddmm_kernel(const LineParams P) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    constexpr int TY = NT / (2 * NZ), G_BYTES = (TY + 2) * NZ * kLineRowB, REGION = G_BYTES + TY * NZ * kLineRowB, O_STAGE = 2 * REGION,
                  STAGE = TY * NZ * kLineValB;
    extern __shared__ uint4 line_smem[];
    char* const sm = reinterpret_cast<char*>(line_smem);
    const unsigned sbase = lat_lds_addr(line_smem);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int lane = tid % kWave;
    const int c = tid & 1, r = tid >> 1;
    const int ly = r / NZ, z = r - ly * NZ;
    auto wrapn = [](int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); };

    int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    const int tyi = (int)(vb % P.tiles_y);
    vb /= P.tiles_y;
    const int seg = (int)(vb % P.nseg);
    const int item = (int)(vb / P.nseg);
    const int x0 = seg * P.seg_len;
    const int L = P.seg_len < P.nx - x0 ? P.seg_len : P.nx - x0;
    const int y0 = tyi * TY, y = y0 + ly;
    const int plane_rows = P.ny * NZ;
    const int64_t item_row0 = (int64_t)item * P.nx * plane_rows;

    // this lane's stored positions of an x-part: c, c + 2, c + 4, c + 6 and (both lanes) 8; per position the dense row of its tap
    unsigned bS[5];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = k / 3 - 1, dz = k % 3 - 1;
        const unsigned a = (unsigned)(((ly + 1 + dy) * NZ + wrapn(z + dz, NZ)) * kLineRowB);
        const int s = 3 * line_rank(y, dy, P.ny) + line_rank(z, dz, NZ);
#pragma unroll
        for (int t = 0; t < 5; ++t)
            if (s == (t < 4 ? c + 2 * t : 8)) bS[t] = a;
    }
    const unsigned srow = (unsigned)(O_STAGE + r * kLineValB + 2 * c);      // (+ 4·t: position c + 2·t;  + 16 - 2·c: position 8)
    const unsigned srow8 = (unsigned)(O_STAGE + r * kLineValB + 16);
    const unsigned own_addr = (unsigned)(G_BYTES + r * kLineRowB);
    constexpr int NG = (TY + 2) * NZ * 2;
    uint32_t goff[kLineKG];
#pragma unroll
    for (int k = 0; k < kLineKG; ++k) {
        const int i = tid + k * NT;
        const int hl = i / (2 * NZ), rem = i - hl * (2 * NZ);
        const int yl = wrapn(y0 - 1 + hl, P.ny);
        goff[k] = i < NG ? (uint32_t)(((int64_t)(yl * NZ + (rem >> 1)) * P.lds_) * 2 + (rem & 1) * 16) : kLatNone;
    }
    const uint32_t ooff = (uint32_t)(((int64_t)(y0 * NZ + r) * P.ldown) * 2 + c * 16);
    // buffer `buf`: the gathered operand's plane xg and the row operand's plane xo
    auto dma_plane = [&](int xg, int xo, int buf) {
        const char* const gsrc = static_cast<const char*>(P.S) + (item_row0 + (int64_t)xg * plane_rows) * P.lds_ * 2;
        const char* const osrc = static_cast<const char*>(P.Own) + (item_row0 + (int64_t)xo * plane_rows) * P.ldown * 2;
        const unsigned dst = sbase + (unsigned)buf * (unsigned)REGION + (unsigned)wave * (kWave * 16);
#pragma unroll
        for (int k = 0; k < kLineKG; ++k)
            if (goff[k] != kLatNone) lat_dma16<false>(gsrc, goff[k], __builtin_amdgcn_readfirstlane(dst + k * (NT * 16)));
        lat_dma16<true>(osrc, ooff, __builtin_amdgcn_readfirstlane(dst + G_BYTES));
    };

    uint32_t own[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) own[0][i] = own[1][i] = own[2][i] = 0u;

    dma_plane(wrapn(x0 - 1, P.nx), x0, 0);
    lat_step_sync();

    constexpr int pieces = kWave / 2 * kLineValB / 16;       // 16-byte pieces of a wave's 32 rows: 108
    // step j: the gathered plane xb = x0 - 1 + j; targets j - 2 (dx = +1, completed: own rows RA), j - 1 (dx = 0), j (dx = -1, new: RC)
    auto run = [&](auto rc, auto bc, int j) {
        constexpr int RA = decltype(rc)::value, RB = (RA + 1) % 3, RC = (RA + 2) % 3, BUF = decltype(bc)::value;
        const int xb = wrapn(x0 - 1 + j, P.nx);
        if (j <= L) dma_plane(wrapn(xb + 1, P.nx), wrapn(xb + 2, P.nx), BUF ^ 1);
        {   // the new target's own row
            const uint4 lo = *reinterpret_cast<const uint4*>(sm + own_addr + BUF * REGION), hi = *reinterpret_cast<const uint4*>(sm + own_addr + BUF * REGION + 16);
            own[RC][0] = lo.x, own[RC][1] = lo.y, own[RC][2] = lo.z, own[RC][3] = lo.w, own[RC][4] = hi.x, own[RC][5] = hi.y, own[RC][6] = hi.z, own[RC][7] = hi.w;
        }
        // stage plane and x-part of the three targets
        const int slotA = (j + 1) % 3, slotB = (j + 2) % 3, slotC = j % 3;
        const unsigned uA = (unsigned)(slotA * STAGE + 18 * line_rank(wrapn(xb - 1, P.nx), 1, P.nx));
        const unsigned uB = (unsigned)(slotB * STAGE + 18 * line_rank(xb, 0, P.nx));
        const unsigned uC = (unsigned)(slotC * STAGE + 18 * line_rank(wrapn(xb + 1, P.nx), -1, P.nx));
        const unsigned wA = srow + uA, wB = srow + uB, wC = srow + uC, wA8 = srow8 + uA, wB8 = srow8 + uB, wC8 = srow8 + uC;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const uint4 lo = *reinterpret_cast<const uint4*>(sm + bS[t] + BUF * REGION), hi = *reinterpret_cast<const uint4*>(sm + bS[t] + BUF * REGION + 16);
            const uint32_t b[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            float dA = 0.f, dB = 0.f, dC = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf2 bv = __builtin_bit_cast(bf2, b[i]);
                dA = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, own[RA][i]), bv, dA, false);
                dB = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, own[RB][i]), bv, dB, false);
                dC = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, own[RC][i]), bv, dC, false);
            }
            dA *= P.alpha, dB *= P.alpha, dC *= P.alpha;
            const uint32_t pab = line_cvt2(dA, dB), pc = line_cvt2(dC, dC);
            *reinterpret_cast<unsigned short*>(sm + (t < 4 ? wA + 4 * t : wA8)) = (unsigned short)pab;
            *reinterpret_cast<unsigned short*>(sm + (t < 4 ? wB + 4 * t : wB8)) = (unsigned short)(pab >> 16);
            *reinterpret_cast<unsigned short*>(sm + (t < 4 ? wC + 4 * t : wC8)) = (unsigned short)pc;
        }
        if (j >= 2) {   // target j - 2 is complete: this wave's 32 rows (written by this wave only) leave as 16-byte pieces
            const int64_t row0 = item_row0 + (int64_t)(x0 + j - 2) * plane_rows + y0 * NZ + wave * (kWave / 2);
            char* const dst = static_cast<char*>(P.gvals) + row0 * kLineValB;
            const char* const src = sm + O_STAGE + slotA * STAGE + wave * (kWave / 2 * kLineValB);
            stream_store16(dst + lane * 16, *reinterpret_cast<const uint4*>(src + lane * 16));
            if (lane + kWave < pieces) stream_store16(dst + (lane + kWave) * 16, *reinterpret_cast<const uint4*>(src + (lane + kWave) * 16));
        }
        lat_step_sync();
    };
    TSGU_LINE_SIX_STEPS(run, L + 1)
}

inline int linemarch_layout(LineParams& P, int threads, int mode) {
    if ((P.nz != 8 && P.nz != 16 && P.nz != 32 && P.nz != 64) || P.ty <= 0 || P.ty * P.nz * 2 != threads) return TSGU_ERR_BAD_ARG;   // (the kernels are compiled per line length)
    if (threads != 256 && threads != 512 && threads != 1024) return TSGU_ERR_BAD_ARG;
    if (P.ny % P.ty) return TSGU_ERR_BAD_ARG;
    const int hl = P.ty + 2;
    P.g_bytes = hl * P.nz * kLineRowB;
    if (hl * P.nz * 2 > kLineKG * threads) return TSGU_ERR_TOO_LARGE;
    // (the kernels derive the same layout from their template arguments; these fields are what the host reports)
    if (mode == kLatSpmmT) {            // two buffers of (dense rows + raw value lines) of the ty + 2 halo lines
        if (hl * (P.nz * kLineValB / 16) > kLineKV * threads) return TSGU_ERR_TOO_LARGE;
        P.buf_stride = P.g_bytes + hl * P.nz * kLineValB;
        P.o_stage = 0, P.stage_bytes = 0;
        P.lds_bytes = 2 * P.buf_stride;
    } else if (mode == kLatSddmm) {     // two buffers of (gathered halo rows + own rows of the row operand), three stage planes
        P.buf_stride = P.g_bytes + P.ty * P.nz * kLineRowB;
        P.o_stage = 2 * P.buf_stride;
        P.stage_bytes = P.ty * P.nz * kLineValB;
        P.lds_bytes = P.o_stage + 3 * P.stage_bytes;
    } else if (mode == kLatSpmm) {      // two buffers of B's halo rows, a ring of four planes of the tile's own value lines
        if (P.ty * (P.nz * kLineValB / 16) > kLineKVOwn * threads) return TSGU_ERR_TOO_LARGE;
        P.buf_stride = P.g_bytes;
        P.o_stage = 2 * P.g_bytes;
        P.stage_bytes = P.ty * P.nz * kLineValB;
        P.lds_bytes = P.o_stage + 4 * P.stage_bytes;
    } else {
        return TSGU_ERR_BAD_ARG;
    }
    if (P.lds_bytes > kLatMaxLds) return TSGU_ERR_TOO_LARGE;
    return P.lds_bytes;
}

template <int NT, int NZ, int MODE>
int linemarch_launch_t(const LineParams& P, hipStream_t stream) {
    static std::atomic<uint64_t> allowed{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TSGU_ERR_RUNTIME;
    void (*kernel)(const LineParams);
    if constexpr (MODE == kLatSpmm) kernel = &linemarch_spmm_kernel<NT, NZ>;
    else if constexpr (MODE == kLatSddmm) kernel = &linemarch_sddmm_kernel<NT, NZ>;
    else kernel = &linemarch_spmmt_kernel<NT, NZ>;
    if (!(allowed.load(std::memory_order_acquire) >> dev & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLatMaxLds) != hipSuccess)
            return TSGU_ERR_RUNTIME;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)P.nblocks), dim3(NT), (size_t)P.lds_bytes, stream, P);
    return check_launch();
}

}  // namespace tsgu
