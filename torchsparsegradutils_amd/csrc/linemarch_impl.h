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
// addresses; rank_x is the same for all rows of a source plane: the three values at +0 / +18 / +36 bytes belong to
// dx = -1 / 0 / +1 in interior planes and are rotated at the two faces (a wave-uniform branch between three copies of the step).
// A tile is TY whole z-lines (the z wrap is inside the line: no z halo, folded into the per-lane addresses as well), two
// planes of (TY + 2) lines are resident: 55 KB at TY = 8 — two workgroups of eight waves per CU, ALL of them computing.
//
// Sum per target row: source planes x-1, x, x+1 in that order (plane by plane as in march_impl.h), inside a plane the taps
// in ascending (dy, dz) by pairs — two entries per v_dot2_f32_bf16, fp32 accumulation, one rounding at the end: every element
// within one bf16 ulp of the exact result, like the sweep's (same tests), not bit-identical to it.
#pragma once

#include "lattice_impl.h"

namespace tsgu {

constexpr int kLineKG = 3;     // 16-byte DMA pieces per thread and plane: dense rows
constexpr int kLineKV = 5;     //                                          value rows
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

// Aᵀ·G.  NT = ty·nz·2 threads: lane pair (c = 0, 1: columns 8c .. 8c+7) per row of the tile.
template <int NT>
__global__ __launch_bounds__(NT, 4) void linemarch_spmmt_kernel(const LineParams P) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    extern __shared__ uint4 line_smem[];
    char* const sm = reinterpret_cast<char*>(line_smem);
    const unsigned sbase = lat_lds_addr(line_smem);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int c = tid & 1, r = tid >> 1;
    const int ly = r / P.nz, z = r - ly * P.nz;
    auto wrapn = [](int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); };

    int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    const int tyi = (int)(vb % P.tiles_y);
    vb /= P.tiles_y;
    const int seg = (int)(vb % P.nseg);
    const int item = (int)(vb / P.nseg);
    const int x0 = seg * P.seg_len;
    const int L = P.seg_len < P.nx - x0 ? P.seg_len : P.nx - x0;
    const int y0 = tyi * P.ty, y = y0 + ly;
    const int plane_rows = P.ny * P.nz;
    const int64_t item_row0 = (int64_t)item * P.nx * plane_rows;

    // ---- per-lane constants of the march ---------------------------------------------------------------------------------
    // rank of the neighbour at displacement d of coordinate s among the three neighbours (wrapped, lattice of n points)
    auto rank3 = [&](int s, int d, int n) {
        const int v = wrapn(s + d, n);
        return (wrapn(s - 1, n) < v) + (s < v) + (wrapn(s + 1, n) < v);
    };
    unsigned gaddr[9], vaddr[9];     // LDS byte addresses in the buffer of the CURRENT plane (toggled after every step)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int ty_ = k / 3 - 1, tz_ = k % 3 - 1;
        const int ys = wrapn(y + ty_, P.ny), zs = wrapn(z + tz_, P.nz);
        const int hrow = (ly + 1 + ty_) * P.nz + zs;
        // the source row at (ys, zs) holds the entry towards (y, z) = its displacement (-ty_, -tz_)
        const int q = 3 * rank3(ys, -ty_, P.ny) + rank3(zs, -tz_, P.nz);
        gaddr[k] = (unsigned)(hrow * kLineRowB + c * 16);
        vaddr[k] = (unsigned)(P.g_bytes + hrow * kLineValB + 2 * q);
    }
    // DMA pieces: piece i of a plane lands at byte 16·i of its region; dense rows [(ty + 2)·nz][2], value lines [(ty + 2)][nz·54 / 16]
    const int NG = (P.ty + 2) * P.nz * 2;
    const int vpl = P.nz * kLineValB / 16;
    const int NV = (P.ty + 2) * vpl;
    uint32_t goff[kLineKG], voff[kLineKV];
#pragma unroll
    for (int k = 0; k < kLineKG; ++k) {
        const int i = tid + k * NT;
        const int hl = i / (2 * P.nz), rem = i - hl * (2 * P.nz);
        const int yl = wrapn(y0 - 1 + hl, P.ny);
        goff[k] = i < NG ? (uint32_t)(((int64_t)(yl * P.nz + (rem >> 1)) * P.lds_) * 2 + (rem & 1) * 16) : kLatNone;
    }
#pragma unroll
    for (int k = 0; k < kLineKV; ++k) {
        const int i = tid + k * NT;
        const int hl = i / vpl, w = i - hl * vpl;
        const int yl = wrapn(y0 - 1 + hl, P.ny);
        voff[k] = i < NV ? (uint32_t)(yl * P.nz * kLineValB + w * 16) : kLatNone;
    }
    auto dma_plane = [&](int xsrc, int buf) {
        const int64_t prow = item_row0 + (int64_t)xsrc * plane_rows;
        const char* const gsrc = static_cast<const char*>(P.S) + prow * P.lds_ * 2;
        const char* const vsrc = static_cast<const char*>(P.val) + prow * kLineValB;
        const unsigned dst = sbase + (unsigned)buf * (unsigned)P.buf_stride + (unsigned)wave * (kWave * 16);
#pragma unroll
        for (int k = 0; k < kLineKG; ++k)
            if (goff[k] != kLatNone) lat_dma16<false>(gsrc, goff[k], dst + k * (NT * 16));
#pragma unroll
        for (int k = 0; k < kLineKV; ++k)
            if (voff[k] != kLatNone) lat_dma16<false>(vsrc, voff[k], dst + P.g_bytes + k * (NT * 16));
    };

    float accA[8], accB[8], accC[8];     // targets x-1 (complete after this step), x, x+1 (new in this step) of source plane x
#pragma unroll
    for (int i = 0; i < 8; ++i) accA[i] = accB[i] = accC[i] = 0.f;

    // p0 / p1 / p2: the accumulators that take the values at position 0 / 1 / 2 of a source row's x-triple; FRESH: which of them is new
    auto body = [&](float (&p0)[8], float (&p1)[8], float (&p2)[8], auto fresh) {
        constexpr int FRESH = decltype(fresh)::value;
        auto dots = [&](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t lo, uint32_t hi, int i, bool first) {
            const bf2 l = __builtin_bit_cast(bf2, lo), h = __builtin_bit_cast(bf2, hi);
            const bf2 v0 = __builtin_bit_cast(bf2, a0), v1 = __builtin_bit_cast(bf2, a1), v2 = __builtin_bit_cast(bf2, a2);
            p0[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(v0, l, first && FRESH == 0 ? 0.f : p0[2 * i], false);
            p0[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(v0, h, first && FRESH == 0 ? 0.f : p0[2 * i + 1], false);
            p1[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(v1, l, first && FRESH == 1 ? 0.f : p1[2 * i], false);
            p1[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(v1, h, first && FRESH == 1 ? 0.f : p1[2 * i + 1], false);
            p2[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(v2, l, first && FRESH == 2 ? 0.f : p2[2 * i], false);
            p2[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(v2, h, first && FRESH == 2 ? 0.f : p2[2 * i + 1], false);
        };
        auto val16 = [&](int k, int t) -> uint32_t { return *reinterpret_cast<const unsigned short*>(sm + vaddr[k] + 18 * t); };
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int k0 = 2 * pp, k1 = 2 * pp + 1;
            const uint4 xr = *reinterpret_cast<const uint4*>(sm + gaddr[k0]);
            const uint4 yr = *reinterpret_cast<const uint4*>(sm + gaddr[k1]);
            uint32_t a[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = val16(k0, t) | (val16(k1, t) << 16);
            const uint32_t xw[4] = {xr.x, xr.y, xr.z, xr.w}, yw[4] = {yr.x, yr.y, yr.z, yr.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t lo = __builtin_amdgcn_perm(yw[i], xw[i], 0x05040100u);   // (x[2i],   y[2i])
                const uint32_t hi = __builtin_amdgcn_perm(yw[i], xw[i], 0x07060302u);   // (x[2i+1], y[2i+1])
                dots(a[0], a[1], a[2], lo, hi, i, pp == 0);
            }
        }
        {   // the ninth tap alone: (value, 0) · (element, 0)
            const uint4 xr = *reinterpret_cast<const uint4*>(sm + gaddr[8]);
            const uint32_t xw[4] = {xr.x, xr.y, xr.z, xr.w};
            const uint32_t a0 = val16(8, 0), a1 = val16(8, 1), a2 = val16(8, 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) dots(a0, a1, a2, xw[i] & 0xffffu, xw[i] >> 16, i, false);
        }
    };

    // ---- prologue: source plane x0 - 1 -> buffer 0 --------------------------------------------------------------------------
    dma_plane(wrapn(x0 - 1, P.nx), 0);
    lat_step_sync();

    // step j: source plane x0 - 1 + j sits in buffer j & 1; its dx = +1 part starts target x0 + j, its dx = -1 part completes x0 + j - 2
    for (int j = 0; j <= L + 1; ++j) {
#pragma unroll
        for (int k = 0; k < kLineKG; ++k) lat_pin(goff[k]);
#pragma unroll
        for (int k = 0; k < kLineKV; ++k) lat_pin(voff[k]);
        const int xs = wrapn(x0 - 1 + j, P.nx);
        if (j <= L) dma_plane(wrapn(xs + 1, P.nx), (j + 1) & 1);
        if (xs == 0)
            body(accB, accC, accA, std::integral_constant<int, 1>{});        // stored x order at the lower face: dx = 0, +1, -1
        else if (xs == P.nx - 1)
            body(accC, accA, accB, std::integral_constant<int, 0>{});        //                  upper face: dx = +1, -1, 0
        else
            body(accA, accB, accC, std::integral_constant<int, 2>{});
        if (j >= 2) {
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = line_cvt2(accA[2 * i], accA[2 * i + 1]);
            const int64_t row = item_row0 + (int64_t)(x0 + j - 2) * plane_rows + y * P.nz + z;
            stream_store16(static_cast<char*>(P.out) + (row * P.ldo + c * 8) * 2, make_uint4(o[0], o[1], o[2], o[3]));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) accA[i] = accB[i], accB[i] = accC[i];
#pragma unroll
        for (int k = 0; k < 9; ++k) gaddr[k] ^= (unsigned)P.buf_stride, vaddr[k] ^= (unsigned)P.buf_stride;
        lat_step_sync();
    }
}

// C = A·B.  The planes of B march through LDS (two buffers); the raw value lines of the tile's OWN rows sit in a ring of four
// planes (the three live targets + the one being filled).  The stored position of (dx, dy, dz) in the own row: 3·rank_y + rank_z
// are per-lane constants, 9·rank_x and the ring slot are wave-uniform per step and target — one add per value read, no face variants.
template <int NT>
__global__ __launch_bounds__(NT, 4) void linemarch_spmm_kernel(const LineParams P) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    extern __shared__ uint4 line_smem[];
    char* const sm = reinterpret_cast<char*>(line_smem);
    const unsigned sbase = lat_lds_addr(line_smem);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int c = tid & 1, r = tid >> 1;
    const int ly = r / P.nz, z = r - ly * P.nz;
    auto wrapn = [](int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); };
    auto rank3 = [&](int s, int d, int n) {
        const int v = wrapn(s + d, n);
        return (wrapn(s - 1, n) < v) + (s < v) + (wrapn(s + 1, n) < v);
    };

    int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    const int tyi = (int)(vb % P.tiles_y);
    vb /= P.tiles_y;
    const int seg = (int)(vb % P.nseg);
    const int item = (int)(vb / P.nseg);
    const int x0 = seg * P.seg_len;
    const int L = P.seg_len < P.nx - x0 ? P.seg_len : P.nx - x0;
    const int y0 = tyi * P.ty, y = y0 + ly;
    const int plane_rows = P.ny * P.nz;
    const int64_t item_row0 = (int64_t)item * P.nx * plane_rows;

    unsigned gaddr[9], vq[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = k / 3 - 1, dz = k % 3 - 1;
        gaddr[k] = (unsigned)(((ly + 1 + dy) * P.nz + wrapn(z + dz, P.nz)) * kLineRowB + c * 16);
        vq[k] = (unsigned)(P.o_stage + r * kLineValB + 2 * (3 * rank3(y, dy, P.ny) + rank3(z, dz, P.nz)));
    }
    const int NG = (P.ty + 2) * P.nz * 2;
    const int NV = P.ty * (P.nz * kLineValB / 16);      // the tile's own value lines: one contiguous run of the plane
    uint32_t goff[kLineKG], voff[kLineKV];
#pragma unroll
    for (int k = 0; k < kLineKG; ++k) {
        const int i = tid + k * NT;
        const int hl = i / (2 * P.nz), rem = i - hl * (2 * P.nz);
        const int yl = wrapn(y0 - 1 + hl, P.ny);
        goff[k] = i < NG ? (uint32_t)(((int64_t)(yl * P.nz + (rem >> 1)) * P.lds_) * 2 + (rem & 1) * 16) : kLatNone;
    }
#pragma unroll
    for (int k = 0; k < kLineKV; ++k) {
        const int i = tid + k * NT;
        voff[k] = i < NV ? (uint32_t)(y0 * P.nz * kLineValB + i * 16) : kLatNone;
    }
    // buffer `buf` <- plane xg of B;  value slot `slot` <- the own lines of plane xv
    auto dma_plane = [&](int xg, int buf, int xv, int slot) {
        const char* const gsrc = static_cast<const char*>(P.S) + (item_row0 + (int64_t)xg * plane_rows) * P.lds_ * 2;
        const char* const vsrc = static_cast<const char*>(P.val) + (item_row0 + (int64_t)xv * plane_rows) * kLineValB;
        const unsigned lane0 = sbase + (unsigned)wave * (kWave * 16);
#pragma unroll
        for (int k = 0; k < kLineKG; ++k)
            if (goff[k] != kLatNone) lat_dma16<false>(gsrc, goff[k], lane0 + (unsigned)buf * (unsigned)P.g_bytes + k * (NT * 16));
#pragma unroll
        for (int k = 0; k < kLineKV; ++k)
            if (voff[k] != kLatNone) lat_dma16<true>(vsrc, voff[k], lane0 + (unsigned)(P.o_stage + slot * P.stage_bytes) + k * (NT * 16));
    };

    float accA[8], accB[8], accC[8];     // targets xb-1 (complete after this step), xb, xb+1 (new in this step) of B's plane xb
#pragma unroll
    for (int i = 0; i < 8; ++i) accA[i] = accB[i] = accC[i] = 0.f;

    dma_plane(wrapn(x0 - 1, P.nx), 0, x0, 0);
    lat_step_sync();

    unsigned bufo = 0;
    for (int j = 0; j <= L + 1; ++j) {
        const int xb = wrapn(x0 - 1 + j, P.nx);
        if (j <= L) dma_plane(wrapn(xb + 1, P.nx), (j + 1) & 1, wrapn(xb + 2, P.nx), (j + 1) & 3);
        // value ring slot and x-part of the three targets j - 2 (dx = +1), j - 1 (dx = 0), j (dx = -1)
        const unsigned uA = (unsigned)(((j + 2) & 3) * P.stage_bytes + 18 * rank3(wrapn(xb - 1, P.nx), 1, P.nx));
        const unsigned uB = (unsigned)(((j + 3) & 3) * P.stage_bytes + 18 * rank3(xb, 0, P.nx));
        const unsigned uC = (unsigned)((j & 3) * P.stage_bytes + 18 * rank3(wrapn(xb + 1, P.nx), -1, P.nx));
        auto val16 = [&](int k, unsigned u) -> uint32_t { return *reinterpret_cast<const unsigned short*>(sm + vq[k] + u); };
        auto dots = [&](uint32_t aA, uint32_t aB, uint32_t aC, uint32_t lo, uint32_t hi, int i, bool first) {
            const bf2 l = __builtin_bit_cast(bf2, lo), h = __builtin_bit_cast(bf2, hi);
            const bf2 vA = __builtin_bit_cast(bf2, aA), vB = __builtin_bit_cast(bf2, aB), vC = __builtin_bit_cast(bf2, aC);
            accA[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(vA, l, accA[2 * i], false);
            accA[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(vA, h, accA[2 * i + 1], false);
            accB[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(vB, l, accB[2 * i], false);
            accB[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(vB, h, accB[2 * i + 1], false);
            accC[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(vC, l, first ? 0.f : accC[2 * i], false);
            accC[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(vC, h, first ? 0.f : accC[2 * i + 1], false);
        };
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int k0 = 2 * pp, k1 = 2 * pp + 1;
            const uint4 xr = *reinterpret_cast<const uint4*>(sm + bufo + gaddr[k0]);
            const uint4 yr = *reinterpret_cast<const uint4*>(sm + bufo + gaddr[k1]);
            const uint32_t aA = val16(k0, uA) | (val16(k1, uA) << 16);
            const uint32_t aB = val16(k0, uB) | (val16(k1, uB) << 16);
            const uint32_t aC = val16(k0, uC) | (val16(k1, uC) << 16);
            const uint32_t xw[4] = {xr.x, xr.y, xr.z, xr.w}, yw[4] = {yr.x, yr.y, yr.z, yr.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t lo = __builtin_amdgcn_perm(yw[i], xw[i], 0x05040100u);
                const uint32_t hi = __builtin_amdgcn_perm(yw[i], xw[i], 0x07060302u);
                dots(aA, aB, aC, lo, hi, i, pp == 0);
            }
        }
        {
            const uint4 xr = *reinterpret_cast<const uint4*>(sm + bufo + gaddr[8]);
            const uint32_t xw[4] = {xr.x, xr.y, xr.z, xr.w};
            const uint32_t aA = val16(8, uA), aB = val16(8, uB), aC = val16(8, uC);
#pragma unroll
            for (int i = 0; i < 4; ++i) dots(aA, aB, aC, xw[i] & 0xffffu, xw[i] >> 16, i, false);
        }
        if (j >= 2) {
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = line_cvt2(accA[2 * i], accA[2 * i + 1]);
            const int64_t row = item_row0 + (int64_t)(x0 + j - 2) * plane_rows + y * P.nz + z;
            stream_store16(static_cast<char*>(P.out) + (row * P.ldo + c * 8) * 2, make_uint4(o[0], o[1], o[2], o[3]));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) accA[i] = accB[i], accB[i] = accC[i];
        bufo = (unsigned)P.g_bytes - bufo;
        lat_step_sync();
    }
}

// gradA = alpha · <R[row], Cm[col]> in A's stored order.  The gathered operand's planes march through LDS as above; a lane pair
// owns a row, its own rows R of the three live targets sit in registers (both halves: a lane takes every second tap with all 16
// columns — no cross-lane sum — and both lanes the ninth), a dot is eight v_dot2_f32_bf16 on the packed pairs.  The stored
// position of (dx, dy, dz) in the OWN row: 9·rank_x (wave-uniform per step and target) + 3·rank_y + rank_z (per-lane constants).
// Results are staged as bf16 in the row's 54-byte slot of its target plane's stage (three planes: a target lives for three
// steps); the rows of a tile plane are consecutive in memory (whole z-lines), so a wave's 32 rows are 1728 contiguous bytes that
// start on a 16-byte boundary: the wave itself flushes what it staged for a finished target as aligned 16-byte pieces — no barrier
// between the last write and the flush, gradA leaves fully coalesced.
template <int NT>
__global__ __launch_bounds__(NT, 4) void linemarch_sddmm_kernel(const LineParams P) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    extern __shared__ uint4 line_smem[];
    char* const sm = reinterpret_cast<char*>(line_smem);
    const unsigned sbase = lat_lds_addr(line_smem);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int lane = tid % kWave;
    const int c = tid & 1, r = tid >> 1;
    const int ly = r / P.nz, z = r - ly * P.nz;
    auto wrapn = [](int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); };
    auto rank3 = [&](int s, int d, int n) {
        const int v = wrapn(s + d, n);
        return (wrapn(s - 1, n) < v) + (s < v) + (wrapn(s + 1, n) < v);
    };

    int64_t vb = xcd_chunked_block(blockIdx.x, P.nblocks);
    const int tyi = (int)(vb % P.tiles_y);
    vb /= P.tiles_y;
    const int seg = (int)(vb % P.nseg);
    const int item = (int)(vb / P.nseg);
    const int x0 = seg * P.seg_len;
    const int L = P.seg_len < P.nx - x0 ? P.seg_len : P.nx - x0;
    const int y0 = tyi * P.ty, y = y0 + ly;
    const int plane_rows = P.ny * P.nz;
    const int64_t item_row0 = (int64_t)item * P.nx * plane_rows;

    // this lane's taps: c, c + 2, c + 4, c + 6 and 8
    unsigned baddr[5], saddr[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int k = t < 4 ? c + 2 * t : 8;
        const int dy = k / 3 - 1, dz = k % 3 - 1;
        baddr[t] = (unsigned)(((ly + 1 + dy) * P.nz + wrapn(z + dz, P.nz)) * kLineRowB);
        saddr[t] = (unsigned)(P.o_stage + r * kLineValB + 2 * (3 * rank3(y, dy, P.ny) + rank3(z, dz, P.nz)));
    }
    const unsigned own_addr = (unsigned)(P.g_bytes + r * kLineRowB);
    const int NG = (P.ty + 2) * P.nz * 2;
    uint32_t goff[kLineKG];
#pragma unroll
    for (int k = 0; k < kLineKG; ++k) {
        const int i = tid + k * NT;
        const int hl = i / (2 * P.nz), rem = i - hl * (2 * P.nz);
        const int yl = wrapn(y0 - 1 + hl, P.ny);
        goff[k] = i < NG ? (uint32_t)(((int64_t)(yl * P.nz + (rem >> 1)) * P.lds_) * 2 + (rem & 1) * 16) : kLatNone;
    }
    const uint32_t ooff = (uint32_t)(((int64_t)(y0 * P.nz + r) * P.ldown) * 2 + c * 16);
    // buffer `buf`: the gathered operand's plane xg and the row operand's plane xo
    auto dma_plane = [&](int xg, int xo, int buf) {
        const char* const gsrc = static_cast<const char*>(P.S) + (item_row0 + (int64_t)xg * plane_rows) * P.lds_ * 2;
        const char* const osrc = static_cast<const char*>(P.Own) + (item_row0 + (int64_t)xo * plane_rows) * P.ldown * 2;
        const unsigned dst = sbase + (unsigned)buf * (unsigned)P.buf_stride + (unsigned)wave * (kWave * 16);
#pragma unroll
        for (int k = 0; k < kLineKG; ++k)
            if (goff[k] != kLatNone) lat_dma16<false>(gsrc, goff[k], dst + k * (NT * 16));
        lat_dma16<true>(osrc, ooff, dst + P.g_bytes);
    };

    uint32_t ownA[8], ownB[8], ownC[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ownA[i] = ownB[i] = ownC[i] = 0u;

    dma_plane(wrapn(x0 - 1, P.nx), x0, 0);
    lat_step_sync();

    const int pieces = kWave / 2 * kLineValB / 16;       // 16-byte pieces of a wave's 32 rows: 108
    unsigned bufo = 0;
    for (int j = 0; j <= L + 1; ++j) {
        const int xb = wrapn(x0 - 1 + j, P.nx);
        if (j <= L) dma_plane(wrapn(xb + 1, P.nx), wrapn(xb + 2, P.nx), (j + 1) & 1);
        {   // the new target's own row
            const uint4 lo = *reinterpret_cast<const uint4*>(sm + bufo + own_addr), hi = *reinterpret_cast<const uint4*>(sm + bufo + own_addr + 16);
            ownC[0] = lo.x, ownC[1] = lo.y, ownC[2] = lo.z, ownC[3] = lo.w, ownC[4] = hi.x, ownC[5] = hi.y, ownC[6] = hi.z, ownC[7] = hi.w;
        }
        // stage plane and x-part of the three targets j - 2 (dx = +1), j - 1 (dx = 0), j (dx = -1)
        const int slotA = (j + 1) % 3, slotB = (j + 2) % 3, slotC = j % 3;
        const unsigned uA = (unsigned)(slotA * P.stage_bytes + 18 * rank3(wrapn(xb - 1, P.nx), 1, P.nx));
        const unsigned uB = (unsigned)(slotB * P.stage_bytes + 18 * rank3(xb, 0, P.nx));
        const unsigned uC = (unsigned)(slotC * P.stage_bytes + 18 * rank3(wrapn(xb + 1, P.nx), -1, P.nx));
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const uint4 lo = *reinterpret_cast<const uint4*>(sm + bufo + baddr[t]), hi = *reinterpret_cast<const uint4*>(sm + bufo + baddr[t] + 16);
            const uint32_t b[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            float dA = 0.f, dB = 0.f, dC = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf2 bv = __builtin_bit_cast(bf2, b[i]);
                dA = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ownA[i]), bv, dA, false);
                dB = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ownB[i]), bv, dB, false);
                dC = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ownC[i]), bv, dC, false);
            }
            dA *= P.alpha, dB *= P.alpha, dC *= P.alpha;
            const uint32_t pab = line_cvt2(dA, dB), pc = line_cvt2(dC, dC);
            *reinterpret_cast<unsigned short*>(sm + saddr[t] + uA) = (unsigned short)pab;
            *reinterpret_cast<unsigned short*>(sm + saddr[t] + uB) = (unsigned short)(pab >> 16);
            *reinterpret_cast<unsigned short*>(sm + saddr[t] + uC) = (unsigned short)pc;
        }
        if (j >= 2) {   // target j - 2 is complete: this wave's 32 rows (written by this wave only) leave as 16-byte pieces
            const int64_t row0 = item_row0 + (int64_t)(x0 + j - 2) * plane_rows + y0 * P.nz + wave * (kWave / 2);
            char* const dst = static_cast<char*>(P.gvals) + row0 * kLineValB;
            const char* const src = sm + P.o_stage + slotA * P.stage_bytes + wave * (kWave / 2 * kLineValB);
            stream_store16(dst + lane * 16, *reinterpret_cast<const uint4*>(src + lane * 16));
            if (lane + kWave < pieces) stream_store16(dst + (lane + kWave) * 16, *reinterpret_cast<const uint4*>(src + (lane + kWave) * 16));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) ownA[i] = ownB[i], ownB[i] = ownC[i];
        bufo ^= (unsigned)P.buf_stride;
        lat_step_sync();
    }
}

inline int linemarch_layout(LineParams& P, int threads, int mode) {
    if (P.nz <= 0 || P.nz % 8 || P.ty <= 0 || P.ty * P.nz * 2 != threads) return TSGU_ERR_BAD_ARG;
    if (threads != 256 && threads != 512 && threads != 1024) return TSGU_ERR_BAD_ARG;
    if (P.ny % P.ty) return TSGU_ERR_BAD_ARG;
    const int hl = P.ty + 2;
    P.g_bytes = hl * P.nz * kLineRowB;
    if (hl * P.nz * 2 > kLineKG * threads) return TSGU_ERR_TOO_LARGE;
    int region;
    if (mode == kLatSpmmT) {
        region = P.g_bytes + hl * P.nz * kLineValB;
        if (hl * (P.nz * kLineValB / 16) > kLineKV * threads) return TSGU_ERR_TOO_LARGE;
    } else if (mode == kLatSddmm) {
        region = P.g_bytes + P.ty * P.nz * kLineRowB;    // + the row operand's plane (own rows only)
    } else if (mode == kLatSpmm) {
        region = P.g_bytes;
        if (P.ty * (P.nz * kLineValB / 16) > kLineKV * threads) return TSGU_ERR_TOO_LARGE;
    } else {
        return TSGU_ERR_BAD_ARG;
    }
    int stride = 1024;
    while (stride < region) stride *= 2;
    P.buf_stride = stride;
    // (reads never leave a region; the DMA of a partial last wave writes nothing beyond its active lanes)
    P.lds_bytes = stride + region;
    if (mode == kLatSddmm) {
        P.buf_stride = region;                           // (no address toggling by xor here: the buffers lie back to back)
        P.o_stage = 2 * region;
        P.stage_bytes = P.ty * P.nz * kLineValB;
        P.lds_bytes = P.o_stage + 3 * P.stage_bytes;
    }
    if (mode == kLatSpmm) {
        P.buf_stride = region;
        P.o_stage = 2 * region;                          // the value ring: four planes of the tile's own lines
        P.stage_bytes = P.ty * P.nz * kLineValB;
        P.lds_bytes = P.o_stage + 4 * P.stage_bytes;
    }
    if (P.lds_bytes > kLatMaxLds) return TSGU_ERR_TOO_LARGE;
    return P.lds_bytes;
}

template <int NT, int MODE>
int linemarch_launch_t(const LineParams& P, hipStream_t stream) {
    static std::atomic<uint64_t> allowed{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TSGU_ERR_RUNTIME;
    auto* const kernel = MODE == kLatSddmm ? &linemarch_sddmm_kernel<NT> : (MODE == kLatSpmm ? &linemarch_spmm_kernel<NT> : &linemarch_spmmt_kernel<NT>);
    if (!(allowed.load(std::memory_order_acquire) >> dev & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLatMaxLds) != hipSuccess)
            return TSGU_ERR_RUNTIME;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)P.nblocks), dim3(NT), (size_t)P.lds_bytes, stream, P);
    return check_launch();
}

}  // namespace tsgu
