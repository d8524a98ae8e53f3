// Fused backward march: gradA (SDDMM) and gradB (Aᵀ·G) of C = A·B in ONE march over the lattice, for the whole-box stencils of
// march_impl.h (27-point, periodic or truncated).
//
// Run as two launches, the SDDMM reads every row of G once as its OWN row (128 MB at C2) and the transposed product reads the
// same rows again as its GATHERED halo rows.  Here the halo ring of G serves both: the transposed product gathers from it, and
// the centre of a halo plane IS the own row the SDDMM needs for that plane.  The two walks are the march_impl.h walks, in one
// loop, with the G ring one plane AHEAD of the B ring:
//
//     step s:   SDDMM   on source plane s   of B  (targets s+1, s, s-1; own rows = centres of G planes s+1, s, s-1 — the one of
//                       s+1 is read from the G ring at the top of the step, the others rotate in registers)
//               SpMM-T  on source plane s+1 of G  (targets s+2, s+1, s; values of the halo rows of plane s+1)
//     requested at the top of step s (asynchronous, waited for at its end): B plane s+1, G plane s+2 and its halo value rows.
//
// Traffic per step at C2: values 108 MB (+ halo), G 128 MB (+ halo), B 128 MB (+ halo) in, gradA 108 MB + gradB 128 MB out —
// the minimum fused backward of SURVEY §8(d) plus the halo re-reads; 128 MB (the SDDMM's own rows) less than the two launches.
// Three rings (B, G, halo values) + the SDDMM's stage rows: 49 KB at the 4 x 8 tile — three 256-thread workgroups per CU.
// Sums run exactly as in march_impl.h: gradA bit-identical to the plan-free SDDMM, gradB bit-identical to the unfused march.
#pragma once

#include "march_impl.h"

namespace tsgu {

template <int CL, int NT, int ROWS>
__global__ __launch_bounds__(NT, 3) void march_bwd_kernel(const MarchParams P) {
    static_assert(ROWS == kRowsUniform || ROWS == kRowsBox, "the whole box: uniform rows or box arithmetic");
    constexpr bool UNIF = ROWS == kRowsUniform, BOXA = ROWS == kRowsBox;
    constexpr int NTAP = 9;
    constexpr int RB = CL * 16;
    constexpr int NG = NT / CL;
    constexpr int RPW = kWave / CL;
    constexpr int NS = 3 * NTAP;
    constexpr int SLOTS = (NS + 3) / 4 * 4;
    constexpr int VP = SLOTS * 4;
    constexpr int VL = SLOTS / 4;
    constexpr int NGI = (RPW * SLOTS + kWave - 1) / kWave;
    constexpr int NPASS = 2;                     // halo rows staged per plane: up to two per row group
    constexpr int RJ = (NS + CL - 1) / CL;
    constexpr int NF = (RPW * VL + kWave - 1) / kWave;
    static_assert(CL == 8, "the transposed reduction of the SDDMM: 8 lanes per row (32 fp32 columns)");

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

    const int64_t vblock = xcd_chunked_block(blockIdx.x, P.nblocks);
    int64_t vb = vblock;
    const int tzi = (int)(vb % P.tiles_z);
    vb /= P.tiles_z;
    const int tyi = (int)(vb % P.tiles_y);
    vb /= P.tiles_y;
    const int seg = (int)(vb % P.nseg);
    const int item = (int)(vb / P.nseg);
    const int xs = seg * P.seg_len;
    const int L = P.seg_len < P.nx - xs ? P.seg_len : P.nx - xs;
    const int y0 = tyi * P.ty, z0 = tzi * P.tz;
    const int item_row0 = item * P.nx * plane_rows;
    auto row_of_x = [&](int x) -> int { return item_row0 + x * plane_rows; };
    auto x_ok = [&](int s) -> bool { return P.per_x || (unsigned)(xs - 1 + s) < (unsigned)P.nx; };
    auto x_of = [&](int s) -> int { return lat_mod(xs - 1 + s, P.nx); };         // lattice plane of ring index s
    auto halo_row = [&](int hy, int hz) -> int {
        const int yy = y0 - P.ry + hy, zz = z0 - P.rz + hz;
        const bool in = (P.per_y || (unsigned)yy < (unsigned)P.ny) && (P.per_z || (unsigned)zz < (unsigned)P.nz);
        return in ? lat_mod(yy, P.ny) * P.nz + lat_mod(zz, P.nz) : -1;
    };
    auto cnt1 = [](int t, int n, int per) -> int { return per ? 3 : 3 - (t == 0) - (t == n - 1); };
    auto pre1 = [](int t, int per) -> int { return per ? 3 * t : 3 * t - (t > 0); };
    const int boxLz = pre1(P.nz, P.per_z) - (P.per_z ? 0 : 1), boxLy = pre1(P.ny, P.per_y) - (P.per_y ? 0 : 1);
    const int boxLyz = boxLy * boxLz, boxLtot = (pre1(P.nx, P.per_x) - (P.per_x ? 0 : 1)) * boxLyz;
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
        else return NS;
    };
    auto plane_base = [&](int x) -> int {
        if constexpr (BOXA) return item * boxLtot + boxLyz * pre1(x, P.per_x);
        else return row_of_x(x) * NS;
    };

    // ---- tables -> LDS ---------------------------------------------------------------------------------------------------
    {
        const int* src = reinterpret_cast<const int*>(P.kidx);
        int* dst = reinterpret_cast<int*>(sm + P.o_tab);
        for (int i = tid; i < P.ncls * 8; i += NT) dst[i] = src[i];
        int* rows = reinterpret_cast<int*>(sm + P.o_rows);
        for (int r = tid; r < HR; r += NT) {
            const int hy = r / HZ, hz = r - hy * HZ;
            rows[r] = halo_row(hy, hz);
        }
    }
    __syncthreads();
    const unsigned char* const kidx_s = reinterpret_cast<const unsigned char*>(sm + P.o_tab);
    const int* const rows_s = reinterpret_cast<const int*>(sm + P.o_rows);

    // ---- descriptors -------------------------------------------------------------------------------------------------------
    const uint32_t ldbb = (uint32_t)P.lds_ * 4u, ldgb = (uint32_t)P.ldown * 4u;      // row pitch of B and G in bytes
    int rrow[kMarchND];                                                                  // halo row of this thread's d-th ring piece
    const int ring_pieces = HR * CL;
#pragma unroll
    for (int d = 0; d < kMarchND; ++d) {
        const int e = d * NT + tid;
        const int hr = e / CL;
        const int hy = hr / HZ, hz = hr - hy * HZ;
        rrow[d] = e < ring_pieces ? halo_row(hy, hz) : -2;                               // -1: beyond a face, -2: no piece
    }
    if (!(P.per_y && P.per_z)) {
#pragma unroll
        for (int d = 0; d < kMarchND; ++d) {
            const int e = d * NT + tid;
            if (rrow[d] == -1) {
#pragma unroll
                for (int slot = 0; slot < 4; ++slot) lat_smem[slot * HR * CL + e] = make_uint4(0, 0, 0, 0);   // B ring, then G ring
            }
        }
        for (int e = tid; e < HR * VL; e += NT) {
            if (rows_s[e / VL] < 0) {
                *reinterpret_cast<uint4*>(sm + P.o_vals + e * 16) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(sm + P.o_vals + HR * VP + e * 16) = make_uint4(0, 0, 0, 0);
            }
        }
    }
    const int ly = g / P.tz, lz = g - ly * P.tz;
    const bool ok = g < NR && y0 + ly < P.ny && z0 + lz < P.nz;
    const int crow = ok ? (y0 + ly) * P.nz + z0 + lz : -1;
    const int hrow = (ly + P.ry) * HZ + lz + P.rz;
    const uint32_t coo = (uint32_t)(ok ? crow : 0) * ((uint32_t)P.ldo * 4u) + (uint32_t)c * 16u;
    const int cen = hrow * RB + c * 16;
    const int own_const = row_const(crow > 0 ? crow : 0);

    int srow[NPASS];
    uint32_t foff[NPASS][NF];
    uint32_t fpo[NF];          // kRowsBox: byte offset of this lane's f-th 16-byte piece inside its row (row constants < 2^24)
#pragma unroll
    for (int f = 0; f < NF; ++f) fpo[f] = (uint32_t)((f * kWave + lane) % VL) * 16u;
    int gconst[BOXA ? NPASS : 1][BOXA ? NGI : 1];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const int r = q * NG + g;
        srow[q] = r < HR ? rows_s[r] : -1;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int piece = f * kWave + lane;
            const int fr = q * NG + wave * RPW + piece / VL;
            const int frow = (piece < RPW * VL && fr < HR) ? rows_s[fr] : -1;
            if constexpr (BOXA) foff[q][f] = frow >= 0 ? (uint32_t)row_const(frow) * 4u : kLatNone;
            else foff[q][f] = frow >= 0 ? (uint32_t)frow * (uint32_t)(NS * 4) + (uint32_t)(piece % VL) * 16u : kLatNone;
        }
        if constexpr (BOXA) {
#pragma unroll
            for (int n = 0; n < NGI; ++n) {
                const int rw = (n * kWave + lane) / SLOTS;
                const int fr = q * NG + wave * RPW + rw;
                const int rr = (rw < RPW && fr < HR) ? rows_s[fr] : -1;
                gconst[q][n] = row_const(rr > 0 ? rr : 0);
            }
        }
    }

    const char* const Bb = static_cast<const char*>(P.S);
    const char* const Gb = static_cast<const char*>(P.Own);
    const char* const valb = static_cast<const char*>(P.val);
    const uint32_t val_bytes = (uint32_t)(P.nnz * 4);
    const int o_gring = 2 * PB;

    // a dense halo plane (rows of `pitch` bytes, first row `prow`) into the ring slot at byte `region`
    auto dma_plane = [&](const char* base, uint32_t pitch, int prow, unsigned region) {
        const char* const pbase = base + (int64_t)prow * pitch;
        const unsigned wb = sbase + region + (unsigned)(wave * kWave * 16);
#pragma unroll
        for (int d = 0; d < kMarchND; ++d) {
            if (d * NT < ring_pieces) {
                if (rrow[d] >= 0) lat_dma16<false>(pbase, (uint32_t)rrow[d] * pitch + (uint32_t)c * 16u, wb + (unsigned)(d * NT * 16));
            }
        }
    };
    // canonical value rows of the halo rows of lattice plane x into the value ring slot at byte `region`
    auto stage_vals = [&](int x, unsigned region, const int (&cls)[NPASS]) {
        const int pbase = plane_base(x), pcx = plane_cx(x);
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int first = q * NG + wave * RPW;
            if (first < HR) {
                const unsigned wbase = sbase + region + (unsigned)(first * VP);
                const bool plain = __builtin_amdgcn_ballot_w64(srow[q] >= 0 && cls[q] != P.ident) == 0;
                if (plain) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        uint32_t fo = foff[q][f];
                        if constexpr (UNIF) {
                            if (fo != kLatNone) fo += (uint32_t)pbase * 4u;
                        } else {
                            if (fo != kLatNone) fo = __umul24((uint32_t)pcx, fo) + ((uint32_t)pbase * 4u + fpo[f]);   // (one v_mad_u32_u24)
                        }
                        if (fo != kLatNone) {
                            if (__builtin_expect(fo + 16u <= val_bytes, 1)) {
                                lat_dma16<false>(valb, fo, wbase + (unsigned)(f * kWave * 16));
                            } else {
                                float* dst = reinterpret_cast<float*>(sm + region + first * VP + (f * kWave + lane) * 16);
#pragma nounroll
                                for (int e = 0; e < 4; ++e)
                                    dst[e] = fo + (e + 1) * 4 <= val_bytes ? *reinterpret_cast<const float*>(valb + fo + e * 4) : 0.f;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int n = 0; n < NGI; ++n) {
                        const int e = n * kWave + lane;
                        const int rw = e / SLOTS, slot = e - rw * SLOTS;
                        const int src_lane = (rw < RPW ? rw * CL : 0) * 4;
                        const int rc = __builtin_amdgcn_ds_bpermute(src_lane, cls[q]);
                        const int rr = (rw < RPW && first + rw < HR) ? rows_s[first + rw] : -1;
                        int rs;
                        if constexpr (BOXA) rs = pbase + (int)__umul24((uint32_t)pcx, (uint32_t)gconst[q][n]);
                        else rs = pbase + pcx * (rr > 0 ? rr : 0);
                        const int k = kidx_s[rc * 32 + (slot & 31)];
                        if (rr >= 0 && slot < NS && (UNIF || k != 0xFF))     // (slots without an entry are never read: march_impl.h)
                            lat_dma4<false>(valb, (uint32_t)rs * 4u + (uint32_t)k * 4u, wbase + (unsigned)(n * kWave * 4));
                    }
                }
            }
        }
    };
    auto load_cls = [&](int prow, int (&cls)[NPASS]) {
        const unsigned char* const cbase = P.rcls + prow;
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            cls[q] = P.ident;
            if (srow[q] >= 0) cls[q] = cbase[(uint32_t)srow[q]];
        }
    };
    auto pin_cls = [&](int (&cls)[NPASS]) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q) lat_pin(cls[q]);
    };

    int tapb[NTAP], tapv[NTAP];
#pragma unroll
    for (int i = 0; i < NTAP; ++i) tapb[i] = P.tap_row[i] * RB, tapv[i] = P.tap_row[i] * VP;
    auto as4 = [](const uint4& raw, float (&f)[4]) {
        f[0] = __uint_as_float(raw.x), f[1] = __uint_as_float(raw.y), f[2] = __uint_as_float(raw.z), f[3] = __uint_as_float(raw.w);
    };
    const int vbuf = HR * VP;

    // ---- prologue: the G plane and the halo value rows of ring index 0 (the transposed product's first source plane); the
    // class bytes of ring index 1, which the first step stages --------------------------------------------------------------
    int cls[NPASS], cld[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) cls[q] = cld[q] = P.ident;
    if (x_ok(0)) {
        load_cls(row_of_x(x_of(0)), cls);
        pin_cls(cls);
        dma_plane(Gb, ldgb, row_of_x(x_of(0)), (unsigned)o_gring);
        stage_vals(x_of(0), (unsigned)P.o_vals, cls);
    }
    load_cls(row_of_x(x_of(1)), cld);
    int clsP = P.ident, clsP_next = P.ident;       // class of the SDDMM target that is staged in this / the next step
    lat_step_sync();

    // SDDMM state
    uint4 oP = make_uint4(0, 0, 0, 0), oC = oP, oN = oP;
    float rP[RJ], rC[RJ], rN[RJ];
#pragma unroll
    for (int j = 0; j < RJ; ++j) rP[j] = rC[j] = rN[j] = 0.f;
    float* const st = reinterpret_cast<float*>(sm + P.o_stage + g * VP);
    bool staged = false;
    int fl_start = 0, fl_len = 0;
    // SpMM-T state
    float accP[4], accC[4], accN[4], done[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) accP[v] = accC[v] = accN[v] = done[v] = 0.f;
    const uint32_t ldob = (uint32_t)P.ldo * 4u;

    auto flush_gvals = [&]() {
        if (crow >= 0) {
            float* const go = static_cast<float*>(P.gvals) + fl_start;
#pragma nounroll
            for (int k0 = c * 4; k0 < fl_len; k0 += CL * 4) {
                const float4 w = *reinterpret_cast<const float4*>(st + k0);
                if (k0 + 4 <= fl_len) {
                    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                    const f4u o = {w.x, w.y, w.z, w.w};
                    __builtin_nontemporal_store(o, reinterpret_cast<f4u*>(go + k0));
                } else {
                    const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k0 + j < fl_len) go[k0 + j] = wv[j];
                }
            }
        }
    };
    auto flush_gradb = [&](int x) {
        if (crow >= 0) {
            char* const obase = static_cast<char*>(P.out) + (int64_t)row_of_x(x) * ldob;
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v o = {done[0], done[1], done[2], done[3]};
            __builtin_nontemporal_store(o, reinterpret_cast<f4v*>(obase + coo));
        }
    };

    for (int s = -1; s <= L + 1; ++s) {
        // 1. what the previous step loaded / completed
        pin_cls(cld);
#pragma unroll
        for (int q = 0; q < NPASS; ++q) cls[q] = cld[q];
        lat_pin(clsP_next);
        clsP = clsP_next;
        if (staged) flush_gvals();
        staged = false;
        if (s - 1 >= 1 && s - 1 <= L) flush_gradb(x_of(s - 1));     // the transposed product's target s-1 was completed by step s-1
        // own rows of the SDDMM: target s+1 is the centre of the G plane that arrived during the previous step
        oP = oC, oC = oN;
        oN = make_uint4(0, 0, 0, 0);
        if (crow >= 0 && s + 1 >= 1 && s + 1 <= L) oN = *reinterpret_cast<const uint4*>(sm + o_gring + ((s + 1) & 1) * PB + cen);
        // 2. requests: B plane s+1, G plane s+2 with the value rows of its halo
        if (s + 1 <= L + 1 && x_ok(s + 1)) dma_plane(Bb, ldbb, row_of_x(x_of(s + 1)), (unsigned)(((s + 1) & 1) * PB));
        if (s + 2 <= L + 1 && x_ok(s + 2)) {
            dma_plane(Gb, ldgb, row_of_x(x_of(s + 2)), (unsigned)(o_gring + (s & 1) * PB));
            stage_vals(x_of(s + 2), (unsigned)(P.o_vals + (s & 1) * vbuf), cls);
        }
        // 3. class bytes for the next step: the halo rows of value plane s+3; the SDDMM target that is staged next (target s)
        if (s + 3 <= L + 1 && x_ok(s + 3)) load_cls(row_of_x(x_of(s + 3)), cld);
        if (crow >= 0 && s >= 1 && s <= L) clsP_next = P.rcls[row_of_x(x_of(s)) + crow];

        if (crow >= 0) {
            // 4. SDDMM: source plane s of B; dots of targets s+1 (N), s (C), s-1 (P) with their own G rows
            if (s >= 0 && x_ok(s)) {
                const char* const bb = sm + (s & 1) * PB + cen;
                typedef float f2v __attribute__((ext_vector_type(2)));
                f2v nc[4];
                float op[4];
                {
                    float on[4], oc[4];
                    as4(oN, on);
                    as4(oC, oc);
                    as4(oP, op);
#pragma unroll
                    for (int v = 0; v < 4; ++v) nc[v] = f2v{on[v], oc[v]};
                }
                float pd[3][NTAP];
                uint4 b[NTAP];
                constexpr int kAhead = 3;
#pragma unroll
                for (int i = 0; i < kAhead && i < NTAP; ++i) b[i] = *reinterpret_cast<const uint4*>(bb + tapb[i]);
#pragma unroll
                for (int i = 0; i < NTAP; ++i) {
                    if (i + kAhead < NTAP) b[i + kAhead] = *reinterpret_cast<const uint4*>(bb + tapb[i + kAhead]);
                    asm volatile("" ::: "memory");
                    float f[4];
                    as4(b[i], f);
                    const f2v f01 = {f[0], f[1]}, f23 = {f[2], f[3]};
                    f2v d2;
                    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d2) : "v"(nc[0]), "v"(f01));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d2) : "v"(nc[1]), "v"(f01));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d2) : "v"(nc[2]), "v"(f23));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d2) : "v"(nc[3]), "v"(f23));
                    float d;
                    asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(op[0]), "v"(f[0]));
#pragma unroll
                    for (int v = 1; v < 4; ++v) asm("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(op[v]), "v"(f[v]));
                    pd[0][i] = d2.x, pd[1][i] = d2.y, pd[2][i] = d;
                }
                lat_static_for<0, RJ>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    auto sl = [&](auto E) -> float {
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
                            const bool whole = lo == 0 && (hi == 8 || 8 * j + hi == NS);
                            r = whole || (c >= lo && c < hi) ? tot : r;
                        }
                    }
                });
            }
            // 5. SDDMM target s-1 is complete: its dots go to the stage row at their stored positions
            if (s >= 2) {
                const bool plain = clsP == P.ident;
#pragma unroll
                for (int j = 0; j < RJ; ++j) {
                    const int slot = j * CL + c;
                    if (slot < NS) {
                        const int k = plain ? slot : (int)kidx_s[clsP * 32 + slot];
                        if (UNIF || k != 0xFF) st[k] = P.alpha * rP[j];
                    }
                }
                const int xt = x_of(s - 1);
                fl_start = plane_base(xt) + plane_cx(xt) * own_const;
                fl_len = UNIF ? NS : (int)kidx_s[clsP * 32 + 31];
                staged = true;
            }
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                rP[j] = rC[j];
                rC[j] = rN[j];
            }
            // 6. transposed product: source plane s+1 of G; targets s+2 (N), s+1 (C), s (P)
#pragma unroll
            for (int v = 0; v < 4; ++v) accN[v] = 0.f;
            if (s + 1 <= L + 1 && x_ok(s + 1)) {
                const char* const gb = sm + o_gring + ((s + 1) & 1) * PB + cen;
                const char* const vb0 = sm + P.o_vals + ((s + 1) & 1) * vbuf + hrow * VP;
                uint4 b[NTAP];
                float a[NTAP][3];
                constexpr int kAhead = 2;
                auto fetch = [&](int i) {
                    b[i] = *reinterpret_cast<const uint4*>(gb + tapb[i]);
                    const char* const vr = vb0 + tapv[i];
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const float*>(vr + ((2 - p) * NTAP + NTAP - 1 - i) * 4);
                };
#pragma unroll
                for (int i = 0; i < kAhead && i < NTAP; ++i) fetch(i);
#pragma unroll
                for (int i = 0; i < NTAP; ++i) {
                    if (i + kAhead < NTAP) fetch(i + kAhead);
                    asm volatile("" ::: "memory");
                    float f[4];
                    as4(b[i], f);
                    const float aN = a[i][0], aC = a[i][1], aP = a[i][2];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        accN[v] = fmaf(aN, f[v], accN[v]);
                        accC[v] = fmaf(aC, f[v], accC[v]);
                        accP[v] = fmaf(aP, f[v], accP[v]);
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                done[v] = accP[v];
                accP[v] = accC[v];
                accC[v] = accN[v];
            }
        }
        lat_step_sync();
    }
    if (staged) flush_gvals();
}

// LDS layout of the fused backward: B ring | G ring | value ring | stage rows | kidx | halo rows.  Returns bytes or a status.
inline int march_bwd_layout(MarchParams& P, int cl, int nt) {
    if (P.ty <= 0 || P.tz <= 0 || P.ry != 1 || P.rz != 1 || P.ncls <= 0 || P.ncls > kMarchMaxCls || cl != 8) return TSGU_ERR_BAD_ARG;
    const int HR = (P.ty + 2) * (P.tz + 2), NR = P.ty * P.tz, RB = cl * 16, VP = 112;
    const int NG = nt / cl;
    if ((int64_t)HR * cl > (int64_t)kMarchND * nt || NR > NG || HR > 2 * NG) return TSGU_ERR_TOO_LARGE;
    int64_t o = 4 * (int64_t)HR * RB;
    P.o_vals = (int)o;
    o += 2 * (int64_t)HR * VP;
    P.o_stage = (int)o;
    o += (int64_t)NR * VP;
    P.o_tab = (int)o;
    o += P.ncls * 32;
    P.o_rows = (int)o;
    o += lat_round16(HR * 4);
    if (o > kLatMaxLds) return TSGU_ERR_TOO_LARGE;
    P.lds_bytes = (int)o;
    return (int)o;
}

template <int CL, int NT, int ROWS>
int march_bwd_launch(const MarchParams& P, hipStream_t stream) {
    static std::atomic<uint64_t> allowed{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TSGU_ERR_RUNTIME;
    if (!(allowed.load(std::memory_order_acquire) >> dev & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&march_bwd_kernel<CL, NT, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLatMaxLds) != hipSuccess)
            return TSGU_ERR_RUNTIME;
        allowed.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL((march_bwd_kernel<CL, NT, ROWS>), dim3((unsigned)P.nblocks), dim3(NT), (size_t)P.lds_bytes, stream, P);
    return check_launch();
}

}  // namespace tsgu
