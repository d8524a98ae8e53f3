// K6: fused BiCGSTAB recurrences on gfx950 (optionally right-preconditioned), every right-hand side at once.
//
// The reference (utils/bicgstab.py:113-247, a pykrylov port) solves the columns one after the other
// in a Python loop, with ~12 small ATen ops and two host reads of the residual norm per iteration and
// column.  The columns are independent problems, so here they advance in lock-step on the device:
// every per-column scalar (rho, alpha, omega, rho_next, threshold, residual norm) and the per-column
// "finished" state live in device arrays, a finished column is simply masked out of every update, and
// one iteration is
//   scalar(BETA) -> update_p -> K1 SpMM(+<r0,v> partials) -> scalar(ALPHA) -> update_s(+|s|^2 partials)
//   -> scalar(HALF) -> K1 SpMM -> dots3 -> scalar(OMEGA) -> update_x(+|r|^2 partials) -> scalar(END)
// The host only polls one 4-byte "all columns finished" word every few iterations.  Reductions are
// two-stage in a fixed order (deterministic).  Per column the arithmetic follows the reference line by
// line (cited below), including its quirks (rho_next = -omega<r0,t>, the early exit after the first
// half step, the matvec budget).
//
// scal  [8][p]  (value type): 0 rho | 1 alpha | 2 omega | 3 rho_next | 4 threshold | 5 beta | 6 resid | 7 resid0
// flags int32 : [0] all finished | [1] iterations | [2..2+p) finished | [2+p..2+2p) finishing after the
//               half step (x += alpha*p still due) | [2+2p..2+3p) matvecs used
#include "krylov_common.h"

namespace tsgu {

enum BicgPhase { kBicgInit = 0, kBicgBeta = 1, kBicgAlpha = 2, kBicgHalf = 3, kBicgOmega = 4, kBicgEnd = 5 };

template <typename V>
__global__ __launch_bounds__(kBlock) void bicg_scalar_kernel(int phase, const V* __restrict__ partial, int64_t n_partial,
                                                             int64_t set_stride, int64_t p, V* __restrict__ scal,
                                                             int* __restrict__ flags, V abstol, V reltol, int matvec_max,
                                                             int nmv0) {
    __shared__ V red[kBlock];
    __shared__ int s_allfin;
    if (phase != kBicgInit && flags[0] != 0) return;
    const int t = threadIdx.x;
    int* fin = flags + 2;
    int* half = flags + 2 + p;
    int* nmv = flags + 2 + 2 * p;
    if (t == 0) s_allfin = 1;
    __syncthreads();
    for (int64_t c0 = 0; c0 < p; c0 += 64) {
        const int w = (int)(p - c0 < 64 ? p - c0 : 64);
        V s0 = 0, s1 = 0, s2 = 0;
        if (phase == kBicgInit || phase == kBicgAlpha || phase == kBicgHalf || phase == kBicgEnd || phase == kBicgOmega)
            s0 = block_colsum<V>(partial, n_partial, p, c0, w, red);
        if (phase == kBicgOmega) {
            s1 = block_colsum<V>(partial + set_stride, n_partial, p, c0, w, red);
            s2 = block_colsum<V>(partial + 2 * set_stride, n_partial, p, c0, w, red);
        }
        if (t < w) {
            const int64_t c = c0 + t;
            V* rho = scal + c;
            V* alpha = scal + p + c;
            V* omega = scal + 2 * p + c;
            V* rho_next = scal + 3 * p + c;
            V* thr = scal + 4 * p + c;
            V* beta = scal + 5 * p + c;
            V* resid = scal + 6 * p + c;
            if (phase == kBicgInit) {
                // rho = alpha = omega = 1; rho_next = <r0,r0>; threshold (bicgstab.py:163-169)
                *rho = 1;
                *alpha = 1;
                *omega = 1;
                *rho_next = s0;
                const V r0n = fabs(sqrt(s0));
                scal[7 * p + c] = r0n;
                *resid = r0n;
                const V th = reltol * r0n > abstol ? reltol * r0n : abstol;
                *thr = th;
                nmv[c] = nmv0;
                half[c] = 0;
                fin[c] = (r0n <= th || nmv0 >= matvec_max) ? 1 : 0;
            } else if (!fin[c]) {
                if (phase == kBicgBeta) {
                    // beta = rho_next/rho * alpha/omega; rho = rho_next (bicgstab.py:183-184)
                    *beta = *rho_next / *rho * *alpha / *omega;
                    *rho = *rho_next;
                } else if (phase == kBicgAlpha) {
                    nmv[c] += 1;
                    *alpha = *rho / s0;  // rho / <r0, v> (bicgstab.py:199)
                } else if (phase == kBicgHalf) {
                    const V rn = sqrt(s0);  // ||s|| (bicgstab.py:203)
                    *resid = rn;
                    if (rn <= *thr) half[c] = 1;               // x += alpha*q, finished (:207-210)
                    else if (nmv[c] >= matvec_max) fin[c] = 1;  // (:212-214)
                } else if (phase == kBicgOmega) {
                    if (!half[c]) {
                        nmv[c] += 1;
                        const V om = s0 / s1;   // <t,s>/<t,t> (bicgstab.py:223)
                        *omega = om;
                        *rho_next = -om * s2;  // -omega <r0,t> (:224)
                    }
                } else if (phase == kBicgEnd) {
                    if (half[c]) {
                        half[c] = 0;
                        fin[c] = 1;
                    } else {
                        const V rn = sqrt(s0);  // ||r|| (bicgstab.py:235)
                        *resid = rn;
                        if (rn <= *thr || nmv[c] >= matvec_max) fin[c] = 1;  // (:239-241)
                    }
                }
            }
            if ((phase == kBicgInit || phase == kBicgEnd) && !fin[c]) atomicAnd(&s_allfin, 0);
        }
        __syncthreads();
    }
    if (t == 0 && (phase == kBicgInit || phase == kBicgEnd)) {
        if (phase == kBicgEnd) flags[1] += 1;
        flags[0] = s_allfin;
    }
}

// p = (p*beta - (beta*omega)*v) + r      (bicgstab.py:187-189)
template <typename V, int VEC>
__global__ __launch_bounds__(kBlock) void bicg_update_p_kernel(int64_t n, int64_t p, V* __restrict__ pv, const V* __restrict__ r,
                                                               const V* __restrict__ v, const V* __restrict__ scal,
                                                               const int* __restrict__ flags, int lpr, int rpp) {
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    if (!(rs < rpp && c < p)) return;
    V beta[VEC], bo[VEC];
    bool act[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        act[k] = flags[2 + c + k] == 0;
        beta[k] = scal[5 * p + c + k];
        bo[k] = beta[k] * scal[2 * p + c + k];
    }
    const int64_t r0 = (int64_t)blockIdx.x * rpp * kPasses;
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t row = r0 + (int64_t)ps * rpp + rs;
        if (row < n) {
            const int64_t o = row * p + c;
            V pp[VEC], rr[VEC], vv[VEC];
            load_vec<V, VEC>(pv + o, pp);
            load_vec<V, VEC>(r + o, rr);
            load_vec<V, VEC>(v + o, vv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const V nw = (pp[k] * beta[k] - bo[k] * vv[k]) + rr[k];
                pp[k] = act[k] ? nw : pp[k];
            }
            store_vec<V, VEC>(pv + o, pp);
        }
    }
}

// s = r - alpha*v ; partial |s|^2      (bicgstab.py:200-203)
template <typename V, int VEC>
__global__ __launch_bounds__(kBlock) void bicg_update_s_kernel(int64_t n, int64_t p, V* __restrict__ s, const V* __restrict__ r,
                                                               const V* __restrict__ v, const V* __restrict__ scal,
                                                               const int* __restrict__ flags, int lpr, int rpp,
                                                               V* __restrict__ partial) {
    __shared__ V red[kBlock * VEC];
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    const bool on = rs < rpp && c < p;
    V alpha[VEC], acc[VEC];
    bool act[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        acc[k] = 0;
        act[k] = on && flags[2 + c + k] == 0;
        alpha[k] = on ? scal[p + c + k] : (V)0;
    }
    const int64_t r0 = (int64_t)blockIdx.x * rpp * kPasses;
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t row = r0 + (int64_t)ps * rpp + rs;
        if (on && row < n) {
            const int64_t o = row * p + c;
            V ss[VEC], rr[VEC], vv[VEC];
            load_vec<V, VEC>(s + o, ss);
            load_vec<V, VEC>(r + o, rr);
            load_vec<V, VEC>(v + o, vv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const V nw = rr[k] - alpha[k] * vv[k];
                ss[k] = act[k] ? nw : ss[k];
                acc[k] = fma(ss[k], ss[k], acc[k]);
            }
            store_vec<V, VEC>(s + o, ss);
        }
    }
    if (rs < rpp) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) red[(rs * lpr + cl) * VEC + k] = acc[k];
    }
    __syncthreads();
    for (int64_t cc = t; cc < p; cc += kBlock) {
        V sum = 0;
        for (int k = 0; k < rpp; ++k) sum += red[k * lpr * VEC + cc];
        partial[(int64_t)blockIdx.x * p + cc] = sum;
    }
}

// partial <t,s>, <t,t>, <r0,t>      (bicgstab.py:223-224); sets are `set_stride` elements apart
template <typename V, int VEC>
__global__ __launch_bounds__(kBlock) void bicg_dots3_kernel(int64_t n, int64_t p, const V* __restrict__ tv, const V* __restrict__ s,
                                                            const V* __restrict__ r0v, const int* __restrict__ flags, int lpr,
                                                            int rpp, V* __restrict__ partial, int64_t set_stride) {
    __shared__ V red[kBlock * VEC];
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    const bool on = rs < rpp && c < p;
    V a0[VEC], a1[VEC], a2[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) a0[k] = a1[k] = a2[k] = 0;
    const int64_t rb = (int64_t)blockIdx.x * rpp * kPasses;
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t row = rb + (int64_t)ps * rpp + rs;
        if (on && row < n) {
            const int64_t o = row * p + c;
            V tt[VEC], ss[VEC], rr[VEC];
            load_vec<V, VEC>(tv + o, tt);
            load_vec<V, VEC>(s + o, ss);
            load_vec<V, VEC>(r0v + o, rr);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                a0[k] = fma(tt[k], ss[k], a0[k]);
                a1[k] = fma(tt[k], tt[k], a1[k]);
                a2[k] = fma(rr[k], tt[k], a2[k]);
            }
        }
    }
    for (int set = 0; set < 3; ++set) {
        if (rs < rpp) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) red[(rs * lpr + cl) * VEC + k] = set == 0 ? a0[k] : set == 1 ? a1[k] : a2[k];
        }
        __syncthreads();
        for (int64_t cc = t; cc < p; cc += kBlock) {
            V sum = 0;
            for (int k = 0; k < rpp; ++k) sum += red[k * lpr * VEC + cc];
            partial[set * set_stride + (int64_t)blockIdx.x * p + cc] = sum;
        }
        __syncthreads();
    }
}

// half-finished columns: x += alpha*q.  active columns: r = s - omega*t; x = (x + omega*z) + alpha*q; partial |r|^2
// (bicgstab.py:207-210, 227-235).  Without a preconditioner q is p and z is s (PRE = false: `pv` serves as q, `s` as z);
// with one, q = M p and z = M s arrive in `pv` and `zv` (bicgstab.py:191-194, 216-219).
template <typename V, int VEC, bool PRE = false>
__global__ __launch_bounds__(kBlock) void bicg_update_x_kernel(int64_t n, int64_t p, V* __restrict__ x, V* __restrict__ r,
                                                               const V* __restrict__ s, const V* __restrict__ tv,
                                                               const V* __restrict__ pv, const V* __restrict__ scal,
                                                               const int* __restrict__ flags, int lpr, int rpp,
                                                               V* __restrict__ partial, const V* __restrict__ zv) {
    __shared__ V red[kBlock * VEC];
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    const bool on = rs < rpp && c < p;
    V alpha[VEC], omega[VEC], acc[VEC];
    bool act[VEC], hf[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        acc[k] = 0;
        const bool f = on ? flags[2 + c + k] != 0 : true;
        hf[k] = on && !f && flags[2 + p + c + k] != 0;
        act[k] = on && !f && !hf[k];
        alpha[k] = on ? scal[p + c + k] : (V)0;
        omega[k] = on ? scal[2 * p + c + k] : (V)0;
    }
    const int64_t rb = (int64_t)blockIdx.x * rpp * kPasses;
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t row = rb + (int64_t)ps * rpp + rs;
        if (on && row < n) {
            const int64_t o = row * p + c;
            V xx[VEC], rr[VEC], ss[VEC], tt[VEC], pp[VEC], zz[VEC];
            load_vec<V, VEC>(x + o, xx);
            load_vec<V, VEC>(r + o, rr);
            load_vec<V, VEC>(s + o, ss);
            load_vec<V, VEC>(tv + o, tt);
            load_vec<V, VEC>(pv + o, pp);
            if constexpr (PRE) load_vec<V, VEC>(zv + o, zz);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const V ap = alpha[k] * pp[k];
                const V xh = xx[k] + ap;
                const V xa = (xx[k] + omega[k] * (PRE ? zz[k] : ss[k])) + ap;
                const V rn = ss[k] - omega[k] * tt[k];
                xx[k] = hf[k] ? xh : (act[k] ? xa : xx[k]);
                rr[k] = act[k] ? rn : rr[k];
                acc[k] = fma(rr[k], rr[k], acc[k]);
            }
            store_vec<V, VEC>(x + o, xx);
            store_vec<V, VEC>(r + o, rr);
        }
    }
    if (rs < rpp) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) red[(rs * lpr + cl) * VEC + k] = acc[k];
    }
    __syncthreads();
    for (int64_t cc = t; cc < p; cc += kBlock) {
        V sum = 0;
        for (int k = 0; k < rpp; ++k) sum += red[k * lpr * VEC + cc];
        partial[(int64_t)blockIdx.x * p + cc] = sum;
    }
}

}  // namespace tsgu

using namespace tsgu;

extern "C" {

int tsgu_bicg_scalar(int vtype, int phase, const void* partial, int64_t n_partial, int64_t set_stride, void* fold,
                     void* scal, int* flags, double abstol, double reltol, int matvec_max, int nmv0, int64_t p,
                     int device, void* stream) {
    if (!scal || !flags || p <= 0 || n_partial < 0 || phase < kBicgInit || phase > kBicgEnd) return TSGU_ERR_BAD_ARG;
    if (phase != kBicgBeta && !partial) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // one partial set with very many rows (the K1 epilogue writes one row per workgroup): fold it first
    const bool do_fold = fold != nullptr && phase != kBicgOmega && phase != kBicgBeta && n_partial > 4 * kFoldRows;
    const int64_t chunk = (n_partial + kFoldRows - 1) / kFoldRows;
#define TSGU_BODY                                                                                                  \
    {                                                                                                              \
        const V* src = (const V*)partial;                                                                          \
        int64_t rows = n_partial;                                                                                  \
        if (do_fold) {                                                                                             \
            hipLaunchKernelGGL((colsum_fold_kernel<V>), dim3(kFoldRows), dim3(kBlock), 0, s, src, n_partial, p,    \
                               chunk, (V*)fold, phase == kBicgInit ? (const int*)nullptr : (const int*)flags);     \
            if (const int rc = check_launch()) return rc;                                                          \
            src = (const V*)fold;                                                                                  \
            rows = kFoldRows;                                                                                      \
        }                                                                                                          \
        hipLaunchKernelGGL((bicg_scalar_kernel<V>), dim3(1), dim3(kBlock), 0, s, phase, src, rows, set_stride, p,  \
                           (V*)scal, flags, (V)abstol, (V)reltol, matvec_max, nmv0);                               \
        return check_launch();                                                                                     \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
    return TSGU_OK;
}

// which: 0 = update_p(pv, r, v)   1 = update_s(s, r, v -> partial)   2 = dots3(t, s, r0 -> partial[3])
//        3 = update_x(x, r, s, t, pv -> partial).  Arrays are contiguous [n][p], 16-byte aligned.
int tsgu_bicg_vector(int vtype, int which, int64_t n, int64_t p, void* a0, void* a1, const void* a2, const void* a3,
                     const void* a4, const void* scal, const int* flags, void* partial, int64_t set_stride,
                     int device, void* stream) {
    if (n <= 0 || p <= 0 || !a0 || !a1 || !a2 || !flags || which < 0 || which > 3) return TSGU_ERR_BAD_ARG;
    if (which != 2 && !scal) return TSGU_ERR_BAD_ARG;
    if (which != 0 && !partial) return TSGU_ERR_BAD_ARG;
    if (which == 3 && (!a3 || !a4)) return TSGU_ERR_BAD_ARG;
    if (!(aligned16(a0) && aligned16(a1) && aligned16(a2) && aligned16(a3) && aligned16(a4))) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TSGU_GO(KERNEL, ...)                                                                                       \
    do {                                                                                                           \
        if (g.vec == 1) hipLaunchKernelGGL((KERNEL<V, 1>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<V, wide>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, __VA_ARGS__);     \
    } while (0)
#define TSGU_BODY                                                                                                  \
    {                                                                                                              \
        constexpr int wide = VT<V>::kWide;                                                                         \
        VecGeom g;                                                                                                 \
        if (!geom_for<V>(n, p, true, g)) return TSGU_ERR_TOO_LARGE;                                                \
        if (which == 0)                                                                                            \
            TSGU_GO(bicg_update_p_kernel, n, p, (V*)a0, (const V*)a1, (const V*)a2, (const V*)scal, flags, g.lpr, g.rpp); \
        else if (which == 1)                                                                                       \
            TSGU_GO(bicg_update_s_kernel, n, p, (V*)a0, (const V*)a1, (const V*)a2, (const V*)scal, flags, g.lpr, g.rpp, \
                    (V*)partial);                                                                                  \
        else if (which == 2)                                                                                       \
            TSGU_GO(bicg_dots3_kernel, n, p, (const V*)a0, (const V*)a1, (const V*)a2, flags, g.lpr, g.rpp, (V*)partial, \
                    set_stride);                                                                                   \
        else                                                                                                       \
            TSGU_GO(bicg_update_x_kernel, n, p, (V*)a0, (V*)a1, (const V*)a2, (const V*)a3, (const V*)a4, (const V*)scal, \
                    flags, g.lpr, g.rpp, (V*)partial, (const V*)nullptr);                                          \
        return check_launch();                                                                                     \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
#undef TSGU_GO
    return TSGU_OK;
}

// The x / r update of a preconditioned iteration: r = s - omega*t; x = (x + omega*z) + alpha*q with q = M p, z = M s
// (columns finishing after the half step: x += alpha*q); partial |r|^2.  Arrays contiguous [n][p], 16-byte aligned.
int tsgu_bicg_update_x_precond(int vtype, int64_t n, int64_t p, void* x, void* r, const void* s_, const void* t,
                               const void* q, const void* z, const void* scal, const int* flags, void* partial,
                               int device, void* stream) {
    if (n <= 0 || p <= 0 || !x || !r || !s_ || !t || !q || !z || !scal || !flags || !partial) return TSGU_ERR_BAD_ARG;
    if (!(aligned16(x) && aligned16(r) && aligned16(s_) && aligned16(t) && aligned16(q) && aligned16(z))) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TSGU_BODY                                                                                                  \
    {                                                                                                              \
        constexpr int wide = VT<V>::kWide;                                                                         \
        VecGeom g;                                                                                                 \
        if (!geom_for<V>(n, p, true, g)) return TSGU_ERR_TOO_LARGE;                                                \
        if (g.vec == 1)                                                                                            \
            hipLaunchKernelGGL((bicg_update_x_kernel<V, 1, true>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n, p, (V*)x, \
                               (V*)r, (const V*)s_, (const V*)t, (const V*)q, (const V*)scal, flags, g.lpr, g.rpp,   \
                               (V*)partial, (const V*)z);                                                          \
        else                                                                                                       \
            hipLaunchKernelGGL((bicg_update_x_kernel<V, wide, true>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n, p, \
                               (V*)x, (V*)r, (const V*)s_, (const V*)t, (const V*)q, (const V*)scal, flags, g.lpr,   \
                               g.rpp, (V*)partial, (const V*)z);                                                   \
        return check_launch();                                                                                     \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
    return TSGU_OK;
}

}  // extern "C"
