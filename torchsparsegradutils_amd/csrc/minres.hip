// K7: fused MINRES recurrences on gfx950, every right-hand side (and every shift) at once.
//
// The reference (utils/minres.py:241-311) runs the Lanczos three-term recurrence and the Givens QR of the
// tridiagonal as ~40 small ATen ops per iteration.  The Lanczos vectors are shared by all shifts; each shift has its
// own rotations, w vectors and solution (arrays [n_shift][n][p]).  `value` scales the operator (value*A + shift*I).
// With a preconditioner the caller applies it between the Lanczos kernel and the Givens step (q_c = M z_c,
// beta_c = sqrt<z_c, q_c>) and hands q_c / q_prev to the update.  Without one q == z, the vector state is
// {z_prev2, z_prev, w_prev2, w_prev, sol} and one iteration is
//   K1 SpMM (+ <z, A z> partials)  ->  scalar(ALPHA)  ->  lanczos (z_c = A z - alpha z - beta z_prev2, |z_c|^2 partials)
//   ->  scalar(GIVENS: beta_c, rotations, sub / subsub / diag / scale)  ->  update (z_c /= beta_c, w_c, sol += w_c scale,
//   and on every 10th iteration |update|^2, |sol|^2 partials)  [->  scalar(STOP)]
// All per-column scalars live on the device; the host reads one word every 10 iterations (where the reference
// also synchronises for its stopping test, minres.py:299-305).  Reductions are two-stage in fixed order.
//
// scal [n_shift][12][p]: 0 alpha | 1 beta (beta_prev on entry of GIVENS, beta_cur after) | 2 c_prev2 | 3 s_prev2 | 4 c_prev |
//               5 s_prev | 6 scale_prev | 7 sub | 8 subsub | 9 diag | 10 scale used by this update | 11 beta_prev (lanczos);
//               rows 0, 1 and 11 are those of block 0 for every shift (the Lanczos scalars are shared)
// flags int32: [0] stop | [1] iterations done
#include "krylov_common.h"

namespace tsgu {

enum MinresPhase { kMrAlpha = 0, kMrGivens = 1, kMrStop = 2 };

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }

template <typename V>
__global__ __launch_bounds__(kBlock) void minres_scalar_kernel(int phase, const V* __restrict__ partial, int64_t n_partial,
                                                               int64_t set_stride, int64_t p, V* __restrict__ scal,
                                                               int* __restrict__ flags, V eps, V tol, V shift0,
                                                               const V* __restrict__ shifts, int n_shift, V value) {
    __shared__ V red[kBlock];
    __shared__ V ratio[kBlock];
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    V my_ratio = 0;
    for (int64_t c0 = 0; c0 < p; c0 += 64) {
        const int w = (int)(p - c0 < 64 ? p - c0 : 64);
        if (phase == kMrStop) {
            // |update| / |sol| per shift and column (minres.py:300-302); sets 2*sh and 2*sh+1
            for (int sh = 0; sh < n_shift; ++sh) {
                const V s0 = block_colsum<V>(partial + (2 * sh) * set_stride, n_partial, p, c0, w, red);
                const V s1 = block_colsum<V>(partial + (2 * sh + 1) * set_stride, n_partial, p, c0, w, red);
                if (t < w) my_ratio += sqrt(s0) / sqrt(s1);
            }
            continue;
        }
        const V s0 = block_colsum<V>(partial, n_partial, p, c0, w, red);
        if (t < w) {
            const int64_t c = c0 + t;
            if (phase == kMrAlpha) {
                scal[c] = value * s0;  // alpha = <value A q, q>   (minres.py:261-262)
            } else {
                const V alpha = scal[c];
                const V beta_p = scal[p + c];
                V beta_c = sqrt(s0);  // (minres.py:268-269)
                beta_c = beta_c < eps ? eps : beta_c;
                for (int sh = 0; sh < n_shift; ++sh) {
                    V* blk = scal + (int64_t)sh * 12 * p;
                    const V c_pp = blk[2 * p + c], s_pp = blk[3 * p + c], c_p = blk[4 * p + c], s_p = blk[5 * p + c];
                    const V scale_p = blk[6 * p + c];
                    // QR of the shifted tridiagonal (minres.py:274-285)
                    const V subsub = s_pp * beta_p;
                    V sub = c_pp * beta_p;
                    const V alpha_s = alpha + (shifts ? shifts[sh] : shift0);
                    V diag = alpha_s * c_p - s_p * sub;
                    sub = sub * c_p + s_p * alpha_s;
                    const V radius = sqrt(diag * diag + beta_c * beta_c);
                    const V c_c = diag / radius;
                    const V s_c = beta_c / radius;
                    diag = diag * c_c + s_c * beta_c;
                    // (minres.py:288-289)
                    const V scale_c = -(scale_p * s_c);
                    const V scale_use = scale_p * c_c;
                    blk[2 * p + c] = c_p;
                    blk[3 * p + c] = s_p;
                    blk[4 * p + c] = c_c;
                    blk[5 * p + c] = s_c;
                    blk[6 * p + c] = scale_c;
                    blk[7 * p + c] = sub;
                    blk[8 * p + c] = subsub;
                    blk[9 * p + c] = diag;
                    blk[10 * p + c] = scale_use;
                }
                scal[11 * p + c] = beta_c;  // beta_prev of the next Lanczos step
                scal[p + c] = beta_c;
            }
        }
    }
    if (phase == kMrStop) {
        ratio[t] = t < 64 ? my_ratio : (V)0;
        __syncthreads();
        if (t == 0) {
            V s = 0;
            const int lim = p < 64 ? (int)p : 64;
            for (int k = 0; k < lim; ++k) s += ratio[k];
            // mean over shifts and columns (minres.py:303-305); NaN compares false
            if (s / ((V)p * (V)n_shift) < tol) flags[0] = 1;
        }
    } else if (phase == kMrGivens && t == 0) {
        flags[1] += 1;
    }
}

// z_c = (value A q - alpha z) - beta_prev z_prev2, written over z_prev2; partial |z_c|^2      (minres.py:261-268)
template <typename V, int VEC>
__global__ __launch_bounds__(kBlock) void minres_lanczos_kernel(int64_t n, int64_t p, V* __restrict__ zpp, const V* __restrict__ zp,
                                                                const V* __restrict__ prod, const V* __restrict__ scal,
                                                                const int* __restrict__ flags, int lpr, int rpp,
                                                                V* __restrict__ partial, V value) {
    __shared__ V red[kBlock * VEC];
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    const bool on = rs < rpp && c < p;
    V alpha[VEC], beta[VEC], acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        acc[k] = 0;
        alpha[k] = on ? scal[c + k] : (V)0;
        beta[k] = on ? scal[11 * p + c + k] : (V)0;
    }
    const int64_t r0 = (int64_t)blockIdx.x * rpp * kPasses;
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t row = r0 + (int64_t)ps * rpp + rs;
        if (on && row < n) {
            const int64_t o = row * p + c;
            V a[VEC], b[VEC], pr[VEC];
            load_vec<V, VEC>(zpp + o, a);
            load_vec<V, VEC>(zp + o, b);
            load_vec<V, VEC>(prod + o, pr);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                a[k] = (mul_rn(value, pr[k]) - alpha[k] * b[k]) - beta[k] * a[k];  // value*A q rounded as the reference's .mul(value)
                acc[k] = fma(a[k], a[k], acc[k]);
            }
            store_vec<V, VEC>(zpp + o, a);
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

// z_c /= beta_c (in place; also q_c when a preconditioner produced one);  per shift: w_c = ((q_p - sub w_p) - subsub w_pp) / diag
// written over w_pp;  sol += w_c * scale;  with_norms: partial |w_c scale|^2 (set 2*shift) and |sol|^2 (set 2*shift+1)
// q_p is z_p without a preconditioner.  w / sol: n_shift planes [n][p], `np` elements apart.   (minres.py:270-271, 290-302)
template <typename V, int VEC>
__global__ __launch_bounds__(kBlock) void minres_update_kernel(int64_t n, int64_t p, V* __restrict__ zc, const V* __restrict__ qp,
                                                               V* __restrict__ wpp, const V* __restrict__ wp, V* __restrict__ sol,
                                                               const V* __restrict__ scal, const int* __restrict__ flags, int lpr,
                                                               int rpp, V* __restrict__ partial, int64_t set_stride, int with_norms,
                                                               V* __restrict__ qc, int n_shift, int64_t np) {
    __shared__ V red[kBlock * VEC];
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    const bool on = rs < rpp && c < p;
    const int64_t r0 = (int64_t)blockIdx.x * rpp * kPasses;
    for (int sh = 0; sh < n_shift; ++sh) {
        const V* blk = scal + (int64_t)sh * 12 * p;
        V* wpp_s = wpp + sh * np;
        const V* wp_s = wp + sh * np;
        V* sol_s = sol + sh * np;
        V beta[VEC], sub[VEC], subsub[VEC], diag[VEC], scale[VEC], au[VEC], as[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            au[k] = as[k] = 0;
            beta[k] = on ? scal[p + c + k] : (V)1;
            sub[k] = on ? blk[7 * p + c + k] : (V)0;
            subsub[k] = on ? blk[8 * p + c + k] : (V)0;
            diag[k] = on ? blk[9 * p + c + k] : (V)1;
            scale[k] = on ? blk[10 * p + c + k] : (V)0;
        }
#pragma unroll
        for (int ps = 0; ps < kPasses; ++ps) {
            const int64_t row = r0 + (int64_t)ps * rpp + rs;
            if (on && row < n) {
                const int64_t o = row * p + c;
                V q[VEC], w2[VEC], w1[VEC], x[VEC];
                if (sh == 0) {
                    V z[VEC];
                    load_vec<V, VEC>(zc + o, z);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) z[k] = z[k] / beta[k];
                    store_vec<V, VEC>(zc + o, z);
                    if (qc) {
                        load_vec<V, VEC>(qc + o, z);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) z[k] = z[k] / beta[k];
                        store_vec<V, VEC>(qc + o, z);
                    }
                }
                load_vec<V, VEC>(qp + o, q);
                load_vec<V, VEC>(wpp_s + o, w2);
                load_vec<V, VEC>(wp_s + o, w1);
                load_vec<V, VEC>(sol_s + o, x);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const V wc = ((q[k] - sub[k] * w1[k]) - subsub[k] * w2[k]) / diag[k];
                    const V up = wc * scale[k];
                    w2[k] = wc;
                    x[k] = x[k] + up;
                    au[k] = fma(up, up, au[k]);
                    as[k] = fma(x[k], x[k], as[k]);
                }
                store_vec<V, VEC>(wpp_s + o, w2);
                store_vec<V, VEC>(sol_s + o, x);
            }
        }
        if (!with_norms) continue;
        for (int set = 0; set < 2; ++set) {
            if (rs < rpp) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) red[(rs * lpr + cl) * VEC + k] = set == 0 ? au[k] : as[k];
            }
            __syncthreads();
            for (int64_t cc = t; cc < p; cc += kBlock) {
                V sum = 0;
                for (int k = 0; k < rpp; ++k) sum += red[k * lpr * VEC + cc];
                partial[(2 * sh + set) * set_stride + (int64_t)blockIdx.x * p + cc] = sum;
            }
            __syncthreads();
        }
    }
}

}  // namespace tsgu

using namespace tsgu;

namespace {

int minres_scalar_launch(int vtype, int phase, const void* partial, int64_t n_partial, int64_t set_stride, void* fold, void* scal,
                         int* flags, double eps, double tol, double shift0, const void* shifts, int n_shift, double value,
                         int64_t p, int device, void* stream) {
    if (!scal || !flags || !partial || p <= 0 || p > 1024 || n_partial < 0 || phase < kMrAlpha || phase > kMrStop || n_shift < 1)
        return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // one partial set with very many rows (the K1 epilogue writes one row per workgroup): fold it first
    const bool do_fold = fold != nullptr && phase != kMrStop && n_partial > 4 * kFoldRows;
    const int64_t chunk = (n_partial + kFoldRows - 1) / kFoldRows;
#define TSGU_BODY                                                                                                  \
    {                                                                                                              \
        const V* src = (const V*)partial;                                                                          \
        int64_t rows = n_partial;                                                                                  \
        if (do_fold) {                                                                                             \
            hipLaunchKernelGGL((colsum_fold_kernel<V>), dim3(kFoldRows), dim3(kBlock), 0, s, src, n_partial, p,    \
                               chunk, (V*)fold, (const int*)flags);                                                \
            if (const int rc = check_launch()) return rc;                                                          \
            src = (const V*)fold;                                                                                  \
            rows = kFoldRows;                                                                                      \
        }                                                                                                          \
        hipLaunchKernelGGL((minres_scalar_kernel<V>), dim3(1), dim3(kBlock), 0, s, phase, src, rows, set_stride, p, \
                           (V*)scal, flags, (V)eps, (V)tol, (V)shift0, (const V*)shifts, n_shift, (V)value);       \
        return check_launch();                                                                                     \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
    return TSGU_OK;
}

int minres_vector_launch(int vtype, int which, int64_t n, int64_t p, void* a0, const void* a1, void* a2, const void* a3, void* a4,
                         void* qc, const void* scal, const int* flags, void* partial, int64_t set_stride, int with_norms,
                         int n_shift, int64_t shift_stride, double value, int device, void* stream) {
    if (n <= 0 || p <= 0 || !a0 || !a1 || !a2 || !scal || !flags || which < 0 || which > 1 || n_shift < 1) return TSGU_ERR_BAD_ARG;
    if (which == 0 && !partial) return TSGU_ERR_BAD_ARG;
    if (which == 1 && (!a3 || !a4 || (with_norms && !partial))) return TSGU_ERR_BAD_ARG;
    if (!(aligned16(a0) && aligned16(a1) && aligned16(a2) && aligned16(a3) && aligned16(a4) && aligned16(qc))) return TSGU_ERR_BAD_ARG;
    if (which == 1 && n_shift > 1 && (shift_stride < n * p || ((shift_stride * (int64_t)(vtype == TSGU_F64 ? 8 : 4)) & 15)))
        return TSGU_ERR_BAD_ARG;
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
            TSGU_GO(minres_lanczos_kernel, n, p, (V*)a0, (const V*)a1, (const V*)a2, (const V*)scal, flags, g.lpr, g.rpp, \
                    (V*)partial, (V)value);                                                                        \
        else                                                                                                       \
            TSGU_GO(minres_update_kernel, n, p, (V*)a0, (const V*)a1, (V*)a2, (const V*)a3, (V*)a4, (const V*)scal, flags, \
                    g.lpr, g.rpp, (V*)partial, set_stride, with_norms, (V*)qc, n_shift, shift_stride);             \
        return check_launch();                                                                                     \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
#undef TSGU_GO
    return TSGU_OK;
}

}  // namespace

extern "C" {

int tsgu_minres_scalar(int vtype, int phase, const void* partial, int64_t n_partial, int64_t set_stride, void* fold,
                       void* scal, int* flags, double eps, double tol, double shift, int64_t p, int device, void* stream) {
    return minres_scalar_launch(vtype, phase, partial, n_partial, set_stride, fold, scal, flags, eps, tol, shift, nullptr, 1, 1.0,
                                p, device, stream);
}

// which: 0 = lanczos(z_prev2 <- z_c, z_prev, prod -> partial)   1 = update(z_c, z_prev, w_prev2 <- w_c, w_prev, sol
//        [-> partial[2] when with_norms]).  Arrays are contiguous [n][p], 16-byte aligned.
int tsgu_minres_vector(int vtype, int which, int64_t n, int64_t p, void* a0, const void* a1, void* a2, const void* a3,
                       void* a4, const void* scal, const int* flags, void* partial, int64_t set_stride, int with_norms,
                       int device, void* stream) {
    return minres_vector_launch(vtype, which, n, p, a0, a1, a2, a3, a4, nullptr, scal, flags, partial, set_stride, with_norms, 1,
                                n * p, 1.0, device, stream);
}

// Several shifts at once, a `value` factor on the operator, and the preconditioned form.
//   scalar: `shifts` = n_shift device values of the value type; phase 0 stores alpha = value * <q, A q>; phase 1 takes
//           |z_c|^2 partials or, preconditioned, <z_c, M z_c>; phase 2 reads 2*n_shift partial sets.
//   vector: which 0: a0 = z_prev2 (<- z_c), a1 = z_prev, a2 = A q (unscaled);  which 1: a0 = z_c, a1 = q_prev (z_prev when
//           there is no preconditioner), a2 = w_prev2 (<- w_c), a3 = w_prev, a4 = sol: n_shift planes [n][p], `shift_stride`
//           elements apart (a multiple of 16 bytes), qc = M z_c to normalise or NULL.
int tsgu_minres_scalar_ms(int vtype, int phase, const void* partial, int64_t n_partial, int64_t set_stride, void* fold,
                          void* scal, int* flags, double eps, double tol, const void* shifts, int n_shift, double value,
                          int64_t p, int device, void* stream) {
    if (!shifts) return TSGU_ERR_BAD_ARG;
    return minres_scalar_launch(vtype, phase, partial, n_partial, set_stride, fold, scal, flags, eps, tol, 0.0, shifts, n_shift,
                                value, p, device, stream);
}

int tsgu_minres_vector_ms(int vtype, int which, int64_t n, int64_t p, void* a0, const void* a1, void* a2, const void* a3,
                          void* a4, void* qc, const void* scal, const int* flags, void* partial, int64_t set_stride,
                          int with_norms, int n_shift, int64_t shift_stride, double value, int device, void* stream) {
    return minres_vector_launch(vtype, which, n, p, a0, a1, a2, a3, a4, qc, scal, flags, partial, set_stride, with_norms, n_shift,
                                shift_stride, value, device, stream);
}

}  // extern "C"
