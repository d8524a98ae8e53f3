// K5: fused vector updates of the Krylov loops (CG recurrences, column dots) on gfx950.
// The reference runs ~15 elementwise/reduce ATen ops and a host sync per CG iteration
// (utils/linear_cg.py:27-95, 372-382); here one iteration is
//   K1(+pᵀAp epilogue) → cg_alpha → cg_update1 → cg_beta → cg_update2
// with every per-column scalar and the convergence decision kept on the device.
// All reductions are two-stage with a fixed summation order (no atomics): deterministic.
#include "krylov_common.h"

namespace tsgu {

// ---- generic column dot --------------------------------------------------------------
template <typename V, int VEC>
__global__ __launch_bounds__(kBlock) void coldot_partial_kernel(int64_t n, int64_t p, const V* __restrict__ X, int64_t ldx,
                                                                const V* __restrict__ Y, int64_t ldy, int lpr, int rpp,
                                                                V* __restrict__ partial) {
    __shared__ V red[kBlock * VEC];
    const int t = threadIdx.x;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    const bool act = rs < rpp && c < p;
    V acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    const int64_t r0 = (int64_t)blockIdx.x * rpp * kPasses;
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t r = r0 + (int64_t)ps * rpp + rs;
        if (act && r < n) {
            V x[VEC], y[VEC];
            load_vec<V, VEC>(X + r * ldx + c, x);
            load_vec<V, VEC>(Y + r * ldy + c, y);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = fma(x[v], y[v], acc[v]);
        }
    }
    if (rs < rpp) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[(rs * lpr + cl) * VEC + v] = acc[v];
    }
    __syncthreads();
    for (int64_t cc = t; cc < p; cc += kBlock) {
        V s = 0;
        for (int k = 0; k < rpp; ++k) s += red[k * lpr * VEC + cc];
        partial[(int64_t)blockIdx.x * p + cc] = s;
    }
}

template <typename V>
__global__ __launch_bounds__(kBlock) void colsum_finalize_kernel(const V* __restrict__ partial, int64_t n_partial, int64_t p,
                                                                 V* __restrict__ out) {
    __shared__ V red[kBlock];
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const int w = (int)(p - c0 < 64 ? p - c0 : 64);
    const V tot = block_colsum<V>(partial, n_partial, p, c0, w, red);
    if ((int)threadIdx.x < w) out[c0 + threadIdx.x] = tot;
}

// ---- CG ---------------------------------------------------------------------------------
// scal: [rr | alpha | beta | rnorm] each [p];  flags: [done, iters, has_converged[p], rhs_is_zero[p]]
template <typename V>
__global__ __launch_bounds__(kBlock) void cg_alpha_kernel(const V* __restrict__ pap_partial, int64_t n_partial, int64_t p,
                                                          V* __restrict__ scal, const int* __restrict__ flags, V eps) {
    __shared__ V red[kBlock];
    if (flags[0] != 0) return;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const int w = (int)(p - c0 < 64 ? p - c0 : 64);
    const V pap = block_colsum<V>(pap_partial, n_partial, p, c0, w, red);
    if ((int)threadIdx.x < w) {
        const int64_t c = c0 + threadIdx.x;
        // safe division (linear_cg.py:67-71) then freeze converged columns (:74)
        V alpha = pap < eps ? (V)0 : scal[c] / pap;
        if (flags[2 + c] != 0) alpha = 0;
        scal[p + c] = alpha;
    }
}

template <typename V, int VEC>
__global__ __launch_bounds__(kBlock) void cg_update1_kernel(int64_t n, int64_t p, V* __restrict__ r, const V* __restrict__ Ap,
                                                            V* __restrict__ x, const V* __restrict__ pv,
                                                            const V* __restrict__ scal, const int* __restrict__ flags,
                                                            int lpr, int rpp, V* __restrict__ rr_partial) {
    __shared__ V red[kBlock * VEC];
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    const bool act = rs < rpp && c < p;
    V alpha[VEC], acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        acc[v] = 0;
        alpha[v] = act ? scal[p + c + v] : (V)0;
    }
    const int64_t r0 = (int64_t)blockIdx.x * rpp * kPasses;
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t row = r0 + (int64_t)ps * rpp + rs;
        if (act && row < n) {
            const int64_t o = row * p + c;
            V rv[VEC], av[VEC], xv[VEC], pvv[VEC];
            load_vec<V, VEC>(r + o, rv);
            load_vec<V, VEC>(Ap + o, av);
            load_vec<V, VEC>(x + o, xv);
            load_vec<V, VEC>(pv + o, pvv);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                rv[v] = fma(-alpha[v], av[v], rv[v]);   // r -= alpha·Ap   (linear_cg.py:78)
                xv[v] = fma(alpha[v], pvv[v], xv[v]);   // x += alpha·p    (linear_cg.py:32)
                acc[v] = fma(rv[v], rv[v], acc[v]);     // rᵀr             (linear_cg.py:36-37)
            }
            store_vec<V, VEC>(r + o, rv);
            store_vec<V, VEC>(x + o, xv);
        }
    }
    if (rs < rpp) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[(rs * lpr + cl) * VEC + v] = acc[v];
    }
    __syncthreads();
    for (int64_t cc = t; cc < p; cc += kBlock) {
        V s = 0;
        for (int k = 0; k < rpp; ++k) s += red[k * lpr * VEC + cc];
        rr_partial[(int64_t)blockIdx.x * p + cc] = s;
    }
}

// Steps 1 + 2 in one launch when K1 left few partial rows (the plane sweep: one per workgroup): EVERY workgroup of the update sums
// the partial rows itself — the same rows in the same order, so all of them hold the same alpha bit for bit — instead of waiting
// for a single-workgroup kernel and a launch boundary (4.9 + 2.2 us of a 98 us iteration at C4).  n_partial x p values are read
// from L2 per workgroup: 8 KB at C4.
template <typename V, int VEC>
__global__ __launch_bounds__(kBlock) void cg_update1_alpha_kernel(int64_t n, int64_t p, V* __restrict__ r, const V* __restrict__ Ap,
                                                                  V* __restrict__ x, const V* __restrict__ pv,
                                                                  const V* __restrict__ pap_partial, int64_t n_partial, V* __restrict__ scal,
                                                                  const int* __restrict__ flags, V eps, int lpr, int rpp,
                                                                  V* __restrict__ rr_partial) {
    __shared__ V red[kBlock * VEC];
    __shared__ V alpha_s[kBlock];        // p <= kBlock columns (checked by the launcher)
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    for (int64_t c0 = 0; c0 < p; c0 += 64) {
        const int w = (int)(p - c0 < 64 ? p - c0 : 64);
        const V pap = block_colsum<V>(pap_partial, n_partial, p, c0, w, red);
        if (t < w) {
            const int64_t c = c0 + t;
            V a = pap < eps ? (V)0 : scal[c] / pap;          // safe division (linear_cg.py:67-71)
            if (flags[2 + c] != 0) a = 0;                     // converged columns are frozen (:74)
            alpha_s[c] = a;
            if (blockIdx.x == 0) scal[p + c] = a;             // (kept in the state for diagnostics; nobody reads it in this launch)
        }
        __syncthreads();
    }
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    const bool act = rs < rpp && c < p;
    V alpha[VEC], acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        acc[v] = 0;
        alpha[v] = act ? alpha_s[c + v] : (V)0;
    }
    const int64_t r0 = (int64_t)blockIdx.x * rpp * kPasses;
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t row = r0 + (int64_t)ps * rpp + rs;
        if (act && row < n) {
            const int64_t o = row * p + c;
            V rv[VEC], av[VEC], xv[VEC], pvv[VEC];
            load_vec<V, VEC>(r + o, rv);
            load_vec<V, VEC>(Ap + o, av);
            load_vec<V, VEC>(x + o, xv);
            load_vec<V, VEC>(pv + o, pvv);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                rv[v] = fma(-alpha[v], av[v], rv[v]);   // r -= alpha·Ap   (linear_cg.py:78)
                xv[v] = fma(alpha[v], pvv[v], xv[v]);   // x += alpha·p    (linear_cg.py:32)
                acc[v] = fma(rv[v], rv[v], acc[v]);     // rᵀr             (linear_cg.py:36-37)
            }
            store_vec<V, VEC>(r + o, rv);
            store_vec<V, VEC>(x + o, xv);
        }
    }
    __syncthreads();
    if (rs < rpp) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[(rs * lpr + cl) * VEC + v] = acc[v];
    }
    __syncthreads();
    for (int64_t cc = t; cc < p; cc += kBlock) {
        V s = 0;
        for (int k = 0; k < rpp; ++k) s += red[k * lpr * VEC + cc];
        rr_partial[(int64_t)blockIdx.x * p + cc] = s;
    }
}

template <typename V>
__global__ __launch_bounds__(kBlock) void cg_beta_kernel(const V* __restrict__ rr_partial, int64_t n_partial, int64_t p,
                                                         V* __restrict__ scal, int* __restrict__ flags, V eps, V stop_after,
                                                         V tolerance, int iter_index, int min_iter_index,
                                                         const V* __restrict__ rz_partial, int64_t n_rz) {
    // single block: needs the mean of the residual norms over ALL columns for the stop test.
    __shared__ V red[kBlock];
    __shared__ V normsum[kBlock];
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    V my_norm_sum = 0;
    for (int64_t c0 = 0; c0 < p; c0 += 64) {
        const int w = (int)(p - c0 < 64 ? p - c0 : 64);
        const V rr_new = block_colsum<V>(rr_partial, n_partial, p, c0, w, red);
        // preconditioned: the recurrences run on <r, z> (z = M r, linear_cg.py:80-84), the stop test still on |r|
        const V ip_new = rz_partial ? block_colsum<V>(rz_partial, n_rz, p, c0, w, red) : rr_new;
        if (t < w) {
            const int64_t c = c0 + t;
            const V rr_old = scal[c];
            // beta = <r,z>_new / <r,z>_old with the reference's safe division (linear_cg.py:35-43)
            const V beta = rr_old < eps ? (V)0 : ip_new / rr_old;
            scal[c] = ip_new;
            scal[2 * p + c] = beta;
            V nrm = sqrt(rr_new);                       // ‖r‖₂ (linear_cg.py:372)
            if (flags[2 + p + c] != 0) nrm = 0;         // rhs_is_zero mask (:373)
            scal[3 * p + c] = nrm;
            flags[2 + c] = nrm < stop_after ? 1 : 0;    // has_converged (:374)
            my_norm_sum += nrm;
        }
    }
    normsum[t] = t < 64 ? my_norm_sum : (V)0;
    __syncthreads();
    if (t == 0) {
        V s = 0;
        const int lim = p < 64 ? (int)p : 64;
        for (int k = 0; k < lim; ++k) s += normsum[k];
        const V mean = s / (V)p;
        // iter_index < 0: the iteration counter lives on the device (flags[1]), so that every iteration is the
        // same launch and a chunk of iterations can be replayed as one hipGraph.
        const int it = iter_index < 0 ? flags[1] : iter_index;
        flags[1] = it + 1;
        // stop rule (linear_cg.py:376-382): k >= min(10, max_iter-1) and mean‖r‖ < tol
        if (it >= min_iter_index && mean < tolerance) flags[0] = 1;
    }
}

template <typename V, int VEC>
__global__ __launch_bounds__(kBlock) void cg_update2_kernel(int64_t n, int64_t p, const V* __restrict__ r, V* __restrict__ pv,
                                                            const V* __restrict__ scal, const int* __restrict__ flags,
                                                            int lpr, int rpp) {
    if (flags[0] != 0) return;
    const int t = threadIdx.x;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    if (!(rs < rpp && c < p)) return;
    V beta[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) beta[v] = scal[2 * p + c + v];
    const int64_t r0 = (int64_t)blockIdx.x * rpp * kPasses;
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t row = r0 + (int64_t)ps * rpp + rs;
        if (row < n) {
            const int64_t o = row * p + c;
            V rv[VEC], pvv[VEC];
            load_vec<V, VEC>(r + o, rv);
            load_vec<V, VEC>(pv + o, pvv);
#pragma unroll
            for (int v = 0; v < VEC; ++v) pvv[v] = fma(pvv[v], beta[v], rv[v]);  // p = r + beta·p (linear_cg.py:47)
            store_vec<V, VEC>(pv + o, pvv);
        }
    }
}

// ---- CG, two launches after K1 (state in two halves that alternate by iteration) ---------------------------------------------
// An iteration of the four-step form above ends with a single-workgroup kernel (beta, norms, stop rule: 5.9 us + a launch boundary
// of a 95 us iteration at C4) between two streaming kernels.  Here EVERY workgroup of the direction update sums the |r|^2 partial
// rows itself (as cg_update1_alpha does for alpha), so that kernel disappears — which needs the per-iteration state to be
// immutable during a launch: the values an iteration reads (rr, has_converged, done) live in half `par` = iteration parity, the
// values it produces go to half `par ^ 1`; only workgroup 0 writes.  x += alpha·p moves from the residual update to the direction
// update, which reads p anyway (32 MB less per iteration at C4).
//   scal2:  [rr half 0 | rr half 1 | alpha | beta | rnorm] each [p]
//   flags2: [done half 0, done half 1, iterations, unused, has_converged half 0 [p], half 1 [p], rhs_is_zero [p]]
template <typename V, int VEC, int GROUPS>      // GROUPS > 0: that many groups of kPasses row passes, all loads issued before the sums
__global__ __launch_bounds__(kBlock) void cg_residual_alpha_kernel(int64_t n, int64_t p, V* __restrict__ r, const V* __restrict__ Ap,
                                                                   const V* __restrict__ pap_partial, int64_t n_partial,
                                                                   V* __restrict__ scal, const int* __restrict__ flags, int par, V eps,
                                                                   int lpr, int rpp, int groups, V* __restrict__ rr_partial) {
    __shared__ V red[kBlock * VEC];
    __shared__ V alpha_s[kBlock];        // p <= kBlock columns (checked by the launcher)
    if (flags[par] != 0) return;
    const int t = threadIdx.x;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    const bool act = rs < rpp && c < p;
    const int64_t r0 = (int64_t)blockIdx.x * rpp * kPasses * groups;
    constexpr int NP = GROUPS > 0 ? GROUPS * kPasses : 1;
    V rv[NP][VEC], av[NP][VEC];
    if constexpr (GROUPS > 0) {
        // the streaming loads go out first: the sum of the partial rows below (two dependent trips to L2) then overlaps them
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int64_t row = r0 + (int64_t)ps * rpp + rs;
            if (act && row < n) {
                load_vec<V, VEC>(r + row * p + c, rv[ps]);
                load_vec<V, VEC>(Ap + row * p + c, av[ps]);
            }
        }
    }
    for (int64_t c0 = 0; c0 < p; c0 += 64) {
        const int w = (int)(p - c0 < 64 ? p - c0 : 64);
        const V pap = block_colsum<V>(pap_partial, n_partial, p, c0, w, red);
        if (t < w) {
            const int64_t cc = c0 + t;
            V a = pap < eps ? (V)0 : scal[(int64_t)par * p + cc] / pap;         // safe division (linear_cg.py:67-71)
            if (flags[4 + (int64_t)par * p + cc] != 0) a = 0;                   // converged columns are frozen (:74)
            alpha_s[cc] = a;
            if (blockIdx.x == 0) scal[2 * p + cc] = a;                          // for the direction update (the next launch)
        }
        __syncthreads();
    }
    V alpha[VEC], acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        acc[v] = 0;
        alpha[v] = act ? alpha_s[c + v] : (V)0;
    }
    if constexpr (GROUPS > 0) {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int64_t row = r0 + (int64_t)ps * rpp + rs;
            if (act && row < n) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    rv[ps][v] = fma(-alpha[v], av[ps][v], rv[ps][v]);   // r -= alpha·Ap   (linear_cg.py:78)
                    acc[v] = fma(rv[ps][v], rv[ps][v], acc[v]);         // rᵀr             (linear_cg.py:36-37)
                }
                store_vec<V, VEC>(r + row * p + c, rv[ps]);
            }
        }
    } else {
        for (int g = 0; g < groups; ++g) {
#pragma unroll
            for (int ps = 0; ps < kPasses; ++ps) {
                const int64_t row = r0 + ((int64_t)g * kPasses + ps) * rpp + rs;
                if (act && row < n) {
                    const int64_t o = row * p + c;
                    load_vec<V, VEC>(r + o, rv[0]);
                    load_vec<V, VEC>(Ap + o, av[0]);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        rv[0][v] = fma(-alpha[v], av[0][v], rv[0][v]);
                        acc[v] = fma(rv[0][v], rv[0][v], acc[v]);
                    }
                    store_vec<V, VEC>(r + o, rv[0]);
                }
            }
        }
    }
    __syncthreads();
    if (rs < rpp) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[(rs * lpr + cl) * VEC + v] = acc[v];
    }
    __syncthreads();
    for (int64_t cc = t; cc < p; cc += kBlock) {
        V s = 0;
        for (int k = 0; k < rpp; ++k) s += red[k * lpr * VEC + cc];
        rr_partial[(int64_t)blockIdx.x * p + cc] = s;
    }
}

template <typename V, int VEC>
__global__ __launch_bounds__(kBlock) void cg_direction_beta_kernel(int64_t n, int64_t p, const V* __restrict__ r, V* __restrict__ pv,
                                                                   V* __restrict__ x, const V* __restrict__ rr_partial, int64_t n_partial,
                                                                   V* __restrict__ scal, int* __restrict__ flags, int par, V eps,
                                                                   V stop_after, V tolerance, int min_iter_index, int lpr, int rpp,
                                                                   V* __restrict__ hist, int n_hist) {
    __shared__ V red[kBlock];
    __shared__ V alpha_s[kBlock], beta_s[kBlock];
    __shared__ V normsum[64];
    const int t = threadIdx.x;
    if (flags[par] != 0) {                 // finished in an earlier iteration: the flag follows the iterations to the other half
        if (blockIdx.x == 0 && t == 0) flags[par ^ 1] = 1;
        return;
    }
    const bool first = blockIdx.x == 0;
    const int cl = t % lpr, rs = t / lpr;
    const int64_t c = (int64_t)cl * VEC;
    const bool act = rs < rpp && c < p;
    const int64_t r0 = (int64_t)blockIdx.x * rpp * kPasses;
    // the streaming loads go out first: the sum of the partial rows (dependent trips to L2) overlaps them
    V rv[kPasses][VEC], pvv[kPasses][VEC], xv[kPasses][VEC];
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t row = r0 + (int64_t)ps * rpp + rs;
        if (act && row < n) {
            const int64_t o = row * p + c;
            load_vec<V, VEC>(r + o, rv[ps]);
            load_vec<V, VEC>(pv + o, pvv[ps]);
            load_vec<V, VEC>(x + o, xv[ps]);
        }
    }
    V my_norm_sum = 0;
    for (int64_t c0 = 0; c0 < p; c0 += 64) {
        const int w = (int)(p - c0 < 64 ? p - c0 : 64);
        const V rr_new = block_colsum<V>(rr_partial, n_partial, p, c0, w, red);
        if (t < w) {
            const int64_t c = c0 + t;
            const V rr_old = scal[(int64_t)par * p + c];
            const V beta = rr_old < eps ? (V)0 : rr_new / rr_old;     // safe division (linear_cg.py:35-43)
            alpha_s[c] = scal[2 * p + c];
            beta_s[c] = beta;
            if (first) {
                // the coefficients of the first n_hist iterations, for the Lanczos tridiagonal matrices (linear_cg.py:385-406):
                // hist[it][0][c] = alpha, hist[it][1][c] = beta
                if (hist != nullptr && flags[2] < n_hist) {
                    hist[((int64_t)flags[2] * 2 + 0) * p + c] = scal[2 * p + c];
                    hist[((int64_t)flags[2] * 2 + 1) * p + c] = beta;
                }
                scal[(int64_t)(par ^ 1) * p + c] = rr_new;
                scal[3 * p + c] = beta;
                V nrm = sqrt(rr_new);                                  // |r|_2 (linear_cg.py:372)
                if (flags[4 + 2 * p + c] != 0) nrm = 0;                // rhs_is_zero mask (:373)
                scal[4 * p + c] = nrm;
                flags[4 + (int64_t)(par ^ 1) * p + c] = nrm < stop_after ? 1 : 0;     // has_converged (:374)
                my_norm_sum += nrm;
            }
        }
    }
    if (first) {
        if (t < 64) normsum[t] = my_norm_sum;
        __syncthreads();
        if (t == 0) {
            V s = 0;
            const int lim = p < 64 ? (int)p : 64;
            for (int k = 0; k < lim; ++k) s += normsum[k];
            const V mean = s / (V)p;
            const int it = flags[2];
            flags[2] = it + 1;
            // stop rule (linear_cg.py:376-382): k >= min(10, max_iter-1) and mean|r| < tol
            flags[par ^ 1] = (it >= min_iter_index && mean < tolerance) ? 1 : 0;
        }
    }
    __syncthreads();
    if (!act) return;
    V alpha[VEC], beta[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        alpha[v] = alpha_s[c + v];
        beta[v] = beta_s[c + v];
    }
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
        const int64_t row = r0 + (int64_t)ps * rpp + rs;
        if (row < n) {
            const int64_t o = row * p + c;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                xv[ps][v] = fma(alpha[v], pvv[ps][v], xv[ps][v]);      // x += alpha·p    (linear_cg.py:32) — with the direction alpha belongs to
                pvv[ps][v] = fma(pvv[ps][v], beta[v], rv[ps][v]);      // p = r + beta·p  (linear_cg.py:47)
            }
            store_vec<V, VEC>(x + o, xv[ps]);
            store_vec<V, VEC>(pv + o, pvv[ps]);
        }
    }
}

}  // namespace tsgu

using namespace tsgu;

extern "C" {

int64_t tsgu_cg_fold_rows(void) { return kFoldRows; }

int64_t tsgu_cg_num_blocks(int vtype, int64_t n, int64_t p) {
    // exact block count of tsgu_cg_update1 (operands must be 16-byte aligned, contiguous)
    VecGeom g;
    const bool ok = vtype == TSGU_F64 ? geom_for<double>(n, p, true, g) : geom_for<float>(n, p, true, g);
    return ok ? g.blocks : -1;
}

int64_t tsgu_coldot_max_blocks(int64_t n, int64_t p) {
    // upper bound on the partial rows tsgu_coldot writes (scalar-lane geometry)
    VecGeom g;
    if (!vec_geom(1, false, n, p, g)) return -1;
    return g.blocks > 0 ? g.blocks : 1;
}

int tsgu_coldot(int vtype, int64_t n, int64_t p, const void* X, int64_t ldx, const void* Y, int64_t ldy,
                void* partial, void* out, int device, void* stream) {
    if (n < 0 || p <= 0 || !X || !Y || !partial || !out || ldx < p || ldy < p) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TSGU_COLDOT_BODY                                                                                          \
    {                                                                                                             \
        constexpr int wide = VT<V>::kWide;                                                                        \
        VecGeom g;                                                                                                \
        const bool can = aligned16(X) && aligned16(Y) && ldx % wide == 0 && ldy % wide == 0 && p % wide == 0;    \
        if (!vec_geom(wide, can, n, p, g)) return TSGU_ERR_TOO_LARGE;                                             \
        if (g.blocks == 0) g.blocks = 1;                                                                          \
        if (g.vec == 1)                                                                                           \
            hipLaunchKernelGGL((coldot_partial_kernel<V, 1>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n, p, \
                               (const V*)X, ldx, (const V*)Y, ldy, g.lpr, g.rpp, (V*)partial);                   \
        else                                                                                                      \
            hipLaunchKernelGGL((coldot_partial_kernel<V, wide>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n, \
                               p, (const V*)X, ldx, (const V*)Y, ldy, g.lpr, g.rpp, (V*)partial);                \
        if (const int rc = check_launch()) return rc;                                                             \
        hipLaunchKernelGGL((colsum_finalize_kernel<V>), dim3((unsigned)((p + 63) / 64)), dim3(kBlock), 0, s,      \
                           (const V*)partial, g.blocks, p, (V*)out);                                              \
        return check_launch();                                                                                    \
    }
    TSGU_VSWITCH(vtype, TSGU_COLDOT_BODY, TSGU_COLDOT_BODY);
#undef TSGU_COLDOT_BODY
    return TSGU_OK;
}

int tsgu_cg_alpha(int vtype, const void* pap_partial, int64_t n_partial, void* fold, void* scal, int* flags,
                  double eps, int64_t p, int device, void* stream) {
    if (!pap_partial || !scal || !flags || p <= 0 || n_partial < 0) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // many partial rows (one per K1 workgroup): fold them to kFoldRows rows first (needs `fold`)
    const bool do_fold = fold != nullptr && n_partial > 4 * kFoldRows;
    const int64_t chunk = (n_partial + kFoldRows - 1) / kFoldRows;
#define TSGU_BODY                                                                                              \
    {                                                                                                          \
        const V* src = (const V*)pap_partial;                                                                  \
        int64_t rows = n_partial;                                                                              \
        if (do_fold) {                                                                                         \
            hipLaunchKernelGGL((colsum_fold_kernel<V>), dim3(kFoldRows), dim3(kBlock), 0, s, src, n_partial, p, \
                               chunk, (V*)fold, (const int*)flags);                                            \
            if (const int rc = check_launch()) return rc;                                                      \
            src = (const V*)fold;                                                                              \
            rows = kFoldRows;                                                                                  \
        }                                                                                                      \
        hipLaunchKernelGGL((cg_alpha_kernel<V>), dim3((unsigned)((p + 63) / 64)), dim3(kBlock), 0, s, src,      \
                           rows, p, (V*)scal, (const int*)flags, (V)eps);                                      \
        return check_launch();                                                                                 \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
    return TSGU_OK;
}

int tsgu_cg_update1(int vtype, int64_t n, int64_t p, void* r, const void* Ap, void* x, const void* pvec,
                    const void* scal, const int* flags, void* rr_partial, int device, void* stream) {
    if (n <= 0 || p <= 0 || !r || !Ap || !x || !pvec || !scal || !flags || !rr_partial) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // rr_partial has tsgu_cg_num_blocks(vtype, n, p) rows; that count assumes aligned operands.
#define TSGU_BODY                                                                                               \
    {                                                                                                           \
        VecGeom g;                                                                                              \
        constexpr int wide = VT<V>::kWide;                                                                      \
        if (!(aligned16(r) && aligned16(Ap) && aligned16(x) && aligned16(pvec))) return TSGU_ERR_BAD_ARG;      \
        if (!geom_for<V>(n, p, true, g)) return TSGU_ERR_TOO_LARGE;                                             \
        if (g.vec == 1)                                                                                         \
            hipLaunchKernelGGL((cg_update1_kernel<V, 1>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n, p,   \
                               (V*)r, (const V*)Ap, (V*)x, (const V*)pvec, (const V*)scal, flags, g.lpr, g.rpp, \
                               (V*)rr_partial);                                                                 \
        else                                                                                                    \
            hipLaunchKernelGGL((cg_update1_kernel<V, wide>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n,   \
                               p, (V*)r, (const V*)Ap, (V*)x, (const V*)pvec, (const V*)scal, flags, g.lpr,     \
                               g.rpp, (V*)rr_partial);                                                          \
        return check_launch();                                                                                  \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
    return TSGU_OK;
}

int tsgu_cg_update1_alpha(int vtype, int64_t n, int64_t p, void* r, const void* Ap, void* x, const void* pvec, const void* pap_partial,
                          int64_t n_partial, void* scal, const int* flags, double eps, void* rr_partial, int device, void* stream) {
    if (n <= 0 || p <= 0 || p > kBlock || !r || !Ap || !x || !pvec || !pap_partial || n_partial <= 0 || !scal || !flags || !rr_partial)
        return TSGU_ERR_BAD_ARG;
    if (n_partial > 1024) return TSGU_ERR_TOO_LARGE;      // (every workgroup reads all partial rows: use tsgu_cg_alpha + tsgu_cg_update1)
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TSGU_BODY                                                                                                     \
    {                                                                                                                 \
        VecGeom g;                                                                                                    \
        constexpr int wide = VT<V>::kWide;                                                                            \
        if (!(aligned16(r) && aligned16(Ap) && aligned16(x) && aligned16(pvec))) return TSGU_ERR_BAD_ARG;            \
        if (!geom_for<V>(n, p, true, g)) return TSGU_ERR_TOO_LARGE;                                                   \
        if (g.vec == 1)                                                                                               \
            hipLaunchKernelGGL((cg_update1_alpha_kernel<V, 1>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n, p,   \
                               (V*)r, (const V*)Ap, (V*)x, (const V*)pvec, (const V*)pap_partial, n_partial,         \
                               (V*)scal, flags, (V)eps, g.lpr, g.rpp, (V*)rr_partial);                               \
        else                                                                                                          \
            hipLaunchKernelGGL((cg_update1_alpha_kernel<V, wide>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n,   \
                               p, (V*)r, (const V*)Ap, (V*)x, (const V*)pvec, (const V*)pap_partial, n_partial,      \
                               (V*)scal, flags, (V)eps, g.lpr, g.rpp, (V*)rr_partial);                               \
        return check_launch();                                                                                        \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
    return TSGU_OK;
}

int tsgu_cg_beta(int vtype, const void* rr_partial, int64_t n_partial, void* scal, int* flags, double eps,
                 double stop_updating_after, double tolerance, int iter_index, int min_iter_index, int64_t p,
                 int device, void* stream) {
    return tsgu_cg_beta_precond(vtype, rr_partial, n_partial, nullptr, 0, scal, flags, eps, stop_updating_after, tolerance,
                                iter_index, min_iter_index, p, device, stream);
}

int tsgu_cg_beta_precond(int vtype, const void* rr_partial, int64_t n_partial, const void* rz_partial, int64_t n_rz, void* scal,
                         int* flags, double eps, double stop_updating_after, double tolerance, int iter_index,
                         int min_iter_index, int64_t p, int device, void* stream) {
    if (!rr_partial || !scal || !flags || p <= 0 || n_partial < 0 || n_rz < 0) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TSGU_BODY                                                                                             \
    {                                                                                                         \
        hipLaunchKernelGGL((cg_beta_kernel<V>), dim3(1), dim3(kBlock), 0, s, (const V*)rr_partial, n_partial, \
                           p, (V*)scal, flags, (V)eps, (V)stop_updating_after, (V)tolerance, iter_index,      \
                           min_iter_index, (const V*)rz_partial, n_rz);                                       \
        return check_launch();                                                                                \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
    return TSGU_OK;
}

int tsgu_cg_update2(int vtype, int64_t n, int64_t p, const void* r, void* pvec, const void* scal,
                    const int* flags, int device, void* stream) {
    if (n <= 0 || p <= 0 || !r || !pvec || !scal || !flags) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TSGU_BODY                                                                                             \
    {                                                                                                         \
        constexpr int wide = VT<V>::kWide;                                                                    \
        VecGeom g;                                                                                            \
        if (!(aligned16(r) && aligned16(pvec))) return TSGU_ERR_BAD_ARG;                                      \
        if (!geom_for<V>(n, p, true, g)) return TSGU_ERR_TOO_LARGE;                                           \
        if (g.vec == 1)                                                                                       \
            hipLaunchKernelGGL((cg_update2_kernel<V, 1>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n, p, \
                               (const V*)r, (V*)pvec, (const V*)scal, flags, g.lpr, g.rpp);                   \
        else                                                                                                  \
            hipLaunchKernelGGL((cg_update2_kernel<V, wide>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n, \
                               p, (const V*)r, (V*)pvec, (const V*)scal, flags, g.lpr, g.rpp);                \
        return check_launch();                                                                                \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
    return TSGU_OK;
}

// ---- the two-launch form (see cg_residual_alpha_kernel) ----
static inline int cg2_groups(int64_t blocks) { return (int)((blocks + 1023) / 1024); }    // so that at most 1024 partial rows are left

int64_t tsgu_cg2_num_blocks(int vtype, int64_t n, int64_t p) {
    VecGeom g;
    const bool ok = vtype == TSGU_F64 ? geom_for<double>(n, p, true, g) : geom_for<float>(n, p, true, g);
    if (!ok || p > kBlock) return -1;
    const int64_t per = (int64_t)g.rpp * kPasses * cg2_groups(g.blocks);
    return (n + per - 1) / per;
}

int tsgu_cg2_residual(int vtype, int64_t n, int64_t p, void* r, const void* Ap, const void* pap_partial, int64_t n_partial, void* scal2,
                      const int* flags2, int parity, double eps, void* rr_partial, int device, void* stream) {
    if (n <= 0 || p <= 0 || p > kBlock || !r || !Ap || !pap_partial || n_partial <= 0 || !scal2 || !flags2 || !rr_partial || (parity & ~1))
        return TSGU_ERR_BAD_ARG;
    if (n_partial > 1024) return TSGU_ERR_TOO_LARGE;      // (every workgroup reads all partial rows)
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TSGU_BODY                                                                                                          \
    {                                                                                                                      \
        VecGeom g;                                                                                                         \
        constexpr int wide = VT<V>::kWide;                                                                                 \
        if (!(aligned16(r) && aligned16(Ap))) return TSGU_ERR_BAD_ARG;                                                     \
        if (!geom_for<V>(n, p, true, g)) return TSGU_ERR_TOO_LARGE;                                                        \
        const int groups = cg2_groups(g.blocks);                                                                           \
        const int64_t per = (int64_t)g.rpp * kPasses * groups;                                                             \
        const unsigned blocks = (unsigned)((n + per - 1) / per);                                                           \
        auto go = [&](auto kern) {                                                                                         \
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(kBlock), 0, s, n, p, (V*)r, (const V*)Ap, (const V*)pap_partial,   \
                               n_partial, (V*)scal2, flags2, parity, (V)eps, g.lpr, g.rpp, groups, (V*)rr_partial);       \
        };                                                                                                                 \
        if (g.vec == 1) go(cg_residual_alpha_kernel<V, 1, 0>);                                                             \
        else if (groups == 1) go(cg_residual_alpha_kernel<V, wide, 1>);                                                    \
        else if (groups == 2) go(cg_residual_alpha_kernel<V, wide, 2>);                                                    \
        else go(cg_residual_alpha_kernel<V, wide, 0>);                                                                     \
        return check_launch();                                                                                             \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
    return TSGU_OK;
}

int tsgu_cg2_direction(int vtype, int64_t n, int64_t p, const void* r, void* pvec, void* x, const void* rr_partial, int64_t n_partial,
                       void* scal2, int* flags2, int parity, double eps, double stop_updating_after, double tolerance,
                       int min_iter_index, void* hist, int n_hist, int device, void* stream) {
    if ((hist != nullptr) != (n_hist > 0)) return TSGU_ERR_BAD_ARG;
    if (n <= 0 || p <= 0 || p > kBlock || !r || !pvec || !x || !rr_partial || n_partial <= 0 || !scal2 || !flags2 || (parity & ~1))
        return TSGU_ERR_BAD_ARG;
    if (n_partial > 1024) return TSGU_ERR_TOO_LARGE;
    if (const int rc = set_device(device)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TSGU_BODY                                                                                                          \
    {                                                                                                                      \
        VecGeom g;                                                                                                         \
        constexpr int wide = VT<V>::kWide;                                                                                 \
        if (!(aligned16(r) && aligned16(pvec) && aligned16(x))) return TSGU_ERR_BAD_ARG;                                   \
        if (!geom_for<V>(n, p, true, g)) return TSGU_ERR_TOO_LARGE;                                                        \
        if (g.vec == 1)                                                                                                    \
            hipLaunchKernelGGL((cg_direction_beta_kernel<V, 1>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n, p,       \
                               (const V*)r, (V*)pvec, (V*)x, (const V*)rr_partial, n_partial, (V*)scal2, flags2, parity,   \
                               (V)eps, (V)stop_updating_after, (V)tolerance, min_iter_index, g.lpr, g.rpp, (V*)hist,       \
                               n_hist);                                                                                    \
        else                                                                                                               \
            hipLaunchKernelGGL((cg_direction_beta_kernel<V, wide>), dim3((unsigned)g.blocks), dim3(kBlock), 0, s, n, p,    \
                               (const V*)r, (V*)pvec, (V*)x, (const V*)rr_partial, n_partial, (V*)scal2, flags2, parity,   \
                               (V)eps, (V)stop_updating_after, (V)tolerance, min_iter_index, g.lpr, g.rpp, (V*)hist,       \
                               n_hist);                                                                                    \
        return check_launch();                                                                                             \
    }
    TSGU_VSWITCH(vtype, TSGU_BODY, TSGU_BODY);
#undef TSGU_BODY
    return TSGU_OK;
}

}  // extern "C"
