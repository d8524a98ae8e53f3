// Wave-pipelined, LDS-tiled K1/K2/K3 ("wavetile") for patterns whose neighbouring rows share
// columns (stencils, banded matrices).
//
// Why: the gather kernels push 27 × 128 B per output row through the per-CU L1/TA path (3.5 GB at
// C2 for 0.48 GB of HBM traffic) and the rocprof counters show that path, not HBM, saturated
// (profiles/r01_*).  Here the operand rows a *wave* needs for its RPT consecutive matrix rows are
// fetched ONCE into a wave-private LDS tile (2.4x fewer bytes through L1 at C2) and every entry
// reads its row from LDS (ds_read_b128 = 4x the L1 rate).
//
// How it stays busy: a single-buffered workgroup-level tile was measured SLOWER than the gather
// kernel (LDS capacity x load latency bounds it).  So each wave is its own persistent software
// pipeline over a contiguous run of tasks (task = RPT rows), with NO inter-wave synchronisation:
//     step t:  rows(t) regs -> LDS tile | issue row loads for t+2 | issue index loads for t+4 |
//              compute task t from LDS
// The data in flight lives in VGPRs (the 512 KiB register file is the largest on-chip store), two
// tasks deep; all loads are unconditional (indices clamped) so the compiler keeps counted vmcnt
// waits instead of draining the queue.
//
// Needs a per-pattern plan (built once, cached), laid out for wide aligned loads:
//   tmeta[ntask]      int2  {first entry, entry count} of the task
//   tcols[ntask][CAP] int32 distinct columns of the task's rows, padded with the last valid one
//   lidx [ntask][256] uint8 tile row of each entry (task-relative entry order), zero padded
// Every task must have <= CAP distinct columns and <= 256 entries, otherwise the host selects the
// gather kernels.  Same summation order as K1/K3: results are bit-identical.
#pragma once

#include "tsgu_common.h"

namespace tsgu {

enum WtMode { kWtSpmm = 0, kWtSpmmPerm = 1, kWtSddmm = 2, kWtBwd = 3 };

constexpr bool wt_has_perm(int mode) { return mode == kWtSpmmPerm || mode == kWtBwd; }
constexpr bool wt_has_dots(int mode) { return mode == kWtSddmm || mode == kWtBwd; }
constexpr bool wt_has_acc(int mode) { return mode != kWtSddmm; }

constexpr int kWtLT = 12;        // tile rows per loader group (16-byte loads a lane keeps in flight per task)
constexpr int kWtEnt = 4;        // entries per lane per task  -> 256 entries per task
constexpr int kWtWaves = 4;      // waves per workgroup (independent pipelines)

struct WtParams {
    int64_t n_rows, nnz, p, ntask;
    const void* crow;
    const void* val;
    const void* perm;
    const int2* tmeta;           // [ntask] {entry begin, entry count}
    const int* tcols;            // [ntask][CAP]
    const unsigned char* lidx;   // [ntask][256]
    const void* X;               // gathered operand rows
    int64_t ldx;
    const void* R;               // sddmm row operand
    int64_t ldr;
    void* out;                   // spmm / bwd: dense rows [n_rows][ldo]; sddmm: values [nnz]
    void* out2;                  // bwd: gradA values [nnz], addressed through perm
    int64_t ldo;
    double alpha;
    int64_t nblocks;             // workgroups launched
    int64_t tasks_per_wave;
};

template <int CL>
struct WtGeom {
    static constexpr int NG = kWave / CL;                                  // loader groups per wave
    static constexpr int CAP = (kWtLT * NG) < 256 ? (kWtLT * NG) : 256;    // tile rows per wave
    static constexpr int LT = (CAP + NG - 1) / NG;
};

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct __attribute__((packed, aligned(4))) WtU4 {  // 16 bytes at 4-byte alignment (A's value / perm arrays)
    unsigned w[4];
};

template <typename V, int VEC, int CL, int MODE>
struct WtStage {
    int2 cur, ent, idx, pre;  // {entry begin, count} of tasks t, t+2, t+4, t+6 (this stage's parity)
    u32x4_t rows[WtGeom<CL>::LT];
    i32x4_t tcs[(WtGeom<CL>::LT + 3) / 4];
    unsigned lid4;            // 4 local indices
    float ent_v[kWtEnt];
    int q[kWtEnt];            // perm positions (SpmmPerm / Bwd)
    int qs[kWtEnt];           // perm positions of the task whose entries are in flight (Bwd scatter)
    int shift;                // entries the 4-wide loads were shifted back to stay inside the arrays
    int rs, re;               // entry range of this lane's row (absolute)
    bool row_ok;
    float own[2 * VEC];       // sddmm row operand (compute geometry: up to 2 vectors per lane)
};

// The 4-wide entry loads of a lane are clamped to start at nnz-4; when that moved the window back by d
// elements (only the tail lanes of the very last task), element u of the lane is w[u+d].
template <typename T>
__device__ __forceinline__ void wt_unshift(T (&w)[4], int d) {
    if (d > 0) {
        T r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = u + d;
            r[u] = s == 1 ? w[1] : s == 2 ? w[2] : w[3];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = r[u];
    }
}

template <typename V, typename I, int VEC, int CL, int EP, int MODE>
struct WtCtx {
    using G = WtGeom<CL>;
    static constexpr int GROUP = CL * EP;
    static constexpr int RPT = kWave / GROUP;
    static constexpr int TW = CL * VEC;
    static constexpr int NT4 = (G::LT + 3) / 4;
    static constexpr unsigned kRowBytes = TW * sizeof(V);
    // Compute geometry: half as many column lanes as the loader geometry, each covering NV vectors, and
    // twice the entry lanes.  The per-entry overhead (slot read, address, masks) is paid by half as many
    // lanes per entry -> ~1/3 fewer instructions per task, which is what bounds this kernel.
    static constexpr int CC = CL >= 2 ? CL / 2 : CL;
    static constexpr int NV = CL / CC;
    static constexpr int EC = EP * NV;
    static_assert(G::LT * G::NG == G::CAP || true, "");

    const WtParams& P;
    const I* crow;
    const V* val;
    const I* perm;
    const V* X;
    uint32_t ldx;
    unsigned char* tile;  // wave-private LDS tile (bytes)
    uint2* slots;         // {tile row byte offset, value bits}
    float* dots;
    int lane, cl, ep, grp, lg;
    int clc, epc;  // compute-geometry lane coordinates inside the row group
    int64_t c0;
    int64_t t_base, t_stride, k_last;  // this wave's tasks: t_base + k*t_stride, k = 0..k_last
    int nnz_m4;                        // nnz - 4 (clamp for the 4-wide entry loads)

    // task id of local step k (clamped to the wave's last task so that prefetches stay in bounds)
    __device__ __forceinline__ int64_t clamp_task(int64_t k) const { return t_base + (k < k_last ? k : k_last) * t_stride; }

    __device__ __forceinline__ int2 load_meta(int64_t task) const { return P.tmeta[task]; }

    // 16-byte operand-row loads (st.tcs of the same task were loaded two steps earlier).
    // Loader group lg owns tile rows lg*LT .. lg*LT+LT-1.
    __device__ __forceinline__ void issue_rows(WtStage<V, VEC, CL, MODE>& st) const {
#pragma unroll
        for (int i = 0; i < G::LT; ++i) {
            const int j = st.tcs[i / 4][i % 4];
            const V* src = X + row_off(j, ldx);
            if constexpr (VEC == 1) {
                unsigned w = 0;
                __builtin_memcpy(&w, src, sizeof(V));
                st.rows[i] = (u32x4_t){w, 0u, 0u, 0u};
            } else {
                st.rows[i] = *reinterpret_cast<const u32x4_t*>(src);
            }
        }
    }

    // first entry index of this lane's 4-wide window for a task starting at e0, and how far it was shifted
    __device__ __forceinline__ int entry_window(int e0, int& shift) const {
        int k = e0 + lane * 4;
        const int kc = k < nnz_m4 ? k : nnz_m4;
        shift = k - kc;
        return kc > 0 ? kc : 0;
    }

    // entry loads (4 consecutive entries per lane), row bounds and the row operand of `task`
    __device__ __forceinline__ void issue_entries(WtStage<V, VEC, CL, MODE>& st, int64_t task, int2 m) const {
        const int e0 = __builtin_amdgcn_readfirstlane(m.x);
        st.lid4 = *reinterpret_cast<const unsigned*>(P.lidx + task * 256 + lane * 4);
        int shift;
        const int k = entry_window(e0, shift);
        if constexpr (MODE == kWtSpmm) {
            st.shift = shift;  // applied to the values when they are drained
            if constexpr (sizeof(V) == 4) {
                const WtU4 raw = *reinterpret_cast<const WtU4*>(val + k);
#pragma unroll
                for (int u = 0; u < kWtEnt; ++u) st.ent_v[u] = __uint_as_float(raw.w[u]);
            } else {
#pragma unroll
                for (int u = 0; u < kWtEnt; ++u) st.ent_v[u] = VT<V>::up(val[k + u]);
            }
        }
        if constexpr (wt_has_perm(MODE)) {
            // st.q (loaded two steps ago for this task) arrived long ago: undo its window shift, then gather
            wt_unshift(st.q, st.shift);
            st.shift = 0;
#pragma unroll
            for (int u = 0; u < kWtEnt; ++u) {
                int qq = st.q[u];
                qq = qq < 0 ? 0 : (qq < (int)P.nnz ? qq : (int)P.nnz - 1);
                st.q[u] = qq;
                st.ent_v[u] = VT<V>::up(val[qq]);
            }
            if constexpr (MODE == kWtBwd) {
#pragma unroll
                for (int u = 0; u < kWtEnt; ++u) st.qs[u] = st.q[u];
            }
        }
        int64_t row = task * RPT + grp;
        const bool ok = row < P.n_rows;
        row = ok ? row : P.n_rows - 1;
        st.rs = (int)crow[row];  // unconditional (row is clamped); masked with row_ok when consumed
        st.re = (int)crow[row + 1];
        st.row_ok = ok;
        if constexpr (wt_has_dots(MODE)) {
#pragma unroll
            for (int nv = 0; nv < NV; ++nv) {
                const int64_t cb = (int64_t)(clc * NV + nv) * VEC;
                float o[VEC];
                load_vec<V, VEC>(static_cast<const V*>(P.R) + row * P.ldr + (cb < P.p ? cb : 0), o);  // masked when consumed
#pragma unroll
                for (int v = 0; v < VEC; ++v) st.own[nv * VEC + v] = o[v];
            }
        }
    }

    // distinct-column loads (and perm loads) of `task`
    __device__ __forceinline__ void issue_index(WtStage<V, VEC, CL, MODE>& st, int64_t task, int2 m) const {
        const int* src = P.tcols + task * G::CAP + lg * G::LT;
#pragma unroll
        for (int i = 0; i < NT4; ++i) {
            if constexpr (G::LT % 4 == 0) {
                st.tcs[i] = *reinterpret_cast<const i32x4_t*>(src + 4 * i);
            } else {
#pragma unroll
                for (int w = 0; w < 4; ++w) st.tcs[i][w] = (4 * i + w < G::LT) ? src[4 * i + w] : 0;
            }
        }
        if constexpr (wt_has_perm(MODE)) {
            const int e0 = __builtin_amdgcn_readfirstlane(m.x);
            const int k = entry_window(e0, st.shift);
            if constexpr (sizeof(I) == 4) {
                const WtU4 raw = *reinterpret_cast<const WtU4*>(perm + k);
#pragma unroll
                for (int u = 0; u < kWtEnt; ++u) st.q[u] = (int)raw.w[u];
            } else {
#pragma unroll
                for (int u = 0; u < kWtEnt; ++u) st.q[u] = (int)perm[k + u];
            }
        }
    }

    // registers -> wave-private LDS (tile rows + entry slots).  Branch-free: rows / slots beyond the
    // task's counts hold padding and land in tile rows / slots that are never read.
    __device__ __forceinline__ void drain(WtStage<V, VEC, CL, MODE>& st) const {
        unsigned char* dst = tile + (size_t)(lg * G::LT) * kRowBytes + (c0 < P.p ? c0 : 0) * sizeof(V);
#pragma unroll
        for (int i = 0; i < G::LT; ++i) {
            if constexpr (VEC == 1) {
                const unsigned w = st.rows[i].x;
                __builtin_memcpy(dst + (size_t)i * kRowBytes, &w, sizeof(V));
            } else {
                *reinterpret_cast<u32x4_t*>(dst + (size_t)i * kRowBytes) = st.rows[i];
            }
        }
        float ev[kWtEnt];
#pragma unroll
        for (int u = 0; u < kWtEnt; ++u) ev[u] = st.ent_v[u];
        if constexpr (MODE == kWtSpmm) wt_unshift(ev, st.shift);
        u32x4_t lo, hi;
        lo.x = (st.lid4 & 0xffu) * kRowBytes;
        lo.y = __float_as_uint(ev[0]);
        lo.z = ((st.lid4 >> 8) & 0xffu) * kRowBytes;
        lo.w = __float_as_uint(ev[1]);
        hi.x = ((st.lid4 >> 16) & 0xffu) * kRowBytes;
        hi.y = __float_as_uint(ev[2]);
        hi.z = (st.lid4 >> 24) * kRowBytes;
        hi.w = __float_as_uint(ev[3]);
        u32x4_t* sl = reinterpret_cast<u32x4_t*>(slots + lane * 4);
        sl[0] = lo;
        sl[1] = hi;
    }

    // one batch of up to U entries of this lane (MASKED: clamp slot indices to the row's last entry and
    // zero-weight the padding)
    template <bool MASKED, int U>
    __device__ __forceinline__ void batch(int i, int iend, const unsigned char* trow, const float (&own)[2 * VEC],
                                          float (&acc)[2 * VEC], const bool (&vok)[2]) const {
        uint2 e[U];
        float b[U][NV][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int k = i + u * EC;
            if constexpr (MASKED) k = k < iend ? k : iend - 1;
            e[u] = slots[k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int nv = 0; nv < NV; ++nv)
                load_vec<V, VEC>(reinterpret_cast<const V*>(trow + e[u].x + nv * VEC * sizeof(V)), b[u][nv]);
        }
        if constexpr (wt_has_dots(MODE)) {
            float d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                d[u] = 0;
#pragma unroll
                for (int nv = 0; nv < NV; ++nv) {
                    float t = 0;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) t = fma(own[nv * VEC + v], b[u][nv][v], t);
                    d[u] += vok[nv] ? t : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = group_sum<float, CC>(d[u]);
            if (clc == 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!MASKED || i + u * EC < iend) dots[i + u * EC] = d[u];
                }
            }
        }
        if constexpr (wt_has_acc(MODE)) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float a = __uint_as_float(e[u].y);
                if constexpr (MASKED) a = (i + u * EC < iend) ? a : 0.f;
#pragma unroll
                for (int nv = 0; nv < NV; ++nv) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[nv * VEC + v] = fma(a, b[u][nv][v], acc[nv * VEC + v]);
                }
            }
        }
    }

    // consume task from LDS (compute geometry CC x EC lanes per row).  Full batches run without clamps
    // or masks; at most one masked batch finishes the row.
    __device__ __forceinline__ void compute(int64_t task, int eb, int ne, int rs, int re, const float (&own)[2 * VEC],
                                            const int (&qcur)[kWtEnt]) const {
#ifndef TSGU_WT_U
#define TSGU_WT_U 4
#endif
        constexpr int U = MODE == kWtBwd ? 2 : TSGU_WT_U;  // the fused mode is at the register limit
        const int64_t cb0 = (int64_t)clc * NV * VEC;
        bool vok[2];
        vok[0] = cb0 < P.p;
        vok[1] = NV > 1 && cb0 + VEC < P.p;
        float acc[2 * VEC];
#pragma unroll
        for (int v = 0; v < 2 * VEC; ++v) acc[v] = 0;
        int i = rs - eb + epc;
        const int iend = re - eb;
        const unsigned char* trow = tile + (vok[0] ? cb0 : 0) * sizeof(V);
        for (; i + (U - 1) * EC < iend; i += U * EC) batch<false, U>(i, iend, trow, own, acc, vok);
        if (i < iend) batch<true, U>(i, iend, trow, own, acc, vok);
        if constexpr (MODE == kWtSddmm) {
            V* __restrict__ outv = static_cast<V*>(P.out);
            const float alpha = (float)P.alpha;
#pragma unroll
            for (int u = 0; u < kWtEnt; ++u) {
                const int e = lane + u * kWave;
                if (e < ne) outv[(int64_t)eb + e] = VT<V>::down(alpha * dots[e]);
            }
        }
        if constexpr (MODE == kWtBwd) {
            // gradA[perm[k]] = <G[i,:], B[j,:]>: this lane owns entries 4·lane .. 4·lane+3 of the task
            V* __restrict__ ga = static_cast<V*>(P.out2);
#pragma unroll
            for (int u = 0; u < kWtEnt; ++u) {
                const int e = lane * 4 + u;
                if (e < ne) ga[qcur[u]] = VT<V>::down(dots[e]);
            }
        }
        if constexpr (wt_has_acc(MODE)) {
            if constexpr (EC > 1) {
#pragma unroll
                for (int v = 0; v < NV * VEC; ++v) acc[v] = ep_sum<float, CC, EC>(acc[v]);
            }
            const int64_t row = task * RPT + grp;
            if (row < P.n_rows && epc == 0) {
#pragma unroll
                for (int nv = 0; nv < NV; ++nv) {
                    if (vok[nv]) {
                        float o[VEC];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) o[v] = acc[nv * VEC + v];
                        store_vec<V, VEC, true>(static_cast<V*>(P.out) + row * P.ldo + cb0 + nv * VEC, o);
                    }
                }
            }
        }
    }

    // one pipeline step for task t using register stage `st`
    __device__ __forceinline__ void step(WtStage<V, VEC, CL, MODE>& st, int64_t t) const {
        st.cur = st.ent;  // meta(t)
        st.ent = st.idx;  // meta(t+2)
        st.idx = st.pre;  // meta(t+4)
        st.pre = load_meta(clamp_task(t + 6));
        drain(st);
        const int eb = __builtin_amdgcn_readfirstlane(st.cur.x);
        const int ne = __builtin_amdgcn_readfirstlane(st.cur.y);
        const int rs = st.row_ok ? st.rs : 0, re = st.row_ok ? st.re : 0;
        float own[2 * VEC];
#pragma unroll
        for (int v = 0; v < 2 * VEC; ++v) own[v] = (wt_has_dots(MODE) && st.row_ok && v < NV * VEC) ? st.own[v] : 0.f;
        int qcur[kWtEnt];
#pragma unroll
        for (int u = 0; u < kWtEnt; ++u) qcur[u] = (MODE == kWtBwd) ? st.qs[u] : 0;
        issue_rows(st);                                      // step t+2 (st.tcs were loaded two steps ago)
        issue_entries(st, clamp_task(t + 2), st.ent);        // step t+2 (st.q were loaded two steps ago)
        issue_index(st, clamp_task(t + 4), st.idx);          // step t+4
        compute(clamp_task(t), eb, ne, rs, re, own, qcur);
    }

    // fill stage `st` for the tasks tA, tA+2, tA+4 it will process first
    __device__ __forceinline__ void prime(WtStage<V, VEC, CL, MODE>& st, int64_t tA) const {
        st.ent = load_meta(clamp_task(tA));
        st.idx = load_meta(clamp_task(tA + 2));
        st.pre = load_meta(clamp_task(tA + 4));
        issue_index(st, clamp_task(tA), st.ent);
        issue_rows(st);
        issue_entries(st, clamp_task(tA), st.ent);
        issue_index(st, clamp_task(tA + 2), st.idx);
    }
};

template <typename V, typename I, int VEC, int CL, int EP, int MODE>
__global__ __launch_bounds__(kWave* kWtWaves, 2) void csr_wavetile_kernel(const WtParams P) {
    using Ctx = WtCtx<V, I, VEC, CL, EP, MODE>;
    using G = WtGeom<CL>;
    constexpr int TW = CL * VEC;
    constexpr size_t kTileBytes = (size_t)G::CAP * TW * sizeof(V);
    constexpr size_t kWaveLds = kTileBytes + 256 * sizeof(uint2) + (wt_has_dots(MODE) ? 256 * sizeof(float) : 0);

    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    unsigned char* base = dsm + (size_t)wave * kWaveLds;

    const int lane = threadIdx.x & (kWave - 1);
    // Task order: the waves of one XCD sweep that XCD's contiguous slice of tasks round-robin
    // (wave w takes tasks w, w+W, w+2W, ...), so at any moment the whole XCD works inside a narrow
    // window of rows and the operand rows shared between neighbouring tasks (the other stencil
    // bands) are still in its L2.  (A contiguous run of tasks per wave was measured slower than
    // the gather kernel: every band became a separate HBM stream.)  nblocks is a multiple of 8.
    const int64_t per_xcd = P.nblocks / kXcd;              // workgroups per XCD
    const int64_t xcd = blockIdx.x % kXcd;
    const int64_t wx = (blockIdx.x / kXcd) * kWtWaves + wave;  // wave index inside the XCD
    const int64_t W = per_xcd * kWtWaves;
    const int64_t tx = (P.ntask + kXcd - 1) / kXcd;        // tasks per XCD slice
    const int64_t x0 = xcd * tx;
    int64_t x1 = x0 + tx;
    x1 = x1 < P.ntask ? x1 : P.ntask;
    const int64_t t_first = x0 + wx;
    if (t_first >= x1) return;
    const int64_t nsteps = (x1 - t_first + W - 1) / W;
    const int64_t t0 = 0, t1 = nsteps;

    Ctx c{P};
    c.crow = static_cast<const I*>(P.crow);
    c.val = static_cast<const V*>(P.val);
    c.perm = static_cast<const I*>(P.perm);
    c.lane = lane;
    c.cl = lane % CL;
    c.lg = lane / CL;
    c.ep = (lane % Ctx::GROUP) / CL;
    c.grp = lane / Ctx::GROUP;
    c.clc = (lane % Ctx::GROUP) % Ctx::CC;
    c.epc = (lane % Ctx::GROUP) / Ctx::CC;
    c.c0 = (int64_t)c.cl * VEC;
    c.X = static_cast<const V*>(P.X) + (c.c0 < P.p ? c.c0 : 0);
    c.ldx = (uint32_t)P.ldx;
    c.tile = base;
    c.nnz_m4 = (int)(P.nnz - 4);
    c.slots = reinterpret_cast<uint2*>(base + kTileBytes);
    c.dots = reinterpret_cast<float*>(base + kTileBytes + 256 * sizeof(uint2));
    c.t_base = t_first;
    c.t_stride = W;
    c.k_last = nsteps - 1;

    WtStage<V, VEC, CL, MODE> s0, s1;
    c.prime(s0, t0);
    c.prime(s1, t0 + 1);
    int64_t t = t0;
    // hipcc's waitcnt pass drains the VMEM queue at every loop back-edge (it cannot carry counted
    // waits across it), which exposes one full load latency per iteration: amortise it over several
    // pipeline steps per iteration; inside the body the waits stay counted.
#ifndef TSGU_WT_PAIRS
#define TSGU_WT_PAIRS 1
#endif
    for (; t + 2 * TSGU_WT_PAIRS - 1 < t1; t += 2 * TSGU_WT_PAIRS) {
#pragma unroll
        for (int k = 0; k < TSGU_WT_PAIRS; ++k) {
            c.step(s0, t + 2 * k);
            c.step(s1, t + 2 * k + 1);
        }
    }
    for (; t + 1 < t1; t += 2) {
        c.step(s0, t);
        c.step(s1, t + 1);
    }
    if (t < t1) c.step(s0, t);
}

template <typename V>
inline size_t wavetile_lds_bytes(int cl, int vec, bool sddmm) {
    const int ng = kWave / cl;
    const int cap = (kWtLT * ng) < 256 ? (kWtLT * ng) : 256;
    return (size_t)kWtWaves * ((size_t)cap * cl * vec * sizeof(V) + 256 * sizeof(uint2) + (sddmm ? 256 * sizeof(float) : 0));
}

template <typename V, typename I, int MODE>
int wavetile_launch(WtParams P, bool can_wide, int n_cu, hipStream_t stream) {
    constexpr int wide = VT<V>::kWide;
    const RowGeom g = pick_geom(wide, can_wide, P.p);
    if (g.col_tiles != 1) return TSGU_ERR_BAD_ARG;
    const int rpt = kWave / (g.cl * g.ep);
    P.ntask = (P.n_rows + rpt - 1) / rpt;
    if (P.ldx > 0xffffffffLL || P.nnz > 0x7fffffffLL || P.nnz < 4) return TSGU_ERR_TOO_LARGE;
    const size_t lds = wavetile_lds_bytes<V>(g.cl, g.vec, wt_has_dots(MODE));
    if (lds > 80 * 1024) return TSGU_ERR_TOO_LARGE;
    // persistent grid: 2 workgroups (8 independent wave pipelines) per CU
    int64_t blocks = (int64_t)n_cu * 2;
    const int64_t need = (P.ntask + kWtWaves - 1) / kWtWaves;
    if (blocks > need) blocks = need;
    blocks = (blocks + kXcd - 1) / kXcd * kXcd;  // whole workgroups per XCD
    P.nblocks = blocks;
    P.tasks_per_wave = 0;
    const dim3 grid((unsigned)blocks, 1, 1);
    return dispatch_geom(g, [&](auto cl, auto ep) -> int {
        constexpr int CL = decltype(cl)::value, EP = decltype(ep)::value;
        if (g.vec == 1)
            hipLaunchKernelGGL((csr_wavetile_kernel<V, I, 1, CL, EP, MODE>), grid, dim3(kWave * kWtWaves), lds, stream, P);
        else
            hipLaunchKernelGGL((csr_wavetile_kernel<V, I, wide, CL, EP, MODE>), grid, dim3(kWave * kWtWaves), lds, stream, P);
        return check_launch();
    });
}

}  // namespace tsgu
