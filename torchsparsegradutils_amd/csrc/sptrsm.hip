// K4: sparse triangular solve X = op(A)^{-1} B as ONE persistent, dependency-driven launch.
//
// Level scheduling needs one grid-wide barrier (or launch) per level; the reference-sized
// problem has ~2.7k levels of ~100 rows, so barriers would dominate.  Instead every wave
// draws rows in dependency order from a device ticket and waits only on the x-entries its
// own row reads ("sync-free" sweep).  The solution array itself is the synchronisation
// medium: X is pre-filled with a signalling bit pattern (a NaN with a private payload),
// producers publish each x[row, c] with ONE agent-scope (sc1, write-through) store and
// consumers poll the element with agent-scope relaxed loads until the payload disappears —
// the 4/8-byte value is its own ready-flag, so no fences and no flag array are needed
// (naturally aligned single-store granules; see MI355X guide "handoff-1to1").
//
// Forward progress: ticket t is only handed out after tickets < t were handed to waves that
// are already running, a wave handles one row at a time (no intra-wave dependencies), and
// rows are ticketed in an order in which all dependencies of a row have smaller tickets
// (ascending for lower, descending for upper).  Every spin is bounded by a wall-clock
// timeout that raises an error word instead of hanging the device.
#include "tsgu_common.h"

namespace tsgu {

constexpr int kTrsmWaves = kBlock / kWave;   // waves of a workgroup = row classes (below)

struct TrsmWork {
    unsigned long long ticket[64];  // (kept for the layout: the error word stays at byte 512)
    int error;
    int pad[15];
    // Row tickets: rows are dealt to CLASSES, class c draws the rows ≡ c (mod classes) in order from its own counter (128 bytes
    // apart).  A class is (wave slot of a workgroup, workgroup index mod `wgc`) — up to 4 x 64 classes; with several column tiles
    // (p > 64) it is (wave slot, tile) as before.  ONE counter serves ~80 M same-address atomics per second: with one counter for
    // all rows that was the solve time of C3 (262144 rows: 3.2 ms whatever the pollers did), with four it still is the solve time
    // of a SHALLOW pattern (the reference's published shape, one off-diagonal entry per row: 65536 atomics per counter = 0.84 ms
    // whatever the number of waves).  Forward progress, per class: a counter hands its rows out in order, so every row of the class
    // below the lowest unfinished one is finished and the waves that held them are free to take it; every class has a resident wave
    // (the grid is persistent and never smaller than `wgc` workgroups).
    unsigned long long class_ticket[kTrsmWaves * 64 * 16];
};

struct TrsmParams {
    int64_t n, p;
    const void* ptr;
    const void* idx;
    const void* perm;
    const void* val;
    const void* B;
    int64_t ldb, bcs;  // row / column stride of B (elements)
    void* X;
    int64_t ldx;
    TrsmWork* work;
    int lower, unit;
    int wgc;                  // workgroup classes (1 … 64; 1 when there are several column tiles)
    long long timeout_ticks;  // wall_clock64 ticks (100 MHz)
};

template <typename V>
struct Sentinel;
template <>
struct Sentinel<float> {
    using Bits = unsigned int;
    static constexpr Bits kTag = 0x7fc5a5a5u;    // quiet NaN, private payload
    static constexpr Bits kCanon = 0x7fc00000u;  // what a genuine NaN result is stored as
    __device__ static __forceinline__ Bits bits(float v) { return __float_as_uint(v); }
    __device__ static __forceinline__ float val(Bits b) { return __uint_as_float(b); }
};
template <>
struct Sentinel<double> {
    using Bits = unsigned long long;
    static constexpr Bits kTag = 0x7ff8a5a5a5a5a5a5ull;
    static constexpr Bits kCanon = 0x7ff8000000000000ull;
    __device__ static __forceinline__ Bits bits(double v) { return (Bits)__double_as_longlong(v); }
    __device__ static __forceinline__ double val(Bits b) { return __longlong_as_double((long long)b); }
};

template <>
struct Sentinel<bf16_t> {   // bf16 elements, fp32 arithmetic: x is rounded once, when it is published
    using Bits = unsigned short;
    static constexpr Bits kTag = 0x7fc5u;
    static constexpr Bits kCanon = 0x7fc0u;
    __device__ static __forceinline__ Bits bits(float v) { return VT<bf16_t>::down(v).bits; }
    __device__ static __forceinline__ float val(Bits b) { return __uint_as_float((unsigned int)b << 16); }
};

template <typename V>
__global__ __launch_bounds__(kBlock) void sptrsm_fill_kernel(void* X, int64_t ldx, int64_t n, int64_t p, TrsmWork* work) {
    using S = Sentinel<V>;
    using Bits = typename S::Bits;
    Bits* x = static_cast<Bits*>(X);
    const int64_t total = n * p;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / p, c = i - r * p;
        __hip_atomic_store(x + r * ldx + c, S::kTag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < kTrsmWaves * 64; i += kBlock)
            __hip_atomic_store(&work->class_ticket[i * 16], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        __hip_atomic_store(&work->ticket[threadIdx.x], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) __hip_atomic_store(&work->error, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Sum over the EP consecutive lanes of a group (every lane gets the total): DPP for groups up to a row of 16, wave shuffles beyond
template <typename A, int EP>
__device__ __forceinline__ A entry_sum(A x) {
    if constexpr (EP <= 16) {
        return group_sum<A, EP>(x);
    } else {
        x = group_sum<A, 16>(x);
#pragma unroll
        for (int m = 16; m < EP; m <<= 1) x += shfl_xor_acc(x, m);
        return x;
    }
}

template <typename V, typename I, int CL>
__global__ __launch_bounds__(kBlock) void sptrsm_syncfree_kernel(const TrsmParams P) {

    using S = Sentinel<V>;
    using Bits = typename S::Bits;
    using A = typename VT<V>::Acc;
    constexpr int EP = kWave / CL;

    const int lane = threadIdx.x & (kWave - 1);
    // the EP entry lanes of a column are NEIGHBOURS (lane = column·EP + entry): their sums are DPP row operations (a few cycles
    // each) instead of ds_bpermute round trips — this reduction sits on the critical path of every dependency hop
    const int cl = lane / EP;
    const int ep = lane % EP;
    const int tile = blockIdx.y;
    const int64_t c = (int64_t)tile * CL + cl;
    const bool col_ok = c < P.p;

    const I* __restrict__ ptr = static_cast<const I*>(P.ptr);
    const I* __restrict__ idx = static_cast<const I*>(P.idx);
    const I* __restrict__ perm = static_cast<const I*>(P.perm);
    const V* __restrict__ val = static_cast<const V*>(P.val);
    const V* __restrict__ B = static_cast<const V*>(P.B);
    Bits* X = static_cast<Bits*>(P.X);
    TrsmWork* work = P.work;

    const int slot = threadIdx.x / kWave;
    const int wg = P.wgc > 1 ? (int)(blockIdx.x % (unsigned)P.wgc) : 0;
    const int cls = wg * kTrsmWaves + slot;                  // this wave's class: rows ≡ cls (mod classes)
    const int ncls = P.wgc * kTrsmWaves;
    unsigned long long* const counter = &work->class_ticket[(slot * 64 + (P.wgc > 1 ? wg : tile)) * 16];
    for (;;) {
        unsigned long long t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __shfl(t, 0, kWave) * ncls + cls;
        if (t >= (unsigned long long)P.n) break;
        const int64_t row = P.lower ? (int64_t)t : P.n - 1 - (int64_t)t;
        const int64_t s = (int64_t)ptr[row];
        const int64_t e = (int64_t)ptr[row + 1];

        A acc = 0;
        A diag = 0;
        A dsum = 1;           // the diagonal (unit: 1): every value type DIVIDES by it, like the reference's backend (_compat.py:42-48) —
                              // a reciprocal prepared ahead of the last dependency rounds differently (<= 1 ulp per row, compounding
                              // along C3's 2 673-level chains) and buys nothing measurable on a 0.8 us hop
        bool dead = false;
        // the right-hand side is requested BEFORE the row waits for its dependencies: its latency is off the critical path
        A rhs = 0;
        if (ep == 0 && col_ok) rhs = VT<V>::up(B[row * P.ldb + c * P.bcs]);
        for (int64_t base = s; base < e; base += EP) {
            // entries are visited farthest-dependency first: ascending columns for a lower sweep, descending for an
            // upper one.  The nearest rows are the ones solved last (the critical path), so everything else of the row
            // is already accumulated when they arrive (upper sweeps walked ascending cost 5.1 instead of 3.2 ms at C3).
            const int64_t k = P.lower ? base + ep : (e - 1) - (base - s) - ep;
            bool need = false;
            int64_t j = 0;
            A a = 0;
            if (k >= s && k < e) {
                j = (int64_t)idx[k];
                a = VT<V>::up(val[perm ? (int64_t)perm[k] : k]);
                if (j == row) {
                    diag += a;
                } else {
                    need = col_ok && (P.lower ? j < row : j > row);
                }
            }
            if (base + EP >= e && !P.unit) dsum = entry_sum<A, EP>(diag);      // last round: the diagonal is among these entries
            Bits xb = S::kTag;
            // poll: back-to-back agent-scope loads (the hop latency of the solve's critical path is the
            // time between the producer's store and the first poll that sees it); the wall clock and the
            // error word are only consulted every 256 spins.
#ifndef TSGU_TRSM_SLEEP
#define TSGU_TRSM_SLEEP 1
#endif
            long long t0 = 0;
            for (unsigned spin = 0;; ++spin) {
                if (need && xb == S::kTag) {
                    xb = __hip_atomic_load(X + j * P.ldx + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!__any(need && xb == S::kTag)) break;
                if (TSGU_TRSM_SLEEP > 0) __builtin_amdgcn_s_sleep(TSGU_TRSM_SLEEP);
                if ((spin & 255u) == 255u) {
                    const long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    if (now - t0 > P.timeout_ticks ||
                        __hip_atomic_load(&work->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        dead = true;
                        break;
                    }
                }
            }
            if (dead) break;
            if (need) acc = fma(a, S::val(xb), acc);
        }
        if (__any(dead)) {
            if (lane == 0) __hip_atomic_store(&work->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if (s == e && !P.unit) dsum = 0;   // an empty row of a non-unit solve: a zero diagonal — inf / NaN like the reference's backend,
                                           // not a silent x = rhs
        acc = entry_sum<A, EP>(acc);
        if (ep == 0 && col_ok) {
            const A x = (rhs - acc) / dsum;      // (correctly rounded division; unit: dsum = 1)
            Bits xb = S::bits(x);
            if (x != x) xb = S::kCanon;
            __hip_atomic_store(X + row * P.ldx + c, xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename V, typename I>
int sptrsm_launch(const TrsmParams& P, int n_cu, int wg_per_cu, hipStream_t stream) {
    // fill X with the "not ready" tag and reset tickets
    {
        int64_t nb = (P.n * P.p + kBlock - 1) / kBlock;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL((sptrsm_fill_kernel<V>), dim3((unsigned)nb), dim3(kBlock), 0, stream, P.X, P.ldx, P.n, P.p, P.work);
        if (const int rc = check_launch()) return rc;
    }
    const int cl = P.p >= 64 ? 64 : next_pow2(P.p);
    const int64_t tiles = (P.p + cl - 1) / cl;
    if (tiles > 64) return TSGU_ERR_TOO_LARGE;
    // persistent grid, `wg_per_cu` workgroups (4 waves each) per CU, default ONE.  Every resident wave polls, and the hop latency is
    // paid in the consumer CU's memory queue: 32 polling waves per CU measured 1.46 us per dependency level at C3 (2 673 levels,
    // ~100 rows each), 4 per CU 1.21 us.  A SHALLOW pattern (the reference's published shape: one random off-diagonal entry per
    // row, a few dozen levels of thousands of rows) is bound by the rows in flight instead — a wave spends ~3 us of dependent
    // loads per row — and wants every wave the CU can hold.  The caller chooses (a measured choice per pattern in
    // sparse_solve.py); the result does not depend on it.  Never more waves than rows.
    if (wg_per_cu < 1) wg_per_cu = 1;
    if (wg_per_cu > 8) wg_per_cu = 8;
    int64_t blocks = (int64_t)n_cu * wg_per_cu;

    const int64_t need = (P.n + 3) / 4;
    if (blocks > need) blocks = need;
    if (tiles > 1) {
        blocks = blocks / tiles;
        if (blocks < 1) blocks = 1;
    }
    TrsmParams Q = P;
    Q.wgc = tiles == 1 ? (int)(blocks < 64 ? blocks : 64) : 1;
    const dim3 grid((unsigned)blocks, (unsigned)tiles, 1);
#define TSGU_TRSM_CASE(N)                                                                                 \
    case N:                                                                                               \
        hipLaunchKernelGGL((sptrsm_syncfree_kernel<V, I, N>), grid, dim3(kBlock), 0, stream, Q);          \
        break;
    switch (cl) {
        TSGU_TRSM_CASE(1)
        TSGU_TRSM_CASE(2)
        TSGU_TRSM_CASE(4)
        TSGU_TRSM_CASE(8)
        TSGU_TRSM_CASE(16)
        TSGU_TRSM_CASE(32)
        TSGU_TRSM_CASE(64)
    }
#undef TSGU_TRSM_CASE
    return check_launch();
}

}  // namespace tsgu

using namespace tsgu;

extern "C" {

int64_t tsgu_sptrsm_work_bytes(int64_t, int64_t) { return (int64_t)sizeof(TrsmWork); }

int tsgu_csr_sptrsm(int vtype, int itype, int64_t n, int64_t nnz,
                    const void* ptr, const void* idx, const void* perm, const void* val,
                    int lower, int unit,
                    const void* B, int64_t ldb, int64_t b_col_stride, void* X, int64_t ldx, int64_t p,
                    void* work, int workgroups_per_cu, int device, void* stream) {
    if (n < 0 || nnz < 0 || p < 0) return TSGU_ERR_BAD_ARG;
    if (n == 0 || p == 0) return TSGU_OK;
    if (!ptr || !B || !X || !work || (nnz > 0 && (!idx || !val))) return TSGU_ERR_BAD_ARG;
    if (b_col_stride < 1 || ldb < 1 || (b_col_stride == 1 && ldb < p) || ldx < p || B == X) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    // compute-unit count per device ordinal: queried once (hipDeviceGetAttribute costs microseconds on every solve)
    static int cu_cache[64] = {0};
    int n_cu = device < 64 ? cu_cache[device] : 0;
    if (n_cu == 0) {
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return TSGU_ERR_RUNTIME;
        if (device < 64) cu_cache[device] = n_cu;
    }
    TrsmParams P{};
    P.n = n;
    P.p = p;
    P.ptr = ptr;
    P.idx = idx;
    P.perm = perm;
    P.val = val;
    P.B = B;
    P.ldb = ldb;
    P.bcs = b_col_stride;
    P.X = X;
    P.ldx = ldx;
    P.work = static_cast<TrsmWork*>(work);
    P.lower = lower;
    P.unit = unit;
    P.timeout_ticks = 400000000LL;  // 4 s at the 100 MHz wall clock
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vtype == TSGU_F32) {
        if (itype == TSGU_I32) return sptrsm_launch<float, int32_t>(P, n_cu, workgroups_per_cu, s);
        if (itype == TSGU_I64) return sptrsm_launch<float, int64_t>(P, n_cu, workgroups_per_cu, s);
    } else if (vtype == TSGU_F64) {
        if (itype == TSGU_I32) return sptrsm_launch<double, int32_t>(P, n_cu, workgroups_per_cu, s);
        if (itype == TSGU_I64) return sptrsm_launch<double, int64_t>(P, n_cu, workgroups_per_cu, s);
    } else if (vtype == TSGU_BF16) {
        if (itype == TSGU_I32) return sptrsm_launch<bf16_t, int32_t>(P, n_cu, workgroups_per_cu, s);
        if (itype == TSGU_I64) return sptrsm_launch<bf16_t, int64_t>(P, n_cu, workgroups_per_cu, s);
    }
    return TSGU_ERR_BAD_DTYPE;
}

}  // extern "C"
