// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes,
// 256-thread workgroups (one wave per SIMD), 8 XCDs with private L2s.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/tsgu_hip.h"

namespace tsgu {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kXcd = 8;

// bf16 is storage only: one 16-bit word, arithmetic in fp32.
struct bf16_t {
    uint16_t bits;
};

template <typename V>
struct VT;
template <>
struct VT<float> {
    using Acc = float;
    static constexpr int kWide = 4;  // elements per 16-byte access
    __device__ static __forceinline__ float up(float v) { return v; }
    __device__ static __forceinline__ float down(float a) { return a; }
};
template <>
struct VT<double> {
    using Acc = double;
    static constexpr int kWide = 2;
    __device__ static __forceinline__ double up(double v) { return v; }
    __device__ static __forceinline__ double down(double a) { return a; }
};
template <>
struct VT<bf16_t> {
    using Acc = float;
    static constexpr int kWide = 8;
    __device__ static __forceinline__ float up(bf16_t v) { return __uint_as_float(((uint32_t)v.bits) << 16); }
    // round-to-nearest-even, NaN kept quiet
    __device__ static __forceinline__ bf16_t down(float a) {
        uint32_t u = __float_as_uint(a);
        bf16_t r;
        if ((u & 0x7fffffffu) > 0x7f800000u) {
            r.bits = (uint16_t)((u >> 16) | 0x0040u);
        } else {
            u += 0x7fffu + ((u >> 16) & 1u);
            r.bits = (uint16_t)(u >> 16);
        }
        return r;
    }
};

// Workgroup b is observed to run on XCD (b % 8).  Give every XCD one contiguous run of
// virtual blocks so that neighbouring row blocks (which gather overlapping RHS rows for
// banded / stencil matrices) share one L2.  Bijective for any nblocks; speed only.
__device__ __forceinline__ int64_t xcd_chunked_block(int64_t bid, int64_t nblocks) {
    const int64_t q = nblocks / kXcd, r = nblocks % kXcd;
    const int64_t x = bid % kXcd, l = bid / kXcd;
    return x * q + (x < r ? x : r) + l;
}

// ---- streaming (non-temporal) accesses --------------------------------------------------
// Data that is touched exactly once (the col/val stream, the output rows) is loaded/stored with
// the `nt` policy so that it does not evict the gathered RHS rows, which are the only operand
// with reuse, from the 4 MiB per-XCD L2.
#ifndef TSGU_NT_STAGE
#define TSGU_NT_STAGE 1
#endif
#ifndef TSGU_NT_STORE
#define TSGU_NT_STORE 1
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
#if TSGU_NT_STAGE
    if constexpr (sizeof(T) == 2) {
        const unsigned short b = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(p));
        T r;
        __builtin_memcpy(&r, &b, 2);
        return r;
    } else {
        return __builtin_nontemporal_load(p);
    }
#else
    return *p;
#endif
}

__device__ __forceinline__ void stream_store16(void* ptr, uint4 raw) {
#if TSGU_NT_STORE
    u32x4_t v = {raw.x, raw.y, raw.z, raw.w};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(ptr));
#else
    *reinterpret_cast<uint4*>(ptr) = raw;
#endif
}

// ---- VEC-element loads/stores of value type V into accumulator registers ------------
template <typename V, int VEC>
__device__ __forceinline__ void load_vec(const V* __restrict__ ptr, typename VT<V>::Acc (&out)[VEC]) {
    if constexpr (VEC == 1) {
        out[0] = VT<V>::up(*ptr);
    } else {
        static_assert(VEC == VT<V>::kWide, "vector width must be 1 or 16 bytes");
        const uint4 raw = *reinterpret_cast<const uint4*>(ptr);
        if constexpr (std::is_same<V, float>::value) {
            out[0] = __uint_as_float(raw.x);
            out[1] = __uint_as_float(raw.y);
            out[2] = __uint_as_float(raw.z);
            out[3] = __uint_as_float(raw.w);
        } else if constexpr (std::is_same<V, double>::value) {
            out[0] = __hiloint2double((int)raw.y, (int)raw.x);
            out[1] = __hiloint2double((int)raw.w, (int)raw.z);
        } else {
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                out[2 * i] = __uint_as_float(w[i] << 16);
                out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            }
        }
    }
}

template <typename V, int VEC, bool STREAM = false>
__device__ __forceinline__ void store_vec(V* __restrict__ ptr, const typename VT<V>::Acc (&in)[VEC]) {
    if constexpr (VEC == 1) {
        *ptr = VT<V>::down(in[0]);
    } else {
        uint4 raw;
        if constexpr (std::is_same<V, float>::value) {
            raw.x = __float_as_uint(in[0]);
            raw.y = __float_as_uint(in[1]);
            raw.z = __float_as_uint(in[2]);
            raw.w = __float_as_uint(in[3]);
        } else if constexpr (std::is_same<V, double>::value) {
            raw.x = (uint32_t)__double2loint(in[0]);
            raw.y = (uint32_t)__double2hiint(in[0]);
            raw.z = (uint32_t)__double2loint(in[1]);
            raw.w = (uint32_t)__double2hiint(in[1]);
        } else {
            // gfx950 converts two floats to packed bf16 (round to nearest even, NaN kept quiet) in ONE instruction; the
            // bit-twiddled VT::down costs ~6 VALU per value, i.e. ~90 per lane at the end of every bf16 kernel
            uint32_t w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(in[2 * i]), "v"(in[2 * i + 1]));
            }
            raw.x = w[0];
            raw.y = w[1];
            raw.z = w[2];
            raw.w = w[3];
        }
        if constexpr (STREAM) stream_store16(ptr, raw);
        else *reinterpret_cast<uint4*>(ptr) = raw;
    }
}

// Offset (in elements) of dense row j with a leading dimension < 2^32: one v_mad_u64_u32.
__device__ __forceinline__ uint64_t row_off(int j, uint32_t ld) { return (uint64_t)(uint32_t)j * ld; }

// ---- cross-lane sums inside an aligned group of CL lanes (CL <= 16: DPP, no LDS traffic) ----
// DPP controls: quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141
// (lane i <-> 7-i of each 8), row_mirror = 0x140 (lane i <-> 15-i of each 16).  After each step
// both partners hold the same partial sum, so the sequence is an all-reduce over the group.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) { return dpp_f32<CTRL>(x); }
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) { return dpp_f64<CTRL>(x); }

template <typename Acc, int CL>
__device__ __forceinline__ Acc group_sum(Acc x) {
    static_assert(CL >= 1 && CL <= 64 && (CL & (CL - 1)) == 0, "CL must be a power of two");
    if constexpr (CL >= 2) x += dpp_move<0xB1>(x);
    if constexpr (CL >= 4) x += dpp_move<0x4E>(x);
    if constexpr (CL >= 8) x += dpp_move<0x141>(x);
    if constexpr (CL >= 16) x += dpp_move<0x140>(x);
    if constexpr (CL >= 32) x += __shfl_xor(x, 16, 64);
    if constexpr (CL >= 64) x += __shfl_xor(x, 32, 64);
    return x;
}

// The same sum for groups of 32 / 64 lanes WITHOUT the LDS round trip of `__shfl_xor` (ds_bpermute): the rows of 16 are reduced by
// the butterfly above, then `row_bcast:15` adds lane 15 of rows 0 / 2 into every lane of rows 1 / 3 and (64 lanes) `row_bcast:31`
// adds lane 31 into rows 2, 3.  NOT an all-reduce: the total is in the lanes cl >= group_total_lane<CL>() of the group only — enough
// for a dot product that ONE lane stores.  Same association as the butterfly (S(row 0) + S(row 1)): the same bits.
template <int CL>
constexpr int group_total_lane() { return CL <= 16 ? 0 : (CL == 32 ? 16 : 48); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_rows_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));      // (rows outside the mask: 0)
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_rows_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float dpp_bcast15(float x) { return dpp_rows_f32<0x142, 0xA>(x); }
__device__ __forceinline__ double dpp_bcast15(double x) { return dpp_rows_f64<0x142, 0xA>(x); }
__device__ __forceinline__ float dpp_bcast31(float x) { return dpp_rows_f32<0x143, 0xC>(x); }
__device__ __forceinline__ double dpp_bcast31(double x) { return dpp_rows_f64<0x143, 0xC>(x); }

template <typename Acc, int CL>
__device__ __forceinline__ Acc group_total(Acc x) {
    static_assert(CL >= 1 && CL <= 64 && (CL & (CL - 1)) == 0, "CL must be a power of two");
    if constexpr (CL <= 16) {
        return group_sum<Acc, CL>(x);
    } else {
        x = group_sum<Acc, 16>(x);
        x += dpp_bcast15(x);
        if constexpr (CL == 64) x += dpp_bcast31(x);
        return x;
    }
}

// Sum over the EP entry-lanes of a row group (lanes cl + CL*e, e < EP): xor strides CL, 2CL, ... below
// CL*EP.  The geometries in use have 8-lane groups ((CL,EP) = (1,8), (2,4), (4,2)): strides 1 and 2 are
// quad permutes, stride 4 is half-mirror followed by a quad reversal (i -> 7-i -> i^4); no LDS traffic.
template <typename Acc, int CL, int EP>
__device__ __forceinline__ Acc ep_sum(Acc x) {
    if constexpr (EP == 1) {
        return x;
    } else if constexpr (CL * EP == 8) {
        if constexpr (CL == 1) x += dpp_move<0xB1>(x);
        if constexpr (CL <= 2) x += dpp_move<0x4E>(x);
        x += dpp_move<0x1B>(dpp_move<0x141>(x));
        return x;
    } else {
#pragma unroll
        for (int m = CL; m < CL * EP; m <<= 1) x += __shfl_xor(x, m, 64);
        return x;
    }
}

// xor-shuffle for float / double accumulators (all 64 lanes participate)
__device__ __forceinline__ float shfl_xor_acc(float v, int mask) { return __shfl_xor(v, mask, kWave); }
__device__ __forceinline__ double shfl_xor_acc(double v, int mask) { return __shfl_xor(v, mask, kWave); }

// ---- host side -----------------------------------------------------------------------
inline int check_launch() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? TSGU_OK : TSGU_ERR_LAUNCH;
}

inline int set_device(int device) {
    if (device < 0) return TSGU_ERR_BAD_ARG;
    return hipSetDevice(device) == hipSuccess ? TSGU_OK : TSGU_ERR_RUNTIME;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int next_pow2(int64_t x) {
    int r = 1;
    while (r < x) r <<= 1;
    return r;
}

// Lane geometry for "a group of lanes owns a row":  CL column lanes × EP entry lanes,
// every lane holding VEC consecutive columns.  Groups are at least 8 lanes wide so the
// staged column/value stream is read by few distinct LDS addresses per wave.
struct RowGeom {
    int vec;      // 1 or kWide
    int cl;       // column lanes per row (power of two, <= 64)
    int ep;       // entry-parallel lanes per row
    int64_t col_tiles;  // grid.z : tiles of cl*vec columns
};

inline RowGeom pick_geom(int wide, bool can_wide, int64_t p) {
    RowGeom g;
    g.vec = can_wide ? wide : 1;
    const int64_t ncl = (p + g.vec - 1) / g.vec;
    g.cl = ncl >= 64 ? 64 : next_pow2(ncl);
    g.ep = g.cl >= 8 ? 1 : 8 / g.cl;
    g.col_tiles = (ncl + g.cl - 1) / g.cl;
    return g;
}

// One lane per row (CL = EP = 1) for operands whose dense row is a single 16-byte access and whose sparse rows are
// short: consecutive lanes own consecutive rows, so for banded / stencil patterns the k-th gathers of a wave fall on
// consecutive dense rows — one contiguous kilobyte instead of 64 separate lines through L1 (C4's 7-point Laplacian
// with 4 right-hand sides: K1 with the dot epilogue 85 -> 43 us).  Longer rows keep 8 entry lanes per row.
// `max_row_nnz` (0 = unknown) guards ragged patterns: one very long row in a short-row matrix would be walked serially
// by a single lane (K1 and the CG dot epilogue with it), so the geometry is only chosen when no row is much longer
// than the average.
inline void prefer_row_per_lane(RowGeom& g, int64_t n_rows, int64_t nnz, int64_t max_row_nnz = 0) {
    if (g.cl == 1 && g.vec > 1 && n_rows > 0 && nnz <= 16 * n_rows &&
        (max_row_nnz <= 0 || max_row_nnz <= 4 * (nnz / n_rows + 1) + 16))
        g.ep = 1;
}

template <typename F>
inline int dispatch_geom(const RowGeom& g, F&& f) {
    // f.template operator()<CL, EP>()
    switch (g.cl) {
        case 1:
            if (g.ep == 1) return f(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            return f(std::integral_constant<int, 1>{}, std::integral_constant<int, 8>{});
        case 2: return f(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
        case 4: return f(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
        case 8: return f(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
        case 16: return f(std::integral_constant<int, 16>{}, std::integral_constant<int, 1>{});
        case 32: return f(std::integral_constant<int, 32>{}, std::integral_constant<int, 1>{});
        case 64: return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 1>{});
    }
    return TSGU_ERR_BAD_ARG;
}

}  // namespace tsgu
