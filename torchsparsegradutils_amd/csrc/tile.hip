// Row-block tile kernels: extern "C" entry points (declared in include/tsgu_hip.h) and launch configuration.
#include "tile_impl.h"

#include <cstdlib>

using namespace tsgu;

namespace {

int n_cu_of(int device) {
    static int cache[64] = {0};
    int n = device < 64 ? cache[device] : 0;
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
        if (device < 64) cache[device] = n;
    }
    return n;
}

int fill(TileParams& P, const tsgu_tile_plan* pl, int64_t p) {
    if (!pl || pl->n_rows < 0 || pl->n_cols < 0 || pl->nnz < 0 || pl->n_blocks < 0) return TSGU_ERR_BAD_ARG;
    // (max_union is also the tile row the padding of the entry records names — the kernel's zero row: it must be the kernel's own limit)
    if (pl->rows_per_block != kTileRows || pl->max_union != kTileUMax || pl->max_entries > kTileEMax) return TSGU_ERR_BAD_ARG;
    if (pl->n_blocks > 0 && (!pl->desc || !pl->ucol || !pl->lidx || !pl->rptr || !pl->ent || !pl->xrow)) return TSGU_ERR_BAD_ARG;
    if ((pl->cpos == nullptr) != (pl->cslot == nullptr)) return TSGU_ERR_BAD_ARG;       // (value chunks and their slots: both or neither)
    if (pl->cpos && pl->nnz > 0 && pl->nnz < 4) return TSGU_ERR_BAD_ARG;                // (a chunk is four values inside the value array)
    if (pl->n_rows > 0x7fffffffLL || pl->nnz > 0x7fffffffLL) return TSGU_ERR_TOO_LARGE;
    if (p <= 0 || p % 32) return TSGU_ERR_BAD_ARG;        // (column tiles of 32: dense rows of 128 bytes per launch)
    P.n_rows = pl->n_rows;
    P.n_cols = pl->n_cols;
    P.nnz = pl->nnz;
    P.n_blocks = pl->n_blocks;
    P.desc = static_cast<const TileDesc*>(pl->desc);
    P.ucol = static_cast<const int*>(pl->ucol);
    P.lidx = static_cast<const unsigned char*>(pl->lidx);
    P.rptr = static_cast<const int*>(pl->rptr);
    P.ent = static_cast<const uint4*>(pl->ent);
    P.xrow = static_cast<const unsigned short*>(pl->xrow);
    P.cpos = static_cast<const int*>(pl->cpos);
    P.cslot = static_cast<const uint2*>(pl->cslot);
    P.ncol = (int)(p / 32);
    return TSGU_OK;
}

template <int MODE>
int launch(const TileParams& P0, int device, hipStream_t s) {
    TileParams P = P0;
    const int n_cu = n_cu_of(device);
    if (n_cu <= 0) return TSGU_ERR_RUNTIME;
    // persistent workgroups, two per CU (76 KB of LDS each): neighbouring blocks share tile rows, consecutive workgroups share an
    // XCD's L2 (xcd_chunked_block)
    const int64_t slots = (int64_t)n_cu * 2;
    int64_t per = (P.n_blocks + slots - 1) / slots;
    if (per < 1) per = 1;
    const int64_t grid = (P.n_blocks + per - 1) / per;
    P.blocks_per_wg = (int)per;
    // block schedule: cyclic by default (TSGU_TILE_CYCLIC=0: a run of consecutive blocks per workgroup) — see tile_impl.h
    static const int cyclic = [] {
        const char* e = getenv("TSGU_TILE_CYCLIC");
        return e ? (e[0] != '0') : 1;
    }();
    P.cyclic = cyclic;
    using L = TileLds<128>;
    // the opt-in to more than 64 KB of dynamic LDS is a per-DEVICE attribute of the kernel: one bit per (variant, device)
    static std::atomic<uint64_t> attr_set[6] = {{0}, {0}, {0}, {0}, {0}, {0}};
    if (device < 0 || device >= 64) return TSGU_ERR_BAD_ARG;
    const bool perm = P.cpos != nullptr;
    auto go = [&](auto kern, int slot) -> int {
        if (!(attr_set[slot].load(std::memory_order_acquire) >> device & 1ull)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::kTotal) != hipSuccess)
                return TSGU_ERR_RUNTIME;
            attr_set[slot].fetch_or(1ull << device, std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kTileThreads), L::kTotal, s, P);
        return check_launch();
    };
    const bool wide = P.ncol > 1;
    if constexpr (MODE == kTileSpmm) {
        if (perm) return wide ? go(tile_kernel<float, 8, kTileSpmm, true, true>, 0) : go(tile_kernel<float, 8, kTileSpmm, true, false>, 1);
        return wide ? go(tile_kernel<float, 8, kTileSpmm, false, true>, 2) : go(tile_kernel<float, 8, kTileSpmm, false, false>, 3);
    } else {
        return wide ? go(tile_kernel<float, 8, kTileSddmm, false, true>, 4) : go(tile_kernel<float, 8, kTileSddmm, false, false>, 5);
    }
}

}  // namespace

extern "C" {

int tsgu_tile_geometry(int vtype, int64_t p, int* rows_per_block, int* max_union, int* max_entries) {
    if (vtype != TSGU_F32 || p <= 0 || p % 32 || p > 1024) return TSGU_ERR_BAD_DTYPE;
    if (rows_per_block) *rows_per_block = kTileRows;
    if (max_union) *max_union = kTileUMax;
    if (max_entries) *max_entries = kTileEMax;
    return TSGU_OK;
}

int tsgu_csr_spmm_tile(int vtype, const tsgu_tile_plan* plan, const void* val, const void* B, int64_t ldb, void* C, int64_t ldc,
                       int64_t p, int device, void* stream) {
    if (vtype != TSGU_F32) return TSGU_ERR_BAD_DTYPE;
    TileParams P{};
    if (const int rc = fill(P, plan, p)) return rc;
    if (P.n_rows == 0) return TSGU_OK;
    if (!B || !C || (P.nnz > 0 && !val) || ldb < p || ldc < p) return TSGU_ERR_BAD_ARG;
    if (!aligned16(B) || !aligned16(C) || ldb % 4 || ldc % 4 || !aligned16(plan->ent)) return TSGU_ERR_BAD_ARG;
    if ((uint64_t)plan->n_cols * (uint64_t)ldb * 4u > 0xffffffffull || (uint64_t)P.nnz * 4u > 0xffffffffull) return TSGU_ERR_TOO_LARGE;
    if (plan->n_cols >= (1 << 24) || ldb * 4 >= (1 << 24)) return TSGU_ERR_TOO_LARGE;      // (tile row offsets are 24-bit products)
    if (const int rc = set_device(device)) return rc;
    P.val = val;
    P.lds_ = ldb;
    P.ldo = ldc;
    P.S = B;                                      // wide operands (the reference's SuiteSparse width is 128): the column tiles of 32 are steps
    P.out = C;                                    // of ONE launch — a block's values, entry bytes and row pointers are staged once
    return launch<kTileSpmm>(P, device, static_cast<hipStream_t>(stream));
}

int tsgu_csr_sddmm_tile(int vtype, const tsgu_tile_plan* plan, const void* R, int64_t ldr, const void* Cm, int64_t ldc, void* out_vals,
                        double alpha, int64_t p, int device, void* stream) {
    if (vtype != TSGU_F32) return TSGU_ERR_BAD_DTYPE;
    TileParams P{};
    if (const int rc = fill(P, plan, p)) return rc;
    if (P.n_rows == 0 || P.nnz == 0) return TSGU_OK;
    if (plan->cpos || !R || !Cm || !out_vals || ldr < p || ldc < p) return TSGU_ERR_BAD_ARG;
    if (!aligned16(R) || !aligned16(Cm) || ldr % 4 || ldc % 4) return TSGU_ERR_BAD_ARG;
    if ((uint64_t)plan->n_cols * (uint64_t)ldc * 4u > 0xffffffffull) return TSGU_ERR_TOO_LARGE;
    if (plan->n_cols >= (1 << 24) || ldc * 4 >= (1 << 24)) return TSGU_ERR_TOO_LARGE;
    if (const int rc = set_device(device)) return rc;
    P.ldown = ldr;
    P.lds_ = ldc;
    P.gvals = out_vals;
    P.alpha = (float)alpha;
    P.Own = R;                                    // (the dots of the column tiles are added up inside the kernel)
    P.S = Cm;
    return launch<kTileSddmm>(P, device, static_cast<hipStream_t>(stream));
}

}  // extern "C"
