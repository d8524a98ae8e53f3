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
    if (pl->rows_per_block != kTileRows || pl->max_union > kTileUMax || pl->max_entries > kTileEMax) return TSGU_ERR_BAD_ARG;
    if (pl->n_blocks > 0 && (!pl->desc || !pl->ucol || !pl->lidx || !pl->rptr)) return TSGU_ERR_BAD_ARG;
    if ((pl->perm == nullptr) != (pl->slot == nullptr)) return TSGU_ERR_BAD_ARG;        // (source positions and their slots: both or neither)
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
    P.perm = static_cast<const int*>(pl->perm);
    P.slot = static_cast<const unsigned short*>(pl->slot);
    return TSGU_OK;
}

template <int MODE>
int launch(const TileParams& P0, int device, hipStream_t s) {
    TileParams P = P0;
    const int n_cu = n_cu_of(device);
    if (n_cu <= 0) return TSGU_ERR_RUNTIME;
    // persistent workgroups, two per CU (76 KB of LDS each), every one a run of consecutive blocks: neighbouring blocks share tile
    // rows, consecutive workgroups share an XCD's L2 (xcd_chunked_block)
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
    static std::atomic<int> attr_set[4] = {{0}, {0}, {0}, {0}};
    const bool perm = P.perm != nullptr;
    auto go = [&](auto kern, int slot) -> int {
        if (!attr_set[slot].load(std::memory_order_relaxed)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::kTotal) != hipSuccess)
                return TSGU_ERR_RUNTIME;
            attr_set[slot].store(1, std::memory_order_relaxed);
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kTileThreads), L::kTotal, s, P);
        return check_launch();
    };
    if constexpr (MODE == kTileSpmm) {
        if (perm) return go(tile_kernel<float, 8, kTileSpmm, true>, 0);
        return go(tile_kernel<float, 8, kTileSpmm, false>, 1);
    } else {
        return go(tile_kernel<float, 8, kTileSddmm, false>, 2);
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
    if (!aligned16(B) || !aligned16(C) || ldb % 4 || ldc % 4) return TSGU_ERR_BAD_ARG;
    if ((uint64_t)plan->n_cols * (uint64_t)ldb * 4u > 0xffffffffull || (uint64_t)P.nnz * 4u > 0xffffffffull) return TSGU_ERR_TOO_LARGE;
    if (plan->n_cols >= (1 << 24) || ldb * 4 >= (1 << 24)) return TSGU_ERR_TOO_LARGE;      // (tile row offsets are 24-bit products)
    if (const int rc = set_device(device)) return rc;
    P.val = val;
    P.lds_ = ldb;
    P.ldo = ldc;
    for (int64_t c = 0; c < p; c += 32) {        // wide operands: one launch per tile of 32 columns (the reference's SuiteSparse width is 128)
        P.S = static_cast<const float*>(B) + c;
        P.out = static_cast<float*>(C) + c;
        if (const int rc = launch<kTileSpmm>(P, device, static_cast<hipStream_t>(stream))) return rc;
    }
    return TSGU_OK;
}

int tsgu_csr_sddmm_tile(int vtype, const tsgu_tile_plan* plan, const void* R, int64_t ldr, const void* Cm, int64_t ldc, void* out_vals,
                        double alpha, int64_t p, int device, void* stream) {
    if (vtype != TSGU_F32) return TSGU_ERR_BAD_DTYPE;
    TileParams P{};
    if (const int rc = fill(P, plan, p)) return rc;
    if (P.n_rows == 0 || P.nnz == 0) return TSGU_OK;
    if (plan->perm || !R || !Cm || !out_vals || ldr < p || ldc < p) return TSGU_ERR_BAD_ARG;
    if (!aligned16(R) || !aligned16(Cm) || ldr % 4 || ldc % 4) return TSGU_ERR_BAD_ARG;
    if ((uint64_t)plan->n_cols * (uint64_t)ldc * 4u > 0xffffffffull) return TSGU_ERR_TOO_LARGE;
    if (plan->n_cols >= (1 << 24) || ldc * 4 >= (1 << 24)) return TSGU_ERR_TOO_LARGE;
    if (const int rc = set_device(device)) return rc;
    P.ldown = ldr;
    P.lds_ = ldc;
    P.gvals = out_vals;
    P.alpha = (float)alpha;
    for (int64_t c = 0; c < p; c += 32) {        // the dots of the later column tiles are added to those of the first
        P.Own = static_cast<const float*>(R) + c;
        P.S = static_cast<const float*>(Cm) + c;
        P.accumulate = c > 0;
        if (const int rc = launch<kTileSddmm>(P, device, static_cast<hipStream_t>(stream))) return rc;
    }
    return TSGU_OK;
}

}  // extern "C"
