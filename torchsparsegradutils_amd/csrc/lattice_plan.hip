// Row analysis for lattice plans (plan building, once per sparsity pattern): the per-entry part of
// _lattice.build_lattice_plan as two small kernels instead of ~40 tensor ops over the nnz entries.
//
//   stored-order walk (kind 0): one thread per row computes the displacement code of every stored entry
//       row = ((item*nx + x)*ny + y)*nz + z,  col = the point at (dx, dy, dz) of it, |dx| <= 1, |dy|, |dz| <= 2, wraps allowed
//       code = ((dx+1)*5 + (dy+2))*5 + (dz+2)
//     and either the 64-bit hash of the row's code sequence (rows with equal hashes are a class candidate) or — when a
//     class table is given — compares the sequence with the table row of the row's class (the exact check).
//   transposed walk (kind 1): one thread per row j of the TRANSPOSED pattern finds its entries without any sort: the
//     candidates are the rows i = j - d for every displacement d that occurs in the pattern; row i is searched for
//     column j; hits are ordered by i (the order of the stable transposition) and carry code(j -> i)*32 + position in row i.
// The hash is _lattice._mix bit for bit, so the classes (and their numbering) are the ones the tensor-op builder finds.
#include "tsgu_common.h"

namespace tsgu {

constexpr int kLatMaxLen = 32;

struct LatDims {
    int nb, nx, ny, nz;
};

__host__ __device__ inline int64_t lat_mix(int64_t a, int64_t b) {
    // wrap-around int64 arithmetic (as torch's), arithmetic right shift
    uint64_t x = (uint64_t)a * (uint64_t)(-7046029254386353131LL) + (uint64_t)b * (uint64_t)(-4417276706812531889LL) + (uint64_t)1609587929392839161LL;
    int64_t sx = (int64_t)x;
    sx = sx ^ (sx >> 29);
    return (int64_t)((uint64_t)sx * (uint64_t)(-49064778989728563LL));
}

// displacement code of (row -> col) on the lattice, or -1 when col is not a neighbour within reach
__device__ __forceinline__ int lat_code(int64_t row, int64_t col, const LatDims& D, int& ady, int& adz) {
    const int64_t d2 = (int64_t)D.ny * D.nz;
    const int64_t X = row / d2, Xc = col / d2;
    const int rem = (int)(row - X * d2), remc = (int)(col - Xc * d2);
    const int y = rem / D.nz, z = rem - y * D.nz;
    const int yc = remc / D.nz, zc = remc - yc * D.nz;
    const int64_t dX = Xc - X;
    int dx;
    if (dX >= -1 && dX <= 1) dx = (int)dX;
    else if (D.nx > 1 && (dX == D.nx - 1 || dX == -(int64_t)(D.nx - 1))) dx = dX > 0 ? -1 : 1;
    else return -1;
    int dy = (yc - y + D.ny / 2) % D.ny;
    if (dy < 0) dy += D.ny;
    dy -= D.ny / 2;
    int dz = (zc - z + D.nz / 2) % D.nz;
    if (dz < 0) dz += D.nz;
    dz -= D.nz / 2;
    if (dy < -2 || dy > 2 || dz < -2 || dz > 2) return -1;
    // the displacement must lead back to col (same item, wraps inside the item)
    const int64_t item = X / D.nx;
    const int x = (int)(X - item * D.nx);
    int xx = (x + dx) % D.nx;
    if (xx < 0) xx += D.nx;
    int yy = (y + dy) % D.ny;
    if (yy < 0) yy += D.ny;
    int zz = (z + dz) % D.nz;
    if (zz < 0) zz += D.nz;
    if (((item * D.nx + xx) * D.ny + yy) * D.nz + zz != col) return -1;
    ady = dy < 0 ? -dy : dy;
    adz = dz < 0 ? -dz : dz;
    return ((dx + 1) * 5 + (dy + 2)) * 5 + (dz + 2);
}

constexpr int kLatSlots = 1024;                       // open-addressing table of distinct row hashes (a pattern has <= 255 classes)
constexpr int64_t kLatEmpty = (int64_t)0x8000000000000000LL;

// Insert hash h into the table; returns its slot, or -1 when the table is full (more distinct rows than any plan may have).
__device__ __forceinline__ int lat_slot_of(int64_t h, int64_t row, unsigned long long* __restrict__ thash, int* __restrict__ trep) {
    if (h == kLatEmpty) h = 1;   // the sentinel itself is not a legal key
    unsigned s = (unsigned)((uint64_t)h * 0x9E3779B97F4A7C15ull >> 54);   // 10 bits
    for (int probe = 0; probe < kLatSlots; ++probe) {
        const unsigned long long cur = thash[s];
        if (cur == (unsigned long long)h || (cur == (unsigned long long)kLatEmpty &&
                                             (atomicCAS(thash + s, (unsigned long long)kLatEmpty, (unsigned long long)h) == (unsigned long long)kLatEmpty ||
                                              thash[s] == (unsigned long long)h))) {
            // (an atomic only when it can change the word: one address serves ~80 M atomics per second, and the interior class
            // alone would send a million rows to one slot — a stale read only costs a redundant atomic)
            if ((int)row < *(volatile int*)(trep + s)) atomicMin(trep + s, (int)row);
            return (int)s;
        }
        s = (s + 1) & (kLatSlots - 1);
    }
    return -1;
}

// status words: [0] rows that are not lattice rows / do not match their class, [1] max |dy|, [2] max |dz|, [3] longest row,
//               [4] (pass 2, box_mask != 0) rows that are not "the displacements of box_mask that lead to an existing neighbour"
// Pass 1 (ctable == NULL): hash of the row's code sequence -> slot[row] in the hash table (thash / trep = smallest row per slot).
//                          A row that stores a column twice is not a lattice row (every plan assumes duplicate-free rows: the
//                          transposed walk would find the column once and drop the second entry).
// Pass 2 (ctable given):   rcls[row] = remap[slot[row]]; the row's codes and length are compared with its class (exact check).
//   box_mask (27 bits, bit (dx+1)·9 + (dy+1)·3 + dz+1) and `periodic` (bit 0 / 1 / 2: x / y / z wrap) state the plane-march
//   condition: the row holds exactly the displacements of the mask whose neighbour exists — all of them in a periodic dimension,
//   the ones that stay inside the lattice in a truncated one.  Checked per row (the entries are distinct, lie in the mask by the
//   class check and must not wrap in a truncated dimension; then equal counts mean equal sets).
template <typename I>
__global__ __launch_bounds__(256) void lat_rows_kernel(int64_t n_rows, const I* __restrict__ crow, const I* __restrict__ col, LatDims D,
                                                        unsigned short* __restrict__ slot, unsigned long long* __restrict__ thash,
                                                        int* __restrict__ trep, const unsigned char* __restrict__ remap,
                                                        const int* __restrict__ ctable, const unsigned char* __restrict__ lens,
                                                        unsigned char* __restrict__ rcls, int* __restrict__ status, unsigned box_mask,
                                                        int periodic) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const int64_t e0 = (int64_t)crow[row], e1 = (int64_t)crow[row + 1];
    const int len = (int)(e1 - e0);
    bool bad = len < 0 || len > kLatMaxLen;
    int my = 0, mz = 0;
    int64_t h = 0;
    int cls = 0;
    const int* trow = nullptr;
    if (ctable) {
        cls = remap[slot[row]];
        rcls[row] = (unsigned char)cls;
        trow = ctable + (int64_t)cls * kLatMaxLen;
    }
    bool off_box = false;
    if (!bad) {
        // position of the row (the plane-march condition)
        const int64_t d2 = (int64_t)D.ny * D.nz;
        const int64_t X = row / d2;
        const int rem = (int)(row - X * d2);
        const int y = rem / D.nz, z = rem - y * D.nz;
        const int x = (int)(X % D.nx);
        const bool px = periodic & 1, py = periodic & 2, pz = periodic & 4;
        unsigned long long seen0 = 0, seen1 = 0;          // codes < 75
        for (int k = 0; k < len; ++k) {
            int ady = 0, adz = 0;
            const int code = lat_code(row, (int64_t)col[e0 + k], D, ady, adz);
            if (code < 0) {
                bad = true;
                break;
            }
            my = ady > my ? ady : my;
            mz = adz > mz ? adz : mz;
            const unsigned long long bit = 1ull << (code & 63);
            unsigned long long& seen = code < 64 ? seen0 : seen1;
            bad |= (seen & bit) != 0;                      // the same column twice
            seen |= bit;
            if (trow) bad |= trow[k] != code;
            else h += lat_mix(code, k + 17);
            if (box_mask) {
                const int dz = code % 5 - 2, dy = (code / 5) % 5 - 2, dx = code / 25 - 1;
                off_box |= dy < -1 || dy > 1 || dz < -1 || dz > 1;
                off_box |= (!px && (unsigned)(x + dx) >= (unsigned)D.nx) || (!py && (unsigned)(y + dy) >= (unsigned)D.ny) ||
                           (!pz && (unsigned)(z + dz) >= (unsigned)D.nz);
            }
        }
        if (trow) bad |= lens[cls] != len;
        else h += lat_mix(len, 3);
        if (box_mask) {
            int expect = 0;
            for (int b = 0; b < 27; ++b) {
                if (!(box_mask >> b & 1u)) continue;
                const int dx = b / 9 - 1, dy = (b / 3) % 3 - 1, dz = b % 3 - 1;
                expect += (px || (unsigned)(x + dx) < (unsigned)D.nx) && (py || (unsigned)(y + dy) < (unsigned)D.ny) &&
                          (pz || (unsigned)(z + dz) < (unsigned)D.nz);
            }
            off_box |= expect != len;
        }
    }
    if (!ctable) {
        const int s = bad ? 0 : lat_slot_of(h, row, thash, trep);
        bad |= s < 0;
        slot[row] = (unsigned short)(s < 0 ? 0 : s);
    }
    // status maxima: atomics only while they still raise the word (every row hitting three addresses was most of a pass: 12 ms at
    // 1e6 rows)
    volatile int* const st = status;
    if (bad) atomicAdd(status + 0, 1);
    if (box_mask && (bad || off_box) && st[4] == 0) atomicAdd(status + 4, 1);
    if (my > st[1]) atomicMax(status + 1, my);
    if (mz > st[2]) atomicMax(status + 2, mz);
    if (len > st[3]) atomicMax(status + 3, len);
}

// code sequences of a few rows (class representatives): out[r][k] = code, -1 beyond the row
template <typename I>
__global__ void lat_row_codes_kernel(int nrows, const int64_t* __restrict__ rows, const I* __restrict__ crow, const I* __restrict__ col,
                                     LatDims D, int* __restrict__ out) {
    const int r = blockIdx.x;
    const int k = threadIdx.x;
    if (r >= nrows || k >= kLatMaxLen) return;
    const int64_t row = rows[r];
    const int64_t e0 = (int64_t)crow[row], e1 = (int64_t)crow[row + 1];
    int code = -1;
    if (k < e1 - e0) {
        int a, b;
        code = lat_code(row, (int64_t)col[e0 + k], D, a, b);
    }
    out[r * kLatMaxLen + k] = code;
}

// Transposed rows.  disp[nd] = the displacement codes that occur in the pattern; for row j and displacement d the only row that
// can hold an entry (i, j) with code d is i = j - d.  Entries are ordered by i; seq = code(j -> i)*32 + (position of j in row i).
template <typename I>
__device__ __forceinline__ int lat_trow_entries(int64_t j, const I* __restrict__ crow, const I* __restrict__ col, const LatDims& D,
                                                const unsigned char* __restrict__ disp, int nd, int64_t (&src)[kLatMaxLen],
                                                int (&seq)[kLatMaxLen], bool& overflow) {
    const int64_t d2 = (int64_t)D.ny * D.nz;
    const int64_t X = j / d2;
    const int rem = (int)(j - X * d2);
    const int y = rem / D.nz, z = rem - y * D.nz;
    const int64_t item = X / D.nx;
    const int x = (int)(X - item * D.nx);
    int cnt = 0;
    for (int t = 0; t < nd; ++t) {
        const int code = disp[t];
        const int dz = code % 5 - 2, dy = (code / 5) % 5 - 2, dx = code / 25 - 1;
        int xx = (x - dx) % D.nx;
        if (xx < 0) xx += D.nx;
        int yy = (y - dy) % D.ny;
        if (yy < 0) yy += D.ny;
        int zz = (z - dz) % D.nz;
        if (zz < 0) zz += D.nz;
        const int64_t i = ((item * D.nx + xx) * D.ny + yy) * D.nz + zz;
        bool dup = false;
        for (int u = 0; u < cnt; ++u) dup |= src[u] == i;
        if (dup) continue;   // tiny lattices: two displacements can name the same row
        const int64_t e0 = (int64_t)crow[i], e1 = (int64_t)crow[i + 1];
        for (int64_t e = e0; e < e1; ++e) {
            if ((int64_t)col[e] == j) {
                int a, b;
                const int c2 = lat_code(j, i, D, a, b);   // where the source row lies, seen from the transposed row
                if (cnt >= kLatMaxLen || c2 < 0 || e - e0 >= kLatMaxLen) {
                    overflow = true;
                    return cnt;
                }
                // insert by ascending source row
                int pos = cnt;
                while (pos > 0 && src[pos - 1] > i) {
                    src[pos] = src[pos - 1];
                    seq[pos] = seq[pos - 1];
                    --pos;
                }
                src[pos] = i;
                seq[pos] = c2 * 32 + (int)(e - e0);
                ++cnt;
                break;   // (a row holds a column once: sorted, duplicate-free rows are a precondition of every plan)
            }
        }
    }
    return cnt;
}

template <typename I>
__global__ __launch_bounds__(256) void lat_trows_kernel(int64_t n_rows, const I* __restrict__ crow, const I* __restrict__ col, LatDims D,
                                                         const unsigned char* __restrict__ disp, int nd, unsigned short* __restrict__ slot,
                                                         unsigned long long* __restrict__ thash, int* __restrict__ trep,
                                                         const unsigned char* __restrict__ remap, const int* __restrict__ ctable,
                                                         const unsigned char* __restrict__ lens, unsigned char* __restrict__ rcls,
                                                         int* __restrict__ status) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_rows) return;
    int64_t src[kLatMaxLen];
    int seq[kLatMaxLen];
    bool bad = false;
    const int cnt = lat_trow_entries<I>(j, crow, col, D, disp, nd, src, seq, bad);
    if (ctable) {
        const int cls = remap[slot[j]];
        rcls[j] = (unsigned char)cls;
        if (!bad) {
            const int* trow = ctable + (int64_t)cls * kLatMaxLen;
            for (int k = 0; k < cnt; ++k) bad |= trow[k] != seq[k];
            bad |= lens[cls] != cnt;
        }
    } else {
        int64_t h = 0;
        for (int k = 0; k < cnt; ++k) h += lat_mix(seq[k], k + 17);
        h += lat_mix(cnt, 3);
        const int s = bad ? 0 : lat_slot_of(h, j, thash, trep);
        bad |= s < 0;
        slot[j] = (unsigned short)(s < 0 ? 0 : s);
    }
    if (bad) atomicAdd(status + 0, 1);
    if (cnt > *(volatile int*)(status + 3)) atomicMax(status + 3, cnt);
}

// The classes of the rows of every workgroup of a launch configuration: mask[block] = 256-bit set of class ids
// (block = ((item*nseg + x/seg_len)*tiles_y + y/ty)*tiles_z + z/tz, as in lattice_kernel).
__global__ __launch_bounds__(256) void lat_block_classes_kernel(int64_t n_rows, const unsigned char* __restrict__ rcls, LatDims D, int ty, int tz,
                                                                 int nseg, unsigned long long* __restrict__ mask) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const int64_t d2 = (int64_t)D.ny * D.nz;
    const int64_t X = row / d2;
    const int rem = (int)(row - X * d2);
    const int y = rem / D.nz, z = rem - y * D.nz;
    const int64_t item = X / D.nx;
    const int x = (int)(X - item * D.nx);
    const int seg_len = (D.nx + nseg - 1) / nseg;
    const int tiles_y = (D.ny + ty - 1) / ty, tiles_z = (D.nz + tz - 1) / tz;
    const int64_t blk = ((item * nseg + x / seg_len) * tiles_y + y / ty) * tiles_z + z / tz;
    const int c = rcls[row];
    const unsigned long long bit = 1ull << (c & 63);
    unsigned long long* w = mask + blk * 4 + (c >> 6);
    if (!(*w & bit)) atomicOr(w, bit);
}

template <typename I>
__global__ void lat_trow_codes_kernel(int nrows, const int64_t* __restrict__ rows, const I* __restrict__ crow, const I* __restrict__ col,
                                      LatDims D, const unsigned char* __restrict__ disp, int nd, int* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    int64_t src[kLatMaxLen];
    int seq[kLatMaxLen];
    bool bad = false;
    const int cnt = lat_trow_entries<I>(rows[r], crow, col, D, disp, nd, src, seq, bad);
    for (int k = 0; k < kLatMaxLen; ++k) out[r * kLatMaxLen + k] = (!bad && k < cnt) ? seq[k] : -1;
}

}  // namespace tsgu

using namespace tsgu;

namespace {

int dims_ok(int64_t n_rows, int nb, int nx, int ny, int nz) {
    if (nb <= 0 || nx <= 0 || ny <= 0 || nz <= 0) return TSGU_ERR_BAD_ARG;
    if ((int64_t)nb * nx * ny * nz != n_rows) return TSGU_ERR_BAD_ARG;
    return n_rows > 0x7fffffffLL ? TSGU_ERR_TOO_LARGE : TSGU_OK;
}

}  // namespace

extern "C" {

int tsgu_lattice_slots(void) { return kLatSlots; }

int tsgu_lattice_rows(int itype, int64_t n_rows, const void* crow, const void* col, int nb, int nx, int ny, int nz, const void* disp, int nd,
                      void* slot, void* thash, void* trep, const void* remap, const void* ctable, const void* lens, void* rcls, void* status,
                      int box_mask, int periodic, int device, void* stream) {
    if (const int rc = dims_ok(n_rows, nb, nx, ny, nz)) return rc;
    if (!crow || !col || !status || !slot || (nd > 0 && !disp) || nd > 75) return TSGU_ERR_BAD_ARG;
    if (box_mask < 0 || box_mask >= (1 << 27) || (box_mask && (nd > 0 || ctable == nullptr))) return TSGU_ERR_BAD_ARG;
    const bool pass2 = ctable != nullptr;
    if (pass2 ? (!remap || !lens || !rcls) : (!thash || !trep)) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    const LatDims D{nb, nx, ny, nz};
    const dim3 grid((unsigned)((n_rows + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* sl = static_cast<unsigned short*>(slot);
    unsigned long long* th = static_cast<unsigned long long*>(thash);
    int* tr = static_cast<int*>(trep);
    const unsigned char* rm = static_cast<const unsigned char*>(remap);
    const int* ct = static_cast<const int*>(ctable);
    const unsigned char* ln = static_cast<const unsigned char*>(lens);
    unsigned char* rc_ = static_cast<unsigned char*>(rcls);
    int* st = static_cast<int*>(status);
    const unsigned char* dp = static_cast<const unsigned char*>(disp);
    if (itype != TSGU_I32 && itype != TSGU_I64) return TSGU_ERR_BAD_DTYPE;
    if (nd <= 0) {
        if (itype == TSGU_I32)
            hipLaunchKernelGGL(lat_rows_kernel<int>, grid, block, 0, s, n_rows, static_cast<const int*>(crow), static_cast<const int*>(col), D, sl,
                               th, tr, rm, ct, ln, rc_, st, (unsigned)box_mask, periodic);
        else
            hipLaunchKernelGGL(lat_rows_kernel<int64_t>, grid, block, 0, s, n_rows, static_cast<const int64_t*>(crow),
                               static_cast<const int64_t*>(col), D, sl, th, tr, rm, ct, ln, rc_, st, (unsigned)box_mask, periodic);
    } else {
        if (itype == TSGU_I32)
            hipLaunchKernelGGL(lat_trows_kernel<int>, grid, block, 0, s, n_rows, static_cast<const int*>(crow), static_cast<const int*>(col), D,
                               dp, nd, sl, th, tr, rm, ct, ln, rc_, st);
        else
            hipLaunchKernelGGL(lat_trows_kernel<int64_t>, grid, block, 0, s, n_rows, static_cast<const int64_t*>(crow),
                               static_cast<const int64_t*>(col), D, dp, nd, sl, th, tr, rm, ct, ln, rc_, st);
    }
    return check_launch();
}

int tsgu_lattice_row_codes(int itype, int64_t n_rows, const void* crow, const void* col, int nb, int nx, int ny, int nz, const void* disp,
                           int nd, const void* rows, int nrows, void* out, int device, void* stream) {
    if (const int rc = dims_ok(n_rows, nb, nx, ny, nz)) return rc;
    if (!crow || !col || !rows || !out || nrows <= 0 || (nd > 0 && !disp)) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    const LatDims D{nb, nx, ny, nz};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t* rw = static_cast<const int64_t*>(rows);
    int* o = static_cast<int*>(out);
    const unsigned char* dp = static_cast<const unsigned char*>(disp);
    if (itype != TSGU_I32 && itype != TSGU_I64) return TSGU_ERR_BAD_DTYPE;
    if (nd <= 0) {
        if (itype == TSGU_I32)
            hipLaunchKernelGGL(lat_row_codes_kernel<int>, dim3(nrows), dim3(kLatMaxLen), 0, s, nrows, rw, static_cast<const int*>(crow),
                               static_cast<const int*>(col), D, o);
        else
            hipLaunchKernelGGL(lat_row_codes_kernel<int64_t>, dim3(nrows), dim3(kLatMaxLen), 0, s, nrows, rw, static_cast<const int64_t*>(crow),
                               static_cast<const int64_t*>(col), D, o);
    } else {
        const dim3 grid((unsigned)((nrows + 63) / 64)), block(64);
        if (itype == TSGU_I32)
            hipLaunchKernelGGL(lat_trow_codes_kernel<int>, grid, block, 0, s, nrows, rw, static_cast<const int*>(crow),
                               static_cast<const int*>(col), D, dp, nd, o);
        else
            hipLaunchKernelGGL(lat_trow_codes_kernel<int64_t>, grid, block, 0, s, nrows, rw, static_cast<const int64_t*>(crow),
                               static_cast<const int64_t*>(col), D, dp, nd, o);
    }
    return check_launch();
}

int tsgu_lattice_block_classes(int64_t n_rows, const void* rcls, int nb, int nx, int ny, int nz, int ty, int tz, int nseg, void* mask,
                               int device, void* stream) {
    if (const int rc = dims_ok(n_rows, nb, nx, ny, nz)) return rc;
    if (!rcls || !mask || ty <= 0 || tz <= 0 || nseg <= 0 || nseg > nx) return TSGU_ERR_BAD_ARG;
    if (const int rc = set_device(device)) return rc;
    const LatDims D{nb, nx, ny, nz};
    hipLaunchKernelGGL(lat_block_classes_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n_rows,
                       static_cast<const unsigned char*>(rcls), D, ty, tz, nseg, static_cast<unsigned long long*>(mask));
    return check_launch();
}

}  // extern "C"
