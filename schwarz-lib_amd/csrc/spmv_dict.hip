// Dictionary-coded CSR tiles: a lossless re-encoding of (val, col) built once at upload.
//
// Many matrices on this path (the Poisson stencils, FEM operators on regular meshes) repeat a
// handful of values and a handful of (column - row) offsets inside every run of 256 rows.  For
// each row tile whose entries use at most 256 distinct values and 256 distinct offsets the 12
// bytes per nonzero of CSR become 2: one byte indexing a per-tile value dictionary, one byte
// indexing a per-tile offset dictionary.
// Tiles that do not qualify keep using val/col.  The arithmetic is unchanged -- the same
// products are summed in the same (ascending column) order -- so results are bit-identical to
// the plain CSR kernels; only the HBM traffic differs.
//
// Kernel: the tile's codes (<= 4 KiB) and dictionaries are staged in LDS with coalesced loads;
// then ONE LANE PER ROW decodes its entries and gathers x.  Consecutive lanes hold consecutive
// rows, so for banded matrices each gather instruction reads one contiguous run of x (the k-th
// neighbour of 64 consecutive rows) instead of the ~20 scattered lines of an entry-parallel
// gather.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "device_utils.hpp"
#include "schwz_internal.hpp"

namespace schwz {

constexpr int kDictMax = 256;
constexpr int kChunk = 8;  // gathers issued back to back per lane

template <int MODE>
__global__ __launch_bounds__(kBlock) void spmv_dict_kernel(CsrView A, SpmvArgs a)
{
    // every SpMV variant rounds each product before it is added (no FMA contraction), so
    // that all variants -- and the sequential CPU oracle -- produce the same bits
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) uint16_t codes[kTileNnz + 8];
    __shared__ double vdict[kDictMax];
    __shared__ int ddict[kDictMax];
    __shared__ double red[4];
    if (MODE == kSpmvDot || MODE == kSpmvResidInit) {
        if (a.stop_iter && a.it >= *a.stop_iter) return;
    }
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x % kXcds;
    const int slot = blockIdx.x / kXcds;
    const int per_xcd = gridDim.x / kXcds;
    const int chunk = xcd_slots(A);
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    const bool dual = (MODE == kSpmvResidDual) && a.x2 != nullptr;

    for (int t = slot; t < chunk; t += per_xcd) {
        const int tile = xcd_tile(A, xcd, t);
        if (tile < 0) continue;
        const int r0 = A.tile_row[tile], r1 = A.tile_row[tile + 1];
        const int s = A.rp[r0], e = A.rp[r1];
        const int v0 = A.vdict_ptr[tile], nv = A.vdict_ptr[tile + 1] - v0;
        const int d0 = A.ddict_ptr[tile], nd = A.ddict_ptr[tile + 1] - d0;
        const int row = r0 + tid;
        const bool have_row = (r1 - r0 > 1) ? row < r1 : tid == 0;
        const bool dual_t = dual && (!A.tile_dual || A.tile_dual[tile]);
        double sum = 0.0, sum2 = 0.0;
        if (nd > 0) {
            // ---- coded tile ------------------------------------------------------------------
            const int s4 = s & ~3;  // 8-byte aligned window of 2-byte codes
            int b0 = 0, b1 = 0;
            if (row < r1) {
                b0 = A.rp[row] - s4;
                b1 = A.rp[row + 1] - s4;
            }
            const int last = max((e - 1) & ~3, s4);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = min(s4 + 4 * (tid + kBlock * k), last);
                *reinterpret_cast<uint2 *>(&codes[4 * (tid + kBlock * k)]) =
                    *reinterpret_cast<const uint2 *>(A.code + idx);
            }
            if (tid < nv) vdict[tid] = A.vdict[v0 + tid];
            if (tid < nd) ddict[tid] = A.ddict[d0 + tid];
            lds_barrier();
            if (row < r1) {
                for (int j = b0; j < b1; j += kChunk) {
                    double v[kChunk], xg[kChunk], xg2[kChunk];
                    int c[kChunk];
                    const int jl = b1 - 1;
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) {
                        const unsigned cd = codes[min(j + k, jl)];
                        v[k] = vdict[cd & 255u];
                        c[k] = row + ddict[cd >> 8];
                    }
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) xg[k] = a.x[c[k]];
                    if (dual_t) {
#pragma unroll
                        for (int k = 0; k < kChunk; ++k) xg2[k] = a.x2[c[k]];
                    }
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) {
                        // product rounded on its own (no FMA contraction): the plain CSR
                        // kernels round it when they stage it in LDS, and the two must agree
                        if (j + k < b1) {
                            sum += __dmul_rn(v[k], xg[k]);
                            if (dual_t) sum2 += __dmul_rn(v[k], xg2[k]);
                        }
                    }
                }
            }
            lds_barrier();
        } else if (r1 - r0 > 1) {
            // ---- raw tile: one lane per row straight from val/col --------------------------------
            if (row < r1) {
                for (int j = A.rp[row]; j < A.rp[row + 1]; ++j) {
                    const double vv = A.val[j];
                    const int cc = A.col[j];
                    sum += __dmul_rn(vv, a.x[cc]);
                    if (dual_t) sum2 += __dmul_rn(vv, a.x2[cc]);
                }
            }
        } else {
            // ---- a single long row: the whole workgroup reduces it ------------------------------
            double part = 0.0, part2 = 0.0;
            for (int i = s + tid; i < e; i += kBlock) {
                part += __dmul_rn(A.val[i], a.x[A.col[i]]);
                if (dual_t) part2 += __dmul_rn(A.val[i], a.x2[A.col[i]]);
            }
            sum = block_sum(part, red);
            if (dual_t) sum2 = block_sum(part2, red);
        }
        if (have_row) {
            const int rw = (r1 - r0 > 1) ? row : r0;
            if (MODE == kSpmvPlain) {
                a.y[rw] = (a.beta == 0.0) ? a.alpha * sum : a.alpha * sum + a.beta * a.y[rw];
            } else if (MODE == kSpmvDot) {
                a.y[rw] = sum;
                acc0 += a.x[rw] * sum;
            } else if (MODE == kSpmvResidInit || MODE == kSpmvResidDual) {
                const double bb = a.b[rw];
                const double r = bb - sum;
                const double z = a.dinv ? a.dinv[rw] * r : r;
                a.y[rw] = r;
                a.p[rw] = z;
                acc0 += r * z;
                acc1 += r * r;
                if (MODE == kSpmvResidDual && rw < a.row_limit) {
                    const double r2 = dual_t ? bb - sum2 : r;
                    acc2 += r2 * r2;
                }
            } else {  // kSpmvResidNorm
                if (rw < a.row_limit) {
                    const double r = a.b[rw] - sum;
                    acc1 += r * r;
                }
            }
        }
    }
    if (MODE != kSpmvPlain) {
        const double s0 = block_sum(acc0, red);
        const double s1 = block_sum(acc1, red);
        if (tid == 0) {
            a.partials[blockIdx.x] = s0;
            a.partials[gridDim.x + blockIdx.x] = s1;
        }
        if (MODE == kSpmvResidDual) {
            const double s2v = block_sum(acc2, red);
            if (tid == 0) a.partials[2 * gridDim.x + blockIdx.x] = s2v;
        }
    }
}

int launch_spmv_dict(const CsrView &A, int mode, const SpmvArgs &a, int grid, hipStream_t s)
{
    switch (mode) {
    case kSpmvPlain:
        hipLaunchKernelGGL(spmv_dict_kernel<kSpmvPlain>, dim3(grid), dim3(kBlock), 0, s, A, a);
        break;
    case kSpmvDot:
        hipLaunchKernelGGL(spmv_dict_kernel<kSpmvDot>, dim3(grid), dim3(kBlock), 0, s, A, a);
        break;
    case kSpmvResidInit:
        hipLaunchKernelGGL(spmv_dict_kernel<kSpmvResidInit>, dim3(grid), dim3(kBlock), 0, s, A, a);
        break;
    case kSpmvResidDual:
        hipLaunchKernelGGL(spmv_dict_kernel<kSpmvResidDual>, dim3(grid), dim3(kBlock), 0, s, A, a);
        break;
    default:
        hipLaunchKernelGGL(spmv_dict_kernel<kSpmvResidNorm>, dim3(grid), dim3(kBlock), 0, s, A, a);
        break;
    }
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

// ---------------------------------------------------------------------------------------------
// host: build the coding
// ---------------------------------------------------------------------------------------------

namespace {

template <typename T>
int up(const std::vector<T> &h, void **d, size_t pad = 0)
{
    *d = nullptr;
    SCHWZ_HIP_TRY(hipMalloc(d, (h.size() + pad ? h.size() + pad : 1) * sizeof(T)));
    if (!h.empty()) SCHWZ_HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    if (pad) SCHWZ_HIP_TRY(hipMemset((char *)*d + h.size() * sizeof(T), 0, pad * sizeof(T)));
    return SCHWZ_OK;
}

// small open-addressing map from a 64-bit key to a code < 256, reset per tile
struct TinyMap {
    uint64_t key[512];
    int16_t code[512];
    int used[256];
    int n = 0;
    TinyMap() { std::memset(code, -1, sizeof(code)); }
    void reset()
    {
        for (int i = 0; i < n; ++i) code[used[i]] = -1;
        n = 0;
    }
    // returns the code, or -1 when a 257th distinct key arrives
    int get(uint64_t k)
    {
        uint64_t h = k * 0x9E3779B97F4A7C15ull;
        int slot = (int)(h >> 55);  // 9 bits
        while (code[slot] >= 0) {
            if (key[slot] == k) return code[slot];
            slot = (slot + 1) & 511;
        }
        if (n == kDictMax) return -1;
        key[slot] = k;
        code[slot] = (int16_t)n;
        used[n] = slot;
        return n++;
    }
};

}  // namespace

// =============================================================================================
// Row-pattern coding: the second level of the same idea.  In a stencil-like matrix almost every
// row of a tile is one of a few (values, col - row offsets) sequences; coding the SEQUENCE costs
// one byte per ROW.  A tile's table of <= 64 patterns (<= 1024 entries) is staged in LDS, and
// because most tiles share one table (tables are de-duplicated at build time) a workgroup
// reloads it only when the table id changes -- the row loop then runs without any barrier:
// per row 1 byte of matrix, the coalesced x gathers, 8 bytes of y.  Summation order is the
// row's entry order, products rounded individually: bit-identical to the CSR kernels again.
// =============================================================================================

constexpr int kPatMax = 64;       // patterns per table
constexpr int kPatEntries = 1024;  // entries per table (npat * lmax)

// One staged table entry: the value and the BYTE offset of the column relative to the row
// (col - row) * 8 (WIDE: col - row), read together by one ds_read_b128.  Pattern p starts at entry p * ls with
// ls = lmax rounded up to kChunk, so entry k of a chunk sits at an immediate offset; the padding
// entries are (0.0, 0): a lane past its row's length gathers x[row] and discards it.
struct __attribute__((aligned(16))) PatEntry {
    double val;
    int off8;
    int pad;
};

__host__ __device__ inline int pat_stride(int lmax) { return (lmax + kChunk - 1) / kChunk * kChunk; }

// NT: tiles a workgroup works on at a time (NT rows per lane; 2 was measured and is slower: the
// registers cost occupancy).  WIDE: x needs 64-bit offsets (>= 2^28 columns); otherwise the
// gathers use the scalar-base + 32-bit-lane-offset form.
template <int MODE, int NT, bool WIDE>
__global__ __launch_bounds__(kBlock) void spmv_pattern_kernel(CsrView A, SpmvArgs a)
{
#pragma clang fp contract(off)
    __shared__ PatEntry ptab[kPatEntries];
    __shared__ int plen[kPatMax];
    __shared__ double red[4];
    if (MODE == kSpmvDot || MODE == kSpmvResidInit) {
        if (a.stop_iter && a.it >= *a.stop_iter) return;
    }
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x % kXcds;
    const int slot = blockIdx.x / kXcds;
    const int per_xcd = gridDim.x / kXcds;
    const int chunk = xcd_slots(A);
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    const bool dual = (MODE == kSpmvResidDual) && a.x2 != nullptr;
    int cached = -1, ls = 0;  // staged table and its row stride

    auto stage_table = [&](int tb) {  // workgroup-uniform
        if (tb == cached) return;
        const int eoff = A.tbl_desc[4 * tb], loff = A.tbl_desc[4 * tb + 1];
        const int npat = A.tbl_desc[4 * tb + 2], lmax = A.tbl_desc[4 * tb + 3];
        ls = pat_stride(lmax);
        lds_barrier();  // everyone is done with the previous table
        for (int i = tid; i < npat * ls; i += kBlock) {
            const int pt = i / ls, k = i - pt * ls;
            PatEntry e;
            e.val = k < lmax ? A.tbl_val[eoff + pt * lmax + k] : 0.0;
            e.off8 = k < lmax ? A.tbl_delta[eoff + pt * lmax + k] * (WIDE ? 1 : 8) : 0;
            e.pad = 0;
            ptab[i] = e;
        }
        if (tid < npat) plen[tid] = A.tbl_len[loff + tid];
        lds_barrier();
        cached = tb;
    };
    auto xload = [&](const double *xv, int row, int off8) -> double {
        if (WIDE) return xv[(int64_t)row + off8];  // the entry holds col - row itself
        const uint32_t o = (uint32_t)row * 8u + (uint32_t)off8;
        return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(xv) + o);
    };
    // the row's own operands of the fused epilogue (requested ahead of the x gathers)
    auto own_load = [&](bool have, int rw, double &ox, double &ob, double &od, bool with_x) {
        ox = 0.0, ob = 0.0, od = 1.0;
        if (!have) return;
        if (MODE == kSpmvDot && with_x) ox = a.x[rw];
        if (MODE == kSpmvResidInit || MODE == kSpmvResidDual) ob = a.b[rw];
        if (MODE == kSpmvResidNorm && rw < a.row_limit) ob = a.b[rw];
        if ((MODE == kSpmvResidInit || MODE == kSpmvResidDual) && a.dinv) od = a.dinv[rw];
        if (MODE == kSpmvPlain && a.beta != 0.0) ob = a.y[rw];
    };
    auto epilogue = [&](int rw, double sum, double sum2, bool dual_t, double ox, double ob, double od) {
        if (MODE == kSpmvPlain) {
            a.y[rw] = (a.beta == 0.0) ? a.alpha * sum : a.alpha * sum + a.beta * ob;
        } else if (MODE == kSpmvDot) {
            a.y[rw] = sum;
            acc0 += ox * sum;
        } else if (MODE == kSpmvResidInit || MODE == kSpmvResidDual) {
            const double r = ob - sum;
            const double z = a.dinv ? od * r : r;
            a.y[rw] = r;
            a.p[rw] = z;
            acc0 += r * z;
            acc1 += r * r;
            if (MODE == kSpmvResidDual && rw < a.row_limit) {
                const double r2 = dual_t ? ob - sum2 : r;
                acc2 += r2 * r2;
            }
        } else {  // kSpmvResidNorm
            if (rw < a.row_limit) {
                const double r = ob - sum;
                acc1 += r * r;
            }
        }
    };
    // entries [j0, len) of the lane's pattern against vector xv, in entry order
    auto row_tail = [&](const double *xv, int row, int base, int len, int j0, double &sum) {
        for (int j = j0; j < len; j += kChunk) {
            double v[kChunk], xg[kChunk];
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const PatEntry e = ptab[base + j + k];
                v[k] = e.val;
                xg[k] = xload(xv, row, e.off8);
            }
#pragma unroll
            for (int k = 0; k < kChunk; ++k)
                if (j + k < len) sum += v[k] * xg[k];
        }
    };
    // one tile on its own: any kind (pattern table, plain rows, one long row)
    auto one_tile = [&](int tile, int r0, int r1, int tb) {
        const int row = r0 + tid;
        const bool single = r1 - r0 == 1;
        const bool have_row = single ? tid == 0 : row < r1;
        const int rw = single ? r0 : row;
        const bool dual_t = dual && (!A.tile_dual || A.tile_dual[tile]);
        double ox, ob, od;
        own_load(have_row, rw, ox, ob, od, true);
        double sum = 0.0, sum2 = 0.0;
        if (tb >= 0) {
            stage_table(tb);
            if (row < r1) {
                const int pid = A.pat_id[row];
                const int len = plen[pid];
                row_tail(a.x, row, pid * ls, len, 0, sum);
                if (dual_t) row_tail(a.x2, row, pid * ls, len, 0, sum2);
            }
        } else if (!single) {
            // tile without a pattern table: one lane per row straight from val/col
            if (row < r1) {
                for (int j = A.rp[row]; j < A.rp[row + 1]; ++j) {
                    const double vv = A.val[j];
                    const int cc = A.col[j];
                    sum += vv * a.x[cc];
                    if (dual_t) sum2 += vv * a.x2[cc];
                }
            }
        } else {
            // a single long row: the whole workgroup reduces it
            const int s = A.rp[r0], e = A.rp[r1];
            double part = 0.0, part2 = 0.0;
            for (int i = s + tid; i < e; i += kBlock) {
                part += A.val[i] * a.x[A.col[i]];
                if (dual_t) part2 += A.val[i] * a.x2[A.col[i]];
            }
            sum = block_sum(part, red);
            if (dual_t) sum2 = block_sum(part2, red);
        }
        if (have_row) epilogue(rw, sum, sum2, dual_t, ox, ob, od);
    };

    // Per tile the fetches depend on each other (tile bounds -> pattern ids -> x -> y); with
    // NT > 1 each step is issued for NT tiles of the workgroup's sequence before it is waited for.
    for (int t = slot; t < chunk; t += NT * per_xcd) {
        int tile[NT], r0[NT], r1[NT], tb[NT];
        bool together = true;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int tt = t + u * per_xcd;
            tile[u] = tt < chunk ? xcd_tile(A, xcd, tt) : -1;
            r0[u] = r1[u] = 0;
            tb[u] = -1;
            if (tile[u] >= 0) {
                r0[u] = A.tile_row[tile[u]];
                r1[u] = A.tile_row[tile[u] + 1];
                tb[u] = A.tile_table[tile[u]];
            }
            together = together && tile[u] >= 0 && tb[u] >= 0 && tb[u] == tb[0];
        }
        if (!together) {
#pragma unroll
            for (int u = 0; u < NT; ++u)
                if (tile[u] >= 0) one_tile(tile[u], r0[u], r1[u], tb[u]);
            continue;
        }
        stage_table(tb[0]);
        int row[NT], pid[NT];
        bool act[NT], dual_t[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            row[u] = r0[u] + tid;
            act[u] = row[u] < r1[u];
            pid[u] = act[u] ? (int)A.pat_id[row[u]] : 0;
            dual_t[u] = dual && (!A.tile_dual || A.tile_dual[tile[u]]);
        }
        double ox[NT], ob[NT], od[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) own_load(act[u], row[u], ox[u], ob[u], od[u], false);
        int len[NT], base[NT];
        double v[NT][kChunk], xg[NT][kChunk];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            len[u] = act[u] ? plen[pid[u]] : 0;
            base[u] = pid[u] * ls;
            // One gather instruction per entry; the fused dot takes x[row] from the gathers instead
            // of loading it again.  Measured in-box A/B (tools/ab.sh, 256^3, CG iteration time):
            // this kernel -1..-3 % against the unpacked-table version; two tiles per workgroup at a
            // time, an LDS ring of the in-plane x window, and lane shuffles for the near-diagonal
            // entries all land within +-3 %, i.e. inside the drift of one box.
            const int rsafe = act[u] ? row[u] : r0[u];  // idle lanes read a valid address
            bool diag_seen = false;
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const PatEntry e = ptab[base[u] + k];
                v[u][k] = e.val;
                xg[u][k] = xload(a.x, rsafe, act[u] ? e.off8 : 0);
                if (MODE == kSpmvDot && e.off8 == 0) {
                    ox[u] = xg[u][k];
                    diag_seen = true;
                }
            }
            if (MODE == kSpmvDot && act[u] && !diag_seen) ox[u] = a.x[row[u]];
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            double sum = 0.0, sum2 = 0.0;
#pragma unroll
            for (int k = 0; k < kChunk; ++k)
                if (k < len[u]) sum += v[u][k] * xg[u][k];
            if (len[u] > kChunk) row_tail(a.x, row[u], base[u], len[u], kChunk, sum);
            if (dual_t[u] && act[u]) row_tail(a.x2, row[u], base[u], len[u], 0, sum2);
            if (act[u]) epilogue(row[u], sum, sum2, dual_t[u], ox[u], ob[u], od[u]);
        }
    }
    if (MODE != kSpmvPlain) {
        const double s0 = block_sum(acc0, red);
        const double s1 = block_sum(acc1, red);
        if (tid == 0) {
            a.partials[blockIdx.x] = s0;
            a.partials[gridDim.x + blockIdx.x] = s1;
        }
        if (MODE == kSpmvResidDual) {
            const double s2v = block_sum(acc2, red);
            if (tid == 0) a.partials[2 * gridDim.x + blockIdx.x] = s2v;
        }
    }
}



int launch_spmv_pattern(const CsrView &A, int mode, const SpmvArgs &a, int grid, hipStream_t s)
{
    const bool wide = A.ncols >= (int64_t(1) << 28);
#define SCHWZ_PAT_LAUNCH(M)                                                                          \
    if (wide)                                                                                        \
        hipLaunchKernelGGL((spmv_pattern_kernel<M, 1, true>), dim3(grid), dim3(kBlock), 0, s, A, a); \
    else                                                                                             \
        hipLaunchKernelGGL((spmv_pattern_kernel<M, 1, false>), dim3(grid), dim3(kBlock), 0, s, A, a);
    switch (mode) {
    case kSpmvPlain: SCHWZ_PAT_LAUNCH(kSpmvPlain) break;
    case kSpmvDot: SCHWZ_PAT_LAUNCH(kSpmvDot) break;
    case kSpmvResidInit: SCHWZ_PAT_LAUNCH(kSpmvResidInit) break;
    case kSpmvResidDual: SCHWZ_PAT_LAUNCH(kSpmvResidDual) break;
    default: SCHWZ_PAT_LAUNCH(kSpmvResidNorm) break;
    }
#undef SCHWZ_PAT_LAUNCH
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

namespace {

struct RowPat {
    std::vector<uint64_t> bits;
    std::vector<schwz_idx> delta;
    bool operator==(const RowPat &o) const { return bits == o.bits && delta == o.delta; }
};

struct Table {
    int npat = 0, lmax = 0;
    std::vector<uint8_t> len;
    std::vector<double> val;       // [npat][lmax]
    std::vector<schwz_idx> delta;  // [npat][lmax]
    uint64_t hash = 0;
    bool same(const Table &o) const
    {
        return npat == o.npat && lmax == o.lmax && len == o.len && delta == o.delta &&
               std::memcmp(val.data(), o.val.data(), val.size() * sizeof(double)) == 0;
    }
};

}  // namespace

// Returns SCHWZ_OK; leaves A->v.pat_id null when the coding does not cover >= 90 % of the
// nonzeros (SCHWZ_SPMV_PATTERN=0 disables, =2 forces whatever the coverage).
static int build_spmv_pattern(schwz_csr *A, const schwz_idx *rp, const schwz_idx *col, const double *val,
                              const std::vector<schwz_idx> &tiles)
{
    const char *env = std::getenv("SCHWZ_SPMV_PATTERN");
    if (env && env[0] == '0') return SCHWZ_OK;
    const int ntiles = (int)tiles.size() - 1;
    const int64_t nrows = tiles.back(), nnz = rp[nrows];
    if (ntiles == 0 || nnz == 0) return SCHWZ_OK;
    std::vector<uint8_t> pat_id((size_t)nrows, 0);
    std::vector<schwz_idx> tile_table((size_t)ntiles, -1);
    std::vector<Table> tables;
    std::unordered_multimap<uint64_t, int> by_hash;
    int64_t coded = 0;
    // every tile's table by all threads (a tile's rows, ids and table depend on that tile alone), then the tables
    // are de-duplicated in tile order -- the ids a sequential pass would give
    std::vector<Table> tile_tb((size_t)ntiles);
    std::vector<char> tile_ok((size_t)ntiles, 0);
    parallel_blocks(ntiles, 256, [&](int, int, int64_t t_begin, int64_t t_end) {
        std::vector<RowPat> pats;
        for (int t = (int)t_begin; t < (int)t_end; ++t) {
            const schwz_idx r0 = tiles[(size_t)t], r1 = tiles[(size_t)t + 1];
            if (rp[r1] == rp[r0] || (r1 - r0 == 1 && rp[r1] - rp[r0] > kTileNnz - 2)) continue;
            pats.clear();
            int lmax = 0;
            bool ok = true;
            RowPat p;
            for (schwz_idx r = r0; r < r1 && ok; ++r) {
                const int len = rp[r + 1] - rp[r];
                if (len > 255) {
                    ok = false;
                    break;
                }
                p.bits.resize((size_t)len);
                p.delta.resize((size_t)len);
                for (int k = 0; k < len; ++k) {
                    std::memcpy(&p.bits[(size_t)k], &val[rp[r] + k], 8);
                    p.delta[(size_t)k] = col[rp[r] + k] - r;
                }
                int id = -1;
                for (size_t q = 0; q < pats.size(); ++q)
                    if (pats[q] == p) {
                        id = (int)q;
                        break;
                    }
                if (id < 0) {
                    if ((int)pats.size() == kPatMax) {
                        ok = false;
                        break;
                    }
                    id = (int)pats.size();
                    lmax = std::max(lmax, len);
                    pats.push_back(p);
                }
                pat_id[(size_t)r] = (uint8_t)id;
            }
            if (!ok || (int64_t)pats.size() * pat_stride(std::max(lmax, 1)) > kPatEntries) continue;
            Table &tb = tile_tb[(size_t)t];
            tb.npat = (int)pats.size();
            tb.lmax = std::max(lmax, 1);
            tb.len.resize((size_t)tb.npat);
            tb.val.assign((size_t)tb.npat * tb.lmax, 0.0);
            tb.delta.assign((size_t)tb.npat * tb.lmax, 0);
            uint64_t h = 1469598103934665603ull;
            for (int q = 0; q < tb.npat; ++q) {
                tb.len[(size_t)q] = (uint8_t)pats[(size_t)q].bits.size();
                for (size_t k = 0; k < pats[(size_t)q].bits.size(); ++k) {
                    std::memcpy(&tb.val[(size_t)q * tb.lmax + k], &pats[(size_t)q].bits[k], 8);
                    tb.delta[(size_t)q * tb.lmax + k] = pats[(size_t)q].delta[k];
                    h = (h ^ pats[(size_t)q].bits[k]) * 1099511628211ull;
                    h = (h ^ (uint64_t)(int64_t)pats[(size_t)q].delta[k]) * 1099511628211ull;
                }
                h = (h ^ 0xffull ^ (uint64_t)tb.len[(size_t)q]) * 1099511628211ull;
            }
            tb.hash = h;
            tile_ok[(size_t)t] = 1;
        }
    });
    for (int t = 0; t < ntiles; ++t) {
        if (!tile_ok[(size_t)t]) continue;
        Table &tb = tile_tb[(size_t)t];
        int id = -1;
        auto range = by_hash.equal_range(tb.hash);
        for (auto it = range.first; it != range.second; ++it)
            if (tables[(size_t)it->second].same(tb)) {
                id = it->second;
                break;
            }
        if (id < 0) {
            id = (int)tables.size();
            by_hash.emplace(tb.hash, id);
            tables.push_back(std::move(tb));
        }
        tile_table[(size_t)t] = id;
        coded += rp[tiles[(size_t)t + 1]] - rp[tiles[(size_t)t]];
    }
    tile_tb.clear();
    tile_tb.shrink_to_fit();
    A->pattern_fraction = (double)coded / (double)nnz;
    if (A->pattern_fraction < 0.9 && !(env && env[0] == '2')) return SCHWZ_OK;
    // a table pays off only when it is shared: with one table per tile the "coding" is just the
    // raw data in another layout
    if (tables.size() * 4 > (size_t)ntiles && !(env && env[0] == '2')) return SCHWZ_OK;
    std::vector<schwz_idx> desc;
    std::vector<uint8_t> lens;
    std::vector<double> vals;
    std::vector<schwz_idx> deltas;
    for (const Table &tb : tables) {
        desc.push_back((schwz_idx)vals.size());
        desc.push_back((schwz_idx)lens.size());
        desc.push_back(tb.npat);
        desc.push_back(tb.lmax);
        lens.insert(lens.end(), tb.len.begin(), tb.len.end());
        vals.insert(vals.end(), tb.val.begin(), tb.val.end());
        deltas.insert(deltas.end(), tb.delta.begin(), tb.delta.end());
    }
    int rc;
    if ((rc = up(pat_id, &A->d_pat_id)) || (rc = up(tile_table, &A->d_tile_table)) || (rc = up(desc, &A->d_tbl_desc)) ||
        (rc = up(lens, &A->d_tbl_len)) || (rc = up(vals, &A->d_tbl_val)) || (rc = up(deltas, &A->d_tbl_delta)))
        return rc;
    A->v.pat_id = (const uint8_t *)A->d_pat_id;
    A->v.tile_table = (const schwz_idx *)A->d_tile_table;
    A->v.tbl_desc = (const schwz_idx *)A->d_tbl_desc;
    A->v.tbl_len = (const uint8_t *)A->d_tbl_len;
    A->v.tbl_val = (const double *)A->d_tbl_val;
    A->v.tbl_delta = (const schwz_idx *)A->d_tbl_delta;
    return SCHWZ_OK;
}

int build_spmv_dict(schwz_csr *A, const schwz_idx *rp, const schwz_idx *col, const double *val,
                    const std::vector<schwz_idx> &tiles)
{
    {
        int rc;
        {
            StageTimer t("codings: row patterns");
            rc = build_spmv_pattern(A, rp, col, val, tiles);
        }
        if (!rc && A->v.pat_id) {
            StageTimer t("codings: row pairs + walk tables");
            rc = build_spmv_pair(A, rp, col, val, tiles);  // stencil-like: pairs too
        }
        if (rc) {
            free_spmv_dict(A);
            return rc;
        }
    }
    const char *env = std::getenv("SCHWZ_SPMV_DICT");
    if (env && env[0] == '0') return SCHWZ_OK;
    // the row-pattern coding supersedes the per-entry one unless that is forced too
    if (A->v.pat_id && !(env && env[0] == '2')) return SCHWZ_OK;
    const int ntiles = (int)tiles.size() - 1;
    const int64_t nnz = rp[tiles.back()];
    if (ntiles == 0 || nnz == 0) return SCHWZ_OK;
    std::vector<uint16_t> code((size_t)nnz, 0);
    std::vector<schwz_idx> vptr((size_t)ntiles + 1, 0), dptr((size_t)ntiles + 1, 0);
    std::vector<double> vdata;
    std::vector<schwz_idx> ddata;
    std::vector<double> tv;
    std::vector<schwz_idx> td;
    TinyMap vm, dm;
    int64_t coded = 0;
    for (int t = 0; t < ntiles; ++t) {
        const schwz_idx r0 = tiles[(size_t)t], r1 = tiles[(size_t)t + 1];
        const int64_t s = rp[r0], e = rp[r1];
        // the kernel stages the codes of the 8-byte aligned window [s & ~3, e) : 2048 at most
        bool ok = e > s && (s & 3) + (e - s) <= kTileNnz;
        vm.reset();
        dm.reset();
        tv.clear();
        td.clear();
        if (ok) {
            for (schwz_idx r = r0; r < r1 && ok; ++r) {
                for (int64_t j = rp[r]; j < rp[r + 1]; ++j) {
                    uint64_t bits;
                    std::memcpy(&bits, &val[j], 8);
                    const int before_v = vm.n, before_d = dm.n;
                    const int cv = vm.get(bits);
                    const int cdv = dm.get((uint64_t)(int64_t)(col[j] - r));
                    if (cv < 0 || cdv < 0) {
                        ok = false;
                        break;
                    }
                    if (vm.n > before_v) tv.push_back(val[j]);
                    if (dm.n > before_d) td.push_back(col[j] - r);
                    code[(size_t)j] = (uint16_t)(cv | (cdv << 8));
                }
            }
        }
        if (!ok) {
            vptr[(size_t)t + 1] = vptr[(size_t)t];
            dptr[(size_t)t + 1] = dptr[(size_t)t];
            continue;
        }
        coded += e - s;
        vdata.insert(vdata.end(), tv.begin(), tv.end());
        ddata.insert(ddata.end(), td.begin(), td.end());
        vptr[(size_t)t + 1] = (schwz_idx)vdata.size();
        dptr[(size_t)t + 1] = (schwz_idx)ddata.size();
    }
    A->dict_fraction = (double)coded / (double)nnz;
    // worth it only when (nearly) the whole matrix is coded; SCHWZ_SPMV_DICT=2 forces it (tests)
    if (A->dict_fraction < 0.9 && !(env && env[0] == '2')) return SCHWZ_OK;
    int rc;
    if ((rc = up(code, &A->d_code, 8)) || (rc = up(vptr, &A->d_vptr)) || (rc = up(dptr, &A->d_dptr)) ||
        (rc = up(vdata, &A->d_vdict)) || (rc = up(ddata, &A->d_ddict))) {
        free_spmv_dict(A);
        return rc;
    }
    A->v.code = (const uint16_t *)A->d_code;
    A->v.vdict_ptr = (const schwz_idx *)A->d_vptr;
    A->v.ddict_ptr = (const schwz_idx *)A->d_dptr;
    A->v.vdict = (const double *)A->d_vdict;
    A->v.ddict = (const schwz_idx *)A->d_ddict;
    return SCHWZ_OK;
}

void free_spmv_dict(schwz_csr *A)
{
    void *ptrs[] = {A->d_code, A->d_vptr, A->d_dptr, A->d_vdict, A->d_ddict, A->d_pat_id, A->d_tile_table,
                    A->d_tbl_desc, A->d_tbl_len, A->d_tbl_val, A->d_tbl_delta};
    for (void *p : ptrs) (void)hipFree(p);
    free_spmv_pair(A);
    A->d_code = A->d_vptr = A->d_dptr = A->d_vdict = A->d_ddict = nullptr;
    A->d_pat_id = A->d_tile_table = A->d_tbl_desc = A->d_tbl_len = A->d_tbl_val = A->d_tbl_delta = nullptr;
    A->v.code = nullptr;
    A->v.pat_id = nullptr;
}

}  // namespace schwz
