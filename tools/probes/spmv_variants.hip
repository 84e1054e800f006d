// SpMV experiments that were measured and NOT kept: one row per lane, the first tiled
// kernel, software-pipelined tiles, wave-private tiles and the ablation builds of the tiled kernel -- variant ids
// 1-5 and 10-73 of schwz_csr_spmv -- plus the dispatch of the ablation builds of spmv_stream.hip (80 + bits).
// Linked into the measurement build libschwz_hip_probes.so only (`make -C schwarz-lib_amd probes`), which
// tools/spmv_probe.py, tools/stream_ablate.py and tools/r03_stream_probe.py load through SCHWZ_HIP_LIB; the
// product library holds none of this.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "schwz_internal.hpp"
#include "device_utils.hpp"

namespace schwz {

constexpr int kPairsPerLane = kTileNnz / (2 * kBlock);  // 16-byte (value, column) pairs per lane and tile: 4

// ---------------------------------------------------------------------------
// CSR SpMV, tiled: each workgroup owns a run of consecutive rows whose nonzeros
// (<= kTileNnz) are read with unit stride, multiplied with the gathered x and
// staged in LDS; one lane per row then sums its LDS segment.  Tiles are dealt to
// workgroups so that each XCD (blockIdx % 8) sweeps one contiguous eighth of the
// matrix: the x entries a tile shares with its neighbours (i+-1, i+-nx, i+-nx*ny
// for the Poisson stencils) stay in that XCD's 4 MiB L2.
// ---------------------------------------------------------------------------

template <int MODE>
__global__ __launch_bounds__(kBlock) void spmv_tiled_kernel(CsrView A, SpmvArgs a)
{
    // every SpMV variant rounds each product before it is added (no FMA contraction), so
    // that all variants -- and the sequential CPU oracle -- produce the same bits
#pragma clang fp contract(off)
    __shared__ double prod[kTileNnz];
    __shared__ double red[4];
    if (MODE == kSpmvDot || MODE == kSpmvResidInit) {
        if (a.stop_iter && a.it >= *a.stop_iter) return;
    }
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x % kXcds;
    const int slot = blockIdx.x / kXcds;
    const int per_xcd = gridDim.x / kXcds;
    const int chunk = (A.ntiles + kXcds - 1) / kXcds;
    double acc0 = 0.0, acc1 = 0.0;

    for (int t = slot; t < chunk; t += per_xcd) {
        const int tile = xcd * chunk + t;
        if (tile >= A.ntiles) break;
        const int r0 = A.tile_row[tile], r1 = A.tile_row[tile + 1];
        const int s = A.rp[r0], e = A.rp[r1];
        const int cnt = e - s;
        double sum = 0.0;
        int row = r0 + tid;
        bool have_row = false;
        if (cnt <= kTileNnz) {
            for (int i = tid; i < cnt; i += kBlock)
                prod[i] = A.val[s + i] * a.x[A.col[s + i]];
            __syncthreads();
            if (row < r1) {
                have_row = true;
                const int b0 = A.rp[row] - s, b1 = A.rp[row + 1] - s;
                for (int j = b0; j < b1; ++j) sum += prod[j];
            }
            __syncthreads();
        } else {
            // a single row longer than a tile: the whole workgroup reduces it
            double part = 0.0;
            for (int i = tid; i < cnt; i += kBlock) part += __dmul_rn(A.val[s + i], a.x[A.col[s + i]]);
            part = block_sum(part, red);
            row = r0;
            if (tid == 0) {
                have_row = true;
                sum = part;
            }
        }
        if (have_row) {
            if (MODE == kSpmvPlain) {
                a.y[row] = (a.beta == 0.0) ? a.alpha * sum : a.alpha * sum + a.beta * a.y[row];
            } else if (MODE == kSpmvDot) {
                a.y[row] = sum;
                acc0 += a.x[row] * sum;
            } else if (MODE == kSpmvResidInit) {
                const double r = a.b[row] - sum;
                const double z = a.dinv ? a.dinv[row] * r : r;
                a.y[row] = r;
                a.p[row] = z;
                acc0 += r * z;
                acc1 += r * r;
            } else {  // kSpmvResidNorm
                if (row < a.row_limit) {
                    const double r = a.b[row] - sum;
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
    }
}


// ---------------------------------------------------------------------------
// Software-pipelined tiled SpMV (variant 4; measured no faster than variant 0).  Same tiles and arithmetic
// as spmv_tiled2_kernel, but a workgroup's tile loop is a two-stage pipeline:
// as soon as the products of tile t are in LDS, the 16-byte loads of tile t+1 are
// issued into the SAME registers, so they fly during the barrier / row-sum / store
// phase of tile t; the tile descriptors (row range, nonzero range) are fetched one
// tile ahead as well.  Each workgroup thus always has ~24 KiB of matrix stream in
// flight instead of stalling through a load -> gather -> LDS -> barrier chain.
// ---------------------------------------------------------------------------

struct TileDesc {
    int r0, r1, s, e;
    bool valid, regular;
};

__device__ __forceinline__ TileDesc tile_desc(const CsrView &A, int xcd, int chunk, int t)
{
    TileDesc d;
    d.r0 = d.r1 = d.s = d.e = 0;
    d.regular = false;
    const int tile = t < chunk ? xcd_tile(A, xcd, t) : -1;
    d.valid = tile >= 0;
    if (d.valid) {
        const int tl = A.tile_order ? A.tile_order[tile] : tile;
        d.r0 = A.tile_row[tl];
        d.r1 = A.tile_row[tl + 1];
        d.s = A.rp[d.r0];
        d.e = A.rp[d.r1];
        d.regular = (d.e - d.s) <= kTileNnz - 2;
    }
    return d;
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void spmv_pipe_kernel(CsrView A, SpmvArgs a)
{
    // every SpMV variant rounds each product before it is added (no FMA contraction), so
    // that all variants -- and the sequential CPU oracle -- produce the same bits
#pragma clang fp contract(off)
    __shared__ double prod[kTileNnz + 2];
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

    double2 v[kPairsPerLane];
    int2 c[kPairsPerLane];
    // No per-pair bounds branch: a lane whose pair lies past the tile re-reads the
    // tile's last pair instead (its product lands in an LDS slot no row sums), so
    // all loads of a phase issue back to back and one wait covers them.
#define SCHWZ_ISSUE_LOADS(D)                                                     \
    {                                                                            \
        const int s2_ = (D).s & ~1;                                              \
        const int last_ = max(((D).e - 1) & ~1, s2_);                            \
        _Pragma("unroll") for (int k = 0; k < kPairsPerLane; ++k)                \
        {                                                                        \
            const int idx = min(s2_ + 2 * (tid + kBlock * k), last_);            \
            v[k] = *reinterpret_cast<const double2 *>(A.val + idx);              \
            c[k] = *reinterpret_cast<const int2 *>(A.col + idx);                 \
        }                                                                        \
    }
#define SCHWZ_PRODUCTS(XV, D)                                                    \
    {                                                                            \
        double xg[2 * kPairsPerLane];                                            \
        _Pragma("unroll") for (int k = 0; k < kPairsPerLane; ++k)                \
        {                                                                        \
            xg[2 * k] = (XV)[c[k].x];                                            \
            xg[2 * k + 1] = (XV)[c[k].y];                                        \
        }                                                                        \
        _Pragma("unroll") for (int k = 0; k < kPairsPerLane; ++k)                \
        {                                                                        \
            double2 pr;                                                          \
            pr.x = v[k].x * xg[2 * k];                                           \
            pr.y = v[k].y * xg[2 * k + 1];                                       \
            *reinterpret_cast<double2 *>(&prod[2 * (tid + kBlock * k)]) = pr;    \
        }                                                                        \
    }

    // past-the-end slots only occur in the last run of the block-cyclic deal
    int t = slot;
    while (t < chunk && xcd_tile(A, xcd, t) < 0) t += per_xcd;
    TileDesc cur = tile_desc(A, xcd, chunk, t);
    if (cur.valid && cur.regular) SCHWZ_ISSUE_LOADS(cur)
    while (cur.valid) {
        t += per_xcd;
        while (t < chunk && xcd_tile(A, xcd, t) < 0) t += per_xcd;
        const TileDesc nxt = tile_desc(A, xcd, chunk, t);
        const int r0 = cur.r0, r1 = cur.r1, s = cur.s, e = cur.e;
        const int cnt = e - s;
        double sum = 0.0, sum2 = 0.0;
        int row = r0 + tid;
        bool have_row = false;
        if (cur.regular) {
            const int s2 = s & ~1;
            int b0 = 0, b1 = 0;
            if (row < r1) {
                b0 = A.rp[row] - s2;
                b1 = A.rp[row + 1] - s2;
            }
            SCHWZ_PRODUCTS(a.x, cur)
            if (!dual && nxt.valid && nxt.regular) SCHWZ_ISSUE_LOADS(nxt)
            lds_barrier();
            if (row < r1) {
                have_row = true;
                for (int j = b0; j < b1; ++j) sum += prod[j];
            }
            lds_barrier();
            if (dual) {
                SCHWZ_PRODUCTS(a.x2, cur)
                if (nxt.valid && nxt.regular) SCHWZ_ISSUE_LOADS(nxt)
                lds_barrier();
                if (row < r1)
                    for (int j = b0; j < b1; ++j) sum2 += prod[j];
                lds_barrier();
            }
        } else {
            if (r1 - r0 > 1) {
                // fits kTileNnz but not the aligned window: plain staging
                for (int i = tid; i < cnt; i += kBlock) prod[i] = A.val[s + i] * a.x[A.col[s + i]];
                lds_barrier();
                const int b0 = row < r1 ? A.rp[row] - s : 0, b1 = row < r1 ? A.rp[row + 1] - s : 0;
                if (row < r1) {
                    have_row = true;
                    for (int j = b0; j < b1; ++j) sum += prod[j];
                }
                lds_barrier();
                if (dual) {
                    for (int i = tid; i < cnt; i += kBlock) prod[i] = A.val[s + i] * a.x2[A.col[s + i]];
                    lds_barrier();
                    for (int j = b0; j < b1; ++j) sum2 += prod[j];
                    lds_barrier();
                }
            } else {
                // a single long row: the whole workgroup reduces it
                double part = 0.0, part2 = 0.0;
                for (int i = tid; i < cnt; i += kBlock) {
                    part += __dmul_rn(A.val[s + i], a.x[A.col[s + i]]);
                    if (dual) part2 += __dmul_rn(A.val[s + i], a.x2[A.col[s + i]]);
                }
                part = block_sum(part, red);
                if (dual) part2 = block_sum(part2, red);
                row = r0;
                if (tid == 0) {
                    have_row = true;
                    sum = part;
                    sum2 = part2;
                }
            }
            if (nxt.valid && nxt.regular) SCHWZ_ISSUE_LOADS(nxt)
        }
        if (have_row) {
            if (MODE == kSpmvPlain) {
                a.y[row] = (a.beta == 0.0) ? a.alpha * sum : a.alpha * sum + a.beta * a.y[row];
            } else if (MODE == kSpmvDot) {
                a.y[row] = sum;
                acc0 += a.x[row] * sum;
            } else if (MODE == kSpmvResidInit || MODE == kSpmvResidDual) {
                const double bb = a.b[row];
                const double r = bb - sum;
                const double z = a.dinv ? a.dinv[row] * r : r;
                a.y[row] = r;
                a.p[row] = z;
                acc0 += r * z;
                acc1 += r * r;
                if (MODE == kSpmvResidDual && row < a.row_limit) {
                    const double r2 = dual ? bb - sum2 : r;
                    acc2 += r2 * r2;
                }
            } else {  // kSpmvResidNorm
                if (row < a.row_limit) {
                    const double r = a.b[row] - sum;
                    acc1 += r * r;
                }
            }
        }
        cur = nxt;
    }
#undef SCHWZ_ISSUE_LOADS
#undef SCHWZ_PRODUCTS
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

// ---------------------------------------------------------------------------
// Ablation build of the tiled kernel (plain mode only, WRONG results on purpose):
// used by tools/spmv_probe.py --variants 10.. to price the pieces of a tile.
//   bit 0: no x gather (x := 1)      bit 1: no LDS staging / barriers / row sums
//   bit 2: no y store                bit 3: no column-index stream
// ---------------------------------------------------------------------------
template <int WHAT>
__global__ __launch_bounds__(kBlock) void spmv_ablate_kernel(CsrView A, SpmvArgs a)
{
    __shared__ double prod[kTileNnz + 2];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x % kXcds;
    const int slot = blockIdx.x / kXcds;
    const int per_xcd = gridDim.x / kXcds;
    const int chunk = xcd_slots(A);
    double keep = 0.0;
    for (int t = slot; t < chunk; t += per_xcd) {
        const int tile = xcd_tile(A, xcd, t);
        if (tile < 0) continue;
        const int r0 = A.tile_row[tile], r1 = A.tile_row[tile + 1];
        const int s = A.rp[r0], e = A.rp[r1];
        if (e - s > kTileNnz - 2) continue;
        const int row = r0 + tid;
        const int s2 = (WHAT & 32) ? s : (s & ~1);
        int b0 = 0, b1 = 0;
        if (row < r1) {
            b0 = A.rp[row] - s2;
            b1 = A.rp[row + 1] - s2;
        }
        double sum = 0.0;
        if (WHAT & 32) {
            // consecutive lanes take consecutive entries (8-byte / 4-byte loads)
            const int last = max(e - 1, s);
            double v[2 * kPairsPerLane];
            int c[2 * kPairsPerLane];
#pragma unroll
            for (int k = 0; k < 2 * kPairsPerLane; ++k) {
                const int idx = min(s + tid + kBlock * k, last);
                v[k] = A.val[idx];
                c[k] = A.col[idx];
            }
            double xg[2 * kPairsPerLane];
#pragma unroll
            for (int k = 0; k < 2 * kPairsPerLane; ++k) xg[k] = a.x[c[k]];
#pragma unroll
            for (int k = 0; k < 2 * kPairsPerLane; ++k) prod[tid + kBlock * k] = v[k] * xg[k];
            lds_barrier();
            if (row < r1)
                for (int j = b0; j < b1; ++j) sum += prod[j];
            lds_barrier();
        } else {
            const int last = max((e - 1) & ~1, s2);
            double2 v[kPairsPerLane];
            int2 c[kPairsPerLane];
#pragma unroll
            for (int k = 0; k < kPairsPerLane; ++k) {
                const int idx = min(s2 + 2 * (tid + kBlock * k), last);
                v[k] = *reinterpret_cast<const double2 *>(A.val + idx);
                if (WHAT & 8) {
                    c[k].x = idx & 1023;
                    c[k].y = (idx + 1) & 1023;
                } else {
                    c[k] = *reinterpret_cast<const int2 *>(A.col + idx);
                }
            }
            double xg[2 * kPairsPerLane];
#pragma unroll
            for (int k = 0; k < kPairsPerLane; ++k) {
                if (WHAT & 1) {
                    xg[2 * k] = 1.0 + c[k].x;
                    xg[2 * k + 1] = 1.0 + c[k].y;
                } else {
                    xg[2 * k] = a.x[c[k].x];
                    xg[2 * k + 1] = a.x[c[k].y];
                }
            }
            if (WHAT & 2) {
#pragma unroll
                for (int k = 0; k < kPairsPerLane; ++k) sum += v[k].x * xg[2 * k] + v[k].y * xg[2 * k + 1];
                sum += b0 + b1;
            } else {
#pragma unroll
                for (int k = 0; k < kPairsPerLane; ++k) {
                    double2 pr;
                    pr.x = v[k].x * xg[2 * k];
                    pr.y = v[k].y * xg[2 * k + 1];
                    *reinterpret_cast<double2 *>(&prod[2 * (tid + kBlock * k)]) = pr;
                }
                lds_barrier();
                if (row < r1)
                    for (int j = b0; j < b1; ++j) sum += prod[j];
                lds_barrier();
            }
        }
        if (WHAT & 4) {
            keep += sum;
        } else if (WHAT & 16) {
            if (row < r1) __builtin_nontemporal_store(sum, &a.y[row]);
        } else if (row < r1) {
            a.y[row] = sum;
        }
    }
    if ((WHAT & 4) && keep == 123.456) a.y[0] = keep;
}

// ---------------------------------------------------------------------------
// Wave-tiled SpMV: the same idea with a WAVE as the unit of work.  Each 64-lane
// wave owns tiles of <= 64 consecutive rows / <= 510 nonzeros, stages the
// products in its private 4 KiB LDS slice and sums one row per lane.  There is no
// workgroup barrier in the loop (LDS operations of one wave complete in order),
// so the four waves of a workgroup -- and the 32 of a CU -- drift apart and keep
// loads in flight while others are in their LDS phase.
// NT: matrix entries are read once per SpMV; loading them non-temporally keeps
// them from evicting the x planes that neighbouring tiles re-read from L2.
// ---------------------------------------------------------------------------

typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

template <typename T, bool NT>
__device__ __forceinline__ T stream_load(const T *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kWavePairs = kWaveTileNnz / 128;  // 16-byte pairs per lane per tile (4)

template <int MODE, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_wave_kernel(CsrView A, SpmvArgs a)
{
    // every SpMV variant rounds each product before it is added (no FMA contraction), so
    // that all variants -- and the sequential CPU oracle -- produce the same bits
#pragma clang fp contract(off)
    __shared__ double prod_all[kBlock / 64][kWaveTileNnz + 2];
    __shared__ double red[4];
    if (MODE == kSpmvDot || MODE == kSpmvResidInit) {
        if (a.stop_iter && a.it >= *a.stop_iter) return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double *prod = prod_all[wave];
    const int xcd = blockIdx.x % kXcds;
    const int slot = (blockIdx.x / kXcds) * (kBlock / 64) + wave;
    const int per_xcd = (gridDim.x / kXcds) * (kBlock / 64);
    const int chunk = (A.nwtiles + kXcds - 1) / kXcds;
    double acc0 = 0.0, acc1 = 0.0;

    for (int t = slot; t < chunk; t += per_xcd) {
        const int tile = xcd * chunk + t;
        if (tile >= A.nwtiles) break;
        const int r0 = A.wtile_row[tile], r1 = A.wtile_row[tile + 1];
        const int s = A.rp[r0], e = A.rp[r1];
        const int cnt = e - s;
        double sum = 0.0;
        int row = r0 + lane;
        bool have_row = false;
        if (cnt <= kWaveTileNnz - 2) {
            const int s2 = s & ~1;
            int b0 = 0, b1 = 0;
            if (row < r1) {
                b0 = A.rp[row] - s2;
                b1 = A.rp[row + 1] - s2;
            }
            v2d v[kWavePairs];
            v2i c[kWavePairs];
#pragma unroll
            for (int k = 0; k < kWavePairs; ++k) {
                const int idx = s2 + 2 * (lane + 64 * k);
                if (idx < e) {
                    v[k] = stream_load<v2d, NT>(reinterpret_cast<const v2d *>(A.val + idx));
                    c[k] = stream_load<v2i, NT>(reinterpret_cast<const v2i *>(A.col + idx));
                }
            }
#pragma unroll
            for (int k = 0; k < kWavePairs; ++k) {
                const int idx = s2 + 2 * (lane + 64 * k);
                if (idx < e) {
                    v2d pr;
                    pr.x = v[k].x * a.x[c[k].x];
                    pr.y = v[k].y * a.x[c[k].y];
                    *reinterpret_cast<v2d *>(&prod[2 * (lane + 64 * k)]) = pr;
                }
            }
            wave_lds_sync();
            if (row < r1) {
                have_row = true;
                for (int j = b0; j < b1; ++j) sum += prod[j];
            }
            wave_lds_sync();
        } else {
            // a single row longer than a wave tile: the wave reduces it
            double part = 0.0;
            for (int i = lane; i < cnt; i += 64) part += __dmul_rn(A.val[s + i], a.x[A.col[s + i]]);
            part = wave_sum(part);
            row = r0;
            if (lane == 0) {
                have_row = true;
                sum = part;
            }
        }
        if (have_row) {
            if (MODE == kSpmvPlain) {
                a.y[row] = (a.beta == 0.0) ? a.alpha * sum : a.alpha * sum + a.beta * a.y[row];
            } else if (MODE == kSpmvDot) {
                a.y[row] = sum;
                acc0 += a.x[row] * sum;
            } else if (MODE == kSpmvResidInit) {
                const double r = a.b[row] - sum;
                const double z = a.dinv ? a.dinv[row] * r : r;
                a.y[row] = r;
                a.p[row] = z;
                acc0 += r * z;
                acc1 += r * r;
            } else {  // kSpmvResidNorm
                if (row < a.row_limit) {
                    const double r = a.b[row] - sum;
                    acc1 += r * r;
                }
            }
        }
    }
    if (MODE != kSpmvPlain) {
        const double s0 = block_sum(acc0, red);
        const double s1 = block_sum(acc1, red);
        if (threadIdx.x == 0) {
            a.partials[blockIdx.x] = s0;
            a.partials[gridDim.x + blockIdx.x] = s1;
        }
    }
}

// Baseline for A/B runs: one row per lane, no staging (what a direct port of a
// row-parallel CPU loop would do).  Only the plain mode.
__global__ __launch_bounds__(kBlock) void spmv_rowlane_kernel(CsrView A, SpmvArgs a)
{
    // every SpMV variant rounds each product before it is added (no FMA contraction), so
    // that all variants -- and the sequential CPU oracle -- produce the same bits
#pragma clang fp contract(off)
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < A.nrows; row += stride) {
        double sum = 0.0;
        for (int j = A.rp[row]; j < A.rp[row + 1]; ++j) sum += A.val[j] * a.x[A.col[j]];
        a.y[row] = (a.beta == 0.0) ? a.alpha * sum : a.alpha * sum + a.beta * a.y[row];
    }
}

static int launch_spmv_probe(const CsrView &A, int mode, const SpmvArgs &a, int variant, int grid, hipStream_t s)
{
#define SCHWZ_LAUNCH_WAVE(NTV)                                                                          \
    switch (mode) {                                                                                    \
    case kSpmvPlain:                                                                                   \
        hipLaunchKernelGGL((spmv_wave_kernel<kSpmvPlain, NTV>), dim3(wgrid), dim3(kBlock), 0, s, A, a); \
        break;                                                                                         \
    case kSpmvDot:                                                                                     \
        hipLaunchKernelGGL((spmv_wave_kernel<kSpmvDot, NTV>), dim3(wgrid), dim3(kBlock), 0, s, A, a);   \
        break;                                                                                         \
    case kSpmvResidInit:                                                                               \
        hipLaunchKernelGGL((spmv_wave_kernel<kSpmvResidInit, NTV>), dim3(wgrid), dim3(kBlock), 0, s, A, a); \
        break;                                                                                         \
    default:                                                                                           \
        hipLaunchKernelGGL((spmv_wave_kernel<kSpmvResidNorm, NTV>), dim3(wgrid), dim3(kBlock), 0, s, A, a); \
        break;                                                                                         \
    }
    if (variant >= 80) {
        if (mode != kSpmvPlain) {
            set_error("schwz_csr_spmv: the stream ablation builds are y = A x only");
            return SCHWZ_ERR_INVALID;
        }
        return launch_spmv_stream_ablate(A, a, variant - 80, s);
    }
    if (variant >= 10 && variant < 74) {
        if (mode != kSpmvPlain) {
            set_error("schwz_csr_spmv: the ablation builds are y = A x only");
            return SCHWZ_ERR_INVALID;
        }
        switch (variant - 10) {
#define SCHWZ_ABL(W) \
    case W: hipLaunchKernelGGL(spmv_ablate_kernel<W>, dim3(grid), dim3(kBlock), 0, s, A, a); break;
            SCHWZ_ABL(0) SCHWZ_ABL(1) SCHWZ_ABL(2) SCHWZ_ABL(3) SCHWZ_ABL(4) SCHWZ_ABL(5) SCHWZ_ABL(6) SCHWZ_ABL(7)
            SCHWZ_ABL(8) SCHWZ_ABL(9) SCHWZ_ABL(10) SCHWZ_ABL(11) SCHWZ_ABL(12) SCHWZ_ABL(13) SCHWZ_ABL(14) SCHWZ_ABL(15)
            SCHWZ_ABL(16) SCHWZ_ABL(32) SCHWZ_ABL(48) SCHWZ_ABL(36)
#undef SCHWZ_ABL
        }
    } else if (variant == 3 || variant == 5) {
        const int wgrid = spmv_grid(A, variant);
        if (variant == 3) {
            SCHWZ_LAUNCH_WAVE(false)
        } else {
            SCHWZ_LAUNCH_WAVE(true)
        }
    } else if (variant == 1) {
        if (mode != kSpmvPlain) {
            set_error("schwz_csr_spmv: variant 1 is y = A x only");
            return SCHWZ_ERR_INVALID;
        }
        hipLaunchKernelGGL(spmv_rowlane_kernel, dim3(kMaxGrid), dim3(kBlock), 0, s, A, a);
    } else if (variant == 4) {
        switch (mode) {
        case kSpmvPlain: hipLaunchKernelGGL(spmv_pipe_kernel<kSpmvPlain>, dim3(grid), dim3(kBlock), 0, s, A, a); break;
        case kSpmvDot: hipLaunchKernelGGL(spmv_pipe_kernel<kSpmvDot>, dim3(grid), dim3(kBlock), 0, s, A, a); break;
        case kSpmvResidInit: hipLaunchKernelGGL(spmv_pipe_kernel<kSpmvResidInit>, dim3(grid), dim3(kBlock), 0, s, A, a); break;
        case kSpmvResidDual: hipLaunchKernelGGL(spmv_pipe_kernel<kSpmvResidDual>, dim3(grid), dim3(kBlock), 0, s, A, a); break;
        default: hipLaunchKernelGGL(spmv_pipe_kernel<kSpmvResidNorm>, dim3(grid), dim3(kBlock), 0, s, A, a); break;
        }
    } else {  // variant 2
        switch (mode) {
        case kSpmvPlain: hipLaunchKernelGGL(spmv_tiled_kernel<kSpmvPlain>, dim3(grid), dim3(kBlock), 0, s, A, a); break;
        case kSpmvDot: hipLaunchKernelGGL(spmv_tiled_kernel<kSpmvDot>, dim3(grid), dim3(kBlock), 0, s, A, a); break;
        case kSpmvResidInit: hipLaunchKernelGGL(spmv_tiled_kernel<kSpmvResidInit>, dim3(grid), dim3(kBlock), 0, s, A, a); break;
        default: hipLaunchKernelGGL(spmv_tiled_kernel<kSpmvResidNorm>, dim3(grid), dim3(kBlock), 0, s, A, a); break;
        }
    }
#undef SCHWZ_LAUNCH_WAVE
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

// linking this file into the library is what switches the measurement variants on
static const bool g_registered = [] {
    g_spmv_probe_hook = launch_spmv_probe;
    return true;
}();

}  // namespace schwz
