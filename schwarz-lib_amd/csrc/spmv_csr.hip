// Plain-CSR SpMV for gfx950 (MI355X): the tiled kernel (LDS-staged products, 16-byte batched loads, XCD-aware
// tile deal) behind the straight-line kernel of spmv_stream.hip, and the dispatch of the SpMV variants.
// The rejected experiments (one row per lane, wave-private tiles, pipelined tiles, ablation builds) live in
// tools/probes/spmv_variants.hip and are only linked into the measurement build libschwz_hip_probes.so
// (`make probes`).  HBM-bandwidth bound fp64 / int32 streaming work: no dense contraction exists on this
// path, MFMA is deliberately unused (BASELINE.json north_star).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "schwz_internal.hpp"
#include "device_utils.hpp"

namespace schwz {

// ---------------------------------------------------------------------------
// Default (variant 0) tiled SpMV: the same tile table as the first version
// (spmv_tiled_kernel, kept as variant 2 for A/B runs), but the streaming phase is
// issued as 16-byte loads (double2 values, int2 columns) and all of a lane's
// loads are in flight before the first use (8 nonzeros per lane per tile), so a
// wave keeps ~1.5 KiB outstanding instead of one 12-byte pair.  The tile window
// starts at the even index below rp[r0]; the at most two foreign entries at the
// window's ends are multiplied like the others and simply never summed (the
// arrays carry four padding entries, see schwz_csr_create).
// ---------------------------------------------------------------------------

constexpr int kQuadsPerLane = kTileNnz / (4 * kBlock);  // 2

template <int MODE>
__global__ __launch_bounds__(kBlock) void spmv_tiled2_kernel(CsrView A, SpmvArgs a)
{
    // every SpMV variant rounds each product before it is added (no FMA contraction), so
    // that all variants -- and the sequential CPU oracle -- produce the same bits
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) double prod[kTileNnz + 4];
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
        const int tl = A.tile_order ? A.tile_order[tile] : tile;
        const int r0 = A.tile_row[tl], r1 = A.tile_row[tl + 1];
        const int s = A.rp[r0], e = A.rp[r1];
        const int cnt = e - s;
        const bool dual_t = dual && (!A.tile_dual || A.tile_dual[tl]);
        double sum = 0.0, sum2 = 0.0;
        int row = r0 + tid;
        bool have_row = false;
        if ((s & 3) + cnt <= kTileNnz) {
            const int s2 = s & ~3;  // 16-byte aligned window of the column indices (32-byte of the values)
            // own row bounds first: independent of the streaming loads
            int b0 = 0, b1 = 0;
            if (row < r1) {
                b0 = A.rp[row] - s2;
                b1 = A.rp[row + 1] - s2;
            }
            // Four consecutive entries per lane and trip: two 16-byte value loads and ONE 16-byte
            // index load (the vector-memory pipe is priced per instruction).  No bounds branch: lanes
            // past the tile re-read its last quad (their products land in LDS slots no row sums).
            const int last = max((e - 1) & ~3, s2);
            double2 v[2 * kQuadsPerLane];
            int4 c[kQuadsPerLane];
#pragma unroll
            for (int k = 0; k < kQuadsPerLane; ++k) {
                const int idx = min(s2 + 4 * (tid + kBlock * k), last);
                v[2 * k] = *reinterpret_cast<const double2 *>(A.val + idx);
                v[2 * k + 1] = *reinterpret_cast<const double2 *>(A.val + idx + 2);
                c[k] = *reinterpret_cast<const int4 *>(A.col + idx);
            }
            double xg[4 * kQuadsPerLane];
#pragma unroll
            for (int k = 0; k < kQuadsPerLane; ++k) {
                xg[4 * k] = a.x[c[k].x];
                xg[4 * k + 1] = a.x[c[k].y];
                xg[4 * k + 2] = a.x[c[k].z];
                xg[4 * k + 3] = a.x[c[k].w];
            }
#pragma unroll
            for (int k = 0; k < kQuadsPerLane; ++k) {
                double2 p0, p1;
                p0.x = v[2 * k].x * xg[4 * k];
                p0.y = v[2 * k].y * xg[4 * k + 1];
                p1.x = v[2 * k + 1].x * xg[4 * k + 2];
                p1.y = v[2 * k + 1].y * xg[4 * k + 3];
                double2 *dst = reinterpret_cast<double2 *>(&prod[4 * (tid + kBlock * k)]);
                dst[0] = p0;
                dst[1] = p1;
            }
            lds_barrier();
            if (row < r1) {
                have_row = true;
                for (int j = b0; j < b1; ++j) sum += prod[j];
            }
            lds_barrier();
            if (dual_t) {
                // second vector, same matrix entries (still in registers)
#pragma unroll
                for (int k = 0; k < kQuadsPerLane; ++k) {
                    xg[4 * k] = a.x2[c[k].x];
                    xg[4 * k + 1] = a.x2[c[k].y];
                    xg[4 * k + 2] = a.x2[c[k].z];
                    xg[4 * k + 3] = a.x2[c[k].w];
                }
#pragma unroll
                for (int k = 0; k < kQuadsPerLane; ++k) {
                    double2 p0, p1;
                    p0.x = v[2 * k].x * xg[4 * k];
                    p0.y = v[2 * k].y * xg[4 * k + 1];
                    p1.x = v[2 * k + 1].x * xg[4 * k + 2];
                    p1.y = v[2 * k + 1].y * xg[4 * k + 3];
                    double2 *dst = reinterpret_cast<double2 *>(&prod[4 * (tid + kBlock * k)]);
                    dst[0] = p0;
                    dst[1] = p1;
                }
                lds_barrier();
                if (row < r1)
                    for (int j = b0; j < b1; ++j) sum2 += prod[j];
                lds_barrier();
            }
        } else if (r1 - r0 > 1) {
            // tile that fits kTileNnz but not the aligned window: plain staging
            for (int i = tid; i < cnt; i += kBlock) prod[i] = A.val[s + i] * a.x[A.col[s + i]];
            lds_barrier();
            const int b0 = row < r1 ? A.rp[row] - s : 0, b1 = row < r1 ? A.rp[row + 1] - s : 0;
            if (row < r1) {
                have_row = true;
                for (int j = b0; j < b1; ++j) sum += prod[j];
            }
            lds_barrier();
            if (dual_t) {
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
                if (dual_t) part2 += __dmul_rn(A.val[s + i], a.x2[A.col[s + i]]);
            }
            part = block_sum(part, red);
            if (dual_t) part2 = block_sum(part2, red);
            row = r0;
            if (tid == 0) {
                have_row = true;
                sum = part;
                sum2 = part2;
            }
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
                    const double r2 = dual_t ? bb - sum2 : r;
                    acc2 += r2 * r2;
                }
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
        if (MODE == kSpmvResidDual) {
            const double s2v = block_sum(acc2, red);
            if (tid == 0) a.partials[2 * gridDim.x + blockIdx.x] = s2v;
        }
    }
}


int spmv_grid(const CsrView &A, int variant)
{
    // wave variants: four wave tiles per workgroup
    const int units = (variant == 3 || variant == 5) ? (A.nwtiles + 3) / 4 : A.ntiles;
    static const int cap = [] {  // SCHWZ_SPMV_GRID: smaller grids for occupancy experiments
        const char *e = std::getenv("SCHWZ_SPMV_GRID");
        const int v = e ? std::atoi(e) : 0;
        return (v >= kXcds && v < kMaxGrid) ? v : kMaxGrid;
    }();
    int g = units < cap ? units : cap;
    g = ((g + kXcds - 1) / kXcds) * kXcds;
    return g < kXcds ? kXcds : g;
}

// set by the measurement build (tools/probes/spmv_variants.hip); null in libschwz_hip.so
SpmvProbeHook g_spmv_probe_hook = nullptr;

int launch_spmv(const CsrView &A, int mode, const SpmvArgs &a, int variant, hipStream_t s)
{
    if (A.nrows == 0) return SCHWZ_OK;
    if (mode == kSpmvResidDual && variant != 0 && variant != 4 && variant != 6 && variant != 7 && variant != 8 &&
        variant != 9) {
        set_error("launch_spmv: the fused dual-residual mode exists for variants 0, 4, 6, 7, 8 and 9 only");
        return SCHWZ_ERR_INVALID;
    }
    const int grid = spmv_grid(A, variant);
    if (variant == 0 && A.pair_id) return launch_spmv_pair(A, mode, a, grid, s);
    if (mode == kSpmvDotOnly || mode == kSpmvDotSym || mode == kSpmvDirDotSym || mode == kSpmvDirDotSymVec ||
        mode == kSpmvCgUpdate) {
        set_error("launch_spmv: the q-free CG modes exist for row-pair coded matrices only");
        return SCHWZ_ERR_INVALID;
    }
    if ((variant == 0 || variant == 8) && A.pat_id) return launch_spmv_pattern(A, mode, a, grid, s);
    if ((variant == 0 || variant == 7) && A.code) return launch_spmv_dict(A, mode, a, grid, s);
    if (variant == 1 || variant == 2 || variant == 3 || variant == 4 || variant == 5 || (variant >= 10 && variant < 74) ||
        variant >= 80) {
        // measurement variants: tools/probes/spmv_variants.hip and the ablation builds of spmv_stream.hip
        if (!g_spmv_probe_hook) {
            set_error("schwz_csr_spmv: variant " + std::to_string(variant) + " is a measurement build; it is linked into "
                      "libschwz_hip_probes.so only (make -C schwarz-lib_amd probes; SCHWZ_HIP_LIB selects the library)");
            return SCHWZ_ERR_INVALID;
        }
        return g_spmv_probe_hook(A, mode, a, variant, grid, s);
    }
    {
        // variants 0 / 6: the straight-line pipeline of spmv_stream.hip where it applies (rows of at most 32
        // entries, not the fused dual residual); variant 9 keeps spmv_tiled2_kernel for A/B runs
        if (variant == 0 || variant == 6) {
            bool done = false;
            const int rc = launch_spmv_stream(A, mode, a, grid, s, &done);
            if (rc || done) return rc;
        }
        switch (mode) {
        case kSpmvPlain:
            hipLaunchKernelGGL(spmv_tiled2_kernel<kSpmvPlain>, dim3(grid), dim3(kBlock), 0, s, A, a);
            break;
        case kSpmvDot:
            hipLaunchKernelGGL(spmv_tiled2_kernel<kSpmvDot>, dim3(grid), dim3(kBlock), 0, s, A, a);
            break;
        case kSpmvResidInit:
            hipLaunchKernelGGL(spmv_tiled2_kernel<kSpmvResidInit>, dim3(grid), dim3(kBlock), 0, s, A, a);
            break;
        case kSpmvResidDual:
            hipLaunchKernelGGL(spmv_tiled2_kernel<kSpmvResidDual>, dim3(grid), dim3(kBlock), 0, s, A, a);
            break;
        default:
            hipLaunchKernelGGL(spmv_tiled2_kernel<kSpmvResidNorm>, dim3(grid), dim3(kBlock), 0, s, A, a);
            break;
        }
    }
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}


}  // namespace schwz
