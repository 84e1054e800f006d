// Level-scheduled sparse triangular solves: the direct local solve (y = P^T L^-T L^-1 P b) and the
// sweeps of the ILU(0) preconditioner.  One workgroup for small factors, a launch plan (replayed
// as a hipGraph) for large ones.
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
// level-scheduled sparse triangular solves in ONE workgroup: rows of a level are
// independent; levels are separated by a workgroup barrier.  Used for the
// direct local solve, whose factors are small (BASELINE config 4: ~500 rows).
// ---------------------------------------------------------------------------

constexpr int kTrsBlock = 1024;

// out[i] = in[perm[i]]   (gko Permutation row_permute)
// L t = out ; U out = t ; y[perm[i]] = out[i]
__global__ __launch_bounds__(kTrsBlock) void trs_solve_kernel(
    int64_t n, const schwz_idx *__restrict__ perm, const schwz_idx *__restrict__ l_rp,
    const schwz_idx *__restrict__ l_col, const double *__restrict__ l_val,
    const schwz_idx *__restrict__ l_order, const schwz_idx *__restrict__ l_lvl, int l_nlvl,
    const schwz_idx *__restrict__ u_rp, const schwz_idx *__restrict__ u_col,
    const double *__restrict__ u_val, const schwz_idx *__restrict__ u_order,
    const schwz_idx *__restrict__ u_lvl, int u_nlvl, const double *__restrict__ b,
    double *__restrict__ y, double *w0, double *w1)
{
    const int tid = threadIdx.x;
    for (int64_t i = tid; i < n; i += kTrsBlock) w0[i] = b[perm[i]];
    __threadfence_block();
    __syncthreads();
    for (int lv = 0; lv < l_nlvl; ++lv) {
        for (int k = l_lvl[lv] + tid; k < l_lvl[lv + 1]; k += kTrsBlock) {
            const int row = l_order[k];
            const int e = l_rp[row + 1] - 1;
            double s = w0[row];
            for (int j = l_rp[row]; j < e; ++j) s -= l_val[j] * w1[l_col[j]];
            w1[row] = s / l_val[e];
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int lv = 0; lv < u_nlvl; ++lv) {
        for (int k = u_lvl[lv] + tid; k < u_lvl[lv + 1]; k += kTrsBlock) {
            const int row = u_order[k];
            const int s0 = u_rp[row];
            double s = w1[row];
            for (int j = s0 + 1; j < u_rp[row + 1]; ++j) s -= u_val[j] * w0[u_col[j]];
            w0[row] = s / u_val[s0];
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int64_t i = tid; i < n; i += kTrsBlock) y[perm[i]] = w0[i];
}

// ---- the same solves level by level, for factors that do not fit one workgroup ----------------

// w[i] = b[perm[i]] (perm == nullptr: identity)
__global__ void trs_permute_in_kernel(int64_t n, const schwz_idx *__restrict__ perm, const double *__restrict__ b,
                                      double *__restrict__ w)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        w[i] = perm ? b[perm[i]] : b[i];
}

// y[perm[i]] = w[i]
__global__ void trs_permute_out_kernel(int64_t n, const schwz_idx *__restrict__ perm, const double *__restrict__ w,
                                       double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (perm)
            y[perm[i]] = w[i];
        else
            y[i] = w[i];
    }
}

// rows order[k0:k1) of one level: out[row] = (rhs[row] - sum_{deps} val * out[col]) / diag.
// LOWER: diagonal is the row's last entry; else (upper) its first.  All dependencies belong to
// earlier levels, i.e. to earlier launches.
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void trs_level_kernel(int k0, int k1, const schwz_idx *__restrict__ order,
                                                           const schwz_idx *__restrict__ rp,
                                                           const schwz_idx *__restrict__ col,
                                                           const double *__restrict__ val,
                                                           const double *__restrict__ rhs, double *out)
{
#pragma clang fp contract(off)
    const int k = k0 + blockIdx.x * kBlock + threadIdx.x;
    if (k >= k1) return;
    const int row = order[k];
    const int s0 = rp[row], e = rp[row + 1];
    double s = rhs[row];
    if (LOWER) {
        for (int j = s0; j < e - 1; ++j) s -= val[j] * out[col[j]];
        out[row] = s / val[e - 1];
    } else {
        for (int j = s0 + 1; j < e; ++j) s -= val[j] * out[col[j]];
        out[row] = s / val[s0];
    }
}

// a run of narrow levels [lv0, lv1) in one workgroup
template <bool LOWER>
__global__ __launch_bounds__(kTrsBlock) void trs_narrow_kernel(int lv0, int lv1, const schwz_idx *__restrict__ lvl,
                                                               const schwz_idx *__restrict__ order,
                                                               const schwz_idx *__restrict__ rp,
                                                               const schwz_idx *__restrict__ col,
                                                               const double *__restrict__ val,
                                                               const double *__restrict__ rhs, double *out)
{
#pragma clang fp contract(off)
    for (int lv = lv0; lv < lv1; ++lv) {
        for (int k = lvl[lv] + (int)threadIdx.x; k < lvl[lv + 1]; k += kTrsBlock) {
            const int row = order[k];
            const int s0 = rp[row], e = rp[row + 1];
            double s = rhs[row];
            if (LOWER) {
                for (int j = s0; j < e - 1; ++j) s -= val[j] * out[col[j]];
                out[row] = s / val[e - 1];
            } else {
                for (int j = s0 + 1; j < e; ++j) s -= val[j] * out[col[j]];
                out[row] = s / val[s0];
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---- the same solves WITHOUT level-by-level launches: one persistent launch per sweep ---------------
// The launch plan above spends ~6.6 us per level on dependent-kernel spacing (760 levels for the ILU(0)
// factors of a 128^3 grid: 5.1 ms per application).  Here every row of the level-sorted order is taken by
// one lane of a resident wave (chunks of 64 positions dealt round robin to the waves), and a lane waits
// for its dependencies row by row: the solution vector itself is the flag.  It starts out as a NaN no
// arithmetic produces (kTrsUnset), every finished row is stored with ONE 8-byte agent-scope store
// (write-through, MI355X_MICROARCH.md "R2 granule": the data is the flag, no fence), and waiting lanes
// poll with agent-scope loads that bypass their L1.  Progress: the chunk holding the smallest unfinished
// position is being worked on by its wave (every wave takes its chunks in ascending order and the whole
// grid is resident), its rows depend on smaller positions only, and lanes whose dependencies are met
// finish INSIDE the wait loop, so no lane waits for a lane of its own wave.  Each row sums its entries
// in CSR order like trs_level_kernel: the same bits.  A wait that exceeds kTrsTimeoutTicks gives up with
// a NaN (the CG's residual check then fails loudly) instead of hanging the GPU.
#ifndef SCHWZ_TRS_SLEEP_EVERY
#define SCHWZ_TRS_SLEEP_EVERY 4  // a power of two, 0: never sleep between polls
#endif
constexpr unsigned long long kTrsUnset = 0x7ff8dead7ff8deadull;
constexpr unsigned long long kTrsTimeoutTicks = 300000000ull;  // 3 s of the 100 MHz constant clock
typedef unsigned long long u64;

// The factor is stored a second time in LEVEL ORDER (rows permuted to their position in the level-sorted
// order, columns translated to positions, entries of a row in their original order): lane k of the grid
// owns position k, its dependencies sit at smaller positions, and the 64 lanes of a wave read row data,
// poll and publish at (nearly) consecutive addresses -- polls that would otherwise touch up to 64 lines
// per instruction are what limits how many waves may wait at once.  rhs / y are gathered / scattered
// through one index per row (src / dst).
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void trs_flag_kernel(int64_t n, const schwz_idx *__restrict__ rp,
                                                          const schwz_idx *__restrict__ col,
                                                          const double *__restrict__ val, const double *rhs,
                                                          const schwz_idx *__restrict__ src, u64 *out, u64 *reset,
                                                          double *y_out, const schwz_idx *__restrict__ dst, int *err)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const u64 t_start = __builtin_amdgcn_s_memrealtime();
    for (int64_t c = wave; c * 64 < n; c += nwaves) {
        const int64_t k = c * 64 + lane;
        bool done = k >= n;
        int j = 0, jend = 0;
        double s = 0.0, diag = 1.0;
        if (!done) {
            const int s0 = rp[k], e = rp[k + 1];
            if (LOWER) {
                j = s0;
                jend = e - 1;
                diag = val[e - 1];
            } else {
                j = s0 + 1;
                jend = e;
                diag = val[s0];
            }
            s = rhs[src[k]];
        }
        unsigned spins = 0;
        // the dependencies are requested four at a time and consumed in the row's entry order; their
        // positions and values are fetched once per group, only the polls repeat
        int cc[4] = {0, 0, 0, 0};
        double vv[4] = {0.0, 0.0, 0.0, 0.0};
        int loaded_at = -1;
        while (true) {
            if (!done) {
                if (loaded_at != j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool in = j + q < jend;
                        cc[q] = in ? col[j + q] : (int)k;
                        vv[q] = in ? val[j + q] : 0.0;
                    }
                    loaded_at = j;
                }
                u64 got[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    got[q] = j + q < jend ? __hip_atomic_load(out + cc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                bool ok = true;
                int used = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (j + q < jend && ok) {
                        if (got[q] != kTrsUnset) {
                            s -= vv[q] * __longlong_as_double((long long)got[q]);
                            ++used;
                        } else {
                            ok = false;
                        }
                    }
                }
                j += used;  // a group that moved on is fetched anew (loaded_at != j)
                if (j == jend) {
                    const double res = s / diag;
                    __hip_atomic_store(out + k, (u64)__double_as_longlong(res), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // the other sweep's vector, ready for the next solve: the upper sweep clears the entry it
                    // took its right-hand side from (nobody else reads it), the lower sweep any one entry
                    reset[LOWER ? k : (int64_t)src[k]] = kTrsUnset;
                    if (!LOWER) y_out[dst[k]] = res;
                    done = true;
                }
            }
            if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
            if ((++spins & 255u) == 0 && __builtin_amdgcn_s_memrealtime() - t_start > kTrsTimeoutTicks) {
                if (!done) {
                    __hip_atomic_store(out + k, 0x7ff8000000000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    reset[LOWER ? k : (int64_t)src[k]] = kTrsUnset;
                    if (!LOWER) y_out[dst[k]] = __longlong_as_double(0x7ff8000000000000ll);
                    *err = 1;
                    done = true;
                }
            } else if (SCHWZ_TRS_SLEEP_EVERY && (spins & (SCHWZ_TRS_SLEEP_EVERY - 1)) == 0) {
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
}

// A flag-driven sweep that waited longer than kTrsTimeoutTicks gave up with NaNs and set the error word: read
// and clear it (callers stand at a point where the stream is idle: the NaN has already reached the host).
int trs_take_error(schwz_trs *t)
{
    if (!t || !t->flags || !t->d_err) return SCHWZ_OK;
    int e = 0;
    SCHWZ_HIP_TRY(hipMemcpy(&e, t->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (!e) return SCHWZ_OK;
    e = 0;
    SCHWZ_HIP_TRY(hipMemcpy(t->d_err, &e, sizeof(int), hipMemcpyHostToDevice));
    set_error("trs flag sweep timed out: a row waited more than 3 s for its dependencies (workgroups of the "
              "persistent triangular sweep not resident together?); SCHWZ_TRS_FLAGS=0 selects the launch plan");
    return SCHWZ_ERR_HIP;
}

}  // namespace schwz

using namespace schwz;

extern "C" {

// ---- triangular solves --------------------------------------------------------

// levels of a triangular CSR: lower => forward dependencies on columns < row
static void level_schedule(int64_t n, const schwz_idx *rp, const schwz_idx *col, bool lower,
                           std::vector<schwz_idx> &order, std::vector<schwz_idx> &lvl_ptr)
{
    std::vector<schwz_idx> level((size_t)n, 0);
    schwz_idx nl = 0;
    if (lower) {
        for (int64_t i = 0; i < n; ++i) {
            schwz_idx l = 0;
            for (schwz_idx j = rp[i]; j < rp[i + 1]; ++j)
                if (col[j] < i && level[col[j]] + 1 > l) l = level[col[j]] + 1;
            level[i] = l;
            if (l + 1 > nl) nl = l + 1;
        }
    } else {
        for (int64_t i = n - 1; i >= 0; --i) {
            schwz_idx l = 0;
            for (schwz_idx j = rp[i]; j < rp[i + 1]; ++j)
                if (col[j] > i && level[col[j]] + 1 > l) l = level[col[j]] + 1;
            level[i] = l;
            if (l + 1 > nl) nl = l + 1;
        }
    }
    lvl_ptr.assign((size_t)nl + 1, 0);
    for (int64_t i = 0; i < n; ++i) lvl_ptr[level[i] + 1]++;
    for (schwz_idx l = 0; l < nl; ++l) lvl_ptr[l + 1] += lvl_ptr[l];
    order.resize((size_t)n);
    std::vector<schwz_idx> fill(lvl_ptr.begin(), lvl_ptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) order[fill[level[i]]++] = (schwz_idx)i;
}

int schwz_trs_create(int64_t n, const schwz_idx *l_rp, const schwz_idx *l_col, const double *l_val,
                     const schwz_idx *u_rp, const schwz_idx *u_col, const double *u_val,
                     const schwz_idx *perm, schwz_trs **out)
{
    SCHWZ_REQUIRE(out && n >= 0 && l_rp && u_rp, "schwz_trs_create: bad arguments");
    for (int64_t i = 0; i < n; ++i) {
        SCHWZ_REQUIRE(l_rp[i + 1] > l_rp[i] && l_col[l_rp[i + 1] - 1] == i,
                      "schwz_trs_create: L must hold its diagonal last in each row");
        SCHWZ_REQUIRE(u_rp[i + 1] > u_rp[i] && u_col[u_rp[i]] == i,
                      "schwz_trs_create: U must hold its diagonal first in each row");
        SCHWZ_REQUIRE(!perm || (perm[i] >= 0 && perm[i] < n), "schwz_trs_create: permutation out of range");
    }
    schwz_trs *t = new schwz_trs();
    t->n = n;
    std::vector<schwz_idx> lo, ll, uo, ul;
    level_schedule(n, l_rp, l_col, true, lo, ll);
    level_schedule(n, u_rp, u_col, false, uo, ul);
    t->l_nlvl = (int)ll.size() - 1;
    t->u_nlvl = (int)ul.size() - 1;
    // One workgroup handles the whole solve while the factor is small; otherwise wide levels get
    // a multi-workgroup launch each and runs of narrow levels share a one-workgroup launch.
    t->fused = n <= 8192;
    // a level of >= wide_min rows gets a launch of its own (SCHWZ_TRS_WIDE overrides the threshold)
    const char *wenv = std::getenv("SCHWZ_TRS_WIDE");
    const int wide_min = (wenv && std::atoi(wenv) > 0) ? std::atoi(wenv) : 256;  // measured: 4096 -> 11.8, 1024 -> 5.4, 256 -> 5.1 ms per ILU-CG iteration at 128^3
    auto plan = [wide_min](const std::vector<schwz_idx> &lvl, std::vector<schwz_trs::Seg> &out) {
        const int nl = (int)lvl.size() - 1;
        int l = 0;
        while (l < nl) {
            if (lvl[(size_t)l + 1] - lvl[(size_t)l] >= wide_min) {
                out.push_back({l, l + 1, true});
                ++l;
            } else {
                int e = l;
                while (e < nl && lvl[(size_t)e + 1] - lvl[(size_t)e] < wide_min) ++e;
                out.push_back({l, e, false});
                l = e;
            }
        }
    };
    plan(ll, t->l_plan);
    plan(ul, t->u_plan);
    t->h_l_lvl = ll;
    t->h_u_lvl = ul;
    int rc = 0;
    void *d;
#define UP(dst, src, cnt, T)                        \
    if (!rc) {                                      \
        rc = upload<T>((src), (size_t)(cnt), &d);   \
        dst = (decltype(dst))d;                     \
    }
    UP(t->l_rp, l_rp, n + 1, schwz_idx)
    UP(t->l_col, l_col, l_rp[n], schwz_idx)
    UP(t->l_val, l_val, l_rp[n], double)
    UP(t->u_rp, u_rp, n + 1, schwz_idx)
    UP(t->u_col, u_col, u_rp[n], schwz_idx)
    UP(t->u_val, u_val, u_rp[n], double)
    if (perm) {
        UP(t->perm, perm, n, schwz_idx)
    }
    UP(t->l_order, lo.data(), lo.size(), schwz_idx)
    UP(t->l_lvl, ll.data(), ll.size(), schwz_idx)
    UP(t->u_order, uo.data(), uo.size(), schwz_idx)
    UP(t->u_lvl, ul.data(), ul.size(), schwz_idx)
#undef UP
    if (rc) {
        schwz_trs_destroy(t);
        return rc;
    }
    SCHWZ_HIP_TRY(hipMalloc((void **)&t->w0, sizeof(double) * (size_t)(n ? n : 1)));
    SCHWZ_HIP_TRY(hipMalloc((void **)&t->w1, sizeof(double) * (size_t)(n ? n : 1)));
    // flag-driven sweeps (one persistent launch per factor) for factors beyond the one-workgroup kernel
    // whose rows are short enough for one lane each; SCHWZ_TRS_FLAGS=0: the level-by-level plan
    {
        const char *fenv = std::getenv("SCHWZ_TRS_FLAGS");
        int64_t longest = 0;
        for (int64_t i = 0; i < n; ++i)
            longest = std::max<int64_t>(longest, std::max<int64_t>(l_rp[i + 1] - l_rp[i], u_rp[i + 1] - u_rp[i]));
        if (!t->fused && !(fenv && fenv[0] == '0') && longest <= 64) {
            // the factors in level order (see trs_flag_kernel)
            std::vector<schwz_idx> lpos((size_t)n), upos((size_t)n);
            for (int64_t k = 0; k < n; ++k) {
                lpos[(size_t)lo[(size_t)k]] = (schwz_idx)k;
                upos[(size_t)uo[(size_t)k]] = (schwz_idx)k;
            }
            auto permuted = [&](const schwz_idx *rp0, const schwz_idx *col0, const double *val0,
                                const std::vector<schwz_idx> &ord, const std::vector<schwz_idx> &pos,
                                std::vector<schwz_idx> &rp1, std::vector<schwz_idx> &col1, std::vector<double> &val1) {
                rp1.assign((size_t)n + 1, 0);
                col1.resize((size_t)rp0[n]);
                val1.resize((size_t)rp0[n]);
                for (int64_t k = 0; k < n; ++k) {
                    const schwz_idx r = ord[(size_t)k];
                    schwz_idx w = rp1[(size_t)k];
                    for (schwz_idx j = rp0[r]; j < rp0[r + 1]; ++j, ++w) {
                        col1[(size_t)w] = pos[(size_t)col0[j]];
                        val1[(size_t)w] = val0[j];
                    }
                    rp1[(size_t)k + 1] = w;
                }
            };
            std::vector<schwz_idx> prp, pcol, lsrc((size_t)n), usrc((size_t)n), udst((size_t)n);
            std::vector<double> pval;
            int rc2 = 0;
            permuted(l_rp, l_col, l_val, lo, lpos, prp, pcol, pval);
            if (!rc2) rc2 = upload(prp.data(), prp.size(), &d), t->fl_rp = (schwz_idx *)d;
            if (!rc2) rc2 = upload(pcol.data(), pcol.size(), &d), t->fl_col = (schwz_idx *)d;
            if (!rc2) rc2 = upload(pval.data(), pval.size(), &d), t->fl_val = (double *)d;
            permuted(u_rp, u_col, u_val, uo, upos, prp, pcol, pval);
            if (!rc2) rc2 = upload(prp.data(), prp.size(), &d), t->fu_rp = (schwz_idx *)d;
            if (!rc2) rc2 = upload(pcol.data(), pcol.size(), &d), t->fu_col = (schwz_idx *)d;
            if (!rc2) rc2 = upload(pval.data(), pval.size(), &d), t->fu_val = (double *)d;
            for (int64_t k = 0; k < n; ++k) {
                lsrc[(size_t)k] = perm ? perm[lo[(size_t)k]] : lo[(size_t)k];
                usrc[(size_t)k] = lpos[(size_t)uo[(size_t)k]];
                udst[(size_t)k] = perm ? perm[uo[(size_t)k]] : uo[(size_t)k];
            }
            if (!rc2) rc2 = upload(lsrc.data(), lsrc.size(), &d), t->fl_src = (schwz_idx *)d;
            if (!rc2) rc2 = upload(usrc.data(), usrc.size(), &d), t->fu_src = (schwz_idx *)d;
            if (!rc2) rc2 = upload(udst.data(), udst.size(), &d), t->fu_dst = (schwz_idx *)d;
            std::vector<unsigned long long> unset((size_t)n, kTrsUnset);
            if (!rc2) rc2 = upload(unset.data(), unset.size(), &d), t->f0 = (unsigned long long *)d;
            if (!rc2) {
                rc2 = upload(unset.data(), unset.size(), &d);
                t->f1 = (unsigned long long *)d;
            }
            const int zero = 0;
            if (!rc2) {
                rc2 = upload(&zero, 1, &d);
                t->d_err = (int *)d;
            }
            if (rc2) {
                schwz_trs_destroy(t);
                return rc2;
            }
            t->flags = true;
            int dev = 0, cus = 256;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
            // Every workgroup must be resident (the waits are on other workgroups' rows).  The grid is one
            // workgroup per CU (SCHWZ_TRS_FLAG_GRID overrides, at most 4 per CU) and is clamped to what the
            // occupancy query says both sweeps can keep resident -- less one per CU beyond the first, because
            // the query may answer one workgroup per CU too many at some SGPR counts (MI355X_MICROARCH.md,
            // "Correctness boundaries").  No residency at all: the level-by-level launch plan.
            int occ_l = 0, occ_u = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_l, trs_flag_kernel<true>, kBlock, 0) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_u, trs_flag_kernel<false>, kBlock, 0) != hipSuccess) {
                (void)hipGetLastError();
                occ_l = occ_u = 0;
            }
            int per_cu = std::min(occ_l, occ_u);
            if (per_cu > 1) --per_cu;
            const char *genv = std::getenv("SCHWZ_TRS_FLAG_GRID");
            int g = genv ? std::atoi(genv) : cus;
            g = std::max(1, std::min(g, std::min(4, per_cu) * cus));
            t->flag_grid = (int)std::min<int64_t>(g, (n + kBlock - 1) / kBlock);
            if (per_cu < 1) t->flags = false;
        }
    }
    *out = t;
    return SCHWZ_OK;
}

void schwz_trs_destroy(schwz_trs *t)
{
    if (!t) return;
    void *ptrs[] = {t->l_rp, t->l_col, t->l_val, t->u_rp, t->u_col, t->u_val, t->perm,
                    t->l_order, t->l_lvl, t->u_order, t->u_lvl, t->w0, t->w1, t->f0, t->f1, t->d_err,
                    t->fl_rp, t->fl_col, t->fl_val, t->fu_rp, t->fu_col, t->fu_val, t->fl_src, t->fu_src, t->fu_dst};
    for (void *p : ptrs) (void)hipFree(p);
    for (auto &g : t->graphs) (void)hipGraphExecDestroy(g.exec);
    if (t->capture_stream) (void)hipStreamDestroy(t->capture_stream);
    delete t;
}

int schwz_trs_solve(schwz_trs *t, const double *d_b, double *d_y, schwz_stream stream)
{
    SCHWZ_REQUIRE(t && d_b && d_y, "schwz_trs_solve: null argument");
    if (t->n == 0) return SCHWZ_OK;
    hipStream_t st = (hipStream_t)stream;
    if (t->fused && t->perm) {
        hipLaunchKernelGGL(trs_solve_kernel, dim3(1), dim3(kTrsBlock), 0, st, t->n, t->perm, t->l_rp, t->l_col,
                           t->l_val, t->l_order, t->l_lvl, t->l_nlvl, t->u_rp, t->u_col, t->u_val, t->u_order,
                           t->u_lvl, t->u_nlvl, d_b, d_y, t->w0, t->w1);
        SCHWZ_HIP_TRY(hipGetLastError());
        return SCHWZ_OK;
    }
    if (t->flags) {
        // L f1 = P b ; U f0 = f1 ; y = P^T f0 -- two persistent launches, rows hand over through the vectors
        hipLaunchKernelGGL((trs_flag_kernel<true>), dim3(t->flag_grid), dim3(kBlock), 0, st, t->n, t->fl_rp, t->fl_col,
                           t->fl_val, d_b, t->fl_src, t->f1, t->f0, (double *)nullptr, (const schwz_idx *)nullptr, t->d_err);
        hipLaunchKernelGGL((trs_flag_kernel<false>), dim3(t->flag_grid), dim3(kBlock), 0, st, t->n, t->fu_rp, t->fu_col,
                           t->fu_val, reinterpret_cast<const double *>(t->f1), t->fu_src, t->f0, t->f1, d_y, t->fu_dst,
                           t->d_err);
        SCHWZ_HIP_TRY(hipGetLastError());
        return SCHWZ_OK;
    }
    // The level-by-level plan is a fixed sequence of launches for given (b, y): it is captured once
    // into a hipGraph (on a private stream: the caller's may be the legacy default stream, which
    // cannot capture) and replayed with one graph launch afterwards.  SCHWZ_TRS_GRAPH=0 disables.
    static const bool graphs_on = [] {
        const char *e = std::getenv("SCHWZ_TRS_GRAPH");
        return !(e && e[0] == '0');
    }();
    hipStream_t user_stream = st;
    bool capturing = false;
    if (graphs_on && !t->graphs_failed && t->l_plan.size() + t->u_plan.size() > 8) {
        for (const auto &g : t->graphs)
            if (g.b == d_b && g.y == d_y) {
                SCHWZ_HIP_TRY(hipGraphLaunch(g.exec, user_stream));
                return SCHWZ_OK;
            }
        if (t->graphs.size() < 4) {
            if (!t->capture_stream) SCHWZ_HIP_TRY(hipStreamCreateWithFlags(&t->capture_stream, hipStreamNonBlocking));
            if (hipStreamBeginCapture(t->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                capturing = true;
                st = t->capture_stream;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    // w0 = P b ; L w1 = w0 ; U w0 = w1 ; y = P^T w0
    hipLaunchKernelGGL(trs_permute_in_kernel, dim3(grid_for(t->n)), dim3(kBlock), 0, st, t->n, t->perm, d_b, t->w0);
    for (const auto &sg : t->l_plan) {
        if (sg.wide) {
            const int k0 = t->h_l_lvl[(size_t)sg.lvl0], k1 = t->h_l_lvl[(size_t)sg.lvl1];
            hipLaunchKernelGGL((trs_level_kernel<true>), dim3((k1 - k0 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, k0,
                               k1, t->l_order, t->l_rp, t->l_col, t->l_val, t->w0, t->w1);
        } else {
            hipLaunchKernelGGL((trs_narrow_kernel<true>), dim3(1), dim3(kTrsBlock), 0, st, sg.lvl0, sg.lvl1, t->l_lvl,
                               t->l_order, t->l_rp, t->l_col, t->l_val, t->w0, t->w1);
        }
    }
    for (const auto &sg : t->u_plan) {
        if (sg.wide) {
            const int k0 = t->h_u_lvl[(size_t)sg.lvl0], k1 = t->h_u_lvl[(size_t)sg.lvl1];
            hipLaunchKernelGGL((trs_level_kernel<false>), dim3((k1 - k0 + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                               k0, k1, t->u_order, t->u_rp, t->u_col, t->u_val, t->w1, t->w0);
        } else {
            hipLaunchKernelGGL((trs_narrow_kernel<false>), dim3(1), dim3(kTrsBlock), 0, st, sg.lvl0, sg.lvl1, t->u_lvl,
                               t->u_order, t->u_rp, t->u_col, t->u_val, t->w1, t->w0);
        }
    }
    hipLaunchKernelGGL(trs_permute_out_kernel, dim3(grid_for(t->n)), dim3(kBlock), 0, st, t->n, t->perm, t->w0, d_y);
    if (capturing) {
        hipGraph_t graph = nullptr;
        if (hipStreamEndCapture(t->capture_stream, &graph) != hipSuccess || !graph) {
            (void)hipGetLastError();
            t->graphs_failed = true;
            return schwz_trs_solve(t, d_b, d_y, stream);
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {  // no graph on this system: launch by launch from now on
            (void)hipGetLastError();
            t->graphs_failed = true;
            return schwz_trs_solve(t, d_b, d_y, stream);
        }
        t->graphs.push_back({d_b, d_y, exec});
        SCHWZ_HIP_TRY(hipGraphLaunch(exec, user_stream));
        return SCHWZ_OK;
    }
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}


}  // extern "C"
