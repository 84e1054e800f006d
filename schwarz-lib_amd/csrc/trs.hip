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
    *out = t;
    return SCHWZ_OK;
}

void schwz_trs_destroy(schwz_trs *t)
{
    if (!t) return;
    void *ptrs[] = {t->l_rp, t->l_col, t->l_val, t->u_rp, t->u_col, t->u_val, t->perm,
                    t->l_order, t->l_lvl, t->u_order, t->u_lvl, t->w0, t->w1};
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
