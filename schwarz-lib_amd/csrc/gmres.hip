// Restarted GMRES(m) with right preconditioning: the local solver of the non-symmetric branch of
// Solve::setup_local_solver / Solve::local_solve (source/solve.cpp:486-520, 750-753;
// gko::solver::Gmres with krylov_dim = settings.restart_iter and the Combined(Iteration,
// ResidualNormReduction) criterion of solve.cpp:469-479).  Ginkgo's source is not part of the
// reference checkout: the algorithm is Saad & Schultz (1986) with modified Gram-Schmidt and Givens
// rotations, the same restatement as oracle/schwz_oracle.c::schwz_or_gmres, which scipy's GMRES
// reproduces iteration for iteration.
//
// Device resident like the CG (cg.hip): all scalars -- the Hessenberg column, the rotations,
// the rotated right-hand side, the stop decision -- live in HBM; every vector kernel folds the
// per-workgroup partial sums of the launch before it in a fixed order (bit-reproducible, no
// atomics), and once the tolerance test has fired the remaining launches of the cycle return at
// once.  The host looks at the 32-byte state one cycle behind the launches.
//
// Per Krylov vector j: z = M^-1 v_j, w = A z, j + 1 projection passes
// (w -= h_i v_i fused with the next dot), one normalisation pass.  Vector traffic grows with j;
// the SpMV and the preconditioner are the kernels of the CG path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <vector>

#include "device_utils.hpp"
#include "schwz_hip.h"
#include "schwz_internal.hpp"

namespace schwz {

struct GmresState {
    double r0;      // ||b - A x_start||, the reference of the reduction test
    double resn;    // current residual norm (true at a cycle start, rotated rhs inside a cycle)
    int stop_it;    // Krylov vectors after which the solve stops (INT_MAX: not decided)
    int iters;      // Krylov vectors built so far
};

static int gm_grid(int64_t n)
{
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > kMaxGrid) g = kMaxGrid;
    return g < 1 ? 1 : (int)g;
}

// after the residual SpMV of a cycle start: beta, the stop tests of the loop top, g = beta e_1
__global__ void gmres_start_kernel(GmresState *st, const double *rr_partials, int nparts, double rtol, int it_base,
                                   int first, int max_iters, double *g, int m)
{
    __shared__ double red[4];
    if (!first && it_base >= st->stop_it) return;
    const double beta = sqrt(fold_partials(rr_partials, nparts, red));
    if (threadIdx.x == 0) {
        if (first) {
            st->r0 = beta;
            st->iters = 0;
            st->stop_it = INT_MAX;
        }
        st->resn = beta;
        if (it_base >= max_iters || beta <= rtol * st->r0 || beta == 0.0) st->stop_it = it_base;
        g[0] = beta;
        for (int i = 1; i <= m; ++i) g[i] = 0.0;
    }
}

// v_0 = r / beta (in place)
__global__ __launch_bounds__(kBlock) void gmres_scale_kernel(int64_t n, double *__restrict__ v, const GmresState *st,
                                                             int it_base)
{
    if (it_base >= st->stop_it) return;
    const double beta = st->resn;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) v[i] /= beta;
}

// One projection pass of modified Gram-Schmidt.  With `vprev`: h = fold(partials_in) is the
// coefficient against vprev (stored by workgroup 0), w -= h vprev.  Then the partial sums of
// w . vnext (vnext == nullptr: of w . w) for the next pass.
__global__ __launch_bounds__(kBlock) void gmres_project_kernel(int64_t n, double *__restrict__ w,
                                                               const double *__restrict__ vprev,
                                                               const double *__restrict__ vnext,
                                                               const double *partials_in, int nparts,
                                                               double *partials_out, double *h_out,
                                                               const GmresState *st, int it)
{
    __shared__ double red[4];
    if (it >= st->stop_it) return;
    double h = 0.0;
    if (vprev) {
        h = fold_partials(partials_in, nparts, red);
        if (blockIdx.x == 0 && threadIdx.x == 0) *h_out = h;
    }
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double wv = w[i];
        if (vprev) {
            wv -= h * vprev[i];
            w[i] = wv;
        }
        acc += wv * (vnext ? vnext[i] : wv);
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials_out[blockIdx.x] = s;
}

// v_{j+1} = w / ||w||; workgroup 0 finishes column j of the Hessenberg matrix: the old rotations,
// the new one, the rotated right-hand side and the stop test.
__global__ __launch_bounds__(kBlock) void gmres_finish_kernel(int64_t n, const double *__restrict__ w,
                                                              double *__restrict__ vnext, const double *ww_partials,
                                                              int nparts, double *H, double *cs, double *sn, double *g,
                                                              int m, int j, GmresState *st, int it, double rtol)
{
    __shared__ double red[4];
    if (it >= st->stop_it) return;
    const double hn = sqrt(fold_partials(ww_partials, nparts, red));
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        vnext[i] = hn != 0.0 ? w[i] / hn : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double *h = H + (size_t)j * (size_t)(m + 1);
        h[j + 1] = hn;
        for (int i = 0; i < j; ++i) {
            const double t = cs[i] * h[i] + sn[i] * h[i + 1];
            h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
            h[i] = t;
        }
        double c = 1.0, s = 0.0;
        if (h[j + 1] != 0.0) {
            const double rr = hypot(h[j], h[j + 1]);
            c = h[j] / rr;
            s = h[j + 1] / rr;
        }
        cs[j] = c;
        sn[j] = s;
        h[j] = c * h[j] + s * h[j + 1];
        h[j + 1] = 0.0;
        g[j + 1] = -s * g[j];
        g[j] = c * g[j];
        const double resn = fabs(g[j + 1]);
        st->resn = resn;
        st->iters = it + 1;
        // other workgroups of this launch compare `it` with stop_it concurrently: it only ever
        // drops to it + 1, which they read as "not yet"
        if (resn <= rtol * st->r0) st->stop_it = it + 1;
    }
}

// End of a cycle that built k = min(launched, stop_it - it_base) vectors: H y = g by back
// substitution (every workgroup, redundantly: k is small), t = sum_i y_i v_i.
__global__ __launch_bounds__(kBlock) void gmres_combine_kernel(int64_t n, const double *__restrict__ V, int64_t ldv,
                                                               double *__restrict__ t, const double *H,
                                                               const double *g, int m, int launched,
                                                               const GmresState *st, int it_base)
{
    extern __shared__ double y[];  // m
    if (it_base >= st->stop_it) return;
    const int k = min(launched, st->stop_it - it_base);
    if (threadIdx.x == 0) {
        for (int i = k - 1; i >= 0; --i) {
            double s = g[i];
            for (int q = i + 1; q < k; ++q) s -= H[i + (size_t)q * (size_t)(m + 1)] * y[q];
            y[i] = s / H[i + (size_t)i * (size_t)(m + 1)];
        }
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double s = 0.0;
        for (int q = 0; q < k; ++q) s += y[q] * V[(size_t)q * (size_t)ldv + (size_t)i];
        t[i] = s;
    }
}

// x += z
__global__ __launch_bounds__(kBlock) void gmres_axpy_kernel(int64_t n, double *__restrict__ x,
                                                            const double *__restrict__ z, const GmresState *st,
                                                            int it_base)
{
    if (it_base >= st->stop_it) return;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) x[i] += z[i];
}

}  // namespace schwz

using namespace schwz;

struct schwz_gmres {
    const schwz_csr *A = nullptr;
    schwz_pcg *pre = nullptr;  // owns the preconditioner (and the SpMV partial-sum banks)
    int64_t n = 0, ldv = 0;
    int m = 1;
    double *V = nullptr;  // (m + 1) vectors of ldv
    double *w = nullptr, *z = nullptr;
    double *H = nullptr, *cs = nullptr, *sn = nullptr, *g = nullptr;
    double *part = nullptr;  // 2 banks of kMaxGrid
    GmresState *state = nullptr, *h_state = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int variant = 0;
};

extern "C" {

int schwz_gmres_create(const schwz_csr *A, int precond, int block_size, int restart, schwz_gmres **out)
{
    SCHWZ_REQUIRE(A && out, "schwz_gmres_create: null argument");
    SCHWZ_REQUIRE(A->v.nrows == A->v.ncols, "schwz_gmres_create: matrix must be square");
    SCHWZ_REQUIRE(restart >= 1, "schwz_gmres_create: restart (krylov_dim) must be >= 1");
    SCHWZ_REQUIRE(restart <= 2048, "schwz_gmres_create: restart above 2048 is not supported");
    schwz_gmres *s = new schwz_gmres();
    s->A = A;
    s->n = A->v.nrows;
    s->m = restart;
    s->ldv = (s->n + 1) & ~int64_t(1);  // 16-byte aligned columns
    int rc = schwz_pcg_create_ex(A, precond, block_size, &s->pre);
    if (rc) {
        delete s;
        return rc;
    }
    const size_t ld = (size_t)(s->ldv ? s->ldv : 2), m = (size_t)restart;
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 8);
    };
    alloc((void **)&s->V, sizeof(double) * ld * (m + 1));
    alloc((void **)&s->w, sizeof(double) * ld);
    alloc((void **)&s->z, sizeof(double) * ld);
    alloc((void **)&s->H, sizeof(double) * (m + 1) * m);
    alloc((void **)&s->cs, sizeof(double) * m);
    alloc((void **)&s->sn, sizeof(double) * m);
    alloc((void **)&s->g, sizeof(double) * (m + 1));
    alloc((void **)&s->part, sizeof(double) * 2 * kMaxGrid);
    alloc((void **)&s->state, sizeof(GmresState));
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_state, 2 * sizeof(GmresState), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev[1], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMemset(s->H, 0, sizeof(double) * (m + 1) * m);
    if (e != hipSuccess) {
        set_error(std::string("schwz_gmres_create: ") + hipGetErrorString(e));
        schwz_gmres_destroy(s);
        return SCHWZ_ERR_HIP;
    }
    *out = s;
    return SCHWZ_OK;
}

void schwz_gmres_destroy(schwz_gmres *s)
{
    if (!s) return;
    void *ptrs[] = {s->V, s->w, s->z, s->H, s->cs, s->sn, s->g, s->part, s->state};
    for (void *p : ptrs) (void)hipFree(p);
    if (s->h_state) (void)hipHostFree(s->h_state);
    for (hipEvent_t ev : s->ev)
        if (ev) (void)hipEventDestroy(ev);
    schwz_pcg_destroy(s->pre);
    delete s;
}

int schwz_gmres_last_stats(schwz_gmres *s, int *h_iters, double *h_resnorm)
{
    SCHWZ_REQUIRE(s && h_iters && h_resnorm, "schwz_gmres_last_stats: null argument");
    SCHWZ_HIP_TRY(hipDeviceSynchronize());
    SCHWZ_HIP_TRY(hipMemcpy(&s->h_state[0], s->state, sizeof(GmresState), hipMemcpyDeviceToHost));
    *h_iters = s->h_state[0].iters;
    *h_resnorm = s->h_state[0].resn;
    return SCHWZ_OK;
}

int schwz_gmres_solve(schwz_gmres *s, const double *d_b, double *d_x, double rtol, int max_iters, int *h_iters,
                      double *h_resnorm, schwz_stream stream)
{
    SCHWZ_REQUIRE(s && d_b && d_x, "schwz_gmres_solve: null argument");
    SCHWZ_REQUIRE(max_iters >= 0, "schwz_gmres_solve: negative max_iters");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = s->n;
    if (n == 0) {
        if (h_iters) *h_iters = 0;
        if (h_resnorm) *h_resnorm = 0.0;
        return SCHWZ_OK;
    }
    const CsrView &A = s->A->v;
    const int m = s->m;
    const int gs = spmv_grid(A, s->variant), gv = gm_grid(n);
    double *bank[2] = {s->part, s->part + kMaxGrid};
    int pending = -1, slot = 0;
    bool stopped = false;
    for (int c = 0; !stopped; ++c) {
        const int it_base = c * m;
        // ---- r = b - A x -> v_0, beta, loop-top tests ----
        SpmvArgs a;
        a.x = d_x;
        a.b = d_b;
        a.y = s->V;
        a.p = s->w;  // the kernel also writes p := r; w is scratch here
        a.partials = s->pre->partials;
        int rc = launch_spmv(A, kSpmvResidInit, a, s->variant, st);
        if (rc) return rc;
        hipLaunchKernelGGL(gmres_start_kernel, dim3(1), dim3(kBlock), 0, st, s->state, s->pre->partials + gs, gs, rtol,
                           it_base, c == 0 ? 1 : 0, max_iters, s->g, m);
        hipLaunchKernelGGL(gmres_scale_kernel, dim3(gv), dim3(kBlock), 0, st, n, s->V, s->state, it_base);
        const int launched = std::min(m, max_iters - it_base);
        for (int j = 0; j < launched; ++j) {
            const int it = it_base + j;
            double *vj = s->V + (size_t)j * (size_t)s->ldv;
            if ((rc = precond_apply(s->pre, vj, s->z, st))) return rc;
            SpmvArgs q;
            q.x = s->z;
            q.y = s->w;
            q.partials = s->pre->partials;  // z . w, unused
            q.stop_iter = &s->state->stop_it;
            q.it = it;
            if ((rc = launch_spmv(A, kSpmvDot, q, s->variant, st))) return rc;
            double *hcol = s->H + (size_t)j * (size_t)(m + 1);
            for (int i = 0; i <= j + 1; ++i) {
                // pass i: subtract the component along v_{i-1}, start the dot with v_i (or w.w)
                const double *vprev = i > 0 ? s->V + (size_t)(i - 1) * (size_t)s->ldv : nullptr;
                const double *vnext = i <= j ? s->V + (size_t)i * (size_t)s->ldv : nullptr;
                hipLaunchKernelGGL(gmres_project_kernel, dim3(gv), dim3(kBlock), 0, st, n, s->w, vprev, vnext,
                                   bank[(i + 1) & 1], gv, bank[i & 1], i > 0 ? hcol + (i - 1) : nullptr, s->state, it);
            }
            hipLaunchKernelGGL(gmres_finish_kernel, dim3(gv), dim3(kBlock), 0, st, n, s->w,
                               s->V + (size_t)(j + 1) * (size_t)s->ldv, bank[(j + 1) & 1], gv, s->H, s->cs, s->sn, s->g,
                               m, j, s->state, it, rtol);
        }
        // ---- x += M^-1 (V y) ----
        hipLaunchKernelGGL(gmres_combine_kernel, dim3(gv), dim3(kBlock), sizeof(double) * (size_t)m, st, n, s->V,
                           s->ldv, s->w, s->H, s->g, m, launched, s->state, it_base);
        if ((rc = precond_apply(s->pre, s->w, s->z, st))) return rc;
        hipLaunchKernelGGL(gmres_axpy_kernel, dim3(gv), dim3(kBlock), 0, st, n, d_x, s->z, s->state, it_base);
        SCHWZ_HIP_TRY(hipGetLastError());
        if (it_base + launched >= max_iters) break;
        // the host reads the state one cycle behind the launches
        if (pending >= 0) {
            SCHWZ_HIP_TRY(hipEventSynchronize(s->ev[pending]));
            if (s->h_state[pending].stop_it != INT_MAX) stopped = true;
        }
        SCHWZ_HIP_TRY(hipMemcpyAsync(&s->h_state[slot], s->state, sizeof(GmresState), hipMemcpyDeviceToHost, st));
        SCHWZ_HIP_TRY(hipEventRecord(s->ev[slot], st));
        pending = slot;
        slot ^= 1;
    }
    if (h_iters || h_resnorm) {
        SCHWZ_HIP_TRY(hipMemcpyAsync(&s->h_state[0], s->state, sizeof(GmresState), hipMemcpyDeviceToHost, st));
        SCHWZ_HIP_TRY(hipStreamSynchronize(st));
        if (h_iters) *h_iters = s->h_state[0].iters;
        if (h_resnorm) *h_resnorm = s->h_state[0].resn;
    }
    return SCHWZ_OK;
}

}  // extern "C"
