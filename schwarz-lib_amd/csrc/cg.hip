// Device-resident preconditioned CG: the fused vector kernels, the preconditioner applications
// (Jacobi, block-Jacobi, ILU(0) sweeps, ISAI products), the stored-q and the q-free iteration,
// hipGraph replay for small systems, and the profiling hooks of bench.py's roofline leg.
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
// CG vector kernels.  One CG iteration = spmv_tiled_kernel<kSpmvDot> + these two.
// Scalars live in CgState in HBM; nothing returns to the host inside the loop.
// ---------------------------------------------------------------------------

__global__ void cg_init_finalize_kernel(CgState *st, const double *partials, int nparts, double rtol,
                                        double *norm_sq_out, int norm_bank)
{
    __shared__ double red[4];
    const double rho = fold_partials(partials, nparts, red);
    const double rr = fold_partials(partials + nparts, nparts, red);
    if (norm_sq_out) {
        const double n2 = fold_partials(partials + norm_bank * nparts, nparts, red);
        if (threadIdx.x == 0) {
            norm_sq_out[0] = n2;  // may be mapped host memory
            __threadfence_system();
        }
    }
    if (threadIdx.x == 0) {
        st->rho[0] = rho;
        st->rho[1] = 0.0;
        st->rr = rr;
        st->r0 = sqrt(rr);
        st->iters = 0;
        // loop-top test of iteration 0: ||r|| <= rtol*||r_initial||
        st->stop_iter = (sqrt(rr) <= rtol * sqrt(rr)) ? 0 : INT_MAX;
    }
}

typedef double vd2 __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ void store2(double *base, int64_t i, vd2 v)
{
    if (NT)
        __builtin_nontemporal_store(v, reinterpret_cast<vd2 *>(base) + i);
    else
        reinterpret_cast<vd2 *>(base)[i] = v;
}

// 1/diag of rows 2i and 2i+1 in whichever representation the solver holds
__device__ __forceinline__ vd2 diag_pair(const DiagView &dg, const vd2 *full2, const uint16_t *code2,
                                         const double *ddict, int64_t i)
{
    vd2 d;
    if (dg.mode == 1) {
        d = full2[i];
    } else if (dg.mode == 2) {
        const unsigned c = code2[i];
        d.x = ddict[c & 255u];
        d.y = ddict[c >> 8];
    } else {
        d.x = d.y = dg.uniform;
    }
    return d;
}

__device__ __forceinline__ double diag_one(const DiagView &dg, const double *ddict, int64_t i)
{
    if (dg.mode == 1) return dg.full[i];
    if (dg.mode == 2) return ddict[dg.code[i]];
    return dg.uniform;
}

// x += alpha p ; r -= alpha q ; z = dinv r ; partials: r.z and r.r
// U: 16-byte elements per lane in flight per trip; NT: non-temporal stores; NOX: x and p are not touched
// (deferred x update of the stored-q iteration: alpha goes to *alpha_out instead, see cg_flush_x_kernel)
template <int U, bool NT, bool NOX = false>
__global__ __launch_bounds__(kBlock) void cg_update_kernel(int64_t n, double *__restrict__ x,
                                                           double *__restrict__ r,
                                                           const double *__restrict__ p,
                                                           const double *__restrict__ q,
                                                           const DiagView dg,
                                                           const double *pq_partials, int nparts_in,
                                                           const CgState *st, int it,
                                                           double *partials_out, double *alpha_out = nullptr)
{
    __shared__ double red[4];
    __shared__ double ddict[256];
    if (it >= st->stop_iter) return;
    if (dg.mode == 2) {
        if (threadIdx.x < dg.ndict) ddict[threadIdx.x] = dg.dict[threadIdx.x];
        __syncthreads();
    }
    const double *__restrict__ dinv = dg.mode == 1 ? dg.full : nullptr;
    const uint16_t *dc2 = reinterpret_cast<const uint16_t *>(dg.code);
    const double pq = fold_partials(pq_partials, nparts_in, red);
    const double alpha = st->rho[it & 1] / pq;
    if (NOX && alpha_out && blockIdx.x == 0 && threadIdx.x == 0) *alpha_out = alpha;
    double a0 = 0.0, a1 = 0.0;
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const vd2 *x2 = reinterpret_cast<const vd2 *>(x);
    const vd2 *r2 = reinterpret_cast<const vd2 *>(r);
    const vd2 *p2 = reinterpret_cast<const vd2 *>(p);
    const vd2 *q2 = reinterpret_cast<const vd2 *>(q);
    const vd2 *d2 = reinterpret_cast<const vd2 *>(dinv);
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n2; i0 += stride * U) {
        vd2 xv[U], rv[U], pv[U], qv[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n2) {
                if (!NOX) {
                    xv[u] = x2[i];
                    pv[u] = p2[i];
                }
                rv[u] = r2[i];
                qv[u] = q2[i];
                dv[u] = diag_pair(dg, d2, dc2, ddict, i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n2) {
                if (!NOX) {
                    // one fused multiply-add per element, spelled out: cg_flush_x_kernel (fused = 1) repeats it
                    xv[u].x = __builtin_fma(alpha, pv[u].x, xv[u].x);
                    xv[u].y = __builtin_fma(alpha, pv[u].y, xv[u].y);
                    store2<NT>(x, i, xv[u]);
                }
                rv[u] -= alpha * qv[u];
                store2<NT>(r, i, rv[u]);
                vd2 z = rv[u];
                if (dg.mode) z *= dv[u];
                a0 += rv[u].x * z.x;
                a0 += rv[u].y * z.y;
                a1 += rv[u].x * rv[u].x;
                a1 += rv[u].y * rv[u].y;
            }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        if (!NOX) x[i] = __builtin_fma(alpha, p[i], x[i]);
        const double rv = r[i] - alpha * q[i];
        r[i] = rv;
        const double z = dg.mode ? diag_one(dg, ddict, i) * rv : rv;
        a0 += rv * z;
        a1 += rv * rv;
    }
    const double s0 = block_sum(a0, red);
    const double s1 = block_sum(a1, red);
    if (threadIdx.x == 0) {
        partials_out[blockIdx.x] = s0;
        partials_out[gridDim.x + blockIdx.x] = s1;
    }
}

// beta = rho'/rho ; p = dinv r + beta p ; state update by workgroup 0
template <int U, bool NT>
__global__ __launch_bounds__(kBlock) void cg_direction_kernel(int64_t n, double *p, const double *__restrict__ r,
                                                              const DiagView dg,
                                                              const double *partials_in, int nparts,
                                                              CgState *st, int it, double rtol, double *p_out = nullptr)
{
    // p_out: the new direction goes to another vector (the ring of the deferred x update); each
    // element is read and written by the same lane, so p_out == p (in place) is the default
    if (!p_out) p_out = p;
    // with a general preconditioner `r` is already z = M^-1 r and dg.mode is 0
    __shared__ double red[4];
    __shared__ double ddict[256];
    if (it >= st->stop_iter) return;
    if (dg.mode == 2) {
        if (threadIdx.x < dg.ndict) ddict[threadIdx.x] = dg.dict[threadIdx.x];
        __syncthreads();
    }
    const double *__restrict__ dinv = dg.mode == 1 ? dg.full : nullptr;
    const uint16_t *dc2 = reinterpret_cast<const uint16_t *>(dg.code);
    const double rho_new = fold_partials(partials_in, nparts, red);
    const double rr = fold_partials(partials_in + nparts, nparts, red);
    const double beta = rho_new / st->rho[it & 1];
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const vd2 *p2 = reinterpret_cast<const vd2 *>(p);
    const vd2 *r2 = reinterpret_cast<const vd2 *>(r);
    const vd2 *d2 = reinterpret_cast<const vd2 *>(dinv);
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n2; i0 += stride * U) {
        vd2 pv[U], zv[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n2) {
                pv[u] = p2[i];
                zv[u] = r2[i];
                dv[u] = diag_pair(dg, d2, dc2, ddict, i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n2) {
                if (dg.mode) zv[u] *= dv[u];
                store2<NT>(p_out, i, zv[u] + beta * pv[u]);
            }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double z = dg.mode ? diag_one(dg, ddict, i) * r[i] : r[i];
        p_out[i] = z + beta * p[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // other workgroups read rho[it&1] concurrently: the slot written here is the other one
        st->rho[(it + 1) & 1] = rho_new;
        st->rr = rr;
        st->iters = st->iters + 1;
        // 0, not it + 1: every later launch leaves at once whatever iteration index it carries (the
        // recorded launches of a replayed hipGraph carry 0..15 again and again).  Workgroups of THIS
        // launch that read the 0 early skip their part of p, which nobody will read any more.
        if (sqrt(rr) <= rtol * st->r0) st->stop_iter = 0;
    }
}

// What workgroup 0 of cg_direction_kernel does, alone: the LAST iteration of a solve needs the state
// (rho, ||r||^2, the iteration count the deferred x update goes by) but no new search direction --
// 24 n bytes and 0.067 ms at 256^3 that nobody reads.
__global__ __launch_bounds__(kBlock) void cg_state_advance_kernel(const double *partials_in, int nparts, CgState *st,
                                                                   int it, double rtol)
{
    __shared__ double red[4];
    if (it >= st->stop_iter) return;
    const double rho_new = fold_partials(partials_in, nparts, red);
    const double rr = fold_partials(partials_in + nparts, nparts, red);
    if (threadIdx.x == 0) {
        st->rho[(it + 1) & 1] = rho_new;
        st->rr = rr;
        st->iters = st->iters + 1;
        if (sqrt(rr) <= rtol * st->r0) st->stop_iter = 0;
    }
}

// The last direction of a fixed-work solve is never stored (pcg_iterate, vlast_on): when somebody asks for the
// statistics of the postponed last iteration after all, its update launch needs that direction in memory --
// p = fma(beta, p_prev, D^-1 r), what the fused direction launch computed (the same operations: the same bits).
__global__ __launch_bounds__(kBlock) void cg_rebuild_direction_kernel(int64_t n, const double *__restrict__ p_prev,
                                                                      double prev_scale, const double *__restrict__ r,
                                                                      int dmode, double dsc, const CgState *st, int it,
                                                                      double *__restrict__ p_out)
{
#pragma clang fp contract(off)
    const double beta = st->rho[it & 1] / st->rho[(it - 1) & 1];
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double pp = prev_scale * p_prev[i];
        const double zv = dmode ? dsc * r[i] : r[i];
        p_out[i] = __builtin_fma(beta, pp, zv);
    }
}

// Measured on MI355X (256^3): U = 2/4 and non-temporal stores change the PCG iteration time by
// < 1 % (0.403 / 0.406 / 0.417 ms for U = 1 / 2 / 4): these kernels sit at the mixed
// read+write HBM ceiling (~5.0-5.5 TB/s), so the plain shape is used.


// Deferred x update: x_i += alpha_k p_k,i for the iterations b0 <= k < b0 + count that were actually
// carried out, in iteration order and with the product rounded before the sum -- the very
// operations kSpmvCgUpdate performs when it updates x itself, so the bits are the same.
// Carried out: k < st->iters (the direction launch counts), plus iteration `pending` when this
// launch sits between its update launch and its direction launch and the update launch ran
// (pending < stop_iter, which only direction launches lower).
struct PRing {
    const double *slot[kDeferDepth];
};

// The launch covers the pairs [pair0, pair1) (and the odd last row when `tail` is set): the last update of a
// solve is cut into the caller's priority rows and the rest (schwz_pcg::prio_*).
// lazy_it >= 0: the update launch of iteration lazy_it was not run (schwz_pcg::LazyLast); its direction counts
// as carried out and its alpha = rho / (p.Ap) is formed here from the partial sums that launch would have folded
// (the same expression, the same bits).  x2: rows [0, x2_rows) of the result are stored there as well (the
// restricted write-back of the RAS step) and the rows [x2_rows, x2_total) are copied from x2_src -- the overlap and
// halo entries of the current x~ --, so that x2 is the complete x~ after the restriction; with nothing to add the
// launch only copies.
__global__ __launch_bounds__(kBlock) void cg_flush_x_kernel(int64_t n, double *__restrict__ x, const PRing ring,
                                                            const double *__restrict__ alpha_hist,
                                                            const CgState *st, int b0, int count, int pending,
                                                            int64_t pair0, int64_t pair1, int tail, int fused,
                                                            int lazy_it, const double *pq_partials, int pq_nparts,
                                                            double *__restrict__ x2, int64_t x2_rows,
                                                            const double *__restrict__ x2_src, int64_t x2_total,
                                                            double p0_scale, int vlast_it,
                                                            const double *__restrict__ vlast_r, int vlast_dmode,
                                                            double vlast_dsc)
{
#pragma clang fp contract(off)
    // p0_scale: direction 0 of the solve is p0_scale x (slot 0 of the ring) -- 1.0 for a stored p0 (the product is
    // exact), D^-1's uniform factor when slot 0 is r0 and p0 = D^-1 r0 was never stored (virtual first direction:
    // the same rounded product the first-direction launch formed)
    // vlast_it >= 0: the direction of iteration vlast_it (the last one of the solve) was never stored either: it is
    // p = fma(beta, p_prev, D^-1 r) with beta = rho_new / rho_old as the fused direction launch formed it (both rho
    // slots still hold those values: the update launch of that iteration is the postponed one), p_prev the
    // direction before it and r = vlast_r the current residual -- the same operations, the same bits.
    __shared__ double alpha[kDeferDepth];
    __shared__ double red[4];
    const int stop = st->stop_iter;
    int done = st->iters + ((pending >= 0 && pending < stop) ? 1 : 0);
    if (lazy_it >= 0 && lazy_it < stop && done <= lazy_it) done = lazy_it + 1;
    const int kmax = min(count, done - b0);
    if (kmax <= 0 && !x2) return;
    double lazy_alpha = 0.0;
    if (kmax > 0 && lazy_it >= b0 && lazy_it < b0 + kmax)  // workgroup-uniform
        lazy_alpha = st->rho[lazy_it & 1] / fold_partials(pq_partials, pq_nparts, red);
    if ((int)threadIdx.x < kmax)
        alpha[threadIdx.x] = (b0 + (int)threadIdx.x == lazy_it) ? lazy_alpha : alpha_hist[(b0 + threadIdx.x) % kDeferDepth];
    __syncthreads();
    const int64_t n2 = pair1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    vd2 *x2v = reinterpret_cast<vd2 *>(x);
    const bool vlast = vlast_it >= b0 && vlast_it < b0 + kmax;  // workgroup-uniform
    const double vbeta = vlast ? st->rho[vlast_it & 1] / st->rho[(vlast_it - 1) & 1] : 0.0;
    const double *const vprev_slot = ring.slot[(vlast_it + kDeferDepth - 1) % kDeferDepth];
    const double vprev_scale = vlast_it == 1 ? p0_scale : 1.0;
    for (int64_t i = pair0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
        vd2 xv = __builtin_nontemporal_load(x2v + i);
        vd2 carry = {0.0, 0.0};  // direction of the iteration before the current group of four
        for (int k0 = 0; k0 < kmax; k0 += 4) {
            vd2 pv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k0 + k < kmax) {
                    if (vlast && b0 + k0 + k == vlast_it) {
                        pv[k] = __builtin_nontemporal_load(reinterpret_cast<const vd2 *>(vlast_r) + i);
                    } else {
                        pv[k] = __builtin_nontemporal_load(reinterpret_cast<const vd2 *>(ring.slot[(b0 + k0 + k) % kDeferDepth]) + i);
                        if (b0 + k0 + k == 0) pv[k] = p0_scale * pv[k];
                    }
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k0 + k < kmax) {
                    if (vlast && b0 + k0 + k == vlast_it) {
                        vd2 pp;
                        if (k > 0)
                            pp = pv[k - 1];
                        else if (k0 > 0)
                            pp = carry;
                        else
                            pp = vprev_scale * __builtin_nontemporal_load(reinterpret_cast<const vd2 *>(vprev_slot) + i);
                        const vd2 dsc2 = {vlast_dsc, vlast_dsc};
                        const vd2 zv = vlast_dmode ? dsc2 * pv[k] : pv[k];
                        pv[k].x = __builtin_fma(vbeta, pp.x, zv.x);
                        pv[k].y = __builtin_fma(vbeta, pp.y, zv.y);
                    }
                    if (fused) {  // the stored-q iteration's update: x = fma(alpha, p, x)
                        xv.x = __builtin_fma(alpha[k0 + k], pv[k].x, xv.x);
                        xv.y = __builtin_fma(alpha[k0 + k], pv[k].y, xv.y);
                    } else {      // the q-free update launch: product rounded, then added
                        const vd2 inc = alpha[k0 + k] * pv[k];
                        xv = xv + inc;
                    }
                }
            carry = pv[3];
        }
        if (kmax > 0) __builtin_nontemporal_store(xv, x2v + i);
        if (x2) {
            if (2 * i + 1 < x2_rows) {
                __builtin_nontemporal_store(xv, reinterpret_cast<vd2 *>(x2) + i);
            } else {
                if (2 * i < x2_rows)
                    x2[2 * i] = xv.x;
                else if (2 * i < x2_total)
                    x2[2 * i] = x2_src[2 * i];
                if (2 * i + 1 < x2_total) x2[2 * i + 1] = x2_src[2 * i + 1];
            }
        }
    }
    // entries of x2 beyond the solve's vector (the halo of x~): copied by the launch that takes the tail
    if (x2 && tail)
        for (int64_t j = 2 * pair1 + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < x2_total; j += stride)
            if (!((n & 1) && j == n - 1)) x2[j] = x2_src[j];
    if (tail && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double xv = x[n - 1];
        double pprev = vlast ? vprev_scale * vprev_slot[n - 1] : 0.0;
        for (int k = 0; k < kmax; ++k) {
            double pk;
            if (vlast && b0 + k == vlast_it) {
                const double rv = vlast_r[n - 1];
                const double zv = vlast_dmode ? vlast_dsc * rv : rv;
                pk = __builtin_fma(vbeta, pprev, zv);
            } else {
                pk = ring.slot[(b0 + k) % kDeferDepth][n - 1];
                if (b0 + k == 0) pk = p0_scale * pk;
            }
            pprev = pk;
            if (fused) {
                xv = __builtin_fma(alpha[k], pk, xv);
            } else {
                const double inc = alpha[k] * pk;
                xv = xv + inc;
            }
        }
        if (kmax > 0) x[n - 1] = xv;
        if (x2) x2[n - 1] = n - 1 < x2_rows ? xv : x2_src[n - 1];
    }
}

// dinv[i] = 1 / A[i][i] (1 when the row stores no diagonal): scalar Jacobi, i.e.
// block-Jacobi with max_block_size 1
__global__ void extract_dinv_kernel(CsrView A, double *__restrict__ dinv)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.nrows; i += stride) {
        double d = 1.0;
        for (int j = A.rp[i]; j < A.rp[i + 1]; ++j)
            if (A.col[j] == i) d = A.val[j];
        dinv[i] = 1.0 / d;
    }
}

// ---- block-Jacobi apply and the vector pieces of the general preconditioned CG -------------

// z[i] = sum_j inv[blk_id[i / bs]][i % bs][j] * r[(i / bs) * bs + j]
__global__ __launch_bounds__(kBlock) void block_jacobi_apply_kernel(int64_t n, int bs,
                                                                    const schwz_idx *__restrict__ blk_id,
                                                                    const double *__restrict__ blk_inv,
                                                                    const double *__restrict__ r,
                                                                    double *__restrict__ z)
{
#pragma clang fp contract(off)
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int64_t b = i / bs, r0 = b * bs;
        const double *row = blk_inv + ((int64_t)blk_id[b] * bs + (i - r0)) * bs;
        double s = 0.0;
        for (int j = 0; j < bs && r0 + j < n; ++j) s += row[j] * r[r0 + j];
        z[i] = s;
    }
}

// the same for detected blocks of different sizes (supervariable agglomeration): block of the row, its
// first row and size from blk_ptr
__global__ __launch_bounds__(kBlock) void block_jacobi_var_apply_kernel(int64_t n, int bs,
                                                                        const schwz_idx *__restrict__ row_blk,
                                                                        const schwz_idx *__restrict__ blk_ptr,
                                                                        const schwz_idx *__restrict__ blk_id,
                                                                        const double *__restrict__ blk_inv,
                                                                        const double *__restrict__ r,
                                                                        double *__restrict__ z)
{
#pragma clang fp contract(off)
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int b = row_blk[i];
        const int64_t r0 = blk_ptr[b];
        const int k = blk_ptr[b + 1] - blk_ptr[b];
        const double *row = blk_inv + ((int64_t)blk_id[b] * bs + (i - r0)) * bs;
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += row[j] * r[r0 + j];
        z[i] = s;
    }
}

// partial sums of r.z and r.r (banks 0 and 1), optionally p := z
__global__ __launch_bounds__(kBlock) void dot_rz_kernel(int64_t n, const double *__restrict__ r,
                                                        const double *__restrict__ z, double *__restrict__ p_out,
                                                        const CgState *st, int it, double *partials_out)
{
    __shared__ double red[4];
    if (st && it >= st->stop_iter) return;
    double a0 = 0.0, a1 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double rv = r[i], zv = z[i];
        a0 += rv * zv;
        a1 += rv * rv;
        if (p_out) p_out[i] = zv;
    }
    const double s0 = block_sum(a0, red);
    const double s1 = block_sum(a1, red);
    if (threadIdx.x == 0) {
        partials_out[blockIdx.x] = s0;
        partials_out[gridDim.x + blockIdx.x] = s1;
    }
}

}  // namespace schwz

using namespace schwz;

// ---- profiling hooks (bench.py roofline leg) ------------------------------------
// HIP-event pairs around the SpMV and the update launch of every CG iteration, on the stream
// they are launched on.

namespace {
struct ProfState {
    bool on = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> kind;  // per event pair: 0 = the SpMV launch of a CG iteration, 1 = its update launch
    size_t used = 0;
    double total[2] = {0.0, 0.0};
    int64_t count[2] = {0, 0};
} g_prof;
}  // namespace

extern "C" {

int schwz_profile_begin(int capacity)
{
    SCHWZ_REQUIRE(capacity > 0, "schwz_profile_begin: capacity must be positive");
    while (g_prof.ev.size() < (size_t)2 * capacity) {
        hipEvent_t e;
        SCHWZ_HIP_TRY(hipEventCreate(&e));
        g_prof.ev.push_back(e);
    }
    g_prof.kind.assign((size_t)capacity, 0);
    g_prof.used = 0;
    g_prof.on = true;
    return SCHWZ_OK;
}

int schwz_profile_end(double *h_total_ms, int64_t *h_launches)
{
    SCHWZ_REQUIRE(h_total_ms && h_launches, "schwz_profile_end: null output");
    g_prof.on = false;
    SCHWZ_HIP_TRY(hipDeviceSynchronize());
    g_prof.total[0] = g_prof.total[1] = 0.0;
    g_prof.count[0] = g_prof.count[1] = 0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        float ms = 0.f;
        SCHWZ_HIP_TRY(hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]));
        const int k = g_prof.kind[i / 2] ? 1 : 0;
        g_prof.total[k] += ms;
        g_prof.count[k] += 1;
    }
    *h_total_ms = g_prof.total[0];
    *h_launches = g_prof.count[0];
    g_prof.used = 0;
    return SCHWZ_OK;
}

int schwz_profile_kind(int kind, double *h_total_ms, int64_t *h_launches)
{
    SCHWZ_REQUIRE(h_total_ms && h_launches && (kind == 0 || kind == 1), "schwz_profile_kind: bad arguments");
    *h_total_ms = g_prof.total[kind];
    *h_launches = g_prof.count[kind];
    return SCHWZ_OK;
}

// ---- PCG --------------------------------------------------------------------

int schwz_pcg_flavour(const schwz_pcg *s) { return s ? s->last_flavour : 0; }

int schwz_pcg_create(const schwz_csr *A, int precond, schwz_pcg **out)
{
    return schwz_pcg_create_ex(A, precond, 1, out);
}

// block-Jacobi / ILU(0) setup: the (setup-time) host copy of the matrix comes back from HBM
static int pcg_setup_general(schwz_pcg *s)
{
    const CsrView &A = s->A->v;
    const int64_t n = s->n;
    std::vector<schwz_idx> rp((size_t)n + 1), col((size_t)A.nnz);
    std::vector<double> val((size_t)A.nnz);
    SCHWZ_HIP_TRY(hipMemcpy(rp.data(), A.rp, rp.size() * sizeof(schwz_idx), hipMemcpyDeviceToHost));
    if (A.nnz) {
        SCHWZ_HIP_TRY(hipMemcpy(col.data(), A.col, col.size() * sizeof(schwz_idx), hipMemcpyDeviceToHost));
        SCHWZ_HIP_TRY(hipMemcpy(val.data(), A.val, val.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    SCHWZ_HIP_TRY(hipMalloc((void **)&s->z, (size_t)(n ? n : 1) * sizeof(double)));
    if (s->precond == SCHWZ_PRECOND_ILU || s->precond == SCHWZ_PRECOND_ISAI) {
        schwz_idx *l_rp, *l_col, *u_rp, *u_col;
        double *l_val, *u_val;
        int rc = schwz_ilu0(n, rp.data(), col.data(), val.data(), &l_rp, &l_col, &l_val, &u_rp, &u_col, &u_val);
        if (rc) return rc;
        if (s->precond == SCHWZ_PRECOND_ISAI) {
            // Ilu<LowerIsai, UpperIsai> (solve.cpp:616-638): z = W_U (W_L r), two CSR products on
            // the patterns of L and U, stored like any other matrix of this library
            double *wl = nullptr, *wu = nullptr;
            rc = schwz_isai(n, l_rp, l_col, l_val, 1, &wl);
            if (!rc) rc = schwz_isai(n, u_rp, u_col, u_val, 0, &wu);
            if (!rc) rc = schwz_csr_create(n, n, l_rp, l_col, wl, &s->isai_l);
            if (!rc) rc = schwz_csr_create(n, n, u_rp, u_col, wu, &s->isai_u);
            if (!rc && hipMalloc((void **)&s->isai_tmp, (size_t)(n ? n : 1) * sizeof(double)) != hipSuccess) {
                set_error("schwz_pcg_create: out of device memory (ISAI work vector)");
                rc = SCHWZ_ERR_HIP;
            }
            schwz_free(wl);
            schwz_free(wu);
        } else {
            rc = schwz_trs_create(n, l_rp, l_col, l_val, u_rp, u_col, u_val, nullptr, &s->ilu);
        }
        schwz_free(l_rp);
        schwz_free(l_col);
        schwz_free(l_val);
        schwz_free(u_rp);
        schwz_free(u_col);
        schwz_free(u_val);
        return rc;
    }
    // block-Jacobi.  Blocks as gko::preconditioner::Jacobi finds them when only max_block_size is given
    // (solve.cpp:490-505; Ginkgo's published find_blocks = find_natural_blocks + agglomerate_supervariables,
    // restated in oracle/schwz_oracle.c::detect_jacobi_blocks): maximal runs of consecutive rows with the
    // same column pattern, cut at bs rows, then merged left to right while a merged block stays within bs
    // rows.  Stencil matrices get consecutive blocks of exactly bs rows.  Every block is inverted by
    // Gauss-Jordan with partial pivoting; identical inverse blocks are stored once.
    const int bs = s->block_size;
    std::vector<schwz_idx> bptr;
    bptr.push_back(0);
    if (n > 0) {
        std::vector<schwz_idx> nat;
        nat.push_back(0);
        schwz_idx cur = 1;
        for (int64_t i = 0; i + 1 < n; ++i) {
            const schwz_idx la = rp[(size_t)i + 1] - rp[(size_t)i], lb = rp[(size_t)i + 2] - rp[(size_t)i + 1];
            bool same = la == lb;
            for (schwz_idx k = 0; same && k < la; ++k) same = col[(size_t)(rp[(size_t)i] + k)] == col[(size_t)(rp[(size_t)i + 1] + k)];
            if (cur < bs && same) {
                ++cur;
            } else {
                nat.push_back(nat.back() + cur);
                cur = 1;
            }
        }
        nat.push_back(nat.back() + cur);
        cur = nat[1] - nat[0];
        for (size_t i = 1; i + 1 < nat.size(); ++i) {
            const schwz_idx size = nat[i + 1] - nat[i];
            if (cur + size <= bs) {
                cur += size;
            } else {
                bptr.push_back(bptr.back() + cur);
                cur = size;
            }
        }
        bptr.push_back(bptr.back() + cur);
    }
    const int64_t nb = (int64_t)bptr.size() - 1;
    bool uniform = true;
    for (int64_t b = 0; b < nb && uniform; ++b) uniform = bptr[(size_t)b] == b * bs;
    std::vector<schwz_idx> id((size_t)nb), row_blk;
    if (!uniform) row_blk.resize((size_t)n);
    std::vector<double> uniq, blk((size_t)bs * bs), inv((size_t)bs * bs), padded((size_t)bs * bs);
    std::unordered_multimap<uint64_t, schwz_idx> seen;
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t r0 = bptr[(size_t)b];
        const int k = (int)(bptr[(size_t)b + 1] - r0);
        std::fill(blk.begin(), blk.end(), 0.0);
        for (int i = 0; i < k; ++i) {
            if (!uniform) row_blk[(size_t)(r0 + i)] = (schwz_idx)b;
            for (schwz_idx j = rp[(size_t)(r0 + i)]; j < rp[(size_t)(r0 + i) + 1]; ++j)
                if (col[(size_t)j] >= r0 && col[(size_t)j] < r0 + k)
                    blk[(size_t)i * k + (col[(size_t)j] - r0)] = val[(size_t)j];
        }
        // invert the k x k block (the copy in blk is destroyed)
        std::vector<double> &a = blk;
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) inv[(size_t)i * k + j] = i == j ? 1.0 : 0.0;
        for (int c = 0; c < k; ++c) {
            int piv = c;
            for (int r = c + 1; r < k; ++r)
                if (std::fabs(a[(size_t)r * k + c]) > std::fabs(a[(size_t)piv * k + c])) piv = r;
            if (a[(size_t)piv * k + c] == 0.0) {
                set_error("block-Jacobi: singular diagonal block");
                return SCHWZ_ERR_NOT_SPD;
            }
            if (piv != c)
                for (int j = 0; j < k; ++j) {
                    std::swap(a[(size_t)c * k + j], a[(size_t)piv * k + j]);
                    std::swap(inv[(size_t)c * k + j], inv[(size_t)piv * k + j]);
                }
            const double d = a[(size_t)c * k + c];
            for (int j = 0; j < k; ++j) {
                a[(size_t)c * k + j] /= d;
                inv[(size_t)c * k + j] /= d;
            }
            for (int r = 0; r < k; ++r) {
                if (r == c) continue;
                const double f = a[(size_t)r * k + c];
                if (f == 0.0) continue;
                for (int j = 0; j < k; ++j) {
                    a[(size_t)r * k + j] -= f * a[(size_t)c * k + j];
                    inv[(size_t)r * k + j] -= f * inv[(size_t)c * k + j];
                }
            }
        }
        // stored in the top left corner of a bs x bs slot
        std::fill(padded.begin(), padded.end(), 0.0);
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) padded[(size_t)i * bs + j] = inv[(size_t)i * k + j];
        uint64_t h = 1469598103934665603ull ^ (uint64_t)k;
        for (double v : padded) {
            uint64_t bits;
            std::memcpy(&bits, &v, 8);
            h = (h ^ bits) * 1099511628211ull;
        }
        schwz_idx found = -1;
        auto range = seen.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (std::memcmp(&uniq[(size_t)it->second * bs * bs], padded.data(), sizeof(double) * bs * bs) == 0) {
                found = it->second;
                break;
            }
        if (found < 0) {
            found = (schwz_idx)(uniq.size() / ((size_t)bs * bs));
            uniq.insert(uniq.end(), padded.begin(), padded.end());
            seen.emplace(h, found);
        }
        id[(size_t)b] = found;
    }
    int rc;
    void *d;
    if ((rc = upload(id.data(), id.size(), &d))) return rc;
    s->d_blk_id = (schwz_idx *)d;
    if ((rc = upload(uniq.data(), uniq.size(), &d))) return rc;
    s->d_blk_inv = (double *)d;
    if (!uniform) {
        if ((rc = upload(row_blk.data(), row_blk.size(), &d))) return rc;
        s->d_row_blk = (schwz_idx *)d;
        if ((rc = upload(bptr.data(), bptr.size(), &d))) return rc;
        s->d_blk_ptr = (schwz_idx *)d;
    }
    return SCHWZ_OK;
}

// everything of schwz_pcg_create_ex that can fail half way: the caller destroys `s` on error
static int pcg_build(schwz_pcg *s, const schwz_csr *A, int precond)
{
    const size_t nb = (size_t)(s->n ? s->n : 1) * sizeof(double);
    SCHWZ_HIP_TRY(hipMalloc((void **)&s->r, nb));
    SCHWZ_HIP_TRY(hipMalloc((void **)&s->p, nb));
    SCHWZ_HIP_TRY(hipMalloc((void **)&s->q, nb));
    SCHWZ_HIP_TRY(hipMalloc((void **)&s->partials, sizeof(double) * 5 * kMaxGrid));
    SCHWZ_HIP_TRY(hipMalloc((void **)&s->d_norm_sq, sizeof(double) * 2));
    SCHWZ_HIP_TRY(hipMalloc((void **)&s->state, sizeof(CgState)));
    SCHWZ_HIP_TRY(hipHostMalloc((void **)&s->h_state, 2 * sizeof(CgState), hipHostMallocDefault));
    SCHWZ_HIP_TRY(hipEventCreateWithFlags(&s->ev[0], hipEventDisableTiming));
    SCHWZ_HIP_TRY(hipEventCreateWithFlags(&s->ev[1], hipEventDisableTiming));
    if (precond == SCHWZ_PRECOND_JACOBI) {
        SCHWZ_HIP_TRY(hipMalloc((void **)&s->dinv, nb));
        if (s->n) {
            hipLaunchKernelGGL(extract_dinv_kernel, dim3(grid_for(s->n)), dim3(kBlock), 0, 0, A->v, s->dinv);
            SCHWZ_HIP_TRY(hipGetLastError());
            SCHWZ_HIP_TRY(hipDeviceSynchronize());
        }
        // compact representation for the per-iteration vector kernels (DiagView)
        s->diag.mode = 1;
        s->diag.full = s->dinv;
        const char *env = std::getenv("SCHWZ_DIAG_DICT");
        if (s->n && !(env && env[0] == '0')) {
            std::vector<double> h((size_t)s->n);
            SCHWZ_HIP_TRY(hipMemcpy(h.data(), s->dinv, (size_t)s->n * sizeof(double), hipMemcpyDeviceToHost));
            std::vector<double> dict;
            std::vector<uint8_t> code((size_t)s->n + 2, 0);
            bool ok = true;
            for (int64_t i = 0; i < s->n && ok; ++i) {
                int c = -1;
                for (size_t k = 0; k < dict.size(); ++k)
                    if (std::memcmp(&dict[k], &h[(size_t)i], 8) == 0) {
                        c = (int)k;
                        break;
                    }
                if (c < 0) {
                    if (dict.size() == 256) {
                        ok = false;
                        break;
                    }
                    c = (int)dict.size();
                    dict.push_back(h[(size_t)i]);
                }
                code[(size_t)i] = (uint8_t)c;
            }
            if (ok && dict.size() == 1) {
                s->diag.mode = 3;
                s->diag.uniform = dict[0];
            } else if (ok && dict.size() <= 16 && !A->v.pair_id) {  // linear search above stays cheap
                // (row-pair coded matrices keep the full vector: their q-free iteration reads it as
                // such and beats the stored-q iteration the codes would select)
                int rc;
                if ((rc = upload(code.data(), code.size(), &s->d_dcode)) ||
                    (rc = upload(dict.data(), dict.size(), &s->d_ddict)))
                    return rc;
                s->diag.mode = 2;
                s->diag.code = (const uint8_t *)s->d_dcode;
                s->diag.dict = (const double *)s->d_ddict;
                s->diag.ndict = (int)dict.size();
            }
        }
    }
    if (precond == SCHWZ_PRECOND_BLOCK_JACOBI || precond == SCHWZ_PRECOND_ILU || precond == SCHWZ_PRECOND_ISAI) {
        int rc = pcg_setup_general(s);
        if (rc) return rc;
    }
    return SCHWZ_OK;
}

int schwz_pcg_create_ex(const schwz_csr *A, int precond, int block_size, schwz_pcg **out)
{
    StageTimer timer_all("pcg_create (diagonal, preconditioner data, vectors)");
    SCHWZ_REQUIRE(A && out, "schwz_pcg_create: null argument");
    SCHWZ_REQUIRE(A->v.nrows == A->v.ncols, "schwz_pcg_create: matrix not square");
    SCHWZ_REQUIRE(precond >= SCHWZ_PRECOND_NONE && precond <= SCHWZ_PRECOND_ISAI,
                  "schwz_pcg_create: unknown preconditioner");
    SCHWZ_REQUIRE(block_size >= 1 && block_size <= 32, "schwz_pcg_create: block size must be in 1..32");
    if (precond == SCHWZ_PRECOND_BLOCK_JACOBI && block_size == 1) precond = SCHWZ_PRECOND_JACOBI;
    schwz_pcg *s = new schwz_pcg();
    s->A = A;
    s->precond = precond;
    s->block_size = block_size;
    s->n = A->v.nrows;
    const int rc = pcg_build(s, A, precond);
    if (rc) {
        schwz_pcg_destroy(s);
        return rc;
    }
    *out = s;
    return SCHWZ_OK;
}

void schwz_pcg_destroy(schwz_pcg *s)
{
    if (!s) return;
    if (s->prio_event) (void)hipEventDestroy(s->prio_event);
    (void)hipFree(s->r);
    (void)hipFree(s->r_alt);
    (void)hipFree(s->p);
    (void)hipFree(s->q);
    (void)hipFree(s->p_ring);
    (void)hipFree(s->alpha_hist);
    (void)hipFree(s->dinv);
    (void)hipFree(s->z);
    (void)hipFree(s->d_blk_id);
    (void)hipFree(s->d_blk_inv);
    (void)hipFree(s->d_row_blk);
    (void)hipFree(s->d_blk_ptr);
    for (auto &g : s->graphs) (void)hipGraphExecDestroy(g.exec);
    if (s->capture_stream) (void)hipStreamDestroy(s->capture_stream);
    schwz_trs_destroy(s->ilu);
    schwz_csr_destroy(s->isai_l);
    schwz_csr_destroy(s->isai_u);
    (void)hipFree(s->isai_tmp);
    (void)hipFree(s->d_dcode);
    (void)hipFree(s->d_ddict);
    (void)hipFree(s->partials);
    (void)hipFree(s->d_norm_sq);
    (void)hipFree(s->state);
    (void)hipHostFree(s->h_state);
    if (s->ev[0]) (void)hipEventDestroy(s->ev[0]);
    if (s->ev[1]) (void)hipEventDestroy(s->ev[1]);
    delete s;
}

}  // extern "C"

namespace schwz {

// z = M^-1 r for the preconditioners that are operators of their own
__global__ __launch_bounds__(kBlock) void diag_scale_kernel(int64_t n, const double *__restrict__ dinv,
                                                            const double *__restrict__ in, double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = dinv[i] * in[i];
}

// out = M^-1 in for whichever preconditioner the object holds (in != out)
int precond_apply(schwz_pcg *s, const double *in, double *out, hipStream_t st)
{
    if (s->n == 0) return SCHWZ_OK;
    if (s->precond == SCHWZ_PRECOND_ILU) return schwz_trs_solve(s->ilu, in, out, (schwz_stream)st);
    if (s->precond == SCHWZ_PRECOND_ISAI) {
        SpmvArgs a;
        a.x = in;
        a.y = s->isai_tmp;
        int rc = launch_spmv(s->isai_l->v, kSpmvPlain, a, 0, st);
        if (rc) return rc;
        a.x = s->isai_tmp;
        a.y = out;
        return launch_spmv(s->isai_u->v, kSpmvPlain, a, 0, st);
    }
    if (s->precond == SCHWZ_PRECOND_BLOCK_JACOBI && s->d_row_blk) {
        hipLaunchKernelGGL(block_jacobi_var_apply_kernel, dim3(grid_for(s->n)), dim3(kBlock), 0, st, s->n, s->block_size,
                           s->d_row_blk, s->d_blk_ptr, s->d_blk_id, s->d_blk_inv, in, out);
    } else if (s->precond == SCHWZ_PRECOND_BLOCK_JACOBI) {
        hipLaunchKernelGGL(block_jacobi_apply_kernel, dim3(grid_for(s->n)), dim3(kBlock), 0, st, s->n, s->block_size,
                           s->d_blk_id, s->d_blk_inv, in, out);
    } else if (s->precond == SCHWZ_PRECOND_JACOBI) {
        hipLaunchKernelGGL(diag_scale_kernel, dim3(grid_for(s->n)), dim3(kBlock), 0, st, s->n, s->dinv, in, out);
    } else {
        return launch_copy(s->n, in, out, st);
    }
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

static int pcg_apply_general(schwz_pcg *s, hipStream_t st) { return precond_apply(s, s->r, s->z, st); }

int pcg_take_trs_error(schwz_pcg *s) { return s && s->ilu ? trs_take_error(s->ilu) : SCHWZ_OK; }

// The residual update and state advance a solve postponed (schwz_pcg::LazyLast): launched now, on the solve's
// stream, for a caller that wants the iteration count or the final residual norm.
int pcg_finish_lazy(schwz_pcg *s)
{
    if (!s || !s->lazy.pending) return SCHWZ_OK;
    s->lazy.pending = false;
    return s->lazy_run ? s->lazy_run(s->lazy.stream) : SCHWZ_OK;
}

int pcg_last_stats(schwz_pcg *s, int *h_iters, double *h_resnorm)
{
    const int rc_lazy = pcg_finish_lazy(s);
    if (rc_lazy) return rc_lazy;
    SCHWZ_HIP_TRY(hipDeviceSynchronize());
    SCHWZ_HIP_TRY(hipMemcpy(&s->h_state[0], s->state, sizeof(CgState), hipMemcpyDeviceToHost));
    *h_iters = s->h_state[0].iters;
    *h_resnorm = sqrt(s->h_state[0].rr);
    if (*h_resnorm != *h_resnorm) return pcg_take_trs_error(s);
    return SCHWZ_OK;
}

static bool pcg_is_general(const schwz_pcg *s)
{
    return s->precond == SCHWZ_PRECOND_BLOCK_JACOBI || s->precond == SCHWZ_PRECOND_ILU ||
           s->precond == SCHWZ_PRECOND_ISAI;
}

// How a solve on this system iterates (decided once per solve, the same way in pcg_begin and pcg_iterate).
struct CgPlan {
    bool qfree = false;         // row-pair coded matrix: q = A p is recomputed, never stored
    int dot_mode = kSpmvDot;    // launch that yields p.(A p)
    bool sweep_on = false;      // z-sweep walk of the update launch
    bool sweep_dirdot = false;  // ... and of the fused direction + p.(A p) launch
    bool fusedir = false;       // two launches per iteration
    bool deferx = false;        // x += sum alpha_k p_k applied once per kDeferDepth iterations
    bool sweep_start = false;   // the solve can start in the walk too (INIT / FIRST forms, spmv_pair.hip)
    int flavour = 0;
};

static CgPlan pcg_plan(schwz_pcg *s)
{
    CgPlan pl;
    const CsrView &A = s->A->v;
    const int64_t n = s->n;
    const int gs = spmv_grid(A, s->variant);
    const bool general = pcg_is_general(s);
    // SCHWZ_CG_QFREE=0 keeps the stored-q iteration for row-pair coded matrices too (A/B runs)
    static const bool qfree_on = [] {
        const char *e = std::getenv("SCHWZ_CG_QFREE");
        return !(e && e[0] == '0');
    }();
    pl.qfree = qfree_on && !general && A.pair_id && s->variant == 0 && s->diag.mode != 2;
    // p.(A p) from the upper triangle when the upload found the matrix symmetric (SCHWZ_CG_SYM=0: full rows)
    static const bool sym_on = [] {
        const char *e = std::getenv("SCHWZ_CG_SYM");
        return !(e && e[0] == '0');
    }();
    pl.dot_mode = !pl.qfree ? kSpmvDot : (sym_on && A.pair_sym_base > 0 ? kSpmvDotSym : kSpmvDotOnly);
    // Two launches per iteration: the direction update and the NEXT iteration's p.(A p) share one
    // launch (kSpmvDirDotSym), p alternating between two buffers (a launch that recomputes its
    // neighbours' new p must not overwrite the old one): s->p and the otherwise unused s->q, or the
    // slots of the deferred-x ring.  The first p.(A p) of a solve is launched on its own.
    // On launch-bound systems (up to kGraphRows rows) this is -11 to -13 % per outer iteration.  On
    // large ones the fused launch costs what the two it replaces cost -- 0.102 vs 0.067 + 0.045 ms on the
    // 256^3 cube (+1 % on the bench line), 0.111 vs 0.067 + 0.043 ms on the 512 x 512 x 64 slab of the
    // multi-GPU runs (-1 %) -- so they keep three launches.  SCHWZ_CG_FUSEDIR=0: never, =2: every size
    // (it combines with the deferred x update).
    static const int fusedir_mode = [] {
        const char *e = std::getenv("SCHWZ_CG_FUSEDIR");
        return e ? std::atoi(e) : 1;
    }();
    const char *dx_env = std::getenv("SCHWZ_CG_DEFERX");  // read per solve: tests switch it
    const int dx_mode = dx_env ? std::atoi(dx_env) : 1;
    // ... unless the matrix takes the z-sweep walk (spmv_pair.hip): there the fused launch loads every
    // element of r and p once instead of gathering both at every entry, and replaces 32 n bytes of the
    // two launches by 24 n (SCHWZ_CG_SWEEP=0: chunk-by-chunk launches only).
    const char *sweep_env = std::getenv("SCHWZ_CG_SWEEP");
    pl.sweep_on = !(sweep_env && sweep_env[0] == '0') && A.sweep_nslots > 0 && s->variant == 0 &&
                  A.ncols < (int64_t(1) << 28) && A.sweep_nslots + A.sweep_gen_blocks <= gs;
    pl.sweep_dirdot = pl.sweep_on && A.canon_sym_val && (s->diag.mode == 0 || s->diag.mode == 3) &&
                      A.sweep_nslots_dir + A.sweep_gen_blocks <= gs;
    pl.fusedir = pl.dot_mode == kSpmvDotSym &&
                 (fusedir_mode == 2 || (fusedir_mode == 1 && (n <= kGraphRows || pl.sweep_dirdot)));
    pl.flavour = !pl.qfree ? 0 : (pl.fusedir ? 2 : 1);
    // Large systems: x is not touched inside the iteration.  The search directions of up to
    // kDeferDepth iterations stay in a ring (slots 0 and 1 are s->p and the otherwise unused s->q),
    // the update launch stores alpha_k instead of updating x, and one launch per kDeferDepth
    // iterations (and one at the end) applies x += sum_k alpha_k p_k in iteration order -- the same
    // bits, (depth + 2) / depth vectors of traffic per iteration instead of 2.
    // SCHWZ_CG_DEFERX=0: never, =2: every size (tests).
    // The stored-q iteration of the scalar-Jacobi / unpreconditioned CG (plain CSR and the other codings)
    // defers x the same way; its ring cannot use s->q (it holds q), so it takes one more vector.
    pl.deferx = !general && !s->ring_failed && (dx_mode == 2 || (dx_mode == 1 && n > kGraphRows));
    if (pl.deferx && !s->p_ring) {
        const size_t nb = (size_t)((n + 1) & ~int64_t(1)) * sizeof(double);
        s->ring_has_q = pl.qfree;
        if (hipMalloc((void **)&s->p_ring, nb * (kDeferDepth - (pl.qfree ? 2 : 1))) != hipSuccess ||
            hipMalloc((void **)&s->alpha_hist, kDeferDepth * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();  // not enough memory for the ring: the plain iteration
            (void)hipFree(s->p_ring);
            s->p_ring = nullptr;
            s->ring_failed = true;
            pl.deferx = false;
        }
    }
    // A solve whose iterations all run in the walk starts in it as well: the start launch takes the walk
    // (32 n -> 24 n bytes: p is not stored) and the first p.(A p) comes from the FIRST form of the fused
    // direction launch, which builds p = D^-1 r from the r it reads anyway (SCHWZ_CG_SWEEPSTART=0: the
    // chunk-by-chunk start launch + kSpmvDotSym).
    const char *start_env = std::getenv("SCHWZ_CG_SWEEPSTART");
    pl.sweep_start = !(start_env && start_env[0] == '0') && pl.deferx && pl.sweep_dirdot && pl.fusedir &&
                     pair_sweep_start_ok(A, gs);
    return pl;
}

// First half of a solve: r = b - A x, p = M^-1 r, rho, ||r||^2 -> CgState.  With
// `fused` the same pass over the matrix also yields ||b - A x2||^2 over the rows
// below row_limit in s->d_norm_sq[0] (x2 == nullptr: x2 is x).
int pcg_begin(schwz_pcg *s, const double *d_b, double *d_x, double rtol, bool fused, const double *d_x2,
              int64_t row_limit, hipStream_t st)
{
    const CsrView &A = s->A->v;
    const int gs = spmv_grid(A, s->variant);
    s->lazy.pending = false;  // nobody asked for the last solve's final residual: it is recomputed from b - A x now
    SpmvArgs a;
    a.x = d_x;
    a.x2 = d_x2;
    a.b = d_b;
    a.y = s->r;
    a.p = s->p;
    a.dinv = s->dinv;
    a.partials = s->partials;
    a.row_limit = row_limit;
    // x2 == x over all rows: the check residual IS the start residual (rr bank)
    const bool same = fused && d_x2 == nullptr && row_limit >= s->n;
    if (pcg_is_general(s)) a.dinv = nullptr;  // p := r for now, z follows
    if (!pcg_is_general(s) && A.pair_id && s->variant == 0 && s->diag.mode == 3) {
        // row-pair kernel: a uniform Jacobi diagonal travels as a scalar, not as a vector of n equal values
        a.dinv = nullptr;
        a.diag_mode = 3;
        a.diag_uniform = s->diag.uniform;
    }
    // the start launch in the z-sweep walk where the whole solve runs in it (not with the second product of
    // the fused check residual, which the walk does not have)
    s->p_pending = false;
    // (with the second product of the fused check residual only where the upload built its plane flags)
    const bool walk_start = !pcg_is_general(s) && (a.diag_mode == 3 || !a.dinv) &&
                            (!(fused && !same) || pair_sweep_dual_ok(A, gs));
    if (walk_start && pcg_plan(s).sweep_start) {
        a.sweep_init = 1;
        a.p = nullptr;
        s->p_pending = true;
    }
    int rc = launch_spmv(A, (fused && !same) ? kSpmvResidDual : kSpmvResidInit, a, s->variant, st);
    if (rc) return rc;
    if (pcg_is_general(s)) {
        // the check-residual norm (bank 1 or 2 of the SpMV partials) first, then z = M^-1 r,
        // p = z and rho = r.z, ||r||^2 from the vector-kernel partials
        if (fused) {
            if ((rc = launch_final_norm(s->partials + (same ? 1 : 2) * gs, gs, s->d_norm_sq, st))) return rc;
        }
        if ((rc = pcg_apply_general(s, st))) return rc;
        const int gv = grid_for(s->n);
        double *part_vec = s->partials + 3 * kMaxGrid;
        hipLaunchKernelGGL(dot_rz_kernel, dim3(gv), dim3(kBlock), 0, st, s->n, s->r, s->z, s->p, nullptr, 0, part_vec);
        hipLaunchKernelGGL(cg_init_finalize_kernel, dim3(1), dim3(kBlock), 0, st, s->state, part_vec, gv, rtol, nullptr,
                           1);
        SCHWZ_HIP_TRY(hipGetLastError());
        return SCHWZ_OK;
    }
    hipLaunchKernelGGL(cg_init_finalize_kernel, dim3(1), dim3(kBlock), 0, st, s->state, s->partials, gs, rtol,
                       fused ? s->d_norm_sq : nullptr, same ? 1 : 2);
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

// Second half: up to max_iters CG updates.  With a positive tolerance the host
// looks at the state every `chunk` iterations, one chunk behind the launches, so
// the queue never drains.
int pcg_iterate(schwz_pcg *s, double *d_x, double rtol, int max_iters, hipStream_t st)
{
    const CsrView &A = s->A->v;
    const int64_t n = s->n;
    const int gs = spmv_grid(A, s->variant);
    const int gv = grid_for((n + 1) / 2);
    double *part_spmv = s->partials;                // [3][gs]
    double *part_vec = s->partials + 3 * kMaxGrid;  // [2][gv]
    const bool poll = rtol > 0.0;
    const bool general = pcg_is_general(s);
    const CgPlan plan = pcg_plan(s);
    const bool qfree = plan.qfree, sweep_on = plan.sweep_on, sweep_dirdot = plan.sweep_dirdot, fusedir = plan.fusedir;
    const bool deferx = plan.deferx;
    const int dot_mode = plan.dot_mode, flavour = plan.flavour;
    s->last_flavour = flavour | (deferx ? 4 : 0) | (sweep_on && deferx && s->diag.mode != 2 ? 8 : 0) |
                      (sweep_dirdot && fusedir ? 16 : 0) | (s->p_pending ? 32 : 0);
    // Virtual first direction (round 3; SCHWZ_CG_P0VIRTUAL=0: stored as before).  A solve that started in the walk
    // has p0 = D^-1 r0 with D^-1 uniform or absent, and r0 is in memory: the first-direction launch then stores
    // nothing (windows and the sums of p0.(A p0) only), the first update walk builds its windows from r0 x D^-1 and
    // writes r1 to the OTHER residual buffer, and r0 -- intact -- serves as p0 for the first fused direction launch
    // and for the x update.  16 n bytes per solve less (the p0 store and one p0 read), the same bits: every reader
    // forms the same rounded product D^-1 r0 the store would have held.
    const char *p0_env = std::getenv("SCHWZ_CG_P0VIRTUAL");  // read per solve: tests switch it
    bool p0_virtual = !(p0_env && p0_env[0] == '0') && s->p_pending && fusedir && plan.sweep_start && sweep_on &&
                      sweep_dirdot && qfree && deferx && !general && max_iters >= 2 &&
                      (s->diag.mode == 0 || s->diag.mode == 3);
    if (p0_virtual && !s->r_alt) {
        const size_t nb = (size_t)((n + 1) & ~int64_t(1)) * sizeof(double);
        if (hipMalloc((void **)&s->r_alt, nb) != hipSuccess) {
            (void)hipGetLastError();
            s->r_alt = nullptr;
            p0_virtual = false;  // no room for the second residual: the stored form
        }
    }
    const double p0_scale = p0_virtual && s->diag.mode != 0 ? s->diag.uniform : 1.0;
    if (p0_virtual) s->last_flavour |= 64;
    double *const r0_vec = s->r;  // where the start launch left r0
    PRing ring;
    const int64_t n_pad = (n + 1) & ~int64_t(1);
    for (int k = 0; k < kDeferDepth; ++k)
        ring.slot[k] = k == 0 ? s->p
                                : (!s->p_ring ? s->p
                                              : (s->ring_has_q ? (k == 1 ? s->q : s->p_ring + (int64_t)(k - 2) * n_pad)
                                                               : s->p_ring + (int64_t)(k - 1) * n_pad));
    if (p0_virtual) ring.slot[0] = r0_vec;  // (a later direction 16, 32, ... simply lands there: r0 is done with by then)
    auto slot = [&](int it) -> double * { return const_cast<double *>(ring.slot[it % kDeferDepth]); };
    bool prio_recorded = false;
    const int fused_x = qfree ? 0 : 1;  // how the in-launch update of this iteration forms x + alpha p
    // The last iteration of a solve of exactly max_iters iterations (rtol == 0: no stopping test can fire), x
    // deferred: what its result needs is alpha = rho / (p.Ap) and x += alpha p, and both happen inside the last x
    // update.  The residual update r -= alpha A p with rho' and ||r||^2, and the state advance, produce nothing
    // anybody reads -- the next solve starts from b - A y -- unless the caller asks for the iteration count or
    // the residual norm: they are postponed (schwz_pcg::lazy, pcg_finish_lazy) instead of launched.  Same x bit
    // for bit.  SCHWZ_CG_LAZYLAST=0: every iteration is launched in full.
    const char *lazy_env = std::getenv("SCHWZ_CG_LAZYLAST");  // read per solve: tests switch it
    const bool lazy_on = !(lazy_env && lazy_env[0] == '0');
    const char *ld_env0 = std::getenv("SCHWZ_CG_LASTDIR");
    const bool lazy_last = lazy_on && rtol == 0.0 && deferx && !general && max_iters > 0 && !(ld_env0 && ld_env0[0] == '1');
    const int lazy_it = lazy_last ? max_iters - 1 : -1;
    const double *const lazy_pq = part_spmv;  // p.(A p) of the last iteration: SpMV bank 0, gs slots
    // ... and the direction of that last iteration is never stored (round 3; SCHWZ_CG_PLASTVIRTUAL=0: stored): its
    // only readers are the x update -- which rebuilds it from the direction before it, which it reads anyway, and
    // the current residual: the same fma the fused direction launch performed -- and the postponed update launch,
    // which gets it rebuilt first (cg_rebuild_direction_kernel).  8 n bytes of stores per solve less.
    const char *pl_env = std::getenv("SCHWZ_CG_PLASTVIRTUAL");
    const bool vlast_on = lazy_last && !(pl_env && pl_env[0] == '0') && max_iters >= 2 && fusedir && sweep_dirdot &&
                          sweep_on && qfree && plan.sweep_start && (s->diag.mode == 0 || s->diag.mode == 3);
    if (vlast_on) s->last_flavour |= 128;
    s->x2_written = false;
    auto flush_x = [&](int b0, int count, int pending, hipStream_t q, bool last = false) {
        const int64_t n2 = n >> 1;
        // the second output only from the launches that finish x (`last`)
        double *const x2 = last ? s->x2_out : nullptr;
        const int64_t x2_rows = last ? s->x2_rows : 0;
        const int lz = last ? lazy_it : -1;
        const int vl = last && vlast_on ? lazy_it : -1;  // the never-stored last direction (see vlast_on)
        const double *const vlast_r = s->r;
        if (last && s->prio_on && s->prio_event && q == st && !prio_recorded) {
            // the caller's priority rows first, the event, then the rest (the same bits: every element is
            // updated by exactly one lane of exactly one of the launches)
            const int64_t lo = std::min(s->prio_lo >> 1, n2), hi = std::max(std::min(s->prio_hi >> 1, n2), lo);
            if (lo > 0)
                hipLaunchKernelGGL(cg_flush_x_kernel, dim3(grid_for(lo)), dim3(kBlock), 0, q, n, d_x, ring, s->alpha_hist,
                                   s->state, b0, count, pending, (int64_t)0, lo, 0, fused_x, lz, lazy_pq, gs, x2, x2_rows, s->x2_src, s->x2_total, p0_scale, vl, vlast_r, s->diag.mode, s->diag.uniform);
            // (this launch also copies the entries of x2 beyond the solve's vector -- the halo of x~, two planes of a
            // slab --, element by element over its whole grid: sized for that too.  With one workgroup, which is what
            // the upper priority range of a subdomain without an upper neighbour asks for, that copy took 0.44 ms.)
            const int64_t tail_items = x2 ? std::max<int64_t>(s->x2_total - 2 * n2, 0) : 0;
            hipLaunchKernelGGL(cg_flush_x_kernel, dim3(std::max(grid_for(n2 - hi + 1), grid_for(tail_items))), dim3(kBlock), 0, q, n, d_x, ring,
                               s->alpha_hist, s->state, b0, count, pending, hi, n2, 1, fused_x, lz, lazy_pq, gs, x2, x2_rows, s->x2_src, s->x2_total, p0_scale, vl, vlast_r, s->diag.mode, s->diag.uniform);
            if (hipEventRecord(s->prio_event, q) == hipSuccess) prio_recorded = true;
            if (hi > lo)
                hipLaunchKernelGGL(cg_flush_x_kernel, dim3(grid_for(hi - lo)), dim3(kBlock), 0, q, n, d_x, ring,
                                   s->alpha_hist, s->state, b0, count, pending, lo, hi, 0, fused_x, lz, lazy_pq, gs, x2, x2_rows, s->x2_src, s->x2_total, p0_scale, vl, vlast_r, s->diag.mode, s->diag.uniform);
            if (x2) s->x2_written = true;
            return;
        }
        hipLaunchKernelGGL(cg_flush_x_kernel, dim3(gv), dim3(kBlock), 0, q, n, d_x, ring, s->alpha_hist, s->state, b0,
                           count, pending, (int64_t)0, n2, 1, fused_x, lz, lazy_pq, gs, x2, x2_rows, s->x2_src, s->x2_total, p0_scale, vl, vlast_r, s->diag.mode, s->diag.uniform);
        if (x2) s->x2_written = true;
    };
    double *const pbuf[2] = {s->p, fusedir ? s->q : s->p};
    // SCHWZ_CG_LASTDIR=1: the last iteration of a solve updates the search direction like every other one
    const char *ld_env = std::getenv("SCHWZ_CG_LASTDIR");
    const bool last_state_only = !(ld_env && ld_env[0] == '1');
    // one CG iteration on stream `q`; `it` only enters through its parity (rho slot) and through
    // "it >= stop_iter", and stop_iter is 0 once the tolerance test has fired: a recorded sequence
    // of an even number of iterations can therefore be replayed as a hipGraph
    auto launch_iteration = [&](int it, hipStream_t q, bool instrument) -> int {
        SpmvArgs a;
        a.x = deferx ? slot(it) : s->p;
        a.y = s->q;
        a.partials = part_spmv;
        a.stop_iter = &s->state->stop_iter;
        a.it = it;
        const bool prof = instrument && g_prof.on && g_prof.used + 2 <= g_prof.ev.size();
        int rc = SCHWZ_OK;
        if (!fusedir) {
            if (prof) SCHWZ_HIP_TRY(hipEventRecord(g_prof.ev[g_prof.used], q));
            if ((rc = launch_spmv(A, dot_mode, a, s->variant, q))) return rc;
            if (prof) {
                SCHWZ_HIP_TRY(hipEventRecord(g_prof.ev[g_prof.used + 1], q));
                g_prof.kind[g_prof.used / 2] = 0;
                g_prof.used += 2;
            }
        }
        if (qfree) {
            // q = A p is never stored: the update pass recomputes (A p)_i row by row while it
            // streams x and r (spmv_pair.hip, kSpmvCgUpdate): 16 B per row less HBM traffic, a
            // third of the stores of these two launches
            SpmvArgs u;
            u.x = deferx ? slot(it) : pbuf[it & 1];
            u.cg_x = deferx ? nullptr : d_x;
            u.alpha_out = deferx ? s->alpha_hist + it % kDeferDepth : nullptr;
            u.cg_r = s->r;
            u.cg_state = s->state;
            u.pq_partials = part_spmv;
            u.pq_nparts = gs;
            u.diag_mode = s->diag.mode;
            u.diag_uniform = s->diag.uniform;
            u.dinv = s->dinv;
            u.partials = part_vec;
            u.it = it;
            const bool first_virtual = p0_virtual && instrument && it == 0;
            if (first_virtual) {
                u.x = r0_vec;  // windows = ring_scale x r0 = p0
                u.ring_scale = p0_scale;
                u.cg_r_out = s->r_alt;
                u.p0_virtual = 1;
            }
            const bool lazy_now = instrument && it == lazy_it;
            const bool prof2 = !lazy_now && instrument && g_prof.on && g_prof.used + 2 <= g_prof.ev.size();
            if (prof2) SCHWZ_HIP_TRY(hipEventRecord(g_prof.ev[g_prof.used], q));
            if (!lazy_now && (rc = launch_spmv(A, kSpmvCgUpdate, u, s->variant, q))) return rc;
            if (first_virtual) std::swap(s->r, s->r_alt);  // r1 (and every later residual) lives in the other buffer
            if (prof2) {
                SCHWZ_HIP_TRY(hipEventRecord(g_prof.ev[g_prof.used + 1], q));
                g_prof.kind[g_prof.used / 2] = 1;
                g_prof.used += 2;
            }
            if (fusedir && !(instrument && it == max_iters - 1)) {
                // p' = z + beta p into the other buffer and the partial sums of p'.(A p') for the next
                // iteration (the last iteration of a solve only needs the direction kernel's state
                // update; recorded graphs replay mid-solve, so they keep the fused launch)
                if (deferx && (it + 1) % kDeferDepth == 0) flush_x(it + 1 - kDeferDepth, kDeferDepth, it, q);
                SpmvArgs f;
                f.x = deferx ? slot(it) : pbuf[it & 1];
                f.y = deferx ? slot(it + 1) : pbuf[(it + 1) & 1];
                f.cg_r = s->r;
                f.cg_state = s->state;
                f.pq_partials = part_vec;
                f.pq_nparts = gs;
                f.diag_mode = s->diag.mode;
                f.diag_uniform = s->diag.uniform;
                f.dinv = s->dinv;
                f.partials = part_spmv;
                f.it = it;
                f.cg_rtol = rtol;
                if (first_virtual) {
                    f.x = r0_vec;  // p0 = p_scale x r0
                    f.p_scale = p0_scale;
                    f.p0_virtual = 1;
                }
                if (vlast_on && instrument && it + 1 == lazy_it) {
                    f.y = nullptr;     // windows and sums only: nobody reads this direction from memory
                    f.p0_virtual = 1;  // (walk kernels only)
                }
                const bool prof3 = instrument && g_prof.on && g_prof.used + 2 <= g_prof.ev.size();
                if (prof3) SCHWZ_HIP_TRY(hipEventRecord(g_prof.ev[g_prof.used], q));
                if ((rc = launch_spmv(A, s->diag.mode == 1 ? kSpmvDirDotSymVec : kSpmvDirDotSym, f, s->variant, q)))
                    return rc;
                if (prof3) {
                    SCHWZ_HIP_TRY(hipEventRecord(g_prof.ev[g_prof.used + 1], q));
                    g_prof.kind[g_prof.used / 2] = 0;
                    g_prof.used += 2;
                }
            } else if (deferx) {
                // the ring is full: apply its kDeferDepth increments before slot (it + 1) % depth,
                // the oldest direction, is overwritten
                if ((it + 1) % kDeferDepth == 0) flush_x(it + 1 - kDeferDepth, kDeferDepth, it, q, instrument && it == max_iters - 1);
                if (lazy_now) {
                    // the residual update and the state advance of this iteration wait until somebody asks
                    const CsrView Av = A;
                    const int variant = s->variant;
                    double *const pv = part_vec;
                    const bool rebuild = vlast_on;
                    const double *const p_prev = slot(it - 1 >= 0 ? it - 1 : 0);
                    const double prev_scale = it == 1 ? p0_scale : 1.0;
                    double *const p_last = slot(it);
                    const int gvv = gv;
                    const int64_t nn = n;
                    s->lazy_run = [Av, u, variant, pv, gs, s, it, rtol, rebuild, p_prev, prev_scale, p_last, gvv, nn](hipStream_t qq) -> int {
                        if (rebuild)
                            hipLaunchKernelGGL(cg_rebuild_direction_kernel, dim3(gvv), dim3(kBlock), 0, qq, nn, p_prev, prev_scale,
                                               (const double *)s->r, s->diag.mode, s->diag.uniform, (const CgState *)s->state, it, p_last);
                        const int rc2 = launch_spmv(Av, kSpmvCgUpdate, u, variant, qq);
                        if (rc2) return rc2;
                        hipLaunchKernelGGL(cg_state_advance_kernel, dim3(1), dim3(kBlock), 0, qq, pv, gs, s->state, it, rtol);
                        SCHWZ_HIP_TRY(hipGetLastError());
                        return SCHWZ_OK;
                    };
                    s->lazy.pending = true;
                    s->lazy.it = it;
                    s->lazy.rtol = rtol;
                    s->lazy.stream = q;
                } else if (instrument && it == max_iters - 1 && last_state_only)  // nobody reads the direction after the last iteration
                    hipLaunchKernelGGL(cg_state_advance_kernel, dim3(1), dim3(kBlock), 0, q, part_vec, gs, s->state, it, rtol);
                else
                    hipLaunchKernelGGL((cg_direction_kernel<1, false>), dim3(gv), dim3(kBlock), 0, q, n, slot(it), s->r,
                                       s->diag, part_vec, gs, s->state, it, rtol, slot(it + 1));
            } else {
                hipLaunchKernelGGL((cg_direction_kernel<1, false>), dim3(gv), dim3(kBlock), 0, q, n, pbuf[it & 1], s->r,
                                   s->diag, part_vec, gs, s->state, it, rtol);
            }
        } else if (!general && deferx) {
            // stored q, x deferred: r -= alpha q (24 n bytes instead of 48-56 n), alpha to the history, the new
            // direction into the next ring slot; the last iteration of a solve only advances the state
            const bool lazy_now = instrument && it == lazy_it;
            if (!lazy_now)
                hipLaunchKernelGGL((cg_update_kernel<1, false, true>), dim3(gv), dim3(kBlock), 0, q, n, (double *)nullptr, s->r,
                                   (const double *)nullptr, s->q, s->diag, part_spmv, gs, s->state, it, part_vec,
                                   s->alpha_hist + it % kDeferDepth);
            if ((it + 1) % kDeferDepth == 0) flush_x(it + 1 - kDeferDepth, kDeferDepth, it, q, instrument && it == max_iters - 1);
            if (lazy_now) {
                double *const ps = part_spmv, *const pv = part_vec;
                s->lazy_run = [s, n, gv, gs, ps, pv, it, rtol](hipStream_t qq) -> int {
                    hipLaunchKernelGGL((cg_update_kernel<1, false, true>), dim3(gv), dim3(kBlock), 0, qq, n, (double *)nullptr,
                                       s->r, (const double *)nullptr, s->q, s->diag, ps, gs, s->state, it, pv,
                                       s->alpha_hist + it % kDeferDepth);
                    hipLaunchKernelGGL(cg_state_advance_kernel, dim3(1), dim3(kBlock), 0, qq, pv, gv, s->state, it, rtol);
                    SCHWZ_HIP_TRY(hipGetLastError());
                    return SCHWZ_OK;
                };
                s->lazy.pending = true;
                s->lazy.it = it;
                s->lazy.rtol = rtol;
                s->lazy.stream = q;
            } else if (instrument && it == max_iters - 1 && last_state_only)
                hipLaunchKernelGGL(cg_state_advance_kernel, dim3(1), dim3(kBlock), 0, q, part_vec, gv, s->state, it, rtol);
            else
                hipLaunchKernelGGL((cg_direction_kernel<1, false>), dim3(gv), dim3(kBlock), 0, q, n, slot(it), s->r,
                                   s->diag, part_vec, gv, s->state, it, rtol, slot(it + 1));
        } else if (!general) {
            hipLaunchKernelGGL((cg_update_kernel<1, false>), dim3(gv), dim3(kBlock), 0, q, n, d_x, s->r, s->p, s->q,
                               s->diag, part_spmv, gs, s->state, it, part_vec);
            hipLaunchKernelGGL((cg_direction_kernel<1, false>), dim3(gv), dim3(kBlock), 0, q, n, s->p, s->r, s->diag,
                               part_vec, gv, s->state, it, rtol);
        } else {
            // x, r update without a preconditioner; z = M^-1 r; rho' = r.z; p = z + beta p
            const DiagView none;
            const int gz = grid_for(n);
            hipLaunchKernelGGL((cg_update_kernel<1, false>), dim3(gv), dim3(kBlock), 0, q, n, d_x, s->r, s->p, s->q,
                               none, part_spmv, gs, s->state, it, part_vec);
            if ((rc = pcg_apply_general(s, q))) return rc;
            hipLaunchKernelGGL(dot_rz_kernel, dim3(gz), dim3(kBlock), 0, q, n, s->r, s->z, (double *)nullptr, s->state,
                               it, part_vec);
            hipLaunchKernelGGL((cg_direction_kernel<1, false>), dim3(gv), dim3(kBlock), 0, q, n, s->p, s->z, none,
                               part_vec, gz, s->state, it, rtol);
        }
        return SCHWZ_OK;
    };
    // Small systems are bound by launches, not bytes (33 k rows: 3 launches of ~3 us work each):
    // kGraphIters iterations are captured once per (x, rtol) into a hipGraph -- on a private stream,
    // the caller's may be the legacy default stream -- and replayed.  SCHWZ_CG_GRAPH=0 disables,
    // =2 uses graphs for every size.
    static const int graph_mode = [] {
        const char *e = std::getenv("SCHWZ_CG_GRAPH");
        return e ? std::atoi(e) : 1;
    }();
    const bool graphable = graph_mode != 0 && !general && !g_prof.on && !deferx && (graph_mode == 2 || n <= kGraphRows);
    hipGraphExec_t replay = nullptr;
    if (graphable && max_iters >= kGraphIters) {
        for (const auto &g : s->graphs)
            if (g.x == d_x && g.rtol == rtol && g.variant == s->variant && g.qfree == flavour) replay = g.exec;
        if (!replay && s->graphs.size() < 4) {
            if (!s->capture_stream) SCHWZ_HIP_TRY(hipStreamCreateWithFlags(&s->capture_stream, hipStreamNonBlocking));
            if (hipStreamBeginCapture(s->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                int rc = SCHWZ_OK;
                for (int k = 0; k < kGraphIters && !rc; ++k) rc = launch_iteration(k, s->capture_stream, false);
                hipGraph_t graph = nullptr;
                const hipError_t e1 = hipStreamEndCapture(s->capture_stream, &graph);
                if (rc) {
                    if (graph) (void)hipGraphDestroy(graph);
                    return rc;
                }
                if (e1 == hipSuccess && hipGraphInstantiate(&replay, graph, nullptr, nullptr, 0) == hipSuccess)
                    s->graphs.push_back({d_x, rtol, s->variant, flavour, replay});
                else
                    replay = nullptr;
                if (graph) (void)hipGraphDestroy(graph);
            }
            (void)hipGetLastError();
        }
    }
    if (s->p_pending && !(fusedir && plan.sweep_start)) {
        set_error("pcg_iterate: the start launch left p to a first-direction launch this solve does not run");
        return SCHWZ_ERR_INVALID;
    }
    if (fusedir && max_iters > 0) {
        // p0.(A p0): every later p.(A p) comes out of the fused direction launch
        SpmvArgs a;
        a.partials = part_spmv;
        a.it = 0;
        int rc;
        if (s->p_pending) {
            // z-sweep start: p0 = D^-1 r0 is built here, from the r the launch reads anyway
            a.y = p0_virtual ? nullptr : slot(0);
            a.cg_r = s->r;
            a.cg_state = s->state;
            a.diag_mode = s->diag.mode;
            a.diag_uniform = s->diag.uniform;
            a.sweep_first = 1;
            rc = launch_spmv(A, kSpmvDirDotSym, a, s->variant, st);
        } else {
            a.x = s->p;
            a.stop_iter = &s->state->stop_iter;
            rc = launch_spmv(A, kSpmvDotSym, a, s->variant, st);
        }
        if (rc) return rc;
    }
    s->p_pending = false;
    int chunk = 16;
    int it = 0, pending = -1, bank = 0;
    bool stopped = false;
    while (it < max_iters && !stopped) {
        const int end = (poll && it + chunk < max_iters) ? it + chunk : max_iters;
        while (it < end) {
            if (replay && it % kGraphIters == 0 && end - it >= kGraphIters) {
                SCHWZ_HIP_TRY(hipGraphLaunch(replay, st));
                it += kGraphIters;
                continue;
            }
            int rc = launch_iteration(it, st, true);
            if (rc) return rc;
            ++it;
        }
        SCHWZ_HIP_TRY(hipGetLastError());
        if (poll && it < max_iters) {
            if (pending >= 0) {
                SCHWZ_HIP_TRY(hipEventSynchronize(s->ev[pending]));
                if (s->h_state[pending].stop_iter != INT_MAX) stopped = true;
            }
            SCHWZ_HIP_TRY(hipMemcpyAsync(&s->h_state[bank], s->state, sizeof(CgState), hipMemcpyDeviceToHost, st));
            SCHWZ_HIP_TRY(hipEventRecord(s->ev[bank], st));
            pending = bank;
            bank ^= 1;
            if (chunk < 64) chunk *= 2;
        }
    }
    // the increments of the last, partly filled ring (iterations past a tolerance stop are not
    // counted by CgState::iters and add nothing)
    if (deferx && it % kDeferDepth != 0) {
        flush_x(it - it % kDeferDepth, it % kDeferDepth, -1, st, true);
        SCHWZ_HIP_TRY(hipGetLastError());
    }
    // second output asked for, but no launch above was the last x update (ring just emptied, a stop between two
    // rings, no iteration at all): a launch that adds nothing and copies
    if (deferx && s->x2_out && !s->x2_written) {
        flush_x(it, 0, -1, st, true);
        SCHWZ_HIP_TRY(hipGetLastError());
    }
    // priority rows without a split update (x updated inside the iteration, or no iteration at all): final
    // behind the last launch
    if (s->prio_on && s->prio_event && !prio_recorded) SCHWZ_HIP_TRY(hipEventRecord(s->prio_event, st));
    return SCHWZ_OK;
}

}  // namespace schwz

extern "C" {

int schwz_pcg_solve(schwz_pcg *s, const double *d_b, double *d_x, double rtol, int max_iters,
                    int *h_iters, double *h_resnorm, schwz_stream stream)
{
    SCHWZ_REQUIRE(s && d_b && d_x, "schwz_pcg_solve: null argument");
    SCHWZ_REQUIRE(max_iters >= 0, "schwz_pcg_solve: negative max_iters");
    SCHWZ_REQUIRE((reinterpret_cast<uintptr_t>(d_x) & 15) == 0, "schwz_pcg_solve: x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (s->n == 0) {
        if (h_iters) *h_iters = 0;
        if (h_resnorm) *h_resnorm = 0.0;
        return SCHWZ_OK;
    }
    int rc = pcg_begin(s, d_b, d_x, rtol, false, nullptr, 0, st);
    if (rc) return rc;
    if ((rc = pcg_iterate(s, d_x, rtol, max_iters, st))) return rc;
    if (h_iters || h_resnorm) {
        if ((rc = pcg_finish_lazy(s))) return rc;
        SCHWZ_HIP_TRY(hipMemcpyAsync(&s->h_state[0], s->state, sizeof(CgState), hipMemcpyDeviceToHost, st));
        SCHWZ_HIP_TRY(hipStreamSynchronize(st));
        if (h_iters) *h_iters = s->h_state[0].iters;
        if (h_resnorm) *h_resnorm = sqrt(s->h_state[0].rr);
        if (s->h_state[0].rr != s->h_state[0].rr) return pcg_take_trs_error(s);
    }
    return SCHWZ_OK;
}

}  // extern "C"
