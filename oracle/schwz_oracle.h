/*
 * schwz_oracle.h -- CPU ORACLE for the Restricted Additive Schwarz hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * It is a plain-C restatement of the algorithm in pratikvn/schwarz-lib (paths
 * below are relative to the reference checkout); every function cites the
 * reference lines it follows.
 *
 * PARITY STATUS: "parity unpinned" by the reference itself -- the reference
 * ships no tests, golden vectors or known-answer files (TESTING.md:1-2), and
 * it cannot be built here (Ginkgo `expt-develop`, gflags, METIS, CHOLMOD are
 * absent; CMakeLists.txt:109-110).  The oracle is pinned instead by
 *   (i)  independent scipy direct solves committed under tests/golden/
 *        (tests/golden/make_golden.py), and
 *   (ii) structural known answers derived by hand from the reference's index
 *        set construction (source/restricted_schwarz.cpp:56-304).
 * Third-party arithmetic the reference delegates to Ginkgo (CG and GMRES
 * recurrences, Jacobi / block-Jacobi, ILU(0) standing in for ParILU, ISAI,
 * triangular solves) and CHOLMOD (LL^T) is restated from the published
 * algorithms.  GMRES is additionally pinned by scipy's implementation (same
 * inner iteration counts and solution, tests/test_oracle_golden.py), ILU(0) and
 * ISAI by their defining identities on the sparsity pattern.
 */
#ifndef SCHWZ_ORACLE_H
#define SCHWZ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t or_idx; /* the reference driver fixes IndexType=int (bench_ras.cpp:204) */

/* ---- problem generation -------------------------------------------------- */

/* 2-D 5-point Laplacian, exactly source/initialization.cpp:214-265 (A.4).
 * Two-call pattern: with col==NULL only row_ptr (n*n+1) is filled.  Returns nnz. */
int64_t schwz_or_laplacian2d(int n, or_idx *row_ptr, or_idx *col, double *val);

/* 3-D 7-point Dirichlet Laplacian (diag 6, off -1), x fastest; same convention
 * extended to 3-D (the reference has no 3-D generator). */
int64_t schwz_or_laplacian3d(int nx, int ny, int nz, or_idx *row_ptr,
                             or_idx *col, double *val);

/* rhs: all ones (source/schwarz_base.cpp:169). */
void schwz_or_rhs_ones(int64_t n, double *rhs);
/* Initialize::generate_rhs (source/initialization.cpp:88-96): the libstdc++ sequence of
 * uniform_real_distribution<double>(0,1) over a default-seeded default_random_engine */
void schwz_or_rhs_random(int64_t n, double *rhs);


/* ---- partitioning -------------------------------------------------------- */

/* contiguous row blocks nb=ceil(N/P) (source/restricted_schwarz.cpp:84,97-102). */
void schwz_or_first_rows_regular(int64_t N, int P, or_idx *first_row);

/* regular2d partition vector (include/partition_tools.hpp:70-106). */
int schwz_or_partition_regular2d(int n1d, int P, uint32_t *part);

/* permutation from a partition vector and the permuted matrix
 * (source/restricted_schwarz.cpp:105-152).  perm is new->old. */
void schwz_or_apply_partition(int64_t N, int P, const uint32_t *part,
                              const or_idx *rp, const or_idx *col,
                              const double *val, or_idx *perm, or_idx *iperm,
                              or_idx *first_row, or_idx *out_rp,
                              or_idx *out_col, double *out_val);

/* ---- per-subdomain index sets and matrices (A.1, A.2) --------------------- */

typedef struct or_subdomain or_subdomain;

or_subdomain *schwz_or_subdomain_setup(int64_t N, const or_idx *rp,
                                       const or_idx *col, const double *val,
                                       int P, int me, int overlap,
                                       const or_idx *first_row);
void schwz_or_subdomain_free(or_subdomain *sd);

/* sizes: [0]=local_size [1]=local_size_x [2]=overlap_size [3]=halo_size
 *        [4]=nnz_local [5]=nnz_interface [6]=num_neighbors_in
 *        [7]=num_neighbors_out(-1 until put lists set) [8]=num_recv [9]=num_send */
void schwz_or_subdomain_sizes(const or_subdomain *sd, int64_t *sizes10);
const or_idx *schwz_or_sd_local_to_global(const or_subdomain *sd); /* lsx+halo */
const or_idx *schwz_or_sd_local_rp(const or_subdomain *sd);
const or_idx *schwz_or_sd_local_col(const or_subdomain *sd);
const double *schwz_or_sd_local_val(const or_subdomain *sd);
const or_idx *schwz_or_sd_iface_rp(const or_subdomain *sd);
const or_idx *schwz_or_sd_iface_col(const or_subdomain *sd); /* GLOBAL ids */
const double *schwz_or_sd_iface_val(const or_subdomain *sd);
/* neighbour k (ascending rank order): returns rank, *count, *ids (global) */
int schwz_or_sd_get_list(const or_subdomain *sd, int k, or_idx *count,
                         const or_idx **ids);
int schwz_or_sd_put_list(const or_subdomain *sd, int k, or_idx *count,
                         const or_idx **ids);
/* the handshake of restricted_schwarz.cpp:400-472: neighbour p's get list for
 * me becomes my put list for p.  Call in ascending p. */
void schwz_or_sd_add_put_list(or_subdomain *sd, int p, or_idx count,
                              const or_idx *ids);

/* ---- solver settings ------------------------------------------------------ */

enum { OR_SOLVER_ITERATIVE = 0, OR_SOLVER_DIRECT = 1 };
enum { OR_PRECOND_NONE = 0, OR_PRECOND_JACOBI = 1, OR_PRECOND_BLOCK_JACOBI = 2, OR_PRECOND_ILU = 3, OR_PRECOND_ISAI = 4 };

typedef struct {
    int32_t max_iters;        /* metadata.max_iters (--num_iters) */
    double tol;               /* metadata.tolerance (--set_tol) */
    int32_t overlap;          /* settings.overlap */
    int32_t local_solver;     /* OR_SOLVER_* (settings.local_solver) */
    int32_t precond;          /* OR_PRECOND_* */
    double local_tol;         /* metadata.local_solver_tolerance */
    int32_t local_max_iters;  /* -1 => local_size_x (solve.cpp:458-463) */
    int32_t enable_global_check;     /* convergence_settings.enable_global_check */
    int32_t enable_onesided;         /* comm_settings.enable_onesided (local test) */
    int32_t global_check_iter_offset;/* enable_global_check_iter_offset */
    int32_t natural_factor_ordering; /* settings.naturally_ordered_factor */
    int32_t num_threads;             /* OpenMP threads for the kernels (0=default) */
    /* comm_settings.enable_overlap together with enable_onesided: the deterministic
     * asynchronous model of this build -- halos are consumed one iteration late and the
     * stop decision is flooded over neighbour messages (see schwz_or_ras_run) */
    int32_t enable_overlap;
    /* settings.use_mixed_precision with MixedValueType=float: halo values are rounded to fp32
     * on the wire (restricted_schwarz.cpp:898-903,952-954) */
    int32_t use_mixed_precision;
    int32_t precond_block_size; /* metadata.precond_max_block_size for OR_PRECOND_BLOCK_JACOBI */
    /* two-stage local solves (solve.cpp:723-742): once iter_count > reset_local_crit_iter the
     * inner iteration cap becomes updated_max_iters (-1 => local_size_x); -1 disables */
    int32_t reset_local_crit_iter;
    int32_t updated_max_iters;
    /* settings.non_symmetric_matrix: GMRES(settings.restart_iter) instead of CG (solve.cpp:486-520) */
    int32_t non_symmetric;
    int32_t restart_iter;
} or_settings;

/* ---- per-subdomain state and the five loop steps (A.3) -------------------- */

typedef struct or_state or_state;

or_state *schwz_or_state_create(or_subdomain *sd, const double *global_rhs,
                                const or_settings *s);
void schwz_or_state_free(or_state *st);
double *schwz_or_state_global_solution(or_state *st); /* length N, like the reference */
double *schwz_or_state_local_solution(or_state *st);  /* length local_size_x */
const double *schwz_or_state_local_rhs(or_state *st);
/* direct path factors (solve.cpp:284-320): U=L^T CSR, L CSR, perm */
int schwz_or_state_factors(or_state *st, const or_idx **l_rp, const or_idx **l_col,
                           const double **l_val, const or_idx **u_rp,
                           const or_idx **u_col, const double **u_val,
                           const or_idx **perm);

/* step 0 halves: restricted_schwarz.cpp:884-911 (pack) and :950-962 (unpack) */
void schwz_or_pack(or_state *st, int k_out, double *send);
void schwz_or_unpack(or_state *st, int k_in, const double *recv);
/* step 1: restricted_schwarz.cpp:992-1017 */
void schwz_or_update_boundary(or_state *st);
/* step 2 (local half): solve.cpp:796-856; returns ||r||_2 */
double schwz_or_local_residual(or_state *st);
/* step 3: solve.cpp:667-792; returns inner iterations (0 for direct) */
int schwz_or_local_solve(or_state *st);
/* step 4: communicate.cpp:65-94 */
void schwz_or_restrict(or_state *st);
/* the inner iteration cap of later local solves (solve.cpp:723-742 rebuilds the criterion) */
void schwz_or_state_set_local_max_iters(or_state *st, int32_t max_iters);

/* ---- whole run: all P subdomains in lockstep (schwarz_base.cpp:387-452) ---- */

typedef struct {
    int32_t iter_count;
    int32_t converged;
    double residual_norm; /* ||b - A x|| (solve.cpp:1025-1085) */
    double rhs_norm;
    double sol_norm;
    double elapsed_s;     /* loop only, like schwarz_base.cpp:384,453-455 */
    double setup_s;
} or_result;

/* hist_global[max_iters], hist_local[max_iters*P], hist_inner[max_iters*P]
 * may be NULL. */
int schwz_or_ras_run(int64_t N, const or_idx *rp, const or_idx *col,
                     const double *val, const double *rhs, int P,
                     const or_idx *first_row, const or_settings *s,
                     double *solution, double *hist_global, double *hist_local,
                     int32_t *hist_inner, or_result *res);

/* ---- stand-alone kernels (used to check the HIP kernels one by one) -------- */

/* y = alpha*A*x + beta*y (Csr::apply 4-arg) */
void schwz_or_spmv(int64_t nrows, const or_idx *rp, const or_idx *col,
                   const double *val, double alpha, const double *x,
                   double beta, double *y);
/* gather/scatter with the four ops of include/gather.hpp:86-113 /
 * include/scatter.hpp:86-113; op: 0=add 1=copy 2=diff 3=avg
 * (enum order of include/collective_common.hpp) */
void schwz_or_gather(int64_t n, const or_idx *idx, const double *from,
                     double *into, int op);
void schwz_or_scatter(int64_t n, const or_idx *idx, const double *from,
                      double *into, int op);
/* PCG on A x = b from x (warm start); returns iterations done. */
int schwz_or_pcg(int64_t n, const or_idx *rp, const or_idx *col,
                 const double *val, const double *b, double *x, int precond,
                 double rtol, int max_iters, double *final_resnorm);
int schwz_or_pcg_ex(int64_t n, const or_idx *rp, const or_idx *col, const double *val, const double *b,
                    double *x, int precond, int block_size, double rtol, int max_iters,
                    double *final_resnorm);
/* restarted GMRES(restart), right preconditioned; same stop rule and return value as the CG */
int schwz_or_gmres(int64_t n, const or_idx *rp, const or_idx *col, const double *val, const double *b,
                   double *x, int precond, int block_size, int restart, double rtol, int max_iters,
                   double *final_resnorm);
/* ISAI of a triangular factor on its own pattern (LowerIsai / UpperIsai, solve.cpp:616-638) */
void schwz_or_isai(int64_t n, const or_idx *rp, const or_idx *col, const double *val, int lower, double **w_val);
/* Block boundaries gko::preconditioner::Jacobi detects for max_block_size (source/solve.cpp:490-505):
 * Ginkgo's published find_natural_blocks + agglomerate_supervariables.  ptr: n + 1 entries. */
int64_t schwz_or_jacobi_blocks(int64_t n, const or_idx *rp, const or_idx *col, int max_block_size, int64_t *ptr);

void schwz_or_ilu0(int64_t n, const or_idx *rp, const or_idx *col, const double *val, or_idx **l_rp,
                   or_idx **l_col, double **l_val, or_idx **u_rp, or_idx **u_col, double **u_val);
/* sparse LL^T of A(perm,perm); outputs malloc'd CSR L and U=L^T; perm is
 * old index per new row.  natural=1 => identity ordering, else RCM. */
int schwz_or_cholesky(int64_t n, const or_idx *rp, const or_idx *col,
                      const double *val, int natural, or_idx **l_rp,
                      or_idx **l_col, double **l_val, or_idx **u_rp,
                      or_idx **u_col, double **u_val, or_idx **perm);
/* y = P^T L^-T L^-1 P b (solve.cpp:709-720, solver_tools.hpp:69-87) */
void schwz_or_direct_solve(int64_t n, const or_idx *l_rp, const or_idx *l_col,
                           const double *l_val, const or_idx *u_rp,
                           const or_idx *u_col, const double *u_val,
                           const or_idx *perm, const double *b, double *y,
                           double *work2n);
void schwz_or_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
