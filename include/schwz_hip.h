/*
 * schwz_hip.h -- C ABI of libschwz_hip.so, the MI355X (gfx950) implementation of
 * the Restricted Additive Schwarz hot path of pratikvn/schwarz-lib.
 *
 * Every entry point names the reference interface it replaces (paths relative
 * to the reference checkout).  The reference itself is a C++ template library
 * over Ginkgo types (no FFI); this header is the boundary a host layer binds:
 * the C++ mirror of SolverRAS/SchwarzBase (schwarz-lib_amd/host/), the Python
 * host used by bench.py/tests (schwarz-lib_amd/schwz_amd/), or the binding
 * sketched in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / Ginkgo types.
 *   - `d_` pointers are device (HBM) pointers, `h_` pointers are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     All device entry points are asynchronous on `stream` unless noted.
 *   - values are fp64 (bench_ras.cpp:204 fixes BenchRas<double,int>), local
 *     indices int32, global row ids int64 (1024^3 rows fit int32, global nnz
 *     does not -- SURVEY F7).
 *   - return value: SCHWZ_OK or an error code; schwz_last_error() holds the
 *     message (the reference throws the Error hierarchy of
 *     include/exception.hpp:42-210; the host mirrors re-throw from the code).
 *   - there is NO CPU fallback: device entry points fail with SCHWZ_ERR_HIP
 *     when no GPU is present.
 */
#ifndef SCHWZ_HIP_H
#define SCHWZ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t schwz_idx;
typedef void *schwz_stream;

enum schwz_status {
    SCHWZ_OK = 0,
    SCHWZ_ERR_INVALID = 1,         /* BadDimension / bad argument */
    SCHWZ_ERR_HIP = 2,             /* CudaError equivalent (exception.hpp:133-152) */
    SCHWZ_ERR_NOT_IMPLEMENTED = 3, /* NotImplemented (exception.hpp:81-96) */
    SCHWZ_ERR_NOT_SPD = 4,         /* factorization failed */
    SCHWZ_ERR_IO = 5,              /* matrix file missing (initialization.cpp:266-271) */
    SCHWZ_ERR_DIVERGED = 6         /* NaN / >1e12 residual (schwarz_base.cpp:424-428) */
};

/* include/collective_common.hpp:36 */
enum schwz_op { SCHWZ_OP_ADD = 0, SCHWZ_OP_COPY = 1, SCHWZ_OP_DIFF = 2, SCHWZ_OP_AVG = 3 };

/* Settings::local_solver_settings (include/settings.hpp) as used by
 * Solve::local_solve (source/solve.cpp:667-792) */
enum schwz_local_solver { SCHWZ_SOLVER_ITERATIVE = 0, SCHWZ_SOLVER_DIRECT = 1 };
/* metadata.local_precond (source/solve.cpp:486-652): "null", "block-jacobi"
 * (gko::preconditioner::Jacobi, :490-505,575-589) and "ilu" (ParIlu + LowerTrs/UpperTrs,
 * :506-532,590-615) and "isai" (Ilu<LowerIsai, UpperIsai>, :616-638) */
enum schwz_precond {
    SCHWZ_PRECOND_NONE = 0,
    SCHWZ_PRECOND_JACOBI = 1,       /* block-jacobi, precond_max_block_size = 1 */
    SCHWZ_PRECOND_BLOCK_JACOBI = 2, /* block-jacobi, consecutive blocks of precond_max_block_size rows */
    SCHWZ_PRECOND_ILU = 3,          /* ilu: ILU(0) + lower/upper triangular sweeps per iteration */
    SCHWZ_PRECOND_ISAI = 4          /* isai: ILU(0) factors replaced by their ISAIs, two SpMVs per application */
};

const char *schwz_last_error(void);
const char *schwz_version(void);
/* threads the host-side setup loops run on (restricted_schwarz.cpp:56-304 is serial in the reference): SCHWZ_SETUP_THREADS,
 * else the smallest of the process's CPUs, its cgroup CPU quota and 32, divided among the ranks of a node when a
 * launcher says how many there are (LOCAL_WORLD_SIZE, OMPI_COMM_WORLD_LOCAL_SIZE, MPI_LOCALNRANKS).  Fixed at first use. */
int schwz_setup_threads(void);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int schwz_device_count(void);
/* device_guard (include/device_guard.hpp:47-101): bind the calling thread */
int schwz_set_device(int device);

/* ------------------------------------------------------------------------ */
/* 1. Stand-alone device kernels                                             */
/* ------------------------------------------------------------------------ */

/* Gather: into[i] = op(into[i], from[idx[i]])
 * replaces gather_kernel + gather_{,add_,diff_,avg_}values
 * (source/gather_kernel.cu:46-109, include/gather.hpp:70-142). */
int schwz_gather(int64_t n, const schwz_idx *d_idx, const double *d_from,
                 double *d_into, int op, schwz_stream stream);

/* Scatter: into[idx[i]] = op(into[idx[i]], from[i])
 * replaces scatter_kernel + launchers (source/scatter_kernel.cu:43-107,
 * include/scatter.hpp:70-143).  avg follows the CPU definition (y+x)/2. */
int schwz_scatter(int64_t n, const schwz_idx *d_idx, const double *d_from,
                  double *d_into, int op, schwz_stream stream);

/* The other instantiations of Gather / Scatter (source/gather_kernel.cu:112-146,
 * source/scatter_kernel.cu:109-142: {float, double, int, long} values x {int, long} indices); the hot
 * path itself only uses fp64 values with int32 indices (schwz_gather / schwz_scatter) and the fp32 wire
 * format of the mixed-precision halo. */
enum schwz_value_type { SCHWZ_VALUE_F32 = 0, SCHWZ_VALUE_F64 = 1, SCHWZ_VALUE_I32 = 2, SCHWZ_VALUE_I64 = 3 };
enum schwz_index_type { SCHWZ_INDEX_I32 = 0, SCHWZ_INDEX_I64 = 1 };
int schwz_gather_typed(int64_t n, const void *d_idx, int index_type, const void *d_from, void *d_into,
                       int value_type, int op, schwz_stream stream);
int schwz_scatter_typed(int64_t n, const void *d_idx, int index_type, const void *d_from, void *d_into,
                        int value_type, int op, schwz_stream stream);

/* CSR matrix resident in HBM, with the row-tile table the SpMV kernel needs.
 * Replaces gko::matrix::Csr<double,int> on the device executor. */
typedef struct schwz_csr schwz_csr;
int schwz_csr_create(int64_t nrows, int64_t ncols, const schwz_idx *h_row_ptr,
                     const schwz_idx *h_col, const double *h_val, schwz_csr **out);
void schwz_csr_destroy(schwz_csr *A);
int64_t schwz_csr_nnz(const schwz_csr *A);
/* encoding the default SpMV (variant 0) uses for this matrix: 0 = plain CSR, 1 = per-entry
 * dictionary tiles, 2 = row-pattern tiles, 3 = row-PAIR pattern tiles (lossless re-encodings
 * built at upload, csrc/spmv_dict.hip and csrc/spmv_pair.hip) */
int schwz_csr_format(const schwz_csr *A);
/* 1 when the upload found a row-pair coded matrix symmetric bit for bit and built the
 * upper-triangle tables the CG iteration takes p.(A p) from (about half the gathers of that
 * launch); such a matrix need not come from the reference's symmetric problems -- nothing is
 * assumed, the check runs on every upload.  0 otherwise. */
int schwz_csr_symmetric(const schwz_csr *A);
/* > 0 when the upload found the matrix to be a 3-D stencil in the canonical row-pair layout
 * {-PL, -NX, -1, 0, +1, +NX, +PL} and prepared the z-sweep walk of the CG update launch (bands of rows
 * swept through consecutive planes, operands from an LDS ring of plane windows; csrc/spmv_pair.hip):
 * the number of workgroup slots of that walk.  0: chunk-by-chunk gathers. */
int schwz_csr_sweep_slots(const schwz_csr *A);
/* chunks of 512 rows that walk leaves to a companion launch of the chunk-by-chunk kernel (rows whose
 * couplings do not fit the plane chains: 0 for a grid in natural order and for the slabs of its z-slab
 * partition, whose appended overlap planes are chained to the interior) */
int schwz_csr_sweep_left_out(const schwz_csr *A);
/* bytes of MATRIX data one SpMV pass reads in the coding `variant` launches (0: the coding the upload
 * chose, schwz_csr_format; anything else: plain CSR = 12 nnz + 4 (rows + 1), the figure of SURVEY 8(d)).
 * bench.py prices its roofline fraction on these bytes (+ the vector bytes of the launch). */
int64_t schwz_csr_matrix_bytes(const schwz_csr *A, int variant);

/* y = alpha*A*x + beta*y : gko Csr::apply(alpha,x,beta,y), call sites
 * source/restricted_schwarz.cpp:1014-1015, source/solve.cpp:834-835,1079-1080.
 * variant: 0 = default: the best coded tiles the matrix allows (row pairs, row patterns,
 * per-entry dictionaries; lossless, bit-identical results; see csrc/spmv_pair.hip,
 * csrc/spmv_dict.hip), else 6; 8 = row-pattern tiles, 7 = per-entry dictionary tiles,
 * 6 = plain CSR, LDS-staged row tiles with 16-byte batched loads,
 * 1 = one-row-per-lane baseline, 2 = first tiled version (scalar loads), 3/5 = wave-private
 * tiles (5: non-temporal matrix loads), 4 = software-pipelined tiles; 10.. = ablation builds
 * for tools/spmv_probe.py (deliberately wrong results). */
int schwz_csr_spmv(const schwz_csr *A, double alpha, const double *d_x,
                   double beta, double *d_y, int variant, schwz_stream stream);

/* Preconditioned CG on A x = b, device resident (no per-iteration host sync).
 * Replaces gko::solver::Cg + stop::Combined(Iteration, ResidualNormReduction)
 * as configured in source/solve.cpp:456-478,571-652 and applied by
 * SolverTools::solve_iterative_ginkgo (include/solver_tools.hpp:91-98).
 * x is the warm start on entry.  rtol<=0 runs exactly max_iters updates.
 * h_iters / h_resnorm (may be NULL) are written after an internal stream
 * sync only when requested. */
typedef struct schwz_pcg schwz_pcg;
int schwz_pcg_create(const schwz_csr *A, int precond, schwz_pcg **out);
/* the same with the block size of SCHWZ_PRECOND_BLOCK_JACOBI (metadata.precond_max_block_size,
 * 1..32) */
int schwz_pcg_create_ex(const schwz_csr *A, int precond, int block_size, schwz_pcg **out);
void schwz_pcg_destroy(schwz_pcg *s);
/* how the last solve iterated (bench.py needs it to name and price its launches): bits 0-1: 0 = q = A p
 * stored, 1 = q-free, three launches per iteration, 2 = q-free with the direction update fused into the
 * next iteration's p.(A p) launch (two launches); 4: x += sum alpha_k p_k deferred; 8 / 16: the update /
 * the fused launch walked the matrix in z-sweeps (csrc/spmv_pair.hip); 32: so did the start residual and
 * the first direction of the solve */
int schwz_pcg_flavour(const schwz_pcg *s);
int schwz_pcg_solve(schwz_pcg *s, const double *d_b, double *d_x, double rtol,
                    int max_iters, int *h_iters, double *h_resnorm,
                    schwz_stream stream);

/* Restarted GMRES(restart) with right preconditioning, device resident like the CG.
 * Replaces gko::solver::Gmres with krylov_dim = settings.restart_iter and the same
 * Combined(Iteration, ResidualNormReduction) criterion, the local solver of
 * settings.non_symmetric_matrix (source/solve.cpp:486-520, applied at :750-753).
 * max_iters counts Krylov vectors over all cycles; h_resnorm is the residual norm the
 * stop test saw (the rotated right-hand side inside a cycle). */
typedef struct schwz_gmres schwz_gmres;
int schwz_gmres_create(const schwz_csr *A, int precond, int block_size, int restart, schwz_gmres **out);
void schwz_gmres_destroy(schwz_gmres *s);
int schwz_gmres_solve(schwz_gmres *s, const double *d_b, double *d_x, double rtol, int max_iters,
                      int *h_iters, double *h_resnorm, schwz_stream stream);
/* iterations and residual norm of the last solve (device synchronisation) */
int schwz_gmres_last_stats(schwz_gmres *s, int *h_iters, double *h_resnorm);

/* Profiling hooks for bench.py's roofline leg: between begin and end, every
 * launch of the dominant kernel (the CSR SpMV inside schwz_pcg_solve) is
 * bracketed by a HIP event pair on its launch stream; end synchronises and
 * returns the summed kernel time and the launch count.  Replaces nothing in
 * the reference (its MEASURE_ELAPSED_FUNC_TIME, include/settings.hpp:508-523,
 * times host calls without a device sync). */
int schwz_profile_begin(int capacity);
/* STREAM-style probe for the measured HBM ceiling quoted beside the 8 TB/s spec
 * (SURVEY 8d): mode 0 copies n doubles src->dst in a grid-stride loop of 2048
 * persistent workgroups, mode 1 reads n doubles and writes one partial sum per
 * workgroup (dst needs 2048 doubles), mode 2 copies with one 16-byte element
 * per thread and a grid that covers the buffer once (the form that reaches the
 * ~6.2 TB/s copy ceiling), mode 3 / 4 four elements per thread, plain /
 * non-temporal. */
int schwz_stream_probe(int64_t n, int mode, const double *d_src, double *d_dst,
                       schwz_stream stream);
int schwz_profile_end(double *h_total_ms, int64_t *h_launches);
/* after schwz_profile_end: the same figures per launch kind of a CG iteration, 0 = the SpMV (+ p.Ap)
 * launch, 1 = the fused recompute-and-update launch of the q-free iteration (zero launches when the
 * solver stores q) */
int schwz_profile_kind(int kind, double *h_total_ms, int64_t *h_launches);

/* Sparse triangular solves y = P^T L^-T L^-1 P b.
 * Replaces gko::solver::LowerTrs/UpperTrs + Permutation::apply as used by
 * Solve::local_solve (source/solve.cpp:709-720) and
 * SolverTools::solve_direct_ginkgo (include/solver_tools.hpp:69-87); also the
 * intent of the dead CusparseWrappers (include/cusparse_helpers.hpp:200-212).
 * L: CSR lower, diagonal last in each row; U: CSR upper, diagonal first (U = L^T for the
 * Cholesky path; any L, U pair for ILU).  h_perm may be NULL (identity).  Factors whose level
 * structure does not fit one workgroup are solved level by level with multi-workgroup launches. */
typedef struct schwz_trs schwz_trs;
int schwz_trs_create(int64_t n, const schwz_idx *h_l_rp, const schwz_idx *h_l_col,
                     const double *h_l_val, const schwz_idx *h_u_rp,
                     const schwz_idx *h_u_col, const double *h_u_val,
                     const schwz_idx *h_perm, schwz_trs **out);
void schwz_trs_destroy(schwz_trs *t);
int schwz_trs_solve(schwz_trs *t, const double *d_b, double *d_y, schwz_stream stream);

/* ------------------------------------------------------------------------ */
/* 2. Host-side setup (no GPU needed)                                        */
/* ------------------------------------------------------------------------ */

/* Global problem description: either an explicit CSR or an analytic stencil
 * whose rows are produced on demand, so that no rank ever holds a global
 * matrix (the reference replicates it, source/schwarz_base.cpp:142-147). */
typedef struct schwz_problem schwz_problem;

/* Initialize::setup_global_matrix, Laplacian branch
 * (source/initialization.cpp:214-265): dim=2 is the reference's 5-point
 * stencil on an nx*nx grid; dim=3 is the 7-point extension on nx*ny*nz. */
int schwz_problem_laplacian(int dim, int64_t nx, int64_t ny, int64_t nz,
                            schwz_problem **out);
/* explicit CSR with int64 row_ptr (copied) */
int schwz_problem_from_csr(int64_t N, const int64_t *h_row_ptr,
                           const schwz_idx *h_col, const double *h_val,
                           schwz_problem **out);
/* Distributed ingest: the reference reads the whole file on every rank and keeps the whole matrix there
 * (initialization.cpp:204-213).  Here one rank may parse and partition, cut out what every
 * subdomain reads -- its interior and overlap rows, schwz_problem_extract_rows -- and hand each rank a row
 * source that holds only that part (row ids ascending, CSR over those rows, columns global and ascending);
 * schwz_subdomain_setup on it gives the same subdomain as on the whole matrix and fails if it needs a row
 * that is not there.  extract_rows: first call with h_col == NULL fills h_row_ptr (nrows + 1 entries). */
int schwz_problem_from_rows(int64_t N, int64_t nrows, const int64_t *h_row_ids, const int64_t *h_row_ptr,
                            const schwz_idx *h_col, const double *h_val, schwz_problem **out);
int schwz_problem_extract_rows(const schwz_problem *p, int64_t nrows, const int64_t *h_row_ids,
                               int64_t *h_row_ptr, schwz_idx *h_col, double *h_val);
/* Matrix-Market file branch (initialization.cpp:204-213): gko::read +
 * sort_by_column_index */
int schwz_problem_from_matrix_market(const char *path, schwz_problem **out);
void schwz_problem_destroy(schwz_problem *p);
int64_t schwz_problem_size(const schwz_problem *p);
int64_t schwz_problem_nnz(const schwz_problem *p);
/* copy one row out (count returned through *n; cols ascending) */
int schwz_problem_row(const schwz_problem *p, int64_t row, int *n,
                      int64_t *cols, double *vals, int capacity);
/* Apply a partition vector: A <- A(perm,perm) with perm grouping rows by
 * part id, stable within a part (source/restricted_schwarz.cpp:105-152).
 * h_perm (new->old, length N) and first_row (P+1) are outputs. */
int schwz_problem_permute(const schwz_problem *p, int P, const uint32_t *h_part,
                          int64_t *h_perm, int64_t *h_first_row,
                          schwz_problem **out);

/* Initialize::generate_rhs (source/initialization.cpp:88-96): entry g of the sequence
 * std::uniform_real_distribution<double>(0,1) draws from a default-seeded
 * std::default_random_engine (libstdc++: minstd_rand0, two draws per value).  Random access by
 * global row id (LCG jump-ahead), so no rank generates or broadcasts the whole vector
 * (source/schwarz_base.cpp:169-182 does). */
int schwz_rhs_random(int64_t count, const int64_t *h_global_ids, double *h_out);
/* contiguous row blocks (source/restricted_schwarz.cpp:84,97-102) */
int schwz_partition_regular(int64_t N, int P, int64_t *h_first_row);
/* PartitionTools::PartitionRegular2D (include/partition_tools.hpp:70-106) */
int schwz_partition_regular2d(int64_t n1d, int P, uint32_t *h_part);
/* graph partition standing in for PartitionTools::PartitionMetis
 * (include/partition_tools.hpp:110-202; METIS itself is absent): multilevel
 * recursive bisection of the (symmetrised) matrix graph -- heavy-edge
 * coarsening, greedy growing on the coarsest graph, Fiduccia-Mattheyses
 * refinement on every level -- to part sizes that differ by at most one row.
 * Deterministic.  SCHWZ_PART_MULTILEVEL=0: the single-level bisection of
 * rounds 1-2. */
int schwz_partition_graph(const schwz_problem *p, int P, uint32_t *h_part);

/* Subdomain: index sets, local/interface matrices, comm lists, device state.
 * SolverRAS::setup_local_matrices (source/restricted_schwarz.cpp:56-304) and
 * setup_comm_buffers (:308-604). */
typedef struct schwz_subdomain schwz_subdomain;

int schwz_subdomain_setup(const schwz_problem *p, int P, int me, int overlap,
                          const int64_t *h_first_row, schwz_subdomain **out);
void schwz_subdomain_destroy(schwz_subdomain *sd);

/* sizes[0]=local_size [1]=local_size_x [2]=overlap_size [3]=halo_size
 * [4]=nnz_local [5]=nnz_interface [6]=num_neighbors_in [7]=num_neighbors_out
 * [8]=num_recv [9]=num_send  (Metadata fields, include/settings.hpp:318-496) */
int schwz_subdomain_sizes(const schwz_subdomain *sd, int64_t *sizes10);
/* local id -> global id, length local_size_x + halo_size */
int schwz_subdomain_local_to_global(const schwz_subdomain *sd, int64_t *h_out);
/* local_matrix / interface_matrix on the host (interface columns are
 * returned as GLOBAL ids, like the reference stores them) */
int schwz_subdomain_local_matrix(const schwz_subdomain *sd, schwz_idx *h_rp,
                                 schwz_idx *h_col, double *h_val);
int schwz_subdomain_interface_matrix(const schwz_subdomain *sd, schwz_idx *h_rp,
                                     int64_t *h_col_global, double *h_val);
/* comm_struct.neighbors_in / global_get (include/communicate.hpp:67-224):
 * k-th in-neighbour; ids ascending global */
int schwz_subdomain_get_list(const schwz_subdomain *sd, int k, int *rank,
                             int64_t *count, int64_t *h_ids /* may be NULL */);
/* the index handshake of restricted_schwarz.cpp:400-472: neighbour p's get
 * list for me is my put list for p.  Add in ascending p. */
int schwz_subdomain_add_put_list(schwz_subdomain *sd, int p, int64_t count,
                                 const int64_t *h_ids);
int schwz_subdomain_put_list(const schwz_subdomain *sd, int k, int *rank,
                             int64_t *count, int64_t *h_ids /* may be NULL */);
/* offsets of neighbour k inside the packed send / recv buffers (prefix sums in
 * neighbour order, restricted_schwarz.cpp:870,909,913,940) */
int schwz_subdomain_send_offset(const schwz_subdomain *sd, int k, int64_t *offset);
int schwz_subdomain_recv_offset(const schwz_subdomain *sd, int k, int64_t *offset);

/* Host sparse LL^T of the local matrix, standing in for CHOLMOD simplicial
 * LL^T (Solve::compute_local_factors, source/solve.cpp:75-143): identity
 * A(perm,perm) = L L^T.  Outputs are malloc'd; free with schwz_free. */
int schwz_cholesky(int64_t n, const schwz_idx *h_rp, const schwz_idx *h_col,
                   const double *h_val, int natural_ordering, schwz_idx **l_rp,
                   schwz_idx **l_col, double **l_val, schwz_idx **u_rp,
                   schwz_idx **u_col, double **u_val, schwz_idx **perm);
/* Host ILU(0) on the pattern of A (columns sorted): L unit lower with the 1 stored LAST in each
 * row, U upper with its diagonal FIRST -- the layout schwz_trs_create expects.  Stands in for
 * gko::factorization::ParIlu (source/solve.cpp:506-532), whose sweeps converge to these
 * factors.  Outputs are malloc'd; free with schwz_free. */
int schwz_ilu0(int64_t n, const schwz_idx *h_rp, const schwz_idx *h_col, const double *h_val,
               schwz_idx **l_rp, schwz_idx **l_col, double **l_val, schwz_idx **u_rp,
               schwz_idx **u_col, double **u_val);
/* Incomplete sparse approximate inverse of a triangular CSR factor on the factor's own pattern:
 * row i of W solves W(i,S) T(S,S) = e_i(S), S = pattern of row i of T.  Replaces
 * gko::preconditioner::LowerIsai / UpperIsai with sparsity power 1 (source/solve.cpp:616-638).
 * *w_val (nnz(T) values on T's pattern) is malloc'd; free with schwz_free. */
int schwz_isai(int64_t n, const schwz_idx *h_rp, const schwz_idx *h_col, const double *h_val, int lower,
               double **w_val);
void schwz_free(void *p);

/* ------------------------------------------------------------------------ */
/* 3. Device-resident RAS iteration of one subdomain                         */
/* ------------------------------------------------------------------------ */

typedef struct {
    int32_t local_solver;    /* schwz_local_solver */
    int32_t precond;         /* schwz_precond */
    double local_tol;        /* metadata.local_solver_tolerance */
    int32_t local_max_iters; /* -1 => local_size_x (solve.cpp:458-463) */
    int32_t natural_factor_ordering; /* settings.naturally_ordered_factor */
    int32_t spmv_variant;    /* 0 default (the best coding the matrix allows); see schwz_csr_spmv */
    int32_t precond_block_size; /* metadata.precond_max_block_size (block-jacobi) */
    int32_t non_symmetric;   /* settings.non_symmetric_matrix: GMRES instead of CG */
    int32_t restart_iter;    /* settings.restart_iter (GMRES krylov_dim), >= 1 */
} schwz_solver_options;

/* Upload matrices / index lists, allocate x~=[interior|overlap|halo] (zero,
 * SURVEY F9), local rhs, CG work vectors; factor on the host for the direct
 * path.  h_local_rhs has local_size_x entries: [rhs[interior]; rhs[overlap_row]]
 * (Initialize::setup_vectors, source/initialization.cpp:332-359).
 * Requires all put lists to be set.  Binds to the current HIP device. */
int schwz_subdomain_to_device(schwz_subdomain *sd, const double *h_local_rhs,
                              const schwz_solver_options *opt);

/* step 0a: send[off_k + i] = x~[put_k[i]] for every out-neighbour
 * (exchange_boundary_twosided pack half, restricted_schwarz.cpp:884-911;
 * CommHelpers::pack_buffer, include/comm_helpers.hpp:93-118).
 * d_send has num_send entries. */
int schwz_ras_pack(schwz_subdomain *sd, double *d_send, schwz_stream stream);
/* step 0b: x~[get_k[i]] = recv[off_k + i] for every in-neighbour
 * (restricted_schwarz.cpp:950-962; CommHelpers::unpack_buffer,
 * include/comm_helpers.hpp:154-177).  d_recv has num_recv entries. */
int schwz_ras_unpack(schwz_subdomain *sd, const double *d_recv, schwz_stream stream);
/* Mixed-precision variants of step 0 (settings.use_mixed_precision with MixedValueType =
 * float): the packed halo travels as fp32 -- the reference converts the fp64 send buffer with
 * Dense::convert_to before MPI_Isend and back after the receive
 * (restricted_schwarz.cpp:898-903, 929-933, 952-954).  Half the xGMI bytes. */
int schwz_ras_pack_f32(schwz_subdomain *sd, float *d_send, schwz_stream stream);
int schwz_ras_unpack_f32(schwz_subdomain *sd, const float *d_recv, schwz_stream stream);
/* One neighbour's share of step 0, to / from any device address: the one-sided exchange without a
 * matched receive.  "put" (CommHelpers::transfer_buffer with enable_put, include/comm_helpers.hpp:122-150
 * after pack_buffer :93-118): the pack kernel of out-neighbour k stores straight into that neighbour's
 * receive window (mapped with schwz_window_open), over xGMI when it lives on another GPU.  "get": the
 * unpack kernel of in-neighbour k loads from that neighbour's send window.  single != 0: fp32 wire
 * format (MixedValueType = float). */
/* The pack of the next exchange while the local solve is still finishing: the solve makes the rows of
 * the put lists final first and records an event; this call makes `stream` (a side stream) wait for it
 * and gathers the send buffer from the solve's result (double, or float with `single`) -- the values
 * schwz_ras_pack reads after schwz_ras_restrict.  The reference posts its MPI_Isend only after the local
 * solve and the restriction have finished (restricted_schwarz.cpp:884-943 inside schwarz_base.cpp:387-452);
 * here the RCCL send / recv runs beside the tail of the solve.  schwz_ras_early_pack_ok: 1 when the
 * subdomain's solver (CG) records that event. */
int schwz_ras_early_pack_ok(const schwz_subdomain *sd);
int schwz_ras_pack_early(schwz_subdomain *sd, void *d_send, int single, schwz_stream stream);
int schwz_ras_pack_neighbor(schwz_subdomain *sd, int k, void *d_dst, int single, schwz_stream stream);
int schwz_ras_unpack_neighbor(schwz_subdomain *sd, int k, const void *d_src, int single, schwz_stream stream);
/* Windows: device buffers that other rank processes of the node map into their address space -- the
 * MPI_Win_create / MPI_Win_lock_all of Communicate::setup_windows (source/communicate.cpp,
 * include/communicate.hpp:67-224) over HIP IPC.  alloc: a zeroed allocation of its own; export: its
 * 64-byte handle (send it to the peers by any host channel); open / close: map / unmap a peer's window. */
int schwz_window_alloc(int64_t bytes, void **d_ptr);
int schwz_window_free(void *d_ptr);
int schwz_window_export(void *d_ptr, unsigned char *h_handle64);
int schwz_window_open(const unsigned char *h_handle64, void **d_ptr);
int schwz_window_close(void *d_ptr);
/* Host-side windows (window_convergence, window_residual_vector of include/conv_tools.hpp:56-275) are
 * shared-memory segments mapped by every rank of the node; these are the remote update operations on
 * them: MPI_Accumulate(MPI_SUM) on an int, MPI_Put / local read of an int, MPI_Accumulate(MPI_MIN) on a
 * double. */
int32_t schwz_host_atomic_add_i32(int32_t *p, int32_t v);
int32_t schwz_host_atomic_load_i32(const int32_t *p);
void schwz_host_atomic_store_i32(int32_t *p, int32_t v);
double schwz_host_atomic_min_f64(double *p, double v);
/* step 1: b~ = b_loc - A_Gamma x~ (SolverRAS::update_boundary,
 * restricted_schwarz.cpp:992-1017) */
int schwz_ras_update_boundary(schwz_subdomain *sd, schwz_stream stream);
/* step 2 (local half): rho = || b~ - A_loc [x~_int; x~_ovl] ||_2
 * (Solve::check_local_convergence, source/solve.cpp:796-856).  The norm is
 * written to *h_resnorm after a stream sync (the reference copies the scalar
 * to the host at solve.cpp:842-843). */
int schwz_ras_local_residual(schwz_subdomain *sd, double *h_resnorm, schwz_stream stream);
/* The same in two halves: launch enqueues the kernels and the scalar copy and
 * records an event; wait blocks on that event only.  A host layer may enqueue
 * the local solve between the two, so the GPU never idles while the host reads
 * the norm and runs the global check (schwz_ras_local_solve does not touch x~,
 * so a converged verdict simply skips step 4). */
int schwz_ras_local_residual_launch(schwz_subdomain *sd, schwz_stream stream);
int schwz_ras_local_residual_wait(schwz_subdomain *sd, double *h_resnorm);
/* step 3: y = solve(A_loc, b~), warm-started (Solve::local_solve,
 * source/solve.cpp:667-792).  h_inner_iters may be NULL (no sync). */
int schwz_ras_local_solve(schwz_subdomain *sd, int *h_inner_iters, schwz_stream stream);
/* Two-stage local solves: Solve::local_solve rebuilds the stopping criterion with a new
 * iteration cap once iter_count > settings.reset_local_crit_iter (source/solve.cpp:723-742);
 * max_iters = -1 means local_size_x.  Takes effect from the next local solve. */
int schwz_ras_set_local_max_iters(schwz_subdomain *sd, int max_iters);
/* settings.enable_logging (source/solve.cpp:751-771): inner iterations and final residual norm of the
 * last local solve (0 and 0.0 for the direct path).  Synchronises the solver's stream. */
int schwz_ras_last_inner_stats(schwz_subdomain *sd, int *h_iters, double *h_resnorm);
/* steps 2+3 in one enqueue: the check residual of step 2 and the start residual of
 * the CG solve of step 3 come out of ONE pass over A_loc (two gathers per entry,
 * one matrix read), the norm's device->host copy is queued right behind it and
 * the CG iterations behind that.  Follow with schwz_ras_local_residual_wait. */
int schwz_ras_check_and_solve_launch(schwz_subdomain *sd, schwz_stream stream);
/* The input of a DEVICE-side all-gather of the residual norms (the MPI_Allgather of source/solve.cpp:890-891 as
 * an RCCL collective): makes `stream` wait for the norm of the last schwz_ras_local_residual_launch /
 * schwz_ras_check_and_solve_launch -- for that scalar only, not for the local solve enqueued behind the check --
 * and writes its SQUARE to d_norm_sq[0] (device memory) on that stream.  The host value stays available through
 * schwz_ras_local_residual_wait. */
int schwz_ras_norm_sq_to_device(schwz_subdomain *sd, double *d_norm_sq, schwz_stream stream);
/* step 4: x~[interior] = y[0:local_size] (Communicate::local_to_global_vector,
 * source/communicate.cpp:65-94) */
int schwz_ras_restrict(schwz_subdomain *sd, schwz_stream stream);

/* device pointers of the state vectors (for tests and host layers):
 * which: 0 = x~ (local_size_x+halo), 1 = b~ / local_solution (local_size_x),
 *        2 = y / init_guess (local_size_x), 3 = local_rhs (local_size_x) */
int schwz_ras_vector(schwz_subdomain *sd, int which, double **d_ptr, int64_t *len);
/* the HBM-resident local_matrix of the subdomain (borrowed handle, owned by sd) */
int schwz_ras_local_csr(schwz_subdomain *sd, schwz_csr **out);
/* how the scalar-Jacobi diagonal of the local CG reaches its vector kernels: 0 no Jacobi scaling,
 * 1 a full 1/diag vector (8 B per row and launch), 2 one-byte codes into a small dictionary, 3 one
 * scalar (every stencil matrix).  Same values, same bits; bench.py needs it to count the bytes of a
 * launch.  Replaces nothing in the reference (gko::preconditioner::Jacobi stores blocks). */
int schwz_ras_jacobi_form(const schwz_subdomain *sd);
/* schwz_pcg_flavour of the subdomain's local CG (0 for the other local solvers) */
int schwz_ras_cg_flavour(const schwz_subdomain *sd);
/* copy x~[0:local_size] to the host (synchronous) -- the rank's piece of the
 * solution assembled in Solve::compute_residual_norm (solve.cpp:1025-1085) */
int schwz_ras_get_interior(schwz_subdomain *sd, double *h_out, schwz_stream stream);
/* || b_int - (A x)_int ||^2 contribution of this subdomain's interior rows to
 * the final true residual (solve.cpp:1079-1084) given up-to-date halo values
 * in x~; returns the squared partial norm on the host (synchronous). */
int schwz_ras_true_residual_sq(schwz_subdomain *sd, double *h_out, schwz_stream stream);

/* HBM bytes one launch of the dominant kernels moves algorithmically
 * (SURVEY 8d): which: 0 = local SpMV, 1 = one PCG iteration, used by bench.py */
int64_t schwz_ras_algorithmic_bytes(const schwz_subdomain *sd, int which);

#ifdef __cplusplus
}
#endif
#endif
