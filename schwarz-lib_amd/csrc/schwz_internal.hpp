// Internal declarations shared by the translation units of libschwz_hip.so.
// Not part of the ABI (include/schwz_hip.h is).
#pragma once

#include <hip/hip_runtime_api.h>

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <new>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unordered_map>
#include <vector>

#include "schwz_hip.h"

namespace schwz {

// std::allocator whose construct() default-initialises: resize() of a vector of arithmetic type leaves the new
// elements unwritten instead of zero-filling them on one thread.
template <typename T>
struct NoInitAlloc : std::allocator<T> {
    template <typename U>
    struct rebind {
        using other = NoInitAlloc<U>;
    };
    NoInitAlloc() = default;
    template <typename U>
    NoInitAlloc(const NoInitAlloc<U> &) {}
    template <typename U>
    void construct(U *p) { ::new ((void *)p) U; }
    template <typename U, typename... Args>
    void construct(U *p, Args &&...args) { ::new ((void *)p) U(std::forward<Args>(args)...); }
    // Large blocks (the local matrix: 1.4 GB at 256^3) come 2 MiB-aligned and marked for transparent huge pages:
    // their first touch is then one fault per 2 MiB instead of 512 -- on a kernel in `madvise` mode the setup's
    // page faults were a third of its time (SCHWZ_SETUP_HUGEPAGES=0: plain operator new).
    static bool huge_pages()
    {
        static const bool v = [] {
            const char *e = std::getenv("SCHWZ_SETUP_HUGEPAGES");
            return !(e && e[0] == '0');
        }();
        return v;
    }
    static constexpr size_t kHugeMin = size_t(8) << 20, kHugeAlign = size_t(2) << 20;
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (bytes >= kHugeMin && huge_pages()) {
            void *p = nullptr;
            const size_t padded = (bytes + kHugeAlign - 1) / kHugeAlign * kHugeAlign;
            if (posix_memalign(&p, kHugeAlign, padded) == 0) {
                (void)madvise(p, padded, MADV_HUGEPAGE);
                return static_cast<T *>(p);
            }
            p = std::malloc(bytes);  // (deallocate() frees blocks of this size with std::free)
            if (!p) throw std::bad_alloc();
            return static_cast<T *>(p);
        }
        return static_cast<T *>(::operator new(bytes));
    }
    void deallocate(T *p, size_t n)
    {
        if (n * sizeof(T) >= kHugeMin && huge_pages())
            std::free(p);
        else
            ::operator delete(p);
    }
};

// Threads for the host-side setup loops: SCHWZ_SETUP_THREADS, else the smallest of the CPUs this process may run on,
// its cgroup CPU quota (a GPU box shows all 256 hardware threads of the host to a job that may use 16 of them:
// an OpenMP team of 256 on that quota made every setup stage 2-5 x slower and erratic) and 32 -- divided by the
// ranks of this node when a launcher says how many there are.
inline int setup_threads()
{
    static const int v = [] {
        const char *e = std::getenv("SCHWZ_SETUP_THREADS");
        int n = e ? std::atoi(e) : 0;
        if (n > 0) return n > 256 ? 256 : n;
        cpu_set_t set;
        n = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        auto quota = [](const char *path_quota, const char *path_period) -> double {
            // cgroup v2: "cpu.max" holds "<quota|max> <period>"; v1: two files
            double q = -1.0, per = -1.0;
            if (FILE *f = std::fopen(path_quota, "r")) {
                char word[64] = {0};
                if (path_period == nullptr) {
                    if (std::fscanf(f, "%63s %lf", word, &per) == 2 && word[0] != 'm') q = std::atof(word);
                } else if (std::fscanf(f, "%lf", &q) != 1) {
                    q = -1.0;
                }
                std::fclose(f);
            }
            if (path_period)
                if (FILE *f = std::fopen(path_period, "r")) {
                    if (std::fscanf(f, "%lf", &per) != 1) per = -1.0;
                    std::fclose(f);
                }
            return q > 0.0 && per > 0.0 ? q / per : -1.0;
        };
        double cpus = quota("/sys/fs/cgroup/cpu.max", nullptr);
        if (cpus <= 0.0) cpus = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
        if (cpus > 0.0) n = std::min(n, (int)(cpus + 0.999));
        // one rank per GPU of a node (torch.distributed.run / mpiexec set these): the ranks share those CPUs
        for (const char *var : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS"}) {
            const char *w = std::getenv(var);
            const int ranks = w ? std::atoi(w) : 0;
            if (ranks > 1) {
                n = std::max(1, n / ranks);
                break;
            }
        }
        return n < 1 ? 1 : (n > 32 ? 32 : n);
    }();
    return v;
}

// Host-side setup loops of the .hip translation units (hipcc builds them without an OpenMP runtime): fn(t, nt,
// begin, end) on nt threads over contiguous blocks of [0, n), thread t taking the t-th block.  nt = setup_threads(),
// 1 below min_per_thread items per thread.  Returns nt.
template <typename F>
inline int parallel_blocks(int64_t n, int64_t min_per_thread, F fn)
{
    const int cap = setup_threads();
    int nt = (int)std::min<int64_t>(cap, n / std::max<int64_t>(min_per_thread, 1));
    if (nt < 1) nt = 1;
    if (nt == 1) {
        fn(0, 1, (int64_t)0, n);
        return 1;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)nt - 1);
    for (int t = 1; t < nt; ++t) th.emplace_back([=, &fn] { fn(t, nt, n * t / nt, n * (t + 1) / nt); });
    fn(0, nt, (int64_t)0, n / nt);
    for (auto &x : th) x.join();
    return nt;
}

// SCHWZ_SETUP_TIMING=1: wall time of the setup stages on stderr ("[schwz setup] <stage>: <ms> ms"), what
// tools/setup_probe.py reads.  Zero cost when off.
struct StageTimer {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit StageTimer(const char *n) : name(n), on(false)
    {
        static const bool enabled = [] {
            const char *e = std::getenv("SCHWZ_SETUP_TIMING");
            return e && e[0] == '1';
        }();
        on = enabled;
        if (on) t0 = std::chrono::steady_clock::now();
    }
    void stop()
    {
        if (!on) return;
        on = false;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "[schwz setup] %s: %.1f ms\n", name, ms);
    }
    ~StageTimer() { stop(); }
};

void set_error(const std::string &msg);

#define SCHWZ_HIP_TRY(expr)                                                        \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            ::schwz::set_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + \
                               ": " #expr " -> " + hipGetErrorString(e_));         \
            return SCHWZ_ERR_HIP;                                                  \
        }                                                                          \
    } while (0)

#define SCHWZ_REQUIRE(cond, msg)                      \
    do {                                              \
        if (!(cond)) {                                \
            ::schwz::set_error(std::string(msg));     \
            return SCHWZ_ERR_INVALID;                 \
        }                                             \
    } while (0)

// ---- SpMV tiling constants (see DESIGN.md "CSR SpMV") ------------------------
constexpr int kBlock = 256;      // threads per workgroup = 4 waves of 64
constexpr int kTileNnz = 2048;   // products staged in LDS per tile (16 KiB fp64)
constexpr int kTileRows = 256;   // one lane per row in the reduction phase
constexpr int kMaxGrid = 2048;   // 256 CUs x 8 workgroups, grid-stride beyond
constexpr int kXcds = 8;
constexpr int kGraphIters = 16;          // CG iterations per recorded hipGraph (even: rho slots repeat)
constexpr int64_t kGraphRows = 1 << 21;  // systems up to this many rows replay their CG loop as graphs
constexpr int kDeferDepth = 16;          // search directions kept before x += sum alpha_k p_k is applied
constexpr int kWaveTileNnz = 512;  // products staged per wave (4 KiB fp64)

struct CsrView {
    int64_t nrows = 0, ncols = 0, nnz = 0;
    const schwz_idx *rp = nullptr;
    const schwz_idx *col = nullptr;
    const double *val = nullptr;
    int ntiles = 0;
    const schwz_idx *tile_row = nullptr;  // ntiles+1 row boundaries
    const schwz_idx *tile_order = nullptr;  // optional BFS visiting order (SCHWZ_TILE_ORDER=1)
    // spmv_stream.hip: first nonzero of every tile (rp[tile_row[t]], ntiles+1 entries) and the masked-add count
    // of its row sums (8 / 16 / 32 >= the longest row; 0: the matrix keeps spmv_tiled2_kernel)
    const schwz_idx *tile_nz = nullptr;
    int stream_cap = 0;
    // spmv_stream.hip: per-workgroup partial sums of a launch whose grid exceeds the kMaxGrid slots the consumers
    // fold (2 banks of stream_part_cap doubles; one launch at a time per matrix)
    double *stream_part = nullptr;
    int stream_part_cap = 0;
    // spmv_stream.hip on wide planes: the tile sequence of every XCD, laid out explicitly (entry [xcd * stream_nper + j],
    // -1 past the end) so that a tile's +-plane neighbours are close in the sequence; null: the block-cyclic deal
    const schwz_idx *stream_order = nullptr;
    int stream_nper = 0;
    int xcd_block = 0;  // tiles are dealt to the 8 XCDs block-cyclically in runs of this many (a power of two)
    int xcd_shift = 0;  // log2(xcd_block)
    // kSpmvResidDual: 1 where the tile's rows or columns reach past `dual_split` (where x2 may
    // differ from x); elsewhere the second product is skipped (nullptr: every tile)
    const uint8_t *tile_dual = nullptr;
    int nwtiles = 0;                        // wave tiles: <= 64 rows, <= kWaveTileNnz-2 nnz
    const schwz_idx *wtile_row = nullptr;
    // dictionary-coded copy of (val, col) for the tiles that allow it (spmv_dict.hip):
    // code[j] = value code | delta code << 8 ; per tile a value and a (col - row) dictionary
    const uint16_t *code = nullptr;
    const schwz_idx *vdict_ptr = nullptr;   // ntiles+1 ; vdict_ptr[t] == vdict_ptr[t+1] => raw tile
    const schwz_idx *ddict_ptr = nullptr;   // ntiles+1
    const double *vdict = nullptr;
    const schwz_idx *ddict = nullptr;
    // row-pattern coding (spmv_dict.hip): one byte per row selects a (values, col - row offsets)
    // pattern from a small table shared by many tiles
    const uint8_t *pat_id = nullptr;        // nrows
    const schwz_idx *tile_table = nullptr;  // ntiles: table id, -1 = tile not pattern coded
    const schwz_idx *tbl_desc = nullptr;    // per table: {entry offset, len offset, npat, lmax}
    const uint8_t *tbl_len = nullptr;       // pattern lengths, concatenated
    const double *tbl_val = nullptr;        // [npat][lmax] per table, concatenated
    const schwz_idx *tbl_delta = nullptr;
    // row-pair pattern coding (spmv_pair.hip): one byte per PAIR of adjacent rows selects a merged
    // (col - row, value of row r, value of row r+1, presence bits) sequence
    const uint8_t *pair_id = nullptr;        // (nrows + 1) / 2: pattern of rows (2i, 2i + 1)
    // the same ids run-length coded per chunk: 8 x (first pair of the run | id << 8), 16 bytes a chunk
    // read with one wave-uniform load instead of one byte per lane; first word 0xffff: more than 8
    // runs, the chunk's ids come from pair_id (nullptr: not built)
    const uint4 *pair_rle = nullptr;
    int sweep_gen_mode = 0;  // z-sweep walk over planes that are not whole 512-row chunks: byte ids, partial last band
    int pair_rle_runs = 8;  // runs per chunk record: 8 (16 bytes) or 16 (32 bytes, x lines shorter than ~170 entries)
    const schwz_idx *chunk_ptable = nullptr; // per chunk of 512 rows: pair table id, -1 = not pair coded
    const uint8_t *chunk_dual = nullptr;     // per chunk: the fused dual residual needs its second product
    int pair_shift = 0;                      // log2 of the run length of the XCD deal of the chunks
    int pair_single = 0;                     // 1: every chunk uses table 0
    const schwz_idx *ptbl_desc = nullptr;    // per table: {entry offset, len offset, npat, lmax, max |col - row|}
    const uint8_t *ptbl_len = nullptr;
    const double *ptbl_val = nullptr;        // 2 per entry
    const schwz_idx *ptbl_meta = nullptr;    // 2 per entry: offset, flags
    // symmetric matrices only: for table t, table pair_sym_base + t holds the entries with col >= row,
    // the strictly upper ones doubled: x.(A x) = sum_i x_i (a_ii x_i + 2 sum_{j>i} a_ij x_j) with about
    // half the gathers (kSpmvDotSym); 0 = not built
    int pair_sym_base = 0;
    // single-table matrices whose commonest pattern is a stencil row pair {n2, n1, -1, 0, +1, p1, p2}:
    // that offset layout (pair_canon[0..6]; pair_canon[7] != 0: valid).  Waves whose lanes all have
    // patterns inside the layout gather its 6 outer slots and take the operands of the offset-0 slot
    // from the -1 and +1 gathers.
    int pair_canon[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // z-sweep ("brick") walk of a canonical 3-D stencil layout {-PL, -NX, -1, 0, +1, +NX, +PL} (spmv_pair.hip):
    // a workgroup keeps a band of sweep_T rows and walks it through consecutive planes (stride PL), the
    // window [band - NX, band + T + NX) of each plane staged once in an LDS ring.  sweep_seg: one
    // {band, z0, z1, 0} per workgroup slot (z0 == z1: empty); sweep_gen: the chunks of 512 rows no segment
    // covers (boundary planes of a subdomain, overlap rows), walked the generic way by sweep_gen_blocks more
    // workgroups of the same launch.  sweep_nslots == 0: not built.
    int sweep_T = 0, sweep_nx = 0, sweep_nslots = 0, sweep_ngen = 0, sweep_gen_blocks = 0;
    int64_t sweep_pl = 0;
    const int4 *sweep_seg = nullptr;
    // the fused direction launch's own band height and segment table (spmv_pair_dirdot_sweep_kernel); equal to
    // the update launch's unless SCHWZ_SWEEP_TDIR asks for taller bands
    int sweep_T_dir = 0, sweep_nslots_dir = 0;
    const int4 *sweep_seg_dir = nullptr;
    // ... and the first-direction launch of a solve (one vector read, bound by latency: more, shorter workgroups)
    int sweep_T_first = 0, sweep_nslots_first = 0;
    const int4 *sweep_seg_first = nullptr;
    const schwz_idx *sweep_gen = nullptr;
    // per pattern of table 0: its entries in the nine slots [far before 0, far before 1, -NX, -1, 0, +1, +NX,
    // far after 0, far after 1] (9 PairVal) and the presence mask (bit k: row r has slot k, bit 16 + k: row r + 1)
    const double *canon_val = nullptr;
    const int *canon_mask = nullptr;
    int canon_npat = 0;
    // the same for the upper-triangle twin of the table (symmetric matrices): slots [0, +1, +NX, far after 0,
    // far after 1], 5 PairVal
    const double *canon_sym_val = nullptr;
    const int *canon_sym_mask = nullptr;
    // chains of planes (see build_sweep): plane index per chain position (-1: none) and, per position, which
    // window each far slot reads (2 bits per slot B0, B1, A0, A1: 0 none, 1 previous position, 2 next)
    const int *chain_plane = nullptr;
    const int *chain_far = nullptr;
    // fused dual residual in the walk (pair_set_dual_split): 1 per chain position whose plane holds or couples
    // to rows / columns >= the split (there b - A x2 may differ from b - A x); the chunks of those planes,
    // walked chunk by chunk by a second small launch (kSpmvResidNorm, listed) -- nullptr / 0: not built
    const int *chain_dual = nullptr;
    const schwz_idx *dual_chunks = nullptr;
    int dual_nchunks = 0, dual_blocks = 0;
};

// epilogues of the tiled SpMV kernel
enum SpmvMode {
    kSpmvPlain = 0,      // y = alpha*A*x + beta*y
    kSpmvDot = 1,        // y = A*x ; partial[b] = sum x_i*y_i
    kSpmvResidInit = 2,  // r = b - A*x ; p = dinv*r ; partials sum r*z, sum r*r
    kSpmvResidNorm = 3,  // partial sum (b - A*x)^2, nothing stored
    // kSpmvResidInit on x plus, in the same pass over the matrix, the partial sum
    // (b - A*x2)^2 over rows < row_limit (third partial bank): the convergence-check
    // residual (solve.cpp:834-841) and the CG start residual share one matrix read
    kSpmvResidDual = 4,
    // row-pair kernel only (spmv_pair.hip): the CG iteration without a stored q = A p.
    kSpmvDotOnly = 5,    // partial sum x_i*(A x)_i, nothing stored
    // alpha = rho / fold(p.q partials); per row q_i = (A p)_i recomputed, x_i += alpha p_i,
    // r_i -= alpha q_i, partials r.z and r.r (what cg_update_kernel does, minus 16 B/row of q)
    kSpmvCgUpdate = 6,
    // kSpmvDotOnly for a matrix whose upload found it symmetric: the same number from the upper
    // triangle alone (CsrView::pair_sym_base), about half the gathers
    kSpmvDotSym = 7,
    // the direction update and the next iteration's kSpmvDotSym in one launch: beta from the folded
    // partials, p' = z + beta p computed wherever the upper-triangle gathers need it (own entry and
    // neighbours alike, from r and the OLD p, which this launch only reads), own p' stored to a
    // second buffer, partial sums of p'.(A p'); CgState advanced by workgroup 0
    kSpmvDirDotSym = 8,
    kSpmvDirDotSymVec = 9  // ... with the Jacobi diagonal as a full vector (gathered like r and p)
};

struct SpmvArgs {
    double alpha = 1.0, beta = 0.0;
    const double *x = nullptr;
    const double *x2 = nullptr;  // kSpmvResidDual: second vector (nullptr: same as x)
    double *y = nullptr;        // out vector (y / q / r)
    const double *b = nullptr;  // rhs for the residual modes
    double *p = nullptr;        // kSpmvResidInit: search direction out
    const double *dinv = nullptr;
    double *partials = nullptr;  // [2][grid] ([3][grid] for kSpmvResidDual)
    const int *stop_iter = nullptr;  // device flag checked by CG launches
    int it = 0;
    int64_t row_limit = 0;  // rows >= row_limit are skipped in kSpmvResidNorm
    // kSpmvCgUpdate
    double *cg_x = nullptr, *cg_r = nullptr;
    const struct CgState *cg_state = nullptr;
    const double *pq_partials = nullptr;
    int pq_nparts = 0;
    int diag_mode = 0;          // 0 none, 1 full vector (dinv), 3 uniform scalar
    double diag_uniform = 1.0;
    double cg_rtol = 0.0;       // kSpmvDirDotSym: relative tolerance of the stopping test
    // kSpmvCgUpdate with cg_x == nullptr: x is not touched, alpha is stored here instead (workgroup 0)
    double *alpha_out = nullptr;
    // set by launch_spmv_pair for the companion launch of the z-sweep walk: walk the chunks listed in
    // CsrView::sweep_gen only (2: CsrView::dual_chunks, kSpmvResidNorm); partial sums at
    // [part_offset + blockIdx.x] of banks part_stride apart
    int sweep = 0;
    int part_offset = 0, part_stride = 0;
    // kSpmvResidInit in the z-sweep walk (r stored, p left to the first direction launch), and that first
    // direction launch (kSpmvDirDotSym: p' = D^-1 r, CgState untouched); pcg_begin / pcg_iterate set them
    // where pair_sweep_start_ok holds
    int sweep_init = 0, sweep_first = 0;
    // spmv_stream.hip: consecutive tiles per workgroup of a grid that covers the matrix once (0: persistent
    // workgroups striding through their XCD's sequence, the form of round 2)
    int seq = 0;
    // q-free CG on the walks, a solve that started in the walk (round 3, "virtual first direction"): p0 = D^-1 r0 is
    // never stored -- the first update walk builds its windows from r0 (a.x = r0, windows x ring_scale) and writes
    // r1 to cg_r_out, so r0 stays intact for the first fused direction launch (a.x = r0, p read as p_scale x a.x)
    // and for the x update (slot 0 of the ring = r0 x the same factor).  p0_virtual = 1 marks such a launch: it must
    // be served by the walk kernels (launch_spmv_pair fails loudly otherwise).
    int p0_virtual = 0;
    double ring_scale = 1.0;
    double *cg_r_out = nullptr;
    double p_scale = 1.0;
};

// launch grid of the streaming vector kernels: one lane per element up to kMaxGrid workgroups
inline int grid_for(int64_t n)
{
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > kMaxGrid) g = kMaxGrid;
    return g < 1 ? 1 : (int)g;
}

// small launch helpers shared by the translation units (defined in kernels.hip)
int launch_interface_update(int64_t nrows, int64_t row0, const schwz_idx *rp, const schwz_idx *col,
                            const double *val, const double *x, const double *b, double *bt, hipStream_t s);
int launch_final_norm(const double *partials, int nparts, double *out, hipStream_t s);
int launch_copy(int64_t n, const double *src, double *dst, hipStream_t s);
int launch_gather_f32(int64_t n, const schwz_idx *idx, const double *from, float *into, hipStream_t s);
int launch_scatter_f32(int64_t n, const schwz_idx *idx, const float *from, double *into, hipStream_t s);

int spmv_grid(const CsrView &A, int variant);
int launch_spmv(const CsrView &A, int mode, const SpmvArgs &a, int variant, hipStream_t s);
int launch_spmv_dict(const CsrView &A, int mode, const SpmvArgs &a, int grid, hipStream_t s);
int launch_spmv_pattern(const CsrView &A, int mode, const SpmvArgs &a, int grid, hipStream_t s);
int launch_spmv_pair(const CsrView &A, int mode, const SpmvArgs &a, int grid, hipStream_t s);
int launch_spmv_stream(const CsrView &A, int mode, const SpmvArgs &a, int grid, hipStream_t s, bool *done);
int launch_spmv_stream_ablate(const CsrView &A, const SpmvArgs &a, int abl, hipStream_t s);  // measurement build only
// the measurement variants of schwz_csr_spmv (tools/probes/spmv_variants.hip): null in the product library
typedef int (*SpmvProbeHook)(const CsrView &A, int mode, const SpmvArgs &a, int variant, int grid, hipStream_t s);
extern SpmvProbeHook g_spmv_probe_hook;
bool pair_sweep_start_ok(const CsrView &A, int grid);
bool pair_sweep_dual_ok(const CsrView &A, int grid);

// Jacobi scaling as the CG vector kernels see it.  The full 1/diag vector costs 8 B per row and
// per kernel; matrices with few distinct diagonal values (every stencil) get a 1-byte code per
// row into a <= 256-entry dictionary, or a single scalar.  Same values, same bits.
struct DiagView {
    int mode = 0;  // 0 none, 1 full vector, 2 dictionary codes, 3 uniform scalar
    const double *full = nullptr;
    const uint8_t *code = nullptr;
    const double *dict = nullptr;
    int ndict = 0;
    double uniform = 1.0;
};

// device-side CG scalar state
struct CgState {
    double rho[2];
    double rr;
    double r0;
    int iters;
    int stop_iter;
};

}  // namespace schwz

struct schwz_csr;
namespace schwz {
// builds the dictionary coding on the host and uploads it (no-op + SCHWZ_OK when it does not pay)
int build_spmv_dict(schwz_csr *A, const schwz_idx *h_rp, const schwz_idx *h_col, const double *h_val,
                    const std::vector<schwz_idx> &tiles);
void free_spmv_dict(schwz_csr *A);
int build_spmv_pair(schwz_csr *A, const schwz_idx *h_rp, const schwz_idx *h_col, const double *h_val,
                    const std::vector<schwz_idx> &tiles);
void free_spmv_pair(schwz_csr *A);
int pair_set_dual_split(schwz_csr *A, const schwz_idx *h_rp, const schwz_idx *h_col, int64_t split);
// marks the tiles whose rows or columns reach index >= split (see CsrView::tile_dual)
// host_setup.cpp (OpenMP): row_ptr monotone and every column in [0, ncols)
bool csr_is_well_formed(int64_t nrows, int64_t ncols, const schwz_idx *rp, const schwz_idx *col);
// host_setup.cpp (OpenMP): a_ij == a_ji bit for bit (rows with ascending columns)
bool csr_is_symmetric(int64_t nrows, int64_t ncols, const schwz_idx *rp, const schwz_idx *col, const double *val);
int csr_set_dual_split(schwz_csr *A, const schwz_idx *h_rp, const schwz_idx *h_col, int64_t split);
}  // namespace schwz

struct schwz_pcg;
namespace schwz {
int pcg_begin(schwz_pcg *s, const double *d_b, double *d_x, double rtol, bool fused, const double *d_x2,
              int64_t row_limit, hipStream_t st);
int pcg_iterate(schwz_pcg *s, double *d_x, double rtol, int max_iters, hipStream_t st);
int precond_apply(schwz_pcg *s, const double *in, double *out, hipStream_t st);
int pcg_last_stats(schwz_pcg *s, int *h_iters, double *h_resnorm);  // device sync + state copy
int pcg_take_trs_error(schwz_pcg *s);
int pcg_finish_lazy(schwz_pcg *s);  // the postponed residual update / state advance of the last solve, if any  // ILU(0) sweeps: trs_take_error of the factor solves
}  // namespace schwz

// ---- opaque ABI types -------------------------------------------------------

struct schwz_csr {
    schwz::CsrView v;
    void *d_rp = nullptr, *d_col = nullptr, *d_val = nullptr, *d_tile = nullptr, *d_wtile = nullptr, *d_order = nullptr;
    void *d_tile_nz = nullptr;
    void *d_stream_part = nullptr;
    void *d_stream_order = nullptr;
    void *d_code = nullptr, *d_vptr = nullptr, *d_dptr = nullptr, *d_vdict = nullptr, *d_ddict = nullptr;
    void *d_pat_id = nullptr, *d_tile_table = nullptr, *d_tbl_desc = nullptr, *d_tbl_len = nullptr, *d_tbl_val = nullptr,
         *d_tbl_delta = nullptr;
    void *d_pair_rle = nullptr;
    void *d_sweep_seg_dir = nullptr;
    void *d_sweep_seg_first = nullptr;
    void *d_sweep_seg = nullptr, *d_sweep_gen = nullptr, *d_canon_val = nullptr, *d_canon_mask = nullptr,
         *d_canon_sym_val = nullptr, *d_canon_sym_mask = nullptr, *d_chain_plane = nullptr, *d_chain_far = nullptr;
    void *d_pair_id = nullptr, *d_tile_ptable = nullptr, *d_ptbl_desc = nullptr, *d_ptbl_len = nullptr,
         *d_ptbl_val = nullptr, *d_ptbl_meta = nullptr, *d_chunk_dual = nullptr;
    void *d_tile_dual = nullptr;
    void *d_chain_dual = nullptr, *d_dual_chunks = nullptr;
    std::vector<int> h_chain_plane;  // host copy of CsrView::chain_plane (z-sweep walk built)
    std::vector<schwz_idx> h_tiles;  // host copy of the tile boundaries
    int pair_deal_shift = 0;         // log2 of the run length (in tiles) of the XCD deal the pair kernels derive theirs from
    double dict_fraction = 0.0;  // share of the nonzeros that are dictionary coded
    double pattern_fraction = 0.0;  // share of the nonzeros in row-pattern coded tiles
    double pair_fraction = 0.0;     // share of the nonzeros in row-pair coded tiles
    int64_t pair_code_bytes = 0;    // pattern ids (run-length or byte form) + chunk table ids + tables of the pair coding
};

struct schwz_pcg {
    const schwz_csr *A = nullptr;
    int precond = 0;
    int64_t n = 0;
    double *r = nullptr, *p = nullptr, *q = nullptr, *dinv = nullptr;
    double *r_alt = nullptr;  // second residual buffer of the virtual first direction (cg.hip: r1 goes there, r0 stays)
    // general preconditioners (block-Jacobi, ILU(0)): z = M^-1 r is a vector of its own
    double *z = nullptr;
    int block_size = 1;
    schwz_idx *d_blk_id = nullptr;  // block-Jacobi: index of each block's inverse
    double *d_blk_inv = nullptr;    // unique inverse blocks, [nunique][bs][bs] (a k x k inverse in the top left corner)
    schwz_idx *d_row_blk = nullptr, *d_blk_ptr = nullptr;  // detected blocks of different sizes: block of a row, boundaries
    schwz_trs *ilu = nullptr;       // ILU(0): level-scheduled L and U sweeps
    schwz_csr *isai_l = nullptr, *isai_u = nullptr;  // ISAI: approximate inverses of L and U
    double *isai_tmp = nullptr;
    schwz::DiagView diag;
    void *d_dcode = nullptr, *d_ddict = nullptr;
    double *partials = nullptr;  // 3 * kMaxGrid (SpMV banks) + 2 * kMaxGrid (vector banks)
    double *d_norm_sq = nullptr; // kSpmvResidDual result
    schwz::CgState *state = nullptr;
    schwz::CgState *h_state = nullptr;  // pinned
    hipEvent_t ev[2] = {nullptr, nullptr};
    int variant = 0;
    // recorded runs of kGraphIters iterations (launch-bound small systems), see pcg_iterate
    struct Captured {
        double *x;
        double rtol;
        int variant;
        int qfree;  // 0 stored q, 1 q-free (3 launches), 2 q-free with the fused direction + dot launch
        hipGraphExec_t exec;
    };
    std::vector<Captured> graphs;
    hipStream_t capture_stream = nullptr;
    // deferred x update of large systems (pcg_iterate): ring of search directions and their alphas
    double *p_ring = nullptr;      // kDeferDepth - 2 vectors; slots 0 and 1 of the ring are p and q
    double *alpha_hist = nullptr;  // kDeferDepth
    bool ring_failed = false;
    bool ring_has_q = true;  // slot 1 of the ring is s->q (q-free iteration); false: the stored-q iteration's ring
    // the start launch of the running solve left p to the first fused direction launch (z-sweep start)
    bool p_pending = false;
    // Rows the caller wants final FIRST (a subdomain's boundary rows, which its neighbours wait for): the
    // last x update of a solve takes [0, prio_lo) and [prio_hi, n) ahead of the rest and records prio_event
    // in between, so that the halo pack + send can run beside the remaining update (schwz_ras_pack_early).
    // Without a deferred x update the event is recorded behind the last launch of the solve.
    bool prio_on = false;
    // Second output of the LAST x update of a solve (cg_flush_x_kernel): rows [0, x2_rows) of the result also go
    // to x2_out -- the restriction x~[interior] = y[interior] without a launch of its own.  x2_written: the last
    // solve did write it (deferred x update; otherwise the caller copies).
    double *x2_out = nullptr;
    int64_t x2_rows = 0;
    const double *x2_src = nullptr;  // rows [x2_rows, x2_total) of x2_out are copied from here (overlap / halo of x~)
    int64_t x2_total = 0;
    bool x2_written = false;
    // A solve of exactly max_iters iterations (rtol == 0) whose last iteration was cut down to what its result
    // needs: alpha of that iteration is formed inside the x update, and the residual update + state advance --
    // whose results nothing reads unless the caller asks for the iteration count / residual norm -- wait in
    // `lazy` until pcg_finish_lazy runs them (pcg_last_stats, schwz_pcg_solve with outputs) or the next solve
    // drops them.
    struct LazyLast {
        bool pending = false;
        int it = 0;
        double rtol = 0.0;
        hipStream_t stream = nullptr;
    } lazy;
    std::function<int(hipStream_t)> lazy_run;
    int64_t prio_lo = 0, prio_hi = 0;  // even
    hipEvent_t prio_event = nullptr;
    // how the last solve iterated: bits 0-1: 0 stored q, 1 q-free (three launches), 2 q-free with the fused
    // direction + p.(A p) launch; 4: deferred x update; 8: z-sweep walk of the update launch; 16: of the fused launch;
    // 32: of the start launch and the first direction as well
    int last_flavour = 0;
};

struct schwz_trs {
    int64_t n = 0;
    schwz_idx *l_rp = nullptr, *l_col = nullptr, *u_rp = nullptr, *u_col = nullptr;
    double *l_val = nullptr, *u_val = nullptr;
    schwz_idx *perm = nullptr;
    // level schedules: rows sorted by level, level pointers
    schwz_idx *l_order = nullptr, *l_lvl = nullptr, *u_order = nullptr, *u_lvl = nullptr;
    int l_nlvl = 0, u_nlvl = 0;
    double *w0 = nullptr, *w1 = nullptr;
    // launch plan for factors too large for the one-workgroup kernel: runs of narrow levels
    // (one workgroup, barriers in between) and wide levels (one multi-workgroup launch each)
    struct Seg {
        int lvl0, lvl1;
        bool wide;
    };
    std::vector<Seg> l_plan, u_plan;
    std::vector<schwz_idx> h_l_lvl, h_u_lvl;
    bool fused = true;  // whole solve in the single-workgroup kernel
    // multi-launch plan as hipGraphs, one per (b, y) pair it has been asked for: a sweep over a
    // 256^3 subdomain is ~1500 tiny launches, replayed as ONE graph launch
    struct Captured {
        const double *b;
        double *y;
        hipGraphExec_t exec;
    };
    std::vector<Captured> graphs;
    hipStream_t capture_stream = nullptr;
    bool graphs_failed = false;  // capture / instantiate refused once: launch by launch from then on
    // flag-driven sweeps (trs_flag_kernel): the two solution vectors that double as per-row flags
    bool flags = false;
    unsigned long long *f0 = nullptr, *f1 = nullptr;
    int *d_err = nullptr;
    // the factors in level order (rows at their position in the level-sorted order, columns = positions)
    schwz_idx *fl_rp = nullptr, *fl_col = nullptr, *fu_rp = nullptr, *fu_col = nullptr;
    double *fl_val = nullptr, *fu_val = nullptr;
    schwz_idx *fl_src = nullptr, *fu_src = nullptr, *fu_dst = nullptr;  // rhs gather / y scatter per position
    int flag_grid = 0;
};

// host-side global problem (explicit CSR or analytic stencil)
struct schwz_problem {
    int kind = 0;  // 0 csr, 1 csr rows of a subset of the rows (distributed ingest), 2 lap2d, 3 lap3d
    // kind 1: the global ids of the rows this process holds, ascending; rp / col / val cover those rows in
    // that order.  Asking for another row is an error of the caller: row() returns 0 entries and notes it.
    std::vector<int64_t> present;
    mutable int64_t missing_row = -1;
    int64_t N = 0;
    int64_t nx = 0, ny = 0, nz = 0;
    std::vector<int64_t> rp;
    std::vector<schwz_idx> col;
    std::vector<double> val;
    int64_t nnz() const;
    // writes up to 16 (stencil) or row-length entries; returns count
    int row(int64_t g, int64_t *cols, double *vals) const;
    int max_row_nnz = 0;
};

struct schwz_subdomain {
    // ---- host index sets -------------------------------------------------
    int P = 0, me = 0, overlap = 0;
    int64_t N = 0;
    std::vector<int64_t> first_row;
    int64_t local_size = 0, local_size_x = 0, overlap_size = 0, halo_size = 0;
    std::vector<int64_t, schwz::NoInitAlloc<int64_t>> l2g;  // local_size_x + halo (interior part filled by all threads)
    std::unordered_map<int64_t, schwz_idx> g2l_x;  // non-interior global -> local
    // (col / val: no value-initialisation on resize -- every entry is written by the threads that fill the matrix,
    // which also places the pages near them)
    std::vector<schwz_idx, schwz::NoInitAlloc<schwz_idx>> l_rp;
    std::vector<schwz_idx, schwz::NoInitAlloc<schwz_idx>> l_col;
    std::vector<double, schwz::NoInitAlloc<double>> l_val;
    std::vector<schwz_idx, schwz::NoInitAlloc<schwz_idx>> i_rp;  // interface rows (local_size_x+1)
    std::vector<int64_t> i_col_global;
    std::vector<double> i_val;
    std::vector<int> nbr_in, nbr_out;
    std::vector<std::vector<int64_t>> get, put;  // global ids
    int64_t num_recv = 0, num_send = 0;
    schwz_idx to_local(int64_t g) const
    {
        if (g >= first_row[me] && g < first_row[me + 1]) return (schwz_idx)(g - first_row[me]);
        auto it = g2l_x.find(g);
        return it == g2l_x.end() ? -1 : it->second;
    }
    // ---- device state -----------------------------------------------------
    bool on_device = false;
    schwz_solver_options opt{};
    schwz_csr *A = nullptr;  // local_matrix
    schwz_pcg *cg = nullptr;
    schwz_gmres *gmres = nullptr;  // non-symmetric local matrix
    schwz_trs *trs = nullptr;
    // interface rows: only overlap rows are non-empty; stored compactly
    schwz_idx *d_i_rp = nullptr, *d_i_col = nullptr;  // cols index x~ directly
    double *d_i_val = nullptr;
    int64_t nnz_interface = 0;
    schwz_idx *d_put_idx = nullptr, *d_get_idx = nullptr;  // local ids, packed
    double *d_x = nullptr;       // x~ [interior|overlap|halo]
    double *d_x_alt = nullptr;   // the other x~ buffer: the last x update of a CG solve writes its interior, the restriction swaps the two
    double *d_rhs = nullptr;     // b_loc
    double *d_btilde = nullptr;  // b~ = local_solution on entry of the solve
    double *d_y = nullptr;       // init_guess / solve result
    double *d_partials = nullptr;
    double *h_scalar = nullptr;  // pinned, mapped
    double *d_h_scalar = nullptr;  // device alias of h_scalar
    hipEvent_t ev_scalar = nullptr;
};

namespace schwz {
int trs_take_error(schwz_trs *t);  // trs.hip: SCHWZ_ERR_HIP (and the reason) if a flag-driven sweep timed out
}

namespace schwz {
// host vector -> fresh device allocation; `pad` extra zeroed elements follow the data (the 16-byte
// SpMV loads may touch them)
template <typename T>
inline int upload(const T *h, size_t count, void **d, size_t pad = 0)
{
    *d = nullptr;
    SCHWZ_HIP_TRY(hipMalloc(d, (count + pad ? count + pad : 1) * sizeof(T)));
    if (count) SCHWZ_HIP_TRY(hipMemcpy(*d, h, count * sizeof(T), hipMemcpyHostToDevice));
    if (pad) SCHWZ_HIP_TRY(hipMemset((char *)*d + count * sizeof(T), 0, pad * sizeof(T)));
    return SCHWZ_OK;
}
}  // namespace schwz
