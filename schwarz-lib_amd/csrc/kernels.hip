// libschwz_hip.so, general part: gather / scatter (the four reference ops and the fp32 wire
// variants), the interface (boundary) update, copies and STREAM-style probes, and the C ABI of the
// device CSR matrix (upload, tiling, XCD deal, the lossless codings of spmv_pair.hip / spmv_dict.hip).
// The SpMV kernels live in spmv_csr.hip / spmv_dict.hip / spmv_pair.hip, the solvers in cg.hip /
// gmres.hip / trs.hip, the per-subdomain RAS steps in subdomain.hip.
//
// Everything is HBM-bandwidth bound fp64 / int32 streaming work; there is no dense contraction, so
// MFMA is deliberately unused (BASELINE.json north_star).  Wavefronts are 64 lanes; workgroups are
// 256 threads (4 waves).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "schwz_internal.hpp"
#include "device_utils.hpp"

namespace schwz {

// ---------------------------------------------------------------------------
// gather / scatter with the four reference ops
// ---------------------------------------------------------------------------

template <int OP>
__device__ __forceinline__ double combine(double into, double from)
{
    if (OP == SCHWZ_OP_COPY) return from;
    if (OP == SCHWZ_OP_ADD) return from + into;
    if (OP == SCHWZ_OP_DIFF) return from - into;
    return (from + into) / 2;
}

template <int OP>
__global__ void gather_kernel(int64_t n, const schwz_idx *__restrict__ idx,
                              const double *__restrict__ from, double *__restrict__ into)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        into[i] = combine<OP>(OP == SCHWZ_OP_COPY ? 0.0 : into[i], from[idx[i]]);
}

template <int OP>
__global__ void scatter_kernel(int64_t n, const schwz_idx *__restrict__ idx,
                               const double *__restrict__ from, double *__restrict__ into)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const schwz_idx k = idx[i];
        into[k] = combine<OP>(OP == SCHWZ_OP_COPY ? 0.0 : into[k], from[i]);
    }
}

// The reference instantiates its Gather / Scatter for {float, double, int, long} values and {int, long}
// indices (source/gather_kernel.cu:112-146, scatter_kernel.cu:109-142): the same here, one grid-stride
// kernel per (value, index, op) behind schwz_gather_typed / schwz_scatter_typed.  avg follows the CPU
// definition (y + x) / 2 in the value type (integer division for the integer types).
template <typename V, int OP>
__device__ __forceinline__ V combine_t(V into, V from)
{
    if (OP == SCHWZ_OP_COPY) return from;
    if (OP == SCHWZ_OP_ADD) return from + into;
    if (OP == SCHWZ_OP_DIFF) return from - into;
    return (from + into) / 2;
}

template <typename V, typename I, int OP>
__global__ void gather_typed_kernel(int64_t n, const I *__restrict__ idx, const V *__restrict__ from, V *__restrict__ into)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        into[i] = combine_t<V, OP>(OP == SCHWZ_OP_COPY ? V(0) : into[i], from[idx[i]]);
}

template <typename V, typename I, int OP>
__global__ void scatter_typed_kernel(int64_t n, const I *__restrict__ idx, const V *__restrict__ from, V *__restrict__ into)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const I k = idx[i];
        into[k] = combine_t<V, OP>(OP == SCHWZ_OP_COPY ? V(0) : into[k], from[i]);
    }
}

template <typename V, typename I>
int launch_gs_typed(bool scatter, int64_t n, const void *idx, const void *from, void *into, int op, hipStream_t st)
{
    const int g = grid_for(n);
#define SCHWZ_GS(OP_)                                                                                                   \
    if (scatter)                                                                                                        \
        hipLaunchKernelGGL((scatter_typed_kernel<V, I, OP_>), dim3(g), dim3(kBlock), 0, st, n, (const I *)idx,           \
                           (const V *)from, (V *)into);                                                                 \
    else                                                                                                                \
        hipLaunchKernelGGL((gather_typed_kernel<V, I, OP_>), dim3(g), dim3(kBlock), 0, st, n, (const I *)idx,            \
                           (const V *)from, (V *)into);
    switch (op) {
    case SCHWZ_OP_COPY: SCHWZ_GS(SCHWZ_OP_COPY) break;
    case SCHWZ_OP_ADD: SCHWZ_GS(SCHWZ_OP_ADD) break;
    case SCHWZ_OP_DIFF: SCHWZ_GS(SCHWZ_OP_DIFF) break;
    case SCHWZ_OP_AVG: SCHWZ_GS(SCHWZ_OP_AVG) break;
    default: set_error(scatter ? "Undefined scatter operation" : "Undefined gather operation"); return SCHWZ_ERR_INVALID;
    }
#undef SCHWZ_GS
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

// halo pack / unpack with the fp64 <-> fp32 conversion of the mixed-precision exchange
__global__ void gather_f32_kernel(int64_t n, const schwz_idx *__restrict__ idx, const double *__restrict__ from,
                                  float *__restrict__ into)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        into[i] = (float)from[idx[i]];
}

__global__ void scatter_f32_kernel(int64_t n, const schwz_idx *__restrict__ idx, const float *__restrict__ from,
                                   double *__restrict__ into)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        into[idx[i]] = (double)from[i];
}


// b~[row] = b[row] - sum_j A_Gamma[row][j] x~[col_j] for the overlap rows only
// (rows < local_size have no interface entries; their b~ is set once at upload).
__global__ void interface_update_kernel(int64_t nrows, int64_t row0, const schwz_idx *__restrict__ rp,
                                        const schwz_idx *__restrict__ col,
                                        const double *__restrict__ val,
                                        const double *__restrict__ x, const double *__restrict__ b,
                                        double *__restrict__ bt)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += stride) {
        double s = 0.0;
        for (int j = rp[i]; j < rp[i + 1]; ++j) s += val[j] * x[col[j]];
        bt[row0 + i] = b[row0 + i] - s;
    }
}


// STREAM-style probes: the measured copy / read ceiling quoted next to the 8 TB/s
// spec figure in DESIGN.md.
__global__ __launch_bounds__(kBlock) void stream_copy_kernel(int64_t n2, const double2 *__restrict__ src,
                                                             double2 *__restrict__ dst)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) dst[i] = src[i];
}

// one 16-byte element per thread, no loop: the shape the microarchitecture guide's 6.29 TB/s copy figure is
// quoted for (mode 2); four independent 16-byte elements per thread, loads before stores (mode 3); the same
// with non-temporal accesses (mode 4)
__global__ __launch_bounds__(kBlock) void stream_copy1_kernel(int64_t n2, const double2 *__restrict__ src,
                                                              double2 *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n2) dst[i] = src[i];
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void stream_copy4_kernel(int64_t n2, const double2 *__restrict__ src,
                                                              double2 *__restrict__ dst)
{
    typedef double vd2 __attribute__((ext_vector_type(2)));
    const vd2 *s = reinterpret_cast<const vd2 *>(src);
    vd2 *d = reinterpret_cast<vd2 *>(dst);
    const int64_t base = (int64_t)blockIdx.x * (4 * kBlock) + threadIdx.x;
    if (base + 3 * kBlock < n2) {
        vd2 v0, v1, v2, v3;
        if (NT) {
            v0 = __builtin_nontemporal_load(s + base);
            v1 = __builtin_nontemporal_load(s + base + kBlock);
            v2 = __builtin_nontemporal_load(s + base + 2 * kBlock);
            v3 = __builtin_nontemporal_load(s + base + 3 * kBlock);
            __builtin_nontemporal_store(v0, d + base);
            __builtin_nontemporal_store(v1, d + base + kBlock);
            __builtin_nontemporal_store(v2, d + base + 2 * kBlock);
            __builtin_nontemporal_store(v3, d + base + 3 * kBlock);
        } else {
            v0 = s[base];
            v1 = s[base + kBlock];
            v2 = s[base + 2 * kBlock];
            v3 = s[base + 3 * kBlock];
            d[base] = v0;
            d[base + kBlock] = v1;
            d[base + 2 * kBlock] = v2;
            d[base + 3 * kBlock] = v3;
        }
    } else {
        for (int k = 0; k < 4; ++k)
            if (base + k * kBlock < n2) d[base + k * kBlock] = s[base + k * kBlock];
    }
}

__global__ __launch_bounds__(kBlock) void stream_read_kernel(int64_t n2, const double2 *__restrict__ src,
                                                             double *__restrict__ out)
{
    __shared__ double red[4];
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    double a = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
        const double2 v = src[i];
        a += v.x + v.y;
    }
    a = block_sum(a, red);
    if (threadIdx.x == 0) out[blockIdx.x] = a;
}

__global__ __launch_bounds__(kBlock) void copy_kernel(int64_t n, const double *__restrict__ src,
                                                      double *__restrict__ dst)
{
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) d2[i] = s2[i];
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
}

__global__ void final_norm_kernel(const double *partials, int nparts, double *out)
{
    __shared__ double red[4];
    const double s = fold_partials(partials, nparts, red);
    if (threadIdx.x == 0) {
        out[0] = s;  // may be mapped host memory
        __threadfence_system();
    }
}

}  // namespace schwz

// ===========================================================================
// C ABI: stand-alone device objects
// ===========================================================================

using namespace schwz;

extern "C" {

int schwz_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int schwz_set_device(int device)
{
    SCHWZ_HIP_TRY(hipSetDevice(device));
    return SCHWZ_OK;
}

#define LAUNCH_GS(kern, OPV)                                                                       \
    hipLaunchKernelGGL((kern<OPV>), dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, \
                       d_idx, d_from, d_into)

int schwz_gather(int64_t n, const schwz_idx *d_idx, const double *d_from, double *d_into, int op,
                 schwz_stream stream)
{
    SCHWZ_REQUIRE(n >= 0, "schwz_gather: negative length");
    if (n == 0) return SCHWZ_OK;
    switch (op) {
    case SCHWZ_OP_COPY: LAUNCH_GS(gather_kernel, SCHWZ_OP_COPY); break;
    case SCHWZ_OP_ADD: LAUNCH_GS(gather_kernel, SCHWZ_OP_ADD); break;
    case SCHWZ_OP_DIFF: LAUNCH_GS(gather_kernel, SCHWZ_OP_DIFF); break;
    case SCHWZ_OP_AVG: LAUNCH_GS(gather_kernel, SCHWZ_OP_AVG); break;
    default: set_error("Undefined gather operation"); return SCHWZ_ERR_INVALID;
    }
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

int schwz_scatter(int64_t n, const schwz_idx *d_idx, const double *d_from, double *d_into, int op,
                  schwz_stream stream)
{
    SCHWZ_REQUIRE(n >= 0, "schwz_scatter: negative length");
    if (n == 0) return SCHWZ_OK;
    switch (op) {
    case SCHWZ_OP_COPY: LAUNCH_GS(scatter_kernel, SCHWZ_OP_COPY); break;
    case SCHWZ_OP_ADD: LAUNCH_GS(scatter_kernel, SCHWZ_OP_ADD); break;
    case SCHWZ_OP_DIFF: LAUNCH_GS(scatter_kernel, SCHWZ_OP_DIFF); break;
    case SCHWZ_OP_AVG: LAUNCH_GS(scatter_kernel, SCHWZ_OP_AVG); break;
    default: set_error("Undefined scatter operation"); return SCHWZ_ERR_INVALID;
    }
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}
#undef LAUNCH_GS

static int gs_typed(bool scatter, int64_t n, const void *d_idx, int index_type, const void *d_from, void *d_into,
                    int value_type, int op, schwz_stream stream)
{
    SCHWZ_REQUIRE(n >= 0, "schwz_gather_typed / schwz_scatter_typed: negative length");
    SCHWZ_REQUIRE(index_type == SCHWZ_INDEX_I32 || index_type == SCHWZ_INDEX_I64, "unknown index type");
    if (n == 0) return SCHWZ_OK;
    SCHWZ_REQUIRE(d_idx && d_from && d_into, "schwz_gather_typed / schwz_scatter_typed: null argument");
    hipStream_t st = (hipStream_t)stream;
    const bool wide = index_type == SCHWZ_INDEX_I64;
    switch (value_type) {
    case SCHWZ_VALUE_F32:
        return wide ? launch_gs_typed<float, int64_t>(scatter, n, d_idx, d_from, d_into, op, st)
                    : launch_gs_typed<float, int32_t>(scatter, n, d_idx, d_from, d_into, op, st);
    case SCHWZ_VALUE_F64:
        return wide ? launch_gs_typed<double, int64_t>(scatter, n, d_idx, d_from, d_into, op, st)
                    : launch_gs_typed<double, int32_t>(scatter, n, d_idx, d_from, d_into, op, st);
    case SCHWZ_VALUE_I32:
        return wide ? launch_gs_typed<int32_t, int64_t>(scatter, n, d_idx, d_from, d_into, op, st)
                    : launch_gs_typed<int32_t, int32_t>(scatter, n, d_idx, d_from, d_into, op, st);
    case SCHWZ_VALUE_I64:
        return wide ? launch_gs_typed<int64_t, int64_t>(scatter, n, d_idx, d_from, d_into, op, st)
                    : launch_gs_typed<int64_t, int32_t>(scatter, n, d_idx, d_from, d_into, op, st);
    default: set_error("schwz_gather_typed / schwz_scatter_typed: unknown value type"); return SCHWZ_ERR_INVALID;
    }
}

int schwz_gather_typed(int64_t n, const void *d_idx, int index_type, const void *d_from, void *d_into, int value_type,
                       int op, schwz_stream stream)
{
    return gs_typed(false, n, d_idx, index_type, d_from, d_into, value_type, op, stream);
}

int schwz_scatter_typed(int64_t n, const void *d_idx, int index_type, const void *d_from, void *d_into, int value_type,
                        int op, schwz_stream stream)
{
    return gs_typed(true, n, d_idx, index_type, d_from, d_into, value_type, op, stream);
}

// ---- CSR --------------------------------------------------------------------

}  // extern "C"


extern "C" {

int schwz_csr_create(int64_t nrows, int64_t ncols, const schwz_idx *h_rp, const schwz_idx *h_col,
                     const double *h_val, schwz_csr **out)
{
    SCHWZ_REQUIRE(out && nrows >= 0 && ncols >= 0 && h_rp, "schwz_csr_create: bad arguments");
    SCHWZ_REQUIRE(nrows < INT32_MAX, "schwz_csr_create: more than 2^31-1 local rows");
    const int64_t nnz = h_rp[nrows];
    SCHWZ_REQUIRE(h_rp[0] == 0 && nnz >= 0, "schwz_csr_create: malformed row_ptr");
    SCHWZ_REQUIRE(nnz == 0 || (h_col && h_val), "schwz_csr_create: null column / value array");
    // the kernels index x with these columns without a bounds test: refuse a malformed matrix here
    StageTimer timer_all("csr_create total");
    StageTimer timer_chk("csr_create: well-formed check");
    SCHWZ_REQUIRE(csr_is_well_formed(nrows, ncols, h_rp, h_col),
                  "schwz_csr_create: row_ptr not monotone or column index out of range");
    // row tiles: consecutive rows, <= kTileRows rows and <= kTileNnz nonzeros; a
    // row longer than kTileNnz forms a tile of its own.
    timer_chk.stop();
    StageTimer timer_tiles("csr_create: tiles, orders, column check");
    std::vector<schwz_idx> tiles;
    tiles.reserve((size_t)(nrows / 128 + 2));
    tiles.push_back(0);
    int64_t r = 0;
    while (r < nrows) {
        int64_t e = r;
        const int64_t s = h_rp[r];
        while (e < nrows && e - r < kTileRows && h_rp[e + 1] - s <= kTileNnz - 2) ++e;
        if (e == r) e = r + 1;  // long row
        tiles.push_back((schwz_idx)e);
        r = e;
    }
    // Visiting order of the tiles.  Every XCD sweeps one contiguous eighth of the
    // tile list with all its resident workgroups side by side, so a tile's x
    // entries are shared with the tiles that run at about the same time only if
    // "adjacent in the sweep" means "adjacent in the matrix graph".  In the natural
    // order of a 3-D grid the +-nx*ny neighbours are a whole plane of tiles away
    // and their x lines are evicted from the 4 MiB L2 by the matrix stream before
    // they are reused (rocprofv3: FETCH_SIZE 13 % above the algorithmic bytes).  A
    // breadth-first order of the tile graph inside each eighth shortens that
    // distance to one BFS level; matrices whose natural order is already local
    // (small bandwidth) are left almost unchanged.  SCHWZ_TILE_ORDER=0 disables it, =2 forces it for small matrices too.
    std::vector<schwz_idx> order;
    const char *ord_env = std::getenv("SCHWZ_TILE_ORDER");
    const int ntl = (int)tiles.size() - 1;
    const bool ord_force = ord_env && ord_env[0] == '2';  // also for small matrices (tests)
    const bool ord_on = ord_env && (ord_env[0] == '1' || ord_force);  // measured slower than the
    // block-cyclic deal below on the 3-D Poisson matrices, hence off by default
    if (ord_on && (ntl >= 4 * kMaxGrid || (ord_force && ntl > 1)) && nrows == ncols) {
        std::vector<schwz_idx> tile_of((size_t)nrows);
        for (int t = 0; t < ntl; ++t)
            for (schwz_idx rr = tiles[t]; rr < tiles[t + 1]; ++rr) tile_of[(size_t)rr] = t;
        const int chunk = (ntl + kXcds - 1) / kXcds;
        order.reserve((size_t)ntl);
        std::vector<char> seen((size_t)ntl, 0);
        std::vector<schwz_idx> nb;
        for (int x = 0; x < kXcds; ++x) {
            const int lo = x * chunk, hi = std::min(ntl, lo + chunk);
            size_t head = order.size();
            for (int start = lo; start < hi; ++start) {
                if (seen[(size_t)start]) continue;
                seen[(size_t)start] = 1;
                order.push_back(start);
                while (head < order.size()) {
                    const int t = order[head++];
                    nb.clear();
                    for (int64_t j = h_rp[tiles[t]]; j < h_rp[tiles[t + 1]]; ++j) {
                        const int u = tile_of[(size_t)h_col[j]];
                        if (u >= lo && u < hi && !seen[(size_t)u]) {
                            seen[(size_t)u] = 1;
                            nb.push_back(u);
                        }
                    }
                    std::sort(nb.begin(), nb.end());
                    order.insert(order.end(), nb.begin(), nb.end());
                }
            }
        }
    }
    // Brick order (SCHWZ_TILE_ORDER=3; measured and NOT the default, see the end of this comment): for a matrix whose rows see a
    // neighbour a fixed PL rows away (a 3-D grid in natural order) and whose tiles are whole runs of 256
    // rows, tile t sits at (plane t / tpp, in-plane position t % tpp).  The resident workgroups of an XCD
    // (per_xcd of them) take consecutive entries of the XCD's sequence, so the sequence is laid out in
    // bricks of per_xcd tiles -- by in-plane neighbours x bz planes -- and a brick's x lines, shared with
    // its +-PL and in-plane neighbours, are fetched into that XCD's L2 about once (+ the brick's surface)
    // instead of once per plane sweep.  Measured with the plain CSR kernel (variant 6) on one MI355X: 256^3
    // 0.379 -> 0.387 ms, 512 x 512 x 64 0.3955 -> 0.394 ms: like the BFS order, it saves bytes, not time --
    // the launch is not bound by the x lines it re-fetches.
    int brick_sh = -1;
    int64_t plane_rows = 0, uniform_tiles = 0;  // (for the stream kernel's sequence on wide planes, below)
    {
        const bool brick_env = ord_env && ord_env[0] == '3';
        int64_t uniform = 0;
        for (int t = 0; t < ntl; ++t) uniform += tiles[t + 1] - tiles[t] == kTileRows;
        std::vector<int64_t> far;
        const int64_t step = nrows > 8192 ? nrows / 8192 : 1;
        for (int64_t i = 0; i < nrows; i += step)
            if (h_rp[i + 1] > h_rp[i]) far.push_back((int64_t)h_col[h_rp[i + 1] - 1] - i);
        int64_t PL = 0;
        if (!far.empty()) {
            std::sort(far.begin(), far.end());
            size_t best = 0, run = 1, at = 0;
            for (size_t k = 1; k <= far.size(); ++k) {
                if (k < far.size() && far[k] == far[k - 1]) {
                    ++run;
                } else {
                    if (run > best) {
                        best = run;
                        at = k - 1;
                    }
                    run = 1;
                }
            }
            if (best * 2 > far.size()) PL = far[at];
        }
        plane_rows = PL;
        uniform_tiles = uniform;
        const int grid8 = std::max(1, (int)std::min<int64_t>(((int64_t)ntl + kXcds - 1) / kXcds * kXcds, kMaxGrid) / kXcds);
        if (brick_env && order.empty() && nrows == ncols && ntl >= 4 * kMaxGrid && uniform * 100 >= (int64_t)ntl * 95 &&
            PL >= 4 * kTileRows && PL % kTileRows == 0) {
            const int tpp = (int)(PL / kTileRows), nplanes = (ntl + tpp - 1) / tpp;
            int by = 1;
            while (by * by < grid8 && by * 2 <= tpp) by *= 2;
            const int bz = std::max(1, grid8 / by);
            std::vector<schwz_idx> G;
            G.reserve((size_t)ntl);
            for (int z0 = 0; z0 < nplanes; z0 += bz)
                for (int y0 = 0; y0 < tpp; y0 += by)
                    for (int z = z0; z < std::min(z0 + bz, nplanes); ++z)
                        for (int y = y0; y < std::min(y0 + by, tpp); ++y) {
                            const int64_t t = (int64_t)z * tpp + y;
                            if (t < ntl) G.push_back((schwz_idx)t);
                        }
            // the deal of spmv kernels (xcd_tile): XCD x's j-th tile is entry
            // ((j >> sh) << (sh + 3)) + (x << sh) + (j & ((1 << sh) - 1)) of the order array; one
            // contiguous share of G per XCD
            int sh = 0;
            while ((int64_t(2) << sh) <= (ntl + kXcds - 1) / kXcds) ++sh;
            brick_sh = sh;
            order.assign((size_t)ntl, -1);
            std::vector<int> cnt(kXcds, 0);
            const int per = 1 << sh;
            for (int x = 0; x < kXcds; ++x)
                for (int j = 0;; ++j) {
                    const int64_t pos = ((int64_t)(j >> sh) << (sh + 3)) + ((int64_t)x << sh) + (j & (per - 1));
                    if ((j >> sh) > (ntl >> (sh + 3)) + 1) break;
                    if (pos < ntl) cnt[(size_t)x] = j + 1;
                }
            // positions XCD x really owns (pos < ntl), in j order
            size_t g = 0;
            for (int x = 0; x < kXcds; ++x)
                for (int j = 0; j < cnt[(size_t)x]; ++j) {
                    const int64_t pos = ((int64_t)(j >> sh) << (sh + 3)) + ((int64_t)x << sh) + (j & (per - 1));
                    if (pos < ntl && g < G.size()) order[(size_t)pos] = G[g++];
                }
            bool complete = g == G.size();
            for (int t = 0; t < ntl && complete; ++t) complete = order[(size_t)t] >= 0;
            if (!complete) {
                order.clear();
                brick_sh = -1;
            }
        }
    }
    std::vector<schwz_idx> wtiles;
    wtiles.reserve((size_t)(nrows / 32 + 2));
    wtiles.push_back(0);
    r = 0;
    while (r < nrows) {
        int64_t e = r;
        const int64_t s = h_rp[r];
        while (e < nrows && e - r < 64 && h_rp[e + 1] - s <= kWaveTileNnz - 2) ++e;
        if (e == r) e = r + 1;  // long row
        wtiles.push_back((schwz_idx)e);
        r = e;
    }
    // (column indices were range-checked by csr_is_well_formed above)
    // spmv_stream.hip: nonzero offset of every tile, and whether the straight-line kernel applies (every tile
    // fits the 16-byte aligned window, no row longer than 32 entries)
    std::vector<schwz_idx> tile_nz(tiles.size());
    int stream_cap = 8;
    {
        int64_t longest = 0;
        for (int64_t i = 0; i < nrows; ++i) longest = std::max<int64_t>(longest, h_rp[i + 1] - h_rp[i]);
        for (size_t t = 0; t < tiles.size(); ++t) tile_nz[t] = h_rp[tiles[t]];
        for (size_t t = 0; t + 1 < tiles.size(); ++t)
            if ((tile_nz[t] & 3) + (tile_nz[t + 1] - tile_nz[t]) > kTileNnz || tiles[t + 1] - tiles[t] > kTileRows) stream_cap = 0;
        if (longest > 32 || nrows == 0 || !order.empty()) stream_cap = 0;
        else if (stream_cap && longest > 16) stream_cap = 32;
        else if (stream_cap && longest > 8) stream_cap = 16;
    }
    timer_tiles.stop();
    StageTimer timer_up("csr_create: upload of rp / col / val");
    schwz_csr *A = new schwz_csr();
    int rc;
    if ((rc = upload(tile_nz.data(), tile_nz.size(), &A->d_tile_nz)) ||
        (rc = upload(h_rp, (size_t)nrows + 1, &A->d_rp)) || (rc = upload(h_col, (size_t)nnz, &A->d_col, 4)) ||
        (rc = upload(h_val, (size_t)nnz, &A->d_val, 4)) || (rc = upload(tiles.data(), tiles.size(), &A->d_tile)) ||
        (rc = upload(wtiles.data(), wtiles.size(), &A->d_wtile)) ||
        (!order.empty() && (rc = upload(order.data(), order.size(), &A->d_order)))) {
        schwz_csr_destroy(A);
        return rc;
    }
    timer_up.stop();
    A->v.nrows = nrows;
    A->v.ncols = ncols;
    A->v.nnz = nnz;
    A->v.rp = (const schwz_idx *)A->d_rp;
    A->v.col = (const schwz_idx *)A->d_col;
    A->v.val = (const double *)A->d_val;
    A->v.ntiles = (int)tiles.size() - 1;
    A->h_tiles = tiles;
    A->v.tile_row = (const schwz_idx *)A->d_tile;
    A->v.tile_nz = (const schwz_idx *)A->d_tile_nz;
    A->v.stream_cap = stream_cap;
    // Wide planes (more than 2048 tiles = 512 K rows per plane, e.g. 1024 x 1024): with the block-cyclic deal an XCD
    // sweeps its share of a whole plane before it comes back to the same in-plane position of the next one -- 512
    // tiles = 12 MB of matrix stream later, long after the shared x lines have left its 4 MiB L2 (PMC traffic of the
    // plain-CSR SpMV on 1024 x 1024 x 128: 1.17 x the CSR bytes, x lines fetched three times).  The sequence of every
    // XCD is therefore laid out explicitly: its in-plane share in sub-stripes of 128 tiles, each sub-stripe through
    // ALL planes before the next one, so that the +-plane neighbours of a tile are 128 positions away -- in flight
    // together with it.  SCHWZ_STREAM_ORDER=0: the block-cyclic deal.
    {
        const char *so_env = std::getenv("SCHWZ_STREAM_ORDER");
        const int ntl_s = (int)tiles.size() - 1;
        const int64_t tpp = plane_rows > 0 && plane_rows % kTileRows == 0 ? plane_rows / kTileRows : 0;
        if (stream_cap && !(so_env && so_env[0] == '0') && nrows == ncols && tpp > kMaxGrid && tpp % kXcds == 0 &&
            ntl_s >= 4 * kMaxGrid && uniform_tiles * 100 >= (int64_t)ntl_s * 95) {
            const int nplanes_s = (int)((ntl_s + tpp - 1) / tpp);
            const int64_t w = tpp / kXcds;
            const int64_t S = std::min<int64_t>(128, w);
            std::vector<std::vector<schwz_idx>> seq((size_t)kXcds);
            for (int x = 0; x < kXcds; ++x)
                for (int64_t s0 = x * w; s0 < (x + 1) * w; s0 += S)
                    for (int z = 0; z < nplanes_s; ++z)
                        for (int64_t y = s0; y < std::min<int64_t>(s0 + S, (x + 1) * w); ++y) {
                            const int64_t t = (int64_t)z * tpp + y;
                            if (t < ntl_s) seq[(size_t)x].push_back((schwz_idx)t);
                        }
            size_t nper = 0, total = 0;
            for (const auto &l : seq) {
                nper = std::max(nper, l.size());
                total += l.size();
            }
            if (total == (size_t)ntl_s) {
                std::vector<schwz_idx> flat((size_t)kXcds * nper, -1);
                for (int x = 0; x < kXcds; ++x) std::copy(seq[(size_t)x].begin(), seq[(size_t)x].end(), flat.begin() + (size_t)x * nper);
                if (upload(flat.data(), flat.size(), &A->d_stream_order) == SCHWZ_OK) {
                    A->v.stream_order = (const schwz_idx *)A->d_stream_order;
                    A->v.stream_nper = (int)nper;
                }
            }
        }
    }
    if (stream_cap && (int)tiles.size() - 1 > kMaxGrid) {
        // scratch for the per-workgroup partial sums of the short-lived workgroups of spmv_stream.hip
        const int cap = (int)tiles.size() + 8 * kXcds;
        if (hipMalloc(&A->d_stream_part, (size_t)2 * cap * sizeof(double)) == hipSuccess) {
            A->v.stream_part = (double *)A->d_stream_part;
            A->v.stream_part_cap = cap;
        } else {
            (void)hipGetLastError();
        }
    }
    A->v.tile_order = order.empty() ? nullptr : (const schwz_idx *)A->d_order;
    {
        // run length of the block-cyclic deal: 1/8 of the matrix bandwidth in tiles
        // (SCHWZ_XCD_BLOCK overrides; >= tiles/8 reproduces one contiguous eighth per XCD)
        // median over sampled rows: robust against the few rows of a subdomain matrix
        // whose overlap columns sit at the far end of the local numbering
        std::vector<int64_t> bws;
        const int64_t step = nrows > 4096 ? nrows / 4096 : 1;
        for (int64_t i = 0; i < nrows; i += step)
            if (h_rp[i + 1] > h_rp[i])
                bws.push_back(std::max<int64_t>(i - h_col[h_rp[i]], h_col[h_rp[i + 1] - 1] - i));
        int64_t bw = 0;
        if (!bws.empty()) {
            std::nth_element(bws.begin(), bws.begin() + bws.size() / 2, bws.end());
            bw = std::max<int64_t>(0, bws[bws.size() / 2]);
        }
        const int64_t rows_per_tile = std::max<int64_t>(1, nrows / std::max(1, ntl));
        int64_t B = bw / rows_per_tile / kXcds;
        const char *be = std::getenv("SCHWZ_XCD_BLOCK");
        if (be && std::atoi(be) > 0) B = std::atoi(be);
        const int64_t cap = (ntl + kXcds - 1) / kXcds;
        B = std::max<int64_t>(1, std::min<int64_t>(B, cap));
        int sh = 0;  // rounded down to a power of two: the deal is shifts and masks on the device
        while ((int64_t(2) << sh) <= B) ++sh;
        A->pair_deal_shift = sh;  // the row-pair kernels keep the block-cyclic deal of the natural order
        if (brick_sh >= 0) sh = brick_sh;  // the brick order was laid out for this run length
        A->v.xcd_shift = sh;
        A->v.xcd_block = 1 << sh;
    }
    A->v.nwtiles = (int)wtiles.size() - 1;
    A->v.wtile_row = (const schwz_idx *)A->d_wtile;
    StageTimer timer_code("csr_create: matrix codings (patterns, pairs, walk tables)");
    if ((rc = build_spmv_dict(A, h_rp, h_col, h_val, tiles))) {
        schwz_csr_destroy(A);
        return rc;
    }
    *out = A;
    return SCHWZ_OK;
}

void schwz_csr_destroy(schwz_csr *A)
{
    if (!A) return;
    (void)hipFree(A->d_rp);
    (void)hipFree(A->d_col);
    (void)hipFree(A->d_val);
    (void)hipFree(A->d_tile);
    (void)hipFree(A->d_tile_nz);
    (void)hipFree(A->d_stream_part);
    (void)hipFree(A->d_stream_order);
    (void)hipFree(A->d_wtile);
    (void)hipFree(A->d_order);
    (void)hipFree(A->d_tile_dual);
    free_spmv_dict(A);
    delete A;
}

int64_t schwz_csr_nnz(const schwz_csr *A) { return A ? A->v.nnz : 0; }

int schwz_csr_format(const schwz_csr *A)
{
    return !A ? 0 : (A->v.pair_id ? 3 : (A->v.pat_id ? 2 : (A->v.code ? 1 : 0)));
}

int schwz_csr_symmetric(const schwz_csr *A) { return A && A->v.pair_id && A->v.pair_sym_base > 0 ? 1 : 0; }

// workgroup slots of the z-sweep walk the upload prepared (0: the matrix is walked chunk by chunk)
int schwz_csr_sweep_slots(const schwz_csr *A) { return A ? A->v.sweep_nslots : 0; }

// chunks of 512 rows the z-sweep walk leaves to its companion launch (0: it covers the whole matrix)
int schwz_csr_sweep_left_out(const schwz_csr *A) { return A ? A->v.sweep_ngen : 0; }

// Bytes of matrix data one pass of the default (variant 0) SpMV launch has to read in the coding the
// upload chose: what "algorithmic bytes of the launched format" means in bench.py's roofline.
int64_t schwz_csr_matrix_bytes(const schwz_csr *A, int variant)
{
    if (!A) return 0;
    const int64_t n = A->v.nrows, nnz = A->v.nnz;
    const int64_t plain = 12 * nnz + 4 * (n + 1);
    if (variant != 0) return plain;
    if (A->v.pair_id) return A->pair_code_bytes + (int64_t)((1.0 - A->pair_fraction) * (double)plain);
    if (A->v.pat_id) return n + 4 * (int64_t)A->v.ntiles + (int64_t)((1.0 - A->pattern_fraction) * (double)plain);
    if (A->v.code) return (int64_t)(2.0 * A->dict_fraction * (double)nnz) + 4 * (n + 1) +
                          (int64_t)((1.0 - A->dict_fraction) * 12.0 * (double)nnz);
    return plain;
}

int schwz_csr_spmv(const schwz_csr *A, double alpha, const double *d_x, double beta, double *d_y,
                   int variant, schwz_stream stream)
{
    SCHWZ_REQUIRE(A && d_x && d_y, "schwz_csr_spmv: null argument");
    SpmvArgs a;
    a.alpha = alpha;
    a.beta = beta;
    a.x = d_x;
    a.y = d_y;
    return launch_spmv(A->v, kSpmvPlain, a, variant, (hipStream_t)stream);
}

int schwz_stream_probe(int64_t n, int mode, const double *d_src, double *d_dst, schwz_stream stream)
{
    SCHWZ_REQUIRE(n >= 0 && d_src && d_dst, "schwz_stream_probe: bad arguments");
    SCHWZ_REQUIRE(mode >= 0 && mode <= 4,
                  "schwz_stream_probe: mode must be 0 (grid-stride copy), 1 (read), 2 (copy, one element per thread), "
                  "3 (copy, four per thread) or 4 (the same, non-temporal)");
    const int64_t n2 = n / 2;
    if (mode == 2)
        hipLaunchKernelGGL(stream_copy1_kernel, dim3((unsigned)((n2 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           (hipStream_t)stream, n2, (const double2 *)d_src, (double2 *)d_dst);
    else if (mode == 3)
        hipLaunchKernelGGL(stream_copy4_kernel<false>, dim3((unsigned)((n2 + 4 * kBlock - 1) / (4 * kBlock))), dim3(kBlock),
                           0, (hipStream_t)stream, n2, (const double2 *)d_src, (double2 *)d_dst);
    else if (mode == 4)
        hipLaunchKernelGGL(stream_copy4_kernel<true>, dim3((unsigned)((n2 + 4 * kBlock - 1) / (4 * kBlock))), dim3(kBlock),
                           0, (hipStream_t)stream, n2, (const double2 *)d_src, (double2 *)d_dst);
    else if (mode == 0)
        hipLaunchKernelGGL(stream_copy_kernel, dim3(kMaxGrid), dim3(kBlock), 0, (hipStream_t)stream, n / 2,
                           (const double2 *)d_src, (double2 *)d_dst);
    else
        hipLaunchKernelGGL(stream_read_kernel, dim3(kMaxGrid), dim3(kBlock), 0, (hipStream_t)stream, n / 2,
                           (const double2 *)d_src, d_dst);
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

}  // extern "C"

// ===========================================================================
// kernels used by subdomain.hip
// ===========================================================================

namespace schwz {

int launch_interface_update(int64_t nrows, int64_t row0, const schwz_idx *rp, const schwz_idx *col,
                            const double *val, const double *x, const double *b, double *bt, hipStream_t s)
{
    if (nrows == 0) return SCHWZ_OK;
    hipLaunchKernelGGL(interface_update_kernel, dim3(grid_for(nrows)), dim3(kBlock), 0, s, nrows, row0, rp,
                       col, val, x, b, bt);
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

// dst[0:n] = src[0:n]; both 16-byte aligned.  A plain kernel instead of
// hipMemcpyAsync: the runtime's copy path leaves a 20-40 us hole in the stream.
int launch_copy(int64_t n, const double *src, double *dst, hipStream_t s)
{
    if (n == 0) return SCHWZ_OK;
    hipLaunchKernelGGL(copy_kernel, dim3(grid_for((n + 1) / 2)), dim3(kBlock), 0, s, n, src, dst);
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

int csr_set_dual_split(schwz_csr *A, const schwz_idx *h_rp, const schwz_idx *h_col, int64_t split)
{
    const int ntiles = (int)A->h_tiles.size() - 1;
    if (ntiles <= 0) return SCHWZ_OK;
    std::vector<uint8_t> flag((size_t)ntiles, 0);
    for (int t = 0; t < ntiles; ++t) {
        const schwz_idx r0 = A->h_tiles[(size_t)t], r1 = A->h_tiles[(size_t)t + 1];
        bool f = r1 > split;
        for (int64_t j = h_rp[r0]; j < h_rp[r1] && !f; ++j) f = h_col[j] >= split;
        flag[(size_t)t] = f ? 1 : 0;
    }
    (void)hipFree(A->d_tile_dual);
    A->d_tile_dual = nullptr;
    SCHWZ_HIP_TRY(hipMalloc(&A->d_tile_dual, flag.size()));
    SCHWZ_HIP_TRY(hipMemcpy(A->d_tile_dual, flag.data(), flag.size(), hipMemcpyHostToDevice));
    A->v.tile_dual = (const uint8_t *)A->d_tile_dual;
    return pair_set_dual_split(A, h_rp, h_col, split);
}

int launch_gather_f32(int64_t n, const schwz_idx *idx, const double *from, float *into, hipStream_t s)
{
    if (n == 0) return SCHWZ_OK;
    hipLaunchKernelGGL(gather_f32_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, n, idx, from, into);
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

int launch_scatter_f32(int64_t n, const schwz_idx *idx, const float *from, double *into, hipStream_t s)
{
    if (n == 0) return SCHWZ_OK;
    hipLaunchKernelGGL(scatter_f32_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, n, idx, from, into);
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

int launch_final_norm(const double *partials, int nparts, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(final_norm_kernel, dim3(1), dim3(kBlock), 0, s, partials, nparts, out);
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

}  // namespace schwz

