// Host-side setup of the RAS hot path: problem sources, partitioning, the
// per-subdomain index sets / matrices / comm lists, and a sparse LL^T.
// No HIP calls in this file -- it also runs on a CPU-only host.
//
// Unlike the reference, nothing of global length is ever materialised per rank
// (SURVEY F7): the global matrix is a row source (explicit CSR or an analytic
// stencil), interior ids are an arithmetic range and only overlap / halo ids
// go through a hash map.
#include <omp.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <numeric>
#include <queue>
#include <set>
#include <unordered_map>
#include <sstream>

#include "schwz_internal.hpp"

namespace schwz {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

bool csr_is_well_formed(int64_t nrows, int64_t ncols, const schwz_idx *rp, const schwz_idx *col)
{
    bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok) num_threads(schwz::setup_threads())
    for (int64_t i = 0; i < nrows; ++i) {
        bool row_ok = rp[i + 1] >= rp[i];
        for (schwz_idx j = rp[i]; row_ok && j < rp[i + 1]; ++j) row_ok = col[j] >= 0 && col[j] < ncols;
        ok = ok && row_ok;
    }
    return ok;
}

// a_ij == a_ji bit for bit and the same sparsity either side of the diagonal.  Rows must have
// strictly ascending columns (what the row-pair coding requires anyway).
bool csr_is_symmetric(int64_t nrows, int64_t ncols, const schwz_idx *rp, const schwz_idx *col, const double *val)
{
    if (nrows != ncols) return false;
    bool ok = true;
    int64_t upper = 0, lower = 0;
#pragma omp parallel for schedule(static) reduction(&& : ok) reduction(+ : upper, lower) num_threads(schwz::setup_threads())
    for (int64_t i = 0; i < nrows; ++i) {
        if (!ok) continue;
        for (schwz_idx j = rp[i]; j < rp[i + 1]; ++j) {
            const int64_t c = col[j];
            if (c < i) {
                ++lower;
            } else if (c > i) {
                ++upper;
                const schwz_idx *b = col + rp[c], *e = col + rp[c + 1];
                const schwz_idx *f = std::lower_bound(b, e, (schwz_idx)i);
                if (f == e || *f != (schwz_idx)i || std::memcmp(&val[f - col], &val[j], sizeof(double)) != 0) {
                    ok = false;
                    break;
                }
            }
        }
    }
    return ok && upper == lower;
}

}  // namespace schwz

using namespace schwz;

// ---------------------------------------------------------------------------
// problem
// ---------------------------------------------------------------------------

int64_t schwz_problem::nnz() const
{
    if (kind == 0 || kind == 1) return rp.empty() ? 0 : rp.back();
    if (kind == 2) return 5 * N - 4 * nx;
    return 7 * N - 2 * (nx * ny + ny * nz + nx * nz);
}

int schwz_problem::row(int64_t g, int64_t *cols, double *vals) const
{
    int c = 0;
    if (kind == 0) {
        for (int64_t j = rp[g]; j < rp[g + 1]; ++j) {
            cols[c] = col[j];
            vals[c] = val[j];
            ++c;
        }
    } else if (kind == 1) {
        const auto it = std::lower_bound(present.begin(), present.end(), g);
        if (it == present.end() || *it != g) {
            // (rows are read by several threads during setup: the first miss wins, without a race)
            int64_t none = -1;
            __atomic_compare_exchange_n(&missing_row, &none, g, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
            return 0;
        }
        const int64_t k = it - present.begin();
        for (int64_t j = rp[(size_t)k]; j < rp[(size_t)k + 1]; ++j) {
            cols[c] = col[j];
            vals[c] = val[j];
            ++c;
        }
    } else if (kind == 2) {
        // source/initialization.cpp:214-265: offsets in key order {-n,-1,0,1,n},
        // pairs (kn,kn-1),(kn-1,kn) excluded
        const int64_t n = nx;
        if (g - n >= 0) { cols[c] = g - n; vals[c++] = -1.0; }
        if (g % n != 0) { cols[c] = g - 1; vals[c++] = -1.0; }
        cols[c] = g; vals[c++] = 4.0;
        if ((g + 1) % n != 0) { cols[c] = g + 1; vals[c++] = -1.0; }
        if (g + n < N) { cols[c] = g + n; vals[c++] = -1.0; }
    } else {
        const int64_t sxy = nx * ny;
        const int64_t x = g % nx, y = (g / nx) % ny, z = g / sxy;
        if (z > 0) { cols[c] = g - sxy; vals[c++] = -1.0; }
        if (y > 0) { cols[c] = g - nx; vals[c++] = -1.0; }
        if (x > 0) { cols[c] = g - 1; vals[c++] = -1.0; }
        cols[c] = g; vals[c++] = 6.0;
        if (x < nx - 1) { cols[c] = g + 1; vals[c++] = -1.0; }
        if (y < ny - 1) { cols[c] = g + nx; vals[c++] = -1.0; }
        if (z < nz - 1) { cols[c] = g + sxy; vals[c++] = -1.0; }
    }
    return c;
}

static void sort_row(int64_t *c, double *v, int len)
{
    for (int i = 1; i < len; ++i) {
        int64_t cc = c[i];
        double vv = v[i];
        int j = i;
        while (j > 0 && c[j - 1] > cc) {
            c[j] = c[j - 1];
            v[j] = v[j - 1];
            --j;
        }
        c[j] = cc;
        v[j] = vv;
    }
}

template <typename T>
static T *to_malloc(const std::vector<T> &v)
{
    T *p = (T *)std::malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
    if (p && !v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

extern "C" {

const char *schwz_last_error(void) { return g_last_error.c_str(); }
const char *schwz_version(void) { return "schwz-hip 0.1 (gfx950)"; }
int schwz_setup_threads(void) { return schwz::setup_threads(); }
void schwz_free(void *p) { std::free(p); }

int schwz_problem_laplacian(int dim, int64_t nx, int64_t ny, int64_t nz, schwz_problem **out)
{
    SCHWZ_REQUIRE(out, "schwz_problem_laplacian: null output");
    SCHWZ_REQUIRE(dim == 2 || dim == 3, "schwz_problem_laplacian: dim must be 2 or 3");
    SCHWZ_REQUIRE(nx > 0, "schwz_problem_laplacian: grid size must be positive");
    auto *p = new schwz_problem();
    p->kind = dim;
    p->nx = nx;
    if (dim == 2) {
        p->ny = nx;
        p->nz = 1;
        p->N = nx * nx;
        p->max_row_nnz = 5;
    } else {
        SCHWZ_REQUIRE(ny > 0 && nz > 0, "schwz_problem_laplacian: grid size must be positive");
        p->ny = ny;
        p->nz = nz;
        p->N = nx * ny * nz;
        p->max_row_nnz = 7;
    }
    *out = p;
    return SCHWZ_OK;
}

// Distributed ingest: a process holds the rows its subdomain reads -- interior and overlap
// rows -- of a matrix one rank parsed and partitioned; the rest of the setup (index sets, local and
// interface matrices, lists) runs on that part exactly as on the whole matrix.
int schwz_problem_from_rows(int64_t N, int64_t nrows, const int64_t *row_ids, const int64_t *rp,
                            const schwz_idx *col, const double *val, schwz_problem **out)
{
    SCHWZ_REQUIRE(out && N >= 0 && nrows >= 0 && nrows <= N && rp && (nrows == 0 || row_ids),
                  "schwz_problem_from_rows: bad arguments");
    SCHWZ_REQUIRE(rp[0] == 0, "schwz_problem_from_rows: row_ptr must start at 0");
    for (int64_t i = 0; i < nrows; ++i) {
        SCHWZ_REQUIRE(row_ids[i] >= 0 && row_ids[i] < N && (i == 0 || row_ids[i - 1] < row_ids[i]),
                      "schwz_problem_from_rows: row ids must be ascending and inside the matrix");
        SCHWZ_REQUIRE(rp[i + 1] >= rp[i], "schwz_problem_from_rows: row_ptr not monotone");
    }
    const int64_t nnz = rp[nrows];
    SCHWZ_REQUIRE(nnz == 0 || (col && val), "schwz_problem_from_rows: null column / value array");
    auto *p = new schwz_problem();
    p->kind = 1;
    p->N = N;
    p->present.assign(row_ids, row_ids + nrows);
    p->rp.assign(rp, rp + nrows + 1);
    p->col.assign(col, col + nnz);
    p->val.assign(val, val + nnz);
    for (int64_t i = 0; i < nrows; ++i) {
        p->max_row_nnz = std::max(p->max_row_nnz, (int)(rp[i + 1] - rp[i]));
        for (int64_t j = rp[i]; j < rp[i + 1]; ++j)
            if (col[j] < 0 || col[j] >= N || (j > rp[i] && col[j - 1] >= col[j])) {
                delete p;
                set_error("schwz_problem_from_rows: columns must be ascending and inside the matrix");
                return SCHWZ_ERR_INVALID;
            }
    }
    *out = p;
    return SCHWZ_OK;
}

// The rows `row_ids` of a problem as CSR arrays (what the parsing rank sends to the others).  Call once
// with col_out == nullptr for the row pointers (rp_out[nrows] = entries needed), then with the arrays.
int schwz_problem_extract_rows(const schwz_problem *p, int64_t nrows, const int64_t *row_ids, int64_t *rp_out,
                               schwz_idx *col_out, double *val_out)
{
    SCHWZ_REQUIRE(p && nrows >= 0 && rp_out && (nrows == 0 || row_ids), "schwz_problem_extract_rows: bad arguments");
    std::vector<int64_t> c((size_t)p->max_row_nnz + 1);
    std::vector<double> v((size_t)p->max_row_nnz + 1);
    rp_out[0] = 0;
    p->missing_row = -1;  // a miss recorded by an earlier, failed call must not fail this one
    for (int64_t i = 0; i < nrows; ++i) {
        SCHWZ_REQUIRE(row_ids[i] >= 0 && row_ids[i] < p->N, "schwz_problem_extract_rows: row outside the matrix");
        const int len = p->row(row_ids[i], c.data(), v.data());
        if (col_out)
            for (int k = 0; k < len; ++k) {
                col_out[rp_out[i] + k] = (schwz_idx)c[(size_t)k];
                val_out[rp_out[i] + k] = v[(size_t)k];
            }
        rp_out[i + 1] = rp_out[i] + len;
    }
    if (p->missing_row >= 0) {
        set_error("schwz_problem_extract_rows: a requested row is not held by this process (row " +
                  std::to_string(p->missing_row) + ")");
        p->missing_row = -1;
        return SCHWZ_ERR_INVALID;
    }
    return SCHWZ_OK;
}

int schwz_problem_from_csr(int64_t N, const int64_t *rp, const schwz_idx *col, const double *val,
                           schwz_problem **out)
{
    SCHWZ_REQUIRE(out && rp && N >= 0, "schwz_problem_from_csr: bad arguments");
    auto *p = new schwz_problem();
    p->kind = 0;
    p->N = N;
    p->rp.assign(rp, rp + N + 1);
    const int64_t nnz = rp[N];
    p->col.assign(col, col + nnz);
    p->val.assign(val, val + nnz);
    for (int64_t i = 0; i < N; ++i) {
        const int len = (int)(rp[i + 1] - rp[i]);
        p->max_row_nnz = std::max(p->max_row_nnz, len);
        // sort_by_column_index (initialization.cpp:212)
        bool sorted = true;
        for (int64_t j = rp[i] + 1; j < rp[i + 1]; ++j)
            if (p->col[j - 1] > p->col[j]) sorted = false;
        if (!sorted) {
            std::vector<std::pair<schwz_idx, double>> tmp((size_t)len);
            for (int k = 0; k < len; ++k) tmp[k] = {p->col[rp[i] + k], p->val[rp[i] + k]};
            std::stable_sort(tmp.begin(), tmp.end(),
                             [](const auto &a, const auto &b) { return a.first < b.first; });
            for (int k = 0; k < len; ++k) {
                p->col[rp[i] + k] = tmp[k].first;
                p->val[rp[i] + k] = tmp[k].second;
            }
        }
    }
    for (int64_t j = 0; j < nnz; ++j) {
        if (p->col[j] < 0 || p->col[j] >= N) {
            delete p;
            set_error("schwz_problem_from_csr: column index out of range");
            return SCHWZ_ERR_INVALID;
        }
    }
    *out = p;
    return SCHWZ_OK;
}

// Matrix-Market coordinate real/integer/pattern, general/symmetric
int schwz_problem_from_matrix_market(const char *path, schwz_problem **out)
{
    SCHWZ_REQUIRE(path && out, "schwz_problem_from_matrix_market: null argument");
    std::ifstream in(path);
    if (!in) {
        set_error(std::string("Could not find the file \"") + path + "\"");
        return SCHWZ_ERR_IO;
    }
    std::string line;
    std::getline(in, line);
    std::string low = line;
    std::transform(low.begin(), low.end(), low.begin(), ::tolower);
    if (low.find("%%matrixmarket") != 0 || low.find("coordinate") == std::string::npos) {
        set_error("matrix market: only coordinate format is supported");
        return SCHWZ_ERR_IO;
    }
    const bool symmetric = low.find("symmetric") != std::string::npos;
    const bool pattern = low.find("pattern") != std::string::npos;
    while (std::getline(in, line))
        if (!line.empty() && line[0] != '%') break;
    int64_t nr = 0, nc = 0, nz = 0;
    {
        std::istringstream ss(line);
        ss >> nr >> nc >> nz;
    }
    if (nr <= 0 || nr != nc) {
        set_error("matrix market: need a square matrix");
        return SCHWZ_ERR_IO;
    }
    std::vector<int64_t> ri, ci;
    std::vector<double> vv;
    ri.reserve((size_t)nz * (symmetric ? 2 : 1));
    ci.reserve(ri.capacity());
    vv.reserve(ri.capacity());
    for (int64_t k = 0; k < nz; ++k) {
        int64_t r, c;
        double v = 1.0;
        if (!(in >> r >> c)) {
            set_error("matrix market: truncated file");
            return SCHWZ_ERR_IO;
        }
        if (!pattern) in >> v;
        ri.push_back(r - 1);
        ci.push_back(c - 1);
        vv.push_back(v);
        if (symmetric && r != c) {
            ri.push_back(c - 1);
            ci.push_back(r - 1);
            vv.push_back(v);
        }
    }
    std::vector<int64_t> rp((size_t)nr + 1, 0);
    for (int64_t r : ri) rp[r + 1]++;
    for (int64_t i = 0; i < nr; ++i) rp[i + 1] += rp[i];
    std::vector<schwz_idx> col(ri.size());
    std::vector<double> val(ri.size());
    std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
    for (size_t k = 0; k < ri.size(); ++k) {
        col[fill[ri[k]]] = (schwz_idx)ci[k];
        val[fill[ri[k]]] = vv[k];
        fill[ri[k]]++;
    }
    return schwz_problem_from_csr(nr, rp.data(), col.data(), val.data(), out);
}

void schwz_problem_destroy(schwz_problem *p) { delete p; }
int64_t schwz_problem_size(const schwz_problem *p) { return p ? p->N : 0; }
int64_t schwz_problem_nnz(const schwz_problem *p) { return p ? p->nnz() : 0; }

int schwz_problem_row(const schwz_problem *p, int64_t row, int *n, int64_t *cols, double *vals, int capacity)
{
    SCHWZ_REQUIRE(p && n && row >= 0 && row < p->N, "schwz_problem_row: bad arguments");
    SCHWZ_REQUIRE(capacity >= p->max_row_nnz, "schwz_problem_row: capacity too small");
    *n = p->row(row, cols, vals);
    return SCHWZ_OK;
}

// source/restricted_schwarz.cpp:105-152
int schwz_problem_permute(const schwz_problem *p, int P, const uint32_t *part, int64_t *perm,
                          int64_t *first_row, schwz_problem **out)
{
    SCHWZ_REQUIRE(p && part && perm && first_row && out && P > 0, "schwz_problem_permute: bad arguments");
    const int64_t N = p->N;
    std::vector<int64_t> cnt((size_t)P, 0);
    for (int64_t i = 0; i < N; ++i) {
        SCHWZ_REQUIRE(part[i] < (uint32_t)P, "schwz_problem_permute: part id out of range");
        cnt[part[i]]++;
    }
    first_row[0] = 0;
    for (int q = 0; q < P; ++q) first_row[q + 1] = first_row[q] + cnt[q];
    std::vector<int64_t> next(first_row, first_row + P);
    for (int64_t i = 0; i < N; ++i) perm[next[part[i]]++] = i;
    std::vector<int64_t> iperm((size_t)N);
    for (int64_t i = 0; i < N; ++i) iperm[perm[i]] = i;
    std::vector<int64_t> rp((size_t)N + 1, 0);
    std::vector<schwz_idx> col;
    std::vector<double> val;
    col.reserve((size_t)p->nnz());
    val.reserve((size_t)p->nnz());
    std::vector<int64_t> c((size_t)p->max_row_nnz + 1);
    std::vector<double> v((size_t)p->max_row_nnz + 1);
    for (int64_t row = 0; row < N; ++row) {
        const int len = p->row(perm[row], c.data(), v.data());
        for (int k = 0; k < len; ++k) {
            col.push_back((schwz_idx)iperm[c[k]]);
            val.push_back(v[k]);
        }
        rp[row + 1] = (int64_t)col.size();
    }
    // the reference does not re-sort after permuting; schwz_problem_from_csr
    // sorts rows, which only changes the order within a row (sums are re-sorted
    // by local column index later anyway, restricted_schwarz.cpp:297-298)
    return schwz_problem_from_csr(N, rp.data(), col.data(), val.data(), out);
}

// minstd_rand0 (x <- 16807 x mod 2^31-1, seed 1) and libstdc++'s generate_canonical<double,53>
// with two draws, evaluated in the same floating-point order
int schwz_rhs_random(int64_t count, const int64_t *ids, double *out)
{
    SCHWZ_REQUIRE(count >= 0 && (count == 0 || (ids && out)), "schwz_rhs_random: bad arguments");
    const uint64_t m = 2147483647ull, a = 16807ull;
    const long double range = 2147483646.0L;  // max() - min() + 1
    auto powmod = [&](uint64_t e) {
        uint64_t r = 1, b = a;
        while (e) {
            if (e & 1) r = r * b % m;
            b = b * b % m;
            e >>= 1;
        }
        return r;
    };
    for (int64_t i = 0; i < count; ++i) {
        SCHWZ_REQUIRE(ids[i] >= 0, "schwz_rhs_random: negative row id");
        const uint64_t x1 = powmod(2 * (uint64_t)ids[i] + 1);  // state after 2g+1 steps from seed 1
        const uint64_t x2 = x1 * a % m;
        double sum = 0.0, tmp = 1.0;
        sum += (double)(x1 - 1) * tmp;
        tmp = (double)(tmp * range);
        sum += (double)(x2 - 1) * tmp;
        tmp = (double)(tmp * range);
        double ret = sum / tmp;
        if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
        out[i] = ret;  // (b - a) * ret + a with a = 0, b = 1
    }
    return SCHWZ_OK;
}

// source/restricted_schwarz.cpp:84,97-102
int schwz_partition_regular(int64_t N, int P, int64_t *first_row)
{
    SCHWZ_REQUIRE(first_row && P > 0 && N >= 0, "schwz_partition_regular: bad arguments");
    const int64_t nb = (N + P - 1) / P;
    first_row[0] = 0;
    for (int p = 0; p < P; ++p) first_row[p + 1] = first_row[p] + std::min(N - first_row[p], nb);
    return SCHWZ_OK;
}

// include/partition_tools.hpp:70-106
int schwz_partition_regular2d(int64_t n1d, int P, uint32_t *part)
{
    SCHWZ_REQUIRE(part && n1d > 0 && P > 0, "schwz_partition_regular2d: bad arguments");
    const int sq_p = (int)std::sqrt((double)P);
    SCHWZ_REQUIRE(sq_p * sq_p == P && n1d % sq_p == 0,
                  "regular2d needs a square subdomain count that divides the grid (SURVEY F10)");
    const int64_t b = n1d / sq_p;
    for (int j1 = 0; j1 < sq_p; ++j1) {
        const int64_t offset2 = (int64_t)j1 * sq_p * b * b;
        for (int j2 = 0; j2 < sq_p; ++j2) {
            const uint32_t id = (uint32_t)(sq_p * j1 + j2);
            const int64_t offset1 = (int64_t)j2 * n1d / sq_p;
            for (int64_t i1 = 0; i1 < b; ++i1)
                for (int64_t i2 = 0; i2 < b; ++i2) part[offset2 + offset1 + i1 * n1d + i2] = id;
        }
    }
    return SCHWZ_OK;
}

// Recursive bisection by BFS level structure from a pseudo-peripheral node:
// stands in for METIS_PartGraphKway (include/partition_tools.hpp:183-195).
static void bisect(const schwz_problem *p, std::vector<int64_t> &nodes, int parts, uint32_t base,
                   uint32_t *part, std::vector<int32_t> &mark, int32_t &stamp)
{
    if (nodes.empty()) return;
    if (parts == 1) {
        for (int64_t g : nodes) part[g] = base;
        return;
    }
    const int left_parts = parts / 2;
    const int64_t target = (int64_t)((double)nodes.size() * left_parts / parts + 0.5);
    std::vector<int64_t> c((size_t)p->max_row_nnz + 1);
    std::vector<double> v((size_t)p->max_row_nnz + 1);
    // mark membership
    ++stamp;
    const int32_t member = stamp;
    for (int64_t g : nodes) mark[g] = member;
    auto bfs = [&](int64_t start, std::vector<int64_t> &order) {
        ++stamp;
        const int32_t seen = stamp;
        order.clear();
        size_t head = 0;
        order.push_back(start);
        mark[start] = seen;
        for (;;) {
            while (head < order.size()) {
                const int64_t u = order[head++];
                const int len = p->row(u, c.data(), v.data());
                for (int k = 0; k < len; ++k) {
                    const int64_t w = c[k];
                    if (mark[w] == member) {
                        mark[w] = seen;
                        order.push_back(w);
                    }
                }
            }
            if (order.size() == nodes.size()) break;
            // disconnected piece: continue from the first unseen member
            for (int64_t g : nodes)
                if (mark[g] == member) {
                    mark[g] = seen;
                    order.push_back(g);
                    break;
                }
        }
        // restore membership marks for the next sweep
        for (int64_t g : order) mark[g] = member;
    };
    std::vector<int64_t> order;
    bfs(nodes.front(), order);
    int64_t far = order.back();
    bfs(far, order);
    far = order.back();
    bfs(far, order);
    // Refinement of the level-structure cut (what METIS does after its initial bisection): passes of
    // Fiduccia-Mattheyses moves, taken in PAIRS (best node of the left side, then best node of the right
    // side) so that the sizes never change; the pass keeps the best prefix of its move sequence and the
    // passes stop when one brings nothing.  Unit edge weights.  SCHWZ_PART_REFINE=0 disables it; skipped
    // above 4 M nodes per bisection (the regular partitions are the ones used at that size).
    static const bool refine_on = [] {
        const char *e = std::getenv("SCHWZ_PART_REFINE");
        return !(e && e[0] == '0');
    }();
    if (refine_on && nodes.size() >= 4 && nodes.size() <= (size_t)4000000 && target > 0 &&
        target < (int64_t)nodes.size()) {
        // side 0 / 1 of the members, kept in `mark` as member + 1 / member + 2 (both != any other stamp in use)
        stamp += 2;
        const int32_t s0 = stamp - 1, s1 = stamp;
        const size_t nodes_count = nodes.size();
        for (int64_t i = 0; i < (int64_t)order.size(); ++i) mark[order[(size_t)i]] = i < target ? s0 : s1;
        std::unordered_map<int64_t, int> gain;
        gain.reserve(nodes.size() * 2);
        auto compute_gain = [&](int64_t u) {
            const int len = p->row(u, c.data(), v.data());
            int g = 0;
            for (int k = 0; k < len; ++k) {
                const int64_t w = c[k];
                if (w == u || (mark[w] != s0 && mark[w] != s1)) continue;
                g += mark[w] != mark[u] ? 1 : -1;
            }
            return g;
        };
        for (int pass = 0; pass < 8; ++pass) {
            std::set<std::pair<int, int64_t>> q[2];  // (-gain, node): begin() = best move
            for (int64_t u : order) {
                const int g = compute_gain(u);
                gain[u] = g;
                q[mark[u] == s1].insert({-g, u});
            }
            std::vector<int64_t> moved;
            std::unordered_map<int64_t, char> locked;
            int64_t delta = 0, best = 0;
            size_t best_len = 0;
            int stale = 0;
            auto move = [&](int from) -> bool {
                if (q[from].empty()) return false;
                const auto it = q[from].begin();
                const int64_t u = it->second;
                const int g = -it->first;
                q[from].erase(it);
                locked[u] = 1;
                delta -= g;
                mark[u] = from == 0 ? s1 : s0;
                moved.push_back(u);
                const int len = p->row(u, c.data(), v.data());
                for (int k = 0; k < len; ++k) {
                    const int64_t w = c[k];
                    if (w == u || (mark[w] != s0 && mark[w] != s1) || locked.count(w)) continue;
                    const int side_w = mark[w] == s1;
                    int &gw = gain[w];
                    q[side_w].erase({-gw, w});
                    gw += (side_w == from) ? 2 : -2;  // u left w's side: the edge is cut now; or joined it: no longer cut
                    q[side_w].insert({-gw, w});
                }
                return true;
            };
            const int patience = (int)std::min<size_t>(4000, std::max<size_t>(64, nodes_count / 50));
            while (stale < patience) {
                if (!move(0) || !move(1)) break;
                if (delta < best) {
                    best = delta;
                    best_len = moved.size();
                    stale = 0;
                } else {
                    ++stale;
                }
            }
            for (size_t i = moved.size(); i > best_len; --i) {  // undo what came after the best prefix
                const int64_t u = moved[i - 1];
                mark[u] = mark[u] == s0 ? s1 : s0;
            }
            if (best == 0) break;
        }
        size_t wl = 0, wr = (size_t)target;
        std::vector<int64_t> resorted(order.size());
        for (int64_t u : order) resorted[mark[u] == s0 ? wl++ : wr++] = u;
        order.swap(resorted);
        for (int64_t u : order) mark[u] = member;
    }
    std::vector<int64_t> left(order.begin(), order.begin() + target);
    std::vector<int64_t> right(order.begin() + target, order.end());
    std::sort(left.begin(), left.end());
    std::sort(right.begin(), right.end());
    nodes.clear();
    nodes.shrink_to_fit();
    bisect(p, left, left_parts, base, part, mark, stamp);
    bisect(p, right, parts - left_parts, base + (uint32_t)left_parts, part, mark, stamp);
}


// ---------------------------------------------------------------------------
// Multilevel recursive bisection (the scheme of METIS_PartGraphRecursive, which PartitionTools::PartitionMetis
// would call through METIS, include/partition_tools.hpp:110-202; METIS itself is absent):
//   coarsening by heavy-edge matching until ~100 vertices are left, greedy graph growing from several seeds on
//   the coarsest graph, and Fiduccia-Mattheyses boundary refinement at every level on the way back up --
//   weighted while vertices stand for several rows (balance within the heaviest vertex), then on the rows
//   themselves to EXACT part sizes (single moves towards the target, then moves in pairs).
// Deterministic: no random numbers, ties broken by vertex number.  SCHWZ_PART_MULTILEVEL=0 keeps the
// single-level bisection above.
// ---------------------------------------------------------------------------
namespace {

struct MlGraph {
    int n = 0;
    std::vector<int64_t> xadj;   // n + 1
    std::vector<int> adj, ew;    // neighbours and edge weights (symmetric)
    std::vector<int> vw;         // vertex weights
    int64_t total_vw() const
    {
        int64_t t = 0;
        for (int w : vw) t += w;
        return t;
    }
};

// side[v] in {0, 1}.  Fiduccia-Mattheyses passes: every vertex moves at most once per pass, always the best-gain
// move that keeps |w0 - target0| within `slack` (from the heavy side while it is outside), and the pass keeps the
// best prefix of its move sequence -- first by how far the balance is outside the slack, then by the cut.
// slack 0 with unit weights: exact sizes.
static void ml_refine(const MlGraph &g, std::vector<char> &side, int64_t target0, int64_t slack, int passes)
{
    const int n = g.n;
    std::vector<int> gain((size_t)n);
    std::vector<char> locked((size_t)n);
    auto excess = [&](int64_t w0) { return std::max<int64_t>(0, (int64_t)std::llabs(w0 - target0) - slack); };
    for (int pass = 0; pass < passes; ++pass) {
        int64_t w0 = 0;
        std::set<std::pair<int, int>> q[2];  // (-gain, vertex) per side
        for (int v = 0; v < n; ++v) {
            int gsum = 0;
            for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; ++j)
                gsum += side[(size_t)g.adj[(size_t)j]] != side[(size_t)v] ? g.ew[(size_t)j] : -g.ew[(size_t)j];
            gain[(size_t)v] = gsum;
            locked[(size_t)v] = 0;
            q[(int)side[(size_t)v]].insert({-gsum, v});
            if (!side[(size_t)v]) w0 += g.vw[(size_t)v];
        }
        std::vector<int> moved;
        int64_t delta = 0, best_delta = 0, best_excess = excess(w0);
        size_t best_len = 0;
        int stale = 0;
        const int patience = std::max(50, std::min(n / 20, 5000));
        while (stale < patience) {
            int from = -1;
            const int64_t imb = w0 - target0;
            auto stays_inside = [&](int f) {
                if (q[f].empty()) return false;
                const int v = q[f].begin()->second;
                const int64_t nw0 = w0 + (f == 0 ? -(int64_t)g.vw[(size_t)v] : (int64_t)g.vw[(size_t)v]);
                return (int64_t)std::llabs(nw0 - target0) <= slack;
            };
            if (imb > slack) {
                from = q[0].empty() ? -1 : 0;
            } else if (imb < -slack) {
                from = q[1].empty() ? -1 : 1;
            } else {
                const bool f0 = stays_inside(0), f1 = stays_inside(1);
                if (f0 && f1)
                    from = q[0].begin()->first <= q[1].begin()->first ? 0 : 1;
                else if (f0)
                    from = 0;
                else if (f1)
                    from = 1;
                else if (slack == 0 && !q[0].empty() && !q[1].empty())
                    from = imb >= 0 ? 0 : 1;  // exact balance: a move out and, next, the best move back (a pair)
            }
            if (from < 0) break;
            const auto it = q[from].begin();
            const int v = it->second;
            q[from].erase(it);
            locked[(size_t)v] = 1;
            delta -= gain[(size_t)v];
            side[(size_t)v] = (char)(1 - from);
            w0 += from == 0 ? -(int64_t)g.vw[(size_t)v] : (int64_t)g.vw[(size_t)v];
            moved.push_back(v);
            for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; ++j) {
                const int w = g.adj[(size_t)j];
                if (locked[(size_t)w]) continue;
                const int sw = side[(size_t)w];
                q[sw].erase({-gain[(size_t)w], w});
                gain[(size_t)w] += (sw == from ? 2 : -2) * g.ew[(size_t)j];
                q[sw].insert({-gain[(size_t)w], w});
            }
            const int64_t ex = excess(w0);
            if (ex < best_excess || (ex == best_excess && delta < best_delta)) {
                best_excess = ex;
                best_delta = delta;
                best_len = moved.size();
                stale = 0;
            } else {
                ++stale;
            }
        }
        for (size_t i = moved.size(); i > best_len; --i) {
            const int v = moved[i - 1];
            side[(size_t)v] = (char)(1 - side[(size_t)v]);
        }
        if (best_len == 0) break;
    }
}

// heavy-edge matching: cmap[v] = coarse vertex; returns the coarse graph
static MlGraph ml_coarsen(const MlGraph &g, std::vector<int> &cmap)
{
    const int n = g.n;
    cmap.assign((size_t)n, -1);
    // light vertices first (keeps the coarse vertex weights even), ties by number
    std::vector<int> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return g.vw[(size_t)a] < g.vw[(size_t)b]; });
    int nc = 0;
    for (int v : order) {
        if (cmap[(size_t)v] >= 0) continue;
        int best = -1, best_w = -1;
        for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; ++j) {
            const int w = g.adj[(size_t)j];
            if (cmap[(size_t)w] < 0 && w != v && g.ew[(size_t)j] > best_w) {
                best_w = g.ew[(size_t)j];
                best = w;
            }
        }
        cmap[(size_t)v] = nc;
        if (best >= 0) cmap[(size_t)best] = nc;
        ++nc;
    }
    MlGraph c;
    c.n = nc;
    c.vw.assign((size_t)nc, 0);
    for (int v = 0; v < n; ++v) c.vw[(size_t)cmap[(size_t)v]] += g.vw[(size_t)v];
    // members of every coarse vertex, then its merged adjacency
    std::vector<int> first((size_t)nc, -1), second((size_t)nc, -1);
    for (int v = 0; v < n; ++v) {
        const int cv = cmap[(size_t)v];
        if (first[(size_t)cv] < 0)
            first[(size_t)cv] = v;
        else
            second[(size_t)cv] = v;
    }
    c.xadj.assign((size_t)nc + 1, 0);
    std::vector<int> slot((size_t)nc, -1);
    for (int cv = 0; cv < nc; ++cv) {
        const int64_t start = (int64_t)c.adj.size();
        for (int v : {first[(size_t)cv], second[(size_t)cv]}) {
            if (v < 0) continue;
            for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; ++j) {
                const int cw = cmap[(size_t)g.adj[(size_t)j]];
                if (cw == cv) continue;
                if (slot[(size_t)cw] < start) {
                    slot[(size_t)cw] = (int)c.adj.size();
                    c.adj.push_back(cw);
                    c.ew.push_back(g.ew[(size_t)j]);
                } else {
                    c.ew[(size_t)slot[(size_t)cw]] += g.ew[(size_t)j];
                }
            }
        }
        c.xadj[(size_t)cv + 1] = (int64_t)c.adj.size();
    }
    return c;
}

// greedy graph growing on the coarsest graph: side 0 grows from a seed by the vertex with the largest gain until
// it holds target0; several seeds, the smallest cut wins
static void ml_initial(const MlGraph &g, std::vector<char> &side, int64_t target0)
{
    const int n = g.n;
    int64_t best_cut = -1;
    std::vector<char> best;
    const int tries = std::min(n, 12);
    for (int t = 0; t < tries; ++t) {
        const int seed = (int)((int64_t)t * n / tries);
        std::vector<char> s((size_t)n, 1);
        std::vector<int> conn((size_t)n, 0);  // edge weight towards side 0
        std::vector<char> in0((size_t)n, 0);
        int64_t w0 = 0;
        int next = seed;
        while (next >= 0 && w0 < target0) {
            s[(size_t)next] = 0;
            in0[(size_t)next] = 1;
            w0 += g.vw[(size_t)next];
            for (int64_t j = g.xadj[(size_t)next]; j < g.xadj[(size_t)next + 1]; ++j) conn[(size_t)g.adj[(size_t)j]] += g.ew[(size_t)j];
            // the frontier vertex most tied to side 0 (ties: fewest ties to side 1, then number); none: any vertex left
            int pick = -1;
            int64_t pick_score = INT64_MIN;
            for (int v = 0; v < n; ++v) {
                if (in0[(size_t)v]) continue;
                if (conn[(size_t)v] == 0 && pick >= 0) continue;
                int64_t deg = 0;
                for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; ++j) deg += g.ew[(size_t)j];
                const int64_t score = conn[(size_t)v] > 0 ? (int64_t)2 * conn[(size_t)v] - deg + ((int64_t)1 << 40) : -deg;
                if (score > pick_score) {
                    pick_score = score;
                    pick = v;
                }
            }
            next = pick;
        }
        ml_refine(g, s, target0, *std::max_element(g.vw.begin(), g.vw.end()), 4);
        int64_t cut = 0;
        for (int v = 0; v < n; ++v)
            for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; ++j)
                if (s[(size_t)v] != s[(size_t)g.adj[(size_t)j]]) cut += g.ew[(size_t)j];
        if (best_cut < 0 || cut < best_cut) {
            best_cut = cut;
            best = s;
        }
    }
    side = best;
}

// bisection of g into weights (target0, rest): multilevel V-cycle
static void ml_bisect(const MlGraph &g, std::vector<char> &side, int64_t target0)
{
    std::vector<MlGraph> levels;
    std::vector<std::vector<int>> maps;
    const MlGraph *cur = &g;
    while (cur->n > 120) {
        std::vector<int> cmap;
        MlGraph c = ml_coarsen(*cur, cmap);
        if (c.n > cur->n * 0.93) break;  // matching no longer shrinks the graph
        maps.push_back(std::move(cmap));
        levels.push_back(std::move(c));
        cur = &levels.back();
    }
    std::vector<char> s;
    ml_initial(*cur, s, target0);
    for (int l = (int)levels.size() - 1; l >= 0; --l) {
        const MlGraph &fine = l == 0 ? g : levels[(size_t)l - 1];
        std::vector<char> sf((size_t)fine.n);
        for (int v = 0; v < fine.n; ++v) sf[(size_t)v] = s[(size_t)maps[(size_t)l][(size_t)v]];
        const int64_t slack = l == 0 ? 0 : *std::max_element(fine.vw.begin(), fine.vw.end());
        ml_refine(fine, sf, target0, slack, l == 0 ? 10 : 6);
        s.swap(sf);
    }
    if (levels.empty()) ml_refine(g, s, target0, 0, 10);
    side = s;
}

static void ml_recurse(const MlGraph &g, const std::vector<int64_t> &ids, int parts, uint32_t base, uint32_t *part)
{
    if (g.n == 0) return;
    if (parts == 1) {
        for (int64_t id : ids) part[id] = base;
        return;
    }
    const int left_parts = parts / 2;
    const int64_t target0 = (int64_t)((double)g.n * left_parts / parts + 0.5);
    std::vector<char> side;
    ml_bisect(g, side, target0);
    // exact sizes: whatever the refinement left over is moved by gain order (unit weights on this level)
    {
        int64_t w0 = 0;
        for (int v = 0; v < g.n; ++v) w0 += !side[(size_t)v];
        while (w0 != target0) {
            const int from = w0 > target0 ? 0 : 1;
            int pick = -1, pick_gain = INT_MIN;
            for (int v = 0; v < g.n; ++v) {
                if (side[(size_t)v] != from) continue;
                int gsum = 0;
                for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; ++j)
                    gsum += side[(size_t)g.adj[(size_t)j]] != from ? g.ew[(size_t)j] : -g.ew[(size_t)j];
                if (gsum > pick_gain) {
                    pick_gain = gsum;
                    pick = v;
                }
            }
            if (pick < 0) break;
            side[(size_t)pick] = (char)(1 - from);
            w0 += from == 0 ? -1 : 1;
        }
    }
    // the two induced subgraphs
    for (int sd = 0; sd < 2; ++sd) {
        std::vector<int> loc((size_t)g.n, -1);
        MlGraph sub;
        std::vector<int64_t> sub_ids;
        for (int v = 0; v < g.n; ++v)
            if (side[(size_t)v] == sd) {
                loc[(size_t)v] = sub.n++;
                sub_ids.push_back(ids[(size_t)v]);
            }
        sub.vw.assign((size_t)sub.n, 1);
        sub.xadj.assign((size_t)sub.n + 1, 0);
        for (int v = 0; v < g.n; ++v) {
            if (side[(size_t)v] != sd) continue;
            for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; ++j) {
                const int w = loc[(size_t)g.adj[(size_t)j]];
                if (w >= 0) {
                    sub.adj.push_back(w);
                    sub.ew.push_back(g.ew[(size_t)j]);
                }
            }
            sub.xadj[(size_t)loc[(size_t)v] + 1] = (int64_t)sub.adj.size();
        }
        ml_recurse(sub, sub_ids, sd == 0 ? left_parts : parts - left_parts, sd == 0 ? base : base + (uint32_t)left_parts, part);
    }
}

}  // namespace

int schwz_partition_graph(const schwz_problem *p, int P, uint32_t *part)
{
    SCHWZ_REQUIRE(p && part && P > 0, "schwz_partition_graph: bad arguments");
    static const bool multilevel = [] {
        const char *e = std::getenv("SCHWZ_PART_MULTILEVEL");
        return !(e && e[0] == '0');
    }();
    if (multilevel && p->N > 0 && p->N < INT32_MAX / 2) {
        // the matrix graph, symmetrised (an entry in either triangle is an edge), without the diagonal
        const int64_t N = p->N;
        std::vector<std::pair<int, int>> edges;
        {
            std::vector<int64_t> c((size_t)p->max_row_nnz + 1);
            std::vector<double> v((size_t)p->max_row_nnz + 1);
            for (int64_t u = 0; u < N; ++u) {
                const int len = p->row(u, c.data(), v.data());
                for (int k = 0; k < len; ++k)
                    if (c[k] != u) {
                        edges.push_back({(int)u, (int)c[k]});
                        edges.push_back({(int)c[k], (int)u});
                    }
            }
        }
        std::sort(edges.begin(), edges.end());
        edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
        MlGraph g;
        g.n = (int)N;
        g.vw.assign((size_t)N, 1);
        g.xadj.assign((size_t)N + 1, 0);
        g.adj.resize(edges.size());
        g.ew.assign(edges.size(), 1);
        for (size_t e = 0; e < edges.size(); ++e) {
            g.adj[e] = edges[e].second;
            ++g.xadj[(size_t)edges[e].first + 1];
        }
        for (int64_t u = 0; u < N; ++u) g.xadj[(size_t)u + 1] += g.xadj[(size_t)u];
        edges.clear();
        edges.shrink_to_fit();
        std::vector<int64_t> ids((size_t)N);
        std::iota(ids.begin(), ids.end(), 0);
        ml_recurse(g, ids, P, 0, part);
        return SCHWZ_OK;
    }
    std::vector<int64_t> nodes((size_t)p->N);
    std::iota(nodes.begin(), nodes.end(), 0);
    std::vector<int32_t> mark((size_t)p->N, 0);
    int32_t stamp = 0;
    if (p->N > 0) bisect(p, nodes, P, 0, part, mark, stamp);
    return SCHWZ_OK;
}

// ---------------------------------------------------------------------------
// subdomain index sets (restricted_schwarz.cpp:56-304) and get lists (:336-371)
// ---------------------------------------------------------------------------

int schwz_subdomain_setup(const schwz_problem *p, int P, int me, int overlap, const int64_t *first_row,
                          schwz_subdomain **out)
{
    SCHWZ_REQUIRE(p && out && first_row, "schwz_subdomain_setup: null argument");
    SCHWZ_REQUIRE(P > 0 && me >= 0 && me < P, "schwz_subdomain_setup: bad rank / subdomain count");
    SCHWZ_REQUIRE(overlap >= 1, "schwz_subdomain_setup: overlap must be >= 1");
    SCHWZ_REQUIRE(first_row[0] == 0 && first_row[P] == p->N, "schwz_subdomain_setup: first_row must cover all rows");
    StageTimer timer_all("subdomain_setup (index sets, local / interface matrices, get lists)");
    p->missing_row = -1;  // see schwz_problem_extract_rows
    auto *sd = new schwz_subdomain();
    sd->P = P;
    sd->me = me;
    sd->overlap = overlap;
    sd->N = p->N;
    sd->first_row.assign(first_row, first_row + P + 1);
    const int64_t lo = first_row[me], hi = first_row[me + 1];
    sd->local_size = hi - lo;
    if (sd->local_size >= INT32_MAX / 2) {
        delete sd;
        set_error("schwz_subdomain_setup: subdomain too large for int32 local indices");
        return SCHWZ_ERR_INVALID;
    }
    const int mr = p->max_row_nnz + 1;
    std::vector<int64_t> c((size_t)mr);
    std::vector<double> v((size_t)mr);
    auto &l2g = sd->l2g;
    // interior ids are implicit; l2g stores them anyway for the callers
    l2g.reserve((size_t)sd->local_size + (size_t)(sd->local_size / 8) + 1024);  // (room for the overlap and halo ids)
    l2g.resize((size_t)sd->local_size);  // no value-initialisation: the threads below touch the pages first
#pragma omp parallel for schedule(static) num_threads(schwz::setup_threads())
    for (int64_t i = 0; i < sd->local_size; ++i) l2g[(size_t)i] = lo + i;
    // overlap-1 BFS layers in discovery order (:166-180).  The candidates of a layer -- columns outside the
    // interior range -- are collected by all threads over contiguous blocks of the layer's rows and visited block
    // after block, i.e. in row order then column order like the sequential loop: the same discovery order.
    StageTimer t_bfs("  setup: overlap layers");
    int64_t old = 0;
    // An analytic stencil knows its bandwidth (nx, or nx * ny): an interior row can only reach outside [lo, hi) from
    // the first or the last `band` rows of the range, so the scan of the interior skips everything between them
    // (16.6 of 16.8 M rows at 256^3, where the scan used to be a quarter of this function's time).
    const int64_t band = p->kind == 2 ? p->nx : (p->kind == 3 ? p->nx * p->ny : -1);
    auto grow_layer = [&](int64_t from, int64_t to) {
        // the rows to scan, as up to three ascending ranges of local indices
        int64_t rb[3][2];
        int nr = 0;
        const int64_t ie = std::min(to, sd->local_size);  // end of the interior part of [from, to)
        if (from < ie) {
            if (band >= 0 && ie - from > 2 * band) {
                const int64_t low_end = std::max(from, std::min(ie, band));
                const int64_t high_begin = std::max(low_end, sd->local_size - band);
                if (from < low_end) rb[nr][0] = from, rb[nr][1] = low_end, ++nr;
                if (high_begin < ie) rb[nr][0] = high_begin, rb[nr][1] = ie, ++nr;
            } else {
                rb[nr][0] = from, rb[nr][1] = ie, ++nr;
            }
        }
        if (std::max(from, ie) < to) rb[nr][0] = std::max(from, ie), rb[nr][1] = to, ++nr;
        int64_t total = 0;
        for (int k = 0; k < nr; ++k) total += rb[k][1] - rb[k][0];
        auto row_at = [&](int64_t pos) {
            int k = 0;
            while (k + 1 < nr && pos >= rb[k][1] - rb[k][0]) pos -= rb[k][1] - rb[k][0], ++k;
            return rb[k][0] + pos;
        };
        int nthreads = schwz::setup_threads();
        if (total < 4096) nthreads = 1;
        std::vector<std::vector<int64_t>> cand((size_t)nthreads);
#pragma omp parallel num_threads(nthreads)
        {
            const int t = omp_get_thread_num();
            const int64_t a = total * t / nthreads, b = total * (t + 1) / nthreads;
            std::vector<int64_t> cc((size_t)mr);
            std::vector<double> vv((size_t)mr);
            auto &mine = cand[(size_t)t];
            for (int64_t q = a; q < b; ++q) {
                const int len = p->row(l2g[(size_t)row_at(q)], cc.data(), vv.data());
                for (int j = 0; j < len; ++j)
                    if (cc[j] < lo || cc[j] >= hi) mine.push_back(cc[j]);
            }
        }
        for (const auto &lst : cand)
            for (int64_t g : lst)
                if (sd->g2l_x.count(g) == 0) {
                    sd->g2l_x.emplace(g, (schwz_idx)l2g.size());
                    l2g.push_back(g);
                }
    };
    for (int k = 1; k < overlap; ++k) {
        const int64_t now = (int64_t)l2g.size();
        grow_layer(old, now);
        old = now;
    }
    t_bfs.stop();
    StageTimer t_cnt("  setup: row counts");
    sd->local_size_x = (int64_t)l2g.size();
    sd->overlap_size = sd->local_size_x - sd->local_size;
    const int64_t n = sd->local_size_x;

    // local / interface split (:194-284).  Entries of interior rows whose column
    // is unmapped ("invalid edge", only possible for overlap == 1) are dropped
    // like the reference does.
    sd->l_rp.resize((size_t)n + 1);
    sd->i_rp.resize((size_t)n + 1);
#pragma omp parallel for schedule(static) num_threads(schwz::setup_threads())
    for (int64_t r = 0; r <= n; ++r) sd->l_rp[(size_t)r] = sd->i_rp[(size_t)r] = 0;
    std::vector<schwz_idx> lc((size_t)mr);
    std::vector<double> lv((size_t)mr);
    int64_t nnz_l = 0;
    {
        // entries per row (all threads), then the running sum
        std::vector<int> cnt((size_t)n, 0);
#pragma omp parallel for schedule(static) firstprivate(c, v) num_threads(schwz::setup_threads())
        for (int64_t r = 0; r < n; ++r) {
            const int len = p->row(l2g[(size_t)r], c.data(), v.data());
            int k = 0;
            for (int j = 0; j < len; ++j) k += sd->to_local(c[j]) >= 0;
            cnt[(size_t)r] = k;
        }
        for (int64_t r = 0; r < n; ++r) {
            nnz_l += cnt[(size_t)r];
            if (nnz_l >= INT32_MAX) {
                delete sd;
                set_error("schwz_subdomain_setup: local matrix exceeds 2^31-1 nonzeros");
                return SCHWZ_ERR_INVALID;
            }
            sd->l_rp[(size_t)r + 1] = (schwz_idx)nnz_l;
        }
    }
    t_cnt.stop();
    StageTimer t_fill("  setup: local matrix fill");
    sd->l_col.resize((size_t)nnz_l);
    sd->l_val.resize((size_t)nnz_l);
#pragma omp parallel for schedule(static) firstprivate(c, v) num_threads(schwz::setup_threads())
    for (int64_t r = 0; r < n; ++r) {
        const int len = p->row(l2g[(size_t)r], c.data(), v.data());
        int64_t pos = sd->l_rp[(size_t)r];
        const int64_t start = pos;
        for (int j = 0; j < len; ++j) {
            const schwz_idx loc = sd->to_local(c[j]);
            if (loc >= 0) {
                sd->l_col[(size_t)pos] = loc;
                sd->l_val[(size_t)pos] = v[j];
                ++pos;
            }
        }
        // sort_by_column_index (:297)
        for (int64_t a = start + 1; a < pos; ++a) {
            const schwz_idx cc = sd->l_col[(size_t)a];
            const double vv = sd->l_val[(size_t)a];
            int64_t b = a;
            while (b > start && sd->l_col[(size_t)b - 1] > cc) {
                sd->l_col[(size_t)b] = sd->l_col[(size_t)b - 1];
                sd->l_val[(size_t)b] = sd->l_val[(size_t)b - 1];
                --b;
            }
            sd->l_col[(size_t)b] = cc;
            sd->l_val[(size_t)b] = vv;
        }
    }
    t_fill.stop();
    StageTimer t_rest("  setup: interface rows, halo, get lists");
    // interface entries: overlap rows only, GLOBAL columns, ascending (:262-284,298)
    for (int64_t r = sd->local_size; r < n; ++r) {
        const int len = p->row(l2g[(size_t)r], c.data(), v.data());
        int cnt = 0;
        for (int j = 0; j < len; ++j) {
            if (sd->to_local(c[j]) < 0) {
                sd->i_col_global.push_back(c[j]);
                sd->i_val.push_back(v[j]);
                ++cnt;
            }
        }
        sort_row(sd->i_col_global.data() + sd->i_col_global.size() - cnt,
                 sd->i_val.data() + sd->i_val.size() - cnt, cnt);
        sd->i_rp[(size_t)r + 1] = (schwz_idx)sd->i_col_global.size();
    }
    for (int64_t r = 0; r < sd->local_size; ++r) sd->i_rp[(size_t)r + 1] = 0;
    // halo marking: one more BFS step from the last layer (:285-295)
    grow_layer(old, (int64_t)l2g.size());
    sd->halo_size = (int64_t)l2g.size() - sd->local_size_x;

    // get lists (:336-371): every mapped id owned by p, ascending.  Only the
    // non-interior ids can belong to another rank, so sort those once.
    std::vector<int64_t> ext(l2g.begin() + sd->local_size, l2g.end());
    std::sort(ext.begin(), ext.end());
    size_t pos = 0;
    for (int q = 0; q < P; ++q) {
        if (q == me) {
            while (pos < ext.size() && ext[pos] < first_row[q + 1]) ++pos;
            continue;
        }
        std::vector<int64_t> lst;
        while (pos < ext.size() && ext[pos] < first_row[q + 1]) lst.push_back(ext[pos++]);
        if (!lst.empty()) {
            sd->num_recv += (int64_t)lst.size();
            sd->nbr_in.push_back(q);
            sd->get.push_back(std::move(lst));
        }
    }
    if (p->missing_row >= 0) {  // a row source that holds part of the matrix only (schwz_problem_from_rows)
        delete sd;
        set_error("schwz_subdomain_setup: the problem does not hold a row this subdomain reads (row " +
                  std::to_string(p->missing_row) + ")");
        p->missing_row = -1;
        return SCHWZ_ERR_INVALID;
    }
    *out = sd;
    return SCHWZ_OK;
}

int schwz_subdomain_sizes(const schwz_subdomain *sd, int64_t *s)
{
    SCHWZ_REQUIRE(sd && s, "schwz_subdomain_sizes: null argument");
    s[0] = sd->local_size;
    s[1] = sd->local_size_x;
    s[2] = sd->overlap_size;
    s[3] = sd->halo_size;
    s[4] = (int64_t)sd->l_col.size();
    s[5] = (int64_t)sd->i_col_global.size();
    s[6] = (int64_t)sd->nbr_in.size();
    s[7] = (int64_t)sd->nbr_out.size();
    s[8] = sd->num_recv;
    s[9] = sd->num_send;
    return SCHWZ_OK;
}

int schwz_subdomain_local_to_global(const schwz_subdomain *sd, int64_t *out)
{
    SCHWZ_REQUIRE(sd && out, "schwz_subdomain_local_to_global: null argument");
    std::copy(sd->l2g.begin(), sd->l2g.end(), out);
    return SCHWZ_OK;
}

int schwz_subdomain_local_matrix(const schwz_subdomain *sd, schwz_idx *rp, schwz_idx *col, double *val)
{
    SCHWZ_REQUIRE(sd && rp && (col || sd->l_col.empty()), "schwz_subdomain_local_matrix: null argument");
    std::copy(sd->l_rp.begin(), sd->l_rp.end(), rp);
    std::copy(sd->l_col.begin(), sd->l_col.end(), col);
    std::copy(sd->l_val.begin(), sd->l_val.end(), val);
    return SCHWZ_OK;
}

int schwz_subdomain_interface_matrix(const schwz_subdomain *sd, schwz_idx *rp, int64_t *col, double *val)
{
    SCHWZ_REQUIRE(sd && rp, "schwz_subdomain_interface_matrix: null argument");
    std::copy(sd->i_rp.begin(), sd->i_rp.end(), rp);
    std::copy(sd->i_col_global.begin(), sd->i_col_global.end(), col);
    std::copy(sd->i_val.begin(), sd->i_val.end(), val);
    return SCHWZ_OK;
}

static int list_out(const std::vector<int> &ranks, const std::vector<std::vector<int64_t>> &lists, int k,
                    int *rank, int64_t *count, int64_t *ids)
{
    SCHWZ_REQUIRE(k >= 0 && k < (int)ranks.size(), "neighbour index out of range");
    if (rank) *rank = ranks[(size_t)k];
    if (count) *count = (int64_t)lists[(size_t)k].size();
    if (ids) std::copy(lists[(size_t)k].begin(), lists[(size_t)k].end(), ids);
    return SCHWZ_OK;
}

int schwz_subdomain_get_list(const schwz_subdomain *sd, int k, int *rank, int64_t *count, int64_t *ids)
{
    SCHWZ_REQUIRE(sd, "schwz_subdomain_get_list: null argument");
    return list_out(sd->nbr_in, sd->get, k, rank, count, ids);
}

int schwz_subdomain_put_list(const schwz_subdomain *sd, int k, int *rank, int64_t *count, int64_t *ids)
{
    SCHWZ_REQUIRE(sd, "schwz_subdomain_put_list: null argument");
    return list_out(sd->nbr_out, sd->put, k, rank, count, ids);
}

int schwz_subdomain_add_put_list(schwz_subdomain *sd, int p, int64_t count, const int64_t *ids)
{
    SCHWZ_REQUIRE(sd && (ids || count == 0), "schwz_subdomain_add_put_list: null argument");
    SCHWZ_REQUIRE(!sd->on_device, "schwz_subdomain_add_put_list: lists are frozen after to_device");
    SCHWZ_REQUIRE(p >= 0 && p < sd->P && p != sd->me, "schwz_subdomain_add_put_list: bad neighbour rank");
    SCHWZ_REQUIRE(sd->nbr_out.empty() || sd->nbr_out.back() < p,
                  "schwz_subdomain_add_put_list: neighbours must be added in ascending rank order");
    if (count <= 0) return SCHWZ_OK;
    const int64_t lo = sd->first_row[sd->me], hi = sd->first_row[sd->me + 1];
    for (int64_t i = 0; i < count; ++i)
        SCHWZ_REQUIRE(ids[i] >= lo && ids[i] < hi, "schwz_subdomain_add_put_list: id not owned by this subdomain");
    sd->nbr_out.push_back(p);
    sd->put.emplace_back(ids, ids + count);
    sd->num_send += count;
    return SCHWZ_OK;
}

static int offset_of(const std::vector<std::vector<int64_t>> &lists, int k, int64_t *off)
{
    SCHWZ_REQUIRE(off && k >= 0 && k <= (int)lists.size(), "neighbour index out of range");
    int64_t o = 0;
    for (int i = 0; i < k; ++i) o += (int64_t)lists[(size_t)i].size();
    *off = o;
    return SCHWZ_OK;
}

int schwz_subdomain_send_offset(const schwz_subdomain *sd, int k, int64_t *off)
{
    SCHWZ_REQUIRE(sd, "schwz_subdomain_send_offset: null argument");
    return offset_of(sd->put, k, off);
}

int schwz_subdomain_recv_offset(const schwz_subdomain *sd, int k, int64_t *off)
{
    SCHWZ_REQUIRE(sd, "schwz_subdomain_recv_offset: null argument");
    return offset_of(sd->get, k, off);
}

// ---------------------------------------------------------------------------
// sparse LL^T (left-looking, column by column, with linked row lists)
// ---------------------------------------------------------------------------

// Cuthill-McKee reversed; any fill-reducing ordering is admissible here (CHOLMOD
// would choose AMD, solve.cpp:102-105 only lets the user force the natural one).
static void rcm(int64_t n, const schwz_idx *rp, const schwz_idx *col, std::vector<schwz_idx> &perm)
{
    perm.clear();
    perm.reserve((size_t)n);
    std::vector<char> seen((size_t)n, 0);
    std::vector<schwz_idx> nb;
    auto deg = [&](schwz_idx i) { return rp[i + 1] - rp[i]; };
    size_t head = 0;
    while ((int64_t)perm.size() < n) {
        schwz_idx s = -1;
        for (int64_t i = 0; i < n; ++i)
            if (!seen[(size_t)i] && (s < 0 || deg((schwz_idx)i) < deg(s))) s = (schwz_idx)i;
        seen[(size_t)s] = 1;
        perm.push_back(s);
        while (head < perm.size()) {
            const schwz_idx u = perm[head++];
            nb.clear();
            for (schwz_idx j = rp[u]; j < rp[u + 1]; ++j) {
                const schwz_idx w = col[j];
                if (w != u && !seen[(size_t)w]) {
                    seen[(size_t)w] = 1;
                    nb.push_back(w);
                }
            }
            std::sort(nb.begin(), nb.end(), [&](schwz_idx a, schwz_idx b) {
                return deg(a) != deg(b) ? deg(a) < deg(b) : a < b;
            });
            perm.insert(perm.end(), nb.begin(), nb.end());
        }
    }
    std::reverse(perm.begin(), perm.end());
}

int schwz_cholesky(int64_t n, const schwz_idx *rp, const schwz_idx *col, const double *val, int natural,
                   schwz_idx **l_rp_o, schwz_idx **l_col_o, double **l_val_o, schwz_idx **u_rp_o,
                   schwz_idx **u_col_o, double **u_val_o, schwz_idx **perm_o)
{
    SCHWZ_REQUIRE(rp && l_rp_o && l_col_o && l_val_o && u_rp_o && u_col_o && u_val_o && perm_o && n >= 0,
                  "schwz_cholesky: bad arguments");
    std::vector<schwz_idx> perm;
    if (natural) {
        perm.resize((size_t)n);
        std::iota(perm.begin(), perm.end(), 0);
    } else {
        rcm(n, rp, col, perm);
    }
    std::vector<schwz_idx> iperm((size_t)n);
    for (int64_t i = 0; i < n; ++i) iperm[(size_t)perm[(size_t)i]] = (schwz_idx)i;

    // Column j of L is built from column j of B = A(perm,perm) (rows >= j) minus
    // the contributions of every earlier column k with L(j,k) != 0.  `next_in_row`
    // chains the columns that currently have their next unprocessed entry in row j.
    std::vector<std::vector<schwz_idx>> lcol_rows((size_t)n);  // row ids per column (ascending)
    std::vector<std::vector<double>> lcol_vals((size_t)n);
    std::vector<schwz_idx> cursor((size_t)n, 0);       // position of next entry below the diagonal
    std::vector<schwz_idx> head((size_t)n, -1), nxt((size_t)n, -1);
    std::vector<double> w((size_t)n, 0.0);
    std::vector<char> inpat((size_t)n, 0);
    std::vector<schwz_idx> pat;
    for (int64_t j = 0; j < n; ++j) {
        pat.clear();
        // scatter column j of B (lower part): entries (i, j) with i >= j come
        // from row perm[j] of A by symmetry
        const schwz_idx aj = perm[(size_t)j];
        for (schwz_idx t = rp[aj]; t < rp[aj + 1]; ++t) {
            const schwz_idx i = iperm[(size_t)col[t]];
            if (i >= j) {
                if (!inpat[(size_t)i]) {
                    inpat[(size_t)i] = 1;
                    pat.push_back(i);
                }
                w[(size_t)i] += val[t];
            }
        }
        if (!inpat[(size_t)j]) {
            inpat[(size_t)j] = 1;
            pat.push_back((schwz_idx)j);
        }
        // subtract L(j:n,k) * L(j,k) for every k in row j's list
        schwz_idx k = head[(size_t)j];
        while (k != -1) {
            const schwz_idx knext = nxt[(size_t)k];
            const auto &rows = lcol_rows[(size_t)k];
            const auto &vals = lcol_vals[(size_t)k];
            const schwz_idx c0 = cursor[(size_t)k];  // rows[c0] == j
            const double ljk = vals[(size_t)c0];
            for (size_t t = (size_t)c0; t < rows.size(); ++t) {
                const schwz_idx i = rows[t];
                if (!inpat[(size_t)i]) {
                    inpat[(size_t)i] = 1;
                    pat.push_back(i);
                }
                w[(size_t)i] -= vals[t] * ljk;
            }
            // advance column k to its next row
            cursor[(size_t)k] = c0 + 1;
            if ((size_t)(c0 + 1) < rows.size()) {
                const schwz_idx r = rows[(size_t)c0 + 1];
                nxt[(size_t)k] = head[(size_t)r];
                head[(size_t)r] = k;
            }
            k = knext;
        }
        std::sort(pat.begin(), pat.end());
        const double d = w[(size_t)j];
        if (!(d > 0.0)) {
            set_error("schwz_cholesky: matrix is not positive definite");
            return SCHWZ_ERR_NOT_SPD;
        }
        const double ljj = std::sqrt(d);
        auto &rows = lcol_rows[(size_t)j];
        auto &vals = lcol_vals[(size_t)j];
        rows.reserve(pat.size());
        vals.reserve(pat.size());
        for (schwz_idx i : pat) {
            rows.push_back(i);
            vals.push_back(i == j ? ljj : w[(size_t)i] / ljj);
            w[(size_t)i] = 0.0;
            inpat[(size_t)i] = 0;
        }
        cursor[(size_t)j] = 1;  // rows[0] is the diagonal
        if (rows.size() > 1) {
            const schwz_idx r = rows[1];
            nxt[(size_t)j] = head[(size_t)r];
            head[(size_t)r] = (schwz_idx)j;
        }
    }
    // U = L^T in CSR = columns of L (diagonal first); L in CSR = its transpose
    // (solve.cpp:288-304)
    std::vector<schwz_idx> u_rp((size_t)n + 1, 0), l_rp((size_t)n + 1, 0);
    for (int64_t j = 0; j < n; ++j) {
        u_rp[(size_t)j + 1] = u_rp[(size_t)j] + (schwz_idx)lcol_rows[(size_t)j].size();
        for (schwz_idx i : lcol_rows[(size_t)j]) l_rp[(size_t)i + 1]++;
    }
    for (int64_t i = 0; i < n; ++i) l_rp[(size_t)i + 1] += l_rp[(size_t)i];
    const size_t lnz = (size_t)u_rp[(size_t)n];
    std::vector<schwz_idx> u_col(lnz), l_col(lnz);
    std::vector<double> u_val(lnz), l_val(lnz);
    std::vector<schwz_idx> fill(l_rp.begin(), l_rp.end() - 1);
    for (int64_t j = 0; j < n; ++j) {
        size_t o = (size_t)u_rp[(size_t)j];
        for (size_t t = 0; t < lcol_rows[(size_t)j].size(); ++t, ++o) {
            const schwz_idx i = lcol_rows[(size_t)j][t];
            u_col[o] = i;
            u_val[o] = lcol_vals[(size_t)j][t];
            l_col[(size_t)fill[(size_t)i]] = (schwz_idx)j;
            l_val[(size_t)fill[(size_t)i]] = lcol_vals[(size_t)j][t];
            fill[(size_t)i]++;
        }
    }
    *l_rp_o = to_malloc(l_rp);
    *l_col_o = to_malloc(l_col);
    *l_val_o = to_malloc(l_val);
    *u_rp_o = to_malloc(u_rp);
    *u_col_o = to_malloc(u_col);
    *u_val_o = to_malloc(u_val);
    *perm_o = to_malloc(perm);
    return SCHWZ_OK;
}

// ISAI of a triangular factor on its own pattern (Anzt, Huckle, Braeckle, Dongarra 2018): per row
// one small triangular substitution with T(S,S), S the row's pattern; rows are independent.
int schwz_isai(int64_t n, const schwz_idx *rp, const schwz_idx *col, const double *val, int lower, double **w_val)
{
    if (!rp || !w_val || n < 0 || (n > 0 && (!col || !val))) {
        set_error("schwz_isai: bad arguments");
        return SCHWZ_ERR_INVALID;
    }
    const int64_t nnz = n ? rp[n] : 0;
    std::vector<double> w((size_t)nnz);
    auto entry = [&](schwz_idx r, schwz_idx c) {
        const schwz_idx *b = col + rp[r], *e = col + rp[r + 1];
        const schwz_idx *it = std::lower_bound(b, e, c);
        return (it != e && *it == c) ? val[it - col] : 0.0;
    };
    bool singular = false;
#pragma omp parallel for schedule(dynamic, 1024) reduction(|| : singular) num_threads(schwz::setup_threads())
    for (int64_t i = 0; i < n; ++i) {
        const schwz_idx s0 = rp[i], k = rp[i + 1] - rp[i];
        const schwz_idx *S = col + s0;
        double *wi = w.data() + s0;
        if (lower) {
            for (schwz_idx j = k - 1; j >= 0; --j) {
                double acc = (S[j] == (schwz_idx)i) ? 1.0 : 0.0;
                for (schwz_idx a = j + 1; a < k; ++a) acc -= wi[a] * entry(S[a], S[j]);
                const double d = entry(S[j], S[j]);
                if (d == 0.0) singular = true;
                wi[j] = acc / d;
            }
        } else {
            for (schwz_idx j = 0; j < k; ++j) {
                double acc = (S[j] == (schwz_idx)i) ? 1.0 : 0.0;
                for (schwz_idx a = 0; a < j; ++a) acc -= wi[a] * entry(S[a], S[j]);
                const double d = entry(S[j], S[j]);
                if (d == 0.0) singular = true;
                wi[j] = acc / d;
            }
        }
    }
    if (singular) {
        set_error("schwz_isai: zero diagonal in the triangular factor");
        return SCHWZ_ERR_INVALID;
    }
    *w_val = to_malloc(w);
    return SCHWZ_OK;
}

// ILU(0), row by row (IKJ): for every k < i in row i: l_ik = a_ik / u_kk, then row i loses
// l_ik * (row k of U) on the positions it stores.
int schwz_ilu0(int64_t n, const schwz_idx *rp, const schwz_idx *col, const double *val, schwz_idx **l_rp_o,
               schwz_idx **l_col_o, double **l_val_o, schwz_idx **u_rp_o, schwz_idx **u_col_o, double **u_val_o)
{
    SCHWZ_REQUIRE(rp && l_rp_o && l_col_o && l_val_o && u_rp_o && u_col_o && u_val_o && n >= 0,
                  "schwz_ilu0: bad arguments");
    const int64_t nnz = rp[n];
    std::vector<double> a(val, val + nnz);
    std::vector<schwz_idx> diag((size_t)n, -1), where((size_t)n, -1);
    for (int64_t i = 0; i < n; ++i) {
        for (schwz_idx j = rp[i]; j < rp[i + 1]; ++j) {
            SCHWZ_REQUIRE(j == rp[i] || col[j - 1] < col[j], "schwz_ilu0: columns must be sorted and unique");
            where[(size_t)col[j]] = j;
            if (col[j] == i) diag[(size_t)i] = j;
        }
        SCHWZ_REQUIRE(diag[(size_t)i] >= 0, "schwz_ilu0: structurally zero diagonal");
        for (schwz_idx kk = rp[i]; kk < rp[i + 1] && col[kk] < i; ++kk) {
            const schwz_idx k = col[kk];
            a[(size_t)kk] /= a[(size_t)diag[(size_t)k]];
            const double lik = a[(size_t)kk];
            for (schwz_idx j = diag[(size_t)k] + 1; j < rp[k + 1]; ++j) {
                const schwz_idx q = where[(size_t)col[j]];
                if (q >= 0) a[(size_t)q] -= lik * a[(size_t)j];
            }
        }
        if (a[(size_t)diag[(size_t)i]] == 0.0) {
            set_error("schwz_ilu0: zero pivot");
            return SCHWZ_ERR_NOT_SPD;
        }
        for (schwz_idx j = rp[i]; j < rp[i + 1]; ++j) where[(size_t)col[j]] = -1;
    }
    std::vector<schwz_idx> l_rp((size_t)n + 1, 0), u_rp((size_t)n + 1, 0), l_col, u_col;
    std::vector<double> l_val, u_val;
    l_col.reserve((size_t)nnz / 2 + (size_t)n);
    l_val.reserve((size_t)nnz / 2 + (size_t)n);
    u_col.reserve((size_t)nnz / 2 + (size_t)n);
    u_val.reserve((size_t)nnz / 2 + (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        for (schwz_idx j = rp[i]; j < rp[i + 1]; ++j) {
            if (col[j] < i) {
                l_col.push_back(col[j]);
                l_val.push_back(a[(size_t)j]);
            } else {
                u_col.push_back(col[j]);
                u_val.push_back(a[(size_t)j]);
            }
        }
        l_col.push_back((schwz_idx)i);
        l_val.push_back(1.0);
        l_rp[(size_t)i + 1] = (schwz_idx)l_col.size();
        u_rp[(size_t)i + 1] = (schwz_idx)u_col.size();
    }
    *l_rp_o = to_malloc(l_rp);
    *l_col_o = to_malloc(l_col);
    *l_val_o = to_malloc(l_val);
    *u_rp_o = to_malloc(u_rp);
    *u_col_o = to_malloc(u_col);
    *u_val_o = to_malloc(u_val);
    return SCHWZ_OK;
}

}  // extern "C"

// ---- host-side windows in shared memory: the convergence / residual windows of the one-sided mode ----
// (MPI_Accumulate / MPI_Put on window_convergence and window_residual_vector, include/conv_tools.hpp:56-275,
// between rank processes of one node: the "window" is a POSIX shared-memory segment every rank maps)
extern "C" {

int32_t schwz_host_atomic_add_i32(int32_t *p, int32_t v) { return __atomic_add_fetch(p, v, __ATOMIC_SEQ_CST); }

int32_t schwz_host_atomic_load_i32(const int32_t *p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }

void schwz_host_atomic_store_i32(int32_t *p, int32_t v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }

// *p = min(*p, v) (MPI_Accumulate with MPI_MIN); returns the value left in place
double schwz_host_atomic_min_f64(double *p, double v)
{
    uint64_t *q = reinterpret_cast<uint64_t *>(p);
    uint64_t old = __atomic_load_n(q, __ATOMIC_SEQ_CST);
    for (;;) {
        double cur;
        std::memcpy(&cur, &old, 8);
        if (!(v < cur)) return cur;
        uint64_t want;
        std::memcpy(&want, &v, 8);
        if (__atomic_compare_exchange_n(q, &old, want, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return v;
    }
}

}  // extern "C"
