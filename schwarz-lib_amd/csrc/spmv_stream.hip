// Plain-CSR SpMV as a straight-line software pipeline (variants 0 / 6 on matrices whose rows hold at most 32
// entries): the same tiles, the same products and the same per-row summation order as spmv_tiled2_kernel
// (spmv_csr.hip) -- the same bits --, but nothing in a workgroup's tile loop waits for a chain of dependent
// loads any more.
//
// What bounded spmv_tiled2_kernel (0.37 ms at 256^3 = 0.58 of the 8 TB/s peak while its matrix stream alone
// runs at 6.05 TB/s): per tile a workgroup went through tile table -> row pointers -> val / col -> x gather ->
// LDS -> row sums, five dependent memory round trips of 2-3 us each under load, with 8 workgroups per CU to
// hide them, and every guarded load (`if (row < r1)`) made hipcc fall back to `s_waitcnt vmcnt(0)`.  Here
//   * the tile bounds come from two tables indexed by the tile number alone (tile_row, tile_nz = rp[tile_row]),
//     read two tiles ahead with scalar loads;
//   * the val / col quads, the lane's row bounds and the operands of the fused epilogue of tile k + 2 are
//     requested as soon as the products of tile k are in LDS, into the register set tile k has just released
//     (two sets, loop unrolled by two: no register copies), and fly during the barrier, the row sums, the
//     store and the next tile's gathers;
//   * there is no branch in the loop: addresses are clamped instead of guarded (lanes past the tile's last
//     row repeat its last row and store the same value to the same address), the row sum is CAP masked adds
//     (an absent entry adds +0.0, which leaves a sum that began at +0.0 unchanged bit for bit).
// Workgroups are SHORT-LIVED (round 3): a workgroup takes a.seq = 3 consecutive tiles of its XCD's sequence and
// ends, and the grid covers the matrix once.  The round-2 form (persistent workgroups striding through the
// sequence, a.seq = 0, SCHWZ_STREAM_SEQ=0) lets the workgroups drift apart: the rows in flight spread over the
// whole sweep, the x lines neighbouring tiles share are gone from the 4 MiB L2 (a line lives ~5 us there at
// this stream rate) before the neighbour asks, and the y stores reach memory as scattered 2 KiB pieces.  With
// workgroups that end after four tiles the dispatch order keeps the active rows a compact moving window:
// 0.359 -> 0.322 ms at 256^3 and 0.380 -> 0.320 ms on the 512 x 512 x 64 slab together with non-temporal y
// stores (profiles/r03_stream_seq.txt; two tiles per workgroup pay too much pipeline fill, eight drift again).
// The y values are the same bits as before; the partial sums of the fused dots are per workgroup, and a grid
// beyond the kMaxGrid slots the consumers fold is reduced to them by spmv_stream_fold_kernel in a fixed order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "schwz_internal.hpp"
#include "device_utils.hpp"

namespace schwz {

// ABL (measurement builds, variants 80 + ABL of schwz_csr_spmv): bit 0 no y store, bit 1 no x gather, bit 2 no
// row-pointer loads (7-entry rows assumed), bit 3 y stored into a 256 KiB window (the stores are issued, the
// bytes stay in L2), bit 4 x gathered from a 512 KiB window (the gathers are issued and hit L2), bit 5 the y
// store of a tile issued after the NEXT tile's gathers (the in-order vmcnt wait for those gathers then does
// not wait for the store's acknowledgement)
template <int MODE, int CAP, int ABL = 0, bool NTY = false>
__global__ __launch_bounds__(kBlock) void spmv_stream_kernel(CsrView A, SpmvArgs a)
{
#pragma clang fp contract(off)
    // the tile's values and columns as stored (16-byte aligned window of <= kTileNnz entries, + what the
    // masked reads of the last rows may touch)
    __shared__ __attribute__((aligned(16))) double vals[kTileNnz + 4 + CAP];
    __shared__ __attribute__((aligned(16))) int cols[kTileNnz + 4 + CAP];
    __shared__ int rstart[kBlock + 1];  // first entry of every row of the tile (window-relative)
    __shared__ double red[4];
    if (MODE == kSpmvDot || MODE == kSpmvResidInit) {
        if (a.stop_iter && a.it >= *a.stop_iter) return;
    }
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x % kXcds;
    const int slot = blockIdx.x / kXcds;
    const int per_xcd = gridDim.x / kXcds;
    // an XCD's tile sequence: block-cyclic, or the explicit one of wide planes (CsrView::stream_order: entries past
    // the end are -1, mapped to ntiles = "no tile" here); read through the scalar cache
    typedef const int __attribute__((address_space(4))) *const_order;
    const const_order sorder = (const_order)(uintptr_t)A.stream_order;
    const int nslots = sorder ? A.stream_nper : xcd_slots(A);
    const int sh = A.xcd_shift;
    auto tile_at = [&](int j) -> int {
        if (sorder) {
            const int t = sorder[(int64_t)xcd * A.stream_nper + j];
            return t < 0 ? A.ntiles : t;
        }
        return ((j >> sh) << (sh + 3)) + (xcd << sh) + (j & (A.xcd_block - 1));
    };
    // tiles of this workgroup: k = 0 .. ntw - 1 (past-the-end slots only occur at the tail of the deal)
    // a.seq > 0: a workgroup takes a.seq CONSECUTIVE slots of its XCD's sequence and ends; the grid covers the
    // matrix once and the dispatch order keeps the active rows a compact window
    const int seq = a.seq;
    auto slot_at = [&](int k) -> int { return seq ? slot * seq + k : slot + k * per_xcd; };
    int ntw = seq ? max(0, min(seq, nslots - slot * seq)) : (slot < nslots ? (nslots - slot + per_xcd - 1) / per_xcd : 0);
    while (ntw > 0 && tile_at(slot_at(ntw - 1)) >= A.ntiles) --ntw;
    // partial-sum slots the consumer folds (a.part_stride of them per bank) beyond this launch's grid
    if (MODE != kSpmvPlain && blockIdx.x == 0)
        for (int i = (int)gridDim.x + tid; i < a.part_stride; i += kBlock) a.partials[i] = a.partials[a.part_stride + i] = 0.0;
    // the masked reads of a tile's last rows may run up to CAP entries past the window: keep that tail finite
    // and inside x (it is never summed, but its columns are gathered)
    if (tid < 4 + CAP) {
        vals[kTileNnz + tid] = 0.0;
        cols[kTileNnz + tid] = 0;
    }
    double acc0 = 0.0, acc1 = 0.0;
    if (ntw > 0) {
        typedef const int __attribute__((address_space(4))) *const_ints;
        const const_ints trow = (const_ints)(uintptr_t)A.tile_row;
        const const_ints tnz = (const_ints)(uintptr_t)A.tile_nz;
        struct Meta {
            int r0, r1, s, e;
        };
        auto meta = [&](int k) -> Meta {
            // beyond the workgroup's last tile: loaded (no branch in the loop), never computed -- one quad and
            // one row of the last tile, so that every such load instruction touches a single line
            const int t = tile_at(slot_at(min(k, ntw - 1)));
            const bool past = k >= ntw;
            const int r0 = trow[t], s0 = tnz[t];
            return Meta{r0, past ? r0 + 1 : trow[t + 1], s0, past ? s0 : tnz[t + 1]};
        };
        // Two register sets (tiles k and k + 1 in flight / landed), named scalars and native vectors pasted
        // into the step by macro: as members of a struct passed to a lambda they stayed in scratch memory
        // (the allocas were not promoted), which doubled the kernel time.
        typedef double vd2 __attribute__((ext_vector_type(2)));
        typedef int vi4 __attribute__((ext_vector_type(4)));
        const double *const dinv = a.dinv ? a.dinv : a.b;  // a valid address either way (the value is selected away)
#define SCHWZ_STREAM_ISSUE(M, P)                                                                     \
    {                                                                                                \
        const int s2_ = (M).s & ~3;                                                                  \
        const int last_ = max(((M).e - 1) & ~3, s2_);                                                \
        const int i0_ = min(s2_ + 4 * tid, last_), i1_ = min(s2_ + 4 * (tid + kBlock), last_);       \
        if (ABL & 64) {                                                                              \
            P##v0 = __builtin_nontemporal_load(reinterpret_cast<const vd2 *>(A.val + i0_));          \
            P##v1 = __builtin_nontemporal_load(reinterpret_cast<const vd2 *>(A.val + i0_ + 2));      \
            P##c0 = __builtin_nontemporal_load(reinterpret_cast<const vi4 *>(A.col + i0_));          \
            P##v2 = __builtin_nontemporal_load(reinterpret_cast<const vd2 *>(A.val + i1_));          \
            P##v3 = __builtin_nontemporal_load(reinterpret_cast<const vd2 *>(A.val + i1_ + 2));      \
            P##c1 = __builtin_nontemporal_load(reinterpret_cast<const vi4 *>(A.col + i1_));          \
        } else {                                                                                     \
            P##v0 = *reinterpret_cast<const vd2 *>(A.val + i0_);                                     \
            P##v1 = *reinterpret_cast<const vd2 *>(A.val + i0_ + 2);                                 \
            P##c0 = *reinterpret_cast<const vi4 *>(A.col + i0_);                                     \
            P##v2 = *reinterpret_cast<const vd2 *>(A.val + i1_);                                     \
            P##v3 = *reinterpret_cast<const vd2 *>(A.val + i1_ + 2);                                 \
            P##c1 = *reinterpret_cast<const vi4 *>(A.col + i1_);                                     \
        }                                                                                            \
        const int rowc_ = min((M).r0 + tid, (M).r1 - 1);                                             \
        /* ONE row-pointer load per lane: the end of a row is the start of the next lane's, handed */ \
        /* over through LDS in the step; the tile's last row ends where the tile does (M.e) */      \
        P##b0 = (ABL & 4) ? 7 * tid : A.rp[rowc_] - s2_;                                             \
        P##o0 = P##o1 = 0.0;                                                                         \
        if (MODE == kSpmvDot) P##o0 = a.x[rowc_];                                                    \
        if (MODE == kSpmvResidInit || MODE == kSpmvResidNorm) P##o0 = a.b[rowc_];                   \
        if (MODE == kSpmvResidInit) P##o1 = dinv[rowc_];                                             \
    }
        // One tile: set P holds its stream on entry and the stream of tile MN (two tiles on) on exit.  The
        // entries go to LDS as they are stored; then a lane IS a row: entry j of 64 consecutive rows per
        // gather instruction (for a stencil: one contiguous run of x), products rounded one by one and
        // added in CSR order.  The next stream is requested AFTER the first eight gathers (the scheduling
        // barriers keep hipcc from moving it up), so that waiting for them -- an in-order counter -- does
        // not wait for it.  The last two steps of a workgroup request its last tile again: no branch.
        // (Keeping the gathered operands of a tile for a step and summing one step later -- a whole step of
        // slack for the gathers, 126-157 registers -- measured the same: the launch is not bound by the
        // latency of its gathers.)
#define SCHWZ_STREAM_STEP(M, MN, P)                                                                  \
    {                                                                                                \
        const int b0 = P##b0;                                                                        \
        const double o0 = P##o0, o1 = P##o1;                                                         \
        lds_barrier(); /* every lane is done with the previous tile's entries */                     \
        rstart[tid] = b0;                                                                            \
        *reinterpret_cast<vd2 *>(&vals[4 * tid]) = P##v0;                                            \
        *reinterpret_cast<vd2 *>(&vals[4 * tid + 2]) = P##v1;                                        \
        *reinterpret_cast<vi4 *>(&cols[4 * tid]) = P##c0;                                            \
        *reinterpret_cast<vd2 *>(&vals[4 * (tid + kBlock)]) = P##v2;                                 \
        *reinterpret_cast<vd2 *>(&vals[4 * (tid + kBlock) + 2]) = P##v3;                             \
        *reinterpret_cast<vi4 *>(&cols[4 * (tid + kBlock)]) = P##c1;                                 \
        lds_barrier();                                                                               \
        const int b1 = (M).r0 + tid + 1 < (M).r1 ? rstart[tid + 1] : (M).e - ((M).s & ~3);          \
        double sum = 0.0;                                                                            \
        _Pragma("unroll") for (int j0 = 0; j0 < CAP; j0 += 8)                                        \
        {                                                                                            \
            double vv[8], xx[8];                                                                     \
            int cc[8];                                                                               \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                            \
            {                                                                                        \
                vv[j] = vals[b0 + j0 + j];                                                           \
                cc[j] = cols[b0 + j0 + j];                                                           \
            }                                                                                        \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                            \
                xx[j] = (ABL & 2) ? (double)cc[j] : a.x[(ABL & 16) ? (cc[j] & 0xFFFF) : cc[j]];      \
            if (j0 == 0) {                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                   \
                if (ABL & 32) a.y[(ABL & 8) ? (prow & 0x7FFF) : prow] = psum;                        \
                SCHWZ_STREAM_ISSUE(MN, P)                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                   \
            }                                                                                        \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                            \
            {                                                                                        \
                const double pv = vv[j] * xx[j];                                                     \
                sum += (b0 + j0 + j < b1) ? pv : 0.0;                                                \
            }                                                                                        \
        }                                                                                            \
        const int rowc = min((M).r0 + tid, (M).r1 - 1);                                              \
        const bool mine = (M).r0 + tid < (M).r1;                                                     \
        if (MODE == kSpmvPlain) {                                                                    \
            if (ABL & 1)                                                                             \
                acc0 += sum;                                                                         \
            else if (ABL & 32) {                                                                     \
                psum = a.alpha * sum;                                                                \
                prow = rowc;                                                                         \
            } else if ((ABL & 128) || NTY)                                                           \
                __builtin_nontemporal_store(a.alpha * sum, a.y + rowc);                              \
            else                                                                                     \
                a.y[(ABL & 8) ? (rowc & 0x7FFF) : rowc] = a.alpha * sum;                             \
        } else if (MODE == kSpmvDot) {                                                               \
            if (NTY)                                                                                 \
                __builtin_nontemporal_store(sum, a.y + rowc);                                        \
            else                                                                                     \
                a.y[rowc] = sum;                                                                     \
            const double t = o0 * sum;                                                               \
            acc0 += mine ? t : 0.0;                                                                  \
        } else if (MODE == kSpmvResidInit) {                                                         \
            const double r = o0 - sum;                                                               \
            const double z = a.dinv ? o1 * r : r;                                                    \
            a.y[rowc] = r;                                                                           \
            a.p[rowc] = z;                                                                           \
            const double t0 = r * z, t1 = r * r;                                                     \
            acc0 += mine ? t0 : 0.0;                                                                 \
            acc1 += mine ? t1 : 0.0;                                                                 \
        } else { /* kSpmvResidNorm */                                                                \
            const double r = o0 - sum;                                                               \
            const double t1 = r * r;                                                                 \
            acc1 += (mine && rowc < a.row_limit) ? t1 : 0.0;                                         \
        }                                                                                            \
    }
        vd2 Av0, Av1, Av2, Av3, Bv0, Bv1, Bv2, Bv3;
        vi4 Ac0, Ac1, Bc0, Bc1;
        int Ab0, Bb0;
        double Ao0, Ao1, Bo0, Bo1;
        Meta m0 = meta(0), m1 = meta(1);
        // delayed store (ABL bit 5): the first step stores 0.0 to the rows its own tile stores again one step
        // later (same lane, same address, program order)
        double psum = 0.0;
        int prow = min(m0.r0 + tid, m0.r1 - 1);
        SCHWZ_STREAM_ISSUE(m0, A)
        SCHWZ_STREAM_ISSUE(m1, B)
        Meta m2 = meta(2), m3 = meta(3);
        int k = 0;
        for (; k + 1 < ntw; k += 2) {
            SCHWZ_STREAM_STEP(m0, m2, A)
            SCHWZ_STREAM_STEP(m1, m3, B)
            m0 = m2;
            m1 = m3;
            m2 = meta(k + 4);
            m3 = meta(k + 5);
        }
        if (k < ntw) SCHWZ_STREAM_STEP(m0, m2, A)
        if ((ABL & 32) && !(ABL & 1)) a.y[(ABL & 8) ? (prow & 0x7FFF) : prow] = psum;
#undef SCHWZ_STREAM_STEP
#undef SCHWZ_STREAM_ISSUE
    }
    if (MODE == kSpmvPlain && (ABL & 1)) {
        if (acc0 == 123.456) a.y[tid] = acc0;  // keeps the sums alive
    }
    if (MODE != kSpmvPlain) {
        // (one partial sum per WAVE instead -- no barrier at the end of a short-lived workgroup's life, four times
        // the slots to fold -- measured the same step time, round 3)
        const double s0 = block_sum(acc0, red);
        const double s1 = MODE == kSpmvDot ? 0.0 : block_sum(acc1, red);
        if (tid == 0) {
            a.partials[blockIdx.x] = s0;
            a.partials[a.part_stride + blockIdx.x] = s1;
        }
    }
}

#ifdef SCHWZ_WITH_PROBES
// measurement builds of the plain mode (tools/stream_ablate.py, tools/r03_stream_probe.py): compiled into
// libschwz_hip_probes.so only
int launch_spmv_stream_ablate(const CsrView &A, const SpmvArgs &a, int abl, hipStream_t s)
{
    if (A.stream_cap == 0 || A.stream_cap > 8 || !A.tile_nz || A.tile_order) {
        set_error("schwz_csr_spmv: the stream ablation builds need a matrix the stream kernel takes (rows <= 8 entries)");
        return SCHWZ_ERR_INVALID;
    }
    const char *ge = std::getenv("SCHWZ_STREAM_GRID");
    int g = ge ? std::atoi(ge) : kMaxGrid;
    g = std::max(kXcds, std::min(g, kMaxGrid) / kXcds * kXcds);
    SpmvArgs b = a;
    const char *se = std::getenv("SCHWZ_STREAM_SEQ");
    b.seq = se ? std::max(0, std::atoi(se)) : 0;
    if (b.seq) {
        const int sh = A.xcd_shift;
        const int nslots = A.stream_order ? A.stream_nper : ((A.ntiles + (kXcds << sh) - 1) >> (sh + 3)) << sh;
        g = kXcds * ((nslots + b.seq - 1) / b.seq);
    }
    switch (abl) {
#define SCHWZ_ABL(W) \
    case W: hipLaunchKernelGGL((spmv_stream_kernel<kSpmvPlain, 8, W>), dim3(g), dim3(kBlock), 0, s, A, b); break;
        SCHWZ_ABL(0) SCHWZ_ABL(1) SCHWZ_ABL(2) SCHWZ_ABL(3) SCHWZ_ABL(4) SCHWZ_ABL(5) SCHWZ_ABL(6) SCHWZ_ABL(7)
        SCHWZ_ABL(8) SCHWZ_ABL(16) SCHWZ_ABL(24) SCHWZ_ABL(32) SCHWZ_ABL(40) SCHWZ_ABL(48) SCHWZ_ABL(56)
        SCHWZ_ABL(64) SCHWZ_ABL(128) SCHWZ_ABL(192)
    default: set_error("schwz_csr_spmv: no such stream ablation build"); return SCHWZ_ERR_INVALID;
#undef SCHWZ_ABL
    }
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

#endif  // SCHWZ_WITH_PROBES

// Folds the partial sums of a launch (`nin` per bank: one per workgroup or per wave, two banks `nin` apart) into
// the `nout` slots per bank the consumers read.  Slot i takes the partial sums i, i + nout, i + 2 nout, ...; eight
// adjacent lanes share a slot (lane l sums every eighth of them, four loads in flight, then a fixed xor tree), so the
// launch is a few dependent L2 round trips long whatever `nin` is.  Fixed order: reproducible bit for bit.
constexpr int kFoldSub = 8;

template <int MODE>
__global__ __launch_bounds__(kBlock) void spmv_stream_fold_kernel(const double *__restrict__ in, int nin,
                                                                  double *__restrict__ out, int nout,
                                                                  const int *stop_iter, int it)
{
    if (MODE == kSpmvDot || MODE == kSpmvResidInit) {
        if (stop_iter && it >= *stop_iter) return;
    }
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int i = min(t / kFoldSub, nout - 1), l = t % kFoldSub;  // surplus lanes repeat the last slot (whole groups of 8)
    constexpr bool two = MODE != kSpmvDot;                        // the p.q launch fills one bank
    const int64_t step = (int64_t)nout * kFoldSub;
    double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t j = i + (int64_t)nout * l; j < nin; j += 4 * step) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t ju = j + u * step;
            const bool in_range = ju < nin;
            const double a0 = in[in_range ? ju : j];
            s0[u] += in_range ? a0 : 0.0;
            if (two) {
                const double a1 = in[nin + (in_range ? ju : j)];
                s1[u] += in_range ? a1 : 0.0;
            }
        }
    }
    double v0 = (s0[0] + s0[1]) + (s0[2] + s0[3]);
    double v1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
#pragma unroll
    for (int m = 1; m < kFoldSub; m <<= 1) {
        v0 += __shfl_xor(v0, m, 64);
        if (two) v1 += __shfl_xor(v1, m, 64);
    }
    if (l == 0 && t / kFoldSub < nout) {
        out[i] = v0;
        out[nout + i] = two ? v1 : 0.0;
    }
}

// tiles per short-lived workgroup (SCHWZ_STREAM_SEQ; 0: the persistent form) and whether y leaves with
// non-temporal stores (SCHWZ_STREAM_NTY: 0 never, 1 y = A x only, 2 the q = A p of the CG iteration as well -- the
// default: 0.5-1 % of the plain-CSR step on two boxes, profiles/r03_plainloop_ab.txt)
static int stream_seq_max()
{
    static const int v = [] {
        const char *e = std::getenv("SCHWZ_STREAM_SEQ");
        return e ? std::max(0, std::min(std::atoi(e), 64)) : 3;
    }();
    return v;
}

static int stream_nty()
{
    static const int v = [] {
        const char *e = std::getenv("SCHWZ_STREAM_NTY");
        return e ? std::atoi(e) : 2;
    }();
    return v;
}

// Launches the stream kernel where it applies; *done = false leaves the launch to spmv_tiled2_kernel.
int launch_spmv_stream(const CsrView &A, int mode, const SpmvArgs &a, int grid, hipStream_t s, bool *done)
{
    *done = false;
    static const bool on = [] {
        const char *e = std::getenv("SCHWZ_SPMV_STREAM");
        return !(e && e[0] == '0');
    }();
    if (!on || A.stream_cap == 0 || !A.tile_nz || A.tile_order) return SCHWZ_OK;
    if (!(mode == kSpmvPlain || mode == kSpmvDot || mode == kSpmvResidInit || mode == kSpmvResidNorm)) return SCHWZ_OK;
    if (mode == kSpmvPlain && a.beta != 0.0) return SCHWZ_OK;
    SpmvArgs b = a;
    b.part_stride = grid;
    // short-lived workgroups (stream_seq_max() consecutive tiles each) once the matrix has four tiles per
    // workgroup of the persistent grid (2 M rows): below that launches, not memory, set the pace
    int launch_grid = grid;
    bool fold = false;
    const int seq = stream_seq_max();
    if (seq && A.ntiles >= 4 * kMaxGrid) {
        const int sh = A.xcd_shift;
        const int nslots = A.stream_order ? A.stream_nper : ((A.ntiles + (kXcds << sh) - 1) >> (sh + 3)) << sh;
        const int g = kXcds * ((nslots + seq - 1) / seq);
        if (mode == kSpmvPlain || (A.stream_part && g <= A.stream_part_cap)) {
            b.seq = seq;
            launch_grid = g;
            if (mode != kSpmvPlain) {
                fold = true;
                b.partials = A.stream_part;
                b.part_stride = g;
            }
        }
    }
    const bool nty = b.seq && ((mode == kSpmvPlain && stream_nty() >= 1) || (mode == kSpmvDot && stream_nty() >= 2));
#define SCHWZ_STREAM_CASE(M, C)                                                                                   \
    {                                                                                                             \
        if (nty)                                                                                                  \
            hipLaunchKernelGGL((spmv_stream_kernel<M, C, 0, true>), dim3(launch_grid), dim3(kBlock), 0, s, A, b); \
        else                                                                                                      \
            hipLaunchKernelGGL((spmv_stream_kernel<M, C, 0, false>), dim3(launch_grid), dim3(kBlock), 0, s, A, b); \
        if (fold)                                                                                                 \
            hipLaunchKernelGGL((spmv_stream_fold_kernel<M>), dim3((grid * kFoldSub + kBlock - 1) / kBlock), dim3(kBlock), 0, s, \
                               (const double *)b.partials, b.part_stride, a.partials, grid, a.stop_iter, a.it);  \
    }
#define SCHWZ_STREAM_MODE(M)                                  \
    if (A.stream_cap <= 8) SCHWZ_STREAM_CASE(M, 8)            \
    else if (A.stream_cap <= 16) SCHWZ_STREAM_CASE(M, 16)     \
    else SCHWZ_STREAM_CASE(M, 32)
    switch (mode) {
    case kSpmvPlain: SCHWZ_STREAM_MODE(kSpmvPlain) break;
    case kSpmvDot: SCHWZ_STREAM_MODE(kSpmvDot) break;
    case kSpmvResidInit: SCHWZ_STREAM_MODE(kSpmvResidInit) break;
    default: SCHWZ_STREAM_MODE(kSpmvResidNorm) break;
    }
#undef SCHWZ_STREAM_MODE
#undef SCHWZ_STREAM_CASE
    SCHWZ_HIP_TRY(hipGetLastError());
    *done = true;
    return SCHWZ_OK;
}

}  // namespace schwz
