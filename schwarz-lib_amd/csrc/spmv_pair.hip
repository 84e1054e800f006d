// Row-PAIR pattern coding of the CSR SpMV: the third, and fastest, lossless coding of this library.
//
// Why.  tools/probes/gather_probe.hip times the bare gather + store structure of a 7-point SpMV at
// 256^3 with no matrix data at all: one row per lane with 8-byte gathers 0.088-0.096 ms, two adjacent
// rows per lane with 16-byte gathers 0.053-0.055 ms.  The row-pattern kernel (spmv_dict.hip, one row
// per lane) sits at 0.115 ms, i.e. on that floor: it is bound by the number of vector-memory
// instructions, not by bytes.  Here a lane owns rows (r, r+1): for an offset d present in either row
// ONE 16-byte load fetches x[r+d] (row r's operand) and x[r+1+d] (row r+1's), halving the gather
// instructions; the matrix costs one byte per PAIR.
//
// Coding.  For a pair, the entries of both rows are merged by offset d = col - row (both rows must
// have strictly ascending columns, so each row still sees its own entries in CSR order); an entry is
// (d, value for row r, value for row r+1, presence bits).  A tile's distinct merged sequences form
// its table (<= 64 patterns, <= 512 entries, staged in LDS); tables are de-duplicated over the
// matrix.  Products are rounded individually and added in the row's entry order: the result is
// bit-identical to the plain CSR kernels.  Tiles that do not qualify run through the plain row code
// of the same launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "device_utils.hpp"
#include "schwz_hip.h"
#include "schwz_internal.hpp"

namespace schwz {

constexpr int kPairPats = 64;      // patterns per table
constexpr int kPairEntries = 512;  // staged entries per table (npat * stride)
constexpr int kPairChunk = 8;      // gathers issued back to back per lane
// dynamic shared memory of every launch: values (16 B), offsets (4 B) per staged entry, group masks
constexpr int kPairTableLds = kPairEntries * 16 + kPairEntries * 4 + (kPairEntries / 4) * 4;

__host__ __device__ inline int pair_stride(int lmax, int ch = kPairChunk) { return (lmax + ch - 1) / ch * ch; }

struct __attribute__((aligned(16))) PairVal {
    double a, b;
};
struct __attribute__((aligned(8))) PairMeta {
    int off;    // col - row
    int flags;  // bit 0: row r has the entry, bit 1: row r+1 has it
};

typedef double pvd2 __attribute__((ext_vector_type(2)));

constexpr int kPairRows = 2 * kBlock;  // rows per chunk: one pair per lane

// chunk dealt to XCD `xcd` as its j-th one (block-cyclic in runs of 1 << sh), -1 past the end
__device__ __forceinline__ int xcd_chunk(int nchunks, int sh, int xcd, int j)
{
    const int c = ((j >> sh) << (sh + 3)) + (xcd << sh) + (j & ((1 << sh) - 1));
    return c < nchunks ? c : -1;
}

// SINGLE: the whole matrix shares one table (every chunk is coded with table 0): it is staged once
// and no per-chunk lookup precedes the pattern ids.
template <int MODE, bool WIDE, bool SINGLE>
__global__ __launch_bounds__(kBlock) void spmv_pair_kernel(CsrView A, SpmvArgs a)
{
#pragma clang fp contract(off)
    // the staged table lives in dynamic shared memory (kPairTableLds bytes, every launch passes them): the
    // z-sweep walk lays its ring of plane windows over it once the canonical slots have been derived
    extern __shared__ __attribute__((aligned(16))) char pair_lds[];
    PairVal *const pv = reinterpret_cast<PairVal *>(pair_lds);
    int *const poff = reinterpret_cast<int *>(pair_lds + kPairEntries * sizeof(PairVal));  // col - row of the entry
    int *const pmask = poff + kPairEntries;  // per group of CH entries: bit k = row r has entry k, bit 8+k = row r+1
    // gathers issued back to back per lane; the upper-triangle tables of kSpmvDotSym are about half as long
    constexpr bool kDirVec = MODE == kSpmvDirDotSymVec;
    constexpr bool kDir = MODE == kSpmvDirDotSym || kDirVec;
    constexpr bool kSym = MODE == kSpmvDotSym || kDir;
    constexpr int CH = kSym ? kPairChunk / 2 : kPairChunk;
    constexpr bool kDotOnly = MODE == kSpmvDotOnly || kSym;
    __shared__ int plen[kPairPats];            // length | (a gathered entry has col - row == 0) << 16
    __shared__ int pmin[kPairPats], pmax[kPairPats];  // smallest / largest col - row the pattern gathers at
    // canonical stencil layout (CsrView::pair_canon): every pattern expanded to the 7 fixed slots, with
    // presence bits; -1: the pattern has an entry outside the layout
    constexpr bool kCanon = SINGLE && !kSym;
    __shared__ PairVal cpv[kCanon ? kPairPats * 8 : 1];
    __shared__ int cmask[kCanon ? kPairPats : 1];
    const bool canon_on = kCanon && A.pair_canon[7] != 0;
    if (MODE == kSpmvDot || MODE == kSpmvResidInit || kDotOnly) {
        if (a.stop_iter && a.it >= *a.stop_iter) return;
    }
    __shared__ double red[4];
    double cg_alpha = 0.0, cg_beta = 0.0, cg_rho_new = 0.0, cg_rr = 0.0;
    if (kDir) {
        if (a.it >= a.cg_state->stop_iter) return;
        cg_rho_new = fold_partials(a.pq_partials, a.pq_nparts, red);
        cg_rr = fold_partials(a.pq_partials + a.pq_nparts, a.pq_nparts, red);
        cg_beta = cg_rho_new / a.cg_state->rho[a.it & 1];
    }
    if (MODE == kSpmvCgUpdate) {
        if (a.it >= a.cg_state->stop_iter) return;
        cg_alpha = a.cg_state->rho[a.it & 1] / fold_partials(a.pq_partials, a.pq_nparts, red);
        if (a.alpha_out && blockIdx.x == 0 && threadIdx.x == 0) *a.alpha_out = cg_alpha;
    }
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x % kXcds;
    const int slot = blockIdx.x / kXcds;
    const int per_xcd = gridDim.x / kXcds;
    const int nrows = (int)A.nrows;
    const int nchunks = (nrows + kPairRows - 1) / kPairRows;
    const int sh = A.pair_shift;
    const int slots = ((nchunks + (kXcds << sh) - 1) >> (sh + 3)) << sh;  // sequence slots per XCD
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    const bool dual = (MODE == kSpmvResidDual) && a.x2 != nullptr;
    constexpr bool kWantOwn = MODE == kSpmvDot || kDotOnly || MODE == kSpmvCgUpdate;  // x[row] itself
    int cached = -1, ls = 0, tl = 0;  // staged table, its row stride, its longest pattern

    auto stage_table = [&](int tb) {  // workgroup-uniform
        const int td = kSym ? A.pair_sym_base + tb : tb;
        const int eoff = A.ptbl_desc[5 * td], loff = A.ptbl_desc[5 * td + 1];
        const int npat = A.ptbl_desc[5 * td + 2], lmax = A.ptbl_desc[5 * td + 3];
        ls = pair_stride(lmax, CH);
        tl = lmax;
        lds_barrier();  // everyone is done with the previous table
        for (int i = tid; i < npat * ls; i += kBlock) {
            const int pt = i / ls, k = i - pt * ls;
            PairVal v = {0.0, 0.0};
            int off = 0;
            if (k < lmax) {
                const int e = eoff + pt * lmax + k;
                v.a = A.ptbl_val[2 * e];
                v.b = A.ptbl_val[2 * e + 1];
                off = A.ptbl_meta[2 * e];
            }
            pv[i] = v;
            poff[i] = off;
        }
        for (int i = tid; i < npat * ls / CH; i += kBlock) {
            const int pt = i / (ls / CH), k0 = (i - pt * (ls / CH)) * CH;
            int m = 0;
            for (int k = 0; k < CH && k0 + k < lmax; ++k) {
                const int fl = A.ptbl_meta[2 * (eoff + pt * lmax + k0 + k) + 1];
                m |= (fl & 1) << k;
                m |= ((fl >> 1) & 1) << (kPairChunk + k);
            }
            pmask[i] = m;
        }
        lds_barrier();
        if (tid < npat) {
            // does a gather of this pattern (padding included) fetch x[r], x[r + 1] themselves?
            const int len = A.ptbl_len[loff + tid];
            int zero = 0, mn = 0, mx = 0;  // the padding entries gather at offset 0
            // entries at or past the longest pattern of the table are padding for every lane: no
            // launch gathers them
            for (int k = 0; k < min(pair_stride(len, CH), lmax); ++k) {
                const int off = poff[tid * ls + k];
                zero |= off == 0;
                mn = min(mn, off);
                mx = max(mx, off);
            }
            plen[tid] = len | (zero << 16);
            pmin[tid] = mn;
            pmax[tid] = mx;
            if (kCanon && canon_on) {
                // slot of each entry: the first layout slot with its offset (duplicated filler slots
                // therefore never receive one)
                int cm = 0;
                for (int k = 0; k < 8; ++k) cpv[tid * 8 + k] = PairVal{0.0, 0.0};
                const int gm = len ? pmask[(tid * ls) / CH] : 0;
                for (int k = 0; k < len && cm >= 0; ++k) {
                    const int off = poff[tid * ls + k];
                    int slot = -1;
                    for (int q = 6; q >= 0; --q)
                        if (A.pair_canon[q] == off) slot = q;
                    if (slot < 0 || len > CH) {
                        cm = -1;
                        break;
                    }
                    cpv[tid * 8 + slot] = pv[tid * ls + k];
                    cm |= ((gm >> k) & 1) << slot;
                    cm |= ((gm >> (kPairChunk + k)) & 1) << (kPairChunk + slot);
                }
                cmask[tid] = cm;
            }
        }
        lds_barrier();
        cached = tb;
    };
    // x[c], x[c + 1] -- the operands of one merged entry for rows r and r + 1 -- by ONE 16-byte load
    auto ld16 = [&](const double *xv, int c) -> pvd2 {
        pvd2 t;
        if (WIDE) {
            __builtin_memcpy(&t, xv + c, 16);
        } else {
            const uint32_t o = (uint32_t)c * 8u;  // scalar base + 32-bit lane offset
            __builtin_memcpy(&t, reinterpret_cast<const char *>(xv) + o, 16);
        }
        return t;
    };
    // Both rows' dot products with xv.  `safe`: some lane of the wave has a pattern whose 16-byte
    // gathers could leave the vector (first / last column): those waves load each operand that
    // exists on its own.
    auto accumulate = [&](const double *xv, int ra, int base, int len, bool safe, double &s0, double &s1,
                          pvd2 &own, bool want_own) {
        for (int j = 0; j < len; j += CH) {
            const int mask = pmask[(base + j) / CH];
            const bool skip_last = CH == kPairChunk && j + CH - 1 >= tl;  // workgroup-uniform; groups of 8 only
            pvd2 t[CH];
            if (!safe) {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    // the last slot of a group is padding for every lane when the table's longest
                    // pattern ends one short of it (7 entries in groups of 8: the 7-point stencil):
                    // one straight-line copy of the loads without it (a per-slot test would break
                    // the back-to-back issue of the gathers: measured 0.100 -> 0.141 ms)
                    if (k == CH - 1 && skip_last) continue;
                    const int off = poff[base + j + k];
                    t[k] = ld16(xv, ra + off);
                    if (want_own && off == 0) own = t[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int c = ra + poff[base + j + k];
                    t[k].x = (mask >> k) & 1 ? xv[c] : 0.0;
                    t[k].y = (mask >> (kPairChunk + k)) & 1 ? xv[c + 1] : 0.0;
                }
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const PairVal v = pv[base + j + k];
                if ((mask >> k) & 1) s0 += v.a * t[k].x;
                if ((mask >> (kPairChunk + k)) & 1) s1 += v.b * t[k].y;
            }
        }
    };
    // The same sums for a wave whose patterns all live in the canonical stencil layout: six gathers at
    // workgroup-uniform offsets; the operands of the offset-0 slot, x[r] and x[r + 1], are the second
    // half of the -1 gather and the first half of the +1 gather.  Products in slot order = entry order,
    // absent entries masked: the same bits as the generic walk.
    auto accumulate_canon = [&](const double *xv, int ra, int pid, double &s0, double &s1, pvd2 &own) {
        const int mask = cmask[pid];
        pvd2 t[7];
        t[0] = ld16(xv, ra + A.pair_canon[0]);
        t[1] = ld16(xv, ra + A.pair_canon[1]);
        t[2] = ld16(xv, ra - 1);
        t[4] = ld16(xv, ra + 1);
        t[5] = ld16(xv, ra + A.pair_canon[5]);
        t[6] = ld16(xv, ra + A.pair_canon[6]);
        t[3].x = t[2].y;
        t[3].y = t[4].x;
        own = t[3];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const PairVal v = cpv[pid * 8 + k];
            if ((mask >> k) & 1) s0 += v.a * t[k].x;
            if ((mask >> (kPairChunk + k)) & 1) s1 += v.b * t[k].y;
        }
    };
    // kSpmvDirDotSym: the new search direction p' = z + beta p at column c, z = D^-1 r -- one
    // expression for the stored copy and for every neighbour's recomputed one
    auto dir2 = [&](pvd2 rv, pvd2 pv, pvd2 dv) -> pvd2 {
        const pvd2 z = a.diag_mode ? dv * rv : rv;
        pvd2 o;
        o.x = __builtin_fma(cg_beta, pv.x, z.x);
        o.y = __builtin_fma(cg_beta, pv.y, z.y);
        return o;
    };
    auto dir1 = [&](int c) -> double {
        const double rv = a.cg_r[c];
        const double z = kDirVec ? a.dinv[c] * rv : (a.diag_mode ? a.diag_uniform * rv : rv);
        return __builtin_fma(cg_beta, a.x[c], z);
    };
    auto accumulate_dir = [&](int ra, int base, int len, bool safe, double &s0, double &s1, pvd2 &own) {
        const pvd2 du = {a.diag_uniform, a.diag_uniform};
        for (int j = 0; j < len; j += CH) {
            const int mask = pmask[(base + j) / CH];
            pvd2 t[CH];
            if (!safe) {
                pvd2 tr[CH], tp[CH], td[kDirVec ? CH : 1];
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int off = poff[base + j + k];
                    tr[k] = ld16(a.cg_r, ra + off);
                    tp[k] = ld16(a.x, ra + off);
                    if (kDirVec) td[k] = ld16(a.dinv, ra + off);
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    t[k] = dir2(tr[k], tp[k], kDirVec ? td[kDirVec ? k : 0] : du);
                    if (poff[base + j + k] == 0) own = t[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int c = ra + poff[base + j + k];
                    t[k].x = (mask >> k) & 1 ? dir1(c) : 0.0;
                    t[k].y = (mask >> (kPairChunk + k)) & 1 ? dir1(c + 1) : 0.0;
                }
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const PairVal v = pv[base + j + k];
                if ((mask >> k) & 1) s0 += v.a * t[k].x;
                if ((mask >> (kPairChunk + k)) & 1) s1 += v.b * t[k].y;
            }
        }
    };
    // one row straight from val / col (chunks that are not pair coded)
    auto plain_row = [&](int row, bool dual_t, double &sum, double &sum2) {
        for (int j = A.rp[row]; j < A.rp[row + 1]; ++j) {
            double vv = A.val[j];
            const int cc = A.col[j];
            if (kSym) {  // upper triangle, off-diagonal entries twice
                if (cc < row) continue;
                if (cc > row) vv *= 2.0;
            }
            if (kDir) {
                sum += vv * dir1(cc);
                continue;
            }
            sum += vv * a.x[cc];
            if (dual_t) sum2 += vv * a.x2[cc];
        }
    };
    auto finish = [&](int rw, double sum, double sum2, bool dual_t, double ox, double ob, double od, double &yv,
                      double &pz) {
        yv = sum;
        pz = 0.0;
        if (MODE == kSpmvPlain) {
            yv = (a.beta == 0.0) ? a.alpha * sum : a.alpha * sum + a.beta * ob;
        } else if (MODE == kSpmvDot || kDotOnly) {
            acc0 += ox * sum;
        } else if (MODE == kSpmvCgUpdate) {
            // ob: r_i, od: 1/diag_i, ox: p_i, sum: q_i = (A p)_i; yv <- new r_i; pz <- alpha p_i (x increment)
            const double r = ob - cg_alpha * sum;
            const double z = a.diag_mode ? od * r : r;
            yv = r;
            pz = cg_alpha * ox;
            acc0 += r * z;
            acc1 += r * r;
        } else if (MODE == kSpmvResidInit || MODE == kSpmvResidDual) {
            const double r = ob - sum;
            const double z = (a.dinv || a.diag_mode == 3) ? od * r : r;
            yv = r;
            pz = z;
            acc0 += r * z;
            acc1 += r * r;
            if (MODE == kSpmvResidDual && rw < a.row_limit) {
                const double r2 = dual_t ? ob - sum2 : r;
                acc2 += r2 * r2;
            }
        } else {  // kSpmvResidNorm
            if (rw < a.row_limit) {
                const double r = ob - sum;
                acc1 += r * r;
            }
        }
    };

    // run-length record of a chunk's pattern ids (first word 0xffff: none, the ids are bytes per pair).
    // chunk is workgroup-uniform and the table is never written by a kernel: read through the constant
    // address space, i.e. with a scalar load that costs no vector-memory slot
    struct RleRec {
        uint4 lo, hi;  // hi: runs 8-15 of a 16-run record
    };
    const bool rle16 = A.pair_rle_runs == 16;
    auto load_rle = [&](int chunk) -> RleRec {
        RleRec rle = {{0xffffu, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
        if (A.pair_rle) {
            typedef const unsigned __attribute__((address_space(4))) *const_words;
            const const_words q = (const_words)(uintptr_t)(A.pair_rle + (rle16 ? 2 * chunk : chunk));
            rle.lo.x = q[0];
            rle.lo.y = q[1];
            rle.lo.z = q[2];
            rle.lo.w = q[3];
            if (rle16) {
                rle.hi.x = q[4];
                rle.hi.y = q[5];
                rle.hi.z = q[6];
                rle.hi.w = q[7];
            }
        }
        return rle;
    };
    // pattern id of the pair starting at row ra: the id of the last run that starts at or before this
    // lane's pair (runs ascending, unused slots repeat the last run)
    auto pid_of = [&](const RleRec rle, int ra) -> int {
        int pid;
        if ((rle.lo.x & 0xffffu) != 0xffffu) {
            const unsigned w[8] = {rle.lo.x, rle.lo.y, rle.lo.z, rle.lo.w, rle.hi.x, rle.hi.y, rle.hi.z, rle.hi.w};
            pid = (int)((w[0] >> 8) & 0xffu);
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const unsigned e = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                if ((unsigned)tid >= (e & 0xffu)) pid = (int)(e >> 8);
            }
            if (rle16) {  // workgroup-uniform
#pragma unroll
                for (int k = 8; k < 16; ++k) {
                    const unsigned e = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                    if ((unsigned)tid >= (e & 0xffu)) pid = (int)(e >> 8);
                }
            }
        } else {
            pid = A.pair_id[ra >> 1];
        }
        return pid;
    };
    auto pair_pid = [&](int chunk, int ra) -> int { return pid_of(load_rle(chunk), ra); };
    // Fixed chunks of 512 consecutive rows, one pair per lane: nothing but the chunk's table id has
    // to be looked up before the pattern ids and the gathers can be requested.
    if (SINGLE) stage_table(0);
    // listed walk: the chunks no z-sweep segment covers (CsrView::sweep_gen), the companion launch of
    // spmv_pair_sweep_kernel below; its partial sums go behind that kernel's (a.part_offset)
    // (kSpmvResidNorm, a.sweep == 2: the chunks of the planes where the fused check residual needs its own
    // product, CsrView::dual_chunks -- the second launch of a dual start in the walk)
    const bool listed = SINGLE && (MODE == kSpmvCgUpdate || MODE == kSpmvDirDotSym || MODE == kSpmvResidNorm) && a.sweep != 0;
    const bool dual_list = MODE == kSpmvResidNorm && a.sweep == 2;
    const int gen_first = listed ? (int)blockIdx.x : slot;
    const int gen_count = listed ? (dual_list ? A.dual_nchunks : A.sweep_ngen) : slots;
    const int gen_stride = listed ? (int)gridDim.x : per_xcd;
    for (int j = gen_first; j < gen_count; j += gen_stride) {
        const int chunk = listed ? (dual_list ? A.dual_chunks[j] : A.sweep_gen[j]) : xcd_chunk(nchunks, sh, xcd, j);
        if (chunk < 0) continue;
        const int tb = SINGLE ? 0 : A.chunk_ptable[chunk];
        const bool dual_t = dual && (!A.chunk_dual || A.chunk_dual[chunk]);
        if (!SINGLE && tb >= 0 && tb != cached) stage_table(tb);
        const int ra = chunk * kPairRows + 2 * tid;
        if (ra >= nrows) continue;
        const bool has_b = ra + 1 < nrows;
        // own operands of the fused epilogues, requested ahead of the gathers
        double ob0 = 0.0, ob1 = 0.0, od0 = 1.0, od1 = 1.0;
        if (MODE == kSpmvResidNorm) {
            if (ra < a.row_limit) ob0 = a.b[ra];
            if (has_b && ra + 1 < a.row_limit) ob1 = a.b[ra + 1];
        }
        if (MODE == kSpmvResidInit || MODE == kSpmvResidDual) {
            // b and 1/diag only stream through: one 16-byte non-temporal load each (the caller's b
            // need not be 16-byte aligned), so that they do not push the gathered x lines out of L2
            typedef double pvd2u __attribute__((ext_vector_type(2), aligned(8)));
            if (has_b) {
                const pvd2u bb = __builtin_nontemporal_load(reinterpret_cast<const pvd2u *>(a.b + ra));
                ob0 = bb.x;
                ob1 = bb.y;
            } else {
                ob0 = a.b[ra];
            }
            if (a.diag_mode == 3) {
                od0 = od1 = a.diag_uniform;
            } else if (a.dinv) {
                if (has_b) {
                    const pvd2u dd = __builtin_nontemporal_load(reinterpret_cast<const pvd2u *>(a.dinv + ra));
                    od0 = dd.x;
                    od1 = dd.y;
                } else {
                    od0 = a.dinv[ra];
                }
            }
        }
        if (MODE == kSpmvPlain && a.beta != 0.0) {
            ob0 = a.y[ra];
            if (has_b) ob1 = a.y[ra + 1];
        }
        pvd2 cgx = {0.0, 0.0};
        if (MODE == kSpmvCgUpdate) {  // r and x of the pair, 1/diag
            if (has_b) {
                // x and r only stream through (read once, written once per launch): non-temporal, so
                // they do not push the gathered p lines out of the XCD's L2 (in-box A/B of the CG
                // iteration: 0.2525 -> 0.2398 ms at 256^3, 0.2632 -> 0.2475 ms on the 512 x 512 x 64 slab,
                // whose +-plane neighbours are 2 MB apart)
                const pvd2 rr = __builtin_nontemporal_load(reinterpret_cast<const pvd2 *>(a.cg_r + ra));
                if (a.cg_x) cgx = __builtin_nontemporal_load(reinterpret_cast<const pvd2 *>(a.cg_x + ra));
                ob0 = rr.x;
                ob1 = rr.y;
            } else {
                ob0 = a.cg_r[ra];
                if (a.cg_x) cgx.x = a.cg_x[ra];
            }
            if (a.diag_mode == 1) {
                if (has_b) {
                    const pvd2 dd = __builtin_nontemporal_load(reinterpret_cast<const pvd2 *>(a.dinv + ra));
                    od0 = dd.x;
                    od1 = dd.y;
                } else {
                    od0 = a.dinv[ra];
                }
            } else if (a.diag_mode == 3) {
                od0 = od1 = a.diag_uniform;
            }
        }
        double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
        pvd2 own = {0.0, 0.0};
        bool own_from_gathers = false;
        if (tb >= 0) {
            const int pid = pair_pid(chunk, ra);
            const int lenz = plen[pid];
            const int len = lenz & 0xffff;
            const int base = pid * ls;
            // a 16-byte gather at offset d reads x[ra + d] and x[ra + d + 1] whether or not both rows have
            // the entry: the lane is an edge lane when that could leave [0, ncols) for ITS pattern (a
            // subdomain's overlap rows sit at the end of the local numbering, so a table-wide bound
            // would send every wave of an interior subdomain down the safe path)
            const bool edge = ra + pmin[pid] < 0 || ra + 1 + pmax[pid] >= (int)A.ncols;
            const bool safe = __builtin_amdgcn_ballot_w64(edge) != 0;
            // the fused dot needs x[ra], x[ra + 1]: the padding entries (col - row = 0) gather exactly
            // that pair, as does a diagonal entry; waves on the safe path load it themselves
            own_from_gathers = kWantOwn && !safe && (lenz >> 16) != 0;
            pvd2 unused = {0.0, 0.0};
            bool canon = false;
            if (kCanon && canon_on) {
                const bool fits = cmask[pid] >= 0 && ra + A.pair_canon[0] >= 0 &&
                                  ra + 1 + A.pair_canon[6] < (int)A.ncols;
                canon = __builtin_amdgcn_ballot_w64(!fits) == 0;
            }
            if (kCanon && canon) {
                accumulate_canon(a.x, ra, pid, s0, s1, own);
                own_from_gathers = kWantOwn;
                if (dual_t) accumulate_canon(a.x2, ra, pid, t0, t1, unused);
            } else if (kDir) {
                accumulate_dir(ra, base, len, safe, s0, s1, own);
            } else {
                accumulate(a.x, ra, base, len, safe, s0, s1, own, own_from_gathers);
                if (dual_t) accumulate(a.x2, ra, base, len, safe, t0, t1, unused, false);
            }
        } else {
            plain_row(ra, dual_t, s0, t0);
            if (has_b) plain_row(ra + 1, dual_t, s1, t1);
        }
        if (kWantOwn && !own_from_gathers) {
            own.x = kDir ? dir1(ra) : a.x[ra];
            if (has_b) own.y = kDir ? dir1(ra + 1) : a.x[ra + 1];
        }
        double y0, y1 = 0.0, p0, p1 = 0.0;
        finish(ra, s0, t0, dual_t, own.x, ob0, od0, y0, p0);
        if (has_b) finish(ra + 1, s1, t1, dual_t, own.y, ob1, od1, y1, p1);
        if (MODE == kSpmvCgUpdate) {
            if (has_b) {
                const pvd2 rr = {y0, y1};
                const pvd2 xx = {cgx.x + p0, cgx.y + p1};
                __builtin_nontemporal_store(rr, reinterpret_cast<pvd2 *>(a.cg_r + ra));
                if (a.cg_x) __builtin_nontemporal_store(xx, reinterpret_cast<pvd2 *>(a.cg_x + ra));
            } else {
                a.cg_r[ra] = y0;
                if (a.cg_x) a.cg_x[ra] = cgx.x + p0;
            }
        } else if (kDir) {  // the pair's own new direction
            if (has_b)
                __builtin_memcpy(a.y + ra, &own, 16);
            else
                a.y[ra] = own.x;
        } else if (MODE != kSpmvResidNorm && !kDotOnly) {
            if (has_b) {
                const pvd2 yy = {y0, y1};
                if (MODE == kSpmvResidInit || MODE == kSpmvResidDual)  // r of the CG start: written once, read much later
                    __builtin_nontemporal_store(yy, reinterpret_cast<pvd2 *>(a.y + ra));
                else
                    __builtin_memcpy(a.y + ra, &yy, 16);
            } else {
                a.y[ra] = y0;
            }
        }
        if (MODE == kSpmvResidInit || MODE == kSpmvResidDual) {
            if (has_b) {
                const pvd2 pp = {p0, p1};
                __builtin_nontemporal_store(pp, reinterpret_cast<pvd2 *>(a.p + ra));
            } else {
                a.p[ra] = p0;
            }
        }
    }
    if (MODE != kSpmvPlain) {
        const double s0 = block_sum(acc0, red);
        const double s1 = block_sum(acc1, red);
        if (tid == 0) {
            const int stride = a.part_stride ? a.part_stride : (int)gridDim.x;
            a.partials[a.part_offset + blockIdx.x] = s0;
            a.partials[stride + a.part_offset + blockIdx.x] = s1;
        }
        if (MODE == kSpmvResidDual) {
            const double s2v = block_sum(acc2, red);
            if (tid == 0) a.partials[2 * gridDim.x + blockIdx.x] = s2v;
        }
    }
    if (kDir && blockIdx.x == 0 && tid == 0 && !a.sweep) {  // (the z-sweep kernel does it for its companion launch)
        // what cg_direction_kernel's workgroup 0 does: the rho slot written is the one no launch of
        // this iteration reads; stop_iter = 0 makes every later launch leave at once
        CgState *st = const_cast<CgState *>(a.cg_state);
        st->rho[(a.it + 1) & 1] = cg_rho_new;
        st->rr = cg_rr;
        st->iters = st->iters + 1;
        if (sqrt(cg_rr) <= a.cg_rtol * st->r0) st->stop_iter = 0;
    }
}

// SCHWZ_WALK_ZEROCOEF (build-time switch, round 3): the walks stage the slot tables into LDS with a coefficient of
// +0.0 wherever a pattern has no entry and sum EVERY slot's product, instead of selecting +0.0 for the absent ones
// by a mask bit (two bit tests and two 64-bit selects per slot and row pair: a third of the vector instructions of a
// step).  The same bits: a sum that began at +0.0 is never -0.0, so adding the +-0.0 that (+0.0 x a finite operand)
// yields leaves it unchanged, exactly like adding the selected +0.0.  The operands of absent slots are loaded vector
// data (clamped addresses, windows of real planes), finite whenever the vector is; a vector holding Inf / NaN has a
// NaN p.(A p) either way.  0 restores the masked sums.
#ifndef SCHWZ_WALK_ZEROCOEF
#define SCHWZ_WALK_ZEROCOEF 1
#endif

// SCHWZ_WALK_HINTS (build-time switch): 1 = the update walk's r is a non-temporal load and store and p' leaves the
// fused launch with non-temporal stores; 0 = plain loads and stores.  On vectors of 135 MB (256^3, 512 x 512 x 64) the
// hints win: in-box a plain r load in the update launch costs the FOLLOWING fused launch 4 us of 74, a plain r store
// the update launch 3 us of 69.  Where r, p and p' fit the 256 MiB Infinity Cache together (192^3: 3 x 57 MB) they
// LOSE 5 % of the step -- they give up the on-die copy the next launch would have read (profiles/r03_cache_hints.txt).
// A per-launch choice was built and dropped: as a workgroup-uniform branch around the three accesses it cost 2.5 % of
// the 256^3 step (the loop body is no longer one block), as a template parameter it doubles ~100 instantiations; no
// configuration of BASELINE.json has vectors that small.
#ifndef SCHWZ_WALK_HINTS
#define SCHWZ_WALK_HINTS 1
#endif
#ifndef SCHWZ_INIT_PLAIN_STORE
#define SCHWZ_INIT_PLAIN_STORE 1  // start walk: r0 leaves with a plain store -- its next two readers (first direction, first update) follow at once and find part of it on die: -0.5 ... -0.75 % per step at 256^3 / 512 x 512 x 64, neutral on 1 GB vectors (profiles/r03_cache_hints.txt)
#endif

// ---------------------------------------------------------------------------------------------------
// z-sweep ("brick") walk of the q-free CG update launch for matrices in the canonical 3-D stencil layout
// {-PL, -NX, -1, 0, +1, +NX, +PL} (CsrView::sweep_*).  tools/probes/brick_probe.hip priced the memory
// structure first: with six 16-byte gathers per row pair the launch is bound by vector-memory
// instructions and, on wide planes, re-fetches the +-PL lines the 4 MiB L2s cannot hold; here a
// workgroup takes a band of T rows through the planes z0 <= z < z1, fetches each plane's window
// [band - NX, band + T + NX) of p ONCE with coalesced 16-byte loads two planes ahead of its use, and
// reads all seven operands of a row pair from an LDS ring of four windows.  r (and 1/diag) of the
// lane's pairs and the chunk's pattern-id record are requested one plane ahead.  The loop body is
// straight-line code (addresses are clamped instead of guarded, every count is a template
// parameter), so the compiler can give each wait its own vmcnt instead of draining the queue.
// Products, order and presence masks are those of accumulate_canon: the same bits per row.
// NL: 16-byte pieces of a window per lane; NH: chunks of 512 rows per band; DIAGVEC: Jacobi diagonal
// as a full vector.  x is not touched (deferred x update only).
// INIT: the same walk as the CG start launch (kSpmvResidInit without its p store): the ring holds the
// windows of the start vector y, r = b - A y is stored to a.y, partial r.z and r.r -- p = D^-1 r is left
// to the first fused direction launch (spmv_pair_dirdot_sweep_kernel<.., FIRST>), which reads r anyway.
// ---------------------------------------------------------------------------------------------------
// DUAL (with INIT): third partial bank = sum of r_i^2 over the chain positions NOT flagged in chain_dual -- the
// part of the fused check residual ||b - A x2||^2 that coincides with the start residual; the flagged planes
// are added by a listed kSpmvResidNorm launch on x2 (launch_spmv_pair).
// RUNS: runs per chunk record of the pattern ids (8: 16 bytes, 16: 32 bytes; CsrView::pair_rle_runs); 0: planes that
// are not whole 512-row chunks (PL % 512 != 0, round 3): the bands of a plane no longer coincide with the chunks
// the records describe, so a lane reads its pair's pattern id as a byte (one coalesced load per 512 rows, requested
// a plane ahead like the record), and the last band of a plane is partial: its lanes beyond the plane's end repeat
// the band's last pair -- same loads, same value stored to the same address -- and add nothing to the sums.
// (The three-positions-ahead halo schedule that pays in the fused direction launch below was measured here too,
// with a third halo slot: no gain on 256- and 512-wide planes, where the p halo already hits L2 -- r is a
// non-temporal stream in this launch and does not evict it --, and a loss wherever the larger ring costs a
// workgroup per CU: profiles/r03_h3_rle16_ab.txt.  Not kept.)
template <int NHL, int NH, bool DIAGVEC, bool INIT = false, bool DUAL = false, int RUNS = 8>
__global__ __launch_bounds__(kBlock) void spmv_pair_sweep_kernel(CsrView A, SpmvArgs a)
{
#pragma clang fp contract(off)
    // dynamic LDS: own parts of four chain positions [4][T], the +-NX halos of two [2][2 NX] (only the plane
    // being computed needs its halo, which is why it is requested one position ahead, not two: its lines are
    // the own lines of the neighbouring bands, i.e. L2 hits), the slot tables of the matrix
    extern __shared__ __attribute__((aligned(16))) char sweep_lds[];
    constexpr int T = NH * kPairRows;
    const int NX = A.sweep_nx;
    double *const own_ring = reinterpret_cast<double *>(sweep_lds);
    double *const halo_ring = own_ring + 4 * T;
    PairVal *const cpv = reinterpret_cast<PairVal *>(halo_ring + 4 * NX);
    int *const cmask = reinterpret_cast<int *>(cpv + A.canon_npat * 9);
    __shared__ double red[4];
    double cg_alpha = 1.0;
    const int tid = threadIdx.x;
    if (!INIT) {
        if (a.it >= a.cg_state->stop_iter) return;
    }
    // INIT reads the right-hand side where the update launch reads r (the caller's b need only be 8-byte
    // aligned) and stores r to a.y
    typedef double pvd2u __attribute__((ext_vector_type(2), aligned(8)));
    const double *const r_in = INIT ? a.b : a.cg_r;
    double *const r_out = INIT ? a.y : (a.cg_r_out ? a.cg_r_out : a.cg_r);
    const pvd2 ring_scale = {a.ring_scale, a.ring_scale};  // 1.0 except in the first update of a solve with a virtual p0
    // What a workgroup needs before its first step besides its windows -- the slot tables in LDS and the step
    // length alpha (every workgroup folds the partial sums of the previous launch itself) -- is fetched AFTER the
    // first windows have been requested (round 3): the two latencies overlap instead of adding up at the start of
    // every workgroup's life.
    auto tables_and_alpha = [&]() {
        for (int i = tid; i < A.canon_npat * 9; i += kBlock) {
            if (SCHWZ_WALK_ZEROCOEF) {
                const int m = A.canon_mask[i / 9], k = i % 9;
                cpv[i] = PairVal{((m >> k) & 1) ? A.canon_val[2 * i] : 0.0, ((m >> (16 + k)) & 1) ? A.canon_val[2 * i + 1] : 0.0};
            } else {
                cpv[i] = PairVal{A.canon_val[2 * i], A.canon_val[2 * i + 1]};
            }
        }
        if (tid < A.canon_npat) cmask[tid] = A.canon_mask[tid];
        if (!INIT) {
            cg_alpha = a.cg_state->rho[a.it & 1] / fold_partials(a.pq_partials, a.pq_nparts, red);
            if (blockIdx.x == 0 && tid == 0 && a.alpha_out) *a.alpha_out = cg_alpha;
        }
        lds_barrier();
    };
    // partial-sum slots no workgroup of this launch or of its companion writes
    if (blockIdx.x == 0)
        for (int i = (int)gridDim.x + a.part_offset + tid; i < a.part_stride; i += kBlock) {
            a.partials[i] = a.partials[a.part_stride + i] = 0.0;
            if (DUAL) a.partials[2 * a.part_stride + i] = 0.0;
        }
    const int4 sg = A.sweep_seg[blockIdx.x];
    const int64_t PL = A.sweep_pl;
    const int band = sg.x, z0 = sg.y, z1 = sg.z;  // chain positions z0 <= z < z1
    constexpr bool GEN = RUNS == 0;
    const int blen = GEN ? sg.w : T;               // rows of the band inside the plane (even)
    const int64_t gmax = A.ncols - 2;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    typedef const unsigned __attribute__((address_space(4))) *const_words;
    typedef const int __attribute__((address_space(4))) *const_ints;
    if (z0 < z1) {
        // first row of the band at a chain position (scalar loads: the position is workgroup-uniform); a
        // position without a plane (the ends of a chain) reads plane 0 -- valid memory no mask lets through
        const const_ints chain = (const_ints)(uintptr_t)A.chain_plane;
        const const_ints chain_far = (const_ints)(uintptr_t)A.chain_far;
        auto band_row = [&](int z) -> int {  // rows fit 32 bits (the launch is for ncols < 2^28)
            const int k = chain[z];
            return (k < 0 ? 0 : k) * (int)PL + band * T;
        };
        // Pieces beyond the vector's ends (first / last band of a plane) are clamped to a valid address: no
        // row has an entry there, the masks drop what they deliver.
        auto clampg = [&](int g) -> int { return g < 0 ? 0 : (g > (int)gmax ? (int)gmax : g); };
        auto load_own = [&](int base, pvd2 (&reg)[NH]) {
#pragma unroll
            for (int k = 0; k < NH; ++k) __builtin_memcpy(&reg[k], a.x + clampg(base + 2 * (tid + k * kBlock)), 16);
        };
        auto store_own = [&](int z, const pvd2 (&reg)[NH]) {
            double *slot_p = own_ring + (size_t)(z & 3) * T;
#pragma unroll
            for (int k = 0; k < NH; ++k) *reinterpret_cast<pvd2 *>(slot_p + 2 * (tid + k * kBlock)) = ring_scale * reg[k];
        };
        // halo of a plane: the NX rows below the band (pieces 0 .. NX/2) and the NX rows above it
        auto load_halo = [&](int base, pvd2 (&reg)[NHL]) {
            const int lo = base - NX, up = base + T;
#pragma unroll
            for (int k = 0; k < NHL; ++k) {
                const int pc = min(tid + k * kBlock, NX - 1);  // NX pieces: NX/2 below, NX/2 above
                const int g = pc < NX / 2 ? lo + 2 * pc : up + 2 * (pc - NX / 2);
                __builtin_memcpy(&reg[k], a.x + clampg(g), 16);
            }
        };
        auto store_halo = [&](int z, const pvd2 (&reg)[NHL]) {
            double *slot_p = halo_ring + (size_t)(z & 1) * 2 * NX;
#pragma unroll
            for (int k = 0; k < NHL; ++k) {
                const int pc = min(tid + k * kBlock, NX - 1);
                *reinterpret_cast<pvd2 *>(slot_p + 2 * pc) = ring_scale * reg[k];
            }
        };
        struct Ahead {
            pvd2 r, d;
            unsigned w[GEN ? 1 : RUNS / 2];
        };
        auto fetch = [&](int base, int h) -> Ahead {
            Ahead f;
            const int row0 = base + h * kPairRows;
            const int ra = GEN ? base + min(h * kPairRows + 2 * tid, blen - 2) : row0 + 2 * tid;
            if (GEN) {
                f.w[0] = A.pair_id[ra >> 1];
            } else {
                const const_words q = (const_words)(uintptr_t)(A.pair_rle + (row0 / kPairRows) * ((RUNS ? RUNS : 8) / 8));
#pragma unroll
                for (int k = 0; k < (GEN ? 1 : RUNS / 2); ++k) f.w[k] = q[k];
            }
            if (SCHWZ_WALK_HINTS)
                f.r = __builtin_nontemporal_load(reinterpret_cast<const pvd2u *>(r_in + ra));
            else
                f.r = *reinterpret_cast<const pvd2u *>(r_in + ra);
            if (DIAGVEC) f.d = __builtin_nontemporal_load(reinterpret_cast<const pvd2 *>(a.dinv + ra));
            return f;
        };
        // Software pipeline, two steps deep, without register copies (the loop is unrolled by two and the
        // register sets alternate): own(z + 3) is requested while position z is computed, into the set whose
        // content -- own(z + 1) -- has just gone to LDS; r(z + 2) at the end of step z, into the set step z
        // has just consumed; the halo of position z + 1 (L2 hits) first thing in step z.  The band's first
        // rows at the positions z - 1 .. z + 3 travel in scalar registers (brow), one new one per step.
        pvd2 own_a[NH], own_b[NH], hreg[NHL];
        Ahead r_a[NH], r_b[NH];
        int brow[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) brow[k] = band_row(z0 - 1 + k);
        {
            // everything the first two positions need is requested before anything is waited for
            pvd2 first[NH], second[NH];
            load_own(brow[0], first);
            load_own(brow[1], second);
            load_halo(brow[1], hreg);
            load_own(brow[2], own_b);
            load_own(brow[3], own_a);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                r_a[h] = fetch(brow[1], h);
                r_b[h] = fetch(z0 + 1 < z1 ? brow[2] : brow[1], h);
            }
            tables_and_alpha();
            store_own(z0 - 1, first);
            store_own(z0, second);
        }
        // one position: `own_next` holds own(z + 1) on entry and own(z + 3) on exit, `rr` r(z) / r(z + 2)
        auto step = [&](int z, pvd2 (&own_next)[NH], Ahead (&rr)[NH]) {
            // brow[] = first rows at z - 1, z, z + 1, z + 2, z + 3
            store_own(z + 1, own_next);
            store_halo(z, hreg);
            lds_barrier();
            load_halo(brow[2], hreg);
            load_own(brow[4], own_next);
            const int far = chain_far[z];
            const bool plain_pos = DUAL ? ((const_ints)(uintptr_t)A.chain_dual)[z] == 0 : true;
            const double *cur = own_ring + (size_t)(z & 3) * T;
            const double *prv = own_ring + (size_t)((z + 3) & 3) * T;
            const double *nxt = own_ring + (size_t)((z + 1) & 3) * T;
            const double *hlo = halo_ring + (size_t)(z & 1) * 2 * NX, *hup = hlo + NX;
            // window each far slot reads (slots of a plane without that coupling read `cur`: masked anyway)
            const double *fb0 = (far & 3) == 1 ? prv : ((far & 3) == 2 ? nxt : cur);
            const double *fb1 = ((far >> 2) & 3) == 1 ? prv : (((far >> 2) & 3) == 2 ? nxt : cur);
            const double *fa0 = ((far >> 4) & 3) == 1 ? prv : (((far >> 4) & 3) == 2 ? nxt : cur);
            const double *fa1 = ((far >> 6) & 3) == 1 ? prv : (((far >> 6) & 3) == 2 ? nxt : cur);
            const int rbase = z + 2 < z1 ? brow[3] : brow[1];  // past the segment: a valid plane, never used
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const Ahead f = rr[h];
                const int i0u = h * kPairRows + 2 * tid;  // position inside the band
                const bool valid = !GEN || i0u < blen;
                const int i0 = GEN ? min(i0u, blen - 2) : i0u;
                const int ra = brow[1] + i0;
                // the id of the last run that starts at or before this lane's pair
                int pid = GEN ? (int)f.w[0] : (int)((f.w[0] >> 8) & 0xffu);
#pragma unroll
                for (int k = 1; k < RUNS; ++k) {
                    const unsigned e = (f.w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                    if ((unsigned)tid >= (e & 0xffu)) pid = (int)(e >> 8);
                }
                const int mask = SCHWZ_WALK_ZEROCOEF ? 0 : cmask[pid];
                pvd2 t[8];
                t[4] = *reinterpret_cast<const pvd2 *>(cur + i0);
                t[3].x = i0 > 0 ? cur[i0 - 1] : hlo[NX - 1];
                t[3].y = t[4].x;
                t[5].x = t[4].y;
                t[5].y = i0 + 2 < T ? cur[i0 + 2] : hup[0];
                t[2] = *reinterpret_cast<const pvd2 *>(i0 >= NX ? cur + (i0 - NX) : hlo + i0);
                t[6] = *reinterpret_cast<const pvd2 *>(i0 + NX < T ? cur + (i0 + NX) : hup + (i0 + NX - T));
                t[0] = *reinterpret_cast<const pvd2 *>(fb0 + i0);
                t[7] = *reinterpret_cast<const pvd2 *>(fa0 + i0);
                // masked sums without branches: an absent entry adds +0.0, which leaves a sum that began
                // at +0.0 unchanged bit for bit (such a sum is never -0.0).  The second far slots exist on a
                // subdomain's boundary planes only: a plane without them skips both (workgroup-uniform).
                double s0 = 0.0, s1 = 0.0;
                auto add = [&](int k, pvd2 tv) {
                    const PairVal v = cpv[pid * 9 + k];
                    const double p0 = v.a * tv.x, p1 = v.b * tv.y;
                    if (SCHWZ_WALK_ZEROCOEF) {
                        s0 += p0;
                        s1 += p1;
                    } else {
                        s0 += ((mask >> k) & 1) ? p0 : 0.0;
                        s1 += ((mask >> (16 + k)) & 1) ? p1 : 0.0;
                    }
                };
                add(0, t[0]);
                if (far & 0x0c) add(1, *reinterpret_cast<const pvd2 *>(fb1 + i0));
#pragma unroll
                for (int k = 2; k < 8; ++k) add(k, t[k]);
                if (far & 0xc0) add(8, *reinterpret_cast<const pvd2 *>(fa1 + i0));
                // r -= alpha q ; z = D^-1 r ; partial r.z and r.r  (kSpmvCgUpdate's epilogue, x deferred)
                const double r0 = INIT ? f.r.x - s0 : f.r.x - cg_alpha * s0;
                const double zz0 = a.diag_mode ? (DIAGVEC ? f.d.x : a.diag_uniform) * r0 : r0;
                const double r1 = INIT ? f.r.y - s1 : f.r.y - cg_alpha * s1;
                const double zz1 = a.diag_mode ? (DIAGVEC ? f.d.y : a.diag_uniform) * r1 : r1;
                if (GEN) {
                    const double t00 = r0 * zz0, t01 = r0 * r0, t10 = r1 * zz1, t11 = r1 * r1;
                    acc0 += valid ? t00 : 0.0;
                    acc1 += valid ? t01 : 0.0;
                    acc0 += valid ? t10 : 0.0;
                    acc1 += valid ? t11 : 0.0;
                } else {
                    acc0 += r0 * zz0;
                    acc1 += r0 * r0;
                    acc0 += r1 * zz1;
                    acc1 += r1 * r1;
                }
                if (DUAL) {
                    const double q0 = r0 * r0, q1 = r1 * r1;
                    acc2 += plain_pos ? q0 : 0.0;
                    acc2 += plain_pos ? q1 : 0.0;
                }
                const pvd2 rn = {r0, r1};
                if (SCHWZ_WALK_HINTS && !(INIT && SCHWZ_INIT_PLAIN_STORE))
                    __builtin_nontemporal_store(rn, reinterpret_cast<pvd2 *>(r_out + ra));
                else
                    *reinterpret_cast<pvd2 *>(r_out + ra) = rn;
                rr[h] = fetch(rbase, h);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) brow[k] = brow[k + 1];
            brow[4] = band_row(z + 4);
        };
        int z = z0;
        for (; z + 1 < z1; z += 2) {
            step(z, own_b, r_a);
            step(z + 1, own_a, r_b);
        }
        if (z < z1) step(z, own_b, r_a);
    } else if (!INIT && blockIdx.x == 0) {
        tables_and_alpha();  // (an empty first slot still reports alpha)
    }
    const double s0 = block_sum(acc0, red);
    const double s1 = block_sum(acc1, red);
    if (tid == 0) {
        a.partials[blockIdx.x] = s0;
        a.partials[a.part_stride + blockIdx.x] = s1;
    }
    if (DUAL) {
        const double s2 = block_sum(acc2, red);
        if (tid == 0) a.partials[2 * a.part_stride + blockIdx.x] = s2;
    }
}

// ---------------------------------------------------------------------------------------------------
// The same walk for the fused direction update + p.(A p) launch of symmetric matrices (kSpmvDirDotSym):
// beta from the folded partials, p' = z + beta p (z = D^-1 r, uniform or no Jacobi scaling) computed ONCE
// per element from coalesced loads of r and p -- into the LDS ring, and into the output buffer for the
// positions of the segment -- and the upper-triangle sum p'.(A p') = sum_i p'_i (a_ii p'_i + 2 sum_{j>i}
// a_ij p'_j) read from LDS: slots [0, +1, +NX, far after 0, far after 1].  The chunk-by-chunk form of this
// launch gathers r AND p at every entry; here every element of r and p is loaded once (plus the NX-row
// halo, L2 hits).  Same expression for p' everywhere, same products in the same order: the same bits per
// row.  Workgroup 0 advances CgState like cg_direction_kernel.  NHL / NH as above.
// FIRST: the first direction of a solve whose start launch was the INIT walk above: p' = z (nothing is
// read from p, beta does not exist yet), the partial sums of p'.(A p'), CgState untouched.
// ---------------------------------------------------------------------------------------------------
// SCHWZ_DD (build-time switch, tools/dd_ab.sh; round 3): bit 0 p' leaves with non-temporal stores; bit 1 the halo
// of a position is requested THREE positions ahead, i.e. in the same step in which the neighbouring band
// requests the same lines as its own window (a third halo slot in LDS); bit 2 the own r lines are plain loads
// (they are the neighbouring band's halo: a non-temporal load marks them for early eviction).  Before, the halo
// was asked for two steps after the neighbour had fetched the lines -- longer than a line lives in the 4 MiB L2
// at this stream rate -- and the PMC traffic of the launch was 1.17 x (256-wide planes) / 1.25 x (512-wide) its
// bytes: exactly the halo, fetched twice.  All three together, in-box (profiles/r03_dd_ab.txt): launch 0.0789 ->
// 0.0740 ms on the cube, 0.0861 -> 0.0729 ms on the 512 x 512 x 64 slab (0.585 -> 0.69 of peak; the step
// 1.95 -> 1.82 ms), the same bits.  0 restores the round-2 schedule.
#ifndef SCHWZ_DD
#define SCHWZ_DD 7
#endif
template <int NHL, int NH, bool FIRST = false, int RUNS = 8>
__global__ __launch_bounds__(kBlock) void spmv_pair_dirdot_sweep_kernel(CsrView A, SpmvArgs a)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) char sweep_lds[];
    constexpr int T = NH * kPairRows;
    const int NX = A.sweep_nx;
    double *const own_ring = reinterpret_cast<double *>(sweep_lds);  // p' of four chain positions [4][T]
    constexpr int kHaloSlots = (SCHWZ_DD & 2) ? 3 : 2;
    double *const halo_ring = own_ring + 4 * T;                       // p' of the NX rows above the band [kHaloSlots][NX]
    PairVal *const cpv = reinterpret_cast<PairVal *>(halo_ring + kHaloSlots * NX);
    int *const cmask = reinterpret_cast<int *>(cpv + A.canon_npat * 5);
    __shared__ double red[4];
    if (a.it >= a.cg_state->stop_iter) return;
    double cg_rho_new = 0.0, cg_rr = 0.0, cg_beta = 0.0;
    const int tid = threadIdx.x;
    // the slot tables and beta (folded from the previous launch's partial sums by every workgroup): fetched after
    // the first windows have been requested, like in the update walk above
    auto tables_and_beta = [&]() {
        for (int i = tid; i < A.canon_npat * 5; i += kBlock) {
            if (SCHWZ_WALK_ZEROCOEF) {
                const int m = A.canon_sym_mask[i / 5], k = i % 5;
                cpv[i] = PairVal{((m >> k) & 1) ? A.canon_sym_val[2 * i] : 0.0, ((m >> (16 + k)) & 1) ? A.canon_sym_val[2 * i + 1] : 0.0};
            } else {
                cpv[i] = PairVal{A.canon_sym_val[2 * i], A.canon_sym_val[2 * i + 1]};
            }
        }
        if (tid < A.canon_npat) cmask[tid] = A.canon_sym_mask[tid];
        if (!FIRST) {
            cg_rho_new = fold_partials(a.pq_partials, a.pq_nparts, red);
            cg_rr = fold_partials(a.pq_partials + a.pq_nparts, a.pq_nparts, red);
            cg_beta = cg_rho_new / a.cg_state->rho[a.it & 1];
        }
        lds_barrier();
    };
    if (blockIdx.x == 0)
        for (int i = (int)gridDim.x + a.part_offset + tid; i < a.part_stride; i += kBlock)
            a.partials[i] = a.partials[a.part_stride + i] = 0.0;
    const int4 sg = A.sweep_seg_dir[blockIdx.x];  // (its own table: the band height may differ from the update launch's)
    const int64_t PL = A.sweep_pl;
    const int band = sg.x, z0 = sg.y, z1 = sg.z;
    constexpr bool GEN = RUNS == 0;          // planes that are not whole chunks: byte ids, partial last band (see above)
    const int blen = GEN ? sg.w : T;
    const int64_t gmax = A.ncols - 2;
    const pvd2 du = {a.diag_uniform, a.diag_uniform};
    double acc0 = 0.0;
    typedef const unsigned __attribute__((address_space(4))) *const_words;
    typedef const int __attribute__((address_space(4))) *const_ints;
    if (z0 < z1) {
        const const_ints chain = (const_ints)(uintptr_t)A.chain_plane;
        const const_ints chain_far = (const_ints)(uintptr_t)A.chain_far;
        auto band_row = [&](int z) -> int {  // rows fit 32 bits (the launch is for ncols < 2^28)
            const int k = chain[z];
            return (k < 0 ? 0 : k) * (int)PL + band * T;
        };
        auto clampg = [&](int g) -> int { return g < 0 ? 0 : (g > (int)gmax ? (int)gmax : g); };
        // p' at one 16-byte piece: the expression of dir2 / cg_direction_kernel
        const pvd2 p_scale = {a.p_scale, a.p_scale};  // 1.0 except where a.x is r0 and p0 = p_scale x r0 (virtual p0)
        auto newp = [&](pvd2 rv, pvd2 pv) -> pvd2 {
            const pvd2 zv = a.diag_mode ? du * rv : rv;
            if (FIRST) return zv;
            const pvd2 ps = p_scale * pv;
            pvd2 o;
            o.x = __builtin_fma(cg_beta, ps.x, zv.x);
            o.y = __builtin_fma(cg_beta, ps.y, zv.y);
            return o;
        };
        struct Own {
            pvd2 r[NH], p[NH];
        };
        struct Halo {
            pvd2 r[NHL], p[NHL];
        };
        // (GEN: pieces beyond the band's rows inside the plane repeat its last piece)
        auto piece_of = [&](int k) -> int { return GEN ? min(tid + k * kBlock, blen / 2 - 1) : tid + k * kBlock; };
        auto load_own = [&](int base, Own &o) {
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                const int g = clampg(base + 2 * piece_of(k));
                if (SCHWZ_DD & 4)
                    o.r[k] = *reinterpret_cast<const pvd2 *>(a.cg_r + g);
                else
                    o.r[k] = __builtin_nontemporal_load(reinterpret_cast<const pvd2 *>(a.cg_r + g));
                if (FIRST)
                    o.p[k] = o.r[k];
                else
                    __builtin_memcpy(&o.p[k], a.x + g, 16);
            }
        };
        // p' of the band at a position: to the ring, and to the output vector where the position belongs to
        // the segment
        auto store_own = [&](int z, int base, const Own &o, bool out) {
            double *slot_p = own_ring + (size_t)(z & 3) * T;
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                const pvd2 v = newp(o.r[k], o.p[k]);
                *reinterpret_cast<pvd2 *>(slot_p + 2 * (tid + k * kBlock)) = v;
                if (out && a.y) {  // (a.y == nullptr: a direction nobody reads from memory -- windows and sums only)
                    if ((SCHWZ_DD & 1) && SCHWZ_WALK_HINTS)
                        __builtin_nontemporal_store(v, reinterpret_cast<pvd2 *>(a.y + base + 2 * piece_of(k)));
                    else
                        __builtin_memcpy(a.y + base + 2 * piece_of(k), &v, 16);
                }
            }
        };
        auto load_halo = [&](int base, Halo &hh) {
            const int up = base + T;
#pragma unroll
            for (int k = 0; k < NHL; ++k) {
                const int pc = min(tid + k * kBlock, NX / 2 - 1);
                const int g = clampg(up + 2 * pc);
                __builtin_memcpy(&hh.r[k], a.cg_r + g, 16);
                if (FIRST)
                    hh.p[k] = hh.r[k];
                else
                    __builtin_memcpy(&hh.p[k], a.x + g, 16);
            }
        };
        // (with three slots the slot of a position is passed in: position modulo 3, kept incrementally)
        auto store_halo = [&](int hslot, const Halo &hh) {
            double *slot_p = halo_ring + (size_t)hslot * NX;
#pragma unroll
            for (int k = 0; k < NHL; ++k) {
                const int pc = min(tid + k * kBlock, NX / 2 - 1);
                *reinterpret_cast<pvd2 *>(slot_p + 2 * pc) = newp(hh.r[k], hh.p[k]);
            }
        };
        struct Rle {
            unsigned w[GEN ? 1 : RUNS / 2];
        };
        auto fetch_rle = [&](int base, int h) -> Rle {
            Rle f;
            if (GEN) {
                f.w[0] = A.pair_id[(base + min(h * kPairRows + 2 * tid, blen - 2)) >> 1];
            } else {
                const const_words q = (const_words)(uintptr_t)(A.pair_rle + ((base + h * kPairRows) / kPairRows) * ((RUNS ? RUNS : 8) / 8));
#pragma unroll
                for (int k = 0; k < (GEN ? 1 : RUNS / 2); ++k) f.w[k] = q[k];
            }
            return f;
        };
        Own own_a, own_b;
        Halo hreg, hreg_b;
        Rle rle_a[NH], rle_b[NH];
        int brow[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) brow[k] = band_row(z0 - 1 + k);
        {
            // the previous position is read by rows whose far-after slot points backwards along the chain
            Own before, first;
            Halo h0;
            load_own(brow[0], before);
            load_own(brow[1], first);
            load_halo(brow[1], (SCHWZ_DD & 2) ? h0 : hreg);
            load_own(brow[2], own_b);
            load_own(brow[3], own_a);
            if (SCHWZ_DD & 2) {
                // halo of the first position goes to its slot now; the next two stay in flight like the own windows
                load_halo(brow[2], hreg);    // position z0 + 1
                load_halo(brow[3], hreg_b);  // position z0 + 2
            }
            tables_and_beta();
            if (SCHWZ_DD & 2) store_halo(0, h0);
            store_own(z0 - 1, brow[0], before, false);
            store_own(z0, brow[1], first, true);
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            rle_a[h] = fetch_rle(brow[1], h);
            rle_b[h] = fetch_rle(z0 + 1 < z1 ? brow[2] : brow[1], h);
        }
        int hs = 0;  // halo slot of the position being computed (three-slot form)
        // one position: `own_next` holds r, p of position z + 1 on entry and of position z + 3 on exit; in the
        // three-slot form `hal_next` holds the halo of position z + 1 on entry and of position z + 3 on exit
        auto step = [&](int z, Own &own_next, Rle (&rl)[NH], Halo &hal_next) {
            store_own(z + 1, brow[2], own_next, z + 1 < z1);
            const int hs_next = hs == 2 ? 0 : hs + 1;
            if (SCHWZ_DD & 2)
                store_halo(hs_next, hal_next);
            else
                store_halo(z & 1, hreg);
            lds_barrier();
            if (SCHWZ_DD & 2)
                load_halo(brow[4], hal_next);
            else
                load_halo(brow[2], hreg);
            load_own(brow[4], own_next);
            const int far = chain_far[z];
            const double *cur = own_ring + (size_t)(z & 3) * T;
            const double *prv = own_ring + (size_t)((z + 3) & 3) * T;
            const double *nxt = own_ring + (size_t)((z + 1) & 3) * T;
            const double *hup = halo_ring + (size_t)((SCHWZ_DD & 2) ? hs : (z & 1)) * NX;
            const double *fa0 = ((far >> 4) & 3) == 1 ? prv : (((far >> 4) & 3) == 2 ? nxt : cur);
            const double *fa1 = ((far >> 6) & 3) == 1 ? prv : (((far >> 6) & 3) == 2 ? nxt : cur);
            const int rbase = z + 2 < z1 ? brow[3] : brow[1];
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const Rle f = rl[h];
                const int i0u = h * kPairRows + 2 * tid;
                const bool valid = !GEN || i0u < blen;
                const int i0 = GEN ? min(i0u, blen - 2) : i0u;
                int pid = GEN ? (int)f.w[0] : (int)((f.w[0] >> 8) & 0xffu);
#pragma unroll
                for (int k = 1; k < RUNS; ++k) {
                    const unsigned e = (f.w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                    if ((unsigned)tid >= (e & 0xffu)) pid = (int)(e >> 8);
                }
                const int mask = SCHWZ_WALK_ZEROCOEF ? 0 : cmask[pid];
                pvd2 t[4];
                t[0] = *reinterpret_cast<const pvd2 *>(cur + i0);
                t[1].x = t[0].y;
                t[1].y = i0 + 2 < T ? cur[i0 + 2] : hup[0];
                t[2] = *reinterpret_cast<const pvd2 *>(i0 + NX < T ? cur + (i0 + NX) : hup + (i0 + NX - T));
                t[3] = *reinterpret_cast<const pvd2 *>(fa0 + i0);
                double s0 = 0.0, s1 = 0.0;
                auto add = [&](int k, pvd2 tv) {
                    const PairVal v = cpv[pid * 5 + k];
                    const double p0 = v.a * tv.x, p1 = v.b * tv.y;
                    if (SCHWZ_WALK_ZEROCOEF) {
                        s0 += p0;
                        s1 += p1;
                    } else {
                        s0 += ((mask >> k) & 1) ? p0 : 0.0;
                        s1 += ((mask >> (16 + k)) & 1) ? p1 : 0.0;
                    }
                };
#pragma unroll
                for (int k = 0; k < 4; ++k) add(k, t[k]);
                if (far & 0xc0) add(4, *reinterpret_cast<const pvd2 *>(fa1 + i0));
                if (GEN) {
                    const double u0 = t[0].x * s0, u1 = t[0].y * s1;
                    acc0 += valid ? u0 : 0.0;
                    acc0 += valid ? u1 : 0.0;
                } else {
                    acc0 += t[0].x * s0;
                    acc0 += t[0].y * s1;
                }
                rl[h] = fetch_rle(rbase, h);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) brow[k] = brow[k + 1];
            brow[4] = band_row(z + 4);
            hs = hs_next;
        };
        int z = z0;
        for (; z + 1 < z1; z += 2) {
            step(z, own_b, rle_a, hreg);
            step(z + 1, own_a, rle_b, hreg_b);
        }
        if (z < z1) step(z, own_b, rle_a, hreg);
    } else if (!FIRST && blockIdx.x == 0) {
        tables_and_beta();  // (an empty first slot still advances the CG state below)
    }
    const double s0 = block_sum(acc0, red);
    if (tid == 0) {
        a.partials[blockIdx.x] = s0;
        a.partials[a.part_stride + blockIdx.x] = 0.0;
    }
    if (!FIRST && blockIdx.x == 0 && tid == 0) {
        // what cg_direction_kernel's workgroup 0 does (see the chunk-by-chunk kernel's epilogue)
        CgState *st = const_cast<CgState *>(a.cg_state);
        st->rho[(a.it + 1) & 1] = cg_rho_new;
        st->rr = cg_rr;
        st->iters = st->iters + 1;
        if (sqrt(cg_rr) <= a.cg_rtol * st->r0) st->stop_iter = 0;
    }
}

// whether a solve on this matrix can START in the z-sweep walk: the INIT form of the update walk and the
// FIRST form of the fused direction walk exist for the shapes below, and only where the walk covers
// every row (no companion launch: cubes and z-slab subdomains)
bool pair_sweep_start_ok(const CsrView &A, int grid)
{
    if (!(A.pair_single && A.sweep_nslots > 0 && A.canon_sym_val && A.sweep_gen_blocks == 0 && A.sweep_nslots <= grid &&
          A.ncols < (int64_t(1) << 28)))
        return false;
    const int nh = A.sweep_T / kPairRows, nh_dir = A.sweep_T_dir / kPairRows;
    const int nhl_upd = (A.sweep_nx + kBlock - 1) / kBlock, nhl_dir = (A.sweep_nx / 2 + kBlock - 1) / kBlock;
    return (nh == 1 || nh == 2) && (nh_dir == 1 || nh_dir == 2 || nh_dir == 4) && nhl_upd <= 4 && nhl_dir <= 2 &&
           A.sweep_nslots_dir <= grid;
}

// ... and with the fused dual residual: the flagged planes' chunk list exists and its launch fits the
// partial-sum slots behind the walk's
bool pair_sweep_dual_ok(const CsrView &A, int grid)
{
    return pair_sweep_start_ok(A, grid) && A.chain_dual && A.dual_chunks && A.dual_blocks > 0 &&
           A.sweep_nslots + A.dual_blocks <= grid;
}

// One instantiation of the z-sweep update / start walk: raises its dynamic-LDS limit once, launches it.
// false: the device does not grant 96 KiB of dynamic LDS to the kernel (nothing was launched).
template <int L_, int H_, bool DV, bool INIT, bool DUAL, int RUNS>
static bool launch_sweep_instance(const CsrView &A, const SpmvArgs &b, size_t lds, hipStream_t s)
{
    static const hipError_t e = hipFuncSetAttribute((const void *)spmv_pair_sweep_kernel<L_, H_, DV, INIT, DUAL, RUNS>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);
    if (e != hipSuccess || lds > (size_t)(96 << 10)) return false;
    hipLaunchKernelGGL((spmv_pair_sweep_kernel<L_, H_, DV, INIT, DUAL, RUNS>), dim3(A.sweep_nslots), dim3(kBlock), lds, s, A, b);
    return true;
}

// ... chosen by the run-length record of the matrix (8 / 16 runs per chunk)
template <int L_, int H_, bool DV, bool INIT, bool DUAL>
static bool launch_sweep_variant(const CsrView &A, const SpmvArgs &b, size_t lds, hipStream_t s)
{
    if (A.sweep_gen_mode) {
        // planes that are not whole chunks (byte ids, partial last band); no dual form there
        if (DUAL) return false;
        return launch_sweep_instance<L_, H_, DV, INIT, false, 0>(A, b, lds, s);
    }
    return A.pair_rle_runs == 16 ? launch_sweep_instance<L_, H_, DV, INIT, DUAL, 16>(A, b, lds, s)
                                 : launch_sweep_instance<L_, H_, DV, INIT, DUAL, 8>(A, b, lds, s);
}

int launch_spmv_pair(const CsrView &A, int mode, const SpmvArgs &a, int grid, hipStream_t s)
{
    const bool wide = A.ncols >= (int64_t(1) << 28);  // byte offsets of x beyond 32 bits
    if ((mode == kSpmvResidInit || mode == kSpmvResidDual) && a.sweep_init) {
        // CG start in the z-sweep walk (pcg_begin asks for it only where pair_sweep_start_ok holds and the
        // Jacobi diagonal is a scalar or absent): r = b - A x to a.y, partial r.z and r.r; p is not written
        if (!pair_sweep_start_ok(A, grid) || a.diag_mode == 1 || a.diag_mode == 2 || a.dinv) {
            set_error("launch_spmv_pair: the z-sweep start launch does not apply to this matrix");
            return SCHWZ_ERR_INVALID;
        }
        const int nhl = (A.sweep_nx + kBlock - 1) / kBlock, nh = A.sweep_T / kPairRows;
        const size_t lds = (size_t)(4 * A.sweep_T + 4 * A.sweep_nx) * sizeof(double) + (size_t)A.canon_npat * (9 * 16 + 4);
        SpmvArgs b = a;
        b.part_stride = grid;
        b.part_offset = 0;
        // fused check residual on a second vector (a subdomain with neighbours): the planes where it needs its
        // own product are left out of the third partial bank here and added by a listed launch below
        const bool with_dual = mode == kSpmvResidDual && a.x2 != nullptr;
        if (with_dual && !pair_sweep_dual_ok(A, grid)) {
            set_error("launch_spmv_pair: the z-sweep dual start launch does not apply to this matrix");
            return SCHWZ_ERR_INVALID;
        }
#define SCHWZ_SWEEP_INIT(L_, H_)                                                                              \
    ok = with_dual ? launch_sweep_variant<L_, H_, false, true, true>(A, b, lds, s)                            \
                   : launch_sweep_variant<L_, H_, false, true, false>(A, b, lds, s);
        bool ok = false;
        if (nh == 1 && nhl == 1) SCHWZ_SWEEP_INIT(1, 1)
        else if (nh == 1 && nhl == 2) SCHWZ_SWEEP_INIT(2, 1)
        else if (nh == 2 && nhl == 1) SCHWZ_SWEEP_INIT(1, 2)
        else if (nh == 2 && nhl == 2) SCHWZ_SWEEP_INIT(2, 2)
        else if (nh == 1) SCHWZ_SWEEP_INIT(4, 1)
        else SCHWZ_SWEEP_INIT(4, 2)
        if (!ok) {
            set_error("launch_spmv_pair: the z-sweep kernels cannot have their dynamic LDS on this device");
            return SCHWZ_ERR_HIP;
        }
#undef SCHWZ_SWEEP_INIT
        SCHWZ_HIP_TRY(hipGetLastError());
        if (with_dual) {
            // ||b - A x2||^2 over the flagged planes: chunk by chunk on x2, partial sums into the third bank
            // behind the walk's (the launch also writes zeros into the same slots of the second bank)
            SpmvArgs c;
            c.x = a.x2;
            c.b = a.b;
            c.row_limit = a.row_limit;
            c.sweep = 2;
            c.partials = a.partials + grid;
            c.part_stride = grid;
            c.part_offset = A.sweep_nslots;
            hipLaunchKernelGGL((spmv_pair_kernel<kSpmvResidNorm, false, true>), dim3(A.dual_blocks), dim3(kBlock),
                               kPairTableLds, s, A, c);
            SCHWZ_HIP_TRY(hipGetLastError());
        }
        return SCHWZ_OK;
    }
    if (mode == kSpmvDirDotSym && a.sweep_first) {
        // first direction of such a solve: p' = D^-1 r and the partial sums of p'.(A p'), CgState untouched
        const CsrView &A_in = A;
        if (!pair_sweep_start_ok(A, grid) || a.diag_mode == 1 || a.diag_mode == 2) {
            set_error("launch_spmv_pair: the z-sweep first-direction launch does not apply to this matrix");
            return SCHWZ_ERR_INVALID;
        }
        // (the kernel reads its segments from sweep_seg_dir: this launch gets a view whose direction table IS the
        // first-direction table)
        CsrView Af = A_in;
        if (A_in.sweep_seg_first && A_in.sweep_nslots_first <= grid) {
            Af.sweep_T_dir = A_in.sweep_T_first;
            Af.sweep_seg_dir = A_in.sweep_seg_first;
            Af.sweep_nslots_dir = A_in.sweep_nslots_first;
        }
        const CsrView &A = Af;
        const int nhl = (A.sweep_nx / 2 + kBlock - 1) / kBlock, nh = A.sweep_T_dir / kPairRows;
        const size_t lds = (size_t)(4 * A.sweep_T_dir + ((SCHWZ_DD & 2) ? 3 : 2) * A.sweep_nx) * sizeof(double) + (size_t)A.canon_npat * (5 * 16 + 4);
        SpmvArgs b = a;
        b.part_stride = grid;
        b.part_offset = 0;
#define SCHWZ_DIRDOT_FIRST(L_, H_)                                                                                        \
    {                                                                                                                    \
        static const hipError_t e0 = hipFuncSetAttribute((const void *)spmv_pair_dirdot_sweep_kernel<L_, H_, true>,      \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);          \
        if (e0 != hipSuccess) {                                                                                          \
            set_error("launch_spmv_pair: the z-sweep kernels cannot have 96 KiB of dynamic LDS on this device");         \
            return SCHWZ_ERR_HIP;                                                                                        \
        }                                                                                                                \
        if (A.sweep_gen_mode) {                                                                                          \
            static const hipError_t e3 = hipFuncSetAttribute((const void *)spmv_pair_dirdot_sweep_kernel<L_, H_, true, 0>, \
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);      \
            if (e3 != hipSuccess) {                                                                                      \
                set_error("launch_spmv_pair: the z-sweep kernels cannot have 96 KiB of dynamic LDS on this device");     \
                return SCHWZ_ERR_HIP;                                                                                    \
            }                                                                                                            \
            hipLaunchKernelGGL((spmv_pair_dirdot_sweep_kernel<L_, H_, true, 0>), dim3(A.sweep_nslots_dir), dim3(kBlock), lds, s, A, b); \
        } else if (A.pair_rle_runs == 16) {                                                                              \
            static const hipError_t e2 = hipFuncSetAttribute((const void *)spmv_pair_dirdot_sweep_kernel<L_, H_, true, 16>, \
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);      \
            if (e2 != hipSuccess) {                                                                                      \
                set_error("launch_spmv_pair: the z-sweep kernels cannot have 96 KiB of dynamic LDS on this device");     \
                return SCHWZ_ERR_HIP;                                                                                    \
            }                                                                                                            \
            hipLaunchKernelGGL((spmv_pair_dirdot_sweep_kernel<L_, H_, true, 16>), dim3(A.sweep_nslots_dir), dim3(kBlock), lds, s, A, b); \
        } else                                                                                                           \
            hipLaunchKernelGGL((spmv_pair_dirdot_sweep_kernel<L_, H_, true>), dim3(A.sweep_nslots_dir), dim3(kBlock), lds, s, A, b); \
    }
        if (nh == 1 && nhl == 1) SCHWZ_DIRDOT_FIRST(1, 1)
        else if (nh == 1) SCHWZ_DIRDOT_FIRST(2, 1)
        else if (nh == 2 && nhl == 1) SCHWZ_DIRDOT_FIRST(1, 2)
        else if (nh == 2) SCHWZ_DIRDOT_FIRST(2, 2)
        else if (nhl == 1) SCHWZ_DIRDOT_FIRST(1, 4)
        else SCHWZ_DIRDOT_FIRST(2, 4)
#undef SCHWZ_DIRDOT_FIRST
        SCHWZ_HIP_TRY(hipGetLastError());
        return SCHWZ_OK;
    }
    if (mode == kSpmvCgUpdate && !wide && A.pair_single && A.sweep_nslots > 0 && !a.cg_x && a.diag_mode != 2 &&
        A.sweep_nslots + A.sweep_gen_blocks <= grid) {
        const char *sweep_env = std::getenv("SCHWZ_CG_SWEEP");  // read per launch: tests switch it
        if (!(sweep_env && sweep_env[0] == '0')) {
            // z-sweep walk + (where boundary planes or overlap rows exist) the listed walk of the chunks it
            // leaves out; the consumer folds `grid` partial sums per bank, as after a chunk-by-chunk launch
            const int nhl = (A.sweep_nx + kBlock - 1) / kBlock, nh = A.sweep_T / kPairRows;  // halo / own pieces per lane
            const size_t lds = (size_t)(4 * A.sweep_T + 4 * A.sweep_nx) * sizeof(double) + (size_t)A.canon_npat * (9 * 16 + 4);
            SpmvArgs b = a;
            b.part_stride = grid;
            b.part_offset = A.sweep_gen_blocks;  // slots of the companion launch follow this one's
            const bool dv = a.diag_mode == 1;
#define SCHWZ_SWEEP_LAUNCH(L_, H_)                                                                   \
    launched = dv ? launch_sweep_variant<L_, H_, true, false, false>(A, b, lds, s)                 \
                  : launch_sweep_variant<L_, H_, false, false, false>(A, b, lds, s);
            bool launched = true;  // false: no such instantiation, or no LDS for it -- chunk by chunk then
            if (nh == 1 && nhl == 1) SCHWZ_SWEEP_LAUNCH(1, 1)
            else if (nh == 1 && nhl == 2) SCHWZ_SWEEP_LAUNCH(2, 1)
            else if (nh == 2 && nhl == 1) SCHWZ_SWEEP_LAUNCH(1, 2)
            else if (nh == 2 && nhl == 2) SCHWZ_SWEEP_LAUNCH(2, 2)
            else if (nh == 1 && nhl <= 4) SCHWZ_SWEEP_LAUNCH(4, 1)
            else if (nh == 2 && nhl <= 4) SCHWZ_SWEEP_LAUNCH(4, 2)
            else launched = false;
#undef SCHWZ_SWEEP_LAUNCH
            if (launched) {
                SCHWZ_HIP_TRY(hipGetLastError());
                if (A.sweep_gen_blocks > 0) {
                    SpmvArgs c = a;
                    c.sweep = 1;
                    c.part_stride = grid;
                    c.part_offset = A.sweep_nslots;
                    hipLaunchKernelGGL((spmv_pair_kernel<kSpmvCgUpdate, false, true>), dim3(A.sweep_gen_blocks), dim3(kBlock),
                                       kPairTableLds, s, A, c);
                    SCHWZ_HIP_TRY(hipGetLastError());
                }
                return SCHWZ_OK;
            }
        }
    }
    if (mode == kSpmvDirDotSym && !wide && A.pair_single && A.sweep_nslots > 0 && A.canon_sym_val && a.diag_mode != 1 &&
        a.diag_mode != 2 && A.sweep_nslots_dir + A.sweep_gen_blocks <= grid) {
        const char *sweep_env = std::getenv("SCHWZ_CG_SWEEP");
        if (!(sweep_env && sweep_env[0] == '0')) {
            const int nhl = (A.sweep_nx / 2 + kBlock - 1) / kBlock, nh = A.sweep_T_dir / kPairRows;
            const size_t lds = (size_t)(4 * A.sweep_T_dir + ((SCHWZ_DD & 2) ? 3 : 2) * A.sweep_nx) * sizeof(double) + (size_t)A.canon_npat * (5 * 16 + 4);
            if (nhl <= 2) {
                // the companion first: the z-sweep kernel's workgroup 0 advances CgState for both
                if (A.sweep_gen_blocks > 0) {
                    SpmvArgs c = a;
                    c.sweep = 1;
                    c.part_stride = grid;
                    c.part_offset = A.sweep_nslots_dir;
                    hipLaunchKernelGGL((spmv_pair_kernel<kSpmvDirDotSym, false, true>), dim3(A.sweep_gen_blocks), dim3(kBlock),
                                       kPairTableLds, s, A, c);
                    SCHWZ_HIP_TRY(hipGetLastError());
                }
                SpmvArgs b = a;
                b.part_stride = grid;
                b.part_offset = A.sweep_gen_blocks;
#define SCHWZ_DIRDOT_LAUNCH(L_, H_)                                                                                      \
    {                                                                                                                    \
        static const hipError_t e0 = hipFuncSetAttribute((const void *)spmv_pair_dirdot_sweep_kernel<L_, H_>,            \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);          \
        if (e0 != hipSuccess) {                                                                                          \
            set_error("launch_spmv_pair: the z-sweep kernels cannot have 96 KiB of dynamic LDS on this device");         \
            return SCHWZ_ERR_HIP;                                                                                        \
        }                                                                                                                \
        if (A.sweep_gen_mode) {                                                                                          \
            static const hipError_t e3 = hipFuncSetAttribute((const void *)spmv_pair_dirdot_sweep_kernel<L_, H_, false, 0>, \
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);      \
            if (e3 != hipSuccess) {                                                                                      \
                set_error("launch_spmv_pair: the z-sweep kernels cannot have 96 KiB of dynamic LDS on this device");     \
                return SCHWZ_ERR_HIP;                                                                                    \
            }                                                                                                            \
            hipLaunchKernelGGL((spmv_pair_dirdot_sweep_kernel<L_, H_, false, 0>), dim3(A.sweep_nslots_dir), dim3(kBlock), lds, s, A, b); \
        } else if (A.pair_rle_runs == 16) {                                                                              \
            static const hipError_t e2 = hipFuncSetAttribute((const void *)spmv_pair_dirdot_sweep_kernel<L_, H_, false, 16>, \
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);      \
            if (e2 != hipSuccess) {                                                                                      \
                set_error("launch_spmv_pair: the z-sweep kernels cannot have 96 KiB of dynamic LDS on this device");     \
                return SCHWZ_ERR_HIP;                                                                                    \
            }                                                                                                            \
            hipLaunchKernelGGL((spmv_pair_dirdot_sweep_kernel<L_, H_, false, 16>), dim3(A.sweep_nslots_dir), dim3(kBlock), lds, s, A, b); \
        } else                                                                                                           \
            hipLaunchKernelGGL((spmv_pair_dirdot_sweep_kernel<L_, H_>), dim3(A.sweep_nslots_dir), dim3(kBlock), lds, s, A, b); \
    }
                if (nh == 1 && nhl == 1) SCHWZ_DIRDOT_LAUNCH(1, 1)
                else if (nh == 1) SCHWZ_DIRDOT_LAUNCH(2, 1)
                else if (nh == 2 && nhl == 1) SCHWZ_DIRDOT_LAUNCH(1, 2)
                else if (nh == 2) SCHWZ_DIRDOT_LAUNCH(2, 2)
                else if (nhl == 1) SCHWZ_DIRDOT_LAUNCH(1, 4)
                else SCHWZ_DIRDOT_LAUNCH(2, 4)
#undef SCHWZ_DIRDOT_LAUNCH
                SCHWZ_HIP_TRY(hipGetLastError());
                return SCHWZ_OK;
            }
        }
    }
    if (a.p0_virtual) {
        // (the chunk-by-chunk kernels read a stored p0; pcg_iterate only asks for the virtual one where the walks apply)
        set_error("launch_spmv_pair: a launch on the virtual first direction was not served by the z-sweep walk");
        return SCHWZ_ERR_INVALID;
    }
#define SCHWZ_PAIR_LAUNCH(M)                                                                          \
    if (wide)                                                                                         \
        hipLaunchKernelGGL((spmv_pair_kernel<M, true, false>), dim3(grid), dim3(kBlock), kPairTableLds, s, A, a);  \
    else if (A.pair_single)                                                                           \
        hipLaunchKernelGGL((spmv_pair_kernel<M, false, true>), dim3(grid), dim3(kBlock), kPairTableLds, s, A, a);  \
    else                                                                                              \
        hipLaunchKernelGGL((spmv_pair_kernel<M, false, false>), dim3(grid), dim3(kBlock), kPairTableLds, s, A, a);
    switch (mode) {
    case kSpmvPlain: SCHWZ_PAIR_LAUNCH(kSpmvPlain) break;
    case kSpmvDot: SCHWZ_PAIR_LAUNCH(kSpmvDot) break;
    case kSpmvResidInit: SCHWZ_PAIR_LAUNCH(kSpmvResidInit) break;
    case kSpmvResidDual: SCHWZ_PAIR_LAUNCH(kSpmvResidDual) break;
    case kSpmvDotOnly: SCHWZ_PAIR_LAUNCH(kSpmvDotOnly) break;
    case kSpmvDotSym: SCHWZ_PAIR_LAUNCH(kSpmvDotSym) break;
    case kSpmvDirDotSym: SCHWZ_PAIR_LAUNCH(kSpmvDirDotSym) break;
    case kSpmvDirDotSymVec: SCHWZ_PAIR_LAUNCH(kSpmvDirDotSymVec) break;
    case kSpmvCgUpdate: SCHWZ_PAIR_LAUNCH(kSpmvCgUpdate) break;
    default: SCHWZ_PAIR_LAUNCH(kSpmvResidNorm) break;
    }
#undef SCHWZ_PAIR_LAUNCH
    SCHWZ_HIP_TRY(hipGetLastError());
    return SCHWZ_OK;
}

namespace {

struct PairEntryH {
    schwz_idx off;
    int flags;
    uint64_t va, vb;
    bool operator==(const PairEntryH &o) const { return off == o.off && flags == o.flags && va == o.va && vb == o.vb; }
};

struct PairTable {
    int npat = 0, lmax = 0;
    std::vector<uint8_t> len;
    std::vector<PairEntryH> ent;  // [npat][lmax], padded with {0,0,0,0}
    uint64_t hash = 0;
    bool same(const PairTable &o) const { return npat == o.npat && lmax == o.lmax && len == o.len && ent == o.ent; }
};

template <typename T>
int upv(const std::vector<T> &h, void **d)
{
    *d = nullptr;
    SCHWZ_HIP_TRY(hipMalloc(d, (h.empty() ? 1 : h.size()) * sizeof(T)));
    if (!h.empty()) SCHWZ_HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return SCHWZ_OK;
}

}  // namespace

// The z-sweep walk (spmv_pair_sweep_kernel, spmv_pair_dirdot_sweep_kernel): host side.
//
// Geometry.  The matrix is cut into PLANES of PL consecutive rows (PL = the dominant far offset of the
// canonical layout, NX its in-plane line offset).  A row of plane k may couple to {-NX, -1, 0, +1, +NX}
// inside its plane and to the row at the SAME in-plane position of at most two other planes: for the
// interior of a grid in natural order those are k - 1 and k + 1; for a subdomain whose overlap planes are
// appended behind its interior the first interior plane couples to the plane PL rows on and
// to the lower overlap plane far behind it, and so on.  Planes linked like that form CHAINS; a workgroup
// sweeps a band of rows along a chain and keeps the windows of three consecutive chain positions in LDS,
// so every far operand of a row is in the window before or after its own -- wherever the numbering put
// that plane.  Slots of a row pair, in the order the entries are summed (= ascending column, the CSR
// order): [far before 0, far before 1, -NX, -1, 0, +1, +NX, far after 0, far after 1]; which window
// (previous / next chain position) a far slot reads is a property of the plane (sweep_far).
// A plane takes part when all its chunks are full, run-length coded, and every pattern in them fits those
// slots; the rest of the matrix is left to the companion launch (sweep_gen).
static int build_sweep(schwz_csr *A, int64_t nrows, const PairTable &tb, const PairTable *ts,
                       const std::vector<uint8_t> &pair_id, const std::vector<uint16_t> &rle, int64_t ntiles)
{
    // SCHWZ_SWEEP_WHY=1: says on stderr why a matrix gets no z-sweep walk
    auto no_walk = [](const char *why) {
        static const bool say = [] {
            const char *e = std::getenv("SCHWZ_SWEEP_WHY");
            return e && e[0] == '1';
        }();
        if (say) std::fprintf(stderr, "[schwz] no z-sweep walk: %s\n", why);
        return SCHWZ_OK;
    };
    const char *sw_env = std::getenv("SCHWZ_SPMV_SWEEP");
    const int sw_mode = sw_env ? std::atoi(sw_env) : 1;
    const int *cn = A->v.pair_canon;
    // A 5-point (2-D) stencil {-N, -1, 0, +1, +N} in natural order is the same walk with the x LINE in the role of the
    // plane: the canonical layout then repeats its outer offsets (cn[0] == cn[1] == -N, cn[5] == cn[6] == N), the
    // +-N neighbours sit at the same position of the previous / next line (the far slots), and nothing couples rows
    // +-NX apart inside a "plane" -- NX is only the width of the halo the kernels load around a band, 2 rows: the
    // band's left and right neighbour.
    const bool two_d = cn[7] && cn[5] == cn[6] && cn[0] == cn[1] && cn[6] > 2;
    const int64_t NX = two_d ? 2 : cn[5], PL = cn[6];
    // Planes of whole 512-row chunks: a band's sub-bands ARE chunks and the pattern ids come from the chunk's
    // run-length record.  Any other even plane size (200 x 200, 300 x 300, ...; round 3): "gen mode" -- byte ids, a
    // partial last band per plane, and the walk must cover the whole matrix (no companion launch: its unit is the
    // chunk, and chunks straddle planes there).  SCHWZ_SWEEP_GEN=0: whole-chunk planes only.
    const char *gen_env = std::getenv("SCHWZ_SWEEP_GEN");
    const bool gen_mode = PL % kPairRows != 0;
    const bool shape_ok = cn[7] && cn[0] == -PL && (two_d || cn[1] == -NX) && NX >= 2 && PL > NX && NX % 2 == 0 && PL % 2 == 0 &&
                          NX <= 1024 && (!gen_mode || (!(gen_env && gen_env[0] == '0') && nrows % PL == 0 && PL >= kPairRows));
    if (sw_mode == 0) return no_walk("switched off (SCHWZ_SPMV_SWEEP=0)");
    if (!cn[7]) return no_walk("no canonical stencil layout (the patterns do not share one set of offsets)");
    if (!shape_ok) return no_walk("offsets are not those of an x-y-z (or x-y) numbering with even line and plane sizes");
    if (!(nrows >= (int64_t(1) << 20) || sw_mode == 2)) return no_walk("below 2^20 rows (SCHWZ_SPMV_SWEEP=2 walks anyway)");
    if (nrows < 3 * PL || nrows % 2 || A->v.ncols != nrows) return no_walk("fewer than three planes, or not square");
    if (rle.empty()) return no_walk("chunks have no run-length records");
    const int nchunks = (int)((nrows + kPairRows - 1) / kPairRows);
    const int nplanes = (int)(nrows / PL), cpp = (int)(PL / kPairRows);
    // ---- per plane: the far offsets its rows use -------------------------------------------------
    auto in_plane = [&](schwz_idx off) { return off == 0 || off == 1 || off == -1 || off == NX || off == -NX; };
    std::vector<std::vector<schwz_idx>> pat_far((size_t)tb.npat);
    std::vector<uint8_t> pat_bad((size_t)tb.npat, 0);
    for (int q = 0; q < tb.npat; ++q) {
        if ((int)tb.len[(size_t)q] > 9) pat_bad[(size_t)q] = 1;
        for (int k = 0; k < (int)tb.len[(size_t)q]; ++k) {
            const schwz_idx off = tb.ent[(size_t)q * tb.lmax + k].off;
            if (in_plane(off)) continue;
            if (off % PL != 0) pat_bad[(size_t)q] = 1;  // a far entry must keep the in-plane position
            pat_far[(size_t)q].push_back(off);
        }
    }
    std::vector<uint8_t> plane_ok((size_t)nplanes, 1);
    std::vector<std::vector<schwz_idx>> plane_far((size_t)nplanes);  // sorted ascending
    std::vector<std::vector<int>> plane_pats((size_t)nplanes);
    for (int k = 0; k < nplanes; ++k) {
        std::vector<uint8_t> used((size_t)tb.npat, 0);
        if (gen_mode)  // the patterns of the plane's pairs, from the byte ids
            for (int64_t pr = (int64_t)k * PL / 2; pr < (int64_t)(k + 1) * PL / 2; ++pr) used[(size_t)pair_id[(size_t)pr]] = 1;
        for (int c = k * cpp; !gen_mode && c < (k + 1) * cpp && plane_ok[(size_t)k]; ++c) {
            const int R = A->v.pair_rle_runs;
            if (rle[(size_t)c * R] == 0xffffu) {  // ids must run-length code (scalar loads only)
                plane_ok[(size_t)k] = 0;
                break;
            }
            // the patterns of a chunk are the ids of its runs
            for (int r = 0; r < R; ++r) used[(size_t)(rle[(size_t)c * R + r] >> 8)] = 1;
        }
        if (!plane_ok[(size_t)k]) continue;
        std::vector<schwz_idx> far;
        for (int q = 0; q < tb.npat; ++q) {
            if (!used[(size_t)q]) continue;
            plane_pats[(size_t)k].push_back(q);
            if (pat_bad[(size_t)q]) plane_ok[(size_t)k] = 0;
            for (schwz_idx f : pat_far[(size_t)q]) far.push_back(f);
        }
        std::sort(far.begin(), far.end());
        far.erase(std::unique(far.begin(), far.end()), far.end());
        int nb = 0, na = 0;
        for (schwz_idx f : far) {
            const int64_t j = k + f / PL;
            if (j < 0 || j >= nplanes) plane_ok[(size_t)k] = 0;
            (f < 0 ? nb : na)++;
        }
        if (far.size() > 2 || nb > 2 || na > 2) plane_ok[(size_t)k] = 0;
        if (plane_ok[(size_t)k]) plane_far[(size_t)k] = far;
    }
    // ---- chains: planes linked by their far couplings (degree <= 2: paths) ------------------------
    std::vector<std::vector<int>> adj((size_t)nplanes);
    auto link = [&](int x, int y) {
        if (std::find(adj[(size_t)x].begin(), adj[(size_t)x].end(), y) == adj[(size_t)x].end()) adj[(size_t)x].push_back(y);
    };
    for (int k = 0; k < nplanes; ++k)
        for (schwz_idx f : plane_far[(size_t)k]) {
            link(k, (int)(k + f / PL));
            link((int)(k + f / PL), k);
        }
    for (int k = 0; k < nplanes; ++k)
        if (adj[(size_t)k].size() > 2) {  // a plane somebody else points at as a third neighbour: not a path
            plane_ok[(size_t)k] = 0;
            for (int j : adj[(size_t)k]) plane_ok[(size_t)j] = 0;
        }
    std::vector<int> chain_plane;   // concatenated chains, -1 between them and at both ends
    std::vector<int> pos_of((size_t)nplanes, -1);
    chain_plane.push_back(-1);
    std::vector<uint8_t> seen((size_t)nplanes, 0);
    for (int pass = 0; pass < 2; ++pass)   // paths from their ends first, then whatever is left (rings: cut anywhere)
        for (int k0 = 0; k0 < nplanes; ++k0) {
            if (seen[(size_t)k0] || adj[(size_t)k0].size() > 2) continue;
            if (pass == 0 && adj[(size_t)k0].size() != 1 && !adj[(size_t)k0].empty()) continue;
            int prev = -1, k = k0;
            while (k >= 0 && !seen[(size_t)k] && adj[(size_t)k].size() <= 2) {
                seen[(size_t)k] = 1;
                pos_of[(size_t)k] = (int)chain_plane.size();
                chain_plane.push_back(k);
                int next = -1;
                for (int j : adj[(size_t)k])
                    if (j != prev && !seen[(size_t)j]) next = j;
                prev = k;
                k = next;
            }
            chain_plane.push_back(-1);
        }
    const int npos = (int)chain_plane.size();
    for (int k = 0; k < 4; ++k) chain_plane.push_back(-1);  // the kernels look up to four positions ahead
    // ---- per chain position: which window each far slot reads; per pattern: its nine slots --------
    // far code: 2 bits per far slot (B0, B1, A0, A1): 0 none, 1 previous chain position, 2 next
    std::vector<int> chain_far((size_t)npos + 4, 0);
    std::vector<double> cval((size_t)tb.npat * 18, 0.0), sval((size_t)tb.npat * 10, 0.0);
    std::vector<int> cmsk((size_t)tb.npat, 0), smsk((size_t)tb.npat, 0);
    std::vector<int8_t> pat_slot_set((size_t)tb.npat, 0);
    std::vector<std::vector<int8_t>> pat_slots((size_t)tb.npat);
    bool sym_ok = ts != nullptr && ts->npat == tb.npat;
    for (int p = 0; p < npos; ++p) {
        const int k = chain_plane[(size_t)p];
        if (k < 0 || !plane_ok[(size_t)k]) continue;
        const std::vector<schwz_idx> &far = plane_far[(size_t)k];
        std::vector<schwz_idx> fb, fa;
        for (schwz_idx f : far) (f < 0 ? fb : fa).push_back(f);
        int code = 0;
        bool ok = true;
        auto src_of = [&](schwz_idx f) -> int {
            const int j = (int)(k + f / PL);
            if (pos_of[(size_t)j] == p - 1) return 1;
            if (pos_of[(size_t)j] == p + 1) return 2;
            ok = false;
            return 0;
        };
        for (size_t i = 0; i < fb.size(); ++i) code |= src_of(fb[i]) << (2 * (int)i);
        for (size_t i = 0; i < fa.size(); ++i) code |= src_of(fa[i]) << (4 + 2 * (int)i);
        // slots of every pattern of the plane; a pattern shared with another plane must get the same ones
        for (int q : plane_pats[(size_t)k]) {
            std::vector<int8_t> slots;
            for (int e = 0; e < (int)tb.len[(size_t)q] && ok; ++e) {
                const schwz_idx off = tb.ent[(size_t)q * tb.lmax + e].off;
                int slot = -1;
                if (off == -NX) slot = 2;
                else if (off == -1) slot = 3;
                else if (off == 0) slot = 4;
                else if (off == 1) slot = 5;
                else if (off == NX) slot = 6;
                else {
                    for (size_t i = 0; i < fb.size(); ++i)
                        if (fb[i] == off) slot = (int)i;
                    for (size_t i = 0; i < fa.size(); ++i)
                        if (fa[i] == off) slot = 7 + (int)i;
                }
                if (slot < 0) ok = false;
                slots.push_back((int8_t)slot);
            }
            if (!ok) break;
            if (pat_slot_set[(size_t)q] && pat_slots[(size_t)q] != slots) ok = false;
            if (!ok) break;
            pat_slot_set[(size_t)q] = 1;
            pat_slots[(size_t)q] = slots;
        }
        if (!ok) {
            plane_ok[(size_t)k] = 0;
            continue;
        }
        chain_far[(size_t)p] = code;
    }
    for (int q = 0; q < tb.npat; ++q) {
        if (!pat_slot_set[(size_t)q]) continue;
        for (int e = 0; e < (int)tb.len[(size_t)q]; ++e) {
            const PairEntryH &en = tb.ent[(size_t)q * tb.lmax + e];
            const int slot = pat_slots[(size_t)q][(size_t)e];
            std::memcpy(&cval[((size_t)q * 9 + slot) * 2], &en.va, 8);
            std::memcpy(&cval[((size_t)q * 9 + slot) * 2 + 1], &en.vb, 8);
            cmsk[(size_t)q] |= (en.flags & 1) << slot;
            cmsk[(size_t)q] |= ((en.flags >> 1) & 1) << (16 + slot);
        }
        if (sym_ok) {
            // upper-triangle twin: slots [0, +1, +NX, far after 0, far after 1] = slots 4 .. 8 of the full form
            for (int e = 0; e < (int)ts->len[(size_t)q]; ++e) {
                const PairEntryH &en = ts->ent[(size_t)q * ts->lmax + e];
                int slot = -1;
                for (int f = 0; f < (int)tb.len[(size_t)q]; ++f)
                    if (tb.ent[(size_t)q * tb.lmax + f].off == en.off) slot = pat_slots[(size_t)q][(size_t)f] - 4;
                if (slot < 0 || slot > 4) {
                    sym_ok = false;
                    break;
                }
                std::memcpy(&sval[((size_t)q * 5 + slot) * 2], &en.va, 8);
                std::memcpy(&sval[((size_t)q * 5 + slot) * 2 + 1], &en.vb, 8);
                smsk[(size_t)q] |= (en.flags & 1) << slot;
                smsk[(size_t)q] |= ((en.flags >> 1) & 1) << (16 + slot);
            }
        }
    }
    // ---- segments -----------------------------------------------------------------------------------
    const char *t_env = std::getenv("SCHWZ_SWEEP_T"), *l_env = std::getenv("SCHWZ_SWEEP_L");
    int T = t_env ? std::atoi(t_env) : ((NX >= 512 || (two_d && PL % 1024 == 0)) ? 1024 : 512);
    if (T != 512 && T != 1024) T = 512;
    if (PL % T && !gen_mode) T = 512;
    if (gen_mode && PL < T) T = 512;
    // dynamic LDS of the update / start walk: the ring (4 T own + 4 NX halo doubles) and the nine-slot tables of
    // every pattern; the launches raise the kernels' limit to 96 KiB.  Too many patterns for the tall band: the
    // short one; still too much: no walk (the chunk-by-chunk launches take the matrix).
    auto walk_lds = [&](int t) { return (size_t)(4 * t + 4 * NX) * sizeof(double) + (size_t)tb.npat * (9 * 16 + 4); };
    if (walk_lds(T) > (size_t)(96 << 10) && T == 1024 && PL % 512 == 0) T = 512;
    if (walk_lds(T) > (size_t)(96 << 10)) return no_walk("ring and pattern tables exceed 96 KiB of LDS");
    if (NX > T) return no_walk("x line longer than a band");  // the halo of a band is NX rows either side: a band holds at least one x line
    const int bands = (int)((PL + T - 1) / T);  // (gen mode: the last band of a plane is partial)
    const int grid = (int)((std::min<int64_t>(ntiles, kMaxGrid) + kXcds - 1) / kXcds) * kXcds;
    struct Run { int p0, p1; };
    std::vector<Run> runs;
    int64_t steps = 0;
    for (int p = 0; p < npos;) {
        const int k = chain_plane[(size_t)p];
        if (k < 0 || !plane_ok[(size_t)k]) {
            ++p;
            continue;
        }
        int e = p;
        while (e < npos && chain_plane[(size_t)e] >= 0 && plane_ok[(size_t)chain_plane[(size_t)e]]) ++e;
        if (e - p >= 2) {
            runs.push_back({p, e});
            steps += (int64_t)(e - p) * bands;
        }
        p = e;
    }
    if (runs.empty()) return no_walk("no chain of two or more walkable planes");
    std::vector<uint8_t> covered((size_t)nchunks, 0);
    std::vector<schwz_idx> gen;
    if (gen_mode) {
        // every plane must be walked: nothing can be left to the chunk-by-chunk companion launch
        int64_t walked = 0;
        for (const Run &r : runs) walked += r.p1 - r.p0;
        if (walked != nplanes) return no_walk("planes that are not whole chunks: some plane cannot be walked (and nothing can be left to the chunk launches)");
    } else {
        for (const Run &r : runs)
            for (int p = r.p0; p < r.p1; ++p)
                for (int c = 0; c < cpp; ++c) covered[(size_t)chain_plane[(size_t)p] * cpp + c] = 1;
        for (int c = 0; c < nchunks; ++c)
            if (!covered[(size_t)c]) gen.push_back(c);
    }
    // segment length: about three segments per CU (two for bands of 1024 rows, which keep twice the loads
    // in flight) in ONE round of workgroups (measured on MI355X, 256^3 and 512 x 512 x 64; tools/sweep_ab.sh)
    int cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    const int per_cu = T == 1024 ? 2 : 3;
    int L = l_env ? std::atoi(l_env) : (int)std::max<int64_t>(8, (steps + per_cu * cus - 1) / (per_cu * cus));
    if (L < 2) L = 16;
    auto count = [&](int len) {
        int64_t n = 0;
        for (const Run &r : runs) n += (int64_t)((r.p1 - r.p0 + len - 1) / len) * bands;
        return n;
    };
    // the companion launch walks its chunks with one gather round trip after the other: as many workgroups
    // as the partial-sum slots next to the segments allow (up to one per chunk)
    const int seg_slots = (int)((count(L) + kXcds - 1) / kXcds * kXcds) + kXcds;
    const int gen_blocks = (int)std::min<int64_t>((int64_t)gen.size(), std::max(256, std::min(1024, grid - seg_slots)));
    while (count(L) + kXcds > grid - gen_blocks && L < (1 << 20)) L += 4;
    struct Seg { int band, p0, p1; };
    // workgroup slots for bands of Tq rows and segments of about Lq chain positions
    auto make_slots = [&](int Tq, int Lq) -> std::vector<int4> {
        const int bands_q = (int)((PL + Tq - 1) / Tq);
        std::vector<Seg> segs;
        for (const Run &r : runs) {
            const int nseg = (r.p1 - r.p0 + Lq - 1) / Lq, len = (r.p1 - r.p0 + nseg - 1) / nseg;
            for (int b = 0; b < bands_q; ++b)
                for (int p = r.p0; p < r.p1; p += len) segs.push_back({b, p, std::min(p + len, r.p1)});
        }
        // deal: XCD x takes the bands [x * bands / 8, (x + 1) * bands / 8) (a band's window shares its NX-row
        // halos with the neighbouring bands: the same L2), segment by segment along the chain
        std::stable_sort(segs.begin(), segs.end(), [](const Seg &x, const Seg &y) { return x.p0 != y.p0 ? x.p0 < y.p0 : x.band < y.band; });
        std::vector<std::vector<int4>> per_xcd(kXcds);
        for (const Seg &sgm : segs) {
            int4 v;
            v.x = sgm.band;
            v.y = sgm.p0;
            v.z = sgm.p1;
            v.w = (int)std::min<int64_t>(Tq, PL - (int64_t)sgm.band * Tq);  // rows of the band inside the plane
            size_t x;
            if (bands_q >= 2 * kXcds) {
                x = (size_t)((int64_t)sgm.band * kXcds / bands_q);
            } else {  // few bands: round robin
                x = 0;
                for (size_t k = 1; k < (size_t)kXcds; ++k)
                    if (per_xcd[k].size() < per_xcd[x].size()) x = k;
            }
            per_xcd[x].push_back(v);
        }
        size_t depth = 0;
        for (const auto &l : per_xcd) depth = std::max(depth, l.size());
        std::vector<int4> out(depth * kXcds);
        for (size_t q = 0; q < depth; ++q)
            for (int x = 0; x < kXcds; ++x) {
                int4 v;
                v.x = v.y = v.z = v.w = 0;
                if (q < per_xcd[(size_t)x].size()) v = per_xcd[(size_t)x][q];
                out[q * kXcds + x] = v;
            }
        return out;
    };
    const std::vector<int4> slots_v = make_slots(T, L);
    // The fused direction launch may take taller bands than the update launch (SCHWZ_SWEEP_TDIR=512|1024|2048):
    // its window carries an NX-row halo of r AND p per band, so a band of twice the rows halves that share,
    // while the update launch keeps four halo lines per window and prefers the shorter band.  A table of its
    // own; equal to the update launch's when the band heights coincide.
    // Measured in-box (tools/tdir_ab.sh): 256-wide planes, update bands of 512 rows: 1024-row bands for the fused
    // launch -3 % per step (2048: +5 %); 512-wide planes, 1024 / 2048: +3 % (two workgroups per CU); 1024-wide
    // planes, where a 1024-row band is a single x line, 2048: fused launch 0.773 -> 0.696 ms, -4 % per step -- in
    // round 2.  With the halo schedule of round 3 (SCHWZ_DD) the halo lines hit L2 and what counts on 1024-wide
    // planes is the second workgroup per CU a 1024-row band leaves room for: 1024 x 1024 x 128 slab, fused launch
    // 0.681 ms with 2048-row bands, 0.640 ms with 1024 (step 16.5 -> 16.0 ms; profiles/r03_c5slab_ab.txt).
    const char *td_env = std::getenv("SCHWZ_SWEEP_TDIR");
    int T_dir = td_env ? std::atoi(td_env) : (T == 512 ? 1024 : T);
    if ((T_dir != 512 && T_dir != 1024 && T_dir != 2048) || (PL % T_dir && !gen_mode) || (gen_mode && PL < T_dir) ||
        (size_t)(4 * T_dir + ((SCHWZ_DD & 2) ? 3 : 2) * NX) * sizeof(double) + (size_t)tb.npat * 84 > (size_t)(96 << 10))
        T_dir = T;
    std::vector<int4> slots_dir;
    if (T_dir != T) {
        const int bands_d = (int)((PL + T_dir - 1) / T_dir);
        int64_t steps_d = 0;
        for (const Run &r : runs) steps_d += (int64_t)(r.p1 - r.p0) * bands_d;
        const int per_cu_d = T_dir >= 2048 ? 1 : (T_dir == 1024 ? 2 : 3);
        const char *ld_env = std::getenv("SCHWZ_SWEEP_LDIR");
        int Ld = ld_env ? std::atoi(ld_env) : (int)std::max<int64_t>(8, (steps_d + per_cu_d * cus - 1) / (per_cu_d * cus));
        if (Ld < 2) Ld = 16;
        auto count_d = [&](int len) {
            int64_t nq = 0;
            for (const Run &r : runs) nq += (int64_t)((r.p1 - r.p0 + len - 1) / len) * bands_d;
            return nq;
        };
        while (count_d(Ld) + kXcds > grid - gen_blocks && Ld < (1 << 20)) Ld += 4;
        slots_dir = make_slots(T_dir, Ld);
        if ((int64_t)slots_dir.size() + gen_blocks > grid) {
            slots_dir.clear();
            T_dir = T;
        }
    }
    // Rows the walk leaves out cost a companion launch per CG launch: measured with 256 x 256 planes, 8 / 4 / 1
    // slabs on one GPU when the boundary planes of a slab were still left out (tools/sweep_sizes.sh, bench.py
    // --ttr-subdomains): +13 % time at 2.2 M rows, +2 % at 4.3 M, -18 % at 16.8 M; without left-out rows the
    // walk wins from 1 M rows on.
    const bool worth = gen.empty() || nrows >= 6000000 || sw_mode == 2;
    if (!worth || (int64_t)slots_v.size() + gen_blocks > grid || steps * T * 2 < nrows)
        return no_walk("rows left to the companion launch on a small matrix, more segments than partial-sum slots, or less than half of the rows walkable");
    // A table of its own for the first-direction launch of a solve (round 3).  That launch reads ONE vector and does
    // little per row: its time is the latency of a workgroup's steps times the bytes it keeps in flight, and the
    // fused launch's table gives it two workgroups per CU.  Bands of the update launch's height (512 rows where the
    // plane allows) and about SCHWZ_SWEEP_FIRSTPERCU (6; 0: the fused launch's table) workgroups per CU.
    std::vector<int4> slots_first;
    int T_first = 0;
    {
        const char *fe = std::getenv("SCHWZ_SWEEP_FIRSTPERCU");
        const int per_cu_f = fe ? std::atoi(fe) : 6;
        const int Tf = T;  // (a height both walks have instantiations for)
        if (per_cu_f > 0 && Tf <= (slots_dir.empty() ? T : T_dir)) {
            const int bands_f = (int)((PL + Tf - 1) / Tf);
            int64_t steps_f = 0;
            for (const Run &r : runs) steps_f += (int64_t)(r.p1 - r.p0) * bands_f;
            int Lf = (int)std::max<int64_t>(6, (steps_f + (int64_t)per_cu_f * cus - 1) / ((int64_t)per_cu_f * cus));
            auto count_f = [&](int len) {
                int64_t nq = 0;
                for (const Run &r : runs) nq += (int64_t)((r.p1 - r.p0 + len - 1) / len) * bands_f;
                return nq;
            };
            while (count_f(Lf) + kXcds > grid - gen_blocks && Lf < (1 << 20)) Lf += 2;
            slots_first = make_slots(Tf, Lf);
            if ((int64_t)slots_first.size() + gen_blocks > grid) slots_first.clear();
            T_first = Tf;
        }
    }
    int rc;
    if (!slots_dir.empty() && (rc = upv(slots_dir, &A->d_sweep_seg_dir))) return rc;
    if (!slots_first.empty() && (rc = upv(slots_first, &A->d_sweep_seg_first))) return rc;
    if ((rc = upv(slots_v, &A->d_sweep_seg)) || (rc = upv(gen, &A->d_sweep_gen)) || (rc = upv(cval, &A->d_canon_val)) ||
        (rc = upv(cmsk, &A->d_canon_mask)) || (rc = upv(chain_plane, &A->d_chain_plane)) ||
        (rc = upv(chain_far, &A->d_chain_far)))
        return rc;
    A->v.canon_val = (const double *)A->d_canon_val;
    A->v.canon_mask = (const int *)A->d_canon_mask;
    A->v.canon_npat = tb.npat;
    A->v.chain_plane = (const int *)A->d_chain_plane;
    A->h_chain_plane = chain_plane;
    A->v.chain_far = (const int *)A->d_chain_far;
    if (sym_ok) {
        if ((rc = upv(sval, &A->d_canon_sym_val)) || (rc = upv(smsk, &A->d_canon_sym_mask))) return rc;
        A->v.canon_sym_val = (const double *)A->d_canon_sym_val;
        A->v.canon_sym_mask = (const int *)A->d_canon_sym_mask;
    }
    A->v.sweep_gen_mode = gen_mode ? 1 : 0;
    A->v.sweep_T = T;
    A->v.sweep_nx = (int)NX;
    A->v.sweep_pl = PL;
    A->v.sweep_nslots = (int)slots_v.size();
    A->v.sweep_ngen = (int)gen.size();
    A->v.sweep_gen_blocks = gen_blocks;
    A->v.sweep_seg = (const int4 *)A->d_sweep_seg;
    A->v.sweep_gen = (const schwz_idx *)A->d_sweep_gen;
    A->v.sweep_T_dir = slots_dir.empty() ? T : T_dir;
    A->v.sweep_nslots_dir = slots_dir.empty() ? (int)slots_v.size() : (int)slots_dir.size();
    A->v.sweep_seg_dir = slots_dir.empty() ? A->v.sweep_seg : (const int4 *)A->d_sweep_seg_dir;
    A->v.sweep_T_first = slots_first.empty() ? A->v.sweep_T_dir : T_first;
    A->v.sweep_nslots_first = slots_first.empty() ? A->v.sweep_nslots_dir : (int)slots_first.size();
    A->v.sweep_seg_first = slots_first.empty() ? A->v.sweep_seg_dir : (const int4 *)A->d_sweep_seg_first;
    return SCHWZ_OK;
}

// Leaves A->v.pair_id null when fewer than 90 % of the nonzeros sit in pair-coded chunks
// (SCHWZ_SPMV_PAIR=0 disables, =2 forces whatever the coverage).
int build_spmv_pair(schwz_csr *A, const schwz_idx *rp, const schwz_idx *col, const double *val,
                    const std::vector<schwz_idx> &tiles)
{
    const char *env = std::getenv("SCHWZ_SPMV_PAIR");
    if (env && env[0] == '0') return SCHWZ_OK;
    if (tiles.size() < 2) return SCHWZ_OK;
    const int64_t nrows = tiles.back(), nnz = rp[nrows];
    if (nnz == 0 || A->v.ncols < 2 || A->v.ncols >= INT32_MAX || nrows >= INT32_MAX - kPairRows) return SCHWZ_OK;
    const int nchunks = (int)((nrows + kPairRows - 1) / kPairRows);
    std::vector<uint8_t> pair_id((size_t)(nrows + 1) / 2, 0);
    std::vector<schwz_idx> chunk_ptable((size_t)nchunks, -1);
    std::vector<PairTable> tables;
    std::unordered_multimap<uint64_t, int> by_hash;
    std::vector<std::vector<PairEntryH>> pats;
    std::vector<PairEntryH> cur;
    int64_t coded = 0;
    // merged (offset, values, presence) sequence of the pair starting at row ra
    auto merge_pair = [&](int64_t ra, bool has_b) {
        cur.clear();
        schwz_idx ja = rp[ra], ea = rp[ra + 1];
        schwz_idx jb = has_b ? rp[ra + 1] : 0, eb = has_b ? rp[ra + 2] : 0;
        while (ja < ea || jb < eb) {
            const int64_t da = ja < ea ? (int64_t)col[ja] - ra : INT64_MAX;
            const int64_t db = jb < eb ? (int64_t)col[jb] - (ra + 1) : INT64_MAX;
            PairEntryH e = {0, 0, 0, 0};
            const int64_t d = std::min(da, db);
            e.off = (schwz_idx)d;
            if (da == d) {
                e.flags |= 1;
                std::memcpy(&e.va, &val[ja], 8);
                ++ja;
            }
            if (db == d) {
                e.flags |= 2;
                std::memcpy(&e.vb, &val[jb], 8);
                ++jb;
            }
            cur.push_back(e);
        }
    };
    // First choice: ONE table for the whole matrix (a constant-coefficient stencil has a few dozen
    // distinct pairs in total); the kernel then stages it once and looks nothing up per chunk.
    bool single = !(env && env[0] == '3');  // SCHWZ_SPMV_PAIR=3: per-chunk tables even if one would do
    {
        StageTimer t_single("  pairs: one table for the whole matrix");
        std::vector<char> unsorted(64, 0);
        parallel_blocks(nrows, 1 << 16, [&](int t, int, int64_t a, int64_t b) {
            bool bad = false;
            for (int64_t r = a; r < b && !bad; ++r) {
                if (rp[r + 1] - rp[r] > 127) bad = true;
                for (schwz_idx j = rp[r] + 1; j < rp[r + 1] && !bad; ++j)
                    if (col[j] <= col[j - 1]) bad = true;
            }
            unsorted[(size_t)t] = bad;
        });
        bool sorted = true;
        for (char u : unsorted) sorted = sorted && !u;
        single = single && sorted;
        int lmax = 1;
        if (single) {
            // Every thread codes a contiguous block of pairs against a dictionary of its own (ids in ITS order of
            // first appearance); the dictionaries are then merged in block order, which numbers the patterns in
            // the order a sequential pass meets them, and the ids are renumbered.
            const int64_t npairs = (nrows + 1) / 2;
            std::vector<std::vector<std::vector<PairEntryH>>> tpats(64);
            std::vector<char> tfail(64, 0);
            const int nthreads = parallel_blocks(npairs, 1 << 15, [&](int t, int, int64_t a, int64_t b) {
                auto &mine = tpats[(size_t)t];
                std::unordered_multimap<uint64_t, int> seen;
                std::vector<PairEntryH> cur_t;
                for (int64_t pi = a; pi < b && !tfail[(size_t)t]; ++pi) {
                    const int64_t ra = 2 * pi;
                    // (merge_pair on a thread-private sequence)
                    cur_t.clear();
                    {
                        const bool has_b = ra + 1 < nrows;
                        schwz_idx ja = rp[ra], ea = rp[ra + 1];
                        schwz_idx jb = has_b ? rp[ra + 1] : 0, eb = has_b ? rp[ra + 2] : 0;
                        while (ja < ea || jb < eb) {
                            const int64_t da = ja < ea ? (int64_t)col[ja] - ra : INT64_MAX;
                            const int64_t db = jb < eb ? (int64_t)col[jb] - (ra + 1) : INT64_MAX;
                            PairEntryH e = {0, 0, 0, 0};
                            const int64_t d = std::min(da, db);
                            e.off = (schwz_idx)d;
                            if (da == d) {
                                e.flags |= 1;
                                std::memcpy(&e.va, &val[ja], 8);
                                ++ja;
                            }
                            if (db == d) {
                                e.flags |= 2;
                                std::memcpy(&e.vb, &val[jb], 8);
                                ++jb;
                            }
                            cur_t.push_back(e);
                        }
                    }
                    uint64_t h = 1469598103934665603ull;
                    for (const PairEntryH &e : cur_t) {
                        h = (h ^ e.va) * 1099511628211ull;
                        h = (h ^ e.vb) * 1099511628211ull;
                        h = (h ^ (uint64_t)(int64_t)e.off) * 1099511628211ull;
                        h = (h ^ (uint64_t)e.flags) * 1099511628211ull;
                    }
                    int id = -1;
                    auto range = seen.equal_range(h);
                    for (auto it = range.first; it != range.second; ++it)
                        if (mine[(size_t)it->second] == cur_t) {
                            id = it->second;
                            break;
                        }
                    if (id < 0) {
                        id = (int)mine.size();
                        if (id == kPairPats) {
                            tfail[(size_t)t] = 1;
                            break;
                        }
                        seen.emplace(h, id);
                        mine.push_back(cur_t);
                    }
                    pair_id[(size_t)pi] = (uint8_t)id;
                }
            });
            std::vector<std::vector<int>> remap((size_t)nthreads);
            for (int t = 0; t < nthreads && single; ++t) {
                if (tfail[(size_t)t]) single = false;
                for (const auto &pt : tpats[(size_t)t]) {
                    if (!single) break;
                    int id = -1;
                    for (size_t q = 0; q < pats.size(); ++q)
                        if (pats[q] == pt) {
                            id = (int)q;
                            break;
                        }
                    if (id < 0) {
                        id = (int)pats.size();
                        lmax = std::max(lmax, (int)pt.size());
                        if (id == kPairPats || (int64_t)(id + 1) * pair_stride(lmax) > kPairEntries) {
                            single = false;
                            break;
                        }
                        pats.push_back(pt);
                    }
                    remap[(size_t)t].push_back(id);
                }
            }
            if (single) {
                // the same blocks again (parallel_blocks cuts [0, npairs) the same way for the same n and grain)
                parallel_blocks(npairs, 1 << 15, [&](int t, int, int64_t a, int64_t b) {
                    const auto &mp = remap[(size_t)t];
                    for (int64_t pi = a; pi < b; ++pi) pair_id[(size_t)pi] = (uint8_t)mp[(size_t)pair_id[(size_t)pi]];
                });
            } else {
                pats.clear();
            }
        }
        if (single) {
            PairTable tb;
            tb.npat = (int)pats.size();
            tb.lmax = lmax;
            tb.len.resize((size_t)tb.npat);
            tb.ent.assign((size_t)tb.npat * lmax, PairEntryH{0, 0, 0, 0});
            for (int q = 0; q < tb.npat; ++q) {
                tb.len[(size_t)q] = (uint8_t)pats[(size_t)q].size();
                for (size_t k = 0; k < pats[(size_t)q].size(); ++k) tb.ent[(size_t)q * lmax + k] = pats[(size_t)q][k];
            }
            tables.push_back(std::move(tb));
            std::fill(chunk_ptable.begin(), chunk_ptable.end(), 0);
            coded = nnz;
        }
    }
    for (int c = 0; c < nchunks && !single; ++c) {
        const int64_t r0 = (int64_t)c * kPairRows, r1 = std::min<int64_t>(r0 + kPairRows, nrows);
        if (rp[r1] == rp[r0]) continue;
        bool ok = true;
        for (int64_t r = r0; r < r1 && ok; ++r) {
            if (rp[r + 1] - rp[r] > 127) ok = false;
            for (schwz_idx j = rp[r] + 1; j < rp[r + 1] && ok; ++j)
                if (col[j] <= col[j - 1]) ok = false;  // merged order == each row's order needs sorted rows
        }
        if (!ok) continue;
        pats.clear();
        int lmax = 0;
        for (int64_t ra = r0; ra < r1 && ok; ra += 2) {
            merge_pair(ra, ra + 1 < r1);
            int id = -1;
            for (size_t q = 0; q < pats.size(); ++q)
                if (pats[q] == cur) {
                    id = (int)q;
                    break;
                }
            if (id < 0) {
                if ((int)pats.size() == kPairPats) {
                    ok = false;
                    break;
                }
                id = (int)pats.size();
                lmax = std::max(lmax, (int)cur.size());
                pats.push_back(cur);
            }
            pair_id[(size_t)(ra >> 1)] = (uint8_t)id;
        }
        lmax = std::max(lmax, 1);
        if (!ok || (int64_t)pats.size() * pair_stride(lmax) > kPairEntries) continue;
        PairTable tb;
        tb.npat = (int)pats.size();
        tb.lmax = lmax;
        tb.len.resize((size_t)tb.npat);
        tb.ent.assign((size_t)tb.npat * lmax, PairEntryH{0, 0, 0, 0});
        uint64_t h = 1469598103934665603ull;
        for (int q = 0; q < tb.npat; ++q) {
            tb.len[(size_t)q] = (uint8_t)pats[(size_t)q].size();
            for (size_t k = 0; k < pats[(size_t)q].size(); ++k) {
                const PairEntryH &e = pats[(size_t)q][k];
                tb.ent[(size_t)q * lmax + k] = e;
                h = (h ^ e.va) * 1099511628211ull;
                h = (h ^ e.vb) * 1099511628211ull;
                h = (h ^ (uint64_t)(int64_t)e.off) * 1099511628211ull;
                h = (h ^ (uint64_t)e.flags) * 1099511628211ull;
            }
            h = (h ^ 0xffull ^ (uint64_t)tb.len[(size_t)q]) * 1099511628211ull;
        }
        tb.hash = h;
        int id = -1;
        auto range = by_hash.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (tables[(size_t)it->second].same(tb)) {
                id = it->second;
                break;
            }
        if (id < 0) {
            id = (int)tables.size();
            by_hash.emplace(h, id);
            tables.push_back(std::move(tb));
        }
        chunk_ptable[(size_t)c] = id;
        coded += rp[r1] - rp[r0];
    }
    A->pair_fraction = (double)coded / (double)nnz;
    const bool force = env && env[0] == '2';
    if (A->pair_fraction < 0.9 && !force) return SCHWZ_OK;
    if (!single && tables.size() * 4 > (size_t)nchunks && !force) return SCHWZ_OK;  // tables must be shared to pay off
    // Symmetric matrix (checked bit for bit): a second set of tables with the entries on and above the
    // diagonal only, the strictly upper ones doubled (exact), for kSpmvDotSym.  SCHWZ_SPMV_SYM=0 skips it.
    StageTimer t_sym("  pairs: symmetry check, upper-triangle twins");
    int sym_base = 0;
    const char *sym_env = std::getenv("SCHWZ_SPMV_SYM");
    if (!(sym_env && sym_env[0] == '0') && csr_is_symmetric(nrows, A->v.ncols, rp, col, val)) {
        sym_base = (int)tables.size();
        std::vector<PairTable> upper((size_t)sym_base);
        for (int t = 0; t < sym_base; ++t) {
            const PairTable &src = tables[(size_t)t];
            PairTable &u = upper[(size_t)t];
            u.npat = src.npat;
            u.len.assign((size_t)u.npat, 0);
            u.lmax = 1;
            for (int q = 0; q < src.npat; ++q) {
                int cnt = 0;
                for (int k = 0; k < (int)src.len[(size_t)q]; ++k) cnt += src.ent[(size_t)q * src.lmax + k].off >= 0;
                u.len[(size_t)q] = (uint8_t)cnt;
                u.lmax = std::max(u.lmax, cnt);
            }
            u.ent.assign((size_t)u.npat * u.lmax, PairEntryH{0, 0, 0, 0});
            for (int q = 0; q < src.npat; ++q) {
                int w = 0;
                for (int k = 0; k < (int)src.len[(size_t)q]; ++k) {
                    PairEntryH e = src.ent[(size_t)q * src.lmax + k];
                    if (e.off < 0) continue;
                    if (e.off > 0) {
                        double va, vb;
                        std::memcpy(&va, &e.va, 8);
                        std::memcpy(&vb, &e.vb, 8);
                        va *= 2.0;
                        vb *= 2.0;
                        std::memcpy(&e.va, &va, 8);
                        std::memcpy(&e.vb, &vb, 8);
                    }
                    u.ent[(size_t)q * u.lmax + w++] = e;
                }
            }
        }
        for (PairTable &u : upper) tables.push_back(std::move(u));
    }
    t_sym.stop();
    StageTimer t_rest("  pairs: tables, records, canonical layout, walk tables, uploads");
    std::vector<schwz_idx> desc, meta;
    std::vector<uint8_t> lens;
    std::vector<double> vals;
    for (const PairTable &tb : tables) {
        desc.push_back((schwz_idx)(vals.size() / 2));
        desc.push_back((schwz_idx)lens.size());
        desc.push_back(tb.npat);
        desc.push_back(tb.lmax);
        schwz_idx reach = 0;
        for (const PairEntryH &e : tb.ent) reach = std::max<schwz_idx>(reach, e.off < 0 ? -e.off : e.off);
        desc.push_back(reach);
        lens.insert(lens.end(), tb.len.begin(), tb.len.end());
        for (const PairEntryH &e : tb.ent) {
            double va, vb;
            std::memcpy(&va, &e.va, 8);
            std::memcpy(&vb, &e.vb, 8);
            vals.push_back(va);
            vals.push_back(vb);
            meta.push_back(e.off);
            meta.push_back(e.flags);
        }
    }
    int rc;
    if ((rc = upv(pair_id, &A->d_pair_id)) || (rc = upv(chunk_ptable, &A->d_tile_ptable)) ||
        (rc = upv(desc, &A->d_ptbl_desc)) || (rc = upv(lens, &A->d_ptbl_len)) || (rc = upv(vals, &A->d_ptbl_val)) ||
        (rc = upv(meta, &A->d_ptbl_meta)))
        return rc;
    A->v.pair_id = (const uint8_t *)A->d_pair_id;
    // run-length form of the ids, chunk by chunk (SCHWZ_SPMV_RLE=0: byte ids only)
    // Records of 8 runs (16 bytes per chunk) serve x lines of ~170 entries and more; a matrix some of whose
    // chunks need up to 16 runs (a 512-row chunk of a 192 x 192 plane crosses three line ends: ten runs) gets
    // records of 16 runs (32 bytes per chunk) throughout -- SCHWZ_SPMV_RLE=8 keeps the short records.
    const char *rle_env = std::getenv("SCHWZ_SPMV_RLE");
    std::vector<uint16_t> rle;
    int rle_runs = 8;
    if (!(rle_env && rle_env[0] == '0')) {
        auto build_rle = [&](int R, int64_t *coded_out) {
            std::vector<uint16_t> out((size_t)nchunks * R, 0xffffu);
            std::vector<int64_t> coded_t(64, 0);
            parallel_blocks(nchunks, 2048, [&](int t, int, int64_t c_begin, int64_t c_end) {
                int64_t mine = 0;
                for (int64_t c = c_begin; c < c_end; ++c) {
                    if (chunk_ptable[(size_t)c] < 0) continue;
                    const int64_t p0 = c * (kPairRows / 2), p1 = std::min<int64_t>(p0 + kPairRows / 2, (nrows + 1) / 2);
                    uint16_t runs[16];
                    int nr = 0;
                    bool fits = true;
                    for (int64_t p = p0; p < p1 && fits; ++p) {
                        if (nr == 0 || pair_id[(size_t)p] != (uint8_t)(runs[nr - 1] >> 8)) {
                            if (nr == R) {
                                fits = false;
                                break;
                            }
                            runs[nr++] = (uint16_t)((p - p0) | ((int)pair_id[(size_t)p] << 8));
                        }
                    }
                    if (!fits || nr == 0) continue;
                    for (int k = nr; k < R; ++k) runs[k] = runs[nr - 1];
                    if (runs[0] == 0xffffu) continue;  // would read as the "not coded" marker
                    std::copy(runs, runs + R, out.begin() + (size_t)c * R);
                    ++mine;
                }
                coded_t[(size_t)t] = mine;
            });
            int64_t coded = 0;
            for (int64_t v : coded_t) coded += v;
            *coded_out = coded;
            return out;
        };
        int64_t coded8 = 0, coded16 = 0;
        rle = build_rle(8, &coded8);
        if (!(rle_env && rle_env[0] == '8')) {
            std::vector<uint16_t> wide = build_rle(16, &coded16);
            if (coded16 > coded8) {
                rle.swap(wide);
                rle_runs = 16;
            }
        }
        if ((rc = upv(rle, &A->d_pair_rle))) return rc;
        A->v.pair_rle = (const uint4 *)A->d_pair_rle;
        A->v.pair_rle_runs = rle_runs;
    }
    A->v.chunk_ptable = (const schwz_idx *)A->d_tile_ptable;
    A->v.ptbl_desc = (const schwz_idx *)A->d_ptbl_desc;
    A->v.ptbl_len = (const uint8_t *)A->d_ptbl_len;
    A->v.ptbl_val = (const double *)A->d_ptbl_val;
    A->v.ptbl_meta = (const schwz_idx *)A->d_ptbl_meta;
    A->v.pair_single = single ? 1 : 0;
    // canonical stencil layout of a single-table matrix (SCHWZ_SPMV_CANON=0: off): the offsets of its
    // commonest pattern when they read {<= 2 below -1, -1, 0, +1, <= 2 above +1}; missing outer slots
    // repeat their neighbour outwards, so an entry always lands in the lowest slot with its offset and
    // the slots stay in ascending entry order
    for (int k = 0; k < 8; ++k) A->v.pair_canon[k] = 0;
    const char *canon_env = std::getenv("SCHWZ_SPMV_CANON");
    if (single && !(canon_env && canon_env[0] == '0')) {
        const PairTable &tb = tables[0];
        std::vector<int64_t> freq((size_t)tb.npat, 0);
        for (uint8_t id : pair_id) ++freq[(size_t)id];
        const int best = (int)(std::max_element(freq.begin(), freq.end()) - freq.begin());
        std::vector<schwz_idx> neg, pos;
        bool has_m1 = false, has_0 = false, has_p1 = false, ok = true;
        for (int k = 0; k < (int)tb.len[(size_t)best]; ++k) {
            const schwz_idx off = tb.ent[(size_t)best * tb.lmax + k].off;
            if (off == -1) has_m1 = true;
            else if (off == 0) has_0 = true;
            else if (off == 1) has_p1 = true;
            else if (off < 0) neg.push_back(off);
            else pos.push_back(off);
        }
        ok = has_m1 && has_0 && has_p1 && neg.size() <= 2 && pos.size() <= 2;
        if (ok) {
            std::sort(neg.begin(), neg.end());  // most negative first
            std::sort(pos.begin(), pos.end());
            const schwz_idx n2 = neg.size() >= 1 ? neg[0] : -1;
            const schwz_idx n1 = neg.size() == 2 ? neg[1] : n2;
            const schwz_idx p2 = pos.size() >= 1 ? pos.back() : 1;
            const schwz_idx p1 = pos.size() == 2 ? pos[0] : p2;
            const schwz_idx lay[7] = {n2, n1, -1, 0, 1, p1, p2};
            for (int k = 0; k < 7; ++k) A->v.pair_canon[k] = lay[k];
            A->v.pair_canon[7] = 1;
        }
    }
    A->v.pair_sym_base = sym_base;
    // z-sweep segments (CsrView::sweep_*): SCHWZ_SPMV_SWEEP=0 off, =2 also below a million rows (tests);
    // SCHWZ_SWEEP_T (512 / 1024 rows per band), SCHWZ_SWEEP_L (planes per segment) override the defaults.
    if (single && (rc = build_sweep(A, nrows, tables[0], sym_base > 0 ? &tables[(size_t)sym_base] : nullptr, pair_id, rle,
                                    (int64_t)tiles.size() - 1)))
        return rc;
    {
        // what a pass over the coded matrix reads: per chunk its 16-byte run-length record, or one byte per
        // pair where the ids do not run-length code; the chunk's table id unless one table serves the whole
        // matrix; the tables themselves (one set; the upper-triangle twins are read INSTEAD by kSpmvDotSym)
        int64_t bytes = 0;
        const uint16_t *rle_h = rle.empty() ? nullptr : rle.data();
        for (int c = 0; c < nchunks; ++c) {
            if (chunk_ptable[(size_t)c] < 0) continue;
            const bool runs = rle_h && rle_h[(size_t)c * rle_runs] != 0xffffu;
            bytes += runs ? 2 * rle_runs : (rle_h ? 2 * rle_runs : 0) + kPairRows / 2;
            if (!single) bytes += 4;
        }
        const size_t ntab = sym_base ? (size_t)sym_base : tables.size();
        for (size_t t = 0; t < ntab; ++t) bytes += (int64_t)tables[t].ent.size() * 24 + tables[t].npat;
        A->pair_code_bytes = bytes;
    }
    // the XCD deal of the chunks: the tile deal's run length in rows, in chunks (a power of two)
    int sh = A->pair_deal_shift;
    const int64_t rows_per_tile = std::max<int64_t>(1, nrows / std::max<int64_t>(1, (int64_t)tiles.size() - 1));
    for (int64_t f = kPairRows / std::max<int64_t>(1, rows_per_tile); f > 1 && sh > 0; f >>= 1) --sh;
    A->v.pair_shift = sh;
    return SCHWZ_OK;
}

// chunks whose rows or columns reach `split` need the second product of the fused dual residual
int pair_set_dual_split(schwz_csr *A, const schwz_idx *h_rp, const schwz_idx *h_col, int64_t split)
{
    if (!A->v.pair_id) return SCHWZ_OK;
    const int64_t nrows = A->v.nrows;
    const int nchunks = (int)((nrows + kPairRows - 1) / kPairRows);
    std::vector<uint8_t> flag((size_t)nchunks, 0);
    for (int c = 0; c < nchunks; ++c) {
        const int64_t r0 = (int64_t)c * kPairRows, r1 = std::min<int64_t>(r0 + kPairRows, nrows);
        bool f = r1 > split;
        for (int64_t j = h_rp[r0]; j < h_rp[r1] && !f; ++j) f = h_col[j] >= split;
        flag[(size_t)c] = f ? 1 : 0;
    }
    (void)hipFree(A->d_chunk_dual);
    A->d_chunk_dual = nullptr;
    int rc = upv(flag, &A->d_chunk_dual);
    if (rc) return rc;
    A->v.chunk_dual = (const uint8_t *)A->d_chunk_dual;
    // the same for the z-sweep walk: chain positions whose plane has a flagged chunk, and the list of those
    // planes' chunks for the listed kSpmvResidNorm launch (launch_spmv_pair, dual start in the walk)
    (void)hipFree(A->d_chain_dual);
    (void)hipFree(A->d_dual_chunks);
    A->d_chain_dual = A->d_dual_chunks = nullptr;
    A->v.chain_dual = nullptr;
    A->v.dual_chunks = nullptr;
    A->v.dual_nchunks = A->v.dual_blocks = 0;
    if (A->v.sweep_nslots > 0 && !A->h_chain_plane.empty() && A->v.sweep_pl > 0 && !A->v.sweep_gen_mode) {
        const int cpp = (int)(A->v.sweep_pl / kPairRows);
        std::vector<int> cdual(A->h_chain_plane.size(), 0);
        std::vector<schwz_idx> list;
        for (size_t p = 0; p < A->h_chain_plane.size(); ++p) {
            const int k = A->h_chain_plane[p];
            if (k < 0) continue;
            bool f = false;
            for (int c = k * cpp; c < (k + 1) * cpp && c < nchunks; ++c) f = f || flag[(size_t)c];
            if (!f) continue;
            cdual[p] = 1;
            for (int c = k * cpp; c < (k + 1) * cpp && c < nchunks; ++c) list.push_back(c);
        }
        if (!list.empty()) {
            if ((rc = upv(cdual, &A->d_chain_dual)) || (rc = upv(list, &A->d_dual_chunks))) return rc;
            A->v.chain_dual = (const int *)A->d_chain_dual;
            A->v.dual_chunks = (const schwz_idx *)A->d_dual_chunks;
            A->v.dual_nchunks = (int)list.size();
            A->v.dual_blocks = (int)std::min<size_t>(list.size(), 512);
        }
    }
    return SCHWZ_OK;
}

void free_spmv_pair(schwz_csr *A)
{
    (void)hipFree(A->d_sweep_seg);
    (void)hipFree(A->d_sweep_seg_dir);
    A->d_sweep_seg_dir = nullptr;
    (void)hipFree(A->d_sweep_seg_first);
    A->d_sweep_seg_first = nullptr;
    A->v.sweep_seg_first = nullptr;
    A->v.sweep_T_first = A->v.sweep_nslots_first = 0;
    A->v.sweep_seg_dir = nullptr;
    A->v.sweep_T_dir = A->v.sweep_nslots_dir = 0;
    (void)hipFree(A->d_sweep_gen);
    (void)hipFree(A->d_canon_val);
    (void)hipFree(A->d_canon_mask);
    (void)hipFree(A->d_canon_sym_val);
    (void)hipFree(A->d_canon_sym_mask);
    (void)hipFree(A->d_chain_plane);
    (void)hipFree(A->d_chain_far);
    (void)hipFree(A->d_chain_dual);
    (void)hipFree(A->d_dual_chunks);
    A->d_chain_plane = A->d_chain_far = A->d_chain_dual = A->d_dual_chunks = nullptr;
    A->v.chain_dual = nullptr;
    A->v.dual_chunks = nullptr;
    A->v.dual_nchunks = A->v.dual_blocks = 0;
    A->h_chain_plane.clear();
    A->d_sweep_seg = A->d_sweep_gen = A->d_canon_val = A->d_canon_mask = A->d_canon_sym_val = A->d_canon_sym_mask = nullptr;
    A->v.canon_sym_val = nullptr;
    A->v.sweep_nslots = 0;
    (void)hipFree(A->d_pair_rle);
    A->d_pair_rle = nullptr;
    A->v.pair_rle = nullptr;
    void *ptrs[] = {A->d_pair_id, A->d_tile_ptable, A->d_ptbl_desc, A->d_ptbl_len, A->d_ptbl_val, A->d_ptbl_meta,
                    A->d_chunk_dual};
    for (void *p : ptrs) (void)hipFree(p);
    A->d_pair_id = A->d_tile_ptable = A->d_ptbl_desc = A->d_ptbl_len = A->d_ptbl_val = A->d_ptbl_meta = nullptr;
    A->d_chunk_dual = nullptr;
    A->v.pair_id = nullptr;
    A->v.chunk_dual = nullptr;
    A->v.pair_sym_base = 0;
}

}  // namespace schwz
