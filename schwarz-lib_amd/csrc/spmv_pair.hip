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

    // pattern id of the pair starting at row ra of `chunk` (workgroup-uniform)
    auto pair_pid = [&](int chunk, int ra) -> int {
        int pid;
        uint4 rle = {0xffffu, 0u, 0u, 0u};
        if (A.pair_rle) {
            // chunk is workgroup-uniform and the table is never written by a kernel: read it through
            // the constant address space, i.e. with a scalar load that costs no vector-memory slot
            typedef const unsigned __attribute__((address_space(4))) *const_words;
            const const_words q = (const_words)(uintptr_t)(A.pair_rle + chunk);
            rle.x = q[0];
            rle.y = q[1];
            rle.z = q[2];
            rle.w = q[3];
        }
        if ((rle.x & 0xffffu) != 0xffffu) {
            // the id of the last run that starts at or before this lane's pair (runs ascending,
            // unused slots repeat the last run)
            const unsigned w[4] = {rle.x, rle.y, rle.z, rle.w};
            pid = (int)((w[0] >> 8) & 0xffu);
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const unsigned e = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                if ((unsigned)tid >= (e & 0xffu)) pid = (int)(e >> 8);
            }
        } else {
            pid = A.pair_id[ra >> 1];
        }
        return pid;
    };
    // Fixed chunks of 512 consecutive rows, one pair per lane: nothing but the chunk's table id has
    // to be looked up before the pattern ids and the gathers can be requested.
    if (SINGLE) stage_table(0);
    // z-sweep walk (CsrView::sweep_*): the first sweep_nslots workgroups each take one segment -- a band
    // of T rows through the planes z0 <= z < z1 -- and read every operand of the canonical stencil layout
    // from an LDS ring of plane windows [band - NX, band + T + NX), each window fetched ONCE with
    // coalesced 16-byte loads, two planes ahead of its use.  Same products, same order, same masks as
    // accumulate_canon: same bits.  The next sweep_gen_blocks workgroups walk the chunks no segment
    // covers the generic way (loop below); the rest of the grid only writes its zero partial sums.
    constexpr bool kSweepMode = SINGLE && MODE == kSpmvCgUpdate;
    const bool listed = kSweepMode && a.sweep != 0;
    if (kSweepMode && listed && (int)blockIdx.x < A.sweep_nslots) {
        double *ring = reinterpret_cast<double *>(pair_lds);  // over the staged table: only cpv / cmask are used from here on
        const int4 sg = A.sweep_seg[blockIdx.x];
        const int T = A.sweep_T, NX = A.sweep_nx, W = T + 2 * NX, npieces = W / 2;
        const int64_t PL = A.sweep_pl;
        const int band = sg.x, z0 = sg.y, z1 = sg.z;
        constexpr int NL = 4;  // 16-byte pieces of a window per lane (W <= 2048)
        constexpr int kSlots = 4;
        const int nplanes = (int)(A.nrows / PL);
        auto load_window = [&](int z, pvd2 (&reg)[NL]) {
            const int64_t base = (int64_t)z * PL + (int64_t)band * T - NX;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int pc = tid + k * kBlock;
                const int64_t g = base + 2 * pc;
                pvd2 v = {0.0, 0.0};
                if (pc < npieces && z >= 0 && z <= nplanes && g >= 0 && g + 2 <= A.ncols) __builtin_memcpy(&v, a.x + g, 16);
                reg[k] = v;
            }
        };
        auto store_window = [&](int z, const pvd2 (&reg)[NL]) {
            double *slot_p = ring + (size_t)((z + kSlots) % kSlots) * W;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int pc = tid + k * kBlock;
                if (pc < npieces) *reinterpret_cast<pvd2 *>(slot_p + 2 * pc) = reg[k];
            }
        };
        if (z0 < z1) {
            pvd2 reg[NL];
            load_window(z0 - 1, reg);
            store_window(z0 - 1, reg);
            load_window(z0, reg);
            store_window(z0, reg);
            load_window(z0 + 1, reg);
            for (int z = z0; z < z1; ++z) {
                store_window(z + 1, reg);
                lds_barrier();
                if (z + 1 < z1) load_window(z + 2, reg);  // in flight while plane z is computed
                const double *cur = ring + (size_t)((z + kSlots) % kSlots) * W;
                const double *prv = ring + (size_t)((z - 1 + kSlots) % kSlots) * W;
                const double *nxt = ring + (size_t)((z + 1 + kSlots) % kSlots) * W;
                for (int h = 0; h < T / kPairRows; ++h) {
                    const int64_t row0 = (int64_t)z * PL + (int64_t)band * T + h * kPairRows;
                    const int chunk = (int)(row0 / kPairRows);
                    const int ra = (int)row0 + 2 * tid;
                    const int i0 = NX + h * kPairRows + 2 * tid;
                    // r (and x) only stream through: non-temporal, requested ahead of the LDS reads
                    const pvd2 rr = __builtin_nontemporal_load(reinterpret_cast<const pvd2 *>(a.cg_r + ra));
                    pvd2 cgx = {0.0, 0.0};
                    if (a.cg_x) cgx = __builtin_nontemporal_load(reinterpret_cast<const pvd2 *>(a.cg_x + ra));
                    double od0 = 1.0, od1 = 1.0;
                    if (a.diag_mode == 1) {
                        const pvd2 dd = __builtin_nontemporal_load(reinterpret_cast<const pvd2 *>(a.dinv + ra));
                        od0 = dd.x;
                        od1 = dd.y;
                    } else if (a.diag_mode == 3) {
                        od0 = od1 = a.diag_uniform;
                    }
                    const int pid = pair_pid(chunk, ra);
                    const int mask = cmask[pid];
                    pvd2 t[7];
                    t[3] = *reinterpret_cast<const pvd2 *>(cur + i0);
                    t[2].x = cur[i0 - 1];
                    t[2].y = t[3].x;
                    t[4].x = t[3].y;
                    t[4].y = cur[i0 + 2];
                    t[1] = *reinterpret_cast<const pvd2 *>(cur + i0 - NX);
                    t[5] = *reinterpret_cast<const pvd2 *>(cur + i0 + NX);
                    t[0] = *reinterpret_cast<const pvd2 *>(prv + i0);
                    t[6] = *reinterpret_cast<const pvd2 *>(nxt + i0);
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int k = 0; k < 7; ++k) {
                        const PairVal v = cpv[pid * 8 + k];
                        if ((mask >> k) & 1) s0 += v.a * t[k].x;
                        if ((mask >> (kPairChunk + k)) & 1) s1 += v.b * t[k].y;
                    }
                    double y0, y1, p0, p1;
                    finish(ra, s0, 0.0, false, t[3].x, rr.x, od0, y0, p0);
                    finish(ra + 1, s1, 0.0, false, t[3].y, rr.y, od1, y1, p1);
                    const pvd2 rn = {y0, y1};
                    __builtin_nontemporal_store(rn, reinterpret_cast<pvd2 *>(a.cg_r + ra));
                    if (a.cg_x) {
                        const pvd2 xx = {cgx.x + p0, cgx.y + p1};
                        __builtin_nontemporal_store(xx, reinterpret_cast<pvd2 *>(a.cg_x + ra));
                    }
                }
            }
        }
    }
    const int gen_first = listed ? (int)blockIdx.x - A.sweep_nslots : slot;
    const int gen_count = listed ? ((int)blockIdx.x < A.sweep_nslots + A.sweep_gen_blocks ? A.sweep_ngen : 0) : slots;
    const int gen_stride = listed ? A.sweep_gen_blocks : per_xcd;
    for (int j = gen_first; j >= 0 && j < gen_count; j += gen_stride) {
        const int chunk = listed ? A.sweep_gen[j] : xcd_chunk(nchunks, sh, xcd, j);
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
            a.partials[blockIdx.x] = s0;
            a.partials[gridDim.x + blockIdx.x] = s1;
        }
        if (MODE == kSpmvResidDual) {
            const double s2v = block_sum(acc2, red);
            if (tid == 0) a.partials[2 * gridDim.x + blockIdx.x] = s2v;
        }
    }
    if (kDir && blockIdx.x == 0 && tid == 0) {
        // what cg_direction_kernel's workgroup 0 does: the rho slot written is the one no launch of
        // this iteration reads; stop_iter = 0 makes every later launch leave at once
        CgState *st = const_cast<CgState *>(a.cg_state);
        st->rho[(a.it + 1) & 1] = cg_rho_new;
        st->rr = cg_rr;
        st->iters = st->iters + 1;
        if (sqrt(cg_rr) <= a.cg_rtol * st->r0) st->stop_iter = 0;
    }
}

int launch_spmv_pair(const CsrView &A, int mode, const SpmvArgs &a, int grid, hipStream_t s)
{
    const bool wide = A.ncols >= (int64_t(1) << 28);  // byte offsets of x beyond 32 bits
    if (mode == kSpmvCgUpdate && !wide && A.pair_single && A.sweep_nslots > 0 &&
        A.sweep_nslots + A.sweep_gen_blocks <= grid) {
        // z-sweep walk: the LDS ring of plane windows is dynamic shared memory (only this launch form pays for it)
        const char *sweep_env = std::getenv("SCHWZ_CG_SWEEP");  // read per launch: tests switch it
        if (!(sweep_env && sweep_env[0] == '0')) {
            static const hipError_t attr = hipFuncSetAttribute((const void *)spmv_pair_kernel<kSpmvCgUpdate, false, true>,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);
            (void)attr;
            SpmvArgs b = a;
            b.sweep = 1;
            const size_t lds = std::max<size_t>(kPairTableLds, (size_t)4 * (A.sweep_T + 2 * A.sweep_nx) * sizeof(double));
            hipLaunchKernelGGL((spmv_pair_kernel<kSpmvCgUpdate, false, true>), dim3(grid), dim3(kBlock), lds, s, A, b);
            SCHWZ_HIP_TRY(hipGetLastError());
            return SCHWZ_OK;
        }
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
        bool sorted = true;
        for (int64_t r = 0; r < nrows && sorted; ++r) {
            if (rp[r + 1] - rp[r] > 127) sorted = false;
            for (schwz_idx j = rp[r] + 1; j < rp[r + 1] && sorted; ++j)
                if (col[j] <= col[j - 1]) sorted = false;
        }
        single = single && sorted;
        int lmax = 1;
        std::unordered_multimap<uint64_t, int> seen;
        for (int64_t ra = 0; ra < nrows && single; ra += 2) {
            merge_pair(ra, ra + 1 < nrows);
            uint64_t h = 1469598103934665603ull;
            for (const PairEntryH &e : cur) {
                h = (h ^ e.va) * 1099511628211ull;
                h = (h ^ e.vb) * 1099511628211ull;
                h = (h ^ (uint64_t)(int64_t)e.off) * 1099511628211ull;
                h = (h ^ (uint64_t)e.flags) * 1099511628211ull;
            }
            int id = -1;
            auto range = seen.equal_range(h);
            for (auto it = range.first; it != range.second; ++it)
                if (pats[(size_t)it->second] == cur) {
                    id = it->second;
                    break;
                }
            if (id < 0) {
                id = (int)pats.size();
                lmax = std::max(lmax, (int)cur.size());
                if (id == kPairPats || (int64_t)(id + 1) * pair_stride(lmax) > kPairEntries) {
                    single = false;
                    break;
                }
                seen.emplace(h, id);
                pats.push_back(cur);
            }
            pair_id[(size_t)(ra >> 1)] = (uint8_t)id;
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
    const char *rle_env = std::getenv("SCHWZ_SPMV_RLE");
    std::vector<uint16_t> rle;
    if (!(rle_env && rle_env[0] == '0')) {
        rle.assign((size_t)nchunks * 8, 0xffffu);
        for (int c = 0; c < nchunks; ++c) {
            if (chunk_ptable[(size_t)c] < 0) continue;
            const int64_t p0 = (int64_t)c * (kPairRows / 2), p1 = std::min<int64_t>(p0 + kPairRows / 2, (nrows + 1) / 2);
            uint16_t runs[8];
            int nr = 0;
            bool fits = true;
            for (int64_t p = p0; p < p1 && fits; ++p) {
                if (nr == 0 || pair_id[(size_t)p] != (uint8_t)(runs[nr - 1] >> 8)) {
                    if (nr == 8) {
                        fits = false;
                        break;
                    }
                    runs[nr++] = (uint16_t)((p - p0) | ((int)pair_id[(size_t)p] << 8));
                }
            }
            if (!fits || nr == 0) continue;
            for (int k = nr; k < 8; ++k) runs[k] = runs[nr - 1];
            if (runs[0] == 0xffffu) continue;  // would read as the "not coded" marker
            std::copy(runs, runs + 8, rle.begin() + (size_t)c * 8);
        }
        if ((rc = upv(rle, &A->d_pair_rle))) return rc;
        A->v.pair_rle = (const uint4 *)A->d_pair_rle;
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
    // z-sweep segments (CsrView::sweep_*) for a canonical layout {-PL, -NX, -1, 0, +1, +NX, +PL} with
    // PL a multiple of the band length: SCHWZ_SPMV_SWEEP=0 off, =2 also below a million rows (tests);
    // SCHWZ_SWEEP_T (512 / 1024 rows per band), SCHWZ_SWEEP_L (planes per segment) override the defaults.
    {
        const char *sw_env = std::getenv("SCHWZ_SPMV_SWEEP");
        const int sw_mode = sw_env ? std::atoi(sw_env) : 1;
        const int *cn = A->v.pair_canon;
        const int64_t NX = cn[5], PL = cn[6];
        const bool shape_ok = single && cn[7] && cn[0] == -PL && cn[1] == -NX && NX >= 2 && PL > NX && NX % 2 == 0 &&
                              PL % kPairRows == 0;
        if (sw_mode != 0 && shape_ok && (nrows >= (int64_t(1) << 20) || sw_mode == 2) && nrows >= 3 * PL) {
            const PairTable &tb = tables[0];
            // a pattern fits the layout when each of its entries has one of the seven offsets (the device
            // stages cmask the same way, stage_table)
            std::vector<uint8_t> pat_ok((size_t)tb.npat, 1);
            for (int q = 0; q < tb.npat; ++q) {
                if ((int)tb.len[(size_t)q] > kPairChunk) pat_ok[(size_t)q] = 0;
                for (int k = 0; k < (int)tb.len[(size_t)q]; ++k) {
                    const schwz_idx off = tb.ent[(size_t)q * tb.lmax + k].off;
                    bool in = false;
                    for (int t = 0; t < 7; ++t) in = in || cn[t] == off;
                    if (!in) pat_ok[(size_t)q] = 0;
                }
            }
            std::vector<uint8_t> chunk_ok((size_t)nchunks, 0);
            for (int c = 0; c < nchunks; ++c) {
                const int64_t p0 = (int64_t)c * (kPairRows / 2), p1 = p0 + kPairRows / 2;
                if (p1 * 2 > nrows) continue;  // a partial chunk stays generic
                bool ok = true;
                for (int64_t q = p0; q < p1 && ok; ++q) ok = pat_ok[(size_t)pair_id[(size_t)q]] != 0;
                chunk_ok[(size_t)c] = ok ? 1 : 0;
            }
            const char *t_env = std::getenv("SCHWZ_SWEEP_T"), *l_env = std::getenv("SCHWZ_SWEEP_L");
            int T = t_env ? std::atoi(t_env) : (NX >= 512 ? 1024 : 512);
            if (T != 512 && T != 1024) T = 512;
            if (PL % T) T = 512;
            const int64_t W = T + 2 * NX;
            const int nplanes = (int)(nrows / PL), bands = (int)(PL / T), cpb = T / kPairRows, cpp = (int)(PL / kPairRows);
            const int grid = ((std::min<int64_t>((int64_t)tiles.size() - 1, kMaxGrid) + kXcds - 1) / kXcds) * kXcds;
            if (W <= 2048 && bands >= 1 && nplanes >= 3) {
                // maximal runs of planes in which every chunk of the band is canonical
                struct Run { int band, z0, z1; };
                std::vector<Run> runs;
                int64_t steps = 0;
                for (int b = 0; b < bands; ++b) {
                    int z = 0;
                    while (z < nplanes) {
                        auto okz = [&](int zz) {
                            for (int h = 0; h < cpb; ++h)
                                if (!chunk_ok[(size_t)((int64_t)zz * cpp + (int64_t)b * cpb + h)]) return false;
                            return true;
                        };
                        if (!okz(z)) {
                            ++z;
                            continue;
                        }
                        int e = z;
                        while (e < nplanes && okz(e)) ++e;
                        if (e - z >= 2) {
                            runs.push_back({b, z, e});
                            steps += e - z;
                        }
                        z = e;
                    }
                }
                std::vector<uint8_t> covered((size_t)nchunks, 0);
                int L = l_env ? std::atoi(l_env) : 16;
                if (L < 2) L = 16;
                // the segments and the workgroups of the generic walk must fit the launch grid
                auto count = [&](int len) {
                    int64_t n = 0;
                    for (const Run &r : runs) n += (r.z1 - r.z0 + len - 1) / len;
                    return n;
                };
                for (const Run &r : runs)
                    for (int z = r.z0; z < r.z1; ++z)
                        for (int h = 0; h < cpb; ++h) covered[(size_t)((int64_t)z * cpp + (int64_t)r.band * cpb + h)] = 1;
                std::vector<schwz_idx> gen;
                for (int c = 0; c < nchunks; ++c)
                    if (!covered[(size_t)c]) gen.push_back(c);
                const int gen_blocks = (int)std::min<size_t>(gen.size(), 256);
                while (count(L) + kXcds > grid - gen_blocks && L < (1 << 20)) L += 4;
                // deal: XCD x takes the bands [x * bands / 8, (x + 1) * bands / 8) (a band's window shares its
                // NX-row halos with the neighbouring bands: the same L2), segment by segment of the z range
                std::vector<std::vector<int4>> per_xcd(kXcds);
                struct Seg { int band, z0, z1; };
                std::vector<Seg> segs;
                for (const Run &r : runs) {
                    const int nseg = (r.z1 - r.z0 + L - 1) / L, len = (r.z1 - r.z0 + nseg - 1) / nseg;
                    for (int z = r.z0; z < r.z1; z += len) segs.push_back({r.band, z, std::min(z + len, r.z1)});
                }
                std::stable_sort(segs.begin(), segs.end(), [](const Seg &x, const Seg &y) {
                    return x.z0 != y.z0 ? x.z0 < y.z0 : x.band < y.band;
                });
                for (const Seg &sgm : segs) {
                    const int x = bands >= 2 * kXcds ? (int)((int64_t)sgm.band * kXcds / bands) : -1;
                    int4 v;
                    v.x = sgm.band;
                    v.y = sgm.z0;
                    v.z = sgm.z1;
                    v.w = 0;
                    if (x >= 0) {
                        per_xcd[(size_t)x].push_back(v);
                    } else {  // few bands: round robin
                        size_t best = 0;
                        for (size_t k = 1; k < (size_t)kXcds; ++k)
                            if (per_xcd[k].size() < per_xcd[best].size()) best = k;
                        per_xcd[best].push_back(v);
                    }
                }
                size_t depth = 0;
                for (const auto &l : per_xcd) depth = std::max(depth, l.size());
                std::vector<int4> slots_v(depth * kXcds);
                for (size_t q = 0; q < depth; ++q)
                    for (int x = 0; x < kXcds; ++x) {
                        int4 v;
                        v.x = v.y = v.z = v.w = 0;
                        if (q < per_xcd[(size_t)x].size()) v = per_xcd[(size_t)x][q];
                        slots_v[q * kXcds + x] = v;
                    }
                if (!segs.empty() && (int64_t)slots_v.size() + gen_blocks <= grid && steps * T * 2 >= nrows) {
                    if ((rc = upv(slots_v, &A->d_sweep_seg)) || (rc = upv(gen, &A->d_sweep_gen))) return rc;
                    A->v.sweep_T = T;
                    A->v.sweep_nx = (int)NX;
                    A->v.sweep_pl = PL;
                    A->v.sweep_nslots = (int)slots_v.size();
                    A->v.sweep_ngen = (int)gen.size();
                    A->v.sweep_gen_blocks = gen_blocks;
                    A->v.sweep_seg = (const int4 *)A->d_sweep_seg;
                    A->v.sweep_gen = (const schwz_idx *)A->d_sweep_gen;
                }
            }
        }
    }
    {
        // what a pass over the coded matrix reads: per chunk its 16-byte run-length record, or one byte per
        // pair where the ids do not run-length code; the chunk's table id unless one table serves the whole
        // matrix; the tables themselves (one set; the upper-triangle twins are read INSTEAD by kSpmvDotSym)
        int64_t bytes = 0;
        const uint16_t *rle_h = rle.empty() ? nullptr : rle.data();
        for (int c = 0; c < nchunks; ++c) {
            if (chunk_ptable[(size_t)c] < 0) continue;
            const bool runs = rle_h && rle_h[(size_t)c * 8] != 0xffffu;
            bytes += runs ? 16 : (rle_h ? 16 : 0) + kPairRows / 2;
            if (!single) bytes += 4;
        }
        const size_t ntab = sym_base ? (size_t)sym_base : tables.size();
        for (size_t t = 0; t < ntab; ++t) bytes += (int64_t)tables[t].ent.size() * 24 + tables[t].npat;
        A->pair_code_bytes = bytes;
    }
    // the XCD deal of the chunks: the tile deal's run length in rows, in chunks (a power of two)
    int sh = A->v.xcd_shift;
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
    return SCHWZ_OK;
}

void free_spmv_pair(schwz_csr *A)
{
    (void)hipFree(A->d_sweep_seg);
    (void)hipFree(A->d_sweep_gen);
    A->d_sweep_seg = A->d_sweep_gen = nullptr;
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
