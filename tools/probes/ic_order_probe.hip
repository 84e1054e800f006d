// Does the 256 MiB Infinity Cache serve the head of a launch from the tail of the previous one?
// Two streaming launches over n-double vectors, like the two launches of a CG iteration on the walk:
//   U: r -= a * p              (reads p, r; writes r)
//   D: pn = r + b * p          (reads r, p; writes pn)
// timed as the pair U, D with D's workgroups dealt (a) in the same order as U's, (b) in the opposite order, and
// (c) both alternating every launch (U up, D down, U down, D up, ...).  Short-lived workgroups, 16 bytes per lane.
//   hipcc --offload-arch=gfx950 -O3 -o ic_order_probe ic_order_probe.hip && ./ic_order_probe [n_millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef double vd2 __attribute__((ext_vector_type(2)));
constexpr int kBlock = 256, kPer = 4;  // 4 x 16 B per lane

template <bool NT, int NTL = 0>
__global__ __launch_bounds__(kBlock) void upd(int64_t n2, vd2 *__restrict__ r, const vd2 *__restrict__ p, double a, int rev)
{
    const int64_t b = rev ? (int64_t)gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int64_t base = b * kBlock * kPer + threadIdx.x;
    vd2 rv[kPer], pv[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int64_t i = base + (int64_t)k * kBlock;
        if (i < n2) {
            rv[k] = (NTL & 1) ? __builtin_nontemporal_load(r + i) : r[i];
            pv[k] = (NTL & 2) ? __builtin_nontemporal_load(p + i) : p[i];
        }
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int64_t i = base + (int64_t)k * kBlock;
        if (i < n2) {
            vd2 v = rv[k] - a * pv[k];
            if (NT) __builtin_nontemporal_store(v, r + i); else r[i] = v;
        }
    }
}

template <bool NT, int NTL = 0>
__global__ __launch_bounds__(kBlock) void dir(int64_t n2, vd2 *__restrict__ pn, const vd2 *__restrict__ r, const vd2 *__restrict__ p, double bta, int rev)
{
    const int64_t b = rev ? (int64_t)gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int64_t base = b * kBlock * kPer + threadIdx.x;
    vd2 rv[kPer], pv[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int64_t i = base + (int64_t)k * kBlock;
        if (i < n2) {
            rv[k] = (NTL & 1) ? __builtin_nontemporal_load(r + i) : r[i];
            pv[k] = (NTL & 2) ? __builtin_nontemporal_load(p + i) : p[i];
        }
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int64_t i = base + (int64_t)k * kBlock;
        if (i < n2) {
            vd2 v = rv[k] + bta * pv[k];
            if (NT) __builtin_nontemporal_store(v, pn + i); else pn[i] = v;
        }
    }
}

// "blocked" order: `gridDim.x` persistent workgroups, each walking its own contiguous 1/gridDim.x of the vectors from
// its start upwards -- the temporal order of the z-sweep walks (every segment of the grid advances at the same time),
// against the dispatch order of the short-lived workgroups above (one front moving through the vectors)
template <bool NT>
__global__ __launch_bounds__(kBlock) void upd_blocked(int64_t n2, vd2 *__restrict__ r, const vd2 *__restrict__ p, double a)
{
    const int64_t chunk = (n2 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = blockIdx.x * chunk, hi = lo + chunk < n2 ? lo + chunk : n2;
    for (int64_t base = lo + threadIdx.x; base < hi; base += kBlock * kPer) {
        vd2 rv[kPer], pv[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int64_t i = base + (int64_t)k * kBlock;
            if (i < hi) { rv[k] = r[i]; pv[k] = p[i]; }
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int64_t i = base + (int64_t)k * kBlock;
            if (i < hi) {
                vd2 v = rv[k] - a * pv[k];
                if (NT) __builtin_nontemporal_store(v, r + i); else r[i] = v;
            }
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void dir_blocked(int64_t n2, vd2 *__restrict__ pn, const vd2 *__restrict__ r, const vd2 *__restrict__ p, double bta)
{
    const int64_t chunk = (n2 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = blockIdx.x * chunk, hi = lo + chunk < n2 ? lo + chunk : n2;
    for (int64_t base = lo + threadIdx.x; base < hi; base += kBlock * kPer) {
        vd2 rv[kPer], pv[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int64_t i = base + (int64_t)k * kBlock;
            if (i < hi) { rv[k] = r[i]; pv[k] = p[i]; }
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int64_t i = base + (int64_t)k * kBlock;
            if (i < hi) {
                vd2 v = rv[k] + bta * pv[k];
                if (NT) __builtin_nontemporal_store(v, pn + i); else pn[i] = v;
            }
        }
    }
}

int main(int argc, char **argv)
{
    const int64_t n = (int64_t)((argc > 1 ? std::atof(argv[1]) : 16.9) * 1e6) / 2 * 2;
    const int64_t n2 = n / 2;
    const int ring = 4;
    double *r;
    std::vector<double *> p(ring);
    CHECK(hipMalloc(&r, n * 8));
    CHECK(hipMemset(r, 0, n * 8));
    for (auto &q : p) { CHECK(hipMalloc(&q, n * 8)); CHECK(hipMemset(q, 0, n * 8)); }
    const int grid = (int)((n2 + kBlock * kPer - 1) / (kBlock * kPer));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 40;
    std::printf("n = %lld doubles (%.1f MB per vector), grid %d\n", (long long)n, n * 8 / 1e6, grid);
    for (int nt = 0; nt < 2; ++nt)
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                for (int it = 0; it < iters; ++it) {
                    // mode 0: U up, D up.  mode 1: U up, D down.  mode 2: launches alternate: U up, D down, U down... no:
                    // every launch runs opposite to the one before it
                    const int urev = mode == 2 ? (it & 1) : 0;
                    const int drev = mode == 0 ? 0 : (mode == 1 ? 1 : !(it & 1));
                    // mode 2: it even: U up (0), D down (1); it odd: U ... the launch before was D down, so U up again
                    // would meet cold data: U must run UP after D DOWN.  So U is always up, D always down = mode 1;
                    // mode 2 instead: U up, D down, U up, D down is mode 1 -- mode 2 tests U down after D down (same
                    // order as its predecessor) as a control: U dir = D dir of the pair before
                    vd2 *pk = (vd2 *)p[it % ring], *pn = (vd2 *)p[(it + 1) % ring];
                    if (nt) {
                        hipLaunchKernelGGL(upd<true>, dim3(grid), dim3(kBlock), 0, 0, n2, (vd2 *)r, pk, 1e-3, urev);
                        hipLaunchKernelGGL(dir<true>, dim3(grid), dim3(kBlock), 0, 0, n2, pn, (const vd2 *)r, pk, 0.5, drev);
                    } else {
                        hipLaunchKernelGGL(upd<false>, dim3(grid), dim3(kBlock), 0, 0, n2, (vd2 *)r, pk, 1e-3, urev);
                        hipLaunchKernelGGL(dir<false>, dim3(grid), dim3(kBlock), 0, 0, n2, pn, (const vd2 *)r, pk, 0.5, drev);
                    }
                }
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double us_pair = best * 1e3 / iters;
            std::printf("%s stores  %-44s %7.1f us per pair   %.2f TB/s of 48n bytes\n", nt ? "nt   " : "plain",
                        mode == 0 ? "U up, D up" : mode == 1 ? "U up, D down" : "pairs alternate (U and D same way, next pair opposite)",
                        us_pair, 48.0 * n / us_pair / 1e6);
        }
    // non-temporal LOADS (stores non-temporal throughout): which of the two launches' streams may carry the hint
    struct Case { const char *name; int u, d; };
    const Case cases[] = {{"plain loads", 0, 0}, {"U: r nt", 1, 0}, {"U: r, p nt", 3, 0}, {"D: r nt", 0, 1}, {"D: r, p nt", 0, 3}, {"all nt", 3, 3}};
    for (const Case &c : cases) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            for (int it = 0; it < iters; ++it) {
                vd2 *pk = (vd2 *)p[it % ring], *pn = (vd2 *)p[(it + 1) % ring];
                switch (c.u) {
                case 0: hipLaunchKernelGGL((upd<true, 0>), dim3(grid), dim3(kBlock), 0, 0, n2, (vd2 *)r, pk, 1e-3, 0); break;
                case 1: hipLaunchKernelGGL((upd<true, 1>), dim3(grid), dim3(kBlock), 0, 0, n2, (vd2 *)r, pk, 1e-3, 0); break;
                default: hipLaunchKernelGGL((upd<true, 3>), dim3(grid), dim3(kBlock), 0, 0, n2, (vd2 *)r, pk, 1e-3, 0); break;
                }
                switch (c.d) {
                case 0: hipLaunchKernelGGL((dir<true, 0>), dim3(grid), dim3(kBlock), 0, 0, n2, pn, (const vd2 *)r, pk, 0.5, 0); break;
                case 1: hipLaunchKernelGGL((dir<true, 1>), dim3(grid), dim3(kBlock), 0, 0, n2, pn, (const vd2 *)r, pk, 0.5, 0); break;
                default: hipLaunchKernelGGL((dir<true, 3>), dim3(grid), dim3(kBlock), 0, 0, n2, pn, (const vd2 *)r, pk, 0.5, 0); break;
                }
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double us_pair = best * 1e3 / iters;
        std::printf("nt stores, %-14s %7.1f us per pair   %.2f TB/s of 48n bytes\n", c.name, us_pair, 48.0 * n / us_pair / 1e6);
    }
    // blocked order (nt stores, plain loads), workgroups = 3, 4, 6, 8 per CU
    for (int per_cu : {3, 4, 6, 8, 16}) {
        const int g = 256 * per_cu;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            for (int it = 0; it < iters; ++it) {
                vd2 *pk = (vd2 *)p[it % ring], *pn = (vd2 *)p[(it + 1) % ring];
                hipLaunchKernelGGL(upd_blocked<true>, dim3(g), dim3(kBlock), 0, 0, n2, (vd2 *)r, pk, 1e-3);
                hipLaunchKernelGGL(dir_blocked<true>, dim3(g), dim3(kBlock), 0, 0, n2, pn, (const vd2 *)r, pk, 0.5);
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double us_pair = best * 1e3 / iters;
        std::printf("blocked order, %2d workgroups per CU      %7.1f us per pair   %.2f TB/s of 48n bytes\n", per_cu, us_pair, 48.0 * n / us_pair / 1e6);
    }
    return 0;
}
