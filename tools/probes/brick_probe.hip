// Micro-benchmark for a z-sweep ("brick") walk of the stencil-shaped CG launches: a workgroup keeps
// a band of T rows of one plane and sweeps it through L consecutive planes; each plane's window of p
// (the band + nx rows either side) is loaded ONCE with coalesced 16-byte loads into an LDS ring, and
// all seven operands of a row pair come from LDS (own / +-1 / +-nx from the current window, +-plane
// from the own parts of the previous and next window).  Compared in the same process with the
// gather structure the library uses today (six 16-byte gathers per row pair, or four for the
// upper-triangle p.(A p)).  No matrix data: the time is what the memory structure costs.
//   hipcc --offload-arch=gfx950 -O3 brick_probe.hip -o brick_probe ; ./brick_probe [nx ny nz]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double vd2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ vd2 ld2(const double *x, long i)
{
    vd2 v;
    __builtin_memcpy(&v, x + i, 16);
    return v;
}

__device__ __forceinline__ double block_sum(double v, double *red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---- today's structure ------------------------------------------------------------------------
// MODE 0: p.(A p) from the upper triangle (4 gathers), MODE 1: update r -= alpha A p (6 gathers, the
// offset-0 operands from the +-1 gathers) with non-temporal r load / store
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ p, double *__restrict__ r, long n,
                                                     int nx, long plane, double alpha, double *__restrict__ part)
{
    __shared__ double red[4];
    double acc = 0.0;
    const long nchunks = n / 512;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long ra = c * 512 + 2 * threadIdx.x;
        if (MODE == 0) {
            if (ra + plane + 2 > n) continue;
            const vd2 own = ld2(p, ra), a = ld2(p, ra + 1), b = ld2(p, ra + nx), d = ld2(p, ra + plane);
            const vd2 s = 6.0 * own - 2.0 * (a + b + d);
            acc += own.x * s.x + own.y * s.y;
        } else {
            if (ra - plane < 0 || ra + plane + 2 > n) continue;
            const vd2 rr = __builtin_nontemporal_load(reinterpret_cast<const vd2 *>(r + ra));
            const vd2 t0 = ld2(p, ra - plane), t1 = ld2(p, ra - nx), t2 = ld2(p, ra - 1), t4 = ld2(p, ra + 1),
                      t5 = ld2(p, ra + nx), t6 = ld2(p, ra + plane);
            vd2 own;
            own.x = t2.y;
            own.y = t4.x;
            const vd2 s = 6.0 * own - (t0 + t1 + t2 + t4 + t5 + t6);
            const vd2 rn = rr - alpha * s;
            __builtin_nontemporal_store(rn, reinterpret_cast<vd2 *>(r + ra));
            acc += rn.x * rn.x + rn.y * rn.y;
        }
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ---- z-sweep with an LDS ring -----------------------------------------------------------------
// T rows per band (multiple of 512), RPL = T / 256 rows per lane (2 or 4), SLOTS ring slots of
// W = T + 2 nx doubles.  Grid = bands x segments; blockIdx -> (band, segment) so that the bands of
// one XCD are contiguous.
template <int MODE, int RPL>
__global__ __launch_bounds__(256) void sweep_kernel(const double *__restrict__ p, double *__restrict__ r, long n, int nx,
                                                    long plane, int nz, int L, double alpha,
                                                    double *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *ring = reinterpret_cast<double *>(smem);
    __shared__ double red[4];
    constexpr int T = 256 * RPL;
    constexpr int SLOTS = 4;
    const int W = T + 2 * nx;
    const int bands = (int)(plane / T);
    const int nseg = (nz + L - 1) / L;
    const int xcd = blockIdx.x % 8, q = blockIdx.x / 8;
    const int bpx = bands / 8;  // bands per XCD (bands is a multiple of 8 here)
    const int band = xcd * bpx + q % bpx, seg = q / bpx;
    double acc = 0.0;
    if (seg >= nseg) {
        const double s = block_sum(acc, red);
        if (threadIdx.x == 0) part[blockIdx.x] = s;
        return;
    }
    const int z0 = seg * L, z1 = min(nz, z0 + L);
    const int tid = threadIdx.x;
    // the W / 2 16-byte pieces of a window are dealt to the lanes round-robin: NL per lane
    const int npieces = W / 2;
    const int NLmax = 4;  // W <= 2048
    auto win_base = [&](int z) -> long { return (long)z * plane + (long)band * T - nx; };
    auto load_window = [&](int z, vd2 (&reg)[NLmax]) {
        const long base = win_base(z);
#pragma unroll
        for (int k = 0; k < NLmax; ++k) {
            const int pc = tid + k * 256;
            const long g = base + 2L * pc;
            vd2 v = {0.0, 0.0};
            if (pc < npieces && z >= 0 && z < nz && g >= 0 && g + 2 <= n) v = ld2(p, g);
            reg[k] = v;
        }
    };
    auto store_window = [&](int z, const vd2 (&reg)[NLmax]) {
        double *slot = ring + (size_t)((z + SLOTS) % SLOTS) * W;
#pragma unroll
        for (int k = 0; k < NLmax; ++k) {
            const int pc = tid + k * 256;
            if (pc < npieces) *reinterpret_cast<vd2 *>(slot + 2 * pc) = reg[k];
        }
    };
    vd2 reg[NLmax];
    load_window(z0 - 1, reg);
    store_window(z0 - 1, reg);
    load_window(z0, reg);
    store_window(z0, reg);
    load_window(z0 + 1, reg);
    for (int z = z0; z < z1; ++z) {
        store_window(z + 1, reg);
        __syncthreads();
        if (z + 2 <= z1) load_window(z + 2, reg);  // in flight during the compute of plane z
        const double *cur = ring + (size_t)((z + SLOTS) % SLOTS) * W;
        const double *prv = ring + (size_t)((z - 1 + SLOTS) % SLOTS) * W;
        const double *nxt = ring + (size_t)((z + 1 + SLOTS) % SLOTS) * W;
#pragma unroll
        for (int h = 0; h < RPL / 2; ++h) {
            const int i0 = nx + 2 * tid + 512 * h;  // window index of row ra
            const long ra = (long)z * plane + (long)band * T + 2 * tid + 512 * h;
            const vd2 own = *reinterpret_cast<const vd2 *>(cur + i0);
            const double right = cur[i0 + 2];
            const vd2 up = *reinterpret_cast<const vd2 *>(cur + i0 + nx);
            const vd2 fw = *reinterpret_cast<const vd2 *>(nxt + i0);
            if (MODE == 0) {
                vd2 a;
                a.x = own.y;
                a.y = right;
                const vd2 s = 6.0 * own - 2.0 * (a + up + fw);
                acc += own.x * s.x + own.y * s.y;
            } else {
                const double left = cur[i0 - 1];
                const vd2 dn = *reinterpret_cast<const vd2 *>(cur + i0 - nx);
                const vd2 bk = *reinterpret_cast<const vd2 *>(prv + i0);
                const vd2 rr = __builtin_nontemporal_load(reinterpret_cast<const vd2 *>(r + ra));
                vd2 t2, t4;
                t2.x = left;
                t2.y = own.x;
                t4.x = own.y;
                t4.y = right;
                const vd2 s = 6.0 * own - (bk + dn + t2 + t4 + up + fw);
                const vd2 rn = rr - alpha * s;
                __builtin_nontemporal_store(rn, reinterpret_cast<vd2 *>(r + ra));
                acc += rn.x * rn.x + rn.y * rn.y;
            }
        }
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// one launch at a time behind a 600 MB memset: p and r are not in the Infinity Cache when the launch
// starts, as in the CG loop, where four more vectors move between two launches of the same kind
// ---- z-sweep with a REGISTER queue ---------------------------------------------------------------
// a lane keeps its row pair's p of the planes z-1 .. z+PF in registers (the +-plane operands are its
// own earlier / later loads), takes the +-1 operands from the neighbouring lanes (wave shuffles, two
// one-lane loads per wave at the wave's ends) and gathers only the +-nx operands: no LDS, no barrier.
// PF = planes of own loads kept in flight ahead of the plane being computed.
template <int MODE, int PF>
__global__ __launch_bounds__(256) void queue_kernel(const double *__restrict__ p, double *__restrict__ r, long n, int nx,
                                                    long plane, int nz, int L, double alpha,
                                                    double *__restrict__ part)
{
    __shared__ double red[4];
    const int bands = (int)(plane / 512);
    const int nseg = (nz + L - 1) / L;
    const int xcd = blockIdx.x % 8, q = blockIdx.x / 8;
    const int bpx = bands / 8;
    const int band = xcd * bpx + q % bpx, seg = q / bpx;
    double acc = 0.0;
    if (seg < nseg) {
        const int z0 = seg * L, z1 = min(nz, z0 + L);
        const int tid = threadIdx.x, lane = tid & 63;
        const long col0 = (long)band * 512 + 2 * tid;  // position inside the plane
        auto own_at = [&](int z) -> vd2 {
            vd2 v = {0.0, 0.0};
            if (z >= 0 && z < nz) v = ld2(p, (long)z * plane + col0);
            return v;
        };
        constexpr int Q = PF + 2;  // z-1, z, z+1 .. z+PF
        vd2 qv[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) qv[k] = own_at(z0 - 1 + k);
        for (int z = z0; z < z1; ++z) {
            const long ra = (long)z * plane + col0;
            const vd2 bk = qv[0], own = qv[1], fw = qv[2];
            // the newest plane of the queue, requested first
            const vd2 newest = own_at(z + PF + 1);
            vd2 up = {0.0, 0.0}, dn = {0.0, 0.0};
            if (ra + nx + 2 <= n) up = ld2(p, ra + nx);
            double right = __shfl_down(own.x, 1, 64);
            if (lane == 63 && ra + 3 <= n) right = p[ra + 2];
            if (MODE == 0) {
                vd2 a;
                a.x = own.y;
                a.y = right;
                const vd2 s = 6.0 * own - 2.0 * (a + up + fw);
                acc += own.x * s.x + own.y * s.y;
            } else {
                if (ra - nx >= 0) dn = ld2(p, ra - nx);
                double left = __shfl_up(own.y, 1, 64);
                if (lane == 0 && ra >= 1) left = p[ra - 1];
                const vd2 rr = __builtin_nontemporal_load(reinterpret_cast<const vd2 *>(r + ra));
                vd2 t2, t4;
                t2.x = left;
                t2.y = own.x;
                t4.x = own.y;
                t4.y = right;
                const vd2 s = 6.0 * own - (bk + dn + t2 + t4 + up + fw);
                const vd2 rn = rr - alpha * s;
                __builtin_nontemporal_store(rn, reinterpret_cast<vd2 *>(r + ra));
                acc += rn.x * rn.x + rn.y * rn.y;
            }
#pragma unroll
            for (int k = 0; k + 1 < Q; ++k) qv[k] = qv[k + 1];
            qv[Q - 1] = newest;
        }
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

template <typename F>
static float cold_time(F launch, void *junk, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    float tot = 0;
    for (int i = 0; i < reps; ++i) {
        hipMemsetAsync(junk, i, 600L << 20, 0);
        hipEventRecord(a, 0);
        launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        tot += ms;
    }
    hipEventDestroy(a);
    hipEventDestroy(b);
    return tot / reps;
}

template <typename F>
static float timeit(F launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int nx = argc > 3 ? atoi(argv[1]) : 256, ny = argc > 3 ? atoi(argv[2]) : 256, nz = argc > 3 ? atoi(argv[3]) : 256;
    const long plane = (long)nx * ny, n = plane * nz;
    double *p, *r, *part, *junk;
    hipMalloc(&p, n * 8 + 64);
    hipMalloc(&r, n * 8 + 64);
    hipMalloc(&part, 8 * 65536);
    hipMalloc(&junk, 600L << 20);
    std::vector<double> h((size_t)n);
    for (long i = 0; i < n; ++i) h[(size_t)i] = 1.0 + 1e-3 * (double)(i % 977);
    hipMemcpy(p, h.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(r, h.data(), n * 8, hipMemcpyHostToDevice);
    printf("grid %d x %d x %d, n = %ld (%.1f MB per vector); times: back to back / behind a 600 MB memset\n", nx, ny, nz, n,
           n * 8 / 1e6);
    hipFuncSetAttribute((const void *)sweep_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10);
    hipFuncSetAttribute((const void *)sweep_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10);
    hipFuncSetAttribute((const void *)sweep_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10);
    hipFuncSetAttribute((const void *)sweep_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10);
    std::vector<double> hp(65536);
    auto total = [&](int grid) {
        hipMemcpy(hp.data(), part, (size_t)grid * 8, hipMemcpyDeviceToHost);
        double t = 0;
        for (int i = 0; i < grid; ++i) t += hp[(size_t)i];
        return t;
    };
    for (int mode = 0; mode < 2; ++mode) {
        const double bytes = mode == 0 ? 8.0 * n : 24.0 * n;
        for (int grid : {1280, 2048}) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL((gather_kernel<0>), dim3(grid), dim3(256), 0, 0, p, r, n, nx, plane, 1e-9, part);
                else hipLaunchKernelGGL((gather_kernel<1>), dim3(grid), dim3(256), 0, 0, p, r, n, nx, plane, 1e-9, part);
            };
            const float ms = timeit(launch, 30), cold = cold_time(launch, junk, 10);
            printf("mode %d gathers                 grid %5d          : %.4f / %.4f ms (%.2f TB/s cold on %d n bytes)  sum %.6e\n", mode, grid,
                   ms, cold, bytes / cold / 1e9, mode == 0 ? 8 : 24, total(grid));
        }
        for (int pf : {1, 2, 3}) {
            if (plane % 512 || (plane / 512) % 8) continue;
            for (int L : {8, 16, 32, 64}) {
                if (L > nz) continue;
                const int bands = (int)(plane / 512), nseg = (nz + L - 1) / L;
                const int grid = bands * nseg;
                if (grid > 65536) continue;
                auto launch = [&] {
#define QL(M, P) hipLaunchKernelGGL((queue_kernel<M, P>), dim3(grid), dim3(256), 0, 0, p, r, n, nx, plane, nz, L, 1e-9, part)
                    if (mode == 0 && pf == 1) QL(0, 1);
                    if (mode == 0 && pf == 2) QL(0, 2);
                    if (mode == 0 && pf == 3) QL(0, 3);
                    if (mode == 1 && pf == 1) QL(1, 1);
                    if (mode == 1 && pf == 2) QL(1, 2);
                    if (mode == 1 && pf == 3) QL(1, 3);
#undef QL
                };
                const float ms = timeit(launch, 30), cold = cold_time(launch, junk, 10);
                printf("mode %d queue PF=%d     L=%2d grid %5d          : %.4f / %.4f ms (%.2f TB/s cold)  sum %.6e\n", mode, pf, L, grid, ms,
                       cold, bytes / cold / 1e9, total(grid));
            }
        }
        for (int rpl : {2, 4}) {
            const int T = 256 * rpl;
            if (plane % T || (plane / T) % 8) continue;
            const int W = T + 2 * nx;
            if (W > 2048) continue;
            const size_t lds = (size_t)4 * W * 8;
            for (int L : {8, 16, 32, 64}) {
                if (L > nz) continue;
                const int bands = (int)(plane / T), nseg = (nz + L - 1) / L;
                const int grid = bands * nseg;
                if (grid > 65536) continue;
                auto launch = [&] {
                    if (mode == 0 && rpl == 2) hipLaunchKernelGGL((sweep_kernel<0, 2>), dim3(grid), dim3(256), lds, 0, p, r, n, nx, plane, nz, L, 1e-9, part);
                    if (mode == 0 && rpl == 4) hipLaunchKernelGGL((sweep_kernel<0, 4>), dim3(grid), dim3(256), lds, 0, p, r, n, nx, plane, nz, L, 1e-9, part);
                    if (mode == 1 && rpl == 2) hipLaunchKernelGGL((sweep_kernel<1, 2>), dim3(grid), dim3(256), lds, 0, p, r, n, nx, plane, nz, L, 1e-9, part);
                    if (mode == 1 && rpl == 4) hipLaunchKernelGGL((sweep_kernel<1, 4>), dim3(grid), dim3(256), lds, 0, p, r, n, nx, plane, nz, L, 1e-9, part);
                };
                const float ms = timeit(launch, 30), cold = cold_time(launch, junk, 10);
                printf("mode %d sweep T=%4d L=%2d grid %5d lds %2zu KB: %.4f / %.4f ms (%.2f TB/s cold)  sum %.6e\n", mode, T, L, grid,
                       lds >> 10, ms, cold, bytes / cold / 1e9, total(grid));
            }
        }
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
