// Micro-benchmark behind the row-pattern SpMV design: 7-point stencil gathers from x with
// (a) one row per lane and 8-byte loads, (b) two adjacent rows per lane and 16-byte loads,
// (c) four adjacent rows per lane (two 16-byte loads per neighbour).  No matrix data at all:
// the time is what the gather + store structure costs.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double vd2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double ld(const double *x, long i) { return x[i]; }
__device__ __forceinline__ vd2 ld2(const double *x, long i)
{
    vd2 v;
    __builtin_memcpy(&v, x + i, 16);  // 8-byte aligned 16-byte load
    return v;
}

template <int ROWS>
__global__ __launch_bounds__(256) void stencil(const double *__restrict__ x, double *__restrict__ y, long n, int nx,
                                                long plane)
{
    const long per_wg = 256L * ROWS;
    for (long base = (long)blockIdx.x * per_wg; base < n; base += (long)gridDim.x * per_wg) {
        const long r = base + (long)threadIdx.x * ROWS;
        if (r + ROWS > n) continue;
        const long offs[7] = {-plane, -nx, -1, 0, 1, nx, plane};
        const double coef[7] = {-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0};
        if (ROWS == 1) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                long c = r + offs[k];
                c = c < 0 ? 0 : (c >= n ? n - 1 : c);
                s += coef[k] * ld(x, c);
            }
            y[r] = s;
        } else if (ROWS == 2) {
            vd2 s = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                long c = r + offs[k];
                c = c < 0 ? 0 : (c + 2 > n ? n - 2 : c);
                s += coef[k] * ld2(x, c);
            }
            *reinterpret_cast<vd2 *>(y + r) = s;
        } else {
            vd2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                long c = r + offs[k];
                c = c < 0 ? 0 : (c + 4 > n ? n - 4 : c);
                s0 += coef[k] * ld2(x, c);
                s1 += coef[k] * ld2(x, c + 2);
            }
            *reinterpret_cast<vd2 *>(y + r) = s0;
            *reinterpret_cast<vd2 *>(y + r + 2) = s1;
        }
    }
}


// The same two-rows-per-lane gather, made table driven step by step, to price the pieces of
// spmv_pair_kernel: LEVEL 1 = offsets from an LDS table (pattern 0 for every pair), 2 = pattern id
// loaded per pair from global memory (1 byte), 3 = values from LDS and presence-mask predication.
template <int LEVEL>
__global__ __launch_bounds__(256) void stencil_tab(const double *__restrict__ x, double *__restrict__ y, long n,
                                                    const int *__restrict__ g_off, const double *__restrict__ g_val,
                                                    const unsigned char *__restrict__ pid)
{
    __shared__ int off[64 * 8];
    __shared__ double val[64 * 8 * 2];
    __shared__ int mask[64];
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {
        off[i] = g_off[i & 7];
        val[2 * i] = g_val[i & 7];
        val[2 * i + 1] = g_val[i & 7];
    }
    if (threadIdx.x < 64) mask[threadIdx.x] = 0x7f7f;
    __syncthreads();
    const long per_wg = 512;
    for (long base = (long)blockIdx.x * per_wg; base < n; base += (long)gridDim.x * per_wg) {
        const long r = base + (long)threadIdx.x * 2;
        if (r + 2 > n) continue;
        int p = 0;
        if (LEVEL >= 2) p = pid[r >> 1];
        const int b = p * 8;
        const int m = LEVEL >= 3 ? mask[p] : 0x7f7f;
        vd2 t[8];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            long c = r + off[b + k];
            c = c < 0 ? 0 : (c + 2 > n ? n - 2 : c);
            t[k] = ld2(x, c);
        }
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            if (LEVEL >= 3) {
                if ((m >> k) & 1) s0 += val[2 * (b + k)] * t[k].x;
                if ((m >> (8 + k)) & 1) s1 += val[2 * (b + k) + 1] * t[k].y;
            } else {
                const double cf = k == 3 ? 6.0 : -1.0;
                s0 += cf * t[k].x;
                s1 += cf * t[k].y;
            }
        }
        vd2 s = {s0, s1};
        *reinterpret_cast<vd2 *>(y + r) = s;
    }
}

template <int LEVEL>
static float run_tab(const double *x, double *y, long n, const int *off, const double *val, const unsigned char *pid,
                     int grid, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(stencil_tab<LEVEL>, dim3(grid), dim3(256), 0, 0, x, y, n, off, val, pid);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(stencil_tab<LEVEL>, dim3(grid), dim3(256), 0, 0, x, y, n, off, val, pid);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}


// two rows per lane, every neighbour gathered from TWO vectors (r and p) and combined on the fly,
// one 16-byte store: the shape a fused "direction update + p.Ap" launch would have
__global__ __launch_bounds__(256) void stencil2v(const double *__restrict__ r, const double *__restrict__ p,
                                                  double *__restrict__ pn, long n, int nx, long plane, double beta)
{
    const long per_wg = 512;
    for (long base = (long)blockIdx.x * per_wg; base < n; base += (long)gridDim.x * per_wg) {
        const long rr = base + (long)threadIdx.x * 2;
        if (rr + 2 > n) continue;
        const long offs[7] = {-plane, -nx, -1, 0, 1, nx, plane};
        vd2 s = {0.0, 0.0}, own = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            long c = rr + offs[k];
            c = c < 0 ? 0 : (c + 2 > n ? n - 2 : c);
            const vd2 t = 0.1666 * ld2(r, c) + beta * ld2(p, c);
            if (k == 3) own = t;
            s += (k == 3 ? 6.0 : -1.0) * t;
        }
        *reinterpret_cast<vd2 *>(pn + rr) = own;
        if (s.x + s.y == 1.2345) pn[0] = s.x;
    }
}

template <int ROWS>
static float run(const double *x, double *y, long n, int nx, long plane, int grid, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stencil<ROWS>, dim3(grid), dim3(256), 0, 0, x, y, n, nx, plane);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stencil<ROWS>, dim3(grid), dim3(256), 0, 0, x, y, n, nx, plane);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char **argv)
{
    // default: the 256^3 cube; `gather_probe 512 512 64` = the per-GPU slab of the multi-GPU runs
    const int nx = argc > 3 ? atoi(argv[1]) : 256;
    const long plane = (long)nx * (argc > 3 ? atoi(argv[2]) : 256), n = plane * (argc > 3 ? atoi(argv[3]) : 256);
    double *x, *y, *junk;
    hipMalloc(&x, n * 8);
    hipMalloc(&y, n * 8);
    hipMalloc(&junk, 512L << 20);
    std::vector<double> h((size_t)n, 1.0);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    for (int grid : {1024, 2048, 4096}) {
        printf("grid %d: 1 row/lane %.4f ms, 2 rows/lane %.4f ms, 4 rows/lane %.4f ms\n", grid,
               run<1>(x, y, n, nx, plane, grid, 50), run<2>(x, y, n, nx, plane, grid, 50),
               run<4>(x, y, n, nx, plane, grid, 50));
    }
    // the same with the caches flushed by a 512 MiB memset between launches (x not resident)
    for (int rows = 1; rows <= 4; rows *= 2) {
        float tot = 0;
        for (int i = 0; i < 10; ++i) {
            hipMemsetAsync(junk, i, 512L << 20, 0);
            tot += rows == 1 ? run<1>(x, y, n, nx, plane, 2048, 1)
                             : (rows == 2 ? run<2>(x, y, n, nx, plane, 2048, 1) : run<4>(x, y, n, nx, plane, 2048, 1));
        }
        printf("cold-ish (after memset), %d rows/lane: %.4f ms (includes 3 warm launches each)\n", rows, tot / 10);
    }
    {
        int h_off[8] = {(int)-plane, -nx, -1, 0, 1, nx, (int)plane, 0};
        double h_val[8] = {-1, -1, -1, 6, -1, -1, -1, 0};
        int *d_off;
        double *d_val;
        unsigned char *d_pid;
        hipMalloc(&d_off, sizeof(h_off));
        hipMalloc(&d_val, sizeof(h_val));
        hipMalloc(&d_pid, n / 2);
        hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
        hipMemcpy(d_val, h_val, sizeof(h_val), hipMemcpyHostToDevice);
        hipMemset(d_pid, 0, n / 2);
        for (int grid : {1536, 2048}) {
            printf("table driven, grid %d: offsets from LDS %.4f ms, + pattern id from global %.4f ms, + values and masks %.4f ms\n",
                   grid, run_tab<1>(x, y, n, d_off, d_val, d_pid, grid, 50), run_tab<2>(x, y, n, d_off, d_val, d_pid, grid, 50),
                   run_tab<3>(x, y, n, d_off, d_val, d_pid, grid, 50));
        }
    }
    {
        double *p2;
        hipMalloc(&p2, n * 8);
        hipMemset(p2, 0, n * 8);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stencil2v, dim3(2048), dim3(256), 0, 0, x, p2, y, n, nx, plane, 0.5);
        hipEventRecord(a, 0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(stencil2v, dim3(2048), dim3(256), 0, 0, x, p2, y, n, nx, plane, 0.5);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("two rows/lane, gathers from two vectors + 16-byte store: %.4f ms\n", ms / 50);
    }
    return 0;
}
