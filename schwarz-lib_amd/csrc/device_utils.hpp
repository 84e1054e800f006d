// Device-side helpers shared by the kernel translation units (wave64, 256-thread workgroups).
#pragma once

#include <hip/hip_runtime.h>

#include "schwz_internal.hpp"

namespace schwz {

// ---------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over the 256-thread workgroup; result valid in every thread.
// `red` must hold 4 doubles.  Fixed order => bitwise reproducible.
__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();  // protect `red` from a previous use
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a
// workgroup-scope release/acquire fence over ALL address spaces, i.e. an
// s_waitcnt vmcnt(0): every global load or store still in flight (the prefetched
// matrix entries of the next tile, the y stores of the previous one) would have
// to land before the barrier.  The tiles only hand LDS data between waves.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Every workgroup folds the per-workgroup partial sums of the previous launch
// itself (fixed order): no extra launch and no atomics on the critical path.
__device__ __forceinline__ double fold_partials(const double *part, int count, double *red)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += kBlock) s += part[i];
    return block_sum(s, red);
}

// Tile dealt to XCD `xcd` as its j-th one.  Tiles go to the eight XCDs (blockIdx % 8) in runs
// of A.xcd_block (block-cyclic; a run as long as an eighth of the matrix gives each XCD one
// contiguous eighth).  The run length was meant to keep the x lines a tile shares with its
// +-nx*ny neighbours inside one XCD's L2; measured on the 256^3 Poisson matrix it changes neither
// the time nor (by more than 4 %) the fetched bytes -- at ~0.5 MB/us of matrix stream per XCD an
// L2 line lives ~9 us, too short for reuse across tiles -- so it only serves load balance.
// Returns -1 past the end.
__device__ __forceinline__ int xcd_tile(const CsrView &A, int xcd, int j)
{
    const int sh = A.xcd_shift;  // xcd_block = 1 << sh
    const int tile = ((j >> sh) << (sh + 3)) + (xcd << sh) + (j & (A.xcd_block - 1));
    return tile < A.ntiles ? tile : -1;
}

// number of sequence slots per XCD
__device__ __forceinline__ int xcd_slots(const CsrView &A)
{
    const int sh = A.xcd_shift;
    return ((A.ntiles + (kXcds << sh) - 1) >> (sh + 3)) << sh;
}


}  // namespace schwz
