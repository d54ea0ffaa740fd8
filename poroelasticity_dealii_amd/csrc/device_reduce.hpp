// Deterministic two-stage reductions shared by the streaming kernels (gfx950, wave64): 256-thread blocks, grids capped at kMaxPartials blocks so that
// the per-block partial sums fit one 8 KB slab which every consumer re-reduces in a fixed order (no float atomics, bitwise reproducible).
#pragma once
#include "common.hpp"

namespace poro {
namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t n, int per_thread = 4) {
  int64_t g = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return (int)g;
}
inline int reduce_grid(int64_t n) {
  int64_t g = (n + 2047) / 2048;
  if (g < 1) g = 1;
  if (g > kMaxPartials) g = kMaxPartials;
  return (int)g;
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
// block-wide sum in a fixed order (deterministic); result valid in thread 0
__device__ inline double block_sum(double v, double *sh /*[4]*/) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}
__device__ inline double block_max(double v, double *sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) r = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  __syncthreads();
  return r;
}
// block b writes its partial and zeroes the unused tail slots b+G, b+2G, ...
__device__ inline void store_partial(double *partials, double v) {
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = v;
    for (int t = blockIdx.x + gridDim.x; t < kMaxPartials; t += gridDim.x) partials[t] = 0.0;
  }
}

// sum of kMaxPartials block partials in a fixed order, broadcast to every thread of the block (identical bits in every block)
__device__ inline double sum_partials(const double *p, double *sh /*[5]*/) {
  double v = 0;
  for (int i = threadIdx.x; i < kMaxPartials; i += kBlock) v += p[i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) sh[4] = v;
  __syncthreads();
  v = sh[4];
  __syncthreads();
  return v;
}
}  // namespace
}  // namespace poro
