// One row of y = (a M + kappa K) x for the Q1 pressure space on a uniform box .
// M = Mx (x) My (x) Mz, K = Kx (x) My (x) Mz + Mx (x) Ky (x) Mz + Mx (x) My (x) Kz with the tridiagonal 1D Q1 matrices
// M1 = h/6 (1,4,1), K1 = 1/h (-1,2,-1) (boundary rows: h/6 (2,1), 1/h (1,-1)) -- what MatrixCreator::create_mass_matrix /
// create_laplace_matrix (PoroElasticPressureSolver.h:96-101) produce on such a mesh.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace poro {

template <int DIM> __device__ __forceinline__ double p_stencil_row(int n0, int n1, int n2, double h0, double h1, double h2, double a, double kappa, int64_t node,
                                                                   const double *__restrict__ x) {
  auto ld = [&](int64_t i) { return x[i]; };
  const int idx[3] = {(int)(node % n0), (int)((node / n0) % n1), (int)(node / ((int64_t)n0 * n1))};
  const int nd[3] = {n0, n1, n2};
  const double h[3] = {h0, h1, h2};
  double M1[3][3], K1[3][3];   // [direction][offset -1,0,+1]; zero where the neighbour does not exist
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const double L = idx[d] > 0 ? 1.0 : 0.0, R = idx[d] < nd[d] - 1 ? 1.0 : 0.0;
    M1[d][0] = L * h[d] / 6; M1[d][2] = R * h[d] / 6; M1[d][1] = (L + R) * h[d] / 3;
    K1[d][0] = -L / h[d]; K1[d][2] = -R / h[d]; K1[d][1] = (L + R) / h[d];
  }
  double acc = 0;
  if constexpr (DIM == 2) {
#pragma unroll
    for (int dj = 0; dj < 3; ++dj)
#pragma unroll
      for (int di = 0; di < 3; ++di) {
        const double wM = M1[0][di] * M1[1][dj], wK = K1[0][di] * M1[1][dj] + M1[0][di] * K1[1][dj];
        const double w = a * wM + kappa * wK;
        if (w != 0.0) acc = fma(w, ld(node + (di - 1) + (int64_t)(dj - 1) * n0), acc);
      }
  } else {
#pragma unroll
    for (int dk = 0; dk < 3; ++dk)
#pragma unroll
      for (int dj = 0; dj < 3; ++dj) {
        const double mm = M1[1][dj] * M1[2][dk], km = K1[1][dj] * M1[2][dk] + M1[1][dj] * K1[2][dk];
#pragma unroll
        for (int di = 0; di < 3; ++di) {
          const double w = a * (M1[0][di] * mm) + kappa * (K1[0][di] * mm + M1[0][di] * km);
          if (w != 0.0) acc = fma(w, ld(node + (di - 1) + ((int64_t)(dj - 1) + (int64_t)(dk - 1) * n1) * n0), acc);
        }
      }
  }
  return acc;
}

}  // namespace poro
