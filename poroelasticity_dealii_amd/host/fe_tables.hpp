// Reference-cell tables for FE_Q(k) / FESystem(FE_Q(k),dim) / MappingQ1 with QGauss(n),
// i.e. the numbers the reference reads out of deal.II FEValues
// (PoroElasticDisplacementSolver.h:159-173, StrainProjector.h:126-134,
//  PoroElasticPressureSolver.h:96-101).  Conventions: include/poroel_hip.h.
#pragma once
#include <array>
#include <cmath>
#include <stdexcept>
#include <vector>
#include "../../include/poroel_hip.h"

namespace poro_host {

// n-point Gauss-Legendre rule on [0,1]; closed forms for n <= 3 (QGauss<1>(n)).
inline void gauss01(int n, std::vector<double> &x, std::vector<double> &w) {
  x.assign(n, 0.0); w.assign(n, 0.0);
  if (n == 1) { x[0] = 0.5; w[0] = 1.0; }
  else if (n == 2) {
    const double a = 0.5 / std::sqrt(3.0);
    x = {0.5 - a, 0.5 + a}; w = {0.5, 0.5};
  } else if (n == 3) {
    const double a = 0.5 * std::sqrt(3.0 / 5.0);
    x = {0.5 - a, 0.5, 0.5 + a}; w = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
  } else throw std::invalid_argument("gauss01: n in 1..3");
}

// 1D Lagrange basis of degree k on equidistant nodes of [0,1] (FE_Q(k), k<=2).
inline void lagrange1d(int k, double x, double *v, double *d) {
  if (k == 1) { v[0] = 1 - x; v[1] = x; d[0] = -1; d[1] = 1; }
  else if (k == 2) {
    v[0] = 2 * (x - 0.5) * (x - 1); v[1] = 4 * x * (1 - x); v[2] = 2 * x * (x - 0.5);
    d[0] = 4 * x - 3;               d[1] = 4 - 8 * x;       d[2] = 4 * x - 1;
  } else throw std::invalid_argument("lagrange1d: k in 1..2");
}

inline int ipow(int b, int e) { int r = 1; while (e--) r *= b; return r; }

// tensor-product shape values / gradients of Q_k at a reference point
inline void shape_at(int dim, int k, const double *xi, double *val, double *grad /*[ns][dim]*/) {
  double v1[3][3], d1[3][3];
  for (int d = 0; d < dim; ++d) lagrange1d(k, xi[d], v1[d], d1[d]);
  const int n1 = k + 1, ns = ipow(n1, dim);
  for (int s = 0; s < ns; ++s) {
    int idx[3] = {s % n1, (s / n1) % n1, s / (n1 * n1)};
    double v = 1;
    for (int d = 0; d < dim; ++d) v *= v1[d][idx[d]];
    val[s] = v;
    for (int g = 0; g < dim; ++g) {
      double t = 1;
      for (int d = 0; d < dim; ++d) t *= (d == g) ? d1[d][idx[d]] : v1[d][idx[d]];
      grad[s * dim + g] = t;
    }
  }
}

struct FETables {
  int dim = 0, k_u = 0;
  std::vector<double> w_qu, w_qp, w_qf, xi_qu, xi_qp;
  std::vector<double> u_qu, du_qu, du_qp, q1_qu, dq1_qu, q1_qp, dq1_qp, u_qf, dq1_qf;
  poro_fe_tables c{};

  void build(int dim_, int k_u_) {
    dim = dim_; k_u = k_u_;
    const int nqu1 = k_u + 1, nqp1 = 2;
    const int nq_u = ipow(nqu1, dim), nq_p = ipow(nqp1, dim), nq_f = ipow(nqu1, dim - 1);
    const int ns_u = ipow(k_u + 1, dim), ns_p = ipow(2, dim);
    auto volume_rule = [&](int n1, std::vector<double> &xi, std::vector<double> &w) {
      std::vector<double> x1, w1; gauss01(n1, x1, w1);
      const int nq = ipow(n1, dim);
      xi.assign(nq * dim, 0); w.assign(nq, 1);
      for (int q = 0; q < nq; ++q) {
        int idx[3] = {q % n1, (q / n1) % n1, q / (n1 * n1)};
        for (int d = 0; d < dim; ++d) { xi[q * dim + d] = x1[idx[d]]; w[q] *= w1[idx[d]]; }
      }
    };
    volume_rule(nqu1, xi_qu, w_qu);
    volume_rule(nqp1, xi_qp, w_qp);
    u_qu.resize(nq_u * ns_u); du_qu.resize(nq_u * ns_u * dim);
    q1_qu.resize(nq_u * ns_p); dq1_qu.resize(nq_u * ns_p * dim);
    for (int q = 0; q < nq_u; ++q) {
      shape_at(dim, k_u, &xi_qu[q * dim], &u_qu[q * ns_u], &du_qu[q * ns_u * dim]);
      shape_at(dim, 1, &xi_qu[q * dim], &q1_qu[q * ns_p], &dq1_qu[q * ns_p * dim]);
    }
    du_qp.resize(nq_p * ns_u * dim); q1_qp.resize(nq_p * ns_p); dq1_qp.resize(nq_p * ns_p * dim);
    std::vector<double> tmp(ns_u);
    for (int q = 0; q < nq_p; ++q) {
      shape_at(dim, k_u, &xi_qp[q * dim], tmp.data(), &du_qp[q * ns_u * dim]);
      shape_at(dim, 1, &xi_qp[q * dim], &q1_qp[q * ns_p], &dq1_qp[q * ns_p * dim]);
    }
    // faces: f = 2*normal + side; face points tensorised over the remaining directions (low dir fastest)
    std::vector<double> x1, w1; gauss01(nqu1, x1, w1);
    w_qf.assign(nq_f, 1);
    u_qf.assign(2 * dim * nq_f * ns_u, 0); dq1_qf.assign(2 * dim * nq_f * ns_p * dim, 0);
    std::vector<double> gtmp(ns_u * dim), vtmp(ns_p);
    for (int f = 0; f < 2 * dim; ++f) {
      const int nd = f / 2, side = f % 2;
      for (int q = 0; q < nq_f; ++q) {
        double xi[3] = {0, 0, 0}; int rem = q; double w = 1;
        for (int d = 0; d < dim; ++d) {
          if (d == nd) { xi[d] = side; continue; }
          const int i = rem % nqu1; rem /= nqu1;
          xi[d] = x1[i]; w *= w1[i];
        }
        w_qf[q] = w;
        shape_at(dim, k_u, xi, &u_qf[(f * nq_f + q) * ns_u], gtmp.data());
        shape_at(dim, 1, xi, vtmp.data(), &dq1_qf[(f * nq_f + q) * ns_p * dim]);
      }
    }
    c.nq_u = nq_u; c.nq_p = nq_p; c.nq_f = nq_f; c.ns_u = ns_u; c.ns_p = ns_p;
    c.w_qu = w_qu.data(); c.w_qp = w_qp.data(); c.w_qf = w_qf.data();
    c.u_qu = u_qu.data(); c.du_qu = du_qu.data(); c.du_qp = du_qp.data();
    c.q1_qu = q1_qu.data(); c.dq1_qu = dq1_qu.data(); c.q1_qp = q1_qp.data(); c.dq1_qp = dq1_qp.data();
    c.u_qf = u_qf.data(); c.dq1_qf = dq1_qf.data();
  }
};

}  // namespace poro_host
