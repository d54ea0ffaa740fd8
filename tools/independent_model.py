"""A second, independent restatement of the reference's time step on uniform boxes - numpy / scipy only, no code shared with oracle/ or the product.

Where the oracle (and the reference) loop over cells and quadrature points, this model builds every operator of the fixed-stress loop from 1D matrices:
  A_u  = Kronecker sums of the 1D FE_Q(k) mass / stiffness / first-derivative matrices (blocks A_ab of the elasticity operator),
  M_p, K_p = Kronecker sums of the 1D FE_Q(1) matrices,  coupling  b_u = alpha G p  with the rectangular Kronecker operators G_c = (x)_d (d == c ? D_d : N_d),
  projection right-hand sides r_cc = G_c^T u_c,  well source = the reference's QGauss(2) quadrature of the cylinder indicator (right_hand_side.h:99-116),
and solves every sub-problem with a sparse DIRECT solver.  It then walks PoroElasticProblem<dim>::run() (PoroelasticityFSS.h:308-407, quirks Q1 / Q2 / Q7).
Used to generate tests/golden/independent_traces.json (python tools/independent_model.py --write), against which both the oracle (CPU suite) and the HIP
path (GPU suite) are checked: a cross-implementation golden instead of the oracle's own output."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

INPUT = dict(E=1.4e10, nu=0.3, alpha=0.9, poro=0.3, f_comp=5.8e-10, perm_mD=10.0, visc=1e-3, r_well=1.0, flow_rate=1e-5, p_init=10e6, dt=60.0)   # reference input.data:13-40
BCS = {2: [(0, 0, 0.0), (1, 0, -1e-5), (2, 1, 0.0), (3, 1, -1e-5)], 3: [(0, 0, 0.0), (1, 0, -1e-5), (2, 1, 0.0), (3, 1, -1e-5), (4, 2, 0.0), (5, 2, -1e-5)]}


def derived():
    E, nu, al, po, cf = INPUT["E"], INPUT["nu"], INPUT["alpha"], INPUT["poro"], INPUT["f_comp"]      # InputDataPoroel.h:213-222
    lam = E * nu / ((1 + nu) * (1 - 2 * nu)); G = E / (2 * (1 + nu)); K = lam + 2.0 / 3.0 * G
    Ks = K / (1 - al); N = Ks / (al - po); M = (N / cf) / (N * po + 1 / cf)
    return dict(lam=lam, G=G, K=K, M=M, kmu=INPUT["perm_mD"] * 9.869233e-16 / INPUT["visc"])


def lagrange(k, x):
    nodes = np.arange(k + 1) / k
    V = np.ones((k + 1, len(x))); D = np.zeros((k + 1, len(x)))
    for i in range(k + 1):
        for j in range(k + 1):
            if j != i:
                V[i] *= (x - nodes[j]) / (nodes[i] - nodes[j])
        for m in range(k + 1):
            if m != i:
                t = np.ones(len(x)) / (nodes[i] - nodes[m])
                for j in range(k + 1):
                    if j not in (i, m):
                        t *= (x - nodes[j]) / (nodes[i] - nodes[j])
                D[i] += t
    return V, D


def one_d(k, n, h):
    """assembled 1D matrices on n cells: M, K, C (C[m][n] = int phi_m' phi_n) of FE_Q(k); N, Dm (FE_Q(k) x FE_Q(1): int phi psi, int phi' psi); FE_Q(1) mass / stiffness"""
    xg, wg = np.polynomial.legendre.leggauss(4); xg = 0.5 * (xg + 1); wg = 0.5 * wg
    V, D = lagrange(k, xg); P, dP = lagrange(1, xg)
    nu, npn = k * n + 1, n + 1
    M = np.zeros((nu, nu)); K = np.zeros((nu, nu)); C = np.zeros((nu, nu)); N = np.zeros((nu, npn)); Dm = np.zeros((nu, npn)); Mp = np.zeros((npn, npn)); Kp = np.zeros((npn, npn))
    for c in range(n):
        su, spn = slice(k * c, k * c + k + 1), slice(c, c + 2)
        M[su, su] += h * (V * wg) @ V.T; K[su, su] += (D * wg) @ D.T / h; C[su, su] += (D * wg) @ V.T
        N[su, spn] += h * (V * wg) @ P.T; Dm[su, spn] += (D * wg) @ P.T
        Mp[spn, spn] += h * (P * wg) @ P.T; Kp[spn, spn] += (dP * wg) @ dP.T / h
    return dict(M=M, K=K, C=C, N=N, D=Dm, Mp=Mp, Kp=Kp)


def kron(mats):                                    # x fastest: the LAST factor of the Kronecker product is direction 0
    out = sp.csr_matrix(mats[0])
    for m in mats[1:]:
        out = sp.kron(sp.csr_matrix(m), out, format="csr")
    return out


class Model:
    def __init__(self, dim, n, deg, size=10.0):
        self.dim, self.deg = dim, deg
        self.n = [n] * dim if np.isscalar(n) else list(n)
        self.h = [size / m for m in self.n]
        self.c = derived()
        one = [one_d(deg, self.n[d], self.h[d]) for d in range(dim)]
        lam, G = self.c["lam"], self.c["G"]
        nnode = int(np.prod([deg * m + 1 for m in self.n])); self.nnode = nnode
        blocks = [[None] * dim for _ in range(dim)]
        for a in range(dim):
            for b in range(dim):
                if a == b:
                    blk = sp.csr_matrix((nnode, nnode))
                    for d in range(dim):
                        blk = blk + ((lam + 2 * G) if d == a else G) * kron([one[e]["K"] if e == d else one[e]["M"] for e in range(dim)])
                else:
                    t1 = [one[e]["M"] for e in range(dim)]; t1[a] = one[a]["C"]; t1[b] = one[b]["C"].T
                    t2 = [one[e]["M"] for e in range(dim)]; t2[a] = one[a]["C"].T; t2[b] = one[b]["C"]
                    blk = lam * kron(t1) + G * kron(t2)
                blocks[a][b] = blk
        perm = np.arange(nnode * dim).reshape(dim, nnode).T.ravel()          # dof = node * dim + component
        self.A = sp.bmat(blocks).tocsr()[perm][:, perm]
        self.Mp = kron([one[e]["Mp"] for e in range(dim)])
        self.Kp = sum(kron([one[e]["Kp"] if e == d else one[e]["Mp"] for e in range(dim)]) for d in range(dim))
        self.Gc = [kron([one[e]["D"] if e == c else one[e]["N"] for e in range(dim)]) for c in range(dim)]   # (nnode x n_p): int d_c phi_s psi_t
        self.n_p = self.Mp.shape[0]
        # Dirichlet conditions on whole faces, first condition wins (PoroElasticDisplacementSolver.h:117-136)
        nn = [deg * m + 1 for m in self.n]
        idx = np.indices(nn[::-1]).reshape(dim, -1)[::-1]                     # idx[d] = lattice index of every node in direction d (x fastest)
        self.fixed = {}
        for label, comp, val in BCS[dim]:
            d, side = label // 2, label % 2
            for node in np.where(idx[d] == (nn[d] - 1 if side else 0))[0]:
                self.fixed.setdefault(int(node) * dim + comp, val)
        self.fd = np.array(sorted(self.fixed)); self.fv = np.array([self.fixed[i] for i in self.fd])
        self.free = np.setdiff1d(np.arange(nnode * dim), self.fd)
        self.Aff = spla.splu(self.A[self.free][:, self.free].tocsc())
        self.lift = self.A[self.free][:, self.fd] @ self.fv
        self.Mlu = spla.splu(self.Mp.tocsc())
        self.source = self.well_source()

    def well_source(self):                         # VectorTools::create_right_hand_side, QGauss(2), SinglePhaseWell (right_hand_side.h:99-116)
        dim = self.dim; g = 0.5 + np.array([-0.5, 0.5]) / np.sqrt(3.0)
        npn = [m + 1 for m in self.n]
        q = np.zeros(self.n_p)
        cells = np.indices(self.n[::-1]).reshape(dim, -1)[::-1]
        s_val = -INPUT["flow_rate"] / (3.1415926 * INPUT["r_well"] ** 2)
        jxw = np.prod(self.h) / 2 ** dim
        for qi in range(2 ** dim):
            xi = [g[(qi >> d) & 1] for d in range(dim)]
            X = [-5.0 + self.h[d] * (cells[d] + xi[d]) for d in range(dim)]
            inside = (X[0] ** 2 + X[1] ** 2) <= INPUT["r_well"] ** 2
            for v in range(2 ** dim):
                w = np.ones(cells.shape[1]); node = np.zeros(cells.shape[1], np.int64); stride = 1
                for d in range(dim):
                    b = (v >> d) & 1
                    w = w * (xi[d] if b else 1 - xi[d]); node = node + (cells[d] + b) * stride; stride *= npn[d]
                np.add.at(q, node, w * inside * s_val * jxw)
        return q

    def solve_u(self, p):
        b = np.zeros(self.nnode * self.dim)
        for c in range(self.dim):
            b[c::self.dim] = INPUT["alpha"] * (self.Gc[c] @ p)
        u = np.zeros_like(b); u[self.fd] = self.fv
        u[self.free] = self.Aff.solve(b[self.free] - self.lift)
        return u

    def normal_strains(self, u):
        return [self.Mlu.solve(self.Gc[c].T @ u[c::self.dim]) for c in range(self.dim)]

    def run(self, n_steps, fss_tol=1e-8, pressure_tol=1e-8, max_fss=50, max_pres=50):
        c, dt, al = self.c, INPUT["dt"], INPUT["alpha"]
        p = np.full(self.n_p, INPUT["p_init"])                                  # :311
        u = self.solve_u(p)                                                     # :312-313
        eps = self.normal_strains(u); ev = sum(eps); ev0 = ev.copy()            # :315-317
        J = spla.splu((self.Mp / (c["M"] * dt) + c["kmu"] * self.Kp).tocsc())
        res = lambda: -(self.Mp @ ((ev - ev0) * (al / dt) + (p - p_old) / (c["M"] * dt)) + c["kmu"] * (self.Kp @ p) + self.source)
        trace = []
        for step in range(1, n_steps + 1):
            p_old = p.copy()                                                    # :342
            err, fss = 2 * pressure_tol, 0
            while fss < max_fss and err > fss_tol:                              # :347-348
                fss += 1; dp = np.zeros(self.n_p); pit = 0
                while pit < max_pres:                                           # :358
                    pit += 1
                    ev = ev + (al / c["K"]) * dp                                # :360 (update_volumetric_strain with the last solution_update)
                    R = res(); err = np.linalg.norm(R)
                    if err < pressure_tol:
                        break
                    dp = J.solve(R); p = p + dp                                 # :377-379
                u = self.solve_u(p)                                             # :395-396
                eps = self.normal_strains(u)                                    # :398 (get_volumetric_strain() stays commented out, :399)
                err = np.linalg.norm(res())                                     # :402-405
                trace.append(dict(step=step, fss_iteration=fss, pressure_iterations=pit - 1, p_linf=float(np.abs(p).max()), fss_error=float(err)))
        return dict(trace=trace, u_l2=float(np.linalg.norm(u)), p_l2=float(np.linalg.norm(p)), ev_l2=float(np.linalg.norm(ev)),
                    u_probe=[float(v) for v in u[:: max(1, len(u) // 16)][:16]], p_probe=[float(v) for v in p[:: max(1, len(p) // 16)][:16]],
                    eps_probe=[float(v) for v in eps[0][:: max(1, len(p) // 8)][:8]])


CASES = {"2d_q2_4": (2, 4, 2, 3), "2d_q1_6": (2, 6, 1, 2), "2d_q2_7x5": (2, (7, 5), 2, 2), "3d_q1_8": (3, 8, 1, 2), "3d_q2_4": (3, 4, 2, 2)}

if __name__ == "__main__":
    out = {}
    for key, (dim, n, deg, steps) in CASES.items():
        out[key] = dict(dim=dim, n=n, degree=deg, steps=steps, **Model(dim, n, deg).run(steps))
        print(key, [(t["pressure_iterations"], t["fss_iteration"]) for t in out[key]["trace"]], out[key]["u_l2"], out[key]["p_l2"])
    if "--write" in sys.argv:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "independent_traces.json")
        json.dump(out, open(path, "w"), indent=1)
        print("wrote", path)
