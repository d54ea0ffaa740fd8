"""numpy/scipy prototype of the sum-factorised (Kronecker) elasticity operator on a uniform box, checked against the oracle's
assembled matrix.  A = sum of Kronecker products of 1D banded matrices M (mass), K (stiffness), C (C[m][n] = int phi_m' phi_n)."""
import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import numpy as np, scipy.sparse as sp
import poroelasticity_dealii_amd as pk, oracle_py
from common import box_problem, material, csr_to_scipy

def gauss(n):
    x, w = np.polynomial.legendre.leggauss(n); return 0.5 * (x + 1), 0.5 * w

def basis(k, x):
    nodes = np.arange(k + 1) / k
    V = np.ones((k + 1, len(x))); D = np.zeros((k + 1, len(x)))
    for i in range(k + 1):
        for j in range(k + 1):
            if j != i: V[i] *= (x - nodes[j]) / (nodes[i] - nodes[j])
        for m in range(k + 1):
            if m == i: continue
            t = np.ones(len(x)) / (nodes[i] - nodes[m])
            for j in range(k + 1):
                if j != i and j != m: t *= (x - nodes[j]) / (nodes[i] - nodes[j])
            D[i] += t
    return V, D

def mats1d(k, n, h):
    x, w = gauss(k + 2); V, D = basis(k, x)
    Me = h * (V * w) @ V.T; Ke = (D * w) @ D.T / h; Ce = (D * w) @ V.T
    N = k * n + 1
    M = np.zeros((N, N)); K = np.zeros((N, N)); C = np.zeros((N, N))
    for c in range(n):
        s = slice(k * c, k * c + k + 1)
        M[s, s] += Me; K[s, s] += Ke; C[s, s] += Ce
    return M, K, C

def kron3(Tz, Ty, Tx): return sp.kron(sp.csr_matrix(Tz), sp.kron(sp.csr_matrix(Ty), sp.csr_matrix(Tx)))

def check(dim, n, deg):
    P = box_problem(dim, n, deg, bc=[]); O = oracle_py.Oracle(P, hoisted=True)
    O.fill(pk.VEC_P, 0.0); O.disp_assemble_system(True)
    A = csr_to_scipy(*O.export_csr(pk.MAT_A_U))
    m = material(); lam, G = m.lame_lambda, m.shear_G
    n = [n] * dim if np.isscalar(n) else list(n)
    mats = [mats1d(deg, n[d], 10.0 / n[d]) for d in range(dim)]
    NN = [deg * n[d] + 1 for d in range(dim)]
    def op(ts):   # ts[d] = 1D matrix in direction d; x fastest
        out = sp.csr_matrix(ts[0])
        for d in range(1, dim): out = sp.kron(sp.csr_matrix(ts[d]), out)
        return out
    M = [mm[0] for mm in mats]; K = [mm[1] for mm in mats]; C = [mm[2] for mm in mats]
    nnod = int(np.prod(NN)); B = [[None] * dim for _ in range(dim)]
    for a in range(dim):
        for b in range(dim):
            if a == b:
                blk = sp.csr_matrix((nnod, nnod))
                for d in range(dim):
                    ts = [K[e] if e == d else M[e] for e in range(dim)]
                    blk = blk + ((lam + 2 * G) if d == a else G) * op(ts)
            else:
                t1 = [M[e] for e in range(dim)]; t1[a] = C[a]; t1[b] = C[b].T
                t2 = [M[e] for e in range(dim)]; t2[a] = C[a].T; t2[b] = C[b]
                blk = lam * op(t1) + G * op(t2)
            B[a][b] = blk
    # interleave components: dof = node*dim + comp
    Pm = sp.lil_matrix((nnod * dim, nnod * dim))
    Ak = sp.bmat(B).tocsr()
    perm = np.arange(nnod * dim).reshape(dim, nnod).T.ravel()   # new index (node*dim+comp) -> old (comp*nnod+node)
    Ak = Ak[perm][:, perm]
    err = abs(Ak - A).max() / abs(A).max()
    print(f"dim={dim} n={n} deg={deg}: |A_kron - A|/|A| = {err:.2e}")
    # C + C^T = E check
    E = C[0] + C[0].T; E[0, 0] += 1; E[-1, -1] -= 1
    assert abs(E).max() < 1e-14
    O.close(); P.close()
    return err

if __name__ == "__main__":
    for cfg in [(2, 3, 1), (2, 3, 2), (2, (3, 4), 2), (3, 2, 1), (3, 2, 2), (3, (2, 3, 2), 2)]:
        assert check(*cfg) < 1e-13


def staged_check(n, deg):
    """the staged algorithm of the HIP kernel: z-stage (mz,kz,oz,wz) -> y-stage (XK,XM,XO,XD) -> x-stage, with C = O + D, C^T = -O + D"""
    dim = 3
    P = box_problem(dim, n, deg, bc=[]); O_ = oracle_py.Oracle(P, hoisted=True)
    O_.fill(pk.VEC_P, 0.0); O_.disp_assemble_system(True)
    A = csr_to_scipy(*O_.export_csr(pk.MAT_A_U))
    m = material(); lam, G = m.lame_lambda, m.shear_G
    n = [n] * dim if np.isscalar(n) else list(n)
    NN = [deg * n[d] + 1 for d in range(dim)]
    mats = [mats1d(deg, n[d], 10.0 / n[d]) for d in range(dim)]
    M = [mm[0] for mm in mats]; K = [mm[1] for mm in mats]; C = [mm[2] for mm in mats]
    Od = [c - np.diag(np.diag(c)) for c in C]; Dd = [np.diag(np.diag(c)) for c in C]
    rng = np.random.default_rng(0)
    u = rng.standard_normal((NN[2], NN[1], NN[0], 3))       # [k][j][i][comp]
    ax = lambda T, f, d: np.moveaxis(np.tensordot(T, f, axes=([1], [2 - d])), 0, 2 - d)   # apply 1D op along direction d (0=x)
    X, Y, Z = 0, 1, 2
    c1, c2, c3, c4 = -(lam + G), (lam - G), (G - lam), (lam + G)
    ux, uy, uz = u[..., 0], u[..., 1], u[..., 2]
    mz = [ax(M[Z], f, Z) for f in (ux, uy, uz)]; kz = [ax(K[Z], f, Z) for f in (ux, uy, uz)]
    oz = [ax(Od[Z], f, Z) for f in (ux, uy, uz)]; wz = [ax(Dd[Z], f, Z) for f in (ux, uy, uz)]
    My = lambda f: ax(M[Y], f, Y); Ky = lambda f: ax(K[Y], f, Y); Oy = lambda f: ax(Od[Y], f, Y); Dy = lambda f: ax(Dd[Y], f, Y)
    XK = [(lam + 2 * G) * My(mz[0]), G * My(mz[1]), G * My(mz[2])]
    XM = [G * (Ky(mz[0]) + My(kz[0])),
          (lam + 2 * G) * Ky(mz[1]) + G * My(kz[1]) + c1 * Oy(oz[2]) + c2 * Oy(wz[2]) + c3 * Dy(oz[2]) + c4 * Dy(wz[2]),
          (lam + 2 * G) * My(kz[2]) + G * Ky(mz[2]) + c1 * Oy(oz[1]) + c3 * Oy(wz[1]) + c2 * Dy(oz[1]) + c4 * Dy(wz[1])]
    XO = [c1 * Oy(mz[1]) + c2 * Dy(mz[1]) + c1 * My(oz[2]) + c2 * My(wz[2]),
          c1 * Oy(mz[0]) + c3 * Dy(mz[0]),
          c1 * My(oz[0]) + c3 * My(wz[0])]
    XD = [c3 * Oy(mz[1]) + c4 * Dy(mz[1]) + c3 * My(oz[2]) + c4 * My(wz[2]),
          c2 * Oy(mz[0]) + c4 * Dy(mz[0]),
          c2 * My(oz[0]) + c4 * My(wz[0])]
    y = np.stack([ax(K[X], XK[a], X) + ax(M[X], XM[a], X) + ax(Od[X], XO[a], X) + ax(Dd[X], XD[a], X) for a in range(3)], axis=-1)
    y0 = (A @ u.ravel()).reshape(u.shape)
    err = np.abs(y - y0).max() / np.abs(y0).max()
    print(f"staged 3D n={n} deg={deg}: err = {err:.2e}")
    # 1D element matrices (what the kernel's coefficient tables are built from)
    x_, w_ = gauss(deg + 2); V, D = basis(deg, x_)
    print(" Ce =", np.round(6 * (D * w_) @ V.T, 12).tolist(), "/6")
    O_.close(); P.close()
    return err


if __name__ == "__main__":
    for cfg in [((2, 3, 2), 1), ((2, 3, 2), 2), (3, 2)]:
        assert staged_check(*cfg) < 1e-13
