"""Operator micro-benchmark: mean seconds per y = A_u x (HIP events around `reps` back-to-back launches)."""
import json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import poroelasticity_dealii_amd as pk
from bench import material, BC_3D, bytes_per_apply

def run(dim, n, deg, mode, reps=50):
    t0 = time.time()
    cells = [n] * dim if isinstance(n, int) else list(n)
    P = pk.Problem.box(dim, cells, [10.0 * c / cells[0] for c in cells], deg, material(), BC_3D[:2 * dim])
    t1 = time.time()
    G = pk.Context(P, 0, mode)
    t2 = time.time()
    G.fill(pk.VEC_P, 10e6); G.disp_assemble_system(True)
    t3 = time.time()
    sec = G.bench_operator(mode, reps)
    if mode == pk.OP_MATRIX_FREE and dim == 3:       # a few Chebyshev-CG iterations so that the fused kernels (k_kron3_*_cheb) appear in counter passes of this script
        G.disp_solve(abs_tol=1e-12, rel_tol=1e-30, max_iter=7, prec=pk.PREC_CHEBYSHEV, poly_degree=6)   # 1 + 2 + 4 iterations enqueued = the cap: no launch behind the end of the solve
    nu, nc = P.desc.n_dofs_u, P.desc.n_cells
    if mode == pk.OP_MATRIX_FREE:
        b = bytes_per_apply(dim, deg, nu, nc, "matrix_free")
    else:
        rp, col, val = None, None, None
        import ctypes as C
        nr, nnz = C.c_int64(), C.c_int64(); G.L.poro_export_csr_size(G.ptr, pk.MAT_A_U, C.byref(nr), C.byref(nnz)); b = 12.0 * nnz.value + 24.0 * nu
    print(json.dumps({"dim": dim, "n": n, "deg": deg, "mode": "mf" if mode else "csr", "N_u": nu, "us_per_apply": sec * 1e6, "GBs_algorithmic": b / sec / 1e9,
                      "GDoF_per_s": nu / sec / 1e9, "host_mesh_s": t1 - t0, "ctx_s": t2 - t1, "assemble_s": t3 - t2}), flush=True)
    G.close(); P.close()

if __name__ == "__main__":
    for spec in sys.argv[1:]:
        dim, n, deg, mode = spec.split(",")
        run(int(dim), int(n) if "x" not in n else [int(v) for v in n.split("x")], int(deg), pk.OP_MATRIX_FREE if mode == "mf" else pk.OP_CSR)     # n: cells per direction, or nx x ny x nz (a slab of a partitioned box)
