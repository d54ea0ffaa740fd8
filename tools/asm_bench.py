"""K-asm-u / K-spmv micro-benchmark in assembled-CSR mode: seconds per matrix assembly (coloured per-cell kernel, CSR scatter) and per SpMV."""
import json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import poroelasticity_dealii_amd as pk
from bench import material, BC_3D
import ctypes as C

for spec in sys.argv[1:]:
    dim, n, deg = map(int, spec.split(","))
    P = pk.Problem.box(dim, [n] * dim, [10.0] * dim, deg, material(), BC_3D[:2 * dim])
    G = pk.Context(P, 0, pk.OP_CSR)
    G.fill(pk.VEC_P, 10e6)
    G.disp_assemble_system(True)                      # warm-up (pattern is built at context creation)
    G.timers_reset()
    reps = 5 if n < 64 else 1
    for _ in range(reps):
        G.disp_assemble_system(True)
    G.timers_enable(False)
    t_asm, n_asm = G.timer("assemble_u_matrix")
    nr, nnz = C.c_int64(), C.c_int64(); G.L.poro_export_csr_size(G.ptr, pk.MAT_A_U, C.byref(nr), C.byref(nnz))
    sec_spmv = G.bench_operator(pk.OP_CSR, 20)
    nu, nc = P.desc.n_dofs_u, P.desc.n_cells
    dpc = dim * (deg + 1) ** dim
    b_asm = 8.0 * nnz.value + 4.0 * dpc * nc          # CSR values written + element dof indices
    b_spmv = 12.0 * nnz.value + 24.0 * nu
    print(json.dumps({"dim": dim, "n": n, "deg": deg, "N_u": nu, "nnz": nnz.value, "assemble_ms": 1e3 * t_asm / max(n_asm, 1), "assemble_GBs_algorithmic": b_asm / (t_asm / max(n_asm, 1)) / 1e9,
                      "assemble_rows_per_s": nu / (t_asm / max(n_asm, 1)), "spmv_us": 1e6 * sec_spmv, "spmv_GBs_algorithmic": b_spmv / sec_spmv / 1e9}), flush=True)
    G.close(); P.close()
