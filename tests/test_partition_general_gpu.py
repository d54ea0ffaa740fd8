"""General partition on the device: ranks share GPU 0 and exchange through the host-staged callback communicator (gloo); with RCCL the same pack ->
grouped send/recv -> ordered-sum sequence runs on the compute stream.  Assembled (CSR) and matrix-free (general cell-loop kernel) operators, Jacobi- and
Chebyshev-preconditioned CG, on the Gmsh mesh and on boxes cut along the Morton curve; compared with the single-rank oracle."""
import pytest

from test_partition_general_cpu import check_against_single_rank, run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,mesh,deg,backend", [(2, "gmsh", 2, "hip_csr"), (3, "gmsh", 2, "hip_mf"), (3, "gmsh", 1, "hip_mf_cheb"), (4, "box:4,4,4", 2, "hip_mf"),
                                                    (3, "box:4,5,3", 1, "hip_csr"), (2, "box:7,6", 2, "hip_mf_cheb"),
                                                    # hanging-node meshes through the partition: ghost masters, rows folded before the interface sums
                                                    (2, "refined:4,4", 2, "hip_csr"), (3, "refined:4,4,4", 2, "hip_mf"), (3, "refined:6,5", 1, "hip_mf_cheb"), (4, "refined:4,4,4", 1, "hip_csr")])
def test_general_partition_on_one_gpu(tmp_path, world, mesh, deg, backend):
    R = run_ranks(tmp_path, world, mesh, deg, backend)
    check_against_single_rank(R, mesh, deg, tol_u=1e-8)
