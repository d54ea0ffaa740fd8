"""General (unstructured-path) matrix-free operator at scale: a graded 3D box WITHOUT the box tag (rectilinear cells of different sizes, >= 1 M dofs), the kernel
BASELINE's north_star describes (element dof indices, quadrature data and material constants staged in LDS, cell loop of PoroElasticDisplacementSolver.h:206-246
applied to a vector).  Prints seconds per application and the algorithmic HBM rate  (16 N + 4 dpc n_cells + 8 * 2^dim * dim n_cells [vertex coordinates]) / time.
Usage: python tools/mfg_bench.py [cells per direction = 36] [degree = 2]     env PORO_MFG_NO_SUMFAC=1: the one-wave-per-cell kernel of round 2"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
import numpy as np
import poroelasticity_dealii_amd as pk
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 36
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P = pk.Problem.graded_box(3, [n] * 3, [10.0] * 3, deg, bench.material(), bench.BC_3D, [1.0, 0.6, -0.8])
G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
G.fill(pk.VEC_P, 0.0); G.disp_assemble_system(True)
t = G.bench_operator(pk.OP_MATRIX_FREE, int(os.environ.get("REPS", "20")))
N, nc = G.n_u, P.desc.n_cells; dpc = 3 * (deg + 1) ** 3
alg = 16.0 * N + 4.0 * dpc * nc + 8.0 * 8 * 3 * nc
rec = {"mesh": f"graded box {n}^3 cells Q{deg}, no box tag", "N_u": int(N), "n_cells": int(nc), "kernel": "k_mfg<3> (one wave per cell)" if os.environ.get("PORO_MFG_NO_SUMFAC") else "k_mfg3_sf (sum-factorised, 8 Q2 / 32 Q1 cells per workgroup)",
       "seconds_per_application": t, "cells_per_second": nc / t, "DoF_updates_per_s": N / t, "algorithmic_bytes": alg, "algorithmic_GB_per_s": alg / t / 1e9, "frac_of_8_TB_per_s": alg / t / 8e12,
       "note": "8 colour launches per application (coloured scatter, no atomics); time = HIP events over back-to-back applications incl. the memset of y"}
print(json.dumps(rec))
G.close(); P.close()
