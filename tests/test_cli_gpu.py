"""The driver executable poro_run (stand-in for the reference's missing Runner.cpp, code/CMakeLists.txt:8): argv[1] = parameter file
(parse_command_line.h:5-27), log lines of PoroelasticityFSS.h:325-406, output_results files (:285-290)."""
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "poroelasticity_dealii_amd", "lib", "poro_run")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def run(*args):
    r = subprocess.run([EXE, *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_domain_msh_one_step_matches_the_golden_trace(tmp_path):
    """BASELINE config 1 through the GPU path: input.data + domain.msh, 1 time step, Q1/Q1 and the reference's Q2/Q1."""
    with open(os.path.join(GOLDEN, "config1_trace.json")) as f:
        gold = json.load(f)
    for deg, key in ((1, "Q1"), (2, "Q2")):
        out = run(os.path.join(GOLDEN, "input.data"), "--mesh", os.path.join(GOLDEN, "domain.msh"), "--degree", str(deg), "--steps", "1")
        assert "starting time loop" in out and "Time: 60" in out and "Coupling iteration: 1" in out
        its = [int(m) for m in re.findall(r"pressure converged; iterations: (\d+)", out)]
        pinf = [float(m) for m in re.findall(r"Solution limits: ([0-9.eE+-]+)", out)]
        err = [float(m) for m in re.findall(r"Error: ([0-9.eE+-]+)", out)]
        assert its == [gold[key]["pressure_iterations"]]
        assert abs(pinf[0] - gold[key]["p_linf"]) <= 1e-5 * gold[key]["p_linf"]          # 6 significant digits are printed
        assert len(err) == 1 and err[0] < 1e-8                                              # quirk Q1: one fixed-stress iteration per step


def test_default_box_with_output_and_the_reference_preconditioner(tmp_path):
    """The reference's own default: `Initial refinement level = 4` box of input.data, Q2/Q1; --ssor = SolverCG + PreconditionSSOR;
    --output writes solution-NNNN.vtk after every step."""
    out_dir = tmp_path / "solution"; out_dir.mkdir()
    a = run(os.path.join(GOLDEN, "input.data"), "--steps", "2", "--ssor", "--output", str(out_dir))
    b = run(os.path.join(GOLDEN, "input.data"), "--steps", "2", "--matrix-free")
    for flag in ("--chebyshev", "--block-fdm"):                                     # the fast preconditioners give the same printed trace
        c = run(os.path.join(GOLDEN, "input.data"), "--steps", "2", "--matrix-free", flag)
        assert re.findall(r"pressure converged; iterations: (\d+)", c) == re.findall(r"pressure converged; iterations: (\d+)", b)
        lc = [float(m) for m in re.findall(r"Solution limits: ([0-9.eE+-]+)", c)]
        assert all(abs(x - y) <= 1e-5 * abs(y) for x, y in zip(lc, [float(m) for m in re.findall(r"Solution limits: ([0-9.eE+-]+)", b)]))
    c = run(os.path.join(GOLDEN, "input.data"), "--steps", "1", "--matrix-free", "--mesh", os.path.join(GOLDEN, "domain.msh"), "--degree", "1", "--chebyshev")   # general matrix-free operator
    assert len(re.findall(r"pressure converged; iterations: (\d+)", c)) == 1
    # read_mesh()'s grid with the strongest preconditioner it supports (--fastest -> the two-level form through the auxiliary box) and with --two-level by name: the same step
    lim = [float(m) for m in re.findall(r"Solution limits: ([0-9.eE+-]+)", c)]
    for flag in ("--fastest", "--two-level"):
        e = run(os.path.join(GOLDEN, "input.data"), "--steps", "1", "--matrix-free", "--mesh", os.path.join(GOLDEN, "domain.msh"), "--degree", "1", flag)
        le = [float(m) for m in re.findall(r"Solution limits: ([0-9.eE+-]+)", e)]
        assert len(le) == len(lim) >= 1 and all(abs(x - y) <= 1e-5 * abs(y) for x, y in zip(le, lim))
    pa = re.findall(r"pressure converged; iterations: (\d+)", a); pb = re.findall(r"pressure converged; iterations: (\d+)", b)
    assert pa == pb and len(pa) == 2
    la = [float(m) for m in re.findall(r"Solution limits: ([0-9.eE+-]+)", a)]; lb = [float(m) for m in re.findall(r"Solution limits: ([0-9.eE+-]+)", b)]
    assert all(abs(x - y) <= 1e-5 * abs(y) for x, y in zip(la, lb))
    files = sorted(os.listdir(out_dir))
    assert files == ["solution-0001.vtk", "solution-0002.vtk"]
    head = (out_dir / files[0]).read_text().split("\n")[:4]
    assert head[0] == "# vtk DataFile Version 3.0" and head[3] == "DATASET UNSTRUCTURED_GRID"


def test_missing_argument_message():
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 1 and "specify the file name" in r.stderr          # parse_command_line.h:9-13
