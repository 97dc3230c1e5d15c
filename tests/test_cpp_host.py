"""The C++ host side above the C ABI (include/relp.hpp, the mirror of the reference's `MatrixData` / `Tableau` /
`PivotRule` / `OptimizationResult`): tests/cpp/test_tableau.cpp restates the reference's own unit tests of the
pivot path in C++ (tableau/mod.rs, strategy/pivot_rule.rs, two_phase/mod.rs, src/tests/problem_{1,2}.rs).
CPU tier: the header and the test program compile and link against the library.  GPU tier: the program runs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BINARY = os.path.join(CPP, "test_tableau")


def test_cpp_host_header_compiles_and_links():
    import rust_lp_amd  # noqa: F401  (builds the library when it is missing)
    from rust_lp_amd import engine
    engine.load_library()
    subprocess.check_call(["make", "-C", CPP], stdout=subprocess.DEVNULL)
    assert os.access(BINARY, os.X_OK)
    # C++17 only on the host side: no HIP header is pulled in through relp.hpp
    src = open(os.path.join(ROOT, "include", "relp.hpp")).read()
    assert "hip/" not in src


def test_host_lu_factorisation_fusion_and_packing():
    """tests/cpp/test_lu_host.cpp (no GPU): P B Q = L U on the reference's factorisation cases (decomposition/mod.rs:301-491)
    and seeded LP-like bases, FTRAN / BTRAN against dense elimination, and the fused, ELL-packed schedules executed pass by pass
    like the device does, with pivots masked the way a Forrest-Tomlin update masks them."""
    subprocess.check_call(["make", "-C", CPP, "test_lu_host"], stdout=subprocess.DEVNULL)
    res = subprocess.run([os.path.join(CPP, "test_lu_host")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]
    assert " 0 failed" in res.stdout


def test_device_lu_factorisation_as_a_host_model():
    """tests/cpp/test_lu_device_model.cpp (no GPU): the device-side LU factorisation (relp_lu_factor_core.h)
    compiled for the host -- the code the kernel k_lu_factor runs, its parallel loops serial -- on the reference's
    factorisation cases (decomposition/mod.rs:301-491), LP-like random bases and providers with slack, bound and artificial
    columns: P B Q = L U, row and column views, FTRAN / BTRAN against dense solves and against lu_factor."""
    subprocess.check_call(["make", "-C", CPP, "test_lu_device_model"], stdout=subprocess.DEVNULL)
    res = subprocess.run([os.path.join(CPP, "test_lu_device_model")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]
    assert " 0 failed" in res.stdout


@pytest.mark.gpu
def test_cpp_host_tests_pass_on_the_gpu():
    if not os.access(BINARY, os.X_OK):
        subprocess.check_call(["make", "-C", CPP], stdout=subprocess.DEVNULL)
    res = subprocess.run([BINARY], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert " 0 failed" in res.stdout
    for kind in ("BasisInverseRows", "LUDecomposition", "DenseTableau"):
        assert any(line.startswith(kind) and line.rstrip().endswith("ok") for line in res.stdout.splitlines()), res.stdout
