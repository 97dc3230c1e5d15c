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


@pytest.mark.gpu
def test_cpp_host_tests_pass_on_the_gpu():
    if not os.access(BINARY, os.X_OK):
        subprocess.check_call(["make", "-C", CPP], stdout=subprocess.DEVNULL)
    res = subprocess.run([BINARY], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert " 0 failed" in res.stdout
    for kind in ("BasisInverseRows", "LUDecomposition", "DenseTableau"):
        assert any(line.startswith(kind) and line.rstrip().endswith("ok") for line in res.stdout.splitlines()), res.stdout
