"""Host-only check (no GPU): the base order of the evaluation-form quotient (kernels.hpp quot_digit_index) is a permutation and matches
the thread-to-element arithmetic of the last quotient kernel; window counts of the signed-digit recoding.  Compiled with hipcc (the header
pulls in the HIP runtime types), run on the CPU."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_quot_digit_index_is_the_kernels_permutation(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "quot_index_check")
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "native", "quot_index_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "QUOT-INDEX-OK" in out.stdout, out.stdout + out.stderr
