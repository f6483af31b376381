"""Which engine replica serves a small call (csrc/dispatch.hpp) — on CPU, with a stub engine: the policy the library uses for calls of
up to one 64-column batch (Algorithm::prove_batch -> ReplicaPicker) and for the micro-batcher's workers (batcher_take) is compiled with
g++ into a harness that replays the reference's concurrent single-Prove callers (libraries/core_test.go:44-111) against replicas that
only sleep.  The GPU leg (tests/test_gpu_replicas.py) shows the same on real replicas."""
import os
import subprocess

from conftest import ROOT


def test_small_calls_reach_every_replica():
    exe = os.path.join(ROOT, "build", "dispatch_check")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "native", "dispatch_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "DISPATCH-OK" in out.stdout, out.stdout + out.stderr
