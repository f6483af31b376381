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
    # the harness drives real threads against wall-clock windows of 150-300 us: on a loaded host a descheduled caller can miss one, so the
    # whole check gets three attempts (the logical checks — nobody lost, every replica serves, a lone caller never waited for — hold every time)
    for attempt in range(3):
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        if out.returncode == 0 and "DISPATCH-OK" in out.stdout:
            break
    assert out.returncode == 0 and "DISPATCH-OK" in out.stdout, out.stdout + out.stderr
