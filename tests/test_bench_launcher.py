"""bench.py's rank launcher on CPU: `bench.py --gpus 2` run directly (the way the driver runs it) must start two rank processes
before anything touches torch/HIP, rendezvous on 127.0.0.1, gather to rank 0 and report n_gpus = 2.  The prover is replaced by
bench.py's --stub-prover (gloo backend, no GPU): what is under test is the launch path the round-1 bench did not have."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus_2_spawns_two_ranks_and_reports_them():
    out = subprocess.check_output([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64", "--stub-prover"],
                                  env=_env(), timeout=300, stderr=subprocess.DEVNULL).decode()
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1                                  # exactly one JSON line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["batch_per_gpu"] == 64 and line["unit"] == "proofs/s"
    assert abs(line["value"] - 2 * 3 * 64 / (line["ms_per_step"] * 3 / 1e3)) / line["value"] < 0.01     # whole-job aggregate over both ranks


def test_world_size_must_match_gpus():
    env = dict(_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-prover", "--steps", "1", "--warmup", "0", "--batch", "64"], env=env, capture_output=True, timeout=120)
    assert p.returncode != 0 and b"WORLD_SIZE=1 but --gpus 2" in p.stderr


def test_failing_rank_fails_the_launcher():
    # without the stub there is no GPU here: every rank exits non-zero and the launcher must say so instead of printing a line
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "64"], env=_env(), capture_output=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        return
    assert p.returncode != 0 and not p.stdout.strip()


def test_in_library_mode_is_one_process_for_all_devices():
    # `bench.py --gpus N --in-library`: ONE process drives N engine replicas through GSC_DEVICES (what a single Go / node FFI host does);
    # no rank processes, no torch.distributed; one call of N x batch statements per step; n_gpus = N in the line.  (Stub prover: the
    # plumbing only; the real thing is rehearsed on one GPU as --devices 0,0 and kept under profiles/.)
    out = subprocess.check_output([sys.executable, BENCH, "--gpus", "2", "--in-library", "--devices", "0,0", "--steps", "3", "--warmup", "1", "--batch", "64", "--stub-prover"],
                                  env=_env(), timeout=300, stderr=subprocess.DEVNULL).decode()
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["batch_per_gpu"] == 64 and "in-library replicas" in line["config"]["parallelism"] and "GSC_DEVICES=0,0" in line["config"]["parallelism"]
    assert abs(line["value"] - 2 * 3 * 64 / (line["ms_per_step"] * 3 / 1e3)) / line["value"] < 0.01
    # the device list must match --gpus; a distributed launch of the in-library mode is refused
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--in-library", "--devices", "0", "--stub-prover"], env=_env(), capture_output=True, timeout=120)
    assert p.returncode != 0 and b"--devices lists 1 devices" in p.stderr
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--in-library", "--stub-prover"], env=dict(_env(), RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"), capture_output=True, timeout=120)
    assert p.returncode != 0 and b"one process for all GPUs" in p.stderr


def test_eight_ranks_and_eight_in_library_replicas():
    # the driver's scaling run is N = 1, 2, 4, 8 on one node: the same two launch paths at N = 8 (stub prover, gloo; 8 rank processes of
    # this script with one torch import each, so give it time).  Every rank's shard enters the aggregate exactly once.
    env = dict(_env(), MASTER_PORT=str(29650 + os.getpid() % 200))
    out = subprocess.check_output([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "64", "--stub-prover"], env=env, timeout=900, stderr=subprocess.DEVNULL).decode()
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 64
    assert "8 GPU(s), one process per GPU" in line["config"]["parallelism"]
    assert abs(line["value"] - 8 * 2 * 64 / (line["ms_per_step"] * 2 / 1e3)) / line["value"] < 0.01
    out = subprocess.check_output([sys.executable, BENCH, "--gpus", "8", "--in-library", "--steps", "2", "--warmup", "1", "--batch", "64", "--stub-prover"], env=_env(), timeout=300, stderr=subprocess.DEVNULL).decode()
    line = json.loads([l for l in out.splitlines() if l.strip()][-1])
    assert line["n_gpus"] == 8 and "GSC_DEVICES=0,1,2,3,4,5,6,7" in line["config"]["parallelism"] and "one call of 512 statements per step" in line["config"]["parallelism"]
    assert abs(line["value"] - 8 * 2 * 64 / (line["ms_per_step"] * 2 / 1e3)) / line["value"] < 0.01
