#!/usr/bin/env python3
"""bench.py — Groth16 proofs/sec for ChaCha20-V3 single 64-byte blocks on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (witness -> quotient NTTs -> 5 MSMs -> proof assembly) over one batch
of `--batch` synthetic, independent statements per GPU, through the C-ABI (gsc_prove_raw: the binary twin of
Prove; no JSON on the timed path).  Independent proofs shard across ranks (one process per GPU, weak
scaling); the only collective is the gather of the finished proofs to rank 0 (RCCL over xGMI).

Prints ONE JSON line on rank 0 (see the harness contract): metric/value/unit..., plus
  "roofline"      : the dominant kernel (k_msm<Fp29f,false> over the Z digit tables) priced against HBM peak, timed live with HIP events;
  "cpu_baseline"  : the CPU oracle (oracle/, a port — not gnark) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_PROOF = 28_281_728                # algorithmic bytes of the whole path per ChaCha proof
MSM_Z_BYTES_PER_BASE = 64 + 32              # affine G1 base + 32-byte scalar


def golden(name):
    import lzma
    p = os.path.join(GOLDEN, name)
    return lzma.open(p + ".xz").read() if os.path.exists(p + ".xz") else open(p, "rb").read()


def pmc_traffic(batch, engine_desc):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), if they were taken
    on this configuration; None otherwise (PMC counters cannot be collected from inside the timed run)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_msm_z_pmc.json")))
        if d["config"]["batch"] == batch and ("window_z=%d " % d["config"]["window_z"]) in engine_desc:
            return d["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def synthetic_records(n, seed):
    """n x 112 B {key[32], nonce[12], counter u32 LE, input[64]} — uniform bytes from a seeded generator."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(n, 112), dtype=np.uint8).tobytes()


def shard_bounds(total, world, rank):
    """Contiguous block partition of `total` units over `world` ranks (used by --total-proofs strong-scaling runs and tests)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_proofs(dist, local: "torch.Tensor", rank, world, use_dist=None):
    """The path's only exchange: every rank's finished proofs (164 B each) to rank 0."""
    import torch
    if not (world > 1 if use_dist is None else use_dist):
        return [local]
    out = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, out, dst=0)
    return out


def _cpu_worker(args):
    n, seed = args
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import oracle as O
    cs = O.R1CS(golden("r1cs.chacha20")); pk = O.ProvingKey(golden("pk.chacha20"))
    recs = synthetic_records(n, seed)
    t = time.time()
    for i in range(n):
        r = recs[112 * i:112 * (i + 1)]
        O.prove(cs, pk, "chacha20", r[:32], r[32:44], int.from_bytes(r[44:48], "little"), r[48:112], 12345 + i, 67890 + i)
    return time.time() - t


def cpu_baseline(cores, per_core=24):
    """Oracle (CPU port of the same path) on `cores` host cores: independent single-threaded provers, one per core."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    t = time.time()
    with ctx.Pool(cores) as pool:
        busy = pool.map(_cpu_worker, [(per_core, 1000 + i) for i in range(cores)])
    wall = time.time() - t
    rate = cores * per_core / max(busy)        # excludes key decoding; all workers run concurrently
    return {"value": round(rate, 3), "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "%d ChaCha20-V3 proofs (%d per core, single-threaded oracle per core, key decode excluded); wall %.1fs" % (cores * per_core, per_core, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("GSC_BENCH_BATCH", "8192")), help="proofs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank (rehearsal of the multi-GPU code path on a one-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE JSON line (rank 0).  Native libraries print there too (libprove reports errors on stdout like the
    # reference's fmt.Println, RCCL prints a banner), so file descriptor 1 is pointed at stderr for the whole run and the line is
    # written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1); os.dup2(2, 1)
    os.environ["GSC_DEVICE"] = str(local_rank)
    os.environ.setdefault("GSC_MAX_BATCH", str(args.batch))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the prover has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import gsc_loader
    g = gsc_loader.load()
    # Z digit tables: 13-bit digits (172 GB) is the bench configuration; fall back to narrower digits if this device cannot
    # hold them (InitAlgorithm reports the failure and leaves the algorithm uninitialised, so it can simply be retried)
    pk, r1cs = golden("pk.chacha20"), golden("r1cs.chacha20")
    wanted = [os.environ["GSC_WINDOW_Z"]] if os.environ.get("GSC_WINDOW_Z") else ["13", "12", "11", "0"]
    for wz in wanted:
        os.environ["GSC_WINDOW_Z"] = wz
        if g.init_algorithm(g.CHACHA20, pk, r1cs):
            break
    else:
        raise SystemExit("InitAlgorithm failed")

    B = args.batch
    dev = torch.device("cuda", local_rank)

    # statements of every step are made before the clock starts (112 B each: "inputs resident"); output buffers are reused
    recs_of = {i: synthetic_records(B, seed=(rank << 24) + (i & 0xFFFFFF)) for i in [0x800000 + w for w in range(args.warmup)] + list(range(args.steps))}
    import numpy as np
    import threading
    from concurrent.futures import ThreadPoolExecutor
    # Two callers keep the library busy, like concurrent Prove callers do (libraries/core_test.go:44-111): while one call's batch is on the
    # GPU, the other call does its host part (native cipher, CSPRNG draws, packing).  Device work of the two calls is serialised by the
    # library, so a step still means one batch through the whole path; each caller owns a set of output buffers.
    bufs = [g.raw_buffers(B) for _ in range(2)]
    free = [threading.Event() for _ in range(2)]
    for e in free:
        e.set()

    def prove(i, slot):
        free[slot].wait(); free[slot].clear()
        pb, lb, cb = bufs[slot]
        ok = g.prove_raw_into(g.CHACHA20, recs_of[i], B, pb, lb, cb)
        return ok, g.last_msm_z_kernel(g.CHACHA20)

    def run(ids, kernel_ms):
        with ThreadPoolExecutor(2) as pool:
            futs = [pool.submit(prove, i, k % 2) for k, i in enumerate(ids)]
            for k, f in enumerate(futs):          # results are consumed in order on this thread (the only one that talks to RCCL)
                ok, km = f.result()
                if ok != B:
                    raise SystemExit("rank %d: only %d of %d proofs produced" % (rank, ok, B))
                if not (np.frombuffer(bufs[k % 2][1], dtype=np.uint32) == 164).all():
                    raise SystemExit("rank %d: incomplete proofs" % rank)
                local = torch.frombuffer(bufs[k % 2][0], dtype=torch.uint8).to(dev)
                free[k % 2].set()
                gather_proofs(dist, local, rank, world, use_dist)
                kernel_ms.append(km)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run([0x800000 + w for w in range(args.warmup)], [])
    barrier()
    t0 = time.time()
    kernel_ms = []
    run(list(range(args.steps)), kernel_ms)
    barrier()
    elapsed = time.time() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        total = world * args.steps * B
        value = total / elapsed
        ms, kb, nb = zip(*kernel_ms)
        avg_ms = sum(ms) / len(ms)
        alg_bytes = kb[-1] * nb[-1] * MSM_Z_BYTES_PER_BASE
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        line = {
            "metric": "Groth16 proofs/sec (ChaCha20-V3 1-block)", "value": round(value, 2), "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (BN254 Fr/Fp, 254-bit modular integers)",
            "data": "synthetic",
            "config": {"workload": "ChaCha20-V3 single 64-byte block, 1xMI355X per rank: batch of %d independent proofs per GPU per step, "
                                   "reference pk.chacha20/r1cs.chacha20, CSPRNG (r,s)" % B,
                       "batch_per_gpu": B, "parallelism": "proofs sharded over %d GPU(s), gather to rank 0" % world, "engine": g.describe(g.CHACHA20)},
            "roofline": {"kernel": "k_msm<Fp29f,false> (Z-table gather-accumulate)", "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc_traffic(kb[-1], g.describe(g.CHACHA20)),
                         "launch_ms": round(avg_ms, 3), "algorithmic_bytes_per_launch": alg_bytes,
                         "whole_path_frac": round(value / world * BYTES_PER_PROOF / 1e9 / HBM_PEAK_GBS, 6)},
            "stage_ms_last_step": g.last_stage_ms(g.CHACHA20),
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = args.cpu_cores or min(os.cpu_count() or 1, 16)
            try:
                line["cpu_baseline"] = cpu_baseline(cores)
            except Exception as e:      # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": cores, "kind": "port", "sample": "failed: %r" % (e,)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
