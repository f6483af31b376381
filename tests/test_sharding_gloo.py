"""N>1 path on CPU: two processes over gloo exercise bench.py's sharding and its only collective (the gather of
finished proofs to rank 0).  The prover itself is GPU-only, so the per-rank "proofs" here are stand-in byte
strings derived from the rank's own synthetic inputs — what is under test is the partition + gather plumbing."""
import hashlib
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_proofs(records: bytes, n: int) -> bytes:
    return b"".join(hashlib.sha256(records[112 * i:112 * (i + 1)]).digest() * 5 + bytes(4) for i in range(n))   # 164 B each


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard_bounds(total, world, rank)
    recs = bench.synthetic_records(total, seed=42)[112 * lo:112 * hi]      # every rank derives the same global input set
    local = torch.frombuffer(bytearray(_fake_proofs(recs, hi - lo)), dtype=torch.uint8)
    # ranks may own different numbers of proofs: pad to the largest shard for the fixed-size gather
    maxn = max(b - a for a, b in (bench.shard_bounds(total, world, r) for r in range(world)))
    padded = torch.zeros(164 * maxn, dtype=torch.uint8); padded[: local.numel()] = local
    out = bench.gather_proofs(dist, padded, rank, world)
    if rank == 0:
        merged = b"".join(bytes(t.numpy().tobytes())[: 164 * (b - a)] for t, (a, b) in zip(out, (bench.shard_bounds(total, world, r) for r in range(world))))
        q.put(merged)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything_once():
    for total in (0, 1, 7, 1024, 1061):
        for world in (1, 2, 3, 8):
            spans = [bench.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_reassembles_the_batch_in_order():
    world, total = 2, 101                      # ragged: 51 + 50
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert merged == _fake_proofs(bench.synthetic_records(total, seed=42), total)


def test_synthetic_records_are_deterministic_and_distinct_per_seed():
    a, b, c = bench.synthetic_records(8, 1), bench.synthetic_records(8, 1), bench.synthetic_records(8, 2)
    assert a == b and a != c and len(a) == 8 * 112
