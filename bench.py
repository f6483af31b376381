#!/usr/bin/env python3
"""bench.py — Groth16 proofs/sec on N MI355X (BASELINE.json metric), through the libprove C-ABI.

Default (the driver's call): ChaCha20-V3 single 64-byte blocks, one batch of --batch independent statements per GPU per step.
A "step" is one pass of the hot path (witness -> quotient NTTs -> MSMs -> proof assembly) over that batch.  The other BASELINE
configs are selected with --workload {chacha20,aes128,aes256,mixed} and --batch {1,64,1024,8192} (--callers: concurrent caller
threads); every run prints the same JSON line (profiles/r03_bench_*.json hold one per config) and, after the clock stops, checks a
sample of the LAST timed step's proofs with the product's verifier libverify.so ("verified": n; a rejection fails the run).

  * chacha20 / aes128 / aes256 go through gsc_prove_raw (the binary twin of Prove: no JSON on the timed path);
  * mixed sends a JSON array (statement i uses cipher i mod 3) through ProveBatch, all three algorithms resident on the device.

Independent proofs shard across ranks (one process per GPU, weak scaling); the only collective is the gather of the finished
proofs to rank 0 (RCCL over xGMI).  `--gpus N` without RANK in the environment makes this process a launcher: it starts N rank
processes BEFORE anything touches torch or HIP and relays rank 0's line; under `torch.distributed.run` each rank reads
RANK / LOCAL_RANK / WORLD_SIZE and the world size must equal --gpus.  `--gpus N --in-library` is the other multi-GPU path: ONE process,
GSC_DEVICES=0..N-1, one call of N x batch statements per step split over the library's own engine replicas (what a Go / node host does).

The line carries, besides the contract's keys:
  "roofline"      : the dominant kernel (the Z-table MSM gather-accumulate of the slowest algorithm in the workload; the resident witness
                    kernel for calls of a handful of statements) priced against HBM peak with SURVEY.md §8(d)'s algorithmic bytes of the
                    statements the launch proved, timed live with HIP events on the kernel's own stream;
  "msm_stage"     : the whole MSM stage in GB/s on §8(d)'s MSM bytes per proof;
  "cpu_baseline"  : the CPU oracle (oracle/, a port — not gnark) timed on the cores the job may use (CPU quota) on a bounded sample of the same workload;
  "verified"      : proofs of the last timed step accepted by libverify.so under the matching verifying key.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8.0 TB/s spec
MSM_Z_BYTES_PER_BASE = 64 + 32              # affine G1 base + 32-byte scalar
ALGOS = {"chacha20": (0, "chacha20", 32), "aes128": (1, "aes-128-ctr", 16), "aes256": (2, "aes-256-ctr", 32)}
# algorithmic bytes per proof — whole path, and the MSM stage alone (AES: the survey's upper bounds)
BYTES_PER_PROOF = {"chacha20": 28_281_728, "aes128": 115_002_752, "aes256": 129_366_656}
MSM_BYTES_PER_PROOF = {"chacha20": 10_589_440, "aes128": 46_750_944, "aes256": 57_991_904}
SETUP_SEEDS = {"aes128": bytes([1] * 32), "aes256": bytes([2] * 32)}      # CPU-baseline keys (oracle Setup; any valid key costs the same to prove with)


def golden(name):
    import lzma
    p = os.path.join(GOLDEN, name)
    return lzma.open(p + ".xz").read() if os.path.exists(p + ".xz") else open(p, "rb").read()


def pmc_traffic(kernel_tag, batch, engine_desc):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), if they were taken on this
    configuration; None otherwise (PMC counters cannot be collected from inside the timed run)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_msm_z_pmc*.json")), reverse=True):
        try:
            d = json.load(open(path))
            c = d["config"]
            # the profile must be of THIS kernel configuration: algorithm, batch, digit width, number of bases and the quotient form
            # (evaluation form walks the n bases V, coefficient form the n - 1 of the key) — a profile that does not say is not replayed
            if c.get("kernel", "") == kernel_tag and c["batch"] == batch and ("window_z=%d " % c["window_z"]) in engine_desc \
                    and c.get("quotient") and ("quotient=%s" % c["quotient"]) in engine_desc and ("Z=%d " % c.get("nbases", -1)) in engine_desc:
                return d["hbm_bytes_per_launch"], "replayed from %s (rocprofv3 PMC passes of this configuration; counters cannot be read inside the timed run)" % os.path.relpath(path, ROOT)
        except Exception:
            pass
    return None, None


def valu_per_wave_add(nbases, nwin):
    """VALU instructions the Z-table kernel issues per wave-addition (64 lanes x one mixed addition), from the newest committed SQ counter
    pass (profiles/r*_valu_per_add*.json, made by tools/make_valu_per_add.py from SQ_INSTS_VALU of the kernel), replayed like `traffic`
    because counters cannot be read inside the timed run.  The pass must be of the same Z set (number of bases: a circuit no pass was
    taken on gets no figure); a different digit width runs the same gather-accumulate loop (the count is per addition) and says so."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_per_add*.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d["config"]["nbases"] == nbases:
                note = "" if d["config"]["windows"] == nwin else "; counted on the %d-window launch, this run has %d windows (same loop, other digit format)" % (d["config"]["windows"], nwin)
                return float(d["instr_per_wave_add"]), "replayed from %s (SQ_INSTS_VALU / wave-additions of a rocprofv3 --pmc pass%s)" % (os.path.relpath(path, ROOT), note)
        except Exception:
            pass
    return None, None


# VALU-only time of the Z kernel per digit window at 8192 columns: the launch with every gather forced onto one table entry (no misses) took
# 15.4 ms per window at c = 13 and at c = 16 alike (profiles/r02_window_sweep.txt); it scales with the columns.
VALU_ONLY_MS_PER_WINDOW_8192 = 15.4
WAVE64_ISSUE_CYCLES = 4.0      # a wave64 VALU instruction occupies a 16-lane SIMD for four cycles: the issue floor


def roofline_valu_of(roof, rows, simds):
    """The BINDING roofline of the Z-table kernel: VALU issue.  cycles per wave instruction = launch time x shader clock x SIMDs /
    (instructions per wave-addition x wave-additions); frac = 4 / that.  Clock: measured live by the kernel itself (two clock stamps of
    a wave in the middle of the launch, gsc_last_kernel_clock); instruction count: replayed from the committed counter pass."""
    clocks = [r[6] for r in rows if r[6] > 0]; nwin = rows[-1][7]; cols, nb = rows[-1][3], rows[-1][4]
    ipa, src = valu_per_wave_add(nb, nwin)
    if not clocks or not ipa or not nwin or not simds:
        return None
    mhz = sum(clocks) / len(clocks); ms = roof["launch_ms"]
    wave_adds = nb * nwin * (cols // 64)
    cpi = ms * 1e-3 * mhz * 1e6 * simds / (ipa * wave_adds)
    valu_ms = nwin * VALU_ONLY_MS_PER_WINDOW_8192 * (cols / 8192.0) * (nb / 32768.0)
    return {"kernel": roof["kernel"], "bound": "valu-issue", "instr_per_wave_add": ipa, "instr_source": src,
            "wave_adds_per_launch": wave_adds, "windows": nwin, "simds": simds, "clock_mhz": round(mhz, 1),
            "clock_source": "live: shader-clock / 100 MHz-clock stamps of eight waves spread over each timed launch, one on each XCD, averaged",
            "launch_ms": ms, "cycles_per_wave_instr": round(cpi, 4), "issue_floor_cycles": WAVE64_ISSUE_CYCLES, "frac": round(WAVE64_ISSUE_CYCLES / cpi, 4),
            "valu_only_ms": round(valu_ms, 2), "valu_only_source": "%.1f ms per window of 32 768 bases at 8192 columns with every gather forced onto one entry (profiles/r02_window_sweep.txt), scaled by columns and bases" % VALU_ONLY_MS_PER_WINDOW_8192,
            "miss_clock_loss_ms": round(ms - valu_ms, 2)}


XOSHIRO_SEED = 0x9E3779B97F4A7C15


def xoshiro_records(n, first_index):
    """statement number `first_index + i` draws its 112-byte record {key[32], nonce[12], counter u32 LE, input[64]}
    from a xoshiro256** stream of its own, seeded with 0x9E3779B97F4A7C15 + proofIndex (the 256-bit state is the first four outputs of
    splitmix64 on that seed; the record is the stream's first 14 outputs, little-endian).  Anyone can rebuild statement k from k alone."""
    import numpy as np
    M = np.uint64
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + M(first_index & 0xFFFFFFFFFFFFFFFF)) + M(XOSHIRO_SEED)
        st = []
        for _ in range(4):                                   # splitmix64
            x = x + M(0x9E3779B97F4A7C15)
            z = x
            z = (z ^ (z >> M(30))) * M(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> M(27))) * M(0x94D049BB133111EB)
            st.append(z ^ (z >> M(31)))
        s0, s1, s2, s3 = st
        out = np.empty((n, 14), dtype="<u8")
        for k in range(14):                                  # xoshiro256**
            t = s1 * M(5)
            out[:, k] = ((t << M(7)) | (t >> M(57))) * M(9)
            t = s1 << M(17)
            s2 = s2 ^ s0; s3 = s3 ^ s1; s1 = s1 ^ s2; s0 = s0 ^ s3
            s2 = s2 ^ t
            s3 = (s3 << M(45)) | (s3 >> M(19))
    return out.tobytes()


def synthetic_records(n, seed):
    """n x 112 B records of the statements seed * 2^32 + (0 .. n-1) (xoshiro_records)."""
    return xoshiro_records(n, int(seed) << 32)


def engine_env(workload, per_algo, share=1, library_defaults=False):
    """The engine configuration a timed run uses (environment read by libprove at InitAlgorithm).  One place, so that the test of the
    timed configuration (tests/test_gpu_00_bench_config.py) starts its prover with exactly these settings.  library_defaults: only the
    batch capacity is set — the table budgets stay the library's own (48 + 16 GB per algorithm: all three algorithms co-resident), the
    multi-tenant configuration; the default run gives ONE algorithm the device."""
    env = {"GSC_MAX_BATCH": str(max(64, (per_algo + 63) // 64 * 64))}
    if workload != "mixed" and not library_defaults:      # one algorithm alone on the device: widest Z digits that fit (ChaCha c = 17: 137 GB; AES c = 15: 137 GB); mixed keeps the
        env["GSC_Z_TABLE_GB"] = str(140 // share)      # library defaults, under which all three algorithms are resident at once (3 x (48 + 16) GB)
        env["GSC_W_TABLE_GB"] = str(56 // share)       # AES-V2 wide wires (and the wide rows of c, evaluation-form quotient): c = 15 (48 GB) instead of 14
        # share > 1: several engine replicas on ONE device (the one-GPU rehearsal of --in-library --devices 0,0) split its memory
    return env


def sample_indices(n, want, edges=()):
    """`want` statement indices spread evenly over a batch of n, always with 0, 63, 64, n - 1 and the given edges (chunk / replica
    boundaries and their neighbours)."""
    idx = {i for i in (0, 63, 64, n - 1) if 0 <= i < n}
    for e in edges:
        idx.update(i for i in (e - 1, e) if 0 <= i < n)
    if want >= n:
        return list(range(n))
    k = 0
    while len(idx) < want:
        idx.add((k * n) // want); k += 1
        if k > 4 * want:
            break
    return sorted(idx)


def signals_of(name, rec, ct):
    """publicSignals of one statement as the verifier wants them (libraries/verifier/impl/verifiers.go:59-62): ct | nonce | counter | pt,
    the counter little-endian for ChaCha20 and big-endian for AES."""
    ctr = rec[44:48] if name == "chacha20" else rec[44:48][::-1]
    return ct + rec[32:44] + ctr + rec[48:112]


def verify_items(g, items, threads=16):
    """items: (cipher name, proof bytes, publicSignals bytes) -> list of verdicts from the product's libverify.so (CPU, like the
    reference's verifier; ctypes releases the GIL, so the pairings run on `threads` host cores)."""
    import base64
    from concurrent.futures import ThreadPoolExecutor

    def check(it):
        cipher, proof, sig = it
        return g.verify({"cipher": cipher, "proof": base64.b64encode(proof).decode(), "publicSignals": base64.b64encode(sig).decode()})
    with ThreadPoolExecutor(threads) as pool:
        return list(pool.map(check, items))


def provable(rec, name):
    """AES-V2 asserts counter + 4 <= 2^32 - 1 inside the circuit (circuits/aesV2/aes128.go:41-53): keep synthetic counters provable."""
    if name == "chacha20":
        return rec
    b = bytearray(rec)
    for i in range(len(b) // 112):
        b[112 * i + 47] &= 0x7F
    return bytes(b)


def mixed_json(n, first_index):
    """JSON array for ProveBatch: statement i uses cipher i mod 3."""
    import base64
    recs = provable(xoshiro_records(n, first_index), "aes")
    names = ("chacha20", "aes128", "aes256")
    out = []
    for i in range(n):
        r = recs[112 * i:112 * (i + 1)]
        _, cipher, kl = ALGOS[names[i % 3]]
        out.append({"cipher": cipher, "key": base64.b64encode(r[:kl]).decode(), "nonce": base64.b64encode(r[32:44]).decode(),
                    "counter": int.from_bytes(r[44:48], "little"), "input": base64.b64encode(r[48:112]).decode()})
    return json.dumps(out).encode()


def shard_bounds(total, world, rank):
    """Contiguous block partition of `total` units over `world` ranks (used by --total-proofs strong-scaling runs and tests)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_proofs(dist, local: "torch.Tensor", rank, world, use_dist=None):
    """The path's only exchange: every rank's finished proofs (164 / 196 B each) to rank 0."""
    import torch
    if not (world > 1 if use_dist is None else use_dist):
        return [local]
    out = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, out, dst=0)
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` run directly (as the driver does) starts the N ranks itself
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launch_ranks(n):
    """Start n fresh rank processes of this script and relay rank 0's JSON line.  Nothing in this (parent) process has imported
    torch or touched HIP: a process that initialised the GPU must never fork/exec workers."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # wait for all ranks; if one fails the others would wait for it in a collective forever, so they are stopped (exact PIDs)
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=30))
        except subprocess.TimeoutExpired:
            p.kill(); rcs.append(p.wait())
    out0 = procs[0].stdout.read()           # rank 0 writes exactly one line, at the very end
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if failed or bad or not out0.strip():
        raise SystemExit("bench.py: rank(s) failed: %s" % (bad or "no output from rank 0"))
    sys.stdout.write(out0.decode()); sys.stdout.flush()


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle is the checker / baseline, never the product path)
def _cpu_worker(args):
    name, n, seed = args
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import oracle as O
    algo, cipher, kl = ALGOS[name]
    cs = O.R1CS(golden("r1cs." + name))
    pk = O.ProvingKey(golden("pk.chacha20") if name == "chacha20" else O.setup(cs, SETUP_SEEDS[name])[0])
    recs = provable(synthetic_records(n, seed), name)
    t = time.time()
    for i in range(n):
        r = recs[112 * i:112 * (i + 1)]
        O.prove(cs, pk, cipher, r[:kl], r[32:44], int.from_bytes(r[44:48], "little"), r[48:112], 12345 + i, 67890 + i, 424242 + i)
    return time.time() - t


def usable_cores():
    """(cores this process can actually keep busy, how that was found): the scheduler affinity, cut down to the container's CPU quota
    (cgroup v2 cpu.max / v1 cfs quota) — a GPU box shows all 256 host threads in the affinity mask while the job's share is 16."""
    n = len(os.sched_getaffinity(0)); how = "affinity mask"
    for path, v2 in (("/sys/fs/cgroup/cpu.max", True), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", False)):
        try:
            if v2:
                quota, period = open(path).read().split()[:2]
            else:
                quota, period = open(path).read().strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1") and int(period) > 0:
                q = max(1, int(int(quota) / int(period)))
                if q < n:
                    n, how = q, "cgroup CPU quota %s/%s" % (quota, period)
                break
        except (OSError, ValueError):
            pass
    if how == "affinity mask":
        # no quota visible: a GPU box is one slot of a shared host (its share is 16 cores per GPU); oversubscribing it measures the
        # neighbours' load, not the port (256 workers there: 17 proofs/s against 31 with 16)
        try:
            import torch
            share = 16 * max(1, torch.cuda.device_count())
            if torch.cuda.device_count() and share < n:
                n, how = share, "16 cores per visible GPU (shared host, no CPU quota visible; affinity mask %d)" % len(os.sched_getaffinity(0))
        except Exception:
            pass
    return n, how


def cpu_baseline(workload, cores):
    """Oracle (CPU port of the same path) on `cores` host cores: independent single-threaded provers, one per core, on a bounded
    sample of the same workload (about 10-30 s of CPU work)."""
    import multiprocessing as mp
    names = ["chacha20", "aes128", "aes256"] if workload == "mixed" else [workload]
    per_core = {"chacha20": 24, "aes128": 6, "aes256": 5}
    ctx = mp.get_context("spawn")
    t = time.time()
    jobs = [(names[i % len(names)], per_core[names[i % len(names)]] if len(names) == 1 else 4, 1000 + i) for i in range(cores)]
    with ctx.Pool(cores) as pool:
        busy = pool.map(_cpu_worker, jobs)
    wall = time.time() - t
    total = sum(j[1] for j in jobs)
    rate = total / max(busy)                   # excludes key decoding / setup; all workers run concurrently
    return {"value": round(rate, 3), "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "%d %s proofs (%s per core, single-threaded oracle per core, key decode excluded); wall %.1fs"
                      % (total, workload, "/".join(str(j[1]) for j in jobs[:len(names)]), wall)}


# ---------------------------------------------------------------------------------------------------------------------------
class StubProver:
    """CPU stand-in used ONLY by the launcher test (tests/test_bench_launcher.py, --stub-prover): exercises rank spawning, the
    rendezvous, the barrier / gather / max-over-ranks plumbing with gloo.  It proves nothing and is never a fallback: the real
    path raises when there is no GPU."""
    def __init__(self, B):
        import hashlib
        self.h = hashlib
        self.B = B

    def step(self, recs):
        time.sleep(0.01)
        return b"".join(self.h.sha256(recs[112 * i:112 * (i + 1)]).digest() * 5 + bytes(4) for i in range(self.B)), None


WITNESS_BYTES_PER_PROOF = {"chacha20": 3_012_224, "aes128": 9_531_552, "aes256": 12_654_496}      # write W + a, b, c


def roofline_of(n, rows, g):
    """Roofline object of one algorithm's dominant kernel from the per-step records (kernel name, HIP-event ms, statements, columns,
    bases per proof, stage ms): batch kernels -> the Z-table gather-accumulate priced on (64 + 32) B per base and proof; calls on the
    latency path (a handful of statements) -> the resident witness kernel priced on §8(d)'s witness bytes.  Bytes are counted for the
    STATEMENTS the launch proved, not for the 64-column padding."""
    kname = rows[-1][0]; avg_ms = sum(r[1] for r in rows) / len(rows); stmts, cols, nb = rows[-1][2], rows[-1][3], rows[-1][4]
    solver = kname.startswith("k_solver") or kname.startswith("k_wit_")      # calls on the latency path: the witness kernels are the dominant ones
    alg_bytes = stmts * (WITNESS_BYTES_PER_PROOF[n] if solver else nb * MSM_Z_BYTES_PER_BASE)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    msm_ms = sum(r[5]["msm"] for r in rows) / len(rows)
    traffic, src = (None, None) if solver else pmc_traffic(n, cols, g.describe(ALGOS[n][0]))
    return {"kernel": "%s (%s, %s)" % (kname, "witness solver of the latency path" if solver else "Z-table gather-accumulate", n), "bound": "hbm",
            "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
            "traffic": traffic, "traffic_source": src, "launch_ms": round(avg_ms, 3), "algorithmic_bytes_per_launch": alg_bytes,
            "proofs_per_launch": stmts, "columns_per_launch": cols,
            "msm_stage": {"ms": round(msm_ms, 3), "bytes_per_proof": MSM_BYTES_PER_PROOF[n], "GB/s": round(stmts * MSM_BYTES_PER_PROOF[n] / (msm_ms * 1e-3) / 1e9, 2),
                          "frac_of_hbm_peak": round(stmts * MSM_BYTES_PER_PROOF[n] / (msm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)},
            "stage_ms_last_step": rows[-1][5]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["chacha20", "aes128", "aes256", "mixed"], default="chacha20")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("GSC_BENCH_BATCH", "0")), help="proofs per GPU per step (default 8192 ChaCha, 1024 AES, 3072 mixed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="host cores for the CPU baseline (default: all the process may run on)")
    ap.add_argument("--verify", type=int, default=1024, help="proofs of the LAST timed step checked with libverify.so after the clock stops (0 = none)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank (rehearsal of the multi-GPU code path on a one-GPU box)")
    ap.add_argument("--in-library", action="store_true", help="ONE process drives all --gpus devices through the library's own replicas (GSC_DEVICES=0..N-1): "
                                                               "what a single FFI host (Go / node) would do; one call of N x batch statements per step")
    ap.add_argument("--devices", default="", help="with --in-library: the device list itself (e.g. 0,0 rehearses two replicas on a one-GPU box)")
    ap.add_argument("--callers", type=int, default=2, help="concurrent caller threads that keep the library busy (each owns output buffers); small batches need several in flight")
    ap.add_argument("--library-defaults", action="store_true", help="no table-budget overrides: the library's default (multi-tenant) configuration, under which all three circuits fit the device together")
    ap.add_argument("--stub-prover", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.in_library:
        devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
        if len(devices) != args.gpus:
            raise SystemExit("bench.py: --devices lists %d devices but --gpus is %d" % (len(devices), args.gpus))
        if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("bench.py: --in-library is one process for all GPUs; do not start it under torch.distributed.run")
    elif "RANK" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)

    replicas = args.gpus if args.in_library else 1
    rank = 0 if args.in_library else int(os.environ.get("RANK", "0"))
    world = 1 if args.in_library else int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.in_library else int(os.environ.get("LOCAL_RANK", "0"))
    if not args.in_library and world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d: start one rank per GPU (or run `bench.py --gpus N` directly)" % (world, args.gpus))
    workload = args.workload
    B = args.batch or {"chacha20": 8192, "aes128": 1024, "aes256": 1024, "mixed": 3072}[workload]      # per GPU
    BT = B * replicas                                                                               # statements per call of this process
    # stdout carries exactly ONE JSON line (rank 0).  Native libraries print there too (libprove reports errors on stdout like the
    # reference's fmt.Println, RCCL prints a banner), so file descriptor 1 is pointed at stderr for the whole run and the line is
    # written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1); os.dup2(2, 1)
    if args.in_library:
        os.environ["GSC_DEVICES"] = ",".join(str(d) for d in devices)
    else:
        os.environ["GSC_DEVICE"] = str(local_rank)
    names = ["chacha20", "aes128", "aes256"] if workload == "mixed" else [workload]
    per_algo = B if workload != "mixed" else (B + 2) // 3
    share = max(devices.count(d) for d in devices) if args.in_library else 1
    for k, v in engine_env(workload, per_algo, share, args.library_defaults).items():
        os.environ.setdefault(k, v)
    import torch
    import torch.distributed as dist
    stub = args.stub_prover
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the prover has no CPU fallback")
    if not stub:
        torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)

    import numpy as np
    g = None
    vks = {}
    if not stub:
        import gsc_loader
        g = gsc_loader.load()
        for name in names:
            algo = ALGOS[name][0]
            r1cs = golden("r1cs." + name)
            if name == "chacha20":
                pk, vks[name] = golden("pk.chacha20"), golden("vk.chacha20")
            else:
                pk, vks[name] = g.setup(r1cs)                        # the product's own Groth16 Setup (GPU), CSPRNG toxic waste
            if not g.init_algorithm(algo, pk, r1cs):
                raise SystemExit("InitAlgorithm failed for " + name)

    # statements of every step are made before the clock starts (112 B each: "inputs resident"); output buffers are reused.
    # Statement number = rank * 2^44 + step id * 2^20 + position in the batch (xoshiro_records: reproducible by anyone).
    ids = [0x800000 + w for w in range(args.warmup)] + list(range(args.steps))
    first_index = lambda i: (rank << 44) | ((i & 0xFFFFFF) << 20)
    if workload == "mixed":
        inputs_of = {i: mixed_json(BT, first_index(i)) for i in ids}
    else:
        inputs_of = {i: provable(xoshiro_records(BT, first_index(i)), workload) for i in ids}
    import threading
    from concurrent.futures import ThreadPoolExecutor
    # --callers threads (default two) keep the library busy, like concurrent Prove callers do (libraries/core_test.go:44-111): while one call's batch is on the
    # GPU, the other call does its host part (native cipher, CSPRNG draws, packing).  Device work of the two calls is serialised by the
    # library, so a step still means one batch through the whole path; each caller owns a set of output buffers.
    plen = 164 if workload == "chacha20" else 196
    NC = max(1, args.callers)
    bufs = [g.raw_buffers(BT) if g else None for _ in range(NC)]
    free = [threading.Event() for _ in range(NC)]
    for e in free:
        e.set()
    stubp = StubProver(BT) if stub else None
    last_json = {}

    def kernel_stats(n):
        # (the library reports the CALLING thread's own last call: with several callers a step's row is that step's, not a neighbour's)
        kname, ms, stmts, cols, nb = g.last_dominant_kernel(ALGOS[n][0])
        mhz, nwin = g.last_kernel_clock(ALGOS[n][0])
        return (n, (kname, ms, stmts, cols, nb, mhz, nwin), g.last_stage_ms(ALGOS[n][0]))

    def prove(i, slot):
        free[slot].wait(); free[slot].clear()
        if stub:
            return stubp.step(inputs_of[i])
        if workload == "mixed":
            out = g.prove_batch_bytes(inputs_of[i])
            if out.count(b'"proofJson"') != BT:
                raise SystemExit("rank %d: only %d of %d proofs produced" % (rank, out.count(b'"proofJson"'), BT))
            last_json[i] = out
            last_json.pop(i - NC, None)
            return out, [kernel_stats(n) for n in names]
        pb, lb, cb = bufs[slot]
        algo = ALGOS[workload][0]
        ok = g.prove_raw_into(algo, inputs_of[i], BT, pb, lb, cb)
        if ok != BT:
            raise SystemExit("rank %d: only %d of %d proofs produced" % (rank, ok, BT))
        if not (np.frombuffer(lb, dtype=np.uint32) == plen).all():
            raise SystemExit("rank %d: incomplete proofs" % rank)
        return pb, [kernel_stats(workload)]

    def run(step_ids, stats):
        with ThreadPoolExecutor(NC) as pool:
            futs = [pool.submit(prove, i, k % NC) for k, i in enumerate(step_ids)]
            last = len(futs) - 1
            for k, f in enumerate(futs):          # results are consumed in order on this thread (the only one that talks to RCCL)
                payload, st = f.result()
                raw = bytes(payload) if isinstance(payload, (bytes, bytearray)) else None
                if raw is not None:               # JSON (mixed) or stub bytes: fixed-size frame for the gather
                    frame = np.zeros(BT * 416 if workload == "mixed" and not stub else len(raw), dtype=np.uint8)
                    frame[: len(raw)] = np.frombuffer(raw, dtype=np.uint8)[: frame.size]
                    local = torch.from_numpy(frame).to(dev)
                else:
                    local = torch.frombuffer(payload, dtype=torch.uint8).to(dev)
                if k != last or stats is None:    # the last timed step's buffers stay untouched: they are verified after the clock stops
                    free[k % NC].set()
                gather_proofs(dist, local, rank, world, use_dist)
                if stats is not None:
                    stats.append(st)

    def barrier():
        if use_dist:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    run(ids[:args.warmup], None)
    barrier()
    t0 = time.time()
    stats = []
    run(ids[args.warmup:], stats)
    barrier()
    elapsed = time.time() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- parity gate for the timed configuration (libraries/core_test.go:171): a sample of the LAST timed step's proofs,
    # spread over the batch (first / last wave, chunk and replica boundaries), goes through the product's verifier libverify.so under the
    # verifying key that belongs to the proving key in use.  A rejection fails the run: no line is printed.
    verified = 0
    if not stub and args.verify > 0 and args.steps > 0:
        import base64
        for name in names:
            if not g.init_verifier(ALGOS[name][0], vks[name]):
                raise SystemExit("bench.py: InitVerifier failed for " + name)
        last_id = ids[-1]
        edges = [r * B for r in range(1, replicas)]
        items = []
        if workload == "mixed":
            outs = json.loads(last_json[last_id]); reqs = json.loads(inputs_of[last_id])
            for k in sample_indices(BT, args.verify, edges):
                q, o = reqs[k], outs[k]
                cname = {"chacha20": "chacha20", "aes-128-ctr": "aes128", "aes-256-ctr": "aes256"}[q["cipher"]]
                rec = bytes(32) + base64.b64decode(q["nonce"]) + int(q["counter"]).to_bytes(4, "little") + base64.b64decode(q["input"])
                items.append((q["cipher"], base64.b64decode(o["proof"]["proofJson"]), signals_of(cname, rec, base64.b64decode(o["publicSignals"]))))
        else:
            pb, lb, cb = bufs[(args.steps - 1) % NC]
            recs = inputs_of[last_id]; cipher = ALGOS[workload][1]; proofs, cts = pb.raw, cb.raw
            for k in sample_indices(BT, args.verify, edges):
                items.append((cipher, proofs[196 * k:196 * k + plen], signals_of(workload, recs[112 * k:112 * (k + 1)], cts[64 * k:64 * k + 64])))
        res = verify_items(g, items, threads=min(32, 2 * usable_cores()[0]))
        if not all(res):
            raise SystemExit("bench.py: rank %d: %d of %d sampled proofs of the last timed step were REJECTED by libverify.so" % (rank, res.count(False), len(res)))
        # a proof must not verify for a neighbour's statement (the verifier is not a rubber stamp)
        if len(items) > 1 and verify_items(g, [(items[0][0], items[0][1], items[1][2])], 1)[0] and items[0][2] != items[1][2]:
            raise SystemExit("bench.py: libverify.so accepted a proof for another statement")
        verified = len(res)
        if use_dist:
            tv = torch.tensor([verified], dtype=torch.int64, device=dev)
            dist.all_reduce(tv, op=dist.ReduceOp.SUM)
            verified = int(tv.item())

    if rank == 0:
        ngpu = world * replicas
        total = ngpu * args.steps * B
        value = total / elapsed
        metric = {"chacha20": "Groth16 proofs/sec (ChaCha20-V3 1-block)", "aes128": "Groth16 proofs/sec (AES-128-V2 64-byte input)",
                  "aes256": "Groth16 proofs/sec (AES-256-V2 64-byte input)", "mixed": "Groth16 proofs/sec (mixed ChaCha20-V3 / AES-128-V2 / AES-256-V2 batch)"}[workload]
        desc = {"chacha20": "ChaCha20-V3 single 64-byte block, reference pk.chacha20/r1cs.chacha20",
                "aes128": "AES-128-V2 (lookup-table circuit), 64-byte input, reference r1cs.aes128, proving key from the product's Setup (the reference ships none)",
                "aes256": "AES-256-V2 (lookup-table circuit), 64-byte input, reference r1cs.aes256, proving key from the product's Setup (the reference ships none)",
                "mixed": "mixed batch through ProveBatch (JSON in/out): statement i uses cipher i mod 3 of chacha20 / aes-128-ctr / aes-256-ctr, all three algorithms resident"}[workload]
        par = ("in-library replicas: one process, GSC_DEVICES=%s, one call of %d statements per step split over the replicas" % (os.environ["GSC_DEVICES"], BT)) if args.in_library \
            else "proofs sharded over %d GPU(s), one process per GPU, gather to rank 0" % world
        line = {
            "metric": metric, "value": round(value, 2), "unit": "proofs/s",
            "n_gpus": ngpu, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (BN254 Fr/Fp, 254-bit modular integers)",
            "data": "synthetic",
            "config": {"workload": "%s; 1xMI355X per rank: batch of %d independent proofs per GPU per step, CSPRNG (r,s); statements from xoshiro256** seeded 0x9E3779B97F4A7C15 + index" % (desc, B),
                       "batch_per_gpu": B, "callers": NC, "parallelism": par, "tables": "library defaults (multi-tenant)" if args.library_defaults else "single algorithm per device" if workload != "mixed" else "library defaults (multi-tenant)",
                       "engine": {n: g.describe(ALGOS[n][0]) for n in names} if g else "stub"},
            "verified": verified,
        }
        if not stub:
            # dominant kernel of the algorithm whose launch is longest; averaged over the timed steps
            per = {}
            for st in stats:
                for n, (kname, ms, stmts, cols, nb, mhz, nwin), stage in st:
                    per.setdefault(n, []).append((kname, ms, stmts, cols, nb, stage, mhz, nwin))
            roofs = {n: roofline_of(n, rows, g) for n, rows in per.items()}
            dom = max(roofs, key=lambda n: roofs[n]["launch_ms"])
            line["roofline"] = dict(roofs[dom])
            if roofs[dom]["kernel"].startswith("k_msm_win"):
                rv = roofline_valu_of(roofs[dom], per[dom], 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count)
                if rv:
                    line["roofline_valu"] = rv
            bpp = BYTES_PER_PROOF[workload] if workload != "mixed" else sum(BYTES_PER_PROOF.values()) / 3.0
            line["roofline"]["whole_path_frac"] = round(value / ngpu * bpp / 1e9 / HBM_PEAK_GBS, 6)
            line["msm_stage"] = line["roofline"].pop("msm_stage")
            line["stage_ms_last_step"] = line["roofline"].pop("stage_ms_last_step")
            if "+beside-wire-sets" in str(line["config"].get("engine")) and ("+beside-wire-sets(<4096)" not in str(line["config"].get("engine")) or B < 4096):
                line["stage_note"] = ("the quotient kernels run on a stream of their own beside the wire-set MSMs (A, B1, B2, K, c); the span they share is charged to "
                                      "'msm', 'quotient' is what the quotient still ran alone afterwards; per-kernel times: profiles/r04_kernel_stats*.csv")
            if len(roofs) > 1:
                line["roofline_per_algorithm"] = roofs
            if ngpu == 1 and not args.no_cpu_baseline:
                # "all host cores"  = every core this job may keep busy: the affinity mask cut down to the container's CPU quota
                quota_cores, how = usable_cores()
                cores = args.cpu_cores or quota_cores
                try:
                    line["cpu_baseline"] = cpu_baseline(workload, cores)
                except Exception as e:      # the baseline is reported, never required for the GPU number
                    line["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": cores, "kind": "port", "sample": "failed: %r" % (e,)}
                line["cpu_baseline"]["host_cores"] = os.cpu_count() or 1
                line["cpu_baseline"]["cores_from"] = how
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
