"""A finished call leaves no secrets in device memory.  The witness of these circuits is a cipher key: the engine clears
the key wires (32-byte rows and the byte plane of the small-integer witness path), the rows of r, s and -rs, the raw input records, the
prover randomness, its endomorphism split and the commitment mask behind the last kernel of every call (csrc/engine_prove.hip
wipe_secrets); host-side copies of requests clear themselves when they go away (ProofRequest's destructor).  The reference leaves all of
this to Go's garbage collector (libraries/prover/impl/provers.go:79-158).  The test hook gsc_debug_secret_residue counts what is left."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import bench, gsc_loader
g = gsc_loader.load()
algo, name = int(sys.argv[2]), sys.argv[3]
if algo == 0:
    pk = bench.golden("pk.chacha20")
else:
    pk, _vk = g.setup(bench.golden("r1cs." + name))
assert g.init_algorithm(algo, pk, bench.golden("r1cs." + name))
assert g.debug_secret_residue(algo) == 0
cipher = bench.ALGOS[name][1]; kl = bench.ALGOS[name][2]
left = []
for n in (1, 5, 64, 200):                     # latency path, its multi-statement form, one batch, a ragged multi-batch call (byte planes for ChaCha20)
    recs = bench.provable(bench.xoshiro_records(n, (0x5EC << 20) + n), name)
    ok, proofs, lens, cts = g.prove_raw(algo, recs, n)
    assert ok == n
    left.append(g.debug_secret_residue(algo))
r = recs[:112]
out = g.prove({"cipher": cipher, "key": list(r[:kl]), "nonce": list(r[32:44]), "counter": int.from_bytes(r[44:48], "little"), "input": list(r[48:112])})
assert b"proofJson" in (out if isinstance(out, bytes) else out.encode())
left.append(g.debug_secret_residue(algo))
print("RESIDUE", left)
print("CHILD-OK")
"""


def _residues(algo, name, extra=None):
    env = dict(os.environ, GSC_MAX_BATCH="256", GSC_WINDOW_Z="8", GSC_WINDOW_W="8", GSC_W_TABLE_GB="8", GSC_FEW_Z_GB="3", GSC_FEW_WIDE="0")
    env.update(extra or {})
    p = subprocess.run([sys.executable, "-c", _CHILD, ROOT, str(algo), name], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "CHILD-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESIDUE")][0]
    return eval(line.split(" ", 1)[1])


@pytest.mark.parametrize("algo,name", [(0, "chacha20"), (1, "aes128")])
def test_no_key_wire_or_randomness_survives_a_call_in_device_memory(algo, name):
    assert _residues(algo, name) == [0, 0, 0, 0, 0]
    # the hook does see them when the wipe is switched off (a test-hooks-only knob)
    kept = _residues(algo, name, {"GSC_KEEP_SECRETS": "1"})
    assert all(k > 1000 for k in kept), kept
