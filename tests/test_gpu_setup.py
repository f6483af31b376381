"""Groth16 Setup as a product component (gsc_setup; reference keygen.go:345,384,423 calls groth16.Setup): for a fixed seed the
GPU-made keys are byte-identical to the oracle's Setup (oracle/setup.c — same toxic-waste derivation, independent arithmetic and
an independent fixed-base multiplication), keys made for the reference's r1cs.chacha20 have exactly the shape of the reference's
own pk.chacha20 / vk.chacha20, and proofs made under a CSPRNG-toxic-waste key verify with the drop-in verifier under the
matching vk."""
import base64
import random

import pytest

from conftest import AES, golden_bytes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["aes128", "aes256"])
def test_fixed_seed_keys_equal_the_oracles(gsc, aes_keys, name):
    algo = AES[name][0]
    r1cs, opk, ovk = aes_keys[name]               # oracle.setup(r1cs, bytes([algo] * 32)) — see conftest.aes_keys
    pk, vk = gsc.setup(r1cs, bytes([algo] * 32))
    assert vk == ovk
    assert len(pk) == len(opk)
    assert pk == opk


def test_chacha_keys_have_the_shape_of_the_reference_files(gsc, oracle):
    r1cs = golden_bytes("r1cs.chacha20")
    pk, vk = gsc.setup(r1cs, bytes([7] * 32))
    ref_pk, ref_vk = golden_bytes("pk.chacha20"), golden_bytes("vk.chacha20")
    assert len(pk) == len(ref_pk) == 3710459 and len(vk) == len(ref_vk) == 37196
    assert pk[:8] == ref_pk[:8] and pk[8:169] == ref_pk[8:169]              # domain size, n^-1, omega, omega^-1, g, g^-1, flag
    for off in (265, 704301, 1105233, 2153781, 2862009):                   # slice lengths of A, B, Z, K, G2.B
        assert pk[off:off + 4] == ref_pk[off:off + 4]
    assert pk[3663869:3663893] == ref_pk[3663869:3663893]                   # nbWires, NbInfinityA, NbInfinityB
    assert pk[3663893:3710455] == ref_pk[3663893:3710455]                   # InfinityA / InfinityB: a property of the circuit, not of tau
    opk, ovk = oracle.ProvingKey(pk), oracle.VerifyingKey(vk)               # and the oracle's decoders accept them
    assert ovk.nK == 1153


def test_csprng_key_proves_and_verifies(gsc):
    # the production path: toxic waste from the OS CSPRNG; two calls give different keys
    r1cs = golden_bytes("r1cs.aes128")
    pk, vk = gsc.setup(r1cs)
    pk2, vk2 = gsc.setup(r1cs)
    assert vk != vk2 and len(pk) == len(pk2)
    # (AES-128 may already be initialised with the session's oracle-made key: InitAlgorithm is idempotent, so prove in a child process)
    import os, subprocess, sys, tempfile
    from conftest import ROOT
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "pk"), "wb").write(pk); open(os.path.join(td, "vk"), "wb").write(vk)
        code = (
            "import sys, os, random, base64; sys.path.insert(0, %r)\n"
            "import gsc_loader; from bench import golden\n"
            "g = gsc_loader.load(); td = sys.argv[1]\n"
            "assert g.init_algorithm(1, open(os.path.join(td, 'pk'), 'rb').read(), golden('r1cs.aes128'))\n"
            "assert g.init_verifier(1, open(os.path.join(td, 'vk'), 'rb').read())\n"
            "rnd = random.Random(9); n = 70\n"
            "recs = b''.join(rnd.randbytes(16) + bytes(16) + rnd.randbytes(12) + rnd.getrandbits(31).to_bytes(4, 'little') + rnd.randbytes(64) for _ in range(n))\n"
            "ok, proofs, lens, cts = g.prove_raw(1, recs, n); assert ok == n and set(lens) == {196}\n"
            "for k in range(n):\n"
            "    rec = recs[112 * k:112 * (k + 1)]\n"
            "    sig = cts[64 * k:64 * k + 64] + rec[32:44] + rec[44:48][::-1] + rec[48:]\n"
            "    assert g.verify({'cipher': 'aes-128-ctr', 'proof': base64.b64encode(proofs[196 * k:196 * k + 196]).decode(), 'publicSignals': base64.b64encode(sig).decode()}), k\n"
            "    bad = bytearray(sig); bad[3] ^= 1\n"
            "    assert k or not g.verify({'cipher': 'aes-128-ctr', 'proof': base64.b64encode(proofs[:196]).decode(), 'publicSignals': base64.b64encode(bytes(bad)).decode()})\n"
            "print('ALLOK')\n" % ROOT)
        env = dict(os.environ, GSC_MAX_BATCH="128", GSC_Z_TABLE_GB="4", GSC_W_TABLE_GB="4")
        out = subprocess.run([sys.executable, "-c", code, td], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ALLOK" in out.stdout, out.stdout + out.stderr
