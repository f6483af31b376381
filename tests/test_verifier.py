"""CPU tests of libverify.so — the drop-in for the reference's verifier library (libraries/verifier/libverify.go:14-17):
the App. E proofs must be accepted under the reference's own vk.chacha20, tampered inputs rejected, malformed JSON -> false
(core_test.go:127), and its verdicts must agree with the oracle's independent pairing implementation."""
import base64
import json
import random
import subprocess

import pytest

from conftest import AES, KAT, golden_bytes


@pytest.fixture(scope="module")
def verifier(gsc):
    assert gsc.init_verifier(0, golden_bytes("vk.chacha20"))
    return gsc


def _sig(ct, nonce, counter, pt, order="little"):
    return ct + nonce + counter.to_bytes(4, order) + pt


def test_exports_and_independence(gsc):
    out = subprocess.check_output(["nm", "-D", "--defined-only", gsc.VERIFY_LIB_PATH]).decode()
    assert {l.split()[-1] for l in out.splitlines() if " T " in l} == {"Verify", "InitVerifier"}
    assert "liboracle" not in subprocess.check_output(["ldd", gsc.VERIFY_LIB_PATH]).decode()


def test_kat_proofs_verify_under_reference_vk(verifier):
    sig = _sig(KAT["ciphertext"], KAT["nonce"], KAT["counter"], KAT["input"])
    for hexproof in KAT["proofs"].values():
        proof = bytes.fromhex(hexproof)
        assert verifier.verify({"cipher": "chacha20", "proof": proof, "publicSignals": sig})
        # base64 fields, as the Go caller marshals []uint8 (core_test.go:165-170)
        assert verifier.verify(json.dumps({"cipher": "chacha20", "proof": base64.b64encode(proof).decode(), "publicSignals": base64.b64encode(sig).decode()}))
        for pos in (0, 70, 77, 100):
            bad = bytearray(sig); bad[pos] ^= 1
            assert not verifier.verify({"cipher": "chacha20", "proof": proof, "publicSignals": bytes(bad)})
        for pos in (1, 40, 100, 140):
            badp = bytearray(proof); badp[pos] ^= 1
            assert not verifier.verify({"cipher": "chacha20", "proof": bytes(badp), "publicSignals": sig})


def test_malformed_inputs_are_false_never_a_crash(verifier):
    assert not verifier.verify(b'{"cipher":"chacha20"}')                       # core_test.go:127
    assert not verifier.verify(b"")
    assert not verifier.verify(b"{")
    assert not verifier.verify(b"[1,2,3]")
    assert not verifier.verify({"cipher": "chacha21", "proof": [], "publicSignals": []})
    assert not verifier.verify({"cipher": "chacha20", "proof": [1, 2, 3], "publicSignals": [0] * 144})
    assert not verifier.verify({"cipher": "chacha20", "proof": [300], "publicSignals": [0] * 144})
    assert not verifier.verify({"cipher": "chacha20", "proof": "!!!", "publicSignals": "AAAA"})
    assert not verifier.verify({"cipher": "aes-256-ctr", "proof": [0] * 196, "publicSignals": [0] * 144})   # key not loaded
    assert not verifier.init_verifier(0, b"\x00" * 50) and not verifier.init_verifier(9, golden_bytes("vk.chacha20"))


def test_agrees_with_oracle_on_random_chacha_proofs(verifier, oracle, chacha_oracle):
    cs, pk, vk = chacha_oracle
    rnd = random.Random(31)
    key, nonce, pt = rnd.randbytes(32), rnd.randbytes(12), rnd.randbytes(64)
    proof, ct = oracle.prove(cs, pk, "chacha20", key, nonce, 1, pt, rnd.getrandbits(250), rnd.getrandbits(250))
    sig = _sig(ct, nonce, 1, pt)
    assert verifier.verify({"cipher": "chacha20", "proof": proof, "publicSignals": sig}) and oracle.verify(vk, "chacha20", proof, sig)
    wrong = _sig(ct, nonce, 2, pt)
    assert not verifier.verify({"cipher": "chacha20", "proof": proof, "publicSignals": wrong}) and not oracle.verify(vk, "chacha20", proof, wrong)


def test_reference_aes_verifying_keys_load(gsc):
    # the reference ships vk.aes128 / vk.aes256 (not the matching pk): they must parse (143 K points, one commitment key)
    assert gsc.init_verifier(1, golden_bytes("vk.aes128")) and gsc.init_verifier(2, golden_bytes("vk.aes256"))


@pytest.mark.parametrize("name", list(AES))
def test_aes_commitment_proofs(gsc, oracle, aes_keys, name):
    algo, cipher, keylen = AES[name]
    r1cs, pkb, vkb = aes_keys[name]
    assert gsc.init_verifier(algo, vkb)
    cs, pk, vk = oracle.R1CS(r1cs), oracle.ProvingKey(pkb), oracle.VerifyingKey(vkb)
    rnd = random.Random(50 + algo)
    key, nonce, pt = rnd.randbytes(keylen), rnd.randbytes(12), rnd.randbytes(64)
    proof, ct = oracle.prove(cs, pk, cipher, key, nonce, 77, pt, rnd.getrandbits(250), rnd.getrandbits(250), mask=rnd.getrandbits(250))
    sig = _sig(ct, nonce, 77, pt, "big")
    assert gsc.verify({"cipher": cipher, "proof": proof, "publicSignals": sig})
    for pos in (5, 140, 170):                         # Ar, commitment, proof of knowledge
        bad = bytearray(proof); bad[pos] ^= 1
        assert not gsc.verify({"cipher": cipher, "proof": bytes(bad), "publicSignals": sig})
    assert not gsc.verify({"cipher": cipher, "proof": proof, "publicSignals": _sig(ct, nonce, 77, pt, "little")})
