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


def _twist_point_outside_g2():
    """A point of E'(Fp2): y^2 = x^3 + 3/(9+u) that is NOT in the r-torsion subgroup (a random twist point is in G2 with
    probability r / #E' ~ 2^-254), compressed the gnark-crypto way (X.A1 | X.A0, flags from y)."""
    P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    mul = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    inv82 = pow(82, -1, P); bt = (27 * inv82 % P, -3 * inv82 % P)

    def sqrt_fp(a):
        s = pow(a, (P + 1) // 4, P)
        return s if s * s % P == a % P else None

    def sqrt_fp2(a):      # norm method, p = 3 mod 4
        if a[1] == 0:
            s = sqrt_fp(a[0])
            if s is not None:
                return (s, 0)
            s = sqrt_fp(-a[0] % P)
            return None if s is None else (0, s)
        n = sqrt_fp((a[0] * a[0] + a[1] * a[1]) % P)
        if n is None:
            return None
        for t in ((a[0] + n) * pow(2, -1, P) % P, (a[0] - n) * pow(2, -1, P) % P):
            x0 = sqrt_fp(t)
            if x0:
                x = (x0, a[1] * pow(2 * x0, -1, P) % P)
                if mul(x, x) == (a[0] % P, a[1] % P):
                    return x
        return None
    k = 1
    while True:
        x = (k, 1); x3 = mul(mul(x, x), x); y = sqrt_fp2(((x3[0] + bt[0]) % P, (x3[1] + bt[1]) % P))
        if y is not None:
            break
        k += 1
    large = (y[1] > (P - 1) // 2) if y[1] else (y[0] > (P - 1) // 2)
    enc = bytearray(x[1].to_bytes(32, "big") + x[0].to_bytes(32, "big")); enc[0] |= 0xC0 if large else 0x80
    return bytes(enc)


def test_g2_points_outside_the_r_torsion_subgroup_are_rejected(gsc):
    # gnark-crypto's decoder checks subgroup membership (and so does groth16.Verify); a curve-equation check alone accepts
    # twist points of the cofactor subgroup.  A verifying key with such a gamma must not load; a proof with such a Bs is false.
    vk = bytearray(golden_bytes("vk.chacha20")); bad = _twist_point_outside_g2()
    assert gsc.init_verifier(0, bytes(vk))
    vk[128:192] = bad                                  # alpha(32) beta1(32) beta2(64) | gamma2 at 128
    assert not gsc.init_verifier(0, bytes(vk))
    assert gsc.init_verifier(0, golden_bytes("vk.chacha20"))
    proof = bytearray(bytes.fromhex(KAT["proofs"][(0, 0)])); proof[32:96] = bad
    sig = _sig(KAT["ciphertext"], KAT["nonce"], KAT["counter"], KAT["input"])
    assert not gsc.verify({"cipher": "chacha20", "proof": bytes(proof), "publicSignals": sig})


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
