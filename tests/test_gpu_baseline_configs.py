"""BASELINE.json configs at their full sizes under -m gpu (VERDICT r1 "configs_untested"): AES-128-V2 and AES-256-V2 at batch 1024,
and the mixed ChaCha/AES batch of 3 x 1024 statements through one ProveBatch call (statement i uses cipher i mod 3).  EVERY proof
is checked with the drop-in verifier libverify.so under the matching verifying key; with (r, s, mask) fixed, three statements per
cipher are compared byte for byte with the CPU oracle.  (ChaCha20-V3 at 1061 / 2048 / 4096 is covered in test_gpu_parity.py.)"""
import base64
import json
import random
from concurrent.futures import ThreadPoolExecutor

import pytest

from conftest import AES, golden_bytes

pytestmark = pytest.mark.gpu

CIPHERS = {"chacha20": (0, 32), "aes-128-ctr": (1, 16), "aes-256-ctr": (2, 32)}


@pytest.fixture(scope="module")
def everything(gsc, gsc_chacha, aes_keys):
    for name, (algo, cipher, keylen) in AES.items():
        r1cs, pk, vk = aes_keys[name]
        assert gsc.init_algorithm(algo, pk, r1cs), name
        assert gsc.init_verifier(algo, vk)
    assert gsc.init_verifier(0, golden_bytes("vk.chacha20"))
    return gsc


def _signals(cipher, ct, nonce, counter, pt):
    return ct + nonce + counter.to_bytes(4, "little" if cipher == "chacha20" else "big") + pt


def _verify_all(g, items):
    def check(it):
        cipher, proof, sig = it
        return g.verify({"cipher": cipher, "proof": base64.b64encode(proof).decode(), "publicSignals": base64.b64encode(sig).decode()})
    with ThreadPoolExecutor(16) as pool:
        return list(pool.map(check, items))


@pytest.mark.parametrize("name", list(AES))
def test_aes_batch_1024_every_proof_verifies_and_samples_match_the_oracle(everything, oracle, aes_keys, name):
    g = everything; algo, cipher, keylen = AES[name]
    r1cs, pkb, vkb = aes_keys[name]
    rnd = random.Random(500 + algo)
    n = 1024
    recs = b"".join(rnd.randbytes(32) + rnd.randbytes(12) + rnd.getrandbits(31).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
    r, s, mask = rnd.getrandbits(252), rnd.getrandbits(252), rnd.getrandbits(252)
    g.set_deterministic_randomness(r, s, mask)
    ok, proofs, lens, cts = g.prove_raw(algo, recs, n)
    g.set_deterministic_randomness(None)
    assert ok == n and set(lens) == {196}
    items = []
    for k in range(n):
        rec = recs[112 * k:112 * (k + 1)]
        items.append((cipher, proofs[196 * k:196 * k + 196], _signals(cipher, cts[64 * k:64 * k + 64], rec[32:44], int.from_bytes(rec[44:48], "little"), rec[48:])))
    res = _verify_all(g, items)
    assert all(res), [k for k, v in enumerate(res) if not v][:10]
    cs, pk = oracle.R1CS(r1cs), oracle.ProvingKey(pkb)
    for k in (0, 511, 1023):
        rec = recs[112 * k:112 * (k + 1)]
        want, want_ct = oracle.prove(cs, pk, cipher, rec[:keylen], rec[32:44], int.from_bytes(rec[44:48], "little"), rec[48:], r, s, mask)
        assert proofs[196 * k:196 * k + 196] == want and cts[64 * k:64 * k + 64] == want_ct, k
    # a second call with CSPRNG randomness: same statements, different (still verifying) proofs
    ok2, proofs2, lens2, _ = g.prove_raw(algo, recs[:112 * 64], 64)
    assert ok2 == 64 and proofs2[:196] != proofs[:196]
    assert all(_verify_all(g, [(cipher, proofs2[196 * k:196 * k + 196], items[k][2]) for k in range(64)]))


def test_mixed_batch_3x1024_through_prove_batch(everything, oracle, chacha_oracle, aes_keys):
    g = everything
    rnd = random.Random(4711)
    names = list(CIPHERS)
    n = 3 * 1024
    reqs = []
    for i in range(n):
        c = names[i % 3]; kl = CIPHERS[c][1]
        reqs.append({"cipher": c, "key": base64.b64encode(rnd.randbytes(kl)).decode(), "nonce": base64.b64encode(rnd.randbytes(12)).decode(),
                     "counter": rnd.getrandbits(30), "input": base64.b64encode(rnd.randbytes(64)).decode()})
    r, s, mask = rnd.getrandbits(252), rnd.getrandbits(252), rnd.getrandbits(252)
    g.set_deterministic_randomness(r, s, mask)
    outs = g.prove_batch(reqs)
    g.set_deterministic_randomness(None)
    assert len(outs) == n and all(isinstance(o, dict) and "proof" in o for o in outs)
    items = []
    for q, o in zip(reqs, outs):
        proof = base64.b64decode(o["proof"]["proofJson"]); ct = base64.b64decode(o["publicSignals"])
        assert len(proof) == (164 if q["cipher"] == "chacha20" else 196)
        items.append((q["cipher"], proof, _signals(q["cipher"], ct, base64.b64decode(q["nonce"]), q["counter"], base64.b64decode(q["input"]))))
    res = _verify_all(g, items)
    assert all(res), [k for k, v in enumerate(res) if not v][:10]
    # a proof of one cipher is not a proof for another statement of the same cipher
    assert not g.verify({"cipher": items[0][0], "proof": base64.b64encode(items[0][1]).decode(), "publicSignals": base64.b64encode(items[3][2]).decode()})
    oracles = {"chacha20": (chacha_oracle[0], chacha_oracle[1])}
    for name, (algo, cipher, keylen) in AES.items():
        oracles[cipher] = (oracle.R1CS(aes_keys[name][0]), oracle.ProvingKey(aes_keys[name][1]))
    for k in (0, 1, 2, 1536, 1537, 1538, n - 3, n - 2, n - 1):                     # three per cipher
        q = reqs[k]; cs, pk = oracles[q["cipher"]]
        want, want_ct = oracle.prove(cs, pk, q["cipher"], base64.b64decode(q["key"]), base64.b64decode(q["nonce"]), q["counter"], base64.b64decode(q["input"]), r, s, mask)
        assert items[k][1] == want and base64.b64decode(outs[k]["publicSignals"]) == want_ct, k
