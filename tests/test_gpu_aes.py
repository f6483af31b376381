"""GPU tests for the AES-128/256-V2 circuits (lookup tables + Groth16 commitment).

The reference ships no AES proving key, so keys come from the oracle's Setup (gnark layout, fixed seed) and parity is
"HIP path == oracle, bit for bit, for the same (r, s, mask)" plus acceptance by the oracle's verifier under the matching
vk — self-consistency, as DESIGN.md §4 states ("parity unpinned" against gnark for the commitment transcript)."""
import base64
import json
import random

import pytest

from conftest import AES, golden_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gsc_aes(gsc, aes_keys):
    for name, (algo, cipher, keylen) in AES.items():
        r1cs, pk, vk = aes_keys[name]
        assert gsc.init_algorithm(algo, pk, r1cs), name
    return gsc


@pytest.mark.parametrize("name", list(AES))
def test_aes_proof_bit_exact_vs_oracle_and_verifies(gsc_aes, oracle, aes_keys, name):
    g = gsc_aes; algo, cipher, keylen = AES[name]
    r1cs, pkb, vkb = aes_keys[name]
    cs, pk, vk = oracle.R1CS(r1cs), oracle.ProvingKey(pkb), oracle.VerifyingKey(vkb)
    rnd = random.Random(algo)
    # the reference's benchmark literal (core_test.go:265 / :275) plus a random statement
    fixed = (bytes([2] * keylen), bytes([3] * 12), 2 if algo == 1 else 10, bytes(rnd.randbytes(64)))
    cases = [fixed, (rnd.randbytes(keylen), rnd.randbytes(12), rnd.getrandbits(31), rnd.randbytes(64)), (bytes(keylen), bytes(12), 0, bytes(64))]
    r, s, mask = rnd.getrandbits(252), rnd.getrandbits(252), rnd.getrandbits(252)
    g.set_deterministic_randomness(r, s, mask)
    outs = g.prove_batch([{"cipher": cipher, "key": list(k), "nonce": list(n), "counter": c, "input": list(p)} for k, n, c, p in cases])
    g.set_deterministic_randomness(None)
    for (k, n, c, p), out in zip(cases, outs):
        proof = base64.b64decode(out["proof"]["proofJson"]); ct = base64.b64decode(out["publicSignals"])
        want, want_ct = oracle.prove(cs, pk, cipher, k, n, c, p, r, s, mask)
        assert len(proof) == 196 and proof[128:132] == b"\x00\x00\x00\x01"       # one commitment + PoK
        assert ct == want_ct == oracle.aes_ctr_xor(k, n, c, p)
        assert proof == want
        sig = ct + n + c.to_bytes(4, "big") + p                                   # AES counter is big-endian (core_test.go:204-205)
        assert oracle.verify(vk, cipher, proof, sig)
        bad = bytearray(sig); bad[70] ^= 1
        assert not oracle.verify(vk, cipher, proof, bytes(bad))


@pytest.mark.parametrize("name", list(AES))
def test_aes_csprng_batch_verifies_and_unprovable_counter_is_an_error(gsc_aes, oracle, aes_keys, name):
    g = gsc_aes; algo, cipher, keylen = AES[name]
    _, _, vkb = aes_keys[name]
    vk = oracle.VerifyingKey(vkb)
    rnd = random.Random(100 + algo)
    n = 70                                                                        # ragged: not a multiple of 64
    recs = b"".join(rnd.randbytes(32) + rnd.randbytes(12) + rnd.getrandbits(31).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
    ok, proofs, lens, cts = g.prove_raw(algo, recs, n)
    assert ok == n and set(lens) == {196}
    assert len({proofs[196 * k:196 * k + 196] for k in range(n)}) == n
    for k in (0, 63, 64, n - 1):
        rec = recs[112 * k:112 * (k + 1)]
        key, nonce, ctr, pt = rec[:keylen], rec[32:44], int.from_bytes(rec[44:48], "little"), rec[48:]
        ct = cts[64 * k:64 * k + 64]
        assert ct == oracle.aes_ctr_xor(key, nonce, ctr, pt)
        assert oracle.verify(vk, cipher, proofs[196 * k:196 * k + 196], ct + nonce + ctr.to_bytes(4, "big") + pt)
    # counter + 3 must stay below 2^32 (circuits/aesV2/aes128.go:41-53): the solver fails -> the Go side panics with a gnark error -> {}
    out = json.loads(g.prove({"cipher": cipher, "key": [1] * keylen, "nonce": [2] * 12, "counter": 0xFFFFFFFE, "input": [3] * 64}))
    assert out == {}
    # a 32-byte key on the 128-bit circuit (and vice versa) cannot be assigned to the witness
    other = 32 if keylen == 16 else 16
    assert json.loads(g.prove({"cipher": cipher, "key": [1] * other, "nonce": [2] * 12, "counter": 1, "input": [3] * 64})) == {}
    assert json.loads(g.prove({"cipher": cipher, "key": [1] * 15, "nonce": [2] * 12, "counter": 1, "input": [3] * 64})) == "key length must be 16 or 32: 15"


def test_mixed_cipher_batch(gsc_aes, gsc_chacha, oracle, chacha_oracle, aes_keys):
    # BASELINE config 5: a mixed ChaCha/AES batch through one ProveBatch call, proof i uses cipher i mod 3
    g = gsc_aes
    rnd = random.Random(77)
    names = ["chacha20", "aes-128-ctr", "aes-256-ctr"]
    reqs = []
    for i in range(9):
        c = names[i % 3]; kl = 16 if c == "aes-128-ctr" else 32
        reqs.append({"cipher": c, "key": list(rnd.randbytes(kl)), "nonce": list(rnd.randbytes(12)), "counter": rnd.getrandbits(30), "input": list(rnd.randbytes(64))})
    outs = g.prove_batch(reqs)
    vks = {"chacha20": chacha_oracle[2], "aes-128-ctr": oracle.VerifyingKey(aes_keys["aes128"][2]), "aes-256-ctr": oracle.VerifyingKey(aes_keys["aes256"][2])}
    for q, out in zip(reqs, outs):
        proof = base64.b64decode(out["proof"]["proofJson"]); ct = base64.b64decode(out["publicSignals"])
        ctr = q["counter"].to_bytes(4, "little" if q["cipher"] == "chacha20" else "big")
        assert oracle.verify(vks[q["cipher"]], q["cipher"], proof, ct + bytes(q["nonce"]) + ctr + bytes(q["input"]))


def test_concurrent_aes_callers_across_lanes_and_ciphers(gsc_aes, gsc_chacha, oracle, aes_keys):
    # libraries/core_test.go:44-111 (TestProveVerify) proves the three ciphers from goroutines at once.  Here 3 x 40 threads call
    # Prove concurrently plus two ProveBatch callers per AES cipher: single calls are gathered by the per-algorithm batchers (one worker
    # per lane), batches take whichever lane is free, small batches take the lanes-are-bases kernel — every answer must still be the
    # caller's own proof and verify under its cipher's key.
    import threading
    g = gsc_aes
    assert g.init_verifier(0, golden_bytes("vk.chacha20"))
    for name, (algo, cipher, keylen) in AES.items():
        assert g.init_verifier(algo, aes_keys[name][2])
    rnd = random.Random(90210)
    ciphers = [("chacha20", 32), ("aes-128-ctr", 16), ("aes-256-ctr", 32)]
    singles = []
    for i in range(120):
        c, kl = ciphers[i % 3]
        singles.append({"cipher": c, "key": list(rnd.randbytes(kl)), "nonce": list(rnd.randbytes(12)), "counter": rnd.getrandbits(30), "input": list(rnd.randbytes(64))})
    batches = []
    for b in range(4):
        c, kl = ciphers[1 + b % 2]
        batches.append([{"cipher": c, "key": list(rnd.randbytes(kl)), "nonce": list(rnd.randbytes(12)), "counter": rnd.getrandbits(30), "input": list(rnd.randbytes(64))} for _ in range(130 + 7 * b)])
    out_single = [None] * len(singles); out_batch = [None] * len(batches)

    def one(i):
        out_single[i] = json.loads(g.prove(singles[i]))

    def many(b):
        out_batch[b] = g.prove_batch(batches[b])
    threads = [threading.Thread(target=one, args=(i,)) for i in range(len(singles))] + [threading.Thread(target=many, args=(b,)) for b in range(len(batches))]
    for t in threads: t.start()
    for t in threads: t.join()

    def check(qo):
        q, o = qo
        assert isinstance(o, dict) and "proof" in o, o
        proof = base64.b64decode(o["proof"]["proofJson"]); ct = base64.b64decode(o["publicSignals"])
        ctr = q["counter"].to_bytes(4, "little" if q["cipher"] == "chacha20" else "big")
        sig = ct + bytes(q["nonce"]) + ctr + bytes(q["input"])
        return g.verify({"cipher": q["cipher"], "proof": base64.b64encode(proof).decode(), "publicSignals": base64.b64encode(sig).decode()})
    pairs = list(zip(singles, out_single))
    for qs, os_ in zip(batches, out_batch):
        assert len(os_) == len(qs)
        pairs += list(zip(qs, os_))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(16) as pool:
        res = list(pool.map(check, pairs))
    assert all(res), [k for k, v in enumerate(res) if not v][:10]
