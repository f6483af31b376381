"""gnark-symmetric-crypto_amd — MI355X-native Groth16 prover behind the reference's libprove C-ABI.

This package is only the Python-side loader of ``libprove.so`` (built from ``csrc/`` by ``csrc/Makefile``)
plus thin helpers that call it exactly the way a foreign-function host (node.js / Go cgo) would:
``GoSlice`` arguments by value, ``struct Prove_return`` results released with ``Free``
(reference: libraries/prover/libprove.go:17-47).  There is no Python or CPU implementation of the prover
here: if the shared library or a GPU is missing, calls fail loudly.
"""
import ctypes as C
import json
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libprove.so")
VERIFY_LIB_PATH = os.path.join(_HERE, "libverify.so")
CSRC = os.path.join(_HERE, "csrc")

CHACHA20, AES_128, AES_256 = 0, 1, 2                       # prove_impl.go:15-19
ALGORITHM_NAMES = {0: "chacha20", 1: "aes-128-ctr", 2: "aes-256-ctr"}   # prove_impl.go:21-25

EXPORTS = ["enforce_binding", "InitAlgorithm", "Free", "Prove", "ProveBatch", "gsc_prove_raw", "gsc_setup",
           "gsc_set_deterministic_randomness", "gsc_debug_prove", "gsc_debug_vector", "gsc_describe", "gsc_last_stage_ms", "gsc_last_dominant_kernel", "gsc_last_kernel_clock", "gsc_debug_field_ops", "gsc_debug_compute_h", "gsc_debug_compute_d", "gsc_debug_secret_residue", "gsc_debug_clock_trace", "gsc_debug_glv_split"]


class GoSlice(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_longlong), ("cap", C.c_longlong)]


class ProveReturn(C.Structure):
    _fields_ = [("r0", C.c_void_p), ("r1", C.c_longlong)]


def build(jobs=8):
    """Compile csrc/ into libprove.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", CSRC, "-j%d" % jobs], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libprove.so is not built: run gnark-symmetric-crypto_amd.build() (make -C %s)" % CSRC)
        L = C.CDLL(LIB_PATH)
        L.InitAlgorithm.restype = C.c_ubyte
        L.InitAlgorithm.argtypes = [C.c_ubyte, GoSlice, GoSlice]
        L.Free.argtypes = [C.c_void_p]
        L.Prove.restype = ProveReturn
        L.Prove.argtypes = [GoSlice]
        L.ProveBatch.restype = ProveReturn
        L.ProveBatch.argtypes = [GoSlice]
        L.gsc_prove_raw.restype = C.c_longlong
        L.gsc_prove_raw.argtypes = [C.c_ubyte, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gsc_setup.restype = C.c_int
        L.gsc_setup.argtypes = [GoSlice, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.gsc_set_deterministic_randomness.restype = C.c_int
        L.gsc_set_deterministic_randomness.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.gsc_debug_prove.restype = C.c_longlong
        L.gsc_debug_prove.argtypes = [GoSlice]
        L.gsc_debug_vector.restype = C.c_longlong
        L.gsc_debug_vector.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        L.gsc_describe.restype = C.c_size_t
        L.gsc_describe.argtypes = [C.c_ubyte, C.c_char_p, C.c_size_t]
        L.gsc_last_stage_ms.argtypes = [C.c_ubyte, C.POINTER(C.c_float)]
        L.gsc_last_dominant_kernel.restype = C.c_int
        L.gsc_last_dominant_kernel.argtypes = [C.c_ubyte, C.c_char_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.gsc_last_kernel_clock.restype = C.c_int
        L.gsc_last_kernel_clock.argtypes = [C.c_ubyte, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.gsc_debug_field_ops.restype = C.c_int
        L.gsc_debug_field_ops.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int]
        L.enforce_binding()
        _lib = L
    return _lib


def _slice(b: bytes):
    buf = C.create_string_buffer(b, len(b))
    return GoSlice(C.cast(buf, C.c_void_p), len(b), len(b)), buf


def init_algorithm(algorithm_id: int, proving_key: bytes, r1cs: bytes) -> bool:
    """InitAlgorithm(algorithmID, provingKey, r1cs) — libprove.go:20-23."""
    s1, k1 = _slice(proving_key)
    s2, k2 = _slice(r1cs)
    return bool(lib().InitAlgorithm(algorithm_id, s1, s2))


def _take(ret: ProveReturn) -> bytes:
    if not ret.r0:
        return b""
    out = C.string_at(ret.r0, ret.r1)
    lib().Free(ret.r0)
    return out


def prove(params) -> bytes:
    """Prove(params) — libprove.go:30-47.  params: bytes/str JSON or a dict.  Returns the raw JSON bytes."""
    if isinstance(params, dict):
        params = json.dumps(params)
    if isinstance(params, str):
        params = params.encode()
    s, keep = _slice(params)
    return _take(lib().Prove(s))


def prove_batch(params_list) -> list:
    """ProveBatch (addition): list of dicts -> list of decoded JSON results."""
    s, keep = _slice(json.dumps(params_list).encode())
    return json.loads(_take(lib().ProveBatch(s)))


def setup(r1cs: bytes, seed: bytes = None):
    """gsc_setup: Groth16 Setup for an R1CS file -> (pk bytes, vk bytes) in gnark's layouts.  seed (32 bytes) makes TEST keys and
    needs the test hooks; None = CSPRNG toxic waste."""
    s, keep = _slice(r1cs)
    pk, vk, npk, nvk = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
    if lib().gsc_setup(s, seed, C.byref(pk), C.byref(npk), C.byref(vk), C.byref(nvk)) != 0:
        raise RuntimeError("gsc_setup failed (see stdout)")
    out = C.string_at(pk.value, npk.value), C.string_at(vk.value, nvk.value)
    lib().Free(pk); lib().Free(vk)
    return out


def prove_batch_bytes(params_json: bytes) -> bytes:
    """ProveBatch on an already encoded JSON array; returns the raw JSON bytes (bench.py keeps JSON work off the timed path)."""
    s, keep = _slice(params_json)
    return _take(lib().ProveBatch(s))


def prove_raw(cipher: int, records: bytes, n: int):
    """Binary batch path: n records of 112 B {key[32], nonce[12], counter u32 LE, input[64]}.
    Returns (n_ok, proofs[n][196], lens[n], ciphertexts[n][64])."""
    proofs = C.create_string_buffer(196 * n)
    lens = (C.c_uint32 * n)()
    cts = C.create_string_buffer(64 * n)
    ok = lib().gsc_prove_raw(cipher, records, n, proofs, lens, cts)
    return ok, proofs.raw, list(lens), cts.raw


def raw_buffers(n: int):
    """Reusable output buffers for prove_raw_into: (proofs[n][196], lens[n] u32, ciphertexts[n][64])."""
    return C.create_string_buffer(196 * n), (C.c_uint32 * n)(), C.create_string_buffer(64 * n)


def prove_raw_into(cipher: int, records: bytes, n: int, proofs, lens, cts) -> int:
    """prove_raw without per-call allocations and copies: fills caller-owned buffers (raw_buffers), returns the number of proofs produced."""
    return lib().gsc_prove_raw(cipher, records, n, proofs, lens, cts)


def set_deterministic_randomness(r=None, s=None, mask=0):
    """TEST HOOK: fix (r, s, mask) as integers; None restores the CSPRNG.  Needs GSC_ENABLE_TEST_HOOKS=1 in the environment
    before the library is loaded."""
    if r is None:
        rc = lib().gsc_set_deterministic_randomness(None, None, None)
    else:
        rc = lib().gsc_set_deterministic_randomness(int(r).to_bytes(32, "big"), int(s).to_bytes(32, "big"), int(mask).to_bytes(32, "big"))
    if rc != 0:
        raise RuntimeError("test hooks are disabled: set GSC_ENABLE_TEST_HOOKS=1 before loading libprove.so")


def debug_prove(params: dict):
    """TEST HOOK: run one proof and return the device pipeline's intermediate vectors as lists of ints.
    W/A/B/C come back in Montgomery form and are converted here with Python integers."""
    r_mod = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    rinv = pow(1 << 256, -1, r_mod)
    s, keep = _slice(json.dumps(params).encode())
    if lib().gsc_debug_prove(s) != 0:
        raise RuntimeError("debug prove failed")
    out = {}
    for which, name, mont in ((0, "W", True), (1, "A", True), (2, "B", True), (3, "C", True), (4, "h", False)):
        n = lib().gsc_debug_vector(which, None, 0)
        buf = C.create_string_buffer(32 * n)
        lib().gsc_debug_vector(which, buf, 32 * n)
        vals = [int.from_bytes(buf.raw[32 * i:32 * i + 32], "little") for i in range(n)]
        if mont:
            vals = [v * rinv % r_mod for v in vals]
        out[name] = vals
    return out


def describe(algorithm_id: int) -> str:
    buf = C.create_string_buffer(4096)
    lib().gsc_describe(algorithm_id, buf, 4096)
    return buf.value.decode()


def last_stage_ms(algorithm_id: int):
    arr = (C.c_float * 4)()
    if lib().gsc_last_stage_ms(algorithm_id, arr) != 0:
        return None
    return dict(zip(("witness", "quotient", "msm", "assembly"), list(arr)))


def last_dominant_kernel(algorithm_id: int):
    """(kernel name, milliseconds, statements proved, padded columns, Z bases per proof) of the dominant kernel in the batch that
    finished last: the Z-table MSM for batch calls, the resident witness solver for calls on the latency path."""
    name = C.create_string_buffer(96)
    ms, st, cols, nb = C.c_float(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    if lib().gsc_last_dominant_kernel(algorithm_id, name, 96, C.byref(ms), C.byref(st), C.byref(cols), C.byref(nb)) != 0:
        return None
    return name.value.decode(), float(ms.value), st.value, cols.value, nb.value


def debug_secret_residue(algorithm_id: int) -> int:
    """TEST HOOK: bytes of a finished call's secrets still non-zero in device memory (0 expected; -1: hooks disabled / error)."""
    lib().gsc_debug_secret_residue.restype = C.c_longlong
    lib().gsc_debug_secret_residue.argtypes = [C.c_ubyte]
    return int(lib().gsc_debug_secret_residue(algorithm_id))


def debug_clock_trace(n: int, interval_us: int):
    """TEST HOOK: [(seconds since the first sample, shader clock in MHz over the interval before it)] from a resident one-wave sampler; blocks
    for n x interval_us microseconds — run it in a thread beside the calls to be observed."""
    buf = (C.c_ulonglong * (2 * n))()
    lib().gsc_debug_clock_trace.restype = C.c_int
    lib().gsc_debug_clock_trace.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong)]
    if lib().gsc_debug_clock_trace(n, interval_us, buf) != 0:
        raise RuntimeError("gsc_debug_clock_trace failed")
    v = list(buf)
    return [((v[2 * i] - v[0]) / 1e8, 100.0 * (v[2 * i + 1] - v[2 * i - 1]) / max(1, v[2 * i] - v[2 * i - 2])) for i in range(1, n)]


def last_kernel_clock(algorithm_id: int):
    """(shader clock in MHz during the last batch's Z-table kernel — 0.0 when not measured —, digit windows of the Z set)."""
    mhz, nwin = C.c_float(0), C.c_int(0)
    if lib().gsc_last_kernel_clock(algorithm_id, C.byref(mhz), C.byref(nwin)) != 0:
        return None
    return float(mhz.value), int(nwin.value)


def last_msm_z_kernel(algorithm_id: int):
    """(milliseconds, columns in the launch, bases per proof) of the dominant kernel in the last batch (older tools)."""
    k = last_dominant_kernel(algorithm_id)
    return (k[1], k[3], k[4]) if k else None


def served(algorithm_id: int):
    """[(calls, statements)] per engine replica (GSC_DEVICES), parsed from gsc_describe."""
    d = describe(algorithm_id)
    tail = d.split("served(calls/statements)=")[1].split()[0]
    return [tuple(int(x) for x in part.split("/")) for part in tail.split(",")]


def debug_field_ops(field: int, op: int, a, b, chain=1):
    """TEST HOOK: a, b lists of ints (canonical residues) -> list of ints computed by the device's radix-2^29 field code."""
    n = len(a)
    out = C.create_string_buffer(32 * n)
    rc = lib().gsc_debug_field_ops(field, op, b"".join(int(x).to_bytes(32, "little") for x in a), b"".join(int(x).to_bytes(32, "little") for x in b), out, n, chain)
    if rc:
        raise RuntimeError("gsc_debug_field_ops failed")
    return [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(n)]


def debug_glv_split(k: int):
    """TEST HOOK (host arithmetic, no GPU): k -> (k1, k2) with k = k1 + k2 * lambda (mod r), the split of csrc/glv.hpp."""
    L = lib()
    L.gsc_debug_glv_split.restype = C.c_int
    L.gsc_debug_glv_split.argtypes = [C.c_char_p, C.c_void_p]
    out = C.create_string_buffer(44)
    if L.gsc_debug_glv_split(int(k).to_bytes(32, "little"), out) != 0:
        raise RuntimeError("gsc_debug_glv_split failed")
    k1 = int.from_bytes(out.raw[:20], "little"); k2 = int.from_bytes(out.raw[20:40], "little"); neg = out.raw[40]
    return (-k1 if neg & 1 else k1), (-k2 if neg & 2 else k2)


def debug_compute_h(algorithm_id: int, abc_be: bytes, m: int) -> bytes:
    """TEST HOOK: computeH on 64 columns of caller-supplied a|b|c ([m][64] big-endian each) -> [domain][64] little-endian rows in
    bit-reversed order (raw bytes)."""
    L = lib()
    L.gsc_debug_compute_h.restype = C.c_longlong
    L.gsc_debug_compute_h.argtypes = [C.c_ubyte, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    n = L.gsc_debug_compute_h(algorithm_id, None, 0, None, 0)
    if n <= 0:
        raise RuntimeError("algorithm not initialised")
    assert len(abc_be) == 3 * m * 64 * 32
    out = C.create_string_buffer(n * 64 * 32)
    if L.gsc_debug_compute_h(algorithm_id, abc_be, m, out, len(out)) != n:
        raise RuntimeError("gsc_debug_compute_h failed")
    return out.raw


def debug_compute_d(algorithm_id: int, ab_be: bytes, m: int) -> bytes:
    """TEST HOOK: the evaluation-form quotient kernels on 64 columns of caller-supplied a|b ([m][64] big-endian each) ->
    [domain][64] little-endian rows, row i = A(zeta w^i) B(zeta w^i) 2^261 mod r (raw bytes)."""
    L = lib()
    L.gsc_debug_compute_d.restype = C.c_longlong
    L.gsc_debug_compute_d.argtypes = [C.c_ubyte, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    n = L.gsc_debug_compute_d(algorithm_id, None, 0, None, 0)
    if n <= 0:
        raise RuntimeError("algorithm not initialised")
    assert len(ab_be) == 2 * m * 64 * 32
    out = C.create_string_buffer(n * 64 * 32)
    if L.gsc_debug_compute_d(algorithm_id, ab_be, m, out, len(out)) != n:
        raise RuntimeError("gsc_debug_compute_d failed")
    return out.raw


# ---- libverify (CPU-side, libraries/verifier/libverify.go:14-17) ----
_vlib = None


def verify_lib():
    global _vlib
    if _vlib is None:
        if not os.path.exists(VERIFY_LIB_PATH):
            raise RuntimeError("libverify.so is not built: make -C %s" % CSRC)
        L = C.CDLL(VERIFY_LIB_PATH)
        L.Verify.restype = C.c_ubyte
        L.Verify.argtypes = [GoSlice]
        L.InitVerifier.restype = C.c_ubyte
        L.InitVerifier.argtypes = [C.c_ubyte, GoSlice]
        _vlib = L
    return _vlib


def init_verifier(algorithm_id: int, verifying_key: bytes) -> bool:
    s, keep = _slice(verifying_key)
    return bool(verify_lib().InitVerifier(algorithm_id, s))


def verify(params) -> bool:
    """Verify(params) — libverify.go:14-17.  params: bytes/str JSON or a dict with cipher / proof / publicSignals."""
    if isinstance(params, dict):
        params = json.dumps({k: (list(v) if isinstance(v, (bytes, bytearray)) else v) for k, v in params.items()})
    if isinstance(params, str):
        params = params.encode()
    s, keep = _slice(params)
    return bool(verify_lib().Verify(s))
