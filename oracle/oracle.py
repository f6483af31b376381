"""TEST ORACLE — ctypes binding of oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (gnark-symmetric-crypto_amd/) never does.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        # the GPU boxes expose hundreds of hardware threads but give a process a share of ~16: do not let OpenMP oversubscribe and spin
        os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        L = C.CDLL(_LIB)
        vp, sz, u8p, u32, i32 = C.c_void_p, C.c_size_t, C.c_char_p, C.c_uint32, C.c_int
        L.orc_r1cs_new.restype = vp; L.orc_r1cs_new.argtypes = [u8p, sz]
        L.orc_r1cs_free.argtypes = [vp]
        L.orc_r1cs_info.restype = sz; L.orc_r1cs_info.argtypes = [vp, i32]
        L.orc_r1cs_levels_are_permutation.argtypes = [vp]
        L.orc_pk_new.restype = vp; L.orc_pk_new.argtypes = [u8p, sz]
        L.orc_pk_free.argtypes = [vp]
        L.orc_pk_info.restype = sz; L.orc_pk_info.argtypes = [vp, i32]
        L.orc_vk_new.restype = vp; L.orc_vk_new.argtypes = [u8p, sz]
        L.orc_vk_free.argtypes = [vp]
        L.orc_vk_nk.restype = sz; L.orc_vk_nk.argtypes = [vp]
        L.orc_chacha20_xor.argtypes = [u8p, u8p, u32, u8p, u8p, sz]
        L.orc_chacha20_block.argtypes = [u8p, u8p, u32, u8p]
        L.orc_aes_ctr_xor.argtypes = [u8p, i32, u8p, u32, u8p, u8p, sz]
        L.orc_aes_encrypt_block.argtypes = [u8p, i32, u8p, u8p]
        L.orc_sha256.argtypes = [u8p, sz, u8p]
        L.orc_expand_message_xmd.argtypes = [u8p, sz, u8p, sz, u8p, sz]
        L.orc_prove.restype = i32
        L.orc_prove.argtypes = [vp, vp, i32, u8p, u8p, u32, u8p, u8p, u8p, u8p, u8p, C.POINTER(sz), u8p, vp, vp, vp, vp, vp]
        L.orc_solve.restype = C.c_long
        L.orc_solve.argtypes = [vp, i32, u8p, u8p, u32, u8p, u8p, u8p, u8p, vp, vp, vp, vp]
        L.orc_verify.restype = i32; L.orc_verify.argtypes = [vp, i32, u8p, sz, u8p, sz]
        L.orc_setup.restype = i32
        L.orc_setup.argtypes = [vp, u8p, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz)]
        L.orc_free.argtypes = [vp]
        L.orc_compute_h.restype = i32; L.orc_compute_h.argtypes = [vp, u8p, u8p, u8p, sz, u8p]
        L.orc_pairing_selftest.restype = i32
        L.orc_field_const.argtypes = [i32, u8p]
        L.orc_init()
        _lib = L
    return _lib


CIPHERS = {"chacha20": 0, "aes-128-ctr": 1, "aes-256-ctr": 2}


class R1CS:
    def __init__(self, data: bytes):
        self.h = lib().orc_r1cs_new(data, len(data))
        if not self.h:
            raise ValueError("oracle: cannot parse r1cs")
        g = lambda w: lib().orc_r1cs_info(self.h, w)
        self.n_wires, self.n_constraints, self.n_public, self.n_secret = g(0), g(1), g(2), g(3)
        self.n_instr, self.n_levels, self.n_calldata, self.n_coeff = g(4), g(5), g(6), g(7)
        self.n_commit, self.n_committed, self.commit_wire = g(8), g(9), g(10)

    def levels_are_permutation(self):
        return bool(lib().orc_r1cs_levels_are_permutation(self.h))

    def solve(self, cipher, key, nonce, counter, pt, mask=None, commit=None, dump=True):
        """Witness generation only. Returns (rc, ct, W, A, B, C) with vectors as bytes (32-byte BE each)."""
        ct = C.create_string_buffer(64)
        W = C.create_string_buffer(32 * self.n_wires) if dump else None
        A = C.create_string_buffer(32 * self.n_constraints) if dump else None
        B = C.create_string_buffer(32 * self.n_constraints) if dump else None
        Cc = C.create_string_buffer(32 * self.n_constraints) if dump else None
        rc = lib().orc_solve(self.h, CIPHERS[cipher], key, nonce, counter, pt, mask, commit, ct, W, A, B, Cc)
        if not dump:
            return rc, ct.raw
        return rc, ct.raw, W.raw, A.raw, B.raw, Cc.raw


class ProvingKey:
    def __init__(self, data: bytes):
        self.h = lib().orc_pk_new(data, len(data))
        if not self.h:
            raise ValueError("oracle: cannot parse pk")
        g = lambda w: lib().orc_pk_info(self.h, w)
        self.n, self.nA, self.nB, self.nZ, self.nK, self.nB2, self.n_wires, self.n_ck, self.n_basis = [g(i) for i in range(9)]


class VerifyingKey:
    def __init__(self, data: bytes):
        self.h = lib().orc_vk_new(data, len(data))
        if not self.h:
            raise ValueError("oracle: cannot parse vk")
        self.nK = lib().orc_vk_nk(self.h)


def _be32(v):
    if v is None:
        return None
    if isinstance(v, int):
        return v.to_bytes(32, "big")
    return bytes(v)


def prove(cs: R1CS, pk: ProvingKey, cipher, key, nonce, counter, pt, r=0, s=0, mask=0, dump=False):
    """Returns (proof_bytes, ciphertext[, dumps]); raises on failure."""
    out = C.create_string_buffer(512)
    n = C.c_size_t(0)
    ct = C.create_string_buffer(64)
    bufs = [None] * 5
    if dump:
        bufs = [C.create_string_buffer(32 * cs.n_wires)] + [C.create_string_buffer(32 * cs.n_constraints) for _ in range(3)] + [C.create_string_buffer(32 * pk.n)]
    rc = lib().orc_prove(cs.h, pk.h, CIPHERS[cipher], bytes(key), bytes(nonce), counter, bytes(pt), _be32(r), _be32(s), _be32(mask),
                         out, C.byref(n), ct, *bufs)
    if rc:
        raise RuntimeError("oracle prove failed rc=%d" % rc)
    if dump:
        return out.raw[: n.value], ct.raw, dict(zip("WABCh", [b.raw for b in bufs]))
    return out.raw[: n.value], ct.raw


def compute_h(pk: ProvingKey, a_be: bytes, b_be: bytes, c_be: bytes) -> bytes:
    """computeH on canonical big-endian vectors (len m*32 each) -> n*32 bytes, natural coefficient order."""
    m = len(a_be) // 32
    out = C.create_string_buffer(32 * pk.n)
    if lib().orc_compute_h(pk.h, a_be, b_be, c_be, m, out):
        raise RuntimeError("oracle compute_h failed")
    return out.raw


def verify(vk: VerifyingKey, cipher, proof: bytes, public_signals: bytes) -> bool:
    return bool(lib().orc_verify(vk.h, CIPHERS[cipher], proof, len(proof), public_signals, len(public_signals)))


def chacha20_xor(key, nonce, counter, data):
    out = C.create_string_buffer(len(data))
    lib().orc_chacha20_xor(bytes(key), bytes(nonce), counter, bytes(data), out, len(data))
    return out.raw


def aes_ctr_xor(key, nonce, counter, data):
    out = C.create_string_buffer(len(data))
    lib().orc_aes_ctr_xor(bytes(key), len(key), bytes(nonce), counter, bytes(data), out, len(data))
    return out.raw


def setup(cs: R1CS, seed: bytes):
    """TEST keys in gnark layout (pk bytes, vk bytes) for a constraint system, deterministic in `seed` (32 bytes)."""
    pk, vk = C.c_void_p(), C.c_void_p()
    npk, nvk = C.c_size_t(), C.c_size_t()
    rc = lib().orc_setup(cs.h, bytes(seed), C.byref(pk), C.byref(npk), C.byref(vk), C.byref(nvk))
    if rc:
        raise RuntimeError("oracle setup failed rc=%d" % rc)
    out = C.string_at(pk, npk.value), C.string_at(vk, nvk.value)
    lib().orc_free(pk); lib().orc_free(vk)
    return out
