"""CPU tests of the drop-in boundary: libprove.so builds, loads, exports every symbol include/libprove.h declares,
decodes JSON like encoding/json, and refuses to work without a GPU (no CPU fallback, no oracle behind it)."""
import json
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT, golden_bytes


def test_library_exports_every_declared_symbol(gsc):
    hdr = open(os.path.join(ROOT, "include", "libprove.h")).read()
    declared = set(re.findall(r"extern\s+[\w\s\*]+?\b(\w+)\s*\(", hdr))
    assert {"enforce_binding", "InitAlgorithm", "Free", "Prove"} <= declared
    lib = gsc.lib()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert declared == set(gsc.EXPORTS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", gsc.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert exported == declared          # nothing else leaks out of the shared object


def test_product_never_links_the_oracle(gsc):
    out = subprocess.check_output(["ldd", gsc.LIB_PATH]).decode()
    assert "liboracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gnark-symmetric-crypto_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_test_hooks_refuse_without_the_environment_opt_in(gsc):
    # a production host never sets GSC_ENABLE_TEST_HOOKS: fixing (r, s, mask) or reading intermediates must be impossible there
    code = (
        "import os, sys, ctypes as C; os.environ.pop('GSC_ENABLE_TEST_HOOKS', None)\n"
        "sys.path.insert(0, %r); import gsc_loader; g = gsc_loader.load(); L = g.lib()\n"
        "one = (1).to_bytes(32, 'big')\n"
        "assert L.gsc_set_deterministic_randomness(one, one, one) == -1\n"
        "assert L.gsc_debug_vector(0, None, 0) == -1\n"
        "assert L.gsc_debug_field_ops(0, 0, one, one, C.create_string_buffer(32), 1, 1) == -1\n"
        "s, keep = g._slice(b'{}'); assert L.gsc_debug_prove(s) == -1\n"
        "L.gsc_debug_compute_h.restype = C.c_longlong; assert L.gsc_debug_compute_h(0, None, 0, None, 0) == -1\n"
        "L.gsc_debug_compute_d.restype = C.c_longlong; assert L.gsc_debug_compute_d(0, None, 0, None, 0) == -1\n"
        "pk, vk, a, b = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()\n"
        "s2, keep2 = g._slice(b'x'); assert L.gsc_setup(s2, one, C.byref(pk), C.byref(a), C.byref(vk), C.byref(b)) == -1 and not pk.value\n"
        "try:\n    g.set_deterministic_randomness(1, 1)\nexcept RuntimeError: print('refused')\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "GSC_ENABLE_TEST_HOOKS"}
    out = subprocess.check_output([sys.executable, "-c", code], env=env).decode()
    assert "refused\n" in out and out.count("test hooks are disabled") == 8      # (C stdio and Python flush in their own order)
    assert gsc.lib().gsc_set_deterministic_randomness(None, None, None) == 0      # this process opted in (conftest.py)


def test_deeply_nested_json_is_an_error_not_a_stack_overflow(gsc):
    # 20 KB of '[' used to overflow a 1 MiB thread stack inside the recursive-descent parser (and the JsonValue destructor)
    import threading
    res = {}

    def work():
        for depth in (65, 9999, 200000):
            doc = b"[" * depth + b"]" * depth
            res[depth] = json.loads(gsc.prove(doc))
            res[("obj", depth)] = json.loads(gsc.prove(b'{"a":' * depth + b"1" + b"}" * depth))
        res["ok"] = json.loads(gsc.prove(b"[" * 64 + b"]" * 64))
    old = threading.stack_size(256 * 1024)
    try:
        t = threading.Thread(target=work); t.start(); t.join()
    finally:
        threading.stack_size(old)
    for depth in (65, 9999, 200000):
        assert res[depth] == {"Offset": 64} and res[("obj", depth)] == {"Offset": 5 * 64}      # json.SyntaxError-shaped: "exceeded max depth"
    assert res["ok"]["Value"] == "array"             # 64 levels still parse (and are then a type error, like any array)


def test_without_gpu_init_fails_loudly_and_prove_reports_uninitialised(gsc, capfd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert gsc.init_algorithm(gsc.CHACHA20, golden_bytes("pk.chacha20"), golden_bytes("r1cs.chacha20")) is False
    out = json.loads(gsc.prove({"cipher": "chacha20", "key": [0] * 32, "nonce": [0] * 12, "counter": 0, "input": [0] * 64}))
    assert out == "proving params are not initialized for cipher: chacha20"
    assert gsc.prove_raw(gsc.CHACHA20, bytes(112), 1)[0] == -1
    with pytest.raises(RuntimeError):                  # Setup computes its group elements on the GPU: no CPU path either
        gsc.setup(golden_bytes("r1cs.chacha20"))


def test_json_decoding_rules_follow_encoding_json(gsc):
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU tests")
    p = gsc.prove
    # decode errors are reported before the cipher lookup, as an UnmarshalTypeError-shaped object
    e = json.loads(p(b'{"cipher":"chacha20","key":"not base64!","nonce":[],"counter":1,"input":[]}'))
    assert isinstance(e, int)                                           # base64.CorruptInputError marshals as its offset
    e = json.loads(p(b'{"cipher":"chacha20","key":[1,2,256],"nonce":[],"counter":1,"input":[]}'))
    assert e["Value"] == "number 256" and e["Field"] == "key"
    e = json.loads(p(b'{"cipher":"chacha20","key":[1],"nonce":[],"counter":-1,"input":[]}'))
    assert e["Value"] == "number -1" and e["Field"] == "counter"
    e = json.loads(p(b'{"cipher":"chacha20","key":[1],"nonce":[],"counter":1.5,"input":[]}'))
    assert e["Value"] == "number 1.5"
    e = json.loads(p(b'{"cipher":5}'))
    assert e["Value"] == "number" and e["Field"] == "cipher"
    assert json.loads(p(b'[1,2]'))["Value"] == "array"
    assert json.loads(p(b'')) == {"Offset": 0}
    assert json.loads(p(b'{"cipher":"chacha20"} x')) == {"Offset": 22}
    assert json.loads(p(b'null')) == "runtime error: invalid memory address or nil pointer dereference"
    assert json.loads(p(b'{"cipher":"nope"}')) == "could not find prover fornope"
    assert json.loads(p(b'{"CIPHER":"<x>"}')) == "could not find prover for<x>"      # HTML-safe escaping is transparent to a JSON reader


def test_solver_program_builder_and_key_parser_on_reference_files():
    # Host-side decoders are compiled into a tiny CLI so they can be exercised without a GPU.
    exe = os.path.join(ROOT, "build", "host_decode_check")
    src = os.path.join(ROOT, "tools", "host_decode_check.cpp")
    csrc = os.path.join(ROOT, "gnark-symmetric-crypto_amd", "csrc")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", csrc, src, os.path.join(csrc, "formats.cpp"), "-o", exe])
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        for name in ("r1cs.chacha20", "r1cs.aes128", "r1cs.aes256", "pk.chacha20"):
            open(os.path.join(td, name), "wb").write(golden_bytes(name))
        out = subprocess.check_output([exe, td]).decode()
    rep = dict(l.split("=", 1) for l in out.split())
    assert rep["chacha.instr"] == "23954" and rep["chacha.wires"] == "23281" and rep["chacha.constraints"] == "23617" and rep["chacha.inversions"] == "0"
    assert rep["aes128.instr"] == "78430" and rep["aes128.inversions"] == "2080" and rep["aes128.tables"] == "5"
    assert rep["aes256.instr"] == "104106" and rep["aes256.inversions"] == "2384"
    assert rep["pk.A"] == "22001" and rep["pk.B"] == "12529" and rep["pk.Z"] == "32767" and rep["pk.K"] == "22128" and rep["pk.n"] == "32768"
    assert rep["truncated.r1cs"] == "rejected" and rep["truncated.pk"] == "rejected"
    # the latency-path layout (build_few_program): every op once, every level only reads wires of earlier levels — also with the
    # check-only constraints moved into the last level — and every wire gets solved
    for c in ("chacha", "aes128", "aes256"):
        assert rep[c + ".few_ops"] == rep[c + ".few_expected"] and rep[c + ".few_bad"] == "0" and rep[c + ".few_unsolved"] == "0", {k: v for k, v in rep.items() if k.startswith(c + ".few")}
    assert rep["chacha.few_count_ops"] == "0" and rep["aes128.few_count_ops"] == "5"
    assert int(rep["chacha.few_last_level"]) > 12000          # the 12 865 constraints that solve nothing run at the end
