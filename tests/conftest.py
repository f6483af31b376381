import lzma
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GSC_ENABLE_TEST_HOOKS"] = "1"      # read once by libprove.so when it is loaded (include/libprove.h, TEST HOOKS)
# The GPU session keeps ChaCha20, AES-128 and AES-256 loaded at once and starts child processes with their own tables beside them:
# smaller quotient tables than the defaults (c = 14 / 12 instead of 15 / 13; latency layout 8.6 GB + 2 x 4.3 GB) leave room on the
# 288 GB device.  Results do not depend on the table sizes (tests/test_gpu_parity.py checks that).
os.environ.setdefault("GSC_Z_TABLE_GB", "24")
os.environ.setdefault("GSC_W_TABLE_GB", "8")
os.environ.setdefault("GSC_FEW_Z_GB", "9")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_bytes(name):
    path = os.path.join(GOLDEN, name)
    if os.path.exists(path + ".xz"):
        return lzma.open(path + ".xz").read()
    return open(path, "rb").read()


# Known-answer vectors for the ChaCha20-V3 circuit with the reference's shipped pk/r1cs.
# Input = the reference's benchmark literal (libraries/core_test.go:285).
KAT = {
    "key": bytes([2] * 32),
    "nonce": bytes([3] * 12),
    "counter": 3,
    "input": bytes.fromhex("a3f7e592aeda1507a7f51b35812dfc50a263d5a6d2df625e563b02e49c08bf30d0e7483f5b13ff079532224ee8fbc31a"
                           "b1899b18e453d36d9793a8355eb0dee9"),
    "ciphertext": bytes.fromhex("e11ef0b2e6d3e450ab1a3509c0a6a2c79ece1376a8a0a6c09603f26b15b106dee60711d709ca21ac7e545f7d2c040f1b"
                                "a1933d4eff4823a142da7aaffa483224"),
    "sha256_W": "1b458bca80f40f2b9f0b6c6fc1e3b3bf1ca0a386eaaf7963f3387b3e5cd3dffa",
    "sha256_abc": "aef05402d37c97ed2c6840de6c10c484b1f14b1dff03510b7404d7a913dfb3b9",
    "sha256_h": "35037f465d7606a5dc8b86c17df1e0224f40d8e48e1543fb6e7ffdd3c23ecf9f",
    "proofs": {
        (0, 0): "c21d45c12d5fd77bb5211e85938971448e56ce09d02af6c6b889e1edf0c1c39ca2829d5ce21612af4ef03c6a28c380d0348fd790adb3b0"
                "027200ceeac421481a24e3e0710da3d26c41df970ed0e50e6647223d1f9f904b8ed6a3288011dd97d2e8414464d817570d0feb88fb12b7"
                "41c48361a2da0725bde1bd8a66415f0f6f95000000004000000000000000000000000000000000000000000000000000000000000000",
        (0x1234567, 0xabcdef0123456789abcdef):
                "ea49984df7447a7faa42e256b4ba77b18c134f87a9f8938bcc7722f9524b81f0a81727420993af92a92d8e28650e858ff01fbded7396dd"
                "3a41033abac4d97d5c0097d13efc1142d0730950c26c4c55037bb1dc96b9e3422eae0729ef36af113fd8fa21e5ff66d3c144a6d75436c9"
                "b87866463d76e98b68310f3bce6130d699a6000000004000000000000000000000000000000000000000000000000000000000000000",
    },
}


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def chacha_oracle(oracle):
    cs = oracle.R1CS(golden_bytes("r1cs.chacha20"))
    pk = oracle.ProvingKey(golden_bytes("pk.chacha20"))
    vk = oracle.VerifyingKey(golden_bytes("vk.chacha20"))
    return cs, pk, vk


@pytest.fixture(scope="session")
def gsc():
    import gsc_loader
    return gsc_loader.load()


@pytest.fixture(scope="session")
def gsc_chacha(gsc):
    """libprove with ChaCha20-V3 initialised from the reference's shipped key files (GPU only)."""
    assert gsc.init_algorithm(gsc.CHACHA20, golden_bytes("pk.chacha20"), golden_bytes("r1cs.chacha20"))
    return gsc


AES = {"aes128": (1, "aes-128-ctr", 16), "aes256": (2, "aes-256-ctr", 32)}


@pytest.fixture(scope="session")
def aes_keys(oracle):
    """TEST keys for the AES-V2 circuits from the oracle's Setup (the reference ships no pk.aes128 / pk.aes256:
    .MISSING_LARGE_BLOBS).  Deterministic in the seed; cached under build/keys."""
    out = {}
    cache = os.path.join(ROOT, "build", "keys")
    os.makedirs(cache, exist_ok=True)
    for name, (algo, cipher, keylen) in AES.items():
        r1cs = golden_bytes("r1cs." + name)
        pkp, vkp = os.path.join(cache, "pk." + name), os.path.join(cache, "vk." + name)
        if not (os.path.exists(pkp) and os.path.exists(vkp)):
            pk, vk = oracle.setup(oracle.R1CS(r1cs), bytes([algo] * 32))
            open(pkp, "wb").write(pk); open(vkp, "wb").write(vk)
        out[name] = (r1cs, open(pkp, "rb").read(), open(vkp, "rb").read())
    return out
