#!/usr/bin/env python3
"""Key generation for the reference's circuits with the product's Groth16 Setup (gsc_setup) — what `go run keygen.go` does with
gnark (reference keygen.go:341-352, :380-391, :419-430) for the R1CS files it ships.  Toxic waste comes from the OS CSPRNG.

  python tools/gsc_keygen.py <r1cs file> <pk out> <vk out>
e.g. python tools/gsc_keygen.py r1cs.aes128 pk.aes128 vk.aes128      (needs a GPU: the group elements are computed there)"""
import lzma
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if len(sys.argv) != 4:
        raise SystemExit(__doc__)
    import gsc_loader
    g = gsc_loader.load()
    path = sys.argv[1]
    r1cs = lzma.open(path).read() if path.endswith(".xz") else open(path, "rb").read()
    pk, vk = g.setup(r1cs)
    open(sys.argv[2], "wb").write(pk)
    open(sys.argv[3], "wb").write(vk)
    print("pk %d bytes -> %s, vk %d bytes -> %s" % (len(pk), sys.argv[2], len(vk), sys.argv[3]))


if __name__ == "__main__":
    main()
