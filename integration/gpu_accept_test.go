// gpu_accept_test.go — acceptance of GPU-made proofs by gnark itself, for the circuits whose proving keys the reference
// does not ship (AES-128-V2 / AES-256-V2: .MISSING_LARGE_BLOBS).  This is the test that PINS the Groth16 commitment
// transcript ("bsb22-commitment" / "G16-BSB22", SURVEY.md App. H) the GPU prover implements from the protocol description:
// gnark's own Setup makes the keys, the GPU library proves, gnark's own Verify checks — no code of this repository on the
// verifying side.  ChaCha20-V3 is included as the control (its parity is already pinned by vk.chacha20).
//
// Where it goes: libraries/gpu_accept_test.go of the reference checkout, next to core_test.go, with prove_gpu.go
// installed (see that file).  Run on a box with Go, the module cache and an MI355X:
//
//	cd libraries && go test -tags gsc_gpu -run TestGPUAccept -v
//
// Source only here: the image this repository is developed in has no Go toolchain.
//
//go:build gsc_gpu

package libraries

import (
	"bytes"
	"crypto/rand"
	"encoding/binary"
	"encoding/json"
	"os"
	"testing"

	prover "gnark-symmetric-crypto/libraries/prover/impl"
	verifier "gnark-symmetric-crypto/libraries/verifier/impl"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/frontend"
)

func acceptAES(t *testing.T, id uint8, cipher string, keyLen int, r1csPath string) {
	r1csBytes, err := os.ReadFile(r1csPath)
	if err != nil {
		t.Fatal(err)
	}
	cs := groth16.NewCS(ecc.BN254)
	if _, err = cs.ReadFrom(bytes.NewReader(r1csBytes)); err != nil {
		t.Fatal(err)
	}
	pk, vk, err := groth16.Setup(cs) // gnark's Setup: keygen.go:384 / :423
	if err != nil {
		t.Fatal(err)
	}
	var pkBuf bytes.Buffer
	if _, err = pk.WriteTo(&pkBuf); err != nil {
		t.Fatal(err)
	}
	if !prover.InitAlgorithm(id, pkBuf.Bytes(), r1csBytes) {
		t.Fatal("InitAlgorithm (GPU) refused a gnark-made proving key")
	}
	for round := 0; round < 8; round++ {
		key, nonce, pt := make([]byte, keyLen), make([]byte, 12), make([]byte, 64)
		rand.Read(key)
		rand.Read(nonce)
		rand.Read(pt)
		counter := uint32(round * 7919)
		in, _ := json.Marshal(&prover.InputParams{Cipher: cipher, Key: key, Nonce: nonce, Counter: counter, Input: pt})
		var out *prover.OutputParams
		if err = json.Unmarshal(prover.Prove(in), &out); err != nil {
			t.Fatal(err)
		}
		// public witness exactly as verifiers.go:120-152 builds it
		w := &verifier.AESWrapper{}
		for i := 0; i < 64; i++ {
			w.Plaintext[i] = pt[i]
			w.Ciphertext[i] = out.PublicSignals[i]
		}
		for i := 0; i < 12; i++ {
			w.Nonce[i] = nonce[i]
		}
		w.Counter = counter
		pub, err := frontend.NewWitness(w, ecc.BN254.ScalarField(), frontend.PublicOnly())
		if err != nil {
			t.Fatal(err)
		}
		proof := groth16.NewProof(ecc.BN254)
		if _, err = proof.ReadFrom(bytes.NewReader(out.Proof.ProofJson)); err != nil {
			t.Fatalf("gnark cannot decode the GPU proof: %v", err)
		}
		if err = groth16.Verify(proof, vk, pub); err != nil {
			t.Fatalf("gnark rejects the GPU proof (round %d): %v", round, err)
		}
		// and a proof for another statement is rejected
		w.Plaintext[0] = pt[0] + 1
		bad, _ := frontend.NewWitness(w, ecc.BN254.ScalarField(), frontend.PublicOnly())
		if groth16.Verify(proof, vk, bad) == nil {
			t.Fatal("gnark accepted a GPU proof for a different statement")
		}
	}
}

func TestGPUAcceptAES128(t *testing.T) {
	acceptAES(t, prover.AES_128, "aes-128-ctr", 16, "../circuits/generated/r1cs.aes128")
}

func TestGPUAcceptAES256(t *testing.T) {
	acceptAES(t, prover.AES_256, "aes-256-ctr", 32, "../circuits/generated/r1cs.aes256")
}

// Control: the shipped ChaCha20-V3 key, verified by the reference's own embedded vk through verifier.Verify (core_test.go:130-172).
func TestGPUAcceptChaCha20(t *testing.T) {
	pk, _ := os.ReadFile("../circuits/generated/pk.chacha20")
	cs, _ := os.ReadFile("../circuits/generated/r1cs.chacha20")
	if !prover.InitAlgorithm(prover.CHACHA20, pk, cs) {
		t.Fatal("InitAlgorithm failed")
	}
	key, nonce, pt := make([]byte, 32), make([]byte, 12), make([]byte, 64)
	rand.Read(key)
	rand.Read(nonce)
	rand.Read(pt)
	in, _ := json.Marshal(&prover.InputParams{Cipher: "chacha20", Key: key, Nonce: nonce, Counter: 1, Input: pt})
	var out *prover.OutputParams
	json.Unmarshal(prover.Prove(in), &out)
	signals := append([]byte{}, out.PublicSignals...)
	signals = append(signals, nonce...)
	ctr := make([]byte, 4)
	binary.LittleEndian.PutUint32(ctr, 1)
	signals = append(signals, ctr...)
	signals = append(signals, pt...)
	vin, _ := json.Marshal(&verifier.InputVerifyParams{Cipher: "chacha20", Proof: out.Proof.ProofJson, PublicSignals: signals})
	if !verifier.Verify(vin) {
		t.Fatal("the reference verifier rejects the GPU proof")
	}
}
