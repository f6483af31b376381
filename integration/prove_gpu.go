// prove_gpu.go — drop-in bodies for impl.InitAlgorithm / impl.Prove that forward to the MI355X library through cgo.
//
// Where it goes: libraries/prover/impl/prove_gpu.go of the reference checkout, replacing the bodies of
// prove_impl.go:65 (InitAlgorithm) and :116 (Prove): keep that file's declarations (the algorithm ids :15-25, InputParams /
// OutputParams / Proof :45-52 and provers.go:53-59) and put `//go:build !gsc_gpu` on the file(s) holding the two function
// bodies and the gnark-backed provers, so that exactly one implementation is compiled.  libraries/prover/libprove.go (the
// cgo exports FFI hosts bind, :17-47) and libraries/core_test.go keep compiling unchanged.  Build the GPU library first
// (`make -C gnark-symmetric-crypto_amd/csrc`), copy include/libprove.h and libprove.so into <reference>/gpu/, then
//
//	go build -tags gsc_gpu -buildmode=c-shared -o libprove.so libraries/prover/libprove.go      (reference README.md:83-96)
//
// The Go toolchain is not part of the image this repository is developed in, so this file is shipped as source and is not
// compiled by the test suite; integration/ffi_harness.c exercises the same C-ABI from C.
//
//go:build gsc_gpu

package impl

/*
#cgo LDFLAGS: -L${SRCDIR}/../../../gpu -lprove -Wl,-rpath,${SRCDIR}/../../../gpu
#include <stdlib.h>
#include "../../../gpu/libprove.h"
*/
import "C"

import (
	"encoding/json"
	"unsafe"
)

func goSlice(b []byte) C.GoSlice {
	if len(b) == 0 {
		return C.GoSlice{}
	}
	return C.GoSlice{data: unsafe.Pointer(&b[0]), len: C.GoInt(len(b)), cap: C.GoInt(cap(b))}
}

// InitAlgorithm keeps the signature of prove_impl.go:65: true on success or if already initialised.
func InitAlgorithm(algorithmID uint8, provingKey []byte, r1csData []byte) bool {
	return C.InitAlgorithm(C.GoUint8(algorithmID), goSlice(provingKey), goSlice(r1csData)) != 0
}

// Prove keeps the signature and the panic behaviour of prove_impl.go:116 (core_test.go:120-128 relies on the panic):
// the library returns the JSON encoding of the panic value, which is re-raised here so that libprove.go:33-43 recovers it.
func Prove(params []byte) []byte {
	ret := C.Prove(goSlice(params))
	if ret.r0 == nil {
		panic("prover returned no result")
	}
	defer C.Free(ret.r0)
	out := C.GoBytes(ret.r0, C.int(ret.r1))
	var probe struct {
		Proof *json.RawMessage `json:"proof"`
	}
	if json.Unmarshal(out, &probe) != nil || probe.Proof == nil {
		var v interface{}
		_ = json.Unmarshal(out, &v)
		panic(v)
	}
	return out
}

// ProveBatch (addition): many statements per call; element i of the result is what Prove returns for element i.
func ProveBatch(params []byte) []byte {
	ret := C.ProveBatch(goSlice(params))
	if ret.r0 == nil {
		return nil
	}
	defer C.Free(ret.r0)
	return C.GoBytes(ret.r0, C.int(ret.r1))
}

// Setup (addition): Groth16 keys for an R1CS file in gnark's WriteTo layouts, CSPRNG toxic waste (keygen.go:345,384,423).
func Setup(r1csData []byte) (pk []byte, vk []byte, ok bool) {
	var ppk, pvk unsafe.Pointer
	var npk, nvk C.size_t
	if C.gsc_setup(goSlice(r1csData), nil, &ppk, &npk, &pvk, &nvk) != 0 {
		return nil, nil, false
	}
	defer C.Free(ppk)
	defer C.Free(pvk)
	return C.GoBytes(ppk, C.int(npk)), C.GoBytes(pvk, C.int(nvk)), true
}
