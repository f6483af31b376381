// caller.js — a node.js FFI host of libprove.so, the way the reference's consumers bind the Go-built library
// (reference README.md:24-25 "node.js on Linux x64/arm64", :79-97): dlopen, four symbols, GoSlice by value.
// Works unchanged against the reference's own libprove.so and against gnark-symmetric-crypto_amd/libprove.so.
//
//   npm install koffi            (not available offline in this repository's image: the C twin of this file,
//   node caller.js <libprove.so> <pk.chacha20> <r1cs.chacha20>       integration/ffi_harness.c, is what the test suite runs)
const fs = require('fs');
const koffi = require('koffi');

const [libPath, pkPath, r1csPath] = process.argv.slice(2);
const lib = koffi.load(libPath);
const GoSlice = koffi.struct('GoSlice', { data: 'void *', len: 'longlong', cap: 'longlong' });
const ProveReturn = koffi.struct('Prove_return', { r0: 'void *', r1: 'longlong' });
const enforce_binding = lib.func('void enforce_binding()');
const InitAlgorithm = lib.func('uint8 InitAlgorithm(uint8 algorithmID, GoSlice provingKey, GoSlice r1cs)');
const Prove = lib.func('Prove_return Prove(GoSlice params)');
const Free = lib.func('void Free(void *pointer)');
let ProveBatch = null;                              // addition of the GPU library; absent from the reference's build
try { ProveBatch = lib.func('Prove_return ProveBatch(GoSlice params)'); } catch (e) { /* reference library */ }

const slice = (buf) => ({ data: buf, len: buf.length, cap: buf.length });
function take(ret) {                                // malloc'd, NOT NUL-terminated JSON; released with Free (libprove.go:25-28, :40, :46)
  const out = Buffer.from(koffi.decode(ret.r0, koffi.array('uint8', Number(ret.r1))));
  Free(ret.r0);
  return JSON.parse(out.toString());
}

enforce_binding();
if (!InitAlgorithm(0, slice(fs.readFileSync(pkPath)), slice(fs.readFileSync(r1csPath)))) throw new Error('InitAlgorithm failed');

const statement = (i) => ({
  cipher: 'chacha20',
  key: Array(32).fill(2), nonce: Array(12).fill(3), counter: i,
  input: Buffer.alloc(64, i).toString('base64'),   // []uint8 fields accept base64 strings or arrays of numbers (encoding/json)
});
const one = take(Prove(slice(Buffer.from(JSON.stringify(statement(1))))));
if (!one.proof) throw new Error('Prove failed: ' + JSON.stringify(one));   // failures are the JSON-encoded Go panic value
console.log('proof bytes:', Buffer.from(one.proof.proofJson, 'base64').length, 'ciphertext:', one.publicSignals);

// concurrent callers share device batches inside the library (micro-batching); a FFI host simply issues its calls
if (ProveBatch) {
  const batch = take(ProveBatch(slice(Buffer.from(JSON.stringify([...Array(256).keys()].map(statement))))));
  console.log('batch of', batch.length, 'all proved:', batch.every((o) => o.proof));
}
const err = take(Prove(slice(Buffer.from('{"cipher":"nope"}'))));
console.log('error value:', err);                  // "could not find prover fornope" (prove_impl.go:141)
