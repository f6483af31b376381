// caller_napi.js — the node.js caller of integration/caller.js, through the hand-written N-API binding gsc_napi.node instead of
// koffi (which cannot be installed offline).  Same calls, same order as a FFI host of the reference's libprove.so.
//   node integration/caller_napi.js <gsc_napi.node> <libprove.so> [<pk.chacha20> <r1cs.chacha20>]
const fs = require('fs');
const path = require('path');
const [addonPath, libPath, pkPath, r1csPath] = process.argv.slice(2);
const gsc = require(path.resolve(addonPath));
gsc.load(path.resolve(libPath));
const ask = (obj) => JSON.parse(gsc.prove(Buffer.from(typeof obj === 'string' ? obj : JSON.stringify(obj))).toString());

// error values are the JSON-encoded Go panic values (libprove.go:33-43, core_test.go:120-128)
console.log('unknown cipher   ->', JSON.stringify(ask({ cipher: 'nope' })));
console.log('not initialised  ->', JSON.stringify(ask({ cipher: 'chacha20' })));
console.log('syntax error     ->', JSON.stringify(ask('{')));
console.log('bad key element  ->', JSON.stringify(ask({ cipher: 'chacha20', key: [1, 2, 256], nonce: [], counter: 1, input: [] })));
console.log('garbage key file ->', gsc.initAlgorithm(0, Buffer.from('garbage'), Buffer.from('garbage')));
if (!pkPath) { console.log('NODE-ERRORS-DONE'); process.exit(0); }

if (!gsc.initAlgorithm(0, fs.readFileSync(pkPath), fs.readFileSync(r1csPath))) throw new Error('InitAlgorithm failed (no GPU?)');
const statement = (i) => ({ cipher: 'chacha20', key: Array(32).fill(2), nonce: Array(12).fill(3), counter: i, input: Buffer.alloc(64, i).toString('base64') });
const one = ask(statement(1));
if (!one.proof) throw new Error('Prove failed: ' + JSON.stringify(one));
console.log('proof bytes:', Buffer.from(one.proof.proofJson, 'base64').length, 'publicSignals:', one.publicSignals);
const batch = JSON.parse(gsc.proveBatch(Buffer.from(JSON.stringify([...Array(100).keys()].map(statement)))).toString());
console.log('batch of', batch.length, 'all proved:', batch.every((o) => o.proof !== undefined));
console.log('NODE-PROOF-DONE');
