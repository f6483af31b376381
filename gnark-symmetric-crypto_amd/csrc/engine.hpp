// Per-algorithm GPU prover state and the batch pipeline.
//
// Host-side mirror of the reference's per-cipher prover objects (libraries/prover/impl/provers.go:61-77:
// Prover interface, baseProver{r1cs, pk}) — here "SetParams" uploads the decoded key and the solver program
// to HBM and builds the fixed-base tables; "Prove" runs the device pipeline for a batch of independent proofs.
#pragma once
#include <cstdint>
#include <cstring>
#include <atomic>
#include <cstddef>
#include <memory>
#include <string>
#include <vector>

namespace gsc {

enum Cipher : int { CHACHA20 = 0, AES_128 = 1, AES_256 = 2 };   // prove_impl.go:15-19

struct ProofRequest {
    uint8_t key[32]; uint32_t keylen;
    uint8_t nonce[12]; uint32_t counter;
    uint8_t plaintext[64];
    uint8_t ciphertext[64];          // filled by the caller with the native cipher
    uint8_t r[32], s[32], mask[32];  // prover randomness, canonical little-endian, < r
    // every copy of a request (the callers', the micro-batcher's, the staging vectors) clears its key and randomness when it goes away
    ~ProofRequest() { explicit_bzero(this, sizeof *this); }
};
struct ProofResult {
    int status = 0;                  // 0 ok; 1 unsatisfied constraint system; 2 degenerate point
    uint8_t proof[196]; size_t proof_len = 0;   // gnark proof.WriteTo bytes
};

struct EngineConfig {
    int device = 0;              // GSC_DEVICE: the device of a single-replica engine
    std::vector<int> devices;    // GSC_DEVICES=0,1,...: one replica (key tables, batch buffers, host thread) per listed device; every batch is
                                 // split over them in contiguous shares — a single FFI host process drives all the GPUs of a node
    size_t max_batch = 1024;     // GSC_MAX_BATCH: proofs per device batch = capacity of every lane (rounded to a multiple of 64)
    int lanes = 0;               // GSC_LANES: concurrent HIP streams, each with batch buffers of its own, handed to concurrent calls / chunks; 0 = 1 for ChaCha20-V3, 2 for AES-V2
    int small_lanes = -1;        // GSC_SMALL_LANES: extra lanes of `small_lane_cap` proofs for calls that fit them (several mid-size calls in flight at once); -1 = 2 for ChaCha20-V3, 0 for AES-V2
    int small_lane_cap = 0;      // GSC_SMALL_LANE_CAP: their capacity (multiple of 64, at most GSC_MAX_BATCH); 0 = the engine's choice (1024 beside a larger full lane, else 512)
    size_t min_split = 0;        // > 0: a call with at least 2*min_split proofs that is alone on the replica is cut over the lanes (GSC_MIN_SPLIT: test hooks only; off by default, see prove_on_replica)
    int window_z = 0;            // digit width of the Z (quotient) rows: 2^(c-1) multiples of each of the n-1 bases; 0 = largest <= 16 that fits z_table_gb
    int window_w = 0;            // digit width of the A / B1 / B2 / K / commitment sets; 0 = largest <= 16 that fits w_table_gb
    int z_table_gb = 48, w_table_gb = 16;   // per-algorithm HBM budgets used when the widths are not given (all three algorithms of the reference fit one 288 GB device)
    int bit_groups = 1;          // GSC_BIT_GROUPS: 0 no prediction-based layout; 1 bit groups / row lengths from a calibration witness; 2 every wire predicted a bit (test: exercises the fallbacks)
    int row_margin_bits = 1;     // GSC_ROW_MARGIN_BITS (test hooks only): a wire seen with k-bit values gets a row of 2^(k + margin) multiples (capped at 2^(c-1))
    int few_path = 1;            // GSC_FEW_PATH: calls with at most few_max statements use the latency kernels for the MSMs, the quotient and the assembly; 0 = always the batch kernels
    int few_max = 0;             // GSC_FEW_MAX: the largest call the latency kernels take (<= MSM_FEW_PROOFS = 32); 0 = 32 for ChaCha20 (4.6 ms for 1 statement, 7.8 ms for 16, 11.3 ms for 32; the batch kernels need 12.3 ms for anything up to 64), 20 for AES (8.2 ms for 1, +1.6 ms each: 38.4 ms for 20; batch kernels 43.7 ms)
    int few_solver = 1;          // GSC_FEW_SOLVER: such calls also solve the witness with the resident lanes-are-terms kernel (k_solver_few); 0 = one launch per level
    int few_workgroups = 0;      // GSC_FEW_WGS (test hooks only): its grid (workgroups of 8 waves, one per CU, so all are resident); 0 = 128 for 1-2 statements, 256 beyond
    int few_z_gb = 12;           // GSC_FEW_Z_GB: HBM budget of the latency-path layout of the quotient bases (rows per (base, window) of 8-, 6- or 4-bit digits: 8.6 GB ChaCha20 at 8, 11.5 GB AES at 6); 0 = none, such calls run the Horner pass
    int few_wide = 1;            // GSC_FEW_WIDE: the wide wires of the wire sets (AES: ~6 k per set) also get (base, window) rows for the latency path (~7.5 GB per AES algorithm); 0 = such calls run the windowed kernel + Horner for them
    int quotient_eval = 1;       // GSC_QUOTIENT_EVAL: batch calls take the quotient in evaluation form (k_quot_bases.hip: four transforms instead of six, the Z sum over the
                                 // bases V_i plus a flat sum over the solver's c); 0 = coefficient form for every call (six transforms, the key's own Z bases)
    int fuse_z_digits = 1;       // GSC_FUSE_Z_DIGITS: in evaluation form the last quotient kernel writes the signed digits of d itself (no scalar vector, no recoding pass); 0 = it writes d
    int small_witness = 1;       // GSC_SMALL_WITNESS: circuits whose whole witness is small integers (ChaCha20-V3) are solved by the integer kernels on byte planes
                                 // (wit_small.hpp) in every call beyond the latency path; 0 = always the generic field-arithmetic solver;
                                 // 2 (test hooks only) = every constraint row predicted narrow: the kernels notice, the chunk is solved again generically
    int ntt_plain = 1;           // GSC_NTT_PLAIN: with byte planes, the first transform kernel works on the small integers themselves (its first two stages become integer scalings
    int overlap_quotient = 1;    // GSC_OVERLAP_QUOTIENT: batch calls run the quotient kernels on the lane's third stream beside the wire-set MSMs (both only read the witness) when they hold fewer than 4096 statements; 0 = one after the other; 2 = whatever the size
    int stream_priorities = 1;   // GSC_STREAM_PRIORITIES: a lane's three streams at three priority levels = in three pools of hardware queues (alloc_lane); 0 = plain streams
                                 // of twiddles); 0 = on their Montgomery images, like the generic path.  Same bytes either way
    int small_witness_few = 1;   // GSC_SMALL_WITNESS_FEW: calls on the latency path take it too (one workgroup walks the levels: no device-wide barriers); 0 = they keep the resident lanes-are-terms solver
    int win_slice = 256;         // GSC_WIN_SLICE (test hooks): bases per slice of the windowed MSM kernel at full batches
    bool trace_host = false;     // GSC_TRACE_HOST: host-side timing lines on stderr (InitAlgorithm breakdown, per-chunk enqueue / wait / serialise)
    // diagnostics that change what the device does: honoured only when the test hooks were enabled at load time (test_hooks_enabled())
    bool solver_trace = false;   // GSC_SOLVER_TRACE: per-level clock stamps of the witness kernels
    int z_exp_entry_bits = 0;    // GSC_Z_EXP_ENTRY_BITS (test hooks; WRONG proofs, timing only): the Z kernel's gathers are confined to the first 2^bits entries of every row
    bool keep_secrets = false;   // GSC_KEEP_SECRETS: skip the end-of-call wipe of key wires / randomness in device memory (shows that the residue check sees them)
    bool few_test_abort = false; // GSC_FEW_TEST_ABORT: the resident witness kernel's barrier never fills (exercises the give-up path)
};
// Every knob is read HERE, once per InitAlgorithm / gsc_setup call — never on the proving path (getenv there would race with a
// host that calls setenv from another thread, and would let code inside the host flip diagnostics mid-flight).
EngineConfig config_from_env();
// GSC_ENABLE_TEST_HOOKS == "1" in the environment when libprove.so was LOADED (evaluated once, by a load-time initialiser).
bool test_hooks_enabled();

// The dominant kernel of a call, as the engine timed it with HIP events on the kernel's own stream (bench.py's roofline object).
// Batch calls: the Z-table gather-accumulate k_msm_win<Fp29f>.  Calls on the latency path (a handful of statements): the resident
// witness kernel k_solver_few, which is half of such a call.  statements = what the call proved; columns = the 64-padded batch the
// kernels ran on; nbases = fixed bases per proof of the Z set.
// clock_mhz: the shader clock during the Z kernel (stamps of one wave in the middle of the launch), 0 when not measured; nwin: its digit windows.
struct KernelStat { const char* name = ""; float ms = 0; size_t statements = 0, columns = 0, nbases = 0; float stage_ms[4] = {0, 0, 0, 0}; float clock_mhz = 0; int nwin = 0; };

// what: 0 W (Montgomery), 1 A, 2 B, 3 C (Montgomery; valid until computeH overwrites them: only with keep_abc), 4 h (canonical, bit-reversed order)
struct DebugVectors { std::vector<uint8_t> W, A, B, C, H; size_t n_wires = 0, n_constraints = 0, n = 0; };

class AlgorithmImpl;
class Algorithm {
  public:
    // throws std::runtime_error with a printable message on any parse / device failure
    Algorithm(Cipher cipher, const uint8_t* pk, size_t pk_len, const uint8_t* r1cs, size_t r1cs_len, const EngineConfig& cfg);
    ~Algorithm();
    Cipher cipher() const;
    // proves n independent statements; results[i] corresponds to reqs[i].  Thread-safe (serialised per algorithm).
    // whole: the call is a batch the micro-batcher took for one device: it goes to ONE replica when it fits (dispatch.hpp plan_shares)
    void prove_batch(const ProofRequest* reqs, size_t n, ProofResult* results, DebugVectors* debug_first = nullptr, bool whole = false);
    // forget the calling thread's "own last call" statistics (its next gsc_last_* query reports the chunk that finished last anywhere)
    void forget_thread_stat() const;
    size_t max_batch() const;      // over all devices
    size_t devices() const;
    size_t lanes() const;          // lanes of one replica: how many device batches can be in flight per device
    std::string describe() const;    // sizes, table memory — for logs / DESIGN numbers
    // Timing of the calling thread's own last prove_batch call (its chunk with the longest dominant kernel) — or, on a thread that has not
    // called prove_batch itself (callers whose statements rode the micro-batcher), of the chunk that finished last on any replica, any lane:
    // stage milliseconds (solve, ntt, msm, finalize) and the dominant kernel.
    KernelStat last_kernel_stat() const;
    // TEST HOOK: the quotient kernels alone on caller-supplied vectors.  abc_be: three matrices [m][64] of canonical big-endian
    // 32-byte values (a, then b, then c; 64 independent columns), m <= number of constraints.  h_out: [domain][64] 32-byte
    // little-endian canonical values, row k = coefficient bitrev(k).
    void debug_compute_h(const uint8_t* abc_be, size_t m, uint8_t* h_out);
    // TEST HOOK: the evaluation-form quotient kernels alone (launch_compute_d).  ab_be: a, then b, [m][64] canonical big-endian values.
    // d_out: [domain][64] little-endian canonical values, row i = A(zeta w^i) B(zeta w^i) * 2^261 mod r (natural order).
    void debug_compute_d(const uint8_t* ab_be, size_t m, uint8_t* d_out);
    size_t domain_size() const;
    // TEST HOOK: bytes of secrets (key wires, r, s, input records, masks) still non-zero in device memory, all replicas and lanes; 0 after any call
    size_t debug_secret_residue();
  private:
    std::vector<std::unique_ptr<AlgorithmImpl>> impls_;      // one per device
    std::unique_ptr<class ReplicaPicker> picker_;            // dispatch.hpp: least-loaded replica for calls that are not split
    mutable std::atomic<size_t> last_replica_{0};            // the replica whose chunk finished last (last_kernel_stat)
};

// TEST HOOK: samples the device's shader clock for n x interval_us microseconds beside whatever else runs (a resident one-wave kernel on a
// stream of its own); out: n pairs {100 MHz clock, shader clock}.  Blocks until the samples are in.
void debug_clock_trace(int device, uint32_t n, uint32_t interval_us, unsigned long long* out);
// TEST HOOK: runs element-wise operations of the device's radix-2^29 field implementation (see kernels.hpp launch_field_ops).
// a, b, out: n x 32 bytes little-endian canonical values (host memory).  Throws on HIP errors / missing GPU.
void debug_field_ops(int device, int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int chain);

}  // namespace gsc
