// InitAlgorithm-time kernels: point decompression, fixed-base digit tables, NTT constants.
//
// Replaces, on the GPU, the work hidden in groth16.ProvingKey.ReadFrom
// (reference libraries/prover/impl/prove_impl.go:86-87): ~10^5 Fp square roots to
// decompress the key, plus what gnark precomputes in its fft.Domain.  The digit tables are new: they trade
// the 288 GB of HBM for the bucket pass of Pippenger.
#include "kernels.hpp"
#include "bn254_fp29.hpp"

namespace gsc {
using namespace bn254;

namespace {

__device__ __forceinline__ fe fe_from_be_bytes(const uint8_t* b, bool clear_flags) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t* q = b + 28 - 4 * i;
        uint32_t b0 = q[0];
        if (clear_flags && i == 7) b0 &= 0x3F;
        r.l[i] = (b0 << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    }
    return r;
}
__device__ __forceinline__ bool fe_lt_mod_p(const fe& a) {
    for (int i = 7; i >= 0; i--) { if (a.l[i] < FpParams::mod(i)) return true; if (a.l[i] > FpParams::mod(i)) return false; }
    return false;
}
// (p+1)/4 — p = 3 mod 4, so a^((p+1)/4) is a square root of a when one exists
__device__ __noinline__ fe fp_sqrt_candidate(const fe& a) {
    const uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    return Fp::pow(a, e);
}

__global__ void k_decompress_g1(const uint8_t* in, G1Aff* out, uint8_t* status, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + 32 * i;
    uint8_t flag = b[0] & 0xC0;
    G1Aff r; uint8_t st = 0;
    if (flag == 0x40) { r.x = Fp::from_u32(1); r.y = Fp::from_u32(2); st = 2; }
    else {
        fe xc = fe_from_be_bytes(b, true);
        if (!fe_lt_mod_p(xc) || flag == 0) st = 1;
        fe x = Fp::to_mont(xc);
        fe rhs = Fp::add(Fp::mul(Fp::sqr(x), x), Fp::from_u32(3));
        fe y = fp_sqrt_candidate(rhs);
        if (!Fp::eq(Fp::sqr(y), rhs)) st = 1;
        bool large = Fp::lex_large(y);
        if ((flag == 0xC0) != large) y = Fp::neg(y);
        r.x = x; r.y = y;
    }
    out[i] = r; status[i] = st;
}

// Fp2 square root by the norm method: a = (x0 + x1 u)^2 => x0^2 = (a0 +- sqrt(a0^2 + a1^2)) / 2, x1 = a1 / (2 x0)
__device__ __noinline__ bool fp2_sqrt(const fe2& a, fe2& out) {
    fe2 x;
    if (Fp::is_zero(a.a1)) {
        fe s = fp_sqrt_candidate(a.a0);
        if (Fp::eq(Fp::sqr(s), a.a0)) { x.a0 = s; x.a1 = Fp::zero(); }
        else { fe na = Fp::neg(a.a0); s = fp_sqrt_candidate(na); if (!Fp::eq(Fp::sqr(s), na)) return false; x.a0 = Fp::zero(); x.a1 = s; }
    } else {
        fe nrm = Fp::add(Fp::sqr(a.a0), Fp::sqr(a.a1));
        fe s = fp_sqrt_candidate(nrm);
        if (!Fp::eq(Fp::sqr(s), nrm)) return false;
        fe half = Fp::inv(Fp::from_u32(2));
        fe t = Fp::mul(Fp::add(a.a0, s), half);
        fe x0 = fp_sqrt_candidate(t);
        if (!Fp::eq(Fp::sqr(x0), t)) {
            t = Fp::mul(Fp::sub(a.a0, s), half);
            x0 = fp_sqrt_candidate(t);
            if (!Fp::eq(Fp::sqr(x0), t)) return false;
        }
        x.a0 = x0; x.a1 = Fp::mul(a.a1, Fp::inv(Fp::dbl(x0)));
    }
    if (!Fp2::eq(Fp2::sqr(x), a)) return false;
    out = x; return true;
}

__global__ void k_decompress_g2(const uint8_t* in, G2Aff* out, uint8_t* status, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + 64 * i;
    uint8_t flag = b[0] & 0xC0;
    G2Aff r; uint8_t st = 0;
    // twist coefficient b' = 3/(9+u) = 3*(9-u)/82
    fe inv82 = Fp::inv(Fp::from_u32(82));
    fe2 bt; bt.a0 = Fp::mul(Fp::from_u32(27), inv82); bt.a1 = Fp::neg(Fp::mul(Fp::from_u32(3), inv82));
    if (flag == 0x40) {
        // any finite point will do for a base that is never selected: use x = 0 is not on the twist in general, so
        // derive one deterministically: try x = (k, 0) for k = 1, 2, ... until x^3 + b' is a square.
        st = 2;
        for (uint32_t k = 1; k < 64; k++) {
            fe2 x; x.a0 = Fp::from_u32(k); x.a1 = Fp::zero();
            fe2 rhs = Fp2::add(Fp2::mul(Fp2::sqr(x), x), bt), y;
            if (fp2_sqrt(rhs, y)) { r.x = x; r.y = y; break; }
        }
    } else {
        fe x1c = fe_from_be_bytes(b, true), x0c = fe_from_be_bytes(b + 32, false);
        if (!fe_lt_mod_p(x1c) || !fe_lt_mod_p(x0c) || flag == 0) st = 1;
        fe2 x; x.a0 = Fp::to_mont(x0c); x.a1 = Fp::to_mont(x1c);
        fe2 rhs = Fp2::add(Fp2::mul(Fp2::sqr(x), x), bt), y;
        if (!fp2_sqrt(rhs, y)) { st = 1; y = Fp2::zero(); }
        bool large = Fp2::lex_large(y);
        if ((flag == 0xC0) != large) y = Fp2::neg(y);
        r.x = x; r.y = y;
    }
    out[i] = r; status[i] = st;
}

__global__ void k_fr_from_be(const uint8_t* in, fe* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = Fr::to_mont(fe_from_be_bytes(in + 32 * i, false));
}
__global__ void k_fr_inverse(const fe* in, fe* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = Fr::inv(in[i]);
}

// Montgomery value of the 8 x 32-bit domain (R = 2^256, what the decompression kernels produce) -> radix-2^29 domain (R' = 2^261)
__device__ __forceinline__ fe9 to_fp29(const fe& old_mont) { return Fp29::to_mont(Fp29::unpack(Fp::from_mont(old_mont))); }
__device__ __forceinline__ Aff9<Fp29f> base_to_fp29(const Aff<Fp>* b) { return Aff9<Fp29f>{to_fp29(b->x), to_fp29(b->y)}; }
__device__ __forceinline__ Aff9<Fp2x> base_to_fp29(const Aff<Fp2>* b) {
    return Aff9<Fp2x>{fe9x2{to_fp29(b->x.a0), to_fp29(b->x.a1)}, fe9x2{to_fp29(b->y.a0), to_fp29(b->y.a1)}};
}

// Signed subset-sum tables for groups of eight bases whose scalars are -1, 0 or 1 in (almost) every proof: entry v-1 of group g is
// sum_i t_i * P_{8g+i} for the balanced-ternary value v = sum_i t_i 3^i, v = 1 .. (3^8-1)/2, affine (negative v: the negated entry).
// One thread per group.  With k the position of the leading digit (which is +1 for v > 0), v = 3^k + s with |s| <= (3^k-1)/2, so
// T[v] = P_k + sign(s) T[|s|] only needs entries built before; sums are kept in XYZZ in `scratch` (MSM_GROUP_ENTRIES points per group)
// and converted with one batch inversion like k_build_table.  A group where some signed subset sums to the point at infinity
// (equal or opposite bases) cannot be tabulated in affine form: ok[g] = 0 and the MSM kernel treats its bases one by one.
template <class F, class OldF>
__global__ void k_build_subset(const Aff<OldF>* bases, size_t ngroups, fe* table, fe* scratch, uint8_t* ok) {
    using C = Curve9<F>;
    using E = typename F::E;
    constexpr int CW = F::WORDS;
    constexpr uint32_t NT = MSM_GROUP_ENTRIES;
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    fe* out = table + g * NT * (2 * CW);
    fe* sc = scratch + g * NT * (4 * CW);
    bool good = true;
    E prefix = F::one();
    uint32_t k = 0, pw = 1;                       // leading digit position and 3^k
    for (uint32_t v = 1; v <= NT; v++) {
        while (v > (3 * pw - 1) / 2) { k++; pw *= 3; }
        const int32_t s = (int32_t)v - (int32_t)pw;
        const Aff9<F> P = base_to_fp29(bases + 8 * g + k);
        Xyzz9<F> Ev;
        if (s == 0) Ev = C::from_aff(P);
        else {
            Xyzz9<F> T = C::load_xyzz(sc + ((uint32_t)(s < 0 ? -s : s) - 1) * (4 * CW));
            if (s < 0) T.y = F::neg(T.y);
            Ev = C::template madd<true>(T, P);
        }
        if (Ev.inf || F::is_zero(Ev.zzz)) { good = false; Ev = C::from_aff(P); }      // keep the arithmetic defined; the group is disabled
        C::store_xyzz(sc + (v - 1) * (4 * CW), Ev);
        F::store(out + (v - 1) * (2 * CW), prefix);
        prefix = F::mul(prefix, Ev.zzz);
    }
    ok[g] = good ? 1 : 0;
    E inv = F::inv(prefix);
    for (uint32_t v = NT; v >= 1; v--) {
        const Xyzz9<F> Ev = C::load_xyzz(sc + (v - 1) * (4 * CW));
        const E pre = F::load(out + (v - 1) * (2 * CW));
        const E izzz = F::mul(inv, pre);
        inv = F::mul(inv, Ev.zzz);
        const E izz = F::mul(F::sqr(Ev.zz), F::sqr(izzz));
        Aff9<F> a; a.x = F::mul(Ev.x, izz); a.y = F::mul(Ev.y, izzz);
        C::store_aff(out + (v - 1) * (2 * CW), a);
    }
}

__device__ __forceinline__ fe fr_pow_u32(const fe& a, uint32_t e) {
    fe acc = Fr::one(); bool started = false;
    for (int i = 31; i >= 0; i--) {
        if (started) acc = Fr::sqr(acc);
        if ((e >> i) & 1) { acc = started ? Fr::mul(acc, a) : a; started = true; }
    }
    return acc;
}
// Values are produced with the 8 x 32-bit scalar field and stored as memory images of the radix-2^29 field used by the NTT
// kernels (bn254_fp29.hpp): twiddles and den_inv in its Montgomery domain (2^261); scale_mid additionally carries the factor
// 2^261/2^256 that moves the solver's a/b/c values (2^256 domain) into that domain; scale_out is a plain integer so that the
// last product of the pipeline leaves Montgomery form.
// gnark-crypto's 2^28-th primitive root of unity of Fr, canonical, 8 little-endian words
__device__ __forceinline__ fe fr_root_2_28() {
    fe c; const uint32_t w[8] = {0x725b19f0u, 0x9bd61b6eu, 0x41112ed4u, 0x402d111eu, 0x8ef62abcu, 0x00e0a7ebu, 0xa58a7e85u, 0x2a3c09f0u};
    for (int i = 0; i < 8; i++) c.l[i] = w[i];
    return Fr::to_mont(c);
}
__global__ void k_ntt_constants(const fe* omega, const fe* omega_inv, const fe* n_inv, int L,
                                int32_t* tw_fwd, int32_t* tw_inv, fe* scale_mid, fe* scale_out, fe* half_c, int32_t* qr, uint32_t* flag, int32_t* tw_inv_plain, fe* scale_mid_plain) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = 1u << L;
    if (i <= 2 * NTT_QMAX) {      // q*r, q = i - NTT_QMAX, as tight limbs with a signed top limb: the NTT kernels' range reduction subtracts these
        const int64_t q = (int64_t)i - NTT_QMAX; int64_t acc = 0;
        for (int k = 0; k < 12; k++) {
            if (k < 9) acc += q * Fr29Q::PK(0, k);
            qr[12 * (size_t)i + k] = k < 8 ? (int32_t)(acc & ((1 << 29) - 1)) : k == 8 ? (int32_t)acc : 0;
            if (k < 8) acc >>= 29;
        }
    }
    if (i >= n) return;
    if (i < n / 2) {      // twiddles are stored as limbs (12 int32 per entry), ready for the butterflies
        const fe9 f = Fr29::freeze(Fr29::to_mont(Fr29::unpack(Fr::from_mont(fr_pow_u32(*omega, i)))));
        const fe9 b = Fr29::freeze(Fr29::to_mont(Fr29::unpack(Fr::from_mont(fr_pow_u32(*omega_inv, i)))));
        for (int k = 0; k < 12; k++) { tw_fwd[12 * (size_t)i + k] = k < 9 ? f.l[k] : 0; tw_inv[12 * (size_t)i + k] = k < 9 ? b.l[k] : 0; }
        const fe9 bp = Fr29::unpack(Fr::from_mont(fr_pow_u32(*omega_inv, i)));      // the same power as a canonical integer (tight limbs)
        for (int k = 0; k < 12; k++) tw_inv_plain[12 * (size_t)i + k] = k < 9 ? bp.l[k] : 0;
    }
    fe zeta = fr_root_2_28();                       // -> the primitive 2n-th root: 27 - L squarings
    for (int t = 0; t < 27 - L; t++) zeta = Fr::sqr(zeta);
    const fe zeta_inv = Fr::mul(zeta, *omega_inv);                              // zeta^-1 = zeta * omega^-1 (zeta^2 = omega)
    uint32_t br = __brev(i) >> (32 - L);
    {
        fe9 c; for (int k = 0; k < 9; k++) c.l[k] = Fr29Q::FROM_R256(k);
        const fe9 sm = Fr29::to_mont(Fr29::unpack(Fr::from_mont(Fr::mul(*n_inv, fr_pow_u32(zeta, br)))));
        const fe9 smc = Fr29::freeze(Fr29::mul(sm, c));
        scale_mid[i] = Fr29::pack(smc);
        fe one256; for (int k = 0; k < 8; k++) one256.l[k] = FrParams::one(k);      // 2^256 mod r
        scale_mid_plain[i] = Fr29::pack(Fr29::freeze(Fr29::mul(smc, Fr29::to_mont(Fr29::unpack(one256)))));      // scale_mid * 2^256
    }
    const fe half = Fr::inv(Fr::from_u32(2));
    scale_out[i] = Fr::from_mont(Fr::mul(Fr::mul(*n_inv, half), fr_pow_u32(zeta_inv, br)));
    if (i == 0) {
        *half_c = Fr::from_mont(Fr::mul(*n_inv, Fr::from_u32(16)));
        if (!Fr::eq(Fr::sqr(zeta), *omega)) atomicOr(flag, 1u);
    }
}

// TEST HOOK kernel: element-wise field operations on canonical inputs, through the radix-2^29 implementation, canonical outputs.
// op: 0 mul, 1 add, 2 sub, 3 sqr, 4 inv, 5 fmms(a,b,b,a+... see host doc), 6 neg, 8 (Fr only) wave_batch_inverse of k_solver.hip; lazy: apply the op `chain` times on a running value
template <class F>
__global__ void k_field_ops(int op, const fe* a, const fe* b, fe* out, size_t n, int chain) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe9 x = F::to_mont(F::unpack(a[i])), y = F::to_mont(F::unpack(b[i]));
    fe9 r = x;
    for (int c = 0; c < chain; c++) {
        switch (op) {
            case 0: r = F::mul(r, y); break;
            case 1: r = F::norm(F::add(r, y)); break;              // lazy sums: only carries are propagated between steps
            case 2: r = F::norm(F::sub(r, y)); break;
            case 3: r = F::sqr(r); break;
            case 4: r = F::inv(r); break;
            case 5: r = F::fmms(r, y, y, x); break;                // r*y - y*x
            case 6: r = F::norm(F::neg(r)); break;
            case 7: r = F::mul(F::sub(r, y), F::add(x, y)); break; // signed-tight x loose operands
        }
        if ((op == 1 || op == 2) && (c & 3) == 3) r = F::freeze(r);   // keep lazy chains inside freeze()'s documented domain
    }
    out[i] = F::pack(F::from_mont(r));
}

}  // namespace

void launch_field_ops(int field, int op, const fe* a, const fe* b, fe* out, size_t n, int chain, hipStream_t s) {
    if (!n) return;
    if (field == 1 && op == 8) { launch_wave_inverse(a, out, n, s); return; }      // the generic solver's own inversion (8 x 32-bit limbs; k_solver.hip)
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    if (field == 0) hipLaunchKernelGGL(k_field_ops<Fp29>, grid, block, 0, s, op, a, b, out, n, chain);
    else hipLaunchKernelGGL(k_field_ops<Fr29>, grid, block, 0, s, op, a, b, out, n, chain);
}

namespace {
}  // namespace

// Diagnostics: ONE wave that samples {100 MHz clock, shader clock} every `interval` ticks of the 100 MHz clock, `n` times.  Launched on a
// stream of its own before a call, it stays resident beside the call's kernels and shows the shader clock each of them is granted.
namespace {
__global__ __launch_bounds__(64) void k_clock_trace(unsigned long long* out, uint32_t n, uint32_t interval) {
    if (threadIdx.x) return;
    unsigned long long next = wall_clock64();
    for (uint32_t i = 0; i < n; i++) {
        unsigned long long t;
        while ((t = wall_clock64()) < next) __builtin_amdgcn_s_sleep(32);
        out[2 * i] = t; out[2 * i + 1] = clock64();
        next = t + interval;
    }
}
}  // namespace
void launch_clock_trace(unsigned long long* out, uint32_t n, uint32_t interval_100mhz_ticks, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_clock_trace, dim3(1), dim3(64), 0, s, out, n, interval_100mhz_ticks);
}

static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

void launch_decompress_g1(const uint8_t* in, G1Aff* out, uint8_t* status, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_decompress_g1, dim3(blocks_for(n, 64)), dim3(64), 0, s, in, out, status, n);
}
void launch_decompress_g2(const uint8_t* in, G2Aff* out, uint8_t* status, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_decompress_g2, dim3(blocks_for(n, 64)), dim3(64), 0, s, in, out, status, n);
}
void launch_fr_from_be(const uint8_t* in, fe* out, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_fr_from_be, dim3(blocks_for(n, 64)), dim3(64), 0, s, in, out, n);
}
void launch_fr_inverse(const fe* in, fe* out, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_fr_inverse, dim3(blocks_for(n, 64)), dim3(64), 0, s, in, out, n);
}
void launch_build_subset_g1(const G1Aff* bases, size_t ngroups, G1Aff* table, G1Xyzz* scratch, uint8_t* ok, hipStream_t s) {
    if (ngroups) hipLaunchKernelGGL((k_build_subset<Fp29f, Fp>), dim3(blocks_for(ngroups, 64)), dim3(64), 0, s,
                                    reinterpret_cast<const Aff<Fp>*>(bases), ngroups, reinterpret_cast<fe*>(table), reinterpret_cast<fe*>(scratch), ok);
}
void launch_build_subset_g2(const G2Aff* bases, size_t ngroups, G2Aff* table, G2Xyzz* scratch, uint8_t* ok, hipStream_t s) {
    if (ngroups) hipLaunchKernelGGL((k_build_subset<Fp2x, Fp2>), dim3(blocks_for(ngroups, 64)), dim3(64), 0, s,
                                    reinterpret_cast<const Aff<Fp2>*>(bases), ngroups, reinterpret_cast<fe*>(table), reinterpret_cast<fe*>(scratch), ok);
}
void launch_ntt_constants(const fe* omega, const fe* omega_inv, const fe* n_inv, int L,
                          int32_t* tw_fwd, int32_t* tw_inv, fe* scale_mid, fe* scale_out, fe* half_c, int32_t* qr, uint32_t* flag, int32_t* tw_inv_plain, fe* scale_mid_plain, hipStream_t s) {
    size_t n = (size_t)1 << L;
    if (n < 2 * NTT_QMAX + 1) n = 2 * NTT_QMAX + 1;
    hipLaunchKernelGGL(k_ntt_constants, dim3(blocks_for(n, 64)), dim3(64), 0, s, omega, omega_inv, n_inv, L,
                       tw_fwd, tw_inv, scale_mid, scale_out, half_c, qr, flag, tw_inv_plain, scale_mid_plain);
}

}  // namespace gsc
