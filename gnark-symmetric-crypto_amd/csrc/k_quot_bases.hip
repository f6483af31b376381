// InitAlgorithm-time: the quotient bases pk.G1.Z of the proving key, once more in EVALUATION form.
//
// groth16.Prove ends the quotient with sum_k H_k Z_k over the COEFFICIENTS of H (reference libraries/prover/impl/provers.go:148,216;
// gnark backend/groth16/bn254 computeH + the Z multi-exponentiation — SURVEY.md §8(a) a7, a9).  With the identities of k_ntt.hip
//   H = (S - D) / 2,   S = the interpolation of c on the n-th roots of unity,   D = the interpolation of d_i = A(zeta w^i) B(zeta w^i)
//   on the coset zeta * (roots of unity), zeta^n = -1,
// the sum is linear in the VALUES c_i and d_i:
//   sum_k H_k Z_k = sum_i c_i U_i + sum_i d_i V_i,     U_i =  (1 / 2n) sum_k w^(-ik) Z_k,     V_i = -(1 / 2n) sum_k zeta^(-k) w^(-ik) Z_k,
// so the prover needs neither the inverse transform of c nor the one of d (two of its six transforms): c is what the solver wrote
// (three quarters zero, nearly all of the rest +-1: a flat MSM set like the wire sets), d feeds the windowed kernel with the bases
// V_i instead of Z_k.  Same group element, hence the same proof bytes.  U and V are discrete Fourier transforms of the key's points —
// "in the exponent": n log n / 2 butterflies whose twiddle products are scalar multiplications — computed here once per key.
// The kernels' d comes out in the NTT kernels' 2^261 Montgomery domain (d_i * 2^261 mod r as a canonical integer): that constant
// is folded into V as well.
#include "kernels.hpp"
#include "bn254_fp29.hpp"

namespace gsc {
using namespace bn254;

namespace {

__device__ __forceinline__ fe9 qb_to_fp29(const fe& old_mont) { return Fp29::to_mont(Fp29::unpack(Fp::from_mont(old_mont))); }
__device__ __forceinline__ fe qb_from_fp29(const fe9& m) { return Fp::to_mont(Fp29::pack(Fp29::from_mont(m))); }

__device__ __forceinline__ fe qb_fr_pow(const fe& a, uint32_t e) {
    fe acc = Fr::one(); bool started = false;
    for (int i = 31; i >= 0; i--) {
        if (started) acc = Fr::sqr(acc);
        if ((e >> i) & 1) { acc = started ? Fr::mul(acc, a) : a; started = true; }
    }
    return acc;
}

// the primitive 2n-th root of unity the quotient kernels use (k_init.hip k_ntt_constants): gnark-crypto's 2^28-th root (SURVEY.md
// App. I; canonical, 8 little-endian words) squared 27 - L times
__device__ __forceinline__ fe qb_zeta(int L) {
    fe c; const uint32_t w[8] = {0x725b19f0u, 0x9bd61b6eu, 0x41112ed4u, 0x402d111eu, 0x8ef62abcu, 0x00e0a7ebu, 0xa58a7e85u, 0x2a3c09f0u};
    for (int i = 0; i < 8; i++) c.l[i] = w[i];
    fe z = Fr::to_mont(c);
    for (int t = 0; t < 27 - L; t++) z = Fr::sqr(z);
    return z;
}

// k * P for a canonical scalar k (8 little-endian words); exact group law throughout (init-time: clarity over speed)
__device__ __noinline__ Xyzz9<Fp29f> qb_scalar_mul(const Xyzz9<Fp29f>& P, const fe& k) {
    Xyzz9<Fp29f> acc = G1x::infinity();
    int top = 255;
    while (top >= 0 && !((k.l[top >> 5] >> (top & 31)) & 1u)) top--;
    for (int i = top; i >= 0; i--) {
        acc = G1x::dbl(acc);
        if ((k.l[i >> 5] >> (i & 31)) & 1u) acc = G1x::add(acc, P);
    }
    return acc;
}

// tw[e] = w^-e as a canonical integer, e < n/2
__global__ __launch_bounds__(64) void k_qb_twiddles(const fe* omega_inv, uint32_t half_n, fe* tw) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= half_n) return;
    tw[e] = Fr::from_mont(qb_fr_pow(*omega_inv, e));
}

// Y[pos] = lambda_k * Zfile[pos], k = bitrev(pos): the key stores Z in bit-reversed order (Zfile[pos] = Z_bitrev(pos), n - 1 points;
// position n - 1 is the missing top coefficient: the point at infinity).  mode 0 (U): lambda = 1 / 2n;
// mode 1 (V): lambda = -zeta^-k / (2n * 2^261).  zeta: the primitive 2n-th root with zeta^2 = w (k_init.hip derives the same one).
__global__ __launch_bounds__(64) void k_qb_load(const Aff<Fp>* zfile, const uint8_t* status, uint32_t n, int L, int mode,
                                                 const fe* omega_inv, const fe* n_inv, fe* Y) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    fe* dst = Y + 4 * (size_t)pos;
    if (pos == n - 1 || status[pos] == 2) { G1x::store_xyzz(dst, G1x::infinity()); return; }
    const uint32_t k = __brev(pos) >> (32 - L);
    fe lam = Fr::mul(*n_inv, Fr::inv(Fr::from_u32(2)));
    if (mode == 1) {
        const fe zeta_inv = Fr::mul(qb_zeta(L), *omega_inv);                      // zeta^-1 = zeta * w^-1
        const fe r261_inv = Fr::inv(qb_fr_pow(Fr::from_u32(2), 261));
        lam = Fr::neg(Fr::mul(Fr::mul(lam, r261_inv), qb_fr_pow(zeta_inv, k)));
    }
    const Aff9<Fp29f> P{qb_to_fp29(zfile[pos].x), qb_to_fp29(zfile[pos].y)};
    G1x::store_xyzz(dst, qb_scalar_mul(G1x::from_aff(P), Fr::from_mont(lam)));
}

// One decimation-in-time stage s (bit-reversed input, natural output after stage L - 1): pairs (j, j + 2^s) inside blocks of
// 2^(s+1), twiddle w^-((j mod 2^s) << (L-1-s)).
__global__ __launch_bounds__(64) void k_qb_stage(fe* Y, uint32_t half_n, int L, int s, const fe* tw) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= half_n) return;
    const uint32_t half = 1u << s, lo = t & (half - 1), j = ((t >> s) << (s + 1)) + lo, e = lo << (L - 1 - s);
    fe* pu = Y + 4 * (size_t)j; fe* pv = Y + 4 * (size_t)(j + half);
    const Xyzz9<Fp29f> u = G1x::load_xyzz(pu);
    Xyzz9<Fp29f> v = G1x::load_xyzz(pv);
    if (e) v = qb_scalar_mul(v, tw[e]);
    G1x::store_xyzz(pu, G1x::add(u, v));
    v.y = Fp29::norm(Fp29::neg(v.y));
    G1x::store_xyzz(pv, G1x::add(u, v));
}

// XYZZ -> affine bases in the layout the table builders take (8 x 32-bit Montgomery images); status 2 = the point at infinity
// perm (optional): out[i] = point perm[i] — the order the windowed MSM wants its bases in (kernels.hpp quot_digit_index)
__global__ __launch_bounds__(64) void k_qb_finish(const fe* Y, uint32_t n, const uint32_t* perm, Aff<Fp>* out, uint8_t* status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Xyzz9<Fp29f> p = G1x::load_xyzz(Y + 4 * (size_t)(perm ? perm[i] : i));
    if (p.inf) { status[i] = 2; out[i] = Aff<Fp>{Fp::zero(), Fp::zero()}; return; }
    const Aff9<Fp29f> a = G1x::to_aff(p);
    out[i] = Aff<Fp>{qb_from_fp29(a.x), qb_from_fp29(a.y)};
    status[i] = 0;
}

}  // namespace

void launch_quot_bases(const G1Aff* zfile, const uint8_t* zstatus, int L, int mode, const fe* omega_inv, const fe* n_inv,
                       fe* tw, G1Xyzz* scratch, const uint32_t* perm, G1Aff* out, uint8_t* status, hipStream_t s) {
    const uint32_t n = 1u << L, hn = n / 2;
    fe* Y = reinterpret_cast<fe*>(scratch);
    hipLaunchKernelGGL(k_qb_twiddles, dim3((hn + 63) / 64), dim3(64), 0, s, omega_inv, hn, tw);
    hipLaunchKernelGGL(k_qb_load, dim3((n + 63) / 64), dim3(64), 0, s, reinterpret_cast<const Aff<Fp>*>(zfile), zstatus, n, L, mode, omega_inv, n_inv, Y);
    for (int st = 0; st < L; st++) hipLaunchKernelGGL(k_qb_stage, dim3((hn + 63) / 64), dim3(64), 0, s, Y, hn, L, st, tw);
    hipLaunchKernelGGL(k_qb_finish, dim3((n + 63) / 64), dim3(64), 0, s, Y, n, perm, reinterpret_cast<Aff<Fp>*>(out), status);
}

}  // namespace gsc
