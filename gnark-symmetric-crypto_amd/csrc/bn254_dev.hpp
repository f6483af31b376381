// BN254 arithmetic for gfx950 (device only).
//
// Replaces, on the GPU, what the reference's hot path gets from gnark-crypto v0.14.0
// (reference go.mod:9): ecc/bn254/fp, fr Montgomery arithmetic and the G1/G2 group law used by
// groth16.Prove (libraries/prover/impl/provers.go:148,216).  Design notes:
//   * 8 x 32-bit limbs, little-endian, Montgomery form with R = 2^256 (same residues as gnark-crypto's
//     4 x 64-bit limbs, so values can be compared limb-for-limb with the oracle).
//   * v_mad_u64_u32 is the workhorse: measured ~4.8 cycles / wave-instruction / SIMD, i.e. the same
//     price as a carry-propagating 32-bit add (tools/ubench_intmul.hip, profiles/r01_ubench_intmul.txt).
//   * No MFMA: 254-bit modular products are not a dense contraction.
//   * Used by the solver (values are mostly bits, the work is additions), the key-decompression kernels and wherever a
//     value crosses an interface; the G1/G2 group law and the NTT run on the radix-2^29 field of bn254_fp29.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEVFN __device__ __forceinline__
#define DEVNOINL __device__ __noinline__
// Cold translation units (init-time kernels) define BN254_OUTLINE_MUL to keep compile time and code size down.
#ifdef BN254_OUTLINE_MUL
#define DEVMUL __device__ __noinline__
#else
#define DEVMUL __device__ __forceinline__
#endif

namespace bn254 {

struct FpParams {
    DEVFN static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    DEVFN static constexpr uint32_t one(int i) {   // R mod p
        constexpr uint32_t m[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return m[i];
    }
    DEVFN static constexpr uint32_t r2(int i) {    // R^2 mod p
        constexpr uint32_t m[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return m[i];
    }
    static constexpr uint32_t ninv = 0xe4866389u;  // -p^-1 mod 2^32
};
struct FrParams {
    DEVFN static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    DEVFN static constexpr uint32_t one(int i) {   // R mod r
        constexpr uint32_t m[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return m[i];
    }
    DEVFN static constexpr uint32_t r2(int i) {    // R^2 mod r
        constexpr uint32_t m[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return m[i];
    }
    static constexpr uint32_t ninv = 0xefffffffu;
};

struct alignas(16) fe {
    uint32_t l[8];
};

template <class P>
struct Field {
    using E = fe;

    DEVFN static E zero() { E r; for (int i = 0; i < 8; i++) r.l[i] = 0; return r; }
    DEVFN static E one() { E r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = P::one(i);
        return r; }
    DEVFN static E r2() { E r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = P::r2(i);
        return r; }
    DEVFN static bool is_zero(const E& a) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= a.l[i];
        return o == 0;
    }
    DEVFN static bool eq(const E& a, const E& b) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
        return o == 0;
    }
    // r = a - p if a >= p else a     (a < 2p)
    DEVFN static E reduce_once(const E& a) {
        E t; uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)a.l[i] - P::mod(i) - br;
            t.l[i] = (uint32_t)d; br = (d >> 32) & 1;
        }
        E r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = br ? a.l[i] : t.l[i];
        return r;
    }
    DEVFN static E add(const E& a, const E& b) {
        E t; uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (uint64_t)a.l[i] + b.l[i]; t.l[i] = (uint32_t)c; c >>= 32; }
        return reduce_once(t);   // p < 2^254: no carry out of limb 7
    }
    DEVFN static E dbl(const E& a) { return add(a, a); }
    DEVFN static E sub(const E& a, const E& b) {
        E t; uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)a.l[i] - b.l[i] - br;
            t.l[i] = (uint32_t)d; br = (d >> 32) & 1;
        }
        uint32_t mask = (uint32_t)0 - (uint32_t)br;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (uint64_t)t.l[i] + (P::mod(i) & mask); t.l[i] = (uint32_t)c; c >>= 32; }
        return t;
    }
    DEVFN static E neg(const E& a) {
        E t; uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)P::mod(i) - a.l[i] - br;
            t.l[i] = (uint32_t)d; br = (d >> 32) & 1;
        }
        bool z = is_zero(a);
#pragma unroll
        for (int i = 0; i < 8; i++) t.l[i] = z ? 0u : t.l[i];
        return t;
    }
    // acc (96 bits: 64-bit pair + overflow word) += x * y.  One v_mad_u64_u32 whose carry-out feeds a v_addc: the
    // compiler cannot express the carry-out of the 64-bit multiply-add from C, and without it every product costs
    // 4-5 instructions (measured: 585 VALU instructions per Montgomery product from the C version, 291 of them v_mov).
    DEVFN static void mac(uint64_t& acc, uint32_t& ovf, uint32_t x, uint32_t y) {
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(ovf) : "v"(x), "v"(y) : "vcc");
    }
    DEVFN static void mac_first(uint64_t& acc, uint32_t x, uint32_t y) {   // acc = x*y (no overflow possible)
        acc = (uint64_t)x * y;
    }
    DEVFN static void shift(uint64_t& acc, uint32_t& ovf) { acc = (acc >> 32) | ((uint64_t)ovf << 32); ovf = 0; }
    // Montgomery product a*b*R^-1 mod p: product scanning (FIPS), the reduction interleaved column by column.
    DEVMUL static E mul(const E& a, const E& b) {
        uint32_t n[8];
#pragma unroll
        for (int i = 0; i < 8; i++) n[i] = P::mod(i);
        uint32_t m[8];
        uint64_t acc = 0; uint32_t ovf = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int i = 0; i < k; i++) { mac(acc, ovf, a.l[i], b.l[k - i]); mac(acc, ovf, m[i], n[k - i]); }
            mac(acc, ovf, a.l[k], b.l[0]);
            m[k] = (uint32_t)acc * P::ninv;
            mac(acc, ovf, m[k], n[0]);
            shift(acc, ovf);
        }
        E r;
#pragma unroll
        for (int k = 8; k < 16; k++) {
#pragma unroll
            for (int i = k - 7; i < 8; i++) { mac(acc, ovf, a.l[i], b.l[k - i]); mac(acc, ovf, m[i], n[k - i]); }
            r.l[k - 8] = (uint32_t)acc;
            shift(acc, ovf);
        }
        // a, b < p  =>  result < 2p < 2^255: the ninth word is zero
        return reduce_once(r);
    }
    DEVFN static E sqr(const E& a) { return mul(a, a); }
    DEVFN static E to_mont(const E& canon) { return mul(canon, r2()); }
    DEVFN static E from_mont(const E& a) { E o = zero(); o.l[0] = 1; return mul(a, o); }
    DEVFN static E from_u32(uint32_t v) { E o = zero(); o.l[0] = v; return to_mont(o); }
    // a^e, e given as 8 little-endian 32-bit limbs (uniform across lanes)
    DEVNOINL static E pow(const E& a, const uint32_t* e) {
        E acc = one();
        bool started = false;
        for (int i = 255; i >= 0; i--) {
            if (started) acc = sqr(acc);
            if ((e[i >> 5] >> (i & 31)) & 1) { acc = started ? mul(acc, a) : a; started = true; }
        }
        return acc;
    }
    // Fermat inverse a^(p-2); 0 -> 0
    DEVFN static E inv(const E& a) {
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = P::mod(i);
        e[0] -= 2;
        return pow(a, e);
    }
    // canonical value > (p-1)/2 ?   (input in Montgomery form)
    DEVFN static bool lex_large(const E& a) {
        E c = from_mont(a);
        // compare with (p-1)/2
        uint32_t h[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { uint32_t lo = P::mod(i) - (i == 0 ? 1u : 0u); uint32_t hi = i < 7 ? P::mod(i + 1) : 0u; h[i] = (lo >> 1) | (hi << 31); }
        for (int i = 7; i >= 0; i--) { if (c.l[i] > h[i]) return true; if (c.l[i] < h[i]) return false; }
        return false;
    }
};

using Fp = Field<FpParams>;
using Fr = Field<FrParams>;

// ---- Fp2 = Fp[u]/(u^2+1) ----
struct alignas(16) fe2 { fe a0, a1; };
struct Fp2 {
    using E = fe2;
    DEVFN static E zero() { return E{Fp::zero(), Fp::zero()}; }
    DEVFN static E one() { return E{Fp::one(), Fp::zero()}; }
    DEVFN static bool is_zero(const E& a) { return Fp::is_zero(a.a0) && Fp::is_zero(a.a1); }
    DEVFN static bool eq(const E& a, const E& b) { return Fp::eq(a.a0, b.a0) && Fp::eq(a.a1, b.a1); }
    DEVFN static E add(const E& a, const E& b) { return E{Fp::add(a.a0, b.a0), Fp::add(a.a1, b.a1)}; }
    DEVFN static E sub(const E& a, const E& b) { return E{Fp::sub(a.a0, b.a0), Fp::sub(a.a1, b.a1)}; }
    DEVFN static E dbl(const E& a) { return add(a, a); }
    DEVFN static E neg(const E& a) { return E{Fp::neg(a.a0), Fp::neg(a.a1)}; }
    DEVFN static E mul(const E& a, const E& b) {
        fe t0 = Fp::mul(a.a0, b.a0), t1 = Fp::mul(a.a1, b.a1);
        fe t2 = Fp::mul(Fp::add(a.a0, a.a1), Fp::add(b.a0, b.a1));
        return E{Fp::sub(t0, t1), Fp::sub(Fp::sub(t2, t0), t1)};
    }
    DEVFN static E sqr(const E& a) {
        // (a0+a1)(a0-a1) + 2 a0 a1 u
        fe s = Fp::add(a.a0, a.a1), d = Fp::sub(a.a0, a.a1), m = Fp::mul(a.a0, a.a1);
        return E{Fp::mul(s, d), Fp::dbl(m)};
    }
    DEVFN static E inv(const E& a) {
        fe n = Fp::add(Fp::sqr(a.a0), Fp::sqr(a.a1));
        fe ni = Fp::inv(n);
        return E{Fp::mul(a.a0, ni), Fp::neg(Fp::mul(a.a1, ni))};
    }
    DEVFN static bool lex_large(const E& a) { return Fp::is_zero(a.a1) ? Fp::lex_large(a.a0) : Fp::lex_large(a.a1); }
};

// ---- affine point in this representation (what the key-decompression kernels produce); the group law itself lives in
// bn254_fp29.hpp, on the radix-2^29 field ----
template <class F>
struct Aff { typename F::E x, y; };

// ---- global-memory helpers: a field element is 32 B = two 16-B vector accesses ----
DEVFN fe load_fe(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    fe r; r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
DEVFN void store_fe(fe* p, const fe& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
}  // namespace bn254
