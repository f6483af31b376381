#include "host_ciphers.hpp"
#include <array>
#include <cstring>

namespace gsc {
namespace {
inline uint32_t rol(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
inline uint32_t get_le(const uint8_t* p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24; }
inline void quarter(std::array<uint32_t, 16>& x, int a, int b, int c, int d) {
    x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 16);
    x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 12);
    x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 8);
    x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 7);
}
}  // namespace

// RFC 7539 section 2.3/2.4: 32-bit block counter in word 12, 96-bit nonce in words 13..15
void chacha20_xor_stream(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter, const uint8_t* in, uint8_t* out, size_t len) {
    std::array<uint32_t, 16> st{};
    st[0] = 0x61707865u; st[1] = 0x3320646eu; st[2] = 0x79622d32u; st[3] = 0x6b206574u;   // "expand 32-byte k"
    for (int i = 0; i < 8; i++) st[4 + i] = get_le(key + 4 * i);
    for (int i = 0; i < 3; i++) st[13 + i] = get_le(nonce + 4 * i);
    for (size_t off = 0; off < len; off += 64) {
        st[12] = counter++;
        std::array<uint32_t, 16> w = st;
        for (int round = 0; round < 20; round += 2) {
            quarter(w, 0, 4, 8, 12); quarter(w, 1, 5, 9, 13); quarter(w, 2, 6, 10, 14); quarter(w, 3, 7, 11, 15);
            quarter(w, 0, 5, 10, 15); quarter(w, 1, 6, 11, 12); quarter(w, 2, 7, 8, 13); quarter(w, 3, 4, 9, 14);
        }
        const size_t take = len - off < 64 ? len - off : 64;
        for (size_t i = 0; i < take; i++) {
            const uint32_t word = w[i / 4] + st[i / 4];
            out[off + i] = in[off + i] ^ uint8_t(word >> (8 * (i % 4)));
        }
    }
}

namespace {
// FIPS-197 with the S-box derived from GF(2^8) log tables (generator 3)
struct AesTables {
    uint8_t sbox[256];
    AesTables() {
        uint8_t exp[256], log[256]; uint8_t x = 1;
        for (int i = 0; i < 255; i++) { exp[i] = x; log[x] = uint8_t(i); x = uint8_t(x ^ (x << 1) ^ ((x & 0x80) ? 0x1b : 0)); }
        exp[255] = exp[0];
        for (int v = 0; v < 256; v++) {
            uint8_t inv = v ? exp[255 - log[v]] : 0;
            uint8_t s = inv, r = inv;
            for (int k = 0; k < 4; k++) { r = uint8_t((r << 1) | (r >> 7)); s ^= r; }
            sbox[v] = s ^ 0x63;
        }
    }
};
const AesTables& tables() { static AesTables t; return t; }
inline uint8_t xtime(uint8_t v) { return uint8_t((v << 1) ^ ((v & 0x80) ? 0x1b : 0)); }

struct AesKey {
    uint8_t rk[15][16]; int rounds;
    AesKey(const uint8_t* key, size_t keylen) {
        const uint8_t* S = tables().sbox;
        const int nk = int(keylen / 4); rounds = nk + 6;
        uint8_t* w = &rk[0][0];
        memcpy(w, key, keylen);
        uint8_t rc = 1;
        for (int i = nk; i < 4 * (rounds + 1); i++) {
            uint8_t t[4] = {w[4 * (i - 1)], w[4 * (i - 1) + 1], w[4 * (i - 1) + 2], w[4 * (i - 1) + 3]};
            if (i % nk == 0) { const uint8_t t0 = t[0]; t[0] = S[t[1]] ^ rc; t[1] = S[t[2]]; t[2] = S[t[3]]; t[3] = S[t0]; rc = xtime(rc); }
            else if (nk == 8 && i % nk == 4) { for (auto& b : t) b = S[b]; }
            for (int k = 0; k < 4; k++) w[4 * i + k] = w[4 * (i - nk) + k] ^ t[k];
        }
    }
    void encrypt(const uint8_t in[16], uint8_t out[16]) const {
        const uint8_t* S = tables().sbox;
        uint8_t s[16];
        for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[0][i];
        for (int r = 1; r <= rounds; r++) {
            uint8_t t[16];
            for (int col = 0; col < 4; col++) for (int row = 0; row < 4; row++) t[4 * col + row] = S[s[4 * ((col + row) & 3) + row]];
            if (r != rounds) {
                for (int col = 0; col < 4; col++) {
                    const uint8_t* a = t + 4 * col; const uint8_t all = a[0] ^ a[1] ^ a[2] ^ a[3];
                    s[4 * col + 0] = a[0] ^ all ^ xtime(a[0] ^ a[1]); s[4 * col + 1] = a[1] ^ all ^ xtime(a[1] ^ a[2]);
                    s[4 * col + 2] = a[2] ^ all ^ xtime(a[2] ^ a[3]); s[4 * col + 3] = a[3] ^ all ^ xtime(a[3] ^ a[0]);
                }
            } else memcpy(s, t, 16);
            for (int i = 0; i < 16; i++) s[i] ^= rk[r][i];
        }
        memcpy(out, s, 16);
    }
};
}  // namespace

void aes_ctr_xor_stream(const uint8_t* key, size_t keylen, const uint8_t nonce[12], uint32_t counter, const uint8_t* in, uint8_t* out, size_t len) {
    AesKey k(key, keylen);
    uint8_t iv[16], ks[16];
    memcpy(iv, nonce, 12);
    iv[12] = uint8_t(counter >> 24); iv[13] = uint8_t(counter >> 16); iv[14] = uint8_t(counter >> 8); iv[15] = uint8_t(counter);
    for (size_t off = 0; off < len; off += 16) {
        k.encrypt(iv, ks);
        const size_t take = len - off < 16 ? len - off : 16;
        for (size_t i = 0; i < take; i++) out[off + i] = in[off + i] ^ ks[i];
        for (int b = 15; b >= 0 && ++iv[b] == 0; b--) {}   // 128-bit big-endian increment (Go cipher.NewCTR)
    }
}

namespace {
struct Sha256 {
    uint32_t h[8]; uint8_t buf[64]; size_t fill = 0; uint64_t total = 0;
    Sha256() { static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u}; memcpy(h, iv, sizeof h); }
    static uint32_t ror(uint32_t v, int s) { return (v >> s) | (v << (32 - s)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
            0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
            0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
            0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int t = 0; t < 16; t++) w[t] = uint32_t(p[4 * t]) << 24 | uint32_t(p[4 * t + 1]) << 16 | uint32_t(p[4 * t + 2]) << 8 | p[4 * t + 3];
        for (int t = 16; t < 64; t++) w[t] = w[t - 16] + (ror(w[t - 15], 7) ^ ror(w[t - 15], 18) ^ (w[t - 15] >> 3)) + w[t - 7] + (ror(w[t - 2], 17) ^ ror(w[t - 2], 19) ^ (w[t - 2] >> 10));
        uint32_t v[8]; memcpy(v, h, sizeof v);
        for (int t = 0; t < 64; t++) {
            const uint32_t t1 = v[7] + (ror(v[4], 6) ^ ror(v[4], 11) ^ ror(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K[t] + w[t];
            const uint32_t t2 = (ror(v[0], 2) ^ ror(v[0], 13) ^ ror(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
            for (int k = 7; k > 0; k--) v[k] = v[k - 1];
            v[4] += t1; v[0] = t1 + t2;
        }
        for (int k = 0; k < 8; k++) h[k] += v[k];
    }
    void update(const uint8_t* p, size_t n) {
        total += n;
        while (n) { const size_t take = 64 - fill < n ? 64 - fill : n; memcpy(buf + fill, p, take); fill += take; p += take; n -= take; if (fill == 64) { block(buf); fill = 0; } }
    }
    void finish(uint8_t out[32]) {
        const uint64_t bits = total * 8; uint8_t pad[72] = {0x80}; const size_t padlen = (fill < 56 ? 56 : 120) - fill;
        uint8_t len[8]; for (int k = 0; k < 8; k++) len[k] = uint8_t(bits >> (56 - 8 * k));
        update(pad, padlen); update(len, 8);
        for (int k = 0; k < 8; k++) { out[4 * k] = uint8_t(h[k] >> 24); out[4 * k + 1] = uint8_t(h[k] >> 16); out[4 * k + 2] = uint8_t(h[k] >> 8); out[4 * k + 3] = uint8_t(h[k]); }
    }
};
}  // namespace

void sha256_digest(const uint8_t* msg, size_t len, uint8_t out[32]) { Sha256 s; s.update(msg, len); s.finish(out); }

// RFC 9380 section 5.3.1
void expand_message_xmd_sha256(const uint8_t* msg, size_t msg_len, const char* dst, uint8_t* out, size_t out_len) {
    const size_t dst_len = strlen(dst), ell = (out_len + 31) / 32;
    const uint8_t dst_tail = uint8_t(dst_len);
    uint8_t zpad[64] = {0}, lib[3] = {uint8_t(out_len >> 8), uint8_t(out_len), 0}, b0[32], bi[32];
    { Sha256 s; s.update(zpad, 64); s.update(msg, msg_len); s.update(lib, 3); s.update((const uint8_t*)dst, dst_len); s.update(&dst_tail, 1); s.finish(b0); }
    { Sha256 s; const uint8_t one = 1; s.update(b0, 32); s.update(&one, 1); s.update((const uint8_t*)dst, dst_len); s.update(&dst_tail, 1); s.finish(bi); }
    for (size_t i = 1;; i++) {
        const size_t off = (i - 1) * 32, take = out_len - off < 32 ? out_len - off : 32;
        memcpy(out + off, bi, take);
        if (i == ell) break;
        uint8_t x[32]; for (int k = 0; k < 32; k++) x[k] = b0[k] ^ bi[k];
        Sha256 s; const uint8_t idx = uint8_t(i + 1); s.update(x, 32); s.update(&idx, 1); s.update((const uint8_t*)dst, dst_len); s.update(&dst_tail, 1); s.finish(bi);
    }
}
}  // namespace gsc
