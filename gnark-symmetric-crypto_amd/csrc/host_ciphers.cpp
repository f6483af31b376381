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
}  // namespace gsc
