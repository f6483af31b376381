/* TEST ORACLE — not product code (see bn254.h).
 *
 * Native ciphers the reference runs on the host to obtain the ciphertext that becomes part of the
 * public witness (libraries/prover/impl/provers.go:93-101 ChaCha20 via x/crypto/chacha20;
 * :184-192 AES-CTR via crypto/aes + cipher.NewCTR), plus SHA-256 / RFC 9380 expand_message_xmd
 * used by gnark's commitment hash_to_field.  Restated from RFC 7539,
 * FIPS-197, FIPS 180-4 and RFC 9380; pinned by the RFC vectors in tests/test_oracle.py.
 */
#include "ciphers.h"
#include <string.h>

static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d) a += b; d ^= a; d = rotl(d, 16); c += d; b ^= c; b = rotl(b, 12); a += b; d ^= a; d = rotl(d, 8); c += d; b ^= c; b = rotl(b, 7);
static uint32_t ld32le(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

void chacha20_block(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter, uint8_t out[64]) {
    uint32_t s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574}, x[16];
    for (int i = 0; i < 8; i++) s[4 + i] = ld32le(key + 4 * i);
    s[12] = counter;
    for (int i = 0; i < 3; i++) s[13 + i] = ld32le(nonce + 4 * i);
    memcpy(x, s, sizeof x);
    for (int r = 0; r < 10; r++) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) { uint32_t v = x[i] + s[i]; out[4 * i] = (uint8_t)v; out[4 * i + 1] = (uint8_t)(v >> 8); out[4 * i + 2] = (uint8_t)(v >> 16); out[4 * i + 3] = (uint8_t)(v >> 24); }
}
void chacha20_xor(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter, const uint8_t *in, uint8_t *out, size_t len) {
    uint8_t ks[64];
    for (size_t off = 0; off < len; off += 64, counter++) {
        chacha20_block(key, nonce, counter, ks);
        for (size_t i = 0; i < 64 && off + i < len; i++) out[off + i] = in[off + i] ^ ks[i];
    }
}

/* ---- AES (FIPS-197), table-free ---- */
static uint8_t SBOX[256]; static int sbox_ready = 0;
static uint8_t gmul(uint8_t a, uint8_t b) { uint8_t p = 0; for (int i = 0; i < 8; i++) { if (b & 1) p ^= a; uint8_t hi = a & 0x80; a <<= 1; if (hi) a ^= 0x1b; b >>= 1; } return p; }
static void sbox_init(void) {
    if (sbox_ready) return;
    for (int x = 0; x < 256; x++) {
        uint8_t inv = 0;
        if (x) for (int y = 1; y < 256; y++) if (gmul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv;
        for (int k = 1; k <= 4; k++) s ^= (uint8_t)((inv << k) | (inv >> (8 - k)));
        SBOX[x] = s ^ 0x63;
    }
    sbox_ready = 1;
}
void aes_encrypt_block(const uint8_t *key, int keylen, const uint8_t in[16], uint8_t out[16]) {
    sbox_init();
    int nk = keylen / 4, nr = nk + 6;
    uint8_t rk[15 * 16];
    memcpy(rk, key, (size_t)keylen);
    uint8_t rcon = 1;
    for (int i = nk; i < 4 * (nr + 1); i++) {
        uint8_t t[4]; memcpy(t, rk + 4 * (i - 1), 4);
        if (i % nk == 0) { uint8_t u = t[0]; t[0] = SBOX[t[1]] ^ rcon; t[1] = SBOX[t[2]]; t[2] = SBOX[t[3]]; t[3] = SBOX[u]; rcon = gmul(rcon, 2); }
        else if (nk > 6 && i % nk == 4) for (int k = 0; k < 4; k++) t[k] = SBOX[t[k]];
        for (int k = 0; k < 4; k++) rk[4 * i + k] = rk[4 * (i - nk) + k] ^ t[k];
    }
    uint8_t s[16]; for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[i];
    for (int r = 1; r <= nr; r++) {
        uint8_t t[16];
        for (int c = 0; c < 4; c++) for (int row = 0; row < 4; row++) t[4 * c + row] = SBOX[s[4 * ((c + row) % 4) + row]];
        if (r < nr) for (int c = 0; c < 4; c++) {
            uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
            s[4 * c] = gmul(a0, 2) ^ gmul(a1, 3) ^ a2 ^ a3; s[4 * c + 1] = a0 ^ gmul(a1, 2) ^ gmul(a2, 3) ^ a3;
            s[4 * c + 2] = a0 ^ a1 ^ gmul(a2, 2) ^ gmul(a3, 3); s[4 * c + 3] = gmul(a0, 3) ^ a1 ^ a2 ^ gmul(a3, 2);
        } else memcpy(s, t, 16);
        for (int i = 0; i < 16; i++) s[i] ^= rk[16 * r + i];
    }
    memcpy(out, s, 16);
}
/* CTR with IV = nonce(12) || BE32(counter), 128-bit big-endian increment (Go cipher.NewCTR) */
void aes_ctr_xor(const uint8_t *key, int keylen, const uint8_t nonce[12], uint32_t counter, const uint8_t *in, uint8_t *out, size_t len) {
    uint8_t iv[16], ks[16]; memcpy(iv, nonce, 12);
    iv[12] = (uint8_t)(counter >> 24); iv[13] = (uint8_t)(counter >> 16); iv[14] = (uint8_t)(counter >> 8); iv[15] = (uint8_t)counter;
    for (size_t off = 0; off < len; off += 16) {
        aes_encrypt_block(key, keylen, iv, ks);
        for (size_t i = 0; i < 16 && off + i < len; i++) out[off + i] = in[off + i] ^ ks[i];
        for (int k = 15; k >= 0; k--) if (++iv[k]) break;
    }
}

/* ---- SHA-256 ---- */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
void sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t total = ((len + 9 + 63) / 64) * 64;
    for (size_t off = 0; off < total; off += 64) {
        uint8_t blk[64];
        for (size_t i = 0; i < 64; i++) {
            size_t p = off + i;
            if (p < len) blk[i] = msg[p];
            else if (p == len) blk[i] = 0x80;
            else if (p >= total - 8) blk[i] = (uint8_t)(((uint64_t)len * 8) >> (8 * (total - 1 - p)));
            else blk[i] = 0;
        }
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K256[i] + w[i];
            uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}
/* RFC 9380 5.3.1 expand_message_xmd with SHA-256; len_in_bytes <= 255*32, dst_len <= 255 */
int expand_message_xmd(const uint8_t *msg, size_t msg_len, const uint8_t *dst, size_t dst_len, uint8_t *out, size_t out_len) {
    size_t ell = (out_len + 31) / 32;
    if (ell > 255 || dst_len > 255 || msg_len > 4096) return -1;
    uint8_t buf[64 + 4096 + 2 + 1 + 256], b0[32], bi[32];
    size_t n = 0;
    memset(buf, 0, 64); n = 64;
    memcpy(buf + n, msg, msg_len); n += msg_len;
    buf[n++] = (uint8_t)(out_len >> 8); buf[n++] = (uint8_t)out_len; buf[n++] = 0;
    memcpy(buf + n, dst, dst_len); n += dst_len; buf[n++] = (uint8_t)dst_len;
    sha256(buf, n, b0);
    uint8_t t[32 + 1 + 256]; memcpy(t, b0, 32); t[32] = 1; memcpy(t + 33, dst, dst_len); t[33 + dst_len] = (uint8_t)dst_len;
    sha256(t, 34 + dst_len, bi);
    for (size_t i = 1;; i++) {
        size_t off = (i - 1) * 32, take = out_len - off < 32 ? out_len - off : 32;
        memcpy(out + off, bi, take);
        if (i == ell) break;
        for (int k = 0; k < 32; k++) t[k] = b0[k] ^ bi[k];
        t[32] = (uint8_t)(i + 1);
        sha256(t, 34 + dst_len, bi);
    }
    return 0;
}
