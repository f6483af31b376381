#include "json.hpp"
#include <cstring>

namespace gsc {

const char* JsonValue::go_kind() const {
    switch (kind) { case Null: return "null"; case Bool: return "bool"; case Number: return "number"; case String: return "string"; case Array: return "array"; default: return "object"; }
}

namespace {
struct Parser {
    const char* d; size_t n; size_t i = 0; int depth = 0;
    [[noreturn]] void fail(const char* m) { throw JsonSyntaxError(m, i); }
    void ws() { while (i < n && (d[i] == ' ' || d[i] == '\t' || d[i] == '\n' || d[i] == '\r')) i++; }
    static void utf8(std::string& o, uint32_t cp) {
        if (cp < 0x80) o += char(cp);
        else if (cp < 0x800) { o += char(0xC0 | (cp >> 6)); o += char(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += char(0xE0 | (cp >> 12)); o += char(0x80 | ((cp >> 6) & 0x3F)); o += char(0x80 | (cp & 0x3F)); }
        else { o += char(0xF0 | (cp >> 18)); o += char(0x80 | ((cp >> 12) & 0x3F)); o += char(0x80 | ((cp >> 6) & 0x3F)); o += char(0x80 | (cp & 0x3F)); }
    }
    uint32_t hex4() {
        if (i + 4 > n) fail("unexpected end of JSON input");
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) { char c = d[i++]; v <<= 4; if (c >= '0' && c <= '9') v |= c - '0'; else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10; else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10; else fail("invalid character in \\u hexadecimal character escape"); }
        return v;
    }
    std::string str() {
        std::string o; i++;   // opening quote
        for (;;) {
            if (i >= n) fail("unexpected end of JSON input");
            unsigned char c = d[i++];
            if (c == '"') return o;
            if (c < 0x20) fail("invalid character in string literal");
            if (c != '\\') { o += char(c); continue; }
            if (i >= n) fail("unexpected end of JSON input");
            char e = d[i++];
            switch (e) {
                case '"': o += '"'; break; case '\\': o += '\\'; break; case '/': o += '/'; break;
                case 'b': o += '\b'; break; case 'f': o += '\f'; break; case 'n': o += '\n'; break; case 'r': o += '\r'; break; case 't': o += '\t'; break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && i + 1 < n && d[i] == '\\' && d[i + 1] == 'u') { i += 2; uint32_t lo = hex4(); if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); else cp = 0xFFFD; }
                    else if (cp >= 0xD800 && cp < 0xE000) cp = 0xFFFD;
                    utf8(o, cp); break;
                }
                default: fail("invalid character in string escape code");
            }
        }
    }
    JsonValue value() {
        if (++depth > JSON_MAX_DEPTH) fail("exceeded max depth");
        ws();
        if (i >= n) fail("unexpected end of JSON input");
        JsonValue v; char c = d[i]; v.start = i;
        if (c == '{') {
            v.kind = JsonValue::Object; i++; ws();
            if (i < n && d[i] == '}') i++;
            else for (;;) {
                ws(); if (i >= n) fail("unexpected end of JSON input");
                if (d[i] != '"') fail("invalid character looking for beginning of object key string");
                std::string k = str(); ws();
                if (i >= n) fail("unexpected end of JSON input");
                if (d[i] != ':') fail("invalid character after object key");
                i++;
                JsonValue m = value(); v.members.emplace_back(std::move(k), std::move(m)); ws();
                if (i >= n) fail("unexpected end of JSON input");
                if (d[i] == ',') { i++; continue; }
                if (d[i] == '}') { i++; break; }
                fail("invalid character after object key:value pair");
            }
        } else if (c == '[') {
            v.kind = JsonValue::Array; i++; ws();
            if (i < n && d[i] == ']') i++;
            else for (;;) {
                v.items.push_back(value()); ws();
                if (i >= n) fail("unexpected end of JSON input");
                if (d[i] == ',') { i++; continue; }
                if (d[i] == ']') { i++; break; }
                fail("invalid character after array element");
            }
        } else if (c == '"') { v.kind = JsonValue::String; v.text = str(); }
        else if (c == '-' || (c >= '0' && c <= '9')) {
            size_t s = i; if (d[i] == '-') i++;
            if (i >= n) fail("unexpected end of JSON input");
            if (d[i] == '0') i++;
            else if (d[i] >= '1' && d[i] <= '9') { while (i < n && d[i] >= '0' && d[i] <= '9') i++; }
            else fail("invalid character in numeric literal");
            if (i < n && d[i] == '.') { i++; if (i >= n || d[i] < '0' || d[i] > '9') fail("invalid character after decimal point in numeric literal"); while (i < n && d[i] >= '0' && d[i] <= '9') i++; }
            if (i < n && (d[i] == 'e' || d[i] == 'E')) { i++; if (i < n && (d[i] == '+' || d[i] == '-')) i++; if (i >= n || d[i] < '0' || d[i] > '9') fail("invalid character in exponent of numeric literal"); while (i < n && d[i] >= '0' && d[i] <= '9') i++; }
            v.kind = JsonValue::Number; v.text.assign(d + s, i - s);
        } else if (n - i >= 4 && !memcmp(d + i, "true", 4)) { v.kind = JsonValue::Bool; v.b = true; i += 4; }
        else if (n - i >= 5 && !memcmp(d + i, "false", 5)) { v.kind = JsonValue::Bool; v.b = false; i += 5; }
        else if (n - i >= 4 && !memcmp(d + i, "null", 4)) { v.kind = JsonValue::Null; i += 4; }
        else fail("invalid character looking for beginning of value");
        v.offset = i; depth--;
        return v;
    }
};
const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
}  // namespace

JsonValue json_parse(const char* data, size_t len) {
    Parser p{data, len};
    JsonValue v = p.value();
    p.ws();
    if (p.i != len) throw JsonSyntaxError("invalid character after top-level value", p.i);
    return v;
}

std::string json_quote(const std::string& s) {
    static const char* hex = "0123456789abcdef";
    std::string o = "\"";
    for (size_t i = 0; i < s.size(); i++) {
        unsigned char c = s[i];
        if (c == '"' || c == '\\') { o += '\\'; o += char(c); }
        else if (c == '\n') o += "\\n"; else if (c == '\r') o += "\\r"; else if (c == '\t') o += "\\t";
        else if (c < 0x20 || c == '<' || c == '>' || c == '&') { o += "\\u00"; o += hex[c >> 4]; o += hex[c & 15]; }
        else if (c == 0xE2 && i + 2 < s.size() && (unsigned char)s[i + 1] == 0x80 && ((unsigned char)s[i + 2] == 0xA8 || (unsigned char)s[i + 2] == 0xA9)) { o += "\\u202"; o += hex[(unsigned char)s[i + 2] & 15]; i += 2; }
        else o += char(c);
    }
    return o + "\"";
}

std::string base64_encode(const uint8_t* p, size_t n) {
    std::string o; o.reserve((n + 2) / 3 * 4);
    size_t i = 0;
    for (; i + 3 <= n; i += 3) { uint32_t v = (p[i] << 16) | (p[i + 1] << 8) | p[i + 2]; o += B64[v >> 18]; o += B64[(v >> 12) & 63]; o += B64[(v >> 6) & 63]; o += B64[v & 63]; }
    if (n - i == 1) { uint32_t v = p[i] << 16; o += B64[v >> 18]; o += B64[(v >> 12) & 63]; o += "=="; }
    else if (n - i == 2) { uint32_t v = (p[i] << 16) | (p[i + 1] << 8); o += B64[v >> 18]; o += B64[(v >> 12) & 63]; o += B64[(v >> 6) & 63]; o += '='; }
    return o;
}

bool base64_decode(const std::string& s, std::vector<uint8_t>& out, size_t& bad_offset) {
    static int8_t rev[256]; static bool init = false;
    if (!init) { memset(rev, -1, sizeof rev); for (int i = 0; i < 64; i++) rev[(unsigned char)B64[i]] = (int8_t)i; init = true; }
    out.clear();
    // Go's StdEncoding ignores \r and \n
    std::string t; std::vector<size_t> pos;
    for (size_t i = 0; i < s.size(); i++) if (s[i] != '\r' && s[i] != '\n') { t += s[i]; pos.push_back(i); }
    if (t.size() % 4) { bad_offset = pos.empty() ? 0 : (t.size() / 4 * 4 < pos.size() ? pos[t.size() / 4 * 4] : s.size());
        // a trailing partial quantum is reported at its start unless an earlier character is invalid
        for (size_t i = 0; i < t.size(); i++) if (t[i] != '=' && rev[(unsigned char)t[i]] < 0) { bad_offset = pos[i]; break; }
        return false; }
    for (size_t i = 0; i < t.size(); i += 4) {
        int v[4]; int pad = 0;
        for (int k = 0; k < 4; k++) {
            char c = t[i + k];
            if (c == '=') { if (i + 4 != t.size() || k < 2) { bad_offset = pos[i + k]; return false; } pad++; v[k] = 0; }
            else { if (pad || rev[(unsigned char)c] < 0) { bad_offset = pos[i + k]; return false; } v[k] = rev[(unsigned char)c]; }
        }
        uint32_t w = (v[0] << 18) | (v[1] << 12) | (v[2] << 6) | v[3];
        out.push_back(uint8_t(w >> 16)); if (pad < 2) out.push_back(uint8_t(w >> 8)); if (pad < 1) out.push_back(uint8_t(w));
    }
    return true;
}

}  // namespace gsc
