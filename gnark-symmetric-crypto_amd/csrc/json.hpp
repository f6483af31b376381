// Minimal JSON reader / base64 codec for the libprove JSON boundary
// (reference libraries/prover/impl/prove_impl.go:116-143 uses encoding/json; the decoding rules that callers
// can observe are listed in SURVEY.md §8(b)).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <stdexcept>

namespace gsc {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    std::string text;                      // String: decoded text; Number: literal as written
    std::vector<JsonValue> items;          // Array
    std::vector<std::pair<std::string, JsonValue>> members;   // Object, in document order
    size_t start = 0;                      // byte offset of the value's first character
    size_t offset = 0;                     // byte offset just past the value
    const char* go_kind() const;           // "array", "string", "number", "bool", "object", "null"
};

struct JsonSyntaxError : std::runtime_error {
    size_t offset;
    JsonSyntaxError(const std::string& m, size_t off) : std::runtime_error(m), offset(off) {}
};

// Nesting limit of the recursive-descent reader.  InputParams needs 2 levels and a ProveBatch array 3; encoding/json's own
// limit (10000) would need megabytes of native stack, which FFI worker threads (musl, node workers) do not have.
constexpr int JSON_MAX_DEPTH = 64;
JsonValue json_parse(const char* data, size_t len);      // throws JsonSyntaxError
std::string json_quote(const std::string& s);            // Go-compatible string escaping (HTML-safe escapes included)

std::string base64_encode(const uint8_t* p, size_t n);
// std alphabet, padded (Go base64.StdEncoding).  Returns false and sets bad_offset on corrupt input.
bool base64_decode(const std::string& s, std::vector<uint8_t>& out, size_t& bad_offset);

}  // namespace gsc
