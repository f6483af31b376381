/* gsc_napi.c — a minimal N-API binding of the libprove C-ABI, so that the node caller (caller_napi.js) runs where `koffi`
 * cannot be installed (no network): it does by hand what koffi / ffi-napi generate — dlopen, symbols by name, GoSlice by value,
 * struct Prove_return by value, Free (reference libraries/prover/libprove.go:17-47; FFI hosts: reference README.md:24-25, :79-97).
 * Works against the reference's own cgo-built libprove.so as well.
 *
 *   gcc -O2 -shared -fPIC -I/usr/include/node -o build/gsc_napi.node integration/node_addon/gsc_napi.c -ldl
 *   node integration/caller_napi.js build/gsc_napi.node <libprove.so> [pk r1cs]
 */
#define NAPI_VERSION 4
#include <node_api.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

typedef struct { void *data; long long len; long long cap; } GoSlice;
struct Prove_return { void *r0; long long r1; };
static unsigned char (*p_InitAlgorithm)(unsigned char, GoSlice, GoSlice);
static struct Prove_return (*p_Prove)(GoSlice);
static struct Prove_return (*p_ProveBatch)(GoSlice);
static void (*p_Free)(void *);
static void (*p_enforce_binding)(void);

static napi_value fail(napi_env env, const char *msg) { napi_throw_error(env, NULL, msg); return NULL; }
static int buffer_arg(napi_env env, napi_value v, GoSlice *s) {
    void *data; size_t len; bool is;
    if (napi_is_buffer(env, v, &is) != napi_ok || !is || napi_get_buffer_info(env, v, &data, &len) != napi_ok) return 0;
    s->data = data; s->len = s->cap = (long long)len; return 1;
}

/* load(path) -> true; binds the four reference symbols (+ ProveBatch when the library has it) */
static napi_value Load(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1]; char path[4096]; size_t n;
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    if (argc < 1 || napi_get_value_string_utf8(env, argv[0], path, sizeof path, &n) != napi_ok) return fail(env, "load(path)");
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(env, dlerror());
    p_enforce_binding = (void (*)(void))dlsym(h, "enforce_binding");
    p_InitAlgorithm = (unsigned char (*)(unsigned char, GoSlice, GoSlice))dlsym(h, "InitAlgorithm");
    p_Prove = (struct Prove_return (*)(GoSlice))dlsym(h, "Prove");
    p_ProveBatch = (struct Prove_return (*)(GoSlice))dlsym(h, "ProveBatch");
    p_Free = (void (*)(void *))dlsym(h, "Free");
    if (!p_enforce_binding || !p_InitAlgorithm || !p_Prove || !p_Free) return fail(env, "libprove symbols missing");
    p_enforce_binding();
    napi_value t; napi_get_boolean(env, true, &t); return t;
}
/* initAlgorithm(id, pkBuffer, r1csBuffer) -> boolean */
static napi_value InitAlgorithm(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3]; uint32_t id; GoSlice pk, cs;
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    if (!p_InitAlgorithm || argc < 3 || napi_get_value_uint32(env, argv[0], &id) != napi_ok || !buffer_arg(env, argv[1], &pk) || !buffer_arg(env, argv[2], &cs))
        return fail(env, "initAlgorithm(id, Buffer, Buffer) after load()");
    napi_value r; napi_get_boolean(env, p_InitAlgorithm((unsigned char)id, pk, cs) != 0, &r); return r;
}
static napi_value call(napi_env env, napi_callback_info info, struct Prove_return (*fn)(GoSlice)) {
    size_t argc = 1; napi_value argv[1]; GoSlice in;
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    if (!fn || argc < 1 || !buffer_arg(env, argv[0], &in)) return fail(env, "prove(Buffer) after load()");
    struct Prove_return r = fn(in);
    if (!r.r0) return fail(env, "the library returned no result");
    napi_value out; void *copy;
    if (napi_create_buffer_copy(env, (size_t)r.r1, r.r0, &copy, &out) != napi_ok) { p_Free(r.r0); return fail(env, "out of memory"); }
    p_Free(r.r0);                                   /* malloc'd by the library, NOT NUL-terminated: length travels in r1 */
    return out;
}
static napi_value Prove(napi_env env, napi_callback_info info) { return call(env, info, p_Prove); }
static napi_value ProveBatch(napi_env env, napi_callback_info info) { return call(env, info, p_ProveBatch); }

NAPI_MODULE_INIT() {
    napi_property_descriptor d[] = {
        {"load", NULL, Load, NULL, NULL, NULL, napi_default, NULL}, {"initAlgorithm", NULL, InitAlgorithm, NULL, NULL, NULL, napi_default, NULL},
        {"prove", NULL, Prove, NULL, NULL, NULL, napi_default, NULL}, {"proveBatch", NULL, ProveBatch, NULL, NULL, NULL, napi_default, NULL}};
    napi_define_properties(env, exports, 4, d);
    return exports;
}
