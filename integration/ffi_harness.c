/* ffi_harness.c — calls libprove.so exactly the way a foreign-function host does.
 *
 * The reference's prover is consumed through FFI (node.js / Android / iOS, reference README.md:24-25, :79-97) by dlopen-ing
 * the cgo-built library and binding enforce_binding / InitAlgorithm / Prove / Free (libraries/prover/libprove.go:17-47).
 * This harness does the same from plain C with no header of ours: symbols by name, GoSlice passed BY VALUE, struct
 * Prove_return returned by value, result released with Free — what koffi / ffi-napi generate.
 *
 *   ffi_harness <libprove.so> errors                       error-path checks (no GPU needed; exit 0 = all as the reference behaves)
 *   ffi_harness <libprove.so> prove <pk> <r1cs> [json]     InitAlgorithm(0, pk, r1cs) + Prove(json) -> prints the result JSON
 *   ffi_harness <libprove.so> callers <pk> <r1cs> <seconds> <C> [C ...]
 *                                                          C threads, each calling Prove in a closed loop (the next call when the previous one
 *                                                          has returned: libraries/core_test.go:44-111) -> proofs/s and mean latency per C
 *
 * build: gcc -O2 -o build/ffi_harness integration/ffi_harness.c -ldl -lpthread
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { void *data; long long len; long long cap; } GoSlice;
struct Prove_return { void *r0; long long r1; };
typedef void (*enforce_binding_fn)(void);
typedef unsigned char (*InitAlgorithm_fn)(unsigned char, GoSlice, GoSlice);
typedef struct Prove_return (*Prove_fn)(GoSlice);
typedef void (*Free_fn)(void *);

static GoSlice slice(const void *p, size_t n) { GoSlice s = {(void *)p, (long long)n, (long long)n}; return s; }
static void *read_file(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    void *b = malloc(sz ? sz : 1); if (fread(b, 1, sz, f) != (size_t)sz) { perror("read"); exit(2); }
    fclose(f); *n = (size_t)sz; return b;
}
static int expect(Prove_fn Prove, Free_fn Free, const char *in, const char *want) {
    struct Prove_return r = Prove(slice(in, strlen(in)));
    int ok = r.r0 && (size_t)r.r1 == strlen(want) && !memcmp(r.r0, want, strlen(want));
    if (!ok) fprintf(stderr, "MISMATCH for %s\n  got  %.*s\n  want %s\n", in, (int)r.r1, r.r0 ? (char *)r.r0 : "", want);
    Free(r.r0);
    return ok;
}

/* ---- callers mode ---- */
static Prove_fn g_Prove; static Free_fn g_Free;
static double g_stop_at;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
struct caller { int id, stride; long calls, failed; double busy; };
static void *caller_main(void *arg) {
    struct caller *c = (struct caller *)arg;
    char json[1024];
    unsigned k = (unsigned)c->id;
    while (now_s() < g_stop_at) {
        /* every call is a different statement: key byte, nonce byte and counter follow the call number */
        int n = snprintf(json, sizeof json, "{\"cipher\":\"chacha20\",\"key\":[%u,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2],"
                         "\"nonce\":[%u,3,3,3,3,3,3,3,3,3,3,3],\"counter\":%u,"
                         "\"input\":\"o/flkq7aFQen9Rs1gS38UKJj1abS32JeVjsC5JwIvzDQ50g/WxP/B5UyIk7o+8MasYmbGORT022Xk6g1XrDe6Q==\"}", k & 255u, (k >> 8) & 255u, k);
        const double t0 = now_s();
        struct Prove_return r = g_Prove(slice(json, (size_t)n));
        c->busy += now_s() - t0;
        if (!r.r0 || r.r1 < 20 || !memmem(r.r0, (size_t)r.r1, "\"proofJson\"", 11)) c->failed++;
        g_Free(r.r0);
        c->calls++; k += (unsigned)c->stride;
    }
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <libprove.so> errors | prove <pk> <r1cs> [json] | callers <pk> <r1cs> <seconds> <C> [C ...]\n", argv[0]); return 2; }
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    enforce_binding_fn enforce_binding = (enforce_binding_fn)dlsym(h, "enforce_binding");
    InitAlgorithm_fn InitAlgorithm = (InitAlgorithm_fn)dlsym(h, "InitAlgorithm");
    Prove_fn Prove = (Prove_fn)dlsym(h, "Prove");
    Free_fn Free = (Free_fn)dlsym(h, "Free");
    if (!enforce_binding || !InitAlgorithm || !Prove || !Free) { fprintf(stderr, "missing symbol\n"); return 2; }
    enforce_binding();
    if (!strcmp(argv[2], "errors")) {
        int ok = 1;
        /* libprove.go:33-43: a recovered panic is returned as its JSON encoding; core_test.go:120-128 (TestPanic) */
        ok &= expect(Prove, Free, "{\"cipher\":\"nope\"}", "\"could not find prover fornope\"");
        ok &= expect(Prove, Free, "{\"cipher\":\"chacha20\"}", "\"proving params are not initialized for cipher: chacha20\"");
        ok &= expect(Prove, Free, "", "{\"Offset\":0}");
        ok &= expect(Prove, Free, "null", "\"runtime error: invalid memory address or nil pointer dereference\"");
        ok &= expect(Prove, Free, "{\"cipher\":5}", "{\"Value\":\"number\",\"Type\":{},\"Offset\":11,\"Struct\":\"InputParams\",\"Field\":\"cipher\"}");
        /* prove_impl.go:113: unknown algorithm id -> false; :89-90: unreadable key -> false (and no crash on empty slices) */
        ok &= InitAlgorithm(9, slice("x", 1), slice("y", 1)) == 0;
        ok &= InitAlgorithm(0, slice(NULL, 0), slice(NULL, 0)) == 0;
        ok &= InitAlgorithm(0, slice("garbage", 7), slice("garbage", 7)) == 0;
        Free(NULL);
        puts(ok ? "FFI-ERRORS-OK" : "FFI-ERRORS-FAILED");
        return ok ? 0 : 1;
    }
    if (!strcmp(argv[2], "prove") && argc >= 5) {
        size_t npk, ncs; void *pk = read_file(argv[3], &npk), *cs = read_file(argv[4], &ncs);
        if (!InitAlgorithm(0, slice(pk, npk), slice(cs, ncs))) { puts("InitAlgorithm failed"); return 1; }
        if (!InitAlgorithm(0, slice(pk, npk), slice(cs, ncs))) { puts("second InitAlgorithm must report success"); return 1; }   /* prove_impl.go:74-76 */
        free(pk); free(cs);                       /* the library copied what it needs: input slices are only read during the call */
        const char *json = argc > 5 ? argv[5] :
            "{\"cipher\":\"chacha20\",\"key\":[2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2,2],\"nonce\":[3,3,3,3,3,3,3,3,3,3,3,3],\"counter\":3,"
            "\"input\":\"o/flkq7aFQen9Rs1gS38UKJj1abS32JeVjsC5JwIvzDQ50g/WxP/B5UyIk7o+8MasYmbGORT022Xk6g1XrDe6Q==\"}";      /* core_test.go:285 */
        struct Prove_return r = Prove(slice(json, strlen(json)));
        if (!r.r0) { puts("Prove returned NULL"); return 1; }
        fwrite(r.r0, 1, (size_t)r.r1, stdout); putchar('\n');
        Free(r.r0);
        return 0;
    }
    if (!strcmp(argv[2], "callers") && argc >= 7) {
        size_t npk, ncs; void *pk = read_file(argv[3], &npk), *cs = read_file(argv[4], &ncs);
        if (!InitAlgorithm(0, slice(pk, npk), slice(cs, ncs))) { puts("InitAlgorithm failed"); return 1; }
        free(pk); free(cs);
        g_Prove = Prove; g_Free = Free;
        const double secs = atof(argv[5]);
        long bad = 0;
        for (int a = 6; a < argc; a++) {
            const int C = atoi(argv[a]);
            if (C < 1 || C > 4096) { fprintf(stderr, "callers: 1..4096\n"); return 2; }
            struct caller *cs_ = calloc((size_t)C, sizeof *cs_); pthread_t *th = calloc((size_t)C, sizeof *th);
            const double t0 = now_s(); g_stop_at = t0 + secs;
            for (int i = 0; i < C; i++) { cs_[i].id = i; cs_[i].stride = C; pthread_create(&th[i], NULL, caller_main, &cs_[i]); }
            long calls = 0, failed = 0; double busy = 0;
            for (int i = 0; i < C; i++) { pthread_join(th[i], NULL); calls += cs_[i].calls; failed += cs_[i].failed; busy += cs_[i].busy; }
            const double el = now_s() - t0;
            printf("callers %4d  %9.1f proofs/s  mean latency %7.2f ms  (%ld calls in %.2f s, %ld failed)\n", C, (double)calls / el, calls ? 1e3 * busy / (double)calls : 0.0, calls, el, failed);
            fflush(stdout);
            bad += failed; free(cs_); free(th);
        }
        return bad ? 1 : 0;
    }
    fprintf(stderr, "unknown mode\n");
    return 2;
}
