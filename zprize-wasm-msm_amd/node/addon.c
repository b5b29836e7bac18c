/*
 * addon.c -- thin N-API binding of the C ABI in include/msm_amd.h (no logic of its own).
 *
 * Host stays Node.js, as with the reference: the reference's callers reach the MSM through
 *   pb.g1m_multiexpAffine(pBases, pScalars, scalarSize, n, pOut)      (wasmcurves/test/bn128.js:493,
 *                                                                      wasmcurves/benchmarks/multiexp.js:37)
 * i.e. a synchronous call on byte buffers.  This addon exposes the same operation on Uint8Array /
 * Buffer arguments, plus an async variant that returns a Promise (ffjavascript's surface is async).
 * index.js in this directory wraps it in the reference's export names.
 *
 * Build (see __graft_entry__.build_node_addon):  gcc -shared -fPIC -I/usr/include/node addon.c -ldl
 * The engine library is dlopen()ed from ../csrc/libmsm_amd.so (or $MSM_AMD_LIB).
 */
#define _GNU_SOURCE
#define NAPI_VERSION 6
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "msm_amd.h"

/* ---- the engine, resolved at load time ---------------------------------------------------- */
static void *g_lib;
static int (*p_create)(int, msm_amd_ctx **);
static void (*p_destroy)(msm_amd_ctx *);
static const char *(*p_last_error)(const msm_amd_ctx *);
static int (*p_set_window_bits)(msm_amd_ctx *, uint32_t);
static int (*p_multiexp)(msm_amd_ctx *, int, int, uint32_t, const void *, const void *, uint32_t, uint32_t, uint8_t *);
static int (*p_multiexp_chunk)(msm_amd_ctx *, int, int, uint32_t, const void *, const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint8_t *);
static int (*p_normalize)(int, int, const uint8_t *, uint8_t *);
static int (*p_to_affine)(int, int, const uint8_t *, uint8_t *);
static int (*p_eq)(int, int, const uint8_t *, const uint8_t *);
static int (*p_combine)(int, int, const uint8_t *, uint32_t, uint8_t *);
static int (*p_f_to_m)(int, const uint8_t *, uint32_t, uint8_t *);
static int (*p_f_from_m)(int, const uint8_t *, uint32_t, uint8_t *);
static int (*p_dev_alloc)(msm_amd_ctx *, size_t, void **);
static int (*p_dev_free)(msm_amd_ctx *, void *);
static int (*p_h2d)(msm_amd_ctx *, void *, const void *, size_t);
static int (*p_fft)(msm_amd_ctx *, int, int, uint32_t, const void *, uint32_t, void *);
static int (*p_apply_key)(msm_amd_ctx *, int, int, uint32_t, const void *, uint32_t, const uint8_t *, const uint8_t *, void *);
static int (*p_fr_apply_key)(msm_amd_ctx *, int, uint32_t, const void *, uint32_t, const uint8_t *, const uint8_t *, void *);
static int (*p_check[2])(msm_amd_ctx *, int, int, uint32_t, const void *, uint32_t, uint8_t *);
static int (*p_conv[6])(msm_amd_ctx *, int, int, uint32_t, const void *, uint32_t, void *);
static int (*p_glv_dec)(const uint8_t *, uint8_t *, uint32_t *);
static int (*p_glv_endo)(const uint8_t *, uint32_t, uint8_t *);
static int (*p_glv_pre)(msm_amd_ctx *, uint32_t, const void *, const void *, uint32_t, void *, void *);
static int (*p_set_option)(msm_amd_ctx *, const char *, uint32_t);
static int (*p_prepare_bases)(msm_amd_ctx *, int, int, uint32_t, const void *, uint32_t, uint32_t, msm_amd_bases **);
static int (*p_release_bases)(msm_amd_ctx *, msm_amd_bases *);
static int (*p_bases_info)(const msm_amd_bases *, uint32_t *, uint64_t *);
static int (*p_plan_prepared)(int, int, uint32_t, uint32_t, uint32_t, uint32_t *, uint64_t *);
static int (*p_multiexp_prepared)(msm_amd_ctx *, msm_amd_bases *, uint32_t, const void *, uint32_t, uint32_t, uint8_t *);
static int (*p_dist_id)(uint8_t *);
static int (*p_dist_init)(msm_amd_ctx *, int, int, const uint8_t *);
static int (*p_dist_destroy)(msm_amd_ctx *);
static int (*p_multiexp_dist)(msm_amd_ctx *, int, int, uint32_t, const void *, const void *, uint32_t, uint32_t, uint8_t *);
static int (*p_multiexp_dist_async)(msm_amd_ctx *, int, int, uint32_t, const void *, const void *, uint32_t, uint32_t, uint64_t *);
static int (*p_dist_wait)(msm_amd_ctx *, int, int, uint64_t, uint8_t *);
static int (*p_dist_comm_count)(const msm_amd_ctx *, int *, int *);

static const char *load_engine(void) {
    if (g_lib) return NULL;
    const char *path = getenv("MSM_AMD_LIB");
    char buf[4096];
    if (!path) {
        Dl_info info;
        if (!dladdr((void *)&load_engine, &info) || !info.dli_fname) return "cannot locate addon";
        snprintf(buf, sizeof buf, "%s", info.dli_fname);
        char *slash = strrchr(buf, '/');
        if (!slash) return "cannot locate addon directory";
        snprintf(slash, sizeof buf - (size_t)(slash - buf), "/../csrc/libmsm_amd.so");
        path = buf;
    }
    g_lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) return dlerror();
#define SYM(var, name)                                  \
    *(void **)(&var) = dlsym(g_lib, name);              \
    if (!var) return "libmsm_amd.so lacks " name;
    SYM(p_create, "msm_amd_create")
    SYM(p_destroy, "msm_amd_destroy")
    SYM(p_last_error, "msm_amd_last_error")
    SYM(p_set_window_bits, "msm_amd_set_window_bits")
    SYM(p_multiexp, "msm_amd_multiexp")
    SYM(p_multiexp_chunk, "msm_amd_multiexp_chunk")
    SYM(p_normalize, "msm_amd_normalize")
    SYM(p_to_affine, "msm_amd_to_affine")
    SYM(p_eq, "msm_amd_eq")
    SYM(p_combine, "msm_amd_combine")
    SYM(p_f_to_m, "msm_amd_f_to_montgomery")
    SYM(p_f_from_m, "msm_amd_f_from_montgomery")
    SYM(p_dev_alloc, "msm_amd_dev_alloc")
    SYM(p_dev_free, "msm_amd_dev_free")
    SYM(p_h2d, "msm_amd_h2d")
    SYM(p_fft, "msm_amd_fft")
    SYM(p_apply_key, "msm_amd_batch_apply_key")
    SYM(p_fr_apply_key, "msm_amd_fr_batch_apply_key")
    SYM(p_check[0], "msm_amd_batch_in_curve_affine")
    SYM(p_check[1], "msm_amd_batch_in_group_affine")
    SYM(p_conv[4], "msm_amd_batch_lem_to_c")
    SYM(p_conv[5], "msm_amd_batch_c_to_lem")
    SYM(p_conv[0], "msm_amd_batch_to_affine")
    SYM(p_conv[1], "msm_amd_batch_to_jacobian")
    SYM(p_conv[2], "msm_amd_batch_lem_to_u")
    SYM(p_conv[3], "msm_amd_batch_u_to_lem")
    SYM(p_glv_dec, "msm_amd_glv_decompose_scalar")
    SYM(p_glv_endo, "msm_amd_glv_endomorphism")
    SYM(p_glv_pre, "msm_amd_glv_preprocess")
    SYM(p_set_option, "msm_amd_set_option")
    SYM(p_prepare_bases, "msm_amd_prepare_bases")
    SYM(p_release_bases, "msm_amd_release_bases")
    SYM(p_bases_info, "msm_amd_bases_info")
    SYM(p_plan_prepared, "msm_amd_plan_prepared")
    SYM(p_multiexp_prepared, "msm_amd_multiexp_prepared")
    SYM(p_dist_id, "msm_amd_dist_unique_id")
    SYM(p_dist_init, "msm_amd_dist_init")
    SYM(p_dist_destroy, "msm_amd_dist_destroy")
    SYM(p_multiexp_dist, "msm_amd_multiexp_dist")
    SYM(p_multiexp_dist_async, "msm_amd_multiexp_dist_async")
    SYM(p_dist_wait, "msm_amd_dist_wait")
    SYM(p_dist_comm_count, "msm_amd_dist_comm_count")
#undef SYM
    return NULL;
}

#define NAPI_OK(call)                                                    \
    do {                                                                 \
        if ((call) != napi_ok) {                                         \
            napi_throw_error(env, NULL, "N-API call failed: " #call);    \
            return NULL;                                                 \
        }                                                                \
    } while (0)

static int n8_of(int curve) { return curve == MSM_AMD_BLS12381 ? 48 : 32; }

static int get_bytes(napi_env env, napi_value v, const uint8_t **data, size_t *len) {
    bool is;
    if (napi_is_buffer(env, v, &is) == napi_ok && is) {
        void *p;
        if (napi_get_buffer_info(env, v, &p, len) != napi_ok) return -1;
        *data = (const uint8_t *)p;
        return 0;
    }
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type t;
        void *p;
        napi_value ab;
        size_t off, count;
        if (napi_get_typedarray_info(env, v, &t, &count, &p, &ab, &off) != napi_ok) return -1;
        if (t != napi_uint8_array) return -1;
        *data = (const uint8_t *)p;
        *len = count;
        return 0;
    }
    return -1;
}
static int get_u32(napi_env env, napi_value v, uint32_t *out) { return napi_get_value_uint32(env, v, out) == napi_ok ? 0 : -1; }
static int get_i32(napi_env env, napi_value v, int *out) {
    int32_t t;
    if (napi_get_value_int32(env, v, &t) != napi_ok) return -1;
    *out = t;
    return 0;
}

static void ctx_finalize(napi_env env, void *data, void *hint) {
    (void)env;
    (void)hint;
    if (data && p_destroy) p_destroy((msm_amd_ctx *)data);
}

/* create(device) -> external */
static napi_value js_create(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    int device = 0;
    if (argc >= 1) get_i32(env, argv[0], &device);
    const char *err = load_engine();
    if (err) {
        napi_throw_error(env, "MSM_AMD_NOLIB", err);
        return NULL;
    }
    msm_amd_ctx *ctx = NULL;
    int rc = p_create(device, &ctx);
    if (rc != 0) {
        char msg[128];
        snprintf(msg, sizeof msg, "msm_amd_create(device=%d) failed with %d: no usable GPU / HIP runtime (there is no CPU fallback)", device, rc);
        napi_throw_error(env, "MSM_AMD_EHIP", msg);
        return NULL;
    }
    napi_value ext;
    NAPI_OK(napi_create_external(env, ctx, ctx_finalize, NULL, &ext));
    return ext;
}

static napi_value js_set_window_bits(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    uint32_t c;
    if (argc < 2 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_u32(env, argv[1], &c)) {
        napi_throw_type_error(env, NULL, "setWindowBits(ctx, c)");
        return NULL;
    }
    if (p_set_window_bits((msm_amd_ctx *)ctx, c)) napi_throw_range_error(env, NULL, "window bits out of range");
    return NULL;
}

/* setOption(ctx, name, value): msm_amd_set_option -- "slots" (calls in flight: promises beyond it wait), "window_bits", ... */
static napi_value js_set_option(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    char name[64];
    size_t len = 0;
    uint32_t v;
    if (argc < 3 || napi_get_value_external(env, argv[0], &ctx) != napi_ok ||
        napi_get_value_string_utf8(env, argv[1], name, sizeof name, &len) != napi_ok || get_u32(env, argv[2], &v)) {
        napi_throw_type_error(env, NULL, "setOption(ctx, name: string, value: uint32)");
        return NULL;
    }
    if (p_set_option((msm_amd_ctx *)ctx, name, v)) napi_throw_range_error(env, NULL, p_last_error((msm_amd_ctx *)ctx));
    return NULL;
}

/* shared argument block of multiexp / multiexpChunk */
typedef struct {
    msm_amd_ctx *ctx;
    int curve, group;
    uint32_t flags, scalar_size, n, start_bit, chunk_size;
    int is_chunk;
    const uint8_t *bases, *scalars;
    uint8_t out[288];
    int rc;
    char err[256];
    /* async only */
    uint64_t dist_ticket; /* is_chunk == 3: submitted on the JS thread, only the wait runs on the worker */
    napi_async_work work;
    napi_deferred deferred;
    napi_ref ref_bases, ref_scalars;
} call_t;

static int parse_call(napi_env env, napi_callback_info info, call_t *c, int is_chunk) {
    size_t argc = 10;
    napi_value argv[10];
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok) return -1;
    size_t need = is_chunk == 1 ? 10 : 8;   /* is_chunk: 0 = multiexp, 1 = _chunk, 2 = the multi-GPU form (msm_amd_multiexp_dist) */
    void *ctx;
    size_t lb, ls;
    if (argc < need || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_i32(env, argv[1], &c->curve) ||
        get_i32(env, argv[2], &c->group) || get_u32(env, argv[3], &c->flags) || get_bytes(env, argv[4], &c->bases, &lb) ||
        get_bytes(env, argv[5], &c->scalars, &ls) || get_u32(env, argv[6], &c->scalar_size) || get_u32(env, argv[7], &c->n)) {
        napi_throw_type_error(env, NULL, "expected (ctx, curve, group, flags, bases:Uint8Array, scalars:Uint8Array, scalarSize, n[, startBit, chunkSize])");
        return -1;
    }
    c->ctx = (msm_amd_ctx *)ctx;
    c->is_chunk = is_chunk;
    if (is_chunk == 1 && (get_u32(env, argv[8], &c->start_bit) || get_u32(env, argv[9], &c->chunk_size))) {
        napi_throw_type_error(env, NULL, "startBit and chunkSize must be unsigned integers");
        return -1;
    }
    if (c->curve != MSM_AMD_BLS12381 && c->curve != MSM_AMD_BN128) {
        napi_throw_range_error(env, NULL, "unknown curve");
        return -1;
    }
    if (c->group != 1 && c->group != 2) {
        napi_throw_range_error(env, NULL, "group must be 1 or 2");
        return -1;
    }
    c->flags &= MSM_AMD_BASES_JACOBIAN; /* device pointers cannot come from JS */
    size_t stride = (size_t)((c->flags & MSM_AMD_BASES_JACOBIAN) ? 3 : 2) * n8_of(c->curve) * c->group;
    if (lb < (size_t)c->n * stride || ls < (size_t)c->n * c->scalar_size) {
        napi_throw_range_error(env, NULL, "bases/scalars buffer shorter than n elements (the reference would trap on out-of-bounds memory)");
        return -1;
    }
    return 0;
}
static void run_call(call_t *c) {
    if (c->is_chunk == 3)
        c->rc = p_dist_wait(c->ctx, c->curve, c->group, c->dist_ticket, c->out);
    else if (c->is_chunk == 2)
        c->rc = p_multiexp_dist(c->ctx, c->curve, c->group, c->flags, c->bases, c->scalars, c->scalar_size, c->n, c->out);
    else if (c->is_chunk)
        c->rc = p_multiexp_chunk(c->ctx, c->curve, c->group, c->flags, c->bases, c->scalars, c->scalar_size, c->n, c->start_bit, c->chunk_size, c->out);
    else
        c->rc = p_multiexp(c->ctx, c->curve, c->group, c->flags, c->bases, c->scalars, c->scalar_size, c->n, c->out);
    if (c->rc) snprintf(c->err, sizeof c->err, "libmsm_amd error %d: %s", c->rc, p_last_error(c->ctx));
}
static napi_value make_result(napi_env env, const call_t *c) {
    size_t jsz = (size_t)3 * n8_of(c->curve) * c->group;
    napi_value ab, arr;
    void *data;
    if (napi_create_arraybuffer(env, jsz, &data, &ab) != napi_ok) return NULL;
    memcpy(data, c->out, jsz);
    if (napi_create_typedarray(env, napi_uint8_array, jsz, ab, 0, &arr) != napi_ok) return NULL;
    return arr;
}

static napi_value js_multiexp_common(napi_env env, napi_callback_info info, int is_chunk) {
    call_t c;
    memset(&c, 0, sizeof c);
    if (parse_call(env, info, &c, is_chunk)) return NULL;
    run_call(&c);
    if (c.rc) {
        napi_throw_error(env, "MSM_AMD_ERROR", c.err);
        return NULL;
    }
    return make_result(env, &c);
}
static napi_value js_multiexp(napi_env env, napi_callback_info info) { return js_multiexp_common(env, info, 0); }
static napi_value js_multiexp_chunk(napi_env env, napi_callback_info info) { return js_multiexp_common(env, info, 1); }

/* async: the GPU call runs on a libuv worker; input buffers are pinned by references meanwhile */
static void async_exec(napi_env env, void *data) {
    (void)env;
    run_call((call_t *)data);
}
static void async_done(napi_env env, napi_status status, void *data) {
    call_t *c = (call_t *)data;
    if (status != napi_ok || c->rc) {
        napi_value msg, err;
        napi_create_string_utf8(env, c->rc ? c->err : "async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, c->deferred, err);
    } else {
        napi_value res = make_result(env, c);
        napi_resolve_deferred(env, c->deferred, res);
    }
    napi_delete_reference(env, c->ref_bases);
    napi_delete_reference(env, c->ref_scalars);
    napi_delete_async_work(env, c->work);
    free(c);
}
static napi_value js_multiexp_async_common(napi_env env, napi_callback_info info, int is_chunk) {
    call_t *c = (call_t *)calloc(1, sizeof *c);
    if (!c) {
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    if (parse_call(env, info, c, is_chunk)) {
        free(c);
        return NULL;
    }
    if (is_chunk == 2) {
        /* A call of a multi-GPU group is SUBMITTED here, on the JS thread, in the order of the JS calls -- the same order on every
         * rank, which is the order the group's exchanges then take (msm_amd_dist_wait).  Submitted from the libuv workers instead,
         * two promises in flight (two slices of one multiExp, or a G1 and a G2 call of a prover) would reach the engine in an order
         * that depends on thread timing and differs between ranks.  Only the wait runs on the worker. */
        int rc = p_multiexp_dist_async(c->ctx, c->curve, c->group, c->flags, c->bases, c->scalars, c->scalar_size, c->n, &c->dist_ticket);
        if (rc) {
            char msg[320];
            snprintf(msg, sizeof msg, "libmsm_amd error %d: %s", rc, p_last_error(c->ctx));
            free(c);
            napi_throw_error(env, "MSM_AMD_ERROR", msg);
            return NULL;
        }
        c->is_chunk = 3;
    }
    size_t argc = 10;
    napi_value argv[10], promise, name;
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    napi_create_reference(env, argv[4], 1, &c->ref_bases);
    napi_create_reference(env, argv[5], 1, &c->ref_scalars);
    napi_create_promise(env, &c->deferred, &promise);
    napi_create_string_utf8(env, "msm_amd_multiexp", NAPI_AUTO_LENGTH, &name);
    napi_create_async_work(env, NULL, name, async_exec, async_done, c, &c->work);
    napi_queue_async_work(env, c->work);
    return promise;
}
static napi_value js_multiexp_async(napi_env env, napi_callback_info info) { return js_multiexp_async_common(env, info, 0); }
/* multi-GPU (one Node process per GPU): this rank's windows + one RCCL all-gather of the partials, behind msm_amd_multiexp_dist */
static napi_value js_multiexp_dist(napi_env env, napi_callback_info info) { return js_multiexp_common(env, info, 2); }
static napi_value js_multiexp_dist_async(napi_env env, napi_callback_info info) { return js_multiexp_async_common(env, info, 2); }
/* distUniqueId() -> Uint8Array(128): rank 0 creates it and hands it to the other processes (IPC, a file, an env var ...) */
static napi_value js_dist_unique_id(napi_env env, napi_callback_info info) {
    (void)info;
    const char *err = load_engine();
    if (err) {
        napi_throw_error(env, "MSM_AMD_NOLIB", err);
        return NULL;
    }
    uint8_t id[128];
    if (p_dist_id(id)) {
        napi_throw_error(env, "MSM_AMD_EHIP", "msm_amd_dist_unique_id failed (RCCL not loadable?)");
        return NULL;
    }
    napi_value ab, arr;
    void *data;
    NAPI_OK(napi_create_arraybuffer(env, sizeof id, &data, &ab));
    memcpy(data, id, sizeof id);
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, sizeof id, ab, 0, &arr));
    return arr;
}
/* distInit(ctx, rank, world, id) -- collective over the `world` processes; distDestroy(ctx) */
static napi_value js_dist_init(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    int rank, world;
    const uint8_t *id;
    size_t len;
    if (argc < 4 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_i32(env, argv[1], &rank) || get_i32(env, argv[2], &world) ||
        get_bytes(env, argv[3], &id, &len) || len < 128) {
        napi_throw_type_error(env, NULL, "distInit(ctx, rank, world, id:Uint8Array(128))");
        return NULL;
    }
    if (p_dist_init((msm_amd_ctx *)ctx, rank, world, id)) {
        char msg[320];
        snprintf(msg, sizeof msg, "msm_amd_dist_init: %s", p_last_error((msm_amd_ctx *)ctx));
        napi_throw_error(env, "MSM_AMD_ERROR", msg);
    }
    return NULL;
}
static napi_value js_dist_destroy(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    if (argc < 1 || napi_get_value_external(env, argv[0], &ctx) != napi_ok) {
        napi_throw_type_error(env, NULL, "distDestroy(ctx)");
        return NULL;
    }
    p_dist_destroy((msm_amd_ctx *)ctx);
    return NULL;
}
/* distCommCount(ctx) -> {ranks, dead}: the ranks the group's communicator itself counts (ncclCommCount) and whether the group has been
 * aborted (an exchange timed out or failed: every call of the group then rejects at once until distDestroy + distInit) */
static napi_value js_dist_comm_count(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    if (argc < 1 || napi_get_value_external(env, argv[0], &ctx) != napi_ok) {
        napi_throw_type_error(env, NULL, "distCommCount(ctx)");
        return NULL;
    }
    int ranks = 0, dead = 0;
    if (p_dist_comm_count((const msm_amd_ctx *)ctx, &ranks, &dead)) {
        napi_throw_error(env, "MSM_AMD_EINVAL", "distCommCount: the context is not in a group");
        return NULL;
    }
    napi_value obj, v;
    NAPI_OK(napi_create_object(env, &obj));
    NAPI_OK(napi_create_int32(env, ranks, &v));
    NAPI_OK(napi_set_named_property(env, obj, "ranks", v));
    NAPI_OK(napi_get_boolean(env, dead != 0, &v));
    NAPI_OK(napi_set_named_property(env, obj, "dead", v));
    return obj;
}
static napi_value js_multiexp_chunk_async(napi_env env, napi_callback_info info) { return js_multiexp_async_common(env, info, 1); }

/* ---- host-only helpers: normalize / toAffine / eq / combine / Montgomery conversion -------- */
static napi_value bytes_out(napi_env env, const uint8_t *src, size_t len) {
    napi_value ab, arr;
    void *data;
    if (napi_create_arraybuffer(env, len, &data, &ab) != napi_ok) return NULL;
    memcpy(data, src, len);
    if (napi_create_typedarray(env, napi_uint8_array, len, ab, 0, &arr) != napi_ok) return NULL;
    return arr;
}
static napi_value js_point_op(napi_env env, napi_callback_info info, int which) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    int curve, group;
    const uint8_t *a, *b = NULL;
    size_t la, lb = 0;
    const char *err = load_engine();
    if (err) {
        napi_throw_error(env, "MSM_AMD_NOLIB", err);
        return NULL;
    }
    if (argc < 3 || get_i32(env, argv[0], &curve) || get_i32(env, argv[1], &group) || get_bytes(env, argv[2], &a, &la) ||
        (which == 2 && (argc < 4 || get_bytes(env, argv[3], &b, &lb)))) {
        napi_throw_type_error(env, NULL, "expected (curve, group, Uint8Array[, Uint8Array])");
        return NULL;
    }
    if ((curve != 0 && curve != 1) || (group != 1 && group != 2)) {
        napi_throw_range_error(env, NULL, "bad curve/group");
        return NULL;
    }
    size_t fsz = (size_t)n8_of(curve) * group, jsz = 3 * fsz;
    uint8_t out[288];
    if (which == 3) { /* combine: any number of Jacobian points */
        if (la % jsz) {
            napi_throw_range_error(env, NULL, "partials must be a whole number of Jacobian points");
            return NULL;
        }
        if (p_combine(curve, group, a, (uint32_t)(la / jsz), out)) {
            napi_throw_error(env, NULL, "msm_amd_combine failed");
            return NULL;
        }
        return bytes_out(env, out, jsz);
    }
    if (la < jsz || (which == 2 && lb < jsz)) {
        napi_throw_range_error(env, NULL, "Jacobian point buffer too short");
        return NULL;
    }
    if (which == 0) {
        p_normalize(curve, group, a, out);
        return bytes_out(env, out, jsz);
    }
    if (which == 1) {
        p_to_affine(curve, group, a, out);
        return bytes_out(env, out, 2 * fsz);
    }
    napi_value res;
    napi_get_boolean(env, p_eq(curve, group, a, b) == 1, &res);
    return res;
}
static napi_value js_normalize(napi_env env, napi_callback_info info) { return js_point_op(env, info, 0); }
static napi_value js_to_affine(napi_env env, napi_callback_info info) { return js_point_op(env, info, 1); }
static napi_value js_eq(napi_env env, napi_callback_info info) { return js_point_op(env, info, 2); }
static napi_value js_combine(napi_env env, napi_callback_info info) { return js_point_op(env, info, 3); }

static napi_value js_mont(napi_env env, napi_callback_info info, int to) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    int curve;
    const uint8_t *a;
    size_t la;
    const char *err = load_engine();
    if (err) {
        napi_throw_error(env, "MSM_AMD_NOLIB", err);
        return NULL;
    }
    if (argc < 2 || get_i32(env, argv[0], &curve) || get_bytes(env, argv[1], &a, &la) || (curve != 0 && curve != 1) || la % (size_t)n8_of(curve)) {
        napi_throw_type_error(env, NULL, "expected (curve, Uint8Array of whole field elements)");
        return NULL;
    }
    uint8_t *out = (uint8_t *)malloc(la ? la : 1);
    if (!out) return NULL;
    (to ? p_f_to_m : p_f_from_m)(curve, a, (uint32_t)(la / (size_t)n8_of(curve)), out);
    napi_value r = bytes_out(env, out, la);
    free(out);
    return r;
}
static napi_value js_to_mont(napi_env env, napi_callback_info info) { return js_mont(env, info, 1); }
static napi_value js_from_mont(napi_env env, napi_callback_info info) { return js_mont(env, info, 0); }

/* ---- resident bases: uploadBases(ctx, Uint8Array) -> handle ; multiexpResident(ctx, curve, group, flags, handle, scalars, scalarSize, n)
 *      A prover's SRS is uploaded once; each call then moves only the scalars over PCIe (MSM_AMD_DEVICE_BASES). -------- */
typedef struct {
    msm_amd_ctx *ctx;
    void *dev;
    size_t bytes;
    napi_ref ref_ctx; /* keeps the context's external alive as long as the buffer */
} bases_t;
static void bases_finalize(napi_env env, void *data, void *hint) {
    (void)hint;
    bases_t *b = (bases_t *)data;
    if (b) {
        if (b->dev && p_dev_free) p_dev_free(b->ctx, b->dev);
        if (b->ref_ctx) napi_delete_reference(env, b->ref_ctx);
        free(b);
    }
}
static napi_value js_upload_bases(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    const uint8_t *p;
    size_t lp;
    if (argc < 2 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_bytes(env, argv[1], &p, &lp) || lp == 0) {
        napi_throw_type_error(env, NULL, "uploadBases(ctx, bases: Uint8Array)");
        return NULL;
    }
    bases_t *b = (bases_t *)calloc(1, sizeof *b);
    if (!b) return NULL;
    b->ctx = (msm_amd_ctx *)ctx;
    b->bytes = lp;
    if (p_dev_alloc(b->ctx, lp, &b->dev) || p_h2d(b->ctx, b->dev, p, lp)) {
        napi_throw_error(env, "MSM_AMD_ERROR", p_last_error(b->ctx));
        if (b->dev) p_dev_free(b->ctx, b->dev);
        free(b);
        return NULL;
    }
    napi_create_reference(env, argv[0], 1, &b->ref_ctx);
    napi_value ext;
    NAPI_OK(napi_create_external(env, b, bases_finalize, NULL, &ext));
    return ext;
}
static napi_value js_multiexp_resident(napi_env env, napi_callback_info info) {
    size_t argc = 8;
    napi_value argv[8];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx, *hb;
    int curve, group;
    uint32_t flags, ss, n;
    const uint8_t *sc;
    size_t ls;
    if (argc < 8 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_i32(env, argv[1], &curve) || get_i32(env, argv[2], &group) ||
        get_u32(env, argv[3], &flags) || napi_get_value_external(env, argv[4], &hb) != napi_ok || get_bytes(env, argv[5], &sc, &ls) ||
        get_u32(env, argv[6], &ss) || get_u32(env, argv[7], &n) || (curve != 0 && curve != 1) || (group != 1 && group != 2)) {
        napi_throw_type_error(env, NULL, "multiexpResident(ctx, curve, group, flags, basesHandle, scalars: Uint8Array, scalarSize, n)");
        return NULL;
    }
    bases_t *b = (bases_t *)hb;
    flags &= MSM_AMD_BASES_JACOBIAN;
    size_t stride = (size_t)((flags & MSM_AMD_BASES_JACOBIAN) ? 3 : 2) * n8_of(curve) * group;
    if (b->bytes < (size_t)n * stride || ls < (size_t)n * ss) {
        napi_throw_range_error(env, NULL, "resident bases / scalars shorter than n elements");
        return NULL;
    }
    call_t c;
    memset(&c, 0, sizeof c);
    c.curve = curve;
    c.group = group;
    int rc = p_multiexp((msm_amd_ctx *)ctx, curve, group, flags | MSM_AMD_DEVICE_BASES, b->dev, sc, ss, n, c.out);
    if (rc) {
        napi_throw_error(env, "MSM_AMD_ERROR", p_last_error((msm_amd_ctx *)ctx));
        return NULL;
    }
    return make_result(env, &c);
}

/* ---- prepared bases (msm_amd_prepare_bases): prepareBases(ctx, curve, group, flags, bases: Uint8Array, n, scalarBits) -> handle ;
 *      multiexpPrepared(ctx, handle, scalars, scalarSize, n) -> Uint8Array ; multiexpPreparedAsync(...) -> Promise ; basesInfo(handle) ; planPrepared(curve, group, n[, scalarBits[, windowBits]]).
 *      The window table of a prover's fixed points is built on the GPU once; every call then sorts all windows into one bucket set. */
typedef struct {
    msm_amd_ctx *ctx;
    msm_amd_bases *h;
    int curve, group;
    uint32_t n;
    napi_ref ref_ctx; /* the context's external: it must outlive the handle (finalisers run in no particular order) */
} prepared_t;
static void prepared_finalize(napi_env env, void *data, void *hint) {
    (void)hint;
    prepared_t *b = (prepared_t *)data;
    if (b) {
        if (b->h && p_release_bases) p_release_bases(b->ctx, b->h);
        if (b->ref_ctx) napi_delete_reference(env, b->ref_ctx);
        free(b);
    }
}
static napi_value js_prepare_bases(napi_env env, napi_callback_info info) {
    size_t argc = 7;
    napi_value argv[7];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    int curve, group;
    uint32_t flags, n, bits = 0;
    const uint8_t *p;
    size_t lp;
    if (argc < 6 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_i32(env, argv[1], &curve) || get_i32(env, argv[2], &group) ||
        get_u32(env, argv[3], &flags) || get_bytes(env, argv[4], &p, &lp) || get_u32(env, argv[5], &n) || (argc >= 7 && get_u32(env, argv[6], &bits)) ||
        (curve != 0 && curve != 1) || (group != 1 && group != 2)) {
        napi_throw_type_error(env, NULL, "prepareBases(ctx, curve, group, flags, bases: Uint8Array, n[, scalarBits])");
        return NULL;
    }
    flags &= MSM_AMD_BASES_JACOBIAN;
    if (lp < (size_t)n * ((flags & MSM_AMD_BASES_JACOBIAN) ? 3 : 2) * n8_of(curve) * group) {
        napi_throw_range_error(env, NULL, "bases buffer shorter than n points");
        return NULL;
    }
    prepared_t *b = (prepared_t *)calloc(1, sizeof *b);
    if (!b) return NULL;
    b->ctx = (msm_amd_ctx *)ctx;
    b->curve = curve;
    b->group = group;
    b->n = n;
    if (p_prepare_bases(b->ctx, curve, group, flags, p, n, bits, &b->h)) {
        napi_throw_error(env, "MSM_AMD_ERROR", p_last_error(b->ctx));
        free(b);
        return NULL;
    }
    napi_create_reference(env, argv[0], 1, &b->ref_ctx);
    napi_value ext;
    NAPI_OK(napi_create_external(env, b, prepared_finalize, NULL, &ext));
    return ext;
}
static napi_value bases_info_object(napi_env env, const uint32_t *v6, uint64_t tb);
static napi_value js_bases_info(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *hb;
    if (argc < 1 || napi_get_value_external(env, argv[0], &hb) != napi_ok) {
        napi_throw_type_error(env, NULL, "basesInfo(handle)");
        return NULL;
    }
    uint32_t v6[6];
    uint64_t tb = 0;
    p_bases_info(((prepared_t *)hb)->h, v6, &tb);
    return bases_info_object(env, v6, tb);
}
static napi_value bases_info_object(napi_env env, const uint32_t *v6, uint64_t tb) {
    static const char *names[6] = {"n", "scalarBits", "c", "windows", "signedWindows", "buckets"};
    napi_value obj, v;
    NAPI_OK(napi_create_object(env, &obj));
    for (int i = 0; i < 6; ++i) {
        NAPI_OK(napi_create_uint32(env, v6[i], &v));
        NAPI_OK(napi_set_named_property(env, obj, names[i], v));
    }
    NAPI_OK(napi_create_double(env, (double)tb, &v));
    NAPI_OK(napi_set_named_property(env, obj, "tableBytes", v));
    return obj;
}
/* planPrepared(curve, group, n[, scalarBits[, windowBits]]) -> what basesInfo would say of the table prepareBases would build (no GPU) */
static napi_value js_plan_prepared(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    int curve, group;
    uint32_t n, bits = 0, wb = 0;
    const char *err = load_engine();
    if (err) {
        napi_throw_error(env, "MSM_AMD_NOLIB", err);
        return NULL;
    }
    if (argc < 3 || get_i32(env, argv[0], &curve) || get_i32(env, argv[1], &group) || get_u32(env, argv[2], &n) ||
        (argc >= 4 && get_u32(env, argv[3], &bits)) || (argc >= 5 && get_u32(env, argv[4], &wb))) {
        napi_throw_type_error(env, NULL, "planPrepared(curve, group, n[, scalarBits[, windowBits]])");
        return NULL;
    }
    uint32_t v6[6];
    uint64_t tb = 0;
    if (p_plan_prepared(curve, group, n, bits, wb, v6, &tb)) {
        napi_throw_range_error(env, NULL, "planPrepared: bad curve / group / n / scalarBits, or windows x points of the table >= 2^31");
        return NULL;
    }
    return bases_info_object(env, v6, tb);
}
typedef struct {
    prepared_t *b;
    uint32_t scalar_size, n;
    const uint8_t *scalars;
    uint8_t out[288];
    int rc;
    char err[256];
    napi_async_work work;
    napi_deferred deferred;
    napi_ref ref_scalars, ref_handle;
} pcall_t;
static int parse_pcall(napi_env env, napi_callback_info info, pcall_t *c, napi_value *argv_out) {
    size_t argc = 5;
    napi_value argv[5];
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok) return -1;
    void *ctx, *hb;
    size_t ls;
    if (argc < 5 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || napi_get_value_external(env, argv[1], &hb) != napi_ok ||
        get_bytes(env, argv[2], &c->scalars, &ls) || get_u32(env, argv[3], &c->scalar_size) || get_u32(env, argv[4], &c->n)) {
        napi_throw_type_error(env, NULL, "multiexpPrepared(ctx, basesHandle, scalars: Uint8Array, scalarSize, n)");
        return -1;
    }
    c->b = (prepared_t *)hb;
    if (c->b->ctx != (msm_amd_ctx *)ctx || ls < (size_t)c->n * c->scalar_size) {
        napi_throw_range_error(env, NULL, "handle of another context, or scalars shorter than n elements");
        return -1;
    }
    if (argv_out) {
        argv_out[0] = argv[1];
        argv_out[1] = argv[2];
    }
    return 0;
}
static napi_value presult(napi_env env, const pcall_t *c) {
    size_t jsz = (size_t)3 * n8_of(c->b->curve) * c->b->group;
    napi_value ab, arr;
    void *data;
    if (napi_create_arraybuffer(env, jsz, &data, &ab) != napi_ok) return NULL;
    memcpy(data, c->out, jsz);
    if (napi_create_typedarray(env, napi_uint8_array, jsz, ab, 0, &arr) != napi_ok) return NULL;
    return arr;
}
static napi_value js_multiexp_prepared(napi_env env, napi_callback_info info) {
    pcall_t c;
    memset(&c, 0, sizeof c);
    if (parse_pcall(env, info, &c, NULL)) return NULL;
    if (p_multiexp_prepared(c.b->ctx, c.b->h, 0, c.scalars, c.scalar_size, c.n, c.out)) {
        napi_throw_error(env, "MSM_AMD_ERROR", p_last_error(c.b->ctx));
        return NULL;
    }
    return presult(env, &c);
}
static void pcall_exec(napi_env env, void *data) {
    (void)env;
    pcall_t *c = (pcall_t *)data;
    c->rc = p_multiexp_prepared(c->b->ctx, c->b->h, 0, c->scalars, c->scalar_size, c->n, c->out);
    if (c->rc) snprintf(c->err, sizeof c->err, "libmsm_amd error %d: %s", c->rc, p_last_error(c->b->ctx));
}
static void pcall_done(napi_env env, napi_status status, void *data) {
    pcall_t *c = (pcall_t *)data;
    if (status != napi_ok || c->rc) {
        napi_value msg, err;
        napi_create_string_utf8(env, c->rc ? c->err : "async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, c->deferred, err);
    } else {
        napi_resolve_deferred(env, c->deferred, presult(env, c));
    }
    napi_delete_reference(env, c->ref_scalars);
    napi_delete_reference(env, c->ref_handle);   /* (the handle cannot be finalised, and its table freed, while a call uses it) */
    napi_delete_async_work(env, c->work);
    free(c);
}
static napi_value js_multiexp_prepared_async(napi_env env, napi_callback_info info) {
    pcall_t *c = (pcall_t *)calloc(1, sizeof *c);
    if (!c) return NULL;
    napi_value keep[2], promise, name;
    if (parse_pcall(env, info, c, keep)) {
        free(c);
        return NULL;
    }
    napi_create_reference(env, keep[0], 1, &c->ref_handle);
    napi_create_reference(env, keep[1], 1, &c->ref_scalars);
    napi_create_promise(env, &c->deferred, &promise);
    napi_create_string_utf8(env, "msm_amd_multiexp_prepared", NAPI_AUTO_LENGTH, &name);
    napi_create_async_work(env, NULL, name, pcall_exec, pcall_done, c, &c->work);
    napi_queue_async_work(env, c->work);
    return promise;
}

/* ---- batch conversions: batchConvert(ctx, which, curve, group, bytes) -> bytes
 *      which: 0 batchToAffine, 1 batchToJacobian, 2 batchLEMtoU, 3 batchUtoLEM ---------------------------- */
static napi_value js_batch_convert(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    int which, curve, group;
    const uint8_t *in;
    size_t lin;
    if (argc < 5 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_i32(env, argv[1], &which) || get_i32(env, argv[2], &curve) ||
        get_i32(env, argv[3], &group) || get_bytes(env, argv[4], &in, &lin) || which < 0 || which > 5 || (curve != 0 && curve != 1) ||
        (group != 1 && group != 2)) {
        napi_throw_type_error(env, NULL, "batchConvert(ctx, which 0..5, curve, group, Uint8Array)");
        return NULL;
    }
    size_t fsz = (size_t)n8_of(curve) * group;
    size_t in_sz = (which == 0 ? 3 : which == 5 ? 1 : 2) * fsz, out_sz = (which == 1 ? 3 : which == 4 ? 1 : 2) * fsz;   /* 4: LEMtoC, 5: CtoLEM */
    if (lin % in_sz) {
        napi_throw_range_error(env, NULL, "input is not a whole number of points");
        return NULL;
    }
    uint32_t n = (uint32_t)(lin / in_sz);
    uint8_t *out = (uint8_t *)malloc((size_t)n * out_sz + 1);
    if (!out) return NULL;
    int rc = p_conv[which]((msm_amd_ctx *)ctx, curve, group, 0, in, n, out);
    napi_value res = NULL;
    if (rc) napi_throw_error(env, "MSM_AMD_ERROR", p_last_error((msm_amd_ctx *)ctx));
    else res = bytes_out(env, out, (size_t)n * out_sz);
    free(out);
    return res;
}

/* ---- fft(ctx, curve, group 0 = Fr | 1 | 2, inverse 0/1, Uint8Array data) -> Uint8Array  (build_fft.js:178-245) ---- */
static napi_value js_fft(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    int curve, group, inverse;
    const uint8_t *in;
    size_t lin;
    if (argc < 5 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_i32(env, argv[1], &curve) || get_i32(env, argv[2], &group) ||
        get_i32(env, argv[3], &inverse) || get_bytes(env, argv[4], &in, &lin) || (curve != 0 && curve != 1) || group < 0 || group > 2) {
        napi_throw_type_error(env, NULL, "fft(ctx, curve, group 0..2, inverse, Uint8Array)");
        return NULL;
    }
    size_t esz = group == 0 ? 32 : (size_t)3 * n8_of(curve) * group;
    if (lin % esz) {
        napi_throw_range_error(env, NULL, "input is not a whole number of elements");
        return NULL;
    }
    uint32_t n = (uint32_t)(lin / esz);
    uint8_t *out = (uint8_t *)malloc((size_t)n * esz + 1);
    if (!out) return NULL;
    int rc = p_fft((msm_amd_ctx *)ctx, curve, group, inverse ? MSM_AMD_FFT_INVERSE : 0, in, n, out);
    napi_value res = NULL;
    if (rc) napi_throw_error(env, "MSM_AMD_ERROR", p_last_error((msm_amd_ctx *)ctx));
    else res = bytes_out(env, out, (size_t)n * esz);
    free(out);
    return res;
}

/* ---- batchApplyKey(ctx, curve, group 0 = Fr | 1 | 2, jacobianIn 0/1, Uint8Array in, Uint8Array first(32), Uint8Array inc(32))
 *      -> Uint8Array: n normalised Jacobian points (group 1/2) or n Fr elements (group 0)  (build_applykey.js:20-76) ---- */
static napi_value js_batch_apply_key(napi_env env, napi_callback_info info) {
    size_t argc = 7;
    napi_value argv[7];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    int curve, group, jac;
    const uint8_t *in, *first, *inc;
    size_t lin, lf, li;
    if (argc < 7 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_i32(env, argv[1], &curve) || get_i32(env, argv[2], &group) ||
        get_i32(env, argv[3], &jac) || get_bytes(env, argv[4], &in, &lin) || get_bytes(env, argv[5], &first, &lf) || get_bytes(env, argv[6], &inc, &li) ||
        (curve != 0 && curve != 1) || group < 0 || group > 2 || lf < 32 || li < 32) {
        napi_throw_type_error(env, NULL, "batchApplyKey(ctx, curve, group 0..2, jacobianIn, Uint8Array in, first: Uint8Array(32), inc: Uint8Array(32))");
        return NULL;
    }
    size_t fsz = (size_t)n8_of(curve) * (group ? group : 1);
    size_t in_sz = group == 0 ? 32 : (jac ? 3 : 2) * fsz, out_sz = group == 0 ? 32 : 3 * fsz;
    if (lin % in_sz) {
        napi_throw_range_error(env, NULL, "input is not a whole number of elements");
        return NULL;
    }
    uint32_t n = (uint32_t)(lin / in_sz);
    uint8_t *out = (uint8_t *)malloc((size_t)n * out_sz + 1);
    if (!out) return NULL;
    int rc = group == 0 ? p_fr_apply_key((msm_amd_ctx *)ctx, curve, 0, in, n, first, inc, out)
                        : p_apply_key((msm_amd_ctx *)ctx, curve, group, jac ? MSM_AMD_BASES_JACOBIAN : 0, in, n, first, inc, out);
    napi_value res = NULL;
    if (rc) napi_throw_error(env, "MSM_AMD_ERROR", p_last_error((msm_amd_ctx *)ctx));
    else res = bytes_out(env, out, (size_t)n * out_sz);
    free(out);
    return res;
}

/* ---- point validation: batchCheck(ctx, which 0 = inCurveAffine | 1 = inGroupAffine, curve, group, Uint8Array affine points)
 *      -> Uint8Array of 0/1 flags (g?m_inCurveAffine, build_curve_jacobian_a0.js:1330-1353; g?m_inGroupAffine, bls12381 only) ---- */
static napi_value js_batch_check(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    int which, curve, group;
    const uint8_t *in;
    size_t lin;
    if (argc < 5 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_i32(env, argv[1], &which) || get_i32(env, argv[2], &curve) ||
        get_i32(env, argv[3], &group) || get_bytes(env, argv[4], &in, &lin) || which < 0 || which > 1 || (curve != 0 && curve != 1) ||
        (group != 1 && group != 2)) {
        napi_throw_type_error(env, NULL, "batchCheck(ctx, which 0..1, curve, group, Uint8Array)");
        return NULL;
    }
    size_t asz = (size_t)2 * n8_of(curve) * group;
    if (lin % asz) {
        napi_throw_range_error(env, NULL, "input is not a whole number of affine points");
        return NULL;
    }
    uint32_t n = (uint32_t)(lin / asz);
    uint8_t *out = (uint8_t *)malloc((size_t)n + 1);
    if (!out) return NULL;
    int rc = p_check[which]((msm_amd_ctx *)ctx, curve, group, 0, in, n, out);
    napi_value res = NULL;
    if (rc) napi_throw_error(env, "MSM_AMD_ERROR", p_last_error((msm_amd_ctx *)ctx));
    else res = bytes_out(env, out, (size_t)n);
    free(out);
    return res;
}

/* ---- GLV pre-pass (bls12381 G1): g1m_glv_decomposeScalar / _endomorphism / _preprocessEndomorphism ---- */
static napi_value js_glv_decompose(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    const uint8_t *k;
    size_t lk;
    const char *err = load_engine();
    if (err) {
        napi_throw_error(env, "MSM_AMD_NOLIB", err);
        return NULL;
    }
    if (argc < 1 || get_bytes(env, argv[0], &k, &lk) || lk < 32) {
        napi_throw_type_error(env, NULL, "glvDecomposeScalar(scalar: Uint8Array(32))");
        return NULL;
    }
    uint8_t out[64];
    uint32_t sign = 0;
    p_glv_dec(k, out, &sign);
    napi_value res, jsign, jbytes;
    NAPI_OK(napi_create_object(env, &res));
    NAPI_OK(napi_create_uint32(env, sign, &jsign));
    jbytes = bytes_out(env, out, 64);
    NAPI_OK(napi_set_named_property(env, res, "sign", jsign));
    NAPI_OK(napi_set_named_property(env, res, "scalars", jbytes));
    return res;
}
static napi_value js_glv_endomorphism(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    const uint8_t *p;
    size_t lp;
    uint32_t sign;
    const char *err = load_engine();
    if (err) {
        napi_throw_error(env, "MSM_AMD_NOLIB", err);
        return NULL;
    }
    if (argc < 2 || get_bytes(env, argv[0], &p, &lp) || lp < 96 || get_u32(env, argv[1], &sign)) {
        napi_throw_type_error(env, NULL, "glvEndomorphism(point: Uint8Array(96), sign)");
        return NULL;
    }
    uint8_t out[96];
    p_glv_endo(p, sign, out);
    return bytes_out(env, out, 96);
}
static napi_value js_glv_preprocess(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *ctx;
    const uint8_t *pts, *sc;
    size_t lp, ls;
    uint32_t n;
    if (argc < 4 || napi_get_value_external(env, argv[0], &ctx) != napi_ok || get_bytes(env, argv[1], &pts, &lp) ||
        get_bytes(env, argv[2], &sc, &ls) || get_u32(env, argv[3], &n) || lp < (size_t)n * 96 || ls < (size_t)n * 32) {
        napi_throw_type_error(env, NULL, "glvPreprocess(ctx, points: Uint8Array(96 n), scalars: Uint8Array(32 n), n)");
        return NULL;
    }
    uint8_t *po = (uint8_t *)malloc((size_t)n * 192 + 1), *so = (uint8_t *)malloc((size_t)n * 64 + 1);
    if (!po || !so) {
        free(po);
        free(so);
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    int rc = p_glv_pre((msm_amd_ctx *)ctx, 0, pts, sc, n, po, so);
    napi_value res = NULL;
    if (rc) {
        napi_throw_error(env, "MSM_AMD_ERROR", p_last_error((msm_amd_ctx *)ctx));
    } else {
        napi_value jp = bytes_out(env, po, (size_t)n * 192), js = bytes_out(env, so, (size_t)n * 64);
        if (napi_create_object(env, &res) == napi_ok) {
            napi_set_named_property(env, res, "points", jp);
            napi_set_named_property(env, res, "scalars", js);
        }
    }
    free(po);
    free(so);
    return res;
}

static napi_value init(napi_env env, napi_value exports) {
    napi_property_descriptor props[] = {
        {"create", NULL, js_create, NULL, NULL, NULL, napi_enumerable, NULL},
        {"setWindowBits", NULL, js_set_window_bits, NULL, NULL, NULL, napi_enumerable, NULL},
        {"multiexp", NULL, js_multiexp, NULL, NULL, NULL, napi_enumerable, NULL},
        {"multiexpChunk", NULL, js_multiexp_chunk, NULL, NULL, NULL, napi_enumerable, NULL},
        {"multiexpAsync", NULL, js_multiexp_async, NULL, NULL, NULL, napi_enumerable, NULL},
        {"multiexpChunkAsync", NULL, js_multiexp_chunk_async, NULL, NULL, NULL, napi_enumerable, NULL},
        {"multiexpDist", NULL, js_multiexp_dist, NULL, NULL, NULL, napi_enumerable, NULL},
        {"multiexpDistAsync", NULL, js_multiexp_dist_async, NULL, NULL, NULL, napi_enumerable, NULL},
        {"distUniqueId", NULL, js_dist_unique_id, NULL, NULL, NULL, napi_enumerable, NULL},
        {"distInit", NULL, js_dist_init, NULL, NULL, NULL, napi_enumerable, NULL},
        {"distDestroy", NULL, js_dist_destroy, NULL, NULL, NULL, napi_enumerable, NULL},
        {"distCommCount", NULL, js_dist_comm_count, NULL, NULL, NULL, napi_enumerable, NULL},
        {"normalize", NULL, js_normalize, NULL, NULL, NULL, napi_enumerable, NULL},
        {"toAffine", NULL, js_to_affine, NULL, NULL, NULL, napi_enumerable, NULL},
        {"eq", NULL, js_eq, NULL, NULL, NULL, napi_enumerable, NULL},
        {"combine", NULL, js_combine, NULL, NULL, NULL, napi_enumerable, NULL},
        {"toMontgomery", NULL, js_to_mont, NULL, NULL, NULL, napi_enumerable, NULL},
        {"fromMontgomery", NULL, js_from_mont, NULL, NULL, NULL, napi_enumerable, NULL},
        {"setOption", NULL, js_set_option, NULL, NULL, NULL, napi_enumerable, NULL},
        {"prepareBases", NULL, js_prepare_bases, NULL, NULL, NULL, napi_enumerable, NULL},
        {"basesInfo", NULL, js_bases_info, NULL, NULL, NULL, napi_enumerable, NULL},
        {"planPrepared", NULL, js_plan_prepared, NULL, NULL, NULL, napi_enumerable, NULL},
        {"multiexpPrepared", NULL, js_multiexp_prepared, NULL, NULL, NULL, napi_enumerable, NULL},
        {"multiexpPreparedAsync", NULL, js_multiexp_prepared_async, NULL, NULL, NULL, napi_enumerable, NULL},
        {"uploadBases", NULL, js_upload_bases, NULL, NULL, NULL, napi_enumerable, NULL},
        {"multiexpResident", NULL, js_multiexp_resident, NULL, NULL, NULL, napi_enumerable, NULL},
        {"batchConvert", NULL, js_batch_convert, NULL, NULL, NULL, napi_enumerable, NULL},
        {"batchCheck", NULL, js_batch_check, NULL, NULL, NULL, napi_enumerable, NULL},
        {"batchApplyKey", NULL, js_batch_apply_key, NULL, NULL, NULL, napi_enumerable, NULL},
        {"fft", NULL, js_fft, NULL, NULL, NULL, napi_enumerable, NULL},
        {"glvDecomposeScalar", NULL, js_glv_decompose, NULL, NULL, NULL, napi_enumerable, NULL},
        {"glvEndomorphism", NULL, js_glv_endomorphism, NULL, NULL, NULL, napi_enumerable, NULL},
        {"glvPreprocess", NULL, js_glv_preprocess, NULL, NULL, NULL, napi_enumerable, NULL},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
