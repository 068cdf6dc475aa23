// bbx_ctx.hip -- context, workspace and error plumbing of libbbx_hip.so
#include "bbx_common.h"
#include <stdlib.h>

int bbx_hip_fail(bbx_ctx* ctx, hipError_t e, const char* what, int line) {
    if (ctx)
        snprintf(ctx->hip_err, sizeof(ctx->hip_err), "%s: %s (line %d)", what,
                 hipGetErrorString(e), line);
    (void)hipGetLastError();
    return BBX_ERR_HIP;
}

void* bbx_ws(bbx_ctx* ctx, int slot, size_t bytes, int* rc) {
    *rc = BBX_OK;
    if (slot < 0 || slot >= WS_MAX) { *rc = BBX_ERR_ARG; return nullptr; }
    if (ctx->ws_bytes[slot] >= bytes && ctx->d_ws[slot]) return ctx->d_ws[slot];
    if (ctx->d_ws[slot]) {
        // a previous frame may still be using the old block on some stream
        (void)hipDeviceSynchronize();
        (void)hipFree(ctx->d_ws[slot]);
        ctx->d_ws[slot] = nullptr; ctx->ws_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&ctx->d_ws[slot], want);
    if (e != hipSuccess) { *rc = bbx_hip_fail(ctx, e, "hipMalloc(workspace)", __LINE__); return nullptr; }
    ctx->ws_bytes[slot] = want;
    return ctx->d_ws[slot];
}

extern "C" {

int bbx_version(void) { return 100; }

const char* bbx_strerror(int code) {
    switch (code) {
        case BBX_OK: return "ok";
        case BBX_ERR_ARG: return "bad argument or geometry";
        case BBX_ERR_HIP: return "HIP runtime error";
        case BBX_ERR_NOMEM: return "out of memory";
        case BBX_ERR_OVERFLOW: return "device work list overflow";
        case BBX_ERR_NOTCONV: return "device iteration did not converge";
        default: return "unknown error";
    }
}

const char* bbx_last_hip_error(const bbx_ctx* ctx) { return ctx ? ctx->hip_err : ""; }

int bbx_ctx_create(int device, bbx_ctx** out) {
    if (!out) return BBX_ERR_ARG;
    *out = nullptr;
    bbx_ctx* ctx = (bbx_ctx*)calloc(1, sizeof(bbx_ctx));
    if (!ctx) return BBX_ERR_NOMEM;
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_err, 4 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(ctx->d_err, 0, 4 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_counters, CNT_MAX * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(ctx->d_counters, 0, CNT_MAX * sizeof(int32_t));
    if (e != hipSuccess) {
        fprintf(stderr, "bbx_ctx_create: %s\n", hipGetErrorString(e));
        free(ctx);
        return BBX_ERR_HIP;
    }
    *out = ctx;
    return BBX_OK;
}

void bbx_ctx_destroy(bbx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < WS_MAX; i++) if (ctx->d_ws[i]) (void)hipFree(ctx->d_ws[i]);
    if (ctx->d_satlist) (void)hipFree(ctx->d_satlist);
    if (ctx->d_err) (void)hipFree(ctx->d_err);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    free(ctx);
}

int bbx_sync(bbx_ctx* ctx, void* stream) {
    if (!ctx) return BBX_ERR_ARG;
    int32_t err[4] = {0, 0, 0, 0};
    BBX_HIP(hipMemcpyAsync(err, ctx->d_err, sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)stream));
    BBX_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (err[0]) {
        BBX_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(err), (hipStream_t)stream));
        if (err[0] & BBX_DERR_LIST_OVERFLOW) return BBX_ERR_OVERFLOW;
        if (err[0] & BBX_DERR_NOTCONV) return BBX_ERR_NOTCONV;
    }
    return BBX_OK;
}

}  // extern "C"
