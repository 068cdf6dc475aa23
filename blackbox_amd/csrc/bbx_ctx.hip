// bbx_ctx.hip -- context, workspace and error plumbing of libbbx_hip.so
#include "bbx_common.h"
#include <stdlib.h>
#include <time.h>

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

// bbx_profile_enable may be called by another thread than the one launching on this context (a
// bench switches the timers on at the start of its timed region): a pair is kept only if no
// enable / read came between its start and its stop (generation counter)
void bbx_prof_start(bbx_ctx* ctx, int slot, hipStream_t s) {
    ctx->prof_open = 0;
    if (!ctx->prof_on || ctx->prof_n >= BBX_PROF_MAX) return;
    const int n = ctx->prof_n;
    ctx->prof_open_gen = ctx->prof_gen;
    ctx->prof_slot[n] = slot;
    (void)hipEventRecord(ctx->prof_ev[2 * n], s);
    ctx->prof_open = 1; ctx->prof_open_idx = n;
}
void bbx_prof_events(bbx_ctx* ctx, int slot, hipEvent_t* e0, hipEvent_t* e1) {
    *e0 = *e1 = nullptr;
    ctx->prof_open = 0;
    if (!ctx->prof_on || ctx->prof_n >= BBX_PROF_MAX) return;
    const int n = ctx->prof_n;
    ctx->prof_slot[n] = slot;
    *e0 = ctx->prof_ev[2 * n]; *e1 = ctx->prof_ev[2 * n + 1];
    ctx->prof_n = n + 1;
}
void bbx_prof_stop(bbx_ctx* ctx, hipStream_t s) {
    if (!ctx->prof_open) return;
    ctx->prof_open = 0;
    if (!ctx->prof_on || ctx->prof_open_gen != ctx->prof_gen || ctx->prof_open_idx != ctx->prof_n) return;
    (void)hipEventRecord(ctx->prof_ev[2 * ctx->prof_n + 1], s);
    ctx->prof_n++;
}

#define BBX_HIP0(call) do { if ((call) != hipSuccess) return BBX_ERR_HIP; } while (0)

extern "C" {

int bbx_version(void) { return 100; }
int bbx_build_flags(void) { return bbx_build_flags_fpack() | bbx_build_flags_zogy() | bbx_build_flags_bkg() | bbx_build_flags_sat() | bbx_build_flags_canny(); }

int bbx_profile_enable(bbx_ctx* ctx, int on) {
    if (!ctx) return BBX_ERR_ARG;
    if (on && !ctx->prof_ev) {
        ctx->prof_ev = (hipEvent_t*)calloc(2 * BBX_PROF_MAX, sizeof(hipEvent_t));
        ctx->prof_slot = (int*)calloc(BBX_PROF_MAX, sizeof(int));
        if (!ctx->prof_ev || !ctx->prof_slot) return BBX_ERR_NOMEM;
        for (int i = 0; i < 2 * BBX_PROF_MAX; i++) BBX_HIP(hipEventCreate(&ctx->prof_ev[i]));
    }
    ctx->prof_on = 0;
    ctx->prof_gen++;
    if (on == 2) return BBX_OK;          // pause: stop recording, keep what was recorded for bbx_profile_read
    ctx->prof_n = 0;
    ctx->prof_on = on ? 1 : 0;
    return BBX_OK;
}

int bbx_profile_read(bbx_ctx* ctx, double* ms_total, int32_t* calls, int nslots) {
    if (!ctx || !ms_total || !calls || nslots <= 0) return BBX_ERR_ARG;
    for (int i = 0; i < nslots; i++) { ms_total[i] = 0.0; calls[i] = 0; }
    for (int k = 0; k < ctx->prof_n; k++) {
        float ms = 0.f;
        // a pair whose launch never happened (profiling switched while a lane was between reserve and launch) is skipped
        if (hipEventSynchronize(ctx->prof_ev[2 * k + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, ctx->prof_ev[2 * k], ctx->prof_ev[2 * k + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }
        const int sl = ctx->prof_slot[k];
        if (sl >= 0 && sl < nslots) { ms_total[sl] += ms; calls[sl]++; }
    }
    ctx->prof_gen++;
    ctx->prof_n = 0;
    return BBX_OK;
}

const char* bbx_strerror(int code) {
    switch (code) {
        case BBX_OK: return "ok";
        case BBX_ERR_ARG: return "bad argument or geometry";
        case BBX_ERR_HIP: return "HIP runtime error";
        case BBX_ERR_NOMEM: return "out of memory";
        case BBX_ERR_OVERFLOW: return "device work list overflow";
        case BBX_ERR_NOTCONV: return "device iteration did not converge";
        case BBX_ERR_PSFWIN: return "PSF matched-filter kernels exceed their row window";
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
    ctx->num_cus = 0;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ctx->num_cus, hipDeviceAttributeMultiprocessorCount, device);
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
    bbx_zogy_release(ctx);
    bbx_zogy2_release(ctx);
    bbx_fpack_release(ctx);
    for (int i = 0; i < WS_MAX; i++) if (ctx->d_ws[i]) (void)hipFree(ctx->d_ws[i]);
    if (ctx->d_satlist) (void)hipFree(ctx->d_satlist);
    if (ctx->d_nonlin) (void)hipFree(ctx->d_nonlin);
    if (ctx->d_err) (void)hipFree(ctx->d_err);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->prof_ev) { for (int i = 0; i < 2 * BBX_PROF_MAX; i++) (void)hipEventDestroy(ctx->prof_ev[i]); free(ctx->prof_ev); free(ctx->prof_slot); }
    free(ctx);
}

__global__ __launch_bounds__(256) void k_copy_bytes(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n, int words) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    if (words) {
        const size_t nw = n >> 2;
        for (size_t i = i0; i < nw; i += step) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
        for (size_t i = (nw << 2) + i0; i < n; i += step) dst[i] = src[i];
    } else {
        for (size_t i = i0; i < n; i += step) dst[i] = src[i];
    }
}

int bbx_copy_kernel(void* dst, const void* src, size_t nbytes, void* stream) {
    if (!dst || !src) return BBX_ERR_ARG;
    if (nbytes == 0) return BBX_OK;
    const int words = (((uintptr_t)dst | (uintptr_t)src) & 3) == 0;
    const size_t items = words ? (nbytes >> 2) + 3 : nbytes;
    const int blocks = (int)((items + 255) / 256 < 1024 ? (items + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_copy_bytes, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (uint8_t*)dst, nbytes, words);
    if (hipGetLastError() != hipSuccess) return BBX_ERR_HIP;
    return BBX_OK;
}

// poll + sleep until [ev] has completed
static int bbx_poll_event(hipEvent_t ev, int sleep_us) {
    struct timespec ts; ts.tv_sec = 0; ts.tv_nsec = (long)sleep_us * 1000L;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return BBX_OK;
        if (e != hipErrorNotReady) return BBX_ERR_HIP;
        (void)hipGetLastError();
        nanosleep(&ts, nullptr);
    }
}

int bbx_event_wait(void* event, int sleep_us) {
    if (!event) return BBX_ERR_ARG;
    if (sleep_us <= 0) { BBX_HIP0(hipEventSynchronize((hipEvent_t)event)); return BBX_OK; }
    return bbx_poll_event((hipEvent_t)event, sleep_us);
}

// The state of a host wait belongs to the waiting THREAD, not to the context: several threads wait on one context (a lane
// thread inside _lib.fetch while the orchestrating thread calls bbx_sync), and a second hipEventRecord on a shared event
// would let the other thread's poll return before ITS work has finished (round 4 kept one event and one pinned error word
// per context).  One event per thread and device, one pinned error buffer per thread; both live as long as the thread
// (a few bytes per pool thread, never freed: HIP may be gone by the time a thread-local destructor runs).
struct wait_tls { hipEvent_t ev[16]; int32_t* h_err; };
static thread_local wait_tls g_wait;

int bbx_wait(bbx_ctx* ctx, void* stream) {
    if (!ctx) return BBX_ERR_ARG;
    if (ctx->wait_sleep_us <= 0) { BBX_HIP(hipStreamSynchronize((hipStream_t)stream)); return BBX_OK; }
    hipEvent_t& ev = g_wait.ev[ctx->device & 15];
    if (!ev) BBX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    BBX_HIP(hipEventRecord(ev, (hipStream_t)stream));
    const int rc = bbx_poll_event(ev, ctx->wait_sleep_us);
    if (rc) return bbx_hip_fail(ctx, hipGetLastError(), "hipEventQuery", __LINE__);
    return BBX_OK;
}

int bbx_sync(bbx_ctx* ctx, void* stream) {
    if (!ctx) return BBX_ERR_ARG;
    if (!g_wait.h_err) BBX_HIP(hipHostMalloc((void**)&g_wait.h_err, 4 * sizeof(int32_t), hipHostMallocDefault));
    int32_t* err = g_wait.h_err;
    err[0] = 0;
    BBX_HIP(hipMemcpyAsync(err, ctx->d_err, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    { const int rc = bbx_wait(ctx, stream); if (rc) return rc; }
    if (err[0]) {
        BBX_HIP(hipMemsetAsync(ctx->d_err, 0, 4 * sizeof(int32_t), (hipStream_t)stream));
        // (the window check first: the caller repeats that call on all rows, and a list overflow of the same call shows again then)
        if (err[0] & BBX_DERR_PSF_WINDOW) return BBX_ERR_PSFWIN;
        if (err[0] & BBX_DERR_LIST_OVERFLOW) return BBX_ERR_OVERFLOW;
        if (err[0] & BBX_DERR_NOTCONV) return BBX_ERR_NOTCONV;
    }
    return BBX_OK;
}

// *slot = device error flags accumulated since the last mark; flags <- 0 (one thread)
__global__ void k_step_mark(int32_t* err, int32_t* slot) { *slot = err[0]; err[0] = 0; }

int bbx_step_mark(bbx_ctx* ctx, int32_t* d_slot, void* stream) {
    if (!ctx || !d_slot) return BBX_ERR_ARG;
    hipLaunchKernelGGL(k_step_mark, dim3(1), dim3(1), 0, (hipStream_t)stream, ctx->d_err, d_slot);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_set_option(bbx_ctx* ctx, int option, int value) {
    if (!ctx) return BBX_ERR_ARG;
    if (option == BBX_OPT_LAC_LEVEL_FEED) { ctx->lac_feed = value ? 1 : 0; return BBX_OK; }
    if (option == BBX_OPT_ZOGY_CORE) return BBX_OK;               // (one transform core since round 3: accepted, no effect)
    if (option == BBX_OPT_DEBUG_LISTCAP) { ctx->debug_listcap = value > 0 ? value : 0; return BBX_OK; }
    if (option == BBX_OPT_ZOGY_KWIN_OFF) { ctx->zogy_kwin_off = value ? 1 : 0; return BBX_OK; }
    if (option == BBX_OPT_FPACK_ONE_WG) { ctx->fpack_one_wg = value ? 1 : 0; return BBX_OK; }
    if (option == BBX_OPT_FPACK_HIST_ONLY) { ctx->fpack_hist_only = value ? 1 : 0; return BBX_OK; }
    if (option == BBX_OPT_BKG_FULL_SORT) { ctx->bkg_full_sort = value ? 1 : 0; return BBX_OK; }
    if (option == BBX_OPT_WAIT_SLEEP_US) { ctx->wait_sleep_us = value > 0 ? value : 0; return BBX_OK; }
    return BBX_ERR_ARG;
}

// ---- stream plumbing (see bbx.h) ---------------------------------------------------------

int bbx_event_create(void** out_event) {
    if (!out_event) return BBX_ERR_ARG;
    hipEvent_t ev;
    BBX_HIP0(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    *out_event = (void*)ev;
    return BBX_OK;
}

void bbx_event_destroy(void* event) {
    if (event) (void)hipEventDestroy((hipEvent_t)event);
}

int bbx_event_record(void* event, void* stream) {
    if (!event) return BBX_ERR_ARG;
    BBX_HIP0(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return BBX_OK;
}

int bbx_event_query(void* event) {
    if (!event) return BBX_ERR_ARG;
    const hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return BBX_ERR_HIP;
}

int bbx_stream_wait_event(void* stream, void* event) {
    if (!event) return BBX_ERR_ARG;
    BBX_HIP0(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return BBX_OK;
}

int bbx_copy_async(void* dst, const void* src, size_t nbytes, int kind, void* stream) {
    if (!dst || !src || kind < 0 || kind > 2) return BBX_ERR_ARG;
    if (nbytes == 0) return BBX_OK;
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    BBX_HIP0(hipMemcpyAsync(dst, src, nbytes, k, (hipStream_t)stream));
    return BBX_OK;
}

}  // extern "C"
