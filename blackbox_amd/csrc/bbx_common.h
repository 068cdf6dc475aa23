// bbx_common.h -- internal helpers shared by the HIP translation units (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/bbx.h"

#define BBX_WAVE 64

struct bbx_dims {
    int ny_raw, nx_raw, ysz, xsz;   // raw shape, channel data section
    int dy, dx;                     // channel size incl. overscans
    int os_y, os_x;                 // overscan rows / columns per channel
    int vos_x0, vos_w;              // vertical overscan strip: column offset in channel, width
    int hos_rows;                   // rows of os_sec_hori (os_y - 10)
    int ny, nx;                     // reduced frame shape
};

// reference define_sections, blackbox.py:6334-6402 (xbin = ybin = 1)
static inline int bbx_make_dims(const bbx_geom* g, bbx_dims* d) {
    if (!g || g->ny_raw <= 0 || g->nx_raw <= 0 || g->ysize_chan <= 0 || g->xsize_chan <= 0)
        return BBX_ERR_ARG;
    d->ny_raw = g->ny_raw; d->nx_raw = g->nx_raw;
    d->ysz = g->ysize_chan; d->xsz = g->xsize_chan;
    if (d->ny_raw % 2 || d->nx_raw % 8) return BBX_ERR_ARG;
    d->dy = d->ny_raw / 2; d->dx = d->nx_raw / 8;
    d->os_y = (d->ny_raw - 2 * d->ysz) / 2;
    d->os_x = (d->nx_raw - 8 * d->xsz) / 8;
    if (d->os_y != d->dy - d->ysz || d->os_x != d->dx - d->xsz) return BBX_ERR_ARG;
    d->hos_rows = d->os_y - 10;            // ncut_hori = 10
    d->vos_x0 = d->xsz + 5;                // ncut_vert = 5
    d->vos_w = d->dx - 1 - d->vos_x0;      // last column dropped
    if (d->hos_rows <= 0 || d->vos_w <= 0) return BBX_ERR_ARG;
    d->ny = 2 * d->ysz; d->nx = 8 * d->xsz;
    return BBX_OK;
}

// device-side error flags (bits) kept in ctx->d_err[0]
#define BBX_DERR_LIST_OVERFLOW 1
#define BBX_DERR_NOTCONV       2
#define BBX_DERR_PSF_WINDOW    4    // bbx_zogy_frame: the PSFs' matched-filter kernels do not fit their row window

struct bbx_ctx {
    int device;
    char hip_err[256];
    // --- persistent small device state
    int32_t* d_err;            // [4] error flags
    // --- work lists (device).  Capacities in elements.
    uint32_t* d_satlist;  int64_t cap_satlist;   // saturated pixel indices (reduced frame)
    void*     flags_clean_ptr; // LA-Cosmic flag plane known to be all-zero (see bbx_lacosmic)
    size_t    flags_clean_bytes;
    void*     d_nonlin;        // nonlin_tab (bbx_calibrate.hip) or NULL
    int       nonlin_on;
    void*     hash_clean_ptr;  // connected-component key table known to be all-empty (see bbx_cc_count_list)
    size_t    hash_clean_n;
    int32_t*  d_counters;      // [CNT_MAX] device counters (see enum below)
    // --- scratch, (re)allocated on demand by bbx_ws()
    void*  d_ws[24];
    size_t ws_bytes[24];
    int    lac_feed;           // BBX_OPT_LAC_LEVEL_FEED (bbx_set_option)
    int    debug_listcap;      // BBX_OPT_DEBUG_LISTCAP: capacity the LA-Cosmic kernels see (0 = the allocated one)
    void*  zogy_state;         // per-context FFT plans / work buffer of bbx_zogy.hip (NULL until first use)
    void*  zogy2_state;        // twiddle table of bbx_zogy_frame (bbx_zogy3.hip)
    int    zogy_kwin_off;      // BBX_OPT_ZOGY_KWIN_OFF: full-size transforms of the matched-filter kernels (no row window)
    int    sat_attr_set;       // dynamic-LDS attribute of k_trail_segment set through this context
    float  zcand_thr;          // bbx_zogy_candidates: > 0: bbx_zogy_frame lists the pixels with |Scorr| >= thr (WS_ZCAND, CNT_ZCAND)
    const float* zcand_img;    // the Scorr frame the list in WS_ZCAND belongs to (NULL: none); consumed by bbx_find_peaks
    float  zcand_thr_used; size_t zcand_npix;
    const float* bcand_med;    // bbx_zoom_candidates: device scalar m; the next bbx_spline_zoom_sub lists |out| >= (float)(m * bcand_nsig) (WS_BCAND, CNT_BCAND)
    double bcand_nsig;
    const float* bcand_img;    // the frame the list in WS_BCAND belongs to (consumed by bbx_find_peaks), its median scalar and factor
    const float* bcand_img_med; double bcand_img_nsig; size_t bcand_npix;
    int    wait_sleep_us;      // BBX_OPT_WAIT_SLEEP_US: host waits poll an event and sleep this long between polls (0: hipStreamSynchronize)
    int    cc_roots;           // set by a caller of bbx_cc_count_list that reads the roots afterwards (second k_cc_flatten); cleared by the call
    int    box_pp;             // which of the two CNT_BOXFAIL counters the next bbx_bkg_boxstats call fills
    int    bkg_full_sort;      // BBX_OPT_BKG_FULL_SORT: every box of bbx_bkg_boxstats through the full sort (tests: same statistics as the bracket path)
    int    fpack_hist_only;    // BBX_OPT_FPACK_HIST_ONLY: row medians by radix histograms over all keys (tests: same bytes as the bracket path)
    struct { void* stream; void* ptr; size_t bytes; } fphint[16];   // bbx_fpack_tiles: its row-hint table, one per calling stream (bbx_fpack.hip)
    int    fpack_one_wg;       // BBX_OPT_FPACK_ONE_WG: k_fp_tile with the worst-case stream buffer only (tests: both paths make the same bytes)
    int    spf_attr_bytes;     // dynamic-LDS attribute of the spline prefilter kernels set through this context
    int    zogy3_attr_L;       // sub-image side whose kernels have their dynamic-LDS attribute set through this context
    int    num_cus;            // compute units of the device (hipDeviceAttributeMultiprocessorCount)
    // --- optional per-kernel timing (bbx_profile_enable): hipEvent pairs on the launch stream
    int prof_on, prof_n;
    volatile int prof_gen;     // bumped by bbx_profile_enable / _read
    int prof_open, prof_open_gen, prof_open_idx;
    hipEvent_t* prof_ev;       // [2 * BBX_PROF_MAX]
    int* prof_slot;            // [BBX_PROF_MAX]
};
#define BBX_PROF_MAX 8192
void bbx_prof_start(bbx_ctx* ctx, int slot, hipStream_t s);
void bbx_prof_stop(bbx_ctx* ctx, hipStream_t s);
// single kernels: an event pair that the launch itself stamps (hipExtLaunchKernelGGL) -- begin and end of the kernel's
// execution, without the time the launch waits behind other streams' kernels; (nullptr, nullptr) when profiling is off
void bbx_prof_events(bbx_ctx* ctx, int slot, hipEvent_t* e0, hipEvent_t* e1);
#define BBX_LAUNCH_TIMED(ctx, slot, kern, grid, blk, lds, s, ...)                                   \
    do {                                                                                              \
        hipEvent_t e0_ = nullptr, e1_ = nullptr;                                                      \
        bbx_prof_events(ctx, slot, &e0_, &e1_);                                                       \
        hipExtLaunchKernelGGL(kern, grid, blk, lds, s, e0_, e1_, 0, __VA_ARGS__);                     \
    } while (0)

enum {
    // one 64-byte line per counter: same-line atomics serialise (~11 ns each on MI355X)
    CNT_SAT = 0,          // saturated pixels queued by calibrate
    CNT_CAND = 16,        // LA-Cosmic candidates of the current iteration
    CNT_STAGE2 = 32,      // LA-Cosmic pixels that passed the first growth step
    CNT_CRLIST = 48,      // cumulative CR pixel list
    CNT_NEWCR = 64,       // CR pixels flagged in the current iteration
    CNT_CC_N = 80,        // connected-component scratch: list length
    CNT_CC_ROOTS = 96,    // connected-component scratch: roots
    CNT_TILES = 112,      // fill-holes: unresolved tiles
    CNT_BGNEED = 128,     // LA-Cosmic: pixels that needed the background level
    CNT_TMP = 144,
    CNT_CANDOVF = 160,    // LA-Cosmic candidates that did not fit their tile segment
    CNT_CANDRAW = 176,    // LA-Cosmic candidates before the s > sigclip pre-filter
    CNT_TICKET = 192,     // workgroups of the current kernel that have finished (last one does the epilogue)
    CNT_ZCAND = 208,      // bbx_zogy_frame: pixels with |Scorr| >= the candidate threshold (bbx_zogy_candidates)
    CNT_BCAND = 224,      // bbx_spline_zoom_sub: pixels above the catalogue threshold (bbx_zoom_candidates)
    CNT_BOXFAIL0 = 240,   // bbx_bkg_boxstats: boxes left to the full sort; two counters (CNT_BOXFAIL0, + 1) used by alternate calls
    CNT_MAX = 256
};

// workspace slots
enum {
    WS_HASH = 0, WS_CCLIST, WS_PARENT, WS_BITS_M, WS_BITS_C, WS_BITS_R, WS_TILES,
    WS_CAND, WS_FLAGS, WS_STAGE2, WS_CRLIST, WS_HIST, WS_SEL, WS_MISC, WS_STRIP, WS_HVALS, WS_CRORIG, WS_CANNY, WS_ZCAND, WS_BCAND, WS_FPHINT, WS_BOXFAIL, WS_ZSPL,
    WS_MAX
};

int bbx_hip_fail(bbx_ctx* ctx, hipError_t e, const char* what, int line);
void bbx_zogy2_release(bbx_ctx* ctx);
void bbx_fpack_release(bbx_ctx* ctx);     // bbx_fpack.hip: frees the per-stream hint tables (called by bbx_ctx_destroy)
int bbx_zogy3_supported(int L);
int bbx_build_flags_fpack(void); int bbx_build_flags_zogy(void); int bbx_build_flags_bkg(void); int bbx_build_flags_sat(void); int bbx_build_flags_canny(void);
void bbx_zogy_release(bbx_ctx* ctx);      // bbx_zogy.hip: frees ctx->zogy_state (called by bbx_ctx_destroy)
void* bbx_ws(bbx_ctx* ctx, int slot, size_t bytes, int* rc);

// small per-channel parameter vectors travel as by-value kernel arguments
struct f32x16 { float v[16]; };
struct f64x256 { double v[256]; };

#define BBX_HIP(call)                                                         \
    do {                                                                      \
        hipError_t _e = (call);                                               \
        if (_e != hipSuccess) return bbx_hip_fail(ctx, _e, #call, __LINE__);  \
    } while (0)

#define BBX_LAUNCH_CHECK()                                                    \
    do {                                                                      \
        hipError_t _e = hipGetLastError();                                    \
        if (_e != hipSuccess) return bbx_hip_fail(ctx, _e, "kernel launch", __LINE__); \
    } while (0)

// ---- device helpers -------------------------------------------------------------
// Wave-wide sums, result in every lane.  Inside each 16-lane row the partial sums travel by
// DPP row rotations (plain VALU moves, no LDS crossbar like __shfl/ds_bpermute); the four row
// totals are then read with v_readlane and added in row order.
template <int CTRL> __device__ __forceinline__ int dpp_mov_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = dpp_mov_i32<CTRL>(__double2loint(v)), hi = dpp_mov_i32<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
#define BBX_DPP_ROR(n) (0x120 + (n))          // row_ror:n
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_mov_f64<BBX_DPP_ROR(1)>(v);
    v += dpp_mov_f64<BBX_DPP_ROR(2)>(v);
    v += dpp_mov_f64<BBX_DPP_ROR(4)>(v);
    v += dpp_mov_f64<BBX_DPP_ROR(8)>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += dpp_mov_i32<BBX_DPP_ROR(1)>(v);
    v += dpp_mov_i32<BBX_DPP_ROR(2)>(v);
    v += dpp_mov_i32<BBX_DPP_ROR(4)>(v);
    v += dpp_mov_i32<BBX_DPP_ROR(8)>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    // counts: two 32-bit halves would need carries; the callers' totals fit 2^53 exactly
    return (long long)wave_sum_f64((double)v);
}

// byte-wide atomic OR through the aligned 32-bit word (little endian)
__device__ __forceinline__ unsigned atomic_or_u8(uint8_t* base, size_t idx, unsigned bits) {
    size_t a = (size_t)(base + idx);
    unsigned* w = (unsigned*)(a & ~(size_t)3);
    unsigned sh = (unsigned)(a & 3) * 8;
    unsigned old = atomicOr(w, bits << sh);
    return (old >> sh) & 0xffu;
}

// raw pixel fetch: u16 or f32 -> float
template <int RAW_T>
__device__ __forceinline__ float raw_load(const void* raw, size_t i) {
    if (RAW_T == BBX_RAW_U16) return (float)((const uint16_t*)raw)[i];
    return ((const float*)raw)[i];
}

// float -> order-preserving uint32 key (radix select) and back
__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
