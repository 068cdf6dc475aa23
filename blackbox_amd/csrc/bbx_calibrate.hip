// bbx_calibrate.hip -- fused calibration pass over one CCD frame.
//
// One read of the raw data sections (+ flat, bias, BPM), one write of the reduced
// float32 frame and of the uint8 mask:  HBM-bound, algorithmic bytes per pixel
// = raw(2|4) + flat 4 + [bias 4] + bpm 1 + out 4 + mask 1  (SURVEY.md section 8d).
//
// Arithmetic follows numpy's evaluation of the reference statements exactly
// (float32 storage after every statement, float64 where numpy promotes):
//   data *= gain[c]                      f32 * f32            blackbox.py:7460
//   data[chan] -= fit_vos_col[:,None]    f64 subtract -> f32  blackbox.py:6553
//   data[data_sec] -= oscan              f64 subtract -> f32  blackbox.py:6844
//   data -= data_mbias                   f32                  blackbox.py:1679
//   mask_init: non-finite -> 0 / bad, >= satlevel_c -> saturated   4408-4414, 4494-4498
//   data /= data_mflat                   f32 (IEEE division)  blackbox.py:1825
// Compile with -ffp-contract=off: no fused multiply-adds may be formed.
#include "bbx_common.h"
#ifndef CAL_NT
#define CAL_NT 1
#endif

// ---- nonlin_corr (blackbox.py:7394-7437, off upstream): per channel,
//   counts = data / gain[c]                       (float32 / float32)
//   frac   = spline_c(counts) if counts <= 50000 else 1     (float64; sic: the "else 1" halves
//            the uncorrected pixels -- reproduced)
//   data   = float32(float64(data) / (frac + 1))
// spline_c = scipy UnivariateSpline; evaluated like FITPACK splev/fpbspl (de Boor recurrence,
// float64, same operation order) from its knots t[n] and coefficients c[n].
#define NL_MAXKNOTS 256
#define NL_MAXDEG 5
struct nonlin_tab {
    int n[16];                       // knots per channel (0 = table unset)
    int k;                           // degree
    int pad[3];
    double t[16][NL_MAXKNOTS];
    double c[16][NL_MAXKNOTS];
};

__device__ __forceinline__ double nl_splev(const nonlin_tab* __restrict__ tab, int ch, double arg) {
    const double* t = tab->t[ch];
    const double* cf = tab->c[ch];
    const int n = tab->n[ch], k = tab->k, k1 = k + 1, nk1 = n - k1;
    // knot interval t(l) <= arg < t(l+1), 1-based l in [k1, nk1] (splev.f labels 35-40)
    // = the first 0-based index in [k1, nk1] whose knot exceeds arg (nk1 if none): binary search
    int l = k1, hi = nk1;
    while (l < hi) { const int mid = (l + hi) >> 1; if (arg < t[mid]) hi = mid; else l = mid + 1; }
    // non-zero B-splines at arg (fpbspl.f)
    double h[NL_MAXDEG + 1], hh[NL_MAXDEG];
    h[0] = 1.0;
    for (int j = 1; j <= k; j++) {
        for (int i = 0; i < j; i++) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 1; i <= j; i++) {
            const int li = l + i, lj = li - j;           // 1-based
            const double tli = t[li - 1], tlj = t[lj - 1];
            if (tli == tlj) { h[i] = 0.0; continue; }
            const double f = hh[i - 1] / (tli - tlj);
            h[i - 1] = h[i - 1] + f * (tli - arg);
            h[i] = f * (arg - tlj);
        }
    }
    double sp = 0.0;
    for (int j = 0; j < k1; j++) sp = sp + cf[l - k1 + j] * h[j];
    return sp;
}

__device__ __forceinline__ float nl_apply(const nonlin_tab* __restrict__ tab, int ch, float v, float gain) {
    const float counts = v / gain;
    double frac = 1.0;
    if (counts <= 50000.0f) frac = nl_splev(tab, ch, (double)counts);
    return (float)((double)v / (frac + 1.0));
}

struct calib_args {
    const nonlin_tab* nonlin;        // device table or NULL
    const void* raw; const double* vfit; const double* oscan;
    const float* bias; const float* flat; const uint8_t* bpm;
    float* data; uint8_t* mask;
    uint32_t* satlist; int32_t* counters; int32_t* err; uint32_t satcap;
    f32x16 gain, sat;
    bbx_dims d;
};

template <int RAW_T>
__device__ __forceinline__ void calib_pixel(const calib_args& a, float rawv, int c, int rl, int x, size_t o,
                                            float& out, uint8_t& m) {
    float v = rawv;
    if (RAW_T == BBX_RAW_F32 && !isfinite(v)) v = 0.f;
    v = v * a.gain.v[c];
    v = (float)((double)v - a.vfit[c * a.d.dy + rl]);
    v = (float)((double)v - a.oscan[c * a.d.xsz + x]);
    if (a.nonlin) v = nl_apply(a.nonlin, c, v, a.gain.v[c]);
    if (a.bias) v = v - a.bias[o];
    m = a.bpm ? a.bpm[o] : (uint8_t)0;
    if (!isfinite(v)) { v = 0.f; if (m == 0) m |= BBX_MASK_BAD; }
    if (v >= a.sat.v[c]) {
        m |= BBX_MASK_SAT;
        unsigned k = atomicAdd((unsigned*)&a.counters[CNT_SAT], 1u);
        if (k < a.satcap) a.satlist[k] = (uint32_t)o; else atomicOr(a.err, BBX_DERR_LIST_OVERFLOW);
    }
    if (a.flat) v = v / a.flat[o];
    out = v;
}

// Vector path (xsize_chan % 4 == 0): block = 256 threads x 4 pixels wide, CAL_ROWS rows tall.
// A thread keeps its 4 oscan values in registers for all rows; vfit[row] is block-uniform.
#define CAL_ROWS 8
template <int RAW_T, bool NONLIN>
__global__ __launch_bounds__(256) void k_calibrate_v4(calib_args a) {
    const bbx_dims& d = a.d;
    const int X = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (X >= d.nx) return;
    const int ix = X / d.xsz, x = X - ix * d.xsz;
    const int Y0 = blockIdx.y * CAL_ROWS;
    const int iy = Y0 / d.ysz;                                   // CAL_ROWS divides ysize_chan (checked by the host)
    const int c = iy * 8 + ix;
    const double os0 = a.oscan[c * d.xsz + x], os1 = a.oscan[c * d.xsz + x + 1],
                 os2 = a.oscan[c * d.xsz + x + 2], os3 = a.oscan[c * d.xsz + x + 3];
    const float g = a.gain.v[c], sat = a.sat.v[c];
    unsigned satbits = 0;                                        // bit 4*k + q: pixel q of row k is saturated
#pragma unroll
    for (int k = 0; k < CAL_ROWS; k++) {
        const int Y = Y0 + k;
        const int y = Y - iy * d.ysz;
        const int rl = (iy == 0) ? y : (d.os_y + y);
        const size_t ri = (size_t)(iy * d.dy + rl) * d.nx_raw + (size_t)ix * d.dx + x;
        const size_t o = (size_t)Y * d.nx + X;
        float r[4];
        if (RAW_T == BBX_RAW_U16) {
            const ushort4 u = *(const ushort4*)((const uint16_t*)a.raw + ri);
            r[0] = u.x; r[1] = u.y; r[2] = u.z; r[3] = u.w;
        } else {
            const float4 f = *(const float4*)((const float*)a.raw + ri);
            r[0] = f.x; r[1] = f.y; r[2] = f.z; r[3] = f.w;
        }
        const double vf = a.vfit[c * d.dy + rl];
        float fl[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
        uint8_t m[4] = {0, 0, 0, 0};
        // (CAL_NT: the masters and the raw frame stream through once, the outputs are written once: non-temporal, see
        // tools/exp/tile_bw.hip -- plain stores of a read + write stream crowd the reads out of L2)
        typedef float cv4 __attribute__((ext_vector_type(4)));
        if (a.flat) { const cv4 t = CAL_NT ? __builtin_nontemporal_load((const cv4*)(a.flat + o)) : *(const cv4*)(a.flat + o); fl[0] = t.x; fl[1] = t.y; fl[2] = t.z; fl[3] = t.w; }
        if (a.bias) { const cv4 t = CAL_NT ? __builtin_nontemporal_load((const cv4*)(a.bias + o)) : *(const cv4*)(a.bias + o); bi[0] = t.x; bi[1] = t.y; bi[2] = t.z; bi[3] = t.w; }
        if (a.bpm) { const uchar4 t = *(const uchar4*)(a.bpm + o); m[0] = t.x; m[1] = t.y; m[2] = t.z; m[3] = t.w; }
        const double os[4] = {os0, os1, os2, os3};
        float ov[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float v = r[q];
            if (RAW_T == BBX_RAW_F32 && !isfinite(v)) v = 0.f;
            v = v * g;
            v = (float)((double)v - vf);
            v = (float)((double)v - os[q]);
            if (NONLIN) v = nl_apply(a.nonlin, c, v, g);
            if (a.bias) v = v - bi[q];
            if (!isfinite(v)) { v = 0.f; if (m[q] == 0) m[q] |= BBX_MASK_BAD; }
            if (v >= sat) { m[q] |= BBX_MASK_SAT; satbits |= 1u << (4 * k + q); }
            if (a.flat) v = v / fl[q];
            ov[q] = v;
        }
        if (CAL_NT) {
            __builtin_nontemporal_store(cv4{ov[0], ov[1], ov[2], ov[3]}, (cv4*)(a.data + o));
            __builtin_nontemporal_store((unsigned)m[0] | ((unsigned)m[1] << 8) | ((unsigned)m[2] << 16) | ((unsigned)m[3] << 24), (unsigned*)(a.mask + o));
        } else {
            *(float4*)(a.data + o) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            *(uchar4*)(a.mask + o) = make_uchar4(m[0], m[1], m[2], m[3]);
        }
    }
    // saturated-pixel queue: one reservation per wave for all the rows of the block (a returning
    // atomic per pixel on one counter retires at ~11 ns each: a frame with 1 % saturated pixels
    // would spend 12 ms there).  Waves without a saturated pixel -- nearly all -- skip this.
    if (__builtin_amdgcn_ballot_w64(satbits != 0u) != 0ull) {
        const int lane = threadIdx.x & 63;
        const unsigned cnt = (unsigned)__popc(satbits);
        unsigned incl = cnt;                                     // inclusive prefix sum over the wave
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) { const unsigned t = __shfl_up(incl, sft, 64); if (lane >= sft) incl += t; }
        const unsigned total = __shfl(incl, 63, 64);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd((unsigned*)&a.counters[CNT_SAT], total);
        base = __shfl(base, 0, 64) + incl - cnt;
        if (base + cnt > a.satcap) { if (cnt) atomicOr(a.err, BBX_DERR_LIST_OVERFLOW); }
        else {
            unsigned b = satbits;
            while (b) {
                const int bit = __ffs(b) - 1; b &= b - 1;
                a.satlist[base++] = (uint32_t)((size_t)(Y0 + (bit >> 2)) * d.nx + X + (bit & 3));
            }
        }
    }
}

// scalar path for geometries without 4-pixel alignment (small test frames)
template <int RAW_T>
__global__ __launch_bounds__(256) void k_calibrate_s(calib_args a) {
    const bbx_dims& d = a.d;
    const size_t total = (size_t)d.ny * d.nx;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int Y = (int)(t / d.nx), X = (int)(t - (size_t)Y * d.nx);
        const int iy = Y / d.ysz, y = Y - iy * d.ysz;
        const int ix = X / d.xsz, x = X - ix * d.xsz;
        const int c = iy * 8 + ix;
        const int rl = (iy == 0) ? y : (d.os_y + y);           // channel-local row (data_sec origin)
        const size_t ri = (size_t)(iy * d.dy + rl) * d.nx_raw + (size_t)ix * d.dx + x;
        float ov; uint8_t mv;
        calib_pixel<RAW_T>(a, raw_load<RAW_T>(a.raw, ri), c, rl, x, t, ov, mv);
        a.data[t] = ov; a.mask[t] = mv;
    }
}

extern "C" int bbx_calibrate(bbx_ctx* ctx, const bbx_geom* g, const void* d_raw, int raw_type,
                             const float* h_gain, const double* d_vfit, const double* d_oscan,
                             const float* d_bias, const float* d_flat, const uint8_t* d_bpm,
                             const float* h_satlevel, float* d_data, uint8_t* d_mask, void* stream) {
    if (!ctx || !d_raw || !h_gain || !d_vfit || !d_oscan || !h_satlevel || !d_data || !d_mask) return BBX_ERR_ARG;
    if (raw_type != BBX_RAW_U16 && raw_type != BBX_RAW_F32) return BBX_ERR_ARG;
    calib_args a;
    int rc = bbx_make_dims(g, &a.d); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t npix = (size_t)a.d.ny * a.d.nx;
    if (npix >= 0xffffffffull) return BBX_ERR_ARG;
    // saturated-pixel queue: an eighth of the frame (a frame with more saturated pixels than that
    // raises the overflow flag -> MASK-P False; the consumers clamp to the capacity)
    const int64_t satcap = (int64_t)(npix / 8 + 4096);
    if (!ctx->d_satlist || ctx->cap_satlist < satcap) {
        if (ctx->d_satlist) { BBX_HIP(hipDeviceSynchronize()); BBX_HIP(hipFree(ctx->d_satlist)); ctx->d_satlist = nullptr; }
        BBX_HIP(hipMalloc((void**)&ctx->d_satlist, (size_t)satcap * sizeof(uint32_t)));
        ctx->cap_satlist = satcap;
    }
    a.nonlin = ctx->nonlin_on ? (const nonlin_tab*)ctx->d_nonlin : nullptr;
    a.raw = d_raw; a.vfit = d_vfit; a.oscan = d_oscan; a.bias = d_bias; a.flat = d_flat; a.bpm = d_bpm;
    a.data = d_data; a.mask = d_mask; a.satlist = ctx->d_satlist; a.counters = ctx->d_counters;
    a.err = ctx->d_err; a.satcap = (uint32_t)ctx->cap_satlist;
    for (int i = 0; i < 16; i++) { a.gain.v[i] = h_gain[i]; a.sat.v[i] = h_satlevel[i]; }
    BBX_HIP(hipMemsetAsync(&ctx->d_counters[CNT_SAT], 0, sizeof(int32_t), s));
    // vector path needs 4-pixel groups inside one channel row, aligned addresses and
    // row strips that do not straddle the two channel rows
    bool vec = (a.d.xsz % 4 == 0) && (a.d.dx % 4 == 0) && (a.d.ysz % CAL_ROWS == 0) && (((uintptr_t)d_raw) % 16 == 0) &&
               (((uintptr_t)d_data) % 16 == 0) && (((uintptr_t)d_mask) % 4 == 0) &&
               (!d_flat || ((uintptr_t)d_flat) % 16 == 0) && (!d_bias || ((uintptr_t)d_bias) % 16 == 0) &&
               (!d_bpm || ((uintptr_t)d_bpm) % 4 == 0);
    if (vec) {
        dim3 grid((a.d.nx / 4 + 255) / 256, a.d.ny / CAL_ROWS);
        if (a.nonlin) {
            if (raw_type == BBX_RAW_U16) BBX_LAUNCH_TIMED(ctx, BBX_PROF_CALIBRATE, (k_calibrate_v4<BBX_RAW_U16, true>), grid, dim3(256), 0, s, a);
            else BBX_LAUNCH_TIMED(ctx, BBX_PROF_CALIBRATE, (k_calibrate_v4<BBX_RAW_F32, true>), grid, dim3(256), 0, s, a);
        } else {
            if (raw_type == BBX_RAW_U16) BBX_LAUNCH_TIMED(ctx, BBX_PROF_CALIBRATE, (k_calibrate_v4<BBX_RAW_U16, false>), grid, dim3(256), 0, s, a);
            else BBX_LAUNCH_TIMED(ctx, BBX_PROF_CALIBRATE, (k_calibrate_v4<BBX_RAW_F32, false>), grid, dim3(256), 0, s, a);
        }
    } else {
        bbx_prof_start(ctx, BBX_PROF_CALIBRATE, s);
        unsigned grid = (unsigned)((npix + 255) / 256);
        if (grid > 256u * 16u) grid = 256u * 16u;
        if (raw_type == BBX_RAW_U16) hipLaunchKernelGGL(k_calibrate_s<BBX_RAW_U16>, dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_calibrate_s<BBX_RAW_F32>, dim3(grid), dim3(256), 0, s, a);
    }
    bbx_prof_stop(ctx, s);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

// ---- a7 standalone: nonlin_corr on an already overscan-corrected frame ------------------
__global__ __launch_bounds__(256) void k_nonlin(float* data, bbx_dims d, f32x16 gain, const nonlin_tab* __restrict__ tab) {
    const size_t total = (size_t)d.ny * d.nx;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int Y = (int)(t / d.nx), X = (int)(t - (size_t)Y * d.nx);
        const int c = (Y / d.ysz) * 8 + X / d.xsz;
        data[t] = nl_apply(tab, c, data[t], gain.v[c]);
    }
}

extern "C" int bbx_nonlin_set(bbx_ctx* ctx, int degree, const int32_t* h_nknots, const double* h_t, const double* h_c) {
    if (!ctx) return BBX_ERR_ARG;
    if (!h_nknots) { ctx->nonlin_on = 0; return BBX_OK; }                      // disable
    if (!h_t || !h_c || degree < 1 || degree > NL_MAXDEG) return BBX_ERR_ARG;
    nonlin_tab* h = (nonlin_tab*)calloc(1, sizeof(nonlin_tab));
    if (!h) return BBX_ERR_NOMEM;
    h->k = degree;
    for (int ch = 0; ch < 16; ch++) {
        const int n = h_nknots[ch];
        if (n < 2 * (degree + 1) || n > NL_MAXKNOTS) { free(h); return BBX_ERR_ARG; }
        h->n[ch] = n;
        for (int i = 0; i < n; i++) { h->t[ch][i] = h_t[ch * NL_MAXKNOTS + i]; h->c[ch][i] = h_c[ch * NL_MAXKNOTS + i]; }
    }
    if (!ctx->d_nonlin) {
        hipError_t e = hipMalloc(&ctx->d_nonlin, sizeof(nonlin_tab));
        if (e != hipSuccess) { free(h); return bbx_hip_fail(ctx, e, "hipMalloc(nonlin)", __LINE__); }
    }
    hipError_t e = hipMemcpy(ctx->d_nonlin, h, sizeof(nonlin_tab), hipMemcpyHostToDevice);
    free(h);
    if (e != hipSuccess) return bbx_hip_fail(ctx, e, "hipMemcpy(nonlin)", __LINE__);
    ctx->nonlin_on = 1;
    return BBX_OK;
}

extern "C" int bbx_nonlin_corr(bbx_ctx* ctx, const bbx_geom* g, float* d_data, const float* h_gain, void* stream) {
    if (!ctx || !d_data || !h_gain || !ctx->d_nonlin) return BBX_ERR_ARG;
    bbx_dims d; int rc = bbx_make_dims(g, &d); if (rc) return rc;
    f32x16 gain; for (int i = 0; i < 16; i++) gain.v[i] = h_gain[i];
    hipLaunchKernelGGL(k_nonlin, dim3(4096), dim3(256), 0, (hipStream_t)stream, d_data, d, gain, (const nonlin_tab*)ctx->d_nonlin);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}
