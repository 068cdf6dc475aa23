// bbx_lacosmic.hip -- LA-Cosmic (astroscrappy.detect_cosmics as called by
// cosmics_corr, blackbox.py:4323-4332: sepmed=False, cleantype='medmask',
// fsmode='median', gain=1, satlevel=inf) restructured for the GPU.
//
// The textbook algorithm runs two 5x5, one 3x3 and one 7x7 median filter over all
// 111 Mpx in each of the 3 iterations (VALU/LDS-bound, ~6e11 compare-exchanges).
// This implementation is exact but sparse (hard part 1):
//
//   sp = s - medfilt5(s) <= s = L+/(2 noise)   and   noise >= sqrt(float32(rn*rn))
//   => a pixel can only pass  sp > sigclip  if  L+ > T = 2*sigclip*sqrt(rn^2)
//
// so per iteration
//   1. k_lac_cand   (dense, HBM-bound: one read of the frame) computes L+ with a
//      5-point stencil and queues the pixels with L+ > T;
//   2. k_lac_seed   one wave per queued pixel evaluates sp and the fine-structure
//      ratio exactly (lanes = the 25 / 49 neighbourhood positions, per-lane
//      register median networks + a wave-wide bitonic sort);
//   3. k_lac_grow1/2 the two 3x3 growth steps on the sparse flag plane; the second
//      needs sp at neighbours that are not queued and evaluates it on demand;
//   4. k_lac_clean  masked 5x5 median at every pixel of the cumulative CR list.
// All arithmetic is float32 in the order fixed by oracle/lacosmic.py; build with
// -ffp-contract=off.  Results are bit-identical to the dense algorithm.
#include "bbx_common.h"
#include "bbx_mednet.h"

struct sel_query { uint32_t prefix; uint32_t pad; unsigned long long k; unsigned long long n; };
int bbx_select_run(bbx_ctx* ctx, const float* d_data, const uint8_t* d_mask, int ny, int nx, int ysz, int xsz,
                   int nq_per_seg, int rule, sel_query** d_q_out, hipStream_t s);
int bbx_cc_count_list(bbx_ctx* ctx, const uint32_t* d_list, const int32_t* d_cnt, size_t cap, int ny, int nx,
                      int32_t* d_out, hipStream_t s);

#define F_STAGE1 1u    // sp > sigclip & good & sp/f > objlim
#define F_HI     2u    // sp > sigclip & good
#define F_STAGE2 4u    // after first growth
#define F_SEEN  16u    // second growth: sp already evaluated for this pixel

struct lac_par {
    int ny, nx;
    float sigclip, sigcliplow, objlim, rn2, T;
};

// L+ of pixel (j,i): 2x2 replicate -> Laplacian -> clip -> 2x2 mean, closed form with
// the evaluation order of oracle/lacosmic.py (p = 4c; -= right; -= left; -= down; -= up)
__device__ __forceinline__ float lplus_px(float c, float u, float dn, float l, float r, bool hu, bool hd, bool hl,
                                          bool hr) {
    const float c4 = 4.0f * c;
    // tl: right = c, left = l?, down = c, up = u?
    float tl = c4 - c; if (hl) tl -= l; tl -= c; if (hu) tl -= u;
    // tr: right = r?, left = c, down = c, up = u?
    float tr = c4; if (hr) tr -= r; tr -= c; tr -= c; if (hu) tr -= u;
    // bl: right = c, left = l?, down = d?, up = c
    float bl = c4 - c; if (hl) bl -= l; if (hd) bl -= dn; bl -= c;
    // br: right = r?, left = c, down = d?, up = c
    float br = c4; if (hr) br -= r; br -= c; if (hd) br -= dn; br -= c;
    tl = fmaxf(tl, 0.f); tr = fmaxf(tr, 0.f); bl = fmaxf(bl, 0.f); br = fmaxf(br, 0.f);
    return (((tl + tr) + bl) + br) * 0.25f;
}

__device__ __forceinline__ float lplus_at(const float* __restrict__ a, int j, int i, int ny, int nx) {
    const size_t o = (size_t)j * nx + i;
    const bool hu = j > 0, hd = j < ny - 1, hl = i > 0, hr = i < nx - 1;
    const float c = a[o];
    const float u = hu ? a[o - nx] : 0.f, dn = hd ? a[o + nx] : 0.f;
    const float l = hl ? a[o - 1] : 0.f, r = hr ? a[o + 1] : 0.f;
    return lplus_px(c, u, dn, l, r, hu, hd, hl, hr);
}

// ---- 1. dense candidate pass ------------------------------------------------------
__global__ __launch_bounds__(256) void k_lac_cand(const float* __restrict__ a, lac_par p, uint32_t* __restrict__ cand,
                                                  int32_t* counters, uint32_t cap, int32_t* err) {
    const size_t npix = (size_t)p.ny * p.nx;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < npix; o += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(o / p.nx), i = (int)(o - (size_t)j * p.nx);
        // the outer 2-pixel frame has sp == 0 (median filter copies its border)
        if (j < 2 || i < 2 || j >= p.ny - 2 || i >= p.nx - 2) continue;
        const float lp = lplus_at(a, j, i, p.ny, p.nx);
        if (lp > p.T) {
            const unsigned k = atomicAdd((unsigned*)&counters[CNT_CAND], 1u);
            if (k < cap) cand[k] = (uint32_t)o; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
        }
    }
}

// ---- wave-cooperative evaluation ----------------------------------------------------
__device__ __forceinline__ float wave_sort_pick(float v, int width, int rank) {
    // ascending bitonic sort of one value per lane over the first [width] lanes
    // (width = 32 or 64, higher lanes must hold +inf), returns the element of [rank]
    const int lane = threadIdx.x & 63;
    for (int k = 2; k <= width; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            const float o = __shfl_xor(v, jj, 64);
            const bool up = ((lane & k) == 0);
            const bool lower = ((lane & jj) == 0);
            v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
        }
    }
    return __shfl(v, rank, 64);
}

// s(q) = L+(q) / (2 * noise(q)) for q = (j,i) anywhere in the image; also returns noise
__device__ __forceinline__ float s_at(const float* __restrict__ a, int j, int i, const lac_par& p, float* noise_out) {
    float m5;
    if (j < 2 || i < 2 || j >= p.ny - 2 || i >= p.nx - 2) {
        m5 = a[(size_t)j * p.nx + i];                       // median filter border = input
    } else {
        float v[25];
#pragma unroll
        for (int dy = 0; dy < 5; dy++)
#pragma unroll
            for (int dx = 0; dx < 5; dx++) v[dy * 5 + dx] = a[(size_t)(j + dy - 2) * p.nx + (i + dx - 2)];
        BBX_MED25(v);
        m5 = v[12];
    }
    m5 = fmaxf(m5, 0.00001f);
    const float noise = sqrtf(m5 + p.rn2);
    *noise_out = noise;
    return lplus_at(a, j, i, p.ny, p.nx) / (2.0f * noise);
}

// sp at pixel (j,i); all 64 lanes of the wave call this with the same (j,i).
// Returns sp (wave-uniform) and the noise at the pixel.
__device__ __forceinline__ float sp_wave(const float* __restrict__ a, int j, int i, const lac_par& p, float* noise_c) {
    const int lane = threadIdx.x & 63;
    float sv = __builtin_huge_valf(), nz = 0.f;
    if (lane < 25) {
        const int dy = lane / 5 - 2, dx = lane % 5 - 2;
        sv = s_at(a, j + dy, i + dx, p, &nz);
    }
    const float s_c = __shfl(sv, 12, 64);
    *noise_c = __shfl(nz, 12, 64);
    const float med = wave_sort_pick(sv, 32, 12);
    return s_c - med;
}

// fine structure f at (j,i) given the noise there (wave-uniform call)
__device__ __forceinline__ float f_wave(const float* __restrict__ a, int j, int i, const lac_par& p, float noise) {
    const int lane = threadIdx.x & 63;
    float m3 = __builtin_huge_valf();
    const bool inner7 = !(j < 3 || i < 3 || j >= p.ny - 3 || i >= p.nx - 3);
    if (inner7) {
        if (lane < 49) {
            const int qj = j + lane / 7 - 3, qi = i + lane % 7 - 3;
            if (qj < 1 || qi < 1 || qj >= p.ny - 1 || qi >= p.nx - 1) {
                m3 = a[(size_t)qj * p.nx + qi];
            } else {
                float v[9];
#pragma unroll
                for (int dy = 0; dy < 3; dy++)
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) v[dy * 3 + dx] = a[(size_t)(qj + dy - 1) * p.nx + (qi + dx - 1)];
                BBX_MED9(v);
                m3 = v[4];
            }
        }
        const float m3c = __shfl(m3, 24, 64);
        const float med = wave_sort_pick(m3, 64, 24);
        float f = (m3c - med) / noise;
        if (f < 0.01f) f = 0.01f;
        return f;
    }
    // inside the 3-pixel frame medfilt7 returns its input: f = (m3 - m3)/noise
    float f = 0.0f / noise;
    if (f < 0.01f) f = 0.01f;
    return f;
}

__device__ __forceinline__ bool good_px(const uint8_t* mask, size_t o) { return (mask[o] & ~BBX_MASK_COSMIC) == 0; }

// ---- 2. seeds: one wave per candidate ---------------------------------------------------
__global__ __launch_bounds__(256) void k_lac_seed(const float* __restrict__ a, const uint8_t* __restrict__ mask, lac_par p,
                                                  const uint32_t* __restrict__ cand, const int32_t* __restrict__ counters,
                                                  uint32_t cap, uint8_t* __restrict__ flags) {
    const int n = min((uint32_t)counters[CNT_CAND], cap);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int k = wave; k < n; k += nwaves) {
        const uint32_t o = cand[k];
        const int j = o / p.nx, i = o - j * p.nx;
        if (!good_px(mask, o)) continue;                      // wave-uniform
        float noise;
        const float sp = sp_wave(a, j, i, p, &noise);
        if (!(sp > p.sigclip)) continue;
        const float f = f_wave(a, j, i, p, noise);
        const unsigned fl = F_HI | ((sp / f > p.objlim) ? F_STAGE1 : 0u);
        if ((threadIdx.x & 63) == 0) flags[o] = (uint8_t)fl;
    }
}

// ---- 3a. first growth: dilate3(stage1) & good & sp > sigclip ---------------------------------
__global__ __launch_bounds__(256) void k_lac_grow1(lac_par p, const uint32_t* __restrict__ cand,
                                                   const int32_t* __restrict__ counters_ro, uint32_t cap, uint8_t* flags,
                                                   uint32_t* __restrict__ stage2, int32_t* counters, int32_t* err) {
    const int n = min((uint32_t)counters_ro[CNT_CAND], cap);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t o = cand[k];
        const unsigned fl = flags[o];
        if (!(fl & F_HI)) continue;
        const int j = o / p.nx, i = o - j * p.nx;
        bool any = false;
        // candidates lie >= 2 px inside the frame, so the 3x3 window is always interior
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) any |= (flags[(size_t)(j + dy) * p.nx + (i + dx)] & F_STAGE1) != 0;
        if (any) {
            flags[o] = (uint8_t)(fl | F_STAGE2);
            const unsigned q = atomicAdd((unsigned*)&counters[CNT_STAGE2], 1u);
            if (q < cap) stage2[q] = o; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
        }
    }
}

// ---- 3b. second growth: dilate3(stage2) & good & sp > sigcliplow -----------------------------
// one wave per (stage-2 pixel, neighbour); sp is evaluated once per pixel (F_SEEN claim)
__global__ __launch_bounds__(256) void k_lac_grow2(const float* __restrict__ a, uint8_t* mask, lac_par p,
                                                   const uint32_t* __restrict__ stage2, uint32_t cap, uint8_t* flags,
                                                   uint32_t* __restrict__ crlist, int32_t* counters, int32_t* err) {
    const int n = min((uint32_t)counters[CNT_STAGE2], cap);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (long long w = wave; w < (long long)n * 9; w += nwaves) {
        const uint32_t o = stage2[w / 9];
        const int nb = (int)(w % 9);
        const int j = (int)(o / p.nx) + nb / 3 - 1, i = (int)(o % p.nx) + nb % 3 - 1;
        // sp == 0 inside the 2-pixel frame: cannot exceed sigcliplow > 0
        if (j < 2 || i < 2 || j >= p.ny - 2 || i >= p.nx - 2) continue;
        const size_t r = (size_t)j * p.nx + i;
        unsigned old = 0;
        if (lane == 0) old = atomic_or_u8(flags, r, F_SEEN);
        old = __shfl(old, 0, 64);
        if (old & F_SEEN) continue;
        if (!good_px(mask, r)) continue;
        float noise;
        const float sp = sp_wave(a, j, i, p, &noise);
        if (sp > p.sigcliplow && lane == 0) {
            const unsigned om = atomic_or_u8(mask, r, BBX_MASK_COSMIC);
            atomicAdd(&counters[CNT_NEWCR], 1);
            if (!(om & BBX_MASK_COSMIC)) {
                const unsigned q = atomicAdd((unsigned*)&counters[CNT_CRLIST], 1u);
                if (q < cap) crlist[q] = (uint32_t)r; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
            }
        }
    }
}

// ---- 4. clean_medmask on the cumulative CR list -------------------------------------------------
__global__ __launch_bounds__(256) void k_lac_clean(float* a, const uint8_t* __restrict__ mask, lac_par p,
                                                   const uint32_t* __restrict__ crlist, const int32_t* __restrict__ counters,
                                                   uint32_t cap, const sel_query* __restrict__ bg) {
    const int n = min((uint32_t)counters[CNT_CRLIST], cap);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (int k = wave; k < n; k += nwaves) {
        const uint32_t o = crlist[k];
        const int j = o / p.nx, i = o - j * p.nx;
        if (j < 2 || i < 2 || j >= p.ny - 2 || i >= p.nx - 2) continue;
        float v = __builtin_huge_valf();
        int ok = 0;
        if (lane < 25) {
            const size_t q = (size_t)(j + lane / 5 - 2) * p.nx + (i + lane % 5 - 2);
            if (mask[q] == 0) { v = a[q]; ok = 1; }           // neither CR nor masked
        }
        const int cnt = __popcll(__ballot(ok));
        // +inf padding sorts last; a genuine +inf pixel would too, and is then picked in order
        const float med = wave_sort_pick(v, 32, cnt > 0 ? (cnt - 1) / 2 : 0);
        if (lane == 0) a[o] = (cnt > 0) ? med : key2f(bg->prefix);
    }
}

__global__ void k_lac_iter_end(int32_t* counters, int32_t* stats, int it) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stats[it] = counters[CNT_NEWCR];
        stats[7] = counters[CNT_CRLIST];
        counters[CNT_NEWCR] = 0; counters[CNT_CAND] = 0; counters[CNT_STAGE2] = 0;
    }
}

extern "C" int bbx_lacosmic(bbx_ctx* ctx, int ny, int nx, float* d_data, uint8_t* d_mask, float sigclip,
                            float sigfrac, float objlim, int niter, float readnoise, int32_t* d_stats, void* stream) {
    if (!ctx || !d_data || !d_mask || !d_stats || ny < 8 || nx < 8 || niter < 0 || niter > 6) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t npix = (size_t)ny * nx;
    if (npix >= 0xffffffffull || ((uintptr_t)d_mask) % 4) return BBX_ERR_ARG;
    int rc;
    lac_par p;
    p.ny = ny; p.nx = nx; p.sigclip = sigclip; p.sigcliplow = sigfrac * sigclip; p.objlim = objlim;
    p.rn2 = readnoise * readnoise;
    // prune threshold: sp <= L+/(2*sqrtf(rn2)); the (1 - 1e-5) factor absorbs the float32
    // roundings of the division and of 2*noise (LA-Cosmic)
    p.T = 2.0f * sigclip * sqrtf(p.rn2) * (1.0f - 1e-5f);
    const size_t cap = npix / 4 + 4096;
    uint32_t* cand = (uint32_t*)bbx_ws(ctx, WS_CAND, cap * 4, &rc); if (rc) return rc;
    uint32_t* stage2 = (uint32_t*)bbx_ws(ctx, WS_STAGE2, cap * 4, &rc); if (rc) return rc;
    uint32_t* crlist = (uint32_t*)bbx_ws(ctx, WS_CRLIST, cap * 4, &rc); if (rc) return rc;
    uint8_t* flags = (uint8_t*)bbx_ws(ctx, WS_FLAGS, npix + 16, &rc); if (rc) return rc;
    int32_t* cnt = ctx->d_counters;
    BBX_HIP(hipMemsetAsync(d_stats, 0, 8 * sizeof(int32_t), s));
    BBX_HIP(hipMemsetAsync(&cnt[CNT_CAND], 0, 4 * sizeof(int32_t), s));      // CAND, STAGE2, CRLIST, NEWCR
    // background level of the unmasked input pixels (needed when a CR pixel has no good neighbour)
    sel_query* bg;
    rc = bbx_select_run(ctx, d_data, d_mask, ny, nx, ny, nx, 1, 0, &bg, s);
    if (rc) return rc;
    const unsigned gdense = 256u * 16u, gsparse = 256u * 8u;
    for (int it = 0; it < niter; it++) {
        BBX_HIP(hipMemsetAsync(flags, 0, npix, s));
        hipLaunchKernelGGL(k_lac_cand, dim3(gdense), dim3(256), 0, s, d_data, p, cand, cnt, (uint32_t)cap, ctx->d_err);
        hipLaunchKernelGGL(k_lac_seed, dim3(gsparse), dim3(256), 0, s, d_data, d_mask, p, cand, cnt, (uint32_t)cap, flags);
        hipLaunchKernelGGL(k_lac_grow1, dim3(gsparse), dim3(256), 0, s, p, cand, cnt, (uint32_t)cap, flags, stage2, cnt, ctx->d_err);
        hipLaunchKernelGGL(k_lac_grow2, dim3(gsparse), dim3(256), 0, s, d_data, d_mask, p, stage2, (uint32_t)cap, flags, crlist,
                           cnt, ctx->d_err);
        hipLaunchKernelGGL(k_lac_clean, dim3(gsparse), dim3(256), 0, s, d_data, d_mask, p, crlist, cnt, (uint32_t)cap, bg);
        hipLaunchKernelGGL(k_lac_iter_end, dim3(1), dim3(64), 0, s, cnt, d_stats, it);
    }
    BBX_LAUNCH_CHECK();
    // NCOSMICS: 8-connected objects of the CR pixels (blackbox.py:4354-4356)
    return bbx_cc_count_list(ctx, crlist, &cnt[CNT_CRLIST], cap, ny, nx, &d_stats[6], s);
}
