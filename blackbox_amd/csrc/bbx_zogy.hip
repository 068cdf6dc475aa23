// bbx_zogy.hip -- ZOGY image subtraction per sub-image with rocFFT, and PSF photometry.
// (zogy.optimal_subtraction -> run_ZOGY, called at blackbox.py:2350-2354 / 2460-2465;
// Zackay, Ofek & Gal-Yam 2016.  [EXT: parity unpinned, conventions = oracle/zogy_core.py])
//
// All spatial quantities are real, so every transform is R2C / C2R on the half spectrum
// (L x (L/2+1) complex64), batched over the sub-images.  Per sub-image: 8 forward + 8 inverse
// 2-D FFTs (rocFFT, HBM-bound) and four element-wise kernels:
//   spec1 : D^, S^, k_r^, k_n^, (k_n N)^, (k_r R)^ and the F_S sum   from N^, R^, Pn^, Pr^
//   sq    : k_r^2, k_n^2 (real space)
//   spec2 : V(S_r)^ = Vr^ * (k_r^2)^ ,  V(S_n)^ = Vn^ * (k_n^2)^
//   final : D, S, S_corr (incl. the astrometric variance from the finite differences of
//           S_n, S_r), F_psf = S / F_S, F_psf_err = sqrt(V_S) / F_S
// rocFFT transforms are unnormalised: the 1/L^2 of every inverse is folded into the kernels.
#include "bbx_common.h"
#include "bbx_spline.h"
#include <rocfft/rocfft.h>
#include <stdlib.h>
#include <mutex>

struct zogy_scal { float sn, sr, fn, fr, dx, dy; };

struct zogy_plans {
    int L, batch;
    rocfft_plan fwd, inv;
    rocfft_execution_info info;
    void* work; size_t work_bytes;
};
// rocfft_setup() is the only process-wide step (once, under a lock); plans, the work buffer and the
// execution info (which carries the stream) belong to the context: two lanes never share them
static std::mutex g_rocfft_lock;
static int g_rocfft_ready = 0;

#define RFFT(call) do { rocfft_status _s = (call); if (_s != rocfft_status_success) { \
    snprintf(ctx->hip_err, sizeof(ctx->hip_err), "rocFFT: %s -> %d (line %d)", #call, (int)_s, __LINE__); return BBX_ERR_HIP; } } while (0)

static void drop_plans(zogy_plans* P) {
    if (P->fwd) { rocfft_plan_destroy(P->fwd); P->fwd = nullptr; }
    if (P->inv) { rocfft_plan_destroy(P->inv); P->inv = nullptr; }
    if (P->info) { rocfft_execution_info_destroy(P->info); P->info = nullptr; }
    if (P->work) { (void)hipDeviceSynchronize(); (void)hipFree(P->work); P->work = nullptr; }
    P->L = P->batch = 0; P->work_bytes = 0;
}

void bbx_zogy_release(bbx_ctx* ctx) {
    if (!ctx || !ctx->zogy_state) return;
    zogy_plans* P = (zogy_plans*)ctx->zogy_state;
    drop_plans(P);
    free(P);
    ctx->zogy_state = nullptr;
}

static int get_plans(bbx_ctx* ctx, int L, int batch, hipStream_t s, zogy_plans** out) {
    {
        std::lock_guard<std::mutex> lk(g_rocfft_lock);
        if (!g_rocfft_ready) { RFFT(rocfft_setup()); g_rocfft_ready = 1; }
    }
    if (!ctx->zogy_state) {
        ctx->zogy_state = calloc(1, sizeof(zogy_plans));
        if (!ctx->zogy_state) return BBX_ERR_NOMEM;
    }
    zogy_plans* P = (zogy_plans*)ctx->zogy_state;
    if (P->L != L || P->batch != batch) {
        drop_plans(P);
        const size_t len[2] = {(size_t)L, (size_t)L};
        std::lock_guard<std::mutex> lk(g_rocfft_lock);          // plan creation touches rocFFT's kernel cache
        RFFT(rocfft_plan_create(&P->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                rocfft_precision_single, 2, len, (size_t)batch, nullptr));
        RFFT(rocfft_plan_create(&P->inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                rocfft_precision_single, 2, len, (size_t)batch, nullptr));
        size_t w1 = 0, w2 = 0;
        RFFT(rocfft_plan_get_work_buffer_size(P->fwd, &w1));
        RFFT(rocfft_plan_get_work_buffer_size(P->inv, &w2));
        P->work_bytes = w1 > w2 ? w1 : w2;
        if (P->work_bytes) BBX_HIP(hipMalloc(&P->work, P->work_bytes));
        RFFT(rocfft_execution_info_create(&P->info));
        if (P->work_bytes) RFFT(rocfft_execution_info_set_work_buffer(P->info, P->work, P->work_bytes));
        P->L = L; P->batch = batch;
    }
    RFFT(rocfft_execution_info_set_stream(P->info, s));
    *out = P;
    return BBX_OK;
}

static int fft_fwd(bbx_ctx* ctx, zogy_plans* P, float* in, float2* out) {
    void* i[1] = {in}; void* o[1] = {out};
    RFFT(rocfft_execute(P->fwd, i, o, P->info));
    return BBX_OK;
}
static int fft_inv(bbx_ctx* ctx, zogy_plans* P, float2* in, float* out) {       // destroys [in]
    void* i[1] = {in}; void* o[1] = {out};
    RFFT(rocfft_execute(P->inv, i, o, P->info));
    return BBX_OK;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

// spectra are [nsub][L][H], H = L/2+1
__global__ __launch_bounds__(256) void k_zogy_spec1(int L, int H, const float2* __restrict__ Nh, const float2* __restrict__ Rh,
                                                    const float2* __restrict__ Pnh, const float2* __restrict__ Prh,
                                                    const zogy_scal* __restrict__ sc, float2* Dh, float2* Sh, float2* krh,
                                                    float2* knh, float2* SnH, float2* SrH, double* fs_partial) {
    const int sub = blockIdx.y;
    const zogy_scal z = sc[sub];
    const float sn2 = z.sn * z.sn, sr2 = z.sr * z.sr, fn2 = z.fn * z.fn, fr2 = z.fr * z.fr;
    const float fD = z.fr * z.fn / sqrtf(sn2 * fr2 + sr2 * fn2);
    const size_t per = (size_t)L * H, base = (size_t)sub * per;
    double fs = 0.0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < per; k += (size_t)gridDim.x * blockDim.x) {
        const float2 n = Nh[base + k], r = Rh[base + k], pn = Pnh[base + k], pr = Prh[base + k];
        const float pn2 = pn.x * pn.x + pn.y * pn.y, pr2 = pr.x * pr.x + pr.y * pr.y;
        const float den = (sn2 * fr2) * pr2 + (sr2 * fn2) * pn2;
        const float isd = 1.0f / sqrtf(den);
        const float2 a = cscale(cmul(pr, n), z.fr), b = cscale(cmul(pn, r), z.fn);
        const float2 dh = cscale(make_float2(a.x - b.x, a.y - b.y), isd);
        const float2 pdh = cscale(cmul(pr, pn), (z.fr * z.fn / fD) * isd);
        const float2 sh = cmul(cscale(dh, fD), cconj(pdh));
        const float2 kr = cscale(cconj(pr), z.fr * fn2 * pn2 / den);
        const float2 kn = cscale(cconj(pn), z.fn * fr2 * pr2 / den);
        Dh[base + k] = dh; Sh[base + k] = sh; krh[base + k] = kr; knh[base + k] = kn;
        SnH[base + k] = cmul(kn, n); SrH[base + k] = cmul(kr, r);
        // Hermitian weight of this half-spectrum column in the full-spectrum sum
        const int kx = (int)(k % H);
        const double w = (kx == 0 || (L % 2 == 0 && kx == L / 2)) ? 1.0 : 2.0;
        fs += w * (double)(fn2 * pn2 * fr2 * pr2 / den);
    }
    fs = wave_sum_f64(fs);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = fs;
    __syncthreads();
    if (threadIdx.x == 0) fs_partial[(size_t)sub * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void k_zogy_fs(int nblocks, int L, const double* __restrict__ fs_partial, float* __restrict__ FS) {
    const int sub = blockIdx.x;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < nblocks; b++) s += fs_partial[(size_t)sub * nblocks + b];
        FS[sub] = (float)(s / ((double)L * (double)L));
    }
}

// kr, kn come out of the unnormalised inverse: scale by 1/L^2 first, then square
__global__ __launch_bounds__(256) void k_zogy_sq(size_t n, float inv_n2, const float* __restrict__ kr, const float* __restrict__ kn,
                                                 float* kr2, float* kn2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = kr[i] * inv_n2, b = kn[i] * inv_n2;
        kr2[i] = a * a; kn2[i] = b * b;
    }
}

__global__ __launch_bounds__(256) void k_zogy_spec2(size_t n, const float2* __restrict__ Vrh, const float2* __restrict__ kr2h,
                                                    const float2* __restrict__ Vnh, const float2* __restrict__ kn2h,
                                                    float2* VSrh, float2* VSnh) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        VSrh[i] = cmul(Vrh[i], kr2h[i]); VSnh[i] = cmul(Vnh[i], kn2h[i]);
    }
}

__global__ __launch_bounds__(256) void k_zogy_final(int L, float inv_n2, const zogy_scal* __restrict__ sc,
                                                    const float* __restrict__ FS, const float* __restrict__ Draw,
                                                    const float* __restrict__ Sraw, const float* __restrict__ Sn,
                                                    const float* __restrict__ Sr, const float* __restrict__ VSr,
                                                    const float* __restrict__ VSn, float* D, float* S, float* Scorr,
                                                    float* Fpsf, float* Fpsferr) {
    const int sub = blockIdx.y;
    const zogy_scal z = sc[sub];
    const float sn2 = z.sn * z.sn, sr2 = z.sr * z.sr, fn2 = z.fn * z.fn, fr2 = z.fr * z.fr;
    const float fD = z.fr * z.fn / sqrtf(sn2 * fr2 + sr2 * fn2);
    const float fs = FS[sub];
    const size_t per = (size_t)L * L, base = (size_t)sub * per;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < per; k += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(k / L), x = (int)(k - (size_t)y * L);
        const int ym = (y == 0) ? L - 1 : y - 1, xm = (x == 0) ? L - 1 : x - 1;      // np.roll(..., 1)
        const float s = Sraw[base + k] * inv_n2;
        const float snc = Sn[base + k] * inv_n2, src = Sr[base + k] * inv_n2;
        const float dSndy = snc - Sn[base + (size_t)ym * L + x] * inv_n2, dSndx = snc - Sn[base + (size_t)y * L + xm] * inv_n2;
        const float dSrdy = src - Sr[base + (size_t)ym * L + x] * inv_n2, dSrdx = src - Sr[base + (size_t)y * L + xm] * inv_n2;
        const float vast = z.dx * z.dx * (dSndx * dSndx + dSrdx * dSrdx) + z.dy * z.dy * (dSndy * dSndy + dSrdy * dSrdy);
        const float vs = VSr[base + k] * inv_n2 + VSn[base + k] * inv_n2;
        D[base + k] = Draw[base + k] * inv_n2 / fD;
        S[base + k] = s;
        Scorr[base + k] = s / sqrtf(vs + vast);
        Fpsf[base + k] = s / fs;
        Fpsferr[base + k] = sqrtf(fmaxf(vs, 0.f)) / fs;
    }
}

// ---- sub-image cut / stitch --------------------------------------------------------------
// four rows per thread: a quarter of the workgroups, four independent loads in flight
#define CUT_ROWS 4
__global__ __launch_bounds__(256) void k_cut(const float* __restrict__ img, int ny, int nx, int size, int border, int nsx,
                                             float* __restrict__ subs) {
    const int L = size + 2 * border;
    const int sub = blockIdx.z, sy = sub / nsx, sx = sub - sy * nsx;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= L) return;
    const int X = sx * size - border + x;
    const bool xin = X >= 0 && X < nx;
    float v[CUT_ROWS];
#pragma unroll
    for (int j = 0; j < CUT_ROWS; j++) {
        const int y = blockIdx.y * CUT_ROWS + j, Y = sy * size - border + y;
        v[j] = (y < L && xin && Y >= 0 && Y < ny) ? img[(size_t)Y * nx + X] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < CUT_ROWS; j++) {
        const int y = blockIdx.y * CUT_ROWS + j;
        if (y < L) subs[((size_t)sub * L + y) * L + x] = v[j];
    }
}

__global__ __launch_bounds__(256) void k_stitch(const float* __restrict__ subs, int ny, int nx, int size, int border, int nsx,
                                                float* __restrict__ img) {
    const int L = size + 2 * border;
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    if (X >= nx) return;
    const int sx = X / size;
    float v[CUT_ROWS];
#pragma unroll
    for (int j = 0; j < CUT_ROWS; j++) {
        const int Y = blockIdx.y * CUT_ROWS + j, sy = Y / size;
        v[j] = (Y < ny) ? subs[((size_t)(sy * nsx + sx) * L + (Y - sy * size + border)) * L + (X - sx * size + border)] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < CUT_ROWS; j++) {
        const int Y = blockIdx.y * CUT_ROWS + j;
        if (Y < ny) img[(size_t)Y * nx + X] = v[j];
    }
}

// ---- PSF photometry: one wave per source ---------------------------------------------------
// (SIGMA: V holds the sigma image of the background-subtracted frame D; the variance max(D, 0) + sigma^2 is formed here,
// in float32 like k_variance does, instead of being written out for the whole frame first)
template <bool SIGMA>
__global__ __launch_bounds__(256) void k_psf_optflux(int ny, int nx, const float* __restrict__ D, const float* __restrict__ V,
                                                     const float* __restrict__ psfs, int S, int nsrc,
                                                     const int32_t* __restrict__ ys, const int32_t* __restrict__ xs,
                                                     float* __restrict__ flux, float* __restrict__ err) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int h = S / 2;
    for (int k = wave; k < nsrc; k += nwaves) {
        double num = 0.0, den = 0.0;
        for (int t = lane; t < S * S; t += 64) {
            const int j = t / S, i = t - j * S;
            const int y = ys[k] + j - h, x = xs[k] + i - h;
            if (y < 0 || y >= ny || x < 0 || x >= nx) continue;
            float vf = V[(size_t)y * nx + x];
            if (SIGMA) vf = fmaxf(D[(size_t)y * nx + x], 0.f) + vf * vf;
            const double v = (double)vf;
            if (!(v > 0.0)) continue;
            const double p = (double)psfs[((size_t)k * S + j) * S + i];
            num += p * (double)D[(size_t)y * nx + x] / v;
            den += p * p / v;
        }
        num = wave_sum_f64(num); den = wave_sum_f64(den);
        if (lane == 0) {
            flux[k] = den > 0.0 ? (float)(num / den) : 0.f;
            err[k] = den > 0.0 ? (float)(1.0 / sqrt(den)) : 0.f;
        }
    }
}

// the same with sigma read off its mini image (no sigma frame).  A stamp of S <= 64 pixels a side crosses at most four
// coefficient columns (box intervals; one more at a channel border): a wave first works out, per stamp column, which of them
// it lies in and its place t inside (49 divisions instead of 2401), then the cubics of the stamp's rows on those intervals
// (bbx_spl_poly: the float64 fold of the row weights, 4 S of them instead of S^2), and a pixel costs one 16-byte LDS read
// and three multiply-adds.  (First version, one full evaluation per pixel: 0.31 ms for the bench's 11 000 sources against
// 0.13 ms of the kernel that reads a sigma frame.)
#define OPT_SMAX 64
#define OPT_KI 4
__global__ __launch_bounds__(256) void k_psf_optflux_mini(int ny, int nx, const float* __restrict__ D, bbx_spl sp,
                                                          const float* __restrict__ psfs, int S, int nsrc,
                                                          const int32_t* __restrict__ ys, const int32_t* __restrict__ xs,
                                                          float* __restrict__ flux, float* __restrict__ err) {
    __shared__ float4 s_poly[4][OPT_SMAX][OPT_KI];
    __shared__ float s_t[4][OPT_SMAX];
    __shared__ int s_k[4][OPT_SMAX], s_c[4][OPT_KI + 1];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int h = S / 2;
    const bool table = S <= OPT_SMAX;
    for (int k = wave; k < nsrc; k += nwaves) {
        const int y0 = ys[k] - h, x0 = xs[k] - h;
        bool fits = table;
        if (table) {
            // stamp columns -> (compact interval number, t); the distinct coefficient columns in order
            int c = 0, r = 0;
            const int X = min(max(x0 + lane, 0), nx - 1);
            bbx_spl_axis(X, sp.pw, sp.rpw, sp.px, sp.npad, sp.nx1, sp.dx, sp.rdx, c, r);
            const int cprev = __shfl_up(c, 1, 64);
            const bool act = lane < S;
            const unsigned long long chg = __ballot(act && lane > 0 && c != cprev);
            const int kk = __popcll(chg & ((2ull << lane) - 1ull));            // changes at lanes <= this one
            const int nk = __popcll(chg) + 1;
            fits = nk <= OPT_KI;                                                // (wave-uniform)
            if (fits) {
                if (act) { s_k[wib][lane] = kk; s_t[wib][lane] = (float)r * sp.rdx; }
                if (act && (lane == 0 || ((chg >> lane) & 1ull))) s_c[wib][kk] = c;
                // (same wave: LDS writes are visible to its later reads in program order)
                for (int t = lane; t < S * nk; t += 64) {
                    const int j = t / nk, q = t - j * nk;
                    const int Y = min(max(y0 + j, 0), ny - 1);
                    s_poly[wib][j][q] = bbx_spl_poly(sp, Y, s_c[wib][q]);
                }
            }
        }
        double num = 0.0, den = 0.0;
        for (int t = lane; t < S * S; t += 64) {
            const int j = t / S, i = t - j * S;
            const int y = y0 + j, x = x0 + i;
            if (y < 0 || y >= ny || x < 0 || x >= nx) continue;
            const float sg = fits ? bbx_spl_horner(s_poly[wib][j][s_k[wib][i]], s_t[wib][i]) : bbx_spl_eval(sp, y, x);
            const float d = D[(size_t)y * nx + x];
            const double v = (double)(fmaxf(d, 0.f) + sg * sg);
            if (!(v > 0.0)) continue;
            const double p = (double)psfs[((size_t)k * S + j) * S + i];
            num += p * (double)d / v;
            den += p * p / v;
        }
        num = wave_sum_f64(num); den = wave_sum_f64(den);
        if (lane == 0) {
            flux[k] = den > 0.0 ? (float)(num / den) : 0.f;
            err[k] = den > 0.0 ? (float)(1.0 / sqrt(den)) : 0.f;
        }
    }
}

// variance image of a background-subtracted frame: V = max(data, 0) + bkg_std^2
__global__ __launch_bounds__(256) void k_variance(size_t n, const float* __restrict__ d, const float* __restrict__ sd, float* v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float s = sd[i];
        v[i] = fmaxf(d[i], 0.f) + s * s;
    }
}

// PSF stamps [nsub][S][S] (centre at S/2) -> [nsub][L][L] images centred on pixel [0,0] (wrap-around)
__global__ __launch_bounds__(256) void k_embed_psf(int nsub, int S, int L, const float* __restrict__ st, float* __restrict__ out) {
    const size_t n = (size_t)nsub * S * S;
    const int h = S / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int sub = (int)(i / (S * S)), r = (int)(i % (S * S)), j = r / S, k = r % S;
        const int y = ((j - h) % L + L) % L, x = ((k - h) % L + L) % L;
        out[((size_t)sub * L + y) * L + x] = st[i];
    }
}

extern "C" {

int bbx_variance(bbx_ctx* ctx, int64_t n, const float* d_data, const float* d_bkgstd, float* d_var, void* stream) {
    if (!ctx || !d_data || !d_bkgstd || !d_var || n <= 0) return BBX_ERR_ARG;
    hipLaunchKernelGGL(k_variance, dim3(2048), dim3(256), 0, (hipStream_t)stream, (size_t)n, d_data, d_bkgstd, d_var);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_embed_psf(bbx_ctx* ctx, int nsub, int S, int L, const float* d_stamps, float* d_out, void* stream) {
    if (!ctx || !d_stamps || !d_out || nsub < 1 || S < 1 || S > L) return BBX_ERR_ARG;
    BBX_HIP(hipMemsetAsync(d_out, 0, (size_t)nsub * L * L * sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(k_embed_psf, dim3(1024), dim3(256), 0, (hipStream_t)stream, nsub, S, L, d_stamps, d_out);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_cut_subimages(bbx_ctx* ctx, int ny, int nx, int size, int border, const float* d_img, float* d_subs, void* stream) {
    if (!ctx || !d_img || !d_subs || size < 1 || border < 0 || ny % size || nx % size) return BBX_ERR_ARG;
    const int L = size + 2 * border, nsy = ny / size, nsx = nx / size;
    hipLaunchKernelGGL(k_cut, dim3((L + 255) / 256, (L + CUT_ROWS - 1) / CUT_ROWS, nsy * nsx), dim3(256), 0, (hipStream_t)stream, d_img, ny, nx, size, border, nsx, d_subs);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_stitch_subimages(bbx_ctx* ctx, int ny, int nx, int size, int border, const float* d_subs, float* d_img, void* stream) {
    if (!ctx || !d_img || !d_subs || size < 1 || border < 0 || ny % size || nx % size) return BBX_ERR_ARG;
    hipLaunchKernelGGL(k_stitch, dim3((nx + 255) / 256, (ny + CUT_ROWS - 1) / CUT_ROWS), dim3(256), 0, (hipStream_t)stream, d_subs, ny, nx, size, border, nx / size, d_img);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_zogy_subimages(bbx_ctx* ctx, int L, int nsub, float* d_new, float* d_ref, float* d_pn, float* d_pr, float* d_vn,
                       float* d_vr, const float* h_scal /*[nsub][6]: sn sr fn fr dx dy*/, float* d_D, float* d_S,
                       float* d_Scorr, float* d_Fpsf, float* d_Fpsferr, void* stream) {
    if (!ctx || !d_new || !d_ref || !d_pn || !d_pr || !d_vn || !d_vr || !h_scal || !d_D || !d_S || !d_Scorr || !d_Fpsf || !d_Fpsferr)
        return BBX_ERR_ARG;
    if (L < 8 || nsub < 1 || nsub > 4096) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    zogy_plans* P = nullptr;
    int rc = get_plans(ctx, L, nsub, s, &P); if (rc) return rc;
    const int H = L / 2 + 1;
    const size_t nreal = (size_t)nsub * L * L, nspec = (size_t)nsub * L * H;
    const int NB = 64;                                      // partial-sum blocks per sub-image
    // workspace: 10 spectra + 8 real temporaries + scalars
    const size_t bytes = 10 * nspec * sizeof(float2) + 8 * nreal * sizeof(float) + nsub * sizeof(zogy_scal) +
                         (size_t)nsub * NB * sizeof(double) + nsub * sizeof(float) + 4096;
    char* ws = (char*)bbx_ws(ctx, WS_CAND, bytes, &rc); if (rc) return rc;   // shares the big LA-Cosmic slot
    float2* spec[10]; for (int i = 0; i < 10; i++) spec[i] = (float2*)(ws + (size_t)i * nspec * sizeof(float2));
    char* p = ws + 10 * nspec * sizeof(float2);
    float* real[8]; for (int i = 0; i < 8; i++) { real[i] = (float*)p; p += nreal * sizeof(float); }
    zogy_scal* d_sc = (zogy_scal*)p; p += nsub * sizeof(zogy_scal);
    p = (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    double* fs_partial = (double*)p; p += (size_t)nsub * NB * sizeof(double);
    float* FS = (float*)p;
    BBX_HIP(hipMemcpyAsync(d_sc, h_scal, nsub * sizeof(zogy_scal), hipMemcpyHostToDevice, s));
    float2 *Nh = spec[0], *Rh = spec[1], *Pnh = spec[2], *Prh = spec[3], *Dh = spec[4], *Sh = spec[5], *krh = spec[6],
           *knh = spec[7], *SnH = spec[8], *SrH = spec[9];
    float *Draw = real[0], *Sraw = real[1], *kr = real[2], *kn = real[3], *Sn = real[4], *Sr = real[5], *t0 = real[6], *t1 = real[7];
    bbx_prof_start(ctx, BBX_PROF_ZOGY, s);
    if ((rc = fft_fwd(ctx, P, d_new, Nh)) || (rc = fft_fwd(ctx, P, d_ref, Rh)) || (rc = fft_fwd(ctx, P, d_pn, Pnh)) || (rc = fft_fwd(ctx, P, d_pr, Prh))) return rc;
    hipLaunchKernelGGL(k_zogy_spec1, dim3(NB, nsub), dim3(256), 0, s, L, H, Nh, Rh, Pnh, Prh, d_sc, Dh, Sh, krh, knh, SnH, SrH, fs_partial);
    hipLaunchKernelGGL(k_zogy_fs, dim3(nsub), dim3(64), 0, s, NB, L, fs_partial, FS);
    if ((rc = fft_inv(ctx, P, Dh, Draw)) || (rc = fft_inv(ctx, P, Sh, Sraw)) || (rc = fft_inv(ctx, P, krh, kr)) || (rc = fft_inv(ctx, P, knh, kn)) ||
        (rc = fft_inv(ctx, P, SnH, Sn)) || (rc = fft_inv(ctx, P, SrH, Sr))) return rc;
    const float inv_n2 = 1.0f / ((float)L * (float)L);
    hipLaunchKernelGGL(k_zogy_sq, dim3(2048), dim3(256), 0, s, nreal, inv_n2, kr, kn, t0, t1);
    // reuse spectra: Vr^ -> Nh, kr2^ -> Rh, Vn^ -> Pnh, kn2^ -> Prh, products -> Dh, Sh
    if ((rc = fft_fwd(ctx, P, d_vr, Nh)) || (rc = fft_fwd(ctx, P, t0, Rh)) || (rc = fft_fwd(ctx, P, d_vn, Pnh)) || (rc = fft_fwd(ctx, P, t1, Prh))) return rc;
    hipLaunchKernelGGL(k_zogy_spec2, dim3(2048), dim3(256), 0, s, nspec, Nh, Rh, Pnh, Prh, Dh, Sh);
    if ((rc = fft_inv(ctx, P, Dh, t0)) || (rc = fft_inv(ctx, P, Sh, t1))) return rc;           // VSr, VSn
    hipLaunchKernelGGL(k_zogy_final, dim3(256, nsub), dim3(256), 0, s, L, inv_n2, d_sc, FS, Draw, Sraw, Sn, Sr, t0, t1, d_D, d_S,
                       d_Scorr, d_Fpsf, d_Fpsferr);
    bbx_prof_stop(ctx, s);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_psf_optflux(bbx_ctx* ctx, int ny, int nx, const float* d_D, const float* d_V, const float* d_psfs, int S, int nsrc,
                    const int32_t* d_ys, const int32_t* d_xs, float* d_flux, float* d_err, void* stream) {
    if (!ctx || !d_D || !d_V || !d_psfs || !d_ys || !d_xs || !d_flux || !d_err || S < 1 || nsrc < 0) return BBX_ERR_ARG;
    if (nsrc == 0) return BBX_OK;
    unsigned grid = (unsigned)((nsrc + 3) / 4); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_psf_optflux<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, ny, nx, d_D, d_V, d_psfs, S, nsrc, d_ys, d_xs, d_flux, d_err);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_psf_optflux_sigma(bbx_ctx* ctx, int ny, int nx, const float* d_D, const float* d_sigma, const float* d_psfs, int S, int nsrc,
                          const int32_t* d_ys, const int32_t* d_xs, float* d_flux, float* d_err, void* stream) {
    if (!ctx || !d_D || !d_sigma || !d_psfs || !d_ys || !d_xs || !d_flux || !d_err || S < 1 || nsrc < 0) return BBX_ERR_ARG;
    if (nsrc == 0) return BBX_OK;
    unsigned grid = (unsigned)((nsrc + 3) / 4); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_psf_optflux<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, ny, nx, d_D, d_sigma, d_psfs, S, nsrc, d_ys, d_xs, d_flux, d_err);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_psf_optflux_mini(bbx_ctx* ctx, int ny, int nx, const float* d_D, const bbx_spline_image* sigma, const float* d_psfs, int S, int nsrc,
                         const int32_t* d_ys, const int32_t* d_xs, float* d_flux, float* d_err, void* stream) {
    if (!ctx || !d_D || !sigma || !d_psfs || !d_ys || !d_xs || !d_flux || !d_err || S < 1 || nsrc < 0) return BBX_ERR_ARG;
    bbx_spl sp;
    const int rc = bbx_spl_make(sigma, ny, nx, &sp); if (rc) return rc;
    if (nsrc == 0) return BBX_OK;
    unsigned grid = (unsigned)((nsrc + 3) / 4); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_psf_optflux_mini, dim3(grid), dim3(256), 0, (hipStream_t)stream, ny, nx, d_D, sp, d_psfs, S, nsrc, d_ys, d_xs, d_flux, d_err);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

}  // extern "C"
