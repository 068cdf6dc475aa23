// bbx_overscan.hip -- overscan strip reductions (reference os_corr, blackbox.py:6407-6879)
//
// The bulk reductions run here; the float64 polynomial / spline fits on the
// resulting <= 5300-point vectors stay on the host (blackbox_amd/overscan.py),
// as SURVEY.md section 7 plans.  All kernels are HBM-bound strip readers.
#include "bbx_common.h"

#define VOS_VMAX 8      // values per lane -> strips up to 512 columns wide

// ---------------------------------------------------------------------------------
// per (channel,row): astropy sigma_clipped_stats(axis=1, mask_value=0,
// cenfunc='mean') of the gain-corrected vertical overscan (os_corr 6480-6490).
// One wave per row; values live in registers as float64; clip loop follows
// astropy's C gufunc (mean, std ddof=0, closed interval, <= 5 bound updates,
// final bounds applied to all values).
// ---------------------------------------------------------------------------------
template <int RAW_T>
__global__ __launch_bounds__(256) void k_vos_rowstats(const void* __restrict__ raw, bbx_dims d,
                                                      f32x16 gain, double* __restrict__ mean_out) {
    const int lane = threadIdx.x & 63;
    const int row_id = blockIdx.x * 4 + (threadIdx.x >> 6);      // 0 .. 16*dy-1
    if (row_id >= 16 * d.dy) return;
    const int c = row_id / d.dy, r = row_id - c * d.dy;
    const int iy = c >> 3, ix = c & 7;
    const size_t base = (size_t)(iy * d.dy + r) * d.nx_raw + (size_t)ix * d.dx + d.vos_x0;
    const float g = gain.v[c];
    double v[VOS_VMAX];
    bool ok[VOS_VMAX];      // still inside the running clip
    bool valid[VOS_VMAX];   // finite and != mask_value(0)
#pragma unroll
    for (int j = 0; j < VOS_VMAX; j++) {
        int col = lane + 64 * j;
        valid[j] = false; v[j] = 0.0;
        if (col < d.vos_w) {
            float f = raw_load<RAW_T>(raw, base + col);
            if (RAW_T == BBX_RAW_F32 && !isfinite(f)) f = 0.f;   // scrub, blackbox.py:1461-1468
            f = f * g;                                          // gain_corr, float32 multiply
            v[j] = (double)f;
            valid[j] = isfinite(f) && !(fabs(v[j]) <= 1e-8);
        }
        ok[j] = valid[j];
    }
    double lo = __longlong_as_double(0x7ff8000000000000LL), hi = lo;   // NaN until computed
    int n = 0;
#pragma unroll
    for (int j = 0; j < VOS_VMAX; j++) n += ok[j] ? 1 : 0;
    n = wave_sum_i32(n);
    for (int it = 0; it < 5 && n > 0; it++) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < VOS_VMAX; j++) if (ok[j]) s += v[j];
        s = wave_sum_f64(s);
        const double mean = s / (double)n;
        double q = 0.0;
#pragma unroll
        for (int j = 0; j < VOS_VMAX; j++) if (ok[j]) { double t = mean - v[j]; q += t * t; }
        q = wave_sum_f64(q);
        const double sd = sqrt(q / (double)n);
        lo = mean - 3.0 * sd;
        hi = mean + 3.0 * sd;
        int m = 0;
#pragma unroll
        for (int j = 0; j < VOS_VMAX; j++) { ok[j] = ok[j] && v[j] >= lo && v[j] <= hi; m += ok[j] ? 1 : 0; }
        m = wave_sum_i32(m);
        if (m == n) break;
        n = m;
    }
    // final: every valid value inside the last bounds (NaN bounds reject nothing
    // -- only reachable with n == 0, where the mean is NaN anyway)
    double s = 0.0; int cnt = 0;
#pragma unroll
    for (int j = 0; j < VOS_VMAX; j++) {
        bool keep = valid[j] && !(v[j] < lo) && !(v[j] > hi);
        if (keep) { s += v[j]; cnt++; }
    }
    s = wave_sum_f64(s); cnt = wave_sum_i32(cnt);
    if (lane == 0) mean_out[row_id] = s / (double)cnt;            // 0/0 -> NaN like nanmean of all-NaN
}

// The same statistics with 16 lanes per row (4 rows per wave): the strip is 180 columns wide,
// so a 64-lane row leaves most lanes idle, and sums inside a 16-lane DPP row need no
// cross-row step.
#define VOS_V16 4       // up to 4 * VOS_V16 values per lane -> strips up to 256 columns wide
__device__ __forceinline__ double row16_sum_f64(double v) {
    v += dpp_mov_f64<BBX_DPP_ROR(1)>(v);
    v += dpp_mov_f64<BBX_DPP_ROR(2)>(v);
    v += dpp_mov_f64<BBX_DPP_ROR(4)>(v);
    v += dpp_mov_f64<BBX_DPP_ROR(8)>(v);
    return v;
}
__device__ __forceinline__ int row16_sum_i32(int v) {
    v += dpp_mov_i32<BBX_DPP_ROR(1)>(v);
    v += dpp_mov_i32<BBX_DPP_ROR(2)>(v);
    v += dpp_mov_i32<BBX_DPP_ROR(4)>(v);
    v += dpp_mov_i32<BBX_DPP_ROR(8)>(v);
    return v;
}

// NV = values per lane (16 NV >= strip width): the loop is ALU-bound (float64 sums, 5 clip rounds),
// so the production strip (174 columns) gets an 11-value instance instead of the generic 16.
template <int RAW_T, int NV>
__global__ __launch_bounds__(256) void k_vos_rowstats16(const void* __restrict__ raw, bbx_dims d, f32x16 gain,
                                                        double* __restrict__ mean_out) {
    const int l16 = threadIdx.x & 15;
    const int row_id = (blockIdx.x * 256 + threadIdx.x) >> 4;           // 0 .. 16*dy-1 (+ padding rows)
    const bool live = row_id < 16 * d.dy;
    const int rid = live ? row_id : 0;
    const int c = rid / d.dy, r = rid - c * d.dy;
    const int iy = c >> 3, ix = c & 7;
    const size_t base = (size_t)(iy * d.dy + r) * d.nx_raw + (size_t)ix * d.dx + d.vos_x0;
    const float g = gain.v[c];
    double v[NV];
    unsigned valid = 0;                       // bit per value: finite and != mask_value(0)
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int col = l16 + 16 * k;
        float x = 0.f;
        if (col < d.vos_w) x = raw_load<RAW_T>(raw, base + col);
        if (RAW_T == BBX_RAW_F32 && !isfinite(x)) x = 0.f;             // scrub, blackbox.py:1461-1468
        x = x * g;                                                     // gain_corr, float32 multiply
        v[k] = (double)x;
        if (col < d.vos_w && isfinite(x) && !(fabs((double)x) <= 1e-8)) valid |= 1u << k;
    }
    unsigned ok = valid;                                               // still inside the running clip
    double lo = __longlong_as_double(0x7ff8000000000000LL), hi = lo;   // NaN until computed
    int n = row16_sum_i32(__popc(ok));
    bool run = n > 0;
    for (int it = 0; it < 5; it++) {                                   // rows that are done keep their state
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NV; k++) if (ok & (1u << k)) s += v[k];
        s = row16_sum_f64(s);
        const double mean = s / (double)n;
        double q2 = 0.0;
#pragma unroll
        for (int k = 0; k < NV; k++) if (ok & (1u << k)) { const double t = mean - v[k]; q2 += t * t; }
        q2 = row16_sum_f64(q2);
        const double sd = sqrt(q2 / (double)n);
        const double nlo = mean - 3.0 * sd, nhi = mean + 3.0 * sd;
        unsigned nok = 0;
#pragma unroll
        for (int k = 0; k < NV; k++) if ((ok & (1u << k)) && v[k] >= nlo && v[k] <= nhi) nok |= 1u << k;
        const int m = row16_sum_i32(__popc(nok));
        if (run) { lo = nlo; hi = nhi; ok = nok; }
        if (run && m == n) run = false;
        if (run) { n = m; if (n == 0) run = false; }
        if (!__any(run)) break;                                        // every row of the wave has converged
    }
    // final: every valid value inside the last bounds (NaN bounds reject nothing -- only
    // reachable with n == 0, where the mean is NaN anyway)
    double s = 0.0; int cnt = 0;
#pragma unroll
    for (int k = 0; k < NV; k++)
        if ((valid & (1u << k)) && !(v[k] < lo) && !(v[k] > hi)) { s += v[k]; cnt++; }
    s = row16_sum_f64(s); cnt = row16_sum_i32(cnt);
    if (live && l16 == 0) mean_out[row_id] = s / (double)cnt;         // 0/0 -> NaN like nanmean of all-NaN
}

// gain-corrected copy of the horizontal overscan rows (os_sec_hori), all dx columns
template <int RAW_T>
__global__ __launch_bounds__(256) void k_hos_copy(const void* __restrict__ raw, bbx_dims d, f32x16 gain,
                                                  float* __restrict__ hos) {
    const int total = 16 * d.hos_rows * d.dx;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int x = i % d.dx; int t = i / d.dx; int r = t % d.hos_rows; int c = t / d.hos_rows;
        int iy = c >> 3, ix = c & 7;
        // os_sec_hori: lower row channels [dy-cut, dy), upper row channels [dy, dy+cut)
        int gy = (iy == 0) ? (d.dy - d.hos_rows + r) : (d.dy + r);
        float f = raw_load<RAW_T>(raw, (size_t)gy * d.nx_raw + (size_t)ix * d.dx + x);
        if (RAW_T == BBX_RAW_F32 && !isfinite(f)) f = 0.f;
        hos[i] = f * gain.v[c];
    }
}

__global__ __launch_bounds__(256) void k_count_nonfinite(const float* __restrict__ raw, size_t n,
                                                         unsigned long long* __restrict__ out) {
    long long cnt = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        cnt += isfinite(raw[i]) ? 0 : 1;
    cnt = wave_sum_i64(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, (unsigned long long)cnt);
}

// ---------------------------------------------------------------------------------
// read noise: clipped std of the fit-subtracted vertical overscan (os_corr 6572)
// ---------------------------------------------------------------------------------
struct vos_state {          // per channel
    double lo, hi;          // running intersection of the clip intervals
    double mean, std;       // statistics of the current survivors
    long long n;            // survivors counted by the last pass
    int frozen;             // clip loop finished
    int pad;
};
#define VSTD_BLOCKS 128     // partial-sum workgroups per channel

// Block reduction of (s1, s2, n) -> partial[c][b][3].  (The fold over the workgroups is a
// separate tiny kernel: a "last workgroup folds" scheme needs a device-scope fence per
// workgroup, and on this multi-XCD part each of those writes the L2 back.)
__device__ __forceinline__ void vos_pass_finish(double s1, double s2, long long n, int c, int b,
                                                double* __restrict__ partial) {
    __shared__ double sh1[4], sh2[4], shn[4];
    s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
    const double nd = wave_sum_f64((double)n);
    if ((threadIdx.x & 63) == 0) { sh1[threadIdx.x >> 6] = s1; sh2[threadIdx.x >> 6] = s2; shn[threadIdx.x >> 6] = nd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* p = partial + ((size_t)c * VSTD_BLOCKS + b) * 3;
        p[0] = (sh1[0] + sh1[1]) + (sh1[2] + sh1[3]);
        p[1] = (sh2[0] + sh2[1]) + (sh2[2] + sh2[3]);
        p[2] = (shn[0] + shn[1]) + (shn[2] + shn[3]);
    }
}

// Fold of a channel's partials in fixed order (lanes 0..VSTD_BLOCKS-1 of the workgroup) and the
// clip-state update.  Follows SigmaClip._sigmaclip_noaxis: bounds from the survivors' mean/std,
// survivors = survivors inside the closed interval, stop when nothing changed or after 5
// iterations; the pass after the last filter delivers the returned statistics.
// `pass` is the index of the pass that produced `partial`; every thread of the workgroup gets
// the same new state.  Needs blockDim.x >= VSTD_BLOCKS (a multiple of 64).
__device__ __forceinline__ vos_state vos_fold_update(vos_state s, const double* __restrict__ partial, int c, int pass) {
    __shared__ double sh[3][VSTD_BLOCKS / 64];
    if (s.frozen) return s;                                 // uniform over the workgroup
    if (threadIdx.x < VSTD_BLOCKS) {
        const double* p = partial + ((size_t)c * VSTD_BLOCKS + threadIdx.x) * 3;
        double a = wave_sum_f64(p[0]), q = wave_sum_f64(p[1]), m = wave_sum_f64(p[2]);
        if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = q; sh[2][threadIdx.x >> 6] = m; }
    }
    __syncthreads();
    double a = 0.0, q = 0.0, m = 0.0;
    for (int w = 0; w < VSTD_BLOCKS / 64; w++) { a += sh[0][w]; q += sh[1][w]; m += sh[2][w]; }
    const long long cnt = (long long)m;
    const double mean = a / m;
    double var = q / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const double sd = sqrt(var);
    const bool unchanged = (pass > 0 && cnt == s.n);
    s.mean = mean; s.std = sd; s.n = cnt;
    if (unchanged || pass >= 5 || cnt == 0) {
        s.frozen = 1;
    } else {
        const double lo = mean - 3.0 * sd, hi = mean + 3.0 * sd;
        if (lo > s.lo) s.lo = lo;
        if (hi < s.hi) s.hi = hi;
    }
    return s;
}

__device__ __forceinline__ vos_state vos_state_init() {
    vos_state s;
    s.lo = -__builtin_huge_val(); s.hi = __builtin_huge_val();
    s.mean = 0; s.std = 0; s.n = -1; s.frozen = 0; s.pad = 0;
    return s;
}

// state after the update that follows pass `pass`: st_prev = state before it (ignored for pass 0)
__device__ __forceinline__ vos_state vos_state_after(const vos_state* __restrict__ st_prev, const double* __restrict__ partial,
                                                     int c, int pass) {
    const vos_state s = pass == 0 ? vos_state_init() : st_prev[c];
    return vos_fold_update(s, partial, c, pass);
}

// closing launch: the update after pass 5 -> read noise per channel
__global__ __launch_bounds__(VSTD_BLOCKS) void k_vos_std_final(const vos_state* __restrict__ st_prev,
                                                               const double* __restrict__ partial, int pass,
                                                               double* __restrict__ std_out) {
    const vos_state s = vos_state_after(st_prev, partial, blockIdx.x, pass);
    if (threadIdx.x == 0) std_out[blockIdx.x] = s.std;
}

// pass 0: residuals of the vertical overscan after the column fit -> compact float32 strip
// [16][dy][vos_w] (read by the later passes with vector loads), plus the unclipped sums.
// One wave per row; workgroup b of a channel takes the rows b*4+w, +4*VSTD_BLOCKS, ...
template <int RAW_T>
__global__ __launch_bounds__(256) void k_vos_strip(const void* __restrict__ raw, bbx_dims d, f32x16 gain,
                                                   const double* __restrict__ vfit, f32x16 dlevel,
                                                   float* __restrict__ strip, double* __restrict__ partial) {
    const int c = blockIdx.y, b = blockIdx.x;
    const int iy = c >> 3, ix = c & 7;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float g = gain.v[c], dl = dlevel.v[c];
    double s1 = 0.0, s2 = 0.0; long long n = 0;
    for (int r = b * 4 + w; r < d.dy; r += 4 * VSTD_BLOCKS) {
        const double vf = vfit[c * d.dy + r];
        // rows shared with the horizontal overscan section also received `-= dlevel`
        // (os_sec_hori spans the full channel width, blackbox.py:6568)
        const bool in_hos = (iy == 0) ? (r >= d.dy - d.hos_rows) : (r < d.hos_rows);
        const size_t src = (size_t)(iy * d.dy + r) * d.nx_raw + (size_t)ix * d.dx + d.vos_x0;
        float* dst = strip + ((size_t)c * d.dy + r) * d.vos_w;
        for (int col = lane; col < d.vos_w; col += 64) {
            float f = raw_load<RAW_T>(raw, src + col);
            if (RAW_T == BBX_RAW_F32 && !isfinite(f)) f = 0.f;
            f = f * g;
            float x = (float)((double)f - vf);                   // float32 array -= float64 column
            if (in_hos) x = x - dl;
            dst[col] = x;
            const double xd = (double)x;
            if (isfinite(x) && !(fabs(xd) <= 1e-8)) { s1 += xd; s2 += xd * xd; n++; }
        }
    }
    vos_pass_finish(s1, s2, n, c, b, partial);
}

// passes 1..5 over the compact strip.  Each workgroup first derives the clip state from the
// previous state and the previous pass's partials (the same fold in every workgroup, so all
// agree bit for bit; workgroup 0 records it for the next launch) - state and partials ping-pong
// between two buffers, so no launch reads what a concurrent workgroup writes.  Channels whose
// clip loop has finished return at once.
__global__ __launch_bounds__(256) void k_vos_std_pass(const float* __restrict__ strip, bbx_dims d, int pass,
                                                      const vos_state* __restrict__ st_prev, vos_state* __restrict__ st_cur,
                                                      const double* __restrict__ part_prev, double* __restrict__ part_cur) {
    const int c = blockIdx.y, b = blockIdx.x;
    const vos_state s = vos_state_after(st_prev, part_prev, c, pass - 1);
    if (b == 0 && threadIdx.x == 0) st_cur[c] = s;
    if (s.frozen) return;
    const double lo = s.lo, hi = s.hi;
    const size_t total = (size_t)d.dy * d.vos_w;
    const float* x = strip + (size_t)c * total;
    double s1 = 0.0, s2 = 0.0; long long n = 0;
    auto take = [&](float v) {
        const double xd = (double)v;
        if (isfinite(v) && !(fabs(xd) <= 1e-8) && xd >= lo && xd <= hi) { s1 += xd; s2 += xd * xd; n++; }
    };
    if (total % 4 == 0) {
        const float4* x4 = (const float4*)x;            // strip and channel offsets are 16-byte aligned
        for (size_t i = (size_t)b * 256 + threadIdx.x; i < total / 4; i += (size_t)VSTD_BLOCKS * 256) {
            const float4 v = x4[i];
            take(v.x); take(v.y); take(v.z); take(v.w);
        }
    } else {
        for (size_t i = (size_t)b * 256 + threadIdx.x; i < total; i += (size_t)VSTD_BLOCKS * 256) take(x[i]);
    }
    vos_pass_finish(s1, s2, n, c, b, part_cur);
}

// ---------------------------------------------------------------------------------
// BlackGEM: per-column counts of near-saturated pixels (os_corr 6624-6640)
// ---------------------------------------------------------------------------------
template <int RAW_T>
__global__ __launch_bounds__(256) void k_satcol(const void* __restrict__ raw, bbx_dims d, f32x16 gain,
                                                const double* __restrict__ vfit, f32x16 thr, int rows1,
                                                int rows2, int* __restrict__ counts) {
    const int c = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= d.xsz) return;
    const int iy = c >> 3, ix = c & 7;
    const int k0 = blockIdx.y * 64;                 // distance from the overscan edge
    int n1 = 0, n2 = 0;
    for (int k = k0; k < k0 + 64 && k < rows2; k++) {
        // data-section row: upper channels count from row 0, lower from the top row down
        int y = (iy == 1) ? k : (d.ysz - 1 - k);
        int rl = (iy == 0) ? y : (d.os_y + y);      // channel-local row
        float f = raw_load<RAW_T>(raw, (size_t)(iy * d.dy + rl) * d.nx_raw + (size_t)ix * d.dx + x);
        if (RAW_T == BBX_RAW_F32 && !isfinite(f)) f = 0.f;
        f = f * gain.v[c];
        float v = (float)((double)f - vfit[c * d.dy + rl]);
        if (v >= thr.v[c]) { n2++; if (k < rows1) n1++; }
    }
    if (n1) atomicAdd(&counts[(0 * 16 + c) * d.xsz + x], n1);
    if (n2) atomicAdd(&counts[(1 * 16 + c) * d.xsz + x], n2);
}

static f32x16 load16(const float* h) { f32x16 r; for (int i = 0; i < 16; i++) r.v[i] = h[i]; return r; }

extern "C" {

int bbx_overscan_stats(bbx_ctx* ctx, const bbx_geom* g, const void* d_raw, int raw_type,
                       const float* h_gain, double* d_mean_vos_col, float* d_hos,
                       int64_t* d_n_infnan, void* stream) {
    if (!ctx || !d_raw || !h_gain || !d_mean_vos_col || !d_hos || !d_n_infnan) return BBX_ERR_ARG;
    bbx_dims d; int rc = bbx_make_dims(g, &d); if (rc) return rc;
    if (d.vos_w > 64 * VOS_VMAX) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    f32x16 gain = load16(h_gain);
    const int rows = 16 * d.dy;
    dim3 grid((rows + 3) / 4), block(256);
    BBX_HIP(hipMemsetAsync(d_n_infnan, 0, sizeof(int64_t), s));
    // 16-lane variant for strips up to 256 columns
    const bool v16 = d.vos_w <= 64 * VOS_V16;
    const dim3 grid16((rows + 15) / 16);
    if (raw_type == BBX_RAW_U16) {
        if (v16 && d.vos_w <= 176) hipLaunchKernelGGL((k_vos_rowstats16<BBX_RAW_U16, 11>), grid16, block, 0, s, d_raw, d, gain, d_mean_vos_col);
        else if (v16) hipLaunchKernelGGL((k_vos_rowstats16<BBX_RAW_U16, 4 * VOS_V16>), grid16, block, 0, s, d_raw, d, gain, d_mean_vos_col);
        else hipLaunchKernelGGL(k_vos_rowstats<BBX_RAW_U16>, grid, block, 0, s, d_raw, d, gain, d_mean_vos_col);
        hipLaunchKernelGGL(k_hos_copy<BBX_RAW_U16>, dim3(256), block, 0, s, d_raw, d, gain, d_hos);
    } else if (raw_type == BBX_RAW_F32) {
        if (v16 && d.vos_w <= 176) hipLaunchKernelGGL((k_vos_rowstats16<BBX_RAW_F32, 11>), grid16, block, 0, s, d_raw, d, gain, d_mean_vos_col);
        else if (v16) hipLaunchKernelGGL((k_vos_rowstats16<BBX_RAW_F32, 4 * VOS_V16>), grid16, block, 0, s, d_raw, d, gain, d_mean_vos_col);
        else hipLaunchKernelGGL(k_vos_rowstats<BBX_RAW_F32>, grid, block, 0, s, d_raw, d, gain, d_mean_vos_col);
        hipLaunchKernelGGL(k_hos_copy<BBX_RAW_F32>, dim3(256), block, 0, s, d_raw, d, gain, d_hos);
        hipLaunchKernelGGL(k_count_nonfinite, dim3(2048), block, 0, s, (const float*)d_raw,
                           (size_t)d.ny_raw * d.nx_raw, (unsigned long long*)d_n_infnan);
    } else return BBX_ERR_ARG;
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_vos_std(bbx_ctx* ctx, const bbx_geom* g, const void* d_raw, int raw_type, const float* h_gain,
                const double* d_vfit, const float* h_dlevel, double* d_std_vos, void* stream) {
    if (!ctx || !d_raw || !h_gain || !d_vfit || !h_dlevel || !d_std_vos) return BBX_ERR_ARG;
    bbx_dims d; int rc = bbx_make_dims(g, &d); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    f32x16 gain = load16(h_gain), dlev = load16(h_dlevel);
    const size_t nstrip = (size_t)16 * d.dy * d.vos_w;
    // [2] clip states, [2] partial-sum sets (ping-pong by pass parity), then the strip
    const size_t npart = (size_t)16 * VSTD_BLOCKS * 3;
    const size_t o_partial = 2 * 16 * sizeof(vos_state), o_strip = o_partial + 2 * npart * sizeof(double);
    char* ws = (char*)bbx_ws(ctx, WS_STRIP, o_strip + nstrip * sizeof(float), &rc);
    if (rc) return rc;
    vos_state* st = (vos_state*)ws;
    double* partial = (double*)(ws + o_partial);
    float* strip = (float*)(ws + o_strip);
    if (raw_type == BBX_RAW_U16)
        hipLaunchKernelGGL(k_vos_strip<BBX_RAW_U16>, dim3(VSTD_BLOCKS, 16), dim3(256), 0, s, d_raw, d, gain, d_vfit, dlev,
                           strip, partial);
    else if (raw_type == BBX_RAW_F32)
        hipLaunchKernelGGL(k_vos_strip<BBX_RAW_F32>, dim3(VSTD_BLOCKS, 16), dim3(256), 0, s, d_raw, d, gain, d_vfit, dlev,
                           strip, partial);
    else return BBX_ERR_ARG;
    // pass p writes state p-1 into st[(p-1)&1] and its sums into partial[p&1]
    for (int pass = 1; pass < 6; pass++)
        hipLaunchKernelGGL(k_vos_std_pass, dim3(VSTD_BLOCKS, 16), dim3(256), 0, s, strip, d, pass,
                           st + 16 * (pass & 1), st + 16 * ((pass - 1) & 1),
                           partial + npart * ((pass - 1) & 1), partial + npart * (pass & 1));
    hipLaunchKernelGGL(k_vos_std_final, dim3(16), dim3(VSTD_BLOCKS), 0, s, st + 16 * (4 & 1), partial + npart * (5 & 1), 5,
                       d_std_vos);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_satcol_counts(bbx_ctx* ctx, const bbx_geom* g, const void* d_raw, int raw_type, const float* h_gain,
                      const double* d_vfit, const float* h_thr, int rows1, int rows2, int32_t* d_counts,
                      void* stream) {
    if (!ctx || !d_raw || !h_gain || !d_vfit || !h_thr || !d_counts) return BBX_ERR_ARG;
    bbx_dims d; int rc = bbx_make_dims(g, &d); if (rc) return rc;
    if (rows1 <= 0 || rows2 < rows1 || rows2 > d.ysz) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    f32x16 gain = load16(h_gain), thr = load16(h_thr);
    BBX_HIP(hipMemsetAsync(d_counts, 0, (size_t)2 * 16 * d.xsz * sizeof(int32_t), s));
    dim3 grid((d.xsz + 255) / 256, (rows2 + 63) / 64, 16);
    if (raw_type == BBX_RAW_U16)
        hipLaunchKernelGGL(k_satcol<BBX_RAW_U16>, grid, dim3(256), 0, s, d_raw, d, gain, d_vfit, thr, rows1, rows2, d_counts);
    else if (raw_type == BBX_RAW_F32)
        hipLaunchKernelGGL(k_satcol<BBX_RAW_F32>, grid, dim3(256), 0, s, d_raw, d, gain, d_vfit, thr, rows1, rows2, d_counts);
    else return BBX_ERR_ARG;
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

}  // extern "C"
