// bbx_coadd.hip -- reference co-add: what buildref.py does per
// input image before SWarp (prep_inputimages 2442-2777), SWarp's LANCZOS3 resampling onto the
// output frame (buildref.py:1727-1763) and its pixel combination (-COMBINE_TYPE, 1733/1815).
//
// All three are HBM-streaming passes:
//   k_coadd_prep     data, background, background-sigma, mask in; data, weights out (float4)
//   k_resample_l3    one output pixel per thread, 6x6 taps gathered through L1/L2 (adjacent
//                    output pixels share 30 of their 36 taps), coordinates from a coarse grid
//   k_combine        one output pixel per thread over the n resampled planes, float64 sums in
//                    image order; order statistics by rank counting in registers (n <= 32)
#include "bbx_common.h"

// ---------------------------------------------------------------------------------
// prep_inputimages: data -= bkg; data[mask == edge] = 0; weights = 1/bkg_std^2 (0 where
// bkg_std == 0 or the mask holds a discarded type)       buildref.py:2602-2624, 2709-2733
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void prep_one(float& d, float b, float s, unsigned m, int discard, int edge, int has_bkg, float& w) {
    if (has_bkg) d = d - b;
    if ((int)m == edge) d = 0.f;
    w = (s != 0.f) ? 1.0f / (s * s) : 0.f;
    if (m & (unsigned)discard) w = 0.f;
}

__global__ __launch_bounds__(256) void k_coadd_prep(float* __restrict__ data, const float* __restrict__ bkg,
                                                    const float* __restrict__ bstd, const uint8_t* __restrict__ mask,
                                                    size_t npix, int discard, int edge, float* __restrict__ wout) {
    const size_t n4 = npix / 4;
    const int has_bkg = bkg != nullptr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 d = ((const float4*)data)[i];
        const float4 s = ((const float4*)bstd)[i];
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_bkg) b = ((const float4*)bkg)[i];
        const uint32_t m = ((const uint32_t*)mask)[i];
        float4 w;
        prep_one(d.x, b.x, s.x, m & 255u, discard, edge, has_bkg, w.x);
        prep_one(d.y, b.y, s.y, (m >> 8) & 255u, discard, edge, has_bkg, w.y);
        prep_one(d.z, b.z, s.z, (m >> 16) & 255u, discard, edge, has_bkg, w.z);
        prep_one(d.w, b.w, s.w, m >> 24, discard, edge, has_bkg, w.w);
        ((float4*)data)[i] = d;
        ((float4*)wout)[i] = w;
    }
    // tail (npix not a multiple of 4)
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        float d = data[i], w;
        prep_one(d, has_bkg ? bkg[i] : 0.f, bstd[i], mask[i], discard, edge, has_bkg, w);
        data[i] = d; wout[i] = w;
    }
}

// ---------------------------------------------------------------------------------
// LANCZOS3 resampling
// ---------------------------------------------------------------------------------
// taps ix-2 .. ix+3 for a position ix + f (0 <= f < 1): k(t) = sinc(t) sinc(t/3), t = f - i,
// normalised to unit sum (oracle/coadd.py lanczos3_taps).  sin(pi t) = +-sin(pi f) and
// sin(pi t/3) = sin(pi f/3 - i pi/3) by the addition theorem: three sinpi/cospi calls per axis
// instead of twelve sines.  Single precision like SWarp's kernels: the taps agree with the
// float64 oracle to a few 1e-7, which is the tolerance of the resampling tests.
__device__ __forceinline__ void l3_taps(float f, float* k) {
    // sin and cos of y = pi f / 3 (0 <= y < 1.05) from their Taylor polynomials (through y^11 / y^12:
    // below float32 resolution on this interval); sin(pi f) follows from the triple-angle identity
    const float y = f * 1.04719755119659774615f, y2 = y * y;
    float ps = -2.50521083854417187751e-8f;                   // -1/11!
    ps = __builtin_fmaf(ps, y2, 2.75573192239858906526e-6f);  //  1/9!
    ps = __builtin_fmaf(ps, y2, -1.98412698412698412698e-4f); // -1/7!
    ps = __builtin_fmaf(ps, y2, 8.33333333333333333333e-3f);  //  1/5!
    ps = __builtin_fmaf(ps, y2, -1.66666666666666666667e-1f); // -1/3!
    const float s3 = __builtin_fmaf(ps * y2, y, y);
    float pc = 2.08767569878680989792e-9f;                    //  1/12!
    pc = __builtin_fmaf(pc, y2, -2.75573192239858906526e-7f); // -1/10!
    pc = __builtin_fmaf(pc, y2, 2.48015873015873015873e-5f);  //  1/8!
    pc = __builtin_fmaf(pc, y2, -1.38888888888888888889e-3f); // -1/6!
    pc = __builtin_fmaf(pc, y2, 4.16666666666666666667e-2f);  //  1/4!
    pc = __builtin_fmaf(pc, y2, -0.5f);
    const float c3 = __builtin_fmaf(pc, y2, 1.0f);
    const float s1 = s3 * (3.0f - 4.0f * s3 * s3);
    const float H = 0.86602540378443864676f;                 // sin(pi/3)
    const float ci[6] = {-0.5f, 0.5f, 1.0f, 0.5f, -0.5f, -1.0f};   // cos(i pi/3), i = -2..3
    const float si[6] = {-H, -H, 0.0f, H, H, 0.0f};                // sin(i pi/3)
    float v[6], sum = 0.f;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const int i = q - 2;
        const float t = f - (float)i;
        const float sa = (i & 1) ? -s1 : s1;                  // sin(pi t)
        const float sb = s3 * ci[q] - c3 * si[q];             // sin(pi t / 3)
        // (hardware reciprocal, 1 ulp: the taps are normalised afterwards)
        const float val = (t == 0.f) ? 1.0f : (3.0f * sa * sb) * __builtin_amdgcn_rcpf((float)(M_PI * M_PI) * (t * t));
        v[q] = val; sum += val;
    }
    const float rs = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int q = 0; q < 6; q++) k[q] = v[q] * rs;
}

#define RS_BIG 1e30f
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct rs_args {
    const float* in; const float* win;
    int in_ny, in_nx, out_ny, out_nx;
    const double* grid; int gny, gnx, gstep;
    float fscale;
    float* out; float* wout;
};

// input position of output pixel (X, Y): bilinear interpolation (float64) of the coarse grid
__device__ __forceinline__ void rs_position(const rs_args& a, int X, int Y, double* pos) {
    const int gj = Y / a.gstep, gi = X / a.gstep;
    const double fy = (double)(Y - gj * a.gstep) / (double)a.gstep, fx = (double)(X - gi * a.gstep) / (double)a.gstep;
    const double* g00 = a.grid + ((size_t)gj * a.gnx + gi) * 2;
    const double* g10 = g00 + (size_t)a.gnx * 2;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const double p = g00[k] + (g00[2 + k] - g00[k]) * fx;
        const double q = g10[k] + (g10[2 + k] - g10[k]) * fx;
        pos[k] = p + (q - p) * fy;
    }
}

// Output tile of RT_W x RT_H pixels per workgroup.  The input pixels its 6x6 footprints touch
// form a box a little larger than the tile (same pixel scale, small rotation): the workgroup
// stages that box in LDS once -- data and variance, the reciprocal taken once per input pixel
// instead of 36 times -- and every output pixel reads its 36 taps from there.  A box that does
// not fit (strong rotation / magnification) falls back to gathering from global memory.
#define RT_W 64
#define RT_H 16
#define RT_LW 96
#define RT_LH 48
__global__ __launch_bounds__(256) void k_resample_l3(rs_args a) {
    __shared__ float2 tile[RT_LW * RT_LH];                     // (data, variance)
    __shared__ int red[4][4];
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6, wave = threadIdx.x >> 6;
    const int X = blockIdx.x * RT_W + lx;
    double px[4], py[4];
    int fxi[4], fyi[4];
    bool ok[4];                                                // footprint inside the input image
    int mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -0x7fffffff, mxy = -0x7fffffff;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int Y = blockIdx.y * RT_H + ly + 4 * k;
        ok[k] = false; fxi[k] = 0; fyi[k] = 0; px[k] = 0.0; py[k] = 0.0;
        if (X < a.out_nx && Y < a.out_ny) {
            double pos[2];
            rs_position(a, X, Y, pos);
            const double xf = floor(pos[0]), yf = floor(pos[1]);
            // (the comparison also rejects NaN positions)
            if (xf - 2.0 >= 0.0 && xf + 3.0 < (double)a.in_nx && yf - 2.0 >= 0.0 && yf + 3.0 < (double)a.in_ny) {
                ok[k] = true; fxi[k] = (int)xf; fyi[k] = (int)yf; px[k] = pos[0] - xf; py[k] = pos[1] - yf;
                mnx = min(mnx, fxi[k]); mxx = max(mxx, fxi[k]); mny = min(mny, fyi[k]); mxy = max(mxy, fyi[k]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, o, 64)); mny = min(mny, __shfl_xor(mny, o, 64));
        mxx = max(mxx, __shfl_xor(mxx, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64));
    }
    if (lx == 0) { red[wave][0] = mnx; red[wave][1] = mny; red[wave][2] = mxx; red[wave][3] = mxy; }
    __syncthreads();
    mnx = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
    mny = min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1]));
    mxx = max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2]));
    mxy = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
    const bool any = mxx >= mnx;
    const int x0 = mnx - 2, y0 = mny - 2;
    const int W = any ? mxx - mnx + 6 : 0, Hh = any ? mxy - mny + 6 : 0;
    const bool staged = any && W <= RT_LW && Hh <= RT_LH;      // workgroup-uniform
    if (staged) {
        // every pixel of the box lies inside the image (each contributing footprint does)
        for (int idx = threadIdx.x; idx < W * Hh; idx += 256) {
            const int r = idx / W, c = idx - r * W;
            const size_t g = (size_t)(y0 + r) * a.in_nx + (x0 + c);
            const float f = a.in[g], w = a.win[g];
            tile[r * RT_LW + c] = make_float2(f, (w > 0.f) ? 1.0f / w : __builtin_huge_valf());   // inf poisons the sum
        }
    }
    __syncthreads();
    const float fs = a.fscale;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int Y = blockIdx.y * RT_H + ly + 4 * k;
        if (!(X < a.out_nx && Y < a.out_ny)) continue;
        const size_t o = (size_t)Y * a.out_nx + X;
        if (!ok[k]) { a.out[o] = 0.f; a.wout[o] = 0.f; continue; }
        float kx[6], ky[6];
        l3_taps((float)px[k], kx);
        l3_taps((float)py[k], ky);
        float acc = 0.f, vacc = 0.f;
        bool bad = false;
        if (staged) {
            const float2* p = tile + (fyi[k] - 2 - y0) * RT_LW + (fxi[k] - 2 - x0);
            // (data, variance) pairs through packed fused multiply-adds
            // a zero-weight input pixel carries an infinite variance: any tap touching it (even
            // with a zero kernel value: 0 * inf = NaN) leaves the interpolated variance non-finite
            f32x2 acc2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 6; j++) {
                f32x2 row2 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const float2 fv = p[i];
                    const f32x2 v2 = {fv.x, fv.y}, k2 = {kx[i], kx[i]};
                    row2 = __builtin_elementwise_fma(k2, v2, row2);
                }
                const f32x2 ky2 = {ky[j], ky[j]};
                acc2 = __builtin_elementwise_fma(ky2, row2, acc2);
                p += RT_LW;
            }
            acc = acc2.x; vacc = acc2.y;
            bad = !(fabsf(vacc) < __builtin_huge_valf());
        } else {
            const float* p = a.in + (size_t)(fyi[k] - 2) * a.in_nx + (fxi[k] - 2);
            const float* pw = a.win + (size_t)(fyi[k] - 2) * a.in_nx + (fxi[k] - 2);
#pragma unroll
            for (int j = 0; j < 6; j++) {
                float row = 0.f, vrow = 0.f;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const float f = p[i], w = pw[i];
                    const float v = (w > 0.f) ? 1.0f / w : RS_BIG;
                    bad |= !(w > 0.f);
                    row = row + kx[i] * f;
                    vrow = vrow + kx[i] * v;
                }
                acc = acc + ky[j] * row;
                vacc = vacc + ky[j] * vrow;
                p += a.in_nx; pw += a.in_nx;
            }
        }
        const float vout = (fs * fs) * vacc;
        a.out[o] = fs * acc;
        a.wout[o] = (!bad && vout > 0.f) ? 1.0f / vout : 0.f;
    }
}

// ---------------------------------------------------------------------------------
// combination of n resampled planes
// ---------------------------------------------------------------------------------
enum { CB_WEIGHTED = 0, CB_AVERAGE = 1, CB_MEDIAN = 2, CB_CLIPPED = 3, CB_MIN = 4, CB_MAX = 5, CB_SUM = 6 };

struct cb_args {
    const float* cube; const float* wcube;
    long long stride; size_t npix; int n;
    float clip_sigma, clip_ampfrac;
    float* out; float* wout;
    uint8_t* clipmask;               // [n][npix] or null
    float* nsigma;                   // [n][npix] or null: deviation of the dropped pixels in sigma
    unsigned long long* nclip;       // [n] or null
};

// ascending bitonic sort of CAP (power of two) register values: fully unrolled compare-exchange
// network, no data-dependent indexing (invalid entries are +inf and sort last)
template <int CAP>
__device__ __forceinline__ void sort_regs(float (&v)[CAP]) {
#pragma unroll
    for (int k = 2; k <= CAP; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < CAP; i++) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const float a = v[i], b = v[l];
                    const float lo = fminf(a, b), hi = fmaxf(a, b);
                    v[i] = up ? lo : hi;
                    v[l] = up ? hi : lo;
                }
            }
        }
    }
}
// v[r] for a run-time r (select chain)
template <int CAP>
__device__ __forceinline__ float pick_reg(const float (&v)[CAP], int r) {
    float res = v[0];
#pragma unroll
    for (int i = 1; i < CAP; i++) res = (i == r) ? v[i] : res;
    return res;
}

template <int TYPE, int CAP>
__global__ __launch_bounds__(256) void k_combine(cb_args a) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < a.npix;
    float f[CAP], w[CAP];
    unsigned valid = 0;
#pragma unroll
    for (int i = 0; i < CAP; i++) {
        f[i] = 0.f; w[i] = 0.f;
        if (live && i < a.n) {
            f[i] = a.cube[(size_t)i * a.stride + p];
            w[i] = a.wcube[(size_t)i * a.stride + p];
            if (w[i] > 0.f) valid |= 1u << i;
        }
    }
    const int m = __popc(valid);
    double out = 0.0, wout = 0.0;
    unsigned drop = 0;
    float med_keep = 0.f;
    if (m > 0) {
        double sw = 0.0, swf = 0.0, sf = 0.0, sinv = 0.0;
        const bool need_inv = (TYPE == CB_AVERAGE || TYPE == CB_SUM || TYPE == CB_MEDIAN);
        const bool need_w = (TYPE == CB_WEIGHTED || TYPE == CB_CLIPPED);
#pragma unroll
        for (int i = 0; i < CAP; i++)
            if (valid >> i & 1u) {
                const double wd = (double)w[i], fd = (double)f[i];
                if (need_w) { sw += wd; swf += wd * fd; }
                if (TYPE == CB_AVERAGE || TYPE == CB_SUM) sf += fd;
                if (need_inv) sinv += 1.0 / wd;
            }
        if (TYPE == CB_WEIGHTED) { out = swf / sw; wout = sw; }
        else if (TYPE == CB_AVERAGE) { out = sf / (double)m; wout = (double)m * (double)m / sinv; }
        else if (TYPE == CB_SUM) { out = sf; wout = 1.0 / sinv; }
        else if (TYPE == CB_MIN || TYPE == CB_MAX) {
            bool first = true;
#pragma unroll
            for (int i = 0; i < CAP; i++)
                if (valid >> i & 1u) {
                    const bool better = first || (TYPE == CB_MIN ? (double)f[i] < out : (double)f[i] > out);
                    if (better) { out = (double)f[i]; wout = (double)w[i]; first = false; }
                }
        } else {
            float srt[CAP];
#pragma unroll
            for (int i = 0; i < CAP; i++) srt[i] = (valid >> i & 1u) ? f[i] : __builtin_huge_valf();
            sort_regs<CAP>(srt);
            const double lo = (double)pick_reg<CAP>(srt, (m - 1) / 2);
            const double hi = (double)pick_reg<CAP>(srt, m / 2);
            const double med = 0.5 * (lo + hi);
            if (TYPE == CB_MEDIAN) { out = med; wout = (2.0 / M_PI) * (double)m * (double)m / sinv; }
            else {
                // the clip test runs in float32 (SWarp's pixel type): |f - med| > sigma*sqrt(1/w) + A*|med|
                const float med32 = (float)med, amed = a.clip_ampfrac * fabsf(med32);
                med_keep = med32;
                double sw2 = 0.0, swf2 = 0.0;
                int nk = 0;
#pragma unroll
                for (int i = 0; i < CAP; i++)
                    if (valid >> i & 1u) {
                        const float thr = a.clip_sigma * sqrtf(1.0f / w[i]) + amed;
                        const bool d = fabsf(f[i] - med32) > thr;
                        if (d) drop |= 1u << i;
                        else { const double wd = (double)w[i]; sw2 += wd; swf2 += wd * (double)f[i]; nk++; }
                    }
                if (nk > 0) { out = swf2 / sw2; wout = sw2; }
                else { out = swf / sw; wout = sw; drop = 0; }
            }
        }
    }
    if (live) { a.out[p] = (float)out; a.wout[p] = (float)wout; }
    if (TYPE == CB_CLIPPED) {
#pragma unroll
        for (int i = 0; i < CAP; i++) {
            if (i < a.n) {                                      // uniform
                const bool d = live && (drop >> i & 1u);
                if (a.clipmask && live) a.clipmask[(size_t)i * a.npix + p] = d ? 1 : 0;
                if (a.nsigma && live) a.nsigma[(size_t)i * a.npix + p] = d ? (f[i] - med_keep) / sqrtf(1.0f / w[i]) : 0.f;
                if (a.nclip) {
                    const unsigned long long bal = __ballot(d);
                    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&a.nclip[i], (unsigned long long)__popcll(bal));
                }
            }
        }
    }
}

template <int TYPE>
static void launch_combine(const cb_args& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.npix + 255) / 256)), block(256);
    if (a.n <= 4) hipLaunchKernelGGL((k_combine<TYPE, 4>), grid, block, 0, s, a);
    else if (a.n <= 8) hipLaunchKernelGGL((k_combine<TYPE, 8>), grid, block, 0, s, a);
    else if (a.n <= 16) hipLaunchKernelGGL((k_combine<TYPE, 16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_combine<TYPE, 32>), grid, block, 0, s, a);
}

// ---------------------------------------------------------------------------------
// clipped pixels of the first CLIPPED pass -> mask in the frame of an input image
// (buildref.py clipped2mask_loop 3686-3783, pass_filters 3784-3873)
// ---------------------------------------------------------------------------------
struct c2m_point { uint32_t xy; float nsigma; };          // (y0 << 16) | x0, 0-based input pixel

__global__ __launch_bounds__(256) void k_c2m_points(rs_args g, const uint8_t* __restrict__ clip, const float* __restrict__ nsig,
                                                    float min_sigma, c2m_point* __restrict__ pts, unsigned* __restrict__ npts,
                                                    unsigned cap, int32_t* err) {
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    bool keep = false;
    c2m_point pt = {0u, 0.f};
    if (X < g.out_nx && Y < g.out_ny) {
        const size_t o = (size_t)Y * g.out_nx + X;
        if (clip[o] && fabsf(nsig[o]) > min_sigma) {
            double pos[2];
            rs_position(g, X, Y, pos);
            // (x_im + 0.5).astype(uint16) on 1-based positions; kept when 1 <= x <= xsize
            const double x1 = pos[0] + 1.0 + 0.5, y1 = pos[1] + 1.0 + 0.5;
            if (x1 >= 1.0 && y1 >= 1.0 && x1 < 60000.0 && y1 < 60000.0) {
                const int xi = (int)x1, yi = (int)y1;
                if (xi <= g.in_nx && yi <= g.in_ny) { keep = true; pt.xy = ((uint32_t)(yi - 1) << 16) | (uint32_t)(xi - 1); pt.nsigma = nsig[o]; }
            }
        }
    }
    const unsigned long long m = __ballot(keep);
    if (m) {
        const int leader = __ffsll((long long)m) - 1;
        unsigned base = 0;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(npts, (unsigned)__popcll(m));
        base = __shfl(base, leader, 64);
        const unsigned k = base + (unsigned)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
        if (keep) { if (k < cap) pts[k] = pt; else atomicOr(err, BBX_DERR_LIST_OVERFLOW); }
    }
}

__device__ __forceinline__ void add_u8(uint8_t* plane, size_t idx) {
    atomicAdd((unsigned*)(plane + (idx & ~(size_t)3)), 1u << (8 * (idx & 3)));      // (counts stay far below 256)
}

// mode 0: count the boxes of the selected points; 1: mark the back-boxes of box pixels whose count
// reached fmax; 2: clear the boxes; 3: fsize == 1, mark the pixels themselves
__global__ __launch_bounds__(256) void k_c2m_filter(const c2m_point* __restrict__ pts, const unsigned* __restrict__ npts, unsigned cap,
                                                    int ny, int nx, int fsize, float fsigma, int fmax, uint8_t* cnt0, uint8_t* cnt1,
                                                    uint8_t* mask_im, int mode) {
    const unsigned n = min(*npts, cap);
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const c2m_point pt = pts[k];
        if (!(fabsf(pt.nsigma) > fsigma)) continue;
        const int i0 = (int)(pt.xy & 0xffffu), j0 = (int)(pt.xy >> 16);
        if (mode == 3) { mask_im[(size_t)j0 * nx + i0] = 1; continue; }
        if (mode == 0 && mask_im[(size_t)j0 * nx + i0]) continue;          // masked by an earlier filter
        const int i1 = min(i0 + fsize, nx), j1 = min(j0 + fsize, ny);
        uint8_t* cnt = (pt.nsigma > 0.f) ? cnt1 : cnt0;
        for (int j = j0; j < j1; j++)
            for (int i = i0; i < i1; i++) {
                const size_t q = (size_t)j * nx + i;
                if (mode == 0) add_u8(cnt, q);
                else if (mode == 2) { cnt0[q] = 0; cnt1[q] = 0; }
                else if (cnt0[q] >= fmax || cnt1[q] >= fmax) {
                    const int ib = max(i + 1 - fsize, 0), jb = max(j + 1 - fsize, 0);
                    for (int jj = jb; jj <= j; jj++)
                        for (int ii = ib; ii <= i; ii++) mask_im[(size_t)jj * nx + ii] = 1;
                }
            }
    }
}

// masked pixels within sqrt(dist2) of a saturated pixel are released; the weights of the others go to zero
__global__ __launch_bounds__(256) void k_c2m_apply(uint8_t* __restrict__ mask_im, const uint8_t* __restrict__ data_mask, int ny, int nx,
                                                   int sat_bits, float dist2, float* __restrict__ weights,
                                                   unsigned long long* __restrict__ nmasked) {
    const int R = (int)floorf(sqrtf(dist2));
    unsigned long long cnt = 0;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)ny * nx; o += (size_t)gridDim.x * blockDim.x) {
        if (!mask_im[o]) continue;
        const int j = (int)(o / nx), i = (int)(o - (size_t)j * nx);
        bool near = false;
        for (int dj = -R; dj <= R && !near; dj++) {
            const int jj = j + dj;
            if (jj < 0 || jj >= ny) continue;
            for (int di = -R; di <= R; di++) {
                const int ii = i + di;
                if (ii < 0 || ii >= nx) continue;
                if ((float)(di * di + dj * dj) <= dist2 && (data_mask[(size_t)jj * nx + ii] & sat_bits)) { near = true; break; }
            }
        }
        if (near) mask_im[o] = 0;
        else { weights[o] = 0.f; cnt++; }
    }
    cnt = (unsigned long long)wave_sum_i64((long long)cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(nmasked, cnt);
}

extern "C" {

int bbx_clipped2mask(bbx_ctx* ctx, int out_ny, int out_nx, const uint8_t* d_clip, const float* d_nsigma, const double* d_grid,
                     int gny, int gnx, int gstep, int in_ny, int in_nx, const uint8_t* d_data_mask, int sat_bits,
                     float dist2_limit, int nfilt, const int* h_fsize, const float* h_fsigma, const int* h_fmax,
                     float* d_weights, uint8_t* d_mask_im, int64_t* d_nmasked, void* stream) {
    if (!ctx || !d_clip || !d_nsigma || !d_grid || !d_data_mask || !d_weights || !d_mask_im || !d_nmasked || !h_fsize || !h_fsigma || !h_fmax)
        return BBX_ERR_ARG;
    if (out_ny < 1 || out_nx < 1 || in_ny < 1 || in_nx < 1 || in_ny > 60000 || in_nx > 60000 || gstep < 1 || nfilt < 1 || nfilt > 8)
        return BBX_ERR_ARG;
    if ((out_ny - 1) / gstep + 1 >= gny || (out_nx - 1) / gstep + 1 >= gnx || !(dist2_limit >= 0.f) || dist2_limit > 1.0e4f) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t in_npix = (size_t)in_ny * in_nx, in_pad = (in_npix + 3) & ~(size_t)3;
    const size_t cap = (size_t)out_ny * out_nx / 8 + 4096;
    int rc;
    // 2 count planes | point count | points
    char* w = (char*)bbx_ws(ctx, WS_HIST, 2 * in_pad + 64 + cap * sizeof(c2m_point), &rc); if (rc) return rc;
    uint8_t* cnt0 = (uint8_t*)w; uint8_t* cnt1 = cnt0 + in_pad;
    unsigned* npts = (unsigned*)(w + 2 * in_pad);
    c2m_point* pts = (c2m_point*)(w + 2 * in_pad + 64);
    BBX_HIP(hipMemsetAsync(w, 0, 2 * in_pad + 64, s));
    BBX_HIP(hipMemsetAsync(d_mask_im, 0, in_npix, s));
    BBX_HIP(hipMemsetAsync(d_nmasked, 0, sizeof(int64_t), s));
    float min_sigma = h_fsigma[0];
    for (int k = 1; k < nfilt; k++) min_sigma = h_fsigma[k] < min_sigma ? h_fsigma[k] : min_sigma;
    for (int k = 0; k < nfilt; k++)
        if (h_fsize[k] < 1 || h_fsize[k] > 64 || h_fmax[k] < 1 || h_fmax[k] > 255) return BBX_ERR_ARG;
    rs_args g;
    memset(&g, 0, sizeof(g));
    g.in_ny = in_ny; g.in_nx = in_nx; g.out_ny = out_ny; g.out_nx = out_nx; g.grid = d_grid; g.gny = gny; g.gnx = gnx; g.gstep = gstep;
    hipLaunchKernelGGL(k_c2m_points, dim3((out_nx + 63) / 64, (out_ny + 3) / 4), dim3(256), 0, s, g, d_clip, d_nsigma, min_sigma, pts,
                       npts, (unsigned)cap, ctx->d_err);
    for (int k = 0; k < nfilt; k++) {
        if (h_fsize[k] == 1) {
            hipLaunchKernelGGL(k_c2m_filter, dim3(512), dim3(256), 0, s, pts, npts, (unsigned)cap, in_ny, in_nx, 1, h_fsigma[k], h_fmax[k],
                               cnt0, cnt1, d_mask_im, 3);
        } else {
            for (int mode = 0; mode < 3; mode++)
                hipLaunchKernelGGL(k_c2m_filter, dim3(512), dim3(256), 0, s, pts, npts, (unsigned)cap, in_ny, in_nx, h_fsize[k], h_fsigma[k],
                                   h_fmax[k], cnt0, cnt1, d_mask_im, mode);
        }
    }
    hipLaunchKernelGGL(k_c2m_apply, dim3(2048), dim3(256), 0, s, d_mask_im, d_data_mask, in_ny, in_nx, sat_bits & 255, dist2_limit,
                       d_weights, (unsigned long long*)d_nmasked);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_coadd_prep(bbx_ctx* ctx, int64_t npix, float* d_data, const float* d_bkg, const float* d_bkg_std,
                   const uint8_t* d_mask, int discard_bits, int edge_value, float* d_weights, void* stream) {
    if (!ctx || !d_data || !d_bkg_std || !d_mask || !d_weights || npix <= 0) return BBX_ERR_ARG;
    if (((uintptr_t)d_data | (uintptr_t)d_bkg_std | (uintptr_t)d_weights | (uintptr_t)d_bkg) % 16 || ((uintptr_t)d_mask) % 4)
        return BBX_ERR_ARG;
    hipLaunchKernelGGL(k_coadd_prep, dim3(4096), dim3(256), 0, (hipStream_t)stream, d_data, d_bkg, d_bkg_std, d_mask,
                       (size_t)npix, discard_bits & 255, edge_value, d_weights);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_resample_lanczos3(bbx_ctx* ctx, int in_ny, int in_nx, const float* d_in, const float* d_win, int out_ny,
                          int out_nx, const double* d_grid, int gny, int gnx, int gstep, float fscale, float* d_out,
                          float* d_wout, void* stream) {
    if (!ctx || !d_in || !d_win || !d_grid || !d_out || !d_wout) return BBX_ERR_ARG;
    if (in_ny < 6 || in_nx < 6 || out_ny < 1 || out_nx < 1 || gstep < 1) return BBX_ERR_ARG;
    // the grid must hold the node after the last pixel in both directions
    if ((out_ny - 1) / gstep + 1 >= gny || (out_nx - 1) / gstep + 1 >= gnx) return BBX_ERR_ARG;
    if ((size_t)in_ny * in_nx >= 0x7fffffffull * 4) return BBX_ERR_ARG;
    rs_args a;
    a.in = d_in; a.win = d_win; a.in_ny = in_ny; a.in_nx = in_nx; a.out_ny = out_ny; a.out_nx = out_nx;
    a.grid = d_grid; a.gny = gny; a.gnx = gnx; a.gstep = gstep; a.fscale = fscale; a.out = d_out; a.wout = d_wout;
    hipLaunchKernelGGL(k_resample_l3, dim3((out_nx + RT_W - 1) / RT_W, (out_ny + RT_H - 1) / RT_H), dim3(256), 0, (hipStream_t)stream, a);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_coadd_combine(bbx_ctx* ctx, int n, int64_t npix, const float* d_cube, const float* d_wcube, int64_t plane_stride,
                      int combine_type, float clip_sigma, float clip_ampfrac, float* d_out, float* d_wout,
                      uint8_t* d_clipmask, float* d_nsigma, int64_t* d_nclip, void* stream) {
    if (!ctx || !d_cube || !d_wcube || !d_out || !d_wout || n < 1 || n > 32 || npix <= 0 || plane_stride < npix)
        return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    cb_args a;
    a.cube = d_cube; a.wcube = d_wcube; a.stride = plane_stride; a.npix = (size_t)npix; a.n = n;
    a.clip_sigma = clip_sigma; a.clip_ampfrac = clip_ampfrac; a.out = d_out; a.wout = d_wout;
    a.clipmask = combine_type == CB_CLIPPED ? d_clipmask : nullptr;
    a.nsigma = combine_type == CB_CLIPPED ? d_nsigma : nullptr;
    a.nclip = combine_type == CB_CLIPPED ? (unsigned long long*)d_nclip : nullptr;
    if (d_nclip) BBX_HIP(hipMemsetAsync(d_nclip, 0, (size_t)n * sizeof(int64_t), s));
    switch (combine_type) {
        case CB_WEIGHTED: launch_combine<CB_WEIGHTED>(a, s); break;
        case CB_AVERAGE: launch_combine<CB_AVERAGE>(a, s); break;
        case CB_MEDIAN: launch_combine<CB_MEDIAN>(a, s); break;
        case CB_CLIPPED: launch_combine<CB_CLIPPED>(a, s); break;
        case CB_MIN: launch_combine<CB_MIN>(a, s); break;
        case CB_MAX: launch_combine<CB_MAX>(a, s); break;
        case CB_SUM: launch_combine<CB_SUM>(a, s); break;
        default: return BBX_ERR_ARG;
    }
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

}  // extern "C"
