// bbx_spline.h -- a mini (per-box) image read at frame pixels inside the kernels that need it (round 5), instead of a
// full-frame image made by bbx_spline_zoom, written to HBM and read back: zogy.mini2back = scipy.ndimage.zoom(mini, box,
// order 3, mode 'nearest') of every patch (one patch, or one per channel: interp_Xchan False).
//
//   output pixel o of a patch of n boxes (n * box pixels):  c = o (n - 1) / (n box - 1) + npad,  f = floor(c),  t = c - f
//   value = sum_a sum_b coef[f_y - 1 + a][f_x - 1 + b] w_a(t_y) w_b(t_x)        (coef: bbx_spline_prefilter of the padded patch)
//
// Here f and t come from integer arithmetic (o (n - 1) = f (n box - 1) + r, t = r / (n box - 1): exact where scipy's
// float64 c sits within 1e-13 of it).  The row weights are folded into the coefficients in float64 like bbx_spline_zoom
// does; along x the four folded values of an interval become the cubic p0 + t (p1 + t (p2 + t p3)), evaluated in float32
// per pixel: within 3e-7 (relative) of the float64 evaluation rounded to float32 that the zoom kernel writes.  Meant for
// smooth, positive maps (sigma images); a caller that needs the zoom's own bits uses bbx_spline_zoom.
#pragma once
#include "bbx_common.h"

struct bbx_spl {                 // device view of a bbx_spline_image over a frame of (nblky ph) x (nblkx pw) pixels
    const double* coef;
    int cnx;                     // coefficients per row: nblkx * px
    int ph, pw;                  // patch size in pixels (cy box, cx box)
    int py, px;                  // patch size in coefficients (cy + 2 npad, cx + 2 npad)
    int npad;
    int ny1, nx1;                // cy - 1, cx - 1
    int dy, dx;                  // ph - 1, pw - 1 (>= 1)
    float rdy, rdx, rph, rpw;    // 1 / dy, 1 / dx, 1 / ph, 1 / pw
    const float4* poly;          // optional: the cubics of every frame row on every coefficient column, [ny][cnx] (k_spl_polytable)
};

// host: the device view; BBX_ERR_ARG when the image does not tile a frame of ny x nx pixels or the integer arithmetic
// below would leave the range float32 converts exactly
static inline int bbx_spl_make(const bbx_spline_image* im, int ny, int nx, bbx_spl* o) {
    if (!im || !im->d_coef || im->nby < 1 || im->nbx < 1 || im->cy < 1 || im->cx < 1 || im->box < 1 || im->npad < 2) return BBX_ERR_ARG;
    if (im->nby % im->cy || im->nbx % im->cx || (int64_t)im->nby * im->box != ny || (int64_t)im->nbx * im->box != nx) return BBX_ERR_ARG;
    o->coef = im->d_coef;
    o->ph = im->cy * im->box; o->pw = im->cx * im->box;
    o->py = im->cy + 2 * im->npad; o->px = im->cx + 2 * im->npad;
    o->cnx = (im->nbx / im->cx) * o->px;
    o->npad = im->npad;
    o->ny1 = im->cy - 1; o->nx1 = im->cx - 1;
    o->dy = o->ph > 1 ? o->ph - 1 : 1; o->dx = o->pw > 1 ? o->pw - 1 : 1;
    if ((int64_t)o->ph * (o->ny1 + 1) >= (1 << 24) || (int64_t)o->pw * (o->nx1 + 1) >= (1 << 24) || ny >= (1 << 24) || nx >= (1 << 24)) return BBX_ERR_ARG;
    o->poly = nullptr;
    o->rdy = 1.0f / (float)o->dy; o->rdx = 1.0f / (float)o->dx; o->rph = 1.0f / (float)o->ph; o->rpw = 1.0f / (float)o->pw;
    return BBX_OK;
}

// q = n / d, r = n - q d for 0 <= n < 2^24, 1 <= d < 2^24: the float32 quotient is off by one at most
__device__ __forceinline__ void bbx_divmod24(int n, int d, float rd, int& q, int& r) {
    q = (int)((float)n * rd);
    r = n - q * d;
    if (r < 0) { q--; r += d; }
    if (r >= d) { q++; r -= d; }
}
// frame pixel X (or Y) -> coefficient index of tap 1 (f + npad, counted from the frame's first patch) and the remainder r
// of o (n - 1) over d = n box - 1 (t = r / d)
__device__ __forceinline__ void bbx_spl_axis(int X, int pw, float rpw, int px, int npad, int n1, int d, float rd, int& c, int& r) {
    int ix, ox;
    bbx_divmod24(X, pw, rpw, ix, ox);
    int f;
    bbx_divmod24(ox * n1, d, rd, f, r);
    c = ix * px + npad + f;
}
// cubic B-spline weights (scipy's order-3 taps at t, t in [0, 1)), float64
__device__ __forceinline__ void bbx_bspline_w(double t, double w[4]) {
    const double u = 1.0 - t, t2 = t * t, t3 = t2 * t;
    w[0] = u * u * u / 6.0;
    w[1] = (3.0 * t3 - 6.0 * t2 + 4.0) / 6.0;
    w[2] = (-3.0 * t3 + 3.0 * t2 + 3.0 * t + 1.0) / 6.0;
    w[3] = t3 / 6.0;
}
// The cubic of frame row Y on the interval whose tap-1 coefficient column is c: value(t) = p.x + t (p.y + t (p.z + t p.w)).
// (0 when the interval's taps leave the coefficient array: the gaps between two patches' column ranges)
__device__ __forceinline__ float4 bbx_spl_poly(const bbx_spl& sp, int Y, int c) {
    if (c - 1 < 0 || c + 2 >= sp.cnx) return make_float4(0.f, 0.f, 0.f, 0.f);
    int cy, ry;
    bbx_spl_axis(Y, sp.ph, sp.rph, sp.py, sp.npad, sp.ny1, sp.dy, sp.rdy, cy, ry);
    double wy[4];
    bbx_bspline_w((double)ry / (double)sp.dy, wy);
    const double* c0 = sp.coef + (size_t)(cy - 1) * sp.cnx + (c - 1);
    double r[4];
#pragma unroll
    for (int b = 0; b < 4; b++)
        r[b] = ((c0[b] * wy[0] + c0[sp.cnx + b] * wy[1]) + c0[2 * (size_t)sp.cnx + b] * wy[2]) + c0[3 * (size_t)sp.cnx + b] * wy[3];
    return make_float4((float)((r[0] + 4.0 * r[1] + r[2]) / 6.0), (float)((r[2] - r[0]) * 0.5), (float)((r[0] - 2.0 * r[1] + r[2]) * 0.5),
                       (float)((r[3] - r[0] + 3.0 * (r[1] - r[2])) / 6.0));
}
__device__ __forceinline__ float bbx_spl_horner(float4 p, float t) { return p.x + t * (p.y + t * (p.z + t * p.w)); }
// one pixel, everything from global memory (sparse callers: stamps around catalogue sources)
__device__ __forceinline__ float bbx_spl_eval(const bbx_spl& sp, int Y, int X) {
    int c, r;
    bbx_spl_axis(X, sp.pw, sp.rpw, sp.px, sp.npad, sp.nx1, sp.dx, sp.rdx, c, r);
    return bbx_spl_horner(bbx_spl_poly(sp, Y, c), (float)r * sp.rdx);
}
// the cubics of all rows at once: table[Y][c] (a frame of 10560 rows x 368 coefficient columns: 62 MB, made in ~20 us; the
// kernels that read a sigma map then take one 16-byte load per group of pixels instead of folding coefficients themselves)
__global__ __launch_bounds__(256) static void k_spl_polytable(bbx_spl sp, int ny, float4* __restrict__ table) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x), Y = (int)blockIdx.y;
    if (c < sp.cnx && Y < ny) table[(size_t)Y * sp.cnx + c] = bbx_spl_poly(sp, Y, c);
}
