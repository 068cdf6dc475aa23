// bbx_zogy2.hip -- ZOGY on whole frames with a hand-written 2-D FFT (gfx950).
// (zogy.optimal_subtraction -> run_ZOGY, called at blackbox.py:2350-2354 / 2460-2465; Zackay,
// Ofek & Gal-Yam 2016.  [EXT: parity unpinned, conventions = oracle/zogy_core.py])
//
// Why not rocFFT: a batched 2-D real transform of 1400 x 1400 runs as four kernels (two 1-D
// passes and two transposes) that each move the whole batch through HBM, and the element-wise
// ZOGY algebra needs four more passes: 17.4 ms per frame for 3.8 GB of algorithmic I/O
// (profiles/r01_stage_bench.json).  Here every kernel is "load NL lines -> 1-D FFTs in LDS ->
// point-wise algebra in registers -> (more FFTs) -> store", and the transposition between the
// row and the column pass is the store pattern of the producing kernel (NL x 8-byte runs):
//
//   sub-image side L = N1 * N2 (1400 = 35 * 40).  One line of L complex values lives in LDS; a
//   1-D FFT is two steps of register DFTs (tools/gen_fft.py), thread-private in both steps:
//     forward : step 1, thread n2: DFT_N1 over x[N2 n1 + n2], times W_L^(n2 k1), in place;
//               barrier; step 2, thread k1: DFT_N2 over n2 -> X[k1 + N1 k2] in registers,
//               i.e. the spectrum in the digit-swapped order pos(k) = N2 (k mod N1) + k div N1
//     inverse : the same backwards from that order (no reordering pass anywhere: the
//               point-wise algebra between a forward and an inverse transform does not care)
//   Real rows are transformed in pairs (a + i b), the two half spectra come out of the
//   Hermitian split; the inverse row pass packs two half spectra the same way.
//
// Launch sequence for one frame of nsub sub-images (layouts: T = [sub][kx][y], U = [sub][y][kx],
// C = [sub][kx group][k2][k1][line] -- the register order of the column kernels):
//   k_psf_cols   PSF stamps -> Pn^, Pr^ (the row DFT of the few non-zero rows is summed directly)
//                -> coefficient arrays A, B, kn^, kr^ (C) + F_S partial sums; kn^, kr^ back
//                through the inverse column pass (U)
//   k_psf_rows   inverse row pass -> kr, kn (real) -> squares -> forward row pass (T)
//   k_k2_cols    forward column pass of (kr^2)^, (kn^2)^ (C)
//   k_img_rows   cut of the frames + forward row pass of (N, R) and (Vn, Vr); V = max(d,0)+sigma^2
//   k_img_cols   column pass: D^ = A N^ - B R^, Sn^ = kn^ N^, Sr^ = kr^ R^, inverse column pass (U)
//   k_var_cols   column pass: V(S)^ = Vn^ (kn^2)^ + Vr^ (kr^2)^, inverse column pass (U)
//   k_final_rows inverse row pass of (D, V_S) and (Sn, Sr); S = Sn - Sr; S_corr with the
//                astrometric variance from the finite differences of Sn, Sr; F_psf, F_psf_err;
//                written straight into the full-frame outputs (borders dropped)
// HBM traffic: ~44 passes of nsub * L * (L/2+1) * 8 bytes = 22 GB per frame of 64 x 1400^2.
#include "bbx_common.h"
#include "bbx_fft_gen.h"
#include <math.h>
#include <stdlib.h>

namespace z2 {

// compiler-level fence: keeps the scheduler from hoisting every load of an unrolled loop to its top
#define FENCE() asm volatile("" ::: "memory")

struct zscal { float sn, sr, fn, fr, dx, dy; };

constexpr int floor_pow2(int v) { int p = 1; while (2 * p <= v) p *= 2; return p; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cmin(int a, int b) { return a < b ? a : b; }
// line stride in float2: >= lp, and 2 * stride = 8 (mod 64) dwords so that the NL lines of one
// wave-instruction start on different banks
constexpr int line_stride(int lp) { int s = lp; while ((2 * s) % 64 != 8) s++; return s; }

#ifndef Z2_NL_BIG
#define Z2_NL_BIG 6           // 6 lines of 1400: 69 KB of LDS, two workgroups per CU (NL = 8: one)
#endif
template <int N1_, int N2_> struct Plan {
    static constexpr int N1 = N1_, N2 = N2_, L = N1_ * N2_, H = L / 2 + 1;
    static constexpr int NT = cmax(N1_, N2_);                       // thread tasks per line
    static constexpr int NL = cmin(NT >= 32 ? Z2_NL_BIG : 16, floor_pow2(320 / NT));       // lines per workgroup
    static constexpr int THREADS = ((NL * NT + 63) / 64) * 64;
#ifndef Z2_MINW
#define Z2_MINW 2
#endif
    static constexpr int MINW = Z2_MINW;                            // minimum waves per SIMD the register allocation must allow
    static constexpr int G = (H + NL - 1) / NL, HP = G * NL;        // column groups, padded half-spectrum width
    static constexpr int LP = N1_ * (N2_ + 1);                      // padded line: one pad per N2 entries
    static constexpr int LS = line_stride(LP);
    static constexpr int CT = NL * N1_;                             // step-2 thread tasks per workgroup
    static constexpr int LB = (L + NL - 1) / NL;                    // row blocks of NL rows
    // T and U are stored in NL x NL tiles: T[sub][yb][kx][yi] (row kernels write one contiguous
    // block per workgroup, column kernels read 8 NL^2-byte tiles), U[sub][g][y][l] (the reverse)
    static constexpr size_t UNIT = (size_t)LB * NL * HP;            // elements of one T / U / C array per sub-image
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

// knock-out switches for timing experiments (tools/dbg/z2_variants.sh); never set in the product build
#ifdef Z2_SKIP_DFT
#define DFT_RUN(N, x) ((void)0)
#else
#define DFT_RUN(N, x) bbx_dft<N>::run(x)
#endif
#ifdef Z2_SKIP_TW
#define TW(i) make_float2(1.f, 0.f)
#else
#define TW(i) tw[i]
#endif
template <int N> __device__ __forceinline__ void idft(float2 (&x)[N]) {      // inverse = forward on swapped pairs
#pragma unroll
    for (int i = 0; i < N; i++) { const float t = x[i].x; x[i].x = x[i].y; x[i].y = t; }
    DFT_RUN(N, x);
#pragma unroll
    for (int i = 0; i < N; i++) { const float t = x[i].x; x[i].x = x[i].y; x[i].y = t; }
}

template <class P> __device__ __forceinline__ int ppos(int k) { return (P::N2 + 1) * (k % P::N1) + k / P::N1; }   // spectrum order
template <class P> __device__ __forceinline__ int npos(int n) { return n + n / P::N2; }                              // natural order

// ---- the two steps (line = this thread's line in LDS, t = task index) ------------------------
// forward step 1 on values already in registers (x[n1] = sample N2 n1 + t), result to LDS
template <class P> __device__ __forceinline__ void fwd_step1_regs(float2 (&x)[P::N1], float2* line, int t, const float2* __restrict__ tw) {
    DFT_RUN(P::N1, x);
#pragma unroll
    for (int k1 = 0; k1 < P::N1; k1++) line[(P::N2 + 1) * k1 + t] = cmul(x[k1], TW(t * k1));
}
template <class P> __device__ __forceinline__ void fwd_step1(float2* line, int t, const float2* __restrict__ tw) {
    float2 x[P::N1];
#pragma unroll
    for (int n1 = 0; n1 < P::N1; n1++) x[n1] = line[(P::N2 + 1) * n1 + t];
    fwd_step1_regs<P>(x, line, t, tw);
}
// forward step 2: thread t = k1 -> X[k1 + N1 k2] in x[k2]
template <class P> __device__ __forceinline__ void fwd_step2(const float2* line, int t, float2 (&x)[P::N2]) {
#pragma unroll
    for (int n2 = 0; n2 < P::N2; n2++) x[n2] = line[(P::N2 + 1) * t + n2];
    DFT_RUN(P::N2, x);
}
// inverse step 2 from registers (x[k2] of thread t = k1), result to LDS
template <class P> __device__ __forceinline__ void inv_step2(float2 (&x)[P::N2], float2* line, int t, const float2* __restrict__ tw) {
    idft<P::N2>(x);
#pragma unroll
    for (int n2 = 0; n2 < P::N2; n2++) line[(P::N2 + 1) * t + n2] = cmulc(x[n2], TW(n2 * t));
}
// inverse step 1: thread t = n2 -> x[n1] = (unnormalised) sample N2 n1 + t
template <class P> __device__ __forceinline__ void inv_step1(const float2* line, int t, float2 (&x)[P::N1]) {
#pragma unroll
    for (int k1 = 0; k1 < P::N1; k1++) x[k1] = line[(P::N2 + 1) * k1 + t];
    idft<P::N1>(x);
}
// whole transforms for all lines of the workgroup (barriers inside; every thread must call)
template <class P> __device__ __forceinline__ void fwd_lines(float2* s, int l, int t, const float2* tw, float2 (&X)[P::N2]) {
    if (t < P::N2) fwd_step1<P>(s + l * P::LS, t, tw);
    __syncthreads();
    if (t < P::N1) fwd_step2<P>(s + l * P::LS, t, X);
}
template <class P> __device__ __forceinline__ void inv_lines(float2 (&X)[P::N2], float2* s, int l, int t, const float2* tw, float2 (&x)[P::N1]) {
    if (t < P::N1) inv_step2<P>(X, s + l * P::LS, t, tw);
    __syncthreads();
    if (t < P::N2) inv_step1<P>(s + l * P::LS, t, x);
}

// store the inverse column pass (thread t = n2 holds y = N2 n1 + t of column g NL + l) to the U layout:
// U[sub][g][y][l], a wave-instruction writes 64 consecutive entries.  halo != NULL: the last row of
// every row block (and row L - 1) also goes to halo[sub][yb][kx] for the finite differences across blocks.
template <class P> __device__ __forceinline__ void store_u(float2* __restrict__ U, int sub, int g, int l, int t, const float2 (&x)[P::N1],
                                                           float2* __restrict__ halo = nullptr) {
    float2* base = U + (size_t)sub * P::UNIT + (size_t)g * P::L * P::NL + l;
#pragma unroll
    for (int n1 = 0; n1 < P::N1; n1++) base[(size_t)(P::N2 * n1 + t) * P::NL] = x[n1];
    if (halo) {
#pragma unroll
        for (int n1 = 0; n1 < P::N1; n1++) {
            const int y = P::N2 * n1 + t;
            if (y % P::NL == P::NL - 1 || y == P::L - 1) halo[((size_t)sub * P::LB + y / P::NL) * P::HP + g * P::NL + l] = x[n1];
        }
    }
}
// load NL lines of a T-layout array ([sub][kx][y], kx = g NL + l) into LDS, natural order.  All loads
// of a thread are issued before its first LDS write (one wave per SIMD: nothing else hides the latency).
template <class P> __device__ __forceinline__ void load_t_lines(const float2* T, int sub, int g, float2* s) {
    static_assert(P::NL % 2 == 0, "even tile side");
    const float2* src = T + (size_t)sub * P::UNIT + (size_t)g * P::NL * P::NL;
    constexpr int TV = P::NL * P::NL / 2;                           // float4 per tile
    constexpr int NV = P::LB * TV, IT = (NV + P::THREADS - 1) / P::THREADS;
    float4 v[IT];
#pragma unroll
    for (int i = 0; i < IT; i++) {
        const int e = threadIdx.x + i * P::THREADS;
        if (e < NV) { const int yb = e / TV, j = e - yb * TV; v[i] = *reinterpret_cast<const float4*>(src + (size_t)yb * P::HP * P::NL + 2 * j); }
    }
#pragma unroll
    for (int i = 0; i < IT; i++) {
        const int e = threadIdx.x + i * P::THREADS;
        if (e < NV) {
            const int yb = e / TV, j = e - yb * TV, l = (2 * j) / P::NL, y = yb * P::NL + (2 * j) % P::NL;
            if (y < P::L) {
                float2* d = s + l * P::LS + npos<P>(y);             // y even: y and y + 1 share their pad group
                d[0] = make_float2(v[i].x, v[i].y); d[1] = make_float2(v[i].z, v[i].w);
            }
        }
    }
}
// the same tiles as the workgroup's scratch: entry p of its NL * L values
template <class P> __device__ __forceinline__ float2* park_ptr(float2* T, int sub, int g, int p) {
    constexpr int TS = P::NL * P::NL;
    return T + (size_t)sub * P::UNIT + (size_t)(p / TS) * P::HP * P::NL + (size_t)g * TS + p % TS;
}
// Hermitian packing of two U-layout half spectra (rows y0 .. y0+NL-1) into full complex lines in
// spectrum order: Z[k] = a[k] + i b[k]
template <class P> __device__ __forceinline__ void pack_store(float2* line, int kx, float2 a, float2 b) {
    if (kx >= P::H) return;
    line[ppos<P>(kx)] = make_float2(a.x - b.y, a.y + b.x);
    if (kx >= 1 && P::L - kx >= P::H) line[ppos<P>(P::L - kx)] = make_float2(a.x + b.y, b.x - a.y);
}
template <class P> __device__ __forceinline__ void load_u_pair(const float2* __restrict__ Ua, const float2* __restrict__ Ub, int sub, int yb,
                                                               float2* s) {
    constexpr int TV = P::NL * P::NL / 2, NV = P::G * TV, IT = (NV + P::THREADS - 1) / P::THREADS;
    float4 va[IT], vb[IT];
    const size_t base = (size_t)sub * P::UNIT + (size_t)yb * P::NL * P::NL;
#pragma unroll
    for (int i = 0; i < IT; i++) {
        const int e = threadIdx.x + i * P::THREADS;
        if (e < NV) {
            const int g = e / TV, j = e - g * TV;
            const size_t o = base + (size_t)g * P::L * P::NL + 2 * j;
            va[i] = *reinterpret_cast<const float4*>(Ua + o);
            vb[i] = *reinterpret_cast<const float4*>(Ub + o);
        }
    }
#pragma unroll
    for (int i = 0; i < IT; i++) {
        const int e = threadIdx.x + i * P::THREADS;
        if (e < NV) {
            const int g = e / TV, j = e - g * TV, row = (2 * j) / P::NL, l = (2 * j) % P::NL;      // tile entry [row][l], l even
            if (yb * P::NL + row < P::L) {
                float2* line = s + row * P::LS;
                pack_store<P>(line, g * P::NL + l, make_float2(va[i].x, va[i].y), make_float2(vb[i].x, vb[i].y));
                pack_store<P>(line, g * P::NL + l + 1, make_float2(va[i].z, va[i].w), make_float2(vb[i].z, vb[i].w));
            }
        }
    }
}
// one extra line from the halo arrays (row-major over kx)
template <class P> __device__ __forceinline__ void load_halo_pair(const float2* __restrict__ Ha, const float2* __restrict__ Hb, int sub, int yb,
                                                                  float2* line, int tid, int nthreads) {
    const size_t base = ((size_t)sub * P::LB + yb) * P::HP;
    for (int kx = tid; kx < P::H; kx += nthreads) pack_store<P>(line, kx, Ha[base + kx], Hb[base + kx]);
}
// Hermitian split of a packed transform Z (LDS, spectrum order) -> two half spectra, T layout
template <class P> __device__ __forceinline__ void store_t_split(const float2* s, float2* __restrict__ Ta, float2* __restrict__ Tb, int sub, int yb) {
    constexpr int NE = P::NL * P::H, IT = (NE + P::THREADS - 1) / P::THREADS;
    const size_t base = (size_t)sub * P::UNIT + (size_t)yb * P::HP * P::NL;         // the block's entries (kx, row) are contiguous
#pragma unroll 6
    for (int i = 0; i < IT; i++) {
        const int e = threadIdx.x + i * P::THREADS;
        if (e >= NE) break;
        const int kx = e / P::NL, row = e - kx * P::NL;
        const float2* line = s + row * P::LS;
        const float2 zk = line[ppos<P>(kx)], zm = line[ppos<P>(kx ? P::L - kx : 0)];
        Ta[base + e] = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        Tb[base + e] = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
    }
}

// ---- PSF side ---------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(P::THREADS, P::MINW) void k_psf_cols(const float* __restrict__ psf_n, const float* __restrict__ psf_r, int S,
                                                         const zscal* __restrict__ sc, const float2* __restrict__ tw,
                                                         float2* __restrict__ cA, float2* __restrict__ cB, float2* __restrict__ cKn,
                                                         float2* __restrict__ cKr, float2* __restrict__ Ukn, float2* __restrict__ Ukr,
                                                         double* __restrict__ fs_partial) {
    extern __shared__ float2 s[];
    float2* twl = s + P::NL * P::LS;                                   // the twiddle table, in LDS for the gathers of the two steps
    for (int e = threadIdx.x; e < P::L; e += blockDim.x) twl[e] = tw[e];
    __shared__ double red[3][P::THREADS / 64];
    const int g = blockIdx.x, sub = blockIdx.y;
    const int l = threadIdx.x % P::NL, t = threadIdx.x / P::NL;
    const int kx = g * P::NL + l, h = S / 2;
    float2 X[P::N2];
    const size_t cbase = ((size_t)(sub * P::G + g) * P::N2) * P::CT + (size_t)t * P::NL + l;
    for (int pass = 0; pass < 2; pass++) {
        const float* st = (pass ? psf_r : psf_n) + (size_t)sub * S * S;
        for (int e = threadIdx.x; e < P::NL * P::LS; e += P::THREADS) s[e] = make_float2(0.f, 0.f);
        __syncthreads();
        // row DFT of the S non-zero rows, summed directly: line[y] = sum_x p[y][x] W^(kx x)
        for (int e = threadIdx.x; e < P::NL * S; e += P::THREADS) {
            const int ll = e % P::NL, j = e / P::NL;
            const int kk = g * P::NL + ll;
            if (kk >= P::H) continue;
            float2 acc = make_float2(0.f, 0.f);
            for (int i = 0; i < S; i++) {
                const int xw = ((i - h) % P::L + P::L) % P::L;
                const float2 w = tw[(int)(((long long)kk * xw) % P::L)];
                const float p = st[j * S + i];
                acc.x += p * w.x; acc.y += p * w.y;
            }
            const int y = ((j - h) % P::L + P::L) % P::L;
            s[ll * P::LS + npos<P>(y)] = acc;
        }
        __syncthreads();
        fwd_lines<P>(s, l, t, twl, X);
        if (pass == 0 && t < P::N1) {
            // Pn^ waits in the (not yet written) A array while Pr^ is transformed: one spectrum in registers at a time
#pragma unroll
            for (int k2 = 0; k2 < P::N2; k2++) cA[cbase + (size_t)k2 * P::CT] = X[k2];
        }
        __syncthreads();
    }
    const zscal z = sc[sub];
    const float sn2 = z.sn * z.sn, sr2 = z.sr * z.sr, fn2 = z.fn * z.fn, fr2 = z.fr * z.fr;
    double fs = 0.0, sk2n = 0.0, sk2r = 0.0;                      // F_S and the Parseval sums of kn^2, kr^2
    if (t < P::N1) {
        const bool live = kx < P::H;
        const double wgt = (kx == 0 || (P::L % 2 == 0 && kx == P::L / 2)) ? 1.0 : 2.0;
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) {
            const size_t o = cbase + (size_t)k2 * P::CT;
            float2 a = make_float2(0.f, 0.f), b = a, kn = a, kr = a;
            if (live) {
                const float2 pn = cA[o], pr = X[k2];
                const float pn2 = pn.x * pn.x + pn.y * pn.y, pr2 = pr.x * pr.x + pr.y * pr.y;
                const float den = (sn2 * fr2) * pr2 + (sr2 * fn2) * pn2;
                const float isd = 1.0f / sqrtf(den);
                a = cscale(pr, z.fr * isd);                                   // D^ = A N^ - B R^
                b = cscale(pn, z.fn * isd);
                kr = cscale(make_float2(pr.x, -pr.y), z.fr * fn2 * pn2 / den);
                kn = cscale(make_float2(pn.x, -pn.y), z.fn * fr2 * pr2 / den);
                fs += wgt * (double)(fn2 * pn2 * fr2 * pr2 / den);
                sk2n += wgt * (double)(kn.x * kn.x + kn.y * kn.y);
                sk2r += wgt * (double)(kr.x * kr.x + kr.y * kr.y);
            }
            cA[o] = a; cB[o] = b; cKn[o] = kn; cKr[o] = kr;
            X[k2] = kr;
        }
    }
    float2 x[P::N1];
    inv_lines<P>(X, s, l, t, twl, x);
    if (t < P::N2) store_u<P>(Ukr, sub, g, l, t, x);
    __syncthreads();
    if (t < P::N1) {
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) X[k2] = cKn[cbase + (size_t)k2 * P::CT];       // written by this thread above
    }
    inv_lines<P>(X, s, l, t, twl, x);
    if (t < P::N2) store_u<P>(Ukn, sub, g, l, t, x);
    fs = wave_sum_f64(fs); sk2n = wave_sum_f64(sk2n); sk2r = wave_sum_f64(sk2r);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = fs; red[1][threadIdx.x >> 6] = sk2n; red[2][threadIdx.x >> 6] = sk2r; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double tot = 0.0;
        for (int i = 0; i < P::THREADS / 64; i++) tot += red[threadIdx.x][i];
        fs_partial[((size_t)threadIdx.x * gridDim.y + sub) * P::G + g] = tot;      // [3][nsub][G]
    }
}

// inverse row pass of kr^, kn^ -> kr, kn -> squares -> forward row pass, T layout
template <class P>
__global__ __launch_bounds__(P::THREADS, P::MINW) void k_psf_rows(const float2* __restrict__ Ukr, const float2* __restrict__ Ukn, float inv_n2,
                                                         const float2* __restrict__ tw, float2* __restrict__ Tkr2, float2* __restrict__ Tkn2) {
    extern __shared__ float2 s[];
    float2* twl = s + P::NL * P::LS;                                   // the twiddle table, in LDS for the gathers of the two steps
    for (int e = threadIdx.x; e < P::L; e += blockDim.x) twl[e] = tw[e];
    const int sub = blockIdx.y;
    const int l = threadIdx.x % P::NL, t = threadIdx.x / P::NL;
    load_u_pair<P>(Ukr, Ukn, sub, blockIdx.x, s);
    __syncthreads();
    float2 X[P::N2], x[P::N1];
    if (t < P::N1) {
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) X[k2] = s[l * P::LS + (P::N2 + 1) * t + k2];
    }
    inv_lines<P>(X, s, l, t, twl, x);
    __syncthreads();
    if (t < P::N2) {
#pragma unroll
        for (int n1 = 0; n1 < P::N1; n1++) { const float a = x[n1].x * inv_n2, b = x[n1].y * inv_n2; x[n1] = make_float2(a * a, b * b); }
        fwd_step1_regs<P>(x, s + l * P::LS, t, twl);
    }
    __syncthreads();
    if (t < P::N1) {
        fwd_step2<P>(s + l * P::LS, t, X);
    }
    __syncthreads();
    if (t < P::N1) {
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) s[l * P::LS + (P::N2 + 1) * t + k2] = X[k2];
    }
    __syncthreads();
    store_t_split<P>(s, Tkr2, Tkn2, sub, blockIdx.x);
}

// forward column pass of one T-layout array -> C layout
template <class P>
__global__ __launch_bounds__(P::THREADS, P::MINW) void k_cols_fwd(const float2* __restrict__ T, const float2* __restrict__ tw, float2* __restrict__ Cout) {
    extern __shared__ float2 s[];
    float2* twl = s + P::NL * P::LS;                                   // the twiddle table, in LDS for the gathers of the two steps
    for (int e = threadIdx.x; e < P::L; e += blockDim.x) twl[e] = tw[e];
    const int g = blockIdx.x, sub = blockIdx.y;
    const int l = threadIdx.x % P::NL, t = threadIdx.x / P::NL;
    load_t_lines<P>(T, sub, g, s);
    __syncthreads();
    float2 X[P::N2];
    fwd_lines<P>(s, l, t, twl, X);
    if (t < P::N1) {
        const size_t cbase = ((size_t)(sub * P::G + g) * P::N2) * P::CT + (size_t)t * P::NL + l;
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) Cout[cbase + (size_t)k2 * P::CT] = X[k2];
    }
}

// V(S) is ~1e-4 of D in the units both leave the column pass in, and the final inverse row pass
// transforms them as one complex signal (D + i V_S): float32 rounding of the larger part would
// leak into the smaller one (0.1 % of V_S typically, 10 % next to bright residuals).  V(S)^ is
// therefore scaled by a power of two (exact) that brings its sky level -- sigma_n^2 sum(kn^2) +
// sigma_r^2 sum(kr^2), from the Parseval sums of k_psf_cols -- to ~1, and unscaled at the end.
template <class P> __device__ __forceinline__ float vs_scale(const double* __restrict__ fs_partial, int nsub, int sub, const zscal& z) {
    double a = 0.0, b = 0.0;
    for (int g = 0; g < P::G; g++) {
        a += fs_partial[((size_t)1 * nsub + sub) * P::G + g];
        b += fs_partial[((size_t)2 * nsub + sub) * P::G + g];
    }
    const double n2 = (double)P::L * (double)P::L;
    const float level = (float)(((double)z.sn * z.sn * a + (double)z.sr * z.sr * b) / n2);            // Parseval: sum_x k^2 = sum_k |k^|^2 / L^2
    if (!(level > 0.f) || !isfinite(level)) return 1.f;
    return exp2f(-rintf(log2f(level)));
}

// ---- image side -------------------------------------------------------------------------------
struct frame_args {
    const float* a; const float* b;          // the two frames of a pair (new, ref) or their sigma images
    const float* sa; const float* sb;        // sigma images (variance pair) or NULL
    int ny, nx, size, border, nsx, vec4;
};

// cut + forward row pass of a pair of real frames: (N, R) or, with sigma images, (Vn, Vr)
template <class P>
__global__ __launch_bounds__(P::THREADS, P::MINW) void k_img_rows(frame_args f, const float2* __restrict__ tw, float2* __restrict__ Ta,
                                                         float2* __restrict__ Tb) {
    extern __shared__ float2 s[];
    float2* twl = s + P::NL * P::LS;                                   // the twiddle table, in LDS for the gathers of the two steps
    for (int e = threadIdx.x; e < P::L; e += blockDim.x) twl[e] = tw[e];
    const int y0 = blockIdx.x * P::NL, sub = blockIdx.y;
    const int l = threadIdx.x % P::NL, t = threadIdx.x / P::NL;
    const int sy = sub / f.nsx, sx = sub - sy * f.nsx;
    const int Y0 = sy * f.size - f.border, X0 = sx * f.size - f.border;
    if (f.vec4) {
        // groups of four pixels never straddle the frame edge or a pad group (size, border, nx multiples of 4)
        constexpr int NV = P::NL * P::L / 4, IT = (NV + P::THREADS - 1) / P::THREADS;
        float4 va[IT], vb[IT];
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        size_t off[IT];
#pragma unroll
        for (int i = 0; i < IT; i++) {
            const int e = threadIdx.x + i * P::THREADS;
            va[i] = zero; vb[i] = zero; off[i] = (size_t)-1;
            if (e < NV) {
                const int q = 4 * e, ll = q / P::L, x = q - ll * P::L;
                const int Y = Y0 + y0 + ll, X = X0 + x;
                if (y0 + ll < P::L && Y >= 0 && Y < f.ny && X >= 0 && X < f.nx) {
                    off[i] = (size_t)Y * f.nx + X;
                    va[i] = *reinterpret_cast<const float4*>(f.a + off[i]);
                    vb[i] = *reinterpret_cast<const float4*>(f.b + off[i]);
                }
            }
        }
        if (f.sa) {
#pragma unroll
            for (int i = 0; i < IT; i++) {
                if (off[i] != (size_t)-1) {
                    const float4 p = *reinterpret_cast<const float4*>(f.sa + off[i]), q = *reinterpret_cast<const float4*>(f.sb + off[i]);
                    va[i] = make_float4(fmaxf(va[i].x, 0.f) + p.x * p.x, fmaxf(va[i].y, 0.f) + p.y * p.y, fmaxf(va[i].z, 0.f) + p.z * p.z,
                                        fmaxf(va[i].w, 0.f) + p.w * p.w);
                    vb[i] = make_float4(fmaxf(vb[i].x, 0.f) + q.x * q.x, fmaxf(vb[i].y, 0.f) + q.y * q.y, fmaxf(vb[i].z, 0.f) + q.z * q.z,
                                        fmaxf(vb[i].w, 0.f) + q.w * q.w);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < IT; i++) {
            const int e = threadIdx.x + i * P::THREADS;
            if (e < NV) {
                const int q = 4 * e, ll = q / P::L, x = q - ll * P::L;
                float2* d = s + ll * P::LS + npos<P>(x);
                d[0] = make_float2(va[i].x, vb[i].x); d[1] = make_float2(va[i].y, vb[i].y);
                d[2] = make_float2(va[i].z, vb[i].z); d[3] = make_float2(va[i].w, vb[i].w);
            }
        }
    } else
    for (int e = threadIdx.x; e < P::NL * P::L; e += P::THREADS) {
        const int ll = e / P::L, x = e - ll * P::L;
        const int Y = Y0 + y0 + ll, X = X0 + x;
        float2 v = make_float2(0.f, 0.f);
        if (y0 + ll < P::L && Y >= 0 && Y < f.ny && X >= 0 && X < f.nx) {
            const size_t o = (size_t)Y * f.nx + X;
            v.x = f.a[o]; v.y = f.b[o];
            if (f.sa) { const float p = f.sa[o], q = f.sb[o]; v.x = fmaxf(v.x, 0.f) + p * p; v.y = fmaxf(v.y, 0.f) + q * q; }
        }
        s[ll * P::LS + npos<P>(x)] = v;
    }
    __syncthreads();
    float2 X[P::N2];
    fwd_lines<P>(s, l, t, twl, X);
    __syncthreads();
    if (t < P::N1) {
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) s[l * P::LS + (P::N2 + 1) * t + k2] = X[k2];
    }
    __syncthreads();
    store_t_split<P>(s, Ta, Tb, sub, blockIdx.x);
}

// column pass of the image pair: D^ = A N^ - B R^, Sn^ = kn^ N^, Sr^ = kr^ R^ and back (U layout)
template <class P>
__global__ __launch_bounds__(P::THREADS, P::MINW) void k_img_cols(float2* TN, const float2* __restrict__ TR,
                                                         const float2* __restrict__ cA, const float2* __restrict__ cB,
                                                         const float2* __restrict__ cKn, const float2* __restrict__ cKr,
                                                         const float2* __restrict__ tw, float2* __restrict__ UD, float2* __restrict__ USn,
                                                         float2* __restrict__ USr, float2* __restrict__ HSn, float2* __restrict__ HSr) {
    extern __shared__ float2 s[];
    float2* twl = s + P::NL * P::LS;                                   // the twiddle table, in LDS for the gathers of the two steps
    for (int e = threadIdx.x; e < P::L; e += blockDim.x) twl[e] = tw[e];
    const int g = blockIdx.x, sub = blockIdx.y;
    const int l = threadIdx.x % P::NL, t = threadIdx.x / P::NL;
    const size_t cbase = ((size_t)(sub * P::G + g) * P::N2) * P::CT + (size_t)t * P::NL + l;
    float2 X[P::N2], x[P::N1];
    // the partial D^ = A N^ waits in this workgroup's own (already consumed) lines of T_N, in the
    // register order of the C layout (coalesced; it stays in the XCD's L2 for the few microseconds)
#define PARK(k2) (*park_ptr<P>(TN, sub, g, (k2) * P::CT + t * P::NL + l))
    load_t_lines<P>(TN, sub, g, s);
    __syncthreads();
    fwd_lines<P>(s, l, t, twl, X);
    if (t < P::N1) {
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) {
            const size_t o = cbase + (size_t)k2 * P::CT;
            PARK(k2) = cmul(cA[o], X[k2]);
            X[k2] = cmul(cKn[o], X[k2]);
        }
    }
    __syncthreads();
    inv_lines<P>(X, s, l, t, twl, x);
    if (t < P::N2) store_u<P>(USn, sub, g, l, t, x, HSn);
    __syncthreads();
    load_t_lines<P>(TR, sub, g, s);
    __syncthreads();
    fwd_lines<P>(s, l, t, twl, X);
    if (t < P::N1) {
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) {
            const size_t o = cbase + (size_t)k2 * P::CT;
            const float2 br = cmul(cB[o], X[k2]), da = PARK(k2);
            PARK(k2) = make_float2(da.x - br.x, da.y - br.y);
            X[k2] = cmul(cKr[o], X[k2]);
        }
    }
    __syncthreads();
    inv_lines<P>(X, s, l, t, twl, x);
    if (t < P::N2) store_u<P>(USr, sub, g, l, t, x, HSr);
    __syncthreads();
    if (t < P::N1) {
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) X[k2] = PARK(k2);
    }
    inv_lines<P>(X, s, l, t, twl, x);
    if (t < P::N2) store_u<P>(UD, sub, g, l, t, x);
}

// column pass of the variance pair: V(S)^ = Vn^ (kn^2)^ + Vr^ (kr^2)^ and back (U layout)
template <class P>
__global__ __launch_bounds__(P::THREADS, P::MINW) void k_var_cols(float2* TVn, const float2* __restrict__ TVr,
                                                         const float2* __restrict__ cK2n, const float2* __restrict__ cK2r,
                                                         const float2* __restrict__ tw, float2* __restrict__ UVS,
                                                         const zscal* __restrict__ sc, const double* __restrict__ fs_partial) {
    extern __shared__ float2 s[];
    float2* twl = s + P::NL * P::LS;                                   // the twiddle table, in LDS for the gathers of the two steps
    for (int e = threadIdx.x; e < P::L; e += blockDim.x) twl[e] = tw[e];
    __shared__ float s_beta;
    const int g = blockIdx.x, sub = blockIdx.y;
    if (threadIdx.x == 0) s_beta = vs_scale<P>(fs_partial, gridDim.y, sub, sc[sub]);
    const int l = threadIdx.x % P::NL, t = threadIdx.x / P::NL;
    const size_t cbase = ((size_t)(sub * P::G + g) * P::N2) * P::CT + (size_t)t * P::NL + l;
    float2 X[P::N2], x[P::N1];
#define PARKV(k2) (*park_ptr<P>(TVn, sub, g, (k2) * P::CT + t * P::NL + l))
    load_t_lines<P>(TVn, sub, g, s);
    __syncthreads();
    fwd_lines<P>(s, l, t, twl, X);
    if (t < P::N1) {
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) PARKV(k2) = cmul(cK2n[cbase + (size_t)k2 * P::CT], X[k2]);
    }
    __syncthreads();
    load_t_lines<P>(TVr, sub, g, s);
    __syncthreads();
    const float beta = s_beta;
    fwd_lines<P>(s, l, t, twl, X);
    if (t < P::N1) {
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) {
            const float2 v = cmul(cK2r[cbase + (size_t)k2 * P::CT], X[k2]), a = PARKV(k2);
            X[k2] = make_float2((a.x + v.x) * beta, (a.y + v.y) * beta);
        }
    }
    __syncthreads();
    inv_lines<P>(X, s, l, t, twl, x);
    if (t < P::N2) store_u<P>(UVS, sub, g, l, t, x);
}

struct out_args {
    float* D; float* S; float* Scorr; float* Fpsf; float* Fpsferr;      // full frames [ny][nx]; S may be NULL
    int ny, nx, size, border, nsx, vec4;
};

// inverse row pass of (D, V_S) and (Sn, Sr) + the final algebra, written into the full frames.
// A workgroup takes one block of NL rows.  The row above the block (for the y finite difference)
// comes from the halo arrays as one more line, transformed by the 64 extra threads of the workgroup.
// Thread mapping: task index fastest (unlike the other kernels): after the inverse transform a
// thread holds x = N2 n1 + t of its row, so the lanes of a wave-instruction hold up to N2
// consecutive pixels of a row and the five outputs go from registers straight to the frames.
template <class P>
__global__ __launch_bounds__(P::THREADS + 64, P::MINW) void k_final_rows(const float2* __restrict__ UD, const float2* __restrict__ UVS,
                                                                         const float2* __restrict__ USn, const float2* __restrict__ USr,
                                                                         const float2* __restrict__ HSn, const float2* __restrict__ HSr,
                                                                         const zscal* __restrict__ sc, const double* __restrict__ fs_partial,
                                                                         float inv_n2, const float2* __restrict__ tw, out_args o, int yb0) {
    extern __shared__ float2 s[];
    float2* hline = s + P::NL * P::LS;                              // the halo line
    __shared__ float s_fs, s_ibeta;
    const int sub = blockIdx.y, yb = yb0 + blockIdx.x, y0 = yb * P::NL;
    const bool main_thread = threadIdx.x < P::NL * P::NT;
    const int l = main_thread ? threadIdx.x / P::NT : P::NL;       // threads beyond the NL * NT tasks: the halo line (64 of them at least)
    const int t = main_thread ? threadIdx.x % P::NT : threadIdx.x - P::NL * P::NT;
    float2* myline = s + l * P::LS;
    const zscal z = sc[sub];
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int g = 0; g < P::G; g++) tot += fs_partial[(size_t)sub * P::G + g];
        s_fs = (float)(tot / ((double)P::L * (double)P::L));
        s_ibeta = 1.0f / vs_scale<P>(fs_partial, gridDim.y, sub, z);
    }
    const float sn2 = z.sn * z.sn, sr2 = z.sr * z.sr, fn2 = z.fn * z.fn, fr2 = z.fr * z.fr;
    const float fD = z.fr * z.fn / sqrtf(sn2 * fr2 + sr2 * fn2);
    const int sy = sub / o.nsx, sx = sub - sy * o.nsx;
    // this thread's row in the frame (or none)
    const int y = y0 + l;
    const bool row_out = main_thread && y >= o.border && y < o.border + o.size && sy * o.size + (y - o.border) < o.ny;
    const size_t rowbase = row_out ? (size_t)(sy * o.size + (y - o.border)) * o.nx + (size_t)sx * o.size : 0;
    float2 X[P::N2], x[P::N1];
    float vs[P::N1];
    // (D, V_S): D goes out at once, V_S stays in registers
    if (threadIdx.x < P::THREADS) load_u_pair<P>(UD, UVS, sub, yb, s);
    __syncthreads();
    if (main_thread && t < P::N1) {
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) X[k2] = myline[(P::N2 + 1) * t + k2];
        inv_step2<P>(X, myline, t, tw);
    }
    __syncthreads();
    const float ibeta = s_ibeta;
    if (main_thread && t < P::N2) {
        inv_step1<P>(myline, t, x);
#pragma unroll
        for (int n1 = 0; n1 < P::N1; n1++) {
            const int xi = P::N2 * n1 + t - o.border;
            vs[n1] = x[n1].y * inv_n2 * ibeta;
            if (row_out && xi >= 0 && xi < o.size && sx * o.size + xi < o.nx) o.D[rowbase + xi] = x[n1].x * inv_n2 / fD;
        }
    }
    __syncthreads();
    // (Sn, Sr): the block's rows + the row above it (row L - 1 above row 0: np.roll)
    if (threadIdx.x < P::THREADS) load_u_pair<P>(USn, USr, sub, yb, s);
    if (!main_thread) load_halo_pair<P>(HSn, HSr, sub, yb ? yb - 1 : P::LB - 1, hline, threadIdx.x - P::NL * P::NT, (int)blockDim.x - P::NL * P::NT);
    __syncthreads();
    if (t < P::N1) {                                              // the extra threads take part with the halo line
#pragma unroll
        for (int k2 = 0; k2 < P::N2; k2++) X[k2] = myline[(P::N2 + 1) * t + k2];
        inv_step2<P>(X, myline, t, tw);
    }
    __syncthreads();
    if (t < P::N2) inv_step1<P>(myline, t, x);
    __syncthreads();
    // Sn, Sr (scaled) in natural order for the neighbour reads
    if (t < P::N2) {
#pragma unroll
        for (int n1 = 0; n1 < P::N1; n1++) { x[n1] = cscale(x[n1], inv_n2); myline[npos<P>(P::N2 * n1 + t)] = x[n1]; }
    }
    __syncthreads();
    const float fs = s_fs;
    if (row_out && t < P::N2) {
        const float2* upline = l ? s + (l - 1) * P::LS : hline;
#pragma unroll
        for (int n1 = 0; n1 < P::N1; n1++) {
            const int xx = P::N2 * n1 + t, xi = xx - o.border;
            if (xi < 0 || xi >= o.size || sx * o.size + xi >= o.nx) continue;
            const float2 c = x[n1];                                         // (Sn, Sr) here
            const float sval = c.x - c.y;                                   // S = Sn - Sr
            const int xm = xx == 0 ? P::L - 1 : xx - 1;
            const float2 up = upline[npos<P>(xx)], lf = myline[npos<P>(xm)];
            const float dSndy = c.x - up.x, dSndx = c.x - lf.x, dSrdy = c.y - up.y, dSrdx = c.y - lf.y;
            const float vast = z.dx * z.dx * (dSndx * dSndx + dSrdx * dSrdx) + z.dy * z.dy * (dSndy * dSndy + dSrdy * dSrdy);
            if (o.S) o.S[rowbase + xi] = sval;
            o.Scorr[rowbase + xi] = sval / sqrtf(vs[n1] + vast);
            o.Fpsf[rowbase + xi] = sval / fs;
            o.Fpsferr[rowbase + xi] = sqrtf(fmaxf(vs[n1], 0.f)) / fs;
        }
    }
}

// ---- host side --------------------------------------------------------------------------------
struct state {
    int L; float2* d_tw;
};

template <class P>
static int run(bbx_ctx* ctx, state* st, int ny, int nx, int size, int border, const float* d_new, const float* d_ref,
               const float* d_sig_new, const float* d_sig_ref, const float* d_psf_n, const float* d_psf_r, int S, const float* h_scal,
               float* d_D, float* d_S, float* d_Scorr, float* d_Fpsf, float* d_Fpsferr, hipStream_t s) {
    const int nsy = ny / size, nsx = nx / size, nsub = nsy * nsx;
    int rc;
    const size_t unit = (size_t)nsub * P::UNIT;                       // elements of one T / U / C array
    const size_t hunit = (size_t)nsub * P::LB * P::HP;                // elements of one halo array
    // 4 T + 4 U + 6 C arrays + scalars + F_S partial sums
    const size_t bytes = (14 * unit + 2 * hunit) * sizeof(float2) + (size_t)nsub * sizeof(zscal) + 3 * (size_t)nsub * P::G * sizeof(double) + 4096;
    char* ws = (char*)bbx_ws(ctx, WS_CAND, bytes, &rc); if (rc) return rc;
    float2* arr[14]; for (int i = 0; i < 14; i++) arr[i] = (float2*)ws + (size_t)i * unit;
    float2 *HSn = (float2*)ws + 14 * unit, *HSr = HSn + hunit;
    char* p = ws + (14 * unit + 2 * hunit) * sizeof(float2);
    zscal* d_sc = (zscal*)p; p += (size_t)nsub * sizeof(zscal);
    p = (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    double* fs_partial = (double*)p;
    float2 *T0 = arr[0], *T1 = arr[1], *T2 = arr[2], *T3 = arr[3], *U0 = arr[4], *U1 = arr[5], *U2 = arr[6], *U3 = arr[7];
    float2 *cA = arr[8], *cB = arr[9], *cKn = arr[10], *cKr = arr[11], *cK2n = arr[12], *cK2r = arr[13];
    // (pageable host source: the runtime stages it before the call returns)
    BBX_HIP(hipMemcpyAsync(d_sc, h_scal, (size_t)nsub * sizeof(zscal), hipMemcpyHostToDevice, s));
    const float2* tw = st->d_tw;
    const size_t lds_tw = (size_t)P::L * sizeof(float2);
    const size_t lds = (size_t)P::NL * P::LS * sizeof(float2) + lds_tw,
                 lds_fin = (size_t)(P::NL + 1) * P::LS * sizeof(float2);
    static bool attr_set = false;
    if (!attr_set) {
        BBX_HIP(hipFuncSetAttribute((const void*)k_psf_cols<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_psf_rows<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_cols_fwd<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_img_rows<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_img_cols<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_var_cols<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_final_rows<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fin));
        attr_set = true;
    }
    const float inv_n2 = 1.0f / ((float)P::L * (float)P::L);
    const dim3 gcol(P::G, nsub), grow(P::LB, nsub), blk(P::THREADS);
    bbx_prof_start(ctx, BBX_PROF_ZOGY, s);
    // PSF side: U0 = kn (cols^-1), U1 = kr; T0 = (kr^2)^ rows, T1 = (kn^2)^ rows
    hipLaunchKernelGGL(k_psf_cols<P>, gcol, blk, lds, s, d_psf_n, d_psf_r, S, d_sc, tw, cA, cB, cKn, cKr, U0, U1, fs_partial);
    hipLaunchKernelGGL(k_psf_rows<P>, grow, blk, lds, s, U1, U0, inv_n2, tw, T0, T1);
    hipLaunchKernelGGL(k_cols_fwd<P>, gcol, blk, lds, s, T0, tw, cK2r);
    hipLaunchKernelGGL(k_cols_fwd<P>, gcol, blk, lds, s, T1, tw, cK2n);
    // image side
    frame_args fa; fa.a = d_new; fa.b = d_ref; fa.sa = nullptr; fa.sb = nullptr; fa.ny = ny; fa.nx = nx; fa.size = size; fa.border = border; fa.nsx = nsx;
    fa.vec4 = (size % 4 == 0 && border % 4 == 0 && nx % 4 == 0 && P::N2 % 4 == 0 && ((uintptr_t)d_new | (uintptr_t)d_ref | (uintptr_t)d_sig_new | (uintptr_t)d_sig_ref) % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_img_rows<P>, grow, blk, lds, s, fa, tw, T0, T1);
    fa.sa = d_sig_new; fa.sb = d_sig_ref;
    hipLaunchKernelGGL(k_img_rows<P>, grow, blk, lds, s, fa, tw, T2, T3);
    hipLaunchKernelGGL(k_img_cols<P>, gcol, blk, lds, s, T0, T1, cA, cB, cKn, cKr, tw, U0, U1, U2, HSn, HSr);      // D, Sn, Sr
    hipLaunchKernelGGL(k_var_cols<P>, gcol, blk, lds, s, T2, T3, cK2n, cK2r, tw, U3, d_sc, fs_partial);                    // V_S
    out_args oa; oa.D = d_D; oa.S = d_S; oa.Scorr = d_Scorr; oa.Fpsf = d_Fpsf; oa.Fpsferr = d_Fpsferr;
    oa.ny = ny; oa.nx = nx; oa.size = size; oa.border = border; oa.nsx = nsx;
    oa.vec4 = (size % 4 == 0 && border % 4 == 0 && nx % 4 == 0 && P::L % 4 == 0 &&
               ((uintptr_t)d_D | (uintptr_t)d_S | (uintptr_t)d_Scorr | (uintptr_t)d_Fpsf | (uintptr_t)d_Fpsferr) % 16 == 0) ? 1 : 0;
    // row blocks that hold output rows: border .. border + size - 1
    const int yb0 = border / P::NL, yb1 = (border + size - 1) / P::NL;
    const dim3 gfin(yb1 - yb0 + 1, nsub);
    bbx_prof_stop(ctx, s);
    bbx_prof_start(ctx, BBX_PROF_ZOGY_FINAL, s);
    hipLaunchKernelGGL(k_final_rows<P>, gfin, dim3(((P::NL * P::NT + 64 + 63) / 64) * 64), lds_fin, s, U0, U3, U1, U2, HSn, HSr, d_sc, fs_partial, inv_n2, tw, oa, yb0);
    bbx_prof_stop(ctx, s);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

}  // namespace z2

void bbx_zogy2_release(bbx_ctx* ctx) {
    if (!ctx || !ctx->zogy2_state) return;
    z2::state* st = (z2::state*)ctx->zogy2_state;
    if (st->d_tw) (void)hipFree(st->d_tw);
    free(st);
    ctx->zogy2_state = nullptr;
}

extern "C" int bbx_zogy_frame_supported(int L) {
    return (L == 1400 || L == 140 || L == 128 || L == 64) ? 1 : 0;
}

extern "C" int bbx_zogy_frame(bbx_ctx* ctx, int ny, int nx, int size, int border, const float* d_new, const float* d_ref,
                              const float* d_sig_new, const float* d_sig_ref, const float* d_psf_n, const float* d_psf_r, int S,
                              const float* h_scal, float* d_D, float* d_S, float* d_Scorr, float* d_Fpsf, float* d_Fpsferr,
                              void* stream) {
    if (!ctx || !d_new || !d_ref || !d_sig_new || !d_sig_ref || !d_psf_n || !d_psf_r || !h_scal || !d_D || !d_Scorr || !d_Fpsf || !d_Fpsferr)
        return BBX_ERR_ARG;
    if (size < 1 || border < 0 || ny < size || nx < size || ny % size || nx % size || S < 1) return BBX_ERR_ARG;
    const int L = size + 2 * border;
    if (!bbx_zogy_frame_supported(L) || S > L || (ny / size) * (nx / size) > 4096) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->zogy2_state) {
        ctx->zogy2_state = calloc(1, sizeof(z2::state));
        if (!ctx->zogy2_state) return BBX_ERR_NOMEM;
    }
    z2::state* st = (z2::state*)ctx->zogy2_state;
    if (st->L != L) {
        if (st->d_tw) { BBX_HIP(hipDeviceSynchronize()); BBX_HIP(hipFree(st->d_tw)); st->d_tw = nullptr; }
        float2* h = (float2*)malloc((size_t)L * sizeof(float2));
        if (!h) return BBX_ERR_NOMEM;
        for (int m = 0; m < L; m++) {
            const double a = -2.0 * M_PI * (double)m / (double)L;
            h[m] = make_float2((float)cos(a), (float)sin(a));
        }
        hipError_t e = hipMalloc((void**)&st->d_tw, (size_t)L * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(st->d_tw, h, (size_t)L * sizeof(float2), hipMemcpyHostToDevice);
        free(h);
        if (e != hipSuccess) return bbx_hip_fail(ctx, e, "twiddle table", __LINE__);
        st->L = L;
    }
    if (ctx->zogy_core == 1 && bbx_zogy3_supported(L))
        return bbx_zogy3_run(ctx, st->d_tw, L, ny, nx, size, border, d_new, d_ref, d_sig_new, d_sig_ref, d_psf_n, d_psf_r, S, h_scal, d_D, d_S,
                             d_Scorr, d_Fpsf, d_Fpsferr, s);
#define Z2_RUN(N1, N2) return z2::run<z2::Plan<N1, N2>>(ctx, st, ny, nx, size, border, d_new, d_ref, d_sig_new, d_sig_ref, d_psf_n, d_psf_r, S, \
                                                        h_scal, d_D, d_S, d_Scorr, d_Fpsf, d_Fpsferr, s)
    switch (L) {
        case 1400: Z2_RUN(35, 40);
        case 140: Z2_RUN(10, 14);
        case 128: Z2_RUN(8, 16);
        case 64: Z2_RUN(8, 8);
    }
    return BBX_ERR_ARG;
}
