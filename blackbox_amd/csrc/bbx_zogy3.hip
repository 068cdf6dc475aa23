// bbx_zogy3.hip -- ZOGY on whole frames (bbx_zogy_frame): a 2-D FFT pipeline written for this shape.
// Kernel sequence, data layouts (T / U tiles, C coefficient arrays, halo rows) and algebra: DESIGN.md section 4b.
// (The first core of round 2, bbx_zogy2.hip -- lines of 1400 = 35 * 40 points in two steps of register DFTs, one thread per
// sub-transform, ~210 VGPRs, 8 waves per CU, 11.3 ms per frame -- was removed in round 3.)
//
//   1-D transform: L = R0 R1 [R2 [R3]] (1400 = 5 * 7 * 5 * 8); a step is L / R butterflies of radix R
//              per line (decimation in frequency, in place: a butterfly reads and writes the same
//              R positions), any thread takes any butterfly: 45-77 VGPRs, 24 waves per CU, all data
//              movement is loops over the workgroup.  The spectrum is left in the digit-reversed
//              order of the in-place algorithm (pos(k) below); the inverse runs the steps backwards.
//              The butterflies (bbx_fft_gen.h, tools/gen_fft.py) work on (re, im) pairs: packed fp32
//              instructions, the half-crossing operations spelled out with op_sel / neg modifiers.
//
#include "bbx_common.h"
#ifndef Z3_NO_CONTRACT
#pragma clang fp contract(fast)      // the transforms are compared within a tolerance, not bit by bit: let mul + add fuse
#endif
#include "bbx_fft_gen.h"
#include "bbx_spline.h"
#include <math.h>
#include <stdlib.h>

namespace z3 {

struct zscal { float sn, sr, fn, fr, dx, dy; };

// Diagnostic builds only (tools/exp/zvar.sh -DZ3_STAMPS; the Makefile never defines it): thread 0 of every workgroup writes
// the shader clock at the phase boundaries of the kernel into a buffer set by bbx_z3_stamps (tools/dbg/z3_stamps.py).
#ifdef Z3_STAMPS
#define Z3_STAMP_WGS 24576
__device__ unsigned long long* g_z3_stamps;
#define ZSTAMP(kid, i)                                                                                                     \
    do { if (threadIdx.x == 0 && g_z3_stamps && blockIdx.x < Z3_STAMP_WGS)                                                \
            g_z3_stamps[((size_t)(kid) * Z3_STAMP_WGS + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define ZSTAMP_HEAD(kid)                                                                                                   \
    do { if (threadIdx.x == 0 && g_z3_stamps && blockIdx.x < Z3_STAMP_WGS) {                                              \
            unsigned long long* q_ = g_z3_stamps + ((size_t)(kid) * Z3_STAMP_WGS + blockIdx.x) * 16;                       \
            q_[14] = __builtin_amdgcn_s_memrealtime(); q_[15] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20); \
            q_[0] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define ZSTAMP(kid, i)
#define ZSTAMP_HEAD(kid)
#endif

constexpr int cmax(int a, int b) { return a > b ? a : b; }
#ifndef Z3_SPREAD
#define Z3_SPREAD 1
#endif
#ifndef Z3_LSMOD
#define Z3_LSMOD (Z3_SPREAD ? 16 : 8)
#endif
constexpr int line_stride(int lp) { int s = lp; while ((2 * s) % 64 != Z3_LSMOD) s++; return s; }

#ifndef Z3_THREADS
#define Z3_THREADS 768         // two workgroups per CU: 24 waves, <= 85 VGPRs (8 parked float2 per thread in the column kernels)
#endif
#ifndef Z3_FIN_THREADS
#define Z3_FIN_THREADS 768     // k_final_rows: 5 lines = 64 KB, two workgroups per CU
#endif
#ifndef Z3_NL
#define Z3_NL 4                // 4 x 4 tiles of float2 = one 128-byte line; LDS: 4 lines = 51 KB per workgroup
#endif
#ifndef Z3_LOADS_T
#define Z3_LOADS_T 4
#endif
#ifndef Z3_LOADS_U
#define Z3_LOADS_U 2
#endif
#ifndef Z3_IMG_LOADS
#define Z3_IMG_LOADS 3
#endif
#ifndef Z3_LIGHT_THREADS
#define Z3_LIGHT_THREADS Z3_THREADS   // k_psf_rows, k_img_rows (nothing parked in registers)
#endif
#ifndef Z3_VAR_THREADS
#define Z3_VAR_THREADS 768     // k_var_cols keeps two spectra in registers: 2 x 8 float2 per thread, 85 VGPRs without spills
#define Z3_VAR_MINW 6          // since the transforms work on (re, im) pairs (512 threads / 128 VGPRs before)
#endif
#ifndef Z3_MINW
#define Z3_MINW 6              // waves per SIMD the register allocation must allow (two workgroups of 768 threads)
#endif
#ifndef Z3_MINW_LIGHT
#define Z3_MINW_LIGHT Z3_MINW  // the same for the kernels without parked registers (k_psf_rows, k_cols_fwd, k_img_rows)
#endif
#ifndef Z3_MINW_PSF
#define Z3_MINW_PSF Z3_MINW
#endif
#ifndef Z3_FIN_MINW
#define Z3_FIN_MINW 6
#endif
#ifndef Z3_PLAN1400
#define Z3_PLAN1400 5, 7, 5, 8
#endif
#ifndef Z3_KWIN_TOL
#define Z3_KWIN_TOL 1e-6       // share of the energy of k_n, k_r (per sub-image) allowed outside the row window: V(S) is off by that share at most
#endif
#ifndef Z3_TWL
#define Z3_TWL 1               // twiddle table in LDS (1) or read from global memory (0: three 8-byte reads per point
                               // and transform through the vector memory path, +9 % on the whole call)
#endif

template <int R0_, int R1_, int R2_ = 1, int R3_ = 1> struct Plan {
    static constexpr int R0 = R0_, R1 = R1_, R2 = R2_, R3 = R3_;
    static constexpr int L = R0_ * R1_ * R2_ * R3_, H = L / 2 + 1;
    static constexpr int NL = L >= 512 ? Z3_NL : 16;                // lines per workgroup
    static constexpr int THREADS = L >= 512 ? Z3_THREADS : 256, FIN_THREADS = L >= 512 ? Z3_FIN_THREADS : 256;
    static constexpr int LIGHT_THREADS = L >= 512 ? Z3_LIGHT_THREADS : 256;
    static constexpr int VAR_THREADS = L >= 512 ? Z3_VAR_THREADS : 256, VAR_MINW = L >= 512 ? Z3_VAR_MINW : 1;
    static constexpr int G = (H + NL - 1) / NL, HP = G * NL;        // column groups, padded half-spectrum width
    static constexpr int LP = L + L / 8 + 1;                        // padded line: one pad per 8 entries
    static constexpr int LS = line_stride(LP);
    static constexpr int LB = (L + NL - 1) / NL;                    // row blocks of NL rows
    static constexpr size_t UNIT = (size_t)LB * NL * HP;            // elements of one T / U / C array per sub-image
    static constexpr int TWL = Z3_TWL;
    static constexpr int MINW_LIGHT = L >= 512 ? Z3_MINW_LIGHT : 1;
    static constexpr int MINW_PSF = L >= 512 ? Z3_MINW_PSF : 1;
    static constexpr int MINW = L >= 512 ? Z3_MINW : 1, FIN_MINW = L >= 512 ? Z3_FIN_MINW : 1;
};

typedef bbx_v2f __attribute__((may_alias)) v2f_a;     // a float2 of the line buffers read / written as one (re, im) pair
__device__ __forceinline__ bbx_v2f to_v(float2 a) { return bbx_v2f{a.x, a.y}; }
__device__ __forceinline__ float2 from_v(bbx_v2f a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return from_v(bbx_cmul(to_v(a), to_v(b))); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return from_v(bbx_cmulc(to_v(a), to_v(b))); }   // a * conj(b)
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

// Streaming accesses of the big arrays (frames, T / U tiles, the PSF spectra: written once, read once, 0.5 GB each): non-temporal.
// Measured with the kernels' own access shape (tools/exp/tile_bw.hip: a workgroup reads 350 pieces of 128 bytes at a 22.5 KB
// stride and writes 45 KB contiguously, 2 workgroups per CU): plain loads + plain stores 3.5 TB/s, plain loads + non-temporal
// stores 6.1, both non-temporal 6.9 (a contiguous copy of the same shape: 5.9) -- plain stores keep their lines in the
// XCD's L2 and push out the lines the scattered reads still need.
#ifndef Z3_NT
#define Z3_NT 1
#endif
typedef float z3_v4 __attribute__((ext_vector_type(4)));
typedef float z3_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_nt(const float4* p) {
#if Z3_NT
    const z3_v4 v = __builtin_nontemporal_load(reinterpret_cast<const z3_v4*>(p)); return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_nt(float4* p, float4 v) {
#if Z3_NT
    __builtin_nontemporal_store(z3_v4{v.x, v.y, v.z, v.w}, reinterpret_cast<z3_v4*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_nt(float* p, float v) {
#if Z3_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

__device__ __forceinline__ int npos(int n) { return n + (n >> 3); }                       // LDS position of entry n of a line
// position of spectral index k after the forward transform (digit reversal of the in-place passes)
template <class P> __device__ __forceinline__ int ppos(int k) {
    int p = (k % P::R0) * (P::L / P::R0); k /= P::R0;
    p += (k % P::R1) * (P::L / (P::R0 * P::R1)); k /= P::R1;
    if (P::R2 > 1) { p += (k % P::R2) * (P::L / (P::R0 * P::R1 * P::R2)); k /= P::R2; }
    if (P::R3 > 1) p += k;
    return npos(p);
}

// one radix-R pass over blocks of B entries of NLINES lines (B / R = M butterflies per block).  (A thread taking the same
// butterfly of two neighbouring lines -- positions and twiddles worked out once for both -- needs ~50 more registers than
// the 85 that two workgroups of 768 threads leave: it spills and is 15 % slower.)  The
// butterfly's R positions are base + r M: with the padding they stay an affine function of r when M
// is a multiple of 8, or when M = 1 and the block starts at a multiple of 8 (constant offsets).
template <class P, int R, int B, bool INV, int NLINES, int ZP = 0>
__device__ __forceinline__ void fft_step(float2* s, const float2* __restrict__ tw, int zw = 0) {
    constexpr int M = B / R, PER = P::L / R, NTASK = NLINES * PER, TWS = P::L / B;
    constexpr bool AFF8 = M % 8 == 0, AFF1 = M == 1 && (8 % R == 0 || R % 8 == 0);
    // Lanes -> butterflies.  SPREAD (M a multiple of 8, groups of 4 lines): 8 consecutive lanes take 8 consecutive
    // butterflies of one line, the next 8 lanes the same butterflies of the next line.  With the line stride = 8 entries
    // mod 32 the 32 lanes one ds_read_b64 group serves then touch 32 different 8-byte banks (a run of 32 consecutive
    // entries of ONE line covers 35 padded positions: a two-way conflict on every read), and their twiddle reads are the
    // same 8 addresses for the 4 lines.
    constexpr bool SPREAD = Z3_SPREAD && AFF8 && PER % 8 == 0 && NLINES >= 4;
    constexpr int NSPREAD = SPREAD ? (NLINES / 4) * 4 * PER : 0;
    for (int task = threadIdx.x; task < NTASK; task += blockDim.x) {
        int l, j;
        if (SPREAD && task < NSPREAD) {
            const int lg = NLINES >= 8 ? task / (4 * PER) : 0, t = task - lg * (4 * PER);
            l = lg * 4 + ((t >> 3) & 3); j = ((t >> 5) << 3) | (t & 7);
        } else { l = task / PER; j = task - l * PER; }
        const int b = j / M, m = j - b * M;
        // lines that are zero outside their first and last [zw] entries (PSF stamps, windowed kernels): a butterfly of the
        // first pass (ZP 1) or the second (ZP 2) whose inputs all lie in the zero stretch leaves its zeros as they are
        if (ZP == 1 && zw > 0 && m >= zw && m < M - zw) continue;
        if (ZP == 2 && zw > 0 && m >= zw && m + (R - 1) * M < B - zw) continue;
        const int base = b * B + m;
        v2f_a* line = reinterpret_cast<v2f_a*>(s + l * P::LS + ((AFF8 || AFF1) ? npos(base) : 0));
        const v2f_a* twv = reinterpret_cast<const v2f_a*>(tw);
        bbx_v2f u[R];
#pragma unroll
        for (int r = 0; r < R; r++) u[r] = line[AFF8 ? r * (M + M / 8) : AFF1 ? r + r / 8 : npos(base + r * M)];
        // twiddles W^(m k TWS), k = 1 .. R-1: one table read, the powers by multiplication (Z3_TWPOW), or R-1 reads
        bbx_v2f w[R];
        if constexpr (M > 1) {
#ifdef Z3_TWPOW
            w[1] = twv[m * TWS];
#pragma unroll
            for (int k = 2; k < R; k++) w[k] = (k % 2 == 0) ? bbx_cmul(w[k / 2], w[k / 2]) : bbx_cmul(w[k - 1], w[1]);
#else
#pragma unroll
            for (int k = 1; k < R; k++) w[k] = twv[m * k * TWS];
#endif
        }
        if (!INV) {
            bbx_dft<R>::fwd(u);
            if constexpr (M > 1) {
#pragma unroll
                for (int k = 1; k < R; k++) u[k] = bbx_cmul(u[k], w[k]);
            }
        } else {
            if constexpr (M > 1) {
#pragma unroll
                for (int k = 1; k < R; k++) u[k] = bbx_cmulc(u[k], w[k]);
            }
            bbx_dft<R>::inv(u);
        }
#pragma unroll
        for (int r = 0; r < R; r++) line[AFF8 ? r * (M + M / 8) : AFF1 ? r + r / 8 : npos(base + r * M)] = u[r];
    }
}
// forward: natural order -> spectrum at ppos; inverse: back (unnormalised).  Barriers after every pass.
// (zw > 0: the lines are exactly zero outside their first and last zw entries)
template <class P, int NLINES = P::NL> __device__ __forceinline__ void fft_fwd(float2* s, const float2* tw, int zw = 0) {
#ifdef Z3_SKIP_FFT
    __syncthreads(); return;
#endif
#ifdef Z3_NO_ZSKIP
    zw = 0;
#endif
    fft_step<P, P::R0, P::L, false, NLINES, 1>(s, tw, zw); __syncthreads();
    fft_step<P, P::R1, P::L / P::R0, false, NLINES, 2>(s, tw, zw); __syncthreads();
    if constexpr (P::R2 > 1) { fft_step<P, P::R2, P::L / (P::R0 * P::R1), false, NLINES>(s, tw); __syncthreads(); }
    if constexpr (P::R3 > 1) { fft_step<P, P::R3, P::L / (P::R0 * P::R1 * P::R2), false, NLINES>(s, tw); __syncthreads(); }
}
template <class P, int NLINES = P::NL> __device__ __forceinline__ void fft_inv(float2* s, const float2* tw) {
#ifdef Z3_SKIP_FFT
    __syncthreads(); return;
#endif
    if constexpr (P::R3 > 1) { fft_step<P, P::R3, P::L / (P::R0 * P::R1 * P::R2), true, NLINES>(s, tw); __syncthreads(); }
    if constexpr (P::R2 > 1) { fft_step<P, P::R2, P::L / (P::R0 * P::R1), true, NLINES>(s, tw); __syncthreads(); }
    fft_step<P, P::R1, P::L / P::R0, true, NLINES>(s, tw); __syncthreads();
    fft_step<P, P::R0, P::L, true, NLINES>(s, tw); __syncthreads();
}

// per-workgroup tables behind the line buffers: the twiddles (LDS copy, or the global table) and the
// position of every spectral index (ppos costs ~25 VALU instructions; the packing / splitting loops need two per entry)
struct aux_t { const float2* tw; const unsigned short* pp; };
template <class P> __device__ __forceinline__ aux_t aux_setup(float2* after_lines, const float2* __restrict__ twg) {
    aux_t a;
    a.tw = twg;
    if (P::TWL) {
        for (int e = threadIdx.x; e < P::L; e += blockDim.x) after_lines[e] = twg[e];
        a.tw = after_lines;
        after_lines += P::L;
    }
    unsigned short* pp = reinterpret_cast<unsigned short*>(after_lines);
    for (int e = threadIdx.x; e < P::L; e += blockDim.x) pp[e] = (unsigned short)ppos<P>(e);
    a.pp = pp;
    return a;                                                       // visible after the caller's next barrier
}
template <class P> constexpr size_t aux_bytes() { return (P::TWL ? (size_t)P::L * sizeof(float2) : 0) + (((size_t)P::L * 2 + 15) & ~(size_t)15); }

// Workgroup -> (x, y) of an nx * ny task grid, launched as a 1-D grid of a multiple of 8 workgroups.
// Workgroups are dealt round-robin over the 8 XCDs (observed, for speed only): the tasks are numbered so
// that the ones an XCD runs at the same time are neighbours -- they read adjacent 288-byte pieces
// of the same 128-byte lines, which then come from that XCD's L2 instead of HBM a second time.
#ifndef Z3_NO_XCD
__device__ __forceinline__ int xcd_task() { return (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8); }
#else
__device__ __forceinline__ int xcd_task() { return (int)blockIdx.x; }
#endif
#define WG_TASK(nx_, ny_, x_, y_)                       \
    const int task_ = xcd_task();                       \
    if (task_ >= (nx_) * (ny_)) return;                 \
    const int x_ = task_ % (nx_), y_ = task_ / (nx_);
// The row kernels (image cut -> T tiles, U tiles -> output frames) take their tasks in launch order: their neighbours share
// no lines (a task reads and writes whole rows of its own), and the XCD-aware order costs them 1-4 % (measured, round 3).
#ifndef Z3_ROWS_XCD
#define WG_TASK_ROWS(nx_, ny_, x_, y_)                  \
    const int task_ = (int)blockIdx.x;                  \
    if (task_ >= (nx_) * (ny_)) return;                 \
    const int x_ = task_ % (nx_), y_ = task_ / (nx_);
#else
#define WG_TASK_ROWS WG_TASK
#endif
static inline dim3 grid8(int nx, int ny) { return dim3((unsigned)(((size_t)nx * ny + 7) / 8 * 8)); }

// ---- data movement (loops over the workgroup) ---------------------------------------------------
// The final kernel's workgroups take chunks of consecutive row blocks of one sub-image: chunk c of nch covers the blocks
// start[c] .. start[c + 1] - 1 (the same cut for every sub-image; host: chunk_plan); only the row above a chunk's first block
// travels through the halo arrays H[sub][c][HP] (inside a chunk the row above a block is the last row of the block before
// it, still in LDS).
struct chunk_args { const int* start; int nch; };
__device__ __forceinline__ int chunk_start(const chunk_args& ch, int c) { return ch.start[c]; }
// LDS (natural order, after an inverse column pass) -> U tiles U[sub][g][y][l]; halo rows on request
template <class P> __device__ __forceinline__ void store_u(const float2* s, float2* __restrict__ U, int sub, int g, float2* __restrict__ halo = nullptr,
                                                            chunk_args ch = chunk_args{nullptr, 0}) {
    float2* base = U + (size_t)sub * P::UNIT + (size_t)g * P::L * P::NL;
    // two neighbouring lines per item: one 16-byte store
    constexpr int HL = P::NL / 2;
    for (int e = threadIdx.x; e < P::L * HL; e += blockDim.x) {
        const int y = e / HL, l = 2 * (e - y * HL), py = npos(y);
        const float2 v0 = s[l * P::LS + py], v1 = s[(l + 1) * P::LS + py];
        st_nt(reinterpret_cast<float4*>(base + (size_t)y * P::NL + l), make_float4(v0.x, v0.y, v1.x, v1.y));
    }
    if (halo) {
        for (int e = threadIdx.x; e < ch.nch * HL; e += blockDim.x) {
            const int c = e / HL, l = 2 * (e - c * HL), yb = chunk_start(ch, c);
            const int y = yb ? yb * P::NL - 1 : P::L - 1, py = npos(y);                    // row L - 1 above row 0: np.roll
            const float2 v0 = s[l * P::LS + py], v1 = s[(l + 1) * P::LS + py];
            *reinterpret_cast<float4*>(halo + ((size_t)sub * ch.nch + c) * P::HP + g * P::NL + l) = make_float4(v0.x, v0.y, v1.x, v1.y);
        }
    }
}
// The matched-filter kernels k_n, k_r live around the origin: of the inverse column pass of k^ only the rows
// y < wh and y >= L - wh leave the workgroup (U tiles, same layout; the other rows are never read).  The energy
// the dropped rows hold is summed next to the total (Parseval along x: row y of this half spectrum carries
// sum_x k(y, x)^2), so that the host-chosen window can be checked on the device.
template <class P> __device__ __forceinline__ void store_u_win(const float2* s, float2* __restrict__ U, int sub, int g, int wh, double& e_all, double& e_out) {
    float2* base = U + (size_t)sub * P::UNIT + (size_t)g * P::L * P::NL;
    constexpr int HL = P::NL / 2;
    float ea = 0.f, eo = 0.f;
    for (int e = threadIdx.x; e < P::L * HL; e += blockDim.x) {
        const int y = e / HL, l = 2 * (e - y * HL), py = npos(y);
        const float2 v0 = s[l * P::LS + py], v1 = s[(l + 1) * P::LS + py];
        const float en = (v0.x * v0.x + v0.y * v0.y) + (v1.x * v1.x + v1.y * v1.y);
        ea += en;
        if (y < wh || y >= P::L - wh) st_nt(reinterpret_cast<float4*>(base + (size_t)y * P::NL + l), make_float4(v0.x, v0.y, v1.x, v1.y));
        else eo += en;
    }
    e_all += (double)ea; e_out += (double)eo;
}
// T tiles of the row blocks inside the window (rows < wh and >= L - wh; wh a multiple of NL, L a multiple of NL) of
// column group g -> LDS lines; the rows outside the window are zero
template <class P> __device__ __forceinline__ void load_t_lines_win(const float2* T, int sub, int g, float2* s, int wh) {
    const float2* src = T + (size_t)sub * P::UNIT + (size_t)g * P::NL * P::NL;
    constexpr int TV = P::NL * P::NL / 2;
    const int wb = wh / P::NL, NVW = 2 * wb * TV;
    for (int e = threadIdx.x; e < P::NL * (P::L - 2 * wh); e += blockDim.x) {
        const int l = e / (P::L - 2 * wh), y = wh + e - l * (P::L - 2 * wh);
        s[l * P::LS + npos(y)] = make_float2(0.f, 0.f);
    }
    for (int e = threadIdx.x; e < NVW; e += blockDim.x) {
        const int ybw = e / TV, j = e - ybw * TV, yb = ybw < wb ? ybw : P::LB - 2 * wb + ybw;
        const float4 v = ld_nt(reinterpret_cast<const float4*>(src + (size_t)yb * P::HP * P::NL + 2 * j));
        const int l = (2 * j) / P::NL, y = yb * P::NL + (2 * j) % P::NL;
        float2* line = s + l * P::LS;
        line[npos(y)] = make_float2(v.x, v.y); line[npos(y + 1)] = make_float2(v.z, v.w);
    }
}
// T tiles T[sub][yb][kx][yi] of column group g -> LDS lines, natural order
template <class P> __device__ __forceinline__ void load_t_lines(const float2* T, int sub, int g, float2* s) {
    static_assert(P::NL % 2 == 0 && P::L % 2 == 0, "even tile side and line length");
    const float2* src = T + (size_t)sub * P::UNIT + (size_t)g * P::NL * P::NL;
    constexpr int TV = P::NL * P::NL / 2, NV = P::LB * TV;          // float4 per tile, per workgroup
    constexpr int TD = Z3_LOADS_T;                                  // 16-byte loads in flight per thread
    for (int e0 = threadIdx.x; e0 < NV; e0 += TD * (int)blockDim.x) {
        float4 v[TD];
#pragma unroll
        for (int i = 0; i < TD; i++) {
            const int e = e0 + i * (int)blockDim.x;
            if (e < NV) { const int yb = e / TV, j = e - yb * TV; v[i] = ld_nt(reinterpret_cast<const float4*>(src + (size_t)yb * P::HP * P::NL + 2 * j)); }
        }
#pragma unroll
        for (int i = 0; i < TD; i++) {
            const int e = e0 + i * (int)blockDim.x;
            if (e < NV) {
                const int yb = e / TV, j = e - yb * TV, l = (2 * j) / P::NL, y = yb * P::NL + (2 * j) % P::NL;
                if (y < P::L) {
                    float2* line = s + l * P::LS;
                    line[npos(y)] = make_float2(v[i].x, v[i].y); line[npos(y + 1)] = make_float2(v[i].z, v[i].w);
                }
            }
        }
    }
}
// entry e (= l * L + p) of the workgroup's scratch in its own, already consumed T tiles
template <class P> __device__ __forceinline__ float2* park_ptr(float2* T, int sub, int g, int e) {
    constexpr int TS = P::NL * P::NL;
    return T + (size_t)sub * P::UNIT + (size_t)(e / TS) * P::HP * P::NL + (size_t)g * TS + e % TS;
}
template <class P> __device__ __forceinline__ void pack_store(float2* line, const unsigned short* pp, int kx, float2 a, float2 b) {
    if (kx >= P::H) return;
    line[pp[kx]] = make_float2(a.x - b.y, a.y + b.x);
    if (kx >= 1 && P::L - kx >= P::H) line[pp[P::L - kx]] = make_float2(a.x + b.y, b.x - a.y);
}
// Hermitian packing of two U half spectra (row block yb) into full complex lines in spectrum order: Z = a + i b
template <class P> __device__ __forceinline__ void load_u_pair(const float2* __restrict__ Ua, const float2* __restrict__ Ub, int sub, int yb, float2* s,
                                                                const unsigned short* pp) {
    constexpr int TV = P::NL * P::NL / 2, NV = P::G * TV;
    const size_t base = (size_t)sub * P::UNIT + (size_t)yb * P::NL * P::NL;
    constexpr int UD = Z3_LOADS_U;                                  // pairs of 16-byte loads in flight per thread
    for (int e0 = threadIdx.x; e0 < NV; e0 += UD * (int)blockDim.x) {
        float4 va[UD], vb[UD];
#pragma unroll
        for (int i = 0; i < UD; i++) {
            const int e = e0 + i * (int)blockDim.x;
            if (e < NV) {
                const int g = e / TV, j = e - g * TV;
                const size_t o = base + (size_t)g * P::L * P::NL + 2 * j;
                va[i] = ld_nt(reinterpret_cast<const float4*>(Ua + o)); vb[i] = ld_nt(reinterpret_cast<const float4*>(Ub + o));
            }
        }
#pragma unroll
        for (int i = 0; i < UD; i++) {
            const int e = e0 + i * (int)blockDim.x;
            if (e < NV) {
                const int g = e / TV, j = e - g * TV, row = (2 * j) / P::NL, l = (2 * j) % P::NL;      // tile entry [row][l], l even
                if (yb * P::NL + row < P::L) {
                    float2* line = s + row * P::LS;
                    pack_store<P>(line, pp, g * P::NL + l, make_float2(va[i].x, va[i].y), make_float2(vb[i].x, vb[i].y));
                    pack_store<P>(line, pp, g * P::NL + l + 1, make_float2(va[i].z, va[i].w), make_float2(vb[i].z, vb[i].w));
                }
            }
        }
    }
}
template <class P> __device__ __forceinline__ void load_halo_pair(const float2* __restrict__ Ha, const float2* __restrict__ Hb, int sub, int yb, float2* line,
                                                                   const unsigned short* pp) {
    const size_t base = ((size_t)sub * P::LB + yb) * P::HP;
    for (int kx = threadIdx.x; kx < P::H; kx += blockDim.x) pack_store<P>(line, pp, kx, Ha[base + kx], Hb[base + kx]);
}
// The same loads split in two: all of a workgroup's 16-byte loads into registers first (issued before a
// transform of other data, so that their latency hides behind it), the LDS stores later.
template <class P, int T> struct t_regs { static constexpr int N = (P::LB * P::NL * P::NL / 2 + T - 1) / T; float4 v[N]; };
// (I0, I1: the part of a thread's loads to issue / store; a kernel short of registers takes the rest in a second round)
template <class P, int T, int I0 = 0, int I1 = 1 << 30> __device__ __forceinline__ void fetch_t_lines(const float2* __restrict__ Tin, int sub, int g, t_regs<P, T>& r) {
    const float2* src = Tin + (size_t)sub * P::UNIT + (size_t)g * P::NL * P::NL;
    constexpr int TV = P::NL * P::NL / 2, NV = P::LB * TV;
#pragma unroll
    for (int i = I0; i < (I1 < t_regs<P, T>::N ? I1 : t_regs<P, T>::N); i++) {
        const int e = (int)threadIdx.x + i * T;
        if (e < NV) { const int yb = e / TV, j = e - yb * TV; r.v[i] = ld_nt(reinterpret_cast<const float4*>(src + (size_t)yb * P::HP * P::NL + 2 * j)); }
    }
}
template <class P, int T, int I0 = 0, int I1 = 1 << 30> __device__ __forceinline__ void pack_t_lines(const t_regs<P, T>& r, float2* s) {
    constexpr int TV = P::NL * P::NL / 2, NV = P::LB * TV;
#pragma unroll
    for (int i = I0; i < (I1 < t_regs<P, T>::N ? I1 : t_regs<P, T>::N); i++) {
        const int e = (int)threadIdx.x + i * T;
        if (e < NV) {
            const int yb = e / TV, j = e - yb * TV, l = (2 * j) / P::NL, y = yb * P::NL + (2 * j) % P::NL;
            if (y < P::L) {
                float2* line = s + l * P::LS;
                line[npos(y)] = make_float2(r.v[i].x, r.v[i].y); line[npos(y + 1)] = make_float2(r.v[i].z, r.v[i].w);
            }
        }
    }
}
template <class P, int T> struct u_regs { static constexpr int N = (P::G * P::NL * P::NL / 2 + T - 1) / T; float4 a[N], b[N]; };
template <class P, int T> __device__ __forceinline__ void fetch_u_pair(const float2* __restrict__ Ua, const float2* __restrict__ Ub, int sub, int yb,
                                                                        u_regs<P, T>& r, int tid = (int)threadIdx.x) {
    constexpr int TV = P::NL * P::NL / 2, NV = P::G * TV;
    const size_t base = (size_t)sub * P::UNIT + (size_t)yb * P::NL * P::NL;
#pragma unroll
    for (int i = 0; i < u_regs<P, T>::N; i++) {
        const int e = tid + i * T;
        if (e < NV) {
            const int g = e / TV, j = e - g * TV;
            const size_t o = base + (size_t)g * P::L * P::NL + 2 * j;
            r.a[i] = ld_nt(reinterpret_cast<const float4*>(Ua + o)); r.b[i] = ld_nt(reinterpret_cast<const float4*>(Ub + o));
        }
    }
}
template <class P, int T> __device__ __forceinline__ void pack_u_pair(const u_regs<P, T>& r, int yb, float2* s, const unsigned short* pp,
                                                                       int tid = (int)threadIdx.x) {
    constexpr int TV = P::NL * P::NL / 2, NV = P::G * TV;
#pragma unroll
    for (int i = 0; i < u_regs<P, T>::N; i++) {
        const int e = tid + i * T;
        if (e < NV) {
            const int g = e / TV, j = e - g * TV, row = (2 * j) / P::NL, l = (2 * j) % P::NL;
            if (yb * P::NL + row < P::L) {
                float2* line = s + row * P::LS;
                pack_store<P>(line, pp, g * P::NL + l, make_float2(r.a[i].x, r.a[i].y), make_float2(r.b[i].x, r.b[i].y));
                pack_store<P>(line, pp, g * P::NL + l + 1, make_float2(r.a[i].z, r.a[i].w), make_float2(r.b[i].z, r.b[i].w));
            }
        }
    }
}
// Hermitian split of a packed transform Z (LDS, spectrum order) -> two half spectra, T tiles of row block yb
template <class P> __device__ __forceinline__ void store_t_split(const float2* s, const unsigned short* pp, float2* __restrict__ Ta, float2* __restrict__ Tb, int sub,
                                                                  int yb) {
    const size_t base = (size_t)sub * P::UNIT + (size_t)yb * P::HP * P::NL;         // the block's entries (kx, row) are contiguous
    // a thread takes a column kx for all rows of the block: one pair of positions, NL / 2 16-byte stores per output array
    for (int kx = threadIdx.x; kx < P::H; kx += blockDim.x) {
        const int pk = pp[kx], pm = pp[kx ? P::L - kx : 0];
        float4* ta = reinterpret_cast<float4*>(Ta + base + (size_t)kx * P::NL);
        float4* tb = reinterpret_cast<float4*>(Tb + base + (size_t)kx * P::NL);
#pragma unroll
        for (int r = 0; r < P::NL; r += 2) {
            const float2 zk0 = s[r * P::LS + pk], zm0 = s[r * P::LS + pm], zk1 = s[(r + 1) * P::LS + pk], zm1 = s[(r + 1) * P::LS + pm];
            st_nt(ta + r / 2, make_float4(0.5f * (zk0.x + zm0.x), 0.5f * (zk0.y - zm0.y), 0.5f * (zk1.x + zm1.x), 0.5f * (zk1.y - zm1.y)));
            st_nt(tb + r / 2, make_float4(0.5f * (zk0.y + zm0.y), 0.5f * (zm0.x - zk0.x), 0.5f * (zk1.y + zm1.y), 0.5f * (zm1.x - zk1.x)));
        }
    }
}
// spectrum entry e = l * L + p of the C layout <-> LDS
// the same loop with a per-thread register array alongside: iteration k of thread t is entry e = t + k THREADS,
// every time, so a value parked in reg[k] meets the same entry again
// (the thread index goes through an empty asm so that the 17 entry addresses are recomputed in every
// loop instead of being kept in registers across the transforms in between)
__device__ __forceinline__ int opaque_tid() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }
// entry e -> (line l, position p).  With 4 lines of a multiple of 8 entries: 8 consecutive entries of one line, then the
// same 8 of the next line (as the butterflies of fft_step: the 32 lanes of a read group touch 32 different banks).  The
// C arrays are stored in this order of e, by the same loops.
template <class P> __device__ __forceinline__ void e2lp(int e, int& l, int& p) {
    if constexpr (Z3_SPREAD && P::NL == 4 && P::L % 8 == 0) { l = (e >> 3) & 3; p = ((e >> 5) << 3) | (e & 7); }
    else { l = e / P::L; p = e - l * P::L; }
}
#define R_LOOP(k, e, l, p)                                                                     \
    for (int t_ = opaque_tid(), once_ = 1; once_; once_ = 0)                                     \
    _Pragma("unroll") for (int k = 0, e, l, p; k < NE; k++)                                      \
        if (e = t_ + k * RT, e2lp<P>(e, l, p), e < P::NL * P::L)                      /* RT: the kernel's thread count */
#define C_LOOP(e, l, p) for (int e = threadIdx.x, l, p; e2lp<P>(e, l, p), e < P::NL * P::L; e += blockDim.x)

// V(S)^ is scaled by a power of two before it shares a transform with D (float32 rounding leaks ~1e-7 of the larger part of a
// complex transform into the smaller; the scale, chosen per sub-image from Parseval sums of k^2, is exact to undo)
__device__ __forceinline__ float vs_scale_of(double a, double b, const zscal& z, double n2) {
    const float level = (float)(((double)z.sn * z.sn * a + (double)z.sr * z.sr * b) / n2);            // Parseval: sum_x k^2 = sum_k |k^|^2 / L^2
    if (!(level > 0.f) || !isfinite(level)) return 1.f;
    return exp2f(-rintf(log2f(level)));
}
// The per-sub-image scalars every later workgroup needs -- F_S, the scale of V(S)^ and its inverse -- from k_psf_cols'
// partial sums, and the check of the row window: one wave per sub-image (first workgroup of k_psf_rows; rounds 3-4: a
// serial loop over the G partial sums in thread 0 of every workgroup of k_var_cols and k_final_rows).
struct sub_scal { float fs, beta, ibeta, pad; };
template <class P> __device__ __forceinline__ void sub_scalars(const double* __restrict__ fs_partial, int nsub, int sub, const zscal& z, bool win,
                                                                sub_scal* __restrict__ out, int32_t* __restrict__ d_err) {
    // lanes 0 .. 63 of one wave
    const int lane = (int)(threadIdx.x & 63);
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int g = lane; g < P::G; g += 64) {
#pragma unroll
        for (int i = 0; i < 5; i++) if (i < 3 || win) v[i] += fs_partial[((size_t)i * nsub + sub) * P::G + g];
    }
#pragma unroll
    for (int i = 0; i < 5; i++) v[i] = wave_sum_f64(v[i]);
    if (lane == 0) {
        const double n2 = (double)P::L * (double)P::L;
        sub_scal r;
        r.fs = (float)(v[0] / n2);
        r.beta = vs_scale_of(v[1], v[2], z, n2);
        r.ibeta = 1.0f / r.beta; r.pad = 0.f;
        out[sub] = r;
        // the rows the window dropped from k_n, k_r must hold nothing a float32 transform could tell from zero
        if (win && !(v[4] <= Z3_KWIN_TOL * v[3])) atomicOr(d_err, BBX_DERR_PSF_WINDOW);
    }
}

// ---- PSF side ---------------------------------------------------------------------------------
// Row DFT of the S non-zero rows of every PSF stamp, summed directly: Q[which][sub][j][kx] = sum_x p[j][x] W^(kx (x - h)),
// kx < H (the stamp's columns sit at x = -h .. S-1-h mod L).  A thread takes one kx and ZQ_J stamp rows: the twiddle of a
// column is read once for all of them, the stamp values are wave-uniform.  (Round 5: this sum sat inside k_psf_cols, one
// L2 round trip per four terms in every workgroup: 2 x 11 000 of its 75 000 cycles.)
#define ZQ_J 7
template <class P>
__global__ __launch_bounds__(256) void k_psf_rowdft(const float* __restrict__ psf_n, const float* __restrict__ psf_r, int S, const float2* __restrict__ twg,
                                                    float2* __restrict__ Q, int nsub) {
    __shared__ float2 tw[P::L];
    for (int e = threadIdx.x; e < P::L; e += blockDim.x) tw[e] = twg[e];
    __syncthreads();
    const int kx = (int)(blockIdx.x * blockDim.x + threadIdx.x), j0 = (int)blockIdx.y * ZQ_J;
    const int which = (int)blockIdx.z / nsub, sub = (int)blockIdx.z - which * nsub;
    if (kx >= P::H) return;
    const float* st = (which ? psf_r : psf_n) + (size_t)sub * S * S;
    const int h = S / 2;
    float2 acc[ZQ_J];
#pragma unroll
    for (int jj = 0; jj < ZQ_J; jj++) acc[jj] = make_float2(0.f, 0.f);
    int idx = (kx * ((P::L - h % P::L) % P::L)) % P::L;             // kx (L - h) < H L: 32-bit
    for (int x = 0; x < S; x++) {
        const float2 w = tw[idx];
#pragma unroll
        for (int jj = 0; jj < ZQ_J; jj++) {
            if (j0 + jj < S) { const float pv = st[(j0 + jj) * S + x]; acc[jj].x += pv * w.x; acc[jj].y += pv * w.y; }
        }
        idx += kx; if (idx >= P::L) idx -= P::L;
    }
#pragma unroll
    for (int jj = 0; jj < ZQ_J; jj++)
        if (j0 + jj < S) Q[(((size_t)which * nsub + sub) * S + (j0 + jj)) * P::HP + kx] = acc[jj];
}

template <class P>
__global__ __launch_bounds__(P::THREADS, P::MINW_PSF) void k_psf_cols(const float2* __restrict__ Q, int S,
                                                         const zscal* __restrict__ sc, const float2* __restrict__ twg,
                                                         float4* __restrict__ cP,
                                                         float2* __restrict__ Ukn, float2* __restrict__ Ukr,
                                                         double* __restrict__ fs_partial, int nsub, int wh) {
    extern __shared__ float2 s[];
    __shared__ double red[5][P::THREADS / 64];
    WG_TASK(P::G, nsub, g, sub);
    ZSTAMP_HEAD(0);
    const aux_t aux = aux_setup<P>(s + P::NL * P::LS, twg);
    const float2* tw = aux.tw;
    const int h = S / 2;
    const size_t cbase = (size_t)(sub * P::G + g) * P::NL * P::L;
    constexpr int RT = P::THREADS, NE = (P::NL * P::L + RT - 1) / RT;
    float2 park[NE];                                               // Pn^, then kn^
    for (int pass = 0; pass < 2; pass++) {
        const float2* qs = Q + ((size_t)pass * nsub + sub) * S * P::HP;
        for (int e = threadIdx.x; e < P::NL * P::LS; e += blockDim.x) s[e] = make_float2(0.f, 0.f);
        __syncthreads();
        ZSTAMP(0, 1 + 3 * pass);
        // the stamp's S non-zero rows, row-transformed by k_psf_rowdft: line[y] = sum_x p[y][x] W^(kx x)
        for (int e = threadIdx.x; e < P::NL * S; e += blockDim.x) {
            const int ll = e % P::NL, j = e / P::NL;
            const int kk = g * P::NL + ll;
            if (kk >= P::H) continue;
            const int y = ((j - h) % P::L + P::L) % P::L;
            s[ll * P::LS + npos(y)] = qs[(size_t)j * P::HP + kk];
        }
        __syncthreads();
        ZSTAMP(0, 2 + 3 * pass);
        fft_fwd<P>(s, tw, S - h);                                      // the stamp's rows: [0, S - h) and [L - h, L)
        ZSTAMP(0, 3 + 3 * pass);
        if (pass == 0) {
            R_LOOP(k, e, l, p) park[k] = s[l * P::LS + npos(p)];      // Pn^ waits in registers while Pr^ is transformed
            __syncthreads();
        }
    }
    const zscal z = sc[sub];
    const float sn2 = z.sn * z.sn, sr2 = z.sr * z.sr, fn2 = z.fn * z.fn, fr2 = z.fr * z.fr;
    double fs = 0.0, sk2n = 0.0, sk2r = 0.0;                      // F_S and the Parseval sums of kn^2, kr^2
    R_LOOP(k, e, l, p) {
        const int kx = g * P::NL + l;
        float2 kn = make_float2(0.f, 0.f), kr = kn;
        float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kx < P::H) {
            const double wgt = (kx == 0 || (P::L % 2 == 0 && kx == P::L / 2)) ? 1.0 : 2.0;
            const float2 pn = park[k], pr = s[l * P::LS + npos(p)];
            const float pn2 = pn.x * pn.x + pn.y * pn.y, pr2 = pr.x * pr.x + pr.y * pr.y;
            const float den = (sn2 * fr2) * pr2 + (sr2 * fn2) * pn2;
            // one division per entry: 1 / den as the square of 1 / sqrt(den) (a float32 division is ~10 instructions,
            // four of them were a fifth of this kernel's arithmetic)
            const float sd = sqrtf(den), isd = 1.0f / sd, rden = isd * isd;
            cp = make_float4(pn.x, pn.y, pr.x, pr.y);
            kr = cscale(make_float2(pr.x, -pr.y), z.fr * fn2 * pn2 * rden);
            kn = cscale(make_float2(pn.x, -pn.y), z.fn * fr2 * pr2 * rden);
            fs += wgt * (double)(fn2 * pn2 * fr2 * pr2 * rden);
            sk2n += wgt * (double)(kn.x * kn.x + kn.y * kn.y);
            sk2r += wgt * (double)(kr.x * kr.x + kr.y * kr.y);
        }
        // k_img_cols gets the two spectra themselves (16 bytes per entry, one load) and forms 1 / sqrt(den) and the
        // coefficients of D^, S_n^, S_r^ from them again (rounds 3-4: A, B and sqrt(den), 20 bytes in three loads)
        st_nt(cP + cbase + e, cp);
        s[l * P::LS + npos(p)] = kr;
        park[k] = kn;
    }
    __syncthreads();
    ZSTAMP(0, 7);
    fft_inv<P>(s, tw);
    ZSTAMP(0, 8);
    const bool win = 2 * wh < P::L;
    double e_all = 0.0, e_out = 0.0;
    if (win) store_u_win<P>(s, Ukr, sub, g, wh, e_all, e_out); else store_u<P>(s, Ukr, sub, g);
    __syncthreads();
    R_LOOP(k, e, l, p) s[l * P::LS + npos(p)] = park[k];
    __syncthreads();
    ZSTAMP(0, 9);
    fft_inv<P>(s, tw);
    ZSTAMP(0, 10);
    if (win) store_u_win<P>(s, Ukn, sub, g, wh, e_all, e_out); else store_u<P>(s, Ukn, sub, g);
    fs = wave_sum_f64(fs); sk2n = wave_sum_f64(sk2n); sk2r = wave_sum_f64(sk2r);
    e_all = wave_sum_f64(e_all); e_out = wave_sum_f64(e_out);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = fs; red[1][threadIdx.x >> 6] = sk2n; red[2][threadIdx.x >> 6] = sk2r;
        red[3][threadIdx.x >> 6] = e_all; red[4][threadIdx.x >> 6] = e_out;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double tot = 0.0;
        for (int i = 0; i < (int)blockDim.x / 64; i++) tot += red[threadIdx.x][i];
        fs_partial[((size_t)threadIdx.x * nsub + sub) * P::G + g] = tot;      // [3][nsub][G]
    }
    if (win && (threadIdx.x == 3 || threadIdx.x == 4)) {
        // energy of k_n, k_r in this column group: all rows / the rows dropped by the window (summed over the groups
        // and checked by k_var_cols)
        double tot = 0.0;
        for (int i = 0; i < (int)blockDim.x / 64; i++) tot += red[threadIdx.x][i];
        fs_partial[((size_t)threadIdx.x * nsub + sub) * P::G + g] = tot;      // [5][nsub][G]
    }
    ZSTAMP(0, 11);
}

// inverse row pass of kr^, kn^ -> kr, kn -> squares -> forward row pass, T tiles
template <class P>
__global__ __launch_bounds__(P::LIGHT_THREADS, P::MINW_LIGHT) void k_psf_rows(const float2* __restrict__ Ukr, const float2* __restrict__ Ukn, float inv_n2,
                                                         const float2* __restrict__ twg, float2* __restrict__ Tkr2, float2* __restrict__ Tkn2,
                                                         int nsub, int nyb, int wb, const zscal* __restrict__ sc, const double* __restrict__ fs_partial,
                                                         sub_scal* __restrict__ sub_sc, int32_t* __restrict__ d_err) {
    extern __shared__ float2 s[];
    WG_TASK(nyb, nsub, ybw, sub);
    if (ybw == 0 && threadIdx.x < 64) sub_scalars<P>(fs_partial, nsub, sub, sc[sub], nyb != P::LB, sub_sc, d_err);
    // with a row window only the blocks that hold rows < wh or >= L - wh exist: nyb = 2 wb of them
    const int yb = (nyb == P::LB || ybw < wb) ? ybw : P::LB - 2 * wb + ybw;
    ZSTAMP_HEAD(1);
    const aux_t aux = aux_setup<P>(s + P::NL * P::LS, twg);
    const float2* tw = aux.tw;
    __syncthreads();                                                // the position table is used right away
    ZSTAMP(1, 1);
    load_u_pair<P>(Ukr, Ukn, sub, yb, s, aux.pp);
    __syncthreads();
    ZSTAMP(1, 2);
    fft_inv<P>(s, tw);
    ZSTAMP(1, 3);
    C_LOOP(e, l, p) {
        float2* q = s + l * P::LS + npos(p);
        const float a = q->x * inv_n2, b = q->y * inv_n2;
        *q = make_float2(a * a, b * b);
    }
    __syncthreads();
    ZSTAMP(1, 4);
    fft_fwd<P>(s, tw);
    ZSTAMP(1, 5);
    store_t_split<P>(s, aux.pp, Tkr2, Tkn2, sub, yb);
    ZSTAMP(1, 6);
}

// forward column pass of one T array -> C layout
template <class P>
__global__ __launch_bounds__(P::THREADS, P::MINW_LIGHT) void k_cols_fwd(const float2* __restrict__ T, const float2* __restrict__ twg, float2* __restrict__ Cout, int nsub) {
    extern __shared__ float2 s[];
    WG_TASK(P::G, nsub, g, sub);
    const aux_t aux = aux_setup<P>(s + P::NL * P::LS, twg);
    const float2* tw = aux.tw;
    load_t_lines<P>(T, sub, g, s);
    __syncthreads();
    fft_fwd<P>(s, tw);
    const size_t cbase = (size_t)(sub * P::G + g) * P::NL * P::L;
    C_LOOP(e, l, p) Cout[cbase + e] = s[l * P::LS + npos(p)];
}

// ---- image side -------------------------------------------------------------------------------
struct frame_args {
    const float* a; const float* b;          // the two frames of a pair (new, ref)
    const float* sa; const float* sb;        // sigma images (variance pair) or NULL
    int ny, nx, size, border, nsx, vec4;
};

// cut + forward row pass of a pair of real frames: (N, R) or, with sigma images, (Vn, Vr)
template <class P>
__global__ __launch_bounds__(P::LIGHT_THREADS, P::MINW_LIGHT) void k_img_rows(frame_args f, const float2* __restrict__ twg, float2* __restrict__ Ta, float2* __restrict__ Tb, int nsub) {
    extern __shared__ float2 s[];
    WG_TASK_ROWS(P::LB, nsub, yb, sub);
    const aux_t aux = aux_setup<P>(s + P::NL * P::LS, twg);
    const float2* tw = aux.tw;
    const int y0 = yb * P::NL;
    const int sy = sub / f.nsx, sx = sub - sy * f.nsx;
    const int Y0 = sy * f.size - f.border, X0 = sx * f.size - f.border;
    if (f.vec4) {
        // groups of four pixels never straddle the frame edge (size, border, nx multiples of 4)
        constexpr int NV = P::NL * P::L / 4;
        constexpr int IL = Z3_IMG_LOADS;                        // groups of four pixels per thread and round: all of a workgroup's loads in one round trip
        for (int e0 = threadIdx.x; e0 < NV; e0 += IL * (int)blockDim.x) {
            float4 va[IL], vb[IL];
            bool in[IL];
#pragma unroll
            for (int i = 0; i < IL; i++) {
                const int e = e0 + i * (int)blockDim.x;
                va[i] = vb[i] = make_float4(0.f, 0.f, 0.f, 0.f); in[i] = false;
                if (e < NV) {
                    const int q = 4 * e, ll = q / P::L, x = q - ll * P::L;
                    const int Y = Y0 + y0 + ll, X = X0 + x;
                    if (y0 + ll < P::L && Y >= 0 && Y < f.ny && X >= 0 && X < f.nx) {
                        const size_t o = (size_t)Y * f.nx + X;
                        in[i] = true;
                        va[i] = ld_nt(reinterpret_cast<const float4*>(f.a + o)); vb[i] = ld_nt(reinterpret_cast<const float4*>(f.b + o));
                        if (f.sa) {
                            const float4 p = *reinterpret_cast<const float4*>(f.sa + o), q4 = *reinterpret_cast<const float4*>(f.sb + o);
                            va[i] = make_float4(fmaxf(va[i].x, 0.f) + p.x * p.x, fmaxf(va[i].y, 0.f) + p.y * p.y, fmaxf(va[i].z, 0.f) + p.z * p.z,
                                                fmaxf(va[i].w, 0.f) + p.w * p.w);
                            vb[i] = make_float4(fmaxf(vb[i].x, 0.f) + q4.x * q4.x, fmaxf(vb[i].y, 0.f) + q4.y * q4.y,
                                                fmaxf(vb[i].z, 0.f) + q4.z * q4.z, fmaxf(vb[i].w, 0.f) + q4.w * q4.w);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < IL; i++) {
                const int e = e0 + i * (int)blockDim.x;
                if (e < NV) {
                    const int q = 4 * e, ll = q / P::L, x = q - ll * P::L;
                    float2* line = s + ll * P::LS;
                    line[npos(x)] = make_float2(va[i].x, vb[i].x); line[npos(x + 1)] = make_float2(va[i].y, vb[i].y);
                    line[npos(x + 2)] = make_float2(va[i].z, vb[i].z); line[npos(x + 3)] = make_float2(va[i].w, vb[i].w);
                }
            }
        }
    } else
    for (int e = threadIdx.x; e < P::NL * P::L; e += blockDim.x) {
        const int ll = e / P::L, x = e - ll * P::L;
        const int Y = Y0 + y0 + ll, X = X0 + x;
        float2 v = make_float2(0.f, 0.f);
        if (y0 + ll < P::L && Y >= 0 && Y < f.ny && X >= 0 && X < f.nx) {
            const size_t o = (size_t)Y * f.nx + X;
            v.x = f.a[o]; v.y = f.b[o];
            if (f.sa) { const float p = f.sa[o], q = f.sb[o]; v.x = fmaxf(v.x, 0.f) + p * p; v.y = fmaxf(v.y, 0.f) + q * q; }
        }
        s[ll * P::LS + npos(x)] = v;
    }
    __syncthreads();
    fft_fwd<P>(s, tw);
    store_t_split<P>(s, aux.pp, Ta, Tb, sub, yb);
}

// The same for both pairs in one launch (frames whose groups of four pixels are aligned): the four frames are read once;
// the variance pair (max(N, 0) + sigma_N^2, max(R, 0) + sigma_R^2) waits in registers behind the first transform.
// SPL (round 5, bbx_zogy_frame_mini): the sigma images do not exist as frames -- the kernel reads them off their mini images
// (bbx_spline.h): a table of the cubics of every frame row on every box interval (k_spl_polytable, 96 MB for both maps,
// mostly L2 hits here: the ~15 groups of four pixels of an interval read the same entry) instead of the two frames (2 x 4N
// written, 2 x 4N x 1.125 read); a pixel costs a share of a 16-byte load and three multiply-adds per map.
template <class P, bool SPL>
__global__ __launch_bounds__(P::LIGHT_THREADS, P::MINW_LIGHT) void k_img_rows_both(frame_args f, const float2* __restrict__ twg, float2* __restrict__ Ta,
                                                                                float2* __restrict__ Tb, float2* __restrict__ Tva, float2* __restrict__ Tvb, int nsub,
                                                                                bbx_spl spn, bbx_spl spr) {
    extern __shared__ float2 s[];
    WG_TASK_ROWS(P::LB, nsub, yb, sub);
    ZSTAMP_HEAD(2);
    const aux_t aux = aux_setup<P>(s + P::NL * P::LS, twg);
    const float2* tw = aux.tw;
    const int y0 = yb * P::NL;
    const int sy = sub / f.nsx, sx = sub - sy * f.nsx;
    const int Y0 = sy * f.size - f.border, X0 = sx * f.size - f.border;
    constexpr int NV = P::NL * P::L / 4, NP = (NV + P::LIGHT_THREADS - 1) / P::LIGHT_THREADS;
    float4 pa[NP], pb[NP];
    {
        float4 va[NP], vb[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int e = (int)threadIdx.x + i * P::LIGHT_THREADS;
            va[i] = vb[i] = pa[i] = pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < NV) {
                const int q = 4 * e, ll = q / P::L, x = q - ll * P::L;
                const int Y = Y0 + y0 + ll, X = X0 + x;
                if (y0 + ll < P::L && Y >= 0 && Y < f.ny && X >= 0 && X < f.nx) {
                    const size_t o = (size_t)Y * f.nx + X;
                    va[i] = ld_nt(reinterpret_cast<const float4*>(f.a + o)); vb[i] = ld_nt(reinterpret_cast<const float4*>(f.b + o));
                    if (!SPL) { pa[i] = ld_nt(reinterpret_cast<const float4*>(f.sa + o)); pb[i] = ld_nt(reinterpret_cast<const float4*>(f.sb + o)); }
                }
            }
        }
        if (SPL) {
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const int e = (int)threadIdx.x + i * P::LIGHT_THREADS;
                if (e < NV) {
                    const int q = 4 * e, ll = q / P::L, x = q - ll * P::L;
                    const int Y = Y0 + y0 + ll, X = X0 + x;
                    if (y0 + ll < P::L && Y >= 0 && Y < f.ny && X >= 0 && X < f.nx) {
                        // four pixels of one patch (patch widths, X0 multiples of 4): interval and remainder of the first, steps
                        // for the others; the group lies in one interval or in two neighbouring ones: both cubics are loaded
                        int cn, rn, cr, rr;
                        bbx_spl_axis(X, spn.pw, spn.rpw, spn.px, spn.npad, spn.nx1, spn.dx, spn.rdx, cn, rn);
                        bbx_spl_axis(X, spr.pw, spr.rpw, spr.px, spr.npad, spr.nx1, spr.dx, spr.rdx, cr, rr);
                        const float4* tn = spn.poly + (size_t)Y * spn.cnx + cn;
                        const float4* tr = spr.poly + (size_t)Y * spr.cnx + cr;
                        const int stepn = (rn + 3 * spn.nx1 >= spn.dx) ? 1 : 0, stepr = (rr + 3 * spr.nx1 >= spr.dx) ? 1 : 0;
                        const float4 pn0 = tn[0], pn1 = tn[stepn], pr0 = tr[0], pr1 = tr[stepr];
                        float sgn[4], sgr[4];
                        bool wn = false, wr = false;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            sgn[k] = bbx_spl_horner(wn ? pn1 : pn0, (float)rn * spn.rdx);
                            sgr[k] = bbx_spl_horner(wr ? pr1 : pr0, (float)rr * spr.rdx);
                            rn += spn.nx1; if (rn >= spn.dx) { rn -= spn.dx; wn = true; }
                            rr += spr.nx1; if (rr >= spr.dx) { rr -= spr.dx; wr = true; }
                        }
                        pa[i] = make_float4(sgn[0], sgn[1], sgn[2], sgn[3]); pb[i] = make_float4(sgr[0], sgr[1], sgr[2], sgr[3]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int e = (int)threadIdx.x + i * P::LIGHT_THREADS;
            if (e < NV) {
                const int q = 4 * e, ll = q / P::L, x = q - ll * P::L;
                float2* line = s + ll * P::LS;
                line[npos(x)] = make_float2(va[i].x, vb[i].x); line[npos(x + 1)] = make_float2(va[i].y, vb[i].y);
                line[npos(x + 2)] = make_float2(va[i].z, vb[i].z); line[npos(x + 3)] = make_float2(va[i].w, vb[i].w);
                const float4 p = pa[i], q4 = pb[i];
                pa[i] = make_float4(fmaxf(va[i].x, 0.f) + p.x * p.x, fmaxf(va[i].y, 0.f) + p.y * p.y, fmaxf(va[i].z, 0.f) + p.z * p.z,
                                    fmaxf(va[i].w, 0.f) + p.w * p.w);
                pb[i] = make_float4(fmaxf(vb[i].x, 0.f) + q4.x * q4.x, fmaxf(vb[i].y, 0.f) + q4.y * q4.y, fmaxf(vb[i].z, 0.f) + q4.z * q4.z,
                                    fmaxf(vb[i].w, 0.f) + q4.w * q4.w);
            }
        }
    }
    __syncthreads();
    ZSTAMP(2, 1);
    fft_fwd<P>(s, tw);
    ZSTAMP(2, 2);
    store_t_split<P>(s, aux.pp, Ta, Tb, sub, yb);
    __syncthreads();
    ZSTAMP(2, 3);
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int e = (int)threadIdx.x + i * P::LIGHT_THREADS;
        if (e < NV) {
            const int q = 4 * e, ll = q / P::L, x = q - ll * P::L;
            float2* line = s + ll * P::LS;
            line[npos(x)] = make_float2(pa[i].x, pb[i].x); line[npos(x + 1)] = make_float2(pa[i].y, pb[i].y);
            line[npos(x + 2)] = make_float2(pa[i].z, pb[i].z); line[npos(x + 3)] = make_float2(pa[i].w, pb[i].w);
        }
    }
    __syncthreads();
    ZSTAMP(2, 4);
    fft_fwd<P>(s, tw);
    ZSTAMP(2, 5);
    store_t_split<P>(s, aux.pp, Tva, Tvb, sub, yb);
    ZSTAMP(2, 6);
}

// column pass of the image pair: D^ = A N^ - B R^, Sn^ = kn^ N^, Sr^ = kr^ R^ and back (U tiles)
template <class P>
__global__ __launch_bounds__(P::THREADS, P::MINW) void k_img_cols(const float2* __restrict__ TN, const float2* __restrict__ TR, const float4* __restrict__ cP,
                                                         const zscal* __restrict__ sc, const float2* __restrict__ twg,
                                                         float2* __restrict__ UD, float2* __restrict__ USn, float2* __restrict__ USr,
                                                         float2* __restrict__ HSn, float2* __restrict__ HSr, int nsub, chunk_args ch) {
    extern __shared__ float2 s[];
    WG_TASK(P::G, nsub, g, sub);
    ZSTAMP_HEAD(3);
    const aux_t aux = aux_setup<P>(s + P::NL * P::LS, twg);
    const float2* tw = aux.tw;
    const size_t cbase = (size_t)(sub * P::G + g) * P::NL * P::L;
    load_t_lines<P>(TN, sub, g, s);
#ifndef Z3_NO_PREFETCH
    t_regs<P, P::THREADS> rr;                                      // T_R on its way while T_N is transformed
    fetch_t_lines<P, P::THREADS>(TR, sub, g, rr);
#endif
    __syncthreads();
    ZSTAMP(3, 1);
    fft_fwd<P>(s, tw);
    ZSTAMP(3, 2);
    constexpr int RT = P::THREADS, NE = (P::NL * P::L + RT - 1) / RT;
    // N^ waits in registers for R^; then one loop makes D^, S_n^ (both parked) and S_r^ (into the lines) from A, B, sqrt(den)
    float2 park[NE], parkd[NE];
    R_LOOP(k, e, l, p) park[k] = s[l * P::LS + npos(p)];
    __syncthreads();
#ifndef Z3_NO_PREFETCH
    pack_t_lines<P, P::THREADS>(rr, s);
#else
    load_t_lines<P>(TR, sub, g, s);
#endif
    __syncthreads();
    ZSTAMP(3, 3);
    // the PSF spectra of this column group: the first half of a thread's entries is on its way while R is transformed, the
    // second half goes out in one round behind it (a load inside the guarded loop below waits for its own round trip in
    // every iteration: 8 x 2 700 cycles of this kernel's 95 000 in round 4)
    constexpr int NC0 = NE / 2;
    float4 c0[NC0], c1[NE - NC0];
    {
        const int t = opaque_tid();
#pragma unroll
        for (int k = 0; k < NC0; k++) c0[k] = ld_nt(cP + cbase + min(t + k * RT, P::NL * P::L - 1));
    }
    fft_fwd<P>(s, tw);
    ZSTAMP(3, 4);
    {
        const int t = opaque_tid();
#pragma unroll
        for (int k = NC0; k < NE; k++) c1[k - NC0] = ld_nt(cP + cbase + min(t + k * RT, P::NL * P::L - 1));
    }
    const zscal z = sc[sub];
    const float cn = (z.sr * z.sr) * (z.fn * z.fn), cr = (z.sn * z.sn) * (z.fr * z.fr);      // den = cr |Pr^|^2 + cn |Pn^|^2
    const float kfn = z.fn * (z.fr * z.fr), kfr = z.fr * (z.fn * z.fn);
    R_LOOP(k, e, l, p) {
        float2* q = s + l * P::LS + npos(p);
        const float2 r = *q, n = park[k];
        const float4 c = k < NC0 ? c0[k < NC0 ? k : 0] : c1[k >= NC0 ? k - NC0 : 0];      // (Pn^, Pr^); zero in the padding columns kx >= H
        const float2 pn = make_float2(c.x, c.y), pr = make_float2(c.z, c.w);
        const float pn2 = pn.x * pn.x + pn.y * pn.y, pr2 = pr.x * pr.x + pr.y * pr.y;
        const float den = cr * pr2 + cn * pn2;
        const float isd = (g * P::NL + l < P::H) ? __builtin_amdgcn_rsqf(den) : 0.f, rden = isd * isd;
        // D^ = (f_r Pr^ N^ - f_n Pn^ R^) / sqrt(den), S_n^ = k_n^ N^, S_r^ = k_r^ R^ with k_n^ = f_n f_r^2 conj(Pn^) |Pr^|^2 / den
        const float2 an = cmul(pr, n), br = cmul(pn, r);
        const float da = z.fr * isd, db = z.fn * isd;
        parkd[k] = make_float2(an.x * da - br.x * db, an.y * da - br.y * db);
        park[k] = cscale(cmulc(n, pn), kfn * pr2 * rden);           // conj(Pn^) n
        *q = cscale(cmulc(r, pr), kfr * pn2 * rden);
    }
    __syncthreads();
    ZSTAMP(3, 5);
    fft_inv<P>(s, tw);
    ZSTAMP(3, 6);
    store_u<P>(s, USr, sub, g, HSr, ch);
    __syncthreads();
    R_LOOP(k, e, l, p) s[l * P::LS + npos(p)] = park[k];
    __syncthreads();
    ZSTAMP(3, 7);
    fft_inv<P>(s, tw);
    ZSTAMP(3, 8);
    store_u<P>(s, USn, sub, g, HSn, ch);
    __syncthreads();
    R_LOOP(k, e, l, p) s[l * P::LS + npos(p)] = parkd[k];
    __syncthreads();
    ZSTAMP(3, 9);
    fft_inv<P>(s, tw);
    ZSTAMP(3, 10);
    store_u<P>(s, UD, sub, g);
    ZSTAMP(3, 11);
}

// column pass of the variance pair: V(S)^ = Vn^ (kn^2)^ + Vr^ (kr^2)^ and back (U tiles).  The spectra (kn^2)^, (kr^2)^
// get their column pass here as well (their row pass is k_psf_rows): a workgroup needs exactly its own column group
// of them, so they go from the T tiles through the transform into registers and never to HBM as coefficient arrays.
template <class P>
__global__ __launch_bounds__(P::VAR_THREADS, P::VAR_MINW) void k_var_cols(const float2* __restrict__ TVn, const float2* __restrict__ TVr, const float2* __restrict__ Tk2n,
                                                         const float2* __restrict__ Tk2r, const float2* __restrict__ twg,
                                                         float2* __restrict__ UVS, const sub_scal* __restrict__ sub_sc, int nsub, int wh,
                                                         int32_t* __restrict__ ticket) {
    extern __shared__ float2 s[];
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0;          // the final kernel's chunk counter (next launch on this stream)
    WG_TASK(P::G, nsub, g, sub);
    ZSTAMP_HEAD(4);
    const aux_t aux = aux_setup<P>(s + P::NL * P::LS, twg);
    const float2* tw = aux.tw;
    constexpr int RT = P::VAR_THREADS, NE = (P::NL * P::L + RT - 1) / RT;        // two parked arrays: 512 threads, <= 128 VGPRs, no spills
    float2 park[NE], coef[NE];
    const bool win = 2 * wh < P::L;
    if (win) load_t_lines_win<P>(Tk2n, sub, g, s, wh); else load_t_lines<P>(Tk2n, sub, g, s);
    t_regs<P, P::VAR_THREADS> rr;                                  // Vn^ (then Vr^) on its way while the k^2 spectrum is transformed
    fetch_t_lines<P, P::VAR_THREADS>(TVn, sub, g, rr);
    __syncthreads();
    ZSTAMP(4, 1);
    fft_fwd<P>(s, tw, win ? wh : 0);
    ZSTAMP(4, 2);
#if defined(Z3_STAMPS) && defined(Z3_FFT2X)          // diagnostic: the same transform once more (results wrong, timing only)
    fft_fwd<P>(s, tw, win ? wh : 0);
    ZSTAMP(4, 12);
    fft_fwd<P>(s, tw, win ? wh : 0);
    ZSTAMP(4, 13);
#endif
    R_LOOP(k, e, l, p) coef[k] = s[l * P::LS + npos(p)];                       // (kn^2)^
    __syncthreads();
    pack_t_lines<P, P::VAR_THREADS>(rr, s);
    __syncthreads();
    ZSTAMP(4, 3);
    fft_fwd<P>(s, tw);
    ZSTAMP(4, 4);
    R_LOOP(k, e, l, p) park[k] = cmul(coef[k], s[l * P::LS + npos(p)]);
    __syncthreads();
    if (win) load_t_lines_win<P>(Tk2r, sub, g, s, wh); else load_t_lines<P>(Tk2r, sub, g, s);
    constexpr int NH = t_regs<P, P::VAR_THREADS>::N / 4;          // (more of it beside the parked product: spills)
    fetch_t_lines<P, P::VAR_THREADS, 0, NH>(TVr, sub, g, rr);
    __syncthreads();
    ZSTAMP(4, 5);
    fft_fwd<P>(s, tw, win ? wh : 0);
    ZSTAMP(4, 6);
    R_LOOP(k, e, l, p) coef[k] = s[l * P::LS + npos(p)];                       // (kr^2)^
    __syncthreads();
    fetch_t_lines<P, P::VAR_THREADS, NH>(TVr, sub, g, rr);
    pack_t_lines<P, P::VAR_THREADS>(rr, s);
    __syncthreads();
    ZSTAMP(4, 7);
    fft_fwd<P>(s, tw);
    ZSTAMP(4, 8);
    const float beta = sub_sc[sub].beta;
    R_LOOP(k, e, l, p) {
        float2* q = s + l * P::LS + npos(p);
        const float2 v = cmul(coef[k], *q);
        *q = make_float2((park[k].x + v.x) * beta, (park[k].y + v.y) * beta);
    }
    __syncthreads();
    ZSTAMP(4, 9);
    fft_inv<P>(s, tw);
    ZSTAMP(4, 10);
    store_u<P>(s, UVS, sub, g);
    ZSTAMP(4, 11);
}

struct out_args {
    float* D; float* S; float* Scorr; float* Fpsf; float* Fpsferr;      // full frames [ny][nx]; S may be NULL
    int ny, nx, size, border, nsx, vec4;
};

// inverse row pass of (D, V_S) and (Sn, Sr) + the final algebra, written into the full frames.  A workgroup takes a chunk of
// consecutive blocks of NL rows of one sub-image (round 5; one block per workgroup before): the tables are set up once, the
// tiles of the next block are on their way while this one is transformed (registers), and the row above a block -- needed
// for the finite differences of V_ast -- is the last row of the block before it, kept in LDS: only a chunk's first block
// transforms a fifth line, taken from the halo arrays.
template <class P>
__global__ __launch_bounds__(P::FIN_THREADS, P::FIN_MINW) void k_final_rows(const float2* __restrict__ UD, const float2* __restrict__ UVS,
                                                           const float2* __restrict__ USn, const float2* __restrict__ USr,
                                                           const float2* __restrict__ HSn, const float2* __restrict__ HSr,
                                                           const zscal* __restrict__ sc, const sub_scal* __restrict__ sub_sc,
                                                           float inv_n2, const float2* __restrict__ twg, out_args o, chunk_args ch,
                                                           const int2* __restrict__ tasks, int ntasks, int32_t* __restrict__ ticket,
                                                           int nsub, float cand_thr, uint32_t* __restrict__ cand_list,
                                                           int32_t* __restrict__ cand_cnt, uint32_t cand_cap, int32_t* __restrict__ d_err) {
    extern __shared__ float2 s[];
    float2* hline = s + P::NL * P::LS;                              // the row above the block
    __shared__ int s_task;
    ZSTAMP_HEAD(5);
    const aux_t aux = aux_setup<P>(s + (P::NL + 1) * P::LS, twg);
    const float2* tw = aux.tw;
    // chunks are handed out by a ticket counter (zeroed by k_var_cols), longest first: the workgroups of a CU do not run at
    // the same speed (the older one wins the issue slots), a fixed share per workgroup left the slow ones 15 % behind
    for (;;) {
    __syncthreads();                                                // the chunk before is done with the lines and s_task
    if (threadIdx.x == 0) s_task = atomicAdd(ticket, 1);
    __syncthreads();
    const int task_ = __builtin_amdgcn_readfirstlane(s_task);      // wave-uniform: what follows from it lives in scalar registers
    if (task_ >= ntasks) break;
    const int sub = tasks[task_].x, cc = tasks[task_].y;
    const int ybA = chunk_start(ch, cc), ybB = chunk_start(ch, cc + 1);
    const zscal z = sc[sub];
    const sub_scal ss = sub_sc[sub];
    const float sn2 = z.sn * z.sn, sr2 = z.sr * z.sr, fn2 = z.fn * z.fn, fr2 = z.fr * z.fr;
    const float fD = z.fr * z.fn / sqrtf(sn2 * fr2 + sr2 * fn2);
    const float dx2 = z.dx * z.dx, dy2 = z.dy * z.dy;
    const int sy = sub / o.nsx, sx = sub - sy * o.nsx;
    constexpr int KX = (P::L + P::FIN_THREADS - 1) / P::FIN_THREADS;   // pixels of a row per thread: x = border + t + k T
    // the chunk's first block: both pairs and the row above it go out now
    u_regs<P, P::FIN_THREADS> ra, rb;
    fetch_u_pair<P, P::FIN_THREADS>(UD, UVS, sub, ybA, ra);
    fetch_u_pair<P, P::FIN_THREADS>(USn, USr, sub, ybA, rb);
    static_assert(P::H <= 2 * P::FIN_THREADS, "halo line: two entries per thread");
    float2 ha[2], hb[2];
    {
        const size_t hbase = ((size_t)sub * ch.nch + cc) * P::HP;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int kx = (int)threadIdx.x + i * P::FIN_THREADS;
            if (kx < P::H) { ha[i] = HSn[hbase + kx]; hb[i] = HSr[hbase + kx]; }
        }
    }
    __syncthreads();                                                // tables and scalars
    ZSTAMP(5, 1);
    // the row above the chunk (spectrum order) waits in the fifth line: the first (D, V_S) transform leaves it alone
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int kx = (int)threadIdx.x + i * P::FIN_THREADS;
        if (kx < P::H) pack_store<P>(hline, aux.pp, kx, ha[i], hb[i]);
    }
    const float ifs = 1.0f / ss.fs, ifD = inv_n2 / fD, vscale = inv_n2 * ss.ibeta;
    for (int yb = ybA; yb < ybB; yb++) {
        const int y0 = yb * P::NL;
        const bool first = yb == ybA, more = yb + 1 < ybB;
        // (D, V_S): D goes out at once, V_S waits in registers (the same thread takes the same pixels below)
        float vsr[P::NL][KX];
        // (the thread index goes through an empty asm in every phase: the offsets of each phase are worked out again instead of
        // living in registers across the whole loop -- with them the kernel spills)
        pack_u_pair<P, P::FIN_THREADS>(ra, yb, s, aux.pp, opaque_tid());
        __syncthreads();
        fft_inv<P>(s, tw);
        if (more) fetch_u_pair<P, P::FIN_THREADS>(UD, UVS, sub, yb + 1, ra, opaque_tid());         // on its way behind the rest of this block
        const int td = opaque_tid();
#pragma unroll
        for (int l = 0; l < P::NL; l++) {
            const int y = y0 + l, Y = sy * o.size + (y - o.border);
            const bool rowok = y >= o.border && y < o.border + o.size && Y < o.ny;
#pragma unroll
            for (int k = 0; k < KX; k++) {
                const int xi = td + k * P::FIN_THREADS, Xf = sx * o.size + xi;
                vsr[l][k] = 0.f;
                if (xi < o.size) {
                    const float2 v = s[l * P::LS + npos(o.border + xi)];
                    vsr[l][k] = v.y * vscale;
                    if (rowok && Xf < o.nx) st_nt(o.D + (size_t)Y * o.nx + Xf, v.x * ifD);
                }
            }
        }
        __syncthreads();
        // (Sn, Sr): the block's rows; the row above them is in hline already (real space) unless this is the chunk's first block
        pack_u_pair<P, P::FIN_THREADS>(rb, yb, s, aux.pp, opaque_tid());
        __syncthreads();
        if (first) fft_inv<P, P::NL + 1>(s, tw); else fft_inv<P>(s, tw);
        if (more) fetch_u_pair<P, P::FIN_THREADS>(USn, USr, sub, yb + 1, rb, opaque_tid());
        const int tf = opaque_tid();
#pragma unroll
        for (int l = 0; l < P::NL; l++) {
            const int y = y0 + l, Y = sy * o.size + (y - o.border);
            if (!(y >= o.border && y < o.border + o.size && Y < o.ny)) continue;
            const float2* line = s + l * P::LS;
            const float2* upline = l ? line - P::LS : hline;
#pragma unroll
            for (int k = 0; k < KX; k++) {
                const int xi = tf + k * P::FIN_THREADS, Xf = sx * o.size + xi, xx = o.border + xi;
                if (xi >= o.size || Xf >= o.nx) continue;
                const float2 c = cscale(line[npos(xx)], inv_n2);                              // (Sn, Sr) here
                const float2 up = cscale(upline[npos(xx)], inv_n2), lf = cscale(line[npos(xx == 0 ? P::L - 1 : xx - 1)], inv_n2);
                const float sval = c.x - c.y;                                                  // S = Sn - Sr
                const float dSndy = c.x - up.x, dSndx = c.x - lf.x, dSrdy = c.y - up.y, dSrdx = c.y - lf.y;
                const float vast = dx2 * (dSndx * dSndx + dSrdx * dSrdx) + dy2 * (dSndy * dSndy + dSrdy * dSrdy);
                const float vs = vsr[l][k];
                const size_t q = (size_t)Y * o.nx + Xf;
                if (o.S) st_nt(o.S + q, sval);
#ifndef Z3_EXACT_SQRT
                // v_rsq_f32 / v_sqrt_f32 (1 ulp) instead of the correctly rounded division and square roots (~30 instructions per
                // pixel: 8 % of this kernel); the transforms in front are good to ~1e-6 of the image scale
                const float scv = sval * __builtin_amdgcn_rsqf(vs + vast);
                st_nt(o.Scorr + q, scv);
                st_nt(o.Fpsf + q, sval * ifs);
                st_nt(o.Fpsferr + q, __builtin_amdgcn_sqrtf(fmaxf(vs, 0.f)) * ifs);
#else
                const float scv = sval / sqrtf(vs + vast);
                o.Scorr[q] = scv;
                o.Fpsf[q] = sval * ifs;
                o.Fpsferr[q] = sqrtf(fmaxf(vs, 0.f)) * ifs;
#endif
                // transient candidates (bbx_zogy_candidates): the pixels bbx_find_peaks would collect in a pass of its own over
                // the Scorr frame.  They are rare (a few thousand per frame): one reservation per wave that holds any.
                if (cand_thr > 0.f) {
                    const bool hit = fabsf(scv) >= cand_thr;             // NaN compares false
                    const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
                    if (hm) {
                        const int leader = (int)__builtin_ctzll(hm);
                        unsigned base = 0;
                        if ((tf & 63) == leader) base = atomicAdd((unsigned*)cand_cnt, (unsigned)__popcll(hm));
                        base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
                        if (hit) {
                            const unsigned k2 = base + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
                            if (k2 < cand_cap) cand_list[k2] = (uint32_t)q; else atomicOr(d_err, BBX_DERR_LIST_OVERFLOW);
                        }
                    }
                }
            }
        }
        if (more) {
            // the block's last row is the row above the next block
            __syncthreads();
            const float2* last = s + (P::NL - 1) * P::LS;
            for (int e = opaque_tid(); e < P::LP; e += P::FIN_THREADS) hline[e] = last[e];
            __syncthreads();
        }
    }
    }
    ZSTAMP(5, 8);
}

// ---- host side --------------------------------------------------------------------------------
}  // namespace z3
// Chunks of the final kernel: every sub-image's nyb row blocks are cut the same way into runs of decreasing length (guided
// schedule: a round of nslots / nsub chunks per sub-image takes two thirds of what is left, down to 2 blocks), and the tasks
// (sub-image, chunk) are listed longest first: the first round fills every workgroup slot, the short ones at the end even
// out the workgroups' different speeds.  Built once per geometry, kept with the context's twiddle table.
struct zogy_chunk_plan { int key[4]; int nch, ntasks; int* d_start; int2* d_tasks; };
static int chunk_plan(bbx_ctx* ctx, zogy_chunk_plan* pl, int yb0, int nyb, int nsub, int nslots) {
    if (pl->d_start && pl->key[0] == yb0 && pl->key[1] == nyb && pl->key[2] == nsub && pl->key[3] == nslots) return BBX_OK;
    int per = nslots / nsub; if (per < 1) per = 1;
    int* len = (int*)malloc((size_t)(nyb + 1) * sizeof(int));
    if (!len) return BBX_ERR_NOMEM;
    int nch = 0, rem = nyb;
    while (rem > 0) {
        int sz = (2 * rem) / (3 * per); if (sz < 2) sz = 2;
        for (int i = 0; i < per && rem > 0; i++) { const int l = sz < rem ? sz : rem; len[nch++] = l; rem -= l; }
    }
    int* start = (int*)malloc((size_t)(nch + 1) * sizeof(int));
    int2* tasks = (int2*)malloc((size_t)nch * nsub * sizeof(int2));
    if (!start || !tasks) { free(len); free(start); free(tasks); return BBX_ERR_NOMEM; }
    start[0] = yb0;
    for (int c = 0; c < nch; c++) start[c + 1] = start[c] + len[c];
    int nt = 0;
    for (int c = 0; c < nch; c++)                                   // len[] is non-increasing: chunk-major = longest first
        for (int sub = 0; sub < nsub; sub++) tasks[nt++] = make_int2(sub, c);
    if (pl->d_start) { (void)hipDeviceSynchronize(); (void)hipFree(pl->d_start); (void)hipFree(pl->d_tasks); pl->d_start = nullptr; pl->d_tasks = nullptr; }
    hipError_t e = hipMalloc((void**)&pl->d_start, (size_t)(nch + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&pl->d_tasks, (size_t)nt * sizeof(int2));
    if (e == hipSuccess) e = hipMemcpy(pl->d_start, start, (size_t)(nch + 1) * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d_tasks, tasks, (size_t)nt * sizeof(int2), hipMemcpyHostToDevice);
    free(len); free(start); free(tasks);
    if (e != hipSuccess) return bbx_hip_fail(ctx, e, "chunk plan", __LINE__);
    pl->key[0] = yb0; pl->key[1] = nyb; pl->key[2] = nsub; pl->key[3] = nslots; pl->nch = nch; pl->ntasks = nt;
    return BBX_OK;
}
namespace z3 {
template <class P>
static int run(bbx_ctx* ctx, const float2* d_tw, zogy_chunk_plan* plan, int ny, int nx, int size, int border, const float* d_new, const float* d_ref,
               const float* d_sig_new, const float* d_sig_ref, const bbx_spl* spn, const bbx_spl* spr, const float* d_psf_n, const float* d_psf_r, int S, const float* h_scal,
               float* d_D, float* d_S, float* d_Scorr, float* d_Fpsf, float* d_Fpsferr, hipStream_t s) {
    const int nsy = ny / size, nsx = nx / size, nsub = nsy * nsx;
    int rc;
    // chunks of the final kernel (chunk_plan; cached with the twiddle table)
    const int yb0 = border / P::NL, yb1 = (border + size - 1) / P::NL, nyb = yb1 - yb0 + 1;
    const int nslots = 2 * (ctx->num_cus > 0 ? ctx->num_cus : 256);
    rc = chunk_plan(ctx, plan, yb0, nyb, nsub, nslots); if (rc) return rc;
    const int nch = plan->nch;
    const z3::chunk_args ch{plan->d_start, nch};
    const size_t unit = (size_t)nsub * P::UNIT, hunit = (size_t)nsub * nch * P::HP;
    // 4 T + 4 U arrays + the PSF spectra (Pn^, Pr^) as float4 (two arrays) + 2 row-transformed k^2 arrays + 2 halo arrays +
    // the row DFTs of the stamps + scalars + partial sums
    constexpr int NARR = 12;
    const size_t qunit = 2 * (size_t)nsub * S * P::HP;
    const size_t bytes = (NARR * unit + 2 * hunit + qunit) * sizeof(float2) + (size_t)nsub * sizeof(zscal) + 5 * (size_t)nsub * P::G * sizeof(double) + (size_t)nsub * sizeof(z3::sub_scal) + 4096;
    char* ws = (char*)bbx_ws(ctx, WS_CAND, bytes, &rc); if (rc) return rc;
    float2* arr[NARR]; for (int i = 0; i < NARR; i++) arr[i] = (float2*)ws + (size_t)i * unit;
    float2 *HSn = (float2*)ws + NARR * unit, *HSr = HSn + hunit, *Qdft = HSr + hunit;
    char* p = ws + (NARR * unit + 2 * hunit + qunit) * sizeof(float2);
    zscal* d_sc = (zscal*)p; p += (size_t)nsub * sizeof(zscal);
    p = (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    double* fs_partial = (double*)p; p += 5 * (size_t)nsub * P::G * sizeof(double);
    sub_scal* sub_sc = (sub_scal*)p;
    float2 *T0 = arr[0], *T1 = arr[1], *T2 = arr[2], *T3 = arr[3], *U0 = arr[4], *U1 = arr[5], *U2 = arr[6], *U3 = arr[7];
    float4* cP = (float4*)arr[8];                            // arr[8], arr[9]
    float2 *cK2n = arr[10], *cK2r = arr[11];
    BBX_HIP(hipMemcpyAsync(d_sc, h_scal, (size_t)nsub * sizeof(zscal), hipMemcpyHostToDevice, s));      // pageable source: staged before the call returns
#ifndef Z3_LDS_PAD
#define Z3_LDS_PAD 0           // experiments: extra dynamic LDS per workgroup (forces one workgroup per CU)
#endif
    const size_t lds = (size_t)P::NL * P::LS * sizeof(float2) + aux_bytes<P>() + Z3_LDS_PAD,
                 lds_fin = (size_t)(P::NL + 1) * P::LS * sizeof(float2) + aux_bytes<P>() + Z3_LDS_PAD;
    if (ctx->zogy3_attr_L != P::L) {                       // per context (= per device and issuing thread)
        BBX_HIP(hipFuncSetAttribute((const void*)k_psf_cols<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_psf_rows<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_cols_fwd<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_img_rows<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_img_rows_both<P, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_img_rows_both<P, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_img_cols<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_var_cols<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        BBX_HIP(hipFuncSetAttribute((const void*)k_final_rows<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fin));
        ctx->zogy3_attr_L = P::L;
    }
    const float inv_n2 = 1.0f / ((float)P::L * (float)P::L);
    const dim3 gcol = grid8(P::G, nsub), grow = grid8(P::LB, nsub), blk(P::THREADS);
    const float2* tw = d_tw;
    // Row window of the matched-filter kernels k_n, k_r (real space): they are as compact as the PSFs they are made
    // of, so only 2 wh of their L rows go through the inverse row pass, the squares and the forward row pass; the rest
    // is below float32 rounding (checked on the device, Z3_KWIN_TOL).  wh: a multiple of NL; W = 2 wh >= 4 S + 32.
    int wh = P::L;                                          // 2 wh >= L: no window
    if (!ctx->zogy_kwin_off && P::L % P::NL == 0) {
        int w = ((4 * S + 32) / 2 + P::NL - 1) / P::NL * P::NL;
        if (w < 32) w = (32 + P::NL - 1) / P::NL * P::NL;
        if (2 * w < P::L) wh = w;
    }
    const bool win = 2 * wh < P::L;
    const int wb = win ? wh / P::NL : 0, nyb_psf = win ? 2 * wb : P::LB;
    BBX_LAUNCH_TIMED(ctx, BBX_PROF_Z_PSF_DFT, k_psf_rowdft<P>, dim3((P::H + 255) / 256, (S + ZQ_J - 1) / ZQ_J, 2 * nsub), dim3(256), 0, s, d_psf_n, d_psf_r, S, tw, Qdft, nsub);
    BBX_LAUNCH_TIMED(ctx, BBX_PROF_Z_PSF_COLS, k_psf_cols<P>, gcol, blk, lds, s, Qdft, S, d_sc, tw, cP, U0, U1, fs_partial, nsub, wh);
    float2 *TK2r = cK2r, *TK2n = cK2n;                      // row-transformed (kr^2)^, (kn^2)^: T layout, column pass inside k_var_cols
    BBX_LAUNCH_TIMED(ctx, BBX_PROF_Z_PSF_ROWS, k_psf_rows<P>, grid8(nyb_psf, nsub), dim3(P::LIGHT_THREADS), lds, s, U1, U0, inv_n2, tw, TK2r, TK2n, nsub,
                     nyb_psf, wb, d_sc, fs_partial, sub_sc, ctx->d_err);
    frame_args fa; fa.a = d_new; fa.b = d_ref; fa.sa = nullptr; fa.sb = nullptr; fa.ny = ny; fa.nx = nx; fa.size = size; fa.border = border; fa.nsx = nsx;
    fa.vec4 = (size % 4 == 0 && border % 4 == 0 && nx % 4 == 0 && P::L % 4 == 0 && ((uintptr_t)d_new | (uintptr_t)d_ref | (uintptr_t)d_sig_new | (uintptr_t)d_sig_ref) % 16 == 0) ? 1 : 0;
    if (spn) {
        // sigma maps read off their mini images: aligned groups of four pixels inside one patch, at most one interval step
        // inside a group (boxes wider than 3 pixels)
        if (!fa.vec4 || spn->pw % 4 || spr->pw % 4 || 3 * spn->nx1 >= spn->dx || 3 * spr->nx1 >= spr->dx) return BBX_ERR_ARG;
        bbx_spl tn = *spn, tr = *spr;
        const size_t nbn = (size_t)ny * tn.cnx * sizeof(float4), nbr = (size_t)ny * tr.cnx * sizeof(float4);
        char* tab = (char*)bbx_ws(ctx, WS_ZSPL, nbn + nbr + 256, &rc); if (rc) return rc;
        tn.poly = (const float4*)tab; tr.poly = (const float4*)(tab + ((nbn + 255) & ~(size_t)255));
        hipLaunchKernelGGL(k_spl_polytable, dim3((tn.cnx + 255) / 256, ny), dim3(256), 0, s, tn, ny, (float4*)tn.poly);
        hipLaunchKernelGGL(k_spl_polytable, dim3((tr.cnx + 255) / 256, ny), dim3(256), 0, s, tr, ny, (float4*)tr.poly);
        BBX_LAUNCH_TIMED(ctx, BBX_PROF_Z_IMG_ROWS, (k_img_rows_both<P, true>), grow, dim3(P::LIGHT_THREADS), lds, s, fa, tw, T0, T1, T2, T3, nsub, tn, tr);
    } else
#ifndef Z3_ROWS_SPLIT
    if (fa.vec4) {
        fa.sa = d_sig_new; fa.sb = d_sig_ref;
        BBX_LAUNCH_TIMED(ctx, BBX_PROF_Z_IMG_ROWS, (k_img_rows_both<P, false>), grow, dim3(P::LIGHT_THREADS), lds, s, fa, tw, T0, T1, T2, T3, nsub, bbx_spl{}, bbx_spl{});
    } else
#endif
    {
        BBX_LAUNCH_TIMED(ctx, BBX_PROF_Z_IMG_ROWS, k_img_rows<P>, grow, dim3(P::LIGHT_THREADS), lds, s, fa, tw, T0, T1, nsub);
        fa.sa = d_sig_new; fa.sb = d_sig_ref;
        BBX_LAUNCH_TIMED(ctx, BBX_PROF_Z_IMG_ROWS, k_img_rows<P>, grow, dim3(P::LIGHT_THREADS), lds, s, fa, tw, T2, T3, nsub);
    }
    BBX_LAUNCH_TIMED(ctx, BBX_PROF_Z_IMG_COLS, k_img_cols<P>, gcol, blk, lds, s, T0, T1, cP, d_sc, tw, U0, U1, U2, HSn, HSr, nsub, ch);      // D, Sn, Sr
    BBX_LAUNCH_TIMED(ctx, BBX_PROF_Z_VAR_COLS, k_var_cols<P>, gcol, dim3(P::VAR_THREADS), lds, s, T2, T3, TK2n, TK2r, tw, U3, sub_sc, nsub, wh, &ctx->d_counters[CNT_TICKET]);            // V_S
    out_args oa; oa.D = d_D; oa.S = d_S; oa.Scorr = d_Scorr; oa.Fpsf = d_Fpsf; oa.Fpsferr = d_Fpsferr;
    oa.ny = ny; oa.nx = nx; oa.size = size; oa.border = border; oa.nsx = nsx; oa.vec4 = 0;
    const dim3 gfin = grid8(plan->ntasks < nslots ? plan->ntasks : nslots, 1);
    // transient candidates on request (bbx_zogy_candidates): listed by the kernel that writes Scorr
    float cand_thr = 0.f; uint32_t* cand_list = nullptr; uint32_t cand_cap = 0;
    int32_t* cand_cnt = &ctx->d_counters[CNT_ZCAND];
    ctx->zcand_img = nullptr;
    if (ctx->zcand_thr > 0.f && (size_t)ny * nx < 0xffffffffull) {
        cand_cap = (uint32_t)((size_t)ny * nx / 16 + 1024);
        cand_list = (uint32_t*)bbx_ws(ctx, WS_ZCAND, (size_t)cand_cap * sizeof(uint32_t), &rc); if (rc) return rc;
        BBX_HIP(hipMemsetAsync(cand_cnt, 0, sizeof(int32_t), s));
        cand_thr = ctx->zcand_thr;
    }
    BBX_LAUNCH_TIMED(ctx, BBX_PROF_ZOGY_FINAL, k_final_rows<P>, gfin, dim3(P::FIN_THREADS), lds_fin, s, U0, U3, U1, U2, HSn, HSr, d_sc, sub_sc,
                     inv_n2, tw, oa, ch, plan->d_tasks, plan->ntasks, &ctx->d_counters[CNT_TICKET], nsub, cand_thr, cand_list, cand_cnt, cand_cap,
                     ctx->d_err);
    if (cand_thr > 0.f) { ctx->zcand_img = d_Scorr; ctx->zcand_thr_used = cand_thr; ctx->zcand_npix = (size_t)ny * nx; }
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

}  // namespace z3

int bbx_zogy3_supported(int L) { return (L == 1400 || L == 140 || L == 128 || L == 100 || L == 64) ? 1 : 0; }

static int bbx_zogy3_run(bbx_ctx* ctx, const float2* d_tw, zogy_chunk_plan* plan, int L, int ny, int nx, int size, int border, const float* d_new, const float* d_ref,
                  const float* d_sig_new, const float* d_sig_ref, const bbx_spl* spn, const bbx_spl* spr, const float* d_psf_n, const float* d_psf_r, int S, const float* h_scal,
                  float* d_D, float* d_S, float* d_Scorr, float* d_Fpsf, float* d_Fpsferr, hipStream_t s) {
#define Z3_RUN(...) return z3::run<z3::Plan<__VA_ARGS__>>(ctx, d_tw, plan, ny, nx, size, border, d_new, d_ref, d_sig_new, d_sig_ref, spn, spr, d_psf_n, d_psf_r, S, \
                                                          h_scal, d_D, d_S, d_Scorr, d_Fpsf, d_Fpsferr, s)
    switch (L) {
        case 1400: Z3_RUN(Z3_PLAN1400);
        case 140: Z3_RUN(5, 7, 4);
        case 128: Z3_RUN(8, 16);
        case 100: Z3_RUN(5, 5, 4);
        case 64: Z3_RUN(8, 8);
    }
    return BBX_ERR_ARG;
}

// ---- entry points (include/bbx.h) ------------------------------------------------------------------------------------
struct zogy_tw_state { float2* d_tw; int L; zogy_chunk_plan plan; };

void bbx_zogy2_release(bbx_ctx* ctx) {          // (name kept: bbx_ctx_destroy calls it) frees the context's twiddle table
    if (!ctx || !ctx->zogy2_state) return;
    zogy_tw_state* st = (zogy_tw_state*)ctx->zogy2_state;
    if (st->d_tw) (void)hipFree(st->d_tw);
    if (st->plan.d_start) (void)hipFree(st->plan.d_start);
    if (st->plan.d_tasks) (void)hipFree(st->plan.d_tasks);
    free(st);
    ctx->zogy2_state = nullptr;
}

#ifdef Z3_STAMPS
// diagnostic builds: buf = device buffer of 6 * Z3_STAMP_WGS * 16 uint64 (or NULL: stamps off)
extern "C" int bbx_z3_stamps(void* buf) {
    unsigned long long* p = (unsigned long long*)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(z3::g_z3_stamps), &p, sizeof(p)) == hipSuccess ? BBX_OK : BBX_ERR_HIP;
}
#endif

extern "C" int bbx_zogy_frame_supported(int L) { return bbx_zogy3_supported(L); }

extern "C" int bbx_zogy_candidates(bbx_ctx* ctx, float thr) {
    if (!ctx || !(thr >= 0.f)) return BBX_ERR_ARG;
    ctx->zcand_thr = thr;
    ctx->zcand_img = nullptr;
    return BBX_OK;
}

static int zogy_frame_entry(bbx_ctx* ctx, int ny, int nx, int size, int border, const float* d_new, const float* d_ref,
                            const float* d_sig_new, const float* d_sig_ref, const bbx_spline_image* sig_new, const bbx_spline_image* sig_ref,
                            const float* d_psf_n, const float* d_psf_r, int S,
                            const float* h_scal, float* d_D, float* d_S, float* d_Scorr, float* d_Fpsf, float* d_Fpsferr, void* stream) {
    if (!ctx || !d_new || !d_ref || !d_psf_n || !d_psf_r || !h_scal || !d_D || !d_Scorr || !d_Fpsf || !d_Fpsferr) return BBX_ERR_ARG;
    if (size < 1 || border < 0 || ny < size || nx < size || ny % size || nx % size || S < 1) return BBX_ERR_ARG;
    const int L = size + 2 * border;
    if (!bbx_zogy_frame_supported(L) || S > L || (ny / size) * (nx / size) > 4096) return BBX_ERR_ARG;
    bbx_spl spn, spr;
    const bool spl = sig_new != nullptr;
    if (spl) {
        int rc = bbx_spl_make(sig_new, ny, nx, &spn); if (rc) return rc;
        rc = bbx_spl_make(sig_ref, ny, nx, &spr); if (rc) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->zogy2_state) {
        ctx->zogy2_state = calloc(1, sizeof(zogy_tw_state));
        if (!ctx->zogy2_state) return BBX_ERR_NOMEM;
    }
    zogy_tw_state* st = (zogy_tw_state*)ctx->zogy2_state;
    if (st->L != L) {
        // twiddle table W^m = exp(-2 pi i m / L), float64 on the host, rounded once
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
    return bbx_zogy3_run(ctx, st->d_tw, &st->plan, L, ny, nx, size, border, d_new, d_ref, d_sig_new, d_sig_ref, spl ? &spn : nullptr, spl ? &spr : nullptr,
                         d_psf_n, d_psf_r, S, h_scal, d_D, d_S, d_Scorr, d_Fpsf, d_Fpsferr, s);
}

extern "C" int bbx_zogy_frame(bbx_ctx* ctx, int ny, int nx, int size, int border, const float* d_new, const float* d_ref,
                              const float* d_sig_new, const float* d_sig_ref, const float* d_psf_n, const float* d_psf_r, int S,
                              const float* h_scal, float* d_D, float* d_S, float* d_Scorr, float* d_Fpsf, float* d_Fpsferr,
                              void* stream) {
    if (!d_sig_new || !d_sig_ref) return BBX_ERR_ARG;
    return zogy_frame_entry(ctx, ny, nx, size, border, d_new, d_ref, d_sig_new, d_sig_ref, nullptr, nullptr, d_psf_n, d_psf_r, S, h_scal, d_D, d_S, d_Scorr,
                            d_Fpsf, d_Fpsferr, stream);
}

extern "C" int bbx_zogy_frame_mini(bbx_ctx* ctx, int ny, int nx, int size, int border, const float* d_new, const float* d_ref,
                                   const bbx_spline_image* sig_new, const bbx_spline_image* sig_ref, const float* d_psf_n, const float* d_psf_r,
                                   int S, const float* h_scal, float* d_D, float* d_S, float* d_Scorr, float* d_Fpsf, float* d_Fpsferr,
                                   void* stream) {
    if (!sig_new || !sig_ref) return BBX_ERR_ARG;
    return zogy_frame_entry(ctx, ny, nx, size, border, d_new, d_ref, nullptr, nullptr, sig_new, sig_ref, d_psf_n, d_psf_r, S, h_scal, d_D, d_S, d_Scorr,
                            d_Fpsf, d_Fpsferr, stream);
}

// bbx_build_flags (bbx_ctx.hip): any timing / diagnostic switch of this file compiled in?
int bbx_build_flags_zogy(void) {
#if defined(Z3_SKIP_FFT) || defined(Z3_STAMPS) || defined(Z3_FFT2X) || defined(Z3_NO_PREFETCH) || defined(Z3_NO_XCD) || defined(Z3_NO_CONTRACT) || defined(Z3_NO_ZSKIP) || \
    defined(Z3_TWPOW) || defined(Z3_ROWS_SPLIT) || defined(Z3_ROWS_XCD) || defined(Z3_EXACT_SQRT)
    return 2;
#else
    return 0;
#endif
}
