// bbx_canny.hip -- the edge detector in front of the satellite-trail Hough transform (row a12).
// acstools.satdet.detsat (what sat_detect calls, blackbox.py:4183-4186: sigma=3, h_thresh=0.2, low_thresh
// default 0.1) prepares its edge map with numpy.percentile(4.5, 93) + skimage.exposure.rescale_intensity +
// skimage.feature.canny + skimage.morphology.remove_small_objects(60).  This file does exactly those
// steps on the 2x2-binned frame, with the libraries' arithmetic order (oracle/sattrail.py `edges`,
// pinned against scikit-image 0.18.3 by tests/golden/sat_front.npz):
//
//   percentiles    exact order statistics by a 3-pass radix select of four ranks (12 + 10 + 10 key bits),
//                  numpy's linear rule: a + float32(b - a) * g, or b - float32(b - a) * (1 - g) for g >= 0.5
//   rescale        float32: (clip(x, p1, p2) - p1) / float32(p2 - p1)
//   smoothing      scipy.ndimage.gaussian_filter(sigma, mode='constant'): axis 0 then axis 1, double
//                  accumulation centre first, then the pairs from the outside in, float32 after each axis;
//                  divided by the same filter of an all-ones image (float64) + eps
//   gradients      ndimage.sobel on the float64 image, reflecting borders; magnitude sqrt(i^2 + j^2)
//   suppression    four direction sectors, linear interpolation between the two neighbours next to the
//                  gradient direction; where sectors overlap the later one decides (gradients, magnitudes and
//                  suppression of a 64 x 16 tile in LDS: the magnitude image never goes to HBM)
//   hysteresis     8-connected components of the low mask that hold a pixel of the high mask, and of those
//                  the ones with at least 60 pixels (remove_small_objects) -- bbx_mask.hip's sparse union-find
//
// Traffic on a 5280^2 binned frame (112 MB as float32): ~1.3 GB in 8 passes; everything after the suppression
// works on a sparse pixel list.
#include "bbx_common.h"

#define CANNY_MAXR 32

struct canny_par {
    unsigned prefix[4];                    // radix select: key bits fixed so far, per rank
    unsigned long long rank[4];            // rank inside the current bin
    double p1, p2, low, high;
    float p1f, p2f, den;
    int degenerate;
    double w[CANNY_MAXR + 1];              // Gaussian weights, centre first
    int radius;
};

__device__ __forceinline__ unsigned fkey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void k_canny_init(canny_par* p, size_t n, double q1, double q2, const double* __restrict__ w, int radius, unsigned* hist) {
    const int t = threadIdx.x;
    for (int i = t; i < 4 * 4096; i += blockDim.x) hist[i] = 0;
    if (t <= radius) p->w[t] = w[t];
    if (t == 0) {
        // numpy: virtual index (n - 1) q, neighbours floor / floor + 1
        const double v1 = (double)(n - 1) * q1, v2 = (double)(n - 1) * q2;
        const unsigned long long l1 = (unsigned long long)floor(v1), l2 = (unsigned long long)floor(v2);
        p->rank[0] = l1; p->rank[1] = l1 + 1 < n ? l1 + 1 : n - 1;
        p->rank[2] = l2; p->rank[3] = l2 + 1 < n ? l2 + 1 : n - 1;
        for (int q = 0; q < 4; q++) p->prefix[q] = 0;
        p->radius = radius; p->degenerate = 0;
    }
}

// one digit of the select: pass 0 counts the top 12 key bits of every pixel (one histogram for all ranks),
// passes 1 and 2 the next 10 bits of the pixels that match a rank's prefix (one histogram per rank)
template <int PASS>
__global__ __launch_bounds__(256) void k_canny_hist(const float* __restrict__ b, size_t n, const canny_par* __restrict__ p,
                                                    unsigned* __restrict__ hist) {
    __shared__ unsigned h[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) h[i] = 0;
    __syncthreads();
    unsigned pre[4];
#pragma unroll
    for (int q = 0; q < 4; q++) pre[q] = p->prefix[q];
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    const float4* b4 = (const float4*)b;
    auto count = [&](float f) {
        const unsigned k = fkey(f);
        if (PASS == 0) {
            // the top 12 key bits of a sky frame fall into two or three bins: 64 lanes adding to one LDS word queue up one
            // after the other (this pass took 60 us for 111 MB).  Up to two rounds in which the bin of the first pending lane
            // is counted once for every lane that shares it; what is left goes one by one.
            const unsigned bin = k >> 20;
            bool pending = true;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const unsigned long long act = __builtin_amdgcn_ballot_w64(pending);
                if (!act) break;
                const int leader = (int)__builtin_ctzll(act);
                const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                const bool same = pending && bin == b0;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(same);
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&h[b0], (unsigned)__popcll(m));
                pending = pending && !same;
            }
            if (pending) atomicAdd(&h[bin], 1u);
        } else {
            // (the four ranks -- two neighbours each of the 4.5 % and the 93 % point -- mostly share their prefix: one histogram
            // per distinct prefix, k_canny_scan reads the first one of a run of equal prefixes)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q > 0 && pre[q] == pre[q - 1]) continue;
                if (PASS == 1 && (k >> 20) == pre[q]) atomicAdd(&h[q * 1024 + ((k >> 10) & 1023u)], 1u);
                if (PASS == 2 && (k >> 10) == pre[q]) atomicAdd(&h[q * 1024 + (k & 1023u)], 1u);
            }
        }
    };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 f = b4[i];
        count(f.x); count(f.y); count(f.z); count(f.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) count(b[n4 * 4 + threadIdx.x]);      // the last n mod 4 pixels
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) if (h[i]) atomicAdd(&hist[i], h[i]);
}

// locate the bin of every rank, extend its prefix; after the last pass: the four order statistics -> parameters
template <int PASS>
__global__ __launch_bounds__(256) void k_canny_scan(canny_par* p, unsigned* hist) {
    __shared__ unsigned sh[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = hist[i];
    __syncthreads();
    // wave q serves rank q: every lane sums a chunk of the histogram, the wave locates the chunk, lane 0 the bin
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int qh = q;                                                    // the histogram of this rank's prefix: the first of a run of equal prefixes
    if (PASS != 0) { while (qh > 0 && p->prefix[qh] == p->prefix[qh - 1]) qh--; }
    __syncthreads();                                               // (all prefixes read before any is extended below)
    const unsigned* h = PASS == 0 ? sh : sh + qh * 1024;
    constexpr int NB = PASS == 0 ? 4096 : 1024, CH = NB / 64;
    unsigned long long mine = 0;
    for (int i = 0; i < CH; i++) mine += h[lane * CH + i];
    unsigned long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    const unsigned long long r = p->rank[q];
    const unsigned long long m = __builtin_amdgcn_ballot_w64(incl > r);       // first chunk whose running total exceeds the rank
    const int chunk = m ? __ffsll((long long)m) - 1 : 63;
    const unsigned long long before = __shfl(incl - mine, chunk, 64);
    if (lane == 0) {
        unsigned long long cum = before;
        int bin = chunk * CH + CH - 1;
        for (int i = 0; i < CH; i++) { const unsigned v = h[chunk * CH + i]; if (cum + v > r) { bin = chunk * CH + i; break; } cum += v; }
        p->rank[q] = r - cum;
        p->prefix[q] = PASS == 0 ? (unsigned)bin : ((p->prefix[q] << 10) | (unsigned)bin);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) hist[i] = 0;
}

__global__ void k_canny_params(canny_par* p, size_t n, double q1, double q2, double low_frac, double high_frac) {
    if (threadIdx.x != 0) return;
    float os[4];
    for (int q = 0; q < 4; q++) os[q] = fkey_inv(p->prefix[q]);
    double pc[2];
    for (int j = 0; j < 2; j++) {
        const double v = (double)(n - 1) * (j ? q2 : q1), g = v - floor(v);
        const float a = os[2 * j], b = os[2 * j + 1];
        const double diff = (double)(b - a);                       // the difference is taken in float32
        pc[j] = g < 0.5 ? (double)a + diff * g : (double)b - diff * (1.0 - g);
    }
    if (pc[0] < 0.0) pc[0] = 0.0;                                  // acstools: "if p1 < 0: p1 = 0.0"
    p->p1 = pc[0]; p->p2 = pc[1];
    p->p1f = (float)pc[0]; p->p2f = (float)pc[1]; p->den = (float)(pc[1] - pc[0]);
    p->degenerate = !(pc[1] > pc[0]);
    const float immax = (p->p2f - p->p1f) / p->den;                // the rescaled value of every pixel >= p2
    p->low = (double)immax * low_frac; p->high = (double)immax * high_frac;
}

// rescale + Gaussian along y: a thread produces RY consecutive rows of one column from a register window (x across lanes)
#define GV_RY 16
template <int R>
__global__ __launch_bounds__(256) void k_canny_gauss_v(const float* __restrict__ img, int ny, int nx, const canny_par* __restrict__ p,
                                                       float* __restrict__ tmp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = blockIdx.y * GV_RY;
    if (x >= nx) return;
    double w[R + 1];
#pragma unroll
    for (int j = 0; j <= R; j++) w[j] = p->w[j];
    const float p1 = p->p1f, p2 = p->p2f, den = p->den;
    double in[GV_RY + 2 * R];
#pragma unroll
    for (int k = 0; k < GV_RY + 2 * R; k++) {
        const int y = y0 - R + k;
        // the rescaled pixel (float32: np.clip, subtract, divide), zero outside the frame
        in[k] = (y >= 0 && y < ny) ? (double)((fminf(fmaxf(img[(size_t)y * nx + x], p1), p2) - p1) / den) : 0.0;
    }
#pragma unroll
    for (int r = 0; r < GV_RY; r++) {
        if (y0 + r >= ny) break;
        double t = in[r + R] * w[0];
#pragma unroll
        for (int j = R; j >= 1; j--) t += (in[r + R - j] + in[r + R + j]) * w[j];
        tmp[(size_t)(y0 + r) * nx + x] = (float)t;
    }
}

// Gaussian along x + normalisation by the filtered all-ones image -> smoothed image, float64.  A workgroup takes 256
// pixels of one row: the row piece with its halo goes through LDS; the all-ones image depends on the row (v0) and,
// away from the left / right border, not on the column, so its interior value is computed once per thread from v0.
#ifndef GH_ROWS
#define GH_ROWS 8                                          // rows per workgroup: one barrier, one set of weights for all of them
#endif
// (tried in round 4: four neighbouring pixels per thread from one register window, the row in four x-mod-4 planes so that
// the lanes read consecutive words -- 7 LDS reads per pixel instead of 25 -- 144 us against 116: the kernel is bound by its
// chains of dependent float64 operations and the per-workgroup set-up, not by the LDS)
template <int R>
__global__ __launch_bounds__(256) void k_canny_gauss_h(const float* __restrict__ tmp, int ny, int nx, const canny_par* __restrict__ p,
                                                       double* __restrict__ sm) {
    __shared__ double row[GH_ROWS][256 + 2 * R];            // the row pieces, converted once
    __shared__ double s_v0[GH_ROWS], s_bl[GH_ROWS];         // the all-ones image: after the first pass (per row), after both (away from the borders)
    const int x0 = blockIdx.x * 256, x = x0 + threadIdx.x, yb = blockIdx.y * GH_ROWS;
    for (int i = threadIdx.x; i < GH_ROWS * (256 + 2 * R); i += 256) {
        const int r = i / (256 + 2 * R), c = i - r * (256 + 2 * R), xx = x0 - R + c, yy = yb + r;
        row[r][c] = (yy < ny && xx >= 0 && xx < nx) ? (double)tmp[(size_t)yy * nx + xx] : 0.0;
    }
    double w[R + 1];
#pragma unroll
    for (int j = 0; j <= R; j++) w[j] = p->w[j];
    if (threadIdx.x < GH_ROWS) {
        // the all-ones image through the same two passes (float64 throughout): column-independent after the first pass,
        // and after the second one too wherever the window does not touch the left / right border
        const int y = yb + threadIdx.x;
        double v0 = 1.0 * w[0];
#pragma unroll
        for (int j = R; j >= 1; j--) v0 += ((y - j >= 0 ? 1.0 : 0.0) + (y + j < ny ? 1.0 : 0.0)) * w[j];
        double bl = v0 * w[0];
#pragma unroll
        for (int j = R; j >= 1; j--) bl += (v0 + v0) * w[j];
        s_v0[threadIdx.x] = v0; s_bl[threadIdx.x] = bl;
    }
    __syncthreads();
    if (x >= nx) return;
    const bool inner = x >= R && x + R < nx;
    for (int r = 0; r < GH_ROWS; r++) {
        const int y = yb + r;
        if (y >= ny) break;
        double t = row[r][threadIdx.x + R] * w[0];
#pragma unroll
        for (int j = R; j >= 1; j--) t += (row[r][threadIdx.x + R - j] + row[r][threadIdx.x + R + j]) * w[j];
        const float s32 = (float)t;
        double bl = s_bl[r];
        if (!inner) {
            const double v0 = s_v0[r];
            bl = v0 * w[0];
#pragma unroll
            for (int j = R; j >= 1; j--) bl += ((x - j >= 0 ? v0 : 0.0) + (x + j < nx ? v0 : 0.0)) * w[j];
        }
        sm[(size_t)y * nx + x] = (double)s32 / (bl + 2.220446049250313e-16);
    }
}

// Gradients, magnitude, non-maximum suppression and both thresholds on a 64 x 16 tile: the smoothed image with a
// 2-pixel halo lives in LDS (float64 + a float32 copy for the screen); the low-mask pixels of the tile go to the list in
// one reservation, each with a flag "also in the high mask".
//   ndimage.sobel(axis): derivative along the axis ([-1, 0, 1]), [1, 2, 1] across it, reflecting borders (index -1 -> 0:
//   the clamped loads below); magnitude sqrt(i^2 + j^2)
#define CT_X 64
#define CT_Y 16
#ifndef CT_NT
#define CT_NT 8
#endif
//                                                        tiles (stacked in y) per workgroup: one list reservation for all of them
#define CT_Q 2048                                          // (every workgroup's returning atomic on the one list counter costs ~11 ns)
__global__ __launch_bounds__(256) void k_canny_tile(const double* __restrict__ sm, int ny, int nx, const canny_par* __restrict__ p,
                                                    uint32_t* list, uint8_t* hflag, int32_t* cnt, uint32_t cap, int32_t* err) {
    __shared__ double s_sm[CT_Y + 4][CT_X + 4];
    __shared__ float s_smf[CT_Y + 4][CT_X + 4];             // the same tile in float32: the screen below
    __shared__ uint32_t q[CT_Q];
    __shared__ uint16_t s_cand[CT_X * CT_Y];                // pixels of the tile that passed the screen
    __shared__ unsigned qn, gbase, ncand;
    const int tid = threadIdx.x, x0 = blockIdx.x * CT_X;
    if (tid == 0) { qn = 0; ncand = 0; }
    const double low = p->low, high = p->high;
    const bool live = !p->degenerate;
    // Screen: 99 % of the pixels are far below the low threshold.  Their float32 gradient magnitude (squared; inputs in
    // [0, 1], error ~2e-5 of low^2 at the threshold) settles that with a margin of 2e-3; only the pixels that pass it -- and,
    // for the non-maximum test, the neighbours they look at -- get the float64 Sobel sums and the correctly rounded square
    // root that skimage's float64 arithmetic demands (they were 60 float64 operations for every pixel of the frame).
    const float low2f = (float)(low * low) * (1.f - 2e-3f) - 1e-12f;
    // exact gradient / magnitude at s_mag position (r, c) = image position (y0 - 1 + r, x0 - 1 + c): the operations of rounds 2-3
    auto grad = [&](int r, int c, double& is, double& js) {
        const double d0 = s_sm[r][c + 2] - s_sm[r][c], d1 = s_sm[r + 1][c + 2] - s_sm[r + 1][c], d2 = s_sm[r + 2][c + 2] - s_sm[r + 2][c];
        js = d1 * 2.0 + (d0 + d2);
        const double e0 = s_sm[r + 2][c] - s_sm[r][c], e1 = s_sm[r + 2][c + 1] - s_sm[r][c + 1], e2 = s_sm[r + 2][c + 2] - s_sm[r][c + 2];
        is = e1 * 2.0 + (e0 + e2);
    };
    auto mag = [&](int r, int c) { double is, js; grad(r, c, is, js); return sqrt(is * is + js * js); };
    for (int it = 0; it <= CT_NT; it++) {
        const int y0 = (blockIdx.y * CT_NT + it) * CT_Y;
        const bool more = it < CT_NT && y0 < ny;
        __syncthreads();
        // flush when the next tile might not fit, and at the end
        const unsigned n = qn;
        if (n && (!more || n + CT_X * CT_Y > CT_Q)) {
            if (tid == 0) gbase = atomicAdd((unsigned*)cnt, n);
            __syncthreads();
            for (unsigned t = tid; t < n; t += 256) {
                if (gbase + t < cap) { list[gbase + t] = q[t] & 0x7fffffffu; hflag[gbase + t] = (uint8_t)(q[t] >> 31); }
                else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
            }
            __syncthreads();
            if (tid == 0) qn = 0;
        }
        if (!more) break;
        for (int i = tid; i < (CT_Y + 4) * (CT_X + 4); i += 256) {
            const int r = i / (CT_X + 4), c = i - r * (CT_X + 4);
            const int yy = min(max(y0 - 2 + r, 0), ny - 1), xx = min(max(x0 - 2 + c, 0), nx - 1);
            const double v = sm[(size_t)yy * nx + xx];
            s_sm[r][c] = v; s_smf[r][c] = (float)v;
        }
        __syncthreads();
        // (a) the screen, every pixel of the tile: those that pass go to a list of the tile ...
        for (int i = tid; i < CT_X * CT_Y; i += 256) {
            const int ty = i / CT_X, tx = i - ty * CT_X, y = y0 + ty, x = x0 + tx;
            if (!live || y < 1 || y > ny - 2 || x < 1 || x > nx - 2) continue;      // binary_erosion of the all-ones mask
            const int r = ty + 1, c = tx + 1;                                      // s_mag position; s_sm centre is [r + 1][c + 1]
            const float d0 = s_smf[r][c + 2] - s_smf[r][c], d1 = s_smf[r + 1][c + 2] - s_smf[r + 1][c], d2 = s_smf[r + 2][c + 2] - s_smf[r + 2][c];
            const float e0 = s_smf[r + 2][c] - s_smf[r][c], e1 = s_smf[r + 2][c + 1] - s_smf[r][c + 1], e2 = s_smf[r + 2][c + 2] - s_smf[r][c + 2];
            const float jf = d1 * 2.f + (d0 + d2), if_ = e1 * 2.f + (e0 + e2);
            if (if_ * if_ + jf * jf >= low2f) s_cand[atomicAdd(&ncand, 1u)] = (uint16_t)i;
        }
        __syncthreads();
        // (b) ... which all threads then work through in float64 (a wave that met one such pixel among its 64 would otherwise
        // run the whole exact path for it: more float64 work than without the screen)
        const unsigned nc = ncand;
        for (unsigned t = tid; t < nc; t += 256) {
            const int i = s_cand[t];
            const int ty = i / CT_X, tx = i - ty * CT_X, y = y0 + ty, x = x0 + tx;
            const int r = ty + 1, c = tx + 1;
            double is, js;
            grad(r, c, is, js);
            const double m = sqrt(is * is + js * js);
            if (!(m > 0.0 && m >= low)) continue;
            const double ai = fabs(is), aj = fabs(js);
            const bool same = (is >= 0 && js >= 0) || (is <= 0 && js <= 0), opp = (is <= 0 && js >= 0) || (is >= 0 && js <= 0);
            // c2 w + c1 (1 - w) <= m on both sides of the gradient direction, c2 the diagonal neighbour, c1 the one along an
            // axis; of the four sectors (skimage tests them in this order, a later one overwrites an earlier one) the last
            // that applies decides
            int sec = 0;
            if (same && ai >= aj) sec = 1;
            if (same && ai <= aj) sec = 2;
            if (opp && ai <= aj) sec = 3;
            if (opp && ai >= aj) sec = 4;
            if (!sec) continue;
            const int dr2 = sec <= 2 ? 1 : -1, dr1 = (sec == 1) ? 1 : (sec == 4 ? -1 : 0), dc1 = (sec == 2 || sec == 3) ? 1 : 0;
            const double w = (sec == 1 || sec == 4) ? aj / ai : ai / aj;
            const bool loc = (mag(r + dr2, c + 1) * w + mag(r + dr1, c + dc1) * (1 - w) <= m) &&
                             (mag(r - dr2, c - 1) * w + mag(r - dr1, c - dc1) * (1 - w) <= m);
            if (loc) { const unsigned k = atomicAdd(&qn, 1u); q[k] = (uint32_t)((size_t)y * nx + x) | (m >= high ? 0x80000000u : 0u); }
        }
        __syncthreads();
        if (tid == 0) ncand = 0;
    }
}

int bbx_cc_filter_list(bbx_ctx* ctx, const uint32_t* d_list, const int32_t* d_cnt, size_t cap, int ny, int nx, const uint8_t* d_flag,
                       int min_size, uint32_t* d_out, int32_t* d_out_cnt, uint32_t out_cap, hipStream_t s);

// binned frame -> list of edge pixels (after hysteresis and remove_small_objects) in d_out / *d_out_cnt
int bbx_canny_edges(bbx_ctx* ctx, const float* d_bin, int ny, int nx, const double* h_gauss, int radius, double low_frac, double high_frac,
                    int min_size, uint32_t* d_out, int32_t* d_out_cnt, uint32_t out_cap, hipStream_t s) {
    if (radius < 1 || radius > CANNY_MAXR || ny < 3 || nx < 3) return BBX_ERR_ARG;
    const size_t n = (size_t)ny * nx;
    int rc;
    const size_t cap = n / 8 + 4096;                               // low-mask pixels
    const size_t o_par = 0, o_hist = 4096, o_w = o_hist + 4 * 4096 * 4, o_tmp = o_w + 1024;
    const size_t o_sm = o_tmp + ((n * 4 + 255) & ~(size_t)255);
    const size_t o_list = o_sm + ((n * 8 + 255) & ~(size_t)255), o_flag = o_list + ((cap * 4 + 255) & ~(size_t)255);
    const size_t total = o_flag + cap + 256;
    char* ws = (char*)bbx_ws(ctx, WS_CANNY, total, &rc); if (rc) return rc;
    canny_par* par = (canny_par*)(ws + o_par); unsigned* hist = (unsigned*)(ws + o_hist); double* d_w = (double*)(ws + o_w);
    float* tmp = (float*)(ws + o_tmp); double* sm = (double*)(ws + o_sm);
    uint32_t* list = (uint32_t*)(ws + o_list); uint8_t* hflag = (uint8_t*)(ws + o_flag);
    int32_t* cnt = &ctx->d_counters[CNT_TMP];
    const double q1 = 4.5 / 100.0, q2 = 93.0 / 100.0;             // np.percentile(image, (4.5, 93.0))
    BBX_HIP(hipMemcpyAsync(d_w, h_gauss, (size_t)(radius + 1) * 8, hipMemcpyHostToDevice, s));
    BBX_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t), s));
#ifdef SATV                                                        // timing knock-outs (tools/exp/satvar.sh; never in the product build)
    const int satv = getenv("BBX_DBG_SAT") ? atoi(getenv("BBX_DBG_SAT")) : 0;
    if (satv >= 5) return BBX_OK;
#endif
    hipLaunchKernelGGL(k_canny_init, dim3(1), dim3(256), 0, s, par, n, q1, q2, d_w, radius, hist);
    hipLaunchKernelGGL(k_canny_hist<0>, dim3(1024), dim3(256), 0, s, d_bin, n, par, hist);
    hipLaunchKernelGGL(k_canny_scan<0>, dim3(1), dim3(256), 0, s, par, hist);
    hipLaunchKernelGGL(k_canny_hist<1>, dim3(1024), dim3(256), 0, s, d_bin, n, par, hist);
    hipLaunchKernelGGL(k_canny_scan<1>, dim3(1), dim3(256), 0, s, par, hist);
    hipLaunchKernelGGL(k_canny_hist<2>, dim3(1024), dim3(256), 0, s, d_bin, n, par, hist);
    hipLaunchKernelGGL(k_canny_scan<2>, dim3(1), dim3(256), 0, s, par, hist);
    hipLaunchKernelGGL(k_canny_params, dim3(1), dim3(64), 0, s, par, n, q1, q2, low_frac, high_frac);
    const dim3 gx((nx + 255) / 256, (ny + GH_ROWS - 1) / GH_ROWS), gv((nx + 255) / 256, (ny + GV_RY - 1) / GV_RY);
#ifdef SATV
    if (satv >= 4) return BBX_OK;
#endif
    if (radius == 12) {
        hipLaunchKernelGGL(k_canny_gauss_v<12>, gv, dim3(256), 0, s, d_bin, ny, nx, par, tmp);
        hipLaunchKernelGGL(k_canny_gauss_h<12>, gx, dim3(256), 0, s, tmp, ny, nx, par, sm);
    } else {
        return BBX_ERR_ARG;                                        // sigma = 3 (radius 12) is what sat_detect asks for
    }
#ifdef SATV
    if (satv >= 3) return BBX_OK;
#endif
    hipLaunchKernelGGL(k_canny_tile, dim3((nx + CT_X - 1) / CT_X, (ny + CT_Y * CT_NT - 1) / (CT_Y * CT_NT)), dim3(256), 0, s, sm, ny, nx, par, list, hflag, cnt,
                       (uint32_t)cap, ctx->d_err);
    BBX_LAUNCH_CHECK();
#ifdef SATV
    if (satv >= 2) return BBX_OK;
#endif
    return bbx_cc_filter_list(ctx, list, cnt, cap, ny, nx, hflag, min_size, d_out, d_out_cnt, out_cap, s);
}

// the edge map itself (tests; the trail detector consumes the list): d_map[ny * nx] = 1 at edge pixels, 0 elsewhere
__global__ __launch_bounds__(256) void k_canny_scatter(const uint32_t* __restrict__ list, const int32_t* __restrict__ cnt, uint32_t cap,
                                                       uint8_t* __restrict__ map) {
    const uint32_t n = min((uint32_t)*cnt, cap);
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) map[list[e]] = 1;
}

extern "C" int bbx_canny_edge_map(bbx_ctx* ctx, int ny, int nx, const float* d_img, const double* h_gauss, int gauss_radius,
                                  double low_frac, double high_frac, int min_size, uint8_t* d_map, int32_t* d_count, void* stream) {
    if (!ctx || !d_img || !h_gauss || !d_map || !d_count || ((uintptr_t)d_img) % 16) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const size_t n = (size_t)ny * nx, cap = n / 8 + 4096;
    uint32_t* out = (uint32_t*)bbx_ws(ctx, WS_CRLIST, cap * sizeof(uint32_t), &rc); if (rc) return rc;
    BBX_HIP(hipMemsetAsync(d_map, 0, n, s));
    rc = bbx_canny_edges(ctx, d_img, ny, nx, h_gauss, gauss_radius, low_frac, high_frac, min_size, out, d_count, (uint32_t)cap, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_canny_scatter, dim3(256), dim3(256), 0, s, out, d_count, (uint32_t)cap, d_map);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

// bbx_build_flags (bbx_ctx.hip): this file's timing knock-outs compiled in?
int bbx_build_flags_canny(void) {
#ifdef SATV
    return 8;
#else
    return 0;
#endif
}
