// bbx_select.hip -- exact order statistics over image segments (3-pass radix select)
// and the edge fill that uses them.
//
//   * LA-Cosmic background level = element (n-1)/2 of the sorted unmasked pixels
//     (astroscrappy's quick-select median; oracle/lacosmic.py lower_median)
//   * edge fill (blackbox.py:1968-1974): edge pixels <- np.median(channel) = mean of
//     the two middle elements of the 5280x1320 values (float32)
//
// A query = (segment, rank k).  Each pass histograms 11/11/10 key bits of the pixels
// whose higher key bits match the query's prefix; a one-wave scan kernel then picks
// the bin that holds rank k.  Keys are the order-preserving uint32 image of float32.
// Every pass is one HBM-bound read of the frame (4N bytes, + N with a mask filter).
#include "bbx_common.h"

#define SEL_MAXQ 32
#define SEL_BINS 2048

struct sel_query { uint32_t prefix; uint32_t pad; unsigned long long k; unsigned long long n; };

struct sel_args {
    const float* data; const uint8_t* mask;   // mask != NULL: only pixels with (mask & ~2) == 0 count
    int ny, nx, ysz, xsz, SX;                 // segments: rectangles ysz x xsz, SX per row
    int nq_per_seg;                           // 1 or 2 queries per segment
    int shift, bits;                          // current digit
    uint32_t himask;                          // mask of the key bits already fixed
    sel_query* q; uint32_t* hist;             // hist[query][SEL_BINS]
};

__global__ __launch_bounds__(256) void k_sel_hist(sel_args a) {
    __shared__ uint32_t lh[2][SEL_BINS];
    // block = (row Y, segment column sx)
    const int Y = blockIdx.x, sx = blockIdx.y;
    const int seg = (Y / a.ysz) * a.SX + sx;
    const int nq = a.nq_per_seg;
    for (int i = threadIdx.x; i < nq * SEL_BINS; i += blockDim.x) (&lh[0][0])[i] = 0;
    __syncthreads();
    uint32_t pre[2];
    for (int k = 0; k < nq; k++) pre[k] = a.q[seg * nq + k].prefix;
    const size_t row = (size_t)Y * a.nx + (size_t)sx * a.xsz;
    const uint32_t dmask = (1u << a.bits) - 1u;
    const int xend = ((a.xsz + 63) / 64) * 64;          // keep whole waves in the loop (ballots)
    for (int x = threadIdx.x; x < xend; x += blockDim.x) {
        bool in = x < a.xsz;
        if (in && a.mask && (a.mask[row + x] & ~BBX_MASK_COSMIC)) in = false;
        const uint32_t key = in ? f2key(a.data[row + x]) : 0u;
        for (int k = 0; k < nq; k++) {
            bool hit = in && ((key & a.himask) == pre[k]);
            const uint32_t bin = (key >> a.shift) & dmask;
            // sky-dominated frames put most of a wave into one bin: fold up to two
            // popular bins per wave into single LDS atomics, the rest go direct
            for (int round = 0; round < 2; round++) {
                const unsigned long long act = __ballot(hit);
                if (!act) break;
                const int leader = __ffsll((long long)act) - 1;
                const uint32_t b0 = __shfl(bin, leader, 64);
                const unsigned long long m = __ballot(hit && bin == b0);
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&lh[k][b0], (uint32_t)__popcll(m));
                if (bin == b0) hit = false;
            }
            if (hit) atomicAdd(&lh[k][bin], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nq * SEL_BINS; i += blockDim.x) {
        const uint32_t v = (&lh[0][0])[i];
        if (v) atomicAdd(&a.hist[(size_t)(seg * nq + i / SEL_BINS) * SEL_BINS + (i % SEL_BINS)], v);
    }
}

// one wave per query: locate the bin that contains rank k
// first pass (himask == 0) also fixes n and turns the rank rule into k
//   rule 0: k = (n-1)/2         rule 1: k = n/2 - 1 (even n) or (n-1)/2     rule 2: k = n/2
__global__ void k_sel_scan(sel_query* q, uint32_t* hist, int nquery, int shift, int bits, int first, int rule_base,
                           int nq_per_seg) {
    const int qi = blockIdx.x;
    if (qi >= nquery) return;
    uint32_t* h = hist + (size_t)qi * SEL_BINS;
    if (threadIdx.x == 0) {
        sel_query s = q[qi];
        const int nb = 1 << bits;
        if (first) {
            unsigned long long n = 0;
            for (int b = 0; b < nb; b++) n += h[b];
            s.n = n;
            const int rule = (nq_per_seg == 2) ? (1 + (qi & 1)) : rule_base;
            if (n == 0) s.k = 0;
            else if (rule == 0) s.k = (n - 1) / 2;
            else if (rule == 1) s.k = (n & 1) ? (n - 1) / 2 : n / 2 - 1;
            else s.k = n / 2;
        }
        unsigned long long acc = 0; int b = 0;
        for (; b < nb; b++) { if (acc + h[b] > s.k) break; acc += h[b]; }
        if (b == nb) b = nb - 1;
        s.k -= acc;
        s.prefix |= ((uint32_t)b) << shift;
        q[qi] = s;
        for (int i = 0; i < nb; i++) h[i] = 0;
    }
}

__global__ void k_sel_init(sel_query* q, uint32_t* hist, int nquery) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nquery * SEL_BINS; i += gridDim.x * blockDim.x) hist[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x < nquery) { q[threadIdx.x].prefix = 0; q[threadIdx.x].k = 0; q[threadIdx.x].n = 0; q[threadIdx.x].pad = 0; }
}

// results live in ctx workspace WS_SEL as sel_query[]; value = key2f(prefix)
int bbx_select_run(bbx_ctx* ctx, const float* d_data, const uint8_t* d_mask, int ny, int nx, int ysz, int xsz,
                   int nq_per_seg, int rule, sel_query** d_q_out, hipStream_t s) {
    int rc;
    if (ny % ysz || nx % xsz) return BBX_ERR_ARG;
    const int SX = nx / xsz, SY = ny / ysz;
    const int nquery = SX * SY * nq_per_seg;
    if (nquery > SEL_MAXQ || nq_per_seg < 1 || nq_per_seg > 2) return BBX_ERR_ARG;
    char* ws = (char*)bbx_ws(ctx, WS_SEL, SEL_MAXQ * sizeof(sel_query) + (size_t)SEL_MAXQ * SEL_BINS * 4, &rc);
    if (rc) return rc;
    sel_args a;
    a.data = d_data; a.mask = d_mask; a.ny = ny; a.nx = nx; a.ysz = ysz; a.xsz = xsz; a.SX = SX;
    a.nq_per_seg = nq_per_seg; a.q = (sel_query*)ws; a.hist = (uint32_t*)(ws + SEL_MAXQ * sizeof(sel_query));
    hipLaunchKernelGGL(k_sel_init, dim3(64), dim3(256), 0, s, a.q, a.hist, nquery);
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    uint32_t himask = 0;
    for (int p = 0; p < 3; p++) {
        a.shift = shifts[p]; a.bits = nbits[p]; a.himask = himask;
        hipLaunchKernelGGL(k_sel_hist, dim3(ny, SX), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_sel_scan, dim3(nquery), dim3(64), 0, s, a.q, a.hist, nquery, a.shift, a.bits, p == 0, rule,
                           nq_per_seg);
        himask |= ((1u << nbits[p]) - 1u) << shifts[p];
    }
    BBX_LAUNCH_CHECK();
    *d_q_out = a.q;
    return BBX_OK;
}

// np.median of a float32 array with an even count: float32 mean of the two middle values
__global__ void k_chan_median(const sel_query* __restrict__ q, float* __restrict__ med) {
    const int c = threadIdx.x;
    if (c < 16) {
        const float lo = key2f(q[2 * c].prefix), hi = key2f(q[2 * c + 1].prefix);
        med[c] = (q[2 * c].n & 1) ? lo : (lo + hi) * 0.5f;
    }
}

__global__ __launch_bounds__(256) void k_edge_fill(float* data, const uint8_t* __restrict__ mask, bbx_dims d,
                                                   const float* __restrict__ med) {
    const size_t npix = (size_t)d.ny * d.nx;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        if ((mask[i] & BBX_MASK_EDGE) == BBX_MASK_EDGE) {
            const int Y = (int)(i / d.nx), X = (int)(i - (size_t)Y * d.nx);
            data[i] = med[(Y / d.ysz) * 8 + X / d.xsz];
        }
    }
}

extern "C" int bbx_edge_fill(bbx_ctx* ctx, const bbx_geom* g, float* d_data, const uint8_t* d_mask,
                             float* d_chan_median, void* stream) {
    if (!ctx || !d_data || !d_mask || !d_chan_median) return BBX_ERR_ARG;
    bbx_dims d; int rc = bbx_make_dims(g, &d); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    sel_query* q;
    rc = bbx_select_run(ctx, d_data, nullptr, d.ny, d.nx, d.ysz, d.xsz, 2, 1, &q, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_chan_median, dim3(1), dim3(64), 0, s, q, d_chan_median);
    hipLaunchKernelGGL(k_edge_fill, dim3(2048), dim3(256), 0, s, d_data, d_mask, d, d_chan_median);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}
