// bbx_select.hip -- exact order statistics over image segments, and the edge fill.
//
//   * LA-Cosmic background level = element (n-1)/2 of the sorted unmasked pixels
//     (astroscrappy's quick-select median; oracle/lacosmic.py lower_median)
//   * edge fill (blackbox.py:1968-1974): edge pixels <- np.median(channel) = float32
//     mean of the two middle elements of the channel's 5280x1320 values
//
// Fast path = bracketed select (bbx_bsel.h): the frame is not re-read; the kernels that
// stream it anyway feed a ~5 % side buffer, and a 3-digit (11/11/10 bit) radix select runs
// on that buffer.  Slow path (exact fallback, taken only when a rank falls outside its
// bracket) = the same radix select over the whole frame.  Keys are the order-preserving
// uint32 image of float32.
#include <algorithm>
#include "bbx_bsel.h"

#define SEL_BINS 2048

// fold popular bins per wave into single LDS atomics (sky-dominated data put most of a
// wave into one bin), the rest go direct.  All 64 lanes must call.
__device__ __forceinline__ void hist_add(uint32_t* lh, uint32_t bin, bool hit) {
    for (int round = 0; round < 2; round++) {
        const unsigned long long act = __ballot(hit);
        if (!act) return;
        const int leader = __ffsll((long long)act) - 1;
        const uint32_t b0 = __shfl(bin, leader, 64);
        const unsigned long long m = __ballot(hit && bin == b0);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&lh[b0], (uint32_t)__popcll(m));
        if (bin == b0) hit = false;
    }
    if (hit) atomicAdd(&lh[bin], 1u);
}

// wave-level: locate the bin holding rank rk in histogram h[nb]; returns (bin, rank inside)
__device__ __forceinline__ bool wave_find_bin(const uint32_t* h, int nb, unsigned long long rk, int* bin,
                                              unsigned long long* rk_in, unsigned long long* total) {
    const int lane = threadIdx.x & 63, per = nb / 64;
    unsigned long long mine = 0;
    for (int k = 0; k < per; k++) mine += h[lane * per + k];
    unsigned long long incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    *total = __shfl(incl, 63, 64);
    const unsigned long long excl = incl - mine;
    if (rk >= excl && rk < incl) {
        unsigned long long acc = excl; int b = lane * per;
        for (int k = 0; k < per; k++, b++) { if (acc + h[b] > rk) break; acc += h[b]; }
        *bin = b; *rk_in = rk - acc;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------
// workgroup-wide radix select of two ranks in a small float array (the samples)
// ---------------------------------------------------------------------------------
__device__ void wg_select2(const float* __restrict__ v, uint32_t count, unsigned long long r0, unsigned long long r1,
                           float* out0, float* out1) {
    __shared__ uint32_t lh[2][SEL_BINS];
    __shared__ uint32_t s_prefix[2];
    __shared__ unsigned long long s_rank[2];
    const int tid = threadIdx.x;
    if (tid == 0) { s_prefix[0] = s_prefix[1] = 0; s_rank[0] = r0; s_rank[1] = r1; }
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    uint32_t himask = 0;
    for (int p = 0; p < 3; p++) {
        for (int i = tid; i < 2 * SEL_BINS; i += blockDim.x) (&lh[0][0])[i] = 0;
        __syncthreads();
        const uint32_t pre0 = s_prefix[0], pre1 = s_prefix[1];
        const uint32_t dmask = (1u << nbits[p]) - 1u;
        const uint32_t cend = ((count + 63u) / 64u) * 64u;
        for (uint32_t i = tid; i < cend; i += blockDim.x) {
            const bool in = i < count;
            const uint32_t key = in ? f2key(v[i]) : 0u;
            const uint32_t bin = (key >> shifts[p]) & dmask;
            hist_add(lh[0], bin, in && ((key & himask) == pre0));
            if (pre1 != pre0) hist_add(lh[1], bin, in && ((key & himask) == pre1));
        }
        __syncthreads();
        const int q = tid >> 6;
        if (q < 2) {
            const uint32_t* h = (q == 1 && pre1 == pre0) ? lh[0] : lh[q];
            int b; unsigned long long rin, total;
            if (wave_find_bin(h, 1 << nbits[p], s_rank[q], &b, &rin, &total)) {
                s_rank[q] = rin;
                s_prefix[q] |= ((uint32_t)b) << shifts[p];
            }
        }
        __syncthreads();
        himask |= dmask << shifts[p];
    }
    if (tid == 0) { *out0 = key2f(s_prefix[0]); *out1 = key2f(s_prefix[1]); }
    __syncthreads();
}

// ---------------------------------------------------------------------------------
// bracketed select: sample, bracket
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bsel_init(bsel_seg* seg, bsel_shard* shard, int nseg, uint32_t* hist, int nhist,
                                                   int* anyfail) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = i; k < nhist; k += gridDim.x * blockDim.x) hist[k] = 0;
    if (i == 0) *anyfail = 0;
    if (i < nseg * BSEL_NSH) { bsel_shard z; memset(&z, 0, sizeof(z)); shard[i] = z; }
    if (i < nseg) {
        bsel_seg s;
        s.lo = 0.f; s.hi = 0.f; s.nsample = 0; s.nbuf = 0; s.below = 0; s.n = 0; s.fail = 0; s.pad = 0;
        s.result[0] = s.result[1] = 0.f;
        s.wlo = -__builtin_huge_val(); s.whi = __builtin_huge_val();
        seg[i] = s;
    }
}

__global__ __launch_bounds__(256) void k_bsel_sample(const float* __restrict__ data, const uint8_t* __restrict__ mask,
                                                     int nx, int ysz, int xsz, int SX, bsel_seg* seg,
                                                     float* __restrict__ samples, int skip_zero) {
    const int sg = blockIdx.y;
    const int sy = sg / SX, sx = sg - sy * SX;
    const unsigned long long npix = (unsigned long long)ysz * xsz;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // 0 .. BSEL_S-1
    float v = __builtin_huge_valf();
    int ok = 0;
    if (npix >= BSEL_S || (unsigned long long)i < npix) {
        const unsigned long long j = (npix >= BSEL_S) ? ((unsigned long long)i * npix) / BSEL_S : (unsigned long long)i;
        const int y = (int)(j / xsz), x = (int)(j - (unsigned long long)y * xsz);
        const size_t o = (size_t)(sy * ysz + y) * nx + (size_t)sx * xsz + x;
        if (!mask || !(mask[o] & ~BBX_MASK_COSMIC)) {
            v = data[o];
            ok = bsel_value_ok(v, seg[sg].wlo, seg[sg].whi, skip_zero) ? 1 : 0;
            if (!ok) v = __builtin_huge_valf();
        }
    }
    samples[(size_t)sg * BSEL_S + i] = v;
    ok = wave_sum_i32(ok);
    if ((threadIdx.x & 63) == 0 && ok) atomicAdd(&seg[sg].nsample, (unsigned)ok);
}

__global__ __launch_bounds__(1024) void k_bsel_bracket(bsel_seg* seg, const float* __restrict__ samples) {
    const int sg = blockIdx.x;
    __shared__ float lo, hi;
    const uint32_t m = seg[sg].nsample;
    if (m < 256) {                       // too few valid samples: go straight to the full select
        if (threadIdx.x == 0) { seg[sg].fail = 1; seg[sg].lo = __builtin_huge_valf(); seg[sg].hi = -__builtin_huge_valf(); }
        return;
    }
    const long long target = ((long long)m - 1) / 2;
    const long long margin = (long long)(3.0f * sqrtf((float)m)) + 8;      // +-6 sigma of the sample rank
    const long long rlo = target - margin < 0 ? 0 : target - margin;
    const long long rhi = target + margin > (long long)m - 1 ? (long long)m - 1 : target + margin;
    wg_select2(samples + (size_t)sg * BSEL_S, BSEL_S, (unsigned long long)rlo, (unsigned long long)rhi, &lo, &hi);
    if (threadIdx.x == 0) {
        seg[sg].lo = (rlo == 0) ? -__builtin_huge_valf() : lo;
        seg[sg].hi = (rhi == (long long)m - 1) ? __builtin_huge_valf() : hi;
    }
}

// ---------------------------------------------------------------------------------
// finish: ranks, then 3 digit passes; source = side buffer, or the frame for segments
// whose bracket failed
// ---------------------------------------------------------------------------------
// Digits are taken from key - klo, where [klo, khi] are the keys of the bracket: the values
// in a side buffer span only B = bitlength(khi - klo) bits (typically 16-18 for a sky
// bracket), so ceil(B/11) passes resolve the rank and their histograms are spread over
// many bins (no hot bin, no atomic pile-up).  Failed segments use klo = 0, B = 32.
struct sel_args {
    const float* data; const uint8_t* mask;
    int nx, ysz, xsz, SX;
    int pass;
    bsel_dev b;
    uint32_t* prefix;                         // prefix[seg][2]
    unsigned long long* rank;                 // rank[seg][2] (remaining)
    uint32_t* hist;                           // hist[seg][2][SEL_BINS]
    uint32_t* klo;                            // klo[seg]
    int* nbits;                               // B[seg]
    int* anyfail;                             // != 0: some segment needs the exact select over the frame
};

// digit of pass p for a segment with B significant bits: false when the pass is not needed
struct sel_digit { int shift, bits; uint32_t dmask; bool fold; };
__device__ __forceinline__ bool sel_digit_of(int B, int pass, sel_digit* d) {
    const int rem = B - 11 * pass;
    if (rem <= 0) return false;
    d->bits = rem < 11 ? rem : 11;
    d->shift = rem - d->bits;
    d->dmask = (1u << d->bits) - 1u;
    d->fold = (pass == 0 && B > 24);          // exponent-level digit: sky-dominated data pile into few bins
    return true;
}
// do the bits above the current digit equal the prefix chosen so far?
__device__ __forceinline__ bool sel_match(uint32_t key, uint32_t prefix, const sel_digit& d) {
    const int hs = d.shift + d.bits;
    return hs >= 32 ? true : ((key >> hs) == (prefix >> hs));
}

__global__ void k_sel_plan(sel_args a, int nseg) {
    const int sg = threadIdx.x;
    if (sg >= nseg) return;
    bsel_seg* s = &a.b.seg[sg];
    unsigned long long n = 0, below = 0, nbuf = 0;
    bool over = false;
    for (int sh = 0; sh < BSEL_NSH; sh++) {
        const bsel_shard* q = &a.b.shard[sg * BSEL_NSH + sh];
        n += q->n; below += q->below; nbuf += q->nbuf;
        over |= q->nbuf > a.b.capS;
    }
    s->n = n; s->below = below; s->nbuf = (uint32_t)nbuf;
    const unsigned long long k0 = n ? (n - 1) / 2 : 0, k1 = n / 2;
    if (s->fail || over || k0 < below || k1 >= below + nbuf) s->fail = 1;
    a.rank[sg * 2] = s->fail ? k0 : k0 - below;
    a.rank[sg * 2 + 1] = s->fail ? k1 : k1 - below;
    a.prefix[sg * 2] = a.prefix[sg * 2 + 1] = 0;
    uint32_t klo = 0; int B = 32;
    if (!s->fail) {
        klo = f2key(s->lo);
        const uint32_t range = f2key(s->hi) - klo;
        B = range ? 32 - __clz(range) : 1;
    }
    a.klo[sg] = klo; a.nbits[sg] = B;
    if (s->fail) atomicOr(a.anyfail, 1);
}

__device__ __forceinline__ void sel_hist_one(uint32_t* lh0, uint32_t* lh1, uint32_t key, bool in, uint32_t pre0, uint32_t pre1,
                                             const sel_digit& d) {
    const uint32_t bin = (key >> d.shift) & d.dmask;
    const bool h0 = in && sel_match(key, pre0, d);
    const bool h1 = in && pre1 != pre0 && sel_match(key, pre1, d);
    if (d.fold) {
        hist_add(lh0, bin, h0);
        if (pre1 != pre0) hist_add(lh1, bin, h1);
    } else {
        if (h0) atomicAdd(&lh0[bin], 1u);
        if (h1) atomicAdd(&lh1[bin], 1u);
    }
}

// histogram of one key digit.  Workgroups [0, bufblocks * nseg): the side buffers of the
// segments that did not fail (bufblocks = BSEL_NSH * bps: bps workgroups share one shard's
// region).  The remaining workgroups: the exact path over the frame for segments flagged
// `fail` -- they return at once when no segment is (the usual case).
__global__ __launch_bounds__(256) void k_sel_hist(sel_args a, int nseg, int bufblocks, int ny) {
    __shared__ uint32_t lh[2][SEL_BINS];
    const int nbuf = bufblocks * nseg;
    if ((int)blockIdx.x < nbuf) {
        const int sg = blockIdx.x / bufblocks, bx = blockIdx.x - sg * bufblocks;
        const bsel_seg* s = &a.b.seg[sg];
        if (s->fail) return;
        sel_digit d;
        if (!sel_digit_of(a.nbits[sg], a.pass, &d)) return;
        const int bps = bufblocks / BSEL_NSH, sh = bx / bps;
        const uint32_t count = min(a.b.shard[sg * BSEL_NSH + sh].nbuf, a.b.capS);
        const uint32_t per = (count + bps - 1) / bps;
        const uint32_t i0 = (bx % bps) * per, i1 = min(count, i0 + per);
        if (i0 >= i1) return;
        for (int i = threadIdx.x; i < 2 * SEL_BINS; i += blockDim.x) (&lh[0][0])[i] = 0;
        __syncthreads();
        const uint32_t pre0 = a.prefix[sg * 2], pre1 = a.prefix[sg * 2 + 1], klo = a.klo[sg];
        const float* v = bsel_region(a.b, sg, sh);
        const uint32_t span = ((i1 - i0 + 63u) / 64u) * 64u;
        for (uint32_t k = threadIdx.x; k < span; k += blockDim.x) {
            const bool in = i0 + k < i1;
            const uint32_t key = in ? f2key(v[i0 + k]) - klo : 0u;
            sel_hist_one(lh[0], lh[1], key, in, pre0, pre1, d);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * SEL_BINS; i += blockDim.x) {
            const uint32_t c = (&lh[0][0])[i];
            if (c) atomicAdd(&a.hist[(size_t)sg * 2 * SEL_BINS + i], c);
        }
        return;
    }
    if (*a.anyfail == 0) return;
    const int nfb = gridDim.x - nbuf;
    for (int pair = blockIdx.x - nbuf; pair < ny * a.SX; pair += nfb) {          // (row, segment column)
        const int Y = pair / a.SX, sx = pair - Y * a.SX;
        const int sg = (Y / a.ysz) * a.SX + sx;
        sel_digit d;
        if (!a.b.seg[sg].fail || !sel_digit_of(a.nbits[sg], a.pass, &d)) continue;   // workgroup-uniform
        for (int i = threadIdx.x; i < 2 * SEL_BINS; i += blockDim.x) (&lh[0][0])[i] = 0;
        __syncthreads();
        const uint32_t pre0 = a.prefix[sg * 2], pre1 = a.prefix[sg * 2 + 1], klo = a.klo[sg];
        const size_t row = (size_t)Y * a.nx + (size_t)sx * a.xsz;
        const int xend = ((a.xsz + 63) / 64) * 64;
        for (int x = threadIdx.x; x < xend; x += blockDim.x) {
            bool in = x < a.xsz;
            if (in && a.mask && (a.mask[row + x] & ~BBX_MASK_COSMIC)) in = false;
            if (in && !bsel_value_ok(a.data[row + x], a.b.seg[sg].wlo, a.b.seg[sg].whi, a.b.skip_zero)) in = false;
            const uint32_t key = in ? f2key(a.data[row + x]) - klo : 0u;
            sel_hist_one(lh[0], lh[1], key, in, pre0, pre1, d);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * SEL_BINS; i += blockDim.x) {
            const uint32_t c = (&lh[0][0])[i];
            if (c) atomicAdd(&a.hist[(size_t)sg * 2 * SEL_BINS + i], c);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(128) void k_sel_scan(sel_args a) {
    const int sg = blockIdx.x;
    sel_digit d;
    if (!sel_digit_of(a.nbits[sg], a.pass, &d)) return;
    const int q = threadIdx.x >> 6;
    const uint32_t pre0 = a.prefix[sg * 2], pre1 = a.prefix[sg * 2 + 1];
    uint32_t* h0 = a.hist + (size_t)sg * 2 * SEL_BINS;
    const uint32_t* h = (q == 1 && pre1 == pre0) ? h0 : h0 + q * SEL_BINS;
    const int nb = d.bits < 6 ? 64 : (1 << d.bits);           // wave_find_bin wants a multiple of 64 (upper bins are empty)
    int b = 0; unsigned long long rin = 0, total = 0;
    const bool found = wave_find_bin(h, nb, a.rank[sg * 2 + q], &b, &rin, &total);
    __syncthreads();                                         // both waves have read prefix/hist
    if (found) {
        a.rank[sg * 2 + q] = rin;
        const uint32_t pre = (q ? pre1 : pre0) | (((uint32_t)b) << d.shift);
        a.prefix[sg * 2 + q] = pre;
        if (d.shift == 0) a.b.seg[sg].result[q] = key2f(a.klo[sg] + pre);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * nb; k += 128) h0[(k / nb) * SEL_BINS + (k % nb)] = 0;
}

static int ws_layout(bbx_ctx* ctx, int nseg, uint32_t cap, bsel_seg** seg, bsel_shard** shard, float** samples, float** buf,
                     uint32_t** prefix, unsigned long long** rank, uint32_t** hist, uint32_t** klo, int** nbits,
                     int** anyfail = nullptr) {
    int rc;
    static_assert(sizeof(bsel_shard) == 64, "one cache line per shard");
    const size_t o_shard = 0;
    const size_t o_seg = o_shard + (size_t)BSEL_MAXSEG * BSEL_NSH * sizeof(bsel_shard);
    const size_t o_rank = o_seg + BSEL_MAXSEG * sizeof(bsel_seg);
    const size_t o_prefix = o_rank + BSEL_MAXSEG * 2 * sizeof(unsigned long long);
    const size_t o_klo = o_prefix + BSEL_MAXSEG * 2 * sizeof(uint32_t);
    const size_t o_nbits = o_klo + BSEL_MAXSEG * sizeof(uint32_t);
    const size_t o_anyfail = o_nbits + BSEL_MAXSEG * sizeof(int);
    const size_t o_hist = o_anyfail + 64;
    const size_t o_samples = o_hist + (size_t)BSEL_MAXSEG * 2 * SEL_BINS * 4;
    const size_t o_buf = o_samples + (size_t)nseg * BSEL_S * 4;
    const size_t total = o_buf + (size_t)nseg * cap * 4;
    char* ws = (char*)bbx_ws(ctx, WS_SEL, total, &rc);
    if (rc) return rc;
    *shard = (bsel_shard*)(ws + o_shard);
    *seg = (bsel_seg*)(ws + o_seg); *rank = (unsigned long long*)(ws + o_rank);
    *prefix = (uint32_t*)(ws + o_prefix); *hist = (uint32_t*)(ws + o_hist);
    *klo = (uint32_t*)(ws + o_klo); *nbits = (int*)(ws + o_nbits);
    if (anyfail) *anyfail = (int*)(ws + o_anyfail);
    *samples = (float*)(ws + o_samples); *buf = (float*)(ws + o_buf);
    return BBX_OK;
}

int bbx_bsel_prepare(bbx_ctx* ctx, const float* d_data, const uint8_t* d_mask, int ny, int nx, int ysz, int xsz,
                     bsel_dev* out, hipStream_t s, int stride) {
    if (ysz < 1 || xsz < 1 || ny % ysz || nx % xsz) return BBX_ERR_ARG;
    if (stride == 0) stride = nx;
    if (stride < nx) return BBX_ERR_ARG;
    const int SX = nx / xsz, nseg = SX * (ny / ysz);
    if (nseg > BSEL_MAXSEG) return BBX_ERR_ARG;
    // side buffer: 1/8 of the segment (the bracket holds ~5 %), at least 64k values
    const size_t segpix = (size_t)ysz * xsz;
    // split over BSEL_NSH shards (workgroups pick shards round-robin; 2x headroom per shard)
    uint32_t capS = (uint32_t)(((segpix / 8 + 65536) / BSEL_NSH + 1023) / 1024 * 1024);
    // ... and room for one feeding workgroup whose strip lies entirely inside the bracket (an already
    // edge-filled frame: whole rows equal to the median) on top of the shard's usual share
    const uint32_t strip = (uint32_t)((size_t)16 * xsz + capS / 2 + 1023) / 1024 * 1024;
    if (strip > capS) capS = strip;
    const uint32_t cap = capS * BSEL_NSH;
    bsel_seg* seg; bsel_shard* shard; float *samples, *buf; uint32_t *prefix, *hist, *klo; int* nbits; unsigned long long* rank;
    int* anyfail;
    int rc = ws_layout(ctx, nseg, cap, &seg, &shard, &samples, &buf, &prefix, &rank, &hist, &klo, &nbits, &anyfail);
    if (rc) return rc;
    // (also zeroes the digit histograms: every scan leaves them zero again, so once per select is enough)
    hipLaunchKernelGGL(k_bsel_init, dim3((nseg * BSEL_NSH + 255) / 256), dim3(256), 0, s, seg, shard, nseg, hist,
                       nseg * 2 * SEL_BINS, anyfail);
    hipLaunchKernelGGL(k_bsel_sample, dim3(BSEL_S / 256, nseg), dim3(256), 0, s, d_data, d_mask, stride, ysz, xsz, SX, seg, samples, 0);
    hipLaunchKernelGGL(k_bsel_bracket, dim3(nseg), dim3(1024), 0, s, seg, samples);
    BBX_LAUNCH_CHECK();
    out->seg = seg; out->shard = shard; out->buf = buf; out->cap = cap; out->capS = capS; out->ysz = ysz; out->xsz = xsz; out->SX = SX; out->stride = stride; out->skip_zero = 0;
    return BBX_OK;
}

int bbx_bsel_finish(bbx_ctx* ctx, const bsel_dev& b, const float* d_data, const uint8_t* d_mask, int ny, int nx,
                    hipStream_t s) {
    const int SX = b.SX, nseg = SX * (ny / b.ysz);
    bsel_seg* seg; bsel_shard* shard; float *samples, *buf; uint32_t *prefix, *hist, *klo; int* nbits; unsigned long long* rank;
    int* anyfail;
    int rc = ws_layout(ctx, nseg, b.cap, &seg, &shard, &samples, &buf, &prefix, &rank, &hist, &klo, &nbits, &anyfail);
    if (rc) return rc;
    sel_args a;
    a.data = d_data; a.mask = d_mask; a.nx = b.stride; a.ysz = b.ysz; a.xsz = b.xsz; a.SX = SX;
    a.b = b; a.prefix = prefix; a.rank = rank; a.hist = hist; a.klo = klo; a.nbits = nbits; a.pass = 0; a.anyfail = anyfail;
    hipLaunchKernelGGL(k_sel_plan, dim3(1), dim3(64), 0, s, a, nseg);
    const int bufblocks = BSEL_NSH * (nseg == 1 ? 8 : 1);
    const int framebl = std::min(ny * SX, 4096);               // exact-path workgroups (idle unless a segment failed)
    // up to 3 digits of <= 11 bits; segments whose bracket spans fewer bits skip the later passes
    for (int p = 0; p < 3; p++) {
        a.pass = p;
        hipLaunchKernelGGL(k_sel_hist, dim3(bufblocks * nseg + framebl), dim3(256), 0, s, a, nseg, bufblocks, ny);
        hipLaunchKernelGGL(k_sel_scan, dim3(nseg), dim3(128), 0, s, a);
    }
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

// ---------------------------------------------------------------------------------
// edge fill
// ---------------------------------------------------------------------------------
// standalone feeder (used when no other kernel streams the frame after its last change)
__global__ __launch_bounds__(256) void k_bsel_feed_frame(const float* __restrict__ data, const uint8_t* __restrict__ mask,
                                                         int ny, int nx, bsel_dev b) {
    __shared__ bsel_lds L;
    // block = (group of 4 rows, segment column sx)
    const int sx = blockIdx.y;
    bsel_lds_init(L);
    bsel_acc acc = {0, 0};
    const int xend = ((b.xsz + 63) / 64) * 64;
    for (int Y = blockIdx.x * 4; Y < min(ny, blockIdx.x * 4 + 4); Y++) {
        const int sg = (Y / b.ysz) * b.SX + sx;
        const int sg_next = ((Y + 1) / b.ysz) * b.SX + sx;
        const float lo = b.seg[sg].lo, hi = b.seg[sg].hi;
        const double wlo = b.seg[sg].wlo, whi = b.seg[sg].whi;
        const size_t row = (size_t)Y * nx + (size_t)sx * b.xsz;
        for (int x0 = 0; x0 < xend; x0 += 1024) {
            for (int x = x0 + threadIdx.x; x < min(xend, x0 + 1024); x += blockDim.x) {
                const bool in = x < b.xsz;
                bool valid = in;
                if (in && mask && (mask[row + x] & ~BBX_MASK_COSMIC)) valid = false;
                const float v = in ? data[row + x] : 0.f;
                if (!bsel_value_ok(v, wlo, whi, b.skip_zero)) valid = false;  // NaN (np.nanmedian), clip window, mask_value
                bsel_feed(L, lo, hi, v, valid, acc);
            }
            // at most 1024 appends until the next drain point; force at the end of a
            // segment / of the block's rows
            const bool lastchunk = x0 + 1024 >= xend;
            const bool force = lastchunk && (Y + 1 >= min(ny, (int)blockIdx.x * 4 + 4) || sg_next != sg);
            bsel_drain(b, sg, L, 1024, force);
        }
        if (sg_next != sg || Y + 1 >= min(ny, (int)blockIdx.x * 4 + 4)) bsel_flush(b, sg, acc);
    }
}

// fast feeder for 4-pixel-aligned segments: block = (segment column, strip of FEED_ROWS rows inside
// one segment); a thread walks float4 groups, stages its in-bracket values in private LDS slots
// (no atomics in the loop) and the block compacts them with one global atomic at the end.
#define FEED_ROWS 16
#define FEEDQ 24
__global__ __launch_bounds__(256) void k_bsel_feed_v4(const float* __restrict__ data, const uint8_t* __restrict__ mask,
                                                      int nx, bsel_dev b) {
    __shared__ float stage[FEEDQ * 256];
    __shared__ unsigned wsum[4], gbase;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int sx = blockIdx.y, Y0 = blockIdx.x * FEED_ROWS;
    const int sg = (Y0 / b.ysz) * b.SX + sx;
    const float lo = b.seg[sg].lo, hi = b.seg[sg].hi;
    const double wlo = b.seg[sg].wlo, whi = b.seg[sg].whi;
    const unsigned sh = bsel_my_shard();
    float* reg = bsel_region(b, sg, sh);
    const int ng = b.xsz / 4;
    unsigned nst = 0, nvalid = 0, nbelow = 0;
    // one flat loop over the strip's (row, group) pairs: with a loop per row, a 330-group row
    // (1320 px channels) leaves 71 % of the lanes idle in its second round
    for (int idx = tid; idx < FEED_ROWS * ng; idx += 256) {
        const int r = idx / ng, g = idx - r * ng;
        const size_t row = (size_t)(Y0 + r) * nx + (size_t)sx * b.xsz;
        {
            const float4 f = *(const float4*)(data + row + 4 * g);
            uchar4 m = make_uchar4(0, 0, 0, 0);
            if (mask) m = *(const uchar4*)(mask + row + 4 * g);
            const float v[4] = {f.x, f.y, f.z, f.w};
            const uint8_t mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bool valid = !(mm[q] & ~BBX_MASK_COSMIC) && bsel_value_ok(v[q], wlo, whi, b.skip_zero);
                nvalid += valid ? 1u : 0u;
                nbelow += (valid && v[q] < lo) ? 1u : 0u;
                if (valid && v[q] >= lo && v[q] <= hi) {
                    if (nst < FEEDQ) { stage[nst * 256 + tid] = v[q]; nst++; }
                    else { const unsigned k = bsel_reserve(b, sg, sh, 1u); if (k < b.capS) reg[k] = v[q]; }
                }
            }
        }
    }
    // block compaction
    unsigned incl = nst;
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    unsigned off = 0, tot = 0;
    for (int w = 0; w < 4; w++) { if (w < wid) off += wsum[w]; tot += wsum[w]; }
    if (tid == 0) gbase = tot ? bsel_reserve(b, sg, sh, tot) : 0u;
    __syncthreads();
    const unsigned base = gbase + off + incl - nst;
    for (unsigned k = 0; k < nst; k++) { const unsigned pos = base + k; if (pos < b.capS) reg[pos] = stage[k * 256 + tid]; }
    bsel_acc acc = {nvalid, nbelow};
    bsel_flush(b, sg, acc);
}

int bbx_bsel_feed_frame(const float* d_data, const uint8_t* d_mask, int ny, int nx, const bsel_dev& b, hipStream_t s) {
    const bool vec = (b.xsz % 4 == 0) && (b.xsz <= 2048) && (b.ysz % FEED_ROWS == 0) && (((uintptr_t)d_data) % 16 == 0) && (b.stride % 4 == 0) &&
                     (!d_mask || ((uintptr_t)d_mask) % 4 == 0);
    (void)nx;
    if (vec) hipLaunchKernelGGL(k_bsel_feed_v4, dim3(ny / FEED_ROWS, b.SX), dim3(256), 0, s, d_data, d_mask, b.stride, b);
    else hipLaunchKernelGGL(k_bsel_feed_frame, dim3((ny + 3) / 4, b.SX), dim3(256), 0, s, d_data, d_mask, ny, b.stride, b);
    return BBX_OK;
}

// np.median of a float32 array: odd n -> middle element, even n -> float32 mean of the two
__global__ void k_chan_median(const bsel_seg* __restrict__ seg, float* __restrict__ med) {
    const int c = threadIdx.x;
    if (c < 16) {
        const float lo = seg[c].result[0], hi = seg[c].result[1];
        med[c] = (seg[c].n & 1) ? lo : (lo + hi) * 0.5f;
    }
}

__global__ __launch_bounds__(256) void k_edge_fill(float* data, const uint8_t* __restrict__ mask, bbx_dims d,
                                                   const float* __restrict__ med) {
    const size_t npix = (size_t)d.ny * d.nx;
    // the mask is read sixteen pixels at a time (edge pixels are a thin frame: nearly every group
    // is skipped after one test); groups never straddle the end because the tail is done apart
    const size_t n16 = (((uintptr_t)mask) % 16 == 0) ? npix / 16 : 0;
    const uint4* m16 = (const uint4*)mask;
    const uint32_t E4 = 0x01010101u * BBX_MASK_EDGE;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n16; g += (size_t)gridDim.x * blockDim.x) {
        const uint4 m = m16[g];
        if (((m.x | m.y | m.z | m.w) & E4) == 0) continue;
        const uint32_t w[4] = {m.x, m.y, m.z, m.w};
        for (int k = 0; k < 16; k++) {
            if (((w[k >> 2] >> (8 * (k & 3))) & BBX_MASK_EDGE) == BBX_MASK_EDGE) {
                const size_t i = g * 16 + k;
                const int Y = (int)(i / d.nx), X = (int)(i - (size_t)Y * d.nx);
                data[i] = med[(Y / d.ysz) * 8 + X / d.xsz];
            }
        }
    }
    for (size_t i = n16 * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
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
    bsel_dev b;
    rc = bbx_bsel_prepare(ctx, d_data, nullptr, d.ny, d.nx, d.ysz, d.xsz, &b, s); if (rc) return rc;
    bbx_bsel_feed_frame(d_data, nullptr, d.ny, d.nx, b, s);
    rc = bbx_bsel_finish(ctx, b, d_data, nullptr, d.ny, d.nx, s); if (rc) return rc;
    hipLaunchKernelGGL(k_chan_median, dim3(1), dim3(64), 0, s, b.seg, d_chan_median);
    hipLaunchKernelGGL(k_edge_fill, dim3(2048), dim3(256), 0, s, d_data, d_mask, d, d_chan_median);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}


// ---------------------------------------------------------------------------------
// a14 / a8 statistics: per-segment median, mean, sigma and "lower-half" sigma
// (get_flatstats, blackbox.py:3661-3820)
// ---------------------------------------------------------------------------------
#define RS_ROWS 8
// block (chunk of RS_ROWS rows, segment column): float64 sums of the valid pixels of one
// segment: n, sum x, sum x^2, and for x <= median: n_low, sum (x - median)^2
__global__ __launch_bounds__(256) void k_seg_moments(const float* __restrict__ data, const uint8_t* __restrict__ mask,
                                                     int stride, bsel_dev b, int nchunk, double* __restrict__ partial) {
    const int sx = blockIdx.y;
    const int sy = blockIdx.x / nchunk, cin = blockIdx.x - sy * nchunk;      // strip [cin] of segment row [sy]
    const int Y0 = sy * b.ysz + cin * RS_ROWS, nrow = min(RS_ROWS, b.ysz - cin * RS_ROWS);
    const int sg = sy * b.SX + sx;
    const float lo = b.seg[sg].result[0], hi = b.seg[sg].result[1];
    const float med = (b.seg[sg].n & 1) ? lo : (lo + hi) * 0.5f;
    const double wlo = b.seg[sg].wlo, whi = b.seg[sg].whi;
    double s[5] = {0, 0, 0, 0, 0};
    for (int r = 0; r < nrow; r++) {
        const size_t row = (size_t)(Y0 + r) * stride + (size_t)sx * b.xsz;
        for (int x = threadIdx.x; x < b.xsz; x += 256) {
            const float v = data[row + x];
            const bool valid = bsel_value_ok(v, wlo, whi, b.skip_zero) && !(mask && (mask[row + x] & ~BBX_MASK_COSMIC));
            if (valid) {
                const double d = (double)v;
                s[0] += 1.0; s[1] += d; s[2] += d * d;
                if (v <= med) { const double e = (double)(v - med); s[3] += 1.0; s[4] += e * e; }
            }
        }
    }
    __shared__ double sh[5][4];
#pragma unroll
    for (int k = 0; k < 5; k++) { const double w = wave_sum_f64(s[k]); if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = w; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int k = threadIdx.x;
        // partial[sg][strip within the segment][5]
        partial[((size_t)sg * nchunk + cin) * 5 + k] = (sh[k][0] + sh[k][1]) + (sh[k][2] + sh[k][3]);
    }
}

// one wave per segment: fold the partials in fixed order -> out[sg][8] =
// {n, median, mean, sigma (ddof 0), n_low, sigma_low (ddof 1 about the median), 0, 0}
__global__ __launch_bounds__(64) void k_seg_finalize(bsel_dev b, int nchunk, const double* __restrict__ partial,
                                                     double* __restrict__ out, double clip_sigma) {
    const int sg = blockIdx.x, lane = threadIdx.x;
    double s[5] = {0, 0, 0, 0, 0};
    for (int c = lane; c < nchunk; c += 64)
        for (int k = 0; k < 5; k++) s[k] += partial[((size_t)sg * nchunk + c) * 5 + k];
    for (int k = 0; k < 5; k++) s[k] = wave_sum_f64(s[k]);
    if (lane == 0) {
        const float lo = b.seg[sg].result[0], hi = b.seg[sg].result[1];
        const float med = (b.seg[sg].n & 1) ? lo : (lo + hi) * 0.5f;
        const double n = s[0], mean = s[1] / n;
        double var = s[2] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        double* o = out + (size_t)sg * 8;
        o[0] = n; o[1] = n > 0 ? (double)med : __longlong_as_double(0x7ff8000000000000LL);
        o[2] = mean; o[3] = sqrt(var); o[4] = s[3]; o[5] = sqrt(s[4] / (s[3] - 1.0)); o[6] = 0.0; o[7] = 0.0;
        if (clip_sigma > 0.0 && n > 0) {
            // sigma clipping about the median (astropy sigma_clip, cenfunc='median'): the next
            // round's survivors are this round's survivors inside median +- sigma * std
            const double lo2 = (double)med - clip_sigma * o[3], hi2 = (double)med + clip_sigma * o[3];
            if (lo2 > b.seg[sg].wlo) b.seg[sg].wlo = lo2;
            if (hi2 < b.seg[sg].whi) b.seg[sg].whi = hi2;
        }
    }
}

extern "C" int bbx_rect_stats(bbx_ctx* ctx, int ny, int nx, int stride, const float* d_data, const uint8_t* d_mask,
                              int ysz, int xsz, double* d_out, void* stream) {
    if (!ctx || !d_data || !d_out || ny < 1 || nx < 1 || stride < nx || ysz < 1 || xsz < 1) return BBX_ERR_ARG;
    if (ny % ysz || nx % xsz) return BBX_ERR_ARG;
    const int SX = nx / xsz, nseg = SX * (ny / ysz);
    if (nseg > BSEL_MAXSEG) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    bsel_dev b;
    int rc = bbx_bsel_prepare(ctx, d_data, d_mask, ny, nx, ysz, xsz, &b, s, stride); if (rc) return rc;
    bbx_bsel_feed_frame(d_data, d_mask, ny, nx, b, s);
    rc = bbx_bsel_finish(ctx, b, d_data, d_mask, ny, nx, s); if (rc) return rc;
    const int nchunk = (ysz + RS_ROWS - 1) / RS_ROWS;
    double* partial = (double*)bbx_ws(ctx, WS_HIST, (size_t)nseg * nchunk * 5 * sizeof(double), &rc); if (rc) return rc;
    hipLaunchKernelGGL(k_seg_moments, dim3((ny / ysz) * nchunk, SX), dim3(256), 0, s, d_data, d_mask, stride, b, nchunk, partial);
    hipLaunchKernelGGL(k_seg_finalize, dim3(nseg), dim3(64), 0, s, b, nchunk, partial, d_out, 0.0);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

// counters back to zero for another select over the same segments; the clip window stays
__global__ __launch_bounds__(256) void k_bsel_reset(bsel_seg* seg, bsel_shard* shard, int nseg, int* anyfail) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *anyfail = 0;
    if (i < nseg * BSEL_NSH) { bsel_shard z; memset(&z, 0, sizeof(z)); shard[i] = z; }
    if (i < nseg) { seg[i].lo = 0.f; seg[i].hi = 0.f; seg[i].nsample = 0; seg[i].nbuf = 0; seg[i].below = 0; seg[i].n = 0; seg[i].fail = 0; }
}

// astropy.stats.sigma_clipped_stats(x, sigma, maxiters, mask_value=0) per segment (cenfunc
// median, stdfunc std): [maxiters] rounds of (median, std of the survivors -> clip), then
// the statistics of the final survivors.  All rounds are enqueued without a host round trip.
extern "C" int bbx_rect_clipped_stats(bbx_ctx* ctx, int ny, int nx, int stride, const float* d_data, const uint8_t* d_mask,
                                      int ysz, int xsz, double sigma, int maxiters, int skip_zero, double* d_out,
                                      void* stream) {
    if (!ctx || !d_data || !d_out || ny < 1 || nx < 1 || stride < nx || ysz < 1 || xsz < 1) return BBX_ERR_ARG;
    if (ny % ysz || nx % xsz || maxiters < 0 || maxiters > 20 || !(sigma > 0.0)) return BBX_ERR_ARG;
    const int SX = nx / xsz, nseg = SX * (ny / ysz);
    if (nseg > BSEL_MAXSEG) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    bsel_dev b;
    int rc = bbx_bsel_prepare(ctx, d_data, d_mask, ny, nx, ysz, xsz, &b, s, stride); if (rc) return rc;
    b.skip_zero = skip_zero ? 1 : 0;
    const int nchunk = (ysz + RS_ROWS - 1) / RS_ROWS;
    double* partial = (double*)bbx_ws(ctx, WS_HIST, (size_t)nseg * nchunk * 5 * sizeof(double), &rc); if (rc) return rc;
    bsel_seg* seg; bsel_shard* shard; float *samples, *buf; uint32_t *prefix, *hist, *klo; int* nbits; unsigned long long* rank;
    int* anyfail;
    rc = ws_layout(ctx, nseg, b.cap, &seg, &shard, &samples, &buf, &prefix, &rank, &hist, &klo, &nbits, &anyfail); if (rc) return rc;
    for (int it = 0; it <= maxiters; it++) {
        // (prepare sampled without the mask value rule; sample again with the current window)
        hipLaunchKernelGGL(k_bsel_reset, dim3((nseg * BSEL_NSH + 255) / 256), dim3(256), 0, s, seg, shard, nseg, anyfail);
        hipLaunchKernelGGL(k_bsel_sample, dim3(BSEL_S / 256, nseg), dim3(256), 0, s, d_data, d_mask, stride, ysz, xsz, SX, seg,
                           samples, b.skip_zero);
        hipLaunchKernelGGL(k_bsel_bracket, dim3(nseg), dim3(1024), 0, s, seg, samples);
        bbx_bsel_feed_frame(d_data, d_mask, ny, nx, b, s);
        rc = bbx_bsel_finish(ctx, b, d_data, d_mask, ny, nx, s); if (rc) return rc;
        hipLaunchKernelGGL(k_seg_moments, dim3((ny / ysz) * nchunk, SX), dim3(256), 0, s, d_data, d_mask, stride, b, nchunk, partial);
        hipLaunchKernelGGL(k_seg_finalize, dim3(nseg), dim3(64), 0, s, b, nchunk, partial, d_out, it < maxiters ? sigma : 0.0);
    }
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

// in-place scaling of a rectangle: data[y][x] /= f or *= f (float32, IEEE)
__global__ __launch_bounds__(256) void k_rect_scale(float* data, int ny, int nx, int stride, float f, int divide) {
    const size_t total = (size_t)ny * nx;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t y = t / nx, x = t - y * nx;
        float* p = data + y * stride + x;
        *p = divide ? (*p / f) : (*p * f);
    }
}

extern "C" int bbx_rect_scale(bbx_ctx* ctx, int ny, int nx, int stride, float* d_data, float factor, int divide, void* stream) {
    if (!ctx || !d_data || ny < 1 || nx < 1 || stride < nx) return BBX_ERR_ARG;
    hipLaunchKernelGGL(k_rect_scale, dim3(2048), dim3(256), 0, (hipStream_t)stream, d_data, ny, nx, stride, factor, divide);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}
