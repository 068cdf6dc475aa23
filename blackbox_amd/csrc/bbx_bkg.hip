// bbx_bkg.hip -- 2-D background mesh (zogy.get_back / mini2back as used by BlackBOX:
// buildref.py:2398-2405, 2480-2495).  [EXT: parity unpinned, the
// conventions are the ones written down in oracle/zogy_core.py]
//
//   k_bkg_boxstats_fast  one wave per bkg_boxsize x bkg_boxsize box: the astropy-style clip (centre = median, std about
//                    the mean, 3 sigma, <= 5 iterations) of the usable pixels (mask == 0, objmask == 0, value != 0) from a
//                    sorted bracket around the median + the list of the wing pixels (round 4; vector-issue-bound);
//   k_bkg_boxstats(_list)  the same from a full sort of the box in registers (rounds 2-3): the boxes the bracket kernel
//                    lists, or all of them (BBX_OPT_BKG_FULL_SORT).  One read of the frame + mask: 5N bytes.
//   k_mini_fill_filter  NaN boxes <- nan-median of 3x3 neighbours (repeated), then a 3x3
//                    median filter with replicated edges, on the 176x176 mini image.
//   k_spline_zoom    scipy.ndimage.zoom(order=3, mode='nearest') evaluation: the cubic
//                    B-spline coefficients (tiny; k_spf_axis0/1 below: scipy's prefilter, same bits) are combined with
//                    per-row / per-column tap weights in float64; optionally fused with the
//                    subtraction from the frame (read 4N + write 4N).
#include "bbx_common.h"
#include "bbx_mednet.h"

#define BOX_NS 4096                 // 64 lanes x 64 registers: boxes up to 64 x 64 pixels
#define BOX_PAD 0xffffffffu         // key of a slot without usable pixel (above the key of +inf)

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t o;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
__device__ __forceinline__ float box_key_value(uint32_t u) { return __uint_as_float((u >> 31) ? (u ^ 0x80000000u) : ~u); }

// compare-exchange stage between registers r and r^J of every lane, directions fixed by the
// element index (levels of 2..32 elements of the bitonic network)
template <int K, int J> __device__ __forceinline__ void box_stage_static(uint32_t (&k)[64]) {
#pragma unroll
    for (int r = 0; r < 64; r++) {
        const int q = r ^ J;
        if (q > r) {
            const uint32_t a = k[r], b = k[q];
            const uint32_t lo = min(a, b), hi = max(a, b);
            if ((r & K) == 0) { k[r] = lo; k[q] = hi; } else { k[r] = hi; k[q] = lo; }
        }
    }
}
// the same with the direction of the lane: clo = 0 sorts ascending (med3(a,b,0) = min), ~0 descending
template <int J> __device__ __forceinline__ void box_stage_lane(uint32_t (&k)[64], uint32_t clo) {
    const uint32_t chi = ~clo;
#pragma unroll
    for (int r = 0; r < 64; r++) {
        const int q = r ^ J;
        if (q > r) {
            const uint32_t a = k[r], b = k[q];
            k[r] = umed3(a, b, clo); k[q] = umed3(a, b, chi);
        }
    }
}

// index of sorted element i in LDS: one word of padding per lane's 64 so that the lanes'
// stores fall in different banks
__device__ __forceinline__ int box_slot(int i) { return i + (i >> 6); }

// number of entries of the ascending v[a, b) that are < lim (UPPER = false) or <= lim (true):
// a 64-way search over the blocks of 64, then inside the block that holds the boundary
template <bool UPPER> __device__ __forceinline__ int box_count_below(const float* v, int a, int b, double lim, int lane) {
    const int nblk = (b - a + 63) >> 6;                   // <= 64
    bool below = false;
    if (lane < nblk) { const double x = (double)v[box_slot(a + lane * 64)]; below = UPPER ? (x <= lim) : (x < lim); }
    const int nb = __popcll(__ballot(below));             // blocks whose first entry is below: the boundary is in block nb-1
    if (nb == 0) return 0;
    const int base = a + (nb - 1) * 64;
    bool in = false;
    if (base + lane < b) { const double x = (double)v[box_slot(base + lane)]; in = UPPER ? (x <= lim) : (x < lim); }
    return (nb - 1) * 64 + __popcll(__ballot(in));
}


__device__ __forceinline__ void box_full_sort(const float* __restrict__ data, const uint8_t* __restrict__ mask,
                                              const uint8_t* __restrict__ objmask, int nx, int box,
                                              int nbx, int ibox, float limfrac, float* mini_med, float* mini_std) {
    // one wave per box.  The usable pixels become order-preserving integer keys, 64 per lane
    // in registers, and are sorted once by a bitonic network: stages between registers of a
    // lane are v_min/v_max (or v_med3 against a per-lane 0 / ~0 when the direction depends on
    // the lane), the 21 stages between lanes fetch the partner through ds_bpermute.  The sorted
    // pixels go to LDS; the clip iterations then only move the two ends of the index range.
    __shared__ float v[BOX_NS + 64];
    const int lane = threadIdx.x;
    const int by = ibox / nbx, bx = ibox - by * nbx;
    const int npx = box * box;
    const int q64 = 64 / box, r64 = 64 - q64 * box;
    uint32_t k[64];
    int cnt = 0;
    {
        // element c*64 + lane: consecutive lanes read along a box row.  All loads of the box are
        // issued before the first use and without branches (slots beyond the box re-read its
        // last pixel and are discarded): the kernel runs few waves per CU (LDS) and needs the
        // memory parallelism
        int yy = lane / box, xx = lane - yy * box;
        const size_t o0 = (size_t)(by * box) * nx + (size_t)bx * box;
        const float* pd = data + o0; const uint8_t* pm = mask + o0; const uint8_t* po = objmask + o0;
        const uint32_t omax = (uint32_t)((box - 1) * nx + box - 1);
        unsigned mk[64];
#pragma unroll
        for (int c = 0; c < 64; c++) {
            const uint32_t o = min((uint32_t)(yy * nx + xx), omax);    // < 64 nx: the host checks that it fits
            k[c] = __float_as_uint(pd[o]);
            mk[c] = pm[o];
            if (objmask) mk[c] |= po[o];
            yy += q64; xx += r64;
            if (xx >= box) { xx -= box; yy++; }
        }
#pragma unroll
        for (int c = 0; c < 64; c++) {
            const uint32_t u = k[c];
            const float d = __uint_as_float(u);
            const bool ok = (c * 64 + lane < npx) & (mk[c] == 0) & (d != 0.f) & (d == d);
            k[c] = ok ? (u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u)) : BOX_PAD;
            cnt += ok;
        }
    }
    const int n0 = wave_sum_i32(cnt);
    if ((float)n0 < limfrac * (float)npx || n0 == 0) {
        if (lane == 0) { const float nanv = __uint_as_float(0x7fc00000u); mini_med[ibox] = nanv; mini_std[ibox] = nanv; }
        return;
    }
    // sorted position of element (lane, r) is lane*64 + r
    box_stage_static<2, 1>(k);
    box_stage_static<4, 2>(k); box_stage_static<4, 1>(k);
    box_stage_static<8, 4>(k); box_stage_static<8, 2>(k); box_stage_static<8, 1>(k);
    box_stage_static<16, 8>(k); box_stage_static<16, 4>(k); box_stage_static<16, 2>(k); box_stage_static<16, 1>(k);
    box_stage_static<32, 16>(k); box_stage_static<32, 8>(k); box_stage_static<32, 4>(k); box_stage_static<32, 2>(k);
    box_stage_static<32, 1>(k);
    for (int ll = 0; ll <= 6; ll++) {                     // runs of 64 << ll elements
        const bool asc = (lane & (1 << ll)) == 0;         // ll = 6: one ascending run
        for (int m = (1 << ll) >> 1; m > 0; m >>= 1) {
            const uint32_t c = (((lane & m) == 0) == asc) ? 0u : 0xffffffffu;   // keep the smaller / the larger
#pragma unroll
            for (int r = 0; r < 64; r++) k[r] = umed3(k[r], (uint32_t)__shfl_xor((int)k[r], m), c);
        }
        const uint32_t clo = asc ? 0u : 0xffffffffu;
        box_stage_lane<32>(k, clo); box_stage_lane<16>(k, clo); box_stage_lane<8>(k, clo);
        box_stage_lane<4>(k, clo); box_stage_lane<2>(k, clo); box_stage_lane<1>(k, clo);
    }
#pragma unroll
    for (int r = 0; r < 64; r++) v[lane * 65 + r] = box_key_value(k[r]);
    __syncthreads();
    // sums about a pivot close to the final mean (the median of all usable pixels); the clip
    // iterations subtract what they remove
    const double piv = (double)v[box_slot(n0 >> 1)];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < 64; r++) {
        const double t = (k[r] != BOX_PAD) ? (double)box_key_value(k[r]) - piv : 0.0;
        s1 += t; s2 += t * t;
    }
    s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
    int a = 0, b = n0;
    double sd = 0.0, med = 0.0;
    for (int it = 0; it <= 5; it++) {
        const int n = b - a;
        const double m1 = s1 / (double)n;
        const double var = s2 / (double)n - m1 * m1;      // std about the mean, ddof 0
        sd = sqrt(var > 0.0 ? var : 0.0);
        med = (n & 1) ? (double)v[box_slot(a + n / 2)] : ((double)v[box_slot(a + n / 2 - 1)] + (double)v[box_slot(a + n / 2)]) * 0.5;
        if (it == 5) break;                               // statistics of the survivors after 5 clips
        const double lo = med - 3.0 * sd, hi = med + 3.0 * sd;
        const int tlo = box_count_below<false>(v, a, b, lo, lane);
        const int thi = n - box_count_below<true>(v, a, b, hi, lane);
        if (tlo == 0 && thi == 0) break;                  // nothing clipped: these are the final statistics
        double d1 = 0.0, d2 = 0.0;
        for (int i = a + lane; i < a + tlo; i += 64) { const double t = (double)v[box_slot(i)] - piv; d1 += t; d2 += t * t; }
        for (int i = b - thi + lane; i < b; i += 64) { const double t = (double)v[box_slot(i)] - piv; d1 += t; d2 += t * t; }
        s1 -= wave_sum_f64(d1); s2 -= wave_sum_f64(d2);
        a += tlo; b -= thi;
        if (b - a <= 0) break;
    }
    if (lane == 0) {
        if (b - a <= 0) { const float nanv = __uint_as_float(0x7fc00000u); mini_med[ibox] = nanv; mini_std[ibox] = nanv; }
        else { mini_med[ibox] = (float)med; mini_std[ibox] = (float)sd; }
    }
}

// all boxes, one workgroup each (BBX_OPT_BKG_FULL_SORT) ...
__global__ __launch_bounds__(64) void k_bkg_boxstats(const float* __restrict__ data, const uint8_t* __restrict__ mask,
                                                     const uint8_t* __restrict__ objmask, int ny, int nx, int box,
                                                     int nbx, int nboxes, float limfrac, float* mini_med, float* mini_std) {
    // workgroups go round-robin over the 8 XCDs: give each XCD a contiguous run of boxes, so that
    // the cache lines which neighbouring boxes share are fetched into one L2 only
    const int per = (nboxes + 7) >> 3;
    const int ibox = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (ibox < nboxes) box_full_sort(data, mask, objmask, nx, box, nbx, ibox, limfrac, mini_med, mini_std);
}
// ... or the boxes k_bkg_boxstats_fast has listed (half a percent of a frame's).  The list counters alternate between
// calls: this launch clears the one the next call will fill, so no call needs a memset.
__global__ __launch_bounds__(64) void k_bkg_boxstats_list(const float* __restrict__ data, const uint8_t* __restrict__ mask,
                                                          const uint8_t* __restrict__ objmask, int ny, int nx, int box,
                                                          int nbx, int nboxes, float limfrac, float* mini_med, float* mini_std,
                                                          const int32_t* __restrict__ nlist, const int32_t* __restrict__ list, int32_t* nlist_next) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *nlist_next = 0;
    const int n = min(*nlist, nboxes);
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        box_full_sort(data, mask, objmask, nx, box, nbx, list[t], limfrac, mini_med, mini_std);
        __syncthreads();
    }
}

// ---- the same statistics without sorting the box ---------------------------------------------------------------------
// The clip only ever needs (i) the middle order statistic(s) of the survivors and (ii) the pixels it removes, which sit in
// the wings.  So: 512 of the box's slots are sorted as a sample (8 registers per lane); its order statistics give a
// bracket [lo, hi] around the median (sample ranks 0.44 .. 0.56) and two wing limits (0.05 / 0.85).  Two passes over the 64
// registers, neither with a step that waits for another lane: the first counts per lane the keys of the bracket, of the
// wings and below the bracket and sums the pixels between the wings about a pivot; after one scan of the lane counts the
// second writes the bracket (<= 1024 keys) and the wings (<= 1024) into LDS.  The bracket alone is sorted (8 or 16 registers
// per lane).  A clip round then reads its median(s) from the sorted bracket by rank and scans the wing list for what it
// removes.  Whenever an assumption fails -- bracket or wings overflow (ties, constant boxes), a median outside the bracket,
// a clip limit inside the unlisted middle, too few samples -- the box is listed and k_bkg_boxstats_list sorts it in full:
// the medians are the same order statistics whichever path ran.
#define BOXF_NB 1024
#ifndef BOXF_QLO
#define BOXF_QLO 0.44f            // sample ranks of the bracket's ends: 12 % of the keys, mostly <= 512 -> the 8-register sort;
#define BOXF_QHI 0.56f            // (0.42 / 0.58: 16 boxes of a frame's 30 976 fall back instead of 209, but 9 % slower; 0.45 / 0.55: 676)
#endif
#define BOXF_NT 1024
template <int NR, int K, int J> __device__ __forceinline__ void bs_static(uint32_t (&k)[NR]) {
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int q = r ^ J;
        if (q > r) {
            const uint32_t a = k[r], b = k[q];
            const uint32_t lo = min(a, b), hi = max(a, b);
            if ((r & K) == 0) { k[r] = lo; k[q] = hi; } else { k[r] = hi; k[q] = lo; }
        }
    }
}
template <int NR, int J> __device__ __forceinline__ void bs_lane(uint32_t (&k)[NR], uint32_t clo) {
    const uint32_t chi = ~clo;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int q = r ^ J;
        if (q > r) {
            const uint32_t a = k[r], b = k[q];
            k[r] = umed3(a, b, clo); k[q] = umed3(a, b, chi);
        }
    }
}
// 64 * NR keys, NR per lane, ascending: sorted position of (lane, r) is lane * NR + r
template <int NR> __device__ __forceinline__ void bs_sort(uint32_t (&k)[NR], int lane) {
    bs_static<NR, 2, 1>(k);
    bs_static<NR, 4, 2>(k); bs_static<NR, 4, 1>(k);
    if (NR >= 16) { bs_static<NR, 8, 4>(k); bs_static<NR, 8, 2>(k); bs_static<NR, 8, 1>(k); }
    for (int ll = 0; ll <= 6; ll++) {                     // runs of NR << ll elements
        const bool asc = (lane & (1 << ll)) == 0;
        for (int m = (1 << ll) >> 1; m > 0; m >>= 1) {
            const uint32_t c = (((lane & m) == 0) == asc) ? 0u : 0xffffffffu;
#pragma unroll
            for (int r = 0; r < NR; r++) k[r] = umed3(k[r], (uint32_t)__shfl_xor((int)k[r], m), c);
        }
        const uint32_t clo = asc ? 0u : 0xffffffffu;
        if (NR >= 16) bs_lane<NR, 8>(k, clo);
        bs_lane<NR, 4>(k, clo); bs_lane<NR, 2>(k, clo); bs_lane<NR, 1>(k, clo);
    }
}
__device__ __forceinline__ uint32_t box_float_key(float f) { const uint32_t u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ int box_bslot(int i) { return i + (i >> 4); }       // sorted bracket in LDS: one pad word per lane's 16

#ifndef BOXF_MINW
#define BOXF_MINW 3
#endif
__global__ __launch_bounds__(64, BOXF_MINW) void k_bkg_boxstats_fast(const float* __restrict__ data, const uint8_t* __restrict__ mask,
                                                          const uint8_t* __restrict__ objmask, int ny, int nx, int box,
                                                          int nbx, int nboxes, float limfrac, float* __restrict__ mini_med,
                                                          float* __restrict__ mini_std, int32_t* nfail, int32_t* __restrict__ flist) {
    __shared__ uint32_t s_br[BOXF_NB + 128];                 // list (later: sorted, one pad word per 16) | a slot per lane for the stores that do not count
    __shared__ uint32_t s_tl[BOXF_NT + 64];
    const int lane = threadIdx.x;
    const int per = (nboxes + 7) >> 3;
    const int ibox = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (ibox >= nboxes) return;
    const int by = ibox / nbx, bx = ibox - by * nbx;
    const int npx = box * box;
    uint32_t k[64];
    int cnt = 0;
    {
        // register c = row c of the box, lane = column (the lanes beyond the box width re-read its last column and are
        // discarded; rows beyond the box are not loaded): the address of a load is the row's scalar base + 4 * lane, and no
        // element needs a bounds test of its own.  All loads are issued before the first use: the kernel lives on the
        // memory parallelism of its few waves.
        const int xl = min(lane, box - 1);
        const size_t o0 = (size_t)(by * box) * nx + (size_t)bx * box + (size_t)xl;
        const float* pd = data + o0; const uint8_t* pm = mask + o0; const uint8_t* po = objmask + o0;
        const bool col = lane < box;
        unsigned mk[64];
#pragma unroll
        for (int c = 0; c < 64; c++) {
            k[c] = 0; mk[c] = 1;
            if (c < box) {                                         // (uniform)
                k[c] = __float_as_uint(pd[(size_t)c * nx]);
                mk[c] = pm[(size_t)c * nx];
                if (objmask) mk[c] |= po[(size_t)c * nx];
            }
        }
#pragma unroll
        for (int c = 0; c < 64; c++) {
            const uint32_t u = k[c];
            const float d = __uint_as_float(u);
            const bool ok = col & (mk[c] == 0) & (d != 0.f) & (d == d);
            k[c] = ok ? (u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u)) : BOX_PAD;
            cnt += ok;
        }
    }
    const int n0 = wave_sum_i32(cnt);
    if ((float)n0 < limfrac * (float)npx || n0 == 0) {
        if (lane == 0) { const float nanv = __uint_as_float(0x7fc00000u); mini_med[ibox] = nanv; mini_std[ibox] = nanv; }
        return;
    }
#if defined(BOXK_NOFAIL)                                           // (timing knock-outs, tools/dbg/box_knock.sh; never in the product build)
#define BOXF_LEAVE(why) do { } while (0)
#elif defined(BOXV)                                                // (tools/dbg/box_fail.py: why boxes are left to the full sort; never in the product build)
#define BOXF_LEAVE(why) do { if (lane == 0) { flist[atomicAdd(nfail, 1)] = ibox; mini_std[ibox] = __uint_as_float(0x7fc0b0b0u); mini_med[ibox] = (float)(why); } return; } while (0)
#else
#define BOXF_LEAVE(why) do { if (lane == 0) flist[atomicAdd(nfail, 1)] = ibox; return; } while (0)
#endif
    // ---- the sample: slots 8 j * 64 + lane, spread over the rows of the box
    uint32_t lo_key, hi_key, tl_key, th_key;
    double piv;
    {
        uint32_t sm[8];
        int nv = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { sm[j] = k[8 * j]; nv += sm[j] != BOX_PAD; }
        const int nsv = wave_sum_i32(nv);
        if (nsv < 128) BOXF_LEAVE(1);
#ifndef BOXK_NOSAMP
        bs_sort<8>(sm, lane);
#endif
#pragma unroll
        for (int j = 0; j < 8; j++) s_br[lane * 8 + j] = sm[j];
        __syncthreads();
        const int jl = (int)((float)nsv * BOXF_QLO), jh = min((int)((float)nsv * BOXF_QHI) + 1, nsv - 1);
        // wings: below the sample's 5 % point and above its 85 % point (stars make the upper wing the one a clip reaches into)
        const int jtl = (int)((float)nsv * 0.05f), jth = min((int)((float)nsv * 0.85f), nsv - 1);
        lo_key = s_br[jl]; hi_key = s_br[jh]; tl_key = s_br[jtl]; th_key = s_br[jth];
        piv = (double)box_key_value(s_br[nsv >> 1]);
        __syncthreads();
    }
    // ---- pass 1 (no lane talks to another): per-lane counts of bracket / wing keys and of keys below the bracket, and the
    // sums about the pivot of the pixels BETWEEN the wings in float32 (they lie within a sigma or two of the pivot: the
    // float32 partial sums of 64 such terms carry ~1e-7; a star pixel would leave the rounding error of its 1e9-sized square
    // behind after the clip has removed it -- the wing pixels' sums are taken in float64 from their list below)
    int c_lo = 0, nb = 0, nt = 0;
    double s1 = 0.0, s2 = 0.0;
    const uint32_t wbr = hi_key - lo_key, wth = BOX_PAD - 1u - th_key;
    int nbl = 0, ntl = 0;
    {
        const float pivf = (float)piv;
        float f1 = 0.f, f2 = 0.f;
#ifdef BOXK_NOCLASS
        for (int c = 0; c < 1; c++) {
#else
#pragma unroll
        for (int c = 0; c < 64; c++) {
#endif
            const uint32_t u = k[c];
            const bool inb = (u - lo_key) <= wbr;
            const bool inw = (u < tl_key) || ((u - th_key - 1u) < wth);          // below the low limit, or th_key < u < BOX_PAD
            nbl += inb ? 1 : 0; ntl += inw ? 1 : 0;
            c_lo += (u < lo_key) ? 1 : 0;
            const uint32_t xb = u ^ ((uint32_t)((int32_t)~u >> 31) | 0x80000000u);          // box_key_value
            const float t = (inw || u == BOX_PAD) ? 0.f : __uint_as_float(xb) - pivf;
            f1 += t; f2 = fmaf(t, t, f2);
        }
        c_lo = wave_sum_i32(c_lo);
        s1 = wave_sum_f64((double)f1); s2 = wave_sum_f64((double)f2);
    }
    // lane offsets into the two lists
    int ob = nbl, ow = ntl;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int tb = __shfl_up(ob, o), tw = __shfl_up(ow, o);
        if (lane >= o) { ob += tb; ow += tw; }
    }
    nb = __builtin_amdgcn_readlane(ob, 63); nt = __builtin_amdgcn_readlane(ow, 63);
    ob -= nbl; ow -= ntl;
    if (nb > BOXF_NB || nt > BOXF_NT) BOXF_LEAVE(nb > BOXF_NB ? 2 : 3);
    // ---- pass 2: the keys into the lists (lane after lane); a key of neither list goes to a slot of the lane's own behind
    // the list, so that no store needs a mask of its own
    {
        const uint32_t dumpb = BOXF_NB + 64 + lane, dumpw = BOXF_NT + lane;
#ifdef BOXK_NOCLASS
        for (int c = 0; c < 1; c++) {
#else
#pragma unroll
        for (int c = 0; c < 64; c++) {
#endif
            uint32_t u = k[c];
            asm volatile("" : "+v"(u));                             // (the compiler would carry pass 1's 128 lane masks over in scalar registers, and spill them)
            const bool inb = (u - lo_key) <= wbr;
            const bool inw = (u < tl_key) || ((u - th_key - 1u) < wth);
            s_br[inb ? (uint32_t)ob : dumpb] = u; ob += inb ? 1 : 0;
            s_tl[inw ? (uint32_t)ow : dumpw] = u; ow += inw ? 1 : 0;
        }
    }
    __syncthreads();
    {
        double d1 = 0.0, d2 = 0.0;
        for (int i = lane; i < nt; i += 64) { const double t = (double)box_key_value(s_tl[i]) - piv; d1 += t; d2 += t * t; }
        s1 += wave_sum_f64(d1); s2 += wave_sum_f64(d2);
    }
    // ---- the bracket, sorted, back into LDS
    if (nb <= 512) {                                               // (uniform) half the network for a bracket that fits 8 registers per lane
        uint32_t kb[8];
#pragma unroll
        for (int r = 0; r < 8; r++) kb[r] = (r * 64 + lane < nb) ? s_br[r * 64 + lane] : BOX_PAD;
        __syncthreads();
#ifndef BOXK_NOBR
        bs_sort<8>(kb, lane);
#endif
#pragma unroll
        for (int r = 0; r < 8; r++) s_br[box_bslot(lane * 8 + r)] = kb[r];
        __syncthreads();
    } else {
        uint32_t kb[16];
#pragma unroll
        for (int r = 0; r < 16; r++) kb[r] = (r * 64 + lane < nb) ? s_br[r * 64 + lane] : BOX_PAD;
        __syncthreads();
#ifndef BOXK_NOBR
        bs_sort<16>(kb, lane);
#endif
#pragma unroll
        for (int r = 0; r < 16; r++) s_br[lane * 17 + r] = kb[r];
        __syncthreads();
    }
    // ---- the clip: survivors = keys in [cl, ch]; a = how many were removed below
    int a = 0, n = n0;
    uint32_t cl = 0u, ch = BOX_PAD - 1u;
    double sd = 0.0, med = 0.0;
    for (int it = 0; it <= 5; it++) {
        const double m1 = s1 / (double)n;
        const double var = s2 / (double)n - m1 * m1;      // std about the mean, ddof 0
        sd = sqrt(var > 0.0 ? var : 0.0);
        const int r1 = a + n / 2 - c_lo, r0 = (n & 1) ? r1 : r1 - 1;
        if (r0 < 0 || r1 >= nb) BOXF_LEAVE(4);            // a median outside the bracket
        const double v1 = (double)box_key_value(s_br[box_bslot(r1)]);
        med = (n & 1) ? v1 : ((double)box_key_value(s_br[box_bslot(r0)]) + v1) * 0.5;
        if (it == 5) break;
        const double lo = med - 3.0 * sd, hi = med + 3.0 * sd;
        // x < lo  <=>  key(x) < nl;  x > hi  <=>  key(x) > nh   (for float x: the limits rounded to float, stepped inwards where the rounding went outwards)
        const float flo = (float)lo, fhi = (float)hi;
        const uint32_t nl = box_float_key(flo) + (((double)flo < lo) ? 1u : 0u);
        const uint32_t nh = box_float_key(fhi) - (((double)fhi > hi) ? 1u : 0u);
        if (nl > tl_key || nh < th_key) BOXF_LEAVE(5);    // a limit inside the unlisted middle
        int tlo = 0, thi = 0;
        double d1 = 0.0, d2 = 0.0;
        for (int i = lane; i < nt; i += 64) {
            const uint32_t u = s_tl[i];
            const bool surv = u >= cl && u <= ch;
            const bool lc = surv && u < nl, hc = surv && u > nh;
            tlo += lc; thi += hc;
            if (lc || hc) { const double t = (double)box_key_value(u) - piv; d1 += t; d2 += t * t; }
        }
        tlo = wave_sum_i32(tlo); thi = wave_sum_i32(thi);
        if (tlo == 0 && thi == 0) break;                  // nothing clipped: these are the final statistics
        s1 -= wave_sum_f64(d1); s2 -= wave_sum_f64(d2);
        a += tlo; n -= tlo + thi;
        cl = max(cl, nl); ch = min(ch, nh);
        if (n <= 0) break;
    }
    if (lane == 0) {
        if (n <= 0) { const float nanv = __uint_as_float(0x7fc00000u); mini_med[ibox] = nanv; mini_std[ibox] = nanv; }
        else { mini_med[ibox] = (float)med; mini_std[ibox] = (float)sd; }
    }
#undef BOXF_LEAVE
}

// one workgroup; [mini] is updated in place, [tmp] is scratch of the same size
__global__ __launch_bounds__(1024) void k_mini_fill_filter(float* mini, float* tmp, int nby, int nbx, int32_t* err) {
    __shared__ int nbad, nfixed;
    const int n = nby * nbx;
    float* cur = mini; float* nxt = tmp;
    for (int iter = 0; iter < n + 1; iter++) {
        if (threadIdx.x == 0) { nbad = 0; nfixed = 0; }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            float val = cur[i];
            if (!(val == val)) {
                const int y = i / nbx, x = i - y * nbx;
                float w[9]; int m = 0;
                for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || xx < 0 || yy >= nby || xx >= nbx) continue;
                    const float t = cur[yy * nbx + xx];
                    if (t == t) w[m++] = t;
                }
                if (m > 0) {
                    for (int a = 1; a < m; a++) { const float t = w[a]; int b = a - 1; while (b >= 0 && w[b] > t) { w[b + 1] = w[b]; b--; } w[b + 1] = t; }
                    val = (m & 1) ? w[m / 2] : (float)(((double)w[m / 2 - 1] + (double)w[m / 2]) * 0.5);
                    atomicAdd(&nfixed, 1);
                } else atomicAdd(&nbad, 1);
            }
            nxt[i] = val;
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
        const int nb = nbad, nf = nfixed;
        __syncthreads();
        if (nb == 0 && nf == 0) break;                    // nothing left to fill
        if (nf == 0) break;                               // all-NaN image: cannot be filled
    }
    // 3x3 median, replicated edges: cur -> nxt
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / nbx, x = i - y * nbx;
        float w[9]; int m = 0;
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            const int yy = min(max(y + dy, 0), nby - 1), xx = min(max(x + dx, 0), nbx - 1);
            w[m++] = cur[yy * nbx + xx];
        }
        BBX_MED9(w);
        nxt[i] = w[4];
    }
    __syncthreads();
    if (nxt != mini) for (int i = threadIdx.x; i < n; i += blockDim.x) mini[i] = nxt[i];
    (void)err;
}

// The same with the mini image in LDS (a 176 x 176 mini image: 124 KB of the CU's 160): every neighbour read of the fill
// passes and of the 3 x 3 median is an LDS read instead of a ~1 us trip to L2 by one lonely workgroup (91 -> ~20 us per
// launch, two launches per frame, all of it latency of the frame).  A fill pass reads the snapshot, then writes: all new
// values are formed (registers) before the first is stored, as the two-buffer version does.
#define MINI_LDS_MAX (36 * 1024)                              // floats: 144 KB
#define MINI_PER_THREAD ((MINI_LDS_MAX + 1023) / 1024)
__global__ __launch_bounds__(1024) void k_mini_fill_filter_lds(float* mini, int nby, int nbx) {
    extern __shared__ float s_mini[];
    __shared__ int nbad, nfixed;
    const int n = nby * nbx;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_mini[i] = mini[i];
    __syncthreads();
    for (int iter = 0; iter < n + 1; iter++) {
        if (threadIdx.x == 0) { nbad = 0; nfixed = 0; }
        __syncthreads();
        float nv[MINI_PER_THREAD];                           // the new values of this thread's NaN entries (few)
        unsigned long long fixed = 0ull;
        int k = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x, k++) {
            const float val = s_mini[i];
            if (!(val == val)) {
                const int y = i / nbx, x = i - y * nbx;
                float w[9]; int m = 0;
                for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || xx < 0 || yy >= nby || xx >= nbx) continue;
                    const float t = s_mini[yy * nbx + xx];
                    if (t == t) w[m++] = t;
                }
                if (m > 0) {
                    for (int a = 1; a < m; a++) { const float t = w[a]; int b = a - 1; while (b >= 0 && w[b] > t) { w[b + 1] = w[b]; b--; } w[b + 1] = t; }
                    nv[k] = (m & 1) ? w[m / 2] : (float)(((double)w[m / 2 - 1] + (double)w[m / 2]) * 0.5);
                    fixed |= 1ull << k;
                    atomicAdd(&nfixed, 1);
                } else atomicAdd(&nbad, 1);
            }
        }
        __syncthreads();                                     // every read of the snapshot is done
        k = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x, k++) if ((fixed >> k) & 1ull) s_mini[i] = nv[k];
        const int nb = nbad, nf = nfixed;
        __syncthreads();
        if (nb == 0 && nf == 0) break;                    // nothing left to fill
        if (nf == 0) break;                               // all-NaN image: cannot be filled
    }
    // 3x3 median, replicated edges
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / nbx, x = i - y * nbx;
        float w[9]; int m = 0;
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            const int yy = min(max(y + dy, 0), nby - 1), xx = min(max(x + dx, 0), nbx - 1);
            w[m++] = s_mini[yy * nbx + xx];
        }
        BBX_MED9(w);
        mini[i] = w[4];
    }
}

// np.median of a float32 array (one workgroup): exact radix select of the middle element(s) on order-preserving keys,
// an even count gives the float32 mean of the two (numpy: np.mean of the pair in float32); NaN when the array holds one.
// The same for n <= 32768 (a 176 x 176 mini image): the keys stay in registers, 32 per thread, and the middle element is
// found bit by bit -- ans = the largest value with count(keys < ans) <= rank -- with compare-and-count passes over the
// registers (the histogram version below spends its time in LDS atomics: 169 us against 15).
__device__ __forceinline__ unsigned mm_count_below(const unsigned (&k)[32], int nk, unsigned cand, int* sh) {
    int c = 0;
#pragma unroll
    for (int r = 0; r < 32; r++) if (r < nk) c += k[r] < cand ? 1 : 0;
    c = wave_sum_i32(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) t += sh[w];
    return (unsigned)t;
}
__global__ __launch_bounds__(1024) void k_mini_median_regs(const float* __restrict__ a, int n, float* __restrict__ out) {
    __shared__ int sh[16];
    __shared__ unsigned s_nan;
    const int tid = threadIdx.x;
    unsigned k[32];
    const int nk = (n - tid + 1023) / 1024;                      // elements tid, tid + 1024, ... of this thread
    bool nan = false;
#pragma unroll
    for (int r = 0; r < 32; r++) {
        k[r] = 0xffffffffu;
        if (r < nk) { const float v = a[tid + 1024 * r]; nan |= !(v == v); const unsigned u = __float_as_uint(v); k[r] = (u >> 31) ? ~u : (u | 0x80000000u); }
    }
    if (tid == 0) s_nan = 0;
    __syncthreads();
    if (nan) s_nan = 1;
    __syncthreads();
    if (s_nan) { if (tid == 0) out[0] = __uint_as_float(0x7fc00000u); return; }
    // lower middle element: rank (n - 1) / 2
    const unsigned rank = (unsigned)((n - 1) / 2);
    unsigned ans = 0;
    for (int b = 31; b >= 0; b--) {
        const unsigned cand = ans | (1u << b);
        if (mm_count_below(k, nk, cand, sh) <= rank) ans = cand;
    }
    unsigned ans2 = ans;
    if (!(n & 1)) {
        // upper middle element (rank n / 2): the same value when enough keys equal it, else the smallest key above it
        const unsigned le = ans == 0xffffffffu ? (unsigned)n : mm_count_below(k, nk, ans + 1u, sh);
        if (le <= (unsigned)(n / 2)) {
            unsigned m = 0xffffffffu;
#pragma unroll
            for (int r = 0; r < 32; r++) if (r < nk && k[r] > ans) m = min(m, k[r]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o, 64));
            __syncthreads();
            if ((tid & 63) == 0) sh[tid >> 6] = (int)m;
            __syncthreads();
            for (int w = 0; w < 16; w++) m = min(m, (unsigned)sh[w]);
            ans2 = m;
        }
    }
    if (tid == 0) {
        const float v1 = __uint_as_float((ans >> 31) ? (ans & 0x7fffffffu) : ~ans), v2 = __uint_as_float((ans2 >> 31) ? (ans2 & 0x7fffffffu) : ~ans2);
        out[0] = (n & 1) ? v1 : (v1 + v2) * 0.5f;
    }
}

__global__ __launch_bounds__(1024) void k_mini_median(const float* __restrict__ a, int n, float* __restrict__ out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_rank, s_nan;
    __shared__ float s_val[2];
    const int tid = threadIdx.x;
    if (tid == 0) s_nan = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) if (!(a[i] == a[i])) s_nan = 1;
    __syncthreads();
    if (s_nan || n < 1) { if (tid == 0) out[0] = __uint_as_float(0x7fc00000u); return; }
    for (int which = 0; which < 2; which++) {
        const unsigned rank = which == 0 ? (unsigned)((n - 1) / 2) : (unsigned)(n / 2);
        if (tid == 0) { s_prefix = 0; s_rank = rank; }
        __syncthreads();
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += 1024) hist[i] = 0;
            __syncthreads();
            const unsigned pre = s_prefix;
            for (int i = tid; i < n; i += 1024) {
                const unsigned u = __float_as_uint(a[i]);
                const unsigned k = (u >> 31) ? ~u : (u | 0x80000000u);
                if (shift == 24 || (k >> (shift + 8)) == (pre >> (shift + 8))) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned r = s_rank, b = 0;
                while (r >= hist[b]) { r -= hist[b]; b++; }
                s_rank = r; s_prefix = pre | (b << shift);
            }
            __syncthreads();
        }
        if (tid == 0) { const unsigned k = s_prefix; s_val[which] = __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k); }
        __syncthreads();
    }
    if (tid == 0) out[0] = (n & 1) ? s_val[0] : (s_val[0] + s_val[1]) * 0.5f;
}

// One workgroup = 256 columns x ZOOM_ROWS rows: the column taps (index + 4 float64 weights per
// pixel, 36 bytes) are read once per thread and reused down the rows -- with a workgroup per row
// they came through the L2 again for every row, 4.5x the bytes of the image itself.
#ifndef ZOOM_ROWS
#define ZOOM_ROWS 32                // (16: 0.157 / 0.340 ms write-only / with the subtraction; 32: 0.144 / 0.316; 64: 0.148 / 0.299)
#endif
#define ZOOM_CQ 2048                // LDS queue of candidate pixels per workgroup (k_spline_zoom4)
static_assert(ZOOM_ROWS * 32 <= 1024, "k_spline_zoom folds ZOOM_ROWS x 32 coefficients into rowc[2][512]");
__global__ __launch_bounds__(256) void k_spline_zoom(int ny, int nx, const double* __restrict__ coef, int cnx,
                                                     const int32_t* __restrict__ fy, const double* __restrict__ wy,
                                                     const int32_t* __restrict__ fx, const double* __restrict__ wx,
                                                     float* data, float* bkg, const float* __restrict__ src) {
    // (src: where the pixels to subtract from are read -- data itself, or another frame that stays as it is)
    // the 4 coefficient rows of an output row are first folded with the row weights into a
    // short LDS vector (a 256-pixel span touches only a handful of coefficient columns); each
    // pixel then needs 4 taps instead of 16 strided float64 loads
    __shared__ double rowc[2][512];
    const int X0 = blockIdx.x * blockDim.x, X = X0 + threadIdx.x;
    const int j0 = fx[X0] - 1, j1 = fx[min(X0 + (int)blockDim.x - 1, nx - 1)] + 2;      // fx is non-decreasing
    const int span = j1 - j0 + 1;
    const bool live = X < nx;
    const int ix = live ? fx[X] : 0;
    double wxr[4] = {0, 0, 0, 0};
    if (live) { wxr[0] = wx[X * 4]; wxr[1] = wx[X * 4 + 1]; wxr[2] = wx[X * 4 + 2]; wxr[3] = wx[X * 4 + 3]; }
    const int k = ix - 1 - j0;
    const int Y0 = blockIdx.y * ZOOM_ROWS, Y1 = min(ny, Y0 + ZOOM_ROWS);
    if (span <= 32) {
        // the usual case (zoom factor >> 1: a 256-pixel span touches ~10 coefficient columns): the folded
        // coefficient vectors of all ZOOM_ROWS rows at once, one barrier, then 4 taps per pixel and row
        double* rc = &rowc[0][0];                               // [ZOOM_ROWS][32]
        for (int e = threadIdx.x; e < ZOOM_ROWS * 32; e += blockDim.x) {
            const int r = e >> 5, j = e & 31, Y = Y0 + r;
            if (Y < Y1 && j < span) {
                const double* c0 = coef + (size_t)(fy[Y] - 1) * cnx + (j0 + j);
                rc[e] = ((c0[0] * wy[Y * 4] + c0[cnx] * wy[Y * 4 + 1]) + c0[2 * (size_t)cnx] * wy[Y * 4 + 2]) + c0[3 * (size_t)cnx] * wy[Y * 4 + 3];
            }
        }
        __syncthreads();
        if (!live) return;
        for (int Y = Y0; Y < Y1; Y++) {
            const double* r4 = rc + ((Y - Y0) << 5) + k;
            double t = 0.0;
#pragma unroll
            for (int b = 0; b < 4; b++) t += r4[b] * wxr[b];
            const float v = (float)t;
            const size_t o = (size_t)Y * nx + X;
            if (bkg) bkg[o] = v;
            if (data) data[o] = (src ? src[o] : data[o]) - v;
        }
        return;
    }
    for (int Y = Y0; Y < Y1; Y++) {
        const int iy = fy[Y];
        double t = 0.0;
        if (span <= 512) {
            double* rc = rowc[(Y - Y0) & 1];
            const double w0 = wy[Y * 4], w1 = wy[Y * 4 + 1], w2 = wy[Y * 4 + 2], w3 = wy[Y * 4 + 3];
            for (int j = threadIdx.x; j < span; j += blockDim.x) {
                const double* c0 = coef + (size_t)(iy - 1) * cnx + (j0 + j);
                rc[j] = ((c0[0] * w0 + c0[cnx] * w1) + c0[2 * (size_t)cnx] * w2) + c0[3 * (size_t)cnx] * w3;
            }
            __syncthreads();                                     // (the other buffer is free again after the next barrier)
            if (live) {
#pragma unroll
                for (int b = 0; b < 4; b++) t += rc[k + b] * wxr[b];
            }
        } else if (live) {
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const double wa = wy[Y * 4 + a];
                const double* row = coef + (size_t)(iy - 1 + a) * cnx + (ix - 1);
#pragma unroll
                for (int b = 0; b < 4; b++) t += row[b] * (wa * wxr[b]);
            }
        }
        if (live) {
            const float v = (float)t;
            const size_t o = (size_t)Y * nx + X;
            if (bkg) bkg[o] = v;
            if (data) data[o] = (src ? src[o] : data[o]) - v;
        }
    }
}

// The same with four neighbouring pixels per thread (frames whose rows are multiples of 4 pixels, 16-byte aligned): one
// 16-byte load / store per thread and row instead of four 4-byte ones -- the scalar kernel streams at 2.8 TB/s, this one
// at the rate of the memory system.  Same operations in the same order: the same bits.
__global__ __launch_bounds__(256) void k_spline_zoom4(int ny, int nx, const double* __restrict__ coef, int cnx,
                                                      const int32_t* __restrict__ fy, const double* __restrict__ wy,
                                                      const int32_t* __restrict__ fx, const double* __restrict__ wx,
                                                      float* data, float* bkg, const float* __restrict__ src,
                                                      const float* __restrict__ cand_med, double cand_nsig, uint32_t* __restrict__ cand_list,
                                                      int32_t* __restrict__ cand_cnt, uint32_t cand_cap, int32_t* __restrict__ d_err) {
    // cand_list: the pixels of the subtracted frame with |value| >= (float)(cand_med[0] * cand_nsig) are listed as they
    // are written (the catalogue search's pass over the frame, bbx_zoom_candidates).  They come in clusters (stars): a
    // workgroup queues them in LDS, one reservation per wave and row, and appends the queue with one global
    // reservation at its end; a wave that finds the queue full appends its own directly.
    __shared__ uint32_t cq[ZOOM_CQ];
    __shared__ unsigned cqn, cqbad;
    const float cthr = cand_list ? (float)((double)cand_med[0] * cand_nsig) : 0.f;
    if (cand_list) { if (threadIdx.x == 0) { cqn = 0; cqbad = 0xffffffffu; } }
    __shared__ double rc[ZOOM_ROWS * 64];      // (64 columns: a span of 1024 pixels that crosses a channel border takes in its two padded patch edges)
    const int X0 = blockIdx.x * 1024, X = X0 + 4 * (int)threadIdx.x;
    const int j0 = fx[X0] - 1, j1 = fx[min(X0 + 1023, nx - 1)] + 2;
    const int span = j1 - j0 + 1;
    const bool live = X < nx;                                   // nx is a multiple of 4: all four pixels or none
    const int Y0 = blockIdx.y * ZOOM_ROWS, Y1 = min(ny, Y0 + ZOOM_ROWS);
    int kc[4] = {0, 0, 0, 0};
    double w[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int b = 0; b < 4; b++) w[c][b] = live ? wx[(size_t)(X + c) * 4 + b] : 0.0;
        kc[c] = live ? fx[X + c] - 1 - j0 : 0;
    }
    if (span <= 64) {
        for (int e = threadIdx.x; e < ZOOM_ROWS * 64; e += blockDim.x) {
            const int r = e >> 6, j = e & 63, Y = Y0 + r;
            if (Y < Y1 && j < span) {
                const double* c0 = coef + (size_t)(fy[Y] - 1) * cnx + (j0 + j);
                rc[e] = ((c0[0] * wy[Y * 4] + c0[cnx] * wy[Y * 4 + 1]) + c0[2 * (size_t)cnx] * wy[Y * 4 + 2]) + c0[3 * (size_t)cnx] * wy[Y * 4 + 3];
            }
        }
        __syncthreads();
        for (int Y = Y0; Y < Y1 && live; Y++) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double* r4 = rc + ((Y - Y0) << 6) + kc[c];
                double t = 0.0;
#pragma unroll
                for (int b = 0; b < 4; b++) t += r4[b] * w[c][b];
                v[c] = (float)t;
            }
            const size_t o = (size_t)Y * nx + X;
            typedef float zv4 __attribute__((ext_vector_type(4)));
            if (bkg) __builtin_nontemporal_store(zv4{v[0], v[1], v[2], v[3]}, reinterpret_cast<zv4*>(bkg + o));
            if (data) {
                // (streamed once in, once out: non-temporal -- plain stores of a read + write stream keep their lines in L2 and
                // cost the reads their share of it, tools/exp/tile_bw.hip)
                const zv4 d = __builtin_nontemporal_load(reinterpret_cast<const zv4*>((src ? src : data) + o));
                const float r0 = d.x - v[0], r1 = d.y - v[1], r2 = d.z - v[2], r3 = d.w - v[3];
                __builtin_nontemporal_store(zv4{r0, r1, r2, r3}, reinterpret_cast<zv4*>(data + o));
                if (cand_list) {
                    const float rr[4] = {r0, r1, r2, r3};
                    bool hit[4]; unsigned long long hm[4]; unsigned tot = 0;
#pragma unroll
                    for (int c = 0; c < 4; c++) { hit[c] = fabsf(rr[c]) >= cthr; hm[c] = __builtin_amdgcn_ballot_w64(hit[c]); tot += (unsigned)__popcll(hm[c]); }
                    if (tot) {                                       // wave-uniform
                        const int lane = threadIdx.x & 63;
                        const unsigned long long any = hm[0] | hm[1] | hm[2] | hm[3];
                        const int leader = (int)__builtin_ctzll(any);
                        unsigned base = 0;
                        if (lane == leader) base = atomicAdd(&cqn, tot);
                        base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
                        const bool fits = base + tot <= ZOOM_CQ;
                        unsigned gbase = 0;
                        if (!fits) {
                            if (lane == leader) { atomicMin(&cqbad, base); gbase = atomicAdd((unsigned*)cand_cnt, tot); }
                            gbase = (unsigned)__builtin_amdgcn_readlane((int)gbase, leader);
                        }
                        unsigned off = 0;
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            if (hit[c]) {
                                const unsigned k = off + __builtin_amdgcn_mbcnt_hi((unsigned)(hm[c] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm[c], 0u));
                                const uint32_t px = (uint32_t)(o + c);
                                if (fits) cq[base + k] = px;
                                else if (gbase + k < cand_cap) cand_list[gbase + k] = px;
                                else atomicOr(d_err, BBX_DERR_LIST_OVERFLOW);
                            }
                            off += (unsigned)__popcll(hm[c]);
                        }
                    }
                }
            }
        }
        if (cand_list) {
            __syncthreads();
            const unsigned nq = min(min(cqn, cqbad), (unsigned)ZOOM_CQ);
            __shared__ unsigned cg;
            if (nq) {
                if (threadIdx.x == 0) cg = atomicAdd((unsigned*)cand_cnt, nq);
                __syncthreads();
                for (unsigned t = threadIdx.x; t < nq; t += blockDim.x) {
                    if (cg + t < cand_cap) cand_list[cg + t] = cq[t]; else atomicOr(d_err, BBX_DERR_LIST_OVERFLOW);
                }
            }
        }
        return;
    }
    // zoom factors so small that 1024 pixels span more than 64 coefficient columns: every pixel from its 16 taps
    if (!live) return;
    for (int Y = Y0; Y < Y1; Y++) {
        const int iy = fy[Y];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            double t = 0.0;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const double wa = wy[Y * 4 + a];
                const double* row = coef + (size_t)(iy - 1 + a) * cnx + (j0 + kc[c]);
#pragma unroll
                for (int b = 0; b < 4; b++) t += row[b] * (wa * w[c][b]);
            }
            const float v = (float)t;
            const size_t o = (size_t)Y * nx + X + c;
            if (bkg) bkg[o] = v;
            if (data) {
                const float r = (src ? src[o] : data[o]) - v;
                data[o] = r;
                if (cand_list && fabsf(r) >= cthr) {                 // (rare geometry: a reservation per pixel)
                    const unsigned k = atomicAdd((unsigned*)cand_cnt, 1u);
                    if (k < cand_cap) cand_list[k] = (uint32_t)o; else atomicOr(d_err, BBX_DERR_LIST_OVERFLOW);
                }
            }
        }
    }
}

// ---- B-spline prefilter of the mini image: scipy.ndimage.spline_filter(np.pad(block, npad, 'edge'), order = 3, mode = 'nearest') ----
// What scipy's C does per line (ni_splines.c: apply_filter, _init_causal_reflect, _init_anticausal_reflect), operation by
// operation in float64 and without fused multiply-adds, so that the coefficients are the same bits: gain, the in-place
// causal start (its last term reads the c[0] it is accumulating into), the two recursions.  The pole is sqrt(3) - 2
// correctly rounded, as gcc folds `sqrt(3.0) - 2.0` when it compiles scipy (the run-time double expression is 2 ulp off);
// pow(z, len) comes from the host's libm, like scipy's.
#define SPF_LINES 32
#define SPF_MAXLEN 512
__device__ __forceinline__ void spf_line(double* c, int n, int stride, double zn) {
    // (the recursions run on blocks of 8 values held in registers: with a load and a store of LDS per step the
    // 200 steps of a line are 200 round trips to LDS -- 0.2 ms for a 200 x 200 patch)
    const double z = -0.2679491924311227064725536584941276330571947461896;
    const double gain = 1.0 * ((1.0 - 1.0 / z) * (1.0 - z));
    for (int i = 0; i < n; i++) c[i * stride] *= gain;
    if (n < 2) return;
    const double c0 = c[0];
    double acc = c[(n - 1) * stride] * zn + c0, zi = z;
    for (int i = 1; i < n; i++) {
        // scipy accumulates into c[0] in place: its last term (i = n - 1) reads the running sum, not c0
        const double back = i == n - 1 ? acc : c[(n - 1 - i) * stride];
        acc += zi * (back * zn + c[i * stride]);
        zi *= z;
    }
    double prev = acc * (z / (1.0 - zn * zn)) + c0;
    c[0] = prev;
    for (int i0 = 1; i0 < n; i0 += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) if (i0 + j < n) v[j] = c[(i0 + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; j++) if (i0 + j < n) { prev = v[j] + z * prev; v[j] = prev; }
#pragma unroll
        for (int j = 0; j < 8; j++) if (i0 + j < n) c[(i0 + j) * stride] = v[j];
    }
    prev *= z / (z - 1.0);                                         // c[n - 1]
    c[(n - 1) * stride] = prev;
    for (int i0 = n - 2; i0 >= 0; i0 -= 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) if (i0 - j >= 0) v[j] = c[(i0 - j) * stride];
#pragma unroll
        for (int j = 0; j < 8; j++) if (i0 - j >= 0) { prev = z * (prev - v[j]); v[j] = prev; }
#pragma unroll
        for (int j = 0; j < 8; j++) if (i0 - j >= 0) c[(i0 - j) * stride] = v[j];
    }
}
// axis 0: strips of SPF_LINES columns of one padded block, read from the mini image with the edge padding
__global__ __launch_bounds__(256) void k_spf_axis0(const float* __restrict__ mini, int nbx, int cy, int cx, int npad, int nblkx, double zn,
                                                    double* __restrict__ coef, int cnx) {
    extern __shared__ double sp[];                    // [py][SPF_LINES + 1]
    const int py = cy + 2 * npad, px = cx + 2 * npad;
    const int nstrip = (px + SPF_LINES - 1) / SPF_LINES;
    const int blk = blockIdx.x / nstrip, j0 = (blockIdx.x - blk * nstrip) * SPF_LINES;
    const int by = blk / nblkx, bx = blk - by * nblkx;
    const int W = SPF_LINES + 1;
    for (int e = threadIdx.x; e < py * SPF_LINES; e += blockDim.x) {
        const int i = e / SPF_LINES, jj = e - i * SPF_LINES, j = j0 + jj;
        if (j < px) {
            const int my = by * cy + min(max(i - npad, 0), cy - 1), mx = bx * cx + min(max(j - npad, 0), cx - 1);
            sp[i * W + jj] = (double)mini[(size_t)my * nbx + mx];
        }
    }
    __syncthreads();
    if (threadIdx.x < SPF_LINES && j0 + (int)threadIdx.x < px) spf_line(sp + threadIdx.x, py, W, zn);
    __syncthreads();
    for (int e = threadIdx.x; e < py * SPF_LINES; e += blockDim.x) {
        const int i = e / SPF_LINES, jj = e - i * SPF_LINES, j = j0 + jj;
        if (j < px) coef[(size_t)(by * py + i) * cnx + bx * px + j] = sp[i * W + jj];
    }
}
// axis 1: strips of SPF_LINES rows, in place
__global__ __launch_bounds__(256) void k_spf_axis1(int cy, int cx, int npad, int nblkx, double zn, double* __restrict__ coef, int cnx) {
    extern __shared__ double sp[];                    // [SPF_LINES][px + 1]
    const int py = cy + 2 * npad, px = cx + 2 * npad;
    const int nstrip = (py + SPF_LINES - 1) / SPF_LINES;
    const int blk = blockIdx.x / nstrip, i0 = (blockIdx.x - blk * nstrip) * SPF_LINES;
    const int by = blk / nblkx, bx = blk - by * nblkx;
    const int W = px + 1;
    for (int e = threadIdx.x; e < SPF_LINES * px; e += blockDim.x) {
        const int ii = e / px, j = e - ii * px;
        if (i0 + ii < py) sp[ii * W + j] = coef[(size_t)(by * py + i0 + ii) * cnx + bx * px + j];
    }
    __syncthreads();
    if (threadIdx.x < SPF_LINES && i0 + (int)threadIdx.x < py) spf_line(sp + threadIdx.x * W, px, 1, zn);
    __syncthreads();
    for (int e = threadIdx.x; e < SPF_LINES * px; e += blockDim.x) {
        const int ii = e / px, j = e - ii * px;
        if (i0 + ii < py) coef[(size_t)(by * py + i0 + ii) * cnx + bx * px + j] = sp[ii * W + j];
    }
}

extern "C" {

int bbx_bkg_boxstats(bbx_ctx* ctx, int ny, int nx, int box, const float* d_data, const uint8_t* d_mask,
                     const uint8_t* d_objmask, float limfrac, float* d_mini_med, float* d_mini_std, void* stream) {
    if (!ctx || !d_data || !d_mask || !d_mini_med || !d_mini_std) return BBX_ERR_ARG;
    if (box < 2 || box > 64 || ny % box || nx % box || (size_t)nx * 64 > 0x7fffffffu) return BBX_ERR_ARG;
    const int nby = ny / box, nbx = nx / box;
    const int nboxes = nby * nbx;
    if (ctx->bkg_full_sort) {
        hipLaunchKernelGGL(k_bkg_boxstats, dim3(((nboxes + 7) / 8) * 8), dim3(64), 0, (hipStream_t)stream, d_data, d_mask, d_objmask,
                           ny, nx, box, nbx, nboxes, limfrac, d_mini_med, d_mini_std);
    } else {
        int rc;
        int32_t* flist = (int32_t*)bbx_ws(ctx, WS_BOXFAIL, (size_t)nboxes * sizeof(int32_t), &rc); if (rc) return rc;
        int32_t* nf = &ctx->d_counters[CNT_BOXFAIL0 + (ctx->box_pp & 1)], *nf_next = &ctx->d_counters[CNT_BOXFAIL0 + ((ctx->box_pp + 1) & 1)];
        ctx->box_pp ^= 1;
        hipLaunchKernelGGL(k_bkg_boxstats_fast, dim3(((nboxes + 7) / 8) * 8), dim3(64), 0, (hipStream_t)stream, d_data, d_mask, d_objmask,
                           ny, nx, box, nbx, nboxes, limfrac, d_mini_med, d_mini_std, nf, flist);
#if defined(BOXV) || defined(BOXK_NOFAIL)
        if (getenv("BBX_DBG_BOX_NOFALLBACK")) return BBX_OK;
#endif
        hipLaunchKernelGGL(k_bkg_boxstats_list, dim3(min(nboxes, 1024)), dim3(64), 0, (hipStream_t)stream, d_data, d_mask, d_objmask,
                           ny, nx, box, nbx, nboxes, limfrac, d_mini_med, d_mini_std, nf, flist, nf_next);
    }
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_mini_fill_filter(bbx_ctx* ctx, int nby, int nbx, float* d_mini, void* stream) {
    if (!ctx || !d_mini || nby < 1 || nbx < 1 || (size_t)nby * nbx > (1u << 22)) return BBX_ERR_ARG;
    int rc;
    float* tmp = (float*)bbx_ws(ctx, WS_MISC, (size_t)nby * nbx * 4 + 256, &rc); if (rc) return rc;
    const size_t n = (size_t)nby * nbx;
    if (n <= MINI_LDS_MAX) {
        static bool attr_set = false;                        // (idempotent: a race only sets it twice)
        if (!attr_set) {
            BBX_HIP(hipFuncSetAttribute((const void*)k_mini_fill_filter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, MINI_LDS_MAX * 4));
            attr_set = true;
        }
        hipLaunchKernelGGL(k_mini_fill_filter_lds, dim3(1), dim3(1024), n * sizeof(float), (hipStream_t)stream, d_mini, nby, nbx);
        BBX_LAUNCH_CHECK();
        return BBX_OK;
    }
    hipLaunchKernelGGL(k_mini_fill_filter, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_mini, tmp + 64, nby, nbx, ctx->d_err);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_spline_zoom(bbx_ctx* ctx, int ny, int nx, const double* d_coef, int cny, int cnx, const int32_t* d_fy,
                    const double* d_wy, const int32_t* d_fx, const double* d_wx, float* d_data, float* d_bkg,
                    void* stream) {
    if (!ctx || !d_coef || !d_fy || !d_wy || !d_fx || !d_wx || (!d_data && !d_bkg) || ny < 1 || nx < 1 || cny < 4 || cnx < 4)
        return BBX_ERR_ARG;
    if (nx % 4 == 0 && (((uintptr_t)d_data | (uintptr_t)d_bkg) & 15) == 0)
        hipLaunchKernelGGL(k_spline_zoom4, dim3((nx + 1023) / 1024, (ny + ZOOM_ROWS - 1) / ZOOM_ROWS), dim3(256), 0, (hipStream_t)stream, ny, nx, d_coef,
                           cnx, d_fy, d_wy, d_fx, d_wx, d_data, d_bkg, (const float*)nullptr, (const float*)nullptr, 0.0, (uint32_t*)nullptr,
                           (int32_t*)nullptr, 0u, ctx->d_err);
    else
        hipLaunchKernelGGL(k_spline_zoom, dim3((nx + 255) / 256, (ny + ZOOM_ROWS - 1) / ZOOM_ROWS), dim3(256), 0, (hipStream_t)stream, ny, nx, d_coef, cnx,
                           d_fy, d_wy, d_fx, d_wx, d_data, d_bkg, (const float*)nullptr);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_spline_zoom_sub(bbx_ctx* ctx, int ny, int nx, const double* d_coef, int cny, int cnx, const int32_t* d_fy,
                        const double* d_wy, const int32_t* d_fx, const double* d_wx, const float* d_in, float* d_out,
                        void* stream) {
    if (!ctx || !d_coef || !d_fy || !d_wy || !d_fx || !d_wx || !d_in || !d_out || ny < 1 || nx < 1 || cny < 4 || cnx < 4)
        return BBX_ERR_ARG;
    // catalogue candidates on request (bbx_zoom_candidates, one-shot): listed by the kernel that writes the subtracted frame
    const float* cmed = ctx->bcand_med; const double cnsig = ctx->bcand_nsig;
    ctx->bcand_med = nullptr; ctx->bcand_img = nullptr;
    if (nx % 4 == 0 && (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0) {
        uint32_t* clist = nullptr; uint32_t ccap = 0; int32_t* ccnt = &ctx->d_counters[CNT_BCAND];
        if (cmed && (size_t)ny * nx < 0xffffffffull) {
            int rc;
            ccap = (uint32_t)((size_t)ny * nx / 16 + 1024);
            clist = (uint32_t*)bbx_ws(ctx, WS_BCAND, (size_t)ccap * sizeof(uint32_t), &rc); if (rc) return rc;
            BBX_HIP(hipMemsetAsync(ccnt, 0, sizeof(int32_t), (hipStream_t)stream));
        }
        hipLaunchKernelGGL(k_spline_zoom4, dim3((nx + 1023) / 1024, (ny + ZOOM_ROWS - 1) / ZOOM_ROWS), dim3(256), 0, (hipStream_t)stream, ny, nx, d_coef,
                           cnx, d_fy, d_wy, d_fx, d_wx, d_out, (float*)nullptr, d_in, cmed, cnsig, clist, ccnt, ccap, ctx->d_err);
        if (clist) { ctx->bcand_img = d_out; ctx->bcand_img_med = cmed; ctx->bcand_img_nsig = cnsig; ctx->bcand_npix = (size_t)ny * nx; }
    } else
        hipLaunchKernelGGL(k_spline_zoom, dim3((nx + 255) / 256, (ny + ZOOM_ROWS - 1) / ZOOM_ROWS), dim3(256), 0, (hipStream_t)stream, ny, nx, d_coef, cnx,
                           d_fy, d_wy, d_fx, d_wx, d_out, (float*)nullptr, d_in);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_spline_prefilter(bbx_ctx* ctx, int nby, int nbx, int cy, int cx, int npad, double zn_y, double zn_x, const float* d_mini,
                         double* d_coef, void* stream) {
    if (!ctx || !d_mini || !d_coef || nby < 1 || nbx < 1 || cy < 1 || cx < 1 || npad < 0 || nby % cy || nbx % cx) return BBX_ERR_ARG;
    const int py = cy + 2 * npad, px = cx + 2 * npad;
    if (py > SPF_MAXLEN || px > SPF_MAXLEN) return BBX_ERR_ARG;
    const int nblky = nby / cy, nblkx = nbx / cx, cnx = nblkx * px;
    const size_t l0 = (size_t)py * (SPF_LINES + 1) * sizeof(double), l1 = (size_t)SPF_LINES * (px + 1) * sizeof(double);
    if (ctx->spf_attr_bytes < (int)(l0 > l1 ? l0 : l1)) {
        BBX_HIP(hipFuncSetAttribute((const void*)k_spf_axis0, hipFuncAttributeMaxDynamicSharedMemorySize, SPF_MAXLEN * (SPF_LINES + 1) * 8));
        BBX_HIP(hipFuncSetAttribute((const void*)k_spf_axis1, hipFuncAttributeMaxDynamicSharedMemorySize, SPF_LINES * (SPF_MAXLEN + 1) * 8));
        ctx->spf_attr_bytes = SPF_MAXLEN * (SPF_LINES + 1) * 8;
    }
    hipLaunchKernelGGL(k_spf_axis0, dim3(nblky * nblkx * ((px + SPF_LINES - 1) / SPF_LINES)), dim3(256), l0, (hipStream_t)stream, d_mini, nbx, cy, cx,
                       npad, nblkx, zn_y, d_coef, cnx);
    hipLaunchKernelGGL(k_spf_axis1, dim3(nblky * nblkx * ((py + SPF_LINES - 1) / SPF_LINES)), dim3(256), l1, (hipStream_t)stream, cy, cx, npad, nblkx,
                       zn_x, d_coef, cnx);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_mini_median(bbx_ctx* ctx, int n, const float* d_a, float* d_med, void* stream) {
    if (!ctx || !d_a || !d_med || n < 1 || n > (1 << 24)) return BBX_ERR_ARG;
    if (n <= 32768) hipLaunchKernelGGL(k_mini_median_regs, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_a, n, d_med);
    else hipLaunchKernelGGL(k_mini_median, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_a, n, d_med);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_zoom_candidates(bbx_ctx* ctx, const float* d_med, double nsigma) {
    if (!ctx || (d_med && !(nsigma > 0.0))) return BBX_ERR_ARG;
    ctx->bcand_med = d_med; ctx->bcand_nsig = nsigma; ctx->bcand_img = nullptr;
    return BBX_OK;
}

}  // extern "C"

// bbx_build_flags (bbx_ctx.hip): any timing knock-out of this file compiled in?
int bbx_build_flags_bkg(void) {
#if defined(BOXK_NOBR) || defined(BOXK_NOCLASS) || defined(BOXK_NOFAIL) || defined(BOXK_NOSAMP)
    return 4;
#else
    return 0;
#endif
}
