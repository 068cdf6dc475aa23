// bbx_lacosmic.hip -- LA-Cosmic (astroscrappy.detect_cosmics as called by
// cosmics_corr, blackbox.py:4323-4332: sepmed=False, cleantype='medmask',
// fsmode='median', gain=1, satlevel=inf) restructured for the GPU.
//
// The textbook algorithm runs two 5x5, one 3x3 and one 7x7 median filter over all
// 111 Mpx in each of the 3 iterations (VALU/LDS-bound, ~6e11 compare-exchanges).
// This implementation is exact but sparse (hard part 1):
//
//   sp = s - medfilt5(s) <= s = L+/(2 noise)   and   noise >= sqrt(float32(rn*rn))
//   => a pixel can only pass  sp > sigclip  if  L+ > T = 2*sigclip*sqrt(rn^2)
//
// so per iteration
//   1. k_lac_cand   (dense, HBM-bound: one read of the frame) computes L+ with a
//      5-point stencil and queues the pixels with L+ > T;
//   2. k_lac_seed   one wave per queued pixel evaluates sp and the fine-structure
//      ratio exactly (lanes = the 25 / 49 neighbourhood positions, per-lane
//      register median networks + a wave-wide bitonic sort);
//   3. k_lac_grow1/2 the two 3x3 growth steps on the sparse flag plane; the second
//      needs sp at neighbours that are not queued and evaluates it on demand;
//   4. k_lac_clean  masked 5x5 median at every pixel of the cumulative CR list.
// All arithmetic is float32 in the order fixed by oracle/lacosmic.py; build with
// -ffp-contract=off.  Results are bit-identical to the dense algorithm.
#include "bbx_bsel.h"
#include "bbx_mednet.h"

int bbx_cc_count_list(bbx_ctx* ctx, const uint32_t* d_list, const int32_t* d_cnt, size_t cap, int ny, int nx,
                      int32_t* d_out, hipStream_t s);

#define F_STAGE1 1u    // sp > sigclip & good & sp/f > objlim
#define F_HI     2u    // sp > sigclip & good
#define F_STAGE2 4u    // after first growth
#define F_SEEN  16u    // second growth: sp already evaluated for this pixel

struct lac_par {
    int ny, nx;
    float sigclip, sigcliplow, objlim;
    const float* rnp;            // device: {float32(rn*rn), prune threshold T}
};

// L+ of pixel (j,i): 2x2 replicate -> Laplacian -> clip -> 2x2 mean, closed form with
// the evaluation order of oracle/lacosmic.py (p = 4c; -= right; -= left; -= down; -= up)
__device__ __forceinline__ float lplus_px(float c, float u, float dn, float l, float r, bool hu, bool hd, bool hl,
                                          bool hr) {
    const float c4 = 4.0f * c;
    // tl: right = c, left = l?, down = c, up = u?
    float tl = c4 - c; if (hl) tl -= l; tl -= c; if (hu) tl -= u;
    // tr: right = r?, left = c, down = c, up = u?
    float tr = c4; if (hr) tr -= r; tr -= c; tr -= c; if (hu) tr -= u;
    // bl: right = c, left = l?, down = d?, up = c
    float bl = c4 - c; if (hl) bl -= l; if (hd) bl -= dn; bl -= c;
    // br: right = r?, left = c, down = d?, up = c
    float br = c4; if (hr) br -= r; br -= c; if (hd) br -= dn; br -= c;
    tl = fmaxf(tl, 0.f); tr = fmaxf(tr, 0.f); bl = fmaxf(bl, 0.f); br = fmaxf(br, 0.f);
    return (((tl + tr) + bl) + br) * 0.25f;
}

__device__ __forceinline__ float lplus_at(const float* __restrict__ a, int j, int i, int ny, int nx) {
    const size_t o = (size_t)j * nx + i;
    const bool hu = j > 0, hd = j < ny - 1, hl = i > 0, hr = i < nx - 1;
    const float c = a[o];
    const float u = hu ? a[o - nx] : 0.f, dn = hd ? a[o + nx] : 0.f;
    const float l = hl ? a[o - 1] : 0.f, r = hr ? a[o + 1] : 0.f;
    return lplus_px(c, u, dn, l, r, hu, hd, hl, hr);
}

// ---- 1. dense candidate pass ------------------------------------------------------
// One read of the frame.  Block = 256 threads x 4 columns, walking CAND_ROWS rows with a
// 3-row register window; left/right neighbours come from the adjacent lanes; the loads of
// CAND_B rows are issued together to keep enough bytes in flight.  With FEED the same pass
// feeds the bracketed select of the background level (reads the mask too; dynamic LDS =
// sizeof(bsel_lds)).
#define CAND_ROWS 32
#define CAND_B 4
#define CAND_Q 8        // per-thread staging slots for candidate indices (LDS)
#define FEED_Q 16       // per-thread staging slots for in-bracket values (LDS)

// exclusive prefix sum of one value per thread over a 256-thread block
__device__ __forceinline__ unsigned block_excl_scan256(unsigned v, unsigned* wsum, unsigned* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    unsigned off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const unsigned x = wsum[w]; if (w < wid) off += x; tot += x; }
    *total = tot;
    __syncthreads();
    return off + incl - v;
}

// No atomic sits in the row loop: hits are staged in per-thread LDS slots (slot-major, no
// bank conflicts) and compacted once per block with ONE global atomic per list; a thread
// that overflows its slots appends directly (rare: dense star cores / degenerate brackets).
template <bool FEED>
__global__ __launch_bounds__(256) void k_lac_cand_v4(const float* __restrict__ a, const uint8_t* __restrict__ mask,
                                                     lac_par p, uint32_t* __restrict__ cand, int32_t* counters,
                                                     uint32_t cap, int32_t* err, bsel_dev b) {
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    uint32_t* lcand = reinterpret_cast<uint32_t*>(dyn_lds);                       // [CAND_Q][256]
    float* lfeed = reinterpret_cast<float*>(dyn_lds + CAND_Q * 256 * 4);          // [FEED_Q][256]
    __shared__ unsigned wsum[4];
    __shared__ unsigned gbase[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const float T = p.rnp[1];
    const int x0 = (blockIdx.x * 256 + tid) * 4;
    const bool act = x0 < p.nx;                                 // nx % 4 == 0 on this path
    const int j0 = blockIdx.y * CAND_ROWS;
    const int j1 = min(j0 + CAND_ROWS, p.ny);
    // loads are unconditional on clamped (always valid) addresses and masked afterwards: a
    // `cond ? *ptr : zero` select makes hipcc pick between a global and a private pointer and
    // fall back to scalar flat loads
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int xc = act ? x0 : 0;
    float4 up = *(const float4*)(a + (size_t)max(j0 - 1, 0) * p.nx + xc);
    float4 cur = *(const float4*)(a + (size_t)j0 * p.nx + xc);
    if (!(act && j0 > 0)) up = zero;
    if (!act) cur = zero;
    float lo = 0.f, hi = 0.f;
    unsigned ncand = 0, nfeed = 0, nvalid = 0, nbelow = 0;
    if (FEED) { lo = b.seg[0].lo; hi = b.seg[0].hi; }
    bool colok[4];
#pragma unroll
    for (int q = 0; q < 4; q++) colok[q] = act && (x0 + q >= 2) && (x0 + q < p.nx - 2);
    struct batch { float4 nxt[CAND_B]; float le[CAND_B], re[CAND_B]; uchar4 mk[CAND_B]; };
    // loads of a batch of CAND_B rows (clamped addresses, masked values)
    auto load_batch = [&](batch& t, int jb) {
#pragma unroll
        for (int k = 0; k < CAND_B; k++) {
            const int j = jb + k;
            const int jn = min(j + 1, p.ny - 1), jc = min(j, p.ny - 1);
            t.nxt[k] = *(const float4*)(a + (size_t)jn * p.nx + xc);
            if (!(act && j + 1 < p.ny)) t.nxt[k] = zero;
            // one lane per wave fetches the pixel left / right of the wave's span
            const int xe = (lane == 0) ? max(x0 - 1, 0) : min(x0 + 4, p.nx - 1);
            float e = 0.f;
            if ((lane == 0 || lane == 63) && act) e = a[(size_t)jc * p.nx + xe];
            t.le[k] = (lane == 0 && x0 > 0) ? e : 0.f;
            t.re[k] = (lane == 63 && x0 + 4 < p.nx) ? e : 0.f;
            t.mk[k] = make_uchar4(0, 0, 0, 0);
            if (FEED) { t.mk[k] = *(const uchar4*)(mask + (size_t)jc * p.nx + xc); }
        }
    };
    auto compute_batch = [&](const batch& t, int jb) {
#pragma unroll
        for (int k = 0; k < CAND_B; k++) {
            const int j = jb + k;
            if (j >= j1) break;                                  // block-uniform
            const float4 dn = t.nxt[k];
            float l = __shfl_up(cur.w, 1, 64), r = __shfl_down(cur.x, 1, 64);
            if (lane == 0) l = t.le[k];
            if (lane == 63) r = t.re[k];
            // the 2-pixel frame of the image never holds candidates (sp == 0 there), so every
            // tested pixel has all four neighbours: interior form of L+
            const bool rowok = (j >= 2 && j < p.ny - 2);
            const float c4[4] = {cur.x, cur.y, cur.z, cur.w};
            const float u4[4] = {up.x, up.y, up.z, up.w};
            const float d4[4] = {dn.x, dn.y, dn.z, dn.w};
            const float l4[4] = {l, cur.x, cur.y, cur.z};
            const float r4[4] = {cur.y, cur.z, cur.w, r};
            const uint8_t mm[4] = {t.mk[k].x, t.mk[k].y, t.mk[k].z, t.mk[k].w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // cheap exact-safe reject: L+ is the mean of four clipped half-Laplacians, hence
                // <= their maximum = 2c - min(u,d) - min(l,r); the float32 roundings of either
                // form stay below 1e-5*|4c|, which the slack term covers
                const float c = c4[q];
                const float quick = (c + c) - fminf(u4[q], d4[q]) - fminf(l4[q], r4[q]);
                if (rowok && colok[q] && quick > T - 4e-5f * fabsf(c)) {
                    const float lp = lplus_px(c, u4[q], d4[q], l4[q], r4[q], true, true, true, true);
                    if (lp > T) {
                        const uint32_t idx = (uint32_t)((size_t)j * p.nx + x0 + q);
                        if (ncand < CAND_Q) { lcand[ncand * 256 + tid] = idx; ncand++; }
                        else {
                            const unsigned kk = atomicAdd((unsigned*)&counters[CNT_CAND], 1u);
                            if (kk < cap) cand[kk] = idx; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
                        }
                    }
                }
                if (FEED) {
                    const bool valid = act && !(mm[q] & ~BBX_MASK_COSMIC);
                    const bool below = valid && c < lo;
                    const bool inb = valid && c >= lo && c <= hi;
                    nvalid += valid ? 1u : 0u;
                    nbelow += below ? 1u : 0u;
                    if (inb) {
                        if (nfeed < FEED_Q) { lfeed[nfeed * 256 + tid] = c; nfeed++; }
                        else {
                            const unsigned kk = atomicAdd(&b.seg[0].nbuf, 1u);
                            if (kk < b.cap) b.buf[kk] = c;
                        }
                    }
                }
            }
            up = cur; cur = dn;
        }
    };
    // software pipeline: the loads of the next batch are in flight while this one is computed
    batch bA, bB;
    load_batch(bA, j0);
    for (int jb = j0; jb < j1; jb += 2 * CAND_B) {
        load_batch(bB, jb + CAND_B);
        compute_batch(bA, jb);
        load_batch(bA, jb + 2 * CAND_B);
        compute_batch(bB, jb + CAND_B);
    }
    // ---- compaction: one global atomic per list and block
    unsigned total;
    unsigned off = block_excl_scan256(ncand, wsum, &total);
    if (tid == 0) gbase[0] = total ? atomicAdd((unsigned*)&counters[CNT_CAND], total) : 0u;
    __syncthreads();
    for (unsigned k = 0; k < ncand; k++) {
        const unsigned pos = gbase[0] + off + k;
        if (pos < cap) cand[pos] = lcand[k * 256 + tid]; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
    }
    if (FEED) {
        off = block_excl_scan256(nfeed, wsum, &total);
        if (tid == 0) gbase[1] = total ? atomicAdd(&b.seg[0].nbuf, total) : 0u;
        __syncthreads();
        for (unsigned k = 0; k < nfeed; k++) {
            const unsigned pos = gbase[1] + off + k;
            if (pos < b.cap) b.buf[pos] = lfeed[k * 256 + tid];
        }
        bsel_acc acc = {nvalid, nbelow};
        bsel_flush(b, 0, acc);
    }
}

// scalar variant for frames whose width is not a multiple of 4 (small test frames)
template <bool FEED>
__global__ __launch_bounds__(256) void k_lac_cand_s(const float* __restrict__ a, const uint8_t* __restrict__ mask,
                                                    lac_par p, uint32_t* __restrict__ cand, int32_t* counters,
                                                    uint32_t cap, int32_t* err, bsel_dev b) {
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    bsel_lds& L = *reinterpret_cast<bsel_lds*>(dyn_lds);
    const size_t npix = (size_t)p.ny * p.nx;
    const float T = p.rnp[1];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nend = ((npix + stride - 1) / stride) * stride;      // every thread runs the same trip count
    float lo = 0.f, hi = 0.f;
    bsel_acc acc = {0, 0};
    if (FEED) { lo = b.seg[0].lo; hi = b.seg[0].hi; bsel_lds_init(L); }
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < nend; o += stride) {
        const bool in = o < npix;
        const int j = in ? (int)(o / p.nx) : 0, i = in ? (int)(o - (size_t)j * p.nx) : 0;
        // the outer 2-pixel frame has sp == 0 (median filter copies its border)
        if (in && !(j < 2 || i < 2 || j >= p.ny - 2 || i >= p.nx - 2)) {
            const float lp = lplus_at(a, j, i, p.ny, p.nx);
            if (lp > T) {
                const unsigned k = atomicAdd((unsigned*)&counters[CNT_CAND], 1u);
                if (k < cap) cand[k] = (uint32_t)o; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
            }
        }
        if (FEED) {
            bsel_feed(L, lo, hi, in ? a[o] : 0.f, in && !(mask[o] & ~BBX_MASK_COSMIC), acc);
            bsel_drain(b, 0, L, 256u, o + stride >= nend);
        }
    }
    if (FEED) bsel_flush(b, 0, acc);
}

// ---- wave-cooperative evaluation ----------------------------------------------------
__device__ __forceinline__ float wave_sort_pick(float v, int width, int rank) {
    // ascending bitonic sort of one value per lane over the first [width] lanes
    // (width = 32 or 64, higher lanes must hold +inf), returns the element of [rank]
    const int lane = threadIdx.x & 63;
    for (int k = 2; k <= width; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            const float o = __shfl_xor(v, jj, 64);
            const bool up = ((lane & k) == 0);
            const bool lower = ((lane & jj) == 0);
            v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
        }
    }
    return __shfl(v, rank, 64);
}

// s(q) = L+(q) / (2 * noise(q)) for q = (j,i) anywhere in the image; also returns noise
__device__ __forceinline__ float s_at(const float* __restrict__ a, int j, int i, const lac_par& p, float* noise_out) {
    float m5;
    if (j < 2 || i < 2 || j >= p.ny - 2 || i >= p.nx - 2) {
        m5 = a[(size_t)j * p.nx + i];                       // median filter border = input
    } else {
        float v[25];
#pragma unroll
        for (int dy = 0; dy < 5; dy++)
#pragma unroll
            for (int dx = 0; dx < 5; dx++) v[dy * 5 + dx] = a[(size_t)(j + dy - 2) * p.nx + (i + dx - 2)];
        BBX_MED25(v);
        m5 = v[12];
    }
    m5 = fmaxf(m5, 0.00001f);
    const float noise = sqrtf(m5 + p.rnp[0]);
    *noise_out = noise;
    return lplus_at(a, j, i, p.ny, p.nx) / (2.0f * noise);
}

// sp at pixel (j,i); all 64 lanes of the wave call this with the same (j,i).
// Returns sp (wave-uniform) and the noise at the pixel.
__device__ __forceinline__ float sp_wave(const float* __restrict__ a, int j, int i, const lac_par& p, float* noise_c) {
    const int lane = threadIdx.x & 63;
    float sv = __builtin_huge_valf(), nz = 0.f;
    if (lane < 25) {
        const int dy = lane / 5 - 2, dx = lane % 5 - 2;
        sv = s_at(a, j + dy, i + dx, p, &nz);
    }
    const float s_c = __shfl(sv, 12, 64);
    *noise_c = __shfl(nz, 12, 64);
    const float med = wave_sort_pick(sv, 32, 12);
    return s_c - med;
}

// fine structure f at (j,i) given the noise there (wave-uniform call)
__device__ __forceinline__ float f_wave(const float* __restrict__ a, int j, int i, const lac_par& p, float noise) {
    const int lane = threadIdx.x & 63;
    float m3 = __builtin_huge_valf();
    const bool inner7 = !(j < 3 || i < 3 || j >= p.ny - 3 || i >= p.nx - 3);
    if (inner7) {
        if (lane < 49) {
            const int qj = j + lane / 7 - 3, qi = i + lane % 7 - 3;
            if (qj < 1 || qi < 1 || qj >= p.ny - 1 || qi >= p.nx - 1) {
                m3 = a[(size_t)qj * p.nx + qi];
            } else {
                float v[9];
#pragma unroll
                for (int dy = 0; dy < 3; dy++)
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) v[dy * 3 + dx] = a[(size_t)(qj + dy - 1) * p.nx + (qi + dx - 1)];
                BBX_MED9(v);
                m3 = v[4];
            }
        }
        const float m3c = __shfl(m3, 24, 64);
        const float med = wave_sort_pick(m3, 64, 24);
        float f = (m3c - med) / noise;
        if (f < 0.01f) f = 0.01f;
        return f;
    }
    // inside the 3-pixel frame medfilt7 returns its input: f = (m3 - m3)/noise
    float f = 0.0f / noise;
    if (f < 0.01f) f = 0.01f;
    return f;
}

__device__ __forceinline__ bool good_px(const uint8_t* mask, size_t o) { return (mask[o] & ~BBX_MASK_COSMIC) == 0; }

// ---- 2. seeds: one wave per candidate ---------------------------------------------------
__global__ __launch_bounds__(256) void k_lac_seed(const float* __restrict__ a, const uint8_t* __restrict__ mask, lac_par p,
                                                  const uint32_t* __restrict__ cand, const int32_t* __restrict__ counters,
                                                  uint32_t cap, uint8_t* __restrict__ flags) {
    const int n = min((uint32_t)counters[CNT_CAND], cap);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int k = wave; k < n; k += nwaves) {
        const uint32_t o = cand[k];
        const int j = o / p.nx, i = o - j * p.nx;
        if (!good_px(mask, o)) continue;                      // wave-uniform
        float noise;
        const float sp = sp_wave(a, j, i, p, &noise);
        if (!(sp > p.sigclip)) continue;
        const float f = f_wave(a, j, i, p, noise);
        const unsigned fl = F_HI | ((sp / f > p.objlim) ? F_STAGE1 : 0u);
        if ((threadIdx.x & 63) == 0) flags[o] = (uint8_t)fl;
    }
}

// ---- 3a. first growth: dilate3(stage1) & good & sp > sigclip ---------------------------------
__global__ __launch_bounds__(256) void k_lac_grow1(lac_par p, const uint32_t* __restrict__ cand,
                                                   const int32_t* __restrict__ counters_ro, uint32_t cap, uint8_t* flags,
                                                   uint32_t* __restrict__ stage2, int32_t* counters, int32_t* err) {
    const int n = min((uint32_t)counters_ro[CNT_CAND], cap);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t o = cand[k];
        const unsigned fl = flags[o];
        if (!(fl & F_HI)) continue;
        const int j = o / p.nx, i = o - j * p.nx;
        bool any = false;
        // candidates lie >= 2 px inside the frame, so the 3x3 window is always interior
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) any |= (flags[(size_t)(j + dy) * p.nx + (i + dx)] & F_STAGE1) != 0;
        if (any) {
            flags[o] = (uint8_t)(fl | F_STAGE2);
            const unsigned q = atomicAdd((unsigned*)&counters[CNT_STAGE2], 1u);
            if (q < cap) stage2[q] = o; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
        }
    }
}

// ---- 3b. second growth: dilate3(stage2) & good & sp > sigcliplow -----------------------------
// One wave per stage-2 pixel p evaluates all 9 neighbours at once: lanes 0..48 hold s on the
// 7x7 block around p (each lane one 5x5 median of the image), then lanes 0..8 gather the 25
// s values of "their" neighbour's window with wave shuffles and run the median network in
// registers.  Neighbours shared by adjacent stage-2 pixels are simply evaluated again (no
// claim/atomic on the critical path); the F_FINAL bit makes the per-iteration count unique.
#define F_FINAL 8u
__global__ __launch_bounds__(256) void k_lac_grow2(const float* __restrict__ a, uint8_t* mask, lac_par p,
                                                   const uint32_t* __restrict__ stage2, uint32_t cap, uint8_t* flags,
                                                   uint32_t* __restrict__ crlist, int32_t* counters, int32_t* err) {
    const int n = min((uint32_t)counters[CNT_STAGE2], cap);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (int w = wave; w < n; w += nwaves) {
        const uint32_t o = stage2[w];
        const int j = (int)(o / p.nx), i = (int)(o % p.nx);      // >= 2 px inside the frame
        // s on the 7x7 block (positions outside the image are never read back)
        float sv = 0.f, nz;
        if (lane < 49) {
            const int qj = j + lane / 7 - 3, qi = i + lane % 7 - 3;
            if (qj >= 0 && qi >= 0 && qj < p.ny && qi < p.nx) sv = s_at(a, qj, qi, p, &nz);
        }
        // neighbour handled by this lane (lanes 0..8)
        const int ry = (lane < 9 ? lane / 3 : 1) - 1, rx = (lane < 9 ? lane % 3 : 1) - 1;
        const int rj = j + ry, ri = i + rx;
        float v[25];
#pragma unroll
        for (int e = 0; e < 25; e++) {
            const int src = (ry + e / 5 - 2 + 3) * 7 + (rx + e % 5 - 2 + 3);
            v[e] = __shfl(sv, src, 64);
        }
        const float s_c = v[12];
        // sp == 0 inside the 2-pixel frame (median filter copies its border): cannot pass
        const bool inner = !(rj < 2 || ri < 2 || rj >= p.ny - 2 || ri >= p.nx - 2);
        if (lane < 9 && inner) {
            const size_t r = (size_t)rj * p.nx + ri;
            if (good_px(mask, r)) {
                BBX_MED25(v);
                const float sp = s_c - v[12];
                if (sp > p.sigcliplow) {
                    const unsigned of = atomic_or_u8(flags, r, F_FINAL);
                    if (!(of & F_FINAL)) {
                        const unsigned om = atomic_or_u8(mask, r, BBX_MASK_COSMIC);
                        atomicAdd(&counters[CNT_NEWCR], 1);
                        if (!(om & BBX_MASK_COSMIC)) {
                            const unsigned q = atomicAdd((unsigned*)&counters[CNT_CRLIST], 1u);
                            if (q < cap) crlist[q] = (uint32_t)r; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
                        }
                    }
                }
            }
        }
    }
}

// ---- 4. clean_medmask on the cumulative CR list -------------------------------------------------
__global__ __launch_bounds__(256) void k_lac_clean(float* a, const uint8_t* __restrict__ mask, lac_par p,
                                                   const uint32_t* __restrict__ crlist, const int32_t* __restrict__ counters,
                                                   uint32_t cap, const bsel_seg* __restrict__ bg) {
    const int n = min((uint32_t)counters[CNT_CRLIST], cap);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (int k = wave; k < n; k += nwaves) {
        const uint32_t o = crlist[k];
        const int j = o / p.nx, i = o - j * p.nx;
        if (j < 2 || i < 2 || j >= p.ny - 2 || i >= p.nx - 2) continue;
        float v = __builtin_huge_valf();
        int ok = 0;
        if (lane < 25) {
            const size_t q = (size_t)(j + lane / 5 - 2) * p.nx + (i + lane % 5 - 2);
            if (mask[q] == 0) { v = a[q]; ok = 1; }           // neither CR nor masked
        }
        const int cnt = __popcll(__ballot(ok));
        // +inf padding sorts last; a genuine +inf pixel would too, and is then picked in order
        const float med = wave_sort_pick(v, 32, cnt > 0 ? (cnt - 1) / 2 : 0);
        if (lane == 0) a[o] = (cnt > 0) ? med : bg->result[0];
    }
}

__global__ void k_lac_iter_end(int32_t* counters, int32_t* stats, int it) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stats[it] = counters[CNT_NEWCR];
        stats[7] = counters[CNT_CRLIST];
        if (it < 4) { stats[8 + 2 * it] = counters[CNT_CAND]; stats[9 + 2 * it] = counters[CNT_STAGE2]; }
        counters[CNT_NEWCR] = 0; counters[CNT_CAND] = 0; counters[CNT_STAGE2] = 0;
    }
}

// readnoise -> {rn2, T} on the device.  With d_rdn16 the read noise is the float32 image of
// np.nanmean of the 16 channel sigmas (header RDNOISE, blackbox.py:6867) evaluated in numpy's
// pairwise order for 16 elements, so no host round trip is needed between os_corr and here.
__global__ void k_lac_rn(float readnoise, const double* __restrict__ rdn16, float sigclip, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float rn = readnoise;
    if (rdn16) {
        double r[8]; int n = 0;
        for (int i = 0; i < 8; i++) {
            const double a = rdn16[i], b = rdn16[i + 8];
            r[i] = ((a == a) ? a : 0.0) + ((b == b) ? b : 0.0);
            n += (a == a) + (b == b);
        }
        const double sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        rn = (float)(sum / (double)n);
    }
    const float rn2 = rn * rn;
    out[0] = rn2;
    // prune threshold: sp <= L+/(2*sqrtf(rn2)); the (1 - 1e-5) factor absorbs the float32
    // roundings of the division and of 2*noise (LA-Cosmic)
    out[1] = 2.0f * sigclip * sqrtf(rn2) * (1.0f - 1e-5f);
    out[2] = rn;
}

extern "C" int bbx_lacosmic(bbx_ctx* ctx, int ny, int nx, float* d_data, uint8_t* d_mask, float sigclip,
                            float sigfrac, float objlim, int niter, float readnoise, const double* d_rdn16,
                            int32_t* d_stats, void* stream) {
    if (!ctx || !d_data || !d_mask || !d_stats || ny < 8 || nx < 8 || niter < 0 || niter > 6) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t npix = (size_t)ny * nx;
    if (npix >= 0xffffffffull || ((uintptr_t)d_mask) % 4) return BBX_ERR_ARG;
    int rc;
    lac_par p;
    p.ny = ny; p.nx = nx; p.sigclip = sigclip; p.sigcliplow = sigfrac * sigclip; p.objlim = objlim;
    float* rnp = (float*)bbx_ws(ctx, WS_MISC, 64, &rc); if (rc) return rc;
    p.rnp = rnp;
    hipLaunchKernelGGL(k_lac_rn, dim3(1), dim3(64), 0, s, readnoise, d_rdn16, sigclip, rnp);
    const size_t cap = npix / 4 + 4096;
    uint32_t* cand = (uint32_t*)bbx_ws(ctx, WS_CAND, cap * 4, &rc); if (rc) return rc;
    uint32_t* stage2 = (uint32_t*)bbx_ws(ctx, WS_STAGE2, cap * 4, &rc); if (rc) return rc;
    uint32_t* crlist = (uint32_t*)bbx_ws(ctx, WS_CRLIST, cap * 4, &rc); if (rc) return rc;
    uint8_t* flags = (uint8_t*)bbx_ws(ctx, WS_FLAGS, npix + 16, &rc); if (rc) return rc;
    int32_t* cnt = ctx->d_counters;
    BBX_HIP(hipMemsetAsync(d_stats, 0, 16 * sizeof(int32_t), s));
    BBX_HIP(hipMemsetAsync(&cnt[CNT_CAND], 0, 4 * sizeof(int32_t), s));      // CAND, STAGE2, CRLIST, NEWCR
    // background level of the unmasked input pixels (needed when a CR pixel has no good
    // neighbour): bracketed select fed by the first candidate pass, no extra read of the frame
    bsel_dev bs;
    rc = bbx_bsel_prepare(ctx, d_data, d_mask, ny, nx, ny, nx, &bs, s);
    if (rc) return rc;
    const unsigned gdense = 256u * 16u, gsparse = 256u * 8u;
    const bool vec = (nx % 4 == 0) && (((uintptr_t)d_data) % 16 == 0);
    const dim3 gvec((nx / 4 + 255) / 256, (ny + CAND_ROWS - 1) / CAND_ROWS);
    for (int it = 0; it < niter; it++) {
        BBX_HIP(hipMemsetAsync(flags, 0, npix, s));
        bbx_prof_start(ctx, BBX_PROF_LAC_DENSE, s);
        if (it == 0) {
            if (vec) hipLaunchKernelGGL(k_lac_cand_v4<true>, gvec, dim3(256), (CAND_Q + FEED_Q) * 256 * 4, s, d_data, d_mask, p, cand, cnt, (uint32_t)cap, ctx->d_err, bs);
            else hipLaunchKernelGGL(k_lac_cand_s<true>, dim3(gdense), dim3(256), sizeof(bsel_lds), s, d_data, d_mask, p, cand, cnt, (uint32_t)cap, ctx->d_err, bs);
            bbx_prof_stop(ctx, s);
            rc = bbx_bsel_finish(ctx, bs, d_data, d_mask, ny, nx, s);
            if (rc) return rc;
        } else {
            if (vec) hipLaunchKernelGGL(k_lac_cand_v4<false>, gvec, dim3(256), CAND_Q * 256 * 4, s, d_data, d_mask, p, cand, cnt, (uint32_t)cap, ctx->d_err, bs);
            else hipLaunchKernelGGL(k_lac_cand_s<false>, dim3(gdense), dim3(256), 0, s, d_data, d_mask, p, cand, cnt, (uint32_t)cap, ctx->d_err, bs);
            bbx_prof_stop(ctx, s);
        }
        bbx_prof_start(ctx, BBX_PROF_LAC_SPARSE, s);
        hipLaunchKernelGGL(k_lac_seed, dim3(gsparse), dim3(256), 0, s, d_data, d_mask, p, cand, cnt, (uint32_t)cap, flags);
        hipLaunchKernelGGL(k_lac_grow1, dim3(gsparse), dim3(256), 0, s, p, cand, cnt, (uint32_t)cap, flags, stage2, cnt, ctx->d_err);
        hipLaunchKernelGGL(k_lac_grow2, dim3(gsparse), dim3(256), 0, s, d_data, d_mask, p, stage2, (uint32_t)cap, flags, crlist,
                           cnt, ctx->d_err);
        hipLaunchKernelGGL(k_lac_clean, dim3(gsparse), dim3(256), 0, s, d_data, d_mask, p, crlist, cnt, (uint32_t)cap, bs.seg);
        hipLaunchKernelGGL(k_lac_iter_end, dim3(1), dim3(64), 0, s, cnt, d_stats, it);
        bbx_prof_stop(ctx, s);
    }
    BBX_LAUNCH_CHECK();
    // NCOSMICS: 8-connected objects of the CR pixels (blackbox.py:4354-4356)
    return bbx_cc_count_list(ctx, crlist, &cnt[CNT_CRLIST], cap, ny, nx, &d_stats[6], s);
}
