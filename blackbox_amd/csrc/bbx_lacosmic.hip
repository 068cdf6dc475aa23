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
#include <algorithm>
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
    int nwx, ntiles;             // dense pass: waves along x, tiles in all
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
// One read of the frame: 5-point stencil on a rolling 3-row register window, left/right
// neighbours from the adjacent lanes.  With FEED the same pass feeds the bracketed select
// of the background level (reads the mask too).
#ifndef CAND_ROWS
#define CAND_ROWS 16
#endif
#ifndef CAND_PF
#define CAND_PF 8       // rows of loads in flight per thread
#endif
#ifndef CAND_WPE
#define CAND_WPE 1
#endif
#define CAND_WQ 512      // per-wave LDS queue of candidate indices

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

// Lane layout: a wave spans 64 x 4 columns but only lanes 1..62 produce results (248
// columns); lanes 0 and 63 exist to hand their edge pixel to the neighbour lane (DPP wave
// shift), so the row loop has no edge loads and no lane-dependent branch.  Out-of-image
// lanes load from a clamped (valid) address and never produce.  Rows: rolling 3-row window
// over CAND_ROWS rows with a ring of CAND_PF rows of loads in flight per thread.
//
// L+ is evaluated exactly for every pixel (packed float32 math, the operation order of
// oracle/lacosmic.py): with T only ~2 sigma of the sky noise any cheaper upper bound of L+
// passes in almost every wave-row, so a pre-filter only adds work.
//
// No global atomic on the common path: candidates are queued per wave in LDS (they come in
// clusters: star cores) and written to the wave's own tile segment (tile_cnt/tile_seg);
// k_lac_compact turns the segments into the dense list.  A tile with more than CAND_TILECAP
// candidates appends the excess to an overflow list with one atomic per wave.
// FEED: in-bracket values go to a second per-wave LDS queue; the wave reserves side-buffer
// space with one atomic on its shard (bbx_bsel.h) and copies the queue out coalesced.
// One wave per workgroup: no barrier anywhere, waves retire independently.
#define CAND_SPAN 248
#define CAND_TILECAP 64
#define FEED_WQ 1024    // per-wave LDS queue of in-bracket values (FEED)

typedef float f2 __attribute__((ext_vector_type(2)));

// lane i <- lane i-1 / lane i+1 across the whole wave (DPP wave_shr:1 / wave_shl:1, one VALU
// op); lane 0 resp. 63 receive 0
__device__ __forceinline__ float lane_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
// max(x, 0) as the hardware does it (IEEE maxNum: NaN -> 0, like fmaxf), without the
// canonicalisation move clang puts in front of fmaxf
__device__ __forceinline__ float relu_hw(float x) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ f2 relu2(f2 x) { f2 r; r.x = relu_hw(x.x); r.y = relu_hw(x.y); return r; }

// L+ of two pixels at once; same operation order as lplus_px with all four neighbours
__device__ __forceinline__ f2 lplus2(f2 c, f2 u, f2 d, f2 l, f2 r) {
    const f2 c4 = c * 4.0f;
    const f2 a2 = (c4 - c) - l;            // left half:  4c - c(right replica) - l
    const f2 a4 = (c4 - r) - c;            // right half: 4c - r - c(left replica)
    const f2 tl = relu2((a2 - c) - u), bl = relu2((a2 - d) - c);
    const f2 tr = relu2((a4 - c) - u), br = relu2((a4 - d) - c);
    return (((tl + tr) + bl) + br) * 0.25f;
}

template <bool FEED>
__global__ __launch_bounds__(64, CAND_WPE) void k_lac_cand_v4(const float* __restrict__ a, const uint8_t* __restrict__ mask,
                                                     lac_par p, uint8_t* __restrict__ tile_cnt,
                                                     uint32_t* __restrict__ tile_seg, uint32_t* __restrict__ ovf,
                                                     int32_t* counters, uint32_t capovf, int32_t* err, bsel_dev b) {
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    uint32_t* lcand = reinterpret_cast<uint32_t*>(dyn_lds);                       // [CAND_WQ]
    const int lane = threadIdx.x;                                                 // one wave per workgroup
    float* wq = reinterpret_cast<float*>(dyn_lds + CAND_WQ * 4);              // queue of in-bracket values (FEED)
    const float T = p.rnp[1];
    // XCD-aware tile order: workgroups go round-robin to the 8 XCDs (each with its own L2), so
    // workgroup L works on tile (L % 8) * chunk + L / 8 -- every XCD sweeps one contiguous band
    // of the frame and finds the halo rows/columns shared with neighbouring tiles in its own L2
    const unsigned chunk = gridDim.x >> 3;
    const unsigned tile = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
    if (tile >= (unsigned)p.ntiles) return;
    const int bx = (int)(tile % (unsigned)p.nwx), by = (int)(tile / (unsigned)p.nwx);
    const int x0 = bx * CAND_SPAN - 4 + lane * 4;
    const bool prod = x0 >= 0 && x0 < p.nx && lane >= 1 && lane <= 62;      // nx % 4 == 0 on this path
    const int xc = min(max(x0, 0), p.nx - 4);
    const int j0 = by * CAND_ROWS;
    const int j1 = min(j0 + CAND_ROWS, p.ny);
    const float* __restrict__ col = a + xc;
    const uint8_t* __restrict__ mcol = mask + xc;
    const size_t nx = (size_t)p.nx;
    const int ylast = p.ny - 1;
    float4 up = *(const float4*)(col + (size_t)max(j0 - 1, 0) * nx);
    float4 cur = *(const float4*)(col + (size_t)j0 * nx);
    float4 ring[CAND_PF];
    uint32_t mring[CAND_PF];
#pragma unroll
    for (int k = 0; k < CAND_PF; k++) {
        ring[k] = *(const float4*)(col + (size_t)min(j0 + 1 + k, ylast) * nx);
        if (FEED) mring[k] = *(const uint32_t*)(mcol + (size_t)min(j0 + k, ylast) * nx);
    }
    float lo = 0.f, hi = 0.f;
    unsigned wcand = 0;                                        // wave-uniform
    unsigned wcount = 0;                                       // wave-uniform (FEED)
    unsigned lvalid = 0, lbelow = 0;                           // per lane (FEED)
    unsigned wbelow = 0;                                       // wave-uniform (FEED): rows without masked pixels
    const unsigned sh = bsel_my_shard();
    if (FEED) { lo = b.seg[0].lo; hi = b.seg[0].hi; }
    bool colok[4];
#pragma unroll
    for (int q = 0; q < 4; q++) colok[q] = prod && (x0 + q >= 2) && (x0 + q < p.nx - 2);
    // bits of a mask word that make a pixel invalid for the background statistics
    const uint32_t badbits = 0x01010101u * (uint32_t)(0xff & ~BBX_MASK_COSMIC);
    const float lo_l = prod ? lo : __builtin_nanf(""), hi_l = prod ? hi : __builtin_nanf("");
    const unsigned vrow = prod ? 4u : 0u;
    // fully unrolled (no loop back-edge): with a rolled loop the register renaming at the
    // back-edge makes hipcc wait for all outstanding loads once per batch
#pragma unroll
    for (int jbi = 0; jbi < CAND_ROWS / CAND_PF; jbi++) {
        const int jb = j0 + jbi * CAND_PF;
#pragma unroll
        for (int k = 0; k < CAND_PF; k++) {
            const int j = jb + k;
            const float4 dn = ring[k];
            uint32_t mk = 0;
            if (FEED) mk = mring[k];
            // refill the slot: row j+1+CAND_PF (and its mask row j+CAND_PF), clamped
            ring[k] = *(const float4*)(col + (size_t)min(j + 1 + CAND_PF, ylast) * nx);
            if (FEED) mring[k] = *(const uint32_t*)(mcol + (size_t)min(j + CAND_PF, ylast) * nx);
            if (j < j1) {                                        // block-uniform
                // the 2-pixel frame of the image never holds candidates (sp == 0 there), so every
                // tested pixel has all four neighbours: interior form of L+
                if (j >= 2 && j < p.ny - 2) {
                    const float l = lane_prev(cur.w), r = lane_next(cur.x);
                    const f2 lp01 = lplus2(f2{cur.x, cur.y}, f2{up.x, up.y}, f2{dn.x, dn.y}, f2{l, cur.x}, f2{cur.y, cur.z});
                    const f2 lp23 = lplus2(f2{cur.z, cur.w}, f2{up.z, up.w}, f2{dn.z, dn.w}, f2{cur.y, cur.z}, f2{cur.w, r});
                    const bool hit[4] = {colok[0] && lp01.x > T, colok[1] && lp01.y > T, colok[2] && lp23.x > T,
                                         colok[3] && lp23.y > T};
                    // wave-level queue in LDS: candidates come in clusters (star cores), so the
                    // staging is per wave, not per lane; wcand is wave-uniform
                    if (__builtin_amdgcn_ballot_w64(hit[0] || hit[1] || hit[2] || hit[3])) {
                        const uint32_t idx0 = (uint32_t)((size_t)j * nx + x0);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit[q]);
                            if (hit[q]) lcand[wcand + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = idx0 + q;
                            wcand += (unsigned)__popcll(m);
                        }
                        if (wcand + 256u > CAND_WQ) {            // no room for another row: spill the queue (rare)
                            unsigned base = 0;
                            if (lane == 0) base = atomicAdd((unsigned*)&counters[CNT_CANDOVF], wcand);
                            base = __shfl(base, 0, 64);
                            for (unsigned i = lane; i < wcand; i += 64) {
                                if (base + i < capovf) ovf[base + i] = lcand[i]; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
                            }
                            wcand = 0;
                        }
                    }
                }
                if (FEED) {
                    const float c4[4] = {cur.x, cur.y, cur.z, cur.w};
                    // room for this row's appends (at most 4 per lane)?  wave-uniform, rare
                    if (wcount + 256u > FEED_WQ) {
                        unsigned base = 0;
                        if (lane == 0) base = bsel_reserve(b, 0, sh, wcount);
                        base = __shfl(base, 0, 64);
                        float* reg = bsel_region(b, 0, sh);
                        for (unsigned i = lane; i < wcount; i += 64)
                            if (base + i < b.capS) reg[base + i] = wq[i];
                        wcount = 0;
                    }
                    // lanes that do not produce compare against NaN bounds (lo_l, hi_l): never
                    // below, never inside.  Rows without any masked pixel (wave-uniform test, the
                    // usual case) skip the per-pixel validity logic.
                    const uint32_t bad = mk & badbits;
                    if (__builtin_amdgcn_ballot_w64(bad != 0) == 0) {
                        lvalid += vrow;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const float c = c4[q];
                            // the compare masks live in scalar registers: counting and combining them
                            // is scalar-unit work, the vector unit only does the two compares
                            const bool below = c < lo_l;
                            const bool inb = !below && c <= hi_l;
                            const unsigned long long mb = __builtin_amdgcn_ballot_w64(below);
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(inb);
                            wbelow += (unsigned)__popcll(mb);
                            if (m) {
                                if (inb)
                                    wq[wcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = c;
                                wcount += (unsigned)__popcll(m);
                            }
                        }
                    } else {
                        // valid pixels of this lane in the row: 4 - (number of non-zero bytes of bad)
                        const uint32_t nzb = (((bad & 0x7f7f7f7fu) + 0x7f7f7f7fu) | bad) & 0x80808080u;
                        if (prod) lvalid += 4u - (unsigned)__popc(nzb);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const float c = c4[q];
                            const bool valid = (nzb & (0x80u << (8 * q))) == 0;
                            const bool below = valid && c < lo_l;
                            const bool inb = valid && !(c < lo_l) && c <= hi_l;
                            lbelow += below ? 1u : 0u;
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(inb);
                            if (inb) wq[wcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = c;
                            wcount += (unsigned)__popcll(m);
                        }
                    }
                }
            }
            up = cur; cur = dn;
        }
    }
    // ---- candidates -> this wave's tile segment; what does not fit goes to the overflow list
    const unsigned nseg = min(wcand, (unsigned)CAND_TILECAP);
    if (lane == 0) tile_cnt[tile] = (uint8_t)nseg;
    for (unsigned i = lane; i < nseg; i += 64) tile_seg[(size_t)tile * CAND_TILECAP + i] = lcand[i];
    if (wcand > nseg) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd((unsigned*)&counters[CNT_CANDOVF], wcand - nseg);
        base = __shfl(base, 0, 64);
        for (unsigned i = nseg + lane; i < wcand; i += 64) {
            const unsigned pos = base + i - nseg;
            if (pos < capovf) ovf[pos] = lcand[i]; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
        }
    }
    if (FEED) {
        // ---- in-bracket values: one reservation on the wave's shard, queue copied coalesced
        unsigned base = 0;
        if (lane == 0 && wcount) base = bsel_reserve(b, 0, sh, wcount);
        base = __shfl(base, 0, 64);
        float* reg = bsel_region(b, 0, sh);
        for (unsigned i = lane; i < wcount; i += 64)
            if (base + i < b.capS) reg[base + i] = wq[i];
        const int wv = wave_sum_i32((int)lvalid), wb = wave_sum_i32((int)lbelow) + (int)wbelow;
        if (lane == 0) bsel_count(b, 0, sh, (unsigned)wv, (unsigned)wb);
    }
}

// tile segments (+ overflow list) -> dense candidate list.  Workgroup g owns the tiles
// [g*tpb, (g+1)*tpb), tpb <= 256 and a multiple of 16: it sums the (byte) counts of all
// earlier tiles itself -- 16 counts per load, a few KB in all -- instead of waiting for a
// scan, then every thread copies the segment of one tile.  Workgroup 0 appends the overflow
// list.
__global__ __launch_bounds__(256) void k_lac_compact(const uint8_t* __restrict__ tile_cnt, const uint32_t* __restrict__ tile_seg,
                                                      int ntiles, int tpb, const uint32_t* __restrict__ ovf, uint32_t capovf,
                                                      int32_t* counters, uint32_t* __restrict__ cand, uint32_t cap,
                                                      int32_t* err) {
    __shared__ unsigned wsum[4];
    __shared__ unsigned red[2][4];
    __shared__ unsigned lexcl[256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int t0 = min((int)blockIdx.x * tpb, ntiles), t1 = min(t0 + tpb, ntiles);
    unsigned before = 0, all = 0;
    // tile_cnt is 16-byte aligned and zero-padded to a multiple of 16 entries; t0 % 16 == 0
    const uint4* tc4 = reinterpret_cast<const uint4*>(tile_cnt);
    const int n16 = (ntiles + 15) / 16;
#pragma unroll 2
    for (int g = tid; g < n16; g += 256) {
        const uint4 c = tc4[g];
        unsigned sum = __builtin_amdgcn_sad_u8(c.x, 0u, 0u);
        sum = __builtin_amdgcn_sad_u8(c.y, 0u, sum);
        sum = __builtin_amdgcn_sad_u8(c.z, 0u, sum);
        sum = __builtin_amdgcn_sad_u8(c.w, 0u, sum);
        all += sum;
        if (16 * g < t0) before += sum;
    }
    before = (unsigned)wave_sum_i32((int)before); all = (unsigned)wave_sum_i32((int)all);
    if (lane == 0) { red[0][wid] = before; red[1][wid] = all; }
    const int t = t0 + tid;
    // a count above the segment size can only come from a producer bug: never index past the segment
    unsigned mine = (t < t1) ? tile_cnt[t] : 0u;
    if (mine > (unsigned)CAND_TILECAP) { mine = CAND_TILECAP; atomicOr(err, BBX_DERR_LIST_OVERFLOW); }
    unsigned ltot;
    const unsigned excl = block_excl_scan256(mine, wsum, &ltot);         // contains the barriers for red[]
    const unsigned base = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const unsigned tot = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    // flat copy of the workgroup's ltot entries: entry e belongs to the last tile with lexcl <= e
    lexcl[tid] = excl;
    __syncthreads();
    for (unsigned e = tid; e < ltot; e += 256) {
        int lo = 0, hi = 255;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (lexcl[mid] <= e) lo = mid; else hi = mid - 1; }
        const unsigned pos = base + e;
        if (pos < cap) cand[pos] = tile_seg[(size_t)(t0 + lo) * CAND_TILECAP + (e - lexcl[lo])];
    }
    if (blockIdx.x == 0) {
        const unsigned novf = min((unsigned)counters[CNT_CANDOVF], capovf);
        for (unsigned i = tid; i < novf; i += 256)
            if (tot + i < cap) cand[tot + i] = ovf[i];
        if (tid == 0) {
            if (tot + novf > cap) atomicOr(err, BBX_DERR_LIST_OVERFLOW);
            counters[CNT_CANDRAW] = (int32_t)min(tot + novf, cap);
        }
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
                const unsigned k = atomicAdd((unsigned*)&counters[CNT_CANDRAW], 1u);
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

// ---- 1a'. candidates of iterations 2..n without another pass over the frame.
// After the first iteration the image only changes at the cleaned pixels (clean_medmask
// rewrites exactly the pixels of the cumulative CR list).  A pixel can seed in iteration k
// only if sp or the fine-structure ratio at it changed -- both read the image within a
// Chebyshev distance of 4 (5x5 median of s, s from a 5x5 median and a 3x3 stencil; 7x7 median
// of the 3x3-median image) -- or if it seeded before, in which case it was cleaned itself.
// First-growth members lie within 1 of a seed.  So every pixel that needs to be on the
// candidate list of iteration k (L+ > T is still necessary) lies within distance 5 of a CR-list
// pixel: test the 11x11 neighbourhoods of the list instead of the whole frame.  Duplicates
// are dropped through a bit of the flag plane; the result is a superset of what matters and
// a subset of the dense list, so the outcome is identical.
#define F_QUEUED 32u
#define SPC_CHUNK 4096
__global__ __launch_bounds__(256) void k_lac_cand_sparse(const float* __restrict__ a, lac_par p,
                                                         const uint32_t* __restrict__ crlist, int32_t* counters, uint32_t cap,
                                                         uint8_t* flags, uint32_t* __restrict__ cand_raw, int32_t* err) {
    __shared__ uint32_t q[SPC_CHUNK];
    __shared__ unsigned qn, qbase;
    const float T = p.rnp[1];
    const unsigned long long total = (unsigned long long)min((uint32_t)counters[CNT_CRLIST], cap) * 121ull;
    for (unsigned long long c0 = (unsigned long long)blockIdx.x * SPC_CHUNK; c0 < total; c0 += (unsigned long long)gridDim.x * SPC_CHUNK) {
        if (threadIdx.x == 0) qn = 0;
        __syncthreads();
        const unsigned long long c1 = min(total, c0 + SPC_CHUNK);
        for (unsigned long long k = c0 + threadIdx.x; k < c1; k += 256) {
            const uint32_t o = crlist[k / 121ull];
            const int e = (int)(k % 121ull);
            const int j = (int)(o / (uint32_t)p.nx) + e / 11 - 5, i = (int)(o % (uint32_t)p.nx) + e % 11 - 5;
            if (j >= 2 && i >= 2 && j < p.ny - 2 && i < p.nx - 2 && lplus_at(a, j, i, p.ny, p.nx) > T) {
                const size_t r = (size_t)j * p.nx + i;
                if (!(atomic_or_u8(flags, r, F_QUEUED) & F_QUEUED)) q[atomicAdd(&qn, 1u)] = (uint32_t)r;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) qbase = qn ? atomicAdd((unsigned*)&counters[CNT_CANDRAW], qn) : 0u;
        __syncthreads();
        for (unsigned k = threadIdx.x; k < qn; k += 256) {
            const unsigned pos = qbase + k;
            if (pos < cap) cand_raw[pos] = q[k]; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
        }
        __syncthreads();
    }
}

// ---- 1b. pre-filter: sp = s - medfilt5(s) <= s (s >= 0 everywhere), so only candidates with
// s > sigclip (and no mask bit) can seed or join the first growth.  With the true local
// noise instead of the read noise alone this drops ~80 % of the L+ > T list before the
// wave-per-candidate stage.  One thread per candidate; survivors are appended wave by wave.
__global__ __launch_bounds__(256) void k_lac_prefilter(const float* __restrict__ a, const uint8_t* __restrict__ mask, lac_par p,
                                                       const uint32_t* __restrict__ raw, int32_t* counters, uint32_t cap,
                                                       uint32_t* __restrict__ cand, int32_t* err) {
    const uint32_t n = min((uint32_t)counters[CNT_CANDRAW], cap);
    const uint32_t npix = (uint32_t)p.ny * (uint32_t)p.nx;
    const uint32_t nround = ((n + 63u) / 64u) * 64u;
    const int lane = threadIdx.x & 63;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nround; k += gridDim.x * blockDim.x) {
        bool keep = false;
        uint32_t o = 0;
        if (k < n) {
            o = raw[k];
            // every later consumer dereferences the indices this kernel lets through: an index off
            // the frame (a producer bug) becomes an error code here, never a device fault
            if (o >= npix) atomicOr(err, BBX_DERR_LIST_OVERFLOW);
            else if (good_px(mask, o)) {
                const int j = (int)(o / p.nx), i = (int)(o - (uint32_t)j * p.nx);
                float noise;
                keep = s_at(a, j, i, p, &noise) > p.sigclip;
            }
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (m) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd((unsigned*)&counters[CNT_CAND], (unsigned)__popcll(m));
            base = __shfl(base, 0, 64);
            if (keep) cand[base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = o;
        }
    }
}

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
// New CR pixels are queued in LDS and appended to the cumulative list with one reservation per
// workgroup: a returning atomic per wave on the one list counter retires at ~11 ns each on
// this part, which made the ~9000 stage-2 pixels of a first iteration cost 80 us.
#define F_FINAL 8u
#define G2_QCAP 1024
__global__ __launch_bounds__(256) void k_lac_grow2(const float* __restrict__ a, uint8_t* mask, lac_par p,
                                                   const uint32_t* __restrict__ stage2, uint32_t cap, uint8_t* flags,
                                                   uint32_t* __restrict__ crlist, int32_t* counters, int32_t* err,
                                                   float* __restrict__ orig, uint32_t caporig) {
    __shared__ uint32_t q[G2_QCAP];
    __shared__ unsigned qn, nnew, qbase;
    if (threadIdx.x == 0) { qn = 0; nnew = 0; qbase = 0; }
    __syncthreads();
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
                        atomicAdd(&nnew, 1u);
                        if (!(om & BBX_MASK_COSMIC)) {
                            const unsigned k = atomicAdd(&qn, 1u);
                            if (k < G2_QCAP) q[k] = (uint32_t)r;
                            else {                                              // queue full: append directly
                                const unsigned g = atomicAdd((unsigned*)&counters[CNT_CRLIST], 1u);
                                if (g < cap) crlist[g] = (uint32_t)r; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
                                if (orig) { if (g < caporig) orig[g] = a[r]; else atomicOr(err, BBX_DERR_LIST_OVERFLOW); }
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const unsigned nq = min(qn, (unsigned)G2_QCAP);
    if (threadIdx.x == 0) {
        if (nq) qbase = atomicAdd((unsigned*)&counters[CNT_CRLIST], nq);
        if (nnew) atomicAdd(&counters[CNT_NEWCR], (int)nnew);
    }
    __syncthreads();
    // (a pixel enters the list once and is cleaned only afterwards: a[] still holds its input value,
    // kept in orig[] for the exact background level on demand)
    for (unsigned k = threadIdx.x; k < nq; k += blockDim.x) {
        if (qbase + k < cap) crlist[qbase + k] = q[k]; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
        if (orig) { if (qbase + k < caporig) orig[qbase + k] = a[q[k]]; else atomicOr(err, BBX_DERR_LIST_OVERFLOW); }
    }
}

// ---- 4. clean_medmask on the cumulative CR list -------------------------------------------------
// A CR pixel without any good neighbour takes the background level (median of the good input
// pixels).  That is rare, and the level costs a select to finish, so such pixels are only
// listed here; k_lac_bg, right after this kernel, produces the level on demand and fills them
// (nothing else reads a CR pixel's value in between: the windows take mask == 0 pixels only).
__global__ __launch_bounds__(256) void k_lac_clean(float* a, const uint8_t* __restrict__ mask, lac_par p,
                                                   const uint32_t* __restrict__ crlist, int32_t* counters,
                                                   uint32_t cap, uint32_t* __restrict__ bglist, uint32_t capbg, int32_t* err) {
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
        if (lane == 0) {
            if (cnt > 0) a[o] = med;
            else {
                const unsigned q = atomicAdd((unsigned*)&counters[CNT_BGNEED], 1u);
                if (q < capbg) bglist[q] = o; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
            }
        }
    }
}

// ---- 4b. background level on demand ------------------------------------------------------------
// One workgroup: lower-middle element (rank (n-1)/2, astroscrappy's quick-select median) of the
// values picked by [from_frame]: the good pixels of the frame, or the side buffer of the
// bracketed select (then [rank] counts inside the buffer).  Three digit passes over the float
// keys with an LDS histogram; slow next to the multi-workgroup select, but it only runs when a
// frame needs the level at all.
// [orig, norig]: with from_frame, pixels carrying the COSMIC bit are skipped (their values were
// replaced) and the saved input values of the CR list stand in for them.
__device__ float wg_lower_median(const bsel_dev& b, const float* __restrict__ a, const uint8_t* __restrict__ mask,
                                 size_t npix, bool from_frame, unsigned long long rank,
                                 const float* __restrict__ orig = nullptr, uint32_t norig = 0) {
    __shared__ uint32_t lh[2048];
    __shared__ uint32_t s_prefix;
    __shared__ unsigned long long s_rank;
    __shared__ int s_empty;
    const int tid = threadIdx.x;
    if (tid == 0) { s_prefix = 0; s_rank = rank; s_empty = 0; }
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    uint32_t himask = 0;
    for (int ps = 0; ps < 3; ps++) {
        for (int i = tid; i < 2048; i += blockDim.x) lh[i] = 0;
        __syncthreads();
        const uint32_t pre = s_prefix, dmask = (1u << nbits[ps]) - 1u;
        const int shift = shifts[ps];
        if (from_frame) {
            const unsigned skipbits = orig ? 0xffu : (0xffu & ~BBX_MASK_COSMIC);
            for (size_t i = tid; i < npix; i += blockDim.x) {
                if (mask[i] & skipbits) continue;
                const uint32_t key = f2key(a[i]);
                if ((key & himask) == pre) atomicAdd(&lh[(key >> shift) & dmask], 1u);
            }
            for (uint32_t i = tid; i < norig; i += blockDim.x) {
                const uint32_t key = f2key(orig[i]);
                if ((key & himask) == pre) atomicAdd(&lh[(key >> shift) & dmask], 1u);
            }
        } else {
            for (unsigned sh = 0; sh < BSEL_NSH; sh++) {
                const uint32_t cnt = min(b.shard[sh].nbuf, b.capS);
                const float* v = bsel_region(b, 0, sh);
                for (uint32_t i = tid; i < cnt; i += blockDim.x) {
                    const uint32_t key = f2key(v[i]);
                    if ((key & himask) == pre) atomicAdd(&lh[(key >> shift) & dmask], 1u);
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (ps == 0 && from_frame) {
                unsigned long long total = 0;
                for (int bin = 0; bin < 2048; bin++) total += lh[bin];
                if (total == 0) s_empty = 1; else s_rank = (total - 1) / 2;
            }
            unsigned long long r = s_rank, acc = 0;
            for (int bin = 0; bin <= (int)dmask; bin++) {
                if (acc + lh[bin] > r) { s_rank = r - acc; s_prefix = pre | ((uint32_t)bin << shift); break; }
                acc += lh[bin];
            }
        }
        __syncthreads();
        if (s_empty) return 0.f;                                // no good pixel at all: level 0
        himask |= dmask << shift;
    }
    return key2f(s_prefix);
}

// mode 0, after the pass that fed the bracketed select and before the frame is modified: sums
// of the shards; if the bracket cannot deliver the rank (see bbx_bsel.h) the exact select over
// the frame runs now.  mode 1, after k_lac_clean: nothing to do unless pixels were listed; then
// the level comes from the side buffer (unless it is known already) and the listed pixels get it.
// seg->pad marks "result[0] holds the level".
__global__ __launch_bounds__(256) void k_lac_bg(float* a, const uint8_t* __restrict__ mask, lac_par p, bsel_dev b,
                                                 int32_t* counters, const uint32_t* __restrict__ bglist, uint32_t capbg, int mode,
                                                 int32_t* __restrict__ stats) {
    __shared__ float s_bg;
    __shared__ int s_fail;
    bsel_seg* sg = b.seg;
    const size_t npix = (size_t)p.ny * p.nx;
    if (mode == 0) {
        if (threadIdx.x < 64) {
            const bsel_shard* q = &b.shard[threadIdx.x];
            const long long n = wave_sum_i64((long long)q->n), below = wave_sum_i64((long long)q->below);
            const long long nbuf = wave_sum_i64((long long)q->nbuf);
            const bool over = __ballot(q->nbuf > b.capS) != 0ull;
            if (threadIdx.x == 0) {
                sg->n = (unsigned long long)n; sg->below = (unsigned long long)below; sg->nbuf = (uint32_t)nbuf;
                const long long k0 = n ? (n - 1) / 2 : 0;
                const int fail = (sg->fail || over || k0 < below || k0 >= below + nbuf) ? 1 : 0;
                sg->fail = fail; sg->pad = 0;
                s_fail = fail;
            }
        }
        __syncthreads();
        if (!s_fail) return;
        const float r = wg_lower_median(b, a, mask, npix, true, 0ull);
        if (threadIdx.x == 0) { sg->result[0] = r; sg->result[1] = r; sg->pad = 1; }
        return;
    }
    const uint32_t n = min((uint32_t)counters[CNT_BGNEED], capbg);
    if (n == 0) return;
    float bg;
    if (!sg->pad) {                                             // workgroup-uniform
        const unsigned long long k0 = sg->n ? (sg->n - 1) / 2 : 0;
        bg = wg_lower_median(b, a, mask, npix, false, k0 - sg->below);
    } else bg = sg->result[0];
    if (threadIdx.x == 0) s_bg = bg;
    __syncthreads();
    bg = s_bg;
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) a[bglist[k]] = bg;
    __syncthreads();
    if (threadIdx.x == 0) { sg->result[0] = bg; sg->result[1] = bg; sg->pad = 1; counters[CNT_BGNEED] = 0; stats[15] = 1; }
}

// The same step when nothing was prepared (BBX_OPT_LAC_LEVEL_FEED = 0): pixels listed by
// k_lac_clean get the level, selected exactly over the good pixels of the frame -- those the run
// has flagged and cleaned so far are taken with their input values from orig[] (saved by
// k_lac_grow2).  lvl[0] = level, lvl[1] = known? (zeroed by k_lac_begin).
// Three digit passes of a radix select (11 + 11 + 10 bits of the order-preserving key), each its
// own launch: k_lac_bgf_hist<ps> adds the LDS histogram of every workgroup to ghist[ps]; the
// kernel boundary is the grid-wide synchronisation (no assumption about which workgroups are
// resident together); the next launch's workgroups each re-derive the prefix and rank from the
// finished histograms.  k_lac_bgf_final resolves the last digit and fills the listed pixels.
// All four return at once unless pixels are listed (the usual frame).
#define BGF_WGS 256
struct bgf_sh {
    unsigned long long part[256];
    unsigned long long rank;
    uint32_t prefix;
    int empty;
};
// prefix / rank after the first [nps] digits, from the global histograms (workgroup-uniform result)
__device__ __forceinline__ void bgf_resolve(const uint32_t* __restrict__ ghist, int nps, bgf_sh& sh, uint32_t& himask) {
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    const int tid = threadIdx.x;
    if (tid == 0) { sh.prefix = 0; sh.rank = 0; sh.empty = 0; }
    himask = 0;
    __syncthreads();
    for (int ps = 0; ps < nps; ps++) {
        const uint32_t* gh = ghist + ps * 2048;
        const uint32_t pre = sh.prefix, dmask = (1u << nbits[ps]) - 1u;
        const int shift = shifts[ps];
        unsigned long long part = 0;
        uint32_t mine[8];
        for (int k = 0; k < 8; k++) { mine[k] = gh[tid * 8 + k]; part += mine[k]; }
        sh.part[tid] = part;
        __syncthreads();
        if (tid == 0) {
            unsigned long long total = 0;
            for (int t = 0; t < 256; t++) total += sh.part[t];
            if (ps == 0) { if (total == 0) sh.empty = 1; else sh.rank = (total - 1) / 2; }
            unsigned long long r = sh.rank, acc = 0;
            int t = 0;
            while (t < 255 && acc + sh.part[t] <= r) { acc += sh.part[t]; t++; }
            sh.part[0] = acc;                                // counts below group t
            sh.prefix = (uint32_t)t;                         // group index, replaced below
        }
        __syncthreads();
        if (sh.empty) return;
        const int grp = (int)sh.prefix;
        const unsigned long long below = sh.part[0];
        __syncthreads();
        if (tid == grp) {
            unsigned long long acc = below;
            int k = 0;
            while (k < 7 && acc + mine[k] <= sh.rank) { acc += mine[k]; k++; }
            sh.rank = sh.rank - acc;
            sh.prefix = pre | ((uint32_t)(grp * 8 + k) << shift);
        }
        __syncthreads();
        himask |= dmask << shift;
    }
}

__global__ __launch_bounds__(256) void k_lac_bgf_hist(const float* __restrict__ a, const uint8_t* __restrict__ mask, lac_par p,
                                                      const int32_t* __restrict__ counters, uint32_t capbg,
                                                      const float* __restrict__ orig, uint32_t caporig,
                                                      const float* __restrict__ lvl, uint32_t* ghist, int ps) {
    __shared__ uint32_t lh[2048];
    __shared__ bgf_sh sh;
    const uint32_t n = min((uint32_t)counters[CNT_BGNEED], capbg);
    if (n == 0 || lvl[1] != 0.f) return;                       // nothing listed, or level known (grid-uniform)
    const int tid = threadIdx.x;
    uint32_t himask;
    bgf_resolve(ghist, ps, sh, himask);
    if (sh.empty) return;
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    const uint32_t pre = sh.prefix, dmask = (1u << nbits[ps]) - 1u;
    const int shift = shifts[ps];
    for (int i = tid; i < 2048; i += 256) lh[i] = 0;
    __syncthreads();
    const size_t npix = (size_t)p.ny * p.nx;
    const uint32_t norig = min((uint32_t)counters[CNT_CRLIST], caporig);
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < npix; i += (size_t)gridDim.x * 256) {
        if (mask[i]) continue;                                  // masked, or flagged (and cleaned) by this run
        const uint32_t key = f2key(a[i]);
        if ((key & himask) == pre) atomicAdd(&lh[(key >> shift) & dmask], 1u);
    }
    for (uint32_t i = blockIdx.x * 256 + tid; i < norig; i += gridDim.x * 256) {
        const uint32_t key = f2key(orig[i]);
        if ((key & himask) == pre) atomicAdd(&lh[(key >> shift) & dmask], 1u);
    }
    __syncthreads();
    uint32_t* gh = ghist + ps * 2048;
    for (int i = tid; i < 2048; i += 256) if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

__global__ __launch_bounds__(256) void k_lac_bgf_final(float* a, lac_par p, int32_t* counters,
                                                       const uint32_t* __restrict__ bglist, uint32_t capbg, float* lvl,
                                                       int32_t* __restrict__ stats, uint32_t* ghist) {
    __shared__ bgf_sh sh;
    const uint32_t n = min((uint32_t)counters[CNT_BGNEED], capbg);
    if (n == 0) return;
    const int tid = threadIdx.x;
    float bg;
    if (lvl[1] != 0.f) bg = lvl[0];                             // level known from an earlier iteration
    else {
        uint32_t himask;
        bgf_resolve(ghist, 3, sh, himask);
        bg = sh.empty ? 0.f : key2f(sh.prefix);                 // no good pixel at all: level 0
    }
    for (uint32_t k = tid; k < n; k += blockDim.x) a[bglist[k]] = bg;
    __syncthreads();
    if (tid == 0) { lvl[0] = bg; lvl[1] = 1.f; counters[CNT_BGNEED] = 0; stats[15] = 1; }
}

// the flag plane is only ever written at listed pixels (raw candidates; 3x3 around stage-2
// pixels): clearing exactly those keeps it all-zero between iterations without a memset
__global__ __launch_bounds__(256) void k_lac_unflag(lac_par p, const uint32_t* __restrict__ cand,
                                                    const uint32_t* __restrict__ stage2, int32_t* counters, uint32_t cap,
                                                    uint8_t* __restrict__ flags, int32_t* __restrict__ stats, int it) {
    const uint32_t n1 = min((uint32_t)counters[CNT_CANDRAW], cap), n2 = min((uint32_t)counters[CNT_STAGE2], cap);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n1; k += stride) flags[cand[k]] = 0;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < 9u * n2; k += stride) {
        const uint32_t o = stage2[k / 9u];                   // >= 2 px inside the frame
        const int e = (int)(k % 9u);
        flags[(size_t)o + (size_t)((e / 3 - 1) * p.nx) + (e % 3 - 1)] = 0;
    }
    // end of the iteration: the workgroup that finishes last (every other one has read the
    // counters by then) records the statistics and resets the per-iteration counters
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&counters[CNT_TICKET], 1) == (int)gridDim.x - 1) {
        stats[it] = counters[CNT_NEWCR];
        stats[7] = counters[CNT_CRLIST];
        if (it < 3) { stats[8 + 2 * it] = counters[CNT_CAND]; stats[9 + 2 * it] = counters[CNT_STAGE2]; }
        counters[CNT_NEWCR] = 0; counters[CNT_CAND] = 0; counters[CNT_STAGE2] = 0; counters[CNT_CANDOVF] = 0;
        counters[CNT_CANDRAW] = 0; counters[CNT_TICKET] = 0;
    }
}

// start of a frame: counters and statistics to zero, and readnoise -> {rn2, T, rn} on the
// device.  With d_rdn16 the read noise is the float32 image of np.nanmean of the 16 channel
// sigmas (header RDNOISE, blackbox.py:6867) evaluated in numpy's pairwise order for 16
// elements, so no host round trip is needed between os_corr and here.
__global__ void k_lac_begin(int32_t* counters, int32_t* stats, uint8_t* tile_cnt_pad, float readnoise,
                            const double* __restrict__ rdn16, float sigclip, float* out, unsigned* gbar) {
    const int t = threadIdx.x;
    if (gbar) for (int i = t; i < 3 * 2048; i += blockDim.x) gbar[i] = 0;   // digit histograms of k_lac_bgf_hist
    if (t < 16) stats[t] = 0;
    if (t < 16 && tile_cnt_pad) tile_cnt_pad[t] = 0;         // k_lac_compact reads the counts sixteen at a time
    if (t != 0) return;
    counters[CNT_CAND] = 0; counters[CNT_STAGE2] = 0; counters[CNT_CRLIST] = 0; counters[CNT_NEWCR] = 0;
    counters[CNT_CANDOVF] = 0; counters[CNT_CANDRAW] = 0; counters[CNT_TICKET] = 0; counters[CNT_BGNEED] = 0;
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
    out[8] = 0.f; out[9] = 0.f;                                // background level (k_lac_bg_frame): value, known?
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
    const size_t cap = (npix / 4 + 4096) & ~(size_t)63;
    // BBX_OPT_DEBUG_LISTCAP: the kernels see a smaller capacity than what is allocated (tests of the overflow path)
    const size_t capk = (ctx->debug_listcap > 0 && (size_t)ctx->debug_listcap < cap) ? (size_t)ctx->debug_listcap : cap;
    const bool vec = (nx % 4 == 0) && (((uintptr_t)d_data) % 16 == 0);
    const int nwx = (nx + CAND_SPAN - 1) / CAND_SPAN;             // waves along x
    const size_t ntiles = (size_t)nwx * ((ny + CAND_ROWS - 1) / CAND_ROWS), capovf = cap / 4 + 4096;
    const dim3 gvec((unsigned)(((ntiles + 7) / 8) * 8));                    // one wave per workgroup, see the tile order in the kernel
    p.nwx = nwx; p.ntiles = (int)ntiles;
    // candidate workspace: filtered list | raw list | overflow list | per-tile segments | per-tile counts
    uint32_t* cand = (uint32_t*)bbx_ws(ctx, WS_CAND, (2 * cap + capovf + 64 + ntiles + 64 + ntiles * CAND_TILECAP) * 4, &rc); if (rc) return rc;
    uint32_t* cand_raw = cand + cap;
    uint32_t* ovf = cand_raw + cap;
    uint32_t* tile_seg = ovf + ((capovf + 63) / 64) * 64;
    uint8_t* tile_cnt = (uint8_t*)(tile_seg + ntiles * CAND_TILECAP);     // 16-byte aligned; one byte per tile (<= CAND_TILECAP)
    uint32_t* stage2 = (uint32_t*)bbx_ws(ctx, WS_STAGE2, cap * 4, &rc); if (rc) return rc;
    uint32_t* crlist = (uint32_t*)bbx_ws(ctx, WS_CRLIST, cap * 4, &rc); if (rc) return rc;
    uint8_t* flags = (uint8_t*)bbx_ws(ctx, WS_FLAGS, npix + 16, &rc); if (rc) return rc;
    int32_t* cnt = ctx->d_counters;
    const bool feed = ctx->lac_feed != 0;
    bsel_dev bs;
    memset(&bs, 0, sizeof(bs));
    float* orig = nullptr;
    const size_t caporig = std::min<size_t>(cap, (size_t)1 << 23);
    if (feed) { rc = bbx_bsel_prepare(ctx, d_data, d_mask, ny, nx, ny, nx, &bs, s); if (rc) return rc; }
    uint32_t* ghist = nullptr;
    if (!feed) {
        // input values of the CR pixels | 3 digit histograms | grid-barrier counter (k_lac_bg_frame)
        char* w = (char*)bbx_ws(ctx, WS_CRORIG, caporig * sizeof(float) + 3 * 2048 * 4 + 64, &rc); if (rc) return rc;
        orig = (float*)w;
        ghist = (uint32_t*)(w + caporig * sizeof(float));
    }
    hipLaunchKernelGGL(k_lac_begin, dim3(1), dim3(64), 0, s, cnt, d_stats, tile_cnt + ntiles, readnoise, d_rdn16, sigclip, rnp,
                       (unsigned*)ghist);
    // the flag plane is kept all-zero between calls (k_lac_unflag); zero it when it is new
    if (ctx->flags_clean_ptr != flags || ctx->flags_clean_bytes < npix) BBX_HIP(hipMemsetAsync(flags, 0, npix, s));
    ctx->flags_clean_ptr = nullptr;
    // background level of the unmasked input pixels (needed when a CR pixel has no good
    // neighbour): bracketed select fed by the first candidate pass, no extra read of the frame
    const unsigned gdense = 256u * 16u, gsparse = 256u * 8u;
    // workgroups of the level select's digit passes
    const unsigned bgf_wgs = (unsigned)std::min(BGF_WGS, std::max(8, ctx->num_cus));
    for (int it = 0; it < niter; it++) {
        if (it == 0) {
            // the one dense pass: candidates of the first iteration (+ the background-level feed)
            if (vec && feed) BBX_LAUNCH_TIMED(ctx, BBX_PROF_LAC_DENSE, k_lac_cand_v4<true>, gvec, dim3(64), CAND_WQ * 4 + FEED_WQ * 4, s, d_data, d_mask, p, tile_cnt, tile_seg, ovf, cnt, (uint32_t)capovf, ctx->d_err, bs);
            else if (vec) BBX_LAUNCH_TIMED(ctx, BBX_PROF_LAC_DENSE, k_lac_cand_v4<false>, gvec, dim3(64), CAND_WQ * 4, s, d_data, d_mask, p, tile_cnt, tile_seg, ovf, cnt, (uint32_t)capovf, ctx->d_err, bs);
            else bbx_prof_start(ctx, BBX_PROF_LAC_DENSE, s);
            if (vec) {}
            else if (feed) hipLaunchKernelGGL(k_lac_cand_s<true>, dim3(gdense), dim3(256), sizeof(bsel_lds), s, d_data, d_mask, p, cand_raw, cnt, (uint32_t)capk, ctx->d_err, bs);
            else hipLaunchKernelGGL(k_lac_cand_s<false>, dim3(gdense), dim3(256), 0, s, d_data, d_mask, p, cand_raw, cnt, (uint32_t)capk, ctx->d_err, bs);
            bbx_prof_stop(ctx, s);
            if (vec) {
                const int tpb = (int)std::min<size_t>(256, std::max<size_t>(16, ((ntiles + 127) / 128 + 15) / 16 * 16));
                hipLaunchKernelGGL(k_lac_compact, dim3((unsigned)((ntiles + tpb - 1) / tpb)), dim3(256), 0, s, tile_cnt, tile_seg,
                                   (int)ntiles, tpb, ovf, (uint32_t)capovf, cnt, cand_raw, (uint32_t)capk, ctx->d_err);
            }
        } else {
            // later iterations: only the surroundings of the pixels cleaned so far can differ
            hipLaunchKernelGGL(k_lac_cand_sparse, dim3(512), dim3(256), 0, s, d_data, p, crlist, cnt, (uint32_t)capk, flags, cand_raw,
                               ctx->d_err);
        }
        hipLaunchKernelGGL(k_lac_prefilter, dim3(256), dim3(256), 0, s, d_data, d_mask, p, cand_raw, cnt, (uint32_t)capk, cand, ctx->d_err);
        if (it == 0 && feed) hipLaunchKernelGGL(k_lac_bg, dim3(1), dim3(256), 0, s, d_data, d_mask, p, bs, cnt, ovf, (uint32_t)capovf, 0, d_stats);
        bbx_prof_start(ctx, BBX_PROF_LAC_SPARSE, s);
        hipLaunchKernelGGL(k_lac_seed, dim3(gsparse), dim3(256), 0, s, d_data, d_mask, p, cand, cnt, (uint32_t)capk, flags);
        hipLaunchKernelGGL(k_lac_grow1, dim3(gsparse), dim3(256), 0, s, p, cand, cnt, (uint32_t)capk, flags, stage2, cnt, ctx->d_err);
        hipLaunchKernelGGL(k_lac_grow2, dim3(512), dim3(256), 0, s, d_data, d_mask, p, stage2, (uint32_t)capk, flags, crlist,
                           cnt, ctx->d_err, orig, (uint32_t)caporig);
        // (the overflow list of the dense pass is free again after k_lac_compact: pixels waiting for the level)
        hipLaunchKernelGGL(k_lac_clean, dim3(gsparse), dim3(256), 0, s, d_data, d_mask, p, crlist, cnt, (uint32_t)capk, ovf,
                           (uint32_t)capovf, ctx->d_err);
        if (feed) hipLaunchKernelGGL(k_lac_bg, dim3(1), dim3(256), 0, s, d_data, d_mask, p, bs, cnt, ovf, (uint32_t)capovf, 1, d_stats);
        else {
            for (int ps = 0; ps < 3; ps++)
                hipLaunchKernelGGL(k_lac_bgf_hist, dim3(bgf_wgs), dim3(256), 0, s, d_data, d_mask, p, cnt, (uint32_t)capovf, orig,
                                   (uint32_t)caporig, rnp + 8, ghist, ps);
            hipLaunchKernelGGL(k_lac_bgf_final, dim3(1), dim3(256), 0, s, d_data, p, cnt, ovf, (uint32_t)capovf, rnp + 8, d_stats, ghist);
        }
        hipLaunchKernelGGL(k_lac_unflag, dim3(256), dim3(256), 0, s, p, cand_raw, stage2, cnt, (uint32_t)capk, flags, d_stats, it);
        bbx_prof_stop(ctx, s);
    }
    ctx->flags_clean_ptr = flags; ctx->flags_clean_bytes = npix;
    BBX_LAUNCH_CHECK();
    // NCOSMICS: 8-connected objects of the CR pixels (blackbox.py:4354-4356)
    return bbx_cc_count_list(ctx, crlist, &cnt[CNT_CRLIST], cap, ny, nx, &d_stats[6], s);
}
