// bbx_mask.hip -- mask_init tail, fill_sat_holes, connected-object counts, mask counts.
//
// Saturated pixels are rare (tens of stars per frame), so everything here works
// on the sparse pixel queue that bbx_calibrate leaves behind, plus a 1-bit-per-
// pixel plane (14 MB for a 10560^2 frame) for the 3x3 closing and the hole fill.
// The full-resolution uint8 mask is only touched at the (few) pixels that change.
#include "bbx_common.h"

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------
// sparse: crosstalk flags + saturated-connected ring + bit plane of M = sat | satcon
// (mask_init, blackbox.py:4504-4531, 4557-4562; fill_sat_holes 4590-4591)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sat_scatter(const uint32_t* __restrict__ satlist,
                                                     const int32_t* __restrict__ counters, bbx_dims d,
                                                     uint8_t* mask, u64* bitsM, int W, int cap) {
    const int n = min(counters[CNT_SAT], cap);                  // an overflowing producer has raised the error flag
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = satlist[i];
        const int Y = p / d.nx, X = p - Y * d.nx;
        const int iy = Y / d.ysz, y = Y - iy * d.ysz;
        const int ix = X / d.xsz, x = X - ix * d.xsz;
        const int c = iy * 8 + ix;
        // victims: same pixel position in the 15 other channels, y-flipped across rows
        for (int v = 0; v < 16; v++) {
            if (v == c) continue;
            const int vy = v >> 3, vx = v & 7;
            const int yy = (vy == iy) ? y : (d.ysz - 1 - y);
            atomic_or_u8(mask, (size_t)(vy * d.ysz + yy) * d.nx + (size_t)vx * d.xsz + x, BBX_MASK_XTALK);
        }
        // 3x3 dilation ring and the bit plane
        for (int dyy = -1; dyy <= 1; dyy++) {
            const int Yn = Y + dyy;
            if (Yn < 0 || Yn >= d.ny) continue;
            for (int dxx = -1; dxx <= 1; dxx++) {
                const int Xn = X + dxx;
                if (Xn < 0 || Xn >= d.nx) continue;
                const size_t q = (size_t)Yn * d.nx + Xn;
                if ((dyy || dxx) && !(mask[q] & BBX_MASK_SAT)) atomic_or_u8(mask, q, BBX_MASK_SATCON);
                atomicOr(&bitsM[(size_t)Yn * W + (Xn >> 6)], 1ull << (Xn & 63));
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// 3x3 binary closing on the bit plane (ndimage.binary_closing, border_value 0)
// ---------------------------------------------------------------------------------
// closing in one pass: word w of row r of erode(dilate(in)).  The dilation of rows r-1..r+1 is
// rebuilt from the words w-1, w, w+1 of rows r-2..r+2 (15 loads that hit the caches; the
// saturated pixels are sparse, so almost every thread finds 15 zeros and stores a zero).
__global__ __launch_bounds__(256) void k_bits_close(const u64* __restrict__ in, u64* __restrict__ out, int ny, int nx, int W) {
    const size_t total = (size_t)ny * W;
    const u64 lastmask = (nx & 63) ? ((1ull << (nx & 63)) - 1) : ~0ull;      // dilation stays inside the image
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)((unsigned)i / (unsigned)W), w = (int)((unsigned)i - (unsigned)r * (unsigned)W);   // ny * W < 2^22 (COARSE_MAX tiles)
        u64 m[5][3]; u64 any = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int rr = r + k - 2;
            const bool in_r = rr >= 0 && rr < ny;
            m[k][0] = (in_r && w > 0) ? in[(size_t)rr * W + w - 1] : 0ull;
            m[k][1] = in_r ? in[(size_t)rr * W + w] : 0ull;
            m[k][2] = (in_r && w < W - 1) ? in[(size_t)rr * W + w + 1] : 0ull;
            any |= m[k][0] | m[k][1] | m[k][2];
        }
        u64 res = 0;
        if (any && r > 0 && r < ny - 1) {                                    // erosion: border_value = 0
            res = ~0ull;
#pragma unroll
            for (int k = 1; k <= 3; k++) {                                   // dilated row r + k - 2
                const u64 nl = m[k - 1][0] | m[k][0] | m[k + 1][0];          // OR of the three rows, per word
                const u64 n = m[k - 1][1] | m[k][1] | m[k + 1][1];
                const u64 nr = m[k - 1][2] | m[k][2] | m[k + 1][2];
                u64 c = n | (n << 1) | (n >> 1) | (nl >> 63) | (nr << 63);
                if (w == W - 1) c &= lastmask;
                // the two bits of the neighbouring words' dilation that the erosion looks at
                u64 l63 = (w > 0) ? (((nl | (nl << 1)) >> 63) | (n & 1ull)) & 1ull : 0ull;
                u64 r0 = 0ull;
                if (w < W - 1) {
                    u64 d = (nr | (nr >> 1) | (n >> 63)) & 1ull;
                    if (w + 1 == W - 1) d &= lastmask;
                    r0 = d;
                }
                res &= c & ((c << 1) | l63) & ((c >> 1) | (r0 << 63));
            }
        }
        out[i] = res;
    }
}

// ---------------------------------------------------------------------------------
// hole filling (ndimage.binary_fill_holes, 3x3 structure): background reached from
// outside the image through 8-connected background is NOT a hole.
//   tile = 64 px (one word) x 64 rows.  Empty tiles 4-connected to the frame edge
//   through empty tiles are resolved wholesale (coarse flood, one workgroup, LDS);
//   the remaining tiles are flooded at pixel level with word-parallel bit tricks.
// ---------------------------------------------------------------------------------
// one wave per 64 consecutive tiles of a tile row: occupancy bytes (for the byte-wise flood)
// and the row's free-tile bit word (for the run-based kernel)
__global__ __launch_bounds__(256) void k_tile_occupancy(const u64* __restrict__ bitsC, uint8_t* __restrict__ occ,
                                                        u64* __restrict__ freebits, int ny, int W, int TH) {
    // four waves share the 64 rows of a tile row (16 independent loads per lane)
    __shared__ u64 part[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ty = blockIdx.y, k = blockIdx.x, tx = 64 * k + lane;
    u64 any = 0;
    if (tx < W) {
        const int r0 = ty * 64 + wv * 16;
#pragma unroll
        for (int j = 0; j < 16; j++) { const int r = r0 + j; if (r < ny) any |= bitsC[(size_t)r * W + tx]; }
    }
    part[wv][lane] = any;
    __syncthreads();
    if (wv) return;
    any = (part[0][lane] | part[1][lane]) | (part[2][lane] | part[3][lane]);
    if (tx < W) occ[ty * W + tx] = any ? 1 : 0;
    const u64 fr = __builtin_amdgcn_ballot_w64(tx < W && any == 0);
    if (lane == 0) freebits[(size_t)ty * gridDim.x + k] = fr;
}

#define COARSE_MAX 49152
// state per tile: 0 = unresolved, 1 = resolved outside (empty + connected to the edge)
__global__ __launch_bounds__(1024) void k_coarse_flood(const uint8_t* __restrict__ occ, uint8_t* __restrict__ state,
                                                       int W, int TH, uint32_t* __restrict__ tiles,
                                                       int32_t* counters, int32_t* err) {
    __shared__ uint8_t st[COARSE_MAX];
    __shared__ int changed;
    const int nt = W * TH;
    for (int t = threadIdx.x; t < nt; t += blockDim.x) {
        const int ty = t / W, tx = t - ty * W;
        const bool edge = (tx == 0 || ty == 0 || tx == W - 1 || ty == TH - 1);
        st[t] = occ[t] ? 2 : (edge ? 1 : 0);          // bit 1 = occupied, bit 0 = resolved outside
    }
    __syncthreads();
    // run-length propagation: one thread sweeps a whole tile row (both directions), then a
    // whole tile column; an empty frame is resolved after a single round
    for (int iter = 0; iter < nt + 4; iter++) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (int ty = threadIdx.x; ty < TH; ty += blockDim.x) {
            bool run = false;
            for (int tx = 0; tx < W; tx++) {
                const int t = ty * W + tx;
                if (st[t] & 2) run = false; else if (st[t]) run = true; else if (run) { st[t] = 1; changed = 1; }
            }
            run = false;
            for (int tx = W - 1; tx >= 0; tx--) {
                const int t = ty * W + tx;
                if (st[t] & 2) run = false; else if (st[t]) run = true; else if (run) { st[t] = 1; changed = 1; }
            }
        }
        __syncthreads();
        for (int tx = threadIdx.x; tx < W; tx += blockDim.x) {
            bool run = false;
            for (int ty = 0; ty < TH; ty++) {
                const int t = ty * W + tx;
                if (st[t] & 2) run = false; else if (st[t]) run = true; else if (run) { st[t] = 1; changed = 1; }
            }
            run = false;
            for (int ty = TH - 1; ty >= 0; ty--) {
                const int t = ty * W + tx;
                if (st[t] & 2) run = false; else if (st[t]) run = true; else if (run) { st[t] = 1; changed = 1; }
            }
        }
        __syncthreads();
        const int c = changed;
        __syncthreads();
        if (!c) break;
    }
    // (single workgroup: the tile counter starts from zero here, no memset before the launch)
    if (threadIdx.x == 0) counters[CNT_TILES] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < nt; t += blockDim.x) {
        state[t] = st[t] & 1;
        if (!(st[t] & 1)) { int k = atomicAdd(&counters[CNT_TILES], 1); tiles[k] = (uint32_t)t; }
    }
    (void)err;
}

// Bit-row helpers for tile grids up to CF_MAX x CF_MAX (a 10560^2 frame has 165 x 165
// tiles; a tile row is CF_WORDS 64-bit words).  "Reached" spreads along the free runs of a
// row in O(1) with the carry trick: for seeds X inside free runs F, (F + X) ^ F marks every
// run from its seed up to the run's end (the carry ripples through the ones); the other
// direction is the same on bit-reversed words.
#define CF_MAX 256
#define CF_WORDS 4
__device__ __forceinline__ bool cf_fill_row(const u64* F, u64* S, int nw) {
    u64 f[CF_WORDS], s0[CF_WORDS], s[CF_WORDS];
#pragma unroll
    for (int k = 0; k < CF_WORDS; k++) { f[k] = k < nw ? F[k] : 0ull; s0[k] = k < nw ? (S[k] & f[k]) : 0ull; s[k] = s0[k]; }
    // towards higher bits
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < CF_WORDS; k++) {
        const u64 x = s[k];
        const u64 a = f[k] + x, b = a + carry;
        carry = (u64)((a < x) | (b < a));
        s[k] |= (b ^ f[k]) & f[k];
    }
    // towards lower bits: the same on the reversed row (word order and bit order)
    carry = 0;
#pragma unroll
    for (int k = CF_WORDS - 1; k >= 0; k--) {
        const u64 fr = __brevll(f[k]), x = __brevll(s[k]);
        const u64 a = fr + x, b = a + carry;
        carry = (u64)((a < x) | (b < a));
        s[k] |= __brevll((b ^ fr) & fr);
    }
    bool ch = false;
#pragma unroll
    for (int k = 0; k < CF_WORDS; k++) if (k < nw) { ch |= (s[k] != s0[k]); S[k] = s[k]; }
    return ch;
}

// Connected components of the free tiles instead of a flood: the free tiles of a row form
// runs; runs of adjacent rows that overlap are joined in a lock-free union-find in LDS, runs
// that touch the frame border are joined to a virtual "outside" node, and a run is reached
// iff its root is the outside's.  Constant depth whatever the shape of the free space (an
// iterative flood needs one round per turn of the longest free path, ~10 on a real frame).  parent[] lives in dynamic LDS: (TH * (W/2 + 1) + 1) words.
__device__ __forceinline__ unsigned cfu_find(unsigned* parent, unsigned i) {
    unsigned p = *(volatile unsigned*)&parent[i];
    while (p != i) {
        const unsigned gp = *(volatile unsigned*)&parent[p];
        if (gp != p) parent[i] = gp;                  // path halving (benign race)
        i = p; p = gp;
    }
    return i;
}
__device__ __forceinline__ void cfu_union(unsigned* parent, unsigned a, unsigned b) {
    for (;;) {
        a = cfu_find(parent, a); b = cfu_find(parent, b);
        if (a == b) return;
        if (a < b) { const unsigned t = a; a = b; b = t; }      // hang the larger index under the smaller
        if (atomicCAS(&parent[a], a, b) == a) return;
    }
}

__global__ __launch_bounds__(CF_MAX) void k_coarse_cc(const u64* __restrict__ freebits, uint8_t* __restrict__ state, int W, int TH,
                                                      uint32_t* __restrict__ tiles, int32_t* counters) {
    extern __shared__ unsigned parent[];
    __shared__ u64 F[CF_MAX][CF_WORDS], R[CF_MAX][CF_WORDS], ST[CF_MAX][CF_WORDS];
    __shared__ unsigned rowbase[CF_MAX + 1];
    __shared__ unsigned wsum[CF_MAX / 64];
    const int t = threadIdx.x;
    const int WW = (W + 63) >> 6;
    const int lane = t & 63, wave = t >> 6, nwave = CF_MAX / 64;
    // free-tile bit rows, written by k_tile_occupancy
    for (int idx = t; idx < TH * WW; idx += CF_MAX) {
        const int ty = idx / WW, k = idx - ty * WW;
        F[ty][k] = freebits[idx]; R[ty][k] = 0;
    }
    __syncthreads();
    // run starts and the number of runs per row; run id = rowbase[row] + (# starts at or below the bit) - 1
    unsigned nrun = 0;
    if (t < TH) {
        u64 carry = 0;
        for (int k = 0; k < WW; k++) {
            const u64 f = F[t][k];
            const u64 st = f & ~((f << 1) | carry);
            carry = f >> 63;
            ST[t][k] = st;
            nrun += (unsigned)__popcll(st);
        }
    }
    {
        unsigned incl = nrun;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; w++) base += wsum[w];
        if (t < TH) rowbase[t] = base + incl - nrun;
        if (t == CF_MAX - 1) rowbase[CF_MAX] = base + incl;         // total number of runs
    }
    __syncthreads();
    const unsigned NR = rowbase[CF_MAX];                            // the virtual outside node
    for (unsigned i = t; i <= NR; i += CF_MAX) parent[i] = i;
    __syncthreads();
    // run id of the free tile (ty, bit pos): starts at or below pos in row ty
    auto run_of = [&](int ty, int pos) -> unsigned {
        unsigned c = 0;
        const int kw = pos >> 6;
        for (int k = 0; k < kw; k++) c += (unsigned)__popcll(ST[ty][k]);
        const u64 m = ((pos & 63) == 63) ? ~0ull : ((1ull << ((pos & 63) + 1)) - 1ull);
        c += (unsigned)__popcll(ST[ty][kw] & m);
        return rowbase[ty] + c - 1u;
    };
    // unions: (a) vertically adjacent free tiles, one union per group of consecutive common bits
    // in a word; (b) runs holding a border tile with the outside
    for (int idx = t; idx < TH * WW; idx += CF_MAX) {
        const int ty = idx / WW, k = idx - ty * WW;
        const u64 f = F[ty][k];
        if (ty > 0) {
            const u64 o = f & F[ty - 1][k];
            u64 g = o & ~(o << 1);                                   // first bit of every group
            while (g) {
                const int b = __ffsll((long long)g) - 1;
                g &= g - 1;
                cfu_union(parent, run_of(ty, 64 * k + b), run_of(ty - 1, 64 * k + b));
            }
        }
        u64 edge = (ty == 0 || ty == TH - 1) ? ~0ull : 0ull;
        if (k == 0) edge |= 1ull;
        if (k == ((W - 1) >> 6)) edge |= 1ull << ((W - 1) & 63);
        u64 e = f & edge;
        e &= ~(e << 1);                                              // one per group is enough
        while (e) {
            const int b = __ffsll((long long)e) - 1;
            e &= e - 1;
            cfu_union(parent, run_of(ty, 64 * k + b), NR);
        }
    }
    __syncthreads();
    // seeds: the start tile of every run connected to the outside; the run fill spreads them
    const unsigned root_out = cfu_find(parent, NR);
    for (int idx = t; idx < TH * WW; idx += CF_MAX) {
        const int ty = idx / WW, k = idx - ty * WW;
        u64 st = ST[ty][k], seed = 0;
        unsigned r = rowbase[ty];
        for (int kk = 0; kk < k; kk++) r += (unsigned)__popcll(ST[ty][kk]);
        while (st) {
            const int b = __ffsll((long long)st) - 1;
            st &= st - 1;
            if (cfu_find(parent, r) == root_out) seed |= 1ull << b;
            r++;
        }
        R[ty][k] = seed;
    }
    __syncthreads();
    if (t < TH) cf_fill_row(F[t], R[t], WW);
    __syncthreads();
    // outputs: state per tile, and the list of unresolved tiles (count pass, wave prefix, writes)
    unsigned mycount = 0;
    for (int idx = wave; idx < TH * WW; idx += nwave) {
        const int ty = idx / WW, k = idx - ty * WW;
        u64 valid = ~0ull;
        if (64 * k + 64 > W) valid = (1ull << (W - 64 * k)) - 1ull;
        mycount += (unsigned)__popcll(~R[ty][k] & valid);
    }
    __syncthreads();
    if (lane == 0) wsum[wave] = mycount;
    __syncthreads();
    unsigned base = 0, total = 0;
    for (int w = 0; w < nwave; w++) { if (w < wave) base += wsum[w]; total += wsum[w]; }
    for (int idx = wave; idx < TH * WW; idx += nwave) {
        const int ty = idx / WW, k = idx - ty * WW, tx = 64 * k + lane;
        const u64 r = R[ty][k];
        u64 valid = ~0ull;
        if (64 * k + 64 > W) valid = (1ull << (W - 64 * k)) - 1ull;
        const u64 un = ~r & valid;
        if (tx < W) {
            const bool reached = (r >> lane) & 1ull;
            state[ty * W + tx] = reached ? 1 : 0;
            if (!reached) tiles[base + (unsigned)__popcll(un & ((1ull << lane) - 1ull))] = (uint32_t)(ty * W + tx);
        }
        base += (unsigned)__popcll(un);
    }
    if (t == 0) counters[CNT_TILES] = (int32_t)total;
}

// R plane init: resolved tiles all ones; unresolved zero; pad bits of the last word one
__global__ __launch_bounds__(256) void k_reach_init(u64* __restrict__ R, const uint8_t* __restrict__ state, int ny, int nx, int W) {
    const size_t total = (size_t)ny * W;
    const u64 pad = (nx & 63) ? ~((1ull << (nx & 63)) - 1) : 0ull;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)((unsigned)i / (unsigned)W), w = (int)((unsigned)i - (unsigned)r * (unsigned)W);   // ny * W < 2^22 (COARSE_MAX tiles)
        u64 v = state[(r >> 6) * W + w] ? ~0ull : 0ull;
        if (w == W - 1) v |= pad;
        R[i] = v;
    }
}

__device__ __forceinline__ u64 reach3(const u64* R, int r, int w, int ny, int W) {
    // OR of rows r-1..r+1 of word w; outside the image everything is reachable
    if (w < 0 || w >= W) return ~0ull;
    u64 v = R[(size_t)r * W + w];
    v |= (r > 0) ? R[(size_t)(r - 1) * W + w] : ~0ull;
    v |= (r < ny - 1) ? R[(size_t)(r + 1) * W + w] : ~0ull;
    return v;
}

__global__ __launch_bounds__(1024) void k_fine_flood(u64* R, const u64* __restrict__ bitsC,
                                                     const uint32_t* __restrict__ tiles, const int32_t* counters,
                                                     int ny, int nx, int W, int32_t* err) {
    __shared__ int changed;
    const int ntile = counters[CNT_TILES];
    const long long nwork = (long long)ntile * 64;
    const u64 pad = (nx & 63) ? ~((1ull << (nx & 63)) - 1) : 0ull;
    const long long max_iter = 64 + nwork;      // every productive sweep adds >= 1 bit to >= 1 word-row
    long long iter = 0;
    for (;; iter++) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (long long k = threadIdx.x; k < nwork; k += blockDim.x) {
            const uint32_t t = tiles[k >> 6];
            const int ty = t / W, w = t - ty * W;
            const int r = ty * 64 + (int)(k & 63);
            if (r >= ny) continue;
            u64 bg = ~bitsC[(size_t)r * W + w];
            if (w == W - 1) bg |= pad;
            const u64 cur = R[(size_t)r * W + w];
            const u64 n = reach3(R, r, w, ny, W);
            const u64 nl = reach3(R, r, w - 1, ny, W);
            const u64 nr = reach3(R, r, w + 1, ny, W);
            u64 x = (cur | n | (n << 1) | (n >> 1) | (nl >> 63) | (nr << 63)) & bg;
            for (;;) {                       // fill along the row inside the word
                const u64 yv = (x | (x << 1) | (x >> 1)) & bg;
                if (yv == x) break;
                x = yv;
            }
            if (x != cur) { R[(size_t)r * W + w] = x; changed = 1; }
        }
        __syncthreads();
        const int c = changed;
        __syncthreads();
        if (!c) break;
        if (iter > max_iter) { if (threadIdx.x == 0) atomicOr(err, BBX_DERR_NOTCONV); break; }
    }
}

// new mask pixels: closing-added or hole pixels (= not reached) where the mask is 0
__global__ __launch_bounds__(256) void k_fill_apply(const u64* __restrict__ R, const uint32_t* __restrict__ tiles,
                                                    const int32_t* counters, uint8_t* mask, int ny, int nx, int W) {
    const long long nwork = (long long)counters[CNT_TILES] * 64;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nwork; k += (long long)gridDim.x * blockDim.x) {
        const uint32_t t = tiles[k >> 6];
        const int ty = t / W, w = t - ty * W;
        const int r = ty * 64 + (int)(k & 63);
        if (r >= ny) continue;
        u64 h = ~R[(size_t)r * W + w];
        while (h) {
            const int b = __ffsll((long long)h) - 1;
            h &= h - 1;
            const int X = w * 64 + b;
            if (X < nx) {
                const size_t q = (size_t)r * nx + X;
                if (mask[q] == 0) mask[q] = BBX_MASK_SATCON;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// number of 8-connected objects in a sparse pixel list: index map + lock-free union-find
// (ndimage.label(..., structure=ones((3,3)))[1])
// ---------------------------------------------------------------------------------
// imap[pixel] = 1 + the list position of a listed pixel, 0 elsewhere: one word per pixel of the frame, of which a call
// touches the lines around its listed pixels -- a plain store per pixel to fill it and at most four plain loads (three of
// them in one line) to find a pixel's earlier neighbours, where the hash table of rounds 1-3 took a compare-and-swap and
// a probe sequence per key.  The map is all-zero between calls: k_cc_count clears the words its call set.
__global__ __launch_bounds__(256) void k_cc_insert(const uint32_t* __restrict__ list, const int32_t* __restrict__ cnt,
                                                   uint32_t* __restrict__ imap, uint32_t npix, int cap, int32_t* err, int32_t* out) {
    int n = *cnt;
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = 0;                 // summed into by k_cc_count
    if (n > cap) { n = cap; if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(err, BBX_DERR_LIST_OVERFLOW); }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = list[i];
        if (p < npix) imap[p] = (uint32_t)i + 1u; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);     // (a pixel outside the frame: never from this library's lists)
    }
}

// (device-scope relaxed loads: served by the L2 the atomics go to; `volatile` would make them
// system-scope accesses)
__device__ __forceinline__ uint32_t cc_load(const uint32_t* q) {
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t cc_find(uint32_t* parent, uint32_t i) {
    uint32_t p = cc_load(&parent[i]);
    while (p != i) {
        const uint32_t gp = cc_load(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[i], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving (benign race)
        i = p; p = gp;
    }
    return i;
}

// Links in two steps (the decision tree of raster-scan labelling).  With a = NW, b = N, c = NE,
// d = W of pixel e, every pixel only has to make sure that e and its earlier neighbours end up
// in one tree, given that each of those does the same for its own earlier neighbours:
//   b present                -> e - b            (a, c, d all touch b)
//   b absent, c present      -> e - c, and c - a (or c - d when a is absent; d touches a)
//   b, c absent              -> e - a, else e - d (d touches a)
// The first link of each pixel is a plain store into its own parent slot (every pixel points
// at an earlier pixel of the raster: a forest), done by k_cc_link for all pixels before any
// tree is searched.  Only the pixels of the middle case need a real union afterwards
// (k_cc_union) -- a few per object instead of four contended compare-and-swap loops per pixel.
__global__ __launch_bounds__(256) void k_cc_link(const uint32_t* __restrict__ list, const int32_t* __restrict__ cnt,
                                                 const uint32_t* __restrict__ imap, uint32_t* __restrict__ parent,
                                                 int32_t* __restrict__ pend, int ny, int nx, int cap) {
    const int n = (*cnt > cap) ? cap : *cnt;
    const uint32_t npix = (uint32_t)ny * (uint32_t)nx;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = list[i];
        const int Y = p / nx, X = p - Y * nx;
        int link = i, other = -1;
        if (p >= npix) { parent[i] = (uint32_t)i; pend[i] = -1; continue; }
        if (Y > 0) {
            const uint32_t up = p - (uint32_t)nx;
            const int jb = (int)imap[up] - 1;
            if (jb >= 0) link = jb;
            else {
                const int jc = (X + 1 < nx) ? (int)imap[up + 1u] - 1 : -1;
                const int ja = (X > 0) ? (int)imap[up - 1u] - 1 : -1;
                if (jc >= 0) {
                    link = jc;
                    other = ja >= 0 ? ja : ((X > 0) ? (int)imap[p - 1u] - 1 : -1);
                } else if (ja >= 0) link = ja;
            }
        }
        if (link == i && X > 0) {                                 // nothing in the row above
            const int jd = (int)imap[p - 1u] - 1;
            if (jd >= 0) link = jd;
        }
        parent[i] = (uint32_t)link;
        pend[i] = other;
    }
}

// every pixel -> the root of its tree.  A chain of first links is as long as the edge it follows (tens of pixels around
// a star, thousands along a trail); the searches of the kernels that follow would each walk it with device-scope loads.
// Here the walk uses plain loads and only the walker's own word is written: whatever a load returns -- the first link or a
// root another thread has put there already -- is an ancestor, so the walk ends at the root whichever it sees.
__global__ __launch_bounds__(256) void k_cc_flatten(const int32_t* __restrict__ cnt, uint32_t* parent, int cap) {
    const int n = (*cnt > cap) ? cap : *cnt;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t p = parent[i];
        if (p == (uint32_t)i) continue;
        for (;;) { const uint32_t gp = parent[p]; if (gp == p) break; p = gp; }
        parent[i] = p;
    }
}

__global__ __launch_bounds__(256) void k_cc_union(const int32_t* __restrict__ cnt, uint32_t* parent,
                                                  const int32_t* __restrict__ pend, int cap) {
    const int n = (*cnt > cap) ? cap : *cnt;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = pend[i];
        if (j < 0) continue;
        uint32_t a = (uint32_t)i, b = (uint32_t)j;
        for (;;) {
            a = cc_find(parent, a); b = cc_find(parent, b);
            if (a == b) break;
            if (a < b) { uint32_t t = a; a = b; b = t; }      // a > b: hang a under b
            const uint32_t old = atomicCAS(&parent[a], a, b);
            if (old == a) break;
        }
    }
}

// roots = objects (a pixel that is its own parent, flattened or not); the pass also clears the words of the
// index map this list set, so the map is all-zero again for the next call (no memset of the frame-sized map per call)
__global__ __launch_bounds__(256) void k_cc_count(const int32_t* __restrict__ cnt, const uint32_t* __restrict__ parent,
                                                  const uint32_t* __restrict__ list, uint32_t* __restrict__ imap,
                                                  uint32_t npix, int32_t* out, int cap) {
    const int n = (*cnt > cap) ? cap : *cnt;
    int c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        c += (parent[i] == (uint32_t)i) ? 1 : 0;
        const uint32_t p = list[i];
        if (p < npix) imap[p] = 0u;
    }
    // one add per workgroup (every returning or same-address atomic queues up at ~11 ns: 45 us for one per wave)
    __shared__ int s_c[4];
    c = wave_sum_i32(c);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { const int t = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]); if (t) atomicAdd(out, t); }
}

// compaction of (mask & bit) pixels into a list
// pixels whose mask byte holds all of [bit] -> list (any order).  16 mask bytes per thread and
// load; a wave that found something reserves its entries with one atomic.  from16/to16 are in
// units of 16 bytes ([mask] 16-byte aligned); k_compact_bit_bytes takes what is left.
__global__ __launch_bounds__(256) void k_compact_bit(const uint8_t* __restrict__ mask, size_t n16, int bit,
                                                     uint32_t* list, int32_t* cnt, uint32_t cap, int32_t* err) {
    const uint4* m16 = (const uint4*)mask;
    const unsigned B = (unsigned)bit * 0x01010101u;
    const int lane = threadIdx.x & 63;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nend = ((n16 + stride - 1) / stride) * stride;          // every wave runs the same number of rounds
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nend; i += stride) {
        uint4 w = make_uint4(0, 0, 0, 0);
        if (i < n16) w = m16[i];
        unsigned h[4]; int c = 0;
        const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned t = (ww[k] & B) ^ B;                          // byte == 0 <=> all bits of [bit] set
            const unsigned nz = (((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;
            h[k] = (i < n16) ? (~nz & 0x80808080u) : 0u;
            c += __popc(h[k]);
        }
        if (__ballot(c > 0) == 0) continue;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        const int total = __shfl(incl, 63);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd((unsigned*)cnt, (unsigned)total);
        base = (unsigned)__shfl((int)base, 0) + (unsigned)(incl - c);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            unsigned hk = h[k];
            while (hk) {
                const int by = (__ffs((int)hk) - 1) >> 3;
                hk &= hk - 1;
                if (base < cap) list[base] = (uint32_t)(i * 16 + k * 4 + by); else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
                base++;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_compact_bit_bytes(const uint8_t* __restrict__ mask, size_t from, size_t npix, int bit,
                                                           uint32_t* list, int32_t* cnt, uint32_t cap, int32_t* err) {
    for (size_t i = from + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        if ((mask[i] & bit) == bit) {
            unsigned k = atomicAdd((unsigned*)cnt, 1u);
            if (k < cap) list[k] = (uint32_t)i; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
        }
    }
}

// count objects of a device-resident list (length in *d_cnt, at most cap entries are
// considered; more raises the overflow flag) -> *d_out
int bbx_cc_count_list(bbx_ctx* ctx, const uint32_t* d_list, const int32_t* d_cnt, size_t cap, int ny, int nx,
                      int32_t* d_out, hipStream_t s) {
    int rc;
    const size_t npix = (size_t)ny * nx;
    if (ny < 1 || nx < 1 || npix >= 0xffffffffull) return BBX_ERR_ARG;
    uint32_t* imap = (uint32_t*)bbx_ws(ctx, WS_HASH, npix * sizeof(uint32_t), &rc); if (rc) return rc;
    uint32_t* parent = (uint32_t*)bbx_ws(ctx, WS_PARENT, 2 * cap * sizeof(uint32_t) + 16, &rc); if (rc) return rc;
    int32_t* pend = (int32_t*)(parent + cap);
    // the index map is kept all-zero between calls (k_cc_count clears what its call set); it is filled once when the
    // workspace block is new or grew
    if (ctx->hash_clean_ptr != (void*)imap || ctx->hash_clean_n < npix) {
        BBX_HIP(hipMemsetAsync(imap, 0, ctx->ws_bytes[WS_HASH], s));
        ctx->hash_clean_ptr = (void*)imap; ctx->hash_clean_n = ctx->ws_bytes[WS_HASH] / sizeof(uint32_t);
    }
    const unsigned grid = 1024;
    hipLaunchKernelGGL(k_cc_insert, dim3(grid), dim3(256), 0, s, d_list, d_cnt, imap, (uint32_t)npix, (int)cap, ctx->d_err, d_out);
    hipLaunchKernelGGL(k_cc_link, dim3(grid), dim3(256), 0, s, d_list, d_cnt, imap, parent, pend, ny, nx, (int)cap);
    hipLaunchKernelGGL(k_cc_flatten, dim3(grid), dim3(256), 0, s, d_cnt, parent, (int)cap);
    hipLaunchKernelGGL(k_cc_union, dim3(grid), dim3(256), 0, s, d_cnt, parent, pend, (int)cap);
    // (the count needs no second flatten: a root is a root; the callers that go on to read every pixel's root ask for it)
    if (ctx->cc_roots) hipLaunchKernelGGL(k_cc_flatten, dim3(grid), dim3(256), 0, s, d_cnt, parent, (int)cap);
    ctx->cc_roots = 0;
    hipLaunchKernelGGL(k_cc_count, dim3(grid / 4), dim3(256), 0, s, d_cnt, parent, d_list, imap, (uint32_t)npix, d_out, (int)cap);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

// ---- hysteresis + small-object filter on a pixel list (Canny edges of the satellite-trail detector) ----
// components (8-connected) of the listed pixels; a component stays when one of its pixels carries a flag (the
// high mask) and it has at least min_size pixels; the pixels of the components that stay go to d_out
__global__ __launch_bounds__(256) void k_ccf_acc(const int32_t* __restrict__ cnt, int cap, const uint32_t* __restrict__ parent, const uint8_t* __restrict__ flag,
                                                 uint32_t* size, uint32_t* has) {
    const int n = (*cnt > cap) ? cap : *cnt;
    const int lane = threadIdx.x & 63;
    const int nround = (n + (int)(gridDim.x * blockDim.x) - 1) / (int)(gridDim.x * blockDim.x);
    for (int r = 0; r < nround; r++) {
        const int i = (r * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        bool on = i < n;
        uint32_t root = 0;
        if (on) { root = parent[i]; if (flag[i]) has[root] = 1u; }       // (its root: k_cc_flatten ran after the unions)
        // neighbours in the list mostly belong to the same component: one add per root and wave (an add per pixel on the
        // few roots of long edges queues up at ~11 ns each: 117 us for the edges of a frame)
        unsigned long long m = __builtin_amdgcn_ballot_w64(on);
        while (m) {
            const int leader = (int)__builtin_ctzll(m);
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)root, leader);
            const unsigned long long same = __builtin_amdgcn_ballot_w64(on && root == r0);
            if (lane == leader) atomicAdd(&size[r0], (unsigned)__popcll(same));
            on = on && root != r0;
            m &= ~same;
        }
    }
}
__global__ __launch_bounds__(256) void k_ccf_emit(const uint32_t* __restrict__ list, const int32_t* __restrict__ cnt, int cap, const uint32_t* __restrict__ parent,
                                                  const uint32_t* __restrict__ size, const uint32_t* __restrict__ has, uint32_t min_size,
                                                  uint32_t* out, int32_t* out_cnt, uint32_t out_cap, int32_t* err) {
    const int n = (*cnt > cap) ? cap : *cnt;
    const int lane = threadIdx.x & 63;
    const int nround = (n + (int)(gridDim.x * blockDim.x) - 1) / (int)(gridDim.x * blockDim.x);
    for (int r = 0; r < nround; r++) {
        const int i = (r * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        bool keep = false;
        if (i < n) { const uint32_t root = parent[i]; keep = has[root] && size[root] >= min_size; }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (!m) continue;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd((unsigned*)out_cnt, (unsigned)__popcll(m));
        base = __shfl(base, 0, 64);
        if (keep) {
            const unsigned pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (pos < out_cap) out[pos] = list[i]; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
        }
    }
}

int bbx_cc_filter_list(bbx_ctx* ctx, const uint32_t* d_list, const int32_t* d_cnt, size_t cap, int ny, int nx, const uint8_t* d_flag,
                       int min_size, uint32_t* d_out, int32_t* d_out_cnt, uint32_t out_cap, hipStream_t s) {
    ctx->cc_roots = 1;
    int rc = bbx_cc_count_list(ctx, d_list, d_cnt, cap, ny, nx, &ctx->d_counters[CNT_CC_ROOTS], s); if (rc) return rc;
    uint32_t* parent = (uint32_t*)ctx->d_ws[WS_PARENT];
    uint32_t* size = (uint32_t*)bbx_ws(ctx, WS_STAGE2, 2 * cap * sizeof(uint32_t), &rc); if (rc) return rc;
    uint32_t* has = size + cap;
    BBX_HIP(hipMemsetAsync(size, 0, 2 * cap * sizeof(uint32_t), s));
    BBX_HIP(hipMemsetAsync(d_out_cnt, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_ccf_acc, dim3(1024), dim3(256), 0, s, d_cnt, (int)cap, parent, d_flag, size, has);
    hipLaunchKernelGGL(k_ccf_emit, dim3(1024), dim3(256), 0, s, d_list, d_cnt, (int)cap, parent, size, has, (uint32_t)min_size, d_out, d_out_cnt,
                       out_cap, ctx->d_err);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

// ---- peaks of connected regions of |img| >= thr (transient candidates on S_corr) ---------
// Hits go through a per-workgroup LDS queue (wave ballots + an LDS counter) that is flushed with
// one global reservation per ~1024 entries: a returning atomic per hit on one global counter
// retires at ~11 ns each, 10 ms for the 10^6 pixels above a detection threshold.
__global__ __launch_bounds__(256) void k_compact_abs(const float* __restrict__ img, size_t npix, float thr,
                                                     uint32_t* list, int32_t* cnt, uint32_t cap, int32_t* err) {
    __shared__ uint32_t q[2048];
    __shared__ unsigned qn, gbase;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) qn = 0;
    __syncthreads();
    const size_t step = (size_t)gridDim.x * 1024;
    const size_t nround = (npix + step - 1) / step * step;
    const bool vec = (((uintptr_t)img) & 15) == 0;
    for (size_t b = (size_t)blockIdx.x * 1024; b < nround; b += step) {
        // 1024 pixels per workgroup and round: one 16-byte load per thread where the frame allows it
        float v[4];
        const size_t i0 = b + (size_t)tid * 4;
        if (vec && i0 + 3 < npix) { const float4 f = *(const float4*)(img + i0); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
        else {
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = i0 + r < npix ? img[i0 + r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const bool hit = i0 + r < npix && fabsf(v[r]) >= thr;   // NaN compares false
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (m) {
                unsigned wb = 0;
                if (lane == 0) wb = atomicAdd(&qn, (unsigned)__popcll(m));
                wb = __shfl(wb, 0, 64);
                if (hit) q[wb + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (uint32_t)(i0 + r);
            }
        }
        __syncthreads();
        const unsigned n = qn;
        if (n >= 1024 || b + step >= nround) {                   // workgroup-uniform
            if (n) {
                if (tid == 0) gbase = atomicAdd((unsigned*)cnt, n);
                __syncthreads();
                const unsigned g0 = gbase;
                for (unsigned t = tid; t < n; t += 256) {
                    if (g0 + t < cap) list[g0 + t] = q[t]; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
                }
            }
            __syncthreads();
            if (tid == 0) qn = 0;
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void k_cc_peak(const uint32_t* __restrict__ list, const int32_t* __restrict__ cnt, int cap,
                                                 const uint32_t* __restrict__ parent, const float* __restrict__ img,
                                                 unsigned long long* __restrict__ best) {
    const int n = (*cnt > cap) ? cap : *cnt;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = list[i];
        const uint32_t root = parent[i];
        // largest |value| wins, ties go to the smallest pixel index (first in C order)
        const unsigned long long key = ((unsigned long long)__float_as_uint(fabsf(img[p])) << 32) | (0xffffffffu - p);
        atomicMax(&best[root], key);
    }
}

__global__ __launch_bounds__(256) void k_cc_peak_emit(const int32_t* __restrict__ cnt, int cap, const uint32_t* __restrict__ parent,
                                                      const unsigned long long* __restrict__ best, const float* __restrict__ img,
                                                      int nx, int max_out, int32_t* yx, float* val, int32_t* nout) {
    const int n = (*cnt > cap) ? cap : *cnt;
    const int lane = threadIdx.x & 63;
    const int step = (int)(gridDim.x * blockDim.x), nround = (n + step - 1) / step;
    for (int r = 0; r < nround; r++) {
        const int i = r * step + (int)(blockIdx.x * blockDim.x + threadIdx.x);
        const bool root = i < n && parent[i] == (uint32_t)i;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(root);          // one reservation per wave
        if (!m) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(nout, (int)__popcll(m));
        base = __shfl(base, 0, 64);
        if (root) {
            const int k = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const uint32_t p = 0xffffffffu - (uint32_t)(best[i] & 0xffffffffull);
            if (k < max_out) { yx[2 * k] = (int)(p / nx); yx[2 * k + 1] = (int)(p % nx); val[k] = img[p]; }
        }
    }
}

// the threshold a candidate list was made with against the one the caller passes
__global__ void k_thr_check(const float* __restrict__ med, double nsig, float thr, int32_t* err) {
    if (threadIdx.x == 0 && !((float)((double)med[0] * nsig) == thr)) atomicOr(err, BBX_DERR_LIST_OVERFLOW);
}

extern "C" int bbx_find_peaks(bbx_ctx* ctx, int ny, int nx, const float* d_img, float thr, int max_out, int32_t* d_yx,
                              float* d_val, int32_t* d_count, void* stream) {
    if (!ctx || !d_img || !d_yx || !d_val || !d_count || ny < 1 || nx < 1 || max_out < 1) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const size_t npix = (size_t)ny * nx;
    if (npix >= 0xffffffffull) return BBX_ERR_ARG;
    const size_t cap = npix / 16 + 1024;
    uint32_t* list = (uint32_t*)bbx_ws(ctx, WS_CCLIST, cap * sizeof(uint32_t), &rc); if (rc) return rc;
    unsigned long long* best = (unsigned long long*)bbx_ws(ctx, WS_STAGE2, cap * sizeof(unsigned long long), &rc); if (rc) return rc;
    int32_t* cnt = &ctx->d_counters[CNT_CC_N];
    BBX_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), s));
    if (ctx->zcand_img == d_img && ctx->zcand_thr_used == thr && ctx->zcand_npix == npix && ctx->d_ws[WS_ZCAND]) {
        // the frame is the Scorr image of the last bbx_zogy_frame call, which listed these pixels as it wrote them
        // (bbx_zogy_candidates): no pass over the frame
        list = (uint32_t*)ctx->d_ws[WS_ZCAND];
        cnt = &ctx->d_counters[CNT_ZCAND];
        ctx->zcand_img = nullptr;
    } else if (ctx->bcand_img == d_img && ctx->bcand_npix == npix && ctx->d_ws[WS_BCAND]) {
        // the frame is the background-subtracted image of the last bbx_spline_zoom_sub call, which listed the pixels above
        // (float)(median * nsigma) as it wrote them (bbx_zoom_candidates): no pass over the frame -- provided the caller's
        // threshold is that very number (checked on the device: the list-overflow flag otherwise, i.e. the caller sees a failed search, not a wrong one)
        list = (uint32_t*)ctx->d_ws[WS_BCAND];
        cnt = &ctx->d_counters[CNT_BCAND];
        hipLaunchKernelGGL(k_thr_check, dim3(1), dim3(64), 0, s, ctx->bcand_img_med, ctx->bcand_img_nsig, thr, ctx->d_err);
        ctx->bcand_img = nullptr;
    } else {
        BBX_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t), s));
        hipLaunchKernelGGL(k_compact_abs, dim3(2048), dim3(256), 0, s, d_img, npix, thr, list, cnt, (uint32_t)cap, ctx->d_err);
    }
    // union-find over the list (the object count itself is not needed: reuse its scratch)
    ctx->cc_roots = 1;
    rc = bbx_cc_count_list(ctx, list, cnt, cap, ny, nx, &ctx->d_counters[CNT_CC_ROOTS], s); if (rc) return rc;
    uint32_t* parent = (uint32_t*)ctx->d_ws[WS_PARENT];
    BBX_HIP(hipMemsetAsync(best, 0, cap * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_cc_peak, dim3(1024), dim3(256), 0, s, list, cnt, (int)cap, parent, d_img, best);
    hipLaunchKernelGGL(k_cc_peak_emit, dim3(1024), dim3(256), 0, s, cnt, (int)cap, parent, best, d_img, nx, max_out, d_yx, d_val, d_count);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

#define MC_BLOCKS 512
// per-workgroup counts of the six reported bits -> partial[block][6]; no atomics (6 counters in
// one cache line serialise ~11 ns per update)
__global__ __launch_bounds__(256) void k_mask_counts(const uint8_t* __restrict__ mask, size_t n4,
                                                     unsigned long long* __restrict__ partial) {
    // bits in reporting order: bad, edge, saturated, saturated-connected, satellite, cosmic
    const unsigned bits[6] = {1u, 32u, 4u, 8u, 16u, 2u};
    unsigned c[6] = {0, 0, 0, 0, 0, 0};                         // < 2^32: a thread sees < 2^28 bytes (checked on the host)
    const uint32_t* m4 = (const uint32_t*)mask;
    const size_t n16 = n4 / 4;                                  // 16-byte groups
    const uint4* m16 = (const uint4*)mask;
    const bool al16 = (((uintptr_t)mask) & 15) == 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (al16) {
        size_t i = t0;
        for (; i + 3 * stride < n16; i += 4 * stride) {
            uint4 w[4];
#pragma unroll
            for (int u = 0; u < 4; u++) w[u] = m16[i + u * stride];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const unsigned b = bits[k] * 0x01010101u;
                    c[k] += __popc(w[u].x & b) + __popc(w[u].y & b) + __popc(w[u].z & b) + __popc(w[u].w & b);
                }
        }
        for (; i < n16; i += stride) {
            const uint4 w = m16[i];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const unsigned b = bits[k] * 0x01010101u;
                c[k] += __popc(w.x & b) + __popc(w.y & b) + __popc(w.z & b) + __popc(w.w & b);
            }
        }
    }
    for (size_t i = (al16 ? n16 * 4 : 0) + t0; i < n4; i += stride) {
        const uint32_t w = m4[i];
#pragma unroll
        for (int k = 0; k < 6; k++) c[k] += __popc(w & (bits[k] * 0x01010101u));
    }
    __shared__ long long sh[6][4];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const long long v = wave_sum_i64((long long)c[k]);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6)
        partial[(size_t)blockIdx.x * 6 + threadIdx.x] = (unsigned long long)(sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// one workgroup: folds the partial counts, adds the bytes behind the last full 4-byte word
__global__ __launch_bounds__(64) void k_mask_counts_fold(const unsigned long long* __restrict__ partial, int nblocks,
                                                         const uint8_t* __restrict__ mask, size_t from, size_t n,
                                                         unsigned long long* __restrict__ out) {
    const unsigned bits[6] = {1u, 32u, 4u, 8u, 16u, 2u};
#pragma unroll
    for (int k = 0; k < 6; k++) {
        long long v = 0;
        for (int b = threadIdx.x; b < nblocks; b += 64) v += (long long)partial[(size_t)b * 6 + k];
        v = wave_sum_i64(v);
        if (threadIdx.x == 0) {
            for (size_t i = from; i < n; i++) if (mask[i] & bits[k]) v += 1;
            out[k] = (unsigned long long)v;
        }
    }
}

extern "C" {

int bbx_mask_finish(bbx_ctx* ctx, const bbx_geom* g, uint8_t* d_mask, int32_t* d_nobj_sat, void* stream) {
    if (!ctx || !d_mask || !d_nobj_sat) return BBX_ERR_ARG;
    bbx_dims d; int rc = bbx_make_dims(g, &d); if (rc) return rc;
    if (!ctx->d_satlist) return BBX_ERR_ARG;                 // bbx_calibrate must have run
    hipStream_t s = (hipStream_t)stream;
    const int W = (d.nx + 63) / 64, TH = (d.ny + 63) / 64;
    if ((size_t)W * TH > COARSE_MAX) return BBX_ERR_ARG;
    const size_t nwords = (size_t)d.ny * W;
    u64* bitsM = (u64*)bbx_ws(ctx, WS_BITS_M, nwords * 8, &rc); if (rc) return rc;
    u64* bitsC = (u64*)bbx_ws(ctx, WS_BITS_C, nwords * 8, &rc); if (rc) return rc;
    u64* bitsR = (u64*)bbx_ws(ctx, WS_BITS_R, nwords * 8, &rc); if (rc) return rc;
    const size_t o_tiles = ((size_t)2 * W * TH + 15) & ~(size_t)15, o_free = o_tiles + (size_t)W * TH * sizeof(uint32_t);
    char* tws = (char*)bbx_ws(ctx, WS_TILES, o_free + (size_t)TH * ((W + 63) / 64) * 8 + 64, &rc); if (rc) return rc;
    uint8_t* occ = (uint8_t*)tws; uint8_t* state = occ + (size_t)W * TH;
    uint32_t* tiles = (uint32_t*)(tws + o_tiles);
    u64* freebits = (u64*)(tws + o_free);
    // (tiles offset may exceed the requested size by the alignment slack; bbx_ws over-allocates by 1/8 + 256)
    BBX_HIP(hipMemsetAsync(bitsM, 0, nwords * 8, s));
    hipLaunchKernelGGL(k_sat_scatter, dim3(512), dim3(256), 0, s, ctx->d_satlist, ctx->d_counters, d, d_mask, bitsM, W, (int)ctx->cap_satlist);
    // NOBJ-SAT: objects of the saturated pixels themselves (blackbox.py:4544-4550)
    rc = bbx_cc_count_list(ctx, ctx->d_satlist, &ctx->d_counters[CNT_SAT],
                           (size_t)ctx->cap_satlist > (1u << 22) ? (size_t)(1u << 22) : (size_t)ctx->cap_satlist,
                           d.ny, d.nx, d_nobj_sat, s);
    if (rc) return rc;
    const unsigned gw = (unsigned)((nwords + 255) / 256 > 4096 ? 4096 : (nwords + 255) / 256);
    hipLaunchKernelGGL(k_bits_close, dim3(gw), dim3(256), 0, s, bitsM, bitsC, d.ny, d.nx, W);
    hipLaunchKernelGGL(k_tile_occupancy, dim3((W + 63) / 64, TH), dim3(256), 0, s, bitsC, occ, freebits, d.ny, W, TH);
    if (W <= CF_MAX && TH <= CF_MAX) {
        const size_t lds = ((size_t)TH * (W / 2 + 1) + 2) * sizeof(unsigned);
        BBX_HIP(hipFuncSetAttribute((const void*)k_coarse_cc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_coarse_cc, dim3(1), dim3(CF_MAX), lds, s, freebits, state, W, TH, tiles, ctx->d_counters);
    }
    else
        hipLaunchKernelGGL(k_coarse_flood, dim3(1), dim3(1024), 0, s, occ, state, W, TH, tiles, ctx->d_counters, ctx->d_err);
    hipLaunchKernelGGL(k_reach_init, dim3(gw), dim3(256), 0, s, bitsR, state, d.ny, d.nx, W);
    hipLaunchKernelGGL(k_fine_flood, dim3(1), dim3(1024), 0, s, bitsR, bitsC, tiles, ctx->d_counters, d.ny, d.nx, W, ctx->d_err);
    hipLaunchKernelGGL(k_fill_apply, dim3(256), dim3(256), 0, s, bitsR, tiles, ctx->d_counters, d_mask, d.ny, d.nx, W);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_count_objects(bbx_ctx* ctx, int ny, int nx, const uint8_t* d_mask, int bit, int32_t* d_count, void* stream) {
    if (!ctx || !d_mask || !d_count || ny <= 0 || nx <= 0 || bit <= 0 || bit > 255) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const size_t npix = (size_t)ny * nx;
    if (npix >= 0xffffffffull) return BBX_ERR_ARG;
    // list capacity: 1/8 of the frame is far beyond any real mask plane; more is
    // reported as BBX_ERR_OVERFLOW by bbx_sync
    const size_t cap = npix / 8 + 1024;
    uint32_t* list = (uint32_t*)bbx_ws(ctx, WS_CCLIST, cap * sizeof(uint32_t), &rc); if (rc) return rc;
    BBX_HIP(hipMemsetAsync(&ctx->d_counters[CNT_CC_N], 0, sizeof(int32_t), s));
    const size_t n16 = (((uintptr_t)d_mask) & 15) ? 0 : (size_t)npix / 16;
    if (n16)
        hipLaunchKernelGGL(k_compact_bit, dim3(2048), dim3(256), 0, s, d_mask, n16, bit, list, &ctx->d_counters[CNT_CC_N],
                           (uint32_t)cap, ctx->d_err);
    if (n16 * 16 < (size_t)npix)
        hipLaunchKernelGGL(k_compact_bit_bytes, dim3(n16 ? 1 : 2048), dim3(256), 0, s, d_mask, n16 * 16, (size_t)npix, bit, list,
                           &ctx->d_counters[CNT_CC_N], (uint32_t)cap, ctx->d_err);
    BBX_LAUNCH_CHECK();
    return bbx_cc_count_list(ctx, list, &ctx->d_counters[CNT_CC_N], cap, ny, nx, d_count, s);
}

int bbx_mask_counts(bbx_ctx* ctx, int64_t npix, const uint8_t* d_mask, int64_t* d_counts, void* stream) {
    if (!ctx || !d_mask || !d_counts || npix <= 0) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t n4 = (size_t)npix / 4;
    if (((uintptr_t)d_mask) % 4 || (size_t)npix > ((size_t)1 << 40)) return BBX_ERR_ARG;
    int rc;
    unsigned long long* partial = (unsigned long long*)bbx_ws(ctx, WS_MISC, (size_t)MC_BLOCKS * 6 * 8, &rc); if (rc) return rc;
    hipLaunchKernelGGL(k_mask_counts, dim3(MC_BLOCKS), dim3(256), 0, s, d_mask, n4, partial);
    hipLaunchKernelGGL(k_mask_counts_fold, dim3(1), dim3(64), 0, s, partial, MC_BLOCKS, d_mask, n4 * 4, (size_t)npix,
                       (unsigned long long*)d_counts);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

}  // extern "C"
