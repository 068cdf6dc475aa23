// bbx_bsel.h -- "bracketed select": exact order statistics of image segments without a
// dedicated pass over the frame.
//
//   1. sample   : BSEL_S strided samples per segment                      (tiny)
//   2. bracket  : [lo, hi] = sample order statistics around the target quantile,
//                 wide enough to hold the wanted rank with overwhelming probability
//   3. feed     : any kernel that streams the segment anyway calls bsel_feed() per pixel:
//                 counts values below lo and stages the values inside the bracket
//                 (~5 % of the pixels) in LDS; bsel_drain() moves them to a side buffer
//                 with ONE global atomic per workgroup -- no extra HBM read of the frame.
//                 Same-cache-line atomics retire at ~11 ns each on MI355X whatever the
//                 address inside the line, so the counters and the side buffer are split
//                 into BSEL_NSH shards (one 64-byte line each, picked by workgroup index)
//   4. finish   : rank' = rank - below; exact radix select inside the small buffer.
//                 If the rank fell outside the bracket (or a buffer overflowed) the
//                 segment is flagged and the full 3-pass radix select over the frame runs
//                 instead, so the result is exact in every case.
#pragma once
#include "bbx_common.h"

#define BSEL_S 16384
#define BSEL_MAXSEG 64
#define BSEL_LBUF 6144            // LDS staging entries per workgroup (24 KB)
#define BSEL_NSH 64               // shards per segment

struct bsel_seg {
    float lo, hi;                 // closed bracket
    uint32_t nsample;             // valid samples
    uint32_t nbuf;                // values in the side buffer    } sums over the shards,
    unsigned long long below;     // valid values < lo            } formed by the plan kernel
    unsigned long long n;         // valid values                 }
    uint32_t fail;                // 1 -> full select required
    uint32_t pad;
    float result[2];              // lower / upper middle element (ranks (n-1)/2 and n/2)
    double wlo, whi;              // value window of the valid pixels (sigma clipping); +-inf = none
};

struct bsel_shard {               // exactly one cache line
    uint32_t nbuf;                // values appended to this shard's region (may exceed capS -> fail)
    uint32_t pad0;
    unsigned long long below, n;
    uint32_t pad[10];
};

struct bsel_dev {                 // passed by value to feeding kernels
    bsel_seg* seg;                // [nseg]
    bsel_shard* shard;            // [nseg][BSEL_NSH]
    float* buf;                   // [nseg][BSEL_NSH][capS]
    uint32_t cap;                 // = BSEL_NSH * capS (segment stride of buf)
    uint32_t capS;
    int ysz, xsz, SX;             // segment rectangles
    int stride;                   // elements between image rows (>= SX * xsz)
    int skip_zero;                // 1: pixels equal to 0 are invalid (astropy mask_value=0)
};

// value-side validity shared by the generic feeders: not NaN, inside the segment's window,
// not the masked value
__device__ __forceinline__ bool bsel_value_ok(float v, double wlo, double whi, int skip_zero) {
    const double d = (double)v;
    return (v == v) && d >= wlo && d <= whi && !(skip_zero && v == 0.f);
}

__device__ __forceinline__ unsigned bsel_my_shard() {
    return (blockIdx.x + blockIdx.y * gridDim.x) & (BSEL_NSH - 1);
}
// reserve [count] slots in shard [sh] of segment [seg]; returns the first index inside the region
__device__ __forceinline__ unsigned bsel_reserve(const bsel_dev& b, int seg, unsigned sh, unsigned count) {
    return atomicAdd(&b.shard[seg * BSEL_NSH + sh].nbuf, count);
}
__device__ __forceinline__ float* bsel_region(const bsel_dev& b, int seg, unsigned sh) {
    return b.buf + (size_t)seg * b.cap + (size_t)sh * b.capS;
}
__device__ __forceinline__ void bsel_count(const bsel_dev& b, int seg, unsigned sh, unsigned n, unsigned below) {
    bsel_shard* s = &b.shard[seg * BSEL_NSH + sh];
    if (n) atomicAdd(&s->n, (unsigned long long)n);
    if (below) atomicAdd(&s->below, (unsigned long long)below);
}

struct bsel_acc { unsigned n, below; };
struct bsel_lds { float v[BSEL_LBUF]; unsigned cnt; unsigned base; };

__device__ __forceinline__ void bsel_lds_init(bsel_lds& L) {
    if (threadIdx.x == 0) { L.cnt = 0; L.base = 0; }
    __syncthreads();
}

// per pixel; every lane of the wave must call it (ballot)
__device__ __forceinline__ void bsel_feed(bsel_lds& L, float lo, float hi, float v, bool valid, bsel_acc& acc) {
    bool app = false;
    if (valid) {
        acc.n++;
        if (v < lo) acc.below++;
        else if (v <= hi) app = true;
    }
    const unsigned long long m = __ballot(app);
    if (m) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)m) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&L.cnt, (unsigned)__popcll(m));
        base = __shfl(base, leader, 64);
        if (app) {
            const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (pos < BSEL_LBUF) L.v[pos] = v;
        }
    }
}

// workgroup-wide (contains barriers): move the staged values to the side buffer when the
// stage could overflow before the next call ([reserve] = max appends until then) or [force]
__device__ __forceinline__ void bsel_drain(const bsel_dev& b, int seg, bsel_lds& L, unsigned reserve, bool force) {
    __syncthreads();
    const unsigned c = L.cnt;
    if (force || c + reserve > BSEL_LBUF) {
        const unsigned sh = bsel_my_shard();
        if (threadIdx.x == 0) {
            L.base = c ? bsel_reserve(b, seg, sh, c) : 0u;
            if (c > BSEL_LBUF) atomicOr(&b.seg[seg].fail, 1u);       // staged values were dropped
        }
        __syncthreads();
        const unsigned base = L.base, n = c < BSEL_LBUF ? c : BSEL_LBUF;
        float* reg = bsel_region(b, seg, sh);
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned pos = base + i;
            if (pos < b.capS) reg[pos] = L.v[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) L.cnt = 0;
        __syncthreads();
    }
}

__device__ __forceinline__ void bsel_flush(const bsel_dev& b, int seg, bsel_acc& acc) {
    const int n = wave_sum_i32((int)acc.n), bl = wave_sum_i32((int)acc.below);
    if ((threadIdx.x & 63) == 0) bsel_count(b, seg, bsel_my_shard(), (unsigned)n, (unsigned)bl);
    acc.n = 0; acc.below = 0;
}

// host-side entry points (bbx_select.hip)
// [stride] = elements between rows of d_data / d_mask (0: = nx); the ny x nx area is cut into
// (ny/ysz) x (nx/xsz) <= BSEL_MAXSEG segments
int bbx_bsel_prepare(bbx_ctx* ctx, const float* d_data, const uint8_t* d_mask, int ny, int nx, int ysz, int xsz,
                     bsel_dev* out, hipStream_t s, int stride = 0);
int bbx_bsel_finish(bbx_ctx* ctx, const bsel_dev& b, const float* d_data, const uint8_t* d_mask, int ny, int nx,
                    hipStream_t s);
