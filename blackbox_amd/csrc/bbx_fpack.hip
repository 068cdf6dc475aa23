// bbx_fpack.hip -- FITS tile compression on the device: what
// `fpack -q Q -D -Y` / `fpack -D -Y` (blackbox.py:812-857) do to the reduced float image
// and to the uint8 mask, so that only the compressed bytes cross PCIe.
//   RICE_1, one tile per image row (fpack's default tiling), block size 32;
//   float rows: CFITSIO fits_quantize_float with SUBTRACTIVE_DITHER_1 -- noise = smallest
//   non-zero of the 2nd/3rd/5th-order MAD estimates of the row (FnNoise5_float),
//   delta = noise / Q, zero point = min rounded to a multiple of delta,
//   q_i = NINT((x_i - zero) / delta + r_i - 0.5) with the fixed 10000-value random table.
// One workgroup per row; the row lives in LDS from quantisation to the packed bit stream.
// Exact order statistics (the three medians) by 4-pass radix select on the float bit
// patterns.  Held byte-for-byte against CFITSIO through oracle/fpack.py
// (tests/golden/fpack.npz).
#include "bbx_common.h"

#define FP_MAXNX 16384
#ifndef FP_THREADS
#define FP_THREADS 768           // threads per row.  Two rows per CU: 24 of its 32 wave slots, 80 registers per thread (1024 threads:
#endif                           // all slots, 64 registers, the same time alone -- and no slot left for the copy and decode kernels beside it)
#ifndef FP_MINW
#define FP_MINW 6                // waves per SIMD the short-buffer kernel's registers are cut for: two workgroups per CU
#endif
#define FP_NRANDOM 10000
#define FP_NRESERVED 10

struct fp_tile {                 // per-row result
    uint32_t nbytes;             // compressed length
    uint32_t flag;               // 0 ok, 1 = not quantisable (zero noise / range), 2 = NaN/Inf in the row
    double zscale, zzero;
};

__device__ __forceinline__ int fp_nint(double x) { return (x >= 0.) ? (int)(x + 0.5) : (int)(x - 0.5); }

// index into the random table for pixel i of a tile: the sequence starts at
// int(rand[iseed] * 500) and, each time it reaches the end of the table, restarts at
// int(rand[++iseed] * 500)
__device__ __forceinline__ int fp_rand_index(const float* __restrict__ rnd, int iseed, int i) {
    int start = (int)((double)rnd[iseed] * 500.);
    while (i >= FP_NRANDOM - start) {
        i -= FP_NRANDOM - start;
        iseed = (iseed + 1 == FP_NRANDOM) ? 0 : iseed + 1;
        start = (int)((double)rnd[iseed] * 500.);
    }
    return start + i;
}

// sum over the 32 lanes of a half wave (lanes 0-31 / 32-63 separately)
__device__ __forceinline__ unsigned long long half_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// exclusive prefix sum over the 32 lanes of a half wave
__device__ __forceinline__ unsigned half_excl_scan_u32(unsigned v, int l32) {
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (l32 >= o) incl += t; }
    return incl - v;
}

// OR [n] bits (n <= 32) of [val] into the big-endian bit stream at bit position [pos]
__device__ __forceinline__ void put_bits(unsigned* words, unsigned pos, unsigned val, int n) {
    if (n == 0) return;
    const unsigned w = pos >> 5, o = pos & 31;
    const unsigned long long v = ((unsigned long long)val << (64 - n)) >> o;       // aligned to the 64-bit window of words w, w+1
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    if (hi) atomicOr(&words[w], hi);
    if (lo) atomicOr(&words[w + 1], lo);
}

template <int BYTEPIX> struct rice_par;
template <> struct rice_par<1> { static constexpr int fsbits = 3, fsmax = 6, bbits = 8; };
template <> struct rice_par<2> { static constexpr int fsbits = 4, fsmax = 14, bbits = 16; };
template <> struct rice_par<4> { static constexpr int fsbits = 5, fsmax = 25, bbits = 32; };

// zig-zag mapped difference of pixel i to its predecessor, in the integer width of the pixel
template <int BYTEPIX>
__device__ __forceinline__ unsigned rice_diff(const int* vals, int i) {
    const int prev = vals[i ? i - 1 : 0];
    int pd = vals[i] - prev;
    if (BYTEPIX == 1) pd = (int)(signed char)pd;
    if (BYTEPIX == 2) pd = (int)(short)pd;
    unsigned d = (pd < 0) ? ~((unsigned)pd << 1) : ((unsigned)pd << 1);
    if (BYTEPIX == 1) d &= 0xffu;
    if (BYTEPIX == 2) d &= 0xffffu;
    return d;
}

// ---- a wave sorts 1024 keys held 16 per lane (sorted position of register r of lane l: 16 l + r): bitonic network,
// stages between registers of a lane as v_min / v_max (v_med3 against a per-lane 0 / ~0 where the direction depends on the
// lane), stages between lanes through ds_bpermute (the machinery of k_bkg_boxstats with 16 instead of 64 keys per lane)
__device__ __forceinline__ uint32_t fp_umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t o;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
template <int K, int J> __device__ __forceinline__ void fp_stage_static(uint32_t (&k)[16]) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int q = r ^ J;
        if (q > r) {
            const uint32_t a = k[r], b = k[q];
            const uint32_t lo = min(a, b), hi = max(a, b);
            if ((r & K) == 0) { k[r] = lo; k[q] = hi; } else { k[r] = hi; k[q] = lo; }
        }
    }
}
template <int J> __device__ __forceinline__ void fp_stage_lane(uint32_t (&k)[16], uint32_t clo) {
    const uint32_t chi = ~clo;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int q = r ^ J;
        if (q > r) {
            const uint32_t a = k[r], b = k[q];
            k[r] = fp_umed3(a, b, clo); k[q] = fp_umed3(a, b, chi);
        }
    }
}
__device__ __forceinline__ void fp_wave_sort1024(uint32_t (&k)[16], int lane) {
    fp_stage_static<2, 1>(k);
    fp_stage_static<4, 2>(k); fp_stage_static<4, 1>(k);
    fp_stage_static<8, 4>(k); fp_stage_static<8, 2>(k); fp_stage_static<8, 1>(k);
#pragma unroll 1
    for (int ll = 0; ll <= 6; ll++) {                     // runs of 16 << ll elements
        const bool asc = (lane & (1 << ll)) == 0;         // ll = 6: one ascending run
#pragma unroll 1
        for (int m = (1 << ll) >> 1; m > 0; m >>= 1) {
            const uint32_t c = (((lane & m) == 0) == asc) ? 0u : 0xffffffffu;   // keep the smaller / the larger
#pragma unroll
            for (int r = 0; r < 16; r++) k[r] = fp_umed3(k[r], (uint32_t)__shfl_xor((int)k[r], m), c);
        }
        const uint32_t clo = asc ? 0u : 0xffffffffu;
        fp_stage_lane<8>(k, clo); fp_stage_lane<4>(k, clo); fp_stage_lane<2>(k, clo); fp_stage_lane<1>(k, clo);
    }
}

// (the result of an asm statement counts as a per-lane value: without this a population count of the mask is done in the
// vector unit, two instructions per word and one more to add)
__device__ __forceinline__ unsigned long long fp_uniform64(unsigned long long m) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}
// the lanes with a <= b as a mask, straight from the compare (a ballot of a condition that also guards a branch is
// otherwise rebuilt from a 0 / 1 register: two more vector instructions per key in the median count)
__device__ __forceinline__ unsigned long long fp_mask_le(unsigned a, unsigned b) {
    unsigned long long m;
    asm("v_cmp_le_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "s"(b));
    return fp_uniform64(m);
}

// a for the lanes outside the mask, b for the lanes in it
__device__ __forceinline__ unsigned fp_select(unsigned long long mask, unsigned a, unsigned b) {
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
typedef float fp_v2f __attribute__((ext_vector_type(2)));
// the value of lane ^ 1 / lane ^ 2 (DPP quad permutes: one vector instruction, no LDS)
__device__ __forceinline__ unsigned fp_quad_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ unsigned fp_quad_xor2(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); }   // quad_perm [2,3,0,1]
typedef __attribute__((address_space(3))) unsigned fp_lds_u32;
// a - b and the lanes where it borrows (a < b)
__device__ __forceinline__ unsigned fp_sub_borrow(unsigned a, unsigned b, unsigned long long& borrow) {
    unsigned r;
    unsigned long long m;
    asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(r), "=s"(m) : "v"(a), "s"(b));
    borrow = fp_uniform64(m);
    return r;
}

// ---- the bit stream of a row is written by threads that own runs of 8 pixels: a thread assembles its codes in a 64-bit
// window and stores whole words; only the first and the last word of its run can be shared with a neighbour (atomic OR
// into the zeroed buffer).  One LDS atomic per thread and end instead of two to four per pixel.
struct fp_bitw {
    unsigned* words;
    unsigned w, nb;                  // current word, bits of it filled
    unsigned long long acc;          // bits of words w, w + 1, left-aligned
    bool shared;                     // word w started before this run: a neighbour writes into it as well
    __device__ __forceinline__ void init(unsigned* wd, unsigned bitpos) { words = wd; w = bitpos >> 5; nb = bitpos & 31u; acc = 0ull; shared = nb != 0u; }
    __device__ __forceinline__ void flush() {
        const unsigned hi = (unsigned)(acc >> 32);
        if (shared) { if (hi) atomicOr(&words[w], hi); } else words[w] = hi;
        shared = false; acc <<= 32; nb -= 32u; w++;
    }
    __device__ __forceinline__ void put(unsigned val, int n) {          // 1 <= n <= 32, val < 2^n
        acc |= (unsigned long long)val << (64u - nb - (unsigned)n);
        nb += (unsigned)n;
        if (nb >= 32u) flush();
    }
    __device__ __forceinline__ void zeros(unsigned z) { nb += z; while (nb >= 32u) flush(); }
    __device__ __forceinline__ void finish() { const unsigned hi = (unsigned)(acc >> 32); if (nb && hi) atomicOr(&words[w], hi); }
};

// FLOAT_IN: src = float32 rows, quantised first; else src = integer rows of BYTEPIX bytes.
// dynamic LDS: int vals[nxpad] | unsigned words[maxwords] | unsigned blkbits[nblk+1] | uint8 fsv[nblk] (+ float mode scratch)
// MODE 0: the bit stream buffer holds the worst case (every pixel at full width): ~100 KB of LDS for a 10560-pixel float row,
//         one workgroup per CU.
// MODE 1: a buffer of [capwords] words (the host gives half the worst case: 16 bits per pixel of a float row; real frames
//         take 7-8) -- ~66 KB, TWO workgroups per CU, whose barrier-separated phases then overlap; a row whose stream does
//         not fit is marked FP_FLAG_RETRY.
// MODE 2: only the rows marked FP_FLAG_RETRY, with the worst-case buffer (same bytes as MODE 0 would have made).
//
// The three exact medians of CFITSIO's noise estimate (lower medians of the |2nd / 3rd / 5th order differences| of the row):
//   round 3 selected them by radix passes over LDS histograms -- an LDS atomic per key and pass, ~64 cycles per wave
//   instruction whatever the addresses: 1.0 of the kernel's 1.4 ms.  Now (rows of >= FP_MIN_ND differences):
//   (1) 1024 keys per kind are sampled at a fixed stride and sorted by one wave per kind in registers; the sample's order
//       statistics 512 -/+ FP_SAMP_HALF bracket the median's value [lo, hi] (+- 3 sigma of the sample median's rank);
//   (2) ONE pass over the differences counts the keys below lo and compacts the keys inside [lo, hi] (ends included; ~9 %
//       of them) into per-wave segments -- masks from one compare of the wrapped difference key - lo, the count below lo
//       from that subtraction's borrow, population counts in the scalar unit, two keys per lane in packed float32
//       arithmetic, a store without a branch (round 5; 25 vector instructions per key, round 4: 77);
//   (3) the rank falls inside: one histogram pass over the ~1000 collected keys on (key - lo) >> s (512 bins), the bin's
//       handful of keys are collected and ranked by counting (hi == lo: every key inside is that value).
//   Whenever a step does not hold (rank outside the bracket: ~0.3 % of the rows; a segment or the last list overflows:
//   rows with many equal differences) the row takes the histogram passes over all keys, as before: the result is the
//   exact order statistic either way.
#define FP_FLAG_RETRY 3u
#define FP_NCOLL 128               // keys collected in the last select pass before it falls back to a histogram
#define FP_NSAMP 1024
#define FP_SAMP_HALF 48
#define FP_MIN_ND 4096             // shorter rows: histogram passes (the bracket of a 1024-key sample is no gain there)
#define FP_MIN_SEG 64              // smallest per-wave segment the bracket path is used with
#define FP_BR_BITS 9               // the bracket's histogram: 512 bins (two 16-bit counters per word) over the ~5 % of a row's keys inside
#define FP_BR_WORDS 256
#define FP_HINT_FRAC 0.06f          // bracket around a hinted median: -/+ 6 %
#define FP_FAST_NMAX 14u            // longest code word of the pair / quad writer: 4 of them + the fs field stay below 64 bits
#define FP_HINT_VALID 0x80000000u  // (keys are bit patterns of non-negative floats: the top bit is free)
// Hints: neighbouring rows of an image have nearly the same noise.  Every workgroup leaves its three exact medians in
// hint[row][3] (relaxed agent-scope atomics, the top bit marks a written word), and starts from the medians of the row one
// or two dispatch generations earlier (row - gen, row - 2 gen; gen = workgroups resident at once) -/+ FP_HINT_FRAC when
// they are there already: ~5 % of the keys inside, no sample, no sort.  Where that bracket misses (or no hint exists yet:
// the first rows) the sampled bracket takes over, then the histograms.  A hint only chooses the path: the medians are the
// exact order statistics whichever way, so the bytes do not depend on the timing of other workgroups.
template <int BYTEPIX, bool FLOAT_IN, int MODE>
__global__ __launch_bounds__(FP_THREADS, MODE == 1 ? FP_MINW : 4) void k_fp_tile(const void* __restrict__ src, int ny, int nx, size_t row_stride_elems,
                                                 float qlevel, int dither_seed, const float* __restrict__ rnd,
                                                 uint8_t* __restrict__ scratch, size_t tile_stride, fp_tile* __restrict__ tiles,
                                                 int capwords, int hist_only, unsigned* __restrict__ hint, int gen, float in_scale) {
    typedef rice_par<BYTEPIX> RP;
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (a scalar: loops over a wave's share stay uniform)
    // MODE 0 / 1: one workgroup per row.  MODE 2: a small grid; workgroup b looks through the flags of rows b, b + G, b + 2 G, ...
    // (64 at a time, a thread each) and takes the marked ones in turn.  Until round 4 this launch had a workgroup per row of
    // the image, each asking for the worst-case LDS only to find its row unmarked: 0.6 ms of a stream's time per image when
    // the CUs are busy.  (No list of marked rows in the context: calls on different streams of one context run at once.)
    __shared__ int s_rows[64], s_nrows;
    int row = blockIdx.x, scan0 = (int)blockIdx.x - 64 * (int)gridDim.x, nmark = 0, kmark = 0;
#define FP_ROW_RETURN do { if (MODE == 2) goto row_done; else return; } while (0)
  next_row:
    if (MODE == 2) {
        while (kmark >= nmark) {                                       // (workgroup-uniform) the next 64 rows of this workgroup
            scan0 += 64 * (int)gridDim.x;
            if (scan0 >= ny) return;
            __syncthreads();
            if (tid == 0) s_nrows = 0;
            __syncthreads();
            const int r = scan0 + tid * (int)gridDim.x;
            if (tid < 64 && r < ny && tiles[r].flag == FP_FLAG_RETRY) s_rows[atomicAdd(&s_nrows, 1)] = r;
            __syncthreads();
            nmark = s_nrows; kmark = 0;
        }
        row = s_rows[kmark++];
        __syncthreads();
    }
    {
    const int nblk = (nx + 31) / 32;
    const int maxwords = MODE == 1 ? capwords : (8 * BYTEPIX + nblk * RP::fsbits + nx * RP::bbits + 31) / 32 + 2;
    int* vals = reinterpret_cast<int*>(lds);
    unsigned* words = reinterpret_cast<unsigned*>(vals + ((nx + 3) & ~3));
    unsigned* blkbits = words + maxwords;
    uint8_t* fsv = reinterpret_cast<uint8_t*>(blkbits + nblk + 1);
    __shared__ __align__(16) unsigned hist[3][1024];                              // 2048 bins each: two 16-bit counters per word; first the samples
    __shared__ unsigned coll[3][FP_NCOLL], ncoll[3];
    __shared__ int s_many, s_fail, s_hint_ok;
    __shared__ unsigned sel_prefix[3], sel_rank[3];
    __shared__ unsigned br_lo[3], br_hi[3], br_need[3], br_shift[3];
    __shared__ unsigned segn[3][FP_THREADS / 64], cpart[FP_THREADS / 64][3];
    __shared__ float red_min[FP_THREADS / 64], red_max[FP_THREADS / 64];
    __shared__ double s_delta, s_zero;
    __shared__ int s_flag;
    __shared__ unsigned wsum[FP_THREADS / 64];
    fp_tile* out = &tiles[row];
    const bool vec4 = FLOAT_IN && (nx & 3) == 0 && (row_stride_elems & 3) == 0 && (((uintptr_t)src) & 15) == 0;
    constexpr int NLD = (FP_MAXNX / 4 + FP_THREADS - 1) / FP_THREADS;

    if (FLOAT_IN) {
        const float* f = (const float*)src + (size_t)row * row_stride_elems;
        float* fv = reinterpret_cast<float*>(vals);
        float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
        int bad = 0;
        if (vec4) {                                                    // 16-byte accesses
            float4* fv4 = reinterpret_cast<float4*>(fv);
            const float4* f4 = reinterpret_cast<const float4*>(f);
            const int n4 = nx >> 2;
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int i = tid + k * FP_THREADS;
                if (i < n4) {
                    float4 v = f4[i];
                    v.x *= in_scale; v.y *= in_scale; v.z *= in_scale; v.w *= in_scale;      // (1.0f: the pixels as they are, bit for bit)
                    fv4[i] = v;
                    if (!isfinite(v.x) || !isfinite(v.y) || !isfinite(v.z) || !isfinite(v.w)) bad = 1;
                    mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w))); mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
                }
            }
        } else {
            for (int i = tid; i < nx; i += FP_THREADS) {
                const float v = f[i] * in_scale;
                fv[i] = v;
                if (!isfinite(v)) bad = 1;
                mn = fminf(mn, v); mx = fmaxf(mx, v);
            }
        }
        if (tid == 0) { s_flag = 0; s_many = 0; s_fail = 0; }
        if (tid == 64) {
            // a hint: the medians of the row one (else two) dispatch generations earlier, if they have been written
            int ok = 0;
            if (hint && !hist_only) {
#pragma unroll 1
                for (int g = 1; g <= 2 && !ok; g++) {
                    const int hr = row - g * gen;
                    if (hr < 0) break;
                    const unsigned h0 = __hip_atomic_load(&hint[3 * hr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned h1 = __hip_atomic_load(&hint[3 * hr + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned h2 = __hip_atomic_load(&hint[3 * hr + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (h0 & h1 & h2 & FP_HINT_VALID) {
                        const unsigned hh[3] = {h0, h1, h2};
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            const float m = __uint_as_float(hh[k] & ~FP_HINT_VALID);
                            br_lo[k] = __float_as_uint(m * (1.f - FP_HINT_FRAC)); br_hi[k] = __float_as_uint(m * (1.f + FP_HINT_FRAC));
                        }
                        ok = 1;
                    }
                }
            }
            s_hint_ok = ok;
        }
        __syncthreads();
        if (bad) s_flag = 2;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
        if (lane == 0) { red_min[wave] = mn; red_max[wave] = mx; }
        const int nd = nx - 8;
        const unsigned rank0 = nd > 0 ? (unsigned)((nd - 1) / 2) : 0u;
        if (tid < 3) { sel_prefix[tid] = 0; sel_rank[tid] = rank0; ncoll[tid] = 0; }
#define FP_KEYS(i)                                                                                                      \
                    const float v1 = fv[i], v3 = fv[i + 2], v5 = fv[i + 4], v7 = fv[i + 6], v9 = fv[i + 8];               \
                    const unsigned k2 = __float_as_uint(fabsf(v5 - v7));                                                    \
                    const unsigned k3 = __float_as_uint(fabsf((2.f * v5) - v3 - v7));                                        \
                    const unsigned k5 = __float_as_uint(fabsf((6.f * v5) - (4.f * v3) - (4.f * v7) + v1 + v9));
        // ---- bracket path ------------------------------------------------------------------------------------------
        const int segcap = maxwords / (3 * (FP_THREADS / 64));            // the segments live in the (still unused) stream buffer
        const bool bracket = !hist_only && nd >= FP_MIN_ND && segcap >= FP_MIN_SEG;    // workgroup-uniform
        bool done = false;                                                 // workgroup-uniform: the medians are in sel_prefix[]
        // (FPV_*: knock-out switches of tools/exp builds; the Makefile never defines them)
#ifdef FPV_NOMED
        if (tid < 3) sel_prefix[tid] = __float_as_uint(tid == 0 ? 8.6f : (tid == 1 ? 14.9f : 50.8f));
        done = true;
#else
        if (bracket) {
            unsigned* samp = &hist[0][0];
            unsigned* seg = words;                                         // [3][16 waves][segcap]
#ifdef FPV_NOHINT
            const int first_attempt = 1;
#else
            const int first_attempt = s_hint_ok ? 0 : 1;                   // (workgroup-uniform)
#endif
#pragma unroll 1
            for (int attempt = first_attempt; attempt < 2 && !done; attempt++) {
            if (attempt == 1) {                                            // (attempt 0: br_lo / br_hi from the hint)
                for (int sidx = tid; sidx < FP_NSAMP; sidx += FP_THREADS) {
                    const int i = (int)(((long long)sidx * nd) >> 10);     // FP_NSAMP = 1024 samples at a fixed stride
                    FP_KEYS(i)
                    samp[sidx] = k2; samp[1024 + sidx] = k3; samp[2048 + sidx] = k5;
                }
                __syncthreads();
                if (wave < 3) {
                    uint32_t k[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) k[r] = samp[1024 * wave + 64 * r + lane];  // any assignment of keys to slots will do
                    fp_wave_sort1024(k, lane);
                    // sorted[16 l + r]: the order statistics 512 - H and 512 + H
                    constexpr int A = FP_NSAMP / 2 - FP_SAMP_HALF, B = FP_NSAMP / 2 + FP_SAMP_HALF;
                    if (lane == A / 16) br_lo[wave] = k[A % 16];
                    if (lane == B / 16) br_hi[wave] = k[B % 16];
                }
            }
            __syncthreads();
            const unsigned lo0 = __builtin_amdgcn_readfirstlane(br_lo[0]), lo1 = __builtin_amdgcn_readfirstlane(br_lo[1]), lo2 = __builtin_amdgcn_readfirstlane(br_lo[2]);
            const unsigned sp0 = __builtin_amdgcn_readfirstlane(br_hi[0]) - lo0, sp1 = __builtin_amdgcn_readfirstlane(br_hi[1]) - lo1,
                           sp2 = __builtin_amdgcn_readfirstlane(br_hi[2]) - lo2;      // (hi >= lo: order statistics / a hint's -/+ 6 %)
            // Per kind a wave counts its keys >= lo (ballot + scalar population count: no vector work besides the compare) and
            // compacts the keys inside [lo, hi], ends included, into its segment.  (Round 4 counted < lo, == lo, == hi per lane
            // -- nine counters and nine wave reductions per thread.)
            unsigned lt0 = 0, lt1 = 0, lt2 = 0;                            // wave-uniform: keys < lo
            unsigned base0 = 0, base1 = 0, base2 = 0;                      // wave-uniform fill of this wave's segments
            const unsigned off0 = (unsigned)((0 * (FP_THREADS / 64) + wave) * segcap), off1 = (unsigned)((1 * (FP_THREADS / 64) + wave) * segcap),
                           off2 = (unsigned)((2 * (FP_THREADS / 64) + wave) * segcap);
            unsigned* seg0 = seg + off0;
            unsigned* seg1 = seg + off1;
            unsigned* seg2 = seg + off2;
            // (a) Whole blocks of 128 keys, two neighbouring keys per lane: the differences in packed float32 arithmetic (the
            // products by 2 and 4 are exact, so fma(-4, v3, 6 v5) rounds once like CFITSIO's (6 v5) - (4 v3)), and every lane
            // stores -- the lanes outside the bracket into 64 spare words behind the stream buffer (blkbits, unused until the
            // Rice pass) -- so that the store needs no second compare and no branch.
            const unsigned seg_lds = (unsigned)(uintptr_t)(fp_lds_u32*)seg;                // the segments' address inside the LDS
            const unsigned trash = seg_lds + 4u * ((unsigned)maxwords + (unsigned)lane);
#ifndef FPV_NOB3
            const int nblk128 = nd >> 7;
#else
            const int nblk128 = 8;
#endif
#define FP_BRACKET2(K, LO, SPAN, LT, BASE, OFF)                                                                         \
                {                                                                                                       \
                    unsigned long long below;                              /* the lanes with K < lo: the subtraction's borrow */ \
                    const unsigned rel = fp_sub_borrow(K, LO, below);                                                   \
                    LT += (unsigned)__popcll(below);                                                                    \
                    const unsigned long long bal = fp_mask_le(rel, SPAN);  /* (rel wraps below lo) */                   \
                    const unsigned cnt = (unsigned)__popcll(bal);                                                       \
                    if (BASE + cnt <= (unsigned)segcap) {                  /* (wave-uniform; past it the kind fails anyway) */ \
                        const unsigned at = __builtin_amdgcn_readfirstlane(seg_lds + 4u * (OFF + BASE));   /* (kept scalar) */ \
                        const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u)); \
                        *reinterpret_cast<fp_lds_u32*>((uintptr_t)fp_select(bal, trash, (pos << 2) + at)) = K;          \
                    }                                                                                                   \
                    BASE += cnt;                                                                                        \
                }
            for (int blk = wave; blk < nblk128; blk += FP_THREADS / 64) {
                const fp_v2f* pv = reinterpret_cast<const fp_v2f*>(fv + 128 * blk + 2 * lane);
                const fp_v2f v1 = pv[0], v3 = pv[1], v5 = pv[2], v7 = pv[3], v9 = pv[4];
                const fp_v2f d2 = v5 - v7;
                const fp_v2f d3 = __builtin_elementwise_fma(fp_v2f{2.f, 2.f}, v5, -v3) - v7;
                const fp_v2f d5 = __builtin_elementwise_fma(fp_v2f{-4.f, -4.f}, v7, __builtin_elementwise_fma(fp_v2f{-4.f, -4.f}, v3, v5 * 6.f)) + v1 + v9;
                const unsigned k2a = __float_as_uint(d2.x) & 0x7fffffffu, k2b = __float_as_uint(d2.y) & 0x7fffffffu;
                const unsigned k3a = __float_as_uint(d3.x) & 0x7fffffffu, k3b = __float_as_uint(d3.y) & 0x7fffffffu;
                const unsigned k5a = __float_as_uint(d5.x) & 0x7fffffffu, k5b = __float_as_uint(d5.y) & 0x7fffffffu;
                FP_BRACKET2(k2a, lo0, sp0, lt0, base0, off0)
                FP_BRACKET2(k2b, lo0, sp0, lt0, base0, off0)
                FP_BRACKET2(k3a, lo1, sp1, lt1, base1, off1)
                FP_BRACKET2(k3b, lo1, sp1, lt1, base1, off1)
                FP_BRACKET2(k5a, lo2, sp2, lt2, base2, off2)
                FP_BRACKET2(k5b, lo2, sp2, lt2, base2, off2)
            }
#undef FP_BRACKET2
            // (b) the keys behind the last whole block, one per lane, by the wave whose turn that block would have been
#ifndef FPV_NOB3
            for (int i0 = 128 * nblk128; i0 < nd && wave == nblk128 % (FP_THREADS / 64); i0 += 64) {
#else
            for (int i0 = 128 * nblk128; i0 < 0; i0 += 64) {
#endif
                const int i = i0 + lane;
#define FP_BRACKET(K, LO, SPAN, GE, BASE, SEG)                                                                          \
                {                                                                                                       \
                    GE += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(in && K < LO));                                \
                    const bool inb = in && K - LO <= SPAN;                                                              \
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(inb);                                    \
                    const unsigned pos = BASE + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u)); \
                    if (inb && pos < (unsigned)segcap) SEG[pos] = K;                                                    \
                    BASE += (unsigned)__popcll(bal);                                                                    \
                }
                const bool in = i < nd;
                const int ii = in ? i : 0;
                FP_KEYS(ii)
                FP_BRACKET(k2, lo0, sp0, lt0, base0, seg0)
                FP_BRACKET(k3, lo1, sp1, lt1, base1, seg1)
                FP_BRACKET(k5, lo2, sp2, lt2, base2, seg2)
#undef FP_BRACKET
            }
            if (lane == 0) {
                cpart[wave][0] = lt0; cpart[wave][1] = lt1; cpart[wave][2] = lt2;                      // keys < lo
                segn[0][wave] = base0; segn[1][wave] = base1; segn[2][wave] = base2;
            }
            for (int i = tid; i < 3 * FP_BR_WORDS; i += FP_THREADS) hist[i / FP_BR_WORDS][i % FP_BR_WORDS] = 0;      // (the samples are done with)
            __syncthreads();
            if (tid < 3) {
                unsigned lt = 0, nin = 0;
                bool over = false;                                         // a segment of this kind did not hold its keys
                for (int w = 0; w < FP_THREADS / 64; w++) { lt += cpart[w][tid]; nin += segn[tid][w]; over |= segn[tid][w] > (unsigned)segcap; }
                const unsigned r = sel_rank[tid], lo = br_lo[tid], hi = br_hi[tid];
                unsigned need = 0;
#ifdef FPV_NOST2
                if (true) sel_prefix[tid] = __float_as_uint(tid == 0 ? 8.6f : (tid == 1 ? 14.9f : 50.8f));
                else
#endif
                if (r < lt || r >= lt + nin) s_fail = 1;                   // the median lies outside the bracket
                else if (hi == lo) sel_prefix[tid] = lo;                   // (every key inside is the same value)
                else if (over) s_fail = 1;
                else {
                    need = 1;
                    sel_rank[tid] = r - lt;
                    const unsigned span = hi - lo;                         // keys inside: lo .. hi -> rel = key - lo in [0, span]
                    const int bits = 32 - __clz((int)span);
                    br_shift[tid] = bits > FP_BR_BITS ? (unsigned)(bits - FP_BR_BITS) : 0u;
                }
#ifdef FPV_NOB3
                s_fail = 0; need = 0; sel_prefix[tid] = __float_as_uint(tid == 0 ? 8.6f : (tid == 1 ? 14.9f : 50.8f));
#endif
                br_need[tid] = need;
            }
            __syncthreads();
            if (!s_fail) {                                                 // workgroup-uniform
                // histogram of the collected keys on (key - lo) >> shift: every wave its own segments
#pragma unroll 1
                for (int kd = 0; kd < 3; kd++) {
                    if (!br_need[kd]) continue;
                    const unsigned n = segn[kd][wave], lo = br_lo[kd], sh = br_shift[kd];
                    const unsigned* sg = seg + (kd * (FP_THREADS / 64) + wave) * segcap;
                    for (unsigned j = lane; j < n; j += 64) { const unsigned d = (sg[j] - lo) >> sh; atomicAdd(&hist[kd][d >> 1], 1u << ((d & 1u) * 16)); }
                }
                __syncthreads();
                if (tid < 192 && br_need[tid >> 6]) {                      // one wave per histogram, 8 bins (4 words) per lane
                    const int kd = tid >> 6;
                    const uint4 h4 = *reinterpret_cast<const uint4*>(&hist[kd][4 * lane]);
                    const unsigned cb[8] = {h4.x & 0xffffu, h4.x >> 16, h4.y & 0xffffu, h4.y >> 16, h4.z & 0xffffu, h4.z >> 16, h4.w & 0xffffu, h4.w >> 16};
                    unsigned mine = 0;
#pragma unroll
                    for (int w = 0; w < 8; w++) mine += cb[w];
                    unsigned incl = mine;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                    const unsigned excl = incl - mine, r = sel_rank[kd];
                    if (r >= excl && r < incl) {                           // the one lane whose bins hold the rank
                        unsigned rr = r - excl, b = 0, found = 0;
#pragma unroll
                        for (int w = 0; w < 8; w++) {
                            const bool here = !found && rr < cb[w];
                            if (here) { b = 8u * lane + w; found = 1; }
                            if (!found) rr -= cb[w];
                        }
                        sel_rank[kd] = rr;
                        sel_prefix[kd] = b;                                // the bin for now
                    }
                }
                __syncthreads();
#pragma unroll 1
                for (int kd = 0; kd < 3; kd++) {
                    if (!br_need[kd]) continue;
                    const unsigned n = segn[kd][wave], lo = br_lo[kd], sh = br_shift[kd], bin = sel_prefix[kd];
                    const unsigned* sg = seg + (kd * (FP_THREADS / 64) + wave) * segcap;
                    for (unsigned j = lane; j < n; j += 64) {
                        const unsigned key = sg[j];
                        if (((key - lo) >> sh) == bin) { const unsigned q = atomicAdd(&ncoll[kd], 1u); if (q < FP_NCOLL) coll[kd][q] = key; }
                    }
                }
                __syncthreads();
                if (tid < 3 && br_need[tid]) {                             // the rank-th smallest of the bin's keys
                    const int n = (int)ncoll[tid];
                    if (n > FP_NCOLL) s_fail = 1;
                    else {
                        const unsigned r = sel_rank[tid];
                        unsigned best = 0;
#pragma unroll 1
                        for (int a = 0; a < n; a++) {
                            const unsigned ka = coll[tid][a];
                            unsigned below = 0, equal = 0;
#pragma unroll 1
                            for (int cc = 0; cc < n; cc++) { below += coll[tid][cc] < ka; equal += coll[tid][cc] == ka; }
                            if (r >= below && r < below + equal) { best = ka; break; }
                        }
                        sel_prefix[tid] = best;
                    }
                }
                __syncthreads();
            }
            done = !s_fail;
            if (!done) {                                                   // next: the sampled bracket, then the histogram passes; from scratch
                __syncthreads();
                if (tid < 3) { sel_prefix[tid] = 0; sel_rank[tid] = rank0; ncoll[tid] = 0; }
                if (tid == 0) { s_many = 0; s_fail = 0; }
#ifdef FPV_STAT
                if (tid == 0 && attempt == 1) s_flag = 16;                 // (experiment: count these rows as refused)
#endif
                __syncthreads();
            }
            }
        }
#endif
        __syncthreads();
        // lower medians of the 2nd / 3rd / 5th order differences, exact: radix select in two histogram passes of 11 bits
        // (2048 bins as pairs of 16-bit counters in one word: a row has fewer than 65536 pixels) and a third pass that just
        // collects the handful of keys that share the 22 leading bits (a histogram pass over the last 10 bits only if they are
        // many: rows with long runs of equal differences).
        if (nd > 0 && !done) {
#pragma unroll 1
            for (int pass = 0; pass < 3; pass++) {
                const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
                const unsigned dmask = pass == 2 ? 1023u : 2047u;
                const bool collect = pass == 2 && !s_many;            // workgroup-uniform
                if (!collect) {
                    for (int i = tid; i < 3 * 1024; i += FP_THREADS) (&hist[0][0])[i] = 0;
                    __syncthreads();
                }
                const unsigned p0 = sel_prefix[0], p1 = sel_prefix[1], p2 = sel_prefix[2];
                const int up = pass == 0 ? 31 : (pass == 1 ? 21 : 10);   // keys take part when they agree above this bit position
                if (collect) {
                    for (int i = tid; i < nd; i += FP_THREADS) {
                        FP_KEYS(i)
                        if ((k2 >> 10) == (p0 >> 10)) { const unsigned j = atomicAdd(&ncoll[0], 1u); if (j < FP_NCOLL) coll[0][j] = k2; }
                        if ((k3 >> 10) == (p1 >> 10)) { const unsigned j = atomicAdd(&ncoll[1], 1u); if (j < FP_NCOLL) coll[1][j] = k3; }
                        if ((k5 >> 10) == (p2 >> 10)) { const unsigned j = atomicAdd(&ncoll[2], 1u); if (j < FP_NCOLL) coll[2][j] = k5; }
                    }
                } else if (pass == 0) {
                    for (int i = tid; i < nd; i += FP_THREADS) {
                        FP_KEYS(i)
                        { const unsigned d = k2 >> 21; atomicAdd(&hist[0][d >> 1], 1u << ((d & 1u) * 16)); }
                        { const unsigned d = k3 >> 21; atomicAdd(&hist[1][d >> 1], 1u << ((d & 1u) * 16)); }
                        { const unsigned d = k5 >> 21; atomicAdd(&hist[2][d >> 1], 1u << ((d & 1u) * 16)); }
                    }
                } else {
                    for (int i = tid; i < nd; i += FP_THREADS) {
                        FP_KEYS(i)
                        if ((k2 >> up) == (p0 >> up)) { const unsigned d = (k2 >> shift) & dmask; atomicAdd(&hist[0][d >> 1], 1u << ((d & 1u) * 16)); }
                        if ((k3 >> up) == (p1 >> up)) { const unsigned d = (k3 >> shift) & dmask; atomicAdd(&hist[1][d >> 1], 1u << ((d & 1u) * 16)); }
                        if ((k5 >> up) == (p2 >> up)) { const unsigned d = (k5 >> shift) & dmask; atomicAdd(&hist[2][d >> 1], 1u << ((d & 1u) * 16)); }
                    }
                }
                __syncthreads();
                if (collect) {
                    if (ncoll[0] > FP_NCOLL || ncoll[1] > FP_NCOLL || ncoll[2] > FP_NCOLL) {
                        // too many keys with these 22 leading bits: the histogram pass after all (workgroup-uniform)
                        __syncthreads();
                        if (tid == 0) s_many = 1;
                        __syncthreads();
                        pass = 1;                                     // (the loop's increment makes it pass 2 again)
                        continue;
                    }
                    if (tid < 3) {                                    // a thread per key type: the rank-th smallest of <= FP_NCOLL keys
                        const int n = (int)ncoll[tid];
                        const unsigned r = sel_rank[tid];
                        unsigned best = 0;
#pragma unroll 1
                        for (int a = 0; a < n; a++) {
                            const unsigned ka = coll[tid][a];
                            unsigned below = 0, equal = 0;
#pragma unroll 1
                            for (int c = 0; c < n; c++) { below += coll[tid][c] < ka; equal += coll[tid][c] == ka; }
                            if (r >= below && r < below + equal) { best = ka; break; }
                        }
                        sel_prefix[tid] = best;
                    }
                } else if (tid < 192) {                               // one wave per histogram, 32 bins (16 words) per lane
                    const int k = tid >> 6;
                    unsigned mine = 0;
#pragma unroll 4
                    for (int w = 0; w < 16; w++) { const unsigned c = hist[k][16 * lane + w]; mine += (c & 0xffffu) + (c >> 16); }
                    unsigned incl = mine;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                    const unsigned excl = incl - mine, r = sel_rank[k];
                    if (r >= excl && r < incl) {                   // the one lane whose bins hold the rank
                        unsigned rr = r - excl, b = 32 * lane;
#pragma unroll 1
                        for (int w = 0; w < 32; w++) {
                            const unsigned c = hist[k][(32 * lane + w) >> 1];
                            const unsigned cnt = (w & 1) ? (c >> 16) : (c & 0xffffu);
                            if (rr < cnt) { b = 32 * lane + w; break; }
                            rr -= cnt;
                        }
                        sel_rank[k] = rr;
                        sel_prefix[k] |= b << shift;
                    }
                }
                __syncthreads();
            }
        }
#undef FP_KEYS
        if (tid == 0) {
            float minv = red_min[0], maxv = red_max[0];
            for (int w = 1; w < FP_THREADS / 64; w++) { minv = fminf(minv, red_min[w]); maxv = fmaxf(maxv, red_max[w]); }
            double n2 = 0., n3 = 0., n5 = 0.;
            if (nd > 0) {
                n2 = 1.0483579 * (double)__uint_as_float(sel_prefix[0]);
                n3 = 0.6052697 * (double)__uint_as_float(sel_prefix[1]);
                n5 = 0.1772048 * (double)__uint_as_float(sel_prefix[2]);
                if (hint) {
#pragma unroll
                    for (int k = 0; k < 3; k++) __hip_atomic_store(&hint[3 * row + k], sel_prefix[k] | FP_HINT_VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            double stdev = n3;
            if (n2 != 0. && n2 < stdev) stdev = n2;
            if (n5 != 0. && n5 < stdev) stdev = n5;
            const double delta = (qlevel == 0.f) ? stdev / 4. : stdev / (double)qlevel;
            double zeropt = 0.;
            int flag = s_flag;
            const double minval = (double)minv, maxval = (double)maxv;
            if (!flag) {
                if (delta == 0. || (maxval - minval) / delta > 2. * 2147483647. - FP_NRESERVED) flag = 1;
                else if ((maxval - minval) / delta < 2147483647. - FP_NRESERVED) {
                    zeropt = minval;
                    const long long iq = (long long)(zeropt / delta + 0.5);
                    zeropt = (double)iq * delta;
                } else zeropt = (minval + maxval) / 2.;
            }
            s_delta = delta; s_zero = zeropt; s_flag = flag;
            out->zscale = delta; out->zzero = zeropt; out->flag = (uint32_t)flag;
        }
        __syncthreads();
        if (s_flag) { if (tid == 0) out->nbytes = 0; FP_ROW_RETURN; }
        const double delta = s_delta, zeropt = s_zero;
        int iseed = (row + dither_seed - 1) % FP_NRANDOM;          // (tile number 1.. + ZDITHER0 - 1 - 1) % N_RANDOM
        if (iseed < 0) iseed += FP_NRANDOM;
        // the float row is replaced in place by the quantised integers (same LDS words; each
        // element is read and rewritten by one thread, and the medians are done with it)
#ifdef FPV_NOQUANT
        for (int i = tid; i < nx; i += FP_THREADS) vals[i] = (int)(fv[i] * 2.f);
#else
        {
            // the dither index of pixel i (fp_rand_index): the table is read from start0 + i until its end, then from start1 on
            // (rows of up to 2 x 9500 pixels take two segments: both starts once per row instead of a table load, a float64
            // product and a loop per pixel)
            const int is1 = (iseed + 1 == FP_NRANDOM) ? 0 : iseed + 1;
            const int start0 = (int)((double)rnd[iseed] * 500.), start1 = (int)((double)rnd[is1] * 500.);
            const int n0 = FP_NRANDOM - start0, n1 = FP_NRANDOM - start1;
            // CFITSIO: NINT((x - zero) / delta + r - 0.5), NINT(y) = (int)(y + 0.5) for y >= 0, (int)(y - 0.5) below.  Here
            // u = (x - zero) (1 / delta) + r in one fused step stands for y + 0.5 and the integer is floor(u): the same
            // integer unless u lies within the rounding error of the shortcut (a few ulp of a quotient below 2^32: < 2^-19)
            // of an integer -- there the two roundings may fall on different sides, and an exact tie rounds away from zero
            // in NINT --, so the pixels with fract(u) within 2^-17 of 0 or 1, one in 10^5, take CFITSIO's expression.
            const double rdelta = 1.0 / delta;
            auto quant = [&](int i, int ri) {
                const double x = (double)fv[i] - zeropt, r = (double)rnd[ri];
                const double u = fma(x, rdelta, r);
                const double fr = __builtin_amdgcn_fract(u);
                int qv = (int)floor(u);
                if (!(fabs(fr - 0.5) < 0.5 - 0x1p-17)) qv = fp_nint((x / delta) + r - 0.5);
                vals[i] = qv;
            };
            int k = 0;
            for (; (k + 1) * FP_THREADS <= min(n0, nx); k++) quant(tid + k * FP_THREADS, start0 + tid + k * FP_THREADS);   // (uniform bounds: no index arithmetic)
            for (int i = tid + k * FP_THREADS; i < nx; i += FP_THREADS) {
                const int j = i - n0;
                quant(i, j < 0 ? start0 + i : (j < n1 ? start1 + j : fp_rand_index(rnd, iseed, i)));
            }
        }
#endif
    } else {
        if (BYTEPIX == 1) {
            const uint8_t* p = (const uint8_t*)src + (size_t)row * row_stride_elems;
            for (int i = tid; i < nx; i += FP_THREADS) vals[i] = (int)(signed char)p[i];
        } else if (BYTEPIX == 2) {
            const short* p = (const short*)src + (size_t)row * row_stride_elems;
            for (int i = tid; i < nx; i += FP_THREADS) vals[i] = (int)p[i];
        } else {
            const int* p = (const int*)src + (size_t)row * row_stride_elems;
            for (int i = tid; i < nx; i += FP_THREADS) vals[i] = p[i];
        }
        if (tid == 0) { out->zscale = 1.0; out->zzero = 0.0; out->flag = 0; }
    }
    {   // (words starts 16-byte aligned: vals holds a multiple of 4 ints)
        uint4* w4 = reinterpret_cast<uint4*>(words);
        for (int i = tid; i < (maxwords >> 2); i += FP_THREADS) w4[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < (maxwords & 3)) words[(maxwords & ~3) + tid] = 0;
    }
    __syncthreads();

    // ---- pass 1: a thread takes 8 consecutive pixels, the 4 threads of a quad one 32-pixel block (the last block of a row
    // may have fewer runs: pixels beyond the row count as absent, the quad works as a whole): zig-zag differences (kept in
    // registers for pass 2), the block's split level fs, the bit lengths
    constexpr int NIT = (FP_MAXNX / 8 + FP_THREADS - 1) / FP_THREADS;     // runs of 8 pixels per thread
    const int nq = (nx + 7) >> 3, sub = tid & 3;
    unsigned dd[NIT][8], ioff[NIT];                                  // differences; bit offset of the run inside its block's codes
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int q = tid + it * FP_THREADS, b = q >> 2;
        if (it * FP_THREADS + 64 * wave >= 4 * nblk) {               // (wave-uniform) no block of the row in this wave's runs
#pragma unroll
            for (int j = 0; j < 8; j++) dd[it][j] = 0;
            ioff[it] = 0;
            continue;
        }
        int prev = (8 * q - 1 < nx) ? vals[q ? 8 * q - 1 : 0] : 0;
        const bool full = 8 * q + 8 <= nx;                           // all 8 pixels inside the row: every run but the last one or two
        if (full) {
            // (two 16-byte LDS reads, no per-pixel bounds)
            const int4 va = *reinterpret_cast<const int4*>(&vals[8 * q]), vb = *reinterpret_cast<const int4*>(&vals[8 * q + 4]);
            const int vv[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int pd = vv[j] - prev;
                if (BYTEPIX == 1) pd = (int)(signed char)pd;
                if (BYTEPIX == 2) pd = (int)(short)pd;
                unsigned d = ((unsigned)pd << 1) ^ (unsigned)(pd >> 31);          // (pd < 0 ? ~(pd << 1) : pd << 1)
                if (BYTEPIX == 1) d &= 0xffu;
                if (BYTEPIX == 2) d &= 0xffffu;
                dd[it][j] = d;
                prev = vv[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = 8 * q + j;
                const bool in = i < nx;
                const int v = in ? vals[i] : prev;
                int pd = v - prev;
                if (BYTEPIX == 1) pd = (int)(signed char)pd;
                if (BYTEPIX == 2) pd = (int)(short)pd;
                unsigned d = ((unsigned)pd << 1) ^ (unsigned)(pd >> 31);
                if (BYTEPIX == 1) d &= 0xffu;
                if (BYTEPIX == 2) d &= 0xffffu;
                dd[it][j] = in ? d : 0u;
                prev = v;
            }
        }
        // the block's sum of differences over the quad's four runs: in 32 bits where no difference has a bit above 2^26 (32
        // of them stay below 2^31) -- quad exchanges as DPP moves, no LDS --, else (wave-uniform) in 64 bits
        unsigned s32 = 0, anyb = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { s32 += dd[it][j]; anyb |= dd[it][j]; }
        s32 += fp_quad_xor1(s32); s32 += fp_quad_xor2(s32);
        anyb |= fp_quad_xor1(anyb); anyb |= fp_quad_xor2(anyb);
        const int thisblock = min(32, nx - 32 * b);
        // CFITSIO: dpsum = (pixelsum - thisblock / 2 - 1) / thisblock in double, < 0 -> 0, psum = (unsigned)dpsum >> 1, fs = bits
        // of psum.  For a full block the division by 32 is a shift of the integer sum (truncation of a non-negative double =
        // floor; sums below 2^36 keep (unsigned)dpsum in range): a few integer instructions instead of a float64 division per run
        unsigned psum;
        bool pszero;
        if (BYTEPIX == 4 && __any(anyb >> 26)) {
            unsigned long long ps = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) ps += dd[it][j];
            ps += __shfl_xor(ps, 1, 64); ps += __shfl_xor(ps, 2, 64);
            pszero = ps == 0ull;
            if (thisblock == 32 && ps < (1ull << 36)) psum = ps >= 17ull ? (unsigned)((ps - 17ull) >> 5) >> 1 : 0u;
            else {
                double dpsum = ((double)ps - (double)(thisblock / 2) - 1.) / (double)thisblock;
                if (dpsum < 0.) dpsum = 0.;
                psum = ((unsigned)dpsum) >> 1;
            }
        } else {
            pszero = s32 == 0u;
            if (thisblock == 32) psum = s32 >= 17u ? (s32 - 17u) >> 6 : 0u;
            else {
                double dpsum = ((double)s32 - (double)(thisblock / 2) - 1.) / (double)thisblock;
                if (dpsum < 0.) dpsum = 0.;
                psum = ((unsigned)dpsum) >> 1;
            }
        }
        const int fs = psum ? 32 - __builtin_clz(psum) : 0;           // (for (fs = 0; psum > 0; fs++) psum >>= 1)
        int code;                                                     // what goes into the fsbits field
        if (fs >= RP::fsmax) code = RP::fsmax + 1;
        else if (fs == 0 && pszero) code = 0;
        else code = fs + 1;
        // bits of this run: per pixel (d >> fs) + 1 + fs, or bbits (code fsmax + 1), or nothing (code 0)
        unsigned top = 0;                                             // (pixels beyond the row hold d = 0)
#pragma unroll
        for (int j = 0; j < 8; j++) top += dd[it][j] >> (fs & 31);
        const unsigned nin = (unsigned)min(8, max(0, nx - 8 * q));
        const unsigned len = code == RP::fsmax + 1 ? nin * (unsigned)RP::bbits : (code == 0 ? 0u : top + nin * (unsigned)(fs + 1));
        unsigned incl = len;
        { const unsigned t = __shfl_up(incl, 1, 64); if (sub >= 1) incl += t; }
        { const unsigned t = __shfl_up(incl, 2, 64); if (sub >= 2) incl += t; }
        ioff[it] = incl - len;
        if (b < nblk && sub == 3) { blkbits[b] = incl + RP::fsbits; fsv[b] = (uint8_t)code; }
    }
    __syncthreads();
    // ---- exclusive scan of the block lengths (nblk <= 512 <= FP_THREADS); the stream starts
    // with the first pixel
    {
        const unsigned mine = (tid < nblk) ? blkbits[tid] : 0u;
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        unsigned base = 8 * BYTEPIX;
        for (int w = 0; w < (tid >> 6); w++) base += wsum[w];
        const unsigned excl = base + incl - mine;
        __syncthreads();
        if (tid < nblk) blkbits[tid] = excl;
        if (tid == FP_THREADS - 1) blkbits[nblk] = excl + mine;   // total bits (threads beyond nblk hold zeros)
    }
    __syncthreads();
    if (MODE == 1 && blkbits[nblk] + 64u > 32u * (unsigned)maxwords) {     // the stream does not fit the short buffer
        if (tid == 0) { out->flag = FP_FLAG_RETRY; out->nbytes = 0; }
        FP_ROW_RETURN;
    }
    // ---- pass 2: write the codes
#ifndef FPV_NOPASS2
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int q = tid + it * FP_THREADS, b = q >> 2;
        if (it * FP_THREADS + (tid & ~63) >= nq) continue;           // (wave-uniform) the whole wave lies beyond the row
#ifndef FPV_NOFASTW
        if (!hist_only) {
            // The common case, decided per wave: every run of the wave has its 8 pixels inside the row, a split level (no
            // all-zero block, no block of raw pixels) and code words of at most FP_FAST_NMAX bits.  Then two code words make
            // a 32-bit pair, two pairs a 64-bit quad -- fixed shifts by the lengths, no window, no flush test per pixel --,
            // the block's fs field rides in front of its first quad, and the run's two quads go into the stream as five
            // words at most: whole words stored, the two end words OR-ed (a neighbour's run may share them).  Same bits as
            // the window below makes; rows of stars and the row's ends take that one.
            const bool act = q < nq;
            const int code = act ? (int)fsv[b] : 1;
            if (!__all(!act || (8 * q + 8 <= nx && code >= 1 && code <= RP::fsmax))) goto window;     // (wave-uniform)
            const int fs = code - 1;
            const unsigned lowmask = (1u << (fs & 31)) - 1u, bit = 1u << (fs & 31);
            unsigned pr[4], mp[4], nmax = 0;                           // pairs of code words and their lengths
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned d0 = dd[it][2 * k], d1 = dd[it][2 * k + 1];
                const unsigned n0 = (d0 >> (fs & 31)) + (unsigned)(fs + 1), n1 = (d1 >> (fs & 31)) + (unsigned)(fs + 1);
                nmax = max(nmax, max(n0, n1));
                pr[k] = (((d0 & lowmask) | bit) << (n1 & 31u)) | ((d1 & lowmask) | bit);
                mp[k] = n0 + n1;
            }
            if (__all(!act || nmax <= FP_FAST_NMAX)) {                                          // (all 64 lanes are here: the loop is wave-uniform)
                if (act) {
                    const unsigned p01 = pr[0], p23 = pr[1], p45 = pr[2], p67 = pr[3], m01 = mp[0], m23 = mp[1], m45 = mp[2], m67 = mp[3];
                    unsigned long long q0 = ((unsigned long long)p01 << m23) | p23, q1 = ((unsigned long long)p45 << m67) | p67;
                    unsigned m0 = m01 + m23;                          // 4 .. 4 FP_FAST_NMAX bits
                    const unsigned m1 = m45 + m67;
                    if (sub == 0) { q0 |= (unsigned long long)(unsigned)code << m0; m0 += RP::fsbits; }
                    if (q == 0) {                                     // the row's first pixel as it is, in front of block 0
                        unsigned first = (unsigned)vals[0];
                        if (BYTEPIX == 1) first &= 0xffu;
                        if (BYTEPIX == 2) first &= 0xffffu;
                        atomicOr(&words[0], first << (32 - 8 * BYTEPIX));
                    }
                    const unsigned P = blkbits[b] + (sub ? RP::fsbits + ioff[it] : 0u);
                    const unsigned long long A = q0 << (64u - m0), B = q1 << (64u - m1);      // left-aligned
                    const unsigned long long Chi = A | (B >> m0), Clo = B << (64u - m0);     // 128 bits, the run's first bit on top
                    const unsigned c0 = (unsigned)(Chi >> 32), c1 = (unsigned)Chi, c2 = (unsigned)(Clo >> 32), c3 = (unsigned)Clo;
                    const unsigned o = P & 31u, end = o + m0 + m1;
                    unsigned* wp = words + (P >> 5);
                    const unsigned T0 = c0 >> o, T1 = __builtin_amdgcn_alignbit(c0, c1, o), T2 = __builtin_amdgcn_alignbit(c1, c2, o),
                                   T3 = __builtin_amdgcn_alignbit(c2, c3, o), T4 = __builtin_amdgcn_alignbit(c3, 0u, o);
                    if (o == 0u && end >= 32u) wp[0] = T0; else atomicOr(&wp[0], T0);
                    if (end >= 64u) wp[1] = T1; else if (end > 32u) atomicOr(&wp[1], T1);
                    if (end >= 96u) wp[2] = T2; else if (end > 64u) atomicOr(&wp[2], T2);
                    if (end >= 128u) wp[3] = T3; else if (end > 96u) atomicOr(&wp[3], T3);
                    if (end > 128u) atomicOr(&wp[4], T4);
                }
                continue;
            }
        }
      window:
#endif
        if (q >= nq) continue;
        const int code = fsv[b];
        fp_bitw bw;
        if (q == 0) {                                                 // the row's first pixel as it is, then block 0
            bw.init(words, 0u);
            unsigned first = (unsigned)vals[0];
            if (BYTEPIX == 1) first &= 0xffu;
            if (BYTEPIX == 2) first &= 0xffffu;
            bw.put(first, 8 * BYTEPIX);
        } else bw.init(words, blkbits[b] + (sub ? RP::fsbits + ioff[it] : 0u));
        if (sub == 0) bw.put((unsigned)code, RP::fsbits);
        if (code != 0) {
            const int npx = min(8, nx - 8 * q);                       // (8 for every run but the row's last one or two)
            const int fs = code - 1;
            const unsigned lowmask = (1u << (fs & 31)) - 1u;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (j < npx) {
                    const unsigned d = dd[it][j];
                    if (code == RP::fsmax + 1) bw.put(d, RP::bbits);
                    else {
                        // `top` zeros, a one, the fs low bits = the number (1 << fs) | low written in top + 1 + fs bits: one
                        // step of the window where that fits 32 bits (a pixel thousands of sigma off its neighbour does not)
                        const unsigned top = d >> fs, n = top + 1u + (unsigned)fs;
                        if (n <= 32u) bw.put((1u << fs) | (d & lowmask), (int)n);
                        else {
                            bw.zeros(top);
                            bw.put(1u, 1);
                            if (fs) bw.put(d & lowmask, fs);
                        }
                    }
                }
            }
        }
        bw.finish();
    }
#endif
    __syncthreads();
    const unsigned totbits = blkbits[nblk];
    const unsigned nbytes = (totbits + 7) >> 3;
    unsigned* dst = reinterpret_cast<unsigned*>(scratch + (size_t)row * tile_stride);
    for (unsigned w = tid; w < (nbytes + 3) / 4; w += FP_THREADS) dst[w] = __builtin_bswap32(words[w]);
    if (tid == 0) out->nbytes = nbytes;
    }
  row_done:
    if (MODE == 2) goto next_row;
#undef FP_ROW_RETURN
}

// tile streams -> contiguous heap
__global__ __launch_bounds__(256) void k_fp_gather(const uint8_t* __restrict__ scratch, size_t tile_stride,
                                                   const fp_tile* __restrict__ tiles, const long long* __restrict__ offsets,
                                                   uint8_t* __restrict__ heap) {
    const int row = blockIdx.x;
    const unsigned n = tiles[row].nbytes;
    const uint8_t* s = scratch + (size_t)row * tile_stride;
    uint8_t* d = heap + offsets[row];
    for (unsigned i = threadIdx.x; i < n; i += 256) d[i] = s[i];
}

static size_t fp_tile_stride(int nx, int bytepix) {
    const size_t nblk = (size_t)(nx + 31) / 32;
    const size_t bits = 8 * (size_t)bytepix + nblk * 5 + (size_t)nx * 8 * bytepix;
    return ((bits + 31) / 32 + 2) * 4 + 64 & ~(size_t)63;
}

extern "C" size_t bbx_fpack_tile_stride(int nx, int bytepix) { return fp_tile_stride(nx, bytepix); }

// d_rnd: the 10000-value random table (float32) on the device.  d_scratch: ny * tile_stride bytes.
// The row-hint table of a call lives with the STREAM the call is made on: the output stage compresses on the lane's stream
// and -- its overflow fallback -- on a writer thread's own stream of the same context at the same time.  (Round 4 took the
// table from the context's workspace slots: bbx_ws frees and reallocates, two threads inside it raced on the slot table, and
// two calls in flight shared one table.)  A table grows only when its own stream needs more; nothing another stream uses is
// ever freed before the context goes.
#include <mutex>
static std::mutex g_fphint_mutex;
static void* fp_hint_table(bbx_ctx* ctx, hipStream_t s, size_t bytes, int* rc) {
    *rc = BBX_OK;
    std::lock_guard<std::mutex> lock(g_fphint_mutex);
    int k = -1;
    for (int i = 0; i < 16; i++) {
        if (ctx->fphint[i].ptr && ctx->fphint[i].stream == (void*)s) { k = i; break; }
        if (k < 0 && !ctx->fphint[i].ptr) k = i;                     // first free entry
    }
    if (k < 0) k = 0;                                                // more than 16 streams: share the first table (hints only choose a path)
    if (ctx->fphint[k].ptr && ctx->fphint[k].stream == (void*)s && ctx->fphint[k].bytes >= bytes) return ctx->fphint[k].ptr;
    if (ctx->fphint[k].ptr && ctx->fphint[k].stream != (void*)s) return ctx->fphint[k].bytes >= bytes ? ctx->fphint[k].ptr : (*rc = BBX_ERR_NOMEM, nullptr);
    if (ctx->fphint[k].ptr) { (void)hipStreamSynchronize(s); (void)hipFree(ctx->fphint[k].ptr); ctx->fphint[k].ptr = nullptr; }
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes + bytes / 8 + 256);
    if (e != hipSuccess) { *rc = bbx_hip_fail(ctx, e, "hipMalloc(fpack hints)", __LINE__); return nullptr; }
    ctx->fphint[k].ptr = p; ctx->fphint[k].stream = (void*)s; ctx->fphint[k].bytes = bytes + bytes / 8 + 256;
    return p;
}
void bbx_fpack_release(bbx_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_fphint_mutex);
    for (int i = 0; i < 16; i++) if (ctx->fphint[i].ptr) { (void)hipFree(ctx->fphint[i].ptr); ctx->fphint[i].ptr = nullptr; }
}

static int fpack_tiles_scaled(bbx_ctx* ctx, int ny, int nx, const void* d_img, int bitpix, float qlevel, int dither_seed,
                              const float* d_rnd, uint8_t* d_scratch, void* d_tiles, float in_scale, void* stream) {
    if (!ctx || !d_img || !d_scratch || !d_tiles || ny < 1 || nx < 1 || nx > FP_MAXNX) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int bytepix = bitpix == -32 ? 4 : bitpix / 8;
    if (!(bitpix == -32 || bitpix == 8 || bitpix == 16 || bitpix == 32)) return BBX_ERR_ARG;
    if (bitpix == -32 && (!d_rnd || dither_seed < 1 || dither_seed > 10000)) return BBX_ERR_ARG;
    const size_t stride = fp_tile_stride(nx, bytepix);
    const int nblk = (nx + 31) / 32;
    const int fsbits = bytepix == 1 ? 3 : (bytepix == 2 ? 4 : 5);
    const size_t maxwords = (8 * (size_t)bytepix + (size_t)nblk * fsbits + (size_t)nx * 8 * bytepix + 31) / 32 + 2;
    const size_t fixed = (size_t)((nx + 3) & ~3) * 4 + ((size_t)nblk + 1) * 4 + (size_t)nblk + 16;
    const size_t ldsbytes = fixed + maxwords * 4;
    if (ldsbytes > 150 * 1024) return BBX_ERR_ARG;
    // first try with half the worst-case stream buffer (two workgroups per CU), then the rows that did not fit
    const size_t capwords = maxwords / 2 + 16, ldshalf = fixed + capwords * 4;
    const bool two = ldshalf + 15 * 1024 <= 80 * 1024 && !ctx->fpack_one_wg;       // (+ the kernel's static LDS: histograms 12 KB, lists, sums)
    fp_tile* tiles = (fp_tile*)d_tiles;
    const int hist_only = ctx->fpack_hist_only;
    // hints (float rows): one slot of three words per row in the context's workspace, zeroed per call; a dispatch generation =
    // the workgroups resident at once (two per CU with the short buffer)
    unsigned* d_hint = nullptr;
    const int ncu = ctx->num_cus > 0 ? ctx->num_cus : 256;
    const int gen = two ? 2 * ncu : ncu;
    if (bitpix == -32 && !hist_only) {
        int rc;
        d_hint = (unsigned*)fp_hint_table(ctx, s, (size_t)ny * 3 * sizeof(unsigned), &rc); if (rc) return rc;
        BBX_HIP(hipMemsetAsync(d_hint, 0, (size_t)ny * 3 * sizeof(unsigned), s));
    }
#ifndef FPV_SKIP_RETRY
#define FPV_SKIP_RETRY 0            // (experiment: what the launch of the retry kernel costs when no row needs it)
#endif
#define FP_LAUNCH(BP, FL)                                                                                              \
    do {                                                                                                               \
        if (two) {                                                                                                     \
            BBX_HIP(hipFuncSetAttribute((const void*)k_fp_tile<BP, FL, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldshalf)); \
            BBX_HIP(hipFuncSetAttribute((const void*)k_fp_tile<BP, FL, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsbytes)); \
            hipLaunchKernelGGL((k_fp_tile<BP, FL, 1>), dim3(ny), dim3(FP_THREADS), ldshalf, s, d_img, ny, nx, (size_t)nx, qlevel, \
                               dither_seed, d_rnd, d_scratch, stride, tiles, (int)capwords, hist_only, d_hint, gen, in_scale);   \
            if (!FPV_SKIP_RETRY) hipLaunchKernelGGL((k_fp_tile<BP, FL, 2>), dim3(min(ny, 256)), dim3(FP_THREADS), ldsbytes, s, d_img, ny, nx, (size_t)nx, qlevel, \
                               dither_seed, d_rnd, d_scratch, stride, tiles, 0, hist_only, d_hint, gen, in_scale);     \
        } else {                                                                                                       \
            BBX_HIP(hipFuncSetAttribute((const void*)k_fp_tile<BP, FL, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsbytes)); \
            hipLaunchKernelGGL((k_fp_tile<BP, FL, 0>), dim3(ny), dim3(FP_THREADS), ldsbytes, s, d_img, ny, nx, (size_t)nx, qlevel, \
                               dither_seed, d_rnd, d_scratch, stride, tiles, 0, hist_only, d_hint, gen, in_scale);     \
        }                                                                                                              \
    } while (0)
    if (bitpix == -32) FP_LAUNCH(4, true);
    else if (bitpix == 8) FP_LAUNCH(1, false);
    else if (bitpix == 16) FP_LAUNCH(2, false);
    else FP_LAUNCH(4, false);
#undef FP_LAUNCH
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

extern "C" int bbx_fpack_tiles(bbx_ctx* ctx, int ny, int nx, const void* d_img, int bitpix, float qlevel, int dither_seed,
                               const float* d_rnd, uint8_t* d_scratch, void* d_tiles, void* stream) {
    return fpack_tiles_scaled(ctx, ny, nx, d_img, bitpix, qlevel, dither_seed, d_rnd, d_scratch, d_tiles, 1.0f, stream);
}

extern "C" int bbx_fpack_gather(bbx_ctx* ctx, int ny, int nx, int bitpix, const uint8_t* d_scratch, const void* d_tiles,
                                const long long* d_offsets, uint8_t* d_heap, void* stream) {
    if (!ctx || !d_scratch || !d_tiles || !d_offsets || !d_heap || ny < 1) return BBX_ERR_ARG;
    const int bytepix = bitpix == -32 ? 4 : bitpix / 8;
    hipLaunchKernelGGL(k_fp_gather, dim3(ny), dim3(256), 0, (hipStream_t)stream, d_scratch, fp_tile_stride(nx, bytepix),
                       (const fp_tile*)d_tiles, d_offsets, d_heap);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

// ---------------------------------------------------------------------------------
// The body of the COMPRESSED_IMAGE binary table as it goes into the file, made on the device in one
// enqueue (no host round trip between the steps): [ny descriptor rows, big-endian][heap of tile streams].
//   descriptor row (FITS 4.0 section 10): int32 length, int32 heap offset of COMPRESSED_DATA
//   [, int32 length, int32 offset of GZIP_COMPRESSED_DATA (0, 0 here), float64 ZSCALE, float64 ZZERO]
// Rows the quantiser refuses (flag != 0: zero noise / range, NaN) get length 0; their indices are listed in
// info[] and the host stores them losslessly (gzip column) behind the heap, as CFITSIO does.
// info (int64): [0] heap bytes, [1] rows listed, [2] 1 = the heap does not fit cap_heap (nothing gathered),
//               [3] longest tile stream, [4 ...] listed rows (any order)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void put_be32(uint8_t* p, unsigned v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
__device__ __forceinline__ void put_be64(uint8_t* p, unsigned long long v) { put_be32(p, (unsigned)(v >> 32)); put_be32(p + 4, (unsigned)v); }

#define FP_SCAN_THREADS 1024
__global__ __launch_bounds__(FP_SCAN_THREADS) void k_fp_scan_table(const fp_tile* __restrict__ tiles, int ny, int quant, uint8_t* __restrict__ table,
                                                                   long long* __restrict__ offsets, long long* __restrict__ info,
                                                                   long long cap_heap, int max_list) {
    __shared__ unsigned long long wsum[FP_SCAN_THREADS / 64];
    __shared__ unsigned wmax[FP_SCAN_THREADS / 64];
    __shared__ int nlist;
    const int tid = threadIdx.x, lane = tid & 63, per = (ny + FP_SCAN_THREADS - 1) / FP_SCAN_THREADS;
    const int r0 = min(tid * per, ny), r1 = min(r0 + per, ny);
    if (tid == 0) nlist = 0;
    unsigned long long mine = 0; unsigned mx = 0;
    for (int r = r0; r < r1; r++) { const unsigned n = tiles[r].nbytes; mine += n; mx = max(mx, n); }
    unsigned long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
    if (lane == 63) wsum[tid >> 6] = incl;
    if (lane == 0) wmax[tid >> 6] = mx;
    __syncthreads();
    unsigned long long base = 0, total = 0; unsigned allmax = 0;
    for (int w = 0; w < FP_SCAN_THREADS / 64; w++) { if (w < (tid >> 6)) base += wsum[w]; total += wsum[w]; allmax = max(allmax, wmax[w]); }
    unsigned long long off = base + incl - mine;
    const int rowlen = quant ? 32 : 8;
    for (int r = r0; r < r1; r++) {
        const fp_tile t = tiles[r];
        offsets[r] = (long long)off;
        uint8_t* d = table + (size_t)r * rowlen;
        put_be32(d, t.nbytes); put_be32(d + 4, (unsigned)off);
        if (quant) {
            put_be32(d + 8, 0u); put_be32(d + 12, 0u);
            const bool listed = t.flag != 0;
            put_be64(d + 16, listed ? 0ull : (unsigned long long)__double_as_longlong(t.zscale));
            put_be64(d + 24, listed ? 0ull : (unsigned long long)__double_as_longlong(t.zzero));
            if (listed) { const int k = atomicAdd(&nlist, 1); if (k < max_list) info[4 + k] = r; }
        }
        off += t.nbytes;
    }
    __syncthreads();
    if (tid == 0) { info[0] = (long long)total; info[1] = nlist; info[2] = ((long long)total > cap_heap || nlist > max_list) ? 1 : 0; info[3] = allmax; }
}

// tile streams -> contiguous heap; the destination is unaligned: aligned 4-byte stores assembled from two source words
__global__ __launch_bounds__(256) void k_fp_gather4(const uint8_t* __restrict__ scratch, size_t tile_stride, const fp_tile* __restrict__ tiles,
                                                    const long long* __restrict__ offsets, const long long* __restrict__ info,
                                                    uint8_t* __restrict__ heap) {
    if (info[2]) return;
    const int row = blockIdx.x;
    const unsigned n = tiles[row].nbytes;
    if (!n) return;
    const uint8_t* s = scratch + (size_t)row * tile_stride;              // 64-byte aligned
    uint8_t* d = heap + offsets[row];
    const unsigned head = min(n, (unsigned)((4 - ((uintptr_t)d & 3)) & 3));
    if (threadIdx.x < head) d[threadIdx.x] = s[threadIdx.x];
    const unsigned nw = (n - head) / 4, sh = 8 * (head & 3);
    const unsigned* sw = reinterpret_cast<const unsigned*>(s);
    unsigned* dw = reinterpret_cast<unsigned*>(d + head);
    for (unsigned k = threadIdx.x; k < nw; k += 256) {
        // destination word k holds source bytes head + 4k .. head + 4k + 3 (little endian words)
        const unsigned lo = sw[k], hi = sw[k + 1];                        // (the row's slot is padded: k + 1 is readable)
        dw[k] = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
    }
    const unsigned done = head + 4 * nw;
    if (threadIdx.x < n - done) d[done + threadIdx.x] = s[done + threadIdx.x];
}

extern "C" int bbx_fpack_body_scaled(bbx_ctx* ctx, int ny, int nx, const void* d_img, int bitpix, float qlevel, int dither_seed, const float* d_rnd,
                                     uint8_t* d_scratch, void* d_tiles, long long* d_offsets, uint8_t* d_body, long long cap_body, long long* d_info,
                                     int max_list, float scale, void* stream);
extern "C" int bbx_fpack_body(bbx_ctx* ctx, int ny, int nx, const void* d_img, int bitpix, float qlevel, int dither_seed, const float* d_rnd,
                              uint8_t* d_scratch, void* d_tiles, long long* d_offsets, uint8_t* d_body, long long cap_body, long long* d_info,
                              int max_list, void* stream) {
    return bbx_fpack_body_scaled(ctx, ny, nx, d_img, bitpix, qlevel, dither_seed, d_rnd, d_scratch, d_tiles, d_offsets, d_body, cap_body, d_info, max_list,
                                 1.0f, stream);
}
extern "C" int bbx_fpack_body_scaled(bbx_ctx* ctx, int ny, int nx, const void* d_img, int bitpix, float qlevel, int dither_seed, const float* d_rnd,
                                     uint8_t* d_scratch, void* d_tiles, long long* d_offsets, uint8_t* d_body, long long cap_body, long long* d_info,
                                     int max_list, float scale, void* stream) {
    if (!d_offsets || !d_body || !d_info || max_list < 0) return BBX_ERR_ARG;
    if (scale != 1.0f && bitpix != -32) return BBX_ERR_ARG;
    const int rc = fpack_tiles_scaled(ctx, ny, nx, d_img, bitpix, qlevel, dither_seed, d_rnd, d_scratch, d_tiles, scale, stream);
    if (rc) return rc;
    const int quant = bitpix == -32, rowlen = quant ? 32 : 8, bytepix = quant ? 4 : bitpix / 8;
    const long long table = (long long)ny * rowlen;
    if (cap_body < table) return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fp_scan_table, dim3(1), dim3(FP_SCAN_THREADS), 0, s, (const fp_tile*)d_tiles, ny, quant, d_body, d_offsets, d_info,
                       cap_body - table, max_list);
    hipLaunchKernelGGL(k_fp_gather4, dim3(ny), dim3(256), 0, s, d_scratch, fp_tile_stride(nx, bytepix), (const fp_tile*)d_tiles, d_offsets, d_info,
                       d_body + table);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

// ---------------------------------------------------------------------------------
// funpack: Rice decode of row tiles (+ un-quantisation of float images).  The codes of a
// tile are sequential (unary prefixes), so one thread walks one tile; the ~10^4 tiles of a
// frame run side by side.  Decoded values are staged per 32-pixel block in registers and
// written as whole blocks.
// ---------------------------------------------------------------------------------
// desc: per row {int32 len, int32 off} (host order), heap: compressed bytes (padded with 8 readable bytes).
// out_kind: 0 = uint8, 1 = int16 -> uint16 with +32768 (BZERO), 2 = int16, 3 = int32, 4 = float32 (dequantised)
//
// A Rice stream is sequential within its tile (a row), so the time of the kernel is the time of
// one row: what matters is that nothing but ALU work sits on that chain.  One wave handles
// FU_ROWS rows: lanes 0..FU_ROWS-1 each decode one row in bursts of FU_OUT pixels (two Rice
// blocks), reading the stream from a ring in LDS and leaving the pixels in LDS; between bursts
// all 64 lanes stage -- four per row, all rows at once -- refilling the rings with 16-byte loads
// and writing the decoded pixels out.  (Reading the stream straight from global memory puts one
// ~1 us load per byte on the chain: 8 ms for a 10600-row frame.)
#define FU_ROWS 16
#define FU_IN 1024           // ring bytes per row (power of two; a burst needs <= 266)
#define FU_OUT 64            // pixels per burst
// Bit reader over the row's ring, written without branches (the 16 decoding lanes of a wave must
// follow one instruction stream) and in 32-bit operations: the window is three byte-swapped ring
// words w0 w1 w2 and the number of bits of w0 already taken; the next 32 bits of the stream are
// one funnel shift of (w0, w1).  The word read from LDS when the window advances is w2's
// successor, needed two advances later: its latency never shows.
struct ringreader {
    const uint32_t* ring;        // this row's ring (FU_IN / 4 words)
    unsigned idx;                // ring word index of w0
    unsigned w0, w1, w2;
    unsigned pos;                // bits of w0 consumed (< 32 after norm())
    __device__ __forceinline__ unsigned bytes_taken() const { return 4u * idx; }
    __device__ __forceinline__ unsigned word(unsigned i) const { return __builtin_bswap32(ring[i & (FU_IN / 4 - 1)]); }
    __device__ __forceinline__ void start() { idx = 0; pos = 0; w0 = word(0); w1 = word(1); w2 = word(2); }
    __device__ __forceinline__ void norm() {
        const bool c = pos >= 32u;
        w0 = c ? w1 : w0; w1 = c ? w2 : w1;
        idx += c ? 1u : 0u; pos -= c ? 32u : 0u;
        w2 = word(idx + 2u);                                    // (the same word again when nothing moved)
    }
    __device__ __forceinline__ unsigned peek() const {          // the next 32 bits
        const unsigned f = __builtin_amdgcn_alignbit(w0, w1, (32u - pos) & 31u);
        return pos ? f : w0;
    }
    __device__ __forceinline__ unsigned get(int n) {            // n <= 32
        norm();
        const unsigned p = peek();
        const unsigned v = n ? (p >> (32 - n)) : 0u;
        pos += (unsigned)n;
        return v;
    }
    // unary part without a branch: valid when the run of zeros is shorter than 32 (else [viol] is
    // raised and the caller decodes the block again with unary())
    __device__ __forceinline__ unsigned unary_fast(bool& viol) {
        norm();
        const unsigned p = peek();
        viol |= p == 0u;
        const unsigned lz = p ? (unsigned)__clz((int)p) : 31u;
        pos += lz + 1u;
        return lz;
    }
    // a whole code word -- `top` zeros, a one, fs low bits -- from ONE look at the next 32 bits: the value (top << fs) | low.
    // Valid when the word fits them (top + 1 + fs <= 32; else [viol] is raised and the caller decodes the block again with
    // unary() + get()).  One window step per pixel instead of two on the chain that is the kernel's time.
    __device__ __forceinline__ unsigned code_fast(int fs, bool& viol) {
        norm();
        const unsigned p = peek();
        const unsigned lz = p ? (unsigned)__clz((int)p) : 31u;
        const unsigned nb = lz + 1u + (unsigned)fs;
        viol |= (p == 0u) | (nb > 32u);
        // the fs bits behind the one: (p << (lz + 1)) >> (32 - fs), written so that lz = 31 and fs = 0 shift by less than 32
        const unsigned low = (((p << lz) << 1) >> 1) >> (31u - (unsigned)fs);
        pos += nb;
        return (lz << fs) | low;
    }
    __device__ __forceinline__ unsigned unary() {              // number of zeros before the next one
        unsigned z = 0;
        for (;;) {
            norm();
            const unsigned p = peek();
            if (p == 0u) {                                      // 32 zeros: rare (a code longer than 32 bits)
                z += 32u; pos += 32u;
                if (z > 0x100000u) return z;                    // (bounded: a corrupt stream ends)
                continue;
            }
            const unsigned lz = (unsigned)__clz((int)p);
            pos += lz + 1u;
            return z + lz;
        }
    }
};

template <int BYTEPIX, bool FLOATOUT>
__global__ __launch_bounds__(64) void k_funpack(const int* __restrict__ desc, const uint8_t* __restrict__ heap, int ny, int nx,
                                                int out_kind, void* __restrict__ out, const double* __restrict__ zscale,
                                                const double* __restrict__ zzero, int dither_seed,
                                                const float* __restrict__ rnd, int* __restrict__ err) {
    typedef rice_par<BYTEPIX> RP;
    __shared__ __align__(16) uint32_t ring[FU_ROWS][FU_IN / 4 + 4];   // (+4 words: rows on different banks, 16-byte aligned)
    __shared__ int obuf[FU_ROWS][FU_OUT + 1];
    __shared__ unsigned s_fill[FU_ROWS], s_rd[FU_ROWS], s_total[FU_ROWS];   // bytes from the row's aligned base
    __shared__ size_t s_base[FU_ROWS];
    __shared__ int s_cnt[FU_ROWS], s_pix[FU_ROWS], s_live;
    const int lane = threadIdx.x;
    const int row = blockIdx.x * FU_ROWS + lane;
    const bool dec = lane < FU_ROWS;
    bool active = false;
    ringreader br; br.ring = ring[dec ? lane : 0]; br.idx = 0; br.pos = 0; br.w0 = 0; br.w1 = 0; br.w2 = 0;
    int last = 0, pix = 0, mis = 0;
    double zs = 0., zz = 0.;
    int iseed = 0, nextrand = 0;
    if (dec) {
        unsigned total = 0; size_t base = 0;
        if (row < ny) {
            const int len = desc[2 * row], off = desc[2 * row + 1];
            if (len > 0) {                                      // len <= 0: stored another way (gzip column), done on the host
                active = true;
                mis = off & 3;
                base = (size_t)(off - mis);
                total = (unsigned)((mis + len + 3) & ~3);
                if (out_kind == 4) {
                    zs = zscale[row]; zz = zzero[row];
                    iseed = (row + dither_seed - 1) % FP_NRANDOM;
                    nextrand = (int)((double)rnd[iseed] * 500.);
                }
            }
        }
        s_base[lane] = base; s_total[lane] = total; s_fill[lane] = 0; s_rd[lane] = 0; s_cnt[lane] = 0; s_pix[lane] = 0;
    }
    // value stored for a decoded pixel: the integer, or the un-quantised float's bits
    auto emit = [&](int v) -> int {
        if (!FLOATOUT) return v;
        const int r = __float_as_int((float)(((double)v - (double)rnd[nextrand] + 0.5) * zs + zz));
        if (++nextrand == FP_NRANDOM) {
            iseed = (iseed + 1 == FP_NRANDOM) ? 0 : iseed + 1;
            nextrand = (int)((double)rnd[iseed] * 500.);
        }
        return r;
    };
    bool first = true;
    // staging role of a lane: FU_LPR lanes per row, each moving 64 bytes of a 256-byte chunk
    constexpr int FU_LPR = 64 / FU_ROWS;
    static_assert(FU_LPR == 4 && FU_OUT == 64, "staging layout: 4 lanes per row, 4 x 16 bytes per lane and chunk");
    const int srow = lane / FU_LPR, ssub = lane % FU_LPR;
    // prime the rings (up to 768 bytes each) before the first burst
    __syncthreads();
    {
        const unsigned total = s_total[srow];
        const uint8_t* src = heap + s_base[srow];
        unsigned fill = 0;
        for (int c = 0; c < 3 && fill < total; c++, fill += 256) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned bo = fill + 64u * ssub + 16u * k;
                if (bo < total) *(uint4*)&ring[srow][(bo >> 2) & (FU_IN / 4 - 1)] = *(const uint4*)(src + bo);
            }
        }
        if (ssub == 0) s_fill[srow] = fill < total ? fill : total;
    }
    for (;;) {
        __syncthreads();
        // ---- refill, first half: the lanes of a row issue the loads of up to two chunks if its ring
        // has room.  They land in registers and go into the ring after the burst, so their latency runs
        // beside the decode; a ring always holds >= 500 bytes ahead of its reader (or the rest of the
        // stream), a burst needs <= 266.  (All rows at once: no loop over the rows.)
        uint4 pre[2][4];
        unsigned pre_fill;
        {
            const unsigned fill = s_fill[srow], rd = s_rd[srow], total = s_total[srow];
            const uint8_t* src = heap + s_base[srow];
            pre_fill = fill;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const unsigned f = fill + 256u * c;
                const bool go = f < total && f - rd <= FU_IN - 256;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned bo = f + 64u * ssub + 16u * k;
                    pre[c][k] = make_uint4(0, 0, 0, 0);
                    if (go && bo < total) pre[c][k] = *(const uint4*)(src + bo);
                }
                if (go) pre_fill = f + 256u;
            }
        }
        // ---- decode a burst ----
        if (dec) {
            s_cnt[lane] = 0;
            if (active) {
                if (first) {
                    br.start();
                    if (mis) br.get(8 * mis);                   // bytes before the stream in its first aligned word
                    last = (int)br.get(8 * BYTEPIX);
                    if (BYTEPIX == 1) last = (int)(signed char)last;
                    if (BYTEPIX == 2) last = (int)(short)last;
                }
                const int n_burst = min(FU_OUT, nx - pix);
                for (int i = 0; i < n_burst; i += 32) {
                    const int fs = (int)br.get(RP::fsbits) - 1;
                    const int n = min(32, n_burst - i);
                    // the three kinds of block; a lane stays in one of them for 32 pixels
                    if (fs < 0) {
                        for (int j = 0; j < n; j++) obuf[lane][i + j] = emit(last);
                    } else if (fs == RP::fsmax) {
                        for (int j = 0; j < n; j++) {
                            const unsigned d = br.get(RP::bbits);
                            last += (int)(d >> 1) ^ -(int)(d & 1u);
                            if (BYTEPIX == 1) last = (int)(signed char)last;
                            if (BYTEPIX == 2) last = (int)(short)last;
                            obuf[lane][i + j] = emit(last);
                        }
                    } else {
                        // straight-line decode of the block; a code with 32 or more leading zeros (rare)
                        // makes the lane decode the block again with the general reader
                        const ringreader br0 = br;
                        const int last0 = last, iseed0 = iseed, nextrand0 = nextrand;
                        bool viol = false;
#pragma unroll 4
                        for (int j = 0; j < n; j++) {
                            const unsigned d = br.code_fast(fs, viol);
                            last += (int)(d >> 1) ^ -(int)(d & 1u);
                            if (BYTEPIX == 1) last = (int)(signed char)last;
                            if (BYTEPIX == 2) last = (int)(short)last;
                            obuf[lane][i + j] = emit(last);
                        }
                        if (viol) {
                            br = br0; last = last0; iseed = iseed0; nextrand = nextrand0;
                            for (int j = 0; j < n; j++) {
                                const unsigned top = br.unary();
                                const unsigned d = (top << fs) | br.get(fs);
                                last += (int)(d >> 1) ^ -(int)(d & 1u);
                                if (BYTEPIX == 1) last = (int)(signed char)last;
                                if (BYTEPIX == 2) last = (int)(short)last;
                                obuf[lane][i + j] = emit(last);
                            }
                        }
                    }
                }
                s_cnt[lane] = n_burst; s_pix[lane] = pix;
                pix += n_burst;
                s_rd[lane] = br.bytes_taken();
                if (br.bytes_taken() > s_total[lane] + 8) { atomicOr(err, 1); active = false; }    // ran past the tile: corrupt stream
                if (pix >= nx) active = false;
            }
        }
        first = false;
        __syncthreads();
        // ---- refill, second half: the loaded words into the rings ----
        {
            const unsigned fill = s_fill[srow], total = s_total[srow];
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned f = fill + 256u * c, bo = f + 64u * ssub + 16u * k;
                    if (f < pre_fill && bo < total) *(uint4*)&ring[srow][(bo >> 2) & (FU_IN / 4 - 1)] = pre[c][k];
                }
            __syncthreads();
            if (ssub == 0) s_fill[srow] = pre_fill < total ? pre_fill : total;
        }
        // ---- write the bursts out: the lanes of a row take its pixels interleaved ----
        {
            const int n = s_cnt[srow];
            const size_t o0 = (size_t)(blockIdx.x * FU_ROWS + srow) * nx + s_pix[srow];
#pragma unroll
            for (int k = 0; k < FU_OUT / FU_LPR; k++) {
                const int i = k * FU_LPR + ssub;
                if (i < n) {
                    const int v = obuf[srow][i];
                    if (out_kind == 0) ((uint8_t*)out)[o0 + i] = (uint8_t)v;
                    else if (out_kind == 1) ((uint16_t*)out)[o0 + i] = (uint16_t)(v + 32768);
                    else if (out_kind == 2) ((short*)out)[o0 + i] = (short)v;
                    else ((int*)out)[o0 + i] = v;                  // int32 or float bits
                }
            }
        }
        if (lane == 0) s_live = 0;
        __syncthreads();
        if (active) s_live = 1;
        __syncthreads();
        if (!s_live) break;
    }
}

extern "C" int bbx_funpack_tiles(bbx_ctx* ctx, int ny, int nx, int bytepix, const int* d_desc, const uint8_t* d_heap,
                                 int out_kind, void* d_out, const double* d_zscale, const double* d_zzero, int dither_seed,
                                 const float* d_rnd, void* stream) {
    if (!ctx || !d_desc || !d_heap || !d_out || ny < 1 || nx < 1 || out_kind < 0 || out_kind > 4) return BBX_ERR_ARG;
    if (out_kind == 4 && (!d_zscale || !d_zzero || !d_rnd || bytepix != 4 || dither_seed < 1 || dither_seed > 10000)) return BBX_ERR_ARG;
    if ((out_kind == 0 && bytepix != 1) || ((out_kind == 1 || out_kind == 2) && bytepix != 2) || (out_kind == 3 && bytepix != 4))
        return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((ny + FU_ROWS - 1) / FU_ROWS);
    if (bytepix == 1) hipLaunchKernelGGL((k_funpack<1, false>), grid, dim3(64), 0, s, d_desc, d_heap, ny, nx, out_kind, d_out, d_zscale, d_zzero, dither_seed, d_rnd, ctx->d_err);
    else if (bytepix == 2) hipLaunchKernelGGL((k_funpack<2, false>), grid, dim3(64), 0, s, d_desc, d_heap, ny, nx, out_kind, d_out, d_zscale, d_zzero, dither_seed, d_rnd, ctx->d_err);
    else if (bytepix == 4 && out_kind == 4) hipLaunchKernelGGL((k_funpack<4, true>), grid, dim3(64), 0, s, d_desc, d_heap, ny, nx, out_kind, d_out, d_zscale, d_zzero, dither_seed, d_rnd, ctx->d_err);
    else if (bytepix == 4) hipLaunchKernelGGL((k_funpack<4, false>), grid, dim3(64), 0, s, d_desc, d_heap, ny, nx, out_kind, d_out, d_zscale, d_zzero, dither_seed, d_rnd, ctx->d_err);
    else return BBX_ERR_ARG;
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

// big-endian int16 + BZERO 32768 (an uncompressed raw frame as it lies in its FITS file) -> uint16: eight pixels per thread where
// both pointers are 16-byte aligned, else one
__global__ __launch_bounds__(256) void k_raw_be16(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, size_t n, int vec) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    if (vec) {
        const uint4* in4 = reinterpret_cast<const uint4*>(in); uint4* out4 = reinterpret_cast<uint4*>(out);
        for (size_t i = t; i < n / 8; i += nt) {
            uint4 v = in4[i];
            // per 32-bit word: swap the bytes of each half (v_perm), flip the top bit of each
            v.x = __builtin_amdgcn_perm(0u, v.x, 0x02030001u) ^ 0x80008000u; v.y = __builtin_amdgcn_perm(0u, v.y, 0x02030001u) ^ 0x80008000u;
            v.z = __builtin_amdgcn_perm(0u, v.z, 0x02030001u) ^ 0x80008000u; v.w = __builtin_amdgcn_perm(0u, v.w, 0x02030001u) ^ 0x80008000u;
            out4[i] = v;
        }
        for (size_t i = (n / 8) * 8 + t; i < n; i += nt) { const uint16_t x = in[i]; out[i] = (uint16_t)(((x >> 8) | (x << 8)) ^ 0x8000u); }
    } else {
        for (size_t i = t; i < n; i += nt) { const uint16_t x = in[i]; out[i] = (uint16_t)(((x >> 8) | (x << 8)) ^ 0x8000u); }
    }
}
extern "C" int bbx_raw_be16(const void* d_file_pixels, uint16_t* d_out, size_t n, void* stream) {
    if (!d_file_pixels || !d_out || ((uintptr_t)d_file_pixels & 1) || ((uintptr_t)d_out & 1)) return BBX_ERR_ARG;
    if (n == 0) return BBX_OK;
    const int vec = (((uintptr_t)d_file_pixels | (uintptr_t)d_out) & 15) == 0;
    const size_t items = vec ? n / 8 + 1 : n;
    const unsigned blocks = (unsigned)((items + 255) / 256 < 8192 ? (items + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_raw_be16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)d_file_pixels, d_out, n, vec);
    if (hipGetLastError() != hipSuccess) return BBX_ERR_HIP;
    return BBX_OK;
}

// big-endian 32-bit words (float32 / int32 pixels as they lie in a FITS file: master flat, master bias, reference image) ->
// host order; in place allowed (d_out == d_in)
__global__ __launch_bounds__(256) void k_be32(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n, int vec) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    if (vec) {
        const uint4* in4 = reinterpret_cast<const uint4*>(in); uint4* out4 = reinterpret_cast<uint4*>(out);
        for (size_t i = t; i < n / 4; i += nt) {
            uint4 v = in4[i];
            v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y); v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
            out4[i] = v;
        }
        for (size_t i = (n / 4) * 4 + t; i < n; i += nt) out[i] = __builtin_bswap32(in[i]);
    } else {
        for (size_t i = t; i < n; i += nt) out[i] = __builtin_bswap32(in[i]);
    }
}
extern "C" int bbx_be32(const void* d_file_words, void* d_out, size_t n, void* stream) {
    if (!d_file_words || !d_out || ((uintptr_t)d_file_words & 3) || ((uintptr_t)d_out & 3)) return BBX_ERR_ARG;
    if (n == 0) return BBX_OK;
    const int vec = (((uintptr_t)d_file_words | (uintptr_t)d_out) & 15) == 0;
    const size_t items = vec ? n / 4 + 1 : n;
    const unsigned blocks = (unsigned)((items + 255) / 256 < 8192 ? (items + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_be32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)d_file_words, (uint32_t*)d_out, n, vec);
    if (hipGetLastError() != hipSuccess) return BBX_ERR_HIP;
    return BBX_OK;
}

// bbx_build_flags (bbx_ctx.hip): any timing knock-out of this file compiled in?
int bbx_build_flags_fpack(void) {
#if defined(FPV_NOB3) || defined(FPV_NOHINT) || defined(FPV_NOMED) || defined(FPV_NOPASS2) || defined(FPV_NOQUANT) || defined(FPV_NOST2) || defined(FPV_SKIP_RETRY) || defined(FPV_STAT)
    return 1;
#else
    return 0;
#endif
}
