// kernels.hip.h -- gfx950 device code of the knn_search hot path.
//
// Reference semantics (paths relative to /root/reference):
//   coarse distances + top-w   src/coarsequantizers.jl:33-37
//   residuals                  src/coarsequantizers.jl:40-45
//   ADC table build            src/index.jl:232-236
//   list scan                  src/index.jl:240-246
//   bounded top-K              src/index.jl:225-226,247-254,257
//
// Float order is the oracle's canonical order: sequential ascending-index sums, one
// rounding per operation, no FMA contraction (the TU is built with -ffp-contract=off).
// Every selection is a k-smallest on 64-bit keys (f32 bits << 32 | visit order): squared
// distances are non-negative, so the bit pattern orders like the value and ties fall to the
// earlier visit, which is exactly the SortedMultiDict behaviour of index.jl:247-254.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "wave_sort.hip.h"

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define KEY_MAX 0xFFFFFFFFFFFFFFFFull

// Register sets of the list-major code stream (striped_scan_range): 1 = one step ahead (the default), n > 1 = n sets in rotation.
// Round 5 measured the deeper forms on the SIFT1B shape (profiles/r05_sift1b_prefetch_depth.txt): 16 384 x w = 1 scan 1.674 ms at
// depth 1, 1.671 at 2, 1.711 at 3, 2.22 at 4 (registers: one workgroup per CU fewer); w = 8: 7.47 against 7.73 ms at depth 3; with the
// loop unrolled per set instead of rotating registers 2.1-2.3 ms (the step body is large: instruction fetch).  The stream's latency is
// NOT what holds these launches at 0.36-0.38 of the HBM peak -- more kilobytes in flight change nothing.
#ifndef IVFADC_PF_DEPTH
#define IVFADC_PF_DEPTH 1
#endif
// (Also measured in round 5 and dropped: adding an entry's four 16-bit fields of the m = 8 integer filter with ONE v_lshl_add_u64 instead
// of two v_add_u32 -- 3.05 against 3.82 SIMD cycles in isolation, tools/micro/add64_rate.hip, but 8.62 against 7.46 ms in the kernel: the
// inline-asm operand pins even-aligned register pairs and the compiler's schedule of the sixteen lookups falls apart.)

namespace ivf {

static __device__ __forceinline__ int lane_id()
{
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Orders LDS traffic between the lanes of ONE wave (no s_barrier needed: a wave's DS
// operations execute in issue order; the fences only pin the compiler).
static __device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

static __device__ __forceinline__ u64 readfirstlane64(u64 v)
{
    u32 lo = __builtin_amdgcn_readfirstlane((u32)v);
    u32 hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}

static __device__ __forceinline__ u64 make_key(float dist, u32 seq)
{
    return ((u64)__float_as_uint(dist) << 32) | (u64)seq;
}

// ---------------------------------------------------------------------------------------
// Wave-level streaming k-smallest selectors.  State is wave-uniform except where noted.
//   WSel<true>   K <= 64: lane i holds the i-th smallest key in registers; a candidate is
//                inserted with one 64-bit compare + one DPP wave-shift (no LDS, no sort).
//   WSel<false>  any K: LDS buffer of cap = pow2 >= K + 64 keys, bitonic-sorted when full.
// Both keep `thr`: keys >= thr can no longer enter the result.
// ---------------------------------------------------------------------------------------
struct Sel {
    int cnt;
    int sorted;   // buf[0 .. sorted) is ascending (the state the last compaction left; newer entries are appended behind)
    u64 thr;
};

static __device__ void wave_bitonic_sort(u64 *buf, int n)
{
    const int lane = lane_id();
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (n >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = ((i & k) == 0);
                const u64 a = buf[i], b = buf[l];
                if ((a > b) == up) { buf[i] = b; buf[l] = a; }
            }
            wave_sync();
        }
    }
}

// Sort the buffer, keep the K smallest, tighten the threshold.
static __device__ void sel_compact(u64 *buf, Sel &s, int cap, int K)
{
    const int lane = lane_id();
    int n = 64;
    while (n < s.cnt) n <<= 1;
    if (n > cap) n = cap;
    for (int t = s.cnt + lane; t < n; t += 64) buf[t] = KEY_MAX;
    wave_sync();
    wave_bitonic_sort(buf, n);
    if (s.cnt > K) s.cnt = K;
    s.sorted = s.cnt;
    if (s.cnt == K) {
        const u64 t = readfirstlane64(buf[K - 1]);
        s.thr = t < s.thr ? t : s.thr;
    }
}

static __device__ __forceinline__ void sel_push(u64 *buf, Sel &s, int cap, int K, bool pred, u64 key)
{
    const u64 mask = __ballot(pred);
    if (mask == 0) return;
    const int lane = lane_id();
    const int pos = s.cnt + __popcll(mask & ((1ull << lane) - 1ull));
    if (pred) buf[pos] = key;
    s.cnt += __popcll(mask);
    if (s.cnt > cap - 64) { wave_sync(); sel_compact(buf, s, cap, K); }
}

static __device__ __forceinline__ u64 readlane64(u64 v, int l)
{
    const u32 lo = __builtin_amdgcn_readlane((u32)v, l);
    const u32 hi = __builtin_amdgcn_readlane((u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

// lane i <- lane i-1 (lane 0 gets 0): v_mov_b32_dpp wave_shr:1 (bound_ctrl: the lane without a source reads 0, so the
// destination needs no initialisation)
static __device__ __forceinline__ u64 wave_shr1_u64(u64 v)
{
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x138, 0xf, 0xf, true);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x138, 0xf, 0xf, true);
    return ((u64)hi << 32) | lo;
}
// min of two wave-uniform 64-bit values on the scalar unit (there is no 64-bit scalar compare: the compiler would take the
// vector ALU for `a < b ? a : b`, five instructions of every selector insertion)
static __device__ __forceinline__ u64 umin64_uniform(u64 a, u64 b)
{
    const u32 ah = (u32)(a >> 32), al = (u32)a, bh = (u32)(b >> 32), bl = (u32)b;
    const bool lt = ah < bh || (ah == bh && al < bl);
    return lt ? a : b;
}

// wave_sort64(v, lane): ascending sort of one key per lane across the wave -- wave_sort.hip.h (DPP + permlane swaps)

template <bool SMALL> struct WSel;

template <> struct WSel<true> {
    u64 top;    // per lane: lane i = i-th smallest so far (KEY_MAX when empty)
    u64 thr_;   // uniform
    u64 ext_;   // uniform: threshold handed in from outside (another workgroup's K-th key)
    __device__ __forceinline__ void init(u64 thr0, u64 *, int, int)
    {
        top = KEY_MAX;
        thr_ = thr0;
        ext_ = thr0;
    }
    __device__ __forceinline__ u64 thr() const { return thr_; }
    // an upper bound of this selector's r-th smallest key, r in [1, 64] (KEY_MAX while it holds fewer): see publish_bound
    __device__ __forceinline__ u64 kth(int r) const { return readlane64(top, r - 1); }
    // adopt a bound found elsewhere (another wave's K-th key is >= the K-th key of the union)
    __device__ __forceinline__ void tighten(u64 t)
    {
        ext_ = t < ext_ ? t : ext_;
        thr_ = t < thr_ ? t : thr_;
    }
    // Before merging OTHER waves' entries: forget the workgroup-shared bound.  It is exclusive only for a
    // wave's own points (no scanned key can equal another wave's K-th key), whereas a merged entry may BE
    // that K-th key.  `hard` = bound from outside the workgroup (or KEY_MAX).
    __device__ __forceinline__ void unshare(u64 hard, int K, int)
    {
        ext_ = hard;
        const u64 t = readlane64(top, K - 1);
        thr_ = t < hard ? t : hard;
    }
    // An EMPTY selector facing a block of many candidates (the first block of a row / of the first list: no
    // threshold yet, ~K(1 + ln(64/K)) serial insertions): sort the block across the wave instead -- 21 bitonic
    // compare-exchange stages on cross-lane shuffles -- and adopt its K smallest in one shot.
    __device__ __forceinline__ void seed_from_block(bool pred, u64 key, int K, int lane)
    {
        const u64 v = wave_sort64(pred ? key : KEY_MAX, lane);
        top = lane < K ? v : KEY_MAX;
        const u64 t = readlane64(top, K - 1);
        thr_ = t < ext_ ? t : ext_;
    }
    __device__ __forceinline__ void push(bool pred, u64 key, int K, int lane)
    {
        // every loop trip is a real insertion: lanes are re-tested against the tightened threshold.  The bound lives in
        // SGPRs inside the loop (an insertion is ~16 vector instructions; with the bound in VGPRs and the generic
        // ballot / 64-bit min idioms it was 26, a quarter of the m = 8 kernel's vector-ALU time)
        u64 thr = readfirstlane64(thr_);
        const u64 pm = __builtin_amdgcn_ballot_w64(pred);
        u64 mask = __builtin_amdgcn_ballot_w64(key < thr) & pm;
        if (mask == 0) return;
        if (__popcll(mask) >= 16 && readlane64(top, 0) == KEY_MAX) {   // uniform: selector still empty
            seed_from_block(pred && key < thr, key, K, lane);
            return;
        }
        const u32 eh = __builtin_amdgcn_readfirstlane((u32)(ext_ >> 32)), el = __builtin_amdgcn_readfirstlane((u32)ext_);
        while (mask) {
            const int src = __builtin_ctzll(mask);
            const u64 x = readlane64(key, src);
            const bool gt = top > x;
            const u64 up = wave_shr1_u64(top);          // lane 0 reads 0: 0 > x is false, so it takes x like any first lane above x
            const bool take_x = !(up > x);
            top = gt ? (take_x ? x : up) : top;
            // thr = min(K-th key, ext) on 32-bit halves: scalar compares and selects (there is no 64-bit scalar compare, and
            // the compiler turns a u64 min of uniform values into five vector instructions)
            u32 th = __builtin_amdgcn_readlane((u32)(top >> 32), K - 1), tl = __builtin_amdgcn_readlane((u32)top, K - 1);
            asm("" : "+s"(th), "+s"(tl));   // keep the halves apart: recombined, the compare goes back to the vector ALU
            const bool lt = th < eh || (th == eh && tl < el);
            thr = ((u64)(lt ? th : eh) << 32) | (lt ? tl : el);
            mask = __builtin_amdgcn_ballot_w64(key < thr) & pm & ~((2ull << src) - 1ull);
        }
        thr_ = thr;
    }
    // sorted already; returns the number of valid entries
    __device__ __forceinline__ int finish(int K, int lane) const { return __popcll(__ballot(lane < K && top != KEY_MAX)); }
    __device__ __forceinline__ void store(u64 *dst, int cnt, int lane) const
    {
        if (lane < cnt) dst[lane] = top;
    }
    template <class F> __device__ __forceinline__ void for_each(int cnt, int lane, F f) const
    {
        if (lane < cnt) f(lane, top);
    }
};

template <> struct WSel<false> {
    u64 *buf;
    Sel s;
    int cap;
    __device__ __forceinline__ void init(u64 thr0, u64 *ldsbuf, int cap_, int)
    {
        buf = ldsbuf;
        cap = cap_;
        s.cnt = 0;
        s.sorted = 0;
        s.thr = thr0;
    }
    __device__ __forceinline__ u64 thr() const { return s.thr; }
    // (as of the last compaction: the entries appended since can only make the true r-th key smaller)
    __device__ __forceinline__ u64 kth(int r) const { return r <= s.sorted ? readfirstlane64(buf[r - 1]) : KEY_MAX; }
    __device__ __forceinline__ void tighten(u64 t) { s.thr = t < s.thr ? t : s.thr; }
    // call after finish() (buffer sorted, cnt <= K): see WSel<true>::unshare
    __device__ __forceinline__ void unshare(u64 hard, int K, int)
    {
        s.thr = hard;
        if (s.cnt >= K) {
            const u64 t = readfirstlane64(buf[K - 1]);
            s.thr = t < hard ? t : hard;
        }
    }
    __device__ __forceinline__ void push(bool pred, u64 key, int K, int) { sel_push(buf, s, cap, K, pred, key); }
    __device__ __forceinline__ int finish(int K, int)
    {
        wave_sync();
        if (s.cnt > 0) sel_compact(buf, s, cap, K);
        return s.cnt;
    }
    __device__ __forceinline__ void store(u64 *dst, int cnt, int lane) const
    {
        if (dst == buf) return;
        for (int i = lane; i < cnt; i += 64) dst[i] = buf[i];
    }
    template <class F> __device__ __forceinline__ void for_each(int cnt, int lane, F f) const
    {
        for (int i = lane; i < cnt; i += 64) f(i, buf[i]);
    }
};

template <class S> static __device__ __forceinline__ void sel_absorb(S &sel, const u64 *src, int n, int K, int lane);

// ---- the workgroup-shared bound of a slot: sthr[s], and behind the QG bounds the four waves' QUARTER keys [QG][4] ------------------
// The four waves of a workgroup select from disjoint points of the same probe(s) with a selector each, and every wave prunes with
// sthr[s], an upper bound of the K-th smallest key of their union.  Two kinds of bound flow into it (atomicMin):
//   * any wave's own K-th key (round 1): the union's K-th key cannot be larger;
//   * T = max over the four waves of each wave's ceil(K / 4)-th smallest key (round 5; LDS selectors, K > 64): every wave then holds at least ceil(K / 4) keys
//     <= T, the union at least K, so its K-th key is <= T.  Waves see statistically equal shares of a list, so T sits near the K-th
//     key of everything the workgroup has seen -- the bound one workgroup-wide selector would have -- while a wave's own K-th key is
//     the K-th of a quarter of it: about a third of the insertions for the same result (a key above the bound can never enter the
//     final K; what is kept is decided by the merge, as before).  A quarter key only ever decreases, so a stale read gives a larger
//     T: still a bound.  Keys are unique, so the exclusive test (key < bound) loses nothing: no scanned key equals another wave's key,
//     and a wave's own key that IS T sits in its selector already.
// Layout: sthr[0 .. QG) bounds, sthr[QG + 4 s + v] quarter key of wave v for slot s; whoever (re)arms sthr[s] for fresh selectors
// re-arms the slot's quarter keys with it (arm_bound).
constexpr int STHR_WORDS = 5;   // u64 words per slot
template <int QG> static __device__ __forceinline__ void arm_bound(u64 *sthr, int s, u64 t0)
{
    sthr[s] = t0;
#pragma unroll
    for (int v = 0; v < 4; ++v) sthr[QG + 4 * s + v] = KEY_MAX;
}
// after a step that offered candidates to sel (uniform call): publish what this wave knows.  improved = its own bound moved.
template <int QG, class S> static __device__ __forceinline__ void publish_bound(u64 *sthr, int s, const S &sel, int K, bool improved, int lane)
{
    // Register selectors (K <= 64) publish their K-th key only.  Measured with the quarter keys as well (profiles/r05_quarter_key_ab.txt):
    // K = 10 and 64 lose 1-3 % -- an insertion there is a dozen vector instructions, fewer of them do not pay for two more LDS round trips per
    // candidate step -- while the LDS selectors, whose every ~100 accepted candidates cost a 36-stage sort, gain 15 % (SIFT1M shape, K = 100).
    if constexpr (!std::is_same<S, WSel<false>>::value) {
        if (lane == 0 && improved) atomicMin(&sthr[s], sel.thr());
        return;
    }
    const u64 qk = sel.kth((K + 3) >> 2);
    if (lane == 0) {
        if (improved) atomicMin(&sthr[s], sel.thr());
        u64 *qs = sthr + QG + 4 * s;
        const int wv = (int)(threadIdx.x >> 6);
        if (qk < qs[wv]) {
            qs[wv] = qk;
            u64 T = qk;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const u64 o = qs[v];
                T = (v != wv && o > T) ? o : T;
            }
            if (T != KEY_MAX) atomicMin(&sthr[s], T);
        }
    }
}

// Merge the per-wave results of one slot (wave v's sorted entries at xch + v*stride, cnts[v*cstride] of them; every
// wave, the caller `me` included, has stored its entries) into the caller's selector.  `hard` = bound from outside
// the workgroup.  Register selectors with 4K <= 64 put all entries in one 64-lane block and sort it once; otherwise
// the caller drops the shared bound and absorbs the other three waves' entries.
template <class S>
static __device__ __forceinline__ void merge_waves(S &sel, const u64 *xch, size_t stride, const int *cnts, int cstride, int K,
                                                   u64 hard, int me, int lane)
{
    if constexpr (std::is_same<S, WSel<true>>::value) {
        if (4 * K <= 64) {
            const int v = lane / K, i = lane - v * K;
            bool pred = lane < 4 * K;
            u64 key = KEY_MAX;
            if (pred) {
                pred = i < cnts[v * cstride];
                if (pred) key = xch[(size_t)v * stride + i];
            }
            sel.init(hard, nullptr, 64, K);
            sel.seed_from_block(pred && key < hard, key, K, lane);
            return;
        }
    }
    sel.unshare(hard, K, lane);
    for (int ow = 0; ow < 4; ++ow)
        if (ow != me) sel_absorb(sel, xch + (size_t)ow * stride, cnts[ow * cstride], K, lane);
}

// pushes n keys that sit in LDS/global memory at src through a selector
template <class S> static __device__ __forceinline__ void sel_absorb(S &sel, const u64 *src, int n, int K, int lane)
{
    for (int b0 = 0; b0 < n; b0 += 64) {
        const int idx = b0 + lane;
        bool pred = idx < n;
        const u64 key = pred ? src[idx] : KEY_MAX;
        pred = pred && key < sel.thr();
        sel.push(pred, key, K, lane);
    }
}

// ---------------------------------------------------------------------------------------
// HOT-1  coarse distances: out[q][c] = sum_i (C[c][i] - Q[q][i])^2   (coarsequantizers.jl:34)
// 64 x 64 (query x centroid) tile per 256-thread workgroup, 4 x 4 per thread, the d axis
// walked sequentially in LDS-staged steps of 16 so every accumulator sees i = 0..d-1 in order.
// ---------------------------------------------------------------------------------------
#define CO_T 64
#define CO_DK 32
#define CO_LD 68

// TQ = queries per workgroup tile (64: 4x4 per thread; 32: 2x4 per thread, twice the workgroups for small batches;
// 16: 1x4, four times).
// One LDS stage + register prefetch: the global loads of d-chunk k+1 are in flight while chunk k is accumulated.
// ldq = leading dimension of the query rows (d for a dense matrix; the trainer passes sub-space slices)
template <int TQ>
__global__ __launch_bounds__(256) void coarse_dist_kernel(const float *__restrict__ Q, const float *__restrict__ Cn,
                                                          float *__restrict__ out, int nq, int kc, int d, int ldq)
{
    constexpr int RQ = TQ / 16;   // query rows per thread
    constexpr int NL = CO_DK / 16;   // float4 loads per thread per operand per chunk
    __shared__ __attribute__((aligned(16))) float Qs[CO_DK][CO_LD];
    __shared__ __attribute__((aligned(16))) float Cs[CO_DK][CO_LD];
    const int tid = threadIdx.x;
    const int tc = tid & 15, tq = tid >> 4;
    const int q0 = blockIdx.y * TQ, c0 = blockIdx.x * CO_T;
    const int lr = tid >> 2, li = (tid & 3) * 4;   // row 0..63, element offset 0,4,8,12 (+16 per extra load)
    float acc[RQ][4];
#pragma unroll
    for (int a = 0; a < RQ; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;

    const bool qrow_ok = lr < TQ && (q0 + lr) < nq, crow_ok = (c0 + lr) < kc;
    const float *qrow = Q + (size_t)(qrow_ok ? q0 + lr : 0) * ldq;
    const float *crow = Cn + (size_t)(crow_ok ? c0 + lr : 0) * d;
    const bool vec_ok = ((d & 3) == 0) && ((ldq & 3) == 0) && ((((size_t)Q) & 15) == 0);

    // two register stages: the global loads of chunks k+1 and k+2 are in flight while chunk k is accumulated (every
    // workgroup of a small batch is resident from the start, so a launch lasts as long as one workgroup's chain of
    // chunk latencies)
    float qA[NL][4], cA[NL][4], qB[NL][4], cB[NL][4];
    auto fetch = [&](float (&qv)[NL][4], float (&cv)[NL][4], int k0) {
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const int i0 = k0 + n * 16 + li;
            if (vec_ok && i0 + 3 < d) {
                const float4 a = qrow_ok ? *(const float4 *)(qrow + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 b = crow_ok ? *(const float4 *)(crow + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
                qv[n][0] = a.x; qv[n][1] = a.y; qv[n][2] = a.z; qv[n][3] = a.w;
                cv[n][0] = b.x; cv[n][1] = b.y; cv[n][2] = b.z; cv[n][3] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = i0 + e;
                    // zero padding past d: (0-0)^2 = +0 and acc + 0 == acc exactly
                    qv[n][e] = (qrow_ok && i < d) ? qrow[i] : 0.0f;
                    cv[n][e] = (crow_ok && i < d) ? crow[i] : 0.0f;
                }
            }
        }
    };
    auto stage = [&](const float (&qv)[NL][4], const float (&cv)[NL][4]) {
        __syncthreads();   // the previous chunk has been consumed
#pragma unroll
        for (int n = 0; n < NL; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (lr < TQ) Qs[n * 16 + li + e][lr] = qv[n][e];
                Cs[n * 16 + li + e][lr] = cv[n][e];
            }
        __syncthreads();
    };
    // LDS operands are read one k-step ahead of their use (two register sets, plain vector values): left to itself the
    // compiler waits for every ds_read right after issuing it, and two waves per SIMD cannot cover that
    auto ld_q = [&](int i) -> float4 {
        if constexpr (RQ == 4) return *(const float4 *)&Qs[i][tq * 4];
        else if constexpr (RQ == 2) { const float2 t = *(const float2 *)&Qs[i][tq * 2]; return make_float4(t.x, t.y, 0.f, 0.f); }
        else return make_float4(Qs[i][tq], 0.f, 0.f, 0.f);
    };
    auto ld_c = [&](int i) -> float4 { return *(const float4 *)&Cs[i][tc * 4]; };
    auto fma_step = [&](const float4 qq, const float4 cc) {
        const float qa[4] = {qq.x, qq.y, qq.z, qq.w};
        const float ca[4] = {cc.x, cc.y, cc.z, cc.w};
#pragma unroll
        for (int a = 0; a < RQ; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float t = ca[b] - qa[a];
                acc[a][b] = acc[a][b] + t * t;
            }
    };
    auto accumulate = [&]() {
        float4 q0r = ld_q(0), c0r = ld_c(0), q1r, c1r;
#pragma unroll
        for (int i = 0; i < CO_DK; i += 2) {
            q1r = ld_q(i + 1);
            c1r = ld_c(i + 1);
            fma_step(q0r, c0r);
            if (i + 2 < CO_DK) {
                q0r = ld_q(i + 2);
                c0r = ld_c(i + 2);
            }
            fma_step(q1r, c1r);
        }
    };
    fetch(qA, cA, 0);
    if (CO_DK < d) fetch(qB, cB, CO_DK);
    for (int k0 = 0; k0 < d; k0 += 2 * CO_DK) {
        stage(qA, cA);
        if (k0 + 2 * CO_DK < d) fetch(qA, cA, k0 + 2 * CO_DK);
        accumulate();
        if (k0 + CO_DK < d) {
            stage(qB, cB);
            if (k0 + 3 * CO_DK < d) fetch(qB, cB, k0 + 3 * CO_DK);
            accumulate();
        }
    }
#pragma unroll
    for (int a = 0; a < RQ; ++a) {
        const int q = q0 + tq * RQ + a;
        if (q >= nq) continue;
        const int c = c0 + tc * 4;
        float *o = out + (size_t)q * kc + c;
        if (c + 3 < kc && ((kc & 3) == 0)) {
            *(float4 *)o = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (c + b < kc) o[b] = acc[a][b];
        }
    }
}

// ---------------------------------------------------------------------------------------
// Exact coarse distances for SMALL problems (every workgroup of the launch resident at once: the launch lasts as long
// as one workgroup, so what counts is the length of a wave's instruction stream, not the tile's operand reuse).
// coarse_dist_kernel<16> spends two LDS reads (a broadcast b32 and a b128) on every 12 VALU operations and is bound by
// the LDS pipe at a third of the VALU rate.  Here a LANE owns a centroid and a WAVE owns QW queries whose elements sit in
// SGPRs (scalar loads through the constant address space: the address is wave-uniform), so a k-step of QW x 3 VALU
// operations reads a quarter of one ds_read_b128 and nothing else: the kernel runs at the VALU rate.
// Workgroup tile: 64 centroids x 4 QW queries; centroid rows staged once per 128-wide d-chunk as Cs[c][k] with a row
// stride of 132 floats (16 lanes of a b128 service group -> 16 different 4-bank groups).  Same sums as above: k
// ascending, (c - q)^2 = sub, mul, add.  Needs d % 8 == 0.
// ---------------------------------------------------------------------------------------
#define CSQ_LD 132
#define CSQ_DK 128
typedef float v8f_a4 __attribute__((ext_vector_type(8), aligned(4)));
// (bx, by): the tile (64 centroids x 4 QW queries); Cs: 64 * CSQ_LD floats of LDS
template <int QW>
static __device__ __forceinline__ void coarse_sgpr_tile(const float *__restrict__ Q, const float *__restrict__ Cn, float *__restrict__ out, int nq,
                                                        int kc, int d, int bx, int by, float *Cs)
{
    typedef const __attribute__((address_space(4))) v8f_a4 *qptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = bx * 64;
    const int q0 = (by * 4 + wv) * QW;
    float acc[QW];
    const float *qrow[QW];
#pragma unroll
    for (int s = 0; s < QW; ++s) {
        acc[s] = 0.0f;
        const int qs = (q0 + s) < nq ? q0 + s : nq - 1;
        qrow[s] = Q + (size_t)qs * d;
    }
    // lgkmcnt(0) only: scalar loads return out of order, so any use of one waits for all of them -- the next step's
    // operands are therefore requested right AFTER this explicit wait and arrive behind the current step's arithmetic
    auto wait_lgkm = [] { __builtin_amdgcn_s_waitcnt(0xc07f); };
    for (int k0 = 0; k0 < d; k0 += CSQ_DK) {
        const int kn = (d - k0) < CSQ_DK ? d - k0 : CSQ_DK;   // multiple of 8
        if (k0) __syncthreads();
        {
            // all eight row segments in flight before the first LDS write; rows past kc and columns past kn are clamped
            // to valid addresses (their sums are never stored / never read)
            float4 st[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = tid + 256 * j, row = idx >> 5, c4 = (idx & 31) * 4;
                const int rr = (c0 + row) < kc ? c0 + row : kc - 1;
                st[j] = *(const float4 *)(Cn + (size_t)rr * d + k0 + (c4 < kn ? c4 : 0));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = tid + 256 * j, row = idx >> 5, c4 = (idx & 31) * 4;
                *(float4 *)&Cs[row * CSQ_LD + c4] = st[j];
            }
        }
        __syncthreads();
        const float *crow = &Cs[lane * CSQ_LD];
        float qa[QW][8], qb[QW][8], ca[8], cb[8];
        auto fetch = [&](float (&qv)[QW][8], float (&cv)[8], int k) {
            const float4 c0v = *(const float4 *)(crow + k), c1v = *(const float4 *)(crow + k + 4);
            cv[0] = c0v.x; cv[1] = c0v.y; cv[2] = c0v.z; cv[3] = c0v.w; cv[4] = c1v.x; cv[5] = c1v.y; cv[6] = c1v.z; cv[7] = c1v.w;
#pragma unroll
            for (int s = 0; s < QW; ++s) {
                const v8f_a4 v = *(qptr_t)(size_t)(qrow[s] + k0 + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) qv[s][e] = v[e];
            }
        };
        auto accumulate = [&](const float (&qv)[QW][8], const float (&cv)[8]) {
#pragma unroll
            for (int s = 0; s < QW; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = cv[e] - qv[s][e];
                    acc[s] = acc[s] + t * t;
                }
        };
        fetch(qa, ca, 0);
        for (int k = 0; k < kn; k += 16) {
            wait_lgkm();
            if (k + 8 < kn) fetch(qb, cb, k + 8);
            accumulate(qa, ca);
            if (k + 8 < kn) {
                wait_lgkm();
                if (k + 16 < kn) fetch(qa, ca, k + 16);
                accumulate(qb, cb);
            }
        }
    }
    if (c0 + lane < kc) {
#pragma unroll
        for (int s = 0; s < QW; ++s)
            if (q0 + s < nq) out[(size_t)(q0 + s) * kc + c0 + lane] = acc[s];
    }
}

template <int QW>
__global__ __launch_bounds__(256) void coarse_sgpr_kernel(const float *__restrict__ Q, const float *__restrict__ Cn,
                                                          float *__restrict__ out, int nq, int kc, int d)
{
    __shared__ __attribute__((aligned(16))) float Cs[64 * CSQ_LD];
    coarse_sgpr_tile<QW>(Q, Cn, out, nq, kc, d, (int)blockIdx.x, (int)blockIdx.y, Cs);
}

// ---------------------------------------------------------------------------------------
// Coarse FILTER on the matrix cores: score[q][c] = ||c||^2 - 2 q.c  (= distance - ||q||^2, approximately).
// f32 MFMA 16x16x4 (exact-f32 fma chain, 64 FLOP/clk/SIMD = 2/3 fewer issue slots than the 3-op VALU form).
// The scores only RANK candidates; the distances that reach the result are recomputed in the oracle's
// 3-op order by refine_probes(), which also certifies that no candidate can be missing.
// Workgroup tile 128 queries x 128 centroids, 4 waves as 2 x 2, each wave 64 x 64 = 4 x 4 MFMA blocks.
// ---------------------------------------------------------------------------------------
// TB = tile edge (128: each wave 64 x 64 = 4 x 4 MFMA blocks; 64: each wave 32 x 32, four times the workgroups
// for batches that would not fill the chip).  BK = depth of a staged k-chunk (16; 64 for small problems, where a
// workgroup's chain of chunk latencies IS the launch: 1024 x 1024 x 128 ran 11 us with eight 16-deep chunks)
template <int TB, int BK>
__global__ __launch_bounds__(256) void coarse_mfma_kernel(const float *__restrict__ Q, const float *__restrict__ Cn,
                                                          const float *__restrict__ cnorm, float *__restrict__ out, int nq, int kc,
                                                          int d, float *__restrict__ tmin, int ntiles)
{
    constexpr int NB = TB / 32;          // 16 x 16 blocks per wave per dimension
    constexpr int LD = TB + 16;          // consecutive k rows start 16 banks apart -> conflict-free fragment reads
    constexpr int NP = TB / 64;          // 64-row load passes per operand
    constexpr int NK = BK / 16;          // 16-deep slices per chunk
    __shared__ __attribute__((aligned(16))) float As[BK][LD];   // [k][query]
    __shared__ __attribute__((aligned(16))) float Bs[BK][LD];   // [k][centroid]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wq = wv >> 1, wc = wv & 1;                 // wave position in the 2 x 2 grid
    const int q0 = blockIdx.y * TB, c0 = blockIdx.x * TB;
    v4f acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    const int lrow = tid >> 2, lk = (tid & 3) * 4;       // rows 0..63 (+64 per pass), k offset 0,4,8,12 (+16 per slice)
    float4 av[NK][NP], bv[NK][NP];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int s = 0; s < NK; ++s)
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const int r = lrow + 64 * h;
                const int qi = q0 + r, ci = c0 + r;
                const int i0 = k0 + 16 * s + lk;
                av[s][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                bv[s][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i0 + 3 < d) {   // d % 4 == 0 is required by the host for this kernel
                    if (qi < nq) av[s][h] = *(const float4 *)(Q + (size_t)qi * d + i0);
                    if (ci < kc) bv[s][h] = *(const float4 *)(Cn + (size_t)ci * d + i0);
                }
            }
    };
    fetch(0);
    for (int k0 = 0; k0 < d; k0 += BK) {
        __syncthreads();   // the previous chunk's fragments have been read
#pragma unroll
        for (int s = 0; s < NK; ++s)
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const int r = lrow + 64 * h, kb = 16 * s + lk;
                As[kb + 0][r] = av[s][h].x; As[kb + 1][r] = av[s][h].y; As[kb + 2][r] = av[s][h].z; As[kb + 3][r] = av[s][h].w;
                Bs[kb + 0][r] = bv[s][h].x; Bs[kb + 1][r] = bv[s][h].y; Bs[kb + 2][r] = bv[s][h].z; Bs[kb + 3][r] = bv[s][h].w;
            }
        __syncthreads();
        if (k0 + BK < d) fetch(k0 + BK);   // in flight under the MFMAs of this chunk
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            // A fragment: lane l holds A[row l&15][k l>>4]; B fragment: B[k l>>4][col l&15]
            float af[NB], bf[NB];
            const int kr = kk + (lane >> 4);
#pragma unroll
            for (int i = 0; i < NB; ++i) af[i] = As[kr][wq * (TB / 2) + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < NB; ++j) bf[j] = Bs[kr][wc * (TB / 2) + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                // centroids as the A operand: the 16 x 16 result block is [centroid][query], so a lane's four
                // registers are four CONSECUTIVE centroids of one query -> one 16-byte store each (epilogue)
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    // C/D layout of 16x16: col (= query) = lane & 15, row (= centroid) = (lane >> 4) * 4 + reg
    float rmin[NB];   // per query of this lane: minimum score over the lane's share of the wave's TB/2 centroids
#pragma unroll
    for (int i = 0; i < NB; ++i) rmin[i] = __builtin_inff();
    const bool vec_ok = (kc & 3) == 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int c = c0 + wc * (TB / 2) + j * 16 + (lane >> 4) * 4;   // first of the lane's four centroids
        float cn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cn[r] = (c + r) < kc ? cnorm[c + r] : 0.f;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = q0 + wq * (TB / 2) + i * 16 + (lane & 15);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = cn[r] - 2.0f * acc[i][j][r];
            if (q < nq) {
                float *o = out + (size_t)q * kc + c;
                if (vec_ok && c < kc) {
                    *(float4 *)o = make_float4(v[0], v[1], v[2], v[3]);   // kc % 4 == 0, c % 4 == 0: c + 3 < kc
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < kc) o[r] = v[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c + r < kc) rmin[i] = fminf(rmin[i], v[r]);
            }
        }
    }
    if (tmin) {
        const int tile = blockIdx.x * 2 + wc;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            float v = rmin[i];   // the query's four lanes sit 16 apart
            v = fminf(v, __shfl_xor(v, 16));
            v = fminf(v, __shfl_xor(v, 32));
            const int q = q0 + wq * (TB / 2) + i * 16 + (lane & 15);
            if (lane < 16 && q < nq && tile < ntiles) tmin[(size_t)q * ntiles + tile] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same filter on the BF16 matrix cores (round 2, large problems): every operand is split into two bf16 pieces,
// x = hi + lo + O(2^-18 |x|)  (hi = bf16(x), lo = bf16(x - hi); x - hi is exact in f32), and
//     q . c  ~  qh . ch + qh . cl + ql . ch          (three v_mfma_f32_16x16x32_bf16, f32 accumulation; ql . cl ~ 2^-18 dropped)
// bf16 MFMA runs at 16x the f32 rate, so three of them cost under a fifth of the f32 kernel's matrix time; the score error
// grows from ~2 (d + 3) u to ~2 (97 + 2.5 d + 8) u (u = 2^-24; coarse_eps_coef), which the certificate of refine_probes
// absorbs exactly as before -- scores only RANK, every distance that reaches a result is recomputed in the oracle's order.
// The split operands are prepared once: centroids at index creation (immutable afterwards), the batch's queries by
// split_bf16_kernel.  Rows are zero-padded to a multiple of 32 dimensions (dp).
// Tile 128 x 128 per workgroup, 4 waves as 2 x 2, 64 x 64 per wave; k-steps of 32; LDS image [k-group of 8][row][8 bf16],
// so the 16 lanes of a ds_read_b128 service group (16 different rows mod 16, one or two k-groups whose planes are a multiple
// of 256 bytes apart) read 16 different bank groups: conflict-free fragments.
// ---------------------------------------------------------------------------------------
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

static __device__ __forceinline__ unsigned short f32_to_bf16_rne(float x)
{
    const u32 b = __float_as_uint(x);
    return (unsigned short)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);   // finite inputs only (the contract excludes NaN)
}

// x[rows][d] f32 -> hi / lo bf16 [rows][dp], zero-padded
__global__ __launch_bounds__(256) void split_bf16_kernel(const float *__restrict__ x, int64_t rows, int d, int dp,
                                                         unsigned short *__restrict__ hi, unsigned short *__restrict__ lo)
{
    const int64_t total = rows * dp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / dp;
        const int i = (int)(e - r * dp);
        unsigned short h = 0, l = 0;
        if (i < d) {
            const float v = x[r * d + i];
            h = f32_to_bf16_rne(v);
            l = f32_to_bf16_rne(v - __uint_as_float((u32)h << 16));
        }
        hi[e] = h;
        lo[e] = l;
    }
}

// TB = tile edge: 128 (each wave 64 x 64) for problems that fill the chip, 64 (each wave 32 x 32, four times the workgroups)
// below that.
// ---- round 5: the same filter with ONE product per score on the F16 matrix path ------------------------------------------------------
// f16 carries 11 significant bits: x^ = fl16(s x) / s with a power-of-two scale s (2^11 <= s max|c| <= 2^12, far from overflow and from the
// subnormals that matter) has |x^ - x| <= 2^-11 |x| + 2^-25 / s, so the single product q^ . c^ (products of two f16 are exact in f32, the
// accumulation is f32) misses q . c by at most (2^-10 + 2^-22) |q||c| + (abs term < 2^-36 sqrt(d) (|q| + |c|)^2): in the score,
// 2^-11 (||c|| + ||q||)^2 = 8192 u against the 97 u of the three-product bf16 split -- eighty times looser, and still far inside what the
// certificate absorbs (Deep1B shape: eps 0.38 on scores whose 32nd and 64th smallest lie 5 apart).  One v_mfma_f32_16x16x32_f16 instead of
// three bf16 ones, two operand arrays instead of four.  A query component that would overflow the f16 range saturates and FLAGS its
// query: the selection recomputes a flagged query exactly (RefineArgs::qflags), so the bound is never relied on where it does not hold.
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void split_f16_kernel(const float *__restrict__ x, int64_t rows, int d, int dp, float scale,
                                                        unsigned short *__restrict__ out, u32 *__restrict__ flags)
{
    const int64_t total = rows * dp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / dp;
        const int i = (int)(e - r * dp);
        unsigned short h = 0;
        if (i < d) {
            float v = x[r * d + i] * scale;
            if (!(fabsf(v) <= 65000.0f)) {          // out of range (or NaN): saturate, and let the selection take the exact path for this query
                v = v > 0.0f ? 65000.0f : (v < 0.0f ? -65000.0f : 0.0f);
                if (flags) atomicOr(&flags[r], 1u);
            }
            const _Float16 hv = (_Float16)v;
            h = __builtin_bit_cast(unsigned short, hv);
        }
        out[e] = h;
    }
}

// score bits <-> unsigned keys that order like the (signed) float
static __device__ __forceinline__ u32 ordered_bits(float f)
{
    const u32 b = __float_as_uint(f);
    return b ^ ((u32)((int)b >> 31) | 0x80000000u);
}
static __device__ __forceinline__ float ordered_to_float(u32 o)
{
    return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}

// ---- listed mode: the four smallest scores of every (query, 64-centroid tile) instead of the score matrix ---------------
// A tile's record is four 32-bit keys, ascending: ordered_bits(score) with its low 6 bits replaced by the centroid's
// position inside the tile (so a key names its centroid and keys of one tile are unique; the score moves by < 64 ulp =
// 2^-17 |score|, which refine_args() adds to the error bound).  0xFFFFFFFF = no centroid (tile cut off by kc).
static __device__ __forceinline__ void cex(u32 &a, u32 &b)
{
    const u32 lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
}
static __device__ __forceinline__ void sort4_u32(u32 (&k)[4])
{
    cex(k[0], k[1]); cex(k[2], k[3]); cex(k[0], k[2]); cex(k[1], k[3]); cex(k[1], k[2]);
}
// a, b ascending -> a = the four smallest of the eight, ascending
static __device__ __forceinline__ void merge4_low(u32 (&a)[4], const u32 (&b)[4])
{
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = a[e] < b[3 - e] ? a[e] : b[3 - e];   // bitonic: holds the four smallest
    cex(a[0], a[2]); cex(a[1], a[3]); cex(a[0], a[1]); cex(a[2], a[3]);
}

// DBG != 0: knock-outs for tools/coarse_probe.py (wrong results by design; instantiated in -DIVFADC_DEBUG builds only, chosen by IVFADC_COARSE_DBG):
// 1 = no epilogue at all (loads and matrix work alone), 2 = scores and a per-lane minimum against a threshold that nothing meets (what a
// threshold filter would cost), 3 = all of the record arithmetic but no store, 5 = no matrix instructions, 6 = no operand loads (5, 6: epilogue of 1),
// 7 = records stored but not the tile minima, 8 = the minima but not the records
// (256, 3): three workgroups per CU -- without the bound the listed epilogue is scheduled into 200 registers (two per CU)
// F16: one f16 product per score (Qh / Ch hold scaled f16 rows, Ql / Cl are not read; neg2 = -2 / (scale_q scale_c)); otherwise the
// three-product bf16 split (neg2 = -2)
template <int TB, int DBG = 0, bool F16 = false>
__global__ __launch_bounds__(256, 3) void coarse_bf16_kernel(const unsigned short *__restrict__ Qh, const unsigned short *__restrict__ Ql,
                                                          const unsigned short *__restrict__ Ch, const unsigned short *__restrict__ Cl,
                                                          const float *__restrict__ cnorm, float *__restrict__ out, int nq, int kc, int dp,
                                                          float *__restrict__ tmin, int ntiles, uint4 *__restrict__ tlist, int ldq, float neg2)
{
    constexpr int NB = TB / 32;           // 16 x 16 blocks per wave per dimension
    constexpr int WT = TB / 2;            // wave tile edge
    constexpr int GPT = TB / 64;          // 16-byte groups per thread and operand part per k-step (4 k-groups x TB rows / 256 threads)
    constexpr int EP_LD = WT + 4;         // floats per staged epilogue row
    // [part: Qh, Ql, Ch, Cl][k-group 0..3][row] x 16 bytes, and room for the epilogue's staging (4 waves x 32 rows x EP_LD floats)
    constexpr int LS_OPER = 4 * 4 * TB, LS_STAGE = (4 * 32 * EP_LD * 4 + 15) / 16;
    __shared__ __attribute__((aligned(16))) uint4 Ls[LS_OPER > LS_STAGE ? LS_OPER : LS_STAGE];
    auto L = [&](int p, int kg, int row) -> uint4 & { return Ls[(p * 4 + kg) * TB + row]; };
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wq = wv >> 1, wc = wv & 1;
    const int q0 = blockIdx.y * TB, c0 = blockIdx.x * TB;
    v4f acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    // staging: TB = 128: thread -> (row = tid >> 1, k-groups 2 (tid & 1), 2 (tid & 1) + 1); TB = 64: (row = tid >> 2, k-group tid & 3)
    const int lrow = TB == 128 ? tid >> 1 : tid >> 2, lkg = TB == 128 ? (tid & 1) * 2 : (tid & 3);
    const int qi = q0 + lrow, ci = c0 + lrow;
    const bool qok = qi < nq, cok = ci < kc;
    const unsigned short *src[4] = {Qh + (size_t)(qok ? qi : 0) * dp, F16 ? Qh : Ql + (size_t)(qok ? qi : 0) * dp,
                                    Ch + (size_t)(cok ? ci : 0) * dp, F16 ? Ch : Cl + (size_t)(cok ? ci : 0) * dp};
    uint4 pre[4][GPT];
    auto fetch = [&](int k0) {
        if constexpr (DBG == 6) {   // no operand traffic: what the loop costs with its loads answered at once
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int g = 0; g < GPT; ++g) pre[p][g] = make_uint4((u32)k0 + (u32)tid, 0x3f803f80u, (u32)p, 0x3f803f80u);
            return;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int g = 0; g < GPT; ++g) {
                if (F16 && (p & 1)) continue;   // no lo parts
                const bool ok = p < 2 ? qok : cok;
                pre[p][g] = ok ? *(const uint4 *)(src[p] + k0 + 8 * (lkg + g)) : make_uint4(0u, 0u, 0u, 0u);
            }
    };
    fetch(0);
    for (int k0 = 0; k0 < dp; k0 += 32) {
        __syncthreads();   // the previous step's fragments have been read
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int g = 0; g < GPT; ++g) {
                if (F16 && (p & 1)) continue;
                L(p, lkg + g, lrow) = pre[p][g];
            }
        __syncthreads();
        if (k0 + 32 < dp) fetch(k0 + 32);   // in flight under this step's MFMAs
        const int kg = lane >> 4, rl = lane & 15;
        if constexpr (F16) {
            v8h qf[NB], cf[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) qf[i] = __builtin_bit_cast(v8h, L(0, kg, wq * WT + i * 16 + rl));
#pragma unroll
            for (int j = 0; j < NB; ++j) cf[j] = __builtin_bit_cast(v8h, L(2, kg, wc * WT + j * 16 + rl));
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cf[j], qf[i], acc[i][j], 0, 0, 0);
            continue;
        }
        v8bf qh[NB], ql[NB], ch[NB], cl[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            qh[i] = __builtin_bit_cast(v8bf, L(0, kg, wq * WT + i * 16 + rl));
            ql[i] = __builtin_bit_cast(v8bf, L(1, kg, wq * WT + i * 16 + rl));
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            ch[j] = __builtin_bit_cast(v8bf, L(2, kg, wc * WT + j * 16 + rl));
            cl[j] = __builtin_bit_cast(v8bf, L(3, kg, wc * WT + j * 16 + rl));
        }
        if constexpr (DBG == 5) {   // no matrix work: loads, LDS traffic and barriers alone
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j][0] += (float)cl[j][0] + (float)qh[i][0] + (float)ch[j][1] + (float)ql[i][1];
            continue;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                // centroids as the A operand: the 16 x 16 result block is [centroid][query] (see coarse_mfma_kernel)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cl[j], qh[i], acc[i][j], 0, 0, 0);   // small terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[j], ql[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[j], qh[i], acc[i][j], 0, 0, 0);
            }
    }
    // Epilogue.  C/D layout: col (= query) = lane & 15, row (= centroid) = (lane >> 4) * 4 + reg, so a lane's four registers are
    // four consecutive scores of one query and a direct store instruction would write 16 rows x 64 bytes -- half cache
    // lines, measured at 2.3 TB/s of the 2.6 GB score matrix (0.53 of this kernel's 1.15 ms on the Deep1B shape).  Each
    // wave therefore transposes its block through its own slice of the (now idle) operand LDS, 32 queries at a time, and
    // stores whole row segments (TB = 128: 256 bytes, two full lines per query row, four rows per instruction).  No
    // workgroup barrier: a wave only touches its own slice (row stride WT + 4 floats: the 16 lanes of a ds_write_b128 group
    // sit in 16 different rows, an odd number of bank groups apart).
    if constexpr (TB == 128) {
        if (tlist) {
            // Listed mode (stand-alone top-w behind this kernel, see select_listed): no score matrix.  The wave's block is ONE
            // (64 queries x 64-centroid tile); query (i, lane & 15) has its 64 scores in the four lanes lane & 15 + 16 g,
            // sixteen each (block j, register r: centroid j * 16 + g * 4 + r of the tile).  Lane-local: sort each block's four,
            // merge keeping the four smallest; then two cross-lane merges.  Records are tile-major, tlist[tile][query]: the
            // wave writes 64 consecutive 16-byte records.  Scores by one fma (they only rank; the error bound covers either form).
            const int tile = blockIdx.x * 2 + wc, g = lane >> 4;
            const int cb = c0 + wc * WT + g * 4;              // + j * 16: the lane's four consecutive centroids of block j
            const int q16 = q0 + wq * WT + (lane & 15);       // + i * 16
            if constexpr (DBG == 1 || DBG == 5 || DBG == 6) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
                if (t == 1.2345e-30f) tmin[0] = t;
                return;
            }
            if constexpr (DBG == 2) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    float m = __builtin_inff();
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const float4 t = *(const float4 *)(cnorm + (cb + j * 16 < kc - 4 ? cb + j * 16 : 0));
                        const float v0 = __builtin_fmaf(neg2, acc[i][j][0], t.x), v1 = __builtin_fmaf(neg2, acc[i][j][1], t.y);
                        const float v2 = __builtin_fmaf(neg2, acc[i][j][2], t.z), v3 = __builtin_fmaf(neg2, acc[i][j][3], t.w);
                        m = fminf(fminf(m, fminf(v0, v1)), fminf(v2, v3));
                    }
                    if (__builtin_amdgcn_ballot_w64(m < -3.0e38f)) tmin[(size_t)(q16 + i * 16) * ntiles + tile] = m;
                }
                return;
            }
            auto run = [&](auto ragged_tag) {
                constexpr bool RAG = decltype(ragged_tag)::value;   // the tile is cut off by kc: clamp loads, blank the keys past kc
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    u32 best[4];
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        // ||c||^2 of the lane's four centroids: re-read per query block (L1 hits) rather than held in 16 VGPRs
                        // across the whole epilogue -- the kernel must stay under 168 for three workgroups per CU
                        const int c = cb + j * 16;
                        float cn[4];
                        if constexpr (!RAG) {
                            const float4 t = *(const float4 *)(cnorm + c);
                            cn[0] = t.x; cn[1] = t.y; cn[2] = t.z; cn[3] = t.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) cn[r] = cnorm[(c + r) < kc ? c + r : kc - 1];
                        }
                        u32 k[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = __builtin_fmaf(neg2, acc[i][j][r], cn[r]);
                            k[r] = (ordered_bits(v) & ~63u) | (u32)(j * 16 + g * 4 + r);
                            if constexpr (RAG) k[r] = (c + r) < kc ? k[r] : 0xFFFFFFFFu;
                        }
                        sort4_u32(k);
                        if (j == 0) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) best[e] = k[e];
                        } else {
                            merge4_low(best, k);
                        }
                    }
                    // rows 0 and 2 of the wave take rows 1 and 3 (v_permlane16_swap: second result = [r1 r1 r3 r3]), then the
                    // lower half takes the upper one (v_permlane32_swap: second result = [hi hi]); only row 0 is stored
                    {
                        u32 o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = __builtin_amdgcn_permlane16_swap(best[e], best[e], false, false)[1];
                        merge4_low(best, o);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = __builtin_amdgcn_permlane32_swap(best[e], best[e], false, false)[1];
                        merge4_low(best, o);
                    }
                    const int q = q16 + i * 16;
                    if (DBG == 3 && best[0] != 0x12345u) continue;
                    if (g == 0 && q < nq && tile < ntiles) {
                        if (DBG != 8) tlist[(size_t)tile * ldq + q] = make_uint4(best[0], best[1], best[2], best[3]);
                        if (DBG != 7) tmin[(size_t)q * ntiles + tile] = ordered_to_float(best[0]);
                    }
                }
            };
            if (c0 + TB <= kc && (kc & 3) == 0) run(std::false_type{});
            else run(std::true_type{});
            return;
        }
    }
    float rmin[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) rmin[i] = __builtin_inff();
    __syncthreads();   // every wave has read its last fragments: the operand LDS is free
    float *stage = (float *)Ls + (size_t)wv * 32 * EP_LD;
    const bool vec_ok = (kc & 3) == 0;
#pragma unroll
    for (int hh = 0; hh < NB / 2; ++hh) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int c = c0 + wc * WT + j * 16 + (lane >> 4) * 4;
            float cn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cn[r] = (c + r) < kc ? cnorm[c + r] : 0.f;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                const int i = hh * 2 + i2;
                const int q = q0 + wq * WT + i * 16 + (lane & 15);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = cn[r] + neg2 * acc[i][j][r];
                *(float4 *)&stage[(i2 * 16 + (lane & 15)) * EP_LD + j * 16 + (lane >> 4) * 4] = make_float4(v[0], v[1], v[2], v[3]);
                if (q < nq) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < kc) rmin[i] = fminf(rmin[i], v[r]);
                }
            }
        }
        wave_sync();
        // 32 rows x WT floats: WT / 4 lanes per row, 256 / WT rows per pass, WT / 8 passes
        constexpr int LPR = WT / 4, RPP = 64 / LPR;
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int rl2 = ps * RPP + lane / LPR;
            const int q = q0 + wq * WT + hh * 32 + rl2;
            const int c = c0 + wc * WT + (lane % LPR) * 4;
            const float4 v = *(const float4 *)&stage[rl2 * EP_LD + (lane % LPR) * 4];
            if (q < nq) {
                float *o = out + (size_t)q * kc + c;
                if (vec_ok && c < kc) {
                    *(float4 *)o = v;   // kc % 4 == 0, c % 4 == 0: c + 3 < kc
                } else {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < kc) o[r] = vv[r];
                }
            }
        }
        wave_sync();
    }
    if (tmin) {
        const int tile = blockIdx.x * 2 + wc;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            float v = rmin[i];
            v = fminf(v, __shfl_xor(v, 16));
            v = fminf(v, __shfl_xor(v, 32));
            const int q = q0 + wq * WT + i * 16 + (lane & 15);
            if (lane < 16 && q < nq && tile < ntiles) tmin[(size_t)q * ntiles + tile] = v;
        }
    }
}

// ---- certified refine ------------------------------------------------------------------------------
// size of the candidate pool kept from the MFMA scores: w + max(16, w) <= 64 for w <= 48 (the certificate needs the
// pool's LAST member to lie beyond the error margin; extra members beyond w are what absorbs near-ties)
static __host__ __device__ __forceinline__ int approx_pool(int w)
{
    const int p = w + (w > 16 ? w : 16);
    return p < 64 ? p : 64;
}

struct RefineArgs {
    const float *queries;      // [nq][d]
    const float *centroids;    // [kc][d]
    int d, kc;
    float cmaxn;               // >= max_c ||c||
    float eps_coef;            // 2 (d+3) u
    float gam;                 // 4 (d+2) u
    u64 *fallbacks;            // statistics: queries whose certificate failed (exact fallback taken)
    // per-(query, centroid tile) minimum score written by coarse_mfma_kernel (null: not available); tile_w centroids
    // per tile: the stand-alone top-w reads only the tiles that can hold one of the pool's keys
    const float *tmin;
    int ntiles, tile_w;
    // listed mode (coarse_bf16_kernel with tlist): tile-major records of each (query, tile)'s four smallest keys instead of a
    // score matrix; null otherwise.  ldq = queries per tile row
    const uint4 *tlist;
    int ldq;
    // f16 filter: queries whose scaled components left the f16 range (split_f16_kernel): their scores carry no bound -- exact path
    const u32 *qflags;
};

// oracle-order exact distance of one centroid row (coarsequantizers.jl:34): sequential, no FMA; d % 4 == 0
static __device__ __forceinline__ float exact_coarse_dist(const float *crow, const float *qv, int d)
{
    // 64 bytes of each operand are requested before the first of them is used: a lane walks its own centroid row,
    // so every 16-byte piece is a separate trip to L2 / HBM and the sum itself is a serial chain
    float acc = 0.0f;
    for (int i = 0; i < d; i += 16) {
        float4 c4[4], q4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = i + 4 * u < d ? i + 4 * u : i;   // d % 4 == 0; past the end: re-read a valid piece, not used
            c4[u] = *(const float4 *)(crow + k);
            q4[u] = *(const float4 *)(qv + k);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + 4 * u < d) {
                float t = c4[u].x - q4[u].x; acc = acc + t * t;
                t = c4[u].y - q4[u].y; acc = acc + t * t;
                t = c4[u].z - q4[u].z; acc = acc + t * t;
                t = c4[u].w - q4[u].w; acc = acc + t * t;
            }
        }
    }
    return acc;
}

// One wave.  `ap` holds the cnt (<= 64) smallest MFMA scores of query q (lane j = j-th smallest; key =
// ordered_bits(score) << 32 | centroid).  Returns the oracle's top-w: exact distances, ties to the lower id.
//
// Why the candidate set is complete.  With u = 2^-24, D_c the real squared distance, O_c the oracle's float and
// A_c = fl(score_c + fl(||q||^2)):  |O_c - D_c| <= g D_c with g = (d+2)u/(1-(d+2)u)  (d non-negative terms, each
// (c-q)^2 rounded twice, then d-1 additions);  |A_c - D_c| <= (d+4)u(||c||+||q||)^2 <= eps  (rounded norms, the
// MFMA's fma chain on q.c, two final roundings).  Let tau = the w-th smallest A.  The w centroids with the smallest A
// all have O <= (tau+eps)(1+g) =: U, so every member of the oracle's top-w has O <= U, hence D <= U/(1-g) and
// A <= U/(1-g) + eps =: T.  Taking every selected centroid with A <= T (T inflated: gam = 4(d+2)u >= 2g, eps doubled) and
// requiring the LAST selected one to have A > T (or every centroid to be selected) therefore loses nobody; the
// selection by score equals the selection by A because x -> fl(x + const) is monotone.  If the requirement fails
// (more than 64 - w near-ties), the wave recomputes all kc distances exactly.
static __device__ WSel<true> refine_probes(const WSel<true> &ap, int cnt, int w, int q, const RefineArgs &r, int lane)
{
    const float *qv = r.queries + (size_t)q * r.d;
    float part = 0.0f;
    for (int i = lane; i < r.d; i += 64) part += qv[i] * qv[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    const float qn = part;
    const int c = (int)(u32)ap.top;
    const float A = ordered_to_float((u32)(ap.top >> 32)) + qn;
    const float tau = __shfl(A, w - 1);
    const float sumn = r.cmaxn + sqrtf(qn) * 1.00001f;
    const float eps = r.eps_coef * sumn * sumn;
    float T = (tau + eps) * (1.0f + r.gam) + eps;
    T = T + fabsf(T) * 1e-6f;
    const float Alast = __shfl(A, cnt - 1);
    const bool flagged = r.qflags != nullptr && r.qflags[q] != 0u;
    const bool complete = !flagged && ((cnt >= r.kc) || (Alast > T));
    WSel<true> ex;
    ex.init(KEY_MAX, nullptr, 64, w);
    if (complete) {
        const bool cand = lane < cnt && A <= T;
        float dist = 0.0f;
        if (cand) dist = exact_coarse_dist(r.centroids + (size_t)c * r.d, qv, r.d);
        ex.push(cand, make_key(dist, (u32)c), w, lane);
    } else {
        if (lane == 0) atomicAdd(r.fallbacks, 1ull);
        for (int c0 = 0; c0 < r.kc; c0 += 64) {
            const int cc = c0 + lane;
            const bool ok = cc < r.kc;
            const float dist = ok ? exact_coarse_dist(r.centroids + (size_t)cc * r.d, qv, r.d) : 0.0f;
            const u64 key = make_key(dist, (u32)cc);
            ex.push(ok && key < ex.thr(), key, w, lane);
        }
    }
    return ex;
}

// One wave, listed mode: the oracle's top-w of query q from the per-tile records of coarse_bf16_kernel -- no score matrix, no
// candidate pool, no certificate.  With A'_c = fl(score'_c + ||q||^2) the approximation carried by a key (score' = the score
// with its low 6 bits replaced, |A' - D| <= eps with the 2^-17 of that replacement inside eps_coef):
//   (a) bound = the w-th smallest tile minimum: w different tiles hold a key <= it, so tau (the w-th smallest A') <= bound;
//   (b) tau' = the w-th smallest LISTED key among the tiles whose minimum is <= bound (>= tau: a subset of all keys);
//   (c) T = (tau' + eps)(1 + gam) + eps: every member of the oracle's top-w has A' <= T (refine_probes' argument, which
//       only needs tau' >= tau);
//   (d) every centroid with A' <= T is enumerated: its tile has a minimum <= T; a record lists, in ascending order, all of its
//       tile's keys up to its last one, so when that last key is > T the record holds every key <= T of the tile, and when it
//       is not (an "overflow" tile: more than three keys under T) the tile's 64 centroids are all taken;
//   (e) the enumerated centroids get their distances in the oracle's order and the w smallest (distance, id) win.
// More qualifying tiles than the LDS list holds (ties en masse, cancelling scores): every distance is recomputed exactly.
constexpr int LISTED_MAX = 128;
static __device__ WSel<true> select_listed(int q, int w, const RefineArgs &r, int lane, int *tl /* LDS, per wave, LISTED_MAX ints */)
{
    const float *qv = r.queries + (size_t)q * r.d;
    const float *tm = r.tmin + (size_t)q * r.ntiles;
    float part = 0.0f;
    for (int i = lane; i < r.d; i += 64) part += qv[i] * qv[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    const float qn = part;
    WSel<true> ex;
    ex.init(KEY_MAX, nullptr, 64, w);
    // tiles whose minimum passes `pass`, in order, into tl; returns how many there are (possibly more than fit)
    auto gather = [&](auto pass) {
        int nt = 0;
        for (int t0 = 0; t0 < r.ntiles; t0 += 64) {
            const int t = t0 + lane;
            const bool qual = t < r.ntiles && pass(tm[t]);
            const u64 mask = __ballot(qual);
            const int pos = nt + __popcll(mask & ((1ull << lane) - 1ull));
            if (qual && pos < LISTED_MAX) tl[pos] = t;
            nt += __popcll(mask);
        }
        wave_sync();
        return nt;
    };
    // lane -> (tile b0 + lane / 4 of the list, entry lane & 3): the 32-bit key, 0xFFFFFFFF when there is none
    auto entry = [&](int b0, int nt, int &tile) -> u32 {
        const int ti = b0 + (lane >> 2);
        tile = ti < nt ? tl[ti] : 0;
        return ti < nt ? ((const u32 *)(r.tlist + (size_t)tile * r.ldq + q))[lane & 3] : 0xFFFFFFFFu;
    };
    bool ok = false;   // uniform
    float T = 0.0f;
    const bool flagged = r.qflags != nullptr && r.qflags[q] != 0u;
    if (!flagged) {
        WSel<true> ts;
        ts.init(KEY_MAX, nullptr, 64, w);
        for (int t0 = 0; t0 < r.ntiles; t0 += 64) {
            const int t = t0 + lane;
            const bool in = t < r.ntiles;
            const u64 key = in ? (((u64)ordered_bits(tm[t]) << 32) | (u32)t) : KEY_MAX;
            ts.push(in && key < ts.thr(), key, w, lane);
        }
        const int tc = ts.finish(w, lane);
        if (tc >= w) {
            const u32 bound = (u32)(readlane64(ts.top, w - 1) >> 32);
            const int nt = gather([&](float v) { return ordered_bits(v) <= bound; });
            if (nt <= LISTED_MAX) {
                WSel<true> es;
                es.init(((u64)bound + 1ull) << 32, nullptr, 64, w);
                for (int b0 = 0; b0 < nt; b0 += 16) {
                    int tile;
                    const u32 k32 = entry(b0, nt, tile);
                    const u64 key = ((u64)k32 << 32) | (u32)(tile * r.tile_w + (int)(k32 & 63u));
                    es.push(k32 != 0xFFFFFFFFu && key < es.thr(), key, w, lane);
                }
                if (es.finish(w, lane) >= w) {   // always: each of the w tiles behind `bound` lists its minimum
                    const float tau = ordered_to_float((u32)(readlane64(es.top, w - 1) >> 32)) + qn;
                    const float sumn = r.cmaxn + sqrtf(qn) * 1.00001f;
                    const float eps = r.eps_coef * sumn * sumn;
                    T = (tau + eps) * (1.0f + r.gam) + eps;
                    T = T + fabsf(T) * 1e-6f;
                    ok = true;
                }
            }
            wave_sync();   // the list is rewritten below
        }
    }
    int nt = 0;
    if (ok) {
        nt = gather([&](float v) { return v + qn <= T; });
        ok = nt <= LISTED_MAX;
    }
    if (!ok) {
        if (lane == 0) atomicAdd(r.fallbacks, 1ull);
        for (int c0 = 0; c0 < r.kc; c0 += 64) {
            const int cc = c0 + lane;
            const bool in = cc < r.kc;
            const float dist = in ? exact_coarse_dist(r.centroids + (size_t)cc * r.d, qv, r.d) : 0.0f;
            const u64 key = make_key(dist, (u32)cc);
            ex.push(in && key < ex.thr(), key, w, lane);
        }
        return ex;
    }
    for (int b0 = 0; b0 < nt; b0 += 16) {
        int tile;
        const u32 k32 = entry(b0, nt, tile);
        const bool under = k32 != 0xFFFFFFFFu && ordered_to_float(k32) + qn <= T;
        const u64 ovf = __ballot(under && (lane & 3) == 3);            // records whose LAST key is still under T
        const bool cand = under && !((ovf >> (lane | 3)) & 1ull);
        const int c = tile * r.tile_w + (int)(k32 & 63u);
        float dist = 0.0f;
        if (cand) dist = exact_coarse_dist(r.centroids + (size_t)c * r.d, qv, r.d);
        const u64 key = make_key(dist, (u32)c);
        ex.push(cand && key < ex.thr(), key, w, lane);
        for (u64 m = ovf; m; m &= m - 1) {                            // uniform: whole tiles
            const int ot = tl[b0 + (__builtin_ctzll(m) >> 2)];
            for (int c0 = 0; c0 < r.tile_w; c0 += 64) {
                const int cc = ot * r.tile_w + c0 + lane;
                const bool in = c0 + lane < r.tile_w && cc < r.kc;
                const float dd = in ? exact_coarse_dist(r.centroids + (size_t)cc * r.d, qv, r.d) : 0.0f;
                const u64 k2 = make_key(dd, (u32)cc);
                ex.push(in && k2 < ex.thr(), k2, w, lane);
            }
        }
    }
    return ex;
}

// The same for a whole workgroup (query-major prologue): wave 0 holds `ap` and decides; the candidates' centroid rows
// and the query are staged through LDS in d-chunks by all 256 threads (coalesced 16-byte loads, many in flight), and
// wave 0's candidate lanes accumulate from LDS in the oracle's order.  A lane walking its own row in global memory pays
// a trip to L2 / HBM per 64 bytes: 78 k cycles for 24 candidates at d = 768 with three waves idle; staged: ~20 k.
// Every thread of the workgroup must call it (barriers inside); the result is valid in wave 0.
// stage: LDS scratch of stage_floats floats (the table area, not yet built); s_cand: 64 + 2 ints of LDS scratch.
// WIDE (wide-code kernels: registers to spare): four staging loads and eight LDS rows in flight instead of one and two -- both loops
// of the narrow form wait out a trip to L2 / LDS per iteration
template <bool WIDE = false>
static __device__ __forceinline__ WSel<true> refine_probes_wg(const WSel<true> &ap, int cnt, int w, int q, const RefineArgs &r, int wv, int lane, int tid,
                                              float *stage, int stage_floats, int *s_cand)
{
    const float *qv = r.queries + (size_t)q * r.d;
    WSel<true> ex;
    ex.init(KEY_MAX, nullptr, 64, w);
    int c = 0, rank = 0;
    bool cand = false;
    if (wv == 0) {
        float part = 0.0f;
        for (int i = lane; i < r.d; i += 64) part += qv[i] * qv[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        const float qn = part;
        c = (int)(u32)ap.top;
        const float A = ordered_to_float((u32)(ap.top >> 32)) + qn;
        const float tau = __shfl(A, w - 1);
        const float sumn = r.cmaxn + sqrtf(qn) * 1.00001f;
        const float eps = r.eps_coef * sumn * sumn;
        float T = (tau + eps) * (1.0f + r.gam) + eps;
        T = T + fabsf(T) * 1e-6f;
        const float Alast = __shfl(A, cnt - 1);
        const bool flagged = r.qflags != nullptr && r.qflags[q] != 0u;
        const bool complete = !flagged && ((cnt >= r.kc) || (Alast > T));
        cand = complete && lane < cnt && A <= T;
        const u64 mask = __ballot(cand);
        rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (cand) s_cand[rank] = c;
        if (lane == 0) {
            s_cand[64] = complete ? 1 : 0;
            s_cand[65] = __popcll(mask);
        }
    }
    __syncthreads();
    const bool complete = s_cand[64] != 0;
    const int ncand = s_cand[65];
    if (!complete) {   // uniform: more near-ties than the pool absorbs -- wave 0 recomputes every distance exactly
        if (wv == 0) {
            if (lane == 0) atomicAdd(r.fallbacks, 1ull);
            for (int c0 = 0; c0 < r.kc; c0 += 64) {
                const int cc = c0 + lane;
                const bool ok = cc < r.kc;
                const float dist = ok ? exact_coarse_dist(r.centroids + (size_t)cc * r.d, qv, r.d) : 0.0f;
                const u64 key = make_key(dist, (u32)cc);
                ex.push(ok && key < ex.thr(), key, w, lane);
            }
        }
        return ex;
    }
    // chunk width: as much of d as fits ncand + 1 rows (row stride CH + 4 floats keeps 16-byte alignment)
    int CH = (stage_floats / (ncand + 1) - 4) & ~3;
    CH = CH > r.d ? r.d : CH;                     // r.d % 4 == 0 on this path
    if (CH < 4) CH = 4;                           // cannot happen with the table area (>= 2 KB) and <= 64 candidates
    const int LDW = CH + 4;
    float acc = 0.0f;
    for (int k0 = 0; k0 < r.d; k0 += CH) {
        const int wdt = min(CH, r.d - k0), w4 = wdt >> 2;
        if constexpr (WIDE) {
            const int tot = (ncand + 1) * w4;
            for (int e0 = tid; e0 < tot; e0 += 1024) {
                // four loads in flight; a slot beyond the end repeats the last element (its store is skipped)
                const int ea = e0, eb = min(e0 + 256, tot - 1), ec = min(e0 + 512, tot - 1), ed = min(e0 + 768, tot - 1);
                const int ra = ea / w4, rb = eb / w4, rc = ec / w4, rd = ed / w4;
                const int ca = ea - ra * w4, cb = eb - rb * w4, cc = ec - rc * w4, cd = ed - rd * w4;
                const float4 va = *(const float4 *)((ra < ncand ? r.centroids + (size_t)s_cand[ra] * r.d : qv) + k0 + ca * 4);
                const float4 vb = *(const float4 *)((rb < ncand ? r.centroids + (size_t)s_cand[rb] * r.d : qv) + k0 + cb * 4);
                const float4 vc = *(const float4 *)((rc < ncand ? r.centroids + (size_t)s_cand[rc] * r.d : qv) + k0 + cc * 4);
                const float4 vd = *(const float4 *)((rd < ncand ? r.centroids + (size_t)s_cand[rd] * r.d : qv) + k0 + cd * 4);
                *(float4 *)&stage[ra * LDW + ca * 4] = va;
                if (e0 + 256 < tot) *(float4 *)&stage[rb * LDW + cb * 4] = vb;
                if (e0 + 512 < tot) *(float4 *)&stage[rc * LDW + cc * 4] = vc;
                if (e0 + 768 < tot) *(float4 *)&stage[rd * LDW + cd * 4] = vd;
            }
        } else {
        for (int e = tid; e < (ncand + 1) * w4; e += 256) {
            const int row = e / w4, c4 = e - row * w4;
            const float *src = row < ncand ? r.centroids + (size_t)s_cand[row] * r.d : qv;
            *(float4 *)&stage[row * LDW + c4 * 4] = *(const float4 *)(src + k0 + c4 * 4);
        }
        }
        __syncthreads();
        if (cand) {
            const float *cr = stage + rank * LDW, *qr = stage + ncand * LDW;
#pragma unroll WIDE ? 8 : 2
            for (int i = 0; i < wdt; i += 4) {   // wdt % 4 == 0; rows are 16-byte aligned
                const float4 c4 = *(const float4 *)(cr + i);
                const float4 q4 = *(const float4 *)(qr + i);
                float t = c4.x - q4.x; acc = acc + t * t;
                t = c4.y - q4.y; acc = acc + t * t;
                t = c4.z - q4.z; acc = acc + t * t;
                t = c4.w - q4.w; acc = acc + t * t;
            }
        }
        __syncthreads();
    }
    if (wv == 0) ex.push(cand, make_key(acc, (u32)c), w, lane);
    return ex;
}

// Streams one row of kc floats through a wave selector; 64-candidate (or 256-candidate, 16-B loads) blocks are
// dealt round-robin to the WPQ waves of the query.  SCORE: values are signed MFMA scores, else distances >= 0.
// `shared` (LDS, may be null): smallest K-th key found by any of the WPQ waves so far -- adopted before every block
// and published after insertions, exactly as in scan_range (callers unshare() before merging the waves).
template <bool SCORE, int WPQ, class S>
static __device__ __forceinline__ void select_row(S &sel, const float *row, int kc, int K, int wv, int lane, u64 *shared = nullptr)
{
    auto obits = [](float f) { return SCORE ? ordered_bits(f) : __float_as_uint(f); };
    auto adopt = [&]() { if (shared) sel.tighten(readfirstlane64(*shared)); };
    auto publish = [&](u64 before) { if (shared && lane == 0 && sel.thr() < before) atomicMin(shared, sel.thr()); };
    if ((kc & 3) == 0) {
        const float4 *row4 = (const float4 *)row;
        const int nblk = (kc + 255) >> 8;
        for (int b0 = 0; b0 * WPQ < nblk; b0 += 4) {
            float4 dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c4 = ((b0 + u) * WPQ + (WPQ == 1 ? 0 : wv)) * 64 + lane;   // float4 index
                dv[u] = (c4 * 4 < kc) ? row4[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = (((b0 + u) * WPQ + (WPQ == 1 ? 0 : wv)) * 64 + lane) * 4;
                const float de[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
                adopt();
                const u32 th = (u32)(sel.thr() >> 32);
                bool anyc = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) anyc = anyc || (c + e < kc && obits(de[e]) <= th);
                if (__any(anyc)) {
                    const u64 before = sel.thr();
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const u64 key = ((u64)obits(de[e]) << 32) | (u32)(c + e);
                        sel.push(c + e < kc && key < sel.thr(), key, K, lane);
                    }
                    publish(before);
                }
            }
        }
    } else {
        for (int b0 = 0; b0 * 64 * WPQ < kc; b0 += 8) {
            float dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = ((b0 + u) * WPQ + (WPQ == 1 ? 0 : wv)) * 64 + lane;
                dv[u] = c < kc ? row[c] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = ((b0 + u) * WPQ + (WPQ == 1 ? 0 : wv)) * 64 + lane;
                adopt();
                const u64 before = sel.thr();
                const u64 key = ((u64)obits(dv[u]) << 32) | (u32)c;
                sel.push(c < kc && key < sel.thr(), key, K, lane);
                publish(before);
            }
        }
    }
}

constexpr int TILED_MAX = 128;

// Row selection through the tile minima of coarse_mfma_kernel (stand-alone top-w, one wave per query): the K-th
// smallest tile minimum bounds the K-th smallest score of the row from above (K different tiles hold a score <= it),
// so only tiles whose minimum is <= that bound are read -- K tiles plus ties, instead of the whole row (Deep1B-shape:
// 64 of 1024 tiles, 16 KB instead of 256 KB per query).  Same set, same order as select_row.
static __device__ __forceinline__ bool select_row_tiled(WSel<true> &sel, const float *row, const float *tm, int ntiles, int tile_w,
                                                        int kc, int K, int lane, int *tlist /* LDS, per wave, TILED_MAX ints */)
{
    WSel<true> ts;
    ts.init(KEY_MAX, nullptr, 64, K);
    for (int t0 = 0; t0 < ntiles; t0 += 64) {
        const int t = t0 + lane;
        const bool ok = t < ntiles;
        const u64 key = ok ? (((u64)ordered_bits(tm[t]) << 32) | (u32)t) : KEY_MAX;
        ts.push(ok && key < ts.thr(), key, K, lane);
    }
    const int tc = ts.finish(K, lane);
    const u32 bound = tc >= K ? (u32)(readlane64(ts.top, K - 1) >> 32) : 0xFFFFFFFFu;
    // the qualifying tiles, in order, into LDS; more than TILED_MAX of them (ties en masse): the caller streams the row
    int nt = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 64) {
        const int t = t0 + lane;
        const bool qual = t < ntiles && ordered_bits(tm[t]) <= bound;
        const u64 mask = __ballot(qual);
        const int pos = nt + __popcll(mask & ((1ull << lane) - 1ull));
        if (qual && pos < TILED_MAX) tlist[pos] = t;
        nt += __popcll(mask);
    }
    if (nt > TILED_MAX) return false;
    wave_sync();
    // every key of the pool has a score <= bound (K tiles hold one): start the selector under that bound, so only the
    // ~K..2K scores that can matter are ever inserted (instead of ~K(1 + ln(N/K)) insertions from a cold start)
    if (bound != 0xFFFFFFFFu) sel.tighten(((u64)bound + 1ull) << 32);
    // 64-lane blocks of scores, eight loads in flight at a time: the blocks sit in different tiles of the row, so every
    // one is a separate trip to L2 / HBM (one block per trip made the Deep1B-shape top-w wait ~1.5 k cycles 64 times)
    const int bpt = (tile_w + 63) >> 6;   // blocks per tile
    const int nblk = nt * bpt;
    constexpr int PF = 8;
    for (int b0 = 0; b0 < nblk; b0 += PF) {
        float v[PF];
        int c[PF];
        bool ok[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int b = b0 + u;
            ok[u] = false;
            c[u] = 0;
            v[u] = 0.0f;
            if (b < nblk) {
                const int tt = tlist[b / bpt];
                const int cend = min(kc, (tt + 1) * tile_w);
                c[u] = tt * tile_w + (b % bpt) * 64 + lane;
                ok[u] = c[u] < cend;
                if (ok[u]) v[u] = row[c[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const u64 key = ok[u] ? (((u64)ordered_bits(v[u]) << 32) | (u32)c[u]) : KEY_MAX;
            sel.push(ok[u] && key < sel.thr(), key, K, lane);
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// Short rows (kc <= 2048, kc % 4 == 0) and few probes (K <= SHORT_ROW_MAXK), all 256 threads of a workgroup on
// ONE row: selection by a sampled bound instead of serial insertions.
//   1. every lane loads its <= 8 values and takes the minimum key;
//   2. each wave sorts its 64 lane minima: the K-th of them bounds the K-th key of the row from above (K different
//      lanes hold a key <= it); T = the smallest of the four waves' bounds;
//   3. keys <= T are compacted into LDS (expected ~4K of them, all of the row's K smallest among them);
//   4. wave 0 sorts that handful: `ws` holds the K smallest keys of the row, exactly as select_row + merge_waves
//      would have left them (keys are unique, so the selection is the same set in the same order).
// Returns false (uniformly, nothing selected) when more than SHORT_ROW_CAND keys pass the bound; the caller then
// takes the streaming path.  Three sorts on the critical path instead of ~1 + 3K insertions per wave.
// ---------------------------------------------------------------------------------------
constexpr int SHORT_ROW_MAXK = 24;
constexpr int SHORT_ROW_CAND = 256;

// NU: 16-byte groups per lane (rows of up to 1024 NU keys; 2 everywhere but in the wide-code kernels, which have the registers)
template <bool SCORE, int NU = 2>
static __device__ __forceinline__ bool select_row_short(WSel<true> &ws, const float *row, int kc, int K, int wv, int lane, int tid,
                                                        u64 *cand /*LDS [SHORT_ROW_CAND]*/, u64 *wbound /*LDS [4]*/,
                                                        u32 *ccnt /*LDS*/)
{
    const float4 *row4 = (const float4 *)row;
    float4 dv[NU];
    int cbase[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int c4 = (u * 4 + wv) * 64 + lane;
        cbase[u] = c4 * 4;
        dv[u] = (cbase[u] < kc) ? row4[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    u64 keys[4 * NU];
    u64 lmin = KEY_MAX;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const float de[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const u32 bits = SCORE ? ordered_bits(de[e]) : __float_as_uint(de[e]);
            const u64 key = (cbase[u] < kc) ? (((u64)bits << 32) | (u32)(cbase[u] + e)) : KEY_MAX;   // kc % 4 == 0
            keys[u * 4 + e] = key;
            lmin = key < lmin ? key : lmin;
        }
    }
    const u64 sorted = wave_sort64(lmin, lane);
    const u64 bound = readlane64(sorted, K - 1);   // KEY_MAX when fewer than K lanes of this wave hold data
    if (lane == 0) wbound[wv] = bound;
    if (tid == 0) *ccnt = 0u;
    __syncthreads();
    u64 T = wbound[0];
#pragma unroll
    for (int v = 1; v < 4; ++v) T = wbound[v] < T ? wbound[v] : T;
    T = readfirstlane64(T);
#pragma unroll
    for (int e = 0; e < 4 * NU; ++e) {
        const bool pred = keys[e] != KEY_MAX && keys[e] <= T;
        const u64 mask = __ballot(pred);
        if (mask) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(ccnt, (u32)__popcll(mask));
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
            const u32 pos = base + (u32)__popcll(mask & ((1ull << lane) - 1ull));
            if (pred && pos < (u32)SHORT_ROW_CAND) cand[pos] = keys[e];
        }
    }
    __syncthreads();
    const int C = (int)*ccnt;
    if (C > SHORT_ROW_CAND) return false;
    if (wv == 0) {
        ws.init(KEY_MAX, nullptr, 64, K);
        sel_absorb(ws, cand, C, K, lane);
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// top-w of the kc coarse distances of one query (one wave per query): sortperm(dist)[1:w]
// is stable, so ties go to the lower cluster index = the low word of the key.  Also emits
// the visit-order base of each probe, the per-list probe histogram (list-major plan only)
// and the B_alg counter.
// ---------------------------------------------------------------------------------------
// WPQ = waves per query: 1 (one wave per query, 4 queries per workgroup; large batches) or 4 (the four
// waves of a workgroup each select over a quarter of the row, wave 0 merges: four times the waves in
// flight when the batch alone cannot fill the chip).
template <bool SMALL, int WPQ, bool APPROX>
__global__ __launch_bounds__(256) void topw_select_kernel(const float *__restrict__ cdist, int nq, int kc, int w, int cap,
                                                          const u32 *__restrict__ list_len, int *__restrict__ probe_list,
                                                          float *__restrict__ probe_dc, u32 *__restrict__ probe_base,
                                                          u32 *__restrict__ list_cnt, u64 *__restrict__ scanned_points,
                                                          const RefineArgs rf, int nparts, int part)
{
    // nparts > 1 (list-partitioned multi-GPU mode): this rank scans the probed lists l with l % nparts == part only -- every probe
    // keeps its place in the probe arrays (visit-order bases are global: keys of different ranks compare), but only this rank's lists
    // enter the probe histogram and the B_alg count
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *sbuf = (u64 *)smem_raw;   // [4][cap]: selector buffer (!SMALL) / staging of each wave's sorted keys
    __shared__ int s_cnt[4];
    __shared__ u64 s_thr;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = (WPQ == 1) ? blockIdx.x * 4 + wv : blockIdx.x;
    if (WPQ == 1 && q >= nq) return;   // WPQ == 1 uses no workgroup barrier
    if (WPQ == 4) {
        if (threadIdx.x == 0) s_thr = KEY_MAX;
        __syncthreads();
    }
    u64 *buf = sbuf + (size_t)wv * cap;
    // APPROX: cdist holds MFMA scores; keep the 64 best, then refine_probes() turns them into the exact top-w
    const int Ksel = APPROX ? approx_pool(w) : w;
    WSel<SMALL> sel;
    sel.init(KEY_MAX, buf, cap, Ksel);
    const float *row = cdist + (size_t)q * kc;
    bool tiled = false, listed = false;
    int cnt = 0;
    if constexpr (APPROX && SMALL && WPQ == 1) {
        if (rf.tlist) {   // listed mode: there is no score matrix; the records give the exact top-w directly
            listed = true;
            const WSel<true> ex = select_listed(q, w, rf, lane, (int *)buf);
            cnt = ex.finish(w, lane);
            wave_sync();
            ex.store(buf, cnt, lane);
        } else if (rf.tmin)   // the tile list lives in the wave's staging area (cap >= 64 keys = 128 ints), which is idle until store()
            tiled = select_row_tiled(sel, row, rf.tmin + (size_t)q * rf.ntiles, rf.ntiles, rf.tile_w, kc, Ksel, lane, (int *)buf);
    }
    if (!listed) {
    bool shortrow = false;   // uniform over the workgroup
    if constexpr (WPQ == 4 && SMALL) {
        // small batches on a short row: the workgroup-wide bound-and-compact selection of the query-major prologue
        // (three sorts instead of a streaming selection per wave and a four-way merge: the latency path of a single query)
        if (Ksel <= SHORT_ROW_MAXK && kc <= 2048 && (kc & 3) == 0) {
            __shared__ u64 s_wb[4];
            __shared__ u32 s_cc;
            shortrow = select_row_short<APPROX>(sel, row, kc, Ksel, wv, lane, (int)threadIdx.x, sbuf, s_wb, &s_cc);
            if (shortrow && wv != 0) return;
        }
    }
    if (!tiled && !shortrow) select_row<APPROX, WPQ>(sel, row, kc, Ksel, wv, lane, WPQ == 4 ? &s_thr : (u64 *)nullptr);
    cnt = sel.finish(Ksel, lane);
    if (shortrow) wave_sync();          // wave 0 alone from here on: the candidate area (sbuf) is about to be reused
    sel.store(buf, cnt, lane);
    if (WPQ == 4 && !shortrow) {
        if (lane == 0) s_cnt[wv] = cnt;
        __syncthreads();
        if (wv != 0) return;
        merge_waves(sel, sbuf, (size_t)cap, s_cnt, 1, Ksel, KEY_MAX, 0, lane);
        cnt = sel.finish(Ksel, lane);   // == min(Ksel, kc)
        sel.store(buf, cnt, lane);
    }
    if constexpr (APPROX && SMALL) {
        const WSel<true> ex = refine_probes(sel, cnt, w, q, rf, lane);
        cnt = ex.finish(w, lane);
        wave_sync();
        ex.store(buf, cnt, lane);
    }
    }   // !listed
    wave_sync();
    u32 running = 0, mine_total = 0;
    for (int j0 = 0; j0 < cnt; j0 += 64) {
        const int j = j0 + lane;
        u32 len = 0;
        int l = 0;
        float dd = 0.0f;
        if (j < cnt) {
            const u64 key = buf[j];
            l = (int)(u32)key;
            dd = __uint_as_float((u32)(key >> 32));
            len = list_len[l];
        }
        const bool mine = nparts <= 1 || (l % nparts) == part;
        u32 incl = len, minc = mine ? len : 0u;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) minc += __shfl_xor(minc, off);
        if (j < cnt) {
            const size_t o = (size_t)q * w + j;
            probe_list[o] = l;
            probe_dc[o] = dd;
            probe_base[o] = running + incl - len;
            if (list_cnt && mine) atomicAdd(&list_cnt[l], 1u);
        }
        running += __shfl(incl, 63);
        mine_total += minc;
    }
    // B_alg statistics: 64 counters, one 64-B line each (a single word would serialise every query's atomic)
    if (lane == 0) atomicAdd(scanned_points + (size_t)(q & 63) * 8, (u64)mine_total);
}

// ---------------------------------------------------------------------------------------
// Group the (query, probe) pairs by inverted list: exclusive scans of the probe histogram
// (bucket offsets) and of the work items per list (ceil(cnt/QG) query groups x chunks).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bucket_scan_kernel(const u32 *__restrict__ list_cnt, const u32 *__restrict__ list_len,
                                                           int kc, int QG, u32 CH, u32 *__restrict__ bucket_off,
                                                           u32 *__restrict__ wi_off, u32 *__restrict__ cursor,
                                                           u32 *__restrict__ queue_head, u32 *__restrict__ item_list = nullptr)
{
    __shared__ u32 sa[1024], sb[1024];
    const int tid = threadIdx.x;
    const int per = (kc + 1023) / 1024;
    const int l0 = tid * per, l1 = min(kc, l0 + per);
    u32 suma = 0, sumb = 0;
    for (int l = l0; l < l1; ++l) {
        const u32 cnt = list_cnt[l];
        const u32 len = list_len[l];
        const u32 ng = (cnt + QG - 1) / QG;
        const u32 nch = (len + CH - 1) / CH;
        suma += cnt;
        sumb += ng * nch;
    }
    sa[tid] = suma;
    sb[tid] = sumb;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        u32 va = 0, vb = 0;
        if (tid >= off) { va = sa[tid - off]; vb = sb[tid - off]; }
        __syncthreads();
        sa[tid] += va;
        sb[tid] += vb;
        __syncthreads();
    }
    u32 runa = sa[tid] - suma, runb = sb[tid] - sumb;
    for (int l = l0; l < l1; ++l) {
        const u32 cnt = list_cnt[l];
        const u32 len = list_len[l];
        const u32 ng = (cnt + QG - 1) / QG;
        const u32 nch = (len + CH - 1) / CH;
        bucket_off[l] = runa;
        wi_off[l] = runb;
        cursor[l] = 0;
        if (item_list)      // (the eight-wave kernel: work item -> list, instead of a binary search over wi_off per item)
            for (u32 i = 0; i < ng * nch; ++i) item_list[runb + i] = (u32)l;
        runa += cnt;
        runb += ng * nch;
    }
    if (tid == 1023) { bucket_off[kc] = sa[1023]; wi_off[kc] = sb[1023]; queue_head[0] = 0; }
}

__global__ __launch_bounds__(256) void bucket_scatter_kernel(const int *__restrict__ probe_list, int nprobe,
                                                             const u32 *__restrict__ bucket_off, u32 *__restrict__ cursor,
                                                             u32 *__restrict__ bucket_items, int nparts, int part)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= nprobe) return;
    const int l = probe_list[p];
    if (nparts > 1 && (l % nparts) != part) return;   // another rank's list (list-partitioned mode)
    const u32 pos = bucket_off[l] + atomicAdd(&cursor[l], 1u);
    bucket_items[pos] = (u32)p;
}

// ---------------------------------------------------------------------------------------
// HOT-2 + HOT-3 building blocks shared by the two scan kernels.
//
// Tables live in LDS interleaved [sub-quantizer][code][QG] so ONE ds_read_b128 (QG=4) returns
// the entries of all queries of the group for a code byte.  Each lane owns whole points
// (16-byte code loads), adds the m entries in ascending sub-quantizer order onto the coarse
// distance (index.jl:242-246), and candidates below the running threshold go to the selector.
// ---------------------------------------------------------------------------------------
struct IndexView {
    const float *centroids;      // [kc][d]
    const float *codebooks;      // [m][ksub][dsub]
    // the same codewords regrouped for the table build: [m][ceil(dsub / 4)][ksub][4] (zero-padded to a multiple of 4),
    // so the 64 lanes of a wave (one codeword each) read consecutive 16-byte groups: 1 KB per load instruction
    // instead of 16 B out of every 64 B of a 4 KB window.  dsub = 6 with an even m: pair-packed, [m / 2][3][ksub][4] --
    // codeword c of sub-quantizers 2p and 2p + 1 back to back, 12 floats in three groups, no padding
    const float *codebooks_t;
    // pair-interleaved copy for the packed-FP32 table build (build_tables_pk; m even, dsub even; null when absent):
    // [m / 2][dsub / 2][ksub][4] = (A[2g], B[2g], A[2g+1], B[2g+1]) with A / B the codewords c of sub-quantizers 2p / 2p + 1
    const float *codebooks_p;
    const uint8_t *labels;       // [m][ksub]
    const uint8_t *codes;        // device layout: list l at codes + list_codeoff[l], stride cs per point
    const int64_t *list_pos;     // [kc] offset of each list in the id array (lists have spare capacity behind them)
    const u32 *list_len;         // [kc] points in each list
    const int64_t *list_codeoff; // [kc]
    const u32 *ids;              // [n] or null (id == position)
    int d, kc, m, ksub, dsub, cs;
    int identity_labels;         // labels[i][c] == c for every block: skip the label fetch
    int dbg_flags;               // diagnostics only (IVFADC_DEBUG_FLAGS): 1 = drop candidates, 2 = skip lookups
};

template <int QG> struct TabV;
template <> struct TabV<1> {
    static __device__ __forceinline__ void ld(const float *t, float *o) { o[0] = t[0]; }
};
template <> struct TabV<2> {
    static __device__ __forceinline__ void ld(const float *t, float *o)
    {
        const float2 v = *(const float2 *)t;
        o[0] = v.x; o[1] = v.y;
    }
};
template <> struct TabV<4> {
    static __device__ __forceinline__ void ld(const float *t, float *o)
    {
        const float4 v = *(const float4 *)t;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};

// residuals r_s = q_s - c_{l_s} (coarsequantizers.jl:40-45) -> LDS resid[i][s]; all 256 threads; caller barriers after.
// List-major groups pass QG queries and one list; query-major rounds pass one query and QG lists.
template <int QG>
static __device__ __forceinline__ void build_residuals(const IndexView &ix, const float *queries, const int (&qi)[QG],
                                                       const int (&li)[QG], float *resid, int tid)
{
    for (int e = tid; e < ix.d * QG; e += 256) {
        const int i = e / QG, s = e - i * QG;
        int qs = qi[0], ls = li[0];
#pragma unroll
        for (int t = 1; t < QG; ++t)
            if (s == t) { qs = qi[t]; ls = li[t]; }
        resid[e] = queries[(size_t)qs * ix.d + i] - ix.centroids[(size_t)ls * ix.d + i];
    }
}

// ADC tables (index.jl:232-236): sum_t (CB_ii[t,c] - r_s[ii*dsub+t])^2 for the QG residuals at once, so every
// codeword fetched from L2 is used QG times.  SEP = false: tab[ii][label][s] (one ds_read_b128 serves the QG
// queries of a list-major group); SEP = true: tab[s][ii][label] (QG independent tables, query-major rounds).
// DSUB > 0 fixes the sub-space width at compile time so all loads of a codeword are issued before its first use.
// Codeword of (ii, c) from codebooks_t through a buffer resource: the descriptor and the uniform part of the address
// (soff = float index of group 0 of sub-quantizer ii) live in SGPRs, the lane contributes ONE 32-bit offset
// (lane_off = c * V * 4 bytes) for every load of the whole table build.  Group g sits g * ksub * V floats further on.
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef u32 v2u __attribute__((ext_vector_type(2)));
template <int DSUB>
static __device__ __forceinline__ void load_codeword(__amdgpu_buffer_rsrc_t rs, u32 soff, u32 lane_off, int ksub, float (&cv)[DSUB > 0 ? DSUB : 1])
{
    // 16-byte groups only: a sub-space that is not a multiple of 4 wide is zero-padded in codebooks_t (dsub = 6: two
    // 16-byte loads instead of three 8-byte ones -- the texture addresser works per lane and instruction, not per byte)
#pragma unroll
    for (int t = 0; t < DSUB; t += 4) {
        const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off, (int)((soff + (u32)(t >> 2) * ksub * 4) * 4u), 0);
        const float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t + e < DSUB) cv[t + e] = f[e];
    }
}

// LAYOUT 0: tab[ii][label][s] (list-major groups, one ds_read of 4 QG bytes per code byte); 1 (SEP): tab[s][ii][label]
// (QG independent tables, query-major rounds); 2 (STRIPED, see striped_scan_step): entry (ii, label) at byte
// (label << stripe_shift<M, QG>()) | (ii << log2(4 QG)) -- sub-quantizer ii owns its own bank stripe.
constexpr int TAB_INTERLEAVED = 0, TAB_SEP = 1, TAB_STRIPED = 2;
template <int M, int QG> static constexpr int stripe_shift()
{
    // bytes per bank row of the ds_read form: 256 (b64 / b128: 64 banks) or 128 (b32: 32 banks); a code's row holds the
    // entries of all M sub-quantizers when they fit, else M * entry bytes / row rows' worth of codes share a row
    return (QG == 4) ? (M == 8 ? 7 : 8) : (QG == 2 ? (M == 8 ? 6 : 7) : (M == 8 ? 5 : 6));
}
// ONLY0 (QG = 2, SEP, compile-time dsub): only the table of residual 0 is wanted (the round's second probe is pruned, deferred or past the
// end): the second residual's arithmetic -- half of the build -- is not executed.  A template parameter, chosen by a uniform branch at the
// call site: the same test inside the codeword stages cost the Deep1B shape 5 % (profiles/r05_first_round_ab.txt).
template <int QG, int DSUB, int LAYOUT, int MS = 0, bool ONLY0 = false>
static __device__ __forceinline__ void build_tables_t(const IndexView &ix, int m, const float *resid, float *tab, int tid)
{
    static_assert(!ONLY0 || (QG == 2 && LAYOUT == TAB_SEP && DSUB > 0), "ONLY0: query-major rounds of two probes");
    constexpr bool SEP = LAYOUT == TAB_SEP;
    // thread = codeword c of every sub-quantizer in turn (m iterations; 256 threads cover the 256 codes)
    const int c = tid;
    if (c >= ix.ksub) return;
    if constexpr (DSUB > 0) {
        // software pipeline without register copies: two codeword stages alternate, the codewords of the next
        // stage are in flight while this one is accumulated
        // padded sub-space width of codebooks_t; dsub = 6 is stored pair-packed instead (codeword_pairs): 6 floats, no padding
        constexpr int DP = DSUB == 6 ? 6 : (DSUB + 3) & ~3;
        const __amdgpu_buffer_rsrc_t cw =
            __builtin_amdgcn_make_buffer_rsrc((void *)ix.codebooks_t, 0, (int)((u32)m * ix.ksub * DP * 4u), 0x00020000);
        const u32 loff = (u32)c * 16u;
        const u32 cstep = (u32)ix.ksub * DP;
        auto accumulate = [&](const float (&cv)[DSUB], int ii) {
            const float *rr = resid + (size_t)ii * DSUB * QG;
            if constexpr (ONLY0) {
                {
                    float sum0 = 0.0f;
#pragma unroll
                    for (int t = 0; t < DSUB; ++t) {
                        const float df = cv[t] - rr[(size_t)t * QG];
                        sum0 = sum0 + df * df;
                    }
                    const int label = ix.identity_labels ? c : (int)ix.labels[ii * ix.ksub + c];
                    tab[(size_t)ii * 256 + label] = sum0;
                    return;
                }
            }
            float sum[QG];
            // scalar on purpose: v_pk_add_f32 / v_pk_mul_f32 are not double-rate on gfx950 (measured: the packed
            // form of this loop took 47k cycles per workgroup against 38k for the scalar one)
#pragma unroll
            for (int s = 0; s < QG; ++s) sum[s] = 0.0f;
#pragma unroll
            for (int t = 0; t < DSUB; ++t) {
                float rv[QG];
                TabV<QG>::ld(rr + (size_t)t * QG, rv);
#pragma unroll
                for (int s = 0; s < QG; ++s) {
                    const float df = cv[t] - rv[s];
                    sum[s] = sum[s] + df * df;
                }
            }
            const int label = ix.identity_labels ? c : (int)ix.labels[ii * ix.ksub + c];
            if constexpr (SEP) {
#pragma unroll
                for (int s = 0; s < QG; ++s) tab[((size_t)s * m + ii) * 256 + label] = sum[s];
            } else if constexpr (LAYOUT == TAB_STRIPED) {
                float *dst = tab + ((((u32)label << stripe_shift<MS, QG>()) | ((u32)ii * 4u * QG)) >> 2);
                if constexpr (QG == 4) *(v4f *)dst = (v4f){sum[0], sum[1], sum[2], sum[3]};
                else {
#pragma unroll
                    for (int s = 0; s < QG; ++s) dst[s] = sum[s];
                }
            } else {
                float *dst = tab + ((size_t)ii * 256 + label) * QG;
#pragma unroll
                for (int s = 0; s < QG; ++s) dst[s] = sum[s];
            }
        };
        // G codewords per stage: a narrow sub-space (dsub = 6: 24 bytes) gives one stage too little in flight and too
        // little arithmetic to cover the next stage's trip to L2 -- measured on the Deep1B shape, the build waited
        // on loads for two thirds of its time with one codeword per stage
        constexpr int G = DSUB >= 16 ? 1 : (DSUB >= 8 ? 2 : 4);
        float ca[G][DSUB], cb[G][DSUB];
        auto load_stage = [&](float (&buf)[G][DSUB], int ii0) {
            if constexpr (DSUB == 6) {
                // codeword c of sub-quantizers 2p and 2p + 1 as THREE 16-byte groups (12 floats) instead of two padded
                // 16-byte groups each: the texture addresser works per lane and instruction, and it is what binds this
                // build on the Deep1B shape -- a quarter fewer instructions.  m is even (host check).
#pragma unroll
                for (int g = 0; g < G; g += 2)
                    if (ii0 + g < m) {
                        const u32 soff = (u32)(ii0 + g) * cstep;   // pair p = (ii0 + g) / 2 starts at float p * ksub * 12
                        v4u v[3];
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            v[k] = __builtin_amdgcn_raw_buffer_load_b128(cw, (int)loff, (int)((soff + (u32)k * ix.ksub * 4) * 4u), 0);
                        buf[g][0] = __uint_as_float(v[0].x); buf[g][1] = __uint_as_float(v[0].y); buf[g][2] = __uint_as_float(v[0].z);
                        buf[g][3] = __uint_as_float(v[0].w); buf[g][4] = __uint_as_float(v[1].x); buf[g][5] = __uint_as_float(v[1].y);
                        buf[g + 1][0] = __uint_as_float(v[1].z); buf[g + 1][1] = __uint_as_float(v[1].w); buf[g + 1][2] = __uint_as_float(v[2].x);
                        buf[g + 1][3] = __uint_as_float(v[2].y); buf[g + 1][4] = __uint_as_float(v[2].z); buf[g + 1][5] = __uint_as_float(v[2].w);
                    }
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (ii0 + g < m) load_codeword<DSUB>(cw, (u32)(ii0 + g) * cstep, loff, ix.ksub, buf[g]);
            }
        };
        auto run_stage = [&](const float (&buf)[G][DSUB], int ii0) {
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (ii0 + g < m) accumulate(buf[g], ii0 + g);
        };
        load_stage(ca, 0);
#pragma unroll 1
        for (int ii = 0; ii < m; ii += 2 * G) {
            if (ii + G < m) load_stage(cb, ii + G);
            run_stage(ca, ii);
            if (ii + G < m) {
                if (ii + 2 * G < m) load_stage(ca, ii + 2 * G);
                run_stage(cb, ii + G);
            }
        }
    } else {
        const int dsub = ix.dsub;
        for (int ii = 0; ii < m; ++ii) {
            const float *cw = ix.codebooks + ((size_t)ii * ix.ksub + c) * dsub;
            const float *rr = resid + (size_t)ii * dsub * QG;
            float sum[QG];
#pragma unroll
            for (int s = 0; s < QG; ++s) sum[s] = 0.0f;
#pragma unroll 2
            for (int t = 0; t < dsub; ++t) {
                const float cvt = cw[t];
                float rv[QG];
                TabV<QG>::ld(rr + (size_t)t * QG, rv);
#pragma unroll
                for (int s = 0; s < QG; ++s) {
                    const float df = cvt - rv[s];
                    sum[s] = sum[s] + df * df;
                }
            }
            const int label = ix.identity_labels ? c : (int)ix.labels[ii * ix.ksub + c];
            if constexpr (SEP) {
#pragma unroll
                for (int s = 0; s < QG; ++s) tab[((size_t)s * m + ii) * 256 + label] = sum[s];
            } else {
                float *dst = tab + ((size_t)ii * 256 + label) * QG;
#pragma unroll
                for (int s = 0; s < QG; ++s) dst[s] = sum[s];
            }
        }
    }
}

// The same build with NBUF rotating codeword buffers (one codeword each; DSUB % 4 == 0, padded layout): a buffer is
// refilled right after its use, so NBUF - 1 stages of arithmetic cover a load's trip to L2.  For kernels with registers to
// spare (m = 48).  Worth 2 % on the HD shape (scan 1.41 -> 1.38 ms; three, four and six buffers within 0.5 % of each
// other): that build is bound by VALU issue (VALUBusy 64 %, four cycles per instruction at three waves per SIMD) and
// the L1 fill rate (TA busy 60 %), not by latency.  SEP layout: tab[s][ii][label].
template <int QG, int DSUB, int NBUF>
static __device__ __forceinline__ void build_tables_deep(const IndexView &ix, int m, const float *resid, float *tab, int tid)
{
    static_assert(DSUB > 0 && DSUB % 4 == 0, "padded codebooks_t layout");
    const int c = tid;
    if (c >= ix.ksub) return;
    const __amdgpu_buffer_rsrc_t cw =
        __builtin_amdgcn_make_buffer_rsrc((void *)ix.codebooks_t, 0, (int)((u32)m * ix.ksub * DSUB * 4u), 0x00020000);
    const u32 loff = (u32)c * 16u;
    const u32 cstep = (u32)ix.ksub * DSUB;
    float buf[NBUF][DSUB];
#pragma unroll
    for (int u = 0; u < NBUF; ++u)
        if (u < m) load_codeword<DSUB>(cw, (u32)u * cstep, loff, ix.ksub, buf[u]);
#pragma unroll 1
    for (int ii = 0; ii < m; ii += NBUF) {
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            const int i0 = ii + u;
            if (i0 < m) {   // uniform
                const float *rr = resid + (size_t)i0 * DSUB * QG;
                float sum[QG];
#pragma unroll
                for (int s = 0; s < QG; ++s) sum[s] = 0.0f;
#pragma unroll
                for (int t = 0; t < DSUB; ++t) {
                    float rv[QG];
                    TabV<QG>::ld(rr + (size_t)t * QG, rv);
#pragma unroll
                    for (int s = 0; s < QG; ++s) {
                        const float df = buf[u][t] - rv[s];
                        sum[s] = sum[s] + df * df;
                    }
                }
                const int label = ix.identity_labels ? c : (int)ix.labels[i0 * ix.ksub + c];
#pragma unroll
                for (int s = 0; s < QG; ++s) tab[((size_t)s * m + i0) * 256 + label] = sum[s];
                if (i0 + NBUF < m) load_codeword<DSUB>(cw, (u32)(i0 + NBUF) * cstep, loff, ix.ksub, buf[u]);
            }
        }
    }
}

// One residual, TWO sub-quantizers per thread and step in packed FP32 (v_pk_add_f32 / v_pk_mul_f32 on (A, B) pairs): the sums of
// entry A and entry B stay sequential in t, each element is rounded exactly like the scalar op (no contraction), so the tables
// are bit-identical -- and the build issues half the vector instructions.  Packed FP32 runs at the scalar ELEMENT rate when the
// SIMD is full (four waves: measured no gain on the m = 8 kernel), but a kernel that LDS holds at three waves per SIMD is
// bound by how often a wave can issue, not by the pipe: there an instruction that carries two elements is worth two.
// resid2: pair-interleaved residuals [(p * DSUB + t) * 2 + h] = r[(2p + h) * DSUB + t].  Two rotating buffers of one pair each.
template <int DSUB>
static __device__ __forceinline__ void build_tables_pk(const IndexView &ix, int m, const float *resid2, float *tab, int tid)
{
    static_assert(DSUB > 0 && DSUB % 2 == 0, "dimension pairs");
    const int c = tid;
    if (c >= ix.ksub) return;
    constexpr int NG = DSUB / 2;   // 16-byte groups per codeword pair
    const __amdgpu_buffer_rsrc_t cw =
        __builtin_amdgcn_make_buffer_rsrc((void *)ix.codebooks_p, 0, (int)((u32)m * ix.ksub * DSUB * 4u), 0x00020000);
    const u32 loff = (u32)c * 16u;
    const u32 pstep = (u32)ix.ksub * DSUB * 2u;   // floats per pair
    v4f buf[2][NG];
    auto load_pair = [&](v4f (&b)[NG], int p) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(cw, (int)loff, (int)(((u32)p * pstep + (u32)g * ix.ksub * 4u) * 4u), 0);
            b[g] = (v4f){__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        }
    };
    auto run_pair = [&](const v4f (&b)[NG], int p) {
        const v2f *rr = (const v2f *)(resid2 + (size_t)p * DSUB * 2);
        v2f sum = (v2f){0.0f, 0.0f};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const v2f r0 = rr[2 * g], r1 = rr[2 * g + 1];
            const v2f d0 = (v2f){b[g].x, b[g].y} - r0;
            sum = sum + d0 * d0;
            const v2f d1 = (v2f){b[g].z, b[g].w} - r1;
            sum = sum + d1 * d1;
        }
        const int la = ix.identity_labels ? c : (int)ix.labels[(2 * p) * ix.ksub + c];
        const int lb = ix.identity_labels ? c : (int)ix.labels[(2 * p + 1) * ix.ksub + c];
        tab[(size_t)(2 * p) * 256 + la] = sum.x;
        tab[(size_t)(2 * p + 1) * 256 + lb] = sum.y;
    };
    const int np = m >> 1;
    load_pair(buf[0], 0);
    if (np > 1) load_pair(buf[1], 1);
#pragma unroll 1
    for (int p = 0; p < np; p += 2) {
        run_pair(buf[0], p);
        if (p + 2 < np) load_pair(buf[0], p + 2);
        if (p + 1 < np) {
            run_pair(buf[1], p + 1);
            if (p + 3 < np) load_pair(buf[1], p + 3);
        }
    }
}

template <int QG, class S>
static __device__ __forceinline__ void scan_emit(const float (&acc)[QG], u32 p, bool valid, int nvalid, const u32 (&sbase)[QG],
                                                 S (&sel)[QG], int K, int lane)
{
#pragma unroll
    for (int s = 0; s < QG; ++s) {
        const u64 key = make_key(acc[s], sbase[s] + p);
        const bool pred = valid && (s < nvalid) && key < sel[s].thr();
        sel[s].push(pred, key, K, lane);
    }
}

// One wave-step worth of code bytes per lane, kept in registers so the next block (or the first block
// of the next list) is in flight while tables are built / the current block is scored.
// points per lane per step: 4 only where registers allow it (m = 8 with at most two queries per code stream);
// measured: m = 16 with 4 costs 40+ VGPRs and is slower everywhere; m = 8, QG = 4 with 2 raises the occupancy
// from 2 to 3 waves/SIMD (SIFT1B-shape scan 14.3 -> 12.1 ms)
template <int M, int QG> static constexpr int ppl_of() { return (M == 8 && QG <= 2) ? 4 : 2; }

template <int M, int P> struct CodeRegs {
    static constexpr int PPL = P;                                      // points per lane per step
    static constexpr int NV = (M == 8) ? P / 2 : P * (M / 16);         // uint4 registers
    static constexpr int STEP = 64 * PPL;                              // points per wave per step
    uint4 v[NV];
    __device__ __forceinline__ void load(const uint8_t *cbase, u32 pb, int lane)
    {
        if constexpr (M == 8) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = *(const uint4 *)(cbase + (size_t)(pb + k * 128 + lane * 2) * 8);
        } else {
#pragma unroll
            for (int r = 0; r < PPL; ++r)
#pragma unroll
                for (int k = 0; k < M / 16; ++k)
                    v[r * (M / 16) + k] = *(const uint4 *)(cbase + (size_t)(pb + r * 64 + lane) * M + 16 * k);
        }
    }
    // position (relative to the list) of register point r
    static __device__ __forceinline__ u32 point(u32 pb, int r, int lane)
    {
        if constexpr (M == 8) return pb + (r >> 1) * 128 + lane * 2 + (r & 1);
        else return pb + r * 64 + lane;
    }
    // the M / 4 code dwords of register point r
    __device__ __forceinline__ void words(int r, u32 (&out)[M / 4]) const
    {
        if constexpr (M == 8) {
            const uint4 q4 = v[r >> 1];
            out[0] = (r & 1) ? q4.z : q4.x;
            out[1] = (r & 1) ? q4.w : q4.y;
        } else {
#pragma unroll
            for (int k = 0; k < M / 16; ++k) {
                const uint4 q4 = v[r * (M / 16) + k];
                out[4 * k + 0] = q4.x; out[4 * k + 1] = q4.y; out[4 * k + 2] = q4.z; out[4 * k + 3] = q4.w;
            }
        }
    }
    // code byte ii of register point r
    __device__ __forceinline__ u32 byte(int r, int ii) const
    {
        const int bi = (M == 8) ? (r & 1) * 8 + ii : ii;              // byte index inside the point's uint4 group
        const uint4 q4 = (M == 8) ? v[r >> 1] : v[r * (M / 16) + (bi >> 4)];
        const int wsel = (bi >> 2) & 3;
        const u32 dw = wsel == 0 ? q4.x : wsel == 1 ? q4.y : wsel == 2 ? q4.z : q4.w;
        return (dw >> (8 * (bi & 3))) & 0xffu;
    }
    // (code byte ii of register point r) << SH in ONE VALU instruction: SDWA selects the byte of the shifted operand
    // (the compiler's own extract + shift-add costs two; the scan issues one of these per table lookup)
    template <int SH> __device__ __forceinline__ u32 byte_shl(int r, int ii) const
    {
        const int bi = (M == 8) ? (r & 1) * 8 + ii : ii;
        const uint4 q4 = (M == 8) ? v[r >> 1] : v[r * (M / 16) + (bi >> 4)];
        const int wsel = (bi >> 2) & 3;
        const u32 dw = wsel == 0 ? q4.x : wsel == 1 ? q4.y : wsel == 2 ? q4.z : q4.w;
        const u32 sh = SH;
        u32 o;
        switch (bi & 3) {   // constant after unrolling
        case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(o) : "s"(sh), "v"(dw)); break;
        case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(o) : "s"(sh), "v"(dw)); break;
        case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(o) : "s"(sh), "v"(dw)); break;
        default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(o) : "s"(sh), "v"(dw)); break;
        }
        return o;
    }
};
// LDS load at an absolute byte address.  The scan kernels own the whole LDS allocation (no static __shared__; the host
// checks hipFuncGetAttributes().sharedSizeBytes == 0), so the dynamic segment starts at address 0 and a table lookup is
// `ds_read vdst, v(code << sh) offset:(table offset)` with nothing added per lookup.
template <class T> static __device__ __forceinline__ T lds_load_abs(u32 byte_addr)
{
    return *(const __attribute__((address_space(3))) T *)(size_t)byte_addr;
}
// byte offset of table entry [ii][code][0..QG) from the table base: ii * 256 * QG * 4 + (code << log2(4 QG))
template <int QG> static constexpr int entry_shift() { return QG == 1 ? 2 : QG == 2 ? 3 : 4; }
template <int P> struct CodeRegs<0, P> {
    static constexpr int STEP = 64;
    __device__ __forceinline__ void load(const uint8_t *, u32, int) {}
};

template <int M, int P>
static __device__ __forceinline__ void scan_prefetch(CodeRegs<M, P> &cr, const uint8_t *cbase, u32 p0, u32 p1, int wv, int lane)
{
    if constexpr (M > 0) {
        const u32 pb = p0 + wv * CodeRegs<M, P>::STEP;
        if (pb < p1) cr.load(cbase, pb, lane);
    }
}

// One step of a wave over the STEP = 64 * PPL points whose codes sit in `cr` (positions pb.. of a list of p1 points):
// ADC sums for the QG queries (tables at absolute LDS offset tab_off), then selection of whatever beats the bounds.
template <int M, int QG, class S>
static __device__ __forceinline__ void scan_step(const CodeRegs<M, ppl_of<M, QG>()> &cr, u32 tab_off, u32 pb, u32 p1,
                                                 const float (&dc)[QG], const u32 (&sbase)[QG], int nvalid, S (&sel)[QG],
                                                 u32 (&thr_hi)[QG], int K, int lane, u64 *sthr, int dbg_flags)
{
#ifndef IVFADC_DEBUG
    dbg_flags = 0;   // knock-outs (wrong results by design) are compiled out of the shipped library
#endif
    using CR = CodeRegs<M, ppl_of<M, QG>()>;
    constexpr int PPL = CR::PPL;
        float acc[PPL][QG];
        if constexpr ((QG & 1) == 0) {
            // packed adds: the QG table entries of a code byte arrive as adjacent registers (one ds_read_b64/b128)
            v2f acc2[PPL][QG / 2];
#pragma unroll
            for (int r = 0; r < PPL; ++r)
#pragma unroll
                for (int h = 0; h < QG / 2; ++h) acc2[r][h] = (v2f){dc[2 * h], dc[2 * h + 1]};
#pragma unroll
            for (int ii = 0; ii < M; ++ii) {
#pragma unroll
                for (int r = 0; r < PPL; ++r) {
                    const u32 ea = cr.template byte_shl<entry_shift<QG>()>(r, ii) + (tab_off + (u32)ii * 1024u * QG);
                    if constexpr (QG == 4) {
                        const v4f t4 = lds_load_abs<v4f>(ea);
                        acc2[r][0] = acc2[r][0] + (v2f){t4.x, t4.y};
                        acc2[r][1] = acc2[r][1] + (v2f){t4.z, t4.w};
                    } else {
                        static_assert(QG == 2 || QG == 4, "packed path");
                        acc2[r][0] = acc2[r][0] + lds_load_abs<v2f>(ea);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < PPL; ++r)
#pragma unroll
                for (int h = 0; h < QG / 2; ++h) { acc[r][2 * h] = acc2[r][h].x; acc[r][2 * h + 1] = acc2[r][h].y; }
        } else {
#pragma unroll
        for (int r = 0; r < PPL; ++r)
#pragma unroll
            for (int s = 0; s < QG; ++s) acc[r][s] = dc[s];
        if (!(dbg_flags & 2)) {
#pragma unroll
        for (int ii = 0; ii < M; ++ii) {
#pragma unroll
            for (int r = 0; r < PPL; ++r) {
                static_assert(QG == 1 || (QG & 1) == 0, "odd QG > 1 is not instantiated");
                acc[r][0] = acc[r][0] + lds_load_abs<float>(cr.template byte_shl<2>(r, ii) + (tab_off + (u32)ii * 1024u));
            }
        }
        } else {
#pragma unroll
            for (int r = 0; r < PPL; ++r) acc[r][0] += __uint_as_float(cr.byte(r, 0) << 10);
        }
        }
        bool anyc = false;
#pragma unroll
        for (int r = 0; r < PPL; ++r)
#pragma unroll
            for (int s = 0; s < QG; ++s)
                anyc = anyc || (CR::point(pb, r, lane) < p1 && s < nvalid && __float_as_uint(acc[r][s]) <= thr_hi[s]);
        if (__any(anyc) && !(dbg_flags & 1)) {
#pragma unroll
            for (int s = 0; s < QG; ++s) sel[s].tighten(readfirstlane64(sthr[s]));
#pragma unroll
            for (int r = 0; r < PPL; ++r) {
                const u32 p = CR::point(pb, r, lane);
                scan_emit<QG>(acc[r], p, p < p1, nvalid, sbase, sel, K, lane);
            }
#pragma unroll
            for (int s = 0; s < QG; ++s) {
                publish_bound<QG>(sthr, s, sel[s], K, (u32)(sel[s].thr() >> 32) < thr_hi[s], lane);
                thr_hi[s] = (u32)(sel[s].thr() >> 32);
            }
        }
}

// Scan points [p0, p1) of one list for the QG queries whose tables are in `tab`; the four waves
// of the workgroup interleave blocks of the range.  sbase[s] + position = visit order of query s.
// `cr` holds the wave's first block (scan_prefetch); later blocks are loaded one step ahead.
template <int M, int QG, class S>
static __device__ __forceinline__ void scan_range(const float *tab, u32 tab_off, const uint8_t *cbase, int cs, int m, u32 p0, u32 p1,
                                                  const float (&dc)[QG], const u32 (&sbase)[QG], int nvalid, S (&sel)[QG],
                                                  int K, int wv, int lane, CodeRegs<M, ppl_of<M, QG>()> cr, u64 *sthr,
                                                  int dbg_flags = 0)
{
    // sthr[s] (LDS): the smallest K-th key any wave of the workgroup has found for slot s -- a valid bound
    // for every wave, so the four per-wave selectors prune like one workgroup-wide selector
    u32 thr_hi[QG];
#pragma unroll
    for (int s = 0; s < QG; ++s) {
        sel[s].tighten(readfirstlane64(sthr[s]));
        thr_hi[s] = (u32)(sel[s].thr() >> 32);
    }

    if constexpr (M > 0) {
        using CR = CodeRegs<M, ppl_of<M, QG>()>;
        constexpr u32 STEP = CR::STEP;
        for (u32 pb = p0 + wv * STEP; pb < p1; pb += 4 * STEP) {
            CR nx;
            const u32 pn = pb + 4 * STEP;
            if (pn < p1) nx.load(cbase, pn, lane);
            else nx = cr;
            scan_step<M, QG>(cr, tab_off, pb, p1, dc, sbase, nvalid, sel, thr_hi, K, lane, sthr, dbg_flags);
            cr = nx;
        }
    } else {
        // generic m: one point per lane, code stride cs (multiple of 4), dword loads
        const int nw = cs >> 2;
        for (u32 pb = p0 + wv * 64; pb < p1; pb += 256) {
            const u32 p = pb + lane;
            const u32 *cp = (const u32 *)(cbase + (size_t)p * cs);
            float acc[QG];
#pragma unroll
            for (int s = 0; s < QG; ++s) acc[s] = dc[s];
            if (p < p1) {
                for (int wd = 0; wd < nw; ++wd) {
                    const u32 dw = cp[wd];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int ii = wd * 4 + b;
                        if (ii < m) {
                            const u32 byte = (dw >> (8 * b)) & 0xffu;
                            float tv[QG];
                            TabV<QG>::ld(tab + ((size_t)ii * 256 + byte) * QG, tv);
#pragma unroll
                            for (int s = 0; s < QG; ++s) acc[s] = acc[s] + tv[s];
                        }
                    }
                }
            }
            bool anyc = false;
#pragma unroll
            for (int s = 0; s < QG; ++s) anyc = anyc || (p < p1 && s < nvalid && __float_as_uint(acc[s]) <= thr_hi[s]);
            if (__any(anyc)) {
#pragma unroll
                for (int s = 0; s < QG; ++s) sel[s].tighten(readfirstlane64(sthr[s]));
                scan_emit<QG>(acc, p, p < p1, nvalid, sbase, sel, K, lane);
#pragma unroll
                for (int s = 0; s < QG; ++s) {
                    publish_bound<QG>(sthr, s, sel[s], K, (u32)(sel[s].thr() >> 32) < thr_hi[s], lane);
                    thr_hi[s] = (u32)(sel[s].thr() >> 32);
                }
            }
        }
    }
}

// =======================================================================================================================
// Striped scan (round 2): table lookups without random bank conflicts, as a FILTER in front of the exact sum.
//
// The round-1 layout tab[ii][code][QG] puts the entry of code c in bank group c mod 16 (ds_read_b128) or bank c mod 32
// (ds_read_b32): all lanes of a ds_read service group (16 / 32 lanes) look the SAME sub-quantizer up with random codes, so
// they collide at random -- measured 2.4x (b128) to 2.9x (b32) the conflict-free time (tools/micro/lds_gather.hip), 52-63 %
// of all LDS cycles of the round-1 scan kernels, which were LDS-bound (97 % LDS busy on the SIFT1B shape).  Here every
// sub-quantizer owns a bank stripe of the table (build_tables_t<..., TAB_STRIPED>) and the lanes of a service group look
// DIFFERENT sub-quantizers up in the same instruction:
//
//   lane l (rotation j = l mod M) looks sub-quantizer (t + j) mod M up at slot t.
//
// Every lane still owns whole points, starts and finishes them with the others (no skew in time), but adds its M entries
// in a rotated order -- NOT the reference's.  Float addition is not associative, so these sums are only a FILTER:
//   * both sums add the same M + 1 non-negative terms, so they differ by less than 2 M ulp (relax_bound): a point whose
//     reference distance is <= the bound has a rotated sum <= bound + 4 (M + 1) bit-pattern steps, which is what the
//     candidate test compares with (no false negatives);
//   * whenever any lane of the wave passes that test (rare once the bound is tight), the wave recomputes the step's sums in
//     the reference's order -- d = dc; d += tab_ii[code_ii], ii ascending (index.jl:242-246) -- and only those exact sums
//     are compared with the true bound and enter the selectors: results stay bit-identical to the oracle.
// Cost per lookup: one more VALU op than round 1 (the stripe bits are OR-ed into the address), two v_perm per point to
// rotate its code bytes.  Banking: ds_read_b128 serves lanes in four groups of 16 whose lane numbers mod 16 are all
// different (MI355X_MICROARCH.md, LDS table), so with M = 16 the 16 lanes of a group read 16 different stripes --
// conflict-free; with M = 8 two lanes share a stripe of two slots (slot = (code & 1) * 8 + sub-quantizer): at most 2-way.
// (A time-skewed variant that keeps the reference's order in every lane -- fma-with-keep restarts, per-slot capture of
// finished sums -- was built first and measured: conflict-free but 2.3x the VALU instructions, 12.5 vs 11.8 ms on the
// SIFT1B shape)
// =======================================================================================================================
template <int M, int QG> static constexpr bool striped_scan() { return (M == 8 || M == 16) && QG == 4; }

template <int M> struct RotConst {
    u32 apre[M];       // ((t + j) mod M) * entry bytes: the stripe part of the lookup address at slot t
    u32 rsel[2];       // v_perm selectors of the byte rotation
    int j;
    template <int QG> __device__ __forceinline__ void init(int lane)
    {
        j = lane & (M - 1);
#pragma unroll
        for (int t = 0; t < M; ++t) apre[t] = (u32)((t + j) & (M - 1)) * 4u * QG;
        if constexpr (M == 8) {
            // out byte t = P[(t + j) mod 8], P = {lo (perm bytes 0..3), hi (4..7)}
            u32 a = 0, b2 = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                a |= (u32)((b + j) & 7) << (8 * b);
                b2 |= (u32)((4 + b + j) & 7) << (8 * b);
            }
            rsel[0] = a; rsel[1] = b2;
        } else {
            // M = 16: dwords rotated by j / 4 first, then every output dword is the byte window of two neighbouring rotated
            // dwords that starts at byte j mod 4 (the same window for all four dwords)
            u32 a = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) a |= (u32)(b + (j & 3)) << (8 * b);
            rsel[0] = a; rsel[1] = 0;
        }
    }
    // code bytes of one point -> out byte t = P[(t + j) mod M]
    __device__ __forceinline__ void rotate(const u32 (&pw)[M / 4], u32 (&out)[M / 4]) const
    {
        if constexpr (M == 8) {
            out[0] = __builtin_amdgcn_perm(pw[1], pw[0], rsel[0]);
            out[1] = __builtin_amdgcn_perm(pw[1], pw[0], rsel[1]);
        } else {
            u32 r1[4], q[4];
            const bool b0 = (j & 4) != 0, b1 = (j & 8) != 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) r1[k] = b0 ? pw[(k + 1) & 3] : pw[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = b1 ? r1[(k + 2) & 3] : r1[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = __builtin_amdgcn_perm(q[(k + 1) & 3], q[k], rsel[0]);
        }
    }
};

// (byte BI of dw) << SH in one VALU instruction (see CodeRegs::byte_shl)
template <int SH, int BI> static __device__ __forceinline__ u32 sdwa_byte_shl(u32 dw)
{
    const u32 sh = SH;
    u32 o;
    if constexpr (BI == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(o) : "s"(sh), "v"(dw));
    else if constexpr (BI == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(o) : "s"(sh), "v"(dw));
    else if constexpr (BI == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(o) : "s"(sh), "v"(dw));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(o) : "s"(sh), "v"(dw));
    return o;
}

template <int N> struct IntC { static constexpr int value = N; };
template <int N, class F> static __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(IntC<N - 1>{});
    }
}

// Bit pattern of a bound, relaxed for the rotated-order filter.  Both the rotated and the reference sum add the same
// M + 1 non-negative terms sequentially, each within M u S of the real sum S (u = 2^-24), so they differ by less than
// 2 M u S < 2 M ulp(S); across a binade boundary an ulp step of the larger value is two of the smaller: 4 (M + 1) steps of
// the bit pattern cover it.  Saturating: KEY_MAX's high word stays put.
template <int M> static __device__ __forceinline__ u32 relax_bound(u32 hi)
{
    constexpr u32 R = 4u * (M + 1);
    return hi > 0xFFFFFFFFu - R ? 0xFFFFFFFFu : hi + R;
}

// ---- what the filter lets through -----------------------------------------------------------------------------------
// Survivors are rare once the bound is tight (a few per thousand points) but not free: each needs its reference-order sum,
// and the selection machinery behind it (bound refresh from LDS, up to QG pushes, bound publication) costs hundreds of
// instructions per visit -- visited for one or two points at a time it took a third of the SIFT1B-shape scan.  So
// survivors are PARKED: their code bytes and list position go to a small per-wave buffer in LDS, and the buffer is
// worked off 64 / M points at a time.  Selection is order-free (k smallest of unique keys), so parking changes nothing
// but the moment a bound tightens.
constexpr int CAND_CAP = 16;   // entries per wave
template <int M> static constexpr int cand_stride() { return M / 4 + 1; }   // dwords: code bytes, list position

// Reference-order sums of parked points, 64 / M per pass: the lanes of segment c (M consecutive lanes) take entry
// base + c, lane ii of the segment looks sub-quantizer ii up -- M different stripes per point in ONE ds_read, where a
// lane walking its own point through sub-quantizer 0, 1, ... would put all 16 lanes of a service group on the two slots
// of one stripe (8-way) -- and the M entries are added in ascending order along the segment: step i adds lane i's entry
// to the running sum handed over from lane i-1 (DPP row_shr:1), so lane M-1 of the segment ends with
// ((dc + t0) + t1) + ... + t_{M-1}, the reference's sum (index.jl:242-246), and offers it to the selectors itself.
template <int M, int QG, class S>
static __device__ __forceinline__ void drain_parked(const u32 *cbuf, int cnt, const float (&dc)[QG], u32 tab_off, const u32 (&sbase)[QG],
                                                    int nvalid, S (&sel)[QG], u32 (&thr_hi)[QG], int K, int lane, u64 *sthr)
{
    static_assert(QG == 4, "striped scan: four queries per code stream");
    constexpr int SHC = stripe_shift<M, QG>();
    constexpr int NS = 64 / M, ES = cand_stride<M>();
    const int seg = lane / M, ii = lane & (M - 1);
    for (int base = 0; base < cnt; base += NS) {   // uniform
        const int e = base + seg;
        const bool ok = e < cnt;
        const u32 *ent = cbuf + (ok ? e : 0) * ES;
        const u32 dw = ent[ii >> 2];
        const u32 pos = ent[M / 4];
        const u32 byte = (dw >> (8 * (ii & 3))) & 0xffu;
        const v4f ev4 = lds_load_abs<v4f>(((byte << SHC) | ((u32)ii * 4u * QG)) + tab_off);
        const float ev[QG] = {ev4.x, ev4.y, ev4.z, ev4.w};
        float x[QG];
#pragma unroll
        for (int s = 0; s < QG; ++s) x[s] = dc[s] + ev[s];
#pragma unroll
        for (int i = 1; i < M; ++i)
#pragma unroll
            for (int s = 0; s < QG; ++s) {
                // lane l <- lane l-1 within a row of 16 (row_shr:1); what lane i reads at step i is lane i-1's value of step i-1
                const float up = __uint_as_float((u32)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x[s]), 0x111, 0xf, 0xf, false));
                x[s] = up + ev[s];
            }
#pragma unroll
        for (int s = 0; s < QG; ++s) sel[s].tighten(readfirstlane64(sthr[s]));
        scan_emit<QG>(x, pos, ok && ii == M - 1, nvalid, sbase, sel, K, lane);
#pragma unroll
        for (int s = 0; s < QG; ++s) {
            publish_bound<QG>(sthr, s, sel[s], K, (u32)(sel[s].thr() >> 32) < thr_hi[s], lane);
            thr_hi[s] = (u32)(sel[s].thr() >> 32);
        }
    }
}

// Striped counterpart of scan_step (QG = 4): rotated-order sums as the filter, reference-order sums for what passes.
// cbuf / ccnt: this wave's parking buffer and its fill (uniform).
// ---- m = 8: the filter in 16-bit integers (round 2) --------------------------------------------------------------------
// The striped f32 filter above is co-bound by the LDS array (79 % busy, half of it 2-way stripe conflicts on 16-byte entries) and
// by vector-ALU issue (77 %: two double-pass v_pk_add_f32 per lookup).  A FILTER does not need floats: the four queries' entries
// of (sub-quantizer, code) are quantised to q = min(4095, floor(t * inv_s)) with inv_s = 4095 / (largest entry of query s's
// tables), four 16-bit fields in ONE 8-byte word at QF_OFF + (sub-quantizer << 11 | code << 3) -- a ds_read_b64 per lookup, half
// the LDS bytes -- and a point's four sums are two 32-bit integer adds per lookup (fields never carry: 8 x 4095 < 2^15; one
// v_lshl_add_u64 instead was measured slower, 8.97 vs 8.88 ms).  Integer addition is associative, so the rotated order costs
// nothing here, and the test is exact arithmetic: with S the reference's float sum (dc, then the entries in ascending order:
// within (m + 1) u of the real sum), S <= thr implies
//   sum_i q_i <= (thr (1 + 2^-18) - dc) inv (1 + 2^-18)        [floor, fl(t inv) <= t inv (1 + u), 9 roundings of S]
// so T_s = floor of the right-hand side + 2, computed in floats.  What makes the float evaluation safe is the RELATIVE slack: the target
// uses thr (1 + 2^-18) where the argument needs thr (1 + 9 u), and 2^-18 thr = 64 u thr dominates everything the evaluation can lose --
// fl(thr c) (1 u thr), the subtraction of dc (exact or 1 u thr), the product with inv and the second factor (2 u of the result) -- whatever
// thr inv is (dc close to thr over tiny tables makes it arbitrarily large: the product then exceeds 32000 and T_s saturates at "every
// point passes").  dc and every entry are >= +0 (sums of squares; the probe arrays never hold -0.0 or NaN: the coarse kernels write
// sums of squares, and qf_targets clamps a negative difference to "nothing passes").  The comparison of four fields at once:
// X = (0x8000 | T_s) - sum_s per field has bit 15 set iff sum_s <= T_s and never borrows (sum_s <= 32760 < 2^15 + T_s).  What passes is
// parked and gets its reference-order float sum from the f32 tables exactly as before (drain_parked): results stay bit-identical
// (tests/test_gpu_parity.py::test_integer_filter_extremes: outlier codewords, zero / denormal / huge tables, dc far above the tables).
constexpr u32 QF_OFF = 8u * 256u * 4u * 4u;   // the 16-bit table sits right behind the m = 8, QG = 4 float tables (32 KB)
constexpr u32 QF_BYTES = 256u * 64u;

// all 256 threads, after the float tables are complete (caller barriers before and after): per-query maxima -> inv -> fields.
// scratch: 8 floats of LDS (the residual area is free once the tables are built): [0..3] maxima (as bits), [4..7] inv
static __device__ __forceinline__ void quantize_tables_m8(float *tabf, unsigned char *qtab, float *scratch, int tid, int lane)
{
    u32 *smax = (u32 *)scratch;
    if (tid < 4) smax[tid] = 0u;
    __syncthreads();
    v4f e[8];
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
        e[ii] = *(const v4f *)((const unsigned char *)tabf + (((u32)tid << 7) | ((u32)ii << 4)));
        mx[0] = fmaxf(mx[0], e[ii].x); mx[1] = fmaxf(mx[1], e[ii].y); mx[2] = fmaxf(mx[2], e[ii].z); mx[3] = fmaxf(mx[3], e[ii].w);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx[s] = fmaxf(mx[s], __shfl_xor(mx[s], off));
        if (lane == 0) atomicMax(&smax[s], __float_as_uint(mx[s]));   // entries are >= +0: the bit pattern orders like the value
    }
    __syncthreads();
    float inv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float m = __uint_as_float(smax[s]);
        inv[s] = m > 0.0f ? 4095.0f / m : 0.0f;
    }
    if (tid < 4) scratch[4 + tid] = inv[tid];
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
        const float ev[4] = {e[ii].x, e[ii].y, e[ii].z, e[ii].w};
        u32 q[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32 v = (u32)floorf(ev[s] * inv[s]);
            q[s] = v < 4095u ? v : 4095u;
        }
        *(uint2 *)(qtab + (((u32)ii << 11) | ((u32)tid << 3))) = make_uint2(q[0] | (q[1] << 16), q[2] | (q[3] << 16));
    }
}

// packed thresholds of the four queries: field s = 0x8000 | T_s (see above); thr_hi = high word of the bound (float bits)
static __device__ __forceinline__ void qf_targets(const u32 (&thr_hi)[4], const float (&dc)[4], const float (&inv)[4], int nvalid, u32 (&tg)[2], u32 (&mk)[2])
{
    u32 f[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32 T = 0x7FFFu;
        if (thr_hi[s] < 0x7F800000u) {   // a finite bound
            const float thr = __uint_as_float(thr_hi[s]);
            const float x = (thr * 1.0000038146972656f - dc[s]) * inv[s] * 1.0000038146972656f;   // (1 + 2^-18)
            T = x < 0.0f ? 0u : (x < 32000.0f ? (u32)x + 2u : 0x7FFFu);
        }
        f[s] = 0x8000u | T;
    }
    tg[0] = f[0] | (f[1] << 16);
    tg[1] = f[2] | (f[3] << 16);
    mk[0] = (nvalid > 0 ? 0x8000u : 0u) | (nvalid > 1 ? 0x80000000u : 0u);
    mk[1] = (nvalid > 2 ? 0x8000u : 0u) | (nvalid > 3 ? 0x80000000u : 0u);
}

// QF (m = 8 only): the 16-bit integer filter (quantize_tables_m8 / qf_targets): kc.apre holds 8-byte stripes, tg / mk the packed
// thresholds of the current bounds (refreshed here whenever a bound moved), inv the per-query scales.
struct QfState { u32 tg[2], mk[2]; float inv[4]; int dbg; };   // dbg: IVFADC_DEBUG_FLAGS in a debug build (1 = drop the filter's candidates)
template <int M, int QG, bool QF, class S>
static __device__ __forceinline__ void striped_scan_step(const CodeRegs<M, ppl_of<M, QG>()> &cr, const RotConst<M> &kc, u32 tab_off, u32 pb,
                                                         u32 p1, const float (&dc)[QG], const u32 (&sbase)[QG], int nvalid, S (&sel)[QG],
                                                         u32 (&thr_hi)[QG], int K, int lane, u64 *sthr, u32 *cbuf, int &ccnt, QfState &qf)
{
    static_assert(QG == 4, "striped scan: four queries per code stream");
    static_assert(!QF || M == 8, "integer filter: m = 8");
    using CR = CodeRegs<M, ppl_of<M, QG>()>;
    constexpr int PPL = CR::PPL;
    constexpr int SHC = stripe_shift<M, QG>();
    constexpr int ES = cand_stride<M>();
    u32 pw[PPL][M / 4], rw[PPL][M / 4];
    u64 fm[PPL], anym = 0;
    if constexpr (QF) {
        // plain [sub-quantizer][code] layout, no rotation: with 8-byte entries the LDS array is no longer what binds (random
        // b64 gathers: 45 query-lookups/clk/CU, the striped form 48; the kernel needs 23), the vector ALU is -- and this form
        // needs neither the stripe OR nor the byte rotation (3 instead of 4 vector instructions per lookup)
        u32 qa[PPL][2];
        static_for<PPL>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            cr.words(r, pw[r]);
            qa[r][0] = 0u;
            qa[r][1] = 0u;
        });
        static_for<M>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            static_for<PPL>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const u32 ea = sdwa_byte_shl<3, (t & 3)>(pw[r][t >> 2]);
                const v2u e = lds_load_abs<v2u>(ea + (tab_off + QF_OFF + (u32)t * 2048u));
                qa[r][0] += e.x;
                qa[r][1] += e.y;
            });
        });
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
            const bool c = (((qf.tg[0] - qa[r][0]) & qf.mk[0]) | ((qf.tg[1] - qa[r][1]) & qf.mk[1])) != 0u;
            fm[r] = __builtin_amdgcn_ballot_w64(c && CR::point(pb, r, lane) < p1);
            anym |= fm[r];
        }
    } else {
    v2f acc2[PPL][2];
    static_for<PPL>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        cr.words(r, pw[r]);
        kc.rotate(pw[r], rw[r]);
        acc2[r][0] = (v2f){dc[0], dc[1]};
        acc2[r][1] = (v2f){dc[2], dc[3]};
    });
    static_for<M>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        static_for<PPL>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const u32 ea = sdwa_byte_shl<SHC, (t & 3)>(rw[r][t >> 2]) | kc.apre[t];
            const v4f t4 = lds_load_abs<v4f>(ea + tab_off);
            acc2[r][0] = acc2[r][0] + (v2f){t4.x, t4.y};
            acc2[r][1] = acc2[r][1] + (v2f){t4.z, t4.w};
        });
    });
    u32 rb[QG];
#pragma unroll
    for (int s = 0; s < QG; ++s) rb[s] = relax_bound<M>(thr_hi[s]);
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
        const float a4[QG] = {acc2[r][0].x, acc2[r][0].y, acc2[r][1].x, acc2[r][1].y};
        bool c = false;
#pragma unroll
        for (int s = 0; s < QG; ++s) c = c || (s < nvalid && __float_as_uint(a4[s]) <= rb[s]);
        fm[r] = __ballot(c && CR::point(pb, r, lane) < p1);
        anym |= fm[r];
    }
    }
#ifdef IVFADC_DEBUG
    if constexpr (QF) {
        if (qf.dbg & 1) anym = 0;   // knock-out (wrong results by design): the filter's fast path alone
    }
#endif
    if (anym) {
        u32 thr_before[QG];
#pragma unroll
        for (int s = 0; s < QG; ++s) thr_before[s] = thr_hi[s];
        int n = 0;
#pragma unroll
        for (int r = 0; r < PPL; ++r) n += __popcll(fm[r]);
        if (n > CAND_CAP / 2) {
            // a crowd (an unset or loose bound at the start of a work item): reference-order sums for the whole step, every
            // lane for its own points (all lanes on one stripe per instruction: slow, and rare)
            float acc[PPL][QG];
#pragma unroll
            for (int r = 0; r < PPL; ++r)
#pragma unroll
                for (int s = 0; s < QG; ++s) acc[r][s] = dc[s];
            static_for<M>([&](auto ic) {
                constexpr int ii = decltype(ic)::value;
                static_for<PPL>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const u32 ea = sdwa_byte_shl<SHC, (ii & 3)>(pw[r][ii >> 2]) + (tab_off + (u32)ii * 4u * QG);
                    const v4f t4 = lds_load_abs<v4f>(ea);
                    acc[r][0] = acc[r][0] + t4.x;
                    acc[r][1] = acc[r][1] + t4.y;
                    acc[r][2] = acc[r][2] + t4.z;
                    acc[r][3] = acc[r][3] + t4.w;
                });
            });
#pragma unroll
            for (int s = 0; s < QG; ++s) sel[s].tighten(readfirstlane64(sthr[s]));
#pragma unroll
            for (int r = 0; r < PPL; ++r) scan_emit<QG>(acc[r], CR::point(pb, r, lane), ((fm[r] >> lane) & 1ull) != 0, nvalid, sbase, sel, K, lane);
#pragma unroll
            for (int s = 0; s < QG; ++s) {
                publish_bound<QG>(sthr, s, sel[s], K, (u32)(sel[s].thr() >> 32) < thr_hi[s], lane);
                thr_hi[s] = (u32)(sel[s].thr() >> 32);
            }
        } else {
            if (ccnt + n > CAND_CAP) {   // uniform
                wave_sync();
                drain_parked<M, QG>(cbuf, ccnt, dc, tab_off, sbase, nvalid, sel, thr_hi, K, lane, sthr);
                ccnt = 0;
                wave_sync();
            }
#pragma unroll
            for (int r = 0; r < PPL; ++r) {
                if (((fm[r] >> lane) & 1ull) != 0) {
                    u32 *ent = cbuf + (ccnt + __popcll(fm[r] & ((1ull << lane) - 1ull))) * ES;
#pragma unroll
                    for (int k = 0; k < M / 4; ++k) ent[k] = pw[r][k];
                    ent[M / 4] = CR::point(pb, r, lane);
                }
                ccnt += __popcll(fm[r]);
            }
            if (ccnt >= 64 / M) {        // a full pass is waiting
                wave_sync();
                drain_parked<M, QG>(cbuf, ccnt, dc, tab_off, sbase, nvalid, sel, thr_hi, K, lane, sthr);
                ccnt = 0;
                wave_sync();
            }
        }
        if constexpr (QF) {
            bool moved = false;
#pragma unroll
            for (int s = 0; s < QG; ++s) moved = moved || thr_hi[s] != thr_before[s];
            if (moved) qf_targets(thr_hi, dc, qf.inv, nvalid, qf.tg, qf.mk);   // uniform
        }
    }
}

// Striped counterpart of scan_range (list-major kernel, striped tables at LDS offset tab_off).  cbuf: CAND_CAP entries of
// cand_stride<M>() dwords per wave.
template <int M, int QG, bool QF, class S>
static __device__ __forceinline__ void striped_scan_range(u32 tab_off, const uint8_t *cbase, u32 p0, u32 p1, const float (&dc)[QG],
                                                          const u32 (&sbase)[QG], int nvalid, S (&sel)[QG], int K, int wv, int lane,
                                                          CodeRegs<M, ppl_of<M, QG>()> cr, u64 *sthr, u32 *cbuf, const float *qf_inv = nullptr,
                                                          int dbg_flags = 0)
{
    using CR = CodeRegs<M, ppl_of<M, QG>()>;
    constexpr u32 STEP = CR::STEP;
    u32 thr_hi[QG];
#pragma unroll
    for (int s = 0; s < QG; ++s) {
        sel[s].tighten(readfirstlane64(sthr[s]));
        thr_hi[s] = (u32)(sel[s].thr() >> 32);
    }
    RotConst<M> kc;
    kc.template init<QG>(lane);   // (the integer filter does not rotate: kc is dead code there)
    QfState qf;
    qf.dbg = dbg_flags;
    if constexpr (QF) {
#pragma unroll
        for (int s = 0; s < QG; ++s) qf.inv[s] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(qf_inv[s])));
        qf_targets(thr_hi, dc, qf.inv, nvalid, qf.tg, qf.mk);
    }
    int ccnt = 0;
#if IVFADC_PF_DEPTH <= 1
    for (u32 pb = p0 + wv * STEP; pb < p1; pb += 4 * STEP) {
        CR nx;
        const u32 pn = pb + 4 * STEP;
        if (pn < p1) nx.load(cbase, pn, lane);
        else nx = cr;
        striped_scan_step<M, QG, QF>(cr, kc, tab_off, pb, p1, dc, sbase, nvalid, sel, thr_hi, K, lane, sthr, cbuf, ccnt, qf);
        cr = nx;
    }
#else
    // Code stream, IVFADC_PF_DEPTH register sets in rotation: a set is refilled (for the step DEPTH wave-steps ahead) the moment its own
    // step has been computed, so DEPTH - 1 loads of 1 KB per wave are in flight at any time and each has DEPTH - 1 steps to arrive.  With
    // one set ahead (round 1-4) a wave held a single kilobyte in flight for the ~0.25 us a step takes: at three workgroups per CU that is
    // 3 MB on the whole chip, i.e. ~3 TB/s at the loaded latency of HBM -- what the SIFT1B-shape scans measured (profiles/r04_sift1b_*:
    // 0.36-0.38 of the HBM peak with SQ_WAIT_ANY at 38 % of the wave cycles and no pipe above 40 %).
    {
        // (the sets rotate by register moves -- a few v_mov per step -- rather than by unrolling the loop DEPTH times: the step body is
        // large, and DEPTH copies of it cost more in instruction fetch than the moves do: measured 2.3 against 1.68 ms)
        constexpr int DEPTH = IVFADC_PF_DEPTH;
        constexpr u32 STR = 4 * STEP;
        CR ahead[DEPTH - 1];
        const u32 pfirst = p0 + wv * STEP;
#pragma unroll
        for (int k = 0; k < DEPTH - 1; ++k) {
            if (pfirst + (u32)(k + 1) * STR < p1) ahead[k].load(cbase, pfirst + (u32)(k + 1) * STR, lane);
            else ahead[k] = cr;
        }
        for (u32 pb = pfirst; pb < p1; pb += STR) {
            CR nx;
            const u32 pn = pb + DEPTH * STR;
            if (pn < p1) nx.load(cbase, pn, lane);
            else nx = cr;
            striped_scan_step<M, QG, QF>(cr, kc, tab_off, pb, p1, dc, sbase, nvalid, sel, thr_hi, K, lane, sthr, cbuf, ccnt, qf);
            cr = ahead[0];
#pragma unroll
            for (int k = 0; k + 1 < DEPTH - 1; ++k) ahead[k] = ahead[k + 1];
            ahead[DEPTH - 2] = nx;
        }
    }
#endif
    if (ccnt > 0) {
        wave_sync();
        drain_parked<M, QG>(cbuf, ccnt, dc, tab_off, sbase, nvalid, sel, thr_hi, K, lane, sthr);
    }
}

// Query-major rounds of two probes: the steps of the two lists are interleaved (list 0 step 0, list 1 step 0, list 0
// step 1, ...).  Both first steps were prefetched while the tables were built, and every later step is requested two
// step computations before it is consumed instead of one -- same registers in flight (current, other list's current,
// next), twice the distance for the code stream's latency, which is what a lone workgroup waits on: probed lists are
// a few steps long.  Selection is order-free (k smallest of unique keys), so the result does not change.
template <int M, class S>
static __device__ __forceinline__ void scan_pair(u32 toff0, u32 toff1, const uint8_t *cb0, const uint8_t *cb1, u32 len0, u32 len1,
                                                 float dc0, float dc1, u32 sb0, u32 sb1, S (&sel)[1], int K, int wv, int lane,
                                                 CodeRegs<M, ppl_of<M, 1>()> a, CodeRegs<M, ppl_of<M, 1>()> b, u64 *sthr,
                                                 int dbg_flags)
{
    using CR = CodeRegs<M, ppl_of<M, 1>()>;
    constexpr u32 STEP = CR::STEP;
    u32 thr_hi[1];
    sel[0].tighten(readfirstlane64(sthr[0]));
    thr_hi[0] = (u32)(sel[0].thr() >> 32);
    const float dca[1] = {dc0}, dcb[1] = {dc1};
    const u32 sba[1] = {sb0}, sbb[1] = {sb1};
    u32 pa = wv * STEP, pb = wv * STEP;
    while (pa < len0 || pb < len1) {   // uniform
        if (pa < len0) {
            CR nx;
            const u32 pn = pa + 4 * STEP;
            if (pn < len0) nx.load(cb0, pn, lane);
            else nx = a;
            scan_step<M, 1>(a, toff0, pa, len0, dca, sba, 1, sel, thr_hi, K, lane, sthr, dbg_flags);
            a = nx;
            pa = pn;
        }
        if (pb < len1) {
            CR nx;
            const u32 pn = pb + 4 * STEP;
            if (pn < len1) nx.load(cb1, pn, lane);
            else nx = b;
            scan_step<M, 1>(b, toff1, pb, len1, dcb, sbb, 1, sel, thr_hi, K, lane, sthr, dbg_flags);
            b = nx;
            pb = pn;
        }
    }
}

// LDS carve shared by both scan kernels:
//   tab    [m][256][QG] f32   (after the scan its first bytes are reused as the exchange area)
//   resid  [d][QG] f32
//   selbuf [4][QG][cap] u64   (only when !SMALL)
//   scnt   [4][QG] int, swi [4] u32
// Exchange area (SMALL): xch[4][QG][64] u64 = 2 KB * QG, always <= the table size (m KB * QG).
struct LdsCarve {
    float *tab, *resid;
    u64 *selbuf, *xch;
    int *scnt;
    u32 *swi;
    u64 *sthr;   // [QG] workgroup-shared thresholds
    int xcap;
};

// NSEL = LDS selectors per wave (K > 64 only): one per query of the group in the list-major kernel, ONE in the query-major kernel, whose
// PG tables all belong to the same query
template <int QG, bool SMALL, int NSEL = QG>
static __device__ __forceinline__ LdsCarve carve_lds(unsigned char *smem, int m, int d, int cap, int extra_bytes = 0)
{
    LdsCarve c;
    c.tab = (float *)smem;
    // the table region is never smaller than the exchange area that later aliases it (m == 1); extra_bytes (a multiple of
    // 16): room behind the tables for the m = 8 list-major kernel's 16-bit filter table
    c.resid = c.tab + (size_t)(m < 2 ? 2 : m) * 256 * QG + (extra_bytes >> 2);
    u64 *after = (u64 *)(c.resid + (((size_t)d * QG + 3) & ~(size_t)3));
    if (SMALL) {
        c.selbuf = nullptr;
        c.xch = (u64 *)smem;
        c.xcap = 64;
        c.scnt = (int *)after;
    } else {
        c.selbuf = after;
        c.xch = after;
        c.xcap = cap;
        c.scnt = (int *)(after + (size_t)4 * NSEL * cap);
    }
    c.swi = (u32 *)(c.scnt + 4 * QG);
    c.sthr = (u64 *)(c.swi + 4);
    return c;
}

// ---------------------------------------------------------------------------------------
// Scan kernel A, "list-major": work item = (inverted list, group of up to QG queries that
// probe it, chunk of the list).  The code stream is read once per group instead of once per
// query; per-(probe, chunk) partial top-K go to HBM and a merge kernel finishes.  For long
// lists (billion-scale shapes).  Workgroups pull items from a device-side queue until empty.
// ---------------------------------------------------------------------------------------
struct ScanArgs {
    IndexView ix;
    const float *queries;
    int w, K, cap;
    const float *probe_dc;
    const u32 *probe_base;
    const u32 *list_cnt;
    const u32 *bucket_off;
    const u32 *wi_off;
    const u32 *bucket_items;
    u32 *queue_head;
    u64 *qthr;
    u64 *part_keys;
    u32 *part_cnt;
    int maxch;
    u32 CH;
    // direct mode (QG == 1 only): work item i = chunk (i % maxch) of probe (i / maxch); no grouping by list is needed
    // when every (query, probe) pair is its own item, so the two bucket kernels are skipped
    const int *probe_list;
    u32 direct_items;   // 0: items come from the bucket arrays
    int prune;          // skip work items whose queries all have a K-th best key below the list's coarse distance (exact)
    u64 *scanned_points;   // statistics: word 8 (q & 63) + 1 counts the points pruning skipped
};

// (M, DS) = compile-time (m, dsub) pair, or (0, 0) for any shape
// STRIPE: bank-striped tables + rotated-order filter sums (striped_scan_step above); m = 8 / 16 with QG = 4 only.
template <int M, int DS, int QG, bool SMALL, bool STRIPE = false>
__global__ __launch_bounds__(256) void scan_kernel(const ScanArgs a)
{
    static_assert(!STRIPE || striped_scan<M, QG>(), "striped tables: m = 8 / 16, four queries per code stream");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const IndexView &ix = a.ix;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = (M > 0) ? M : ix.m;
    const int K = a.K, cap = a.cap;
    // m = 8 striped: the 16-bit integer filter table sits behind the float tables (quantize_tables_m8)
    constexpr bool QF = STRIPE && M == 8 && QG == 4;
    const LdsCarve L = carve_lds<QG, SMALL>(smem_raw, m, ix.d, cap, QF ? (int)QF_BYTES : 0);
    const bool direct = QG == 1 && a.direct_items != 0;
    const u32 total = direct ? a.direct_items : a.wi_off[ix.kc];

    for (;;) {
        __syncthreads();
        if (tid == 0) L.swi[0] = atomicAdd(a.queue_head, 1u);
        __syncthreads();
        const u32 wi = __builtin_amdgcn_readfirstlane(L.swi[0]);
        if (wi >= total) break;   // uniform: every wave of every workgroup reaches this

        int l;
        u32 cnt, chunk, grp, direct_probe = 0;
        if (direct) {
            // rank-major order: every query's closest cell first, then every second-closest, ... -- by the time the farther cells of
            // a query come up, the items of its closer ones have usually published a bound, and the pruning below can fire
            const u32 t = wi / (u32)a.maxch, nqd = a.direct_items / ((u32)a.maxch * (u32)a.w);
            chunk = wi - t * (u32)a.maxch;
            const u32 j = t / nqd;
            direct_probe = (t - j * nqd) * (u32)a.w + j;
            l = a.probe_list[direct_probe];
            cnt = 1;
            grp = 0;
        } else {
            int lo = 0, hi = ix.kc;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (a.wi_off[mid] <= wi) lo = mid; else hi = mid;
            }
            l = lo;
            cnt = a.list_cnt[l];
            const u32 ng = (cnt + QG - 1) / QG;
            const u32 local = wi - a.wi_off[l];
            chunk = local / ng;
            grp = local - chunk * ng;
        }
        const u32 len = ix.list_len[l];
        const u32 p0 = chunk * a.CH;
        if (p0 >= len) continue;   // uniform (direct mode: a chunk slot past the end of a short list)
        const u32 p1 = min(len, p0 + a.CH);
        const int nvalid = min((int)QG, (int)(cnt - grp * QG));

        u32 pidx[QG], sbase[QG];
        int qi[QG];
        float dc[QG];
        u64 hard[QG];
        WSel<SMALL> sel[QG];
#pragma unroll
        for (int s = 0; s < QG; ++s) {
            const int ss = s < nvalid ? s : 0;
            pidx[s] = direct ? direct_probe : a.bucket_items[a.bucket_off[l] + grp * QG + ss];
            qi[s] = (int)(pidx[s] / (u32)a.w);
            dc[s] = a.probe_dc[pidx[s]];
            sbase[s] = a.probe_base[pidx[s]];
            const u64 t0 = readfirstlane64(__hip_atomic_load(&a.qthr[qi[s]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            sel[s].init(t0, SMALL ? nullptr : L.selbuf + ((size_t)wv * QG + s) * cap, cap, K);
            hard[s] = t0;
            if (tid == 0) arm_bound<QG>(L.sthr, s, t0);    // published by the barrier after the residuals
        }

        // Exact pruning (see qscan_kernel): no sum of this list can be below its coarse distance, so a query whose K-th best key
        // so far (the per-query bound in HBM, published by finished work items) already lies below dc has nothing to gain here;
        // when that holds for every query of the group the item is skipped -- its partial results are empty.  Work items follow
        // the lists, not the probe rank, so how often the bound has arrived in time depends on the order; it costs four
        // compares.  Every wave read the bounds for itself (hard[]), and another workgroup may have published in between: thread 0
        // decides for the workgroup and hands the verdict over through LDS.
        if (a.prune) {
            if (tid == 0) {
                bool all = true;
#pragma unroll
                for (int s = 0; s < QG; ++s) all = all && (s >= nvalid || __float_as_uint(dc[s]) > (u32)(hard[s] >> 32));
                L.swi[1] = all ? 1u : 0u;
            }
            __syncthreads();
            if (L.swi[1] != 0u) {   // uniform
                if (tid < nvalid) {
                    u32 pi = pidx[0];
#pragma unroll
                    for (int s = 1; s < QG; ++s) pi = tid == s ? pidx[s] : pi;
                    a.part_cnt[(size_t)pi * a.maxch + chunk] = 0u;
                    atomicAdd(a.scanned_points + (size_t)(pi & 63u) * 8 + 1, (u64)(p1 - p0));
                }
                continue;
            }
        }
        int li[QG];
#pragma unroll
        for (int s = 0; s < QG; ++s) li[s] = l;
        const uint8_t *cbase = ix.codes + ix.list_codeoff[l];
        CodeRegs<M, ppl_of<M, QG>()> cr;
        scan_prefetch(cr, cbase, p0, p1, wv, lane);     // in flight while the tables are built
        build_residuals<QG>(ix, a.queries, qi, li, L.resid, tid);
        __syncthreads();
        if constexpr (STRIPE) build_tables_t<QG, DS, TAB_STRIPED, M>(ix, m, L.resid, L.tab, tid);
        else build_tables_t<QG, DS, TAB_INTERLEAVED>(ix, m, L.resid, L.tab, tid);
        __syncthreads();
        if constexpr (QF) {   // the residuals are consumed: their first 32 bytes serve as scratch (maxima, scales)
            quantize_tables_m8(L.tab, (unsigned char *)L.tab + QF_OFF, L.resid, tid, lane);
            __syncthreads();
        }

        // scanning waves issue first: their few VALU ops feed the LDS pipe, which co-resident table builders would
        // otherwise starve (measured: +4 % on the SIFT1M shape, +1 % on SIFT1B, neutral elsewhere)
        __builtin_amdgcn_s_setprio(3);
        if constexpr (STRIPE)
            striped_scan_range<M, QG, QF>(0u, cbase, p0, p1, dc, sbase, nvalid, sel, K, wv, lane, cr, L.sthr,
                                          (u32 *)(L.sthr + STHR_WORDS * QG) + 192 + wv * (CAND_CAP * cand_stride<M>()),   // behind the 768-B probe cache
                                          L.resid + 4, ix.dbg_flags);
        else scan_range<M, QG>(L.tab, 0u, cbase, ix.cs, m, p0, p1, dc, sbase, nvalid, sel, K, wv, lane, cr, L.sthr);
        __builtin_amdgcn_s_setprio(0);

        // ---- per-wave flush, then wave s merges slot s of the four waves and publishes it
        int mycnt[QG];
#pragma unroll
        for (int s = 0; s < QG; ++s) mycnt[s] = sel[s].finish(K, lane);
        if (SMALL) __syncthreads();   // the exchange area aliases the tables: every wave must be done scanning
#pragma unroll
        for (int s = 0; s < QG; ++s) {
            sel[s].store(L.xch + ((size_t)wv * QG + s) * L.xcap, mycnt[s], lane);
            if (lane == 0) L.scnt[wv * QG + s] = mycnt[s];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < QG; ++s) {
            if (s == wv && s < nvalid) {
                merge_waves(sel[s], L.xch + (size_t)s * L.xcap, (size_t)QG * L.xcap, L.scnt + s, QG, K, hard[s], wv, lane);
                const int fc = sel[s].finish(K, lane);
                const size_t slot = (size_t)pidx[s] * a.maxch + chunk;
                u64 *dst = a.part_keys + slot * K;
                sel[s].for_each(fc, lane, [&](int i, u64 key) { dst[i] = key; });
                if (lane == 0) {
                    a.part_cnt[slot] = (u32)fc;
                    if (fc == K) atomicMin(&a.qthr[qi[s]], sel[s].thr());
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Final merge for kernel A (one wave per query): k-smallest over the per-(probe, chunk)
// partial results, then visit order -> stored id (index.jl:248,252,257).  Re-arms per-call state.
// ---------------------------------------------------------------------------------------
// prow_list / prow_base: the w probes of this query (a row of the global probe arrays, or the LDS copy)
static __device__ __forceinline__ void emit_result(u64 key, int i, int q, int w, int K, const int *prow_list, const u32 *prow_base,
                                                   const int64_t *list_pos, const u32 *ids, u32 *out_ids, float *out_dists)
{
    const u32 seq = (u32)key;
    int lo = 0, hi = w;   // the owning probe is the LAST j with base[j] <= seq (empty probes share a base)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prow_base[mid] <= seq) lo = mid; else hi = mid;
    }
    const int l = prow_list[lo];
    const int64_t pos = list_pos[l] + (int64_t)(seq - prow_base[lo]);
    out_ids[(size_t)q * K + i] = ids ? ids[pos] : (u32)pos;
    out_dists[(size_t)q * K + i] = __uint_as_float((u32)(key >> 32));
}

template <bool SMALL>
__global__ __launch_bounds__(256) void merge_kernel(int nq, int w, int K, int cap, int maxch, u32 CH, int kc,
                                                    const int *__restrict__ probe_list, const u32 *__restrict__ probe_base,
                                                    const int64_t *__restrict__ list_pos, const u32 *__restrict__ list_len,
                                                    const u32 *__restrict__ ids,
                                                    const u64 *__restrict__ part_keys, const u32 *__restrict__ part_cnt,
                                                    u32 *__restrict__ out_ids, float *__restrict__ out_dists,
                                                    int *__restrict__ out_counts, u64 *__restrict__ qthr, u32 *__restrict__ list_cnt,
                                                    u32 *__restrict__ queue_head, int nparts, int part, u64 *__restrict__ out_keys)
{
    // nparts > 1: probes of other ranks' lists have no partial results here; out_keys != null: the K smallest KEYS of this rank's lists
    // leave as they are (distance bits << 32 | visit order) for the merge across ranks (partial_merge_kernel) instead of ids
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *sbuf = (u64 *)smem_raw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // re-arm the probe histogram and the work queue for the next call (the scan kernel has finished)
    for (int l = blockIdx.x * 256 + threadIdx.x; l < kc; l += gridDim.x * 256) list_cnt[l] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *queue_head = 0u;
    const int q = blockIdx.x * 4 + wv;
    if (q >= nq) return;
    WSel<SMALL> sel;
    sel.init(KEY_MAX, sbuf + (size_t)wv * cap, cap, K);
    if (w <= 64) {
        // all (probe, chunk, rank) entries of the query as ONE flat sequence: lane j first learns probe j's share
        // (one round of loads for all probes), then 64 entries are fetched per step -- instead of four dependent
        // loads per probe, one probe after the other (8 us for a single query with 8 probes)
        const size_t pi0 = (size_t)q * w;
        int ent = 0;
        if (lane < w) {
            const int pl = probe_list[pi0 + lane];
            const u32 len = (nparts <= 1 || (pl % nparts) == part) ? list_len[pl] : 0u;
            ent = (int)((len + CH - 1) / CH) * K;
        }
        int incl = ent;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const int excl = incl - ent;
        const int T = __shfl(incl, 63);
        for (int e0 = 0; e0 < T; e0 += 64) {
            const int e = e0 + lane;
            bool pred = e < T;
            int lo = 0, hi = w - 1;   // owning probe: the first j with incl_j > e
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi) >> 1;
                const int v = __shfl(incl, mid);
                if (lo < hi) { if (v > e) hi = mid; else lo = mid + 1; }
            }
            const int local = e - __shfl(excl, lo);
            u64 key = KEY_MAX;
            if (pred) {
                const int c = local / K, i = local - c * K;
                const size_t slot = (pi0 + lo) * maxch + c;
                pred = (u32)i < part_cnt[slot];
                if (pred) key = part_keys[slot * K + i];
            }
            pred = pred && key < sel.thr();
            sel.push(pred, key, K, lane);
        }
    } else
    for (int j = 0; j < w; ++j) {
        const size_t pi = (size_t)q * w + j;
        const int l = probe_list[pi];
        if (nparts > 1 && (l % nparts) != part) continue;   // uniform
        const u32 len = list_len[l];
        const int nch = (int)((len + CH - 1) / CH);
        const int tot = nch * K;
        for (int e0 = 0; e0 < tot; e0 += 64) {
            const int e = e0 + lane;
            bool pred = e < tot;
            u64 key = KEY_MAX;
            if (pred) {
                const int c = e / K, i = e - c * K;
                const size_t slot = pi * maxch + c;
                pred = (u32)i < part_cnt[slot];
                if (pred) key = part_keys[slot * K + i];
            }
            pred = pred && key < sel.thr();
            sel.push(pred, key, K, lane);
        }
    }
    const int cnt = sel.finish(K, lane);
    if (out_keys) sel.for_each(cnt, lane, [&](int i, u64 key) { out_keys[(size_t)q * K + i] = key; });
    else sel.for_each(cnt, lane, [&](int i, u64 key) {
        emit_result(key, i, q, w, K, probe_list + (size_t)q * w, probe_base + (size_t)q * w, list_pos, ids, out_ids, out_dists);
    });
    if (lane == 0) {
        out_counts[q] = cnt;
        qthr[q] = KEY_MAX;
    }
}

// List-partitioned multi-GPU mode: the K-way merge of the ranks' partial top-K (index.jl:247-257 over the union of the ranks' lists: keys are
// unique per query -- a visit order names one stored point -- so the K smallest of the union are the K smallest of the whole scan).
// One wave per query; rank r's keys at keys + r * rank_stride_u64 (query q at + q * K), its counts at counts + r * rank_stride_i32.
template <bool SMALL>
__global__ __launch_bounds__(256) void partial_merge_kernel(int nq, int w, int K, int cap, int nranks, const u64 *__restrict__ keys,
                                                            size_t rank_stride_u64, const int *__restrict__ counts, size_t rank_stride_i32,
                                                            const int *__restrict__ probe_list, const u32 *__restrict__ probe_base,
                                                            const int64_t *__restrict__ list_pos, const u32 *__restrict__ ids,
                                                            u32 *__restrict__ out_ids, float *__restrict__ out_dists, int *__restrict__ out_counts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *sbuf = (u64 *)smem_raw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wv;
    if (q >= nq) return;
    WSel<SMALL> sel;
    sel.init(KEY_MAX, sbuf + (size_t)wv * cap, cap, K);
    for (int r = 0; r < nranks; ++r) {
        const int c = counts[(size_t)r * rank_stride_i32 + q];
        sel_absorb(sel, keys + (size_t)r * rank_stride_u64 + (size_t)q * K, c < K ? c : K, K, lane);
    }
    const int cnt = sel.finish(K, lane);
    sel.for_each(cnt, lane, [&](int i, u64 key) {
        emit_result(key, i, q, w, K, probe_list + (size_t)q * w, probe_base + (size_t)q * w, list_pos, ids, out_ids, out_dists);
    });
    if (lane == 0) out_counts[q] = cnt;
}

// ---------------------------------------------------------------------------------------
// Scan kernel B, "query-major": one workgroup per query walks its w probes in rank order
// (closest cell first, so the threshold tightens early), rebuilding the 1-query table per
// probe; the per-wave selectors persist across probes and the workgroup writes the final
// top-K itself: no partial results, no grouping, no merge kernel.  For short lists, where the
// per-(query, probe) fixed costs dominate the byte stream (SIFT1M-shape).
// ---------------------------------------------------------------------------------------
#include "lbscan.hip.h"
#include "nfscan.hip.h"
#include "wg8scan.hip.h"
#include "wg8q8scan.hip.h"

struct QScanArgs {
    IndexView ix;
    LbView lb;         // lower-bound tables on the matrix cores (LB kernels only)
    const float *queries;
    int nq, w, K, cap;
    const int *probe_list;
    const float *probe_dc;
    const u32 *probe_base;
    u32 *out_ids;
    float *out_dists;
    int *out_counts;
    u64 *dbg;   // diagnostic phase stamps (IVFADC_DEBUG_STAMPS=1), else null: [workgroup][16] cycles
    // fused coarse top-w (w <= 64): when cdist != null the workgroup selects its own probes from its row of the
    // coarse distances (coarsequantizers.jl:35-36) and the probe_* arrays above are not read
    const float *cdist;
    u64 *scanned_points;
    int approx;        // cdist holds MFMA scores: certify + refine (refine_probes)
    RefineArgs rf;
    int prune;         // skip the probes whose coarse distance already lies above the K-th best key (exact: see the round loop)
};

// phase stamps and knock-out flags exist only in a diagnostic build (-DIVFADC_DEBUG): the shipped library carries neither
#ifdef IVFADC_DEBUG
#define STAMP() (a.dbg ? (u64)__builtin_readcyclecounter() : 0ull)
#else
#define STAMP() 0ull
#endif

// Four workgroups per CU (<= 128 VGPRs) for the light shapes: a batch of 1024 queries is then resident at once.
// LB: the rounds of lbscan.hip.h (8-bit lower-bound tables from the matrix cores, exact sums for the survivors) instead of
// the exact f32 tables; K <= 64 and w <= 32 only (register selectors, LDS copy of the probes).
template <int M, int DS, int PG, bool SMALL, bool LB = false>
static __device__ __forceinline__ void qscan_body(const QScanArgs &a, unsigned char *smem_raw, const int q)
{
    const IndexView &ix = a.ix;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = (M > 0) ? M : ix.m;
    const int K = a.K, cap = a.cap, w = a.w;
    static_assert(!LB || (SMALL && M > 0 && DS > 0), "LB rounds: compile-time shape, register selectors");
    // carve: PG tables of m x 256 (same bytes as an interleaved QG = PG table), residuals [d][PG]
    LdsCarve L = carve_lds<PG, SMALL, 1>(smem_raw, m, ix.d, cap);
    int stage_floats = (m < 2 ? 2 : m) * 256 * PG;   // scratch of the prologue: the table area
    if constexpr (LB) {
        // tables, bf16 residuals, seeds, query and parking buffers (LbCfg), then the same tail as carve_lds
        using C = LbCfg<M, DS, PG>;
        stage_floats = (int)(C::TAB_BYTES / 4u);
        L.tab = (float *)smem_raw;
        L.resid = nullptr;
        L.selbuf = nullptr;
        L.xch = (u64 *)smem_raw;
        L.xcap = 64;
        L.scnt = (int *)(smem_raw + C::END);
        L.swi = (u32 *)(L.scnt + 4 * PG);
        L.sthr = (u64 *)(L.swi + 4);
    }

    WSel<SMALL> sel[1];
    sel[0].init(KEY_MAX, SMALL ? nullptr : L.selbuf + (size_t)wv * cap, cap, K);
    if (tid == 0) arm_bound<1>(L.sthr, 0, KEY_MAX);   // published by the first round's barriers (one slot: the PG probes of a round are one query's)
    u64 tph[5] = {0, 0, 0, 0, 0};
    u64 tpro[6] = {0, 0, 0, 0, 0, 0};
    const u64 tstart = STAMP();

    // probes of this query: rows of the global arrays, or selected here and kept in LDS
    const int *prow_list = a.probe_list + (size_t)q * w;
    const float *prow_dc = a.probe_dc + (size_t)q * w;
    const u32 *prow_base = a.probe_base + (size_t)q * w;
    // LDS copy of the query's probes (768 B after the shared thresholds).  w <= 32: list, coarse distance, visit-order
    // base, length and code offset (in 256-B units) of every probe, 32 entries each, so a round's bookkeeping never
    // waits on global memory; the last 128 B are scratch for select_row_short.  32 < w <= 64 (fused top-w only):
    // list, distance and base, 64 entries each.
    const bool cached = w <= 32;
    const int PW = cached ? 32 : 64;
    int *s_list = (int *)(L.sthr + STHR_WORDS * PG);
    float *s_dc = (float *)(s_list + PW);
    u32 *s_base = (u32 *)(s_dc + PW);
    u32 *s_len = s_base + PW;       // cached only
    u32 *s_coff = s_len + PW;       // cached only
    if (a.cdist) {
        const int Ksel = a.approx ? approx_pool(w) : w;
        WSel<true> ws;
        ws.init(KEY_MAX, nullptr, 64, Ksel);
        const float *row = a.cdist + (size_t)q * ix.kc;
        __syncthreads();                                 // L.sthr[0] = KEY_MAX is visible: it is the shared bound of this phase
        const u64 tp0 = STAMP();
        tpro[0] = tp0;
        bool have = false;   // uniform over the workgroup
        if constexpr (LB && M > 16) {
            // rows of up to 8192 scores through the short-row selection (16 / 32 keys per lane: this kernel has the registers)
            if (Ksel <= SHORT_ROW_MAXK && ix.kc > 2048 && ix.kc <= 8192 && (ix.kc & 3) == 0) {
                u64 *wbound = (u64 *)(s_list + 160);
                u32 *ccnt = (u32 *)(s_list + 168);
                if (ix.kc <= 4096)
                    have = a.approx ? select_row_short<true, 4>(ws, row, ix.kc, Ksel, wv, lane, tid, L.xch, wbound, ccnt)
                                    : select_row_short<false, 4>(ws, row, ix.kc, Ksel, wv, lane, tid, L.xch, wbound, ccnt);
                else
                    have = a.approx ? select_row_short<true, 8>(ws, row, ix.kc, Ksel, wv, lane, tid, L.xch, wbound, ccnt)
                                    : select_row_short<false, 8>(ws, row, ix.kc, Ksel, wv, lane, tid, L.xch, wbound, ccnt);
                tph[4] = STAMP() - tp0;
                tpro[1] = tpro[2] = tpro[3] = STAMP();
            }
        }
        if (!have && Ksel <= SHORT_ROW_MAXK && ix.kc <= 2048 && (ix.kc & 3) == 0) {
            // scratch: candidates in the exchange area; bounds + counter in the 128 B behind the five probe arrays
            // (w <= Ksel <= SHORT_ROW_MAXK < 32: the 32-entry layout), which nothing else writes, so late readers are safe
            u64 *wbound = (u64 *)(s_list + 160);
            u32 *ccnt = (u32 *)(s_list + 168);
            have = a.approx ? select_row_short<true>(ws, row, ix.kc, Ksel, wv, lane, tid, L.xch, wbound, ccnt)
                            : select_row_short<false>(ws, row, ix.kc, Ksel, wv, lane, tid, L.xch, wbound, ccnt);
            tph[4] = STAMP() - tp0;
            tpro[1] = tpro[2] = tpro[3] = STAMP();
        }
        if constexpr (M > 16) {   // wide codes only: in the m = 8 / 16 kernels this path costs 40 VGPRs and a workgroup per CU
        if (!have && a.approx && a.rf.tmin != nullptr) {   // uniform
            // behind the MFMA filter the tile minima are there: wave 0 reads the ~Ksel tiles that can matter instead of
            // four waves streaming the row and merging (HD-shape: 39 k -> 27 k cycles)
            int *flag = s_list + 170;                      // scratch word in the last 128 B of the probe area
            if (wv == 0) {
                const bool ok = select_row_tiled(ws, row, a.rf.tmin + (size_t)q * a.rf.ntiles, a.rf.ntiles, a.rf.tile_w, ix.kc, Ksel, lane,
                                                 (int *)L.xch);
                if (lane == 0) *flag = ok ? 1 : 0;
            }
            __syncthreads();
            have = *flag != 0;
            tph[4] = STAMP() - tp0;
            tpro[1] = tpro[2] = tpro[3] = STAMP();
        }
        }
        if (!have) {
            if (a.approx) select_row<true, 4>(ws, row, ix.kc, Ksel, wv, lane, L.sthr);
            else select_row<false, 4>(ws, row, ix.kc, Ksel, wv, lane, L.sthr);
            tph[4] = STAMP() - tp0;
            const int wc = ws.finish(Ksel, lane);
            ws.store(L.xch + (size_t)wv * 64, wc, lane);     // the exchange area aliases the (not yet built) tables
            if (lane == 0) L.scnt[wv] = wc;
            tpro[1] = STAMP();
            __syncthreads();
            tpro[2] = STAMP();
        }
        int fc = 0;
        if (wv == 0) {
            if (lane == 0) arm_bound<1>(L.sthr, 0, KEY_MAX);   // re-armed for the scan (published by the barrier below)
            if (!have) merge_waves(ws, L.xch, (size_t)64, L.scnt, 1, Ksel, KEY_MAX, 0, lane);
            tpro[3] = STAMP();
            fc = ws.finish(Ksel, lane);                 // == min(Ksel, kc)
        }
        if (a.approx) {   // uniform; every wave helps to stage the candidates' rows (the table area is still free)
            ws = refine_probes_wg<(M > 16)>(ws, fc, w, q, a.rf, wv, lane, tid, L.tab, stage_floats, s_list);
            if (wv == 0) fc = ws.finish(w, lane);       // == w
        }
        if (wv == 0) {
            u32 len = 0;
            int l = 0;
            if (lane < fc) {
                l = (int)(u32)ws.top;
                len = ix.list_len[l];
            }
            u32 incl = len;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            if (lane < fc) {
                s_list[lane] = l;
                s_dc[lane] = __uint_as_float((u32)(ws.top >> 32));
                s_base[lane] = incl - len;
                if (cached) {
                    s_len[lane] = len;
                    s_coff[lane] = (u32)(ix.list_codeoff[l] >> 8);   // code blocks are 256-B aligned
                }
            }
            const u32 total = __shfl(incl, 63);
            if (lane == 0) atomicAdd(a.scanned_points + (size_t)(q & 63) * 8, (u64)total);
            tpro[4] = STAMP();
        }
        __syncthreads();
        tpro[5] = STAMP();
        prow_list = s_list;
        prow_dc = s_dc;
        prow_base = s_base;
    } else if (cached) {
        if (tid < w) {
            const int l = prow_list[tid];
            s_list[tid] = l;
            s_dc[tid] = prow_dc[tid];
            s_base[tid] = prow_base[tid];
            s_len[tid] = ix.list_len[l];
            s_coff[tid] = (u32)(ix.list_codeoff[l] >> 8);
        }
        __syncthreads();
        prow_list = s_list;
        prow_dc = s_dc;
        prow_base = s_base;
    }
    // Residual inputs of the NEXT round are fetched into registers before a round's scan and written to LDS after
    // it, so their latency hides behind the scan and a round needs two barriers, not three.  (d * PG <= 512 only;
    // wider rows -- where the table build dwarfs everything else -- take the plain three-barrier round.)
    if constexpr (LB) {
        lb_rounds<M, DS, PG>(ix, a.lb, a.queries, smem_raw, q, w, K, a.prune, a.scanned_points, sel[0], L.sthr, s_list, s_dc, s_base, s_len, s_coff, wv,
                             lane, tid, a.dbg);
    } else {
    constexpr int RU = (M > 0 && M * DS * PG <= 256) ? 1 : 2;
    constexpr bool can_pipe = M == 0 || M * DS * PG <= 256 * RU;   // statically out for wide rows: no dead state in their loops
    const bool pipe = can_pipe && cached && ix.d * PG <= 256 * RU;
    float rq[RU], rc[RU];
    auto resid_fetch = [&](int j0) {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int e = tid + u * 256;
            rq[u] = 0.0f;
            rc[u] = 0.0f;
            if (e < ix.d * PG) {
                const int i = e / PG, sl = e - i * PG;
                const int pj = (j0 + sl) < w ? j0 + sl : j0;
                rq[u] = a.queries[(size_t)q * ix.d + i];
                rc[u] = ix.centroids[(size_t)prow_list[pj] * ix.d + i];
            }
        }
    };
    auto resid_store = [&]() {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int e = tid + u * 256;
            if (e < ix.d * PG) L.resid[e] = rq[u] - rc[u];
        }
    };
    if (pipe) {
        resid_fetch(0);
        resid_store();
    }
    // Probes per round (uniform): PG, except that the FIRST round of a pruning query scans the closest cell alone when the second cell lies
    // far behind it (dc[1] > 2 dc[0]).  A fresh selector prunes nothing, so a first round of two probes builds and scans both -- and on
    // clustered data the bound the closest cell leaves behind prunes every other probe (SIFT1M shape: 4.0 probes per query scanned with
    // pairs from the start, 1.25 one at a time, profiles/r05_pg_probe.txt).  Selection is order-free and the pruning rule is exact for any
    // round structure, so results do not change.  On distance-concentrated data (ratios near 1) every round is a pair, as before.
    int np = PG;
    for (int j0 = 0; j0 < w; j0 += np) {
        np = PG;
        // the PG probes of this round, in rank order (uniform values)
        int li[PG], qi[PG];
        u32 len[PG], sb[PG];
        float dcv[PG];
        const uint8_t *cb[PG];
#pragma unroll
        for (int s = 0; s < PG; ++s) {
            const bool ok = (j0 + s) < w;
            const int pj = ok ? j0 + s : j0;
            li[s] = prow_list[pj];
            qi[s] = q;
            dcv[s] = prow_dc[pj];
            sb[s] = prow_base[pj];
            if (cached) {
                len[s] = ok ? s_len[pj] : 0u;
                cb[s] = ix.codes + ((size_t)s_coff[pj] << 8);
            } else {
                len[s] = ok ? ix.list_len[li[s]] : 0u;
                cb[s] = ix.codes + ix.list_codeoff[li[s]];
            }
        }
        CodeRegs<M, ppl_of<M, 1>()> cr[PG];
#pragma unroll
        for (int s = 0; s < PG; ++s) scan_prefetch(cr[s], cb[s], 0u, len[s], wv, lane);   // in flight while the tables are built
        const u64 t0 = STAMP();
        __syncthreads();          // every wave is done with the previous round's tables (and has written this round's residuals)
        const u64 t1 = STAMP();
        // Exact pruning.  A point's sum starts from its list's coarse distance and only grows (index.jl:242-244: every table
        // entry is a sum of squares, >= +0), so no point of a list whose dc lies above the K-th best key found so far can
        // enter the result; probes come in ascending dc (coarsequantizers.jl:35-36), so the first such list ends the
        // query.  The bound is the workgroup-shared one (nothing writes it between the barrier above and the next scan, so
        // the decision is uniform); strict comparison of the high words: a key with an equal distance may still win on
        // visit order.
        if (a.prune) {
            const u32 thi = (u32)(readfirstlane64(L.sthr[0]) >> 32);
            if (__float_as_uint(dcv[0]) > thi) {
                if (tid == 0) {
                    u64 skipped = 0;
                    for (int pj = j0; pj < w; ++pj) skipped += cached ? s_len[pj] : ix.list_len[prow_list[pj]];
                    atomicAdd(a.scanned_points + (size_t)(q & 63) * 8 + 1, skipped);
                }
                break;
            }
#pragma unroll
            for (int s = 1; s < PG; ++s)
                if (__float_as_uint(dcv[s]) > thi && len[s] != 0) {   // uniform: a later probe of this round alone
                    if (tid == 0) atomicAdd(a.scanned_points + (size_t)(q & 63) * 8 + 1, (u64)len[s]);
                    len[s] = 0;
                }
            if constexpr (PG == 2 && SMALL) {   // (K > 64: the LDS selectors' bound lags a round behind; measured -4 % there)
                if (j0 == 0 && w > 1 && len[1] != 0 && dcv[1] > 2.0f * dcv[0]) {   // uniform: the closest cell alone first
                    len[1] = 0;
                    np = 1;
                }
            }
        }
        const int ns = (PG == 2 && len[1 % PG] == 0) ? 1 : PG;   // tables this round needs
        constexpr bool PKB = M == 48 && DS == 16 && PG == 1;
        const bool pkb = PKB && ix.codebooks_p != nullptr;   // uniform
        if (!pipe) {
            if (pkb) {
                // pair-interleaved residuals for build_tables_pk: [(p * DS + t) * 2 + h] = r[(2p + h) * DS + t]
                for (int i = tid; i < ix.d; i += 256) {
                    const int ii = i / DS, t = i - ii * DS;
                    L.resid[((ii >> 1) * DS + t) * 2 + (ii & 1)] = a.queries[(size_t)q * ix.d + i] - ix.centroids[(size_t)li[0] * ix.d + i];
                }
            } else {
                build_residuals<PG>(ix, a.queries, qi, li, L.resid, tid);
            }
            __syncthreads();
        }
        const u64 t2 = STAMP();
        if constexpr (M == 48 && DS == 16) {
            if (pkb) build_tables_pk<DS>(ix, m, L.resid, L.tab, tid);
            else build_tables_deep<PG, DS, 4>(ix, m, L.resid, L.tab, tid);   // registers to spare: four stages in flight
        } else {
            if constexpr (PG == 2 && DS > 0) {
                if (ns == 1) build_tables_t<PG, DS, TAB_SEP, 0, true>(ix, m, L.resid, L.tab, tid);   // uniform
                else build_tables_t<PG, DS, TAB_SEP>(ix, m, L.resid, L.tab, tid);
            } else build_tables_t<PG, DS, TAB_SEP>(ix, m, L.resid, L.tab, tid);
        }
        __syncthreads();
        const u64 t3 = STAMP();
        const bool more = pipe && (j0 + np) < w;
        if (more) resid_fetch(j0 + np);
        __builtin_amdgcn_s_setprio(3);   // see scan_kernel
        if constexpr (PG == 2 && M > 0 && M <= 16 && SMALL) {   // wider codes / LDS selectors: the extra live state costs a wave per SIMD
            scan_pair<M>(0u, (u32)M * 1024u, cb[0], cb[1], len[0], len[1], dcv[0], dcv[1], sb[0], sb[1], sel, K, wv, lane, cr[0], cr[1],
                         L.sthr, ix.dbg_flags);
        } else {
#pragma unroll
            for (int s = 0; s < PG; ++s) {
                if (len[s] == 0) continue;   // uniform
                const float dc1[1] = {dcv[s]};
                const u32 sb1[1] = {sb[s]};
                scan_range<M, 1>(L.tab + (size_t)s * m * 256, (u32)s * (u32)m * 1024u, cb[s], ix.cs, m, 0u, len[s], dc1, sb1, 1, sel, K, wv,
                                 lane, cr[s], L.sthr, ix.dbg_flags);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (more) resid_store();   // the table build of this round is behind the barrier above: the buffer is free
        const u64 t4 = STAMP();
        tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2; tph[3] += t4 - t3;
    }
    }
    const u64 tloop = STAMP();
    const int mycnt = sel[0].finish(K, lane);
    __syncthreads();              // exchange area aliases the table
    sel[0].store(L.xch + (size_t)wv * L.xcap, mycnt, lane);
    if (lane == 0) L.scnt[wv] = mycnt;
    __syncthreads();
    if (wv == 0) {
        merge_waves(sel[0], L.xch, (size_t)L.xcap, L.scnt, 1, K, KEY_MAX, 0, lane);
        const int fc = sel[0].finish(K, lane);
        sel[0].for_each(fc, lane, [&](int i, u64 key) {
            emit_result(key, i, q, w, K, prow_list, prow_base, ix.list_pos, ix.ids, a.out_ids, a.out_dists);
        });
        if (lane == 0) a.out_counts[q] = fc;
    }
#ifdef IVFADC_DEBUG
    if (a.dbg && tid == 0) {
        const u64 tend = STAMP();
        u64 *o = a.dbg + (size_t)q * 16;
        if constexpr (!LB) { o[0] = tph[0]; o[1] = tph[1]; o[2] = tph[2]; o[3] = tph[3]; }
        o[4] = tloop - tstart; o[5] = tend - tloop; o[6] = tph[4]; o[7] = tend;
        o[8] = tpro[0] - tstart; o[9] = tpro[1] - tpro[0]; o[10] = tpro[2] - tpro[1]; o[11] = tpro[3] - tpro[2];
        o[12] = tpro[4] - tpro[3]; o[13] = tpro[5] - tpro[4];
        if constexpr (!LB) { o[14] = 0; o[15] = 0; }
    }
#endif
}

template <int M, int DS, int PG, bool SMALL, bool LB = false>
#ifndef IVFADC_QSCAN_MINW
#define IVFADC_QSCAN_MINW 1
#endif
__global__ __launch_bounds__(256, LB ? (M <= 16 ? 3 : (PG >= 4 ? 2 : (PG == 3 ? 3 : 4))) : ((M == 8 && SMALL && PG <= 2) ? IVFADC_QSCAN_MINW : 1)) void qscan_kernel(const QScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    qscan_body<M, DS, PG, SMALL, LB>(a, smem_raw, (int)blockIdx.x);
}

// The query-major scan of one batch with the exact small-problem coarse search of the NEXT batch riding behind it in the same grid:
// workgroups [0, a.nq) scan, the rest each take one tile of the next batch's coarse distances as the scanning workgroups retire.
#ifndef IVFADC_RIDER_QW
#define IVFADC_RIDER_QW 4
#endif
constexpr int RIDER_QW = IVFADC_RIDER_QW;   // queries per wave of a rider tile (a tile: 64 centroids x 4 RIDER_QW queries)
struct CoarseNext {
    const float *queries;   // next batch
    float *out;             // its [nq][kc] distance rows
    int nq, ncx;            // ncx = ceil(kc / 64) tiles per 4 RIDER_QW queries
};
template <int M, int DS, int PG>
__global__ __launch_bounds__(256) void qscan_coarse_kernel(const QScanArgs a, const CoarseNext cn)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if ((int)blockIdx.x < a.nq) {
        qscan_body<M, DS, PG, true, false>(a, smem_raw, (int)blockIdx.x);
    } else {
        const int t = (int)blockIdx.x - a.nq;
        coarse_sgpr_tile<RIDER_QW>(cn.queries, a.ix.centroids, cn.out, cn.nq, a.ix.kc, a.ix.d, t % cn.ncx, t / cn.ncx, (float *)smem_raw);
    }
}

#include "smallq.hip.h"

// ---------------------------------------------------------------------------------------
// push! path (utils.jl:148-161): given the nearest centroid of each point, quantize the
// residual: per sub-space the codeword with the smallest SqEuclidean distance, first minimum
// on ties.  One workgroup per point.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void encode_kernel(const float *__restrict__ pts, const int *__restrict__ assign, int d, int m,
                                                     int ksub, int dsub, const float *__restrict__ centroids,
                                                     const float *__restrict__ codebooks, const uint8_t *__restrict__ labels,
                                                     uint8_t *__restrict__ out_codes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *resid = (float *)smem_raw;                               // [d]
    u64 *best = (u64 *)(resid + (((size_t)d + 3) & ~(size_t)3));    // [m] (dist bits << 32 | codeword)
    const int p = blockIdx.x, tid = threadIdx.x;
    const int l = assign[p];
    for (int i = tid; i < d; i += 256) resid[i] = pts[(size_t)p * d + i] - centroids[(size_t)l * d + i];
    for (int i = tid; i < m; i += 256) best[i] = KEY_MAX;
    __syncthreads();
    for (int e = tid; e < m * 256; e += 256) {
        const int ii = e >> 8, c = e & 255;
        u64 key = KEY_MAX;
        if (c < ksub) {
            const float *cw = codebooks + ((size_t)ii * ksub + c) * dsub;
            const float *rr = resid + (size_t)ii * dsub;
            float sum = 0.0f;
            for (int t = 0; t < dsub; ++t) {
                const float df = cw[t] - rr[t];
                sum = sum + df * df;
            }
            key = make_key(sum, (u32)c);
        }
        // wave min, then one LDS atomic per wave (ii is wave-uniform: 256 % 64 == 0)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 o = __shfl_xor(key, off);
            key = o < key ? o : key;
        }
        if ((tid & 63) == 0) atomicMin(&best[ii], key);
    }
    __syncthreads();
    for (int i = tid; i < m; i += 256) out_codes[(size_t)p * m + i] = labels[i * ksub + (int)(u32)best[i]];
}

// argmin over one row of coarse distances, first minimum on ties (coarse_search(cq, p, 1)).
__global__ __launch_bounds__(256) void argmin_rows_kernel(const float *__restrict__ cdist, int n, int kc, int *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= n) return;
    const float *row = cdist + (size_t)r * kc;
    u64 key = KEY_MAX;
    for (int c = lane; c < kc; c += 64) {
        const u64 k2 = make_key(row[c], (u32)c);
        key = k2 < key ? k2 : key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_xor(key, off);
        key = o < key ? o : key;
    }
    if (lane == 0) out[r] = (int)(u32)key;
}

// ---------------------------------------------------------------------------------------
// Synthetic code bytes, written straight into the device layout (bench / test utility).
// One thread per dword of a list's code block.
// ---------------------------------------------------------------------------------------
static __device__ __forceinline__ u64 mix64(u64 x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

__global__ __launch_bounds__(256) void synth_codes_kernel(uint8_t *__restrict__ codes, const int64_t *__restrict__ list_pos,
                                                          const u32 *__restrict__ list_len,
                                                          const int64_t *__restrict__ list_codeoff, int kc, int m, int cs,
                                                          u64 seed)
{
    const int l = blockIdx.x;   // grid.x = lists (kc may exceed the 65535 limit of grid.y)
    const int64_t lpos = list_pos[l];   // synthetic lists have no spare capacity: id offset == canonical global position
    const int64_t len = list_len[l];
    const int64_t ndw = len * (cs >> 2);
    u32 *dst = (u32 *)(codes + list_codeoff[l]);
    const int dpp = cs >> 2;   // dwords per point
    for (int64_t t = (int64_t)blockIdx.y * 256 + threadIdx.x; t < ndw; t += (int64_t)gridDim.y * 256) {
        const int64_t p = t / dpp;
        const int wd = (int)(t - p * dpp);
        u32 out = 0;
        u64 prev_w = ~0ull, h = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ii = wd * 4 + b;
            if (ii < m) {
                const u64 B = (u64)(lpos + p) * (u64)m + (u64)ii;
                const u64 wi = B >> 3;
                if (wi != prev_w) { h = mix64(seed + wi * 0x9E3779B97F4A7C15ull); prev_w = wi; }
                out |= (u32)((h >> (8 * (B & 7))) & 0xff) << (8 * b);
            }
        }
        dst[t] = out;
    }
}

// ---------------------------------------------------------------------------------------
// push! on the device (utils.jl:127-145: push!(list.idxs, id); push!(list.codes, code)): every new point is written
// into the spare capacity behind its list.  dst[2p] = byte offset of the code slot, dst[2p+1] = slot in the id array
// (both computed by the host, which owns the list lengths).  One thread per point.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void append_scatter_kernel(int64_t nnew, int m, const int64_t *__restrict__ dst,
                                                             const uint8_t *__restrict__ new_codes, const u32 *__restrict__ new_ids,
                                                             uint8_t *__restrict__ codes, u32 *__restrict__ ids)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= nnew) return;
    uint8_t *c = codes + dst[2 * p];
    const uint8_t *src = new_codes + (size_t)p * m;
    for (int i = 0; i < m; ++i) c[i] = src[i];
    ids[dst[2 * p + 1]] = new_ids[p];
}

// ---------------------------------------------------------------------------------------
// delete_from_index! / pop! / popfirst! on the device (utils.jl:41-68, 90-105): one workgroup per list removes the
// entries whose id is in the sorted array `rem` (stable, in place) and lowers every surviving id by the number of
// removed ids below it (_shift_inverse_index!).  In-place safety: a chunk of 256 points is handled one dword column
// at a time -- all threads read the column, barrier, the survivors write it to their new slot (never behind the
// read cursor), barrier -- so no slot is overwritten before it has been read.
// ---------------------------------------------------------------------------------------
static __device__ __forceinline__ u32 lower_bound_u32(const u32 *a, u32 n, u32 v)
{
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void delete_compact_kernel(const u32 *__restrict__ rem, u32 nrem, const int64_t *__restrict__ list_pos,
                                                             const int64_t *__restrict__ list_codeoff, u32 *__restrict__ list_len,
                                                             uint8_t *__restrict__ codes, u32 *__restrict__ ids, int cs)
{
    __shared__ u32 s_wave[4];
    __shared__ u32 s_wr;
    const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 len = list_len[l];
    u32 *lid = ids + list_pos[l];
    u32 *lcode = (u32 *)(codes + list_codeoff[l]);
    const int nw = cs >> 2;
    if (tid == 0) s_wr = 0;
    __syncthreads();
    for (u32 c0 = 0; c0 < len; c0 += 256) {
        const u32 p = c0 + tid;
        const bool in = p < len;
        u32 id = in ? lid[p] : 0u;
        const u32 lb = in ? lower_bound_u32(rem, nrem, id) : 0u;
        const bool keep = in && !(lb < nrem && rem[lb] == id);
        const u64 mask = __ballot(keep);
        if (lane == 0) s_wave[wv] = (u32)__popcll(mask);
        __syncthreads();                                   // also: every id of the chunk has been read
        u32 before = 0;
        for (int v = 0; v < wv; ++v) before += s_wave[v];
        const u32 total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        const u32 dst = s_wr + before + (u32)__popcll(mask & ((1ull << lane) - 1ull));
        if (keep) lid[dst] = id - lb;
        for (int wd = 0; wd < nw; ++wd) {
            const u32 v = in ? lcode[(size_t)p * nw + wd] : 0u;
            __syncthreads();
            if (keep) lcode[(size_t)dst * nw + wd] = v;
            __syncthreads();
        }
        if (tid == 0) s_wr += total;
        __syncthreads();
    }
    if (tid == 0) list_len[l] = s_wr;
}

// _shift_up_inverse_index! (pushfirst!, utils.jl:1-9): every stored id moves by `delta`
__global__ __launch_bounds__(256) void shift_ids_kernel(u32 *__restrict__ ids, int64_t nslots, u32 delta)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nslots; i += (int64_t)gridDim.x * 256) ids[i] += delta;
}

__global__ void fill_u64_kernel(u64 *p, size_t n, u64 v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// Query ingest of the host-pointer entries (ivfadc_search, ivfadc_search_batches): the rows are read straight from page-locked host
// memory -- the caller's own array when it is registered (ivfadc_host_register / ivfadc_host_alloc), the library's staging buffer
// otherwise -- by the compute queue.  A copy-engine transfer in front of the first kernel costs the chain of a blocking call ~10 us more
// than this launch does (tools/micro/host_path.hip: hand-off between the SDMA queue and the compute queue).
// nvec 16-byte groups when both pointers are 16-byte aligned (nvec = 0 otherwise), then the remaining dwords.
__global__ __launch_bounds__(256) void host_ingest_kernel(const u32 *__restrict__ src, u32 *__restrict__ dst, size_t nvec, size_t nwords)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const uint4 *s4 = (const uint4 *)src;
    uint4 *d4 = (uint4 *)dst;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) d4[i] = s4[i];
    for (size_t i = nvec * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) dst[i] = src[i];
}

// busy for `us` microseconds of the constant 100 MHz device clock: the probe of stream_overlap() (do two streams run side by side?)
__global__ void spin_us_kernel(unsigned us)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) { }
}

}  // namespace ivf
