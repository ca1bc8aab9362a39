// wg8q8scan.hip.h -- the eight-wave list-major scan (wg8scan.hip.h) with EIGHT queries per code stream.  Included by kernels.hip.h behind
// wg8scan.hip.h (its constants, W8_* knobs and W8Prof macros are shared), namespace ivf.
//
// Reference: src/coarsequantizers.jl:40-45 (residuals), src/index.jl:232-236 (table build), :240-246 (scan), :247-254 (bounded top-K).
//
// wg8_scan_kernel is bound by instruction issue (vector ALU 63 % + LDS instructions 18 % of the SIMD cycles, and the two
// add): a point costs 8 address perms + 8 gathers + 8 three-operand adds per FOUR queries.  With 16-byte table entries -- eight 16-bit
// fields -- the same perm and ONE ds_read_b128 serve EIGHT queries, 16 adds: 32 instructions per point and eight queries instead of 48.
// The table keeps its 64 KB: a code's row is 2 copies x 8 sub-quantizers x 16 B = 256 B, lane l reads sub-quantizer (t + l) mod 8 in copy
// (l / 16) mod 2 -- the four 16-lane service groups of a ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, and the same + 32:
// MI355X_MICROARCH.md) each see 16 different four-bank groups.  Work items are (list, group of <= 8 queries, chunk): half as many table
// builds, set-ups and hand-overs per probed list.  Everything else -- the f32 tables in device memory (64 KB per workgroup here), parked
// candidates and passes, the workgroup pool, bounds from the integer sums, one work queue per XCD -- is wg8scan.hip.h's, eight slots wide.
// The plan takes this kernel where a list is probed by ten queries or more on average.
#pragma once

constexpr u32 W9_TAB_BYTES = 256u * 256u;     // 256 codes x (2 copies x 8 sub-quantizers x 16 bytes)
constexpr u32 W9_GTAB_FLOATS = 8u * 256u * 8u;   // f32 tables of a work item in device memory: [ii][label][8 queries]

struct W9Lds {
    static constexpr u32 RES = W9_TAB_BYTES;                  // f32 residuals [ii][t][s]: 8 x (16 x 8 + 8 of padding) x 4 B (a sub-quantizer's block starts 8 banks on)
    static constexpr u32 SMAX = RES + 4352u;                  // u32 [8]: bits of the per-query largest entry (atomicMax); f32 inv[8] behind
    static constexpr u32 QC = SMAX + 64u;                     // f32 dc[8]; u32 visit-order base[8]; u32 probe index[8]; u32 query[8]
    static constexpr u32 HARD = QC + 128u;                    // u64 [8]: the bounds the item started from
    static constexpr u32 STHR = HARD + 64u;                   // u64 [8]: workgroup-shared bounds
    static constexpr u32 SWI = STHR + 64u;                    // u32 [4]
    static constexpr u32 POOL = SWI + 16u;                    // u64 [8][64]: the workgroup's K smallest keys per slot, unordered (w9_pool_offer)
    static constexpr u32 PARK = POOL + 8u * 64u * 8u;         // u32 [8][32][W8_ES]: the waves' rings of parked points (W8_RING)
    static constexpr u32 COLD = PARK + (u32)W8_NW * 32u * W8_ES * 4u;     // u32 [8][16]: a cold work item's first step, every wave's ceil(K / 8)-th smallest integer sum per slot
    static constexpr u32 END = COLD + 512u;
};
static_assert(W9Lds::END <= 80u * 1024u, "two workgroups per CU");
static_assert((W9Lds::HARD & 7u) == 0 && (W9Lds::STHR & 7u) == 0 && (W9Lds::POOL & 7u) == 0, "8-byte bounds");

typedef W8Prof W9Prof;

static __device__ __forceinline__ u32 w9_perm(u32 s0, u32 s1, u32 sel)
{
    u32 o;
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(o) : "v"(s0), "v"(s1), "s"(sel));
    return o;
}

// the work item's per-slot constants stand in LDS at fixed addresses (the kernel owns the whole allocation: the dynamic segment starts
// at address 0); the rare paths read them there instead of holding two dozen scalars across the scan loop
template <class T> static __device__ __forceinline__ T w9_lds(u32 byte_addr) { return lds_load_abs<T>(byte_addr); }
// a pointer into the workgroup's dynamic LDS segment (for stores and atomics: an integer cast to a generic pointer is NOT an LDS address)
template <class T> static __device__ __forceinline__ T *w9_ptr(u32 byte_off)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char w9_smem[];
    return (T *)(w9_smem + byte_off);
}
static __device__ __forceinline__ float w9_dc(int s) { return __uint_as_float(__builtin_amdgcn_readfirstlane(w9_lds<u32>(W9Lds::QC + 4u * s))); }
static __device__ __forceinline__ u32 w9_sbase(int s) { return __builtin_amdgcn_readfirstlane(w9_lds<u32>(W9Lds::QC + 32u + 4u * s)); }
static __device__ __forceinline__ float w9_inv(int s) { return __uint_as_float(__builtin_amdgcn_readfirstlane(w9_lds<u32>(W9Lds::SMAX + 32u + 4u * s))); }
static __device__ __forceinline__ u64 w9_sthr(int s) { return readfirstlane64(w9_lds<u64>(W9Lds::STHR + 8u * s)); }

// The accumulator BIAS of the eight queries (w8_bias, eight fields in four dwords): lane s (mod 8) evaluates slot s.
static __device__ __forceinline__ void w9_bias(int nvalid, u32 (&bias)[4])
{
    const u32 sl = (u32)lane_id() & 7u;
    const u32 th = w9_lds<u32>(W9Lds::STHR + 8u * sl + 4u);
    const float dc = w9_lds<float>(W9Lds::QC + 4u * sl);
    const float inv = w9_lds<float>(W9Lds::SMAX + 32u + 4u * sl);
    u32 T = 0x7FFFu;
    if (th < 0x7F800000u) {   // a finite bound
        const float thr = __uint_as_float(th);
        const float x = (thr * 1.0000038146972656f - dc) * inv * 1.0000038146972656f;   // (1 + 2^-18): as qf_targets
        T = x < 0.0f ? 0u : (x < 32000.0f ? (u32)x + 2u : 0x7FFFu);
    }
    const u32 B = (int)sl < nvalid ? 0x7FFFu - T : 0x8000u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        bias[i] = (u32)__builtin_amdgcn_readlane((int)B, 2 * i) | ((u32)__builtin_amdgcn_readlane((int)B, 2 * i + 1) << 16);
}

// ---- the workgroup's selection: ONE pool of K keys per slot in LDS, shared by the eight waves ---------------------------------------------
// (index.jl:247-254: the bounded heap of a query.)  Eight per-wave selectors bound the union's K-th key only loosely -- a wave's own K-th
// key is the K-th of an eighth of the points, and max over the waves of their ceil(K / 8)-th keys sits near rank 3.6 K of what the workgroup
// has seen (the 2nd-order statistics' maximum) -- and every candidate the looser bound lets through costs an exact sum and a trip to L2:
// at w = 1, where every work item starts cold, the candidate path was a third of the kernel (knock-out build: 1.65 -> 1.09 ms).  The pool
// is the exact thing: its largest entry IS the K-th smallest key of everything the workgroup has offered.
//   pool[s][0 .. K): unordered, KEY_MAX = empty.  An offer x reads the K entries (one per lane), takes their maximum mx; x >= mx: K keys
//   below x exist, x is out.  Else lane 0 swaps x in for mx (compare-and-swap: another wave may have replaced that entry meanwhile -- an
//   entry only ever DECREASES, so a failed swap means progress elsewhere and the offer starts again; no ABA).  Dropping mx is safe: at the
//   moment of the swap the other K - 1 entries are at or below their snapshot values, all below mx, and so is x.  Keys are unique, so the
//   K smallest keys of all offers are never refused and never dropped: the pool ends as the exact top K in any interleaving (ids and
//   distances bit-identical to the oracle); the order is restored by one 64-lane sort when the work item is done.
//   The maximum of ANY snapshot -- K distinct keys that were offered -- is an upper bound of the K-th key: published with atomicMin.
static __device__ __forceinline__ u32 w9_row_max_u32(u32 x)
{
    // running maximum along each row of 16 lanes (row_shr 1, 2, 4, 8: a lane without a source reads 0), rows' last lanes -> scalar unit
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
    const u32 a = __builtin_amdgcn_readlane(x, 15), b = __builtin_amdgcn_readlane(x, 31), c = __builtin_amdgcn_readlane(x, 47), d = __builtin_amdgcn_readlane(x, 63);
    const u32 ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
static __device__ __forceinline__ u64 w9_wave_max_u64(u64 v)
{
    const u32 hi = (u32)(v >> 32), lo = (u32)v;
    const u32 mh = w9_row_max_u32(hi);
    const u32 ml = w9_row_max_u32(hi == mh ? lo : 0u);
    return ((u64)mh << 32) | ml;
}
// the pool's entries of slot s, one per lane (lanes >= K: 0, below every key)
static __device__ __forceinline__ u64 w9_pool_read(int s, int K, int lane)
{
    return lane < K ? w9_lds<u64>(W9Lds::POOL + 512u * (u32)s + 8u * (u32)lane) : 0ull;
}
// Offers the keys of the lanes in `mask` (uniform, non-empty) to slot s, starting from the snapshot v the caller read a while ago.  A stale
// snapshot is as good as a fresh one for every decision above -- each of its values WAS that entry's, entries only decrease -- it merely
// fails a swap more often, and a failed swap returns the entry's value of the moment: the snapshot is patched and the offer goes on
// without another read.  An offer costs one LDS round trip per swap attempt (under the scan's gathers a round trip is several hundred
// cycles: the dependent trips, not the instructions, were the cost of a pass).  Returns the slot's new bound, KEY_MAX if nothing went in
// or the pool is not full.
static __device__ __forceinline__ u64 w9_pool_offer(int s, u64 v, u64 key, u64 mask, int K, int lane)
{
    bool any = false;
    u64 *pool = w9_ptr<u64>(W9Lds::POOL + 512u * (u32)s);
    u64 mx = w9_wave_max_u64(v);
    for (;;) {   // uniform
        mask &= __builtin_amdgcn_ballot_w64(key < mx);   // (every lane's key against the bound of this moment: most offers of a crowd end here)
        if (mask == 0) break;
        const int src = __builtin_ctzll(mask);
        const u64 x = readlane64(key, src);
        const int idx = __builtin_ctzll(__builtin_amdgcn_ballot_w64(lane < K && v == mx));
        u64 old = 0;
        if (lane == 0) old = atomicCAS((unsigned long long *)&pool[idx], (unsigned long long)mx, (unsigned long long)x);
        old = readfirstlane64(old);
        if (old == mx) {   // (uniform) x is in
            v = lane == idx ? x : v;
            any = true;
            mask &= mask - 1ull;
        } else {
            v = lane == idx ? old : v;   // another wave's key sits there now
        }
        mx = w9_wave_max_u64(v);
    }
    if (!any || mx == KEY_MAX) return KEY_MAX;
    if (lane == 0) atomicMin(w9_ptr<u64>(W9Lds::STHR + 8u * (u32)s), mx);
    return mx;
}

// ---- reference-order sums of parked points, 8 per pass, entries from the work item's f32 tables in device memory -------------------
// (scope: the tables were written by this workgroup before a barrier; the loads go to L2 -- sc1 -- so that no line of an earlier work
// item's tables can be served from this CU's vector cache)
static __device__ __forceinline__ v4f w9_gtab_load(__amdgpu_buffer_rsrc_t rs, u32 ii, u32 byte, int half)
{
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((((ii << 8) | byte) << 5) + 16u * (u32)half), 0, 16);
    return (v4f){__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}

// A pass in flight: lane 8 e + ii holds the f32 entries (four queries) of sub-quantizer ii at parked point e's code byte, and the point's
// position.  Requested when eight points are waiting (w9_pass_issue) and worked off at the top of the NEXT step, right behind the wait
// for that step's code bytes -- older than the gather -- so the trip to L2 costs the wave nothing (a pass worked off where it is
// requested waits for the code stream's request in flight AND its own: ~4 us per pass, measured).
struct W9Pass {
    v4f ev, ew;     // queries 0 .. 3, 4 .. 7
    u32 pos;
    bool ok;
};
static __device__ __forceinline__ void w9_pass_issue(W9Pass &ps, u32 cbuf_addr, int &head, int &cnt, __amdgpu_buffer_rsrc_t gt, int lane)
{
    const int seg = lane >> 3, ii = lane & 7;
    ps.ok = seg < cnt;
    const u32 ea = cbuf_addr + (u32)((head + (ps.ok ? seg : 0)) & (W8_RING - 1)) * (W8_ES * 4u);
    // (parked: the point's ROTATED code bytes -- out byte t = code byte (t + j) mod 8, j = the parking lane's rotation, kept in the position
    // word's top three bits: the scan loop holds no unrotated copy of a step's bytes)
    const u32 pj = w9_lds<u32>(ea + 8u);
    const u32 idx = ((u32)ii - (pj >> 29)) & 7u;
    const u32 dw = w9_lds<u32>(ea + 4u * (idx >> 2));
    ps.pos = pj & 0x1FFFFFFFu;
    ps.ev = w9_gtab_load(gt, (u32)ii, (dw >> (8 * (idx & 3))) & 0xffu, 0);
    ps.ew = w9_gtab_load(gt, (u32)ii, (dw >> (8 * (idx & 3))) & 0xffu, 1);
    const int take = cnt < 8 ? cnt : 8;
    head = (head + take) & (W8_RING - 1);
    cnt -= take;
}

// Works a pass off.  (Measured and dropped: everything the pass needs from LDS -- constants, bounds, a snapshot of every slot's pool, the
// bias arithmetic's operands -- requested in one go ahead of the running sums, the bias computed from registers: 16 384 x w = 8
// 5.87 -> 5.91 ms, w = 1 1.38 -> 1.44.  The pass does not wait for memory -- W8_PROF: 260 of its 7 700 cycles -- it is ~400 dependent
// instructions on a SIMD it shares with three scanning waves.)
static __device__ __forceinline__ void w9_pass_finish(const W9Pass &ps, int nvalid, int K, int lane)
{
    const int ii = lane & 7;
    const float ev[8] = {ps.ev.x, ps.ev.y, ps.ev.z, ps.ev.w, ps.ew.x, ps.ew.y, ps.ew.z, ps.ew.w};
    float x[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) x[s] = w9_dc(s) + ev[s];
#pragma unroll
    for (int i = 1; i < 8; ++i)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // lane l <- lane l-1 within a row of 16 (row_shr:1): the segment's last lane ends with ((dc + t0) + t1) + ... + t7 (index.jl:242-246)
            const float up = __uint_as_float((u32)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x[s]), 0x111, 0xf, 0xf, false));
            x[s] = up + ev[s];
        }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s >= nvalid) continue;   // uniform
        const u64 key = make_key(x[s], w9_sbase(s) + ps.pos);
        const u64 mask = __builtin_amdgcn_ballot_w64(ps.ok && ii == 7 && key < w9_sthr(s));
        if (mask != 0) w9_pool_offer(s, w9_pool_read(s, K, lane), key, mask, K, lane);   // uniform
    }
}

// ---- the scan of one work item by one wave -----------------------------------------------------------------------------------------------
// K-th smallest integer sum of slot S over the step's four points per lane (radix select, as w9_kth_sum)
static __device__ __attribute__((noinline)) u32 w9_kth_sum4(u32 v0, u32 v1, u32 v2, u32 v3, u32 validbits, int need)
{
    u32 prefix = 0;
#pragma unroll 1
    for (int bit = 14; bit >= 0; --bit) {   // uniform
        const u32 want = prefix >> bit;
        const int c0 = __popcll(__builtin_amdgcn_ballot_w64((validbits & 1u) && (v0 >> bit) == want)) +
                       __popcll(__builtin_amdgcn_ballot_w64((validbits & 2u) && (v1 >> bit) == want)) +
                       __popcll(__builtin_amdgcn_ballot_w64((validbits & 4u) && (v2 >> bit) == want)) +
                       __popcll(__builtin_amdgcn_ballot_w64((validbits & 8u) && (v3 >> bit) == want));
        if (c0 < need) {
            need -= c0;
            prefix |= 1u << bit;
        }
    }
    return prefix;
}

static __device__ __forceinline__ void w9_scan_range(__amdgpu_buffer_rsrc_t codes, u32 p0, u32 p1, int nvalid, int K, int wv, int lane,
                                                     v4u ca, v4u cb, __amdgpu_buffer_rsrc_t gt, W9Prof &pr)
{
    // A step of a wave is 256 points: four per lane in two 16-byte registers sets, ca (points pb + 2 lane, + 1) and cb (pb + 128 + 2 lane,
    // + 1), requested by the caller for the first step.  The code stream comes through a buffer resource over the list: the lane's offset
    // (16 lane) is a constant register, the step's offset a scalar -- a request is ONE instruction and no address arithmetic -- and each half
    // of the NEXT step is requested into its register set the moment this step's half has left it (rotated, four v_perm): two requests
    // of 1 KB per wave are in flight at any time, each with a whole step to arrive, and there is no second register set and no move.
    // (One request per wave -- 4 MB on the chip -- at the loaded latency of HBM is 2 TB/s: the conflict-free scan waited on every step.)
    constexpr u32 STEP = 256;
    const u32 cbuf_addr = W9Lds::PARK + (u32)wv * (W8_RING * W8_ES * 4u);
    u32 bias[4];
    w9_bias(nvalid, bias);
    // lane constants: byte rotation of a point's code (out byte t = code byte (t + j) mod 8) and the low address byte of slot t:
    // copy << 7 | ((t + j) mod 8) << 4
    const int j = lane & 7, cpy = (lane >> 4) & 1;
    u32 rsel0 = 0, rsel1 = 0, ap0 = 0, ap1 = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        rsel0 |= (u32)((b + j) & 7) << (8 * b);
        rsel1 |= (u32)((4 + b + j) & 7) << (8 * b);
        ap0 |= ((u32)((b + j) & 7) * 16u + (u32)cpy * 128u) << (8 * b);
        ap1 |= ((u32)((4 + b + j) & 7) * 16u + (u32)cpy * 128u) << (8 * b);
    }
    // address of slot t = perm{byte 0: lane part of slot t, byte 1: rotated code byte t, bytes 2, 3: zero}
    const u32 asel[4] = {0x0C0C0400u, 0x0C0C0501u, 0x0C0C0602u, 0x0C0C0703u};
    const int lane16 = lane * 16;
    int head = 0, ccnt = 0;
    u32 since = 0;
    bool pend = false;
    W9Pass ps;
    ps.ev = (v4f){0.f, 0.f, 0.f, 0.f};
    ps.ew = ps.ev;
    ps.pos = 0;
    ps.ok = false;
    u32 rw[4][2];
    u64 fm[4];
    bool flush = false;
    // A COLD work item (a slot whose query has no bound yet: every point of the first step is a candidate) starts with one exchange between
    // the eight waves: each takes the ceil(K / 8)-th smallest integer sum of ITS first 256 points, T = the largest of the eight -- every
    // wave holds ceil(K / 8) points at or below T, the workgroup K -- and (T + 8) / inv + dc bounds K real distances from above (header):
    // the bound of the 16th-or-so best of 2048 points instead of each wave's own K-th of 256, five times fewer candidates in the steps
    // that follow, and not one exact sum spent on it.  Workgroup-uniform conditions only (the item's own constants in LDS, a range that
    // gives every wave a whole first step), so all eight waves reach the barrier.
    u32 coldmask = 0;
    if (p1 - p0 >= (u32)W8_NW * STEP) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float inv = w9_inv(s);
            const u32 hh = __builtin_amdgcn_readfirstlane(w9_lds<u32>(W9Lds::HARD + 8u * s + 4u));
            if (s < nvalid && hh >= 0x7F800000u && inv > 0.0f && inv < 1.0e30f) coldmask |= 1u << s;
        }
    }
    bool first = true;
    const u32 ptail = p1 > 2u * W8_NW * STEP ? p1 - 2u * W8_NW * STEP : 0u;   // a wave's last two steps start at or behind this point
    for (u32 pb = p0 + wv * STEP;; pb += W8_NW * STEP) {
        bool overflow = false;
        if (pb >= p1) {   // uniform: past the end -- what is still parked gets its sums, then the wave leaves
            if (ccnt == 0 && !pend) break;
            flush = true;
#pragma unroll
            for (int r = 0; r < 4; ++r) fm[r] = 0;
        } else {
            if (__builtin_expect(pend, 0)) {   // uniform: the pass requested during the previous step
                W8_T0(td0);
                W8_CNT(pr, 10, 1);
#ifdef W8_PROF
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pass's wait for its entries (and the code requests ahead of them) on its own
                W8_ADD(pr, 7, td0);
#endif
                pend = false;
                w9_pass_finish(ps, nvalid, K, lane);
                since = 0;
                w9_bias(nvalid, bias);   // (the other waves' offers moved the bounds as well)
                if (ccnt >= W8_TRIG || (ccnt > 0 && pb >= ptail)) {   // the next ones are waiting already (or the range ends)
                    W8_CNT(pr, 11, 8);
                    w9_pass_issue(ps, cbuf_addr, head, ccnt, gt, lane);
                    pend = true;
                }
                W8_ADD(pr, 5, td0);
            } else if (__builtin_expect(ccnt > 0 && pb >= ptail, 0)) {
                // the wave's last two steps: what is parked does not wait for company -- its pass is under way while these steps are
                // scanned, and the end of the range finds an empty ring nine times in ten (a pass worked off THERE is a trip to L2 the
                // wave sits out, with the other seven waiting for it at the barrier behind: the wait was 8 % of the kernel)
                W8_CNT(pr, 11, ccnt < 8 ? ccnt : 8);
                wave_sync();
                w9_pass_issue(ps, cbuf_addr, head, ccnt, gt, lane);
                pend = true;
            } else if (++since >= (u32)W8_REFRESH) {
                // the workgroup's bounds move even when this wave has no candidates of its own
                since = 0;
                w9_bias(nvalid, bias);
            }
            // the next step's offsets: past the end the wave's current halves are read once more (no branch around a request, no second
            // value for a register set to merge with; a half that starts beyond the list repeats the first one: never a byte beyond the
            // 127 points of slack the four-wave kernels read too)
            const u32 pn = pb + W8_NW * STEP;
            const u32 pa = pn < p1 ? pn : pb;
            const u32 pbb = pa + 128u < p1 ? pa + 128u : pa;
            u32 qa[4][4];
            auto half = [&](auto hc, v4u &cx, u32 pnext) __attribute__((always_inline)) {
                constexpr int h = decltype(hc)::value;
                // the half's bytes leave its register set rotated (tied together so that no part of them can sink below the request that
                // follows), and the next step's half is requested INTO it
                rw[2 * h][0] = __builtin_amdgcn_perm(cx.y, cx.x, rsel0);
                rw[2 * h][1] = __builtin_amdgcn_perm(cx.y, cx.x, rsel1);
                rw[2 * h + 1][0] = __builtin_amdgcn_perm(cx.w, cx.z, rsel0);
                rw[2 * h + 1][1] = __builtin_amdgcn_perm(cx.w, cx.z, rsel1);
                asm volatile("" : "+v"(rw[2 * h][0]), "+v"(rw[2 * h][1]), "+v"(rw[2 * h + 1][0]), "+v"(rw[2 * h + 1][1]), "+v"(cx));
                cx = __builtin_amdgcn_raw_buffer_load_b128(codes, lane16, (int)(pnext * 8u), W8_STREAM_AUX);
                // eight gathers (32 registers) in flight: the first point's are issued before the first add; each of its entries, once
                // summed, hands its registers to the same slot's gather of the second point (one fill and one drain per half)
                v4u ev[8];
                static_for<8>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    const u32 ea = w9_perm(rw[2 * h][t >> 2], t < 4 ? ap0 : ap1, asel[t & 3]);
                    ev[t] = lds_load_abs<v4u>(ea);
                });
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) qa[2 * h][i] = bias[i];
                static_for<8>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    qa[2 * h][0] += ev[t].x;
                    qa[2 * h][1] += ev[t].y;
                    qa[2 * h][2] += ev[t].z;
                    qa[2 * h][3] += ev[t].w;
                    const u32 ea = w9_perm(rw[2 * h + 1][t >> 2], t < 4 ? ap0 : ap1, asel[t & 3]);
                    ev[t] = lds_load_abs<v4u>(ea);
                    __builtin_amdgcn_sched_barrier(0);
                });
#pragma unroll
                for (int i = 0; i < 4; ++i) qa[2 * h + 1][i] = bias[i];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    qa[2 * h + 1][0] += ev[t].x;
                    qa[2 * h + 1][1] += ev[t].y;
                    qa[2 * h + 1][2] += ev[t].z;
                    qa[2 * h + 1][3] += ev[t].w;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            half(IntC<0>{}, ca, pa);
            half(IntC<1>{}, cb, pbb);
            // a field below 0x8000 <=> that query's integer sum is within its budget (w9_bias); one compare for the four points
            u32 x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = (qa[r][0] & qa[r][1]) & (qa[r][2] & qa[r][3]);
            u64 anym = __builtin_amdgcn_ballot_w64((((x[0] & x[1]) & (x[2] & x[3])) & 0x80008000u) != 0x80008000u);
#if defined(W8_KO) && (W8_KO & 1)
            asm volatile("" :: "s"(anym));
            anym = 0;   // knock-out build (wrong results by design): the filter's fast path alone
#endif
            W8_CNT(pr, 8, 1);
            if (__builtin_expect(first && coldmask != 0u, 0)) {   // uniform over the WORKGROUP: see above
                const int r8 = (K + W8_NW - 1) / W8_NW;
                static_for<8>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if ((coldmask >> s) & 1u) {   // uniform
                        // (the slot's bias is 0 while it has no bound: the fields are the sums)
                        u32 f[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] = (s & 1) ? (qa[r][s >> 1] >> 16) : (qa[r][s >> 1] & 0xffffu);
                        const u32 V = w9_kth_sum4(f[0], f[1], f[2], f[3], 0xFu, r8);
                        if (lane == 0) *w9_ptr<u32>(W9Lds::COLD + 64u * s + 4u * (u32)wv) = V;
                    }
                });
                __syncthreads();
                static_for<8>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if ((coldmask >> s) & 1u) {   // uniform
                        u32 T = 0;
#pragma unroll
                        for (int v = 0; v < W8_NW; ++v) {
                            const u32 o = __builtin_amdgcn_readfirstlane(w9_lds<u32>(W9Lds::COLD + 64u * s + 4u * (u32)v));
                            T = o > T ? o : T;
                        }
                        const float ub = (w9_dc(s) + (float)(T + 8u) * (1.00001f / w9_inv(s))) * 1.00002f;
                        // (every wave arrives at the same bound; the wave's own atomic is ahead of its own reads of the word)
                        if (ub < 3.0e38f && lane == 0) atomicMin(w9_ptr<u64>(W9Lds::STHR + 8u * s), make_key(ub, 0xFFFFFFFFu));
                    }
                });
                // the step's fields were accumulated under the old bias: re-based on the new one, and the step is tested again
                u32 nb[4];
                w9_bias(nvalid, nb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) qa[r][i] = qa[r][i] - bias[i] + nb[i];
                    x[r] = (qa[r][0] & qa[r][1]) & (qa[r][2] & qa[r][3]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) bias[i] = nb[i];
                anym = __builtin_amdgcn_ballot_w64((((x[0] & x[1]) & (x[2] & x[3])) & 0x80008000u) != 0x80008000u);
                W8_CNT(pr, 12, 1);
            }
            first = false;
            if (__builtin_expect(anym != 0, 0)) {   // uniform; a step in ten once the bounds are tight
                W8_T0(tc0);
                W8_CNT(pr, 9, 1);
                // the lane's four candidate flags; a list's last step masks the points past its end (they carry whatever was loaded)
                bool c[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] = (x[r] & 0x80008000u) != 0x80008000u;
                const u32 pt0 = pb + (u32)lane * 2u;
                if (pb + STEP > p1) {   // uniform
#pragma unroll
                    for (int r = 0; r < 4; ++r) c[r] = c[r] && pt0 + (u32)(r >> 1) * 128u + (u32)(r & 1) < p1;
                }
                u64 m[4];
                int n[4], ntot = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[r] = __builtin_amdgcn_ballot_w64(c[r]);
                    n[r] = __popcll(m[r]);
                    ntot += n[r];
                }
                // a crowd with no bound at all (a cold work item's first step): bounds from the integer sums first (header)
                if (ntot > 8 && (int)min(p1 - pb, STEP) >= K) {
                    bool moved = false;
                    u32 vb = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) vb |= (pt0 + (u32)(r >> 1) * 128u + (u32)(r & 1) < p1) ? (1u << r) : 0u;
                    static_for<8>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        const float inv = w9_inv(s);
                        // (a scale that is not a normal number -- all-zero or denormal tables -- keeps the plain path)
                        if (s < nvalid && (u32)(w9_sthr(s) >> 32) >= 0x7F800000u && inv > 0.0f && inv < 1.0e30f) {   // uniform
                            // the sums themselves: field - bias (no borrow: every field started from its bias)
                            const u32 bs = (s & 1) ? (bias[s >> 1] >> 16) : (bias[s >> 1] & 0xffffu);
                            u32 f[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) f[r] = ((s & 1) ? (qa[r][s >> 1] >> 16) : (qa[r][s >> 1] & 0xffffu)) - bs;
                            const u32 U = w9_kth_sum4(f[0], f[1], f[2], f[3], vb, K);
                            const float ub = (w9_dc(s) + (float)(U + 8u) * (1.00001f / inv)) * 1.00002f;
                            if (ub < 3.0e38f) {
                                if (lane == 0) atomicMin(w9_ptr<u64>(W9Lds::STHR + 8u * s), make_key(ub, 0xFFFFFFFFu));
                                moved = true;
                            }
                        }
                    });
                    if (moved) {
                        W8_CNT(pr, 12, 1);
                        // the step's fields were accumulated under the old bias: re-based on the new one before they are tested again
                        u32 nb[4];
                        w9_bias(nvalid, nb);
                        ntot = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const u32 y = ((qa[r][0] - bias[0] + nb[0]) & (qa[r][1] - bias[1] + nb[1])) & ((qa[r][2] - bias[2] + nb[2]) & (qa[r][3] - bias[3] + nb[3]));
                            c[r] = (y & 0x80008000u) != 0x80008000u && ((vb >> r) & 1u) != 0u;
                            m[r] = __builtin_amdgcn_ballot_w64(c[r]);
                            n[r] = __popcll(m[r]);
                            ntot += n[r];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) bias[i] = nb[i];
                    }
                }
                // park (rotated code bytes, position | rotation << 29: positions stay below 2^28, the code stream's byte offsets are 31-bit)
                if (__builtin_expect(ccnt + ntot <= W8_RING, 1)) {
                    int base = head + ccnt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (n[r] == 0) continue;   // uniform
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((u32)(m[r] >> 32), __builtin_amdgcn_mbcnt_lo((u32)m[r], 0u));
                        if (c[r]) {
                            u32 *ent = w9_ptr<u32>(cbuf_addr + (u32)((base + rank) & (W8_RING - 1)) * (W8_ES * 4u));
                            ent[0] = rw[r][0];
                            ent[1] = rw[r][1];
                            ent[2] = (pt0 + (u32)(r >> 1) * 128u + (u32)(r & 1)) | ((u32)j << 29);
                        }
                        base += n[r];
                    }
                    ccnt += ntot;
                    // a pass is requested when eight points wait and none is in flight; it is worked off at the top of the next step
                    if (!pend && (ccnt >= W8_TRIG || pb >= ptail)) {
                        W8_CNT(pr, 11, 8);
                        wave_sync();
                        w9_pass_issue(ps, cbuf_addr, head, ccnt, gt, lane);
                        pend = true;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) fm[r] = m[r];
                    overflow = true;
                }
                W8_ADD(pr, 4, tc0);
            }
        }
        // No room in the ring (a crowd the integer bound could not thin out), or the end of the range: ONE copy of the code that parks in
        // portions and works passes off here and now (the wave waits for each trip to L2; rare)
        if (__builtin_expect(overflow || flush, 0)) {
            const u32 pt0 = pb + (u32)lane * 2u;
            for (;;) {   // uniform
#pragma unroll 1
                for (int r = 0; r < 4; ++r) {
                    const u64 mm = r == 0 ? fm[0] : (r == 1 ? fm[1] : (r == 2 ? fm[2] : fm[3]));
                    if (mm == 0 || ccnt == W8_RING) continue;
                    const int room = W8_RING - ccnt;
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((u32)(mm >> 32), __builtin_amdgcn_mbcnt_lo((u32)mm, 0u));
                    const bool mine = ((mm >> lane) & 1ull) != 0 && rank < room;
                    if (mine) {
                        u32 *ent = w9_ptr<u32>(cbuf_addr + (u32)((head + ccnt + rank) & (W8_RING - 1)) * (W8_ES * 4u));
                        ent[0] = r == 0 ? rw[0][0] : (r == 1 ? rw[1][0] : (r == 2 ? rw[2][0] : rw[3][0]));
                        ent[1] = r == 0 ? rw[0][1] : (r == 1 ? rw[1][1] : (r == 2 ? rw[2][1] : rw[3][1]));
                        ent[2] = (pt0 + (u32)(r >> 1) * 128u + (u32)(r & 1)) | ((u32)j << 29);
                    }
                    const u64 took = __builtin_amdgcn_ballot_w64(mine);
                    ccnt += __popcll(took);
                    if (r == 0) fm[0] &= ~took; else if (r == 1) fm[1] &= ~took; else if (r == 2) fm[2] &= ~took; else fm[3] &= ~took;
                }
                const bool more = (fm[0] | fm[1] | fm[2] | fm[3]) != 0;
                if (pend) {
                    W8_CNT(pr, 10, 1);
                    pend = false;
                    w9_pass_finish(ps, nvalid, K, lane);
                    w9_bias(nvalid, bias);
                }
                if (ccnt > 0 && (more || flush || ccnt >= 8)) {
                    wave_sync();
                    w9_pass_issue(ps, cbuf_addr, head, ccnt, gt, lane);
                    pend = true;
                    if (more || flush) continue;   // (uniform) worked off at once: room for what is left / nothing may stay behind
                }
                if (!more) break;
            }
            if (flush) break;
        }
    }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------------
// item_list[i] = the list of work item i (bucket_scan_kernel writes it next to wi_off: one load instead of a 13-step binary search
// of dependent loads per work item)
// xq: nranges work-queue heads, 64 B apart, zero at launch.  Work items are ordered by list, so the four or five groups of one list are
// neighbours in the queue: the item range is cut into one contiguous part per XCD and a workgroup pulls from the part of the XCD it runs
// on (HW_REG_XCC_ID) -- the groups that stream the same list then run side by side under ONE L2 and the list crosses the fabric once.
// Placement is a matter of speed only: a workgroup whose part is exhausted moves on to the next one; every wave leaves when all are.
__global__ __launch_bounds__(W8_THREADS, W8_NW / 2) void wg8q8_scan_kernel(const ScanArgs a, float *__restrict__ gtabs, const u32 *__restrict__ item_list,
                                                                 u32 *__restrict__ xq, int nranges)
{
    W9Prof pr;
#ifdef W8_PROF
    pr.zero();
    const u64 tk0 = __builtin_readcyclecounter();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IndexView &ix = a.ix;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // (a scalar: branches on the wave's number and its position in the list are scalar branches)
    const int K = a.K;
    float *res = (float *)(smem + W9Lds::RES);
    u32 *smax = (u32 *)(smem + W9Lds::SMAX);
    float *sinv = (float *)(smem + W9Lds::SMAX) + 8;
    float *sdc = (float *)(smem + W9Lds::QC);
    u32 *ssb = (u32 *)(smem + W9Lds::QC) + 8;
    u32 *spi = (u32 *)(smem + W9Lds::QC) + 16;
    u32 *sqi = (u32 *)(smem + W9Lds::QC) + 24;
    u64 *shard = (u64 *)(smem + W9Lds::HARD);
    u64 *sthr = (u64 *)(smem + W9Lds::STHR);
    u32 *swi = (u32 *)(smem + W9Lds::SWI);
    const u32 total = a.wi_off[ix.kc];
    float *gt = gtabs + (size_t)blockIdx.x * W9_GTAB_FLOATS;
    const __amdgpu_buffer_rsrc_t gtr = __builtin_amdgcn_make_buffer_rsrc((void *)gt, 0, (int)(W9_GTAB_FLOATS * 4u), 0x00020000);

    u64 *pool = (u64 *)(smem + W9Lds::POOL);
    // (thread 0's: the part it pulls from, the parts found empty so far)
    int qcur = nranges > 1 ? (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u) : 0, qtried = 0;
    // the item behind ticket k of the current part; a part that is exhausted hands over to the next one (a trip per part: the tail only)
    auto resolve = [&](u32 k) -> u32 {
        for (;;) {
            // (nranges is 8 or 1: no division -- this runs between two barriers of every work item)
            const u32 r0 = nranges == 1 ? 0u : (u32)(((u64)total * (u32)qcur) >> 3), r1 = nranges == 1 ? total : (u32)(((u64)total * (u32)(qcur + 1)) >> 3);
            if (k < r1 - r0) return r0 + k;
            qcur = qcur + 1 == nranges ? 0 : qcur + 1;
            if (++qtried >= nranges) return 0xFFFFFFFFu;
            k = atomicAdd(xq + qcur * 16, 1u);
        }
    };
    if (tid == 0) swi[0] = resolve(atomicAdd(xq + qcur * 16, 1u));
    __syncthreads();
    u32 wi = __builtin_amdgcn_readfirstlane(swi[0]);
    for (;;) {
        if (wi >= total) break;   // uniform: every wave of every workgroup reaches this
        // the NEXT work item's ticket is pulled now and looked at when this one is done: the atomic's trip is off the critical path
        u32 pulled = 0;
        if (tid == 0 && qtried < nranges) pulled = atomicAdd(xq + qcur * 16, 1u);
        do {   // (one trip: `break` = this work item is finished)
        W8_T0(ts0);
        W8_CNT(pr, 14, 1);
        const int l = __builtin_amdgcn_readfirstlane((int)item_list[wi]);
        const u32 cnt = __builtin_amdgcn_readfirstlane(a.list_cnt[l]);
        const u32 ng = (cnt + 7u) / 8u;
        const u32 local = wi - __builtin_amdgcn_readfirstlane(a.wi_off[l]);
        const u32 chunk = local / ng, grp = local - chunk * ng;
        const u32 len = __builtin_amdgcn_readfirstlane(ix.list_len[l]);
        const u32 p0 = chunk * a.CH;
        if (p0 >= len) break;   // uniform
        const u32 p1 = min(len, p0 + a.CH);
        const int nvalid = min(8, (int)(cnt - grp * 8u));

        // the queries of the group: thread s < 8 fetches slot s (slots past nvalid repeat slot 0 and can never be candidates)
        if (tid < 8) {
            const int ss = tid < nvalid ? tid : 0;
            const u32 pi = a.bucket_items[a.bucket_off[l] + grp * 8u + ss];
            const u32 qq = pi / (u32)a.w;
            spi[tid] = pi;
            sqi[tid] = qq;
            ssb[tid] = a.probe_base[pi];
            sdc[tid] = a.probe_dc[pi];
            const u64 t0 = __hip_atomic_load(&a.qthr[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            shard[tid] = t0;
            sthr[tid] = t0;
            smax[tid] = 0u;
        }
        if (tid < 512) pool[tid] = KEY_MAX;
        __syncthreads();
        // exact pruning of whole work items, as in scan_kernel: no sum of this list lies below its coarse distance
        if (a.prune) {
            bool all = true;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                all = all && (s >= nvalid || __builtin_amdgcn_readfirstlane(__float_as_uint(sdc[s])) > (u32)(readfirstlane64(shard[s]) >> 32));
            if (all) {   // uniform
                if (tid < nvalid) {
                    const u32 pi = spi[tid];
                    a.part_cnt[(size_t)pi * a.maxch + chunk] = 0u;
                    atomicAdd(a.scanned_points + (size_t)(pi & 63u) * 8 + 1, (u64)(p1 - p0));
                }
                break;
            }
        }
        // (a chunk's byte offset pb * 8 stays below 2^31: lists of < 2^28 points)
        const uint8_t *cbase = ix.codes + (int64_t)readfirstlane64((u64)ix.list_codeoff[l]);

        W8_ADD(pr, 1, ts0);
        W8_T0(tb0);
        // (1) residuals r_s = q_s - c (coarsequantizers.jl:40-45), two elements per thread: res[ii][t][s], 17 rows of eight per sub-quantizer
        {
            const int tb = tid & 511;
#pragma unroll
            for (int e = tb; e < 1024; e += 512) {
                const int i = e >> 3, s = e & 7;
                res[(i >> 4) * 136 + (i & 15) * 8 + s] = a.queries[(size_t)sqi[s] * 128 + i] - ix.centroids[(size_t)l * 128 + i];
            }
        }
        // (the thread number passes through an opaque move inside the item loop: the lane-constant addresses it feeds -- codewords, table
        // rows, LDS slots -- would otherwise be hoisted to kernel entry and live, spilled, across the whole persistent loop)
        int tidb = tid & 511;
        asm volatile("" : "+v"(tidb));
        // A thread builds FOUR codewords' entries of ONE sub-quantizer: ii = lane mod 4 (+ 4 for odd waves), codewords cg, cg + 64, + 128,
        // + 192.  A residual row read from LDS serves the four codewords (16 reads of 16 B per thread; one codeword in each of four
        // sub-quantizers per thread was 64 -- on the LDS queue the other workgroup's gathers fill), the four lanes of a quad read four
        // different bank groups (the padding), and the quantised rows below leave conflict-free as they are: the 16 lanes of a store's
        // service group hold 4 sub-quantizers x 4 copies.
        const int ii = (tidb & 3) | (((tidb >> 6) & 1) << 2);
        const int cg = ((tidb >> 2) & 15) | ((tidb >> 7) << 4);
        const float4 *ct = (const float4 *)ix.codebooks_t;        // [ii][g][c][4], ksub = 256
        // The four codewords come four dimensions at a time (g = 0 .. 3), two register sets that take turns inside a REAL loop of two
        // trips: fully unrolled, the scheduler hoists every request of the build above the arithmetic -- 64 registers of codewords next to
        // 64 of residual rows -- and spills them as they arrive, a wait for memory each.
        float4 cwa[4], cwb[4];
        const u32 cofs = (u32)ii * 1024u + (u32)cg;
        auto ldcw = [&](float4 (&d)[4], int g) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = ct[cofs + (u32)(g * 256 + 64 * j)];
        };
        ldcw(cwa, 0);       // on its way while the residuals settle
        __syncthreads();
        W8_ADD(pr, 13, tb0);   // (of the build: residuals up to the barrier)
        // (2) the f32 entries (index.jl:232-236: df = cb - r, sum += df * df for t ascending; no contraction; two queries per packed
        // instruction: the same IEEE operations element by element), to device memory by label; per-query maxima
        v4f ent[4][2];      // [codeword][queries 0 .. 3, 4 .. 7]
        {
            float mx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const u32 roff = W9Lds::RES + (u32)ii * 544u;
            // four queries at a time (wg8scan.hip.h's build on each half of the residual rows; the codewords are requested again)
            static_for<2>([&](auto qc) {
                constexpr int qh = decltype(qc)::value;
                if (qh == 1) ldcw(cwa, 0);
                v2f sum[4][2];
#pragma unroll
                for (int j = 0; j < 4; ++j) sum[j][0] = sum[j][1] = (v2f){0.0f, 0.0f};
                // (the rows of a trip -- eight dimensions -- are requested together at its top)
                v4f rv[8];
                auto grp = [&](const float4 (&cq)[4], int g2) __attribute__((always_inline)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const v2f r01 = (v2f){rv[4 * g2 + t].x, rv[4 * g2 + t].y}, r23 = (v2f){rv[4 * g2 + t].z, rv[4 * g2 + t].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float cv = t == 0 ? cq[j].x : (t == 1 ? cq[j].y : (t == 2 ? cq[j].z : cq[j].w));
                            const v2f c2 = (v2f){cv, cv};
                            const v2f d0 = c2 - r01, d1 = c2 - r23;
                            sum[j][0] = sum[j][0] + d0 * d0;
                            sum[j][1] = sum[j][1] + d1 * d1;
                        }
                    }
                };
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) rv[t] = w9_lds<v4f>(roff + (u32)(8 * h + t) * 32u + 16u * qh);
                    ldcw(cwb, 2 * h + 1);
                    grp(cwa, 0);
                    ldcw(cwa, h == 0 ? 2 : 3);      // (the second trip repeats a request: no branch around one, no second value to merge)
                    grp(cwb, 1);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ent[j][qh] = (v4f){sum[j][0].x, sum[j][0].y, sum[j][1].x, sum[j][1].y};
                    const int c = cg + 64 * j;
                    const int label = ix.identity_labels ? c : (int)ix.labels[ii * 256 + c];
                    *(v4f *)(gt + ((size_t)(ii * 256 + label) << 3) + 4 * qh) = ent[j][qh];
                    mx[4 * qh + 0] = fmaxf(mx[4 * qh + 0], ent[j][qh].x);
                    mx[4 * qh + 1] = fmaxf(mx[4 * qh + 1], ent[j][qh].y);
                    mx[4 * qh + 2] = fmaxf(mx[4 * qh + 2], ent[j][qh].z);
                    mx[4 * qh + 3] = fmaxf(mx[4 * qh + 3], ent[j][qh].w);
                }
            });
            // (entries are >= +0: the bit pattern orders like the value; the wave's maximum on the DPP network and the scalar unit)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32 wm = w9_row_max_u32(__float_as_uint(mx[s]));
                if (lane == 0) atomicMax(&smax[s], wm);
            }
        }
        __syncthreads();
        W8_ADD(pr, 15, tb0);   // (of the build: up to the barrier behind the entries)
        // (3) quantise (quantize_tables_m8's rule: q = min(4095, floor(t * inv)), inv = 4095 / largest entry of the query) and write the two
        // copies: copy (cp + lane / 4) mod 2 of sub-quantizer ii -- the 8 lanes of a 16-byte store's service group write 8 different
        // four-bank groups (consecutive labels are 256 B apart: the same banks).
        {
            float inv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float mxs = __uint_as_float(smax[s]);
                inv[s] = mxs > 0.0f ? 4095.0f / mxs : 0.0f;
            }
            if (tid < 8) sinv[tid] = inv[tid];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ev[8] = {ent[j][0].x, ent[j][0].y, ent[j][0].z, ent[j][0].w, ent[j][1].x, ent[j][1].y, ent[j][1].z, ent[j][1].w};
                u32 f[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const u32 v = (u32)floorf(ev[s] * inv[s]);
                    f[s] = v < 4095u ? v : 4095u;
                }
                const uint4 qv = make_uint4(f[0] | (f[1] << 16), f[2] | (f[3] << 16), f[4] | (f[5] << 16), f[6] | (f[7] << 16));
                const int c = cg + 64 * j;
                const int lb = ix.identity_labels ? c : (int)ix.labels[ii * 256 + c];
                const u32 row = ((u32)lb << 8) | ((u32)ii << 4);
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) *(uint4 *)(smem + (row | ((u32)((cp + (lane >> 2)) & 1) << 7))) = qv;
            }
        }
        // the wave's first two steps of code bytes: requested here, behind the build (held across it they were spilled: a store that waits
        // for the load it saves)
        // (a list's last step reads up to 127 points past p1 -- other lists' bytes or the slack behind the last list, never used: as scan_kernel)
        const __amdgpu_buffer_rsrc_t codes = __builtin_amdgcn_make_buffer_rsrc((void *)cbase, 0, (int)0x7FFFFFF0, 0x00020000);
        v4u ca = (v4u){0u, 0u, 0u, 0u}, cb = ca;
        {
            const u32 pb0 = p0 + (u32)wv * 256u;
            if (pb0 < p1) {
                ca = __builtin_amdgcn_raw_buffer_load_b128(codes, lane * 16, (int)(pb0 * 8u), 0);
                cb = __builtin_amdgcn_raw_buffer_load_b128(codes, lane * 16, (int)((pb0 + 128u < p1 ? pb0 + 128u : pb0) * 8u), 0);
            }
        }
        __syncthreads();   // tables complete (LDS copies; the f32 stores have left for L2: the barrier's release covers them)

        W8_ADD(pr, 2, tb0);
        W8_T0(tsc0);
        __builtin_amdgcn_s_setprio(W8_PRIO_SCAN);
#if defined(W8_KO) && (W8_KO & 2)
        if (K < 0)                    // knock-out build: table build only
#endif
        w9_scan_range(codes, p0, p1, nvalid, K, wv, lane, ca, cb, gtr, pr);
        __builtin_amdgcn_s_setprio(W8_PRIO_REST);
        W8_ADD(pr, 3, tsc0);
        W8_T0(tm0);

        // ---- every wave has offered what it had: wave s < nvalid hands slot s of the pool over as it is -- the entries fill from index 0
        // (an offer takes the first empty one), the merge kernel behind pushes them through a selector in any order
        __syncthreads();
        if (wv < nvalid) {
            const int s = wv;
            const u64 v = lane < K ? pool[64 * s + lane] : 0ull;
            const int fc = __popcll(__builtin_amdgcn_ballot_w64(lane < K && v != KEY_MAX));
            const size_t slot = (size_t)spi[s] * a.maxch + chunk;
            if (lane < fc) a.part_keys[slot * K + lane] = v;
            if (fc == K) {   // uniform
                const u64 kth = w9_wave_max_u64(v);
                if (lane == 0) atomicMin(&a.qthr[sqi[s]], kth);
            }
            if (lane == 0) a.part_cnt[slot] = (u32)fc;
        }
        W8_ADD(pr, 6, tm0);
        } while (false);
        __syncthreads();            // every wave is done with this item's state in LDS
        if (tid == 0) swi[0] = qtried < nranges ? resolve(pulled) : 0xFFFFFFFFu;
        __syncthreads();
        wi = __builtin_amdgcn_readfirstlane(swi[0]);
    }
#ifdef W8_PROF
    pr.c[0] = __builtin_readcyclecounter() - tk0;
    if (lane == 0) {
        u64 *dst = (u64 *)(gtabs + (size_t)gridDim.x * W9_GTAB_FLOATS);
        for (int i = 0; i < 16; ++i) atomicAdd(dst + i, pr.c[i]);
    }
#endif
}
