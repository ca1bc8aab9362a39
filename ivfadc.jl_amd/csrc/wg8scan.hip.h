// wg8scan.hip.h -- list-major scan for long lists, EIGHT waves per workgroup on ONE table set (m = 8, dsub = 16, ksub = 256, K <= 64:
// the SIFT1B shape).  Included by kernels.hip.h, namespace ivf.
//
// Reference: src/coarsequantizers.jl:40-45 (residuals), src/index.jl:232-236 (table build), :240-246 (scan), :247-254 (bounded top-K).
//
// What binds the four-wave kernel (scan_kernel<8, 16, 4, ..., STRIPE>) on this shape, by the counters: the LDS array is
// 81 % busy and two thirds of its cycles are bank-conflict replays -- the 32 lanes of a ds_read_b64 service group look the SAME
// sub-quantizer up at random codes, ~3 distinct rows per bank pair -- at 12 waves per CU (164 registers, 53 KB of LDS per workgroup).
// This kernel changes the three things that follow from that:
//
//   * CONFLICT-FREE GATHERS.  The 16-bit integer filter table (four queries per 8-byte entry, quantize as in quantize_tables_m8)
//     stands in FOUR copies; a code's row is exactly 256 bytes -- 4 copies x 8 sub-quantizers x 8 B = the 64 banks once -- and lane l
//     looks sub-quantizer (t + l) mod 8 up at slot t in copy (l / 8) mod 4: the 32 lanes of a service group read 32 different bank
//     pairs whatever their codes are.  Integer addition is associative, so the rotated order costs nothing, and because a row is
//     256 B the whole address (code << 8 | lane part) is ONE v_perm_b32 of the rotated code dword and a lane-constant dword.
//   * SIXTEEN WAVES PER CU.  512-thread workgroups, two per CU, at most 128 registers: one table build and one quantisation per
//     eight waves, four waves per SIMD to cover the LDS round trips.
//   * The f32 tables leave the LDS (64 KB of integer copies + 5 KB of state: two workgroups per CU).  The build writes them -- the
//     reference's sums, index.jl:232-236 order -- to a 32 KB block of device memory per workgroup (it stays in L2 / the memory-side
//     cache); what the filter lets through is parked and gets its reference-order sum from there, eight points per pass (lane ii of a
//     segment fetches sub-quantizer ii's entry, the running sum walks along the segment: drain as in drain_parked).  Only these sums
//     meet the selectors: ids and distances stay bit-identical to the oracle.
//   * A CROWD (a cold work item: no bound yet, every point a candidate) is not summed exactly point by point: the integer sums bound
//     the distances from ABOVE as well -- S < (dc + (Q + 8) / inv)(1 + 2^-16) -- so the K-th smallest integer sum of a step gives a
//     bound that K real points meet without one exact sum, and only what still passes under it is parked.
//
// Selection, bounds (workgroup-shared word in LDS, per-query word in HBM), partial results and the merge kernel behind it are those of
// scan_kernel; the work items are the same (list, group of <= 4 queries, chunk).
#pragma once

#ifndef W8_NWAVES
#define W8_NWAVES 8
#endif
// Waves per workgroup.  Measured with more (round 6; the table build is laid out for 512 threads, further waves repeat the first ones' share --
// the same values to the same places): TEN (640 threads at <= 96 registers) run one workgroup per CU -- a workgroup's waves spread 3-3-2-2
// over the SIMDs, two of them would need six register sets on a SIMD -- 9.6 ms; TWELVE (<= 80 registers: the scan loop still holds no
// spilled register, the candidate path 150) run two per CU and take 9.1 ms, 8.5 without candidates against 5.0: six waves per SIMD on the
// same LDS are slower than four, whatever the guide's 2 cycles per ds_read_b64 leave free on paper.
constexpr int W8_NW = W8_NWAVES;
constexpr int W8_THREADS = 64 * W8_NW;
constexpr int W8_ES = 3;                      // dwords per parked point: code bytes (2), list position
constexpr u32 W8_TAB_BYTES = 256u * 256u;     // 256 codes x (4 copies x 8 sub-quantizers x 8 bytes)
constexpr u32 W8_GTAB_FLOATS = 8u * 256u * 4u;   // f32 tables of a work item in device memory: [ii][label][4 queries]

struct W8Lds {
    static constexpr u32 RES = W8_TAB_BYTES;                  // f32 residuals [ii][t][s]: 8 x (16 x 4 + 4 of padding) x 4 B (a sub-quantizer's block starts 4 banks on)
    static constexpr u32 SMAX = RES + 2176u;                  // u32 [4]: bits of the per-query largest entry (atomicMax); f32 inv[4] behind
    static constexpr u32 QC = SMAX + 32u;                     // f32 dc[4]; u32 visit-order base[4]; u32 probe index[4]; u32 query[4]
    static constexpr u32 HARD = QC + 64u;                     // u64 [4]: the bounds the item started from
    static constexpr u32 STHR = HARD + 32u;                   // u64 [4]: workgroup-shared bounds
    static constexpr u32 SCNT = STHR + 32u;                   // int [8][4] + [4]
    static constexpr u32 SWI = SCNT + 144u;                   // u32 [4]
    static constexpr u32 POOL = SWI + 16u;                    // u64 [4][64]: the workgroup's K smallest keys per slot, unordered (w8_pool_offer)
    static constexpr u32 PARK = POOL + 4u * 64u * 8u;         // u32 [8][32][W8_ES]: the waves' rings of parked points (W8_RING)
    static constexpr u32 COLD = PARK + (u32)W8_NW * 32u * W8_ES * 4u;     // u32 [4][8]: a cold work item's first step, every wave's ceil(K / 8)-th smallest integer sum per slot
    static constexpr u32 END = COLD + 256u;                   // (u32 [4][16]: up to sixteen waves)
};
static_assert(W8Lds::END <= 80u * 1024u, "two workgroups per CU");
static_assert((W8Lds::HARD & 7u) == 0 && (W8Lds::STHR & 7u) == 0 && (W8Lds::POOL & 7u) == 0, "8-byte bounds");

// W8_PROF (diagnostic builds only: tools/build_variant.sh prof -DW8_PROF): cycle and event counters per wave, summed into 16 words behind the
// f32 table blocks.  0 item loop, 1 setup, 2 build, 3 scan, 4 candidate path, 5 drains, 6 merge, 7 of the drains: the wait for memory; 8 steps, 9 steps with candidates, 10 drains,
// 11 drained points, 12 crowd bounds, 13 refreshes that moved a bound, 14 items
#ifdef W8_PROF
struct W8Prof {
    u64 c[16];
    __device__ __forceinline__ void zero() { for (int i = 0; i < 16; ++i) c[i] = 0; }
};
#define W8_T0(name) const u64 name = __builtin_readcyclecounter()
#define W8_ADD(pr, i, t0) (pr).c[i] += __builtin_readcyclecounter() - (t0)
#define W8_CNT(pr, i, n) (pr).c[i] += (u64)(n)
#else
struct W8Prof {};
#define W8_T0(name)
#define W8_ADD(pr, i, t0)
#define W8_CNT(pr, i, n)
#endif

static __device__ __forceinline__ u32 w8_perm(u32 s0, u32 s1, u32 sel)
{
    u32 o;
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(o) : "v"(s0), "v"(s1), "s"(sel));
    return o;
}

// the work item's per-slot constants stand in LDS at fixed addresses (the kernel owns the whole allocation: the dynamic segment starts
// at address 0); the rare paths read them there instead of holding two dozen scalars across the scan loop
template <class T> static __device__ __forceinline__ T w8_lds(u32 byte_addr) { return lds_load_abs<T>(byte_addr); }
// a pointer into the workgroup's dynamic LDS segment (for stores and atomics: an integer cast to a generic pointer is NOT an LDS address)
template <class T> static __device__ __forceinline__ T *w8_ptr(u32 byte_off)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char w8_smem[];
    return (T *)(w8_smem + byte_off);
}
static __device__ __forceinline__ float w8_dc(int s) { return __uint_as_float(__builtin_amdgcn_readfirstlane(w8_lds<u32>(W8Lds::QC + 4u * s))); }
static __device__ __forceinline__ u32 w8_sbase(int s) { return __builtin_amdgcn_readfirstlane(w8_lds<u32>(W8Lds::QC + 16u + 4u * s)); }
static __device__ __forceinline__ float w8_inv(int s) { return __uint_as_float(__builtin_amdgcn_readfirstlane(w8_lds<u32>(W8Lds::SMAX + 16u + 4u * s))); }
static __device__ __forceinline__ u64 w8_sthr(int s) { return readfirstlane64(w8_lds<u64>(W8Lds::STHR + 8u * s)); }

// The accumulator BIAS of the four queries under the bounds of the moment: field s of a point's accumulators starts at
// B_s = 0x7FFF - T_s (T_s = qf_targets' integer budget of query s; an unused slot gets 0x8000), so "field < 0x8000" <=> "sum_s <= T_s":
// the candidate test of a step is four ANDs and a compare, and the first add of a point absorbs the bias.  sum <= 32760, B <= 0x8000:
// fields never carry.  The bounds are the workgroup's (STHR in LDS: the pool's K-th key, the item's bound from outside, an integer-sum
// bound of a cold start -- whichever is smallest); a wave holds no bound of its own.
// Lane s (mod 4) evaluates slot s -- qf_targets' arithmetic, operation for operation (its argument is what makes the filter exact) -- on the
// slot's constants in LDS: one round trip and a dozen vector instructions for the four slots (evaluated slot by slot on uniform values it
// was eight dependent LDS round trips and ~200 instructions, paid at every refresh and after every pass: most of the candidate path).
static __device__ __forceinline__ void w8_bias_of(u32 th, float dc, float inv, u32 sl, int nvalid, u32 (&bias)[2])
{
    u32 T = 0x7FFFu;
    if (th < 0x7F800000u) {   // a finite bound
        const float thr = __uint_as_float(th);
        const float x = (thr * 1.0000038146972656f - dc) * inv * 1.0000038146972656f;   // (1 + 2^-18): as qf_targets
        T = x < 0.0f ? 0u : (x < 32000.0f ? (u32)x + 2u : 0x7FFFu);
    }
    const u32 B = (int)sl < nvalid ? 0x7FFFu - T : 0x8000u;
    bias[0] = (u32)__builtin_amdgcn_readlane((int)B, 0) | ((u32)__builtin_amdgcn_readlane((int)B, 1) << 16);
    bias[1] = (u32)__builtin_amdgcn_readlane((int)B, 2) | ((u32)__builtin_amdgcn_readlane((int)B, 3) << 16);
}
static __device__ __forceinline__ void w8_bias(int nvalid, u32 (&bias)[2])
{
    const u32 sl = (u32)lane_id() & 3u;
    const u32 th = w8_lds<u32>(W8Lds::STHR + 8u * sl + 4u);
    const float dc = w8_lds<float>(W8Lds::QC + 4u * sl);
    const float inv = w8_lds<float>(W8Lds::SMAX + 16u + 4u * sl);
    w8_bias_of(th, dc, inv, sl, nvalid, bias);
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
static __device__ __forceinline__ u32 w8_row_max_u32(u32 x)
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
static __device__ __forceinline__ u64 w8_wave_max_u64(u64 v)
{
    const u32 hi = (u32)(v >> 32), lo = (u32)v;
    const u32 mh = w8_row_max_u32(hi);
    const u32 ml = w8_row_max_u32(hi == mh ? lo : 0u);
    return ((u64)mh << 32) | ml;
}
// the pool's entries of slot s, one per lane (lanes >= K: 0, below every key)
static __device__ __forceinline__ u64 w8_pool_read(int s, int K, int lane)
{
    return lane < K ? w8_lds<u64>(W8Lds::POOL + 512u * (u32)s + 8u * (u32)lane) : 0ull;
}
// Offers the keys of the lanes in `mask` (uniform, non-empty) to slot s, starting from the snapshot v the caller read a while ago.  A stale
// snapshot is as good as a fresh one for every decision above -- each of its values WAS that entry's, entries only decrease -- it merely
// fails a swap more often, and a failed swap returns the entry's value of the moment: the snapshot is patched and the offer goes on
// without another read.  An offer costs one LDS round trip per swap attempt (under the scan's gathers a round trip is several hundred
// cycles: the dependent trips, not the instructions, were the cost of a pass).  Returns the slot's new bound, KEY_MAX if nothing went in
// or the pool is not full.
static __device__ __forceinline__ u64 w8_pool_offer(int s, u64 v, u64 key, u64 mask, int K, int lane)
{
    bool any = false;
    u64 *pool = w8_ptr<u64>(W8Lds::POOL + 512u * (u32)s);
    u64 mx = w8_wave_max_u64(v);
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
        mx = w8_wave_max_u64(v);
    }
    if (!any || mx == KEY_MAX) return KEY_MAX;
    if (lane == 0) atomicMin(w8_ptr<u64>(W8Lds::STHR + 8u * (u32)s), mx);
    return mx;
}

// ---- reference-order sums of parked points, 8 per pass, entries from the work item's f32 tables in device memory -------------------
// (scope: the tables were written by this workgroup before a barrier; the loads go to L2 -- sc1 -- so that no line of an earlier work
// item's tables can be served from this CU's vector cache)
static __device__ __forceinline__ v4f w8_gtab_load(__amdgpu_buffer_rsrc_t rs, u32 ii, u32 byte)
{
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((ii << 8) | byte) << 4), 0, 16);
    return (v4f){__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}

// A pass in flight: lane 8 e + ii holds the f32 entries (four queries) of sub-quantizer ii at parked point e's code byte, and the point's
// position.  Requested when eight points are waiting (w8_pass_issue) and worked off at the top of the NEXT step, right behind the wait
// for that step's code bytes -- older than the gather -- so the trip to L2 costs the wave nothing (a pass worked off where it is
// requested waits for the code stream's request in flight AND its own: ~4 us per pass, measured).
struct W8Pass {
    v4f ev;
    u32 pos;
    bool ok;
};
// cache policy of the code stream's requests (aux of raw_buffer_load).  Measured with 2 (nt, "streaming": the lines are not kept in L2 ahead
// of the work items' f32 tables, which the passes gather from): 16 384 x w = 8 scan 6.15 -> 7.35 ms, 2048 x w = 8 1.15 -> 1.31 -- the four
// or five groups that stream the same list side by side live on each other's lines in L2 / the memory-side cache.  Default policy.
#ifndef W8_STREAM_AUX
#define W8_STREAM_AUX 0
#endif
#ifndef W8_TRIG
#define W8_TRIG 8        // parked points that trigger a pass
#endif
#ifndef W8_REFRESH
#define W8_REFRESH 8     // steps between looks at the workgroup's shared bounds
#endif
#ifndef W8_PRIO_SCAN
#define W8_PRIO_SCAN 3
#endif
#ifndef W8_PRIO_REST
#define W8_PRIO_REST 0
#endif
constexpr int W8_RING = 32;    // parked points per wave (a ring: entries head .. head + cnt - 1 mod 32)

static __device__ __forceinline__ void w8_pass_issue(W8Pass &ps, u32 cbuf_addr, int &head, int &cnt, __amdgpu_buffer_rsrc_t gt, int lane)
{
    const int seg = lane >> 3, ii = lane & 7;
    ps.ok = seg < cnt;
    const u32 ea = cbuf_addr + (u32)((head + (ps.ok ? seg : 0)) & (W8_RING - 1)) * (W8_ES * 4u);
    // (parked: the point's ROTATED code bytes -- out byte t = code byte (t + j) mod 8, j = the parking lane's rotation, kept in the position
    // word's top three bits: the scan loop holds no unrotated copy of a step's bytes)
    const u32 pj = w8_lds<u32>(ea + 8u);
    const u32 idx = ((u32)ii - (pj >> 29)) & 7u;
    const u32 dw = w8_lds<u32>(ea + 4u * (idx >> 2));
    ps.pos = pj & 0x1FFFFFFFu;
    ps.ev = w8_gtab_load(gt, (u32)ii, (dw >> (8 * (idx & 3))) & 0xffu);
    const int take = cnt < 8 ? cnt : 8;
    head = (head + take) & (W8_RING - 1);
    cnt -= take;
}

// Works a pass off.  (Measured and dropped: everything the pass needs from LDS -- constants, bounds, a snapshot of every slot's pool, the
// bias arithmetic's operands -- requested in one go ahead of the running sums, the bias computed from registers: 16 384 x w = 8
// 5.87 -> 5.91 ms, w = 1 1.38 -> 1.44.  The pass does not wait for memory -- W8_PROF: 260 of its 7 700 cycles -- it is ~400 dependent
// instructions on a SIMD it shares with three scanning waves.)
static __device__ __forceinline__ void w8_pass_finish(const W8Pass &ps, int nvalid, int K, int lane)
{
    const int ii = lane & 7;
    const float ev[4] = {ps.ev.x, ps.ev.y, ps.ev.z, ps.ev.w};
    float x[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) x[s] = w8_dc(s) + ev[s];
#pragma unroll
    for (int i = 1; i < 8; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // lane l <- lane l-1 within a row of 16 (row_shr:1): what lane i reads at step i is lane i-1's value of step i-1, so the
            // segment's last lane ends with ((dc + t0) + t1) + ... + t7 (index.jl:242-246)
            const float up = __uint_as_float((u32)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x[s]), 0x111, 0xf, 0xf, false));
            x[s] = up + ev[s];
        }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s >= nvalid) continue;   // uniform
        // (keys are unique: the exclusive test loses nothing -- a key that IS the bound sits in the pool already, or came from another list)
        const u64 key = make_key(x[s], w8_sbase(s) + ps.pos);
        const u64 mask = __builtin_amdgcn_ballot_w64(ps.ok && ii == 7 && key < w8_sthr(s));
        if (mask != 0) w8_pool_offer(s, w8_pool_read(s, K, lane), key, mask, K, lane);   // uniform: most parked points pass for one query of the four
    }
}

// ---- the scan of one work item by one wave -----------------------------------------------------------------------------------------------
// K-th smallest integer sum of slot S over the step's four points per lane (radix select, as w8_kth_sum)
static __device__ __attribute__((noinline)) u32 w8_kth_sum4(u32 v0, u32 v1, u32 v2, u32 v3, u32 validbits, int need)
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

static __device__ __forceinline__ void w8_scan_range(__amdgpu_buffer_rsrc_t codes, u32 p0, u32 p1, int nvalid, int K, int wv, int lane,
                                                     v4u ca, v4u cb, __amdgpu_buffer_rsrc_t gt, W8Prof &pr)
{
    // A step of a wave is 256 points: four per lane in two 16-byte registers sets, ca (points pb + 2 lane, + 1) and cb (pb + 128 + 2 lane,
    // + 1), requested by the caller for the first step.  The code stream comes through a buffer resource over the list: the lane's offset
    // (16 lane) is a constant register, the step's offset a scalar -- a request is ONE instruction and no address arithmetic -- and each half
    // of the NEXT step is requested into its register set the moment this step's half has left it (rotated, four v_perm): two requests
    // of 1 KB per wave are in flight at any time, each with a whole step to arrive, and there is no second register set and no move.
    // (One request per wave -- 4 MB on the chip -- at the loaded latency of HBM is 2 TB/s: the conflict-free scan waited on every step.)
    constexpr u32 STEP = 256;
    const u32 cbuf_addr = W8Lds::PARK + (u32)wv * (W8_RING * W8_ES * 4u);
    u32 bias[2];
    w8_bias(nvalid, bias);
    // lane constants: byte rotation of a point's code (out byte t = code byte (t + j) mod 8) and the low address byte of slot t:
    // copy << 6 | ((t + j) mod 8) << 3
    const int j = lane & 7, cpy = (lane >> 3) & 3;
    u32 rsel0 = 0, rsel1 = 0, ap0 = 0, ap1 = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        rsel0 |= (u32)((b + j) & 7) << (8 * b);
        rsel1 |= (u32)((4 + b + j) & 7) << (8 * b);
        ap0 |= ((u32)((b + j) & 7) * 8u + (u32)cpy * 64u) << (8 * b);
        ap1 |= ((u32)((4 + b + j) & 7) * 8u + (u32)cpy * 64u) << (8 * b);
    }
    // address of slot t = perm{byte 0: lane part of slot t, byte 1: rotated code byte t, bytes 2, 3: zero}
    const u32 asel[4] = {0x0C0C0400u, 0x0C0C0501u, 0x0C0C0602u, 0x0C0C0703u};
    const int lane16 = lane * 16;
    int head = 0, ccnt = 0;
    u32 since = 0;
    bool pend = false;
    W8Pass ps;
    ps.ev = (v4f){0.f, 0.f, 0.f, 0.f};
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
        for (int s = 0; s < 4; ++s) {
            const float inv = w8_inv(s);
            const u32 hh = __builtin_amdgcn_readfirstlane(w8_lds<u32>(W8Lds::HARD + 8u * s + 4u));
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
                w8_pass_finish(ps, nvalid, K, lane);
                since = 0;
                w8_bias(nvalid, bias);   // (the other waves' offers moved the bounds as well)
                if (ccnt >= W8_TRIG || (ccnt > 0 && pb >= ptail)) {   // the next ones are waiting already (or the range ends)
                    W8_CNT(pr, 11, 8);
                    w8_pass_issue(ps, cbuf_addr, head, ccnt, gt, lane);
                    pend = true;
                }
                W8_ADD(pr, 5, td0);
            } else if (__builtin_expect(ccnt > 0 && pb >= ptail, 0)) {
                // the wave's last two steps: what is parked does not wait for company -- its pass is under way while these steps are
                // scanned, and the end of the range finds an empty ring nine times in ten (a pass worked off THERE is a trip to L2 the
                // wave sits out, with the other seven waiting for it at the barrier behind: the wait was 8 % of the kernel)
                W8_CNT(pr, 11, ccnt < 8 ? ccnt : 8);
                wave_sync();
                w8_pass_issue(ps, cbuf_addr, head, ccnt, gt, lane);
                pend = true;
            } else if (++since >= (u32)W8_REFRESH) {
                // the workgroup's bounds move even when this wave has no candidates of its own
                since = 0;
                w8_bias(nvalid, bias);
            }
            // the next step's offsets: past the end the wave's current halves are read once more (no branch around a request, no second
            // value for a register set to merge with; a half that starts beyond the list repeats the first one: never a byte beyond the
            // 127 points of slack the four-wave kernels read too)
            const u32 pn = pb + W8_NW * STEP;
            const u32 pa = pn < p1 ? pn : pb;
            const u32 pbb = pa + 128u < p1 ? pa + 128u : pa;
            u32 qa[4][2];
            auto half = [&](auto hc, v4u &cx, u32 pnext) __attribute__((always_inline)) {
                constexpr int h = decltype(hc)::value;
                // the half's bytes leave its register set rotated (tied together so that no part of them can sink below the request that
                // follows), and the next step's half is requested INTO it
                rw[2 * h][0] = __builtin_amdgcn_perm(cx.y, cx.x, rsel0);
                rw[2 * h][1] = __builtin_amdgcn_perm(cx.y, cx.x, rsel1);
                rw[2 * h + 1][0] = __builtin_amdgcn_perm(cx.w, cx.z, rsel0);
                rw[2 * h + 1][1] = __builtin_amdgcn_perm(cx.w, cx.z, rsel1);
                asm volatile("" : "+v"(rw[2 * h][0]), "+v"(rw[2 * h][1]), "+v"(rw[2 * h + 1][0]), "+v"(rw[2 * h + 1][1]), "+v"(cx));
#if defined(W8_KO) && (W8_KO & 8)
                asm volatile("" :: "s"(pnext));          // knock-out build: no code stream (every step scans the first step's bytes)
#else
                cx = __builtin_amdgcn_raw_buffer_load_b128(codes, lane16, (int)(pnext * 8u), W8_STREAM_AUX);
#endif
#if defined(W8_KO) && (W8_KO & 16)
                // knock-out build: the code stream alone (no table lookups: the fields are made of the bytes themselves)
                qa[2 * h][0] = bias[0] + rw[2 * h][0]; qa[2 * h][1] = bias[1] + rw[2 * h][1];
                qa[2 * h + 1][0] = bias[0] + rw[2 * h + 1][0]; qa[2 * h + 1][1] = bias[1] + rw[2 * h + 1][1];
                return;
#endif
                // all sixteen gathers of the half are issued before the first add (left alone the compiler waits after every second read)
                v2u ev[2][8];
                static_for<2>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    static_for<8>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        const u32 ea = w8_perm(rw[2 * h + r][t >> 2], t < 4 ? ap0 : ap1, asel[t & 3]);
                        ev[r][t] = lds_load_abs<v2u>(ea);
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    qa[2 * h + r][0] = bias[0];
                    qa[2 * h + r][1] = bias[1];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        qa[2 * h + r][0] += ev[r][t].x;
                        qa[2 * h + r][1] += ev[r][t].y;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            half(IntC<0>{}, ca, pa);
            half(IntC<1>{}, cb, pbb);
            // a field below 0x8000 <=> that query's integer sum is within its budget (w8_bias); one compare for the four points
            u32 x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = qa[r][0] & qa[r][1];
            u64 anym = __builtin_amdgcn_ballot_w64((((x[0] & x[1]) & (x[2] & x[3])) & 0x80008000u) != 0x80008000u);
#if defined(W8_KO) && (W8_KO & 1)
            asm volatile("" :: "s"(anym));
            anym = 0;   // knock-out build (wrong results by design): the filter's fast path alone
#endif
            W8_CNT(pr, 8, 1);
            if (__builtin_expect(first && coldmask != 0u, 0)) {   // uniform over the WORKGROUP: see above
                const int r8 = (K + W8_NW - 1) / W8_NW;
                static_for<4>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if ((coldmask >> s) & 1u) {   // uniform
                        // (the slot's bias is 0 while it has no bound: the fields are the sums)
                        u32 f[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] = (s & 1) ? (qa[r][s >> 1] >> 16) : (qa[r][s >> 1] & 0xffffu);
                        const u32 V = w8_kth_sum4(f[0], f[1], f[2], f[3], 0xFu, r8);
                        if (lane == 0) *w8_ptr<u32>(W8Lds::COLD + 64u * s + 4u * (u32)wv) = V;
                    }
                });
                __syncthreads();
                static_for<4>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if ((coldmask >> s) & 1u) {   // uniform
                        u32 T = 0;
#pragma unroll
                        for (int v = 0; v < W8_NW; ++v) {
                            const u32 o = __builtin_amdgcn_readfirstlane(w8_lds<u32>(W8Lds::COLD + 64u * s + 4u * (u32)v));
                            T = o > T ? o : T;
                        }
                        const float ub = (w8_dc(s) + (float)(T + 8u) * (1.00001f / w8_inv(s))) * 1.00002f;
                        // (every wave arrives at the same bound; the wave's own atomic is ahead of its own reads of the word)
                        if (ub < 3.0e38f && lane == 0) atomicMin(w8_ptr<u64>(W8Lds::STHR + 8u * s), make_key(ub, 0xFFFFFFFFu));
                    }
                });
                // the step's fields were accumulated under the old bias: re-based on the new one, and the step is tested again
                u32 nb[2];
                w8_bias(nvalid, nb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qa[r][0] = qa[r][0] - bias[0] + nb[0];
                    qa[r][1] = qa[r][1] - bias[1] + nb[1];
                    x[r] = qa[r][0] & qa[r][1];
                }
                bias[0] = nb[0];
                bias[1] = nb[1];
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
                    static_for<4>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        const float inv = w8_inv(s);
                        // (a scale that is not a normal number -- all-zero or denormal tables -- keeps the plain path)
                        if (s < nvalid && (u32)(w8_sthr(s) >> 32) >= 0x7F800000u && inv > 0.0f && inv < 1.0e30f) {   // uniform
                            // the sums themselves: field - bias (no borrow: every field started from its bias)
                            const u32 bs = (s & 1) ? (bias[s >> 1] >> 16) : (bias[s >> 1] & 0xffffu);
                            u32 f[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) f[r] = ((s & 1) ? (qa[r][s >> 1] >> 16) : (qa[r][s >> 1] & 0xffffu)) - bs;
                            const u32 U = w8_kth_sum4(f[0], f[1], f[2], f[3], vb, K);
                            const float ub = (w8_dc(s) + (float)(U + 8u) * (1.00001f / inv)) * 1.00002f;
                            if (ub < 3.0e38f) {
                                if (lane == 0) atomicMin(w8_ptr<u64>(W8Lds::STHR + 8u * s), make_key(ub, 0xFFFFFFFFu));
                                moved = true;
                            }
                        }
                    });
                    if (moved) {
                        W8_CNT(pr, 12, 1);
                        // the step's fields were accumulated under the old bias: re-based on the new one before they are tested again
                        u32 nb[2];
                        w8_bias(nvalid, nb);
                        ntot = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const u32 y = (qa[r][0] - bias[0] + nb[0]) & (qa[r][1] - bias[1] + nb[1]);
                            c[r] = (y & 0x80008000u) != 0x80008000u && ((vb >> r) & 1u) != 0u;
                            m[r] = __builtin_amdgcn_ballot_w64(c[r]);
                            n[r] = __popcll(m[r]);
                            ntot += n[r];
                        }
                        bias[0] = nb[0];
                        bias[1] = nb[1];
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
                            u32 *ent = w8_ptr<u32>(cbuf_addr + (u32)((base + rank) & (W8_RING - 1)) * (W8_ES * 4u));
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
                        w8_pass_issue(ps, cbuf_addr, head, ccnt, gt, lane);
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
                        u32 *ent = w8_ptr<u32>(cbuf_addr + (u32)((head + ccnt + rank) & (W8_RING - 1)) * (W8_ES * 4u));
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
                    w8_pass_finish(ps, nvalid, K, lane);
                    w8_bias(nvalid, bias);
                }
                if (ccnt > 0 && (more || flush || ccnt >= 8)) {
                    wave_sync();
                    w8_pass_issue(ps, cbuf_addr, head, ccnt, gt, lane);
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
__global__ __launch_bounds__(W8_THREADS, W8_NW / 2) void wg8_scan_kernel(const ScanArgs a, float *__restrict__ gtabs, const u32 *__restrict__ item_list,
                                                                 u32 *__restrict__ xq, int nranges)
{
    W8Prof pr;
#ifdef W8_PROF
    pr.zero();
    const u64 tk0 = __builtin_readcyclecounter();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IndexView &ix = a.ix;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // (a scalar: branches on the wave's number and its position in the list are scalar branches)
    const int K = a.K;
    float *res = (float *)(smem + W8Lds::RES);
    u32 *smax = (u32 *)(smem + W8Lds::SMAX);
    float *sinv = (float *)(smem + W8Lds::SMAX) + 4;
    float *sdc = (float *)(smem + W8Lds::QC);
    u32 *ssb = (u32 *)(smem + W8Lds::QC) + 4;
    u32 *spi = (u32 *)(smem + W8Lds::QC) + 8;
    u32 *sqi = (u32 *)(smem + W8Lds::QC) + 12;
    u64 *shard = (u64 *)(smem + W8Lds::HARD);
    u64 *sthr = (u64 *)(smem + W8Lds::STHR);
    u32 *swi = (u32 *)(smem + W8Lds::SWI);
    const u32 total = a.wi_off[ix.kc];
    float *gt = gtabs + (size_t)blockIdx.x * W8_GTAB_FLOATS;
    const __amdgpu_buffer_rsrc_t gtr = __builtin_amdgcn_make_buffer_rsrc((void *)gt, 0, (int)(W8_GTAB_FLOATS * 4u), 0x00020000);

    u64 *pool = (u64 *)(smem + W8Lds::POOL);
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
        const u32 ng = (cnt + 3u) / 4u;
        const u32 local = wi - __builtin_amdgcn_readfirstlane(a.wi_off[l]);
        const u32 chunk = local / ng, grp = local - chunk * ng;
        const u32 len = __builtin_amdgcn_readfirstlane(ix.list_len[l]);
        const u32 p0 = chunk * a.CH;
        if (p0 >= len) break;   // uniform
        const u32 p1 = min(len, p0 + a.CH);
        const int nvalid = min(4, (int)(cnt - grp * 4u));

        // the queries of the group: thread s < 4 fetches slot s (slots past nvalid repeat slot 0 and can never be candidates)
        if (tid < 4) {
            const int ss = tid < nvalid ? tid : 0;
            const u32 pi = a.bucket_items[a.bucket_off[l] + grp * 4u + ss];
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
        if (tid >= 256 && tid < 512) pool[tid - 256] = KEY_MAX;
        __syncthreads();
        // exact pruning of whole work items, as in scan_kernel: no sum of this list lies below its coarse distance
        if (a.prune) {
            bool all = true;
#pragma unroll
            for (int s = 0; s < 4; ++s)
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
        // (1) residuals r_s = q_s - c (coarsequantizers.jl:40-45), one element per thread: res[ii][t][s], 17 rows of four per sub-quantizer
        {
            const int tb = tid & 511, i = tb >> 2, s = tb & 3;
            res[tb + (tb >> 6) * 4] = a.queries[(size_t)sqi[s] * 128 + i] - ix.centroids[(size_t)l * 128 + i];
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
        v4f ent[4];
#if defined(W8_KO) && (W8_KO & 4)
        if (K > 0) {                  // knock-out build: no table build (the scan runs on made-up entries)
#pragma unroll
            for (int k = 0; k < 4; ++k) ent[k] = (v4f){(float)(k + cg), (float)(k + 2 * cg), (float)cg, 1.0f};
            if (tid < 4) smax[tid] = __float_as_uint(600.0f);
        } else
#endif
        {
            v2f sum[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) sum[j][0] = sum[j][1] = (v2f){0.0f, 0.0f};
            const u32 roff = W8Lds::RES + (u32)ii * 272u;
            // (the rows of a trip -- eight dimensions -- are requested together at its top: a request waits ~1 000 cycles in the LDS queue behind
            // the other workgroup's gathers, and the build pays that wait once per batch)
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
                for (int t = 0; t < 8; ++t) rv[t] = w8_lds<v4f>(roff + (u32)(8 * h + t) * 16u);
                ldcw(cwb, 2 * h + 1);
                grp(cwa, 0);
                ldcw(cwa, h == 0 ? 2 : 3);      // (the second trip repeats a request: no branch around one, no second value to merge)
                grp(cwb, 1);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) ent[j] = (v4f){sum[j][0].x, sum[j][0].y, sum[j][1].x, sum[j][1].y};
            float mx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = cg + 64 * j;
                const int label = ix.identity_labels ? c : (int)ix.labels[ii * 256 + c];
                *(v4f *)(gt + ((size_t)(ii * 256 + label) << 2)) = ent[j];
                mx[0] = fmaxf(mx[0], ent[j].x);
                mx[1] = fmaxf(mx[1], ent[j].y);
                mx[2] = fmaxf(mx[2], ent[j].z);
                mx[3] = fmaxf(mx[3], ent[j].w);
            }
            // (entries are >= +0: the bit pattern orders like the value.  The wave's maximum on the DPP network and the scalar unit: a shuffle
            // is a trip through the LDS queue -- ~1 000 cycles behind the other workgroup's gathers, six of them in a row per query)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32 wm = w8_row_max_u32(__float_as_uint(mx[s]));
                if (lane == 0) atomicMax(&smax[s], wm);
            }
        }
        __syncthreads();
        W8_ADD(pr, 15, tb0);   // (of the build: up to the barrier behind the entries)
        // (3) quantise (quantize_tables_m8's rule: q = min(4095, floor(t * inv)), inv = 4095 / largest entry of the query) and write the four
        // copies: copy (cp + lane / 4) mod 4 of sub-quantizer ii -- the 16 lanes of a store's service group write 16 different bank pairs
        // (consecutive labels are 256 B apart: the same banks).
        {
            float inv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float mxs = __uint_as_float(smax[s]);
                inv[s] = mxs > 0.0f ? 4095.0f / mxs : 0.0f;
            }
            if (tid < 4) sinv[tid] = inv[tid];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ev[4] = {ent[j].x, ent[j].y, ent[j].z, ent[j].w};
                u32 f[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const u32 v = (u32)floorf(ev[s] * inv[s]);
                    f[s] = v < 4095u ? v : 4095u;
                }
                const uint2 qv = make_uint2(f[0] | (f[1] << 16), f[2] | (f[3] << 16));
                const int c = cg + 64 * j;
                const int lb = ix.identity_labels ? c : (int)ix.labels[ii * 256 + c];
                const u32 row = ((u32)lb << 8) | ((u32)ii << 3);
#pragma unroll
                for (int cp = 0; cp < 4; ++cp) *(uint2 *)(smem + (row | ((u32)((cp + (lane >> 2)) & 3) << 6))) = qv;
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
        w8_scan_range(codes, p0, p1, nvalid, K, wv, lane, ca, cb, gtr, pr);
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
                const u64 kth = w8_wave_max_u64(v);
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
        u64 *dst = (u64 *)(gtabs + (size_t)gridDim.x * W8_GTAB_FLOATS);
        for (int i = 0; i < 16; ++i) atomicAdd(dst + i, pr.c[i]);
    }
#endif
}
