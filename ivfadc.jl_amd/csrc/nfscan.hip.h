// nfscan.hip.h -- list-major scan with a carry-free NARROW-FIELD integer filter, eight queries per code stream (m = 8, dsub = 16,
// ksub = 256: the SIFT1B shape).  Included by kernels.hip.h, namespace ivf.
//
// Reference: src/index.jl:232-236 (table build), :240-246 (scan), :247-254 (bounded top-K).
//
// The round-2 list-major kernel (scan_kernel<8, 16, 4, ..., STRIPE>) filters with 16-bit integer fields, four queries per 8-byte
// table entry, behind resident f32 tables: 48 KB of LDS, three workgroups per CU, and -- measured, round 3 -- LDS 81 % busy with
// 67 % of its cycles bank-conflict replays, vector ALU 74 % busy, every list streamed once per FOUR queries.  Here:
//
//   * EIGHT queries share a 16-byte table entry (eight 16-bit fields of 12-bit quantised entries: the sum of a point's 8 entries is
//     <= 32 760 < 2^15, fields never carry), so a lookup is one ds_read_b128 and four v_add_u32 for eight queries and every list
//     is streamed once per eight queries;
//   * the accumulator starts at bias_s = 0x7FFF - T_s per field (T_s = the integer budget the bound of query s leaves), so
//     "field < 0x8000" <=> "sum_s <= T_s": the candidate test of a point for all eight queries is three ANDs and a compare
//     -- bias + sum <= 0x8000 + 32 760 < 2^16: no wrap, no carry, no false negatives;
//   * the table holds TWO private copies of every entry and lane l looks sub-quantizer (t + l) mod 8 up at slot t in copy
//     (l / 8) mod 2: entry (code, ii, copy) sits at byte code * 256 + ii * 32 + copy * 16, so the 16 lanes of a ds_read_b128
//     service group (their lane numbers are all different mod 16) read 16 different 4-bank groups whatever their codes are --
//     conflict-free gathers -- and because a code's row is exactly 256 bytes the whole address (code << 8 | lane part) is ONE
//     v_perm_b32 of the rotated code dword and a lane-constant dword;
//   * no f32 tables stay resident (64 KB of LDS hold the two copies; two workgroups per CU): whatever the filter lets through is
//     parked as a (point, query) pair and gets its reference-order sum from the f32 codebook -- r = q - c (coarsequantizers.jl:
//     40-45), df = cb - r, T += df * df for t ascending (index.jl:234), dc + T_0 + ... + T_7 in ascending order (index.jl:242-246)
//     -- by four lanes per pair (lbscan.hip.h's scheme).  Only these sums meet the selectors: ids and distances stay bit-identical.
//
// (First built with 4-bit entries in one BYTE per query -- eight queries per 8-byte entry, four copies, ds_read_b64, sums < 128 --:
// the fast path ran at the HBM rate, 4.3 ms for the SIFT1B batch, but 15 levels let 9 400 (point, query) pairs per query through to
// the exact sums instead of a few tens -- the lower tail of the ADC sums is steep: a simulation of the shape gives 700 / 74 / 27 / 13 /
// 10 survivors per query at 4 / 5 / 6 / 8 / 12 bits under the FINAL bound, more while bounds are loose -- and the launch took 24 ms.)
//
// Lower bound.  With E = ||cb - r||^2 in real arithmetic on the f32 operands the reference uses, N = ||cb||^2 + ||r||^2 (2 |cb.r| <= N),
// u = 2^-24, the build evaluates  e = fma(-2, dot, n2 (1 - 2^-17) + rn2 (1 - 2^-17))  with dot a 16-term fma chain (error <= 8 u N),
// n2 = ||cb||^2 rounded from double (1 u), rn2 a 16-term fma chain (16 u), two more roundings (<= 3 u N): |e - (E - 2^-17 N)| <= 36 u N
// < 2^-18.8 N, so e <= E - 2^-18 N.  base_ii = min over the codes of e (exact: a wave reduction), inv_s = 4095 / R_s, and
// q = min(trunc(fl(fl(e - base) inv)), 4095) <= (e - base) inv (1 + 3 u) <= (E - base) inv   (the 2^-18 N pays for the 3 u).
// That is lbscan.hip.h's premise q_i <= (E_i - base_i) inv, so its integer target lb_target(bound, dc, sum of bases, inv) applies
// unchanged: every point whose REFERENCE sum is <= the bound has sum q <= T_s.  R_s is the range of the entries, or -- when the
// query already has a bound at the start of the work item -- the budget that bound leaves, if smaller: entries beyond the budget
// saturate at 4095 > T_s, which is all they need to say (bounds only ever tighten within a search).  Degenerate tables (range
// < 1e-30, or nothing finite) get inv = 0: every field is 0, every point passes and is evaluated exactly -- slow, and correct.
#pragma once

constexpr int NF_QG = 8;                       // queries per code stream
constexpr u32 NF_TAB_BYTES = 256u * 256u;      // 256 codes x (8 sub-quantizers x 2 copies x 16 bytes)
constexpr int NF_POOL = 128;                   // parked (point, query) pairs per wave; drained from 64 on
constexpr int NF_ES = 4;                       // dwords per parked pair: code bytes (2), list position, integer sum << 8 | query slot

// LDS behind the table (byte offsets)
struct NfLds {
    static constexpr u32 RES = NF_TAB_BYTES;                  // f32 residuals [s][t4][ii][4]: 8 x 4 x 8 x 16 B = 4 KB
    static constexpr u32 RN2 = RES + 4096u;                   // f32 [ii][s]: ||r_s,ii||^2 (1 - 2^-17)
    static constexpr u32 SMIN = RN2 + 256u;                   // u32 [ii][s]: ordered bits of min_c e          (atomic)
    static constexpr u32 SMAX = SMIN + 256u;                  // u32 [ii][s]: ordered bits of max_c e          (atomic)
    static constexpr u32 BASE = SMAX + 256u;                  // f32 [ii][s]: base
    static constexpr u32 QC = BASE + 256u;                    // f32 [4][8]: inv_s, sum of bases, dc_s, nn_s (sum over ii of max ||cb||^2 + ||r||^2)
    static constexpr u32 QI = QC + 128u;                      // u32 [3][8]: probe index, query, visit-order base of slot s
    static constexpr u32 HARD = QI + 96u;                     // u64 [8]: the bounds the item started from (from outside the workgroup)
    static constexpr u32 STHR = HARD + 64u;                   // u64 [8]: workgroup-shared bounds
    static constexpr u32 SCNT = STHR + 64u;                   // int [4][8]
    static constexpr u32 SWI = SCNT + 128u;                   // u32 [4]
    static constexpr u32 TW = SWI + 16u;                      // int [4][8]: each wave's integer budgets of the moment (read by lane-varying slot)
    static constexpr u32 POOL = TW + 128u;                    // u32 [4][NF_POOL][NF_ES]
    static constexpr u32 END = POOL + 4u * NF_POOL * NF_ES * 4u;
};
static_assert(NfLds::END <= 80u * 1024u, "two workgroups per CU");
static_assert((NfLds::POOL & 15u) == 0, "16-byte pool entries");
static_assert((NfLds::HARD & 7u) == 0, "8-byte bounds");

static __device__ __forceinline__ u32 nf_perm(u32 s0, u32 s1, u32 sel)
{
    u32 o;
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(o) : "v"(s0), "v"(s1), "s"(sel));
    return o;
}

// ---- table build ---------------------------------------------------------------------------------------------------------------
// Thread c = codeword index; at step k the lane works on sub-quantizer (k + lane) mod 8 (so that the later table writes of a
// wave-instruction fall into different banks).  e[k][s]: lower-bound entry of (sub-quantizer, codeword) for query s.
struct NfBuild {
    float e[8][NF_QG];
};

// reduce e over the codes (min when MIN, else max) and publish per (sub-quantizer, query) through LDS atomics on ordered bits.
// Stage 1 / 2: v_permlane32_swap / v_permlane16_swap exchange halves / odd-even rows between two registers -- min(a', b') then
// holds value A reduced in one half and value B in the other: a transposing reduction, one instruction per value.  Stage 3:
// row_ror:8 pairs lane l with l ^ 8; bank_mask keeps value A's result in lanes 0-7 and value B's in lanes 8-15 of every row.
// Afterwards lane l holds, for s = 0 .. 7, the reduction over the 8 lanes of its class (l mod 8) of e[k0][s], k0 = (l / 8) mod 8.
template <bool MIN> static __device__ __forceinline__ float nf_op(float a, float b) { return MIN ? fminf(a, b) : fmaxf(a, b); }
template <bool MIN> static __device__ __forceinline__ void nf_reduce_publish(const NfBuild &b, unsigned char *smem, int lane)
{
    // value index v = k * 8 + s.  stage 1 pairs v with v + 32 (k with k + 4): lanes < 32 keep k < 4
    float x[32];
#pragma unroll
    for (int v = 0; v < 32; ++v) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(b.e[v >> 3][v & 7]), __float_as_uint(b.e[(v >> 3) + 4][v & 7]), false, false);
        x[v] = nf_op<MIN>(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    // stage 2 pairs v with v + 16 (k with k + 2): even rows keep the lower one
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[v]), __float_as_uint(x[v + 16]), false, false);
        x[v] = nf_op<MIN>(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    // stage 3 pairs v with v + 8 (k with k + 1): lanes 0-7 of a row keep the lower one
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const float a = x[v], bb = x[v + 8];
        const float pa = __uint_as_float((u32)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(a), 0x128, 0xf, 0xf, false));    // row_ror:8
        const float pb = __uint_as_float((u32)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(bb), 0x128, 0xf, 0xf, false));
        x[v] = (lane & 8) ? nf_op<MIN>(bb, pb) : nf_op<MIN>(a, pa);
    }
    // lane l now holds k0 = ((l >> 5) & 1) * 4 + ((l >> 4) & 1) * 2 + ((l >> 3) & 1), sub-quantizer (k0 + l) mod 8, queries s = 0 .. 7
    const int k0 = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
    const int ii = (k0 + lane) & 7;
    u32 *dst = (u32 *)(smem + (MIN ? NfLds::SMIN : NfLds::SMAX)) + ii * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (MIN) atomicMin(&dst[s], ordered_bits(x[s]));
        else atomicMax(&dst[s], ordered_bits(x[s]));
    }
}

// pass 1: e[k][s] for the thread's codeword (all 256 threads; residuals, norms stand in LDS)
static __device__ __forceinline__ void nf_entries(const IndexView &ix, const float *n2, const unsigned char *smem, int tid, int lane, NfBuild &b)
{
    const float4 *ct = (const float4 *)ix.codebooks_t;        // [ii][g][c][4], ksub = 256
    const float4 *rs = (const float4 *)(smem + NfLds::RES);   // [s][t4][ii]
    const float *rn2 = (const float *)(smem + NfLds::RN2);    // [ii][s]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ii = (k + lane) & 7;
        float4 cw[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) cw[g] = ct[(size_t)(ii * 4 + g) * 256 + tid];
        const float nn = n2[ii * 256 + tid] * 0.99999237060546875f;   // (1 - 2^-17)
#pragma unroll
        for (int s = 0; s < NF_QG; ++s) {
            float dot = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 r4 = rs[(s * 4 + g) * 8 + ii];
                dot = __builtin_fmaf(cw[g].x, r4.x, dot);
                dot = __builtin_fmaf(cw[g].y, r4.y, dot);
                dot = __builtin_fmaf(cw[g].z, r4.z, dot);
                dot = __builtin_fmaf(cw[g].w, r4.w, dot);
            }
            b.e[k][s] = __builtin_fmaf(-2.0f, dot, nn + rn2[ii * 8 + s]);
        }
        __builtin_amdgcn_sched_barrier(0);   // one sub-quantizer's loads at a time: hoisting all 256 residual reads costs the registers
    }
}

// pass 2: quantise and write the two copies.  label = table row of the thread's codeword in sub-quantizer ii.
static __device__ __forceinline__ void nf_quantize_store(const IndexView &ix, unsigned char *smem, int tid, int lane, const NfBuild &b)
{
    const float *base = (const float *)(smem + NfLds::BASE);  // [ii][s]
    const float *qc = (const float *)(smem + NfLds::QC);      // inv[8]
    float inv[NF_QG];
#pragma unroll
    for (int s = 0; s < NF_QG; ++s) inv[s] = qc[s];
    const int c0 = (lane >> 3) & 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ii = (k + lane) & 7;
        const float4 b0 = *(const float4 *)(base + ii * 8), b1 = *(const float4 *)(base + ii * 8 + 4);
        const float bs[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        u32 qv[NF_QG];
#pragma unroll
        for (int s = 0; s < NF_QG; ++s) {
            // trunc(y) <= y for y >= 0; negative and NaN products convert to 0, huge ones saturate and are clamped
            const float y = inv[s] > 0.0f ? (b.e[k][s] - bs[s]) * inv[s] : 0.0f;
            const u32 v = (u32)y;
            qv[s] = v < 4095u ? v : 4095u;
        }
        const uint4 ent = make_uint4(qv[0] | (qv[1] << 16), qv[2] | (qv[3] << 16), qv[4] | (qv[5] << 16), qv[6] | (qv[7] << 16));
        const int label = ix.identity_labels ? tid : (int)ix.labels[ii * 256 + tid];
        unsigned char *row = smem + (u32)label * 256u + (u32)ii * 32u;
        *(uint4 *)(row + (u32)c0 * 16u) = ent;
        *(uint4 *)(row + (u32)(c0 ^ 1) * 16u) = ent;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- exact sums of parked (point, query) pairs, 16 per pass (lbscan.hip.h's lb_drain with the residuals in LDS) -------------------------
// Lane 4 e + part: pair e; trip i: the quad holds the reference's table entries of sub-quantizers 4 i .. 4 i + 3, the running sum walks
// along the quad (row_shr:1) and the quad's last lane hands it back.
static __device__ __forceinline__ void nf_drain(const u32 *pool, int cnt, const float *cb_lab, const unsigned char *smem, WSel<true> (&sel)[NF_QG],
                                                int K, int lane)
{
    const float4 *rs = (const float4 *)(smem + NfLds::RES);
    const float *dcs = (const float *)(smem + NfLds::QC) + 16;
    const u32 *sbase = (const u32 *)(smem + NfLds::QI) + 16;
    const int part = lane & 3;
    for (int b0 = 0; b0 < cnt; b0 += 16) {   // uniform
        const int e = b0 + (lane >> 2);
        const bool ok = e < cnt;
        const uint4 ent = *(const uint4 *)(pool + (size_t)(ok ? e : 0) * NF_ES);
        const int s = (int)(ent.w & 7u);                 // (the integer sum sits above bit 8)
        float run = dcs[s];
        float x = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ii = 4 * i + part;
            const u32 byte = ((i == 0 ? ent.x : ent.y) >> (8 * part)) & 0xffu;
            const float4 *cw = (const float4 *)(cb_lab + ((size_t)ii * 256 + byte) * 16);
            float4 c4[4], r4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) { c4[g] = cw[g]; r4[g] = rs[(s * 4 + g) * 8 + ii]; }
            float T = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // df = cb - r, T += df * df for t ascending (index.jl:234, colwise SqEuclidean); r = q - c stands in LDS
                float df = c4[g].x - r4[g].x; T = T + df * df;
                df = c4[g].y - r4[g].y; T = T + df * df;
                df = c4[g].z - r4[g].z; T = T + df * df;
                df = c4[g].w - r4[g].w; T = T + df * df;
            }
            x = run + T;
#pragma unroll
            for (int p = 1; p < 4; ++p) {
                const float up = __uint_as_float((u32)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x111, 0xf, 0xf, false));   // row_shr:1
                x = up + T;
            }
            run = __uint_as_float((u32)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0xFF, 0xf, 0xf, false));   // quad_perm:[3,3,3,3]
        }
        const bool mine = ok && part == 3;
        const u64 key = make_key(x, sbase[s] + ent.z);
#pragma unroll
        for (int t = 0; t < NF_QG; ++t) {
            const bool pred = mine && s == t && key < sel[t].thr();
            if (__builtin_amdgcn_ballot_w64(pred)) sel[t].push(pred, key, K, lane);
        }
    }
}

// ---- the scan of one work item --------------------------------------------------------------------------------------------------------
// integer target of query slot s under a bound (lbscan.hip.h's lb_target on the slot's constants, read from LDS)
static __device__ __forceinline__ int nf_target(const unsigned char *smem, int s, u32 thr_hi)
{
    const float *qc = (const float *)(smem + NfLds::QC);
    return lb_target(thr_hi, qc[16 + s], qc[8 + s], qc[s]);
}

// per wave: the integer budgets of the eight queries under the bounds of the moment, and the accumulator bias that encodes them.
// The budgets also stand in a wave-private LDS row: candidates look theirs up by a lane-varying slot.
struct NfTargets {
    u32 b0, b1, b2, b3;   // 16-bit field s = 0x7FFF - T[s];  T = -1: nothing passes, 0x7FFF: everything does.  (Scalars, not an array: a
                          // lane-varying pick from an array member is turned into an indexed scratch load.)
    __device__ __forceinline__ u32 word(u32 h) const { return h == 0u ? b0 : (h == 1u ? b1 : (h == 2u ? b2 : b3)); }
    __device__ __forceinline__ void set(unsigned char *smem, int nvalid, WSel<true> (&sel)[NF_QG], int wv, int lane)
    {
        int *tw = (int *)(smem + NfLds::TW) + wv * NF_QG;
        u32 f[NF_QG];
        wave_sync();
#pragma unroll
        for (int s = 0; s < NF_QG; ++s) {
            const int T = __builtin_amdgcn_readfirstlane(s < nvalid ? nf_target(smem, s, (u32)(sel[s].thr() >> 32)) : -1);
            f[s] = (u32)(0x7FFF - T);
            if (lane == 0) tw[s] = T;
        }
        b0 = f[0] | (f[1] << 16);
        b1 = f[2] | (f[3] << 16);
        b2 = f[4] | (f[5] << 16);
        b3 = f[6] | (f[7] << 16);
        wave_sync();
    }
    static __device__ __forceinline__ int of(const unsigned char *smem, int wv, int s) { return ((const int *)(smem + NfLds::TW))[wv * NF_QG + s]; }
};

// The pool of one wave is worked on -- it needs room, or the item ends (flush):
//   1. pairs not offered yet give their UPPER bound to the upper-bound selector of their query (flag 0x80 of the slot byte: a pair is
//      offered once -- a key held twice would count one point as two).  The integer sums bound the distances from above as well (no entry
//      of a candidate is saturated: see the header), so the K-th smallest upper bound is a bound that K real points meet, found
//      without one exact sum:  S <= (dc + sum of bases + (Q + 8)(1 + 3 u) / inv + 2^-16.5 NN)(1 + 40 u);
//   2. pairs the bounds of the moment rule out are dropped (the filter's own test on the stored integer sums: most of what was parked
//      under an older, looser bound goes);
//   3. if the pool is still more than half full (or at a flush) the newest pairs get their exact sums, 64 at a time.
static __device__ __forceinline__ void nf_make_room(u32 *pool, int &pcnt, const float *cb_lab, unsigned char *smem, int nvalid,
                                                    WSel<true> (&sel)[NF_QG], WSel<true> (&usel)[NF_QG], int K, int wv, int lane, u32 &nsurv, bool flush,
                                                    bool count_exact = true)
{
    const float *qc = (const float *)(smem + NfLds::QC);
    const u32 *sbase = (const u32 *)(smem + NfLds::QI) + 16;
    wave_sync();
    if (flush && pcnt <= 16) {
        // a handful of pairs at the end of an item (the common case once bounds are tight): one pass of exact sums costs less than
        // offering, re-targeting and compacting them first
        if (count_exact) nsurv += (u32)pcnt;
        nf_drain(pool, pcnt, cb_lab, smem, sel, K, lane);
        pcnt = 0;
        wave_sync();
        return;
    }
    for (int b0 = 0; b0 < pcnt; b0 += 64) {   // uniform
        const int e = b0 + lane;
        const bool have = e < pcnt;
        u32 *slot = pool + (size_t)(have ? e : 0) * NF_ES;
        const uint4 ent = *(const uint4 *)slot;
        const int s = (int)(ent.w & 7u);
        const bool fresh = have && (ent.w & 0x80u) == 0u;
        const float inv = qc[s];
        const float ub = (((qc[16 + s] + qc[8 + s]) + (float)((ent.w >> 8) + 8u) * (1.000001f / inv)) + qc[24 + s] * 2.0e-5f) * 1.0000153f;
        const u64 key = make_key(ub, sbase[s] + ent.z);
        if (fresh) slot[3] = ent.w | 0x80u;
#pragma unroll
        for (int t = 0; t < NF_QG; ++t) {
            const bool pred = fresh && s == t && inv > 0.0f && key < usel[t].thr();
            if (__builtin_amdgcn_ballot_w64(pred)) usel[t].push(pred, key, K, lane);
        }
    }
#pragma unroll
    for (int t = 0; t < NF_QG; ++t) {
        const u64 ut = usel[t].thr();
        if ((u32)(ut >> 32) < 0x7F800000u) sel[t].tighten(ut | 0xFFFFFFFFull);   // K upper bounds stand
    }
    for (;;) {   // uniform
        wave_sync();
        NfTargets tg;
        tg.set(smem, nvalid, sel, wv, lane);
        int kept = 0;
        for (int b0 = 0; b0 < pcnt; b0 += 64) {   // uniform; survivors only move towards the front (kept <= b0)
            const int e = b0 + lane;
            const bool have = e < pcnt;
            const u32 *src = pool + (size_t)(have ? e : 0) * NF_ES;
            const u32 e0 = src[0], e1 = src[1], e2 = src[2], e3 = src[3];
            const u64 keep = __builtin_amdgcn_ballot_w64(have && (int)(e3 >> 8) <= NfTargets::of(smem, wv, (int)(e3 & 7u)));
            wave_sync();   // the block's entries are in registers before one is overwritten
            if ((keep >> lane) & 1ull) {
                u32 *dst = pool + (size_t)(kept + __popcll(keep & ((1ull << lane) - 1ull))) * NF_ES;
                dst[0] = e0, dst[1] = e1, dst[2] = e2, dst[3] = e3;
            }
            kept += __popcll(keep);
            wave_sync();
        }
        pcnt = kept;
        if (pcnt == 0 || (!flush && pcnt <= NF_POOL / 2)) break;
        const int take = pcnt < 64 ? pcnt : 64;
        if (count_exact) nsurv += (u32)take;
        nf_drain(pool + (size_t)(pcnt - take) * NF_ES, take, cb_lab, smem, sel, K, lane);
        pcnt -= take;
        if (!flush) break;   // (a flush goes round: the exact sums have tightened the bounds, the rest is re-tested first)
    }
    wave_sync();
}

template <int PPL>
static __device__ __forceinline__ void nf_scan_range(const uint8_t *cbase, u32 p0, u32 p1, int nvalid, WSel<true> (&sel)[NF_QG], int K, int wv,
                                                     int lane, CodeRegs<8, PPL> cr, unsigned char *smem, const float *cb_lab, u32 &nsurv, int dbg_flags)
{
    using CR = CodeRegs<8, PPL>;
    constexpr u32 STEP = CR::STEP;
    u64 *sthr = (u64 *)(smem + NfLds::STHR);
    u32 *pool = (u32 *)(smem + NfLds::POOL) + (size_t)wv * NF_POOL * NF_ES;
    int pcnt = 0;
    WSel<true> usel[NF_QG];   // K smallest upper bounds this wave has seen, per query (nf_make_room)
#pragma unroll
    for (int s = 0; s < NF_QG; ++s) {
        usel[s].init(KEY_MAX, nullptr, 64, K);
        sel[s].tighten(readfirstlane64(sthr[s]));
    }
    NfTargets tg;
    tg.set(smem, nvalid, sel, wv, lane);
    // lane constants: byte rotation of a point's code (out byte t = code byte (t + j) mod 8), and the low address byte of slot t
    const int j = lane & 7, cpy = (lane >> 3) & 1;
    u32 rsel0 = 0, rsel1 = 0, ap0 = 0, ap1 = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        rsel0 |= (u32)((b + j) & 7) << (8 * b);
        rsel1 |= (u32)((4 + b + j) & 7) << (8 * b);
        ap0 |= ((u32)((b + j) & 7) * 32u + (u32)cpy * 16u) << (8 * b);
        ap1 |= ((u32)((4 + b + j) & 7) * 32u + (u32)cpy * 16u) << (8 * b);
    }
    // address of slot t = perm{byte 0: lane part of slot t, byte 1: rotated code byte t, bytes 2, 3: zero}
    const u32 asel[4] = {0x0C0C0400u, 0x0C0C0501u, 0x0C0C0602u, 0x0C0C0703u};

    // other waves' bounds arrive through LDS; a wave's own improvements leave through it.  Returns whether a bound of this wave moved.
    auto exchange = [&]() __attribute__((always_inline)) {
        u32 moved = 0;
#pragma unroll
        for (int s = 0; s < NF_QG; ++s) {
            const u64 mine = sel[s].thr();
            const u64 shared = readfirstlane64(sthr[s]);
            if (lane == 0 && mine < shared) atomicMin(&sthr[s], mine);
            sel[s].tighten(shared);
            moved |= (u32)(mine >> 32) ^ (u32)(sel[s].thr() >> 32);
        }
        return moved != 0;
    };

    // the code stream runs TWO steps ahead (with two waves per SIMD a wave's step lasts about a microsecond: one step of distance does not
    // cover a trip to HBM under load)
    u32 since = 0;
    CR cr1;
    {
        const u32 pn = p0 + wv * STEP + 4 * STEP;
        if (pn < p1) cr1.load(cbase, pn, lane);
        else cr1 = cr;
    }
    for (u32 pb = p0 + wv * STEP; pb < p1; pb += 4 * STEP) {
        CR nx;
        const u32 pn = pb + 8 * STEP;
        if (pn < p1) nx.load(cbase, pn, lane);
        else nx = cr;
        if (++since == 8u) {
            since = 0;
            if (exchange()) tg.set(smem, nvalid, sel, wv, lane);
        }
        u32 pw[PPL][2], rw[PPL][2], acc[PPL][4];
        static_for<PPL>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            cr.words(r, pw[r]);
            rw[r][0] = __builtin_amdgcn_perm(pw[r][1], pw[r][0], rsel0);
            rw[r][1] = __builtin_amdgcn_perm(pw[r][1], pw[r][0], rsel1);
#pragma unroll
            for (int h = 0; h < 4; ++h) acc[r][h] = h == 0 ? tg.b0 : (h == 1 ? tg.b1 : (h == 2 ? tg.b2 : tg.b3));
        });
        // lookups in groups of GT slots x PPL points, two groups in flight: left to itself the compiler waits for every second read
        // (s_waitcnt lgkmcnt(1) before each pair of adds), and with two waves per SIMD nothing covers the LDS round trip
        constexpr int GT = PPL >= 8 ? 1 : (PPL >= 4 ? 2 : 4), NG = 8 / GT;
        v4u ev[2][GT * PPL];
        auto issue = [&](auto gc, v4u (&dst)[GT * PPL]) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            static_for<GT>([&](auto kc) {
                constexpr int t = g * GT + decltype(kc)::value;
                static_for<PPL>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const u32 ea = nf_perm(rw[r][t >> 2], t < 4 ? ap0 : ap1, asel[t & 3]);
                    dst[decltype(kc)::value * PPL + r] = lds_load_abs<v4u>(ea);
                });
            });
        };
        issue(IntC<0>{}, ev[0]);
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 1 < NG) issue(IntC<g + 1>{}, ev[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < GT; ++k)
#pragma unroll
                for (int r = 0; r < PPL; ++r) {
                    const v4u e = ev[g & 1][k * PPL + r];
                    acc[r][0] += e.x;
                    acc[r][1] += e.y;
                    acc[r][2] += e.z;
                    acc[r][3] += e.w;
                }
            __builtin_amdgcn_sched_barrier(0);
        });
        u64 any = 0;
        u64 cm[PPL];
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
            // a field below 0x8000 <=> that query's integer sum is within its budget
            const bool c = ((acc[r][0] & acc[r][1] & acc[r][2] & acc[r][3]) | 0x7FFF7FFFu) != 0xFFFFFFFFu;
            cm[r] = __builtin_amdgcn_ballot_w64(c && CR::point(pb, r, lane) < p1);
            any |= cm[r];
        }
#ifdef IVFADC_DEBUG
        if (dbg_flags & 1) any = 0;   // knock-out (wrong results by design): the filter's fast path alone
#endif
#ifdef IVFADC_DEBUG
        const u64 tc0 = (dbg_flags & 512) ? (u64)__builtin_readcyclecounter() : 0ull;
#endif
        if (any) {   // uniform, rare once the bounds are tight
#ifdef IVFADC_DEBUG
            if (dbg_flags & 8) nsurv += 1u;    // diagnostic counters instead of the exact-sum count: steps with candidates
#endif
            // Candidates are only PARKED here -- (code bytes, position, integer sum, query) -- and worked on in batches when the pool fills
            // (nf_make_room): a streaming selection moves its bound ~K ln(N / K) times per query, and paying selector insertions,
            // target and bias updates per move was most of this kernel's time before.
            // This is cold code, and its SIZE is what it costs: with one copy of the parking loop (and of nf_make_room inlined in it) per point
            // of a lane, an event walked four stretches of code 20 KB apart and waited for the instruction cache on each -- a third of the
            // kernel's wave-cycles, measured.  Hence ONE copy, which the PPL points of a lane go through in turn.  (Scalars and explicit
            // selects throughout: an array indexed by a lane-varying slot lands in scratch memory.)
            const u32 bs0 = tg.b0, bs1 = tg.b1, bs2 = tg.b2, bs3 = tg.b3;   // the fields were accumulated under these
            // bit s: field s of (g0..g3) is below 0x8000
            auto below = [](u32 g0, u32 g1, u32 g2, u32 g3) __attribute__((always_inline)) {
                return (((~g0) >> 15) & 1u) | (((~g0) >> 30) & 2u) | (((~g1) >> 13) & 4u) | (((~g1) >> 28) & 8u) | (((~g2) >> 11) & 16u) |
                       (((~g2) >> 26) & 32u) | (((~g3) >> 9) & 64u) | (((~g3) >> 24) & 128u);
            };
            for (;;) {   // uniform
                u64 cmr = 0;
                u32 w0 = 0, w1 = 0, f0 = 0, f1 = 0, f2 = 0, f3 = 0, pt = 0;
                static_for<PPL>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if (cmr == 0 && cm[r] != 0) {   // uniform
                        cmr = cm[r], cm[r] = 0;
                        w0 = pw[r][0], w1 = pw[r][1];
                        f0 = acc[r][0], f1 = acc[r][1], f2 = acc[r][2], f3 = acc[r][3];
                        pt = CR::point(pb, r, lane);
                    }
                });
                if (cmr == 0) break;
                asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(w0), "+v"(w1), "+v"(pt));   // (keeps the copy: no second instance of what follows)
                // the integer sums themselves, field by field (no borrow: every field started from its bias)
                const u32 q0 = f0 - bs0, q1 = f1 - bs1, q2 = f2 - bs2, q3 = f3 - bs3;
                u32 m8 = ((cmr >> lane) & 1ull) ? below(f0, f1, f2, f3) : 0u;
                for (u64 bm; (bm = __builtin_amdgcn_ballot_w64(m8 != 0u)) != 0;) {   // one pair per lane and trip (a point rarely passes for two queries)
                    if (pcnt + __popcll(bm) > NF_POOL) {
#ifdef IVFADC_DEBUG
                        if (dbg_flags & 32) nsurv += 1u;              // pool overflows
                        const u64 tm0 = (dbg_flags & 128) ? (u64)__builtin_readcyclecounter() : 0ull;
#endif
                        nf_make_room(pool, pcnt, cb_lab, smem, nvalid, sel, usel, K, wv, lane, nsurv, false, (dbg_flags & 1016) == 0);
#ifdef IVFADC_DEBUG
                        if (dbg_flags & 128) nsurv += (u32)(((u64)__builtin_readcyclecounter() - tm0) >> 6);   // cycles / 64 in overflow handling
#endif
                        exchange();
                        tg.set(smem, nvalid, sel, wv, lane);
                        m8 &= below(q0 + tg.b0, q1 + tg.b1, q2 + tg.b2, q3 + tg.b3);   // the budgets have moved: what still passes (no carry: Q <= 32760, bias <= 0x8000)
                        continue;
                    }
                    const u32 s = m8 ? (u32)__builtin_ctz(m8) : 0u;
                    const u32 h = s >> 1;
                    const u32 qw = h == 0u ? q0 : (h == 1u ? q1 : (h == 2u ? q2 : q3));
                    const u32 Q = (qw >> ((s & 1u) * 16u)) & 0xffffu;
                    if (m8 != 0u) {
                        u32 *ent = pool + (size_t)(pcnt + __popcll(bm & ((1ull << lane) - 1ull))) * NF_ES;
                        *(uint4 *)ent = make_uint4(w0, w1, pt, (Q << 8) | s);
                    }
                    pcnt += __popcll(bm);
#ifdef IVFADC_DEBUG
                    if (dbg_flags & 16) nsurv += (u32)__popcll(bm);   // parked pairs
                    if (dbg_flags & 64) nsurv += 1u;                  // trips of the parking loop
#endif
                    m8 &= m8 - 1u;
                }
            }
        }
#ifdef IVFADC_DEBUG
        if ((dbg_flags & 512) && any) nsurv += (u32)(((u64)__builtin_readcyclecounter() - tc0) >> 6);   // cycles / 64 in the whole candidate path
#endif
        cr = cr1;
        cr1 = nx;
    }
    // what is still viable under the final bounds gets its exact sum (most of what was parked never does)
    exchange();
#ifdef IVFADC_DEBUG
    const u64 tf0 = (dbg_flags & 256) ? (u64)__builtin_readcyclecounter() : 0ull;
#endif
    if (pcnt > 0) nf_make_room(pool, pcnt, cb_lab, smem, nvalid, sel, usel, K, wv, lane, nsurv, true, (dbg_flags & 1016) == 0);
#ifdef IVFADC_DEBUG
    if (dbg_flags & 256) nsurv += (u32)(((u64)__builtin_readcyclecounter() - tf0) >> 6);   // cycles / 64 in the final flush
#endif
    exchange();
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------------
struct NfView {
    const float *n2;       // [8][256] ||codeword||^2 by codeword index (rounded from double)
    const float *cb_lab;   // [8][256][16] f32 codewords by label
    const float *maxn2;    // [8] >= max over the codewords of ||codeword||^2
    u32 *xq;               // [8] work-queue heads, 64 B apart, zero at launch: one per XCD (nranges = 8) or one for all (nranges = 1)
    int nranges;
};

template <int PPL>
__global__ __launch_bounds__(256, 2) void nf_scan_kernel(const ScanArgs a, const NfView nf)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int QG = NF_QG;
    const IndexView &ix = a.ix;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = a.K;
    u64 *sthr = (u64 *)(smem + NfLds::STHR);
    u64 *shard = (u64 *)(smem + NfLds::HARD);
    int *scnt = (int *)(smem + NfLds::SCNT);
    u32 *swi = (u32 *)(smem + NfLds::SWI);
    float *qc = (float *)(smem + NfLds::QC);
    u32 *sqi = (u32 *)(smem + NfLds::QI);        // [0..8) probe index, [8..16) query, [16..24) visit-order base
    const u32 total = a.wi_off[ix.kc];
    u32 nsurv = 0;
    // Work items are ordered by list, so the groups of one list are neighbours in the queue.  The item range is cut into one contiguous
    // part per XCD and a workgroup pulls from the part of the XCD it runs on (HW_REG_XCC_ID): the ~2.5 groups that stream the same list
    // then run side by side under ONE L2 and the list comes from HBM once.  Placement is a matter of speed only: a workgroup whose part
    // is exhausted moves on to the next one, and every wave leaves when all parts are.
    const int nr = nf.nranges;
    int cur = nr > 1 ? (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u) % nr : 0;   // HW_REG_XCC_ID, bits [3:0]
    int tried = 0;

    for (;;) {
        __syncthreads();
        if (tid == 0) {
            u32 got = 0xFFFFFFFFu;
            int c = cur, t = tried;
            while (t < nr) {
                const u32 r0 = (u32)(((u64)total * (u32)c) / (u32)nr), r1 = (u32)(((u64)total * (u32)(c + 1)) / (u32)nr);
                const u32 k = atomicAdd(nf.xq + c * 16, 1u);
                if (k < r1 - r0) { got = r0 + k; break; }
                c = c + 1 == nr ? 0 : c + 1;
                ++t;
            }
            swi[0] = got;
            swi[2] = (u32)c;
            swi[3] = (u32)t;
        }
        __syncthreads();
        const u32 wi = __builtin_amdgcn_readfirstlane(swi[0]);
        cur = (int)__builtin_amdgcn_readfirstlane(swi[2]);
        tried = (int)__builtin_amdgcn_readfirstlane(swi[3]);
        if (wi == 0xFFFFFFFFu) break;   // uniform: every wave of every workgroup reaches this

        int lo = 0, hi = ix.kc;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.wi_off[mid] <= wi) lo = mid; else hi = mid;
        }
        const int l = lo;
        const u32 cnt = a.list_cnt[l];
        const u32 ng = (cnt + QG - 1) / QG;
        const u32 local = wi - a.wi_off[l];
        const u32 chunk = local / ng, grp = local - chunk * ng;
        const u32 len = ix.list_len[l];
        const u32 p0 = chunk * a.CH;
        if (p0 >= len) continue;   // uniform
        const u32 p1 = min(len, p0 + a.CH);
        const int nvalid = min((int)QG, (int)(cnt - grp * QG));

        // the queries of the group: thread s < 8 fetches slot s (slots past nvalid repeat slot 0 and can never be candidates)
        if (tid < QG) {
            const int ss = tid < nvalid ? tid : 0;
            const u32 pi = a.bucket_items[a.bucket_off[l] + grp * QG + ss];
            const u32 qq = pi / (u32)a.w;
            sqi[tid] = pi;
            sqi[8 + tid] = qq;
            sqi[16 + tid] = a.probe_base[pi];
            qc[16 + tid] = a.probe_dc[pi];
            const u64 t0 = __hip_atomic_load(&a.qthr[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            shard[tid] = t0;
            sthr[tid] = t0;
        }
        if (tid < 64) {
            ((u32 *)(smem + NfLds::SMIN))[tid] = 0xFFFFFFFFu;
            ((u32 *)(smem + NfLds::SMAX))[tid] = 0u;
        }
        __syncthreads();
        // exact pruning of whole work items, as in scan_kernel: no sum of this list lies below its coarse distance
        if (a.prune) {
            bool all = true;
#pragma unroll
            for (int s = 0; s < QG; ++s) all = all && (s >= nvalid || __float_as_uint(qc[16 + s]) > (u32)(shard[s] >> 32));
            if (all) {   // uniform: every thread read the same LDS words
                if (tid < nvalid) {
                    const u32 pi = sqi[tid];
                    a.part_cnt[(size_t)pi * a.maxch + chunk] = 0u;
                    atomicAdd(a.scanned_points + (size_t)(pi & 63u) * 8 + 1, (u64)(p1 - p0));
                }
                continue;
            }
        }
        const uint8_t *cbase = ix.codes + ix.list_codeoff[l];
        CodeRegs<8, PPL> cr;
        scan_prefetch(cr, cbase, p0, p1, wv, lane);     // in flight while the tables are built

        // (1) residuals r_s = q_s - c (coarsequantizers.jl:40-45), f32, as [s][t4][ii][4]
        {
            float *res = (float *)(smem + NfLds::RES);
            const float *crow = ix.centroids + (size_t)l * 128;
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                const int e = rep * 256 + tid;            // s * 128 + i
                const int s = e >> 7, i = e & 127;
                const u32 qs = sqi[8 + s];
                const int ii = i >> 4, t4 = (i >> 2) & 3;
                res[(((s * 4 + t4) * 8 + ii) << 2) | (i & 3)] = a.queries[(size_t)qs * 128 + i] - crow[i];
            }
        }
        __syncthreads();
        // (2) ||r_s,ii||^2 (1 - 2^-17), one (ii, s) pair per thread of wave 0
        if (tid < 64) {
            const int ii = tid >> 3, s = tid & 7;
            const float4 *rs = (const float4 *)(smem + NfLds::RES);
            float r2 = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 r4 = rs[(s * 4 + g) * 8 + ii];
                r2 = __builtin_fmaf(r4.x, r4.x, r2); r2 = __builtin_fmaf(r4.y, r4.y, r2);
                r2 = __builtin_fmaf(r4.z, r4.z, r2); r2 = __builtin_fmaf(r4.w, r4.w, r2);
            }
            ((float *)(smem + NfLds::RN2))[tid] = r2 * 0.99999237060546875f;
        }
        __syncthreads();
        // (3) lower-bound entries of the thread's codeword; their minima and maxima over the codes
        // (the thread and lane numbers pass through an opaque move inside the item loop: the 32 lane-constant codeword addresses they
        // feed would otherwise be hoisted to kernel entry as 64 registers and live -- spilled -- across the whole persistent loop)
        int tidb = tid, laneb = lane;
        asm volatile("" : "+v"(tidb), "+v"(laneb));
        NfBuild bld;
#ifdef IVFADC_DEBUG
        if (ix.dbg_flags & 4) {       // knock-out: no table build (the scan runs on whatever the LDS holds)
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int s = 0; s < 8; ++s) bld.e[k][s] = (float)(k + s + tidb);
        } else
#endif
        nf_entries(ix, nf.n2, smem, tidb, laneb, bld);
        nf_reduce_publish<true>(bld, smem, laneb);
        nf_reduce_publish<false>(bld, smem, laneb);
        __syncthreads();
        // (4) per (ii, s): base; per query: scale, sum of bases
        if (tid < 64) {
            const u32 mn = ((const u32 *)(smem + NfLds::SMIN))[tid], mx = ((const u32 *)(smem + NfLds::SMAX))[tid];
            const float fmn = ordered_to_float(mn), fmx = ordered_to_float(mx);
            ((float *)(smem + NfLds::BASE))[tid] = fmn;
            // lanes ii * 8 + s: sum of the bases and the largest range over the sub-quantizers of query s (xor 8, 16, 32: a fixed order)
            float sb = fmn, rg = fmx - fmn;
            float nn = nf.maxn2[tid >> 3] + ((const float *)(smem + NfLds::RN2))[tid] * 1.001f;
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                sb = sb + __shfl_xor(sb, off);
                rg = fmaxf(rg, __shfl_xor(rg, off));
                nn = nn + __shfl_xor(nn, off);
            }
            if (tid < 8) {
                float R = rg;
                // a bound that stands at the start of the item leaves a budget: finer steps, and entries beyond it saturate
                const u32 hs = (u32)(shard[tid] >> 32);
                if (hs < 0x7F800000u) {
                    const float bud = (__uint_as_float(hs) * 1.0001f - qc[16 + tid] - sb) * 1.05f;
                    if (bud > 0.0f && bud < R) R = bud;
                }
                const bool good = R > 1e-30f && R < 3.0e38f && sb > -3.0e38f && sb < 3.0e38f;
                qc[tid] = good ? 4095.0f / R : 0.0f;
                qc[8 + tid] = good ? sb : -3.0e38f;   // inv = 0: every field is 0; a hugely negative sum of bases lets every point through
                qc[24 + tid] = nn;
            }
        }
        __syncthreads();
        // (5) the two copies of the 12-bit table
        nf_quantize_store(ix, smem, tidb, laneb, bld);
        WSel<true> sel[QG];
#pragma unroll
        for (int s = 0; s < QG; ++s) sel[s].init(readfirstlane64(shard[s]), nullptr, 64, K);
        __syncthreads();

        __builtin_amdgcn_s_setprio(3);
#ifdef IVFADC_DEBUG
        if (!(ix.dbg_flags & 2))      // knock-out: table build only
#endif
        nf_scan_range<PPL>(cbase, p0, p1, nvalid, sel, K, wv, lane, cr, smem, nf.cb_lab, nsurv, ix.dbg_flags);
        __builtin_amdgcn_s_setprio(0);

        // ---- per-wave flush, then wave v merges slots v and v + 4 of the four waves and publishes them (as scan_kernel)
        int mycnt[QG];
#pragma unroll
        for (int s = 0; s < QG; ++s) mycnt[s] = sel[s].finish(K, lane);
        __syncthreads();   // the exchange area aliases the table: every wave must be done scanning
        u64 *xch = (u64 *)smem;
#pragma unroll
        for (int s = 0; s < QG; ++s) {
            sel[s].store(xch + ((size_t)wv * QG + s) * 64, mycnt[s], lane);
            if (lane == 0) scnt[wv * QG + s] = mycnt[s];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < QG; ++s) {
            if ((s & 3) == wv && s < nvalid) {
                const u64 hard = readfirstlane64(shard[s]);
                merge_waves(sel[s], xch + (size_t)s * 64, (size_t)QG * 64, scnt + s, QG, K, hard, wv, lane);
                const int fc = sel[s].finish(K, lane);
                const size_t slot = (size_t)sqi[s] * a.maxch + chunk;
                u64 *dst = a.part_keys + slot * K;
                sel[s].for_each(fc, lane, [&](int i, u64 key) { dst[i] = key; });
                if (lane == 0) {
                    a.part_cnt[slot] = (u32)fc;
                    // the workgroup's shared bound (K upper bounds or K exact sums stand behind it) serves the query's other work items
                    u64 pubk = sthr[s];
                    if (fc == K && sel[s].thr() < pubk) pubk = sel[s].thr();
                    if (pubk < hard) atomicMin(&a.qthr[sqi[8 + s]], pubk);
                }
            }
        }
    }
    if (lane == 0 && nsurv) atomicAdd(a.scanned_points + (size_t)(blockIdx.x & 63) * 8 + 2, (u64)nsurv);
}
