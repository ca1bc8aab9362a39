// twolevel.hip.h -- certified two-level coarse search (what coarsequantizers.jl:58-92 uses an HNSW graph for).
//
// The kc centroids are grouped once (k-means over the centroids: kc / 64 groups), and every group keeps its centre g and a radius
// r_g >= max ||c - g|| over its members.  For a query q and any member c of group g the triangle inequality gives
//     ||q - c|| >= ||q - g|| - r_g,
// so lb_g = max(0, ||q - g|| - r_g)^2 bounds every member's squared distance from below.  A query visits the groups in ascending lb_g,
// computes the members' distances EXACTLY (the reference's sum, coarsequantizers.jl:34: sequential, one rounding per operation, no FMA --
// the same bits the exhaustive kernels produce) and keeps the w smallest (distance, cluster id) keys; it stops at the first group whose
// bound lies above the w-th best distance found so far.  Nothing is approximated: a skipped group cannot hold a centroid that belongs
// to the top-w, ties included (a bound EQUAL to the w-th best distance does not skip: a member at exactly that distance with a lower
// cluster id wins the tie, as in the reference's stable sortperm).  The result is the oracle's, bit for bit, whatever the data; what
// depends on the data is how many groups are skipped -- nearly all of them for quantizers trained on clustered data, none for
// N(0,1) centroids in high dimension (distance concentration: every bound lies far below the w-th best distance).
//
// Rounding.  With u = 2^-24: the kernels' float distance of a centroid is >= (true distance)(1 - (d + 2) u); the computed ||q - g||^2
// likewise; sqrtf, the subtraction and the square add a few u.  The bound is therefore deflated by eps = (d + 16) 2^-23 at both ends
// (more than twice what the analysis needs); the radii are computed in double and rounded up.
#pragma once

namespace ivf {

struct TwoLevelView {
    const float *gdist;     // [nq][G] exact squared distances query -> group centre (coarse_dist_kernel on the G centres)
    const u32 *g_off;       // [G + 1] first slot of every group (members side by side, no padding)
    const float *g_rad;     // [G] >= max ||member - centre||, rounded up
    const float *cent_g;    // grouped centroids: group g's block starts at float g_off[g] * d, laid out [d / 4][slots][4]
    const u32 *slot_id;     // [kc] cluster id of a slot
    int G, d;
    float eps;
};

static __device__ __forceinline__ float tl_lower_bound(float gd, float r, float eps)
{
    const float s = sqrtf(gd) * (1.0f - eps) - r;      // (a NaN distance -- non-finite query -- gives 0: everything is visited)
    return s > 0.0f ? (s * s) * (1.0f - eps) : 0.0f;
}

// exact distances of the members of group g to the wave's query (in LDS at qv), into the wave's top-w selector
static __device__ __forceinline__ void tl_scan_group(const TwoLevelView &tv, u32 g, const float *qv, WSel<true> &sel, int w, int lane,
                                                     u32 &visited)
{
    const u32 s0 = tv.g_off[g], ns = tv.g_off[g + 1] - s0;
    const float *blk = tv.cent_g + (size_t)s0 * tv.d;
    const int d4 = tv.d >> 2;
    for (u32 sb = 0; sb < ns; sb += 64) {
        const bool valid = sb + (u32)lane < ns;           // (lanes past the group's end repeat its last member and are not counted)
        const u32 s = valid ? sb + (u32)lane : ns - 1;
        const u32 id = tv.slot_id[s0 + s];
        const float4 *col = (const float4 *)blk + s;      // element i4 at col[i4 * ns]
        float acc = 0.0f;
        for (int i = 0; i < d4; i += 4) {
            float4 c4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c4[u] = col[(size_t)(i + u < d4 ? i + u : i) * ns];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i + u < d4) {
                    const float4 q4 = *(const float4 *)(qv + 4 * (i + u));     // one address for the wave: an LDS broadcast
                    float t = c4[u].x - q4.x; acc = acc + t * t;
                    t = c4[u].y - q4.y; acc = acc + t * t;
                    t = c4[u].z - q4.z; acc = acc + t * t;
                    t = c4[u].w - q4.w; acc = acc + t * t;
                }
            }
        }
        const u64 key = make_key(acc, id);
        sel.push(valid && key < sel.thr(), key, w, lane);
        visited += (u32)__popcll(__ballot(valid));
    }
}

// One wave per query (four queries per workgroup, no workgroup barrier).  Writes what topw_select_kernel writes: the probe arrays, the
// probe histogram of the list-major plan, the B_alg counter -- and, in slot 3 of the sharded counters, the number of exact centroid
// distances it computed.  w <= 64, d % 4 == 0.  Dynamic LDS: 4 x d floats (queries) + 4 x 64 keys.
__global__ __launch_bounds__(256) void twolevel_topw_kernel(const float *__restrict__ queries, const TwoLevelView tv, int nq, int w,
                                                            const u32 *__restrict__ list_len, int *__restrict__ probe_list,
                                                            float *__restrict__ probe_dc, u32 *__restrict__ probe_base,
                                                            u32 *__restrict__ list_cnt, u64 *__restrict__ scanned_points, int nparts, int part)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wv;
    if (q >= nq) return;
    const int d = tv.d, G = tv.G;
    float *qv = (float *)smem_raw + (size_t)wv * d;
    u64 *buf = (u64 *)(smem_raw + (size_t)4 * d * 4) + (size_t)wv * 64;
    for (int i = lane; i < d; i += 64) qv[i] = queries[(size_t)q * d + i];
    wave_sync();
    // the 64 groups with the smallest bounds, ascending (key = bound bits << 32 | group)
    const float *grow = tv.gdist + (size_t)q * G;
    WSel<true> gs;
    gs.init(KEY_MAX, nullptr, 64, 64);
    for (int g0 = 0; g0 < G; g0 += 64) {
        const int g = g0 + lane;
        const bool pred = g < G;
        const u64 key = pred ? make_key(tl_lower_bound(grow[g], tv.g_rad[g], tv.eps), (u32)g) : KEY_MAX;
        gs.push(pred && key < gs.thr(), key, 64, lane);
    }
    const int ng = gs.finish(64, lane);
    WSel<true> sel;
    sel.init(KEY_MAX, nullptr, 64, w);
    u32 visited = 0;
    int j = 0;
    for (; j < ng; ++j) {
        const u64 gk = readlane64(gs.top, j);
        // no member of this group (nor of any later one: bounds ascend) can enter: every member's key is >= bound bits << 32
        if ((gk & 0xFFFFFFFF00000000ull) > sel.thr()) break;
        tl_scan_group(tv, (u32)gk, qv, sel, w, lane, visited);
    }
    if (j == ng && ng == 64 && G > 64) {
        // sixty-four groups went by and the bound has not closed (unstructured centroids): every other group, in index order, unless
        // its own bound excludes it -- still exact, and no slower than looking at everything
        const u64 last = readlane64(gs.top, 63);
        for (int g0 = 0; g0 < G; g0 += 64) {
            const int g = g0 + lane;
            const u64 key = g < G ? make_key(tl_lower_bound(grow[g], tv.g_rad[g], tv.eps), (u32)g) : KEY_MAX;
            u64 todo = __ballot(g < G && key > last);
            while (todo) {
                const int src = __builtin_ctzll(todo);
                todo &= todo - 1;
                const u64 gk = readlane64(key, src);
                if ((gk & 0xFFFFFFFF00000000ull) > sel.thr()) continue;
                tl_scan_group(tv, (u32)gk, qv, sel, w, lane, visited);
            }
        }
    }
    const int cnt = sel.finish(w, lane);
    sel.store(buf, cnt, lane);
    wave_sync();
    // probe arrays, visit-order bases, probe histogram: as topw_select_kernel leaves them
    u32 len = 0, mine_total = 0;
    int l = 0;
    float dd = 0.0f;
    if (lane < cnt) {
        const u64 key = buf[lane];
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
    mine_total = minc;
    if (lane < cnt) {
        const size_t o = (size_t)q * w + lane;
        probe_list[o] = l;
        probe_dc[o] = dd;
        probe_base[o] = incl - len;
        if (list_cnt && mine) atomicAdd(&list_cnt[l], 1u);
    }
    if (lane == 0) {
        atomicAdd(scanned_points + (size_t)(q & 63) * 8, (u64)mine_total);
        atomicAdd(scanned_points + (size_t)(q & 63) * 8 + 3, (u64)visited);
    }
}

}  // namespace ivf
