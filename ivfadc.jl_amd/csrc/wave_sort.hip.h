// wave_sort.hip.h -- ascending sort of one 64-bit key per lane across a 64-lane wavefront.
//
// Bitonic network in its "flip" form: merge step k first pairs lane i with lane i ^ (k-1), then with i ^ j for
// j = k/4 ... 1; every compare keeps the minimum in the lower lane, so no direction bit is needed.  All 21 lane
// exchanges are VALU data movement -- DPP quad_perm / row_half_mirror / row_mirror / row_ror / row_shl+row_shr inside
// a row of 16 lanes, v_permlane16_swap / v_permlane32_swap (new on gfx950) across rows -- instead of ds_bpermute
// round trips through the LDS crossbar, which cost ~200 cycles per stage for a lone wave.
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned long long u64;
typedef unsigned int u32;

namespace ivf {

template <int CTRL> static __device__ __forceinline__ u32 dpp_mov(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// value of lane (i ^ X) for X in {1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 63}
template <int X> static __device__ __forceinline__ u32 lane_xor(u32 v, int lane)
{
    if constexpr (X == 1) return dpp_mov<0xB1>(v);            // quad_perm [1,0,3,2]
    else if constexpr (X == 2) return dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    else if constexpr (X == 3) return dpp_mov<0x1B>(v);       // quad_perm [3,2,1,0]
    else if constexpr (X == 7) return dpp_mov<0x141>(v);      // row_half_mirror
    else if constexpr (X == 15) return dpp_mov<0x140>(v);     // row_mirror
    else if constexpr (X == 8) return dpp_mov<0x128>(v);      // row_ror:8
    else if constexpr (X == 4) {
        const u32 up = dpp_mov<0x104>(v);                     // row_shl:4  lane i <- lane i+4
        const u32 dn = dpp_mov<0x114>(v);                     // row_shr:4  lane i <- lane i-4
        return (lane & 4) ? dn : up;
    } else if constexpr (X == 16) {
        // swap odd rows of the first copy with even rows of the second: a = [r0 r0 r2 r2], b = [r1 r1 r3 r3]
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else if constexpr (X == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // a = [lo lo], b = [hi hi]
        return (lane & 32) ? r[0] : r[1];
    } else if constexpr (X == 31) return lane_xor<15>(lane_xor<16>(v, lane), lane);
    else if constexpr (X == 63) return lane_xor<15>(lane_xor<16>(lane_xor<32>(v, lane), lane), lane);
    else { static_assert(X == 1, "unsupported lane xor"); return v; }
}

// one compare-exchange stage: partner = lane ^ X, the lower lane of each pair keeps the minimum
template <int X> static __device__ __forceinline__ u64 sort_stage(u64 v, int lane)
{
    const u32 plo = lane_xor<X>((u32)v, lane);
    const u32 phi = lane_xor<X>((u32)(v >> 32), lane);
    const u64 pv = ((u64)phi << 32) | plo;
    constexpr int TOP = (X & (X + 1)) == 0 ? (X + 1) >> 1 : X;   // highest bit in which the two lanes differ
    const bool keep_min = (lane & TOP) == 0;
    return (keep_min == (pv < v)) ? pv : v;
}

// NOT inlined: it is reached from every selector push, and inlining it there bloats the scan loops.
static __device__ __attribute__((noinline)) u64 wave_sort64(u64 v, int lane)
{
    v = sort_stage<1>(v, lane);
    v = sort_stage<3>(v, lane);  v = sort_stage<1>(v, lane);
    v = sort_stage<7>(v, lane);  v = sort_stage<2>(v, lane);  v = sort_stage<1>(v, lane);
    v = sort_stage<15>(v, lane); v = sort_stage<4>(v, lane);  v = sort_stage<2>(v, lane);  v = sort_stage<1>(v, lane);
    v = sort_stage<31>(v, lane); v = sort_stage<8>(v, lane);  v = sort_stage<4>(v, lane);  v = sort_stage<2>(v, lane);
    v = sort_stage<1>(v, lane);
    v = sort_stage<63>(v, lane); v = sort_stage<16>(v, lane); v = sort_stage<8>(v, lane);  v = sort_stage<4>(v, lane);
    v = sort_stage<2>(v, lane);  v = sort_stage<1>(v, lane);
    return v;
}

}  // namespace ivf
