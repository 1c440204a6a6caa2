// Lane-to-lane moves inside a wavefront without the LDS crossbar.
#pragma once

#include <hip/hip_runtime.h>

namespace lbl {

// Lane-to-lane moves by data-parallel primitives (DPP): the operand of a VALU instruction is
// taken from another lane of the row of 16 (row_shr:n), from the last lane of the previous row
// (row_bcast:15, rows 1 and 3) or from lane 31 (row_bcast:31, rows 2 and 3) -- a few cycles,
// where a shuffle through the LDS crossbar (ds_bpermute) is a round trip of ~100.  (The pedestal
// chain is one wavefront whose every step waits for the previous one, and the far-field kernels
// reduce 21 sums per wavefront: for both the shuffles were the bottleneck.)  Lanes without a
// source keep `fill`.
template <int CONTROL, int ROWS>
__device__ __forceinline__ double dpp_from(double fill, double value)
{
    const long long f = __double_as_longlong(fill), v = __double_as_longlong(value);
    const int lo = __builtin_amdgcn_update_dpp((int)f, (int)v, CONTROL, ROWS, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(v >> 32), CONTROL, ROWS, 0xf,
                                               false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118;
constexpr int kRowBcast15 = 0x142, kRowBcast31 = 0x143;
constexpr int kWaveShr1 = 0x138;        // the whole wavefront shifted by one lane

// Inclusive prefix sums over the 64 lanes (Kogge-Stone inside the rows, then the row totals).
__device__ __forceinline__ double wave_prefix_sum(double x)
{
    x += dpp_from<kRowShr1, 0xf>(0., x);
    x += dpp_from<kRowShr2, 0xf>(0., x);
    x += dpp_from<kRowShr4, 0xf>(0., x);
    x += dpp_from<kRowShr8, 0xf>(0., x);
    x += dpp_from<kRowBcast15, 0xa>(0., x);
    x += dpp_from<kRowBcast31, 0xc>(0., x);
    return x;
}

// The sums over the 64 lanes of N values per lane at once.  N scans would move every value six
// times; here each exchange step hands half of a lane's values to its partner (lane ^ 1, ^ 2, ...)
// and keeps the sums of the other half, so the lists shrink 21 -> 11 -> 6 -> 3 -> 2 -> 1 -> 1 and
// ~24 additions do the work of 126.  Afterwards lane `l` holds in v[0] the complete sum of the
// value `index`; `valid` says whether that is one of the N (the lists are padded to even length).
// The partner and the order of every addition are fixed: the result is reproducible.
namespace detail {

template <int CONTROL, int BANKS>
__device__ __forceinline__ double dpp_into(double old, double value)
{
    const long long f = __double_as_longlong(old), v = __double_as_longlong(value);
    const int lo = __builtin_amdgcn_update_dpp((int)f, (int)v, CONTROL, 0xf, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(v >> 32), CONTROL, 0xf,
                                               BANKS, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// The value of lane ^ DISTANCE.
template <int DISTANCE>
__device__ __forceinline__ double from_partner(double value)
{
    if constexpr (DISTANCE == 1) return dpp_into<0xB1, 0xf>(value, value);     // quad_perm 1,0,3,2
    else if constexpr (DISTANCE == 2) return dpp_into<0x4E, 0xf>(value, value);    // 2,3,0,1
    else if constexpr (DISTANCE == 4)
    {
        // row_shr:4 into the banks whose lanes have bit 2 set, row_shl:4 into the others
        return dpp_into<0x104, 0x5>(dpp_into<0x114, 0xa>(value, value), value);
    }
    else if constexpr (DISTANCE == 8) return dpp_into<0x128, 0xf>(value, value);   // row_ror:8
    else return __shfl_xor(value, DISTANCE, 64);
}

template <int L, int DISTANCE, int N>
__device__ __forceinline__ void exchange_step(double (&v)[N], int lane, int & index, bool & valid)
{
    if constexpr (DISTANCE < 64)
    {
        constexpr int h = (L + 1)/2;
        const bool upper = (lane & DISTANCE) != 0;
#pragma unroll
        for (int i = 0; i < h; ++i)
        {
            const double low = v[i];
            const double high = (i + h < L) ? v[i + h] : 0.;
            const double keep = upper ? high : low;
            const double send = upper ? low : high;
            v[i] = keep + from_partner<DISTANCE>(send);
        }
        exchange_step<h, DISTANCE*2, N>(v, lane, index, valid);
        // (on the way back: the index of v[0] in this step's list)
        index += upper ? h : 0;
        valid = valid && index < L;
    }
}

}  // namespace detail

template <int N>
__device__ __forceinline__ void wave_sums(double (&v)[N], int & index, bool & valid)
{
    index = 0;
    valid = true;
    detail::exchange_step<N, 1, N>(v, threadIdx.x & 63, index, valid);
}

// First index in [lo, hi] whose wavenumber is > x (ABOVE) or >= x, found by a whole wavefront: 64
// probes per step, so ~3400 candidates take two loads' latency where a binary search takes
// twelve (the search is the serial head of its workgroup: farfield.h, xsec.h).
template <bool ABOVE>
__device__ inline int wave_search(const double * __restrict__ nu, int lo, int hi, double x)
{
    const int lane = threadIdx.x & 63;
    while (hi - lo > 64)
    {
        const int stride = (hi - lo + 63) >> 6;
        const int at = lo + lane*stride;
        const bool below = at < hi && (ABOVE ? nu[at] <= x : nu[at] < x);
        const int count = __builtin_popcountll(__ballot(below));
        if (count == 0)
        {
            return lo;
        }
        const int base = lo + (count - 1)*stride;
        hi = min(base + stride, hi);
        lo = base + 1;
    }
    const int at = lo + lane;
    const bool below = at < hi && (ABOVE ? nu[at] <= x : nu[at] < x);
    return lo + __builtin_popcountll(__ballot(below));
}

}  // namespace lbl
