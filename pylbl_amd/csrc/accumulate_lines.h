// The hot kernel's line loops: what a wavefront does with each of the cut-point ranges of its
// tile (accumulate.h states the ranges) -- fast_ranges (far wing, eight lines per reciprocal),
// clipped_ranges / general_line (windows that end inside the tile), core_lines / core_line (tiles
// that may hold a line's core: voigt.c:74-189 row by row) and inner_ranges / inner_batch (the
// points nearer than xlim1, packed by class through LDS).  Included by accumulate.h after its
// argument structs.
#pragma once

namespace lbl {

// Far-wing loop over two index ranges [a0,a1) and [b0,b1) whose lines all cover the whole
// tile with the tile in their Lorentz wing.  Eight (or four) lines share one reciprocal;
// only whole groups of four are taken here, the caller sends the 0-3 left-over lines of each
// range down the general path.  The records arrive by scalar loads whose latency is covered
// by the other resident wavefronts of the SIMD.
template <int P>
__device__ __forceinline__ void fast_ranges(const LineWing * __restrict__ wing,
                                            int a0, int a1, int b0, int b1,
                                            const double (&v)[P], double (&acc)[P])
{
    const int qa = (a1 - a0) >> 2;
    const int quads = qa + ((b1 - b0) >> 2);
    int q = 0;
#if LBL_WING_GROUP == 8
    // Two groups of four per step: eight lines share one reciprocal.
    for (; q + 2 <= quads; q += 2)
    {
        const int ja = q < qa ? a0 + 4*q : b0 + 4*(q - qa);
        const int jb = (q + 1) < qa ? a0 + 4*(q + 1) : b0 + 4*(q + 1 - qa);
        WingTerm l[8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const LineWing wa = wing[ja + i], wb = wing[jb + i];
            l[i] = WingTerm{wa.centre, wa.g2, wa.bl};
            l[4 + i] = WingTerm{wb.centre, wb.g2, wb.bl};
        }
#pragma unroll
        for (int p = 0; p < P; ++p)
        {
            acc[p] = lorentz_eight(v[p], l, acc[p]);
        }
    }
#endif
    for (; q < quads; ++q)
    {
        const int j = q < qa ? a0 + 4*q : b0 + 4*(q - qa);
        const LineWing l1 = wing[j], l2 = wing[j + 1], l3 = wing[j + 2], l4 = wing[j + 3];
#pragma unroll
        for (int p = 0; p < P; ++p)
        {
            acc[p] = lorentz_four(v[p], l1.centre, l1.g2, l1.bl, l2.centre, l2.g2, l2.bl,
                                  l3.centre, l3.g2, l3.bl, l4.centre, l4.g2, l4.bl, acc[p]);
        }
    }
}

// One line that may clip the tile or have its core in it, row by row (64 points each).
// Per row the decisions are wave-uniform: skip (outside the window), Lorentz (whole row in
// the far wing), or core.  In a core row every lane applies the reference's chain on
// xi = (v-nu')*repwid (voigt.c:76-84): region 0 and w4 region 1 (voigt.c:95-96) are
// evaluated inline.  Lanes closer to the centre than xlim1 add nothing here: the inner points of
// all the lines walked here are summed afterwards, in a pass of their own (inner_ranges), into
// the wavefront's LDS sums.
// Which of a tile's rows (row p = grid indices [i0 + 64 p, i0 + 64 p + 63], p < rows <= 8) meet
// the index range [a, b], and which lie wholly inside it: one bit per row.  The row-by-row
// decisions below are wave-uniform and were, written as comparisons per row, ~19 scalar
// instructions per row and line -- the scalar unit, one per CU, was the busiest unit of the kernel
// wherever the general path carries the work (the far-field option: 183 M scalar against 139 M
// vector instructions per launch).  As masks they cost ~10 scalar instructions per range and line
// and a bit test per row.
__device__ __forceinline__ unsigned row_bits(int lo, int hi, int rows)
{
    lo = max(lo, 0);
    hi = min(hi, rows - 1);
    return lo > hi ? 0u : ((2u << hi) - 1u) & ~((1u << lo) - 1u);
}

__device__ __forceinline__ unsigned rows_meeting(int a, int b, int i0, int rows)
{
    return b < a ? 0u : row_bits((a - i0) >> 6, (b - i0) >> 6, rows);
}

__device__ __forceinline__ unsigned rows_inside(int a, int b, int i0, int rows)
{
    return b < a ? 0u : row_bits((a - i0 + 63) >> 6, (b - 63 - i0) >> 6, rows);
}

template <int P>
__device__ __forceinline__ void general_line(const LineWing & l, const LineCore & c,
                                             int i0, int i1, int lane,
                                             const double (&v)[P], double (&acc)[P])
{
    const unsigned in_window = rows_meeting(l.first, l.last, i0, P);
    if (in_window == 0)
    {
        return;     // also skips empty windows
    }
    // Rows wholly inside the window and in w4 region 1 on every lane (line_prep.h), and rows
    // that may hold a point of the core.
    const unsigned all_region_one = rows_inside(c.mid_first, c.mid_last, i0, P) &
                                    ~rows_meeting(c.hole_first, c.hole_last, i0, P);
    const unsigned near_core = rows_meeting(c.core_first, c.core_last, i0, P);
    const double rsqrpi = 0.56418958354775628695;   // 1/sqrt(pi)
    const double yq = c.y*c.y;
    const double a0 = yq + 0.5;                      // voigt.c:91-93
    const double d0 = a0*a0;
    const double d2 = yq + yq - 1.;
    const double r1_scale = c.amp*rsqrpi*c.y;
#pragma unroll
    for (int p = 0; p < P; ++p)
    {
        if (!(in_window & (1u << p)))
        {
            continue;
        }
        const int r0 = i0 + p*64;
        const double d = v[p] - l.centre;
        if (all_region_one & (1u << p))
        {
            // voigt.c:95-96 with no selection left to make.
            const double xi = d*c.repwid;
            const double xq = xi*xi;
            acc[p] += r1_scale*(a0 + xq)*rcp_newton(__builtin_fma(xq, d2 + xq, d0));
            continue;
        }
        const int i = r0 + lane;
        const bool inside = (i >= l.first) && (i <= l.last);
        double value;
        if (!(near_core & (1u << p)))
        {
            // voigt.c:82 / :24 in wavenumber units: the whole row is in the far wing.
            value = l.bl*rcp_newton(__builtin_fma(d, d, l.g2));
        }
        else
        {
            const double xi = d*c.repwid;               // voigt.c:76
            const double abx = fabs(xi);
            const double xq = abx*abx;
            const bool far = abx >= c.xlim0;
            const bool mid = !far && abx >= c.xlim1;
            value = 0.;
            if (__any(far))
            {
                const double wing = l.bl*rcp_newton(__builtin_fma(d, d, l.g2));
                value = far ? wing : value;
            }
            if (__any(mid))
            {
                // voigt.c:95-96: buf = rsqrpi/(d0 + xq(d2 + xq)) * y * (a0 + xq)
                const double w4 = r1_scale*(a0 + xq)*
                                  rcp_newton(__builtin_fma(xq, d2 + xq, d0));
                value = mid ? w4 : value;
            }
        }
        acc[p] += inside ? value : 0.;
    }
}

// Up to five index ranges of lines in general position, walked as one list.
struct GeneralList
{
    int begin[5];
    int count[5];
};

__device__ __forceinline__ int general_index(const GeneralList & g, int k)
{
    if (k < g.count[0]) return g.begin[0] + k;
    k -= g.count[0];
    if (k < g.count[1]) return g.begin[1] + k;
    k -= g.count[1];
    if (k < g.count[2]) return g.begin[2] + k;
    k -= g.count[2];
    if (k < g.count[3]) return g.begin[3] + k;
    k -= g.count[3];
    return g.begin[4] + k;
}

// The inner points (|x| < xlim1: w4 regions 2-3, CPF12) of a batch of lines, packed twice.
// Row by row they would leave most of a wavefront idle in the most expensive branches of the
// profile: a line's inner points are a few dozen consecutive grid points (27 for a CO2 line at
// 1000 cm-1 and 0.001 cm-1 spacing, a few hundred at 5000 cm-1 and 0.0005 cm-1), and within them
// the three classes of voigt_profile.h -- costing roughly 35, 120 and 190 instruction slots per
// point -- each hold a part of the lanes while the wavefront pays for all three.
//   step 1, lane = line:  fetch the line's scalars, clip its inner index range to tile and
//                         window, inclusive scan of the segment lengths (the segments of all
//                         queued lines laid end to end);
//   step 2, lane = point: for every 64 positions of that sequence, find the line and grid point
//                         of each lane, form x, classify with the reference's comparisons, and
//                         append (line, point) to the class's list (ballot + prefix count);
//   step 3, lane = entry of ONE class list, whenever a list holds 64 entries (and at the end):
//                         evaluate that class's formula, and add to `slab` (LDS sums of the
//                         wavefront, index = point - tile_first) one line at a time: two lines of
//                         a list may cover the same grid point, and the order of additions is
//                         fixed -- results do not depend on scheduling.
constexpr int kInnerQueue = 32;         // lines per batch (lane = line in step 1)
constexpr int kInnerList = 128;         // entries a class list can hold (64 left + 64 new)

struct InnerStage
{
    int first[kInnerQueue];             // first inner point in tile and window
    int begin[kInnerQueue], end[kInnerQueue];   // the line's segment in the packed sequence
    double centre[kInnerQueue], repwid[kInnerQueue], y[kInnerQueue], amp[kInnerQueue];
    double xlim1[kInnerQueue];
    unsigned short list[kInnerClasses][kInnerList];     // (queue slot << 9) | (point - tile_first)
};

template <int C>
__device__ __forceinline__ void inner_evaluate(InnerStage & stage, int count, int i0, int v0,
                                               double dv, int lane, double * slab)
{
    const bool active = lane < count;
    const unsigned record = stage.list[C][active ? lane : 0];
    const int q = record >> 9;
    const int offset = record & 511;
    // absorption.c:39: v[i] = v0 + i*dv (product rounded, then the sum); voigt.c:76-78.
    const double v = (double)v0 + (double)(i0 + offset)*dv;
    const double xi = (v - stage.centre[q])*stage.repwid[q];
    double value = 0.;
    if (active)
    {
        const double y = stage.y[q];
        const double k = C == 0 ? inner_class_a(xi, y)
                       : (C == 1 ? inner_class_b(xi, y) : inner_class_c(xi, y));
        value = stage.amp[q]*k;
    }
    unsigned long long left = __ballot(active);
    while (left != 0)
    {
        const int q0 = __builtin_amdgcn_readlane(q, __builtin_ctzll(left));
        const bool now = active && q == q0;
        if (now)
        {
            slab[offset] += value;
        }
        left &= ~__ballot(now);
    }
}

// `j` is this lane's line of the batch (lanes >= kInnerQueue and lanes beyond the batch: has =
// false).  Returns without touching LDS when no line of the batch has an inner point in the tile:
// the common case at tropospheric pressure, where y >= 8.425 switches the inner regions off for
// most lines, and for the lines of a batch whose cores lie in other tiles.
__device__ __forceinline__ bool inner_batch(const LineWing * __restrict__ wing,
                                            const LineCore * __restrict__ core,
                                            InnerStage & stage, int j, bool has, int i0, int i1,
                                            int v0, int n_per_v, double dv, int lane,
                                            double * slab, bool slab_in_use, int slab_rows)
{
    // Step 1.
    unsigned long long todo;
    int total;
    {
        // Two fields decide for most lines (xlim1 == 0: no inner regions at all; core range
        // elsewhere): the rest of the record is fetched only if some line of the batch passes.
        const double xlim1 = core[j].xlim1;
        const bool candidate = has && xlim1 > 0. && core[j].core_last >= i0 &&
                               core[j].core_first <= i1;
        if (__ballot(candidate) == 0)
        {
            return false;
        }
        const LineWing l = wing[j];
        const LineCore c = core[j];
        int first = 0, last = -1;
        if (candidate)
        {
            inner_index_range(l.centre, c.repwid, c.xlim1, v0, n_per_v, first, last);
            first = max(max(first, l.first), i0);
            last = min(min(last, l.last), i1);
        }
        const int length = last >= first ? last - first + 1 : 0;
        todo = __ballot(length > 0);
        if (todo == 0)
        {
            return false;
        }
        if (!slab_in_use)
        {
            // The wavefront's LDS sums are cleared by the first batch that has something to add
            // (most tiles at tropospheric pressure never get here).
            for (int p = 0; p < slab_rows; ++p) slab[p*64 + lane] = 0.;
        }
        int inclusive = length;
#pragma unroll
        for (int offset = 1; offset < kInnerQueue; offset <<= 1)
        {
            const int below = __shfl_up(inclusive, offset, 64);
            inclusive += (lane >= offset) ? below : 0;
        }
        if (lane < kInnerQueue)
        {
            stage.first[lane] = first;
            stage.begin[lane] = inclusive - length;
            stage.end[lane] = inclusive;
            stage.centre[lane] = l.centre;
            stage.repwid[lane] = c.repwid;
            stage.y[lane] = c.y;
            stage.amp[lane] = c.amp;
            stage.xlim1[lane] = c.xlim1;
        }
        total = __builtin_amdgcn_readlane(inclusive, kInnerQueue - 1);
    }

    int count[kInnerClasses] = {0, 0, 0};
    for (int base = 0; ; base += 64)
    {
        const bool drain = base >= total;
        if (!drain)
        {
            // Step 2.
            const int u = base + lane;
            double m_centre = 0., m_repwid = 1., m_y = 100., m_xlim1 = 0.;
            int m_point = i0, m_line = -1;
            unsigned long long scan = todo;
            while (scan != 0)
            {
                const int q = __builtin_ctzll(scan);
                const int begin = __builtin_amdgcn_readfirstlane(stage.begin[q]);
                if (begin >= base + 64)
                {
                    break;
                }
                const int end = __builtin_amdgcn_readfirstlane(stage.end[q]);
                const bool mine = u >= begin && u < end;
                m_centre = mine ? stage.centre[q] : m_centre;
                m_repwid = mine ? stage.repwid[q] : m_repwid;
                m_y = mine ? stage.y[q] : m_y;
                m_xlim1 = mine ? stage.xlim1[q] : m_xlim1;
                m_point = mine ? stage.first[q] + (u - begin) : m_point;
                m_line = mine ? q : m_line;
                if (end <= base + 64)
                {
                    todo &= ~(1ull << q);       // this line's segment ends here
                }
                scan &= scan - 1;
            }
            const double v = (double)v0 + (double)m_point*dv;
            const double abx = fabs((v - m_centre)*m_repwid);
            const bool take = m_line >= 0 && abx < m_xlim1;
            const int cls = inner_class(abx, inner_limits(m_y));
            const unsigned short record = (unsigned short)((m_line << 9) | (m_point - i0));
#pragma unroll
            for (int c = 0; c < kInnerClasses; ++c)
            {
                const bool member = take && cls == c;
                const unsigned long long mask = __ballot(member);
                if (member)
                {
                    const int rank = __builtin_amdgcn_mbcnt_hi(
                        (unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    stage.list[c][count[c] + rank] = record;
                }
                count[c] += __builtin_popcountll(mask);
            }
        }
        // Step 3.
#pragma unroll
        for (int c = 0; c < kInnerClasses; ++c)
        {
            if (count[c] >= 64 || (drain && count[c] > 0))
            {
                const int now = min(count[c], 64);
                if (c == 0) inner_evaluate<0>(stage, now, i0, v0, dv, lane, slab);
                if (c == 1) inner_evaluate<1>(stage, now, i0, v0, dv, lane, slab);
                if (c == 2) inner_evaluate<2>(stage, now, i0, v0, dv, lane, slab);
                count[c] -= now;
                // What is left (fewer than 64 entries) moves to the front of the list.
                const unsigned short rest = stage.list[c][64 + lane];
                if (lane < count[c])
                {
                    stage.list[c][lane] = rest;
                }
            }
        }
        if (drain)
        {
            break;
        }
    }
    return true;
}

// A line of the core range [c1, c2): its window covers the whole tile (the range lies inside
// [a1, a2)), so what is left to decide per row is `rows`, prepared by core_lines():
//   bit p       row p lies wholly in w4 region 1 (and inside the window): no selection at all;
//   bit 8 + p   row p may hold a point of the core: the reference's chain lane by lane;
//   neither     the whole row is in the far wing.
template <int P>
__device__ __forceinline__ void core_line(const LineWing & l, const LineCore & c, unsigned rows,
                                          const double (&v)[P], double (&acc)[P])
{
    const double rsqrpi = 0.56418958354775628695;   // 1/sqrt(pi)
    const double yq = c.y*c.y;
    const double a0 = yq + 0.5;                      // voigt.c:91-93
    const double d0 = a0*a0;
    const double d2 = yq + yq - 1.;
    const double r1_scale = c.amp*rsqrpi*c.y;
    // (rows that are neither: the far wing -- most rows of most lines, so that is the path the
    // wavefront falls through to: a taken branch costs it its instruction buffer, and the chain
    // "region 1? / near the core? / else" took two per far-wing row)
    const unsigned special = rows | (rows >> 8);
#pragma unroll
    for (int p = 0; p < P; ++p)
    {
        const double d = v[p] - l.centre;
        if (__builtin_expect(!(special & (1u << p)), 1))
        {
            // voigt.c:82 / :24 in wavenumber units: the whole row is in the far wing.
            acc[p] = __builtin_fma(l.bl, rcp_newton(__builtin_fma(d, d, l.g2)), acc[p]);
        }
        else if (rows & (1u << p))
        {
            // voigt.c:95-96 with no selection left to make.
            const double xi = d*c.repwid;
            const double xq = xi*xi;
            acc[p] += r1_scale*(a0 + xq)*rcp_newton(__builtin_fma(xq, d2 + xq, d0));
        }
        else
        {
            const double xi = d*c.repwid;               // voigt.c:76
            const double abx = fabs(xi);
            const double xq = abx*abx;
            const bool far = abx >= c.xlim0;
            const bool mid = !far && abx >= c.xlim1;
            double value = 0.;
            if (__any(far))
            {
                const double wing = l.bl*rcp_newton(__builtin_fma(d, d, l.g2));
                value = far ? wing : value;
            }
            if (__any(mid))
            {
                const double w4 = r1_scale*(a0 + xq)*
                                  rcp_newton(__builtin_fma(xq, d2 + xq, d0));
                value = mid ? w4 : value;
            }
            acc[p] += value;
        }
    }
}

// The core range, two lines at a time.  The row masks of 64 lines are formed at once, lane =
// line, by vector integer arithmetic (half an instruction per line where the scalar unit spent
// ~35), and handed to the walk through v_readlane.
template <int P>
__device__ __forceinline__ void core_lines(const LineWing * __restrict__ wing,
                                           const LineCore * __restrict__ core,
                                           int begin, int count, int i0, int lane,
                                           const double (&v)[P], double (&acc)[P])
{
    for (int base = 0; base < count; base += 64)
    {
        const int n = min(64, count - base);        // even: the caller keeps an odd line back
        const LineCore * __restrict__ mine = core + begin + base + min(lane, n - 1);
        const unsigned all_region_one = rows_inside(mine->mid_first, mine->mid_last, i0, P) &
                                        ~rows_meeting(mine->hole_first, mine->hole_last, i0, P);
        const unsigned near_core = rows_meeting(mine->core_first, mine->core_last, i0, P);
        const unsigned packed = all_region_one | (near_core << 8);
        for (int k = 0; k < n; k += 2)
        {
            const int j = begin + base + k;
            const LineWing la = wing[j], lb = wing[j + 1];
            const LineCore ca = core[j], cb = core[j + 1];
            core_line<P>(la, ca, (unsigned)__builtin_amdgcn_readlane((int)packed, k), v, acc);
            core_line<P>(lb, cb, (unsigned)__builtin_amdgcn_readlane((int)packed, k + 1), v, acc);
        }
    }
}

template <int P>
__device__ __forceinline__ void general_ranges(const LineWing * __restrict__ wing,
                                               const LineCore * __restrict__ core,
                                               const GeneralList & g, int i0, int i1, int lane,
                                               const double (&v)[P], double (&acc)[P])
{
    // The core range (the bulk) ...
    core_lines<P>(wing, core, g.begin[1], g.count[1] & ~1, i0, lane, v, acc);
    // ... and its odd line with the few lines of the other ranges, one at a time.
    GeneralList rest = g;
    rest.begin[1] = g.begin[1] + (g.count[1] & ~1);
    rest.count[1] = g.count[1] & 1;
    const int total = rest.count[0] + rest.count[1] + rest.count[2] + rest.count[3] + rest.count[4];
    for (int k = 0; k < total; ++k)
    {
        const int j = general_index(rest, k);
        const LineWing l = wing[j];
        const LineCore c = core[j];
        general_line<P>(l, c, i0, i1, lane, v, acc);
    }
}

// The lines whose windows END inside the tile (ranges [lo,a1) and [a2,hi)), when the cut-off is
// wide enough that the tile lies in the Lorentz wing of every one of them (the host's test,
// AccumulateArgs::inner_everywhere == 0).  Windows begin and end on integer wavenumbers
// (spectra.c:48-62), so neighbours in the sorted table mostly share theirs: eight lines with the
// same [first, last] are summed like a far-wing group -- one reciprocal, rows outside the window
// skipped, the row the window ends in masked -- instead of line by line, row by row.  On grids
// whose tiles are aligned to the 1 cm-1 cells this is the closing point of ~80 windows on the
// first tile of every cell; on coarse grids (100 points per cm-1) a tenth of all (line, tile)
// pairs.  Groups that straddle a change of window, and what is left over, go line by line.
template <int P>
__device__ __forceinline__ void clipped_ranges(const LineWing * __restrict__ wing,
                                               const LineCore * __restrict__ core,
                                               int begin0, int count0, int begin1, int count1,
                                               int i0, int i1, int lane,
                                               const double (&v)[P], double (&acc)[P])
{
    const int total = count0 + count1;
    int k = 0;
    while (k < total)
    {
        const int j = k < count0 ? begin0 + k : begin1 + (k - count0);
        const bool room = k < count0 ? k + 8 <= count0 : k + 8 <= total;
        if (room)
        {
            // Lanes 0-7 (and their copies) look at the eight windows; the records are asked for
            // at the same time (one round trip, not two: they are rarely not wanted).
            const LineWing * __restrict__ mine = wing + j + (lane & 7);
            const int my_first = mine->first, my_last = mine->last;
            WingTerm l[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                const LineWing w = wing[j + i];
                l[i] = WingTerm{w.centre, w.g2, w.bl};
            }
            const int first = __builtin_amdgcn_readfirstlane(my_first);
            const int last = __builtin_amdgcn_readfirstlane(my_last);
            if (__ballot(my_first != first || my_last != last) == 0)
            {
#pragma unroll
                for (int p = 0; p < P; ++p)
                {
                    const int r0 = i0 + p*64;
                    if (last < r0 || first > r0 + 63)
                    {
                        continue;
                    }
                    const int i = r0 + lane;
                    const double sum = lorentz_eight(v[p], l, acc[p]);
                    acc[p] = (i >= first && i <= last) ? sum : acc[p];
                }
                k += 8;
                continue;
            }
        }
        const LineWing w = wing[j];
        const LineCore c = core[j];
        general_line<P>(w, c, i0, i1, lane, v, acc);
        k += 1;
    }
}

// The second pass over the lines of the general list, kInnerQueue at a time with lane = line
// (their records arrive by one coalesced gather): the inner points the rows left out.  Kept
// apart from the walk above on purpose -- the two need different registers, and as one loop each
// paid for the other's (53 spilled SGPRs against 26, 1-6 % on workloads that have no inner point
// to evaluate).
__device__ __forceinline__ bool inner_ranges(const LineWing * __restrict__ wing,
                                             const LineCore * __restrict__ core,
                                             const GeneralList & g, InnerStage & stage,
                                             int i0, int i1, int v0, int n_per_v, double dv,
                                             int lane, double * slab, int slab_rows)
{
    bool used = false;
    const int total = g.count[0] + g.count[1] + g.count[2] + g.count[3] + g.count[4];
    for (int base = 0; base < total; base += kInnerQueue)
    {
        const int k = base + lane;
        const bool has = lane < kInnerQueue && k < total;
        // general_index() lane by lane.
        int j = g.begin[0], rest = has ? k : 0;
#pragma unroll
        for (int r = 0; r < 5; ++r)
        {
            const bool here = rest >= 0 && rest < g.count[r];
            j = here ? g.begin[r] + rest : j;
            rest = here ? -1 : rest - g.count[r];
        }
        used |= inner_batch(wing, core, stage, j, has, i0, i1, v0, n_per_v, dv, lane, slab, used,
                            slab_rows);
    }
    return used;
}

}  // namespace lbl
