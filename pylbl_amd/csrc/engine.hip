// Host side of the engine and the C ABI declared in include/lbl_amd.h.
//
// Replaces the body of the reference's absorption() (pyLBL/c_lib/absorption.c:19-99):
//   * the per-call SQLite read (absorption.c:44-73) becomes a one-time upload of a
//     wavenumber-sorted struct-of-arrays line table (lbl_molecule_load);
//   * the row loop calling spectra()/voigt() (absorption.c:76-86) becomes two kernel launches
//     per batch of levels -- prologue_kernel (per-line scalars and per-tile cut points) and
//     accumulate_kernel (the Voigt sums; combine_kernel after it where tiles were split) --
//     plus the pedestal kernels when remove_pedestal is set.
// Calls may be queued without waiting (LBL_ASYNC): the engine has several "lanes" (stream pair +
// workspace); calls that can share the GPU rotate over them and are ordered against each other
// only through the memory they write (Lane::note_write, lbl_engine::order_after_writers).
// There is no CPU compute path in this file: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/lbl_amd.h"
#include "accumulate.h"
#include "continuum.h"
#include "xsec.h"
#include "farfield.h"
#include "line_prep.h"
#include "pedestal.h"
#include "tile_schedule.h"

namespace {

using namespace lbl;

thread_local std::string g_create_error;
// The message of the calling thread's last failure and the handle it belongs to: what
// lbl_last_error returns, so that a thread never reads a message another thread is writing.
thread_local std::string g_thread_error;
thread_local const void * g_thread_error_engine = nullptr;

struct HipFailure
{
    std::string message;
};

#define HIP_TRY(call)                                                                     \
    do {                                                                                  \
        hipError_t status_ = (call);                                                      \
        if (status_ != hipSuccess)                                                        \
        {                                                                                 \
            throw HipFailure{std::string(#call) + ": " + hipGetErrorString(status_)};     \
        }                                                                                 \
    } while (0)

template <typename T>
struct DeviceBuffer
{
    T * data = nullptr;
    size_t capacity = 0;   // elements

    void reserve(size_t count)
    {
        if (count <= capacity) return;
        release();
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&data), count*sizeof(T)));
        capacity = count;
    }
    void release()
    {
        if (data != nullptr)
        {
            (void)hipFree(data);
            data = nullptr;
            capacity = 0;
        }
    }
    void upload(const T * host, size_t count, hipStream_t stream)
    {
        reserve(count);
        if (count > 0)
        {
            HIP_TRY(hipMemcpyAsync(data, host, count*sizeof(T), hipMemcpyHostToDevice, stream));
        }
    }
    ~DeviceBuffer() { release(); }
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer & operator=(const DeviceBuffer &) = delete;
};

struct Molecule
{
    long long n_lines = 0;
    // Host copies: row order (for the range rule) and sorted order (host prep, inspection).
    std::vector<double> nu_row;
    bool ascending = true;
    std::vector<int> order;                 // sorted position -> row
    std::vector<double> column[7];          // sorted: nu, sw, gamma_air, gamma_self, n_air, elower, delta_air
    std::vector<int> iso_slot;              // sorted
    double mass[kMassSlots];
    unsigned used_slots = 0;                // bit per isotopologue slot that has lines
    // Rows whose local_iso_id has no mass or no partition-function row.  The reference reads
    // past its tables for them (spectra.c:41-42); here they are an error -- but only when a
    // compute call would actually reach them (rows behind the range `break` never are).
    struct BadRow { int row; double nu; int local_iso_id; };
    std::vector<BadRow> bad_rows;
    double max_abs_delta = 0.;
    // Extremes over the table's rows, for the bound on y below which a level can have inner points.
    double min_gamma_air = 1.e300, min_gamma_self = 1.e300, min_n_air = 1.e300, max_n_air = -1.e300;
    int num_iso = 0, num_t = 0;
    std::vector<double> tips_t, tips_q;
    // Device copies (sorted).
    DeviceBuffer<double> d_column[7];
    DeviceBuffer<int> d_iso_slot, d_row, d_sorted_of_row;

    // Work-item plans, one per (grid, cut_off, tiling) this molecule has been computed on.
    struct Plan
    {
        int v0, vn, n_per_v, cut_off, points, aligned, farfield;
        int pieces = 1;         // the tiles in `pieces` runs of about equal weight (streamed calls)
        int n_items = 0, n_split = 0;
        long long partial_slots = 0;
        DeviceBuffer<WorkItem> items;       // piece-major, heaviest first within a piece
        DeviceBuffer<SplitTile> split;      // piece-major
        std::vector<int> item_begin, split_begin, tile_begin;   // [pieces + 1] each
    };
    std::vector<std::unique_ptr<Plan>> plans;

    LineTableView view() const
    {
        LineTableView v;
        v.nu = d_column[0].data; v.sw = d_column[1].data; v.gamma_air = d_column[2].data;
        v.gamma_self = d_column[3].data; v.n_air = d_column[4].data;
        v.elower = d_column[5].data; v.delta_air = d_column[6].data;
        v.iso_slot = d_iso_slot.data; v.row = d_row.data;
        v.sorted_of_row = d_sorted_of_row.data; v.n_lines = n_lines;
        return v;
    }
};

enum { kTimePrepare = 0, kTimeSchedule = 1, kTimeAccumulate = 2, kTimePedestal = 3,
       kTimeBandSpectra = 4, kTimeContinuum = 5, kTimeXsecModel = 6, kTimeXsec = 7,
       kTimeKinds = 8 };

// One in-flight compute call: its own pair of streams and its own workspace, so that
// several molecules can be in the pipeline at once (the serial pedestal chain of one
// overlaps the accumulate kernels of the others, and its own).
// A call on a tiny grid is three short dependent kernels (prologue -> accumulate -> combine) and its
// cost is their launches: such calls replay an instantiated HIP graph of the three, kept per lane
// and plan, whose kernel arguments are set afresh every call (engine option graphs).
struct SmallGraph
{
    const void * plan = nullptr;        // Molecule::Plan it was built for (items, split tiles)
    int count = 0, points = 0;
    unsigned prologue_blocks = 0, items = 0, combine_blocks = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraphNode_t prologue = nullptr, accumulate = nullptr, combine = nullptr;
    void destroy()
    {
        if (exec != nullptr) (void)hipGraphExecDestroy(exec);
        if (graph != nullptr) (void)hipGraphDestroy(graph);
        exec = nullptr;
        graph = nullptr;
    }
};

struct Lane
{
    std::vector<SmallGraph> graphs;     // most recently used last, at most 8
    hipStream_t main = nullptr;     // prepare, schedule, accumulate, apply, copies
    hipStream_t side = nullptr;     // the pedestal pre-pass
    hipEvent_t prepared = nullptr, pedestal_done = nullptr, levels_copied = nullptr;
    hipEvent_t runs_found = nullptr;
    hipEvent_t queued = nullptr;    // what the copy stream waits for (lbl_copy_rows_to_host)
    hipEvent_t handed_over = nullptr;   // what a caller's stream waits for (lbl_order_stream_after_engine)
    hipEvent_t piece_done[8] = {};      // behind the last kernel of each piece of a streamed call
    hipEvent_t piece_summed[8] = {};    // behind a piece's accumulate launch (pedestal: applied elsewhere)
    // The last part of a call with a pedestal -- the kernels that apply it to the caller's block,
    // piece by piece, and the copies of a streamed call -- kept back until lbl_finish_deferred
    // (LBL_DEFER_FINISH): everything before works in the lane's own buffers, so a long call can be
    // queued FIRST and still be the LAST to add into a block other calls write meanwhile.
    struct Finish
    {
        bool pending = false;
        int pieces = 1, count = 0, n_cells = 0, flags = 0;
        long long point_begin[9] = {};      // piece p covers points [point_begin[p], point_begin[p+1])
        int n_per_v = 0;
        const double * sums = nullptr;
        long long sums_stride = 0;
        double * target = nullptr;
        long long target_stride = 0;
        bool streamed = false, order_writers = false, add_into = false;
        char * host = nullptr;
        long long host_pitch = 0, columns = 0, base = 0;
        double * k = nullptr;
        long long out_bytes = 0;
        hipStream_t finish_stream = nullptr;
    } finish;
    // The last few writes of device output queued on this lane: where, and an event behind the
    // kernel that wrote.  A call on another lane that touches the same memory waits for it.
    struct Write { const char * begin = nullptr; const char * end = nullptr; hipEvent_t done = nullptr; };
    static constexpr int kWrites = 4;
    Write writes[kWrites];
    int next_write = 0;
    bool writes_wrapped = false;
    bool used = false;              // something was queued here since lane 0 last joined it
    bool levels_in_flight = false;
    DeviceBuffer<LineWing> wing;
    DeviceBuffer<LineCore> core;
    DeviceBuffer<TileSchedule> schedule;
    DeviceBuffer<LevelScalars> levels;
    DeviceBuffer<double> staging;   // spectra on their way to host memory
    DeviceBuffer<double> raw;       // un-pedestalled sums when the output must be added to
    DeviceBuffer<double> partial;   // partial sums of split tiles
    DeviceBuffer<double> far_series; // [levels][tiles][kFarTerms]
    DeviceBuffer<double> far_group;  // [levels][groups][kFarParts][kFarTerms]
    DeviceBuffer<GroupCuts> group_cuts;  // [levels][groups]
    DeviceBuffer<double> derived;
    DeviceBuffer<unsigned long long> evals;
    PedestalWorkspace pedestal;
    LevelScalars * pinned_levels = nullptr;
    size_t pinned_capacity = 0;

    void create(bool urgent = false)
    {
        // The pre-pass is short and latency-bound (a serial chain): its queue goes first
        // whenever the accumulate grid frees a slot.
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (urgent)
        {
            HIP_TRY(hipStreamCreateWithPriority(&main, hipStreamNonBlocking, greatest));
        }
        else
        {
            HIP_TRY(hipStreamCreateWithFlags(&main, hipStreamNonBlocking));
        }
        HIP_TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, greatest));
        HIP_TRY(hipEventCreateWithFlags(&prepared, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&runs_found, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&queued, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&handed_over, hipEventDisableTiming));
        for (auto & e : piece_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto & e : piece_summed) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto & w : writes) HIP_TRY(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&pedestal_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&levels_copied, hipEventDisableTiming));
    }
    void drain()
    {
        if (main != nullptr) (void)hipStreamSynchronize(main);
        if (side != nullptr) (void)hipStreamSynchronize(side);
    }
    void destroy()
    {
        drain();
        for (auto & graph : graphs) graph.destroy();
        graphs.clear();
        if (pinned_levels != nullptr) (void)hipHostFree(pinned_levels);
        pinned_levels = nullptr;
        if (prepared != nullptr) (void)hipEventDestroy(prepared);
        if (pedestal_done != nullptr) (void)hipEventDestroy(pedestal_done);
        if (runs_found != nullptr) (void)hipEventDestroy(runs_found);
        if (queued != nullptr) (void)hipEventDestroy(queued);
        if (handed_over != nullptr) (void)hipEventDestroy(handed_over);
        for (auto & e : piece_done) { if (e != nullptr) (void)hipEventDestroy(e); e = nullptr; }
        for (auto & e : piece_summed) { if (e != nullptr) (void)hipEventDestroy(e); e = nullptr; }
        for (auto & w : writes) { if (w.done != nullptr) (void)hipEventDestroy(w.done); w.done = nullptr; }
        if (levels_copied != nullptr) (void)hipEventDestroy(levels_copied);
        if (main != nullptr) (void)hipStreamDestroy(main);
        if (side != nullptr) (void)hipStreamDestroy(side);
        main = side = nullptr;
    }
    void reserve_pinned(size_t count)
    {
        if (count <= pinned_capacity) return;
        if (pinned_levels != nullptr) (void)hipHostFree(pinned_levels);
        pinned_levels = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&pinned_levels),
                              count*sizeof(LevelScalars), hipHostMallocDefault));
        pinned_capacity = count;
    }
    // Everything queued on `stream` (this lane's) so far has written [begin, end).
    void note_write(const void * begin, long long bytes, hipStream_t stream)
    {
        Write & w = writes[next_write];
        w.begin = reinterpret_cast<const char *>(begin);
        w.end = w.begin + bytes;
        HIP_TRY(hipEventRecord(w.done, stream));
        next_write = (next_write + 1) % kWrites;
        if (next_write == 0) writes_wrapped = true;
    }
};

constexpr int kLanes = 8;           // lanes the lines calls rotate over
// One more lane carries the continuum and cross-section calls: short, bandwidth-bound kernels
// on a stream of the highest priority, so that they are dispatched as soon as workgroup slots
// free up instead of queueing behind a resident accumulate grid of another lane (a 6 us
// band_spectra_kernel was seen waiting 0.9 ms for one).
constexpr int kSlotLane = kLanes;
constexpr int kAllLanes = kLanes + 1;

// Level scalars of a batched call on their way to the device: a pinned block, its device
// copy and an event that marks the last kernel reading them (and the per-level workspace that
// goes with them), so that a later call on the same object waits for that only.
template <typename Level>
struct LevelFeed
{
    DeviceBuffer<Level> levels;
    Level * pinned = nullptr;
    size_t pinned_capacity = 0;
    hipEvent_t done = nullptr;      // last kernel of the last call (the destructor waits for it)
    hipEvent_t copied = nullptr;    // last copy out of the pinned block
    bool in_flight = false;

    // The host may refill the pinned block once the copy that read it has run; everything on the
    // device side is ordered by the stream.
    void wait()
    {
        if (in_flight) HIP_TRY(hipEventSynchronize(copied));
        in_flight = false;
    }
    void copied_on(hipStream_t stream)
    {
        if (copied == nullptr) HIP_TRY(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(copied, stream));
        in_flight = true;
    }
    void mark(hipStream_t stream)
    {
        if (done == nullptr) HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(done, stream));
    }
    void reserve_pinned(size_t count)
    {
        if (count <= pinned_capacity) return;
        if (pinned != nullptr) (void)hipHostFree(pinned);
        pinned = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&pinned), count*sizeof(Level),
                              hipHostMallocDefault));
        pinned_capacity = count;
    }
    ~LevelFeed()
    {
        if (done != nullptr) { (void)hipEventSynchronize(done); (void)hipEventDestroy(done); }
        if (copied != nullptr) { (void)hipEventSynchronize(copied); (void)hipEventDestroy(copied); }
        if (pinned != nullptr) (void)hipHostFree(pinned);
    }
    LevelFeed() = default;
    LevelFeed(const LevelFeed &) = delete;
    LevelFeed & operator=(const LevelFeed &) = delete;
};

// One continuum (continuum.h): its bands, their coefficient table and the per-level
// workspace of coarse spectra.
struct ContinuumSet : LevelFeed<ContinuumLevel>
{
    BandSet set;
    DeviceBuffer<double> table;
    DeviceBuffer<double> coarse;        // [levels][set.coarse_points]
    DeviceBuffer<double> slopes;        // same shape: slope of the interval after each knot
    DeviceBuffer<double> staging;       // extinction on its way to host memory
    int widest = 0;                     // points of the largest band
};

// The cross-section bands of one molecule (xsec.h).
struct XsecData : LevelFeed<XsecLevel>
{
    XsecSet set;
    DeviceBuffer<double> fgrid;         // concatenated band frequency grids [Hz]
    DeviceBuffer<double> coeffs;        // per band [4][size]
    DeviceBuffer<double> values;        // [levels][set.total]: the fit on the bands' grids
    DeviceBuffer<double> slopes;        // same shape
    DeviceBuffer<double> staging;
};

struct SpectralGrid
{
    long long n = 0;
    bool ascending = true;
    DeviceBuffer<double> wavenumber;
};

}  // namespace

struct lbl_engine
{
    // Every entry point of the C ABI that takes this handle holds the mutex while it reads or
    // changes the engine's host-side state (lanes, plans, workspaces, write records, options) and
    // queues its work; the GPU work itself runs asynchronously.  The reference's absorption() has
    // no state at all (absorption.c:19-99) and ctypes releases the GIL around it
    // (gas_optics.py:79-91), so any number of threads may call it at once: so may they here.
    // (Recursive: lbl_synchronize finishes a deferred call through the public entry.)
    std::recursive_mutex mutex;
    int device = 0;
    hipStream_t stream = nullptr;   // == lanes[0].main: uploads, and what lbl_stream() returns
    hipStream_t copy_stream = nullptr;  // results on their way to host memory
    // The runtime multiplexes its streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4
    // by default) in an order of its own, and a queue runs its packets one after the other: a copy
    // that shares its queue with the main stream of the lane that is computing waits for that
    // lane's accumulate launches instead of running beside them.  For the call that delivers its
    // result piece by piece that was 1.9 against 1.6 ms (Spectroscopy's "total"), decided by
    // nothing but which lane the call had been dealt (profiles/r04_copy_streams.txt).  The engine
    // asks the GPU once, when it is created, which lanes share a queue with the copy stream
    // (calibrate_delivery_lanes), and a delivering call skips those.  (One copy stream per lane,
    // each chosen to run beside it, was tried first: slower than the best single one -- every
    // further stream in use is one more queue for the hardware to take turns on.)
    bool delivers_badly[kLanes] = {};
    hipEvent_t copies_handed_over = nullptr, taken_over = nullptr;  // lbl_order_*_after_*
    std::string error;
    std::vector<std::unique_ptr<Molecule>> molecules;
    std::vector<std::unique_ptr<ContinuumSet>> continua;
    std::vector<std::unique_ptr<SpectralGrid>> grids;
    std::vector<std::unique_ptr<XsecData>> xsecs;
    Lane lanes[kAllLanes];
    unsigned next_lane = 0;

    // Options.
    int prep = LBL_PREP_DEVICE;
    int points_per_lane = 0;
    int timing = 0;
    long long workspace_bytes = 4ll << 30;
    int ablate = 0;
    int aligned_tiles = 0;          // measured: no gain at 0.001 cm-1
    int overlap_pedestal = 1;       // run the pedestal pre-pass beside the accumulate kernel
    int farfield = 0;               // sum distant lines by their power series (farfield.h)
    int interp_shape = 0;           // continuum_interp_kernel<PT, LV> as 10 PT + LV, 0 = automatic
    int scan_chain = 1;             // pedestal chain by relaxation (pedestal.h), serial chain behind it
    int relax_launches = 0;         // relaxation launches before the serial chain (2 ... 7; 0: by the table)
    int item_floor = 0;             // fewest lines per work item (0 = by grid size; experiments)
    int lanes_in_use = 0;           // lanes the asynchronous calls rotate over; 0: by kind of call
    int graphs = 0;                 // 1: calls on tiny grids replay a HIP graph of their three kernels
    long long small_points = 1ll << 20;    // grids (points x levels) up to this size rotate too
    int order_runs = 1;             // accumulate launch waits for the pedestal's run-finding kernels
    int skip_delivery_lanes = 1;    // delivering calls avoid lanes that share the copy stream's queue

    // Timing.
    struct Span { hipEvent_t begin, end; int kind, counts; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    double time_ms[kTimeKinds] = {};
    long long launches[kTimeKinds] = {};

    hipEvent_t take_event()
    {
        if (!event_pool.empty())
        {
            hipEvent_t e = event_pool.back();
            event_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        return e;
    }

    template <typename F>
    void timed(int kind, hipStream_t on, F && launch, int counts = 1)
    {
        if (!timing || (timing == 2 && kind != kTimeAccumulate))
        {
            launch();
            return;
        }
        Span s{take_event(), take_event(), kind, counts};
        HIP_TRY(hipEventRecord(s.begin, on));
        launch();
        HIP_TRY(hipEventRecord(s.end, on));
        spans.push_back(s);
        if (spans.size() >= 4096) drain_spans();
    }

    void drain_spans()
    {
        for (auto & s : spans)
        {
            HIP_TRY(hipEventSynchronize(s.end));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, s.begin, s.end));
            time_ms[s.kind] += ms;
            launches[s.kind] += s.counts;
            event_pool.push_back(s.begin);
            event_pool.push_back(s.end);
        }
        spans.clear();
    }

    void drain_lanes()
    {
        for (auto & lane : lanes) lane.drain();
        if (copy_stream != nullptr) (void)hipStreamSynchronize(copy_stream);
    }

    // Orders `stream` (a stream of lane `self`) behind every write of [begin, begin + bytes) queued
    // on the other lanes.  Each lane remembers its last few writes; older ones were queued before
    // the oldest it remembers, whose event therefore stands in for them.
    void order_after_writers(hipStream_t stream, const void * begin, long long bytes,
                             const Lane * self)
    {
        const char * b = reinterpret_cast<const char *>(begin);
        const char * e = b + bytes;
        for (auto & lane : lanes)
        {
            if (&lane == self) continue;
            for (const auto & w : lane.writes)
            {
                if (w.begin != nullptr && b < w.end && w.begin < e)
                {
                    HIP_TRY(hipStreamWaitEvent(stream, w.done, 0));
                }
            }
            if (lane.writes_wrapped)
            {
                HIP_TRY(hipStreamWaitEvent(stream, lane.writes[lane.next_write].done, 0));
            }
        }
    }

    Lane * deferred = nullptr;      // the lane whose call waits for lbl_finish_deferred

    // Queues what Lane::Finish describes: apply kernels (+ copies) of every piece, then ties the
    // lane's main stream and the block's write record to the last of them.
    void run_finish(Lane & lane)
    {
        Lane::Finish & f = lane.finish;
        hipStream_t stream = lane.main;
        // (what is queued here is queued NOW: a call that joined this lane since the kept-back
        // call was made has to join it again)
        lane.used = true;
        if (f.order_writers)
        {
            order_after_writers(f.finish_stream, f.k, f.out_bytes, &lane);
        }
        for (int piece = 0; piece < f.pieces; ++piece)
        {
            const long long q0 = f.point_begin[piece], q1 = f.point_begin[piece + 1];
            if (q1 <= q0) continue;
            if (f.finish_stream != stream)
            {
                // (recorded behind this piece's accumulate launch)
                HIP_TRY(hipStreamWaitEvent(f.finish_stream, lane.piece_summed[piece], 0));
            }
            dim3 grid((unsigned)((q1 - q0 + 255)/256), (unsigned)f.count);
            hipLaunchKernelGGL(pedestal_apply_kernel, grid, dim3(256), 0, f.finish_stream, f.sums,
                               f.sums_stride, f.target, f.target_stride,
                               lane.pedestal.cell_sum.data, lane.pedestal.point_sum.data,
                               lane.levels.data, (int)q0, (int)q1, f.n_per_v, f.n_cells,
                               (f.flags & LBL_SCALE_DENSITY) ? 1 : 0, f.add_into ? 1 : 0);
            HIP_TRY(hipGetLastError());
            if (f.streamed && q0 < f.columns)
            {
                // This piece's columns go home beside the kernels of the next.  (The runtime's
                // device-to-host copy is a kernel of its own here, not a DMA engine; queued
                // beside an accumulate grid it costs the grid nothing, and a hand-written copy
                // kernel of 8..1024 workgroups did worse: profiles/r03_perf_deliver.txt.)
                const long long c1 = std::min<long long>(q1, f.columns);
                HIP_TRY(hipEventRecord(lane.piece_done[piece], f.finish_stream));
                HIP_TRY(hipStreamWaitEvent(copy_stream, lane.piece_done[piece], 0));
                HIP_TRY(hipMemcpy2DAsync(f.host + f.base*f.host_pitch + q0*8, (size_t)f.host_pitch,
                                         f.target + q0, (size_t)f.target_stride*8,
                                         (size_t)(c1 - q0)*8, (size_t)f.count,
                                         hipMemcpyDeviceToHost, copy_stream));
            }
        }
        if (f.finish_stream != stream)
        {
            // Later users of the lane's main stream (and of the block) come after the last apply.
            HIP_TRY(hipEventRecord(lane.pedestal_done, f.finish_stream));
            HIP_TRY(hipStreamWaitEvent(stream, lane.pedestal_done, 0));
        }
        if (f.k != nullptr)
        {
            lane.note_write(f.k, f.out_bytes, stream);
        }
        f.pending = false;
        if (deferred == &lane) deferred = nullptr;
    }

    void finish_deferred()
    {
        if (deferred != nullptr && deferred->finish.pending)
        {
            run_finish(*deferred);
        }
        deferred = nullptr;
    }

    // Drops what a call kept back instead of queueing it (lbl_cancel_deferred): its target block
    // and host range are never touched by that call.  What it has queued already works in the
    // lane's own buffers only.
    void cancel_deferred()
    {
        if (deferred != nullptr) deferred->finish.pending = false;
        deferred = nullptr;
    }

    // Orders `stream` (lane 0's) behind everything queued on the other lanes so far, without
    // stopping the host: what a call that adds into its output, or reuses lane 0 after calls
    // that rotated over the lanes, needs.
    void join_lanes(hipStream_t stream)
    {
        for (int i = 1; i < kAllLanes; ++i)
        {
            if (!lanes[i].used) continue;
            HIP_TRY(hipEventRecord(lanes[i].queued, lanes[i].main));
            HIP_TRY(hipStreamWaitEvent(stream, lanes[i].queued, 0));
            lanes[i].used = false;
        }
    }
};

namespace {

int fail(lbl_engine * engine, int code, const std::string & message)
{
    if (engine != nullptr)
    {
        std::lock_guard<std::recursive_mutex> guard(engine->mutex);
        engine->error = message;
        g_thread_error = message;
        g_thread_error_engine = engine;
    }
    else
    {
        g_create_error = message;
    }
    return code;
}

typedef std::lock_guard<std::recursive_mutex> EngineLock;

Molecule * find_molecule(lbl_engine * engine, int32_t handle)
{
    if (handle < 0 || (size_t)handle >= engine->molecules.size()) return nullptr;
    return engine->molecules[handle].get();
}

// spectral_database.c:97-104 with bounds checks the reference lacks.
bool tips_value(const Molecule & m, double temperature, int slot, double * value)
{
    const double * t = m.tips_t.data() + (size_t)slot*m.num_t;
    const double * q = m.tips_q.data() + (size_t)slot*m.num_t;
    const int i = (int)(floor(temperature)) - (int)(t[0]);
    if (i < 0 || i + 1 >= m.num_t) return false;
    *value = q[i] + (q[i+1] - q[i])*(temperature - t[i])/(t[i+1] - t[i]);
    return true;
}

bool fill_level(const Molecule & m, double temperature, double pressure, double vmr,
                LevelScalars & lv, std::string & why)
{
    const double pa_to_atm = 9.86923e-6;            // spectra.c:13
    const double r2 = 2*log(2)*8314.472;            // spectra.c:14
    const double kb = 1.38064852e-23;               // spectroscopy.py:15
    const double vlight = 2.99792458e8;
    const double sqrln2 = sqrt(log(2.));
    if (!(temperature > 0.) || !std::isfinite(temperature) || !std::isfinite(pressure) ||
        !std::isfinite(vmr))
    {
        why = "temperature, pressure and mixing ratio must be finite (temperature > 0).";
        return false;
    }
    std::memset(&lv, 0, sizeof(lv));
    lv.temperature = temperature;
    lv.p_atm = pressure*pa_to_atm;
    lv.p_partial = lv.p_atm*vmr;
    lv.tfact = 296./temperature;
    lv.t_minus_ref = temperature - 296.;
    lv.t_times_ref = temperature*296.;
    lv.density = pressure*vmr/(kb*temperature);
    double widest = 0.;
    for (int slot = 0; slot < kMassSlots; ++slot)
    {
        lv.doppler[slot] = 0.;
        lv.q_ratio[slot] = 0.;
        if (!(m.used_slots & (1u << slot))) continue;
        lv.doppler[slot] = sqrt(r2*temperature/m.mass[slot]);
        widest = std::max(widest, lv.doppler[slot]);
        double q_ref, q_t;
        if (!tips_value(m, 296., slot, &q_ref) || !tips_value(m, temperature, slot, &q_t))
        {
            why = "temperature " + std::to_string(temperature) +
                  " K (or 296 K) lies outside the partition-function table.";
            return false;
        }
        lv.q_ratio[slot] = q_ref/q_t;
    }
    lv.shift_max = fabs(lv.p_atm)*m.max_abs_delta*(1. + 1.e-12) + 1.e-12;
    // voigt.c:34: xlim0 = sqrt(15100 + y(40 - 3.6y)) <= 123.34 for every y.
    lv.core_reach = 123.4/sqrln2/vlight*widest*(1. + 1.e-9);
    return true;
}

int first_row_out_of_range(const Molecule & m, double nu_min, double nu_max)
{
    const auto & nu = m.nu_row;
    if (m.ascending)
    {
        if (nu.empty() || nu.front() < nu_min) return 0;
        return (int)(std::upper_bound(nu.begin(), nu.end(), nu_max) - nu.begin());
    }
    for (size_t i = 0; i < nu.size(); ++i)
    {
        if (nu[i] > nu_max || nu[i] < nu_min) return (int)i;   // absorption.c:80-83
    }
    return (int)nu.size();
}

// Points per lane P (tile = 64*P points) and the tiling.  Cell-aligned tiles are used when
// they waste at most 6 % of the lanes on padding.
int pick_tiling(const lbl_engine * engine, int farfield, int n_per_v, long long n, Tiling & tiling)
{
    const int forced = engine->points_per_lane;
    const bool is_forced = forced == 1 || forced == 2 || forced == 4 || forced == 8;
    const int candidates[4] = {8, 4, 2, 1};
    for (int c = 0; c < 4; ++c)
    {
        const int p = is_forced ? forced : candidates[c];
        const int width = 64*p;
        const int per_cell = (n_per_v + width - 1)/width;
        const double waste = (double)per_cell*width/n_per_v - 1.;
        // (With the far-field series the tiles are cell-aligned wherever that wastes few lanes: no
        // window ends inside a tile then -- bar the closing point -- and the lines left to the
        // direct kernel are few enough for that to show: 0.94 -> 0.86 ms per step on the 5 M-point
        // workload.  The direct kernel alone gains nothing, profiles/r03_ab_tiling.txt.)
        if ((engine->aligned_tiles || farfield) && waste <= 0.06)
        {
            tiling.aligned = 1;
            tiling.per_cell = per_cell;
            tiling.length = (n_per_v + per_cell - 1)/per_cell;
            tiling.n_tiles = (int)(n/n_per_v)*per_cell;
            return p;
        }
        if (is_forced) break;
    }
    int p = forced;
    if (!is_forced)
    {
        // Measured on the 0.001 cm-1 workload: 4 is ~2 % ahead of 8 for the direct kernel,
        // 8 is ahead when the far-field series carries most lines.  With the series a tile does
        // best at 0.5-1.3 cm-1 (the lines within four half-widths stay with the direct kernel):
        // at 0.01 cm-1 two points per lane, 0.467 -> 0.423 ms per step
        // (profiles/r04_farfield_small_grids.txt).
        p = (n_per_v >= 400 && farfield) ? 8 : (n_per_v >= 100 && farfield) ? 2
            : n_per_v >= 100 ? 4 : n_per_v >= 10 ? 2 : 1;
    }
    tiling.aligned = 0;
    tiling.per_cell = 0;
    tiling.length = 64*p;
    tiling.n_tiles = (int)((n + tiling.length - 1)/tiling.length);
    return p;
}

// Splits the tiles into work items of bounded size and orders them heaviest first.  Line
// counts per tile come from the sorted wavenumbers alone (pressure shifts move them by a line
// or two, which does not matter for balance); the exact per-level cut points are still the
// schedule kernel's.
Molecule::Plan & plan_for(lbl_engine * engine, int farfield, Molecule & m, const GridSpec & g,
                          const Tiling & tiling, int points, hipStream_t stream, int pieces = 1)
{
    for (size_t i = 0; i < m.plans.size(); ++i)
    {
        Molecule::Plan * p = m.plans[i].get();
        if (p->v0 == g.v0 && p->vn == g.vn && p->n_per_v == g.n_per_v &&
            p->cut_off == g.cut_off && p->points == points && p->aligned == tiling.aligned &&
            p->farfield == farfield && p->pieces == pieces)
        {
            // Most recently used last.
            std::rotate(m.plans.begin() + i, m.plans.begin() + i + 1, m.plans.end());
            return *m.plans.back();
        }
    }
    // A long-lived process may see many grids: keep the 16 most recent plans per molecule.
    if (m.plans.size() >= 16)
    {
        engine->drain_lanes();      // a queued kernel may still read the oldest plan's items
        m.plans.erase(m.plans.begin());
    }
    std::unique_ptr<Molecule::Plan> plan(new Molecule::Plan());
    plan->v0 = g.v0; plan->vn = g.vn; plan->n_per_v = g.n_per_v; plan->cut_off = g.cut_off;
    plan->points = points; plan->aligned = tiling.aligned; plan->farfield = farfield;
    plan->pieces = pieces;
    const int n_tiles = tiling.n_tiles;
    const std::vector<double> & nu = m.column[0];
    std::vector<long long> weight((size_t)n_tiles);
    std::vector<long long> clipped((size_t)n_tiles, 0);
    long long total = 0;
    for (int t = 0; t < n_tiles; ++t)
    {
        long long i0, i1;
        tile_bounds(tiling, t, g.n_per_v, g.n, i0, i1);
        double lo = (double)((i0 + g.n_per_v - 1)/g.n_per_v + g.v0 - g.cut_off - 1) - 0.05;
        double hi = (double)(i1/g.n_per_v + g.v0 + g.cut_off) + 1.05;
        if (farfield)
        {
            // Only the lines near the tile are evaluated point by point -- and those whose windows
            // end inside it (the ranges [lo,a1) and [a2,hi) of schedule_tile), which the series
            // cannot take: a few dozen on an even table, thousands where a dense cm-1 of lines
            // closes on the tile (counted apart, see below).
            const double full_lo = (double)((i1 + g.n_per_v - 1)/g.n_per_v + g.v0 - g.cut_off - 1);
            const double full_hi = (double)(i0/g.n_per_v + g.v0 + g.cut_off) + 1.;
            clipped[t] = std::max<long long>(0, std::lower_bound(nu.begin(), nu.end(), full_lo + 0.05) -
                                                std::lower_bound(nu.begin(), nu.end(), lo)) +
                         std::max<long long>(0, std::lower_bound(nu.begin(), nu.end(), hi) -
                                                std::lower_bound(nu.begin(), nu.end(), full_hi - 0.05));
            const double u0 = tile_centre(g.v0, g.dv, i0, i1);
            const double radius = kFarRatio*0.5*(double)(i1 - i0)*g.dv + 0.6;
            lo = std::max(lo, u0 - radius);
            hi = std::min(hi, u0 + radius);
        }
        weight[t] = std::lower_bound(nu.begin(), nu.end(), hi) -
                    std::lower_bound(nu.begin(), nu.end(), lo);
        total += weight[t];
    }
    // Aim for ~8 items per workgroup slot of the chip (256 CUs x 6 resident workgroups): enough
    // rounds that the last one costs little, few enough that uniform tables leave their tiles
    // whole -- every extra item pays the kernel's prologue again and every split tile a pass
    // of combine_kernel (A/B on the 5 M-point workloads: 2-3 % against items half that size).
    // Dense bands still get their heavy tiles cut.  On small grids (a launch does not fill the
    // chip; every scalar load is a miss) short chains of lines per wavefront matter more than the
    // per-item overhead: items down to 128 lines.  (Grids in between -- 0.01 cm-1 over 5000 cm-1,
    // ~2000 tiles -- do best with at least 2048 lines per item now that their calls take turns on
    // two lanes: 0.590 -> 0.579 ms per step against 1024, which had been 2 % ahead of 512.)
    const long long floor_lines = engine->item_floor > 0 ? engine->item_floor
                                  : (n_tiles < 1024 || farfield) ? 128 : 2048;
    const long long target = std::max<long long>(floor_lines, total/(8*1536) + 1);
    // Streamed calls (lbl_compute_streamed) launch the tiles in `pieces` runs, each followed by
    // the copy of its columns: runs of about equal weight, every tile counted with a floor that
    // stands for its fixed cost.
    plan->tile_begin.assign((size_t)pieces + 1, n_tiles);
    plan->tile_begin[0] = 0;
    {
        const long long unit = std::max<long long>(total/std::max(n_tiles, 1), 1);
        const long long all = total + unit*n_tiles;
        long long running = 0;
        int piece = 1;
        for (int t = 0; t < n_tiles && piece < pieces; ++t)
        {
            running += weight[t] + unit;
            if (running*pieces >= all*piece)
            {
                plan->tile_begin[piece++] = t + 1;
            }
        }
    }
    std::vector<int> piece_of_tile((size_t)n_tiles, 0);
    for (int piece = 0; piece < pieces; ++piece)
    {
        for (int t = plan->tile_begin[piece]; t < plan->tile_begin[piece + 1]; ++t)
        {
            piece_of_tile[t] = piece;
        }
    }
    std::vector<WorkItem> items;
    std::vector<SplitTile> split;
    std::vector<long long> item_weight;
    long long slots = 0;
    plan->split_begin.assign((size_t)pieces + 1, 0);
    for (int t = 0; t < n_tiles; ++t)
    {
        int parts = (int)std::min<long long>(64, (weight[t] + target - 1)/target);
        if (parts < 1) parts = 1;
        // Far-field plans: a tile on which many windows end (each such line costs about three
        // evaluated ones, row by row) is cut by that load too; the plans of even tables stay as
        // they are.  (A 150 000-line table with 7 330 lines below 1 cm-1: four tiles at 25-27 cm-1
        // held the whole launch, 1.70 -> 0.7 ms per spectrum; profiles/r04_hitran_shaped.txt.)
        long long load = weight[t];
        if (3*clipped[t] > 4*target)
        {
            load += 3*clipped[t];
            parts = (int)std::min<long long>(64, (load + target - 1)/target);
        }
        const int slot = parts > 1 ? (int)slots : -1;
        if (parts > 1)
        {
            split.push_back(SplitTile{t, parts, slot, 0});      // ascending tiles: piece-major
            plan->split_begin[piece_of_tile[t] + 1] += 1;
            slots += parts;
        }
        for (int k = 0; k < parts; ++k)
        {
            items.push_back(WorkItem{t, k, parts, slot});
            item_weight.push_back(load/parts);
        }
    }
    for (int piece = 0; piece < pieces; ++piece)
    {
        plan->split_begin[piece + 1] += plan->split_begin[piece];
    }
    std::vector<int> order(items.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        const int pa = piece_of_tile[items[a].tile], pb = piece_of_tile[items[b].tile];
        return pa != pb ? pa < pb : item_weight[a] > item_weight[b];
    });
    plan->item_begin.assign((size_t)pieces + 1, 0);
    for (const auto & item : items)
    {
        plan->item_begin[piece_of_tile[item.tile] + 1] += 1;
    }
    for (int piece = 0; piece < pieces; ++piece)
    {
        plan->item_begin[piece + 1] += plan->item_begin[piece];
    }
    std::vector<WorkItem> sorted(items.size());
    for (size_t i = 0; i < items.size(); ++i) sorted[i] = items[order[i]];
    plan->n_items = (int)sorted.size();
    plan->n_split = (int)split.size();
    plan->partial_slots = slots;
    plan->items.upload(sorted.data(), sorted.size(), stream);
    plan->split.upload(split.data(), split.size(), stream);
    HIP_TRY(hipStreamSynchronize(stream));      // the host vectors go out of scope
    m.plans.push_back(std::move(plan));
    return *m.plans.back();
}

void launch_accumulate(int points, dim3 grid, hipStream_t stream, const AccumulateArgs & args)
{
    switch (points)
    {
    case 1: hipLaunchKernelGGL(accumulate_kernel<1>, grid, dim3(256), 0, stream, args); break;
    case 2: hipLaunchKernelGGL(accumulate_kernel<2>, grid, dim3(256), 0, stream, args); break;
    case 4: hipLaunchKernelGGL(accumulate_kernel<4>, grid, dim3(256), 0, stream, args); break;
    default: hipLaunchKernelGGL(accumulate_kernel<8>, grid, dim3(256), 0, stream, args); break;
    }
    HIP_TRY(hipGetLastError());
}

// Replays (builds at first use) the graph prologue -> accumulate -> combine of a call on a tiny
// grid with this call's arguments.
void launch_small_graph(Lane & lane, const Molecule::Plan & plan, const Molecule & m,
                        LevelScalars * levels, const InlineLevels & packed, const GridSpec & g,
                        const RangeRule & rule, const Tiling & tiling, int prepare_blocks,
                        unsigned prologue_blocks, int count, int points,
                        const AccumulateArgs & args, hipStream_t stream)
{
    // The kernels' arguments, in their declared order.
    LineTableView view = m.view();
    InlineLevels inline_levels = packed;
    int use_inline = 1, farfield = 0, blocks = prepare_blocks;
    GridSpec grid_spec = g;
    RangeRule range_rule = rule;
    Tiling tiles = tiling;
    LineWing * wing = const_cast<LineWing *>(args.wing);
    LineCore * core = const_cast<LineCore *>(args.core);
    TileSchedule * schedule = const_cast<TileSchedule *>(args.schedule);
    double * derived = nullptr;
    unsigned long long * evals = nullptr;
    void * prologue_args[] = {&view, &levels, &inline_levels, &use_inline, &grid_spec, &range_rule,
                              &tiles, &farfield, &blocks, &wing, &core, &schedule, &derived,
                              &evals};
    AccumulateArgs accumulate = args;
    accumulate.items = plan.items.data;
    void * accumulate_args[] = {&accumulate};
    AccumulateArgs combine = args;
    const SplitTile * split = plan.split.data;
    int n_split = plan.n_split, row_points = 64*points;
    void * combine_args[] = {&combine, &split, &n_split, &row_points};

    const unsigned items = (unsigned)plan.n_items;
    const unsigned combine_blocks = plan.n_split > 0 ? (unsigned)((plan.n_split*points + 3)/4) : 0;
    hipKernelNodeParams node[3] = {};
    node[0].func = reinterpret_cast<void *>(&prologue_kernel);
    node[0].gridDim = dim3(prologue_blocks, (unsigned)count);
    node[0].kernelParams = prologue_args;
    switch (points)
    {
    case 1: node[1].func = reinterpret_cast<void *>(&accumulate_kernel<1>); break;
    case 2: node[1].func = reinterpret_cast<void *>(&accumulate_kernel<2>); break;
    case 4: node[1].func = reinterpret_cast<void *>(&accumulate_kernel<4>); break;
    default: node[1].func = reinterpret_cast<void *>(&accumulate_kernel<8>); break;
    }
    node[1].gridDim = dim3(items, (unsigned)count);
    node[1].kernelParams = accumulate_args;
    node[2].func = reinterpret_cast<void *>(&combine_kernel);
    node[2].gridDim = dim3(std::max(combine_blocks, 1u), (unsigned)count);
    node[2].kernelParams = combine_args;
    for (auto & n : node)
    {
        n.blockDim = dim3(256);
        n.sharedMemBytes = 0;
        n.extra = nullptr;
    }

    SmallGraph * found = nullptr;
    for (size_t i = 0; i < lane.graphs.size(); ++i)
    {
        SmallGraph & c = lane.graphs[i];
        if (c.plan == &plan && c.count == count && c.points == points &&
            c.prologue_blocks == prologue_blocks && c.items == items &&
            c.combine_blocks == combine_blocks)
        {
            std::rotate(lane.graphs.begin() + i, lane.graphs.begin() + i + 1, lane.graphs.end());
            found = &lane.graphs.back();
            break;
        }
    }
    if (found == nullptr)
    {
        if (lane.graphs.size() >= 8)
        {
            HIP_TRY(hipStreamSynchronize(stream));      // the oldest graph may still be running
            lane.graphs.front().destroy();
            lane.graphs.erase(lane.graphs.begin());
        }
        SmallGraph fresh;
        fresh.plan = &plan;
        fresh.count = count;
        fresh.points = points;
        fresh.prologue_blocks = prologue_blocks;
        fresh.items = items;
        fresh.combine_blocks = combine_blocks;
        HIP_TRY(hipGraphCreate(&fresh.graph, 0));
        HIP_TRY(hipGraphAddKernelNode(&fresh.prologue, fresh.graph, nullptr, 0, &node[0]));
        HIP_TRY(hipGraphAddKernelNode(&fresh.accumulate, fresh.graph, &fresh.prologue, 1, &node[1]));
        if (combine_blocks > 0)
        {
            HIP_TRY(hipGraphAddKernelNode(&fresh.combine, fresh.graph, &fresh.accumulate, 1,
                                          &node[2]));
        }
        HIP_TRY(hipGraphInstantiate(&fresh.exec, fresh.graph, nullptr, nullptr, 0));
        lane.graphs.push_back(fresh);
        found = &lane.graphs.back();
    }
    else
    {
        HIP_TRY(hipGraphExecKernelNodeSetParams(found->exec, found->prologue, &node[0]));
        HIP_TRY(hipGraphExecKernelNodeSetParams(found->exec, found->accumulate, &node[1]));
        if (combine_blocks > 0)
        {
            HIP_TRY(hipGraphExecKernelNodeSetParams(found->exec, found->combine, &node[2]));
        }
    }
    HIP_TRY(hipGraphLaunch(found->exec, stream));
}

struct ComputeRequest
{
    int32_t molecule, n_levels;
    const double * temperature, * pressure, * vmr;
    int32_t v0, vn, n_per_v, cut_off, remove_pedestal, range_policy, flags;
    double * k;
    int64_t level_stride;
    int64_t * evals;
    double * derived;      // host, n_lines x 8 in row order, single level only
    // lbl_compute_streamed: the first `columns` points of every level also go to host memory,
    // piece by piece as the kernels of a piece finish.
    char * host = nullptr;
    int64_t host_pitch = 0, columns = 0;
    int32_t pieces = 1;
};

// wait_for: where a blocking call (no LBL_ASYNC) leaves an event behind its last operation instead
// of waiting for it -- the caller waits after it has released the engine's mutex, so that other
// threads queue their calls meanwhile.  nullptr: wait here.
int compute(lbl_engine * engine, const ComputeRequest & rq, hipEvent_t * wait_for = nullptr)
{
    Molecule * m = find_molecule(engine, rq.molecule);
    if (m == nullptr) return fail(engine, LBL_BAD_ARGUMENT, "unknown molecule handle.");
    if (rq.n_levels < 1 || rq.temperature == nullptr || rq.pressure == nullptr ||
        rq.vmr == nullptr)
    {
        return fail(engine, LBL_BAD_ARGUMENT, "need at least one level with T, P and x.");
    }
    if (rq.n_per_v < 1 || rq.vn <= rq.v0 || rq.cut_off < 0)
    {
        return fail(engine, LBL_BAD_ARGUMENT,
                    "need vn > v0, n_per_v >= 1 and cut_off >= 0 (grid must start on an "
                    "integer wavenumber with spacing 1/integer).");
    }
    const long long n_long = (long long)(rq.vn - rq.v0)*rq.n_per_v;
    if (n_long > 0x3fffffff)
    {
        return fail(engine, LBL_BAD_ARGUMENT, "grid has more than 2^30 points.");
    }
    if (rq.k == nullptr && rq.derived == nullptr)
    {
        return fail(engine, LBL_BAD_ARGUMENT, "k is NULL.");
    }
    GridSpec g;
    g.v0 = rq.v0; g.vn = rq.vn; g.n_per_v = rq.n_per_v; g.cut_off = rq.cut_off;
    g.n = (int)n_long;
    g.dv = 1./rq.n_per_v;
    const long long stride = rq.level_stride > 0 ? rq.level_stride : n_long;
    if (stride < n_long) return fail(engine, LBL_BAD_ARGUMENT, "level_stride < grid points.");

    RangeRule rule;
    rule.policy = rq.range_policy == LBL_RANGE_SKIP ? 1 : 0;
    rule.nu_min = rq.v0 - (rq.cut_off + 1);
    rule.nu_max = rq.vn + rq.cut_off + 1;
    rule.row_limit = first_row_out_of_range(*m, rule.nu_min, rule.nu_max);
    for (const auto & bad : m->bad_rows)
    {
        if (line_accepted(rule, bad.nu, bad.row))
        {
            return fail(engine, LBL_OUT_OF_RANGE,
                        "row " + std::to_string(bad.row) + ": local_iso_id " +
                        std::to_string(bad.local_iso_id) +
                        " has no mass or no partition-function row.");
        }
    }

    Tiling tiling;
    // The far-field series: an engine-wide option, or this call's choice (LBL_FARFIELD).
    int farfield = (engine->farfield || (rq.flags & LBL_FARFIELD)) ? 1 : 0;
    int points = pick_tiling(engine, farfield, rq.n_per_v, n_long, tiling);
    if (farfield && kFarRatio*0.5*(double)tiling.length >= (double)rq.cut_off*rq.n_per_v)
    {
        // A tile so wide (coarse grids: 0.1 cm-1 and up) that no line of a window is kFarRatio
        // half-widths away: the series would sum nothing and its two kernels only cost their
        // launches (configs[0]: 0.033 -> 0.041 ms per step, profiles/r04_farfield_small_grids.txt).
        farfield = 0;
        points = pick_tiling(engine, farfield, rq.n_per_v, n_long, tiling);
    }
    const int n_tiles = tiling.n_tiles;
    const int n_cells = rq.vn - rq.v0;
    const bool out_device = (rq.flags & LBL_OUT_DEVICE) != 0;
    const bool want_k = rq.k != nullptr;
    const long long n_lines = m->n_lines;

    try
    {
        HIP_TRY(hipSetDevice(engine->device));
        // Asynchronous calls rotate over the lanes where sharing the GPU pays: calls with a
        // pedestal (their serial chain leaves it almost idle) and calls on small grids; plain
        // calls on large grids run back to back on lane 0, behind everything the other lanes
        // hold, as does anything the caller waits for.
        // Small grids (a launch does not fill the chip, latency rules) gain the same way: up to 2^20
        // points x levels (BASELINE configs[1], 500 k points: 0.625 -> 0.597 ms per step with four
        // calls in flight; at 5 M points it is -1 %, and left alone so that a launch's duration
        // stays what it takes alone).
        const bool small = n_long*rq.n_levels <= engine->small_points;
        // With the far-field series a call is five short kernels whatever the grid: the next call's
        // prologue and series kernels fit beside this one's accumulate kernel (0.83 -> 0.77 ms per
        // step at 5 M points).
        const bool short_kernels = farfield != 0 && engine->small_points > 0;
        // A call that adds into its output can share the GPU too when it removes the pedestal:
        // only its last kernel (pedestal_apply_kernel) touches the output, everything before
        // works in the lane's own buffers.
        const bool add_into_block = (rq.flags & LBL_ACCUMULATE) != 0;
        // (A molecule without lines has no pedestal pass: its accumulate kernel itself adds
        // into the block, so such a call must not leave lane 0's ordering.)
        const bool pedestal_pass = rq.remove_pedestal && n_lines > 0;
        const bool alternate = (rq.flags & LBL_ASYNC) && want_k && rq.derived == nullptr &&
                               ((pedestal_pass && (!add_into_block || out_device)) ||
                                ((small || short_kernels) && out_device && !add_into_block));
        if ((rq.flags & LBL_DEFER_FINISH) ||
            (!alternate && engine->deferred == &engine->lanes[0]))
        {
            // There is one deferral at a time; and a call about to reuse lane 0's buffers must not
            // find a kept-back call still needing them.  (Kept-back calls stay off lane 0, below,
            // so a plain call -- another thread's, say -- leaves a deferral alone: the order in
            // which a pipeline's calls add into their block does not depend on who else uses the
            // engine.  The kept-back kernels order themselves behind every write of their block
            // when they are queued, run_finish.)
            engine->finish_deferred();
        }
        int lane_index = 0;
        if (alternate)
        {
            // How many lanes: the runtime maps streams onto four hardware queues, and calls on more
            // lanes than that only stretch one another's kernels.  Measured (profiles/
            // r03_ab_lanes.txt): calls with a pedestal pass (two streams each) 4 lanes against 8:
            // -1 % on the default workload, -8...-12 % with the far-field series; calls on tiny
            // grids (three short dependent kernels, nothing to overlap but launch gaps) 2 lanes:
            // 23.6 us per call against 33-35 us on 4 or 8.
            const int rotate = engine->lanes_in_use > 0 ? engine->lanes_in_use
                                                        : (pedestal_pass ? 4 : 2);
            lane_index = (int)(engine->next_lane++ % rotate);
            if (&engine->lanes[lane_index] == engine->deferred ||
                (lane_index == 0 && (rq.flags & LBL_DEFER_FINISH)))
            {
                lane_index = (int)(engine->next_lane++ % rotate);
            }
            // A call that delivers its result while it computes: not on a lane whose accumulate
            // launches would hold up its copies (see delivers_badly).
            for (int tries = 1; tries < rotate && rq.host != nullptr && out_device &&
                                engine->skip_delivery_lanes && engine->delivers_badly[lane_index];
                 ++tries)
            {
                const int next = (int)(engine->next_lane % rotate);
                if (&engine->lanes[next] == engine->deferred) break;
                lane_index = next;
                engine->next_lane += 1;
            }
        }
        Lane & lane = engine->lanes[lane_index];
        hipStream_t stream = lane.main;
        const long long out_bytes = ((long long)(rq.n_levels - 1)*stride + n_long)*8;
        if (!alternate)
        {
            if ((rq.flags & LBL_ASYNC) && out_device)
            {
                engine->join_lanes(stream);     // the host keeps queueing
            }
            else
            {
                for (int i = 1; i < kAllLanes; ++i) engine->lanes[i].drain();
            }
        }
        else
        {
            lane.used = true;
            if (out_device && !add_into_block)
            {
                // Calls on different lanes run side by side; two that write the same memory must
                // not: the later one waits for the earlier one's last kernel.  (A call that adds
                // into the block waits later, in front of the one kernel that does the adding.)
                engine->order_after_writers(stream, rq.k, out_bytes, &lane);
            }
        }

        const bool streamed = rq.host != nullptr && out_device && want_k;
        const int pieces = streamed ? std::max(1, std::min<int>(rq.pieces, std::min(n_tiles, 8)))
                                    : 1;
        Molecule::Plan & plan = plan_for(engine, farfield, *m, g, tiling, points, stream, pieces);

        // Levels per pass, bounded by the workspace budget.
        const long long per_level = plan.partial_slots*64*points*8 + n_lines*(long long)(sizeof(LineWing) + sizeof(LineCore)) +
                                    (long long)n_tiles*sizeof(TileSchedule) +
                                    (out_device ? 0 : n_long*8) +
                                    (rq.remove_pedestal ? pedestal_bytes_per_level(n_lines, n_cells, rq.cut_off) : 0);
        long long chunk = std::max(1ll, engine->workspace_bytes/std::max(per_level, 1ll));
        chunk = std::min<long long>(chunk, rq.n_levels);
        if (chunk > 65535) chunk = 65535;

        const bool with_pedestal = rq.remove_pedestal && n_lines > 0 && want_k;
        const bool add_into = (rq.flags & LBL_ACCUMULATE) != 0;
        if (lane.levels_in_flight)
        {
            // The previous call on this lane may still be copying from the pinned block.
            HIP_TRY(hipEventSynchronize(lane.levels_copied));
            lane.levels_in_flight = false;
        }
        lane.reserve_pinned((size_t)chunk);
        lane.levels.reserve((size_t)chunk);
        lane.wing.reserve((size_t)(chunk*std::max(n_lines, 1ll)));
        lane.core.reserve((size_t)(chunk*std::max(n_lines, 1ll)));
        lane.schedule.reserve((size_t)(chunk*n_tiles));
        if (want_k && !out_device) lane.staging.reserve((size_t)(chunk*n_long));
        if (with_pedestal && out_device && add_into) lane.raw.reserve((size_t)(chunk*n_long));
        const int n_groups = (n_tiles + kFarGroup - 1)/kFarGroup;
        if (want_k && farfield)
        {
            lane.far_series.reserve((size_t)(chunk*n_tiles*kFarTerms));
            lane.far_group.reserve((size_t)(chunk*n_groups*kFarParts*kFarTerms));
            lane.group_cuts.reserve((size_t)(chunk*n_groups));
        }
        if (want_k) lane.partial.reserve((size_t)std::max(1ll, chunk*plan.partial_slots*64*points));
        if (rq.evals != nullptr)
        {
            lane.evals.reserve(1);
            HIP_TRY(hipMemsetAsync(lane.evals.data, 0, sizeof(unsigned long long), stream));
        }
        if (rq.derived != nullptr) lane.derived.reserve((size_t)(std::max(n_lines, 1ll)*8));

        std::vector<LineWing> host_wing;
        std::vector<LineCore> host_core;
        std::vector<double> host_derived;
        bool deferred_finish = false;

        for (long long base = 0; base < rq.n_levels; base += chunk)
        {
            const int count = (int)std::min<long long>(chunk, rq.n_levels - base);
            // The previous pass may still be reading the pinned block.
            if (base > 0) HIP_TRY(hipStreamSynchronize(stream));
            for (int l = 0; l < count; ++l)
            {
                std::string why;
                LevelScalars & lv = lane.pinned_levels[l];
                if (!fill_level(*m, rq.temperature[base + l], rq.pressure[base + l],
                                rq.vmr[base + l], lv, why))
                {
                    return fail(engine, LBL_OUT_OF_RANGE,
                                "level " + std::to_string(base + l) + ": " + why);
                }
                // Can any accepted line have y < 8.425 at this level (voigt.c:35-43: below that
                // the inner regions exist)?  y = sqrt(ln2) gamma/alpha with gamma >= the
                // smallest half-widths of the table at this pressure (spectra.c:25-26) and alpha
                // <= the widest Doppler width at the largest accepted wavenumber (spectra.c:29,
                // absorption.c:80-83).  The accumulate kernel skips its look for inner points at
                // levels where the answer is no: all of a CO2 table at 1 atm, for instance.
                lv.inner_possible = 1.;
                const double foreign = lv.p_atm - lv.p_partial;
                if (foreign >= 0. && lv.p_partial >= 0. && m->min_gamma_air >= 0. &&
                    m->min_gamma_self >= 0. && m->n_lines > 0)
                {
                    double widest = 0.;
                    for (int slot = 0; slot < kMassSlots; ++slot)
                    {
                        widest = std::max(widest, lv.doppler[slot]);
                    }
                    const double power = std::min(pow(lv.tfact, m->min_n_air),
                                                  pow(lv.tfact, m->max_n_air));
                    const double gamma = (m->min_gamma_air*foreign +
                                          m->min_gamma_self*lv.p_partial)*power;
                    const double alpha = (rule.nu_max/2.99792458e8)*widest;
                    const double y_lower = sqrt(log(2.))*gamma/alpha*(1. - 1.e-9);
                    if (alpha > 0. && y_lower >= 8.425)
                    {
                        lv.inner_possible = 0.;
                    }
                }
            }
            // A few levels travel as kernel arguments of the prologue kernel (no copy in front
            // of it); more go through the pinned block.
            const bool host_prep = engine->prep == LBL_PREP_HOST;
            const bool inline_levels = !host_prep && count <= kInlineLevels;
            InlineLevels packed;
            int prepare_blocks = 0;
            unsigned prologue_blocks = 1;
            // Tiny grids: the call's three kernels as one replayed graph (SmallGraph).
            const bool graphed = engine->graphs && alternate && small && !with_pedestal &&
                                 !farfield && inline_levels && want_k && out_device && !streamed &&
                                 rq.evals == nullptr && rq.derived == nullptr &&
                                 chunk >= rq.n_levels && engine->timing == 0 && n_lines > 0;
            if (!inline_levels)
            {
                HIP_TRY(hipMemcpyAsync(lane.levels.data, lane.pinned_levels,
                                       count*sizeof(LevelScalars), hipMemcpyHostToDevice, stream));
                HIP_TRY(hipEventRecord(lane.levels_copied, stream));
                lane.levels_in_flight = true;
            }

            // K1: per-line scalars (+ the tile cut points, in the same launch).
            if (host_prep)
            {
                if (n_lines > 0)
                {
                    host_wing.resize((size_t)(count*n_lines));
                    host_core.resize((size_t)(count*n_lines));
                    if (rq.derived != nullptr) host_derived.assign((size_t)(n_lines*8), 0.);
                    unsigned long long total = 0;
                    for (int l = 0; l < count; ++l)
                    {
                        for (long long j = 0; j < n_lines; ++j)
                        {
                            const bool ok = m->iso_slot[j] >= 0 &&
                                            line_accepted(rule, m->column[0][j], m->order[j]);
                            double * d = (rq.derived != nullptr && l == 0)
                                         ? host_derived.data() + j*8 : nullptr;
                            LineWing & w = host_wing[(size_t)(l*n_lines + j)];
                            const int status = prepare_line(
                                lane.pinned_levels[l], g, m->column[0][j], m->column[1][j],
                                m->column[2][j], m->column[3][j], m->column[4][j], m->column[5][j],
                                m->column[6][j], std::max(m->iso_slot[j], 0), ok, w,
                                host_core[(size_t)(l*n_lines + j)], d);
                            if (status == 1 && w.last >= w.first) total += w.last - w.first + 1;
                        }
                    }
                    engine->timed(kTimePrepare, stream, [&] {
                        HIP_TRY(hipMemcpyAsync(lane.wing.data, host_wing.data(),
                                               host_wing.size()*sizeof(LineWing),
                                               hipMemcpyHostToDevice, stream));
                        HIP_TRY(hipMemcpyAsync(lane.core.data, host_core.data(),
                                               host_core.size()*sizeof(LineCore),
                                               hipMemcpyHostToDevice, stream));
                    });
                    HIP_TRY(hipStreamSynchronize(stream));
                    if (rq.evals != nullptr) *rq.evals += (int64_t)total;
                }
                if (want_k)
                {
                    engine->timed(kTimeSchedule, stream, [&] {
                        dim3 grid((unsigned)((8ll*n_tiles + 255)/256), (unsigned)count);
                        hipLaunchKernelGGL(schedule_kernel, grid, dim3(256), 0, stream,
                                           m->d_column[0].data, (int)n_lines, lane.levels.data, g,
                                           tiling, farfield, lane.schedule.data);
                        HIP_TRY(hipGetLastError());
                    });
                }
            }
            else
            {
                if (inline_levels)
                {
                    std::memcpy(packed.level, lane.pinned_levels, count*sizeof(LevelScalars));
                }
                prepare_blocks = (int)((n_lines + 255)/256);
                const int schedule_blocks = want_k ? (int)((8ll*n_tiles + 255)/256) : 0;
                prologue_blocks = (unsigned)std::max(prepare_blocks + schedule_blocks, 1);
                // (A graphed call launches its prologue with the accumulate kernel, below.)
                if (!graphed) engine->timed(kTimePrepare, stream, [&] {
                    dim3 grid((unsigned)std::max(prepare_blocks + schedule_blocks, 1),
                              (unsigned)count);
                    hipLaunchKernelGGL(prologue_kernel, grid, dim3(256), 0, stream, m->view(),
                                       lane.levels.data, packed, inline_levels ? 1 : 0, g, rule,
                                       tiling, farfield, prepare_blocks, lane.wing.data,
                                       lane.core.data, lane.schedule.data,
                                       rq.derived != nullptr ? lane.derived.data : nullptr,
                                       rq.evals != nullptr ? lane.evals.data : nullptr);
                    HIP_TRY(hipGetLastError());
                });
            }

            if (!want_k) continue;

            // The pedestal pre-pass only needs the per-line scalars: it runs on the side
            // stream next to the accumulate kernel (its serial chain keeps one CU busy).
            // Its run-finding kernels go first: once the accumulate grid owns the chip their
            // wide workgroups would wait for it to drain.
            hipStream_t ped_stream = engine->overlap_pedestal ? lane.side : stream;
            if (with_pedestal)
            {
                if (engine->overlap_pedestal)
                {
                    HIP_TRY(hipEventRecord(lane.prepared, stream));
                    HIP_TRY(hipStreamWaitEvent(lane.side, lane.prepared, 0));
                }
                engine->timed(kTimePedestal, ped_stream, [&] {
                    pedestal_find_runs(lane.pedestal, ped_stream, m->view(), lane.wing.data,
                                       count);
                }, 0);
                if (engine->overlap_pedestal)
                {
                    HIP_TRY(hipEventRecord(lane.runs_found, lane.side));
                }
            }

            // Where the spectra of this pass end up, and where the accumulate kernel writes.
            double * target = out_device ? rq.k + base*stride : lane.staging.data;
            const long long target_stride = out_device ? stride : n_long;
            double * sums = target;
            long long sums_stride = target_stride;
            if (with_pedestal && out_device && add_into)
            {
                sums = lane.raw.data;
                sums_stride = n_long;
            }

            AccumulateArgs args;
            args.wing = lane.wing.data;
            args.core = lane.core.data;
            args.schedule = lane.schedule.data;
            args.levels = lane.levels.data;
            args.items = plan.items.data;
            args.far_series = farfield ? lane.far_series.data : nullptr;
            args.partial = lane.partial.data;
            args.partial_slots = plan.partial_slots;
            args.level_stride = sums_stride;
            args.k = sums;
            args.n_lines = n_lines;
            args.tiling = tiling;
            args.n_tiles = n_tiles;
            args.n = g.n;
            args.v0 = g.v0;
            args.n_per_v = g.n_per_v;
            args.dv = g.dv;
            args.v0_real = (double)g.v0;
            // With a pedestal the kernel stores plain sums; pedestal_apply_kernel finishes.
            args.scale_density = (!with_pedestal && (rq.flags & LBL_SCALE_DENSITY)) ? 1 : 0;
            args.accumulate = (!with_pedestal && out_device && add_into) ? 1 : 0;
            // Can a line outside a tile's core range have its core in the tile?  Only if the
            // window (cut_off + 1 on either side) is not much wider than a core can reach
            // (schedule_tile: core_reach x wavenumber, + the widest pressure shift) plus a tile.
            {
                double reach = 0.;
                for (int l = 0; l < count; ++l)
                {
                    const LevelScalars & lv = lane.pinned_levels[l];
                    const double kk = lv.core_reach;
                    reach = std::max(reach, kk < 0.5 ? kk*(rq.vn + 1.)/(1. - kk) + 2.*lv.shift_max
                                                     : 1.e9);
                }
                const double tile_width = (double)tiling.length*g.dv;
                args.inner_everywhere = (rq.cut_off - 1. <= reach + tile_width + 1.) ? 1 : 0;
            }
            args.ablate = engine->ablate;

            if (with_pedestal && engine->overlap_pedestal && engine->order_runs)
            {
                HIP_TRY(hipStreamWaitEvent(stream, lane.runs_found, 0));
            }
            if (farfield)
            {
                engine->timed(kTimeAccumulate, stream, [&] {
                    hipLaunchKernelGGL(farfield_group_kernel,
                                       dim3((unsigned)n_groups, (unsigned)count, kFarParts),
                                       dim3(256), 0, stream, lane.wing.data, lane.schedule.data,
                                       m->d_column[0].data, lane.levels.data, n_lines, tiling,
                                       n_groups, g.v0, g.n_per_v, g.n, g.dv, lane.group_cuts.data,
                                       lane.far_group.data);
                    HIP_TRY(hipGetLastError());
                    dim3 far_grid((unsigned)(((n_tiles + 7)/8)*8), (unsigned)count);
                    hipLaunchKernelGGL(farfield_kernel, far_grid, dim3(256), 0, stream,
                                       lane.wing.data, lane.schedule.data, lane.group_cuts.data,
                                       n_lines, tiling, lane.far_group.data, n_groups, g.v0,
                                       g.n_per_v, g.n, g.dv, lane.far_series.data);
                    HIP_TRY(hipGetLastError());
                }, 0);
            }
            // A piece = a run of tiles: its accumulate launch (+ the sums of its split tiles),
            // then -- once the pedestal chain has been queued -- the kernel that applies the
            // pedestal to its points and, for a streamed call, the copy of its columns, which
            // runs beside the kernels of the next piece.  One piece unless the call is streamed.
            auto point_range = [&](int piece, long long & q0, long long & q1) {
                long long unused = 0;
                q0 = q1 = 0;
                if (plan.tile_begin[piece + 1] > plan.tile_begin[piece])
                {
                    tile_bounds(tiling, plan.tile_begin[piece], g.n_per_v, g.n, q0, unused);
                    tile_bounds(tiling, plan.tile_begin[piece + 1] - 1, g.n_per_v, g.n, unused, q1);
                    q1 += 1;
                }
            };
            // Where a piece is finished: with a pedestal on the stream the chain runs on (the
            // apply kernels follow it there, while the main stream goes on with the accumulate
            // launches of the later pieces), else on the main stream.
            hipStream_t finish_stream = with_pedestal ? ped_stream : stream;
            auto finish_piece = [&](int piece) {        // (without a pedestal: only the copy)
                long long q0, q1;
                point_range(piece, q0, q1);
                if (q1 <= q0) return;
                if (streamed && q0 < rq.columns)
                {
                    const long long c1 = std::min<long long>(q1, rq.columns);
                    HIP_TRY(hipEventRecord(lane.piece_done[piece], stream));
                    HIP_TRY(hipStreamWaitEvent(engine->copy_stream, lane.piece_done[piece], 0));
                    HIP_TRY(hipMemcpy2DAsync(rq.host + base*rq.host_pitch + q0*8,
                                             (size_t)rq.host_pitch, target + q0,
                                             (size_t)target_stride*8, (size_t)(c1 - q0)*8,
                                             (size_t)count, hipMemcpyDeviceToHost,
                                             engine->copy_stream));
                }
            };
            // All accumulate launches first, back to back; the chain is queued behind the first
            // two of a streamed call (the host waits for the run counts inside), and whatever
            // finishes pieces is queued last.
            if (graphed)
            {
                launch_small_graph(lane, plan, *m, lane.levels.data, packed, g, rule, tiling,
                                   prepare_blocks, prologue_blocks, count, points, args, stream);
            }
            for (int piece = 0; piece < (graphed ? 0 : pieces); ++piece)
            {
                const int item0 = plan.item_begin[piece], item1 = plan.item_begin[piece + 1];
                const int split0 = plan.split_begin[piece], split1 = plan.split_begin[piece + 1];
                engine->timed(kTimeAccumulate, stream, [&] {
                    if (item1 > item0)
                    {
                        // One workgroup per work item, heaviest items first.
                        AccumulateArgs mine = args;
                        mine.items = plan.items.data + item0;
                        dim3 grid((unsigned)(item1 - item0), (unsigned)count);
                        launch_accumulate(points, grid, stream, mine);
                    }
                    if (split1 > split0)
                    {
                        const int units = (split1 - split0)*points;     // (split tile, 64-point row)
                        hipLaunchKernelGGL(combine_kernel,
                                           dim3((unsigned)((units + 3)/4), (unsigned)count),
                                           dim3(256), 0, stream, args, plan.split.data + split0,
                                           split1 - split0, 64*points);
                        HIP_TRY(hipGetLastError());
                    }
                });
                if (!with_pedestal)
                {
                    finish_piece(piece);
                }
                else if (finish_stream != stream)
                {
                    HIP_TRY(hipEventRecord(lane.piece_summed[piece], stream));
                }
            }
            if (with_pedestal)
            {
                engine->timed(kTimePedestal, ped_stream, [&] {
                    pedestal_finish(lane.pedestal, ped_stream, m->view(), lane.wing.data,
                                    lane.core.data, g, count, n_cells, engine->scan_chain != 0,
                                    engine->relax_launches);
                });
                Lane::Finish & f = lane.finish;
                f.pieces = pieces;
                f.count = count;
                f.n_cells = n_cells;
                f.n_per_v = g.n_per_v;
                f.flags = rq.flags;
                // (a run of tiles may be empty -- few tiles, uneven weights -- and must not move
                // its neighbours' bounds: entry i is both the end of run i-1 and the start of run i)
                long long reached = 0;
                for (int piece = 0; piece < pieces; ++piece)
                {
                    long long q0, q1;
                    point_range(piece, q0, q1);
                    if (q1 <= q0)
                    {
                        q0 = q1 = reached;
                    }
                    f.point_begin[piece] = q0;
                    f.point_begin[piece + 1] = q1;
                    reached = q1;
                }
                f.sums = sums;
                f.sums_stride = sums_stride;
                f.target = target;
                f.target_stride = target_stride;
                f.streamed = streamed;
                f.order_writers = alternate && out_device && add_into;
                f.add_into = out_device && add_into;
                f.host = rq.host;
                f.host_pitch = rq.host_pitch;
                f.columns = rq.columns;
                f.base = base;
                // The block's write record: only once, behind the call's last pass.
                const bool last_pass = base + count >= rq.n_levels;
                f.k = (out_device && last_pass) ? rq.k : nullptr;
                f.out_bytes = out_bytes;
                f.finish_stream = finish_stream;
                f.pending = true;
                // Kept back for lbl_finish_deferred only if nothing of this call comes after it:
                // one pass, spectra in device memory, the host not waiting.
                deferred_finish = (rq.flags & LBL_DEFER_FINISH) && (rq.flags & LBL_ASYNC) &&
                                  out_device && alternate && chunk >= rq.n_levels &&
                                  rq.evals == nullptr;
                if (deferred_finish)
                {
                    engine->deferred = &lane;
                }
                else
                {
                    engine->run_finish(lane);
                }
            }

            if (!out_device)
            {
                if (stride == n_long && !(rq.flags & LBL_ACCUMULATE))
                {
                    HIP_TRY(hipMemcpyAsync(rq.k + base*stride, lane.staging.data,
                                           (size_t)count*n_long*8, hipMemcpyDeviceToHost,
                                           stream));
                    // (The last pass's copy is waited for at the end of the call.)
                    if (base + count < rq.n_levels || (rq.flags & LBL_ASYNC))
                    {
                        HIP_TRY(hipStreamSynchronize(stream));
                    }
                }
                else
                {
                    std::vector<double> tmp((size_t)count*n_long);
                    HIP_TRY(hipMemcpyAsync(tmp.data(), lane.staging.data, tmp.size()*8,
                                           hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipStreamSynchronize(stream));
                    for (int l = 0; l < count; ++l)
                    {
                        double * dst = rq.k + (base + l)*stride;
                        const double * src = tmp.data() + (size_t)l*n_long;
                        if (rq.flags & LBL_ACCUMULATE)
                        {
                            for (long long i = 0; i < n_long; ++i) dst[i] += src[i];
                        }
                        else
                        {
                            std::memcpy(dst, src, (size_t)n_long*8);
                        }
                    }
                }
            }
        }

        if (out_device && want_k && !with_pedestal)
        {
            lane.note_write(rq.k, out_bytes, stream);   // (with a pedestal: run_finish does)
        }
        if (rq.evals != nullptr && !(engine->prep == LBL_PREP_HOST))
        {
            unsigned long long total = 0;
            HIP_TRY(hipMemcpyAsync(&total, lane.evals.data, sizeof(total),
                                   hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            *rq.evals = (int64_t)total;
        }
        if (rq.derived != nullptr)
        {
            // Back to the reference's row order.
            std::vector<double> sorted((size_t)(n_lines*8));
            if (engine->prep == LBL_PREP_HOST)
            {
                sorted = host_derived;
            }
            else if (n_lines > 0)
            {
                HIP_TRY(hipMemcpyAsync(sorted.data(), lane.derived.data, sorted.size()*8,
                                       hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
            }
            for (long long j = 0; j < n_lines; ++j)
            {
                std::memcpy(rq.derived + (size_t)m->order[j]*8, sorted.data() + (size_t)j*8, 64);
            }
        }
        if (!(rq.flags & LBL_ASYNC))
        {
            if (wait_for != nullptr)
            {
                *wait_for = engine->take_event();
                HIP_TRY(hipEventRecord(*wait_for, stream));
            }
            else
            {
                HIP_TRY(hipStreamSynchronize(stream));
            }
        }
    }
    catch (const HipFailure & f)
    {
        return fail(engine, LBL_ERROR, f.message);
    }
    catch (const std::bad_alloc &)
    {
        return fail(engine, LBL_ERROR, "host allocation failed.");
    }
    catch (const std::exception & e)
    {
        return fail(engine, LBL_ERROR, e.what());
    }
    return LBL_OK;
}

// compute() under the engine's mutex; the wait of a blocking call outside it.
int locked_compute(lbl_engine * engine, const ComputeRequest & rq)
{
    hipEvent_t last = nullptr;
    int status;
    {
        EngineLock lock(engine->mutex);
        status = compute(engine, rq, &last);
    }
    if (last != nullptr)
    {
        const hipError_t waited = hipEventSynchronize(last);
        EngineLock lock(engine->mutex);
        engine->event_pool.push_back(last);
        if (waited != hipSuccess && status == LBL_OK)
        {
            status = fail(engine, LBL_ERROR, hipGetErrorString(waited));
        }
    }
    return status;
}

// Spins for about `ticks` of the constant-rate clock (100 MHz): something that keeps a queue busy.
__global__ void spin_kernel(long long ticks, int * sink)
{
    const long long begin = wall_clock64();
    int turns = 0;
    while (wall_clock64() - begin < ticks && turns < (1 << 26)) turns += 1;
    if (sink != nullptr && turns < 0) *sink = turns;
}

__global__ void touch_kernel() {}

// Which lanes' main streams share a hardware queue with the copy stream (lbl_engine::
// delivers_badly): the lane's main stream is kept busy for ~40 us, an empty kernel goes to the copy
// stream, and if that only gets through when the lane is done, the two are one queue.
void calibrate_delivery_lanes(lbl_engine * e)
{
    hipEvent_t lane_done, through;
    HIP_TRY(hipEventCreate(&lane_done));
    HIP_TRY(hipEventCreate(&through));
    const long long ticks = 4000;       // 40 us at 100 MHz
    for (int l = 0; l < kLanes; ++l)
    {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, e->lanes[l].main, ticks,
                           (int *)nullptr);
        HIP_TRY(hipEventRecord(lane_done, e->lanes[l].main));
        hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, e->copy_stream);
        HIP_TRY(hipEventRecord(through, e->copy_stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventSynchronize(lane_done));
        HIP_TRY(hipEventSynchronize(through));
        float ahead = 0.f;
        HIP_TRY(hipEventElapsedTime(&ahead, through, lane_done));
        e->delivers_badly[l] = !(ahead > 0.01f);
    }
    (void)hipEventDestroy(lane_done);
    (void)hipEventDestroy(through);
}

}  // namespace

extern "C" {

const char * lbl_version(void)
{
    return "pylbl_amd 0.1 (gfx950)";
}

int lbl_engine_create(int device, lbl_engine ** engine)
{
    if (engine == nullptr) return fail(nullptr, LBL_BAD_ARGUMENT, "engine is NULL.");
    *engine = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    {
        return fail(nullptr, LBL_NO_DEVICE,
                    "no HIP device visible: this engine has no CPU fallback.");
    }
    if (device < 0 || device >= count)
    {
        return fail(nullptr, LBL_NO_DEVICE, "device index out of range.");
    }
    try
    {
        HIP_TRY(hipSetDevice(device));
        std::unique_ptr<lbl_engine> e(new lbl_engine());
        e->device = device;
        for (int i = 0; i < kAllLanes; ++i) e->lanes[i].create(i == kSlotLane);
        e->stream = e->lanes[0].main;
        HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
        calibrate_delivery_lanes(e.get());
        HIP_TRY(hipEventCreateWithFlags(&e->copies_handed_over, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->taken_over, hipEventDisableTiming));
        *engine = e.release();
    }
    catch (const HipFailure & f)
    {
        return fail(nullptr, LBL_NO_DEVICE, f.message);
    }
    return LBL_OK;
}

int lbl_engine_destroy(lbl_engine * engine)
{
    if (engine == nullptr) return LBL_OK;
    {
        // (Whoever destroys a handle has made sure no other thread still uses it; the lock only
        // lets calls that are on their way out leave.)
        EngineLock lock(engine->mutex);
        engine->cancel_deferred();
    }
    (void)hipSetDevice(engine->device);
    engine->drain_lanes();
    for (auto & s : engine->spans)
    {
        (void)hipEventDestroy(s.begin);
        (void)hipEventDestroy(s.end);
    }
    for (auto & e : engine->event_pool) (void)hipEventDestroy(e);
    engine->molecules.clear();
    engine->continua.clear();
    engine->xsecs.clear();
    engine->grids.clear();
    for (auto & lane : engine->lanes) lane.destroy();
    if (engine->copy_stream != nullptr) (void)hipStreamDestroy(engine->copy_stream);
    if (engine->copies_handed_over != nullptr) (void)hipEventDestroy(engine->copies_handed_over);
    if (engine->taken_over != nullptr) (void)hipEventDestroy(engine->taken_over);
    delete engine;
    return LBL_OK;
}

const char * lbl_last_error(const lbl_engine * engine)
{
    if (engine == nullptr) return g_create_error.c_str();
    if (g_thread_error_engine != engine)
    {
        // No failure of this thread on this handle yet: the handle's last message, copied under
        // the lock into this thread's own string (the pointer stays valid until the thread's
        // next failure or next call of this function).
        lbl_engine * e = const_cast<lbl_engine *>(engine);
        EngineLock lock(e->mutex);
        g_thread_error = e->error;
        g_thread_error_engine = engine;
    }
    return g_thread_error.c_str();
}

int lbl_molecule_load(lbl_engine * engine, int64_t n_lines,
                      const double * nu, const double * sw, const double * gamma_air,
                      const double * gamma_self, const double * n_air, const double * elower,
                      const double * delta_air, const int32_t * local_iso_id,
                      const double * mass, int32_t num_iso, int32_t num_t,
                      const double * tips_temperature, const double * tips_data,
                      int32_t * molecule)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    if (molecule == nullptr || n_lines < 0 || n_lines > 0x7fffffff || mass == nullptr ||
        tips_temperature == nullptr || tips_data == nullptr || num_iso < 1 || num_t < 2 ||
        (n_lines > 0 && (nu == nullptr || sw == nullptr || gamma_air == nullptr ||
                         gamma_self == nullptr || n_air == nullptr || elower == nullptr ||
                         delta_air == nullptr || local_iso_id == nullptr)))
    {
        return fail(engine, LBL_BAD_ARGUMENT, "lbl_molecule_load: bad argument.");
    }
    try
    {
        HIP_TRY(hipSetDevice(engine->device));
        std::unique_ptr<Molecule> m(new Molecule());
        m->n_lines = n_lines;
        m->num_iso = num_iso;
        m->num_t = num_t;
        std::memcpy(m->mass, mass, sizeof(m->mass));
        m->tips_t.resize((size_t)num_iso*num_t);
        for (int iso = 0; iso < num_iso; ++iso)
        {
            std::memcpy(m->tips_t.data() + (size_t)iso*num_t, tips_temperature, num_t*8);
        }
        m->tips_q.assign(tips_data, tips_data + (size_t)num_iso*num_t);
        m->nu_row.assign(nu, nu + n_lines);
        m->ascending = std::is_sorted(m->nu_row.begin(), m->nu_row.end());
        m->order.resize((size_t)n_lines);
        std::iota(m->order.begin(), m->order.end(), 0);
        if (!m->ascending)
        {
            std::stable_sort(m->order.begin(), m->order.end(),
                             [&](int a, int b) { return nu[a] < nu[b]; });
        }
        const double * source[7] = {nu, sw, gamma_air, gamma_self, n_air, elower, delta_air};
        for (int c = 0; c < 7; ++c)
        {
            m->column[c].resize((size_t)n_lines);
            for (long long j = 0; j < n_lines; ++j) m->column[c][j] = source[c][m->order[j]];
        }
        m->iso_slot.resize((size_t)n_lines);
        for (long long j = 0; j < n_lines; ++j)
        {
            if (!std::isfinite(m->column[0][j]))
            {
                return fail(engine, LBL_BAD_ARGUMENT, "non-finite line position.");
            }
            int iso = local_iso_id[m->order[j]];
            if (iso == 0) iso = 10;                         // spectral_database.c:173-177
            const int slot = iso - 1;
            if (slot < 0 || slot >= kMassSlots || slot >= num_iso || !(m->mass[slot] > 0.))
            {
                m->bad_rows.push_back(Molecule::BadRow{m->order[j], m->column[0][j],
                                                       local_iso_id[m->order[j]]});
                m->iso_slot[j] = -1;        // never evaluated (prepare_kernel / host prep)
                continue;
            }
            m->iso_slot[j] = slot;
            m->used_slots |= 1u << slot;
            m->max_abs_delta = std::max(m->max_abs_delta, fabs(m->column[6][j]));
            m->min_gamma_air = std::min(m->min_gamma_air, m->column[2][j]);
            m->min_gamma_self = std::min(m->min_gamma_self, m->column[3][j]);
            m->min_n_air = std::min(m->min_n_air, m->column[4][j]);
            m->max_n_air = std::max(m->max_n_air, m->column[4][j]);
        }
        hipStream_t stream = engine->stream;
        for (int c = 0; c < 7; ++c) m->d_column[c].upload(m->column[c].data(), n_lines, stream);
        m->d_iso_slot.upload(m->iso_slot.data(), n_lines, stream);
        m->d_row.upload(m->order.data(), n_lines, stream);
        std::vector<int> inverse((size_t)n_lines);
        for (long long j = 0; j < n_lines; ++j) inverse[m->order[j]] = (int)j;
        m->d_sorted_of_row.upload(inverse.data(), n_lines, stream);
        HIP_TRY(hipStreamSynchronize(stream));
        // Reuse a freed slot if there is one.
        size_t slot = engine->molecules.size();
        for (size_t i = 0; i < engine->molecules.size(); ++i)
        {
            if (!engine->molecules[i]) { slot = i; break; }
        }
        if (slot == engine->molecules.size()) engine->molecules.emplace_back();
        engine->molecules[slot] = std::move(m);
        *molecule = (int32_t)slot;
    }
    catch (const HipFailure & f)
    {
        return fail(engine, LBL_ERROR, f.message);
    }
    catch (const std::bad_alloc &)
    {
        return fail(engine, LBL_ERROR, "host allocation failed.");
    }
    return LBL_OK;
}

int lbl_molecule_free(lbl_engine * engine, int32_t molecule)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    if (find_molecule(engine, molecule) == nullptr)
    {
        return fail(engine, LBL_BAD_ARGUMENT, "unknown molecule handle.");
    }
    (void)hipSetDevice(engine->device);
    engine->drain_lanes();
    engine->molecules[molecule].reset();
    return LBL_OK;
}

int lbl_compute(lbl_engine * engine, int32_t molecule, int32_t n_levels,
                const double * temperature, const double * pressure, const double * vmr,
                int32_t v0, int32_t vn, int32_t n_per_v, int32_t cut_off,
                int32_t remove_pedestal, int32_t range_policy, int32_t flags,
                double * k, int64_t level_stride, int64_t * evals)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    if (k == nullptr) return fail(engine, LBL_BAD_ARGUMENT, "k is NULL.");
    if (evals != nullptr) *evals = 0;
    ComputeRequest rq{molecule, n_levels, temperature, pressure, vmr, v0, vn, n_per_v, cut_off,
                      remove_pedestal, range_policy, flags, k, level_stride, evals, nullptr};
    return locked_compute(engine, rq);
}

int lbl_compute_streamed(lbl_engine * engine, int32_t molecule, int32_t n_levels,
                         const double * temperature, const double * pressure, const double * vmr,
                         int32_t v0, int32_t vn, int32_t n_per_v, int32_t cut_off,
                         int32_t remove_pedestal, int32_t range_policy, int32_t flags,
                         double * k, int64_t level_stride, void * host, int64_t host_pitch,
                         int64_t columns, int32_t pieces)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    if (k == nullptr || host == nullptr) return fail(engine, LBL_BAD_ARGUMENT, "k or host is NULL.");
    if (!(flags & LBL_OUT_DEVICE))
    {
        return fail(engine, LBL_BAD_ARGUMENT, "lbl_compute_streamed needs LBL_OUT_DEVICE.");
    }
    if (columns < 0 || host_pitch < columns*8 || pieces < 1)
    {
        return fail(engine, LBL_BAD_ARGUMENT, "need columns >= 0, host_pitch >= 8*columns and "
                                              "pieces >= 1.");
    }
    ComputeRequest rq{molecule, n_levels, temperature, pressure, vmr, v0, vn, n_per_v, cut_off,
                      remove_pedestal, range_policy, flags, k, level_stride, nullptr, nullptr};
    rq.host = static_cast<char *>(host);
    rq.host_pitch = host_pitch;
    rq.columns = columns;
    rq.pieces = pieces;
    const int status = compute(engine, rq);
    if (status == LBL_OK && !(flags & LBL_ASYNC))
    {
        (void)hipStreamSynchronize(engine->copy_stream);
    }
    return status;
}

int lbl_line_scalars(lbl_engine * engine, int32_t molecule, double temperature,
                     double pressure, double vmr, int32_t v0, int32_t vn, int32_t n_per_v,
                     int32_t cut_off, int32_t remove_pedestal, int32_t range_policy,
                     double * derived)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    if (derived == nullptr) return fail(engine, LBL_BAD_ARGUMENT, "derived is NULL.");
    (void)remove_pedestal;
    ComputeRequest rq{molecule, 1, &temperature, &pressure, &vmr, v0, vn, n_per_v, cut_off,
                      0, range_policy, 0, nullptr, 0, nullptr, derived};
    return compute(engine, rq);
}

int lbl_deferred(const lbl_engine * engine)
{
    if (engine == nullptr) return 0;
    EngineLock lock(const_cast<lbl_engine *>(engine)->mutex);
    return (engine->deferred != nullptr && engine->deferred->finish.pending) ? 1 : 0;
}

int lbl_cancel_deferred(lbl_engine * engine)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    engine->cancel_deferred();
    return LBL_OK;
}

int lbl_finish_deferred(lbl_engine * engine)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    try
    {
        HIP_TRY(hipSetDevice(engine->device));
        engine->finish_deferred();
    }
    catch (const HipFailure & f)
    {
        return fail(engine, LBL_ERROR, f.message);
    }
    return LBL_OK;
}

int lbl_synchronize(lbl_engine * engine)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    (void)hipSetDevice(engine->device);
    // A call still kept back (LBL_DEFER_FINISH) is finished first: nothing stays unapplied.
    const int finished = lbl_finish_deferred(engine);
    if (finished != LBL_OK) return finished;
    for (auto & lane : engine->lanes)
    {
        hipError_t status = hipStreamSynchronize(lane.main);
        if (status == hipSuccess) status = hipStreamSynchronize(lane.side);
        if (status != hipSuccess) return fail(engine, LBL_ERROR, hipGetErrorString(status));
    }
    const hipError_t status = hipStreamSynchronize(engine->copy_stream);
    if (status != hipSuccess) return fail(engine, LBL_ERROR, hipGetErrorString(status));
    return LBL_OK;
}

int lbl_set_option(lbl_engine * engine, const char * name, int64_t value)
{
    if (engine == nullptr || name == nullptr) return LBL_BAD_ARGUMENT;
    const std::string key(name);
    if (key == "prep" && (value == LBL_PREP_DEVICE || value == LBL_PREP_HOST))
    {
        engine->prep = (int)value;
    }
    else if (key == "points_per_lane" &&
             (value == 0 || value == 1 || value == 2 || value == 4 || value == 8))
    {
        engine->points_per_lane = (int)value;
    }
    else if (key == "timing" && value >= 0 && value <= 2)
    {
        engine->timing = (int)value;
    }
    else if (key == "interp_shape" && value >= 0 && value < 100)
    {
        engine->interp_shape = (int)value;
    }
    else if (key == "relax_launches" && (value == 0 || (value >= 2 && value <= 7)))
    {
        engine->relax_launches = (int)value;
    }
    else if (key == "scan_chain" && (value == 0 || value == 1))
    {
        engine->scan_chain = (int)value;
    }
    else if (key == "order_runs" && (value == 0 || value == 1))
    {
        engine->order_runs = (int)value;
    }
    else if (key == "skip_delivery_lanes" && (value == 0 || value == 1))
    {
        engine->skip_delivery_lanes = (int)value;
    }
    else if (key == "item_floor" && value >= 0 && value <= 65536)
    {
        engine->item_floor = (int)value;
    }
    else if (key == "graphs" && (value == 0 || value == 1))
    {
        engine->graphs = (int)value;
    }
    else if (key == "small_points" && value >= 0)
    {
        engine->small_points = value;
    }
    else if (key == "lanes" && (value == 0 || (value >= 2 && value <= kLanes)))
    {
        engine->lanes_in_use = (int)value;
    }
    else if (key == "farfield" && (value == 0 || value == 1))
    {
        engine->farfield = (int)value;
    }
    else if (key == "overlap_pedestal" && (value == 0 || value == 1))
    {
        engine->overlap_pedestal = (int)value;
    }
    else if (key == "aligned_tiles" && (value == 0 || value == 1))
    {
        engine->aligned_tiles = (int)value;
    }
    else if (key == "ablate" && value >= 0 && value <= 127)
    {
        engine->ablate = (int)value;    // timing diagnostics only: results are wrong when set
    }
    else if (key == "workspace_bytes" && value >= (1 << 20))
    {
        engine->workspace_bytes = value;
    }
    else
    {
        return fail(engine, LBL_BAD_ARGUMENT, "unknown option or value: " + key);
    }
    return LBL_OK;
}

int lbl_timing(lbl_engine * engine, double ms[8], int64_t launches[8], int32_t reset)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    try
    {
        HIP_TRY(hipSetDevice(engine->device));
        engine->drain_lanes();
        engine->drain_spans();
    }
    catch (const HipFailure & f)
    {
        return fail(engine, LBL_ERROR, f.message);
    }
    for (int i = 0; i < kTimeKinds; ++i)
    {
        if (ms != nullptr) ms[i] = engine->time_ms[i];
        if (launches != nullptr) launches[i] = engine->launches[i];
        if (reset)
        {
            engine->time_ms[i] = 0.;
            engine->launches[i] = 0;
        }
    }
    return LBL_OK;
}

void * lbl_stream(lbl_engine * engine)
{
    return engine != nullptr ? (void *)engine->stream : nullptr;
}

// The two halves of sharing HBM blocks with another HIP user of the device (e.g. the library
// that runs the RCCL exchange of pylbl_amd/distributed.py) without stopping the host: each is a
// handful of event records and stream waits.
int lbl_order_stream_after_engine(lbl_engine * engine, void * stream)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    try
    {
        HIP_TRY(hipSetDevice(engine->device));
        // "Everything queued so far" includes what a call kept back (LBL_DEFER_FINISH): the
        // caller's stream is about to read the block.
        engine->finish_deferred();
        hipStream_t theirs = reinterpret_cast<hipStream_t>(stream);
        // Side streams end in an event their lane's main stream waits for (pedestal_done), so
        // the main streams and the copy stream stand for everything the engine has queued.
        for (auto & lane : engine->lanes)
        {
            HIP_TRY(hipEventRecord(lane.handed_over, lane.main));
            HIP_TRY(hipStreamWaitEvent(theirs, lane.handed_over, 0));
        }
        HIP_TRY(hipEventRecord(engine->copies_handed_over, engine->copy_stream));
        HIP_TRY(hipStreamWaitEvent(theirs, engine->copies_handed_over, 0));
    }
    catch (const HipFailure & f)
    {
        return fail(engine, LBL_ERROR, f.message);
    }
    return LBL_OK;
}

int lbl_order_engine_after_stream(lbl_engine * engine, void * stream)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    try
    {
        HIP_TRY(hipSetDevice(engine->device));
        HIP_TRY(hipEventRecord(engine->taken_over, reinterpret_cast<hipStream_t>(stream)));
        for (auto & lane : engine->lanes)
        {
            HIP_TRY(hipStreamWaitEvent(lane.main, engine->taken_over, 0));
            HIP_TRY(hipStreamWaitEvent(lane.side, engine->taken_over, 0));
        }
        HIP_TRY(hipStreamWaitEvent(engine->copy_stream, engine->taken_over, 0));
    }
    catch (const HipFailure & f)
    {
        return fail(engine, LBL_ERROR, f.message);
    }
    return LBL_OK;
}

int lbl_device_alloc(lbl_engine * engine, int64_t bytes, void ** pointer)
{
    if (engine == nullptr || pointer == nullptr || bytes < 0) return LBL_BAD_ARGUMENT;
    (void)hipSetDevice(engine->device);
    hipError_t status = hipMalloc(pointer, (size_t)std::max<int64_t>(bytes, 8));
    if (status != hipSuccess) return fail(engine, LBL_ERROR, hipGetErrorString(status));
    return LBL_OK;
}

int lbl_device_free(lbl_engine * engine, void * pointer)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    (void)hipSetDevice(engine->device);
    // A call kept back (LBL_DEFER_FINISH) may still have this block to write: it is finished
    // first, like in lbl_synchronize -- never left to run into freed memory.
    try { engine->finish_deferred(); } catch (const HipFailure &) { engine->cancel_deferred(); }
    engine->drain_lanes();
    hipError_t status = hipFree(pointer);
    if (status != hipSuccess) return fail(engine, LBL_ERROR, hipGetErrorString(status));
    return LBL_OK;
}

int lbl_copy_to_host(lbl_engine * engine, void * host, const void * device, int64_t bytes)
{
    if (engine == nullptr || host == nullptr || device == nullptr || bytes < 0)
    {
        return LBL_BAD_ARGUMENT;
    }
    EngineLock lock(engine->mutex);
    (void)hipSetDevice(engine->device);
    try { engine->finish_deferred(); }
    catch (const HipFailure & f) { return fail(engine, LBL_ERROR, f.message); }
    // The memory may have been written on any lane (asynchronous calls with a pedestal rotate
    // over them): wait for all of them, not only for lane 0.
    hipError_t status = hipSuccess;
    for (auto & lane : engine->lanes)
    {
        if (status == hipSuccess) status = hipStreamSynchronize(lane.main);
    }
    if (status == hipSuccess)
    {
        status = hipMemcpyAsync(host, device, (size_t)bytes, hipMemcpyDeviceToHost,
                                engine->stream);
    }
    if (status == hipSuccess) status = hipStreamSynchronize(engine->stream);
    if (status != hipSuccess) return fail(engine, LBL_ERROR, hipGetErrorString(status));
    return LBL_OK;
}

int lbl_fill_zero(lbl_engine * engine, double * k, int32_t n_levels, int64_t n,
                  int64_t level_stride, int32_t flags)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    if (k == nullptr || n_levels < 0 || n < 0 || (level_stride != 0 && level_stride < n))
    {
        return fail(engine, LBL_BAD_ARGUMENT, "lbl_fill_zero: bad argument.");
    }
    const int64_t stride = level_stride > 0 ? level_stride : n;
    if (n_levels == 0 || n == 0) return LBL_OK;
    if (!(flags & LBL_OUT_DEVICE))
    {
        for (int32_t l = 0; l < n_levels; ++l) std::memset(k + l*stride, 0, (size_t)n*8);
        return LBL_OK;
    }
    try
    {
        HIP_TRY(hipSetDevice(engine->device));
        // Ordered like a plain compute call: after everything queued on the other lanes --
        // by events when the caller does not wait either, so that the host keeps queueing.
        if (flags & LBL_ASYNC)
        {
            engine->join_lanes(engine->stream);
        }
        else
        {
            for (int i = 1; i < kAllLanes; ++i) engine->lanes[i].drain();
        }
        HIP_TRY(hipMemset2DAsync(k, (size_t)stride*8, 0, (size_t)n*8, (size_t)n_levels,
                                 engine->stream));
        engine->lanes[0].note_write(k, ((long long)(n_levels - 1)*stride + n)*8, engine->stream);
        if (!(flags & LBL_ASYNC)) HIP_TRY(hipStreamSynchronize(engine->stream));
    }
    catch (const HipFailure & f)
    {
        return fail(engine, LBL_ERROR, f.message);
    }
    return LBL_OK;
}

int lbl_copy_rows_to_host(lbl_engine * engine, void * host, int64_t host_pitch,
                          const void * device, int64_t device_pitch, int64_t row_bytes,
                          int64_t rows, int32_t flags)
{
    if (engine == nullptr || host == nullptr || device == nullptr || row_bytes < 0 || rows < 0 ||
        host_pitch < row_bytes || device_pitch < row_bytes)
    {
        return LBL_BAD_ARGUMENT;
    }
    if (rows == 0 || row_bytes == 0) return LBL_OK;
    EngineLock lock(engine->mutex);
    try
    {
        HIP_TRY(hipSetDevice(engine->device));
        engine->finish_deferred();      // the rows may be what a call kept back still has to write
        // The rows may have been written on any lane: the copy stream waits for what each of
        // them holds now, then copies beside whatever is queued afterwards.
        for (auto & lane : engine->lanes)
        {
            HIP_TRY(hipEventRecord(lane.queued, lane.main));
            HIP_TRY(hipStreamWaitEvent(engine->copy_stream, lane.queued, 0));
        }
        HIP_TRY(hipMemcpy2DAsync(host, (size_t)host_pitch, device, (size_t)device_pitch,
                                 (size_t)row_bytes, (size_t)rows, hipMemcpyDeviceToHost,
                                 engine->copy_stream));
        if (!(flags & LBL_ASYNC)) HIP_TRY(hipStreamSynchronize(engine->copy_stream));
    }
    catch (const HipFailure & f)
    {
        return fail(engine, LBL_ERROR, f.message);
    }
    return LBL_OK;
}

int lbl_host_alloc(lbl_engine * engine, int64_t bytes, void ** pointer)
{
    if (engine == nullptr || pointer == nullptr || bytes < 0) return LBL_BAD_ARGUMENT;
    (void)hipSetDevice(engine->device);
    hipError_t status = hipHostMalloc(pointer, (size_t)std::max<int64_t>(bytes, 8),
                                      hipHostMallocDefault);
    if (status != hipSuccess) return fail(engine, LBL_ERROR, hipGetErrorString(status));
    return LBL_OK;
}

int lbl_host_free(lbl_engine * engine, void * pointer)
{
    if (engine == nullptr) return LBL_BAD_ARGUMENT;
    EngineLock lock(engine->mutex);
    (void)hipSetDevice(engine->device);
    if (engine->deferred != nullptr && engine->deferred->finish.pending)
    {
        // The copies of a streamed call kept back may target this memory.
        try { engine->finish_deferred(); } catch (const HipFailure &) { engine->cancel_deferred(); }
        engine->drain_lanes();
    }
    if (engine->copy_stream != nullptr) (void)hipStreamSynchronize(engine->copy_stream);
    hipError_t status = hipHostFree(pointer);
    if (status != hipSuccess) return fail(engine, LBL_ERROR, hipGetErrorString(status));
    return LBL_OK;
}

}  // extern "C"

#include "continuum_entry.inc"
#include "xsec_entry.inc"
#include "sqlite_entry.inc"
