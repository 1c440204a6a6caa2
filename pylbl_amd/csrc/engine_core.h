// State of the engine: device buffers, a molecule's resident line table and work-item plans,
// the lanes (stream pair + workspace) asynchronous calls rotate over, and struct lbl_engine
// itself -- options, timing spans, the write records calls are ordered by and the part of a
// call that LBL_DEFER_FINISH keeps back.  Included by engine.hip only (one translation unit).
#pragma once

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
    // cell_first of LineTableView (tile_schedule.h): where the sorted table reaches every
    // 1/cell_scale cm-1.
    DeviceBuffer<int> d_cell_first;
    double cell_base = 0., cell_scale = 0.;
    int cell_entries = 0;

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
        v.cell_first = cell_entries > 0 ? d_cell_first.data : nullptr;
        v.cell_base = cell_base; v.cell_scale = cell_scale; v.cell_entries = cell_entries;
        return v;
    }
};

// (kTimeSchedule: schedule_kernel alone, which only host-side prep launches; the far-field series
// kernels are booked there too -- with device prep the index would otherwise stay empty)
enum { kTimePrepare = 0, kTimeSchedule = 1, kTimeAccumulate = 2, kTimePedestal = 3,
       kTimeBandSpectra = 4, kTimeContinuum = 5, kTimeXsecModel = 6, kTimeXsec = 7,
       kTimeKinds = 8 };

// One in-flight compute call: its own pair of streams and its own workspace, so that
// several molecules can be in the pipeline at once (the serial pedestal chain of one
// overlaps the accumulate kernels of the others, and its own).
struct Lane
{
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
        int n_per_v = 0, cut_off = 0;
        const double * sums = nullptr;
        long long sums_stride = 0;
        double * target = nullptr;
        long long target_stride = 0;
        bool streamed = false, order_writers = false, add_into = false;
        char * host = nullptr;
        long long host_pitch = 0, columns = 0, base = 0;
        double * k = nullptr;          // the block, for its write record: set by the call's last pass only
        const double * block = nullptr; // the block, for ordering behind its earlier writers: every pass
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
    // wavenumber[i] == start + (double)i*step for every i, checked element by element at load
    // (what numpy.arange produces): the interpolation kernels then form x in registers.
    bool arithmetic = false;
    double start = 0., step = 0.;
    DeviceBuffer<double> wavenumber;

    GridForm form() const
    {
        return GridForm{wavenumber.data, arithmetic ? 1 : 0, start, step};
    }
};

// Several continua evaluated in one pass (continuum.h, group kernels): the bands of all of them
// in one list on the device, one workspace of coarse spectra, one block of level scalars
// [continuum][level].  Kept per list of handles; dropped when one of them is freed.
struct ContinuumGroup : LevelFeed<ContinuumLevel>
{
    std::vector<int32_t> members;
    DeviceBuffer<GroupBand> bands;
    int n_bands = 0, level_points = 0, widest = 0;
    DeviceBuffer<double> coarse, slopes;    // [levels][level_points]
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
    std::vector<std::unique_ptr<ContinuumGroup>> groups;
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
    int scan_chain = 1;             // pedestal chain by relaxation (pedestal.h), serial chain behind it
    int relax_launches = 0;         // relaxation launches before the serial chain (2 ... 7; 0: by the table)
    int lanes_in_use = 0;           // lanes the asynchronous calls rotate over; 0: by kind of call
    long long small_points = 1ll << 20;    // grids (points x levels) up to this size count as small
    int overlap_plain = 1;          // plain asynchronous calls on larger grids take turns on two lanes too
    int skip_delivery_lanes = 1;    // delivering calls avoid lanes that share the copy stream's queue

    // Timing.
    struct Span { hipEvent_t begin, end; int kind, counts; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    double time_ms[kTimeKinds] = {};
    double busy_ms[kTimeKinds] = {};    // time during which AT LEAST ONE timed span of the kind ran
    double busy_reach[kTimeKinds] = {}; // where the union of the kind's spans ends so far [ms from epoch]
    hipEvent_t epoch = nullptr;         // origin of the spans' positions (reset_epoch)
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
        if (!timing || (timing == 2 && kind != kTimeAccumulate && kind != kTimeSchedule))
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

    // Sums the spans' durations per kind (time_ms) and, because calls on different lanes run side
    // by side, also the length of the UNION of the spans of a kind on the device's clock (busy_ms:
    // two launches that overlap count once).  Positions are milliseconds (double) from `epoch`, an
    // event recorded on the engine's first stream when timing was last reset: it precedes every
    // span on the device's clock, whichever lane a span ran on, and it is the same origin for every
    // batch, so that the reach of the union carries over from one batch to the next (a launch that
    // straddles a batch boundary is counted once).  A span whose position cannot be read (an event
    // the runtime places before the epoch) is clamped to the epoch.
    void drain_spans()
    {
        if (spans.empty()) return;
        std::vector<std::pair<double, double>> placed[kTimeKinds];
        for (auto & s : spans) HIP_TRY(hipEventSynchronize(s.end));
        if (epoch != nullptr) HIP_TRY(hipEventSynchronize(epoch));
        for (auto & s : spans)
        {
            float ms = 0.f, from = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, s.begin, s.end));
            time_ms[s.kind] += ms;
            launches[s.kind] += s.counts;
            if (epoch == nullptr || hipEventElapsedTime(&from, epoch, s.begin) != hipSuccess ||
                !(from > 0.f))
            {
                (void)hipGetLastError();
                from = 0.f;
            }
            placed[s.kind].push_back({(double)from, (double)from + (double)ms});
        }
        for (int kind = 0; kind < kTimeKinds; ++kind)
        {
            auto & list = placed[kind];
            std::sort(list.begin(), list.end());
            double & reach = busy_reach[kind];      // end of the union so far, carried over batches
            for (const auto & interval : list)
            {
                if (interval.first > reach)
                {
                    busy_ms[kind] += interval.second - interval.first;
                    reach = interval.second;
                }
                else if (interval.second > reach)
                {
                    busy_ms[kind] += interval.second - reach;
                    reach = interval.second;
                }
            }
        }
        for (auto & s : spans)
        {
            event_pool.push_back(s.begin);
            event_pool.push_back(s.end);
        }
        spans.clear();
    }

    // A fresh origin for the spans' positions (lbl_timing with reset, option "timing" switched on).
    void reset_epoch()
    {
        if (epoch == nullptr) epoch = take_event();
        HIP_TRY(hipEventRecord(epoch, stream));
        for (int kind = 0; kind < kTimeKinds; ++kind) busy_reach[kind] = 0.;
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

    // Memory handed out by lbl_device_alloc / lbl_host_alloc: base address and size, so that a
    // free knows whether a call kept back (LBL_DEFER_FINISH) still has that memory to write.
    struct Block { const char * begin; long long bytes; };
    std::vector<Block> device_blocks, host_blocks;

    static long long forget(std::vector<Block> & blocks, const void * pointer)
    {
        for (size_t i = 0; i < blocks.size(); ++i)
        {
            if (blocks[i].begin == pointer)
            {
                const long long bytes = blocks[i].bytes;
                blocks.erase(blocks.begin() + i);
                return bytes;
            }
        }
        return -1;      // not one of ours (or freed twice): the caller assumes the worst
    }

    // Does the call kept back write into [begin, begin + bytes)?  bytes < 0: size unknown, yes.
    bool deferred_touches(const void * begin, long long bytes) const
    {
        if (deferred == nullptr || !deferred->finish.pending) return false;
        if (bytes < 0) return true;
        const Lane::Finish & f = deferred->finish;
        const char * b = reinterpret_cast<const char *>(begin);
        const char * e = b + bytes;
        auto meets = [&](const void * p, long long n) {
            const char * q = reinterpret_cast<const char *>(p);
            return p != nullptr && n > 0 && q < e && b < q + n;
        };
        const long long rows = ((long long)(f.count - 1)*f.target_stride + f.point_begin[f.pieces])*8;
        return meets(f.k, f.out_bytes) || meets(f.target, rows) ||
               (f.streamed && meets(f.host + f.base*f.host_pitch,
                                    (long long)(f.count - 1)*f.host_pitch + f.columns*8));
    }

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
            // (every pass of a call in several passes: the FIRST pass's kernels are the ones that
            // must not add into rows another lane's earlier call is still adding into)
            order_after_writers(f.finish_stream, f.block, f.out_bytes, &lane);
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
            constexpr int per_block = 256*kApplyPoints;
            dim3 grid((unsigned)((q1 - q0 + per_block - 1)/per_block), (unsigned)f.count);
            hipLaunchKernelGGL(pedestal_apply_kernel, grid, dim3(256),
                               pedestal_apply_lds_bytes(f.cut_off, f.n_per_v), f.finish_stream,
                               f.sums, f.sums_stride, f.target, f.target_stride,
                               lane.pedestal.bin_sum.data, lane.levels.data, (int)q0, (int)q1,
                               f.n_per_v, f.n_cells + 2*f.cut_off + 3, f.cut_off,
                               pedestal_apply_span(f.n_per_v),
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

}  // namespace
