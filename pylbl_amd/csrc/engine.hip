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

#include "engine_core.h"
#include "lanes_plans.inc"
#include "compute_stages.inc"
#include "compute_call.inc"

// (calibrate_delivery_lanes, used by lbl_engine_create)
#include "delivery.inc"

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
    if (engine->epoch != nullptr) (void)hipEventDestroy(engine->epoch);
    engine->molecules.clear();
    engine->groups.clear();
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
        // Where the sorted table reaches every eighth of a wavenumber (coarser for tables that
        // span more than 131 072 cm-1): the bracket every search of the schedule starts from.
        std::vector<int> cell_first;
        if (n_lines > 0)
        {
            const std::vector<double> & sorted_nu = m->column[0];
            const double base = std::floor(sorted_nu.front());
            const double span = std::ceil(sorted_nu.back()) - base + 1.;
            if (span >= 1. && span < 1.e12)
            {
                const double scale = std::min(8., std::floor(1048576./span*8.)/8.);
                if (scale > 0.)
                {
                    const long long entries = (long long)(span*scale) + 1;
                    cell_first.resize((size_t)entries);
                    long long j = 0;
                    for (long long i = 0; i < entries; ++i)
                    {
                        const double edge = base + (double)i/scale;
                        while (j < n_lines && sorted_nu[(size_t)j] < edge) ++j;
                        cell_first[(size_t)i] = (int)j;
                    }
                    m->cell_base = base;
                    m->cell_scale = scale;
                    m->cell_entries = (int)entries;
                    m->d_cell_first.upload(cell_first.data(), cell_first.size(), stream);
                }
            }
        }
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
    // (compute() reads the options under the same lock: none changes in the middle of a call)
    EngineLock lock(engine->mutex);
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
        if (value != 0 && engine->timing == 0)
        {
            // A fresh origin for the positions of the spans to come, behind everything queued so far.
            try
            {
                HIP_TRY(hipSetDevice(engine->device));
                engine->drain_lanes();
                engine->reset_epoch();
            }
            catch (const HipFailure & f)
            {
                return fail(engine, LBL_ERROR, f.message);
            }
        }
        engine->timing = (int)value;
    }
    else if (key == "relax_launches" && (value == 0 || (value >= 2 && value <= 7)))
    {
        engine->relax_launches = (int)value;
    }
    else if (key == "scan_chain" && (value == 0 || value == 1))
    {
        engine->scan_chain = (int)value;
    }
    else if (key == "skip_delivery_lanes" && (value == 0 || value == 1))
    {
        engine->skip_delivery_lanes = (int)value;
    }
    else if (key == "overlap_plain" && (value == 0 || value == 1))
    {
        engine->overlap_plain = (int)value;
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
#ifdef LBL_ABLATE
    else if (key == "ablate" && value >= 0 && value <= 127)
    {
        engine->ablate = (int)value;    // diagnostics build only: results are wrong when set
    }
#endif
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
            engine->busy_ms[i] = 0.;
        }
    }
    if (reset && engine->timing != 0)
    {
        try
        {
            engine->reset_epoch();      // (every stream has just been drained)
        }
        catch (const HipFailure & f)
        {
            return fail(engine, LBL_ERROR, f.message);
        }
    }
    return LBL_OK;
}

int lbl_timing_busy(lbl_engine * engine, double busy_ms[8])
{
    if (engine == nullptr || busy_ms == nullptr) return LBL_BAD_ARGUMENT;
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
    for (int i = 0; i < kTimeKinds; ++i) busy_ms[i] = engine->busy_ms[i];
    return LBL_OK;
}

}  // extern "C"

#include "continuum_entry.inc"
#include "xsec_entry.inc"
#include "sqlite_entry.inc"
