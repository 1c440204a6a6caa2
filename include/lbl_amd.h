/*
 * lbl_amd.h -- C ABI of the MI355X (gfx950) molecular-lines engine.
 *
 * This is the drop-in boundary for the lines backend of GRIPS-code/pyLBL: the one
 * native entry point the reference binds over ctypes is
 *
 *     int absorption(double pressure, double temperature, double volume_mixing_ratio,
 *                    int v0, int vn, int n_per_v, double *k, char *database,
 *                    char *formula, int cut_off, int remove_pedestal);
 *                                        (pyLBL/c_lib/absorption.c:19-30, bound at
 *                                         pyLBL/c_lib/gas_optics.py:68-91)
 *
 * That call re-opens the SQLite file, re-reads every transition of the molecule and
 * computes ONE (level, molecule) spectrum on one CPU thread.  The replacement splits it
 * into (1) a one-time upload of a molecule's line table to HBM and (2) a batched compute
 * call over many atmospheric levels, and keeps a same-signature compatibility entry.
 *
 * Conventions: plain C types only; every function returns LBL_OK (0) or a non-zero
 * status and never throws; the message for the last failure on a handle is available
 * from lbl_last_error().  All arrays are caller-owned; "host" pointers are ordinary
 * process memory, "device" pointers are HIP device allocations on the engine's GPU.
 *
 * Threads: every entry point may be called from any number of threads on the SAME handle at
 * once, as the reference's absorption() may (no globals or statics, absorption.c:19-99; ctypes
 * releases the GIL around it, gas_optics.py:79-91).  A handle serialises the host side of its
 * calls -- queueing work is microseconds -- with a mutex of its own; a blocking call waits for
 * its result after it has released that mutex, so other threads' calls queue up behind it on
 * the GPU meanwhile, and each call returns exactly what it would have returned alone.
 * lbl_last_error() returns the calling thread's own last message.  What stays with the caller:
 * output blocks that several asynchronous calls share (LBL_ACCUMULATE) see those calls in the
 * order they were made, so threads that add into ONE block without ordering themselves get the
 * sum in a nondeterministic order of additions; and there is one deferred call per handle
 * (LBL_DEFER_FINISH) -- a pipeline that defers should not share its handle with other threads
 * while it does (pylbl_amd.Spectroscopy holds a lock per engine for that).  Distinct handles
 * are independent.
 */
#ifndef LBL_AMD_H_
#define LBL_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBL_OK              0
#define LBL_ERROR           1   /* generic failure (message in lbl_last_error)            */
#define LBL_BAD_ARGUMENT    2
#define LBL_NO_DEVICE       3   /* no usable gfx950 device / HIP runtime failure           */
#define LBL_OUT_OF_RANGE    4   /* temperature outside the TIPS table, iso id without data  */
/* lbl_table_read / lbl_molecule_load_sqlite: the first thing the file lacks for the molecule, in
 * the order the reference looks things up (absorption.c:50-73) */
#define LBL_TABLE_OPEN_FAILED       10  /* not an SQLite file this process can open            */
#define LBL_TABLE_NO_ALIAS          11  /* spectral_database.c:152-156: the reference's rc 1    */
#define LBL_TABLE_NO_TIPS           12  /* absorption.c:53-59: the reference returns zeros      */
#define LBL_TABLE_NOT_RECTANGULAR   13  /* spectral_database.c:85-90                            */
#define LBL_TABLE_NO_ISOTOPOLOGUES  14
#define LBL_TABLE_NO_TRANSITIONS    15

/* Which rows of the table take part (reference: pyLBL/c_lib/absorption.c:80-83). */
#define LBL_RANGE_REFERENCE 0   /* stop at the first row outside [v0-(cut+1), vn+cut+1]    */
#define LBL_RANGE_SKIP      1   /* ignore out-of-range rows, keep going                     */

/* Where the per-line scalars (shifted centre, widths, strength) are evaluated. */
#define LBL_PREP_DEVICE     0   /* HIP kernel (default)                                     */
#define LBL_PREP_HOST       1   /* host libm, same operation order as spectra.c:17-45       */

/* lbl_compute flags */
#define LBL_OUT_DEVICE      1   /* k is a device pointer                                    */
#define LBL_ASYNC           2   /* return after enqueueing; pair with lbl_synchronize       */
#define LBL_SCALE_DENSITY   4   /* multiply by number density P x /(kb T): the lines slot of
                                   Spectroscopy.compute_absorption (spectroscopy.py:181-191) */
#define LBL_ACCUMULATE      8   /* add into k instead of overwriting it                     */
#define LBL_FARFIELD       16   /* this call: lines at least 4 tile half-widths away (and beyond
                                   every line core) enter through one power series per tile
                                   instead of point by point -- truncation <= ~1.5e-11 relative,
                                   3-4x faster at 0.001 cm-1 (same as option "farfield" = 1);
                                   on grids so coarse that a tile is as wide as a line's window
                                   (0.1 cm-1 and up) nothing is far and the call runs without  */
#define LBL_DEFER_FINISH   32   /* with LBL_ASYNC | LBL_OUT_DEVICE and remove_pedestal: everything
                                 * is queued except the last kernels, which apply the pedestal
                                 * to k (and the copies of lbl_compute_streamed): those wait for
                                 * lbl_finish_deferred (or lbl_synchronize).  A long call can
                                 * then be queued FIRST and still be the LAST to add into a block
                                 * other calls write meanwhile -- the reference's loop has no
                                 * such order to keep, its calls are serial (spectroscopy.py:166).
                                 * One deferred call at a time; ignored where it cannot apply
                                 * (no pedestal, several level passes, host output). */

typedef struct lbl_engine lbl_engine;

/* Creates an engine bound to HIP device `device`.  Fails with LBL_NO_DEVICE when the HIP
 * runtime reports no such GPU: there is no CPU fallback. */
int lbl_engine_create(int device, lbl_engine **engine);
int lbl_engine_destroy(lbl_engine *engine);

/* Message of the calling thread's last failure on this handle (of the handle's last failure
 * if this thread has had none; "" if none); engine may be NULL to read the message of a failed
 * lbl_engine_create on the calling thread.  The pointer stays valid until the calling thread's
 * next failure or next call of this function. */
const char *lbl_last_error(const lbl_engine *engine);

/* Uploads one molecule: replaces the per-call SQLite row loop of absorption.c:44-86
 * (tips_data, mass_data, line_parameters of spectral_database.c:49-180).
 *   rows in reference row order; local_iso_id raw (0 means isotopologue 10);
 *   mass[32] indexed isoid-1 (isoid 0 at slot 9);
 *   tips_temperature[num_t]; tips_data[num_iso*num_t] isotopologue-major.
 * On success *molecule receives a small non-negative handle. */
int lbl_molecule_load(lbl_engine *engine, int64_t n_lines,
                      const double *nu, const double *sw, const double *gamma_air,
                      const double *gamma_self, const double *n_air, const double *elower,
                      const double *delta_air, const int32_t *local_iso_id,
                      const double *mass, int32_t num_iso, int32_t num_t,
                      const double *tips_temperature, const double *tips_data,
                      int32_t *molecule);
int lbl_molecule_free(lbl_engine *engine, int32_t molecule);

/* Absorption cross-section spectra [m2 molecule-1] of one molecule at n_levels levels:
 * the batched form of absorption() (absorption.c:19-99).
 *   k: n_levels rows of (vn-v0)*n_per_v doubles, row stride level_stride doubles
 *      (0 means dense); host memory unless LBL_OUT_DEVICE.
 *   evals (optional): sum over accepted lines and levels of last-first+1, the
 *      reference's inner-loop iteration count (spectra.c:48-62), computed in closed form. */
int lbl_compute(lbl_engine *engine, int32_t molecule, int32_t n_levels,
                const double *temperature, const double *pressure, const double *vmr,
                int32_t v0, int32_t vn, int32_t n_per_v, int32_t cut_off,
                int32_t remove_pedestal, int32_t range_policy, int32_t flags,
                double *k, int64_t level_stride, int64_t *evals);

/* lbl_compute into device memory (LBL_OUT_DEVICE required) with the result ALSO delivered to
 * host memory while the call still computes: the grid is worked through in `pieces` runs of
 * tiles (1..8), and the first `columns` points of every level of a finished run are copied to
 * `host` (row l at host + l*host_pitch bytes; page-locked memory from lbl_host_alloc for the
 * copies to overlap) beside the kernels of the next run.  What the reference's callers get --
 * a host array per call (gas_optics.py:65,91) -- without a copy behind the last kernel.  The
 * values are those of lbl_compute (same kernels, same order of additions).  With LBL_ASYNC the
 * copies are complete after lbl_synchronize -- and until then the copied part of `k` must not be
 * written again: the engine orders calls by the memory they WRITE (two calls into one block run
 * in the order they were made), not by what a copy still reads; the same holds for
 * lbl_copy_rows_to_host with LBL_ASYNC.  (An event behind every copy, for later writers to wait
 * for, was measured: it takes the copies' back-to-back rate away, 3.3 -> 3.6 ms for the call that
 * delivers four blocks.) */
int lbl_compute_streamed(lbl_engine *engine, int32_t molecule, int32_t n_levels,
                         const double *temperature, const double *pressure, const double *vmr,
                         int32_t v0, int32_t vn, int32_t n_per_v, int32_t cut_off,
                         int32_t remove_pedestal, int32_t range_policy, int32_t flags,
                         double *k, int64_t level_stride, void *host, int64_t host_pitch,
                         int64_t columns, int32_t pieces);

/* Queues the part of a call that LBL_DEFER_FINISH kept back (no-op without one).  lbl_deferred:
 * 1 while a call is kept back, 0 otherwise -- in particular right after a call whose
 * LBL_DEFER_FINISH could not be honoured and which therefore finished at once. */
int lbl_finish_deferred(lbl_engine *engine);
int lbl_deferred(const lbl_engine *engine);
/* Drops what a call kept back instead of queueing it: the call's target block and host range
 * are then never written by it (what it queued before works in buffers of the engine).  For
 * hosts that fail between a deferred call and its lbl_finish_deferred and are about to release
 * the block.  lbl_copy_to_host, lbl_copy_rows_to_host and lbl_order_stream_after_engine finish a
 * deferred call first (like lbl_synchronize); lbl_device_free and lbl_host_free do so only when
 * the memory they release is memory that call still has to write (its output block, the host
 * range of its delivery) -- a free of anything else, from whichever thread, leaves the deferral
 * and with it the order of a pipeline's additions alone. */
int lbl_cancel_deferred(lbl_engine *engine);

/* Waits for everything enqueued on the engine (all of its streams); a deferred call is finished
 * first. */
int lbl_synchronize(lbl_engine *engine);

/* Zeroes n_levels rows of n doubles (row stride level_stride, 0 = dense) of host memory, or of
 * device memory with LBL_OUT_DEVICE (LBL_ASYNC: queued like a compute call).  What the reference
 * returns for a molecule without partition-function rows or transitions (absorption.c:41,
 * :53-59): hosts use it to give a caller-supplied output buffer those semantics. */
int lbl_fill_zero(lbl_engine *engine, double *k, int32_t n_levels, int64_t n,
                  int64_t level_stride, int32_t flags);

/* Options (thirteen; anything else is LBL_BAD_ARGUMENT):
 *   "prep"                LBL_PREP_DEVICE (default) / LBL_PREP_HOST: where the per-line scalars are formed
 *   "points_per_lane"     0 = by the grid (default), 1/2/4/8 grid points per lane of the accumulate kernel
 *   "timing"              0/1/2: HIP events around every kernel; 2 = only around the accumulate and
 *                         far-field series launches, so that the others keep running back to back
 *   "workspace_bytes"     per-lane workspace that bounds the levels of one pass (default 4 GiB)
 *   "farfield"            0/1: distant lines by their power series (farfield.h); default 0
 *   "aligned_tiles"       0/1: cell-aligned tiles also without the far-field series; default 0
 *   "overlap_pedestal"    0/1: pedestal pre-pass on a side stream beside the accumulate launch; default 1
 *   "scan_chain"          0/1: the pedestal recurrence by relaxation where it applies, the serial
 *                         chain behind it; 0 = the serial chain alone; default 1
 *   "relax_launches"      0, 2..7: relaxation sweeps before the serial chain takes what has not
 *                         settled; 0 = three, or five for tables with more than two runs per window
 *   "lanes"               0, 2..8: streams that asynchronous calls rotate over; 0 = by kind of call
 *   "overlap_plain"       0/1: plain asynchronous calls on grids larger than small_points take turns
 *                         on two lanes, so that one call's last workgroups run beside the next
 *                         call's first; default 1; 0 = back to back on the first stream
 *   "small_points"        grids of up to so many points x levels count as short calls (default 2^20)
 *   "skip_delivery_lanes" 0/1: lbl_compute_streamed avoids the internal streams that share a
 *                         hardware queue with the copy stream; default 1
 * A library built with -DLBL_ABLATE (scripts/ablate_*.sh; never the shipped one) also takes "ablate":
 * parts of the accumulate kernel switched off for timing, results wrong. */
int lbl_set_option(lbl_engine *engine, const char *name, int64_t value);

/* With option timing=1 (or 2): accumulated kernel milliseconds and launch counts since the last
 * reset; index 0 line-scalar prep (+ tile schedule: one prologue launch), 1 the far-field
 * series kernels (and the tile schedule when it is launched alone, LBL_PREP_HOST), 2 Voigt
 * accumulate (+ combine), 3 pedestal,
 * 4 continuum band spectra, 5 continuum interpolation, 6 cross-section fit, 7 cross-section
 * interpolation.  Synchronizes the streams. */
int lbl_timing(lbl_engine *engine, double ms[8], int64_t launches[8], int32_t reset);

/* Same indices: the milliseconds during which AT LEAST ONE timed launch of the kind was running
 * since the last reset (the union of the launches' intervals on the device's clock).  Queued calls
 * take turns on several streams and their launches overlap: the sum lbl_timing returns then counts
 * such a stretch twice, this counts it once -- evals / busy time is the rate of the kernel while any
 * of it runs.  Call it BEFORE the lbl_timing(..., reset = 1) that clears both. */
int lbl_timing_busy(lbl_engine *engine, double busy_ms[8]);

/* The engine's first HIP stream (a hipStream_t), for callers that time with their own events.
 * Blocking calls run on it back to back; asynchronous calls into device memory rotate over
 * several streams, so events on this one do not bracket them (use lbl_synchronize / lbl_timing,
 * or lbl_order_stream_after_engine in front of the caller's own event). */
void *lbl_stream(lbl_engine *engine);

/* Sharing HBM blocks with another HIP user of the same device without stopping the host (the
 * reference has no analogue: it is serial and host-only; this is what lets the collection of a
 * multi-GPU call, pylbl_amd/distributed.py, start behind the kernels instead of behind a
 * lbl_synchronize).  `stream` is a hipStream_t of the caller, NULL for the null stream.
 *   lbl_order_stream_after_engine: work the caller queues on `stream` from now on runs after
 *     everything queued on the engine so far (it may read what the engine wrote);
 *   lbl_order_engine_after_stream: everything the engine queues from now on runs after what
 *     `stream` holds now (the engine may overwrite what that work read). */
int lbl_order_stream_after_engine(lbl_engine *engine, void *stream);
int lbl_order_engine_after_stream(lbl_engine *engine, void *stream);

/* Device memory helpers so that hosts without a HIP binding can keep spectra in HBM. */
int lbl_device_alloc(lbl_engine *engine, int64_t bytes, void **pointer);
int lbl_device_free(lbl_engine *engine, void *pointer);
int lbl_copy_to_host(lbl_engine *engine, void *host, const void *device, int64_t bytes);
/* `rows` rows of `row_bytes` bytes from device memory (row pitch device_pitch bytes) into host
 * memory (row pitch host_pitch bytes): lets a host place the first grid.size columns of every
 * level straight into its final array, e.g. beta[level, mechanism, :]
 * (pyLBL/spectroscopy.py:188-191), without an intermediate copy.  Without LBL_ASYNC it returns
 * when the rows have arrived. */
int lbl_copy_rows_to_host(lbl_engine *engine, void *host, int64_t host_pitch,
                          const void *device, int64_t device_pitch, int64_t row_bytes,
                          int64_t rows, int32_t flags);
/* Page-locked host memory for results: copies into it run at the full rate of the host link and,
 * with LBL_ASYNC in lbl_copy_rows_to_host, beside the kernels queued after them (the copy waits
 * for everything queued before it on any lane; lbl_synchronize waits for the copy). */
int lbl_host_alloc(lbl_engine *engine, int64_t bytes, void **pointer);
int lbl_host_free(lbl_engine *engine, void *pointer);

/* Debug/inspection: per-line scalars for one level as the engine computed them, in
 * reference row order: n_lines x 8 doubles {centre, doppler hwhm, lorentz hwhm, strength,
 * first index, last index, status (1 evaluated, 0 skipped, -1 not accepted), pedestal}. */
int lbl_line_scalars(lbl_engine *engine, int32_t molecule, double temperature,
                     double pressure, double vmr, int32_t v0, int32_t vn, int32_t n_per_v,
                     int32_t cut_off, int32_t remove_pedestal, int32_t range_policy,
                     double *derived);

/* ---------------------------------------------------------------------------------------
 * MT-CKD continua: mechanism slot 1 of Spectroscopy.compute_absorption
 * (pyLBL/spectroscopy.py:193-197).  Replaces BandedContinuum.spectra
 * (pyLBL/mt_ckd/utils.py:157-174) and the Continuum.spectra method of the 16 band classes
 * (water_vapor.py, carbon_dioxide.py, nitrogen.py, oxygen.py, ozone.py).
 * ------------------------------------------------------------------------------------- */

/* Band formulas (the class each one replaces). */
#define LBL_BAND_H2O_SELF        0  /* WaterVaporARMSelfContinuum      columns bs296, bs260          */
#define LBL_BAND_H2O_FOREIGN     1  /* WaterVaporIASIForeignContinuum  columns bfh2o, scale          */
#define LBL_BAND_CO2             2  /* CarbonDioxideHartmannContinuum  columns bfco2, chi, exponent  */
#define LBL_BAND_N2_ROTATION     3  /* NitrogenCIAPureRotationContinuum ct_296, ct_220, sf_296, sf_220 */
#define LBL_BAND_N2_FUNDAMENTAL  4  /* NitrogenCIAFundamentalContinuum xn2_272, xn2_228, a_h2o       */
#define LBL_BAND_N2_OVERTONE     5  /* NitrogenCIAFirstOvertoneContinuum xn2                         */
#define LBL_BAND_O2_FUNDAMENTAL  6  /* OxygenCIAFundamentalContinuum   o2_f, o2_t                    */
#define LBL_BAND_O2_NIR          7  /* OxygenCIANIRContinuum           o2_inf1                       */
#define LBL_BAND_O2_NIR2         8  /* OxygenCIANIR2Continuum          analytic shape / wavenumber   */
#define LBL_BAND_O2_NIR3         9  /* OxygenCIANIR3Continuum          o2_inf3                       */
#define LBL_BAND_O2_VISIBLE     10  /* OxygenVisibleContinuum          o2_invis                      */
#define LBL_BAND_O2_HERZBERG    11  /* OxygenHerzbergContinuum         analytic shape                */
#define LBL_BAND_O2_UV          12  /* OxygenUVContinuum               o2_infuv                      */
#define LBL_BAND_O3_CHAPPUIS    13  /* OzoneChappuisWulfContinuum      x_o3, y_o3, z_o3              */
#define LBL_BAND_O3_HARTLEY     14  /* OzoneHartleyHugginsContinuum    o3_hh0, o3_hh1, o3_hh2        */
#define LBL_BAND_O3_UV          15  /* OzoneUVContinuum                o3_huv                        */
#define LBL_MAX_BANDS            8

/* One band: `size` coefficients per column on the grid lower_bound + j*resolution
 * (utils.py:136-143); column[c] is the offset (in doubles) of column c in `table`, -1 if the
 * formula has fewer columns. */
typedef struct lbl_band
{
    int32_t kind;
    int32_t size;
    double lower_bound;
    double resolution;
    int64_t column[4];
} lbl_band;

/* Mole fractions a level supplies, in this order (the reference passes a dictionary of all
 * gases, spectroscopy.py:173): the gas the continuum belongs to, H2O, O2, N2, and the sum
 * over every gas of the atmosphere (air_number_density, utils.py:16-28). */
#define LBL_VMR_SELF   0
#define LBL_VMR_H2O    1
#define LBL_VMR_O2     2
#define LBL_VMR_N2     3
#define LBL_VMR_TOTAL  4
#define LBL_VMR_COUNT  5

/* Uploads the bands of one continuum (<= LBL_MAX_BANDS) and their coefficient table. */
int lbl_continuum_load(lbl_engine *engine, int32_t n_bands, const lbl_band *bands,
                       const double *table, int64_t table_size, int32_t *continuum);
int lbl_continuum_free(lbl_engine *engine, int32_t continuum);

/* Uploads a spectral grid [cm-1] (any ascending array; the reference interpolates onto the
 * caller's own array, utils.py:171-173) and keeps it in HBM.  A grid whose every element equals
 * first + i*(second - first) in double precision -- what numpy.arange fills -- is recognised
 * (checked element by element), and the interpolation kernels then form the wavenumber in
 * registers, the same bits, instead of reading it. */
int lbl_grid_load(lbl_engine *engine, int64_t n, const double *wavenumber, int32_t *grid);
int lbl_grid_free(lbl_engine *engine, int32_t grid);

/* Continuum extinction [m-1] at n_levels levels on a loaded grid: BandedContinuum.spectra.
 *   pressure in Pa; vmr: n_levels rows of LBL_VMR_COUNT doubles;
 *   extinction: n_levels rows of n doubles, row stride level_stride (0 = dense); host memory
 *   unless LBL_OUT_DEVICE; LBL_ASYNC and LBL_ACCUMULATE as for lbl_compute. */
int lbl_continuum_compute(lbl_engine *engine, int32_t continuum, int32_t grid,
                          int32_t n_levels, const double *temperature, const double *pressure,
                          const double *vmr, int32_t flags, double *extinction,
                          int64_t level_stride);

/* Several continua in ONE pass over the grid: what compute_absorption does with the continua of
 * a gas (spectroscopy.py:193-197 adds each of them into mechanism slot 1; water vapour has two,
 * :58-61) and, in its "gas" / "total" formats, with the slots of all gases (:225-234).
 *   continua: n_continua handles, evaluated and added in this order -- every continuum's bands
 *   summed from zero, then continuum after continuum onto the block: the same values, bit for
 *   bit, as n_continua calls of lbl_continuum_compute (the first writing, the others with
 *   LBL_ACCUMULATE), at 8 (+ 8 with LBL_ACCUMULATE) bytes of HBM traffic per point and level
 *   instead of 16 + 24 (n_continua - 1);
 *   vmr: [n_continua][n_levels][LBL_VMR_COUNT] (LBL_VMR_SELF differs between the continua);
 *   extinction: device memory only (LBL_OUT_DEVICE required); LBL_ASYNC, LBL_ACCUMULATE and
 *   level_stride as for lbl_continuum_compute.  At most 64 bands in all. */
int lbl_continuum_compute_many(lbl_engine *engine, int32_t n_continua, const int32_t *continua,
                               int32_t grid, int32_t n_levels, const double *temperature,
                               const double *pressure, const double *vmr, int32_t flags,
                               double *extinction, int64_t level_stride);

/* Coarse spectra [cm-1] of every band for one level, concatenated in band order:
 * Continuum.spectra(temperature, pressure [mb], vmr) (utils.py:98-108). */
int lbl_continuum_bands(lbl_engine *engine, int32_t continuum, double temperature,
                        double pressure_mb, const double *vmr, double *spectra);

/* ---------------------------------------------------------------------------------------
 * ARTS-crossfit absorption cross-sections: mechanism slot 2 of compute_absorption
 * (pyLBL/spectroscopy.py:199-203).  Replaces CrossSection.absorption_coefficient
 * (pyLBL/arts_crossfit/cross_section.py:19-48) and calculate_xsec_fullmodel
 * (pyLBL/arts_crossfit/xsec_aux_functions.py:14-121).
 * ------------------------------------------------------------------------------------- */
#define LBL_MAX_XSEC_BANDS 16

/* Uploads the bands of one molecule: sizes[n_bands] frequencies per band; `frequency` [Hz]
 * concatenated over the bands, strictly ascending within a band; `coefficients` per band the
 * [4][size] matrix p00, p10, p01, p20 of the fit z = p00 + p10 T + p01 P + p20 T^2,
 * concatenated in band order. */
int lbl_xsec_load(lbl_engine *engine, int32_t n_bands, const int32_t *sizes,
                  const double *frequency, const double *coefficients, int32_t *xsec);
int lbl_xsec_free(lbl_engine *engine, int32_t xsec);

/* Cross sections [m2] at n_levels levels on a loaded grid (lbl_grid_load); with
 * LBL_SCALE_DENSITY multiplied by P vmr /(kb T), i.e. slot 2 itself [m-1] (vmr may be NULL
 * otherwise).  out / level_stride / LBL_OUT_DEVICE / LBL_ASYNC / LBL_ACCUMULATE as for
 * lbl_continuum_compute. */
int lbl_xsec_compute(lbl_engine *engine, int32_t xsec, int32_t grid, int32_t n_levels,
                     const double *temperature, const double *pressure, const double *vmr,
                     int32_t flags, double *out, int64_t level_stride);

/* The fit with negative values removed on the bands' own frequency grids, concatenated in
 * band order: calculate_xsec_fullmodel(temperature, pressure [Pa], coeffs) per band. */
int lbl_xsec_bands(lbl_engine *engine, int32_t xsec, double temperature, double pressure,
                   double *values);

/* Same signature and semantics as the reference's absorption() (absorption.c:19-30):
 * opens the SQLite file, uploads the molecule, computes one level.  Returns 0 on success, 1 on
 * error (message on stderr), 0 with zeros when the molecule has no TIPS rows
 * (absorption.c:53-59).
 *   Device: LBL_DEVICE if set, else LOCAL_RANK / OMPI_COMM_WORLD_LOCAL_RANK / SLURM_LOCALID
 *   modulo the visible devices, else 0 (indices relative to HIP_VISIBLE_DEVICES), fixed at the
 *   first call of the process.
 *   Residency: the uploaded line table is kept per (path, formula) and re-read when the file's
 *   mtime, size or inode changed; at most LBL_COMPAT_CACHE (default 16) molecules stay in HBM,
 *   least recently used evicted first. */
int lbl_absorption(double pressure, double temperature, double volume_mixing_ratio,
                   int v0, int vn, int n_per_v, double *k, char *database, char *formula,
                   int cut_off, int remove_pedestal);

/* The reference's own symbol (pyLBL/c_lib/absorption.c:19-30, bound by
 * pyLBL/c_lib/gas_optics.py:11-12,68-91): an alias of lbl_absorption, exported so that this
 * library, installed as libabsorption*.so beside gas_optics.py, serves an unmodified pyLBL.
 * Define LBL_NO_REFERENCE_SYMBOL before including this header when the reference's own
 * absorption.h is in the same translation unit. */
#ifndef LBL_NO_REFERENCE_SYMBOL
int absorption(double pressure, double temperature, double volume_mixing_ratio,
               int v0, int vn, int n_per_v, double *k, char *database, char *formula,
               int cut_off, int remove_pedestal);
#endif

/* The reader of lbl_absorption by itself -- no GPU involved: one molecule's rows out of an SQLite
 * file in pyLBL's schema (database.py:418-486) through the reference C reader's own SELECTs
 * (absorption.c:69-70; spectral_database.c:55, :113, :143), as host arrays.  Replaces, for hosts,
 * the ORM route of pyLBL/database.py:350-395 (Database.gas / Database.tips) at the speed of the C
 * row loop.  `name` is any alias of the molecule.  Status LBL_OK or LBL_TABLE_* (message:
 * lbl_last_error(NULL)); *table is then NULL.
 *   lbl_table_shape: row counts (transitions; isotopologue rows; TIPS isotopologues x temperatures),
 *     the molecule's id and ordinary formula.
 *   lbl_table_copy: copies out (any pointer may be NULL) the seven fp64 columns nu, sw, gamma_air,
 *     gamma_self, n_air, elower, delta_air as columns[7][n_lines]; local_iso_id[n_lines] raw;
 *     isoid / mass of the isotopologue rows in row order; tips_temperature[num_t];
 *     tips_data[num_iso][num_t]. */
typedef struct lbl_table lbl_table;
int lbl_table_read(const char *path, const char *name, lbl_table **table);
int lbl_table_shape(const lbl_table *table, int64_t *n_lines, int32_t *molecule_id,
                    int32_t *n_isotopologues, int32_t *num_iso, int32_t *num_t,
                    char *formula, int32_t formula_bytes);
int lbl_table_copy(const lbl_table *table, double *columns, int32_t *local_iso_id,
                   int64_t *isoid, double *mass, double *tips_temperature, double *tips_data);
int lbl_table_free(lbl_table *table);

/* File -> HBM in one call (read as above, masses filed under isoid with 0 -> 10, then
 * lbl_molecule_load): what lbl_absorption does at its first call on a (path, formula). */
int lbl_molecule_load_sqlite(lbl_engine *engine, const char *path, const char *name,
                             int32_t *molecule);

/* State of the compatibility entry: the device it computes on (-1 before its first call) and
 * the number of molecules it keeps in HBM. */
int lbl_compat_state(int32_t *device, int32_t *resident);

/* Library version string. */
const char *lbl_version(void);

#ifdef __cplusplus
}
#endif
#endif
