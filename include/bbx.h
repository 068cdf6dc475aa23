/* bbx.h -- C ABI of libbbx_hip.so: the MI355X (gfx950) implementation of the
 * per-image reduction hot path of BlackBOX.
 *
 * The reference (pmvreeswijk/BlackBOX v1.5.1) has no FFI: its boundary is the
 * Python call boundary of the stage functions in blackbox.py.  Each entry point
 * below names the reference function (file:line) whose bulk array work it
 * replaces; the Python mirrors of those functions (blackbox_amd/reduce.py) keep
 * the reference's names and argument meaning and call these through ctypes.
 * INTEGRATION.md shows the binding a BlackBOX maintainer would add.
 *
 * Conventions
 *  - plain C: pointers + sizes, no torch / numpy types.  `d_` = device pointer
 *    (hipMalloc'ed by the caller, e.g. a torch tensor's data_ptr()), `h_` = host
 *    pointer.  The caller owns every buffer.
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*, NULL
 *    = default stream) unless its comment says "synchronises".
 *  - return value: 0 = BBX_OK, negative = error (bbx_strerror).  HIP errors are
 *    reported, never swallowed; there is no CPU fallback.
 *  - images are C-order, row 0 first (numpy layout of the FITS data array).
 *  - one bbx_ctx per worker process / GPU (or per lane of a pipeline); a ctx is not thread-safe, and its calls belong on ONE
 *    stream at a time: the device work lists, counters and scratch of a ctx are shared by its calls.  Exceptions: the host
 *    waits (bbx_wait, bbx_sync: their state belongs to the calling thread), and bbx_fpack_tiles / bbx_fpack_body, whose
 *    per-call state lives in the caller's buffers and in a hint table the library keeps per calling STREAM: the output stage
 *    runs them on a second and -- from a writer thread -- a third stream of a lane's ctx at the same time.
 *
 * Geometry (reference define_sections, blackbox.py:6334-6402): the raw frame is
 * NY x NX = 2 x 8 channels of (dy x dx) pixels; each channel holds a data
 * section (ysize_chan x xsize_chan) at its left, a vertical overscan strip at
 * its right, and horizontal overscan rows between the two channel rows.  The
 * reduced frame is (2*ysize_chan) x (8*xsize_chan).
 */
#ifndef BBX_H
#define BBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBX_OK            0
#define BBX_ERR_ARG      -1   /* bad argument / geometry */
#define BBX_ERR_HIP      -2   /* a HIP runtime call failed (see bbx_last_hip_error) */
#define BBX_ERR_NOMEM    -3
#define BBX_ERR_OVERFLOW -4   /* a device work list overflowed its capacity */
#define BBX_ERR_NOTCONV  -5   /* an iterative device loop hit its bound */
#define BBX_ERR_PSFWIN   -6   /* bbx_zogy_frame: the PSFs' matched-filter kernels exceed their row window (BBX_OPT_ZOGY_KWIN_OFF) */

#define BBX_NCHAN 16

/* mask bits: set_zogy.mask_value */
#define BBX_MASK_BAD        1
#define BBX_MASK_COSMIC     2
#define BBX_MASK_SAT        4
#define BBX_MASK_SATCON     8
#define BBX_MASK_SATELLITE 16
#define BBX_MASK_EDGE      32
#define BBX_MASK_XTALK     64

#define BBX_RAW_U16 0         /* raw pixels uint16 (BZERO-scaled FITS BITPIX 16) */
#define BBX_RAW_F32 1         /* raw pixels float32 */

typedef struct bbx_ctx bbx_ctx;

typedef struct {
    int32_t ny_raw, nx_raw;          /* raw frame shape, e.g. 10600 x 12000 */
    int32_t ysize_chan, xsize_chan;  /* channel data section, e.g. 5280 x 1320 */
} bbx_geom;

/* ---- context ------------------------------------------------------------------ */
int  bbx_ctx_create(int device, bbx_ctx **out);          /* synchronises */
void bbx_ctx_destroy(bbx_ctx *ctx);                      /* synchronises */
const char *bbx_strerror(int code);
const char *bbx_last_hip_error(const bbx_ctx *ctx);
int  bbx_version(void);
/* Timing experiments compile parts of kernels out of scratch builds of this library (tools/exp: -DFPV_*, -DZ3_SKIP_FFT,
 * -DBOXK_*, -DSATV, -DZ3_STAMPS ...; such a build computes wrong results by design).  The Makefile defines none of them;
 * bbx_build_flags() returns 0 for a product build and a bit per family compiled in otherwise (1 fpack, 2 ZOGY, 4 box
 * statistics, 8 satellite stage), so a harness can refuse a knocked-out library. */
int  bbx_build_flags(void);
/* hipStreamSynchronize(stream) + check of the ctx's device-side error flags
 * (list overflow, non-convergence).  Call before trusting host copies. */
int  bbx_sync(bbx_ctx *ctx, void *stream);

/* Options of a context.
 * BBX_OPT_LAC_LEVEL_FEED (default 0): how bbx_lacosmic obtains astroscrappy's background_level
 * (the median of the good input pixels; only CR pixels without a single good 5x5 neighbour take
 * it, which real frames rarely contain).  0: nothing is prepared; when a frame needs the level,
 * 256 workgroups select it exactly over the frame together (about 1 ms for 10^8 pixels, that
 * frame only).  1: the dense candidate pass also feeds a bracketed select (reads the mask plane too,
 * +45 % on that kernel, plus the sample / bracket kernels) and the level costs microseconds
 * when needed.  Results are identical; a host that sees the level being needed
 * (d_stats[15] != 0) can switch the feed on for a while (pipeline.FramePipeline does). */
#define BBX_OPT_LAC_LEVEL_FEED 1
/* BBX_OPT_DEBUG_LISTCAP (default 0 = off): capacity of the LA-Cosmic work lists as the kernels see
 * it, below the allocated one -- lets a test drive the list-overflow path (BBX_ERR_OVERFLOW ->
 * COSMIC-P = False) with an ordinary frame. */
#define BBX_OPT_DEBUG_LISTCAP 2
/* BBX_OPT_ZOGY_CORE: selected between two 1-D transform cores in round 2; the register-DFT core (bbx_zogy2.hip) is
 * gone, the option is accepted and has no effect. */
#define BBX_OPT_ZOGY_CORE 3
/* BBX_OPT_ZOGY_KWIN_OFF (default 0): bbx_zogy_frame takes the matched-filter kernels k_n, k_r (real space) through
 * their inverse row pass, the squares and the forward row pass on a window of 2 wh >= 4 S + 32 rows around the origin
 * only -- they are as compact as the S x S PSF stamps they are made of; the share of their energy outside the
 * window is summed on the device and must stay below 1e-6 (stamp-truncation ringing sits at 1e-9), else the call's step is flagged (device error bit 4).
 * 1: all L rows (the textbook evaluation; same results to float32 rounding). */
#define BBX_OPT_ZOGY_KWIN_OFF 4
/* BBX_OPT_FPACK_ONE_WG (default 0): bbx_fpack_tiles / bbx_fpack_body compress a row with a bit-stream buffer of half the
 * worst case first (two workgroups per CU) and redo the rows that do not fit with the full buffer; 1: full buffer for
 * every row (one workgroup per CU).  Same bytes either way. */
#define BBX_OPT_FPACK_ONE_WG 5
/* BBX_OPT_FPACK_HIST_ONLY (default 0): the three exact row medians of the quantiser's noise estimate come from a sampled
 * bracket + one counting pass (rows the bracket misses fall back to the radix histograms by themselves); 1: radix
 * histograms over all keys for every row (round 3's path).  Same medians, same bytes either way. */
#define BBX_OPT_FPACK_HIST_ONLY 6
/* BBX_OPT_WAIT_SLEEP_US (default 0): how bbx_sync / bbx_wait wait for the device.  0: the runtime's stream synchronisation
 * (it spins on a core).  n > 0: the host thread polls an event and sleeps n microseconds between polls -- a pipeline with
 * several lane, reader and writer threads per GPU would otherwise burn a core per waiting thread (5.4 cores per GPU measured
 * in round 3); costs up to n us of latency per wait.  The wait's event and the pinned copy of the error words belong to
 * the calling thread: any number of threads may wait on one context. */
#define BBX_OPT_WAIT_SLEEP_US 7
/* BBX_OPT_BKG_FULL_SORT (default 0): bbx_bkg_boxstats takes the clipped statistics of a box from a sorted bracket around its
 * median and the list of its wing pixels, and sorts only the boxes where that does not hold (ties, constant boxes); 1: every
 * box is sorted in full (rounds 2-3).  The medians are the same order statistics either way (tests compare the two). */
#define BBX_OPT_BKG_FULL_SORT 8
int  bbx_set_option(bbx_ctx *ctx, int option, int value);
/* Host waits that do not spin.  bbx_wait: everything queued on [stream] so far has finished (no error check: bbx_sync does
 * that); bbx_event_wait: [event] (a hipEvent_t, e.g. of bbx_event_create or a framework's) has completed.  sleep_us > 0:
 * poll + nanosleep; <= 0: for bbx_wait the context's BBX_OPT_WAIT_SLEEP_US, for bbx_event_wait hipEventSynchronize.
 * (what the reference does: nothing -- numpy is synchronous; these replace hipStreamSynchronize / blocking copies.) */
int  bbx_wait(bbx_ctx *ctx, void *stream);
/* Copy [nbytes] between two device-accessible addresses (device memory, or pinned host memory: hipHostMalloc'ed memory is
 * mapped into the device's address space) with a KERNEL on [stream] instead of a copy-engine transfer: the few bytes a stage
 * hands to the host (counts, list heads, scalars) then never queue behind the 100 MB transfers of the output / input stages
 * on the DMA engines.  Meant for small copies (<= a few MB). */
int  bbx_copy_kernel(void *dst, const void *src, size_t nbytes, void *stream);
int  bbx_event_wait(void *event, int sleep_us);

/* Per-step attribution of device-side errors.  Kernels report list overflow / non-convergence by
 * setting bits in a flag word of the context; the calls are asynchronous, so the host cannot see
 * them when a stage function returns.  bbx_step_mark enqueues, on [stream], a move of the flags
 * accumulated so far into *d_slot (device int32, caller-owned; e.g. one slot per stage inside the
 * frame's packed result record) and clears them.  A host that marks after every stage reads, with
 * the frame's scalars, which stage failed (bit 1 = BBX_ERR_OVERFLOW, bit 2 = BBX_ERR_NOTCONV, bit 4 = BBX_ERR_PSFWIN) and
 * applies the reference's convention: `<STEP>-P = False` and carry on (blackbox.py:1866-1878). */
int  bbx_step_mark(bbx_ctx *ctx, int32_t *d_slot, void *stream);

/* Stream plumbing for a host that pipelines frames (the reference runs one frame per worker
 * process, blackbox.py:640-700 pool_func; here one process keeps several frames in flight on
 * HIP streams).  Thin wrappers -- hipEvent without timing, hipStreamWaitEvent,
 * hipMemcpyAsync -- so the per-frame loop does not pay an ML runtime's bookkeeping per call.
 * bbx_event_query: 1 = complete, 0 = not yet, < 0 = BBX_ERR_*.
 * bbx_copy_async kind: 0 host->device, 1 device->host, 2 device->device; host memory should be
 * pinned (hipHostMalloc / hipHostRegister) for the copy to be asynchronous. */
int  bbx_event_create(void **out_event);
void bbx_event_destroy(void *event);
int  bbx_event_record(void *event, void *stream);
int  bbx_event_query(void *event);
int  bbx_stream_wait_event(void *stream, void *event);
int  bbx_copy_async(void *dst, const void *src, size_t nbytes, int kind, void *stream);

/* per-kernel timing: when enabled, the library brackets its main kernels with hipEvents
 * on the launch stream (the reference logs wall time per stage with log_timing_memory,
 * e.g. blackbox.py:4366-4367).  Slots: */
#define BBX_PROF_CALIBRATE 0   /* k_calibrate            (1 launch / frame)  */
#define BBX_PROF_LAC_DENSE 1   /* k_lac_cand             (niter launches)    */
#define BBX_PROF_LAC_SPARSE 2  /* seed+grow+clean        (niter groups)      */
#define BBX_PROF_XTALK 3       /* k_xtalk                                     */
#define BBX_PROF_VOS_STD 4     /* read-noise passes      (1 group / frame)   */
#define BBX_PROF_MASK_FINISH 5 /* mask_init tail + fill  (1 group / frame)   */
#define BBX_PROF_ZOGY 6        /* bbx_zogy_subimages, or (register-DFT core) the kernels of bbx_zogy_frame before its last one */
#define BBX_PROF_ZOGY_FINAL 7  /* k_final_rows of bbx_zogy_frame (1 launch / frame) */
#define BBX_PROF_Z_PSF_COLS 8  /* the other kernels of bbx_zogy_frame (LDS-pass core), one slot each: k_psf_cols, */
#define BBX_PROF_Z_PSF_ROWS 9  /*   k_psf_rows, */
#define BBX_PROF_Z_IMG_ROWS 10 /*   k_img_rows (2 launches / frame), */
#define BBX_PROF_Z_IMG_COLS 11 /*   k_img_cols, */
#define BBX_PROF_Z_VAR_COLS 12 /*   k_var_cols, */
#define BBX_PROF_Z_PSF_DFT 13  /*   k_psf_rowdft (the stamps' row DFTs, in front of k_psf_cols) */
#define BBX_PROF_NSLOTS 14
int  bbx_profile_enable(bbx_ctx *ctx, int on);   /* 1: clear + record; 0: clear + stop; 2: stop, keep the records */
/* synchronises on the recorded events; ms_total/calls have nslots entries; resets */
int  bbx_profile_read(bbx_ctx *ctx, double *ms_total, int32_t *calls, int nslots);

/* ---- a2 + a4 + a5(i): overscan statistics ---------------------------------------
 * replaces: inf/nan scrub blackbox.py:1461-1468, gain_corr 7442-7465 (applied on
 * the fly, raw is not modified), and the strip reductions of os_corr 6480-6490.
 *  d_mean_vos_col [16*dy] f64 : per channel and row, 3-sigma/5-iteration clipped
 *      mean of the gain-corrected vertical overscan (zeros excluded).
 *  d_hos [16*hos_rows*dx] f32 : gain-corrected horizontal-overscan rows
 *      (os_sec_hori) of each channel, BEFORE subtraction of the vertical fit.
 *  d_n_infnan [1] i64 : number of non-finite raw pixels (always 0 for u16).   */
int bbx_overscan_stats(bbx_ctx *ctx, const bbx_geom *g, const void *d_raw,
                       int raw_type, const float *h_gain /*[16]*/,
                       double *d_mean_vos_col, float *d_hos, int64_t *d_n_infnan,
                       void *stream);

/* ---- a5(ii): read noise --------------------------------------------------------
 * replaces os_corr 6572-6573: 3-sigma/5-iteration clipped std (zeros excluded) of
 * each channel's vertical overscan after subtraction of the row fit.
 *  d_vfit [16*dy] f64 : fitted vertical-overscan level per channel row.
 *  h_dlevel [16] f32  : level offset of the horizontal overscan (os_corr 6565-6568);
 *      it was subtracted from the full-width overscan rows, i.e. also from the
 *      corner that the vertical strip shares with them.
 *  d_std_vos [16] f64 (RDN{c});  float64 accumulators (tolerance) */
int bbx_vos_std(bbx_ctx *ctx, const bbx_geom *g, const void *d_raw, int raw_type,
                const float *h_gain, const double *d_vfit, const float *h_dlevel,
                double *d_std_vos, void *stream);

/* ---- a5(iii, BlackGEM): saturated columns near the overscan ----------------------
 * replaces os_corr 6624-6640: per channel and data column, the number of pixels
 * >= h_thr[c] in the [rows1] / [rows2] rows of the data section nearest to the
 * horizontal overscan, after gain and vertical-fit subtraction.
 *  d_counts [2*16*xsize_chan] i32                                               */
int bbx_satcol_counts(bbx_ctx *ctx, const bbx_geom *g, const void *d_raw,
                      int raw_type, const float *h_gain, const double *d_vfit,
                      const float *h_thr /*[16]*/, int rows1, int rows2,
                      int32_t *d_counts, void *stream);

/* ---- a4 + a5(iv) + a6 + a9(first half): fused calibration ------------------------
 * replaces, in one pass over the frame: gain_corr 7460, the per-row and
 * per-column overscan subtractions + crop of os_corr 6553/6844-6847, master-bias
 * subtraction 1679, the non-finite scrub + saturation compare of mask_init
 * 4408-4414/4494-4498/4538, and flat division 1825.
 *  d_oscan [16*xsize_chan] f64 : horizontal-overscan vector per channel.
 *  d_bias, d_flat : reduced-shape f32 masters or NULL.  d_bpm : u8 or NULL.
 *  h_satlevel [16] : saturation thresholds in e- (SATLEV{c}, compared in f32).
 *  d_data [N] f32, d_mask [N] u8 : outputs.  Saturated pixels get bit 4 and are
 *  queued in the ctx for bbx_mask_finish.                                       */
int bbx_calibrate(bbx_ctx *ctx, const bbx_geom *g, const void *d_raw, int raw_type,
                  const float *h_gain, const double *d_vfit, const double *d_oscan,
                  const float *d_bias, const float *d_flat, const uint8_t *d_bpm,
                  const float *h_satlevel, float *d_data, uint8_t *d_mask,
                  void *stream);

/* ---- a14 (+ a8 header statistics): statistics of rectangular segments ---------------
 * replaces the np.nanmedian / np.nanstd / np.ma.median calls of get_flatstats
 * (blackbox.py:3661-3820): the ny x nx area at d_data (row stride [stride] elements; a
 * sub-section of a frame is addressed by offsetting the pointers) is cut into
 * (ny/ysz) x (nx/xsz) <= 64 segments; per segment, over the pixels that are not NaN and
 * (when d_mask is given) have no mask bit other than "cosmic ray":
 *   d_out[seg][8] f64 = { n, median (np.median of float32: float32 mean of the two middle
 *   elements for even n), mean, sigma (ddof 0), n_low = #(x <= median),
 *   sigma_low = sqrt(sum_{x<=median} (x - median)^2 / (n_low - 1)), 0, 0 }
 * The medians are exact order statistics (bracketed select, no sort).                   */
int bbx_rect_stats(bbx_ctx *ctx, int ny, int nx, int stride, const float *d_data,
                   const uint8_t *d_mask, int ysz, int xsz, double *d_out, void *stream);

/* ---- a8 (MBMEAN, MBRDN, MBIASM{c}, MBRDN{c}, MD*): sigma-clipped statistics ------------
 * replaces astropy.stats.sigma_clipped_stats(x, mask_value=0) of master_prep
 * (blackbox.py:5167-5230; sigma 3, maxiters 5, centre = median, spread = std): per segment
 * (same segmentation and validity rules as bbx_rect_stats, plus skip_zero: pixels equal to
 * 0 are masked) [maxiters] rounds of clipping about the exact median, then
 * d_out[seg][8] = { n, median, mean, sigma, ... } of the survivors.  The reference feeds a
 * random 20 % subsample (unseeded); all pixels are used here.                             */
int bbx_rect_clipped_stats(bbx_ctx *ctx, int ny, int nx, int stride, const float *d_data,
                           const uint8_t *d_mask, int ysz, int xsz, double sigma, int maxiters,
                           int skip_zero, double *d_out, void *stream);

/* ---- a16 (Z-SCMED, Z-SCSTD, Z-FPEMED, Z-FPESTD): sigma-clipped statistics of a whole frame -------------------
 * replaces the sigma_clipped_stats calls zogy's optimal_subtraction makes on the Scorr and Fpsferr frames for its
 * header (3 sigma, 5 rounds, centre = median, spread = std, mask_value 0).  zogy feeds a random subset of the
 * pixels; here the sample is the lattice of every [step]-th pixel of both axes of the contiguous [ny][nx] frame.
 * Pixels with mask bits other than the cosmic-ray flag, non-finite values (astropy masks them) and (skip_zero)
 * zeros do not take part.  d_out[8] = { n, median, mean, sigma, 0, 0, 0, 0 } of the survivors of [maxiters] rounds:
 * the first four numbers of bbx_rect_clipped_stats on the lattice as one segment (exact medians; the float64 sums are
 * taken in another order), from one sort instead of a bracketed select per round.                              */
int bbx_frame_clipped_stats(bbx_ctx *ctx, int ny, int nx, const float *d_img, const uint8_t *d_mask, int step,
                            double sigma, int maxiters, int skip_zero, double *d_out, void *stream);

/* ---- a8 (GAINCF): channel scaling of the master flat copy ---------------------------
 * replaces `master_median_corr[data_sec_red[c]] /= med` and `*= ratio` (blackbox.py:5104,
 * 5139-5140): float32 IEEE division / multiplication of a rectangle in place.            */
int bbx_rect_scale(bbx_ctx *ctx, int ny, int nx, int stride, float *d_data, float factor,
                   int divide, void *stream);

/* ---- f3: reference co-add (buildref.py) ----------------------------------------------
 * bbx_coadd_prep replaces the array arithmetic of prep_inputimages (buildref.py:2602-2624,
 * 2709-2733): d_data -= d_bkg (d_bkg may be NULL: image already background-subtracted);
 * d_data[mask == edge_value] = 0; d_weights = 1 / d_bkg_std**2 where d_bkg_std != 0, else 0,
 * and 0 where (mask & discard_bits) != 0 (pass discard_bits = 0 for a single image, as the
 * reference only discards when len(imtable) > 1).  float32 throughout, bit-identical to numpy.
 *
 * bbx_resample_lanczos3 replaces SWarp's resampling step (-RESAMPLING_TYPE LANCZOS3,
 * buildref.py:1748): for each output pixel the input position comes from the bilinear
 * interpolation (float64) of d_grid [gny][gnx][2] = exact (x, y) input coordinates (0-based
 * pixel centres) at output pixels (j*gstep, i*gstep) -- SWarp evaluates the projection on such
 * a lattice too; the grid must extend one node past the last output pixel.  Data: 6x6
 * normalised LANCZOS3 taps times fscale (-FSCALE_KEYWORD); weights travel as variances through
 * the same kernel (x fscale^2); a footprint that leaves the input or holds a zero-weight pixel
 * gives weight 0.
 *
 * bbx_coadd_combine replaces SWarp's -COMBINE_TYPE step (buildref.py:1733, 1815) over n <= 32
 * resampled planes d_cube / d_wcube [n][plane_stride]: pixels with weight > 0 take part;
 * float64 sums in plane order.  CLIPPED (Gruen et al. 2014, -CLIP_SIGMA / -CLIP_AMPFRAC,
 * buildref.py:1780-1788): values further than clip_sigma*sqrt(1/w) + clip_ampfrac*|median|
 * from the median of the valid values are dropped, then WEIGHTED; d_clipmask [n][npix] (1 =
 * dropped), d_nsigma [n][npix] (deviation of the dropped values in sigma, 0 elsewhere) and
 * d_nclip [n] take SWarp's clip log (-CLIP_WRITELOG); all three optional.                    */
#define BBX_COMBINE_WEIGHTED 0
#define BBX_COMBINE_AVERAGE  1
#define BBX_COMBINE_MEDIAN   2
#define BBX_COMBINE_CLIPPED  3
#define BBX_COMBINE_MIN      4
#define BBX_COMBINE_MAX      5
#define BBX_COMBINE_SUM      6
int bbx_coadd_prep(bbx_ctx *ctx, int64_t npix, float *d_data, const float *d_bkg,
                   const float *d_bkg_std, const uint8_t *d_mask, int discard_bits,
                   int edge_value, float *d_weights, void *stream);
int bbx_resample_lanczos3(bbx_ctx *ctx, int in_ny, int in_nx, const float *d_in,
                          const float *d_win, int out_ny, int out_nx, const double *d_grid,
                          int gny, int gnx, int gstep, float fscale, float *d_out,
                          float *d_wout, void *stream);
int bbx_coadd_combine(bbx_ctx *ctx, int n, int64_t npix, const float *d_cube,
                      const float *d_wcube, int64_t plane_stride, int combine_type,
                      float clip_sigma, float clip_ampfrac, float *d_out, float *d_wout,
                      uint8_t *d_clipmask, float *d_nsigma, int64_t *d_nclip, void *stream);
/* bbx_clipped2mask replaces clipped2mask_loop + pass_filters (buildref.py:3686-3873) for one
 * input image: d_clip / d_nsigma = that image's plane of bbx_coadd_combine's clip mask and
 * deviations (the clip log, output frame); every clipped pixel with |nsigma| > min(fsigma) is
 * carried to the input frame through the same lattice d_grid as the resampling (rounding
 * `(x + 0.5).astype(uint16)` on 1-based positions), then the filters run in order: points with
 * |nsigma| > fsigma[k] that are not masked yet add 1 to an fsize x fsize box of a count image
 * (positive and negative deviations apart); where a count reaches fmax, the fsize x fsize box
 * ending at that pixel is masked; fsize == 1 masks the points themselves (the reference uses
 * fsize = [5, 1], fsigma = [nsigma_clip, 4], fmax = [4, 1]).  Masked pixels within
 * sqrt(dist2_limit) = 5 S-FWHM of a pixel with (data_mask & sat_bits) are released; d_weights
 * of the others are set to 0 (the second, WEIGHTED pass then ignores them).  d_mask_im
 * [in_ny][in_nx] receives the mask, d_nmasked the number of zeroed weights.  Integer work:
 * identical to the numpy code. */
int bbx_clipped2mask(bbx_ctx *ctx, int out_ny, int out_nx, const uint8_t *d_clip,
                     const float *d_nsigma, const double *d_grid, int gny, int gnx, int gstep,
                     int in_ny, int in_nx, const uint8_t *d_data_mask, int sat_bits,
                     float dist2_limit, int nfilt, const int *h_fsize, const float *h_fsigma,
                     const int *h_fmax, float *d_weights, uint8_t *d_mask_im,
                     int64_t *d_nmasked, void *stream);

/* ---- f2: FITS tile compression (fpack, blackbox.py:812-857) ------------------------
 * RICE_1, one tile per image row, block size 32; float32 images are quantised like CFITSIO's
 * fits_quantize_float with SUBTRACTIVE_DITHER_1 (noise from the 2nd/3rd/5th order MAD of the
 * row, delta = noise / qlevel) -- byte-identical to CFITSIO for the same ZDITHER0.
 *  bbx_fpack_tiles : d_img [ny][nx] of bitpix -32 (float32) / 8 / 16 / 32, d_rnd = CFITSIO's
 *    10000-value random table (float32) for bitpix -32, dither_seed = ZDITHER0 in 1..10000.
 *    d_scratch: ny * bbx_fpack_tile_stride(nx, bytepix) bytes, receives each row's stream at
 *    its stride; d_tiles [ny] of {u32 nbytes, u32 flag, f64 zscale, f64 zzero} (flag 1: row
 *    not quantisable, 2: non-finite pixel).
 *  bbx_fpack_gather: copies the streams to d_heap at d_offsets[row] (int64).              */
size_t bbx_fpack_tile_stride(int nx, int bytepix);
int bbx_fpack_tiles(bbx_ctx *ctx, int ny, int nx, const void *d_img, int bitpix, float qlevel,
                    int dither_seed, const float *d_rnd, uint8_t *d_scratch, void *d_tiles,
                    void *stream);
int bbx_fpack_gather(bbx_ctx *ctx, int ny, int nx, int bitpix, const uint8_t *d_scratch,
                     const void *d_tiles, const long long *d_offsets, uint8_t *d_heap,
                     void *stream);
/* The same compression in ONE enqueue, down to the bytes of the file: tile streams, their offsets (prefix sum on the
 * device) and the descriptor table of the COMPRESSED_IMAGE extension (big-endian) -- d_body = [ny rows of
 * {int32 len, int32 off [, int32 gzlen = 0, int32 gzoff = 0, float64 ZSCALE, float64 ZZERO]}][heap].  No host round
 * trip between the steps, so a lane can queue it behind the kernels that make the image (reference: fpack of the
 * products that are kept, blackbox.py:812-857, 3933-4035).  d_scratch: ny * bbx_fpack_tile_stride bytes; d_tiles:
 * ny * 24 bytes; d_offsets: ny int64; d_info: (4 + max_list) int64 = [heap bytes, rows listed, 1 = heap larger than
 * cap_body - table (nothing gathered), longest stream, rows the quantiser refused (length 0 in the table: the
 * host stores them gzip-compressed behind the heap, as CFITSIO does)]. */
int bbx_fpack_body(bbx_ctx *ctx, int ny, int nx, const void *d_img, int bitpix, float qlevel, int dither_seed,
                   const float *d_rnd, uint8_t *d_scratch, void *d_tiles, long long *d_offsets, uint8_t *d_body,
                   long long cap_body, long long *d_info, int max_list, void *stream);
/* the same of [scale] x the float image (bitpix -32), the product taken as the pixels are loaded -- float32, the bytes of
 * compressing the multiplied image: `_trans_limmag` = T-NSIGMA x Fpsferr (set_blackbox.py:160-162) is then never made as a
 * frame of its own (a 4N write and a 4N read of HBM per frame less) */
int bbx_fpack_body_scaled(bbx_ctx *ctx, int ny, int nx, const void *d_img, int bitpix, float qlevel, int dither_seed,
                          const float *d_rnd, uint8_t *d_scratch, void *d_tiles, long long *d_offsets, uint8_t *d_body,
                          long long cap_body, long long *d_info, int max_list, float scale, void *stream);


/* ---- a1 / f2: funpack -- reading tile-compressed images (raw frames arrive as .fits.fz;
 * read_hdulist, blackbox.py:1451) -------------------------------------------------------
 * Rice decode of row tiles: d_desc [ny][2] int32 = (length, heap offset) of each row's
 * stream, d_heap = the table heap (+ 16 readable bytes of padding).  out_kind: 0 uint8,
 * 1 uint16 = int16 + 32768 (BZERO), 2 int16, 3 int32, 4 float32 = SUBTRACTIVE_DITHER_1
 * un-quantisation ((q - r + 0.5) * ZSCALE + ZZERO).  Rows with length 0 are left untouched
 * (losslessly stored rows are filled in by the host).                                     */
int bbx_funpack_tiles(bbx_ctx *ctx, int ny, int nx, int bytepix, const int *d_desc,
                      const uint8_t *d_heap, int out_kind, void *d_out, const double *d_zscale,
                      const double *d_zzero, int dither_seed, const float *d_rnd, void *stream);

/* Uncompressed raw frames: FITS stores BITPIX 16 pixels big-endian with BZERO 32768 (read_hdulist, blackbox.py:1451:
 * astropy scales them on the host).  bbx_raw_be16: n pixels as they lie in the file -> uint16 = byteswap(x) ^ 0x8000 in
 * one pass on the device (round 4 did this with four frame-sized tensor passes of the host framework). */
int bbx_raw_be16(const void *d_file_pixels, uint16_t *d_out, size_t n, void *stream);
/* The same for the 32-bit images a process reads before its first frame (master bias and flat: read_hdulist at
 * blackbox.py:1677, 1823; the reference image of the subtraction): n big-endian words as they lie in the file -> host
 * order, in place allowed.  The host uploads the file's bytes untouched instead of swapping 446 MB per image on one core. */
int bbx_be32(const void *d_file_words, void *d_out, size_t n, void *stream);

/* ---- a7: nonlin_corr (blackbox.py:7394-7437; set_bb.correct_nonlin is False upstream) ----
 * per channel: counts = data/gain[c]; frac = spline_c(counts) where counts <= 50000, else 1
 * (sic: uncorrected pixels end up divided by 2, reproduced as written); data /= frac + 1.
 * The splines are scipy UnivariateSpline objects in the reference's pickle; here their
 * (t, c, k) arrays, evaluated like FITPACK splev.
 *  bbx_nonlin_set : degree k (1..5), h_nknots[16], h_t / h_c [16][256] float64 (rows padded to
 *                   256 entries).  h_nknots == NULL switches the correction off again.  While
 *                   set, bbx_calibrate applies it between the overscan and the bias steps
 *                   (its place in blackbox_reduce, 1604-1624).
 *  bbx_nonlin_corr: the function on its own, in place on an overscan-corrected frame.      */
int bbx_nonlin_set(bbx_ctx *ctx, int degree, const int32_t *h_nknots, const double *h_t,
                   const double *h_c);
int bbx_nonlin_corr(bbx_ctx *ctx, const bbx_geom *g, float *d_data, const float *h_gain,
                    void *stream);

/* ---- a9 (second half): mask_init tail + fill_sat_holes ---------------------------
 * replaces mask_init 4504-4562 (crosstalk flags of saturated pixels in the 15
 * other channels, NOBJ-SAT label count, 3x3 dilation -> saturated-connected) and
 * fill_sat_holes 4584-4596 (3x3 closing + hole filling, new pixels where the
 * mask was 0).  Uses the saturated-pixel queue left by bbx_calibrate.
 *  d_nobj_sat [1] i32                                                           */
int bbx_mask_finish(bbx_ctx *ctx, const bbx_geom *g, uint8_t *d_mask,
                    int32_t *d_nobj_sat, void *stream);

/* ---- a10: LA-Cosmic ----------------------------------------------------------------
 * replaces cosmics_corr 4259-4370 = astroscrappy.detect_cosmics(sepmed=False,
 * cleantype='medmask', gain=1, satlevel=inf) + mask update + NCOSMICS label count.
 * In place: d_data becomes the cleaned array, CR pixels get bit 2 in d_mask.
 *  readnoise : RDNOISE in e-; if d_rdn16 != NULL it is instead taken on the device as
 *  float32(nanmean(d_rdn16[0..15])) (the 16 RDN{c} of bbx_vos_std), sparing a host hop.
 *  d_stats [16] i32 : [0..niter-1] pixels flagged per iteration, [6] number of
 *  8-connected CR objects, [7] total CR pixels, [8+2k],[9+2k] (k<4) work-list sizes of
 *  iteration k (pruned candidates, first-growth survivors) for diagnostics.                                  */
int bbx_lacosmic(bbx_ctx *ctx, int ny, int nx, float *d_data, uint8_t *d_mask,
                 float sigclip, float sigfrac, float objlim, int niter,
                 float readnoise, const double *d_rdn16, int32_t *d_stats,
                 void *stream);

/* ---- a11: crosstalk -------------------------------------------------------------------
 * replaces xtalk_corr 7138-7258.  h_coeffs[source*16 + victim], float64.          */
int bbx_xtalk(bbx_ctx *ctx, const bbx_geom *g, float *d_data, const uint8_t *d_mask,
              const double *h_coeffs /*[256]*/, void *stream);

/* ---- a13: mask_header counts + edge fill ---------------------------------------------
 * replaces mask_header 4601-4620 (d_counts[6] i64: bad, edge, saturated,
 * saturated-connected, satellite trail, cosmic ray -- pixels with that bit set)
 * and the edge fill 1959-1974 (edge pixels <- np.median of their channel;
 * d_chan_median [16] f32 also returned).                                        */
int bbx_mask_counts(bbx_ctx *ctx, int64_t npix, const uint8_t *d_mask,
                    int64_t *d_counts, void *stream);
int bbx_edge_fill(bbx_ctx *ctx, const bbx_geom *g, float *d_data,
                  const uint8_t *d_mask, float *d_chan_median, void *stream);

/* ---- a8: master frames ------------------------------------------------------------------
 * replaces master_prep 4906-5073: per-pixel np.median of nframes (<= 32) reduced
 * calibration frames; each frame is first divided by h_norm[i] (flats: MEDSEC of
 * 4929-4941; NULL or 1 for bias/dark; 0 = leave as is); with flat_fix != 0 pixels that
 * are BPM-edge (== 32) or <= 0 become 1 (5071-5073).
 *  h_frames : host array of nframes device pointers, each npix float32.             */
int bbx_median_stack(bbx_ctx *ctx, int64_t npix, int nframes, const float *const *h_frames,
                     const float *h_norm, const uint8_t *d_bpm, int flat_fix,
                     float *d_out, void *stream);

/* ---- a12: satellite trails (sat_detect, blackbox.py:4163-4254) ------------------------------
 * The reference calls acstools.satdet.detsat(buf=40, sigma=3, h_thresh=0.2) + make_mask(sigma=5)
 * [EXT: acstools is not in the image; its probabilistic Hough transform draws random pixels, ASTA
 * is a CNN without weights].  Specified in oracle/sattrail.py: 2x2 sum binning; acstools' front end
 * (percentile (4.5, 93) rescale, skimage Canny sigma 3 with thresholds 0.1 / 0.2 of the maximum,
 * remove_small_objects(60): pinned against scikit-image); full Hough accumulator over ntheta angles
 * with skimage.transform.hough_line's cells (h_cos_sin[2*k], [2*k+1] = cos, sin of theta_k, float64,
 * supplied by the host so that both sides use the same table; acstools' grid is 2 .. 177.5 deg in
 * half-degree steps); strongest line with >= 210 votes whose supporting edge pixels reach within
 * 40 px of both frame borders; perpendicular profile, strip -> bit 16 in d_mask.
 *  h_gauss [gauss_radius + 1] : scipy.ndimage's Gaussian weights for sigma = 3, centre first
 *                               (radius int(4 sigma + 0.5) = 12), float64, from the host like the angles.
 *  d_nsats [1] i32 : 8-connected objects of bit 16 (NSATS).
 *  d_info [8] f32  : level, sigma, votes, theta index, rho, strip lo, strip hi, found.     */
int bbx_sat_trails(bbx_ctx *ctx, int ny, int nx, const float *d_data, uint8_t *d_mask,
                   const double *h_cos_sin, int ntheta, const double *h_gauss, int gauss_radius,
                   int32_t *d_nsats, float *d_info, void *stream);

/* bbx_canny_edge_map: the front end of bbx_sat_trails on its own (parity tests against scikit-image):
 * np.percentile(img, (4.5, 93)) + skimage.exposure.rescale_intensity + skimage.feature.canny(sigma from
 * h_gauss, low / high = low_frac / high_frac x the rescaled maximum) + remove_small_objects(min_size,
 * connectivity 8) of a float32 image -> d_map [ny*nx] u8 (1 = edge), *d_count = number of edge pixels. */
int bbx_canny_edge_map(bbx_ctx *ctx, int ny, int nx, const float *d_img, const double *h_gauss,
                       int gauss_radius, double low_frac, double high_frac, int min_size,
                       uint8_t *d_map, int32_t *d_count, void *stream);

/* ---- a15: background mesh (zogy.get_back / mini2back; buildref.py:2398-2405, 2480-2495) ----
 * [EXT algorithm: conventions in oracle/zogy_core.py]
 * bbx_bkg_boxstats: per box x box tile (box <= 64, divides ny and nx) the sigma-clipped
 *   (median centre, 3 sigma, <= 5 iterations) median and std of the pixels with mask == 0,
 *   objmask == 0 (optional) and value != 0; boxes with fewer than limfrac*box^2 usable
 *   pixels are NaN.  d_mini_* : (ny/box)*(nx/box) float32.
 * bbx_mini_fill_filter: NaN boxes <- nan-median of the 3x3 neighbourhood (repeated),
 *   then 3x3 median filter (edge replicated), in place.
 * bbx_spline_zoom: evaluates scipy.ndimage.zoom(order=3, mode='nearest') from prefiltered
 *   B-spline coefficients d_coef[cny][cnx] (float64) with per-output-row / -column tap
 *   bases d_fy/d_fx (floor index into coef) and weights d_wy/d_wx ([n][4] float64);
 *   writes the background to d_bkg (if non-NULL) and subtracts it from d_data (if non-NULL).
 * bbx_spline_prefilter: the coefficients themselves, on the device: the mini image d_mini[nby][nbx] (float32) cut into
 *   blocks of cy x cx boxes (cy = nby, cx = nbx: one block; the channel blocks for interp_Xchan = False), every block
 *   padded by [npad] edge samples (scipy.ndimage.zoom pads 'nearest' inputs by 12) and run through
 *   scipy.ndimage.spline_filter(order = 3, mode = 'nearest') -> d_coef[(nby/cy)(cy+2 npad)][(nbx/cx)(cx+2 npad)] float64,
 *   bit for bit scipy's values (gcc builds).  zn_y = pow(z, cy + 2 npad), zn_x = pow(z, cx + 2 npad) with
 *   z = sqrt(3) - 2 correctly rounded (-0.2679491924311227), computed by the caller's libm.               */
int bbx_bkg_boxstats(bbx_ctx *ctx, int ny, int nx, int box, const float *d_data,
                     const uint8_t *d_mask, const uint8_t *d_objmask, float limfrac,
                     float *d_mini_med, float *d_mini_std, void *stream);
int bbx_mini_fill_filter(bbx_ctx *ctx, int nby, int nbx, float *d_mini, void *stream);
int bbx_spline_prefilter(bbx_ctx *ctx, int nby, int nbx, int cy, int cx, int npad, double zn_y, double zn_x,
                         const float *d_mini, double *d_coef, void *stream);
/* bbx_mini_median: np.median of a float32 device array of n values (exact order statistics; an even count gives the
 * float32 mean of the middle pair as numpy does; NaN if the array holds one) -> d_med[0].  S-BKGSTD = the median of the
 * sigma mini image (zogy's header value) without a round trip through the host.
 * bbx_zoom_candidates(ctx, d_med, nsigma): the NEXT bbx_spline_zoom_sub call of this context also lists the pixels of
 * the frame it writes with |value| >= (float)(d_med[0] * nsigma) -- the candidates of the source catalogue (peaks above
 * cat_nsigma x S-BKGSTD) -- and bbx_find_peaks on that frame with that threshold starts from the list instead of a
 * pass of its own over the frame (a threshold that is not that number raises the device's list-overflow flag: a failed
 * search, not a wrong one).  d_med = NULL: off. */
int bbx_mini_median(bbx_ctx *ctx, int n, const float *d_a, float *d_med, void *stream);
int bbx_zoom_candidates(bbx_ctx *ctx, const float *d_med, double nsigma);
int bbx_spline_zoom(bbx_ctx *ctx, int ny, int nx, const double *d_coef, int cny, int cnx,
                    const int32_t *d_fy, const double *d_wy, const int32_t *d_fx,
                    const double *d_wx, float *d_data, float *d_bkg, void *stream);
/* the same with separate input and output: d_out = d_in - background (d_in stays as it is) */
int bbx_spline_zoom_sub(bbx_ctx *ctx, int ny, int nx, const double *d_coef, int cny, int cnx,
                        const int32_t *d_fy, const double *d_wy, const int32_t *d_fx,
                        const double *d_wx, const float *d_in, float *d_out, void *stream);

/* ---- a16: ZOGY sub-image subtraction (zogy.optimal_subtraction -> run_ZOGY; call sites
 * blackbox.py:2350-2354, 2460-2465) with rocFFT.  [EXT algorithm: Zackay, Ofek & Gal-Yam
 * 2016; conventions in oracle/zogy_core.py]
 * bbx_cut_subimages / bbx_stitch_subimages: (ny/size)*(nx/size) tiles of size^2 with a
 *   zero-padded border -> [nsub][L][L], L = size + 2*border, and back (borders dropped).
 * bbx_zogy_subimages: per sub-image new N, ref R (background-subtracted), PSFs Pn, Pr
 *   (unit sum, centred on pixel [0,0]), variance images Vn, Vr; h_scal[nsub][6] =
 *   sigma_n, sigma_r, f_n, f_r, dx, dy.  Outputs D, S, S_corr, F_psf, F_psf_err.
 *   Inputs are used as FFT sources (not modified).                                    */
/* helpers: V = max(data,0) + bkg_std^2 ; PSF stamps [nsub][S][S] -> [nsub][L][L] centred on [0,0] */
int bbx_variance(bbx_ctx *ctx, int64_t n, const float *d_data, const float *d_bkgstd,
                 float *d_var, void *stream);
int bbx_embed_psf(bbx_ctx *ctx, int nsub, int S, int L, const float *d_stamps, float *d_out,
                  void *stream);
int bbx_cut_subimages(bbx_ctx *ctx, int ny, int nx, int size, int border,
                      const float *d_img, float *d_subs, void *stream);
int bbx_stitch_subimages(bbx_ctx *ctx, int ny, int nx, int size, int border,
                         const float *d_subs, float *d_img, void *stream);
int bbx_zogy_subimages(bbx_ctx *ctx, int L, int nsub, float *d_new, float *d_ref,
                       float *d_pn, float *d_pr, float *d_vn, float *d_vr,
                       const float *h_scal, float *d_D, float *d_S, float *d_Scorr,
                       float *d_Fpsf, float *d_Fpsferr, void *stream);

/* bbx_zogy_frame: the same subtraction for a whole frame in one call, with the library's own 2-D
 * FFT (bbx_zogy2.hip: 1-D transforms in LDS, transposition folded into the store patterns, the
 * ZOGY algebra in the registers of the column passes) instead of rocFFT + separate element-wise
 * kernels; the cut into (ny/size)*(nx/size) sub-images of side L = size + 2*border (zero-padded
 * at the frame edge), the variance images V = max(d, 0) + sigma^2 and the stitching of the
 * results are part of its kernels.
 *   d_new, d_ref         : background-subtracted frames [ny][nx] (ref on the new frame's grid)
 *   d_sig_new, d_sig_ref : background sigma images [ny][nx]
 *   d_psf_n, d_psf_r     : unit-sum PSF stamps [nsub][S][S] (centre at S/2)
 *   h_scal [nsub][6]     : sigma_n, sigma_r, f_n, f_r, dx, dy
 *   d_D, d_Scorr, d_Fpsf, d_Fpsferr (and d_S, may be NULL) : full frames [ny][nx]
 * Supported sub-image sides: bbx_zogy_frame_supported(L) != 0 (1400 = the reference's
 * 1320 + 2*40, and 64 / 128 / 140 for tests); other sizes go through bbx_zogy_subimages.
 * S = S_n - S_r is formed in real space (identical in exact arithmetic to the inverse transform
 * of S^ that bbx_zogy_subimages takes). */
int bbx_zogy_frame_supported(int L);
/* bbx_zogy_candidates(ctx, thr): thr > 0: the following bbx_zogy_frame calls of this context also list the pixels with
 * |Scorr| >= thr as they write them ([EXT] zogy's get_trans thresholds |Scorr| at transient_nsigma of its settings file,
 * 6 in the deployment: Settings/set_qc.py:387); bbx_find_peaks on that very Scorr frame with the same threshold then starts from the list
 * instead of a pass of its own over the frame (same regions, same peaks).  0 switches the listing off. */
int bbx_zogy_candidates(bbx_ctx *ctx, float thr);
int bbx_zogy_frame(bbx_ctx *ctx, int ny, int nx, int size, int border, const float *d_new,
                   const float *d_ref, const float *d_sig_new, const float *d_sig_ref,
                   const float *d_psf_n, const float *d_psf_r, int S, const float *h_scal,
                   float *d_D, float *d_S, float *d_Scorr, float *d_Fpsf, float *d_Fpsferr,
                   void *stream);

/* bbx_zogy_frame_mini: bbx_zogy_frame with the two sigma images given as their mini images instead of frames.  zogy makes
 * the full-frame sigma image of each side with mini2back(data_bkg_std_mini, ...) [EXT; call sites buildref.py:2480-2495] only
 * to form the variance images: here the kernel that cuts the sub-images reads sigma off the B-spline coefficients of the
 * mini image (bbx_spline_prefilter's output; one patch per channel for the new frame: interp_Xchan False), so the two
 * frames are never written to HBM nor read back (2 x 4N bytes written, 2 x 4N x (L / size)^2 read per frame of N pixels).
 * Per pixel the spline is evaluated as a float32 cubic of the float64-folded coefficients: within 3e-7 (relative) of
 * what bbx_spline_zoom writes (csrc/bbx_spline.h).  Needs frames whose groups of four pixels are aligned (size, border,
 * nx, channel width multiples of 4; 16-byte aligned pointers), BBX_ERR_ARG otherwise: callers then zoom the mini images
 * (bbx_spline_zoom) and use bbx_zogy_frame. */
typedef struct {
    const double *d_coef;      /* bbx_spline_prefilter output: [(nby / cy) * (cy + 2 npad)][(nbx / cx) * (cx + 2 npad)] */
    int32_t nby, nbx;          /* mini image shape in boxes; the frame is nby * box by nbx * box pixels */
    int32_t cy, cx;            /* boxes per patch: (nby, nbx) = one patch, (boxes per channel) = one patch per channel */
    int32_t box, npad;         /* box size [pix]; padding of every patch in boxes (12 = scipy.ndimage.zoom's) */
} bbx_spline_image;
int bbx_zogy_frame_mini(bbx_ctx *ctx, int ny, int nx, int size, int border, const float *d_new,
                        const float *d_ref, const bbx_spline_image *sig_new, const bbx_spline_image *sig_ref,
                        const float *d_psf_n, const float *d_psf_r, int S, const float *h_scal,
                        float *d_D, float *d_S, float *d_Scorr, float *d_Fpsf, float *d_Fpsferr,
                        void *stream);

/* ---- a17: PSFEx model evaluation [EXT: zogy.get_psf / psfex poly] ----------------------
 * stamp[s][p] = sum_k terms[s][k] * basis[k][p]: terms [nsrc][ncoef] f32 = the polynomial
 * terms x'^i y'^j (i + j <= poldeg, PSFEx order) of each source position, basis
 * [ncoef][npix] f32 = the PSF_MASK cube of the .psf file, out [nsrc][npix] f32.  f32 MFMA
 * (v_mfma_f32_32x32x2_f32): k-ordered float32 fma chain.                                  */
int bbx_psf_model(bbx_ctx *ctx, int nsrc, int ncoef, int npix, const float *d_terms,
                  const float *d_basis, float *d_out, void *stream);

/* ---- a17: PSF-weighted optimal flux (zogy.get_psfoptflux) at integer positions:
 * flux = sum(P D / V) / sum(P^2 / V), err = 1 / sqrt(sum(P^2 / V)) over an S x S stamp of
 * the unit-sum PSF model d_psfs[nsrc][S][S] centred on (d_ys, d_xs); pixels off the frame
 * or with V <= 0 are skipped.                                                          */
int bbx_psf_optflux(bbx_ctx *ctx, int ny, int nx, const float *d_D, const float *d_V,
                    const float *d_psfs, int S, int nsrc, const int32_t *d_ys,
                    const int32_t *d_xs, float *d_flux, float *d_err, void *stream);
/* the same on a background-subtracted frame d_D with its sigma image: V = max(D, 0) + sigma^2 is formed
 * per stamp pixel (float32, as bbx_variance would) instead of being written out for the whole frame */
int bbx_psf_optflux_sigma(bbx_ctx *ctx, int ny, int nx, const float *d_D, const float *d_sigma,
                          const float *d_psfs, int S, int nsrc, const int32_t *d_ys,
                          const int32_t *d_xs, float *d_flux, float *d_err, void *stream);
/* the same with sigma read off its mini image at the stamp pixels (bbx_zogy_frame_mini's companion: no sigma frame exists) */
int bbx_psf_optflux_mini(bbx_ctx *ctx, int ny, int nx, const float *d_D, const bbx_spline_image *sigma,
                         const float *d_psfs, int S, int nsrc, const int32_t *d_ys,
                         const int32_t *d_xs, float *d_flux, float *d_err, void *stream);

/* ---- a17: transient candidates: 8-connected regions of |img| >= thr (|S_corr| >= T-NSIGMA,
 * set_qc.py:387); per region the pixel of largest |value| (first in C order on ties).
 * d_yx[max_out][2] (y, x), d_val[max_out], *d_count = number of regions (may exceed max_out,
 * only the first max_out in arbitrary order are stored).                                  */
int bbx_find_peaks(bbx_ctx *ctx, int ny, int nx, const float *d_img, float thr, int max_out,
                   int32_t *d_yx, float *d_val, int32_t *d_count, void *stream);

/* ---- generic: number of 8-connected objects of (mask & bit) --------------------------
 * replaces ndimage.label(..., structure=ones(3,3)) counts (NOBJ-SAT 4545,
 * NCOSMICS 4355, NSATS 4230).                                                     */
int bbx_count_objects(bbx_ctx *ctx, int ny, int nx, const uint8_t *d_mask,
                      int bit, int32_t *d_count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BBX_H */
