/*
 * apgpu.h - C ABI of libapgpu.so: the MI355X (gfx950) implementation of AstroPhotography's per-pixel
 * calibrate -> mask -> arithmetic -> stack-reduce hot path.
 *
 * The reference (DaveStrickland/AstroPhotography v0.5.1) is pure Python and has no FFI of its own;
 * its boundary for this path is the Ap* class API.  Each entry point below replaces
 * the NumPy/astropy/ccdproc arithmetic of one reference method (cited as file:line relative to
 * AstroPhotography/ in the reference tree); the Python Ap* shells in astrophotography_amd/ bind them
 * with ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless its name ends in _host;
 *  - the caller owns every buffer; the library allocates nothing and keeps no state (one exception, stated at
 *    apgpu_stack_args.workspace: a stack call WITHOUT a workspace makes a stream-ordered temporary allocation);
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is asynchronous
 *    with respect to the host, no call synchronises;
 *  - functions return 0 on success or a negative APGPU_E* code; apgpu_last_error() returns a
 *    thread-local message for the last failing call;
 *  - images are row-major [H][W]; frame stacks are contiguous slabs [N][H][W] (P = H*W pixels);
 *  - workspace sizes are returned by the matching *_ws_bytes function; workspaces need 16-byte
 *    alignment (any hipMalloc / torch allocation has it).
 */
#ifndef APGPU_H
#define APGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APGPU_VERSION 130           /* 0.1.3: apgpu_resample_stack_sigclip, APGPU_STACK_NONFINITE_UNCLIPPED, apgpu_combine_ccdproc_f64(form) */

/* error codes */
#define APGPU_OK            0
#define APGPU_EINVAL       -1       /* bad argument (NULL pointer, bad enum, size <= 0 ...) */
#define APGPU_EUNSUPPORTED -2       /* valid request this build cannot serve (e.g. N > APGPU_MAX_STACK) */
#define APGPU_ELAUNCH      -3       /* HIP runtime error at launch */
#define APGPU_EWORKSPACE   -4       /* workspace too small */

/* element types of pixel data */
#define APGPU_F32 0
#define APGPU_U16 1
#define APGPU_F64 2                 /* float64 masters / frames (apgpu_calibrate_mixed, *_f64 entry points) */

/* ApImArith._allowed_ops (core/ApImArith.py:34) */
#define APGPU_OP_ADD 0
#define APGPU_OP_SUB 1
#define APGPU_OP_MUL 2
#define APGPU_OP_DIV 3

/* clip centre / deviation estimators (astropy SigmaClip cenfunc / stdfunc) */
#define APGPU_CENTER_MEDIAN 0
#define APGPU_CENTER_MEAN   1
#define APGPU_DEV_STD       0
#define APGPU_DEV_MAD_STD   1

/* largest N one stack call reduces: up to 128 frames the per-pixel column lives in registers; 129 .. 512 frames are reduced chunk by
 * chunk (64 frames in registers at a time: clipped mean with its median / std planes, plain median, the median / mad_std configuration
 * on raw frames) with an LDS-resident exact kernel behind them for the pixels they are not sure of and for every other option */
#define APGPU_MAX_STACK 512

const char *apgpu_last_error(void);
int apgpu_version(void);

/* ---------------------------------------------------------------------------------------------
 * A1  ApCalibrate._generate_flat (core/ApCalibrate.py:166-190)
 *     norm = np.nanmean(flat) [float32 pairwise sum in 8192-element pieces, / count of non-NaN],
 *     nflat = flat / norm.  Bit-exact with numpy.  norm is written to norm_out[0] (device).
 * ------------------------------------------------------------------------------------------- */
size_t apgpu_flat_normalize_ws_bytes(int64_t n_pixels);
int apgpu_flat_normalize_f32(const float *flat, float *nflat, float *norm_out, int64_t n_pixels,
                             void *ws, size_t ws_bytes, void *stream);
/* The same for a float64 flat (a master written by ccdproc is float64 and _read_fits keeps float data as it is,
 * core/ApCalibrate.py:301-305): float64 pairwise sum in the same 8192-element pieces, float64 division. */
size_t apgpu_flat_normalize_f64_ws_bytes(int64_t n_pixels);
int apgpu_flat_normalize_f64(const double *flat, double *nflat, double *norm_out, int64_t n_pixels,
                             void *ws, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A2  ApCalibrate.calibrate arithmetic block (core/ApCalibrate.py:439-464) incl. the read-time
 *     conversions of _read_fits (core/ApCalibrate.py:304-326), over a slab of N frames:
 *       x = (f32(raw) [+ pedestal[f]]) - bias;  D = dark_still_biased ? dark - bias : dark;
 *       x = x - exp_ratio[f] * D;  out = nflat ? (nflat != 0 ? x / nflat : x) : x
 *     every operation separately rounded to float32 (no FMA contraction, IEEE division).
 *     exp_ratio[N] (float32(EXPTIME_img / EXPTIME_dark)) and pedestal[N] are device arrays;
 *     pedestal may be NULL; a zero pedestal entry adds nothing.  nflat may be NULL (no flat).
 * ------------------------------------------------------------------------------------------- */
int apgpu_calibrate(const void *raw, int raw_dtype, const float *bias, const float *dark,
                    const float *nflat, const float *exp_ratio, const float *pedestal,
                    int dark_still_biased, float *out, int64_t n_frames, int64_t n_pixels, void *stream);

/* A2 with float64 inputs: any mix of raw APGPU_U16 / APGPU_F32 / APGPU_F64 and masters APGPU_F32 / APGPU_F64, evaluated with
 * NumPy's per-operation type promotion exactly as the reference's expressions do (core/ApCalibrate.py:301-305: only
 * non-float FITS data is converted to float32; scripts/ap_combine_darks.py:437: ApMasterCal writes float64 masters):
 *   x = raw - bias in T1 = f64 if raw or bias is f64;  D = dark - bias in T2 = f64 if dark or bias is f64 (or D = dark,
 *   T2 = dark's type);  ds = T2(exp_ratio) * D;  y = x - ds in T3 = T1 | T2;  out = y / nflat in T4 = T3 | nflat's type.
 * exp_ratio[N] and pedestal[N] (may be NULL) are float64 device arrays (python floats; the pedestal is added in the raw
 * frame's own type).  out_dtype must be T4 (APGPU_F64 if any participating array is float64, else APGPU_F32). */
int apgpu_calibrate_mixed(const void *raw, int raw_dtype, const void *bias, int bias_dtype, const void *dark, int dark_dtype,
                          const void *nflat, int nflat_dtype, const double *exp_ratio, const double *pedestal,
                          int dark_still_biased, void *out, int out_dtype, int64_t n_frames, int64_t n_pixels, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A6/A7 + the fused north-star kernel: per-pixel sigma-clipped reduction along N of a slab
 *     [N][P], optionally calibrating each value on the fly (A2) so the raw slab is read once.
 *     Semantics = astropy.stats.sigma_clipped_stats(cube, axis=0, sigma_lower, sigma_upper,
 *     maxiters, cenfunc, stdfunc) (astropy/stats/sigma_clipping.py:298-383, 924-937): non-finite
 *     values dropped; <= maxiters passes of { centre, dev, keep lo <= x <= hi }; final bounds
 *     applied to all values; outputs are the float64 statistics rounded once to float32.
 *     maxiters < 0 iterates to convergence.  With maxiters = 1, centre = median, dev = mad_std and
 *     sigma_lower = sigma_upper = 5 this is the ccdproc.combine configuration of
 *     scripts/ap_combine_darks.py:394-420.
 *     Outputs (each may be NULL): mean/median/std [P] float32, count [P] int32 survivors,
 *     mean_f64/std_f64 [P] float64 (the unrounded statistics: ccdproc.combine and CCDData.write keep float64,
 *     scripts/ap_combine_darks.py:411-439), and the N-shard partial moments of the survivors
 *     (all-reduced over ranks, then apgpu_moments_finalize[_f64]) in one of two layouts:
 *       moments_f64 == 0: float32 moments[3][P] = sum, count, sum of squares (a mean-only exchange all-reduces
 *                         the contiguous [2][P] prefix, 8 bytes per pixel; the float32 sums round per rank);
 *       moments_f64 != 0: double sum[P], double sumsq[P], int32 count[P] laid out back to back in `moments`
 *                         (20 P bytes, 8-byte aligned): ranks add float64 sums and int32 counts, the combined
 *                         mean is the float64 combine rounded once to float32 (12 bytes per pixel mean-only);
 *       moments_f64 == 2: the same layout, but the call ADDS its moments to what the buffer already holds: a stack
 *                         of more than APGPU_MAX_STACK frames is reduced chunk by chunk into one buffer (hierarchical
 *                         clipping: every chunk is clipped against its own statistics);
 *       moments_f64 == 3: packed float64 planes double moments[3][P] = sum, count, sum of squares: the count rides as a
 *                         float64 (exact to 2^53), so ONE all-reduce of the contiguous [2][P] prefix (16 bytes per pixel,
 *                         24 with the third plane for a std) combines the ranks (apgpu_moments_finalize_f64p);
 *       moments_f64 == 4: layout 3, ADDING to what the buffer holds (a rank's share reduced in several chunks).
 * ------------------------------------------------------------------------------------------- */
typedef struct apgpu_stack_args {
    const void *frames;          /* [N][P] APGPU_F32 or APGPU_U16 */
    int32_t dtype;
    int32_t n_frames;            /* 1 .. APGPU_MAX_STACK */
    int64_t n_pixels;
    /* fused calibration: bias == NULL -> frames are reduced as they are */
    const float *bias;           /* [P] */
    const float *dark;           /* [P] */
    const float *nflat;          /* [P] or NULL */
    const float *exp_ratio;      /* [N] */
    const float *pedestal;       /* [N] or NULL */
    int32_t dark_still_biased;
    int32_t center;              /* APGPU_CENTER_* */
    int32_t dev;                 /* APGPU_DEV_* */
    int32_t maxiters;            /* >= 1, or < 0 = until convergence */
    double sigma_lower;
    double sigma_upper;
    const uint8_t *pixmask;      /* [P] or NULL; non-zero = pixel excluded (outputs NaN / 0) */
    float *mean;                 /* [P] or NULL */
    float *median;               /* [P] or NULL */
    float *std;                  /* [P] or NULL */
    int32_t *count;              /* [P] or NULL */
    void *moments;               /* NULL, or the partial moments of the survivors in the layout moments_f64 selects */
    int64_t frame_stride;        /* elements between the starts of consecutive frames; 0 = n_pixels.
                                    > n_pixels lets a call reduce a row stripe of a larger slab */
    double *mean_f64;            /* [P] or NULL */
    double *std_f64;             /* [P] or NULL */
    int32_t moments_f64;         /* layout of `moments`, see above */
    int32_t flags;               /* APGPU_STACK_* bits below; 0 = defaults (was reserved0) */
    void *workspace;             /* NULL, or apgpu_stack_ws_bytes(n_pixels, ..) bytes for the two-kernel scheme (below) */
    size_t workspace_bytes;
} apgpu_stack_args;

/* apgpu_stack_args.flags.  The lean kernels (mean / count / moments outputs, median centre, std deviation, full slot
 * counts) clip on a float32 fast path: moments and bound tests in float32 with an error margin, and every wavefront in
 * which a comparison falls inside the margin is redone on the float64 path - the survivor sets are ALWAYS those of the
 * float64 evaluation; the mean carries the float32 rounding of its sum (a few 1e-3 ulp(float32); still the float64 mean
 * rounded once for > 90 % of the pixels, within 1 ulp otherwise).
 *   APGPU_STACK_EXACT_MOMENTS   force the float64 path: the mean is the float64 mean of the survivors rounded once.
 *   APGPU_STACK_MOMENTS_MEAN    the float64-layout moments will only be used for a mean (sum, count): lets the fast path
 *                               fill them; their sum of squares then has float32 accuracy (not good for a std).
 *                               Without it float64-layout moments always come from the float64 path. */
#define APGPU_STACK_EXACT_MOMENTS 1
#define APGPU_STACK_MOMENTS_MEAN 2
#define APGPU_STACK_SINGLE_KERNEL 4   /* never the fast kernel + redo pass pair described below: one complete kernel, no workspace used */
/*   APGPU_STACK_NONFINITE_UNCLIPPED  (dev == APGPU_DEV_MAD_STD only) a column that holds a non-finite value is not clipped: mean /
 *                               count / std of its finite values.  This is what ccdproc >= 2.2's Combiner.sigma_clipping does
 *                               for scripts/ap_combine_darks.py:394-420: it delegates to astropy.stats.sigma_clip, whose general
 *                               path hands the callables np.ma.median / mad_std a plain array in which invalid values are NaN
 *                               - NaN bounds, nothing rejected (golden group G12, arrays c*_b_*, run for real).  Without the
 *                               flag non-finite values are masked and the rest is clipped (ccdproc <= 2.1's own loop). */
#define APGPU_STACK_NONFINITE_UNCLIPPED 8

/* The two-kernel scheme and its workspace.  A clipped stack of up to 128 frames (and the chunked kernel for 129 .. 512) with
 * lean outputs runs as a FAST kernel - the float32 fast path alone, four wavefronts per SIMD - followed by a REDO PASS of the
 * complete kernel over what the fast kernel could not finish: single pixels (unsure comparisons, non-finite values, masked
 * pixels, operands outside the guards of the fast division) gathered from per-segment lists, and whole 256-pixel tiles that
 * (in 64-pixel blocks) the fast kernel gave up without trying once more than an eighth of the pixels it had seen were failing - so that a stack
 * whose pixels mostly fail costs about one pass of the complete kernel, not both kernels in full.  The lists, counters and
 * tile flags live in `workspace`:
 *   - size: apgpu_stack_ws_bytes(n_pixels, &zero_bytes) bytes (about 4 bytes per pixel), 16-byte aligned;
 *   - its first zero_bytes bytes must be ZERO at the first call (hipMemsetAsync once; zeroing all of it is fine); every call
 *     leaves them zero again (the statistics and one mode word excepted, below).  After a call that returned an error, zero
 *     them again;
 *   - the layout depends on n_pixels: a workspace serves calls of ONE n_pixels (re-zero the prefix before using it for
 *     another image size) and must not be shared by calls that may run concurrently (two streams);
 *   - the 32 bytes at APGPU_STACK_WS_STATS_OFFSET hold four int64 counters, cumulative over the calls that used this
 *     workspace and never reset by the library: calls, pixels, pixels listed, 64-pixel blocks given up (apgpu_stack_ws_stats) -
 *     the caller may copy them out (after the stream has been synchronised) to see which fraction of its data leaves the fast
 *     path: (pixels_listed + 64 * blocks_given_up) / pixels;
 *   - the workspace also remembers, from one call to the next, whether the guard is needed: a call that sent under an eighth
 *     of its pixels to the redo pass (listed, or in blocks given up) lets the next call on the same workspace skip the fast
 *     kernel's look at the counters (it costs 1.5 % of the benchmark, 6 % of a 16-frame stack); the first call after the data has turned bad therefore runs unguarded - both
 *     kernels in full, measured 1.20 - 1.25 x the complete kernel's time when every column fails (profiles/r05/redo_sweep.txt)
 *     - and sets the guard for the calls after it.
 *     The ccdproc.combine configuration (one pass of median / mad_std) keeps a mode word of its own in the same line: after a call
 *     that gave up more than an eighth of its sampled 64-pixel blocks, the next call tries only every 16th tile on the fast kernel.
 * workspace == NULL: the call allocates and frees a stream-ordered temporary of the same size itself (hipMallocAsync /
 * hipMemsetAsync / hipFreeAsync on `stream` - three more runtime calls per stack; the only place where the library
 * allocates); if that fails too, or with APGPU_STACK_SINGLE_KERNEL, the complete kernel reduces the whole stack in one launch. */
#define APGPU_STACK_WS_STATS_OFFSET 16384
typedef struct apgpu_stack_ws_stats {
    int64_t calls, pixels, pixels_listed, blocks_given_up;
} apgpu_stack_ws_stats;
size_t apgpu_stack_ws_bytes(int64_t n_pixels, size_t *zero_bytes);

int apgpu_stack_sigclip(const apgpu_stack_args *args, void *stream);

/* A6 on FLOAT64 frames (BITPIX -64 inputs of ApMasterCal: ccdproc's Combiner is float64 throughout, so such frames must
 * not pass through the float32 stack kernels): base = median, dev = mad_std, ONE strict pass, float64 mean in frame order, std
 * of the kept values, survivor count.  `form` selects which published Combiner.sigma_clipping is followed, operation for
 * operation (oracle/apref.c apref_combine_ccdproc_form, golden group G12 - both forms run for real):
 *   APGPU_CCDPROC_ASTROPY (ccdproc >= 2.2, what requirements.txt:18 resolves to today): astropy.stats.sigma_clip - rejected
 *       where x < base - dev * low or x > base + dev * high; a column holding a non-finite value is not clipped at all;
 *   APGPU_CCDPROC_LEGACY  (ccdproc <= 2.1): rejected where x - base < -low * dev or x - base > high * dev; non-finite values
 *       are masked and the rest is clipped.
 * frames: double [N][P] with frame_stride elements between frames (0 = n_pixels); outputs may be NULL; workspace:
 * double[2][N][P] (apgpu_combine_ccdproc_f64_ws_bytes).  A correctness path (insertion sort per pixel). */
#define APGPU_CCDPROC_ASTROPY 0
#define APGPU_CCDPROC_LEGACY 1
size_t apgpu_combine_ccdproc_f64_ws_bytes(int32_t n_frames, int64_t n_pixels);
int apgpu_combine_ccdproc_f64(const double *frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride, double low,
                              double high, int32_t form, double *mean, int32_t *count, double *std, void *workspace,
                              size_t workspace_bytes, void *stream);

/* Plain median along N (np.nanmedian(axis=0)); config 4.  Optional fused calibration as above. */
int apgpu_stack_median(const apgpu_stack_args *args, void *stream);

/* Name of the kernel variant apgpu_stack_sigclip (median_only == 0) / apgpu_stack_median (!= 0) dispatches for `args`
 * (slot count, raw type, fused calibration, rich / lean, full / padded), as rocprofv3 prints it without the namespace,
 * e.g. "stack_sigclip_kernel<64, float, true, false, true>".  Launches nothing; name_host is a HOST buffer. */
int apgpu_stack_kernel_name(const apgpu_stack_args *args, int median_only, char *name_host, size_t name_bytes);

/* mean = sum / cnt from all-reduced float32 moments (planes sum, cnt; cnt == 0 -> NaN).  `std` must be NULL:
 * sumsq / cnt - mean^2 from float32 sums about zero cancels catastrophically for CCD-range data (APGPU_EUNSUPPORTED);
 * use the float64 layout for a standard deviation. */
int apgpu_moments_finalize(const float *moments, float *mean, float *std, int64_t n_pixels, void *stream);

/* The float64 layout: mean = sum / cnt and std = sqrt(max(sumsq / cnt - mean^2, 0)) evaluated in float64, stored as
 * float32 (mean, std) and/or float64 (mean_f64, std_f64); every output may be NULL; sumsq may be NULL if no std is
 * wanted; cnt == 0 -> NaN. */
int apgpu_moments_finalize_f64(const double *sum, const double *sumsq, const int32_t *count, float *mean, float *std,
                               double *mean_f64, double *std_f64, int64_t n_pixels, void *stream);

/* The packed float64 layout (moments_f64 == 3 / 4): the three planes as pointers (buffer, buffer + P, buffer + 2 P);
 * sumsq may be NULL if no std is wanted; count == 0 -> NaN. */
int apgpu_moments_finalize_f64p(const double *sum, const double *count, const double *sumsq, float *mean, float *std,
                                double *mean_f64, double *std_f64, int64_t n_pixels, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A3  astropy.stats.sigma_clipped_stats(data, sigma) with axis=None as called at
 *     core/ApFindBadPixels.py:191 (astropy/stats/sigma_clipping.py:385-433, numpy nan-functions):
 *     global iterative clip of an image with numpy's arithmetic reproduced exactly (exact median by
 *     radix select, numpy-ordered pairwise sums):
 *       _f32: float32 data, float32 statistics (float32 masters);
 *       _f64: float64 data and statistics - what numpy computes for integer images (the caller widens
 *             uint16/int16 pixels exactly) and for float64 data.
 *     Results stay on the device: stats_out[10] float64 =
 *       { mean, median, std, lo, hi, iterations, survivors, min, max, reserved }
 *     (lo/hi = bounds of the last pass; min/max of the survivors).  maxiters < 0 = until convergence.
 * ------------------------------------------------------------------------------------------- */
size_t apgpu_sigclip_global_ws_bytes(int64_t n_pixels);
int apgpu_sigclip_global_f32(const float *data, int64_t n_pixels, double sigma_lower, double sigma_upper,
                             int maxiters, double *stats_out, void *ws, size_t ws_bytes, void *stream);
size_t apgpu_sigclip_global_f64_ws_bytes(int64_t n_pixels);
int apgpu_sigclip_global_f64(const double *data, int64_t n_pixels, double sigma_lower, double sigma_upper,
                             int maxiters, double *stats_out, void *ws, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * F2  ApImageDifference (scripts/ap_calc_read_noise.py:122): out = float64(a) - float64(b) where neither
 *     bad1 nor bad2 (uint8, non-zero = bad, either may be NULL) flags the pixel, NaN elsewhere; feed
 *     `out` to apgpu_sigclip_global_f64 (maxiters 1, huge sigma) for np.std/mean/median/min/max of
 *     diff[good] in numpy's order.  a, b: APGPU_F32 or APGPU_U16.
 * ------------------------------------------------------------------------------------------- */
int apgpu_image_difference_f64(const void *a, const void *b, int dtype, const uint8_t *bad1, const uint8_t *bad2,
                               double *out, int64_t n_pixels, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A4  ApFindBadPixels._generate_sigmaclip_mask (core/ApFindBadPixels.py:194-216):
 *     mask = (data < f32(lo)) | (data > f32(hi)) as uint8; nbad_out[0] (device int64) = sum(mask).
 *     If thresholds_dev != NULL the two float64 thresholds are read from the device
 *     (thresholds_dev[0], [1]) instead of the by-value arguments, so A3 -> A4 chains without a sync.
 * ------------------------------------------------------------------------------------------- */
int apgpu_threshold_mask_f32(const float *data, int64_t n_pixels, double lothresh, double hithresh,
                             const double *thresholds_dev, uint8_t *mask, int64_t *nbad_out, void *stream);

/* A4 user overlays (core/ApFindBadPixels.py:70-158): mask[r0:r1, c0:c1] += value for n_rects
 * 0-based half-open rectangles rects[n][4] = {r0, r1, c0, c1} (device int32). uint8 wrap-around. */
int apgpu_mask_add_rects_u8(uint8_t *mask, int64_t height, int64_t width, const int32_t *rects,
                            int32_t n_rects, int32_t value, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A5  ApFixBadPixels.fix_bad_pixels (core/ApFixBadPixels.py:292-445): every pixel with mask != 0
 *     becomes the median of the good pixels of the ORIGINAL image inside the (2*deltapix+1)^2 window
 *     clipped to the image, if at least min_valid of them exist; otherwise it is left unchanged.
 *     out may not alias data.  stats_out[3] (device int64) = { nbad, nfixed, nremaining }.
 *     Any deltapix >= 0 (register-resident sorting networks for 1..3, rank counting beyond); any alignment of
 *     data/out (a frame cut out of a slab).  _f64: the same on a float64 image, medians in float64 - what the
 *     reference computes after a float64 calibration (np.median in the input's dtype).
 * ------------------------------------------------------------------------------------------- */
int apgpu_fix_badpix_f32(const float *data, const uint8_t *mask, int64_t height, int64_t width,
                         int32_t deltapix, int32_t min_valid, float *out, int64_t *stats_out, void *stream);
int apgpu_fix_badpix_f64(const double *data, const uint8_t *mask, int64_t height, int64_t width,
                         int32_t deltapix, int32_t min_valid, double *out, int64_t *stats_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A8  ApImArith.process_files op block (core/ApImArith.py:320-333): out = a (op) b, with b an image
 *     (b != NULL) or the scalar float32(scalar).  APGPU_F32: IEEE float32.  APGPU_U16: image (op)
 *     image only, ADD/SUB/MUL wrap modulo 2^16 (numpy); DIV and scalar operands are rejected with
 *     APGPU_EUNSUPPORTED just as numpy raises UFuncTypeError.
 * ------------------------------------------------------------------------------------------- */
int apgpu_imarith(const void *a, const void *b, double scalar, int op, int dtype, void *out,
                  int64_t n_pixels, void *stream);
/* The float64 cases of the same block: a float64 image (op) image / scalar, and float32 (op) float64 image pairs -
 * NumPy computes in float64 as soon as one operand ARRAY is float64 and stores in the dtype of `a`
 * (np.add(data1, data2, out=zeros_like(data1)), same_kind casting).  a_dtype / b_dtype: APGPU_F32 | APGPU_F64. */
int apgpu_imarith_f64(const void *a, int a_dtype, const void *b, int b_dtype, double scalar, int op, void *out,
                      int64_t n_pixels, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A9  RawConv split geometry (core/RawConv.py:111-128, 163-190): planes[k] = where(colour == k,
 *     max(raw - black[k], 0), 0) for k = R0 G1 B2 G2 3, full-size planes [4][H][W].
 *     pattern_host[4] = colour of cell positions (0,0),(0,1),(1,0),(1,1); black_host may be NULL.
 * ------------------------------------------------------------------------------------------- */
int apgpu_bayer_split_u16(const uint16_t *raw, int64_t height, int64_t width, const int32_t *pattern_host,
                          const int32_t *black_host, uint16_t *planes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * F1  FITS payload <-> device arrays: what astropy.io.fits does at core/ApCalibrate.py:270-277 (read,
 *     uint=True) and :392-399 (write).  `payload` is the big-endian data unit exactly as in the file
 *     (device copy of the raw bytes).
 *     decode: BITPIX 16 with unsigned16 != 0 (BSCALE 1, BZERO 32768) -> uint16[n];
 *             BITPIX 16 with unsigned16 == 0 -> float32[n] (integers are converted to float32 at read
 *             time, core/ApCalibrate.py:304-307); BITPIX -32 -> float32[n]; BITPIX 32 -> int32[n];
 *             BITPIX -64 -> float64[n]; BITPIX 64 -> int64[n].
 *     encode: float32[n] -> big-endian BITPIX -32 payload; float64[n] -> BITPIX -64 payload.
 * ------------------------------------------------------------------------------------------- */
int apgpu_fits_decode(const void *payload, int bitpix, int unsigned16, void *out, int64_t n_pixels, void *stream);
int apgpu_fits_encode_f32(const float *data, void *payload, int64_t n_pixels, void *stream);
int apgpu_fits_encode_f64(const double *data, void *payload, int64_t n_pixels, void *stream);

/* ---------------------------------------------------------------------------------------------
 * F3  Affine Lanczos-3 resample of registered frames - the step the reference hands to SWarp
 *     (scripts/resample_all.sh:123-131 RESAMPLING_TYPE LANCZOS3; :298 FSCALE = 1/EXPTIME; :330-342 the call).
 *     frames [n_frames, h_in, w_in] float32; mask [h_in, w_in] uint8 shared by all frames, or NULL;
 *     affines [n_frames][6] float64 (device): xin = A0*x + A1*y + A2, yin = A3*x + A4*y + A5 maps OUTPUT
 *     pixel (x = column, y = row, 0-based) to INPUT coordinates.  With affines_per_tile != 0 the array is
 *     [n_frames][tiles_y][tiles_x][6], one transform per APGPU_RESAMPLE_TILE_H x APGPU_RESAMPLE_TILE_W tile of the
 *     output (tiles_y = ceil(h_out / TILE_H), tiles_x = ceil(w_out / TILE_W); x, y stay absolute): a piecewise-
 *     affine form of a smooth non-linear registration such as TAN -> TAN through the sky (wcs.tile_affines).
 *     conserve_flux != 0 multiplies every pixel by |A0*A4 - A1*A3|, the area of an output pixel in input pixels
 *     (SWarp's FSCALASTRO_TYPE VARIABLE, resample_all.sh:129: total flux, not surface brightness, is preserved); fscale [n_frames] float32 or NULL (= 1);
 *     lut [n_phases + 1][6] float32 (device, 8-byte aligned): row p = normalised Lanczos-3 weights of the
 *     taps floor(xin)-2 .. floor(xin)+3 for the fractional offset p / n_phases (ops.lanczos3_table).
 *     out [n_frames, h_out, w_out] float32 = NaN where any of the 36 taps is outside the frame, masked or
 *     non-finite; weight_out (uint8, same shape, or NULL) = 1 where out is defined.  The co-add itself
 *     (COMBINE_TYPE MEDIAN / AVERAGE / SUM / CLIPPED) is apgpu_stack_median / apgpu_stack_sigclip on `out`:
 *     both skip the NaN pixels.  Exact definition: oracle/apref.c apref_resample_affine_f32.
 * ------------------------------------------------------------------------------------------- */
#define APGPU_RESAMPLE_TILE_H 16
#define APGPU_RESAMPLE_TILE_W 64
int apgpu_resample_affine_f32(const float *frames, int32_t n_frames, int64_t h_in, int64_t w_in, const uint8_t *mask,
                              const double *affines, int32_t affines_per_tile, int32_t conserve_flux, const float *fscale,
                              const float *lut, int32_t n_phases, float *out, uint8_t *weight_out, int64_t h_out,
                              int64_t w_out, void *stream);

/* F3 (continued)  SWarp's OVERSAMPLING n (resample_all.sh:112, 339: 4) in one pass: every output pixel is the mean of n x n
 *     interpolations at the centres of its sub-pixels.  fine_affines maps pixel (u, v) of the n-times FINER grid - output
 *     pixel (x, y) owns u = n x .. n x + n - 1, v = n y .. n y + n - 1 - to input coordinates (ops.oversampled_affines, or
 *     wcs.tile_affines of the n-times finer WCS); with affines_per_tile != 0 there is one transform per TILE_H x TILE_W tile
 *     of the OUTPUT grid, [n_frames][tiles_y][tiles_x][6].  Each sub-sample is apgpu_resample_affine_f32's value at (u, v)
 *     (flux scale, conserve_flux = |det| of the fine transform, NaN rules included); out = (float)(sum * (1.0 / n^2)) with the
 *     sum accumulated in float64 in row-major order (v outer, u inner) - bit-identical to apgpu_block_mean_f32 of the fine
 *     resample, without the n^2-times larger image ever existing.  oversampling 1..16 (1 = apgpu_resample_affine_f32).
 *     Exact definition: oracle/apref.c apref_resample_oversampled_f32. */
int apgpu_resample_oversampled_f32(const float *frames, int32_t n_frames, int64_t h_in, int64_t w_in, const uint8_t *mask,
                                   const double *fine_affines, int32_t affines_per_tile, int32_t conserve_flux, const float *fscale,
                                   const float *lut, int32_t n_phases, int32_t oversampling, float *out, uint8_t *weight_out,
                                   int64_t h_out, int64_t w_out, void *stream);

/* F3 (continued)  The two-step form of OVERSAMPLING, and COMBINE_TYPE WEIGHTED.
 *     apgpu_block_mean_f32: OVERSAMPLING n (resample_all.sh:112, 339: 4) - the frame is resampled onto a grid n times finer
 *     (apgpu_resample_affine_f32 with the transform of the sub-pixel centres) and every output pixel is the mean of its
 *     n x n sub-samples: fine [n*h_out, n*w_out] float32 -> out [h_out, w_out] float32, float64 accumulation in row-major
 *     order inside the block, NaN if any sub-sample is NaN.
 *     apgpu_weighted_mean_f32: COMBINE_TYPE WEIGHTED (resample_all.sh:65-68) with one weight per frame (SWarp with
 *     WEIGHT_TYPE NONE weighs a frame by the inverse variance of its scaled background noise): per pixel
 *     mean = sum_i w_i x_i / sum_i w_i over the frames whose resampled value is finite (float64 accumulation in frame
 *     order), wsum_out = that sum of weights = the -WEIGHTOUT_NAME image (resample_all.sh:342); NaN / 0 where no frame
 *     contributes.  weights [n_frames] float32 on the device, all finite and > 0. */
int apgpu_block_mean_f32(const float *fine, int64_t h_out, int64_t w_out, int32_t oversampling, float *out, void *stream);

/* F3 + A7 in one launch: the co-add of scripts/resample_all.sh:330-342 (one SWarp call: resample every frame, combine) without
 * the resampled slab.  Every output pixel's N values are apgpu_resample_affine_f32's values for it (same tile records, same window
 * arithmetic, bit for bit; oversampling 1) and their reduction is apgpu_stack_sigclip's lean one (mean / count / moments, centre
 * median or mean, deviation std): NaN ("frame absent": a window off the frame, on a bad pixel or on a non-finite value) is
 * skipped as there.  `args` is the stack's argument block with: frames = the INPUT frames [n_frames][h_in * w_in] float32
 * (frame_stride 0 = h_in * w_in), n_frames <= 16 per call, n_pixels = h_out * w_out, no fused calibration, no pixmask, outputs
 * among mean / count / moments, flags among APGPU_STACK_EXACT_MOMENTS / APGPU_STACK_MOMENTS_MEAN; its workspace fields are not
 * used.  The other arguments are apgpu_resample_affine_f32's.  workspace: apgpu_resample_stack_ws_bytes(..) bytes, 64-byte
 * aligned, caller-owned, need not be initialised: tile records, and with a mask the bad-pixel list and one uint32 per output
 * pixel (which frames' windows hold a bad pixel).  HBM traffic 4 n_frames P + 4 P instead of 12 n_frames P + 4 P. */
size_t apgpu_resample_stack_ws_bytes(int32_t n_frames, int64_t h_in, int64_t w_in, int64_t h_out, int64_t w_out, int32_t has_mask);
int apgpu_resample_stack_sigclip(const apgpu_stack_args *args, int64_t h_in, int64_t w_in, const uint8_t *mask,
                                 const double *affines, int32_t affines_per_tile, int32_t conserve_flux, const float *fscale,
                                 const float *lut, int32_t n_phases, int64_t h_out, int64_t w_out, void *workspace,
                                 size_t workspace_bytes, void *stream);
int apgpu_weighted_mean_f32(const float *slab, int32_t n_frames, int64_t n_pixels, const float *weights, float *mean_out,
                            float *wsum_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * F4  Sky-background mesh of ApMeasureBackground (core/ApMeasureBackground.py:142-175, 382-415).  The reference calls
 *     photutils (detect_threshold / detect_sources / make_source_mask, Background2D with MedianBackground + SigmaClip,
 *     BkgZoomInterpolator); photutils is not in the build container, so these entry points implement its published
 *     algorithms as restated in oracle/background_ref.py (parity with photutils itself is unpinned).
 *
 *     apgpu_source_mask_u8: `above` [H][W] uint8 (non-zero = pixel above the detection threshold, e.g. from
 *       apgpu_threshold_mask_f32) -> 8-connected components, components with fewer than min_pixels pixels dropped
 *       (detect_sources(npixels)), the rest dilated with a dilate_size x dilate_size square
 *       (SegmentationImage.make_source_mask(size)); mask_out [H][W] uint8 0/1; nsources_out[0] (device int64, may be
 *       NULL) = surviving components.
 *     apgpu_box_clipped_stats_f32: the image is cut into box_height x box_width boxes (the last row / column of boxes
 *       may stick out of the image: those pixels count as masked, Background2D's edge_method 'pad'); per box astropy's
 *       SigmaClip(sigma, maxiters, median / std) over the unmasked finite pixels; stats_out[ny][nx][4] (device float64)
 *       = { median, std, count of the final survivors, pixels masked before clipping }.
 *     apgpu_spline_zoom_f64: scipy.ndimage.zoom(mesh, (zoom_y, zoom_x), order=3, mode='reflect', grid_mode=True)[:H, :W]
 *       from the prefiltered cubic B-spline coefficients coef[ny][nx] (device float64), clipped to [vmin, vmax].
 * ------------------------------------------------------------------------------------------- */
size_t apgpu_source_mask_ws_bytes(int64_t height, int64_t width);
int apgpu_source_mask_u8(const uint8_t *above, int64_t height, int64_t width, int32_t min_pixels, int32_t dilate_size,
                         uint8_t *mask_out, int64_t *nsources_out, void *ws, size_t ws_bytes, void *stream);
int apgpu_box_clipped_stats_f32(const float *data, const uint8_t *mask, int64_t height, int64_t width, int32_t box_height,
                                int32_t box_width, double sigma, int32_t maxiters, double *stats_out, void *stream);
int apgpu_spline_zoom_f64(const double *coef, int32_t ny, int32_t nx, int32_t zoom_y, int32_t zoom_x, int64_t height,
                          int64_t width, double vmin, double vmax, double *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * F4  L.A.Cosmic, the step ApFixCosmicRays.process hands to ccdproc.cosmicray_lacosmic -> astroscrappy.detect_cosmics
 *     (core/ApFixCosmicRays.py:267-295).  Neither package is in the build container: the entry points implement the
 *     algorithm as restated in oracle/lacosmic_ref.py (parity with astroscrappy itself is unpinned).  Images are float32 in
 *     ELECTRONS (the caller multiplies by the gain, as ccdproc does with gain_apply = True).
 *     apgpu_sepmedfilt_f32: median of `size` (5, 7 or 9) along rows, then along columns, borders copied (astroscrappy's
 *       separable median filters); ws >= 4 * H * W bytes.
 *     apgpu_lacosmic_satmask: astroscrappy's update_mask - cores of saturated stars (data >= satlevel where the 7-median
 *       exceeds satlevel / 10) dilated twice with the 5 x 5 kernel, OR inmask (may be NULL) grown by one pixel.
 *     apgpu_lacosmic_iterate: ONE detect_cosmics iteration, in place on clean / crmask: Laplacian of the 2x subsampled
 *       image, noise model sqrt(max(sepmed7, 1e-5) + readnoise^2), fine-structure image from the 7 x 7 kernel psfk (device,
 *       49 floats; NULL = fsmode 'median'), sp > sigclip and sp / f > objlim, two 3 x 3 growth steps (sp > sigclip, then
 *       sp > sigfrac * sigclip), 'meanmask' cleaning over 5 x 5 with background_level where no good neighbour exists;
 *       ncr_out[0] (device int64) = cosmic-ray pixels found in this iteration (the caller stops at 0).
 *     ws >= apgpu_lacosmic_ws_bytes(H, W) for the last two.
 * ------------------------------------------------------------------------------------------- */
size_t apgpu_lacosmic_ws_bytes(int64_t height, int64_t width);
int apgpu_sepmedfilt_f32(const float *data, int64_t height, int64_t width, int32_t size, float *out, void *ws, size_t ws_bytes,
                         void *stream);
int apgpu_lacosmic_satmask(const float *data, const uint8_t *inmask, int64_t height, int64_t width, float satlevel,
                           uint8_t *mask_out, void *ws, size_t ws_bytes, void *stream);
int apgpu_lacosmic_iterate(float *clean, const uint8_t *mask, uint8_t *crmask, int64_t height, int64_t width, float sigclip,
                           float sigfrac, float objlim, float readnoise, const float *psfk, float background_level,
                           int64_t *ncr_out, void *ws, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* APGPU_H */
