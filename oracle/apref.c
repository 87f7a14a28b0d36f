/*
 * apref.c - CPU ORACLE for the calibrate-and-stack hot path.   *** TEST INFRASTRUCTURE ONLY ***
 *
 * A plain-C restatement of the arithmetic the reference (DaveStrickland/AstroPhotography v0.5.1,
 * pure NumPy/astropy) performs on the path named in BASELINE.json:north_star.  It exists so the
 * HIP kernels can be checked against something that (a) runs without astropy/the reference on the
 * GPU box and (b) is itself pinned: tests/test_oracle_golden.py compares every function here with
 * golden vectors captured from the imported reference (tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (astrophotography_amd/) never imports it.
 *
 * Parity status: A1-A5, A7, A8 pinned by golden vectors generated from the reference + astropy 4.3.1
 * + numpy 1.26.4.  A6 (ccdproc.combine, not installed anywhere in the build container; pinned
 * version ccdproc>=2.1.0, requirements.txt:18) is "parity unpinned": apref_combine_ccdproc()
 * restates ccdproc's published Combiner.sigma_clipping + average_combine algorithm and is
 * anchored only on astropy's median / mad_std building blocks (golden group G6).  F3 (resample, the
 * reference's external SWarp step) is likewise "parity unpinned": apref_resample_affine_f32() is the
 * definition the HIP kernel is held to, checked only against closed-form properties.
 *
 * Citations "ref:" are relative to /root/reference/AstroPhotography/; "astropy:" refers to
 * astropy 4.3.1 (astropy/stats/sigma_clipping.py and its C helper src/compute_bounds.c /
 * src/wirth_select.c); "numpy:" to numpy/core/src/umath/loops_utils.h.src (pairwise summation).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------- */
/* numpy pairwise summation (numpy: loops_utils.h.src @TYPE@_pairwise_sum, PW_BLOCKSIZE = 128).   */
/* np.sum / np.mean / np.nanmean of a contiguous float32 array reduce with exactly this tree,     */
/* starting from the additive identity (0 + pairwise(a, n)).                                      */
/* ------------------------------------------------------------------------------------------- */
#define PW_BLOCKSIZE 128

static float pairwise_f32(const float *a, long n)
{
    if (n < 8) {
        float res = 0.f;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= PW_BLOCKSIZE) {
        float r[8], res;
        long i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_f32(a, n2) + pairwise_f32(a + n2, n - n2);
    }
}

static double pairwise_f64(const double *a, long n)
{
    if (n < 8) {
        double res = 0.;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= PW_BLOCKSIZE) {
        double r[8], res;
        long i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_f64(a, n2) + pairwise_f64(a + n2, n - n2);
    }
}

/* np.add.reduce over a contiguous array: the ufunc machinery feeds the inner loop in buffer-sized
 * pieces of NPY_BUFSIZE = 8192 elements, each reduced pairwise and accumulated sequentially into the
 * running total that starts at the identity 0 (verified against numpy 1.26.4 and 2.2.6: golden G7). */
#define NPY_BUFSIZE 8192

static float npsum_f32(const float *a, long n)
{
    float res = 0.f;
    for (long i = 0; i < n; i += NPY_BUFSIZE) {
        long m = n - i < NPY_BUFSIZE ? n - i : NPY_BUFSIZE;
        res = res + pairwise_f32(a + i, m);
    }
    return res;
}

static double npsum_f64(const double *a, long n)
{
    double res = 0.;
    for (long i = 0; i < n; i += NPY_BUFSIZE) {
        long m = n - i < NPY_BUFSIZE ? n - i : NPY_BUFSIZE;
        res = res + pairwise_f64(a + i, m);
    }
    return res;
}

float apref_pairwise_sum_f32(const float *a, long n) { return npsum_f32(a, n); }
double apref_pairwise_sum_f64(const double *a, long n) { return npsum_f64(a, n); }

/* ------------------------------------------------------------------------------------------- */
/* A1  ApCalibrate._generate_flat  (ref: core/ApCalibrate.py:166-190)                            */
/*   norm = np.nanmean(flat)   -> NaNs replaced by 0, float32 pairwise sum, divided by the count  */
/*          of non-NaN values; numpy 1.26 evaluates float32_scalar / int in float64 and casts the */
/*          result back to float32 (numpy/_core/_methods.py _mean / nanfunctions _divide_by_count)*/
/*   nflat = flat / norm       -> float32 division                                               */
/* ------------------------------------------------------------------------------------------- */
int apref_flat_normalize_f32(const float *flat, long n, float *nflat, float *norm_out)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    if (!tmp) return -1;
    long cnt = 0;
    for (long i = 0; i < n; i++) {
        if (isnan(flat[i])) tmp[i] = 0.f; else { tmp[i] = flat[i]; cnt++; }
    }
    float tot = npsum_f32(tmp, n);
    free(tmp);
    float norm = (float)((double)tot / (double)cnt);      /* cnt == 0 -> nan (numpy warns) */
    if (norm_out) *norm_out = norm;
    if (nflat)
        for (long i = 0; i < n; i++) nflat[i] = flat[i] / norm;
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* A2  ApCalibrate.calibrate arithmetic block (ref: core/ApCalibrate.py:439-464) with the read-   */
/*     time conversions of _read_fits (ref: core/ApCalibrate.py:304-326):                         */
/*       raw u16 -> float32 (exact); PEDESTAL != 0 is ADDED in float32                            */
/*       x  = raw - bias                         (float32)                                        */
/*       D  = dark - bias  if dark_still_biased else dark                                         */
/*       ds = float32(exp_ratio) * D             (python float is a weak scalar vs a f32 array)   */
/*       x  = x - ds                                                                              */
/*       y  = where(nflat != 0, x / nflat, x)    (no flat: y = x)                                 */
/*     Every operation is separately rounded to float32 (compile with -ffp-contract=off).         */
/* ------------------------------------------------------------------------------------------- */
static inline float calib_one(float raw, float ped, int has_ped, float b, float d, int still_biased,
                              float e, const float *nflat_p)
{
    volatile float r = raw;
    if (has_ped) r = r + ped;
    volatile float x = r - b;
    volatile float D = still_biased ? (d - b) : d;
    volatile float ds = e * D;
    volatile float y = x - ds;
    if (nflat_p) {
        float nf = *nflat_p;
        if (nf != 0.f) { volatile float q = y / nf; return q; }   /* NaN != 0 is true -> NaN result */
    }
    return y;
}

/* raw_dtype: 0 = float32, 1 = uint16.  frames laid out [N][P]; masters [P]; e[N], pedestal[N]
 * (pedestal 0 means "no PEDESTAL keyword / zero pedestal": nothing is added).  nflat may be NULL. */
int apref_calibrate(const void *raw, int raw_dtype, const float *bias, const float *dark,
                    const float *nflat, const float *e, const float *pedestal, int dark_still_biased,
                    float *out, long N, long P)
{
    if (raw_dtype != 0 && raw_dtype != 1) return -2;
#pragma omp parallel for schedule(static)
    for (long f = 0; f < N; f++) {
        float ped = pedestal ? pedestal[f] : 0.f;
        int has_ped = (ped != 0.f);
        for (long p = 0; p < P; p++) {
            float r = raw_dtype == 0 ? ((const float *)raw)[f * P + p]
                                     : (float)((const uint16_t *)raw)[f * P + p];
            out[f * P + p] = calib_one(r, ped, has_ped, bias[p], dark[p], dark_still_biased, e[f],
                                       nflat ? &nflat[p] : NULL);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* A1/A2 with float64 inputs (golden group G11).  ApCalibrate._read_fits converts only NON-float  */
/* data to float32 (ref: core/ApCalibrate.py:301-305) and ApMasterCal writes float64 masters      */
/* (ref: scripts/ap_combine_darks.py:437), so the reference's NumPy expressions promote per       */
/* operation: result type = float64 if either operand array is float64, else float32; the python  */
/* float exp_ratio is a weak scalar (takes the type of the dark array it multiplies).             */
/*   T_raw: u16 -> float32 at read; PEDESTAL is added in T_raw (in-place +=, :318-326)            */
/*   x  = raw - bias               in T1 = promote(T_raw, T_bias)             (:439)              */
/*   D  = dark - bias | dark       in T2 = promote(T_dark, T_bias) | T_dark   (:440-445)          */
/*   ds = e * D                    in T2                                      (:450)              */
/*   y  = x - ds                   in T3 = promote(T1, T2)                    (:451)              */
/*   out= where(nf != 0, y / nf, y) in T4 = promote(T3, T_flat)               (:462-464)          */
/* Each operation is evaluated in double and, where NumPy's result type is float32, rounded to    */
/* float32: for + - * / of float32 operands this double rounding is exact (53 >= 2*24 + 2,        */
/* Figueroa 1995), so the float32-only case reproduces apref_calibrate bit for bit (tested).      */
/* dtype tags: 0 float32, 1 uint16 (raw only), 2 float64.  out is float64 if out_f64 else float32;*/
/* out_f64 must equal "T4 is float64" (returns -3 otherwise).                                     */
/* ------------------------------------------------------------------------------------------- */
static inline double rnd_to(double x, int is64)
{
    if (is64) return x;
    volatile float f = (float)x;
    return (double)f;
}

static inline double load_px(const void *a, int dt, long i)
{
    if (dt == 0) return (double)((const float *)a)[i];
    if (dt == 1) return (double)((const uint16_t *)a)[i];
    return ((const double *)a)[i];
}

int apref_calibrate_mixed(const void *raw, int raw_dt, const void *bias, int bias_dt, const void *dark, int dark_dt,
                          const void *nflat, int nflat_dt, const double *e, const double *pedestal,
                          int dark_still_biased, void *out, int out_f64, long N, long P)
{
    const int r64 = raw_dt == 2, b64 = bias_dt == 2, d64 = dark_dt == 2, n64 = nflat_dt == 2;
    const int t1 = r64 || b64;
    const int t2 = dark_still_biased ? (d64 || b64) : d64;
    const int t3 = t1 || t2;
    const int t4 = nflat ? (t3 || n64) : t3;
    if ((out_f64 != 0) != (t4 != 0)) return -3;
#pragma omp parallel for schedule(static)
    for (long f = 0; f < N; f++) {
        const double ped = pedestal ? pedestal[f] : 0.0;
        const double ef = rnd_to(e[f], t2);                    /* weak python scalar: cast to the array's type */
        for (long p = 0; p < P; p++) {
            double r = load_px(raw, raw_dt, f * P + p);
            if (ped != 0.0) r = rnd_to(r + rnd_to(ped, r64), r64);
            const double b = load_px(bias, bias_dt, p), d = load_px(dark, dark_dt, p);
            const double x = rnd_to(r - b, t1);
            const double D = dark_still_biased ? rnd_to(d - b, t2) : d;
            const double ds = rnd_to(ef * D, t2);
            double y = rnd_to(x - ds, t3);
            if (nflat) {
                const double nf = load_px(nflat, nflat_dt, p);
                if (nf != 0.0) y = rnd_to(y / nf, t4);           /* NaN != 0 is true -> NaN */
            }
            if (out_f64) ((double *)out)[f * P + p] = y;
            else ((float *)out)[f * P + p] = (float)y;
        }
    }
    return 0;
}

/* _generate_flat for a float64 flat: np.nanmean(float64) = float64 pairwise sum in 8192-element pieces (NaN -> 0) / count;
 * nflat = flat / norm in float64 (pinned by golden G11 s*_nanmean and f*_nflat). */
int apref_flat_normalize_f64(const double *flat, long n, double *nflat, double *norm_out)
{
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!tmp) return -1;
    long cnt = 0;
    for (long i = 0; i < n; i++) {
        if (isnan(flat[i])) tmp[i] = 0.; else { tmp[i] = flat[i]; cnt++; }
    }
    double tot = npsum_f64(tmp, n);
    free(tmp);
    double norm = tot / (double)cnt;
    if (norm_out) *norm_out = norm;
    if (nflat)
        for (long i = 0; i < n; i++) nflat[i] = flat[i] / norm;
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Exact selection helpers                                                                       */
/* ------------------------------------------------------------------------------------------- */
/* astropy: src/wirth_select.c  kth_smallest / wirth_median (in-place, permutes the buffer).     */
static double kth_smallest(double *a, int n, int k)
{
    int i, j, l = 0, m = n - 1;
    double x;
    while (l < m) {
        x = a[k];
        i = l;
        j = m;
        do {
            while (a[i] < x) i++;
            while (x < a[j]) j--;
            if (i <= j) {
                double t = a[i]; a[i] = a[j]; a[j] = t;
                i++; j--;
            }
        } while (i <= j);
        if (j < k) l = i;
        if (k < i) m = j;
    }
    return a[k];
}

static double wirth_median(double *a, int n)
{
    /* An empty buffer: astropy's wirth_median reads a[0] and a[-1] here - outside the data, undefined.  Defined for this
       build (oracle and kernels alike) as NaN, which is what the NumPy path gives for an empty slice: the bounds become
       NaN, nothing is masked, and every finite value of the pixel survives - the same outcome the C loop reaches by
       itself for stdfunc='std' (0/0 mean and variance). */
    if (n <= 0) return NAN;
    if (n % 2 == 0)
        return 0.5 * (kth_smallest(a, n, n / 2) + kth_smallest(a, n, n / 2 - 1));
    return kth_smallest(a, n, (n - 1) / 2);
}

/* ------------------------------------------------------------------------------------------- */
/* A7  astropy.stats.sigma_clipped_stats(cube, axis=0) - the C fast path                          */
/*     (astropy: sigma_clipping.py:298-383 _sigmaclip_fast, src/compute_bounds.c, then            */
/*      sigma_clipping.py:924-937 nanmean / nanmedian / nanstd of the NaN-masked float64 copy).   */
/* ------------------------------------------------------------------------------------------- */
static void compute_sigma_clipped_bounds(double *buffer, int count, int use_median, int use_mad_std,
                                         int maxiters, double sigma_lower, double sigma_upper,
                                         double *lower_bound, double *upper_bound, double *mad_buffer)
{
    double mean = 0, std, median = 0;
    int i, new_count, iteration = 0;
    while (1) {
        if (use_median || use_mad_std) median = wirth_median(buffer, count);
        if (!use_median || !use_mad_std) {
            mean = 0;
            for (i = 0; i < count; i++) mean += buffer[i];
            mean /= count;
        }
        if (use_mad_std) {
            for (i = 0; i < count; i++) mad_buffer[i] = fabs(buffer[i] - median);
            std = wirth_median(mad_buffer, count) * 1.482602218505602;
        } else {
            std = 0;
            for (i = 0; i < count; i++) std += pow(mean - buffer[i], 2);
            std = sqrt(std / count);
        }
        if (use_median) {
            *lower_bound = median - sigma_lower * std;
            *upper_bound = median + sigma_upper * std;
        } else {
            *lower_bound = mean - sigma_lower * std;
            *upper_bound = mean + sigma_upper * std;
        }
        new_count = 0;
        for (i = 0; i < count; i++) {
            if (buffer[i] >= *lower_bound && buffer[i] <= *upper_bound) {
                buffer[new_count] = buffer[i];
                new_count += 1;
            }
        }
        if (new_count == count) return;
        count = new_count;
        iteration += 1;
        if (maxiters != -1 && iteration >= maxiters) return;
    }
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/*
 * frames: [N][P] float32 (dtype 0) or float64 (dtype 2; astropy converts integer cubes with
 *         astype(float) before the C loop, so u16 callers pass an exactly-converted f64 cube).
 * pixmask: optional [P] u8, non-zero = skip pixel entirely (outputs NaN, count 0) - build extension
 *          used for config 5 "with bad-pixel mask"; NULL reproduces astropy exactly.
 * maxiters: -1 = iterate to convergence (astropy maxiters=None).
 * Outputs (any may be NULL): mean/median/std [P] float64 (np.nanmean / np.nanmedian / np.nanstd
 * along N of the clipped copy, sequential float64 accumulation in frame order), lo/hi [P] float64
 * clipping bounds (NaN when no finite value), count [P] int32 survivors, keep [N][P] u8.
 */
int apref_stack_sigclip(const void *frames, int dtype, long N, long P, const uint8_t *pixmask,
                        double sigma_lower, double sigma_upper, int maxiters, int use_median,
                        int use_mad_std, double *mean_out, double *median_out, double *std_out,
                        double *lo_out, double *hi_out, int32_t *count_out, uint8_t *keep_out)
{
    if (dtype != 0 && dtype != 2) return -2;
    if (N <= 0 || N > (1 << 20)) return -3;
#pragma omp parallel
    {
        double *buffer = (double *)malloc(sizeof(double) * (size_t)N * 3);
        double *mad_buffer = buffer + N;
        double *col = buffer + 2 * N;
#pragma omp for schedule(static)
        for (long p = 0; p < P; p++) {
            int count = 0;
            for (long f = 0; f < N; f++) {
                double v = dtype == 0 ? (double)((const float *)frames)[f * P + p]
                                      : ((const double *)frames)[f * P + p];
                col[f] = v;
                if (isfinite(v)) buffer[count++] = v;
            }
            double lo = NAN, hi = NAN;
            int skip = pixmask && pixmask[p];
            if (count > 0 && !skip)
                compute_sigma_clipped_bounds(buffer, count, use_median, use_mad_std, maxiters,
                                             sigma_lower, sigma_upper, &lo, &hi, mad_buffer);
            /* mask |= ~isfinite; mask |= data < lo; mask |= data > hi  (comparisons with NaN
             * bounds are False, so a pixel with no finite value keeps nothing because all its
             * values are non-finite).  sigma_clipping.py:356-358 */
            int n = 0;
            double sum = 0.0;
            for (long f = 0; f < N; f++) {
                int keep = isfinite(col[f]) && !(col[f] < lo) && !(col[f] > hi) && !skip;
                if (keep_out) keep_out[f * P + p] = (uint8_t)keep;
                if (keep) { sum += col[f]; buffer[n++] = col[f]; }
            }
            double mean = n > 0 ? sum / n : NAN;
            if (mean_out) mean_out[p] = mean;
            if (count_out) count_out[p] = n;
            if (lo_out) lo_out[p] = lo;
            if (hi_out) hi_out[p] = hi;
            if (std_out) {
                /* numpy _nanvar: arr - avg, squared, summed along N in order, / cnt, sqrt */
                double q = 0.0;
                for (int i = 0; i < n; i++) { double d = buffer[i] - mean; q += d * d; }
                std_out[p] = n > 0 ? sqrt(q / n) : NAN;
            }
            if (median_out) {
                if (n == 0) median_out[p] = NAN;
                else {
                    qsort(buffer, (size_t)n, sizeof(double), cmp_double);
                    /* np.nanmedian -> np.median: mean of the two middle values, in float64:
                     * np.mean([a, b]) = (0 + a + b) / 2 */
                    median_out[p] = (n & 1) ? buffer[n / 2] : (buffer[n / 2 - 1] + buffer[n / 2]) / 2.0;
                }
            }
        }
        free(buffer);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* A6  ccdproc.combine as configured at ref: scripts/ap_combine_darks.py:394-420                  */
/*     (method='average', sigma_clip=True, low=high=5, func=np.ma.median, dev_func=mad_std).      */
/*     PARITY UNPINNED - ccdproc (>=2.1.0) is not installed; restated from its published          */
/*     Combiner: data_arr float64 masked cube; ONE pass: base = ma.median over N,                 */
/*     dev = mad_std = 1.482602218505602 * median(|x - median(x)|) (astropy/stats/funcs.py);      */
/*     mask where (x - base) < -low*dev or (x - base) > high*dev (strict); result = masked mean.  */
/*     Non-finite inputs are masked first (ccdproc masks NaN via np.ma.masked_invalid upstream    */
/*     only if the CCDData carries a mask; here: NaN/Inf are treated as masked).                  */
/* ------------------------------------------------------------------------------------------- */
int apref_combine_ccdproc_form(const void *frames, int dtype, long N, long P, double low, double high, int form,
                               double *mean_out, int32_t *count_out, double *std_out)
{
    if (dtype != 0 && dtype != 2) return -2;
#pragma omp parallel
    {
        double *buf = (double *)malloc(sizeof(double) * (size_t)N * 3);
        double *dev = buf + N, *col = buf + 2 * N;
#pragma omp for schedule(static)
        for (long p = 0; p < P; p++) {
            int n = 0;
            for (long f = 0; f < N; f++) {
                double v = dtype == 0 ? (double)((const float *)frames)[f * P + p]
                                      : ((const double *)frames)[f * P + p];
                col[f] = v;
                if (isfinite(v)) buf[n++] = v;
            }
            if (n == 0) {
                if (mean_out) mean_out[p] = NAN;
                if (count_out) count_out[p] = 0;
                if (std_out) std_out[p] = NAN;
                continue;
            }
            qsort(buf, (size_t)n, sizeof(double), cmp_double);
            double base = (n & 1) ? buf[n / 2] : (buf[n / 2 - 1] + buf[n / 2]) / 2.0;
            for (int i = 0; i < n; i++) dev[i] = fabs(buf[i] - base);
            qsort(dev, (size_t)n, sizeof(double), cmp_double);
            double mad = (n & 1) ? dev[n / 2] : (dev[n / 2 - 1] + dev[n / 2]) / 2.0;
            double sd = mad * 1.482602218505602;
            double sum = 0;
            int m = 0;
            /* form 1 ("astropy", ccdproc >= 2.2): bounds as astropy's _compute_bounds forms them (sigma_clipping.py:288-296:
             * min = centre - std * sigma_lower, max = centre + std * sigma_upper), rejected where x < min or x > max; and the
             * general path hands np.ma.median / mad_std a plain array in which non-finite values are NaN, so a column holding
             * one gets NaN bounds: nothing is clipped (sigma_clipping.py:455-507).  form 0 ("legacy", ccdproc <= 2.1). */
            const double lob = base - sd * low, hib = base + sd * high;
            const int unclipped = form == 1 && n < N;
            for (long f = 0; f < N; f++) {
                double v = col[f];
                if (!isfinite(v)) continue;
                if (form == 1) {
                    if (!unclipped && (v < lob || v > hib)) continue;
                } else {
                    double d = v - base;
                    if (d < -low * sd || d > high * sd) continue;
                }
                sum += v; buf[m++] = v;
            }
            double mean = m > 0 ? sum / m : NAN;
            if (mean_out) mean_out[p] = mean;
            if (count_out) count_out[p] = m;
            if (std_out) {
                double q = 0;
                for (int i = 0; i < m; i++) { double d = buf[i] - mean; q += d * d; }
                std_out[p] = m > 0 ? sqrt(q / m) : NAN;
            }
        }
        free(buf);
    }
    return 0;
}

/* The form ApMasterCal uses by default: "astropy" (what ccdproc>=2.1.0, requirements.txt:18, resolves to today). */
int apref_combine_ccdproc(const void *frames, int dtype, long N, long P, double low, double high,
                          double *mean_out, int32_t *count_out, double *std_out)
{
    return apref_combine_ccdproc_form(frames, dtype, N, P, low, high, 1, mean_out, count_out, std_out);
}

/* Plain median along N (np.median / np.nanmedian of the float64-converted cube; config 4). */
int apref_stack_median(const void *frames, int dtype, long N, long P, double *median_out)
{
    if (dtype != 0 && dtype != 2) return -2;
#pragma omp parallel
    {
        double *buf = (double *)malloc(sizeof(double) * (size_t)N);
#pragma omp for schedule(static)
        for (long p = 0; p < P; p++) {
            int n = 0;
            for (long f = 0; f < N; f++) {
                double v = dtype == 0 ? (double)((const float *)frames)[f * P + p]
                                      : ((const double *)frames)[f * P + p];
                if (!isnan(v)) buf[n++] = v;
            }
            if (n == 0) { median_out[p] = NAN; continue; }
            qsort(buf, (size_t)n, sizeof(double), cmp_double);
            median_out[p] = (n & 1) ? buf[n / 2] : (buf[n / 2 - 1] + buf[n / 2]) / 2.0;
        }
        free(buf);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* A3  astropy.stats.sigma_clipped_stats(data, sigma=s) with axis=None as called at               */
/*     ref: core/ApFindBadPixels.py:191  (astropy: sigma_clipping.py:385-433 _sigmaclip_noaxis    */
/*     with numpy nan-functions because the reference environment has no bottleneck).             */
/*     float32 input keeps float32 arithmetic throughout (numpy semantics):                       */
/*       median: exact order statistic; even count -> float32(a + b) / 2                          */
/*       std   : mean = f32(pairwise_f32(x) / n); d = x - mean; f32 pairwise sum of d*d;          */
/*               var = f32(f64(sum) / n); std = sqrtf(var)                                        */
/*       keep  : lo <= x <= hi with lo = med - std*sigma, hi = med + std*sigma computed as        */
/*               numpy-1.26 scalar arithmetic: np.float32 * python float -> float64, compared     */
/*               against the float32 array after demotion of the float64 scalar to float32.       */
/*     Final stats: mean = f32(f64(pairwise_f32)/n), median, std as above, over the survivors.    */
/* ------------------------------------------------------------------------------------------- */
static int cmp_float(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

static float median_sorted_f32(const float *s, long n)
{
    if (n & 1) return s[n / 2];
    volatile float t = s[n / 2 - 1] + s[n / 2];   /* np.mean of 2 float32: f32 sum, then /2 */
    return (float)((double)t / 2.0);
}

static float mean_f32(const float *x, long n)
{
    float tot = npsum_f32(x, n);
    return (float)((double)tot / (double)n);
}

/* np.var on a float32 ndarray: arrmean = true_divide(umr_sum(arr, keepdims=True), n) as a float32
 * array op (f32 / f32(n)); x = arr - arrmean; x*x; ret = f32(f64(pairwise(x)) / n). */
static float std_f32(const float *x, long n, float *scratch)
{
    float tot = npsum_f32(x, n);
    volatile float mean = tot / (float)n;
    for (long i = 0; i < n; i++) { volatile float d = x[i] - mean; volatile float q = d * d; scratch[i] = q; }
    float s2 = npsum_f32(scratch, n);
    float var = (float)((double)s2 / (double)n);
    return sqrtf(var);
}

/* out[0..2] = mean, median, std (as float64 holding float32 values); out[3], out[4] = final clip
 * bounds of the last iteration (float64, as numpy 1.26 produces them); out[5] = iterations run;
 * out[6] = number of survivors. */
int apref_sigclip_global_f32(const float *data, long n, double sigma_lower, double sigma_upper,
                             int maxiters, double *out)
{
    float *x = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * 3);
    if (!x) return -1;
    float *srt = x + n, *scr = x + 2 * n;
    long m = 0;
    for (long i = 0; i < n; i++) if (isfinite(data[i])) x[m++] = data[i];
    int iteration = 0;
    long nchanged = 1;
    double lo = NAN, hi = NAN;
    while (nchanged != 0 && (maxiters < 0 || iteration < maxiters)) {
        iteration++;
        if (m == 0) { lo = hi = NAN; break; }
        memcpy(srt, x, sizeof(float) * (size_t)m);
        qsort(srt, (size_t)m, sizeof(float), cmp_float);
        float med = median_sorted_f32(srt, m);
        float sd = std_f32(x, m, scr);
        /* _compute_bounds: max = cen; min = max - std*sigma_lower; max += std*sigma_upper.
         * np.float32 * python float -> float64 under numpy 1.26 value-based scalar promotion. */
        lo = (double)med - (double)sd * sigma_lower;
        hi = (double)med + (double)sd * sigma_upper;
        /* array(float32) >= float64 scalar: the scalar is demoted to float32 */
        float lof = (float)lo, hif = (float)hi;
        long k = 0;
        for (long i = 0; i < m; i++) if (x[i] >= lof && x[i] <= hif) x[k++] = x[i];
        nchanged = m - k;
        m = k;
    }
    if (m > 0) {
        memcpy(srt, x, sizeof(float) * (size_t)m);
        qsort(srt, (size_t)m, sizeof(float), cmp_float);
        out[0] = mean_f32(x, m);
        out[1] = median_sorted_f32(srt, m);
        out[2] = std_f32(x, m, scr);
    } else {
        out[0] = out[1] = out[2] = NAN;
    }
    out[3] = lo; out[4] = hi; out[5] = iteration; out[6] = (double)m;
    free(x);
    return 0;
}

/* Same call on integer (u16) data: numpy computes every statistic in float64
 * (np.median -> np.mean of ints in f64; np.var with dtype promoted to f64 pairwise). */
int apref_sigclip_global_f64(const double *data, long n, double sigma_lower, double sigma_upper,
                             int maxiters, double *out)
{
    double *x = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * 3);
    if (!x) return -1;
    double *srt = x + n, *scr = x + 2 * n;
    long m = 0;
    for (long i = 0; i < n; i++) if (isfinite(data[i])) x[m++] = data[i];
    int iteration = 0;
    long nchanged = 1;
    double lo = NAN, hi = NAN;
    while (nchanged != 0 && (maxiters < 0 || iteration < maxiters)) {
        iteration++;
        if (m == 0) { lo = hi = NAN; break; }
        memcpy(srt, x, sizeof(double) * (size_t)m);
        qsort(srt, (size_t)m, sizeof(double), cmp_double);
        double med = (m & 1) ? srt[m / 2] : (srt[m / 2 - 1] + srt[m / 2]) / 2.0;
        double mean = npsum_f64(x, m) / (double)m;
        for (long i = 0; i < m; i++) { double d = x[i] - mean; scr[i] = d * d; }
        double sd = sqrt(npsum_f64(scr, m) / (double)m);
        lo = med - sd * sigma_lower;
        hi = med + sd * sigma_upper;
        long k = 0;
        for (long i = 0; i < m; i++) if (x[i] >= lo && x[i] <= hi) x[k++] = x[i];
        nchanged = m - k;
        m = k;
    }
    if (m > 0) {
        memcpy(srt, x, sizeof(double) * (size_t)m);
        qsort(srt, (size_t)m, sizeof(double), cmp_double);
        double mean = npsum_f64(x, m) / (double)m;
        for (long i = 0; i < m; i++) { double d = x[i] - mean; scr[i] = d * d; }
        out[0] = mean;
        out[1] = (m & 1) ? srt[m / 2] : (srt[m / 2 - 1] + srt[m / 2]) / 2.0;
        out[2] = sqrt(npsum_f64(scr, m) / (double)m);
    } else {
        out[0] = out[1] = out[2] = NAN;
    }
    out[3] = lo; out[4] = hi; out[5] = iteration; out[6] = (double)m;
    free(x);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* A4  ApFindBadPixels._generate_sigmaclip_mask (ref: core/ApFindBadPixels.py:194-216):           */
/*     mask = (data < lothresh) | (data > hithresh) as uint8; thresholds are float64 scalars that */
/*     numpy 1.26 demotes to float32 when compared against the float32 array (strict compares).   */
/* ------------------------------------------------------------------------------------------- */
long apref_threshold_mask_f32(const float *data, long n, double lothresh, double hithresh, uint8_t *mask)
{
    float lo = (float)lothresh, hi = (float)hithresh;
    long nbad = 0;
    for (long i = 0; i < n; i++) {
        uint8_t b = (data[i] < lo) || (data[i] > hi);
        mask[i] = b;
        nbad += b;
    }
    return nbad;
}

/* A4 user overlays (ref: core/ApFindBadPixels.py:70-158): mask[r0:r1, c0:c1] += value for
 * 0-based half-open rectangles (the 1-based inclusive -> 0-based conversion and the range checks
 * live in the host code).  uint8 wrap-around as numpy. */
void apref_mask_add_rects(uint8_t *mask, long H, long W, const int32_t *rects, long nrect, int value)
{
    (void)H;
    for (long k = 0; k < nrect; k++) {
        long r0 = rects[4 * k], r1 = rects[4 * k + 1], c0 = rects[4 * k + 2], c1 = rects[4 * k + 3];
        for (long r = r0; r < r1; r++)
            for (long c = c0; c < c1; c++) mask[r * W + c] = (uint8_t)(mask[r * W + c] + value);
    }
}

/* ------------------------------------------------------------------------------------------- */
/* A5  ApFixBadPixels.fix_bad_pixels (ref: core/ApFixBadPixels.py:292-445)                        */
/*     For each pixel with mask != 0: window [r-d, r+d] x [c-d, c+d] clipped to the image, values */
/*     and mask taken from the ORIGINAL arrays; if #good >= min_valid: new = np.median(good)      */
/*     (float32: odd -> middle; even -> float32(a+b)/2 via np.mean) else unchanged.               */
/*     stats[0] = nbad, stats[1] = nfixed, stats[2] = nremaining.                                 */
/* ------------------------------------------------------------------------------------------- */
int apref_fix_badpix_f32(const float *data, const uint8_t *mask, long H, long W, int deltapix,
                         int min_valid, float *out, int64_t *stats)
{
    memcpy(out, data, sizeof(float) * (size_t)(H * W));
    int64_t nbad = 0, nfix = 0;
    int wmax = (2 * deltapix + 1) * (2 * deltapix + 1);
    float *good = (float *)malloc(sizeof(float) * (size_t)wmax);
    for (long r = 0; r < H; r++)
        for (long c = 0; c < W; c++) {
            if (!mask[r * W + c]) continue;
            nbad++;
            long rmin = r - deltapix < 0 ? 0 : r - deltapix;
            long rmax = r + deltapix + 1 > H ? H : r + deltapix + 1;
            long cmin = c - deltapix < 0 ? 0 : c - deltapix;
            long cmax = c + deltapix + 1 > W ? W : c + deltapix + 1;
            int ng = 0;
            for (long rr = rmin; rr < rmax; rr++)
                for (long cc = cmin; cc < cmax; cc++)
                    if (!mask[rr * W + cc]) good[ng++] = data[rr * W + cc];
            if (ng >= min_valid) {
                /* np.median sorts NaN last and returns NaN if any NaN is present */
                int has_nan = 0;
                for (int i = 0; i < ng; i++) if (isnan(good[i])) has_nan = 1;
                if (has_nan) out[r * W + c] = NAN;
                else {
                    qsort(good, (size_t)ng, sizeof(float), cmp_float);
                    out[r * W + c] = median_sorted_f32(good, ng);
                }
                nfix++;
            }
        }
    free(good);
    if (stats) { stats[0] = nbad; stats[1] = nfix; stats[2] = nbad - nfix; }
    return 0;
}

/* A5 on a float64 image (what fix_bad_pixels sees after a float64 calibration): the same window logic, np.median in
 * float64 (even count: (a + b) / 2 in float64).  Pinned by golden G11 (cases with use_mask). */
static int cmp_double_nan_last(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

int apref_fix_badpix_f64(const double *data, const uint8_t *mask, long H, long W, int deltapix,
                         int min_valid, double *out, int64_t *stats)
{
    memcpy(out, data, sizeof(double) * (size_t)(H * W));
    int64_t nbad = 0, nfix = 0;
    int wmax = (2 * deltapix + 1) * (2 * deltapix + 1);
    double *good = (double *)malloc(sizeof(double) * (size_t)wmax);
    for (long r = 0; r < H; r++)
        for (long c = 0; c < W; c++) {
            if (!mask[r * W + c]) continue;
            nbad++;
            long rmin = r - deltapix < 0 ? 0 : r - deltapix;
            long rmax = r + deltapix + 1 > H ? H : r + deltapix + 1;
            long cmin = c - deltapix < 0 ? 0 : c - deltapix;
            long cmax = c + deltapix + 1 > W ? W : c + deltapix + 1;
            int ng = 0;
            for (long rr = rmin; rr < rmax; rr++)
                for (long cc = cmin; cc < cmax; cc++)
                    if (!mask[rr * W + cc]) good[ng++] = data[rr * W + cc];
            if (ng >= min_valid) {
                int has_nan = 0;
                for (int i = 0; i < ng; i++) if (isnan(good[i])) has_nan = 1;
                if (has_nan) out[r * W + c] = NAN;
                else {
                    qsort(good, (size_t)ng, sizeof(double), cmp_double_nan_last);
                    out[r * W + c] = (ng & 1) ? good[ng / 2] : (good[ng / 2 - 1] + good[ng / 2]) / 2.0;
                }
                nfix++;
            }
        }
    free(good);
    if (stats) { stats[0] = nbad; stats[1] = nfix; stats[2] = nbad - nfix; }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* A8  ApImArith.process_files op block (ref: core/ApImArith.py:320-333)                          */
/*     op: 0 ADD, 1 SUB, 2 MUL, 3 DIV.  float32 (+) float32 array or float32(scalar).             */
/*     uint16 (+) uint16 array ADD/SUB/MUL wrap modulo 2^16 (numpy integer ufuncs).               */
/* ------------------------------------------------------------------------------------------- */
int apref_imarith_f32(const float *a, const float *b, double scalar, int use_scalar, int op, float *out, long n)
{
    float s = (float)scalar;
    for (long i = 0; i < n; i++) {
        float y = use_scalar ? s : b[i];
        volatile float r;
        switch (op) {
        case 0: r = a[i] + y; break;
        case 1: r = a[i] - y; break;
        case 2: r = a[i] * y; break;
        case 3: r = a[i] / y; break;
        default: return -2;
        }
        out[i] = r;
    }
    return 0;
}

int apref_imarith_u16(const uint16_t *a, const uint16_t *b, int op, uint16_t *out, long n)
{
    for (long i = 0; i < n; i++) {
        switch (op) {
        case 0: out[i] = (uint16_t)(a[i] + b[i]); break;
        case 1: out[i] = (uint16_t)(a[i] - b[i]); break;
        case 2: out[i] = (uint16_t)((uint32_t)a[i] * (uint32_t)b[i]); break;
        default: return -2;     /* DIV on integers raises UFuncTypeError in the reference */
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* A9  RawConv split geometry (ref: core/RawConv.py:111-128): four FULL-SIZE planes               */
/*     plane[k] = where(color_map == k, raw, 0), k = R 0, G1 1, B 2, G2 3; pattern[4] gives the   */
/*     colour index of the 2x2 cell positions (0,0) (0,1) (1,0) (1,1); optional black-level       */
/*     subtraction clamped at zero (ref: core/RawConv.py:163-190).                                */
/* ------------------------------------------------------------------------------------------- */
int apref_bayer_split_u16(const uint16_t *raw, long H, long W, const int32_t *pattern,
                          const int32_t *black, uint16_t *planes)
{
    memset(planes, 0, sizeof(uint16_t) * (size_t)(4 * H * W));
    for (long r = 0; r < H; r++)
        for (long c = 0; c < W; c++) {
            int k = pattern[(r & 1) * 2 + (c & 1)];
            int v = raw[r * W + c];
            if (black) { v -= black[k]; if (v < 0) v = 0; }
            planes[(long)k * H * W + r * W + c] = (uint16_t)v;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Fused pipeline for the CPU baseline (bench.py cpu_baseline leg): per-frame calibrate (A2) then */
/* the sigma-clipped mean along N (A7), blocked over pixels so the [N][block] working set stays   */
/* in cache.  Identical arithmetic to apref_calibrate + apref_stack_sigclip.                      */
/* ------------------------------------------------------------------------------------------- */
int apref_calibrate_stack(const void *raw, int raw_dtype, const float *bias, const float *dark,
                          const float *nflat, const float *e, const float *pedestal,
                          int dark_still_biased, long N, long P, double sigma_lower, double sigma_upper,
                          int maxiters, int use_median, int use_mad_std, float *mean_out_f32,
                          int32_t *count_out)
{
    if (raw_dtype != 0 && raw_dtype != 1) return -2;
#pragma omp parallel
    {
        double *buffer = (double *)malloc(sizeof(double) * (size_t)N * 3);
        double *mad_buffer = buffer + N, *col = buffer + 2 * N;
#pragma omp for schedule(static)
        for (long p = 0; p < P; p++) {
            int count = 0;
            for (long f = 0; f < N; f++) {
                float ped = pedestal ? pedestal[f] : 0.f;
                float r = raw_dtype == 0 ? ((const float *)raw)[f * P + p]
                                         : (float)((const uint16_t *)raw)[f * P + p];
                double v = (double)calib_one(r, ped, ped != 0.f, bias[p], dark[p], dark_still_biased,
                                             e[f], nflat ? &nflat[p] : NULL);
                col[f] = v;
                if (isfinite(v)) buffer[count++] = v;
            }
            double lo = NAN, hi = NAN;
            if (count > 0)
                compute_sigma_clipped_bounds(buffer, count, use_median, use_mad_std, maxiters,
                                             sigma_lower, sigma_upper, &lo, &hi, mad_buffer);
            int n = 0;
            double sum = 0.0;
            for (long f = 0; f < N; f++)
                if (isfinite(col[f]) && !(col[f] < lo) && !(col[f] > hi)) { sum += col[f]; n++; }
            mean_out_f32[p] = n > 0 ? (float)(sum / n) : NAN;
            if (count_out) count_out[p] = n;
        }
        free(buffer);
    }
    return 0;
}

int apref_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void apref_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---------------------------------------------------------------------------------------------------
 * F3  Affine Lanczos-3 resample of registered frames (the step SWarp performs for the reference:
 *     scripts/resample_all.sh:123-131 RESAMPLING_TYPE LANCZOS3, :330-342 the swarp call; FSCALE_DEFAULT =
 *     1/EXPTIME :298).  No in-tree arithmetic exists - the definition below is this build's own
 *     (parity with SWarp unpinned); the HIP kernel must match it bit for bit.
 *
 *     For output pixel (x, y) (0-based column, row) of frame f with affine A = affines[f][0..5]:
 *         F[k] = llrint(A[k] * 2^32);  xin = F0 x + F1 y + F2,  yin = F3 x + F4 y + F5      (32.32 fixed point, int64)
 *         ix = xin >> 32, px = the top log2(n_phases) bits of the fraction, rounded   (0 .. n_phases), same for y
 *         (n_phases a power of two; round 2 evaluated the transform in float64 - see resample_tile_fx below)
 *         wx = lut[px][0..5], wy = lut[py][0..5]      (row p holds the normalised Lanczos-3 weights of the
 *                                                       taps ix-2 .. ix+3 for a fractional offset p/n_phases)
 *         e_j = fmaf(wx4, s4, fmaf(wx2, s2, wx0 * s0)), o_j = fmaf(wx5, s5, fmaf(wx3, s3, wx1 * s1)),
 *                                                     s_i = src[iy-2+j][ix-2+i]   (even / odd taps of row j)
 *         ve = fmaf chain over j of wy[j] * e_j, vo likewise of o_j (first term a product);
 *         v = ve + vo;   out = v * fscale[f]          (float32; a packed-pair evaluation order)
 *     If any of the 36 taps lies outside the frame, on a masked pixel (mask != 0) or on a non-finite value,
 *     out = NaN.  weight = 1 where out is not NaN, else 0.
 * --------------------------------------------------------------------------------------------------- */
/* 32.32 fixed-point statement of the transform (round 3): F[k] = llrint(A[k] * 2^32); an output pixel (x, y) maps to
 * xin = F0 x + F1 y + F2, yin = F3 x + F4 y + F5 in 64-bit two's-complement arithmetic (exactly reproducible on CPU and GPU,
 * 4 integer instructions where the float64 evaluation needed ~25); floor = xin >> 32, sub-pixel phase = the top log2(n_phases)
 * bits of the fraction, rounded to nearest.  A 64 x 16 output tile whose corner coordinates (float64, fma order below) leave
 * +-1e9 pixels, or whose coefficients are not below 2^30 in magnitude, is undefined (NaN) as a whole. */
/* os = 1, or the oversampling factor: A then maps the pixels of the os-times finer grid, and the tile's corners are the
 * first / last fine pixels of its output pixels. */
static int resample_tile_fx(const double *A, long x0, long y0, long w_out, long h_out, long os, int64_t *F)
{
    const long xl = x0 + 63 < w_out - 1 ? x0 + 63 : w_out - 1, yl = y0 + 15 < h_out - 1 ? y0 + 15 : h_out - 1;
    const double xa = (double)(x0 * os), xb = (double)(xl * os + (os - 1));
    const double ya = (double)(y0 * os), yb = (double)(yl * os + (os - 1));
    const double cx[4] = {xa, xb, xa, xb}, cy[4] = {ya, ya, yb, yb};
    for (int k = 0; k < 4; k++) {
        const double xi = fma(A[0], cx[k], fma(A[1], cy[k], A[2]));
        const double yi = fma(A[3], cx[k], fma(A[4], cy[k], A[5]));
        if (!(xi > -1e9 && xi < 1e9 && yi > -1e9 && yi < 1e9)) return 0;
    }
    for (int k = 0; k < 6; k++) {
        if (!(fabs(A[k]) < 1073741824.0)) return 0;
        F[k] = (int64_t)llrint(A[k] * 4294967296.0);
    }
    return 1;
}

/* The interpolated value at (fine) output pixel (u, v) under the fixed-point transform F, times fs; NaN if undefined. */
static float resample_sample_fx(const float *src, const uint8_t *mask, long h_in, long w_in, const int64_t *F, int64_t u, int64_t v,
                                const float *lut, int sh, float fs)
{
    const int64_t xin = (int64_t)((uint64_t)F[0] * (uint64_t)u + (uint64_t)F[1] * (uint64_t)v + (uint64_t)F[2]);
    const int64_t yin = (int64_t)((uint64_t)F[3] * (uint64_t)u + (uint64_t)F[4] * (uint64_t)v + (uint64_t)F[5]);
    const int64_t ix = xin >> 32, iy = yin >> 32;
    /* the 6x6 window must lie inside the frame */
    if (!(ix >= 2 && iy >= 2 && ix <= w_in - 4 && iy <= h_in - 4)) return NAN;
    const uint32_t frx = (uint32_t)xin, fry = (uint32_t)yin;
    const int px = (int)((frx >> sh) + ((frx >> (sh - 1)) & 1u));
    const int py = (int)((fry >> sh) + ((fry >> (sh - 1)) & 1u));
    const float *wx = lut + 6 * px, *wy = lut + 6 * py;
    int ok = 1;
    float ve = 0.f, vo = 0.f;
    for (int j = 0; j < 6; j++) {
        const long row = iy - 2 + j;
        const float *s = src + row * w_in + (ix - 2);
        for (int i = 0; i < 6; i++)
            if (!isfinite(s[i]) || (mask && mask[row * w_in + ix - 2 + i])) ok = 0;
        const float e = fmaf(wx[4], s[4], fmaf(wx[2], s[2], wx[0] * s[0]));   /* even taps */
        const float o = fmaf(wx[5], s[5], fmaf(wx[3], s[3], wx[1] * s[1]));   /* odd taps  */
        ve = (j == 0) ? wy[0] * e : fmaf(wy[j], e, ve);
        vo = (j == 0) ? wy[0] * o : fmaf(wy[j], o, vo);
    }
    const float val = ve + vo;
    return (ok && val == val) ? val * fs : NAN;
}

/* os = 1: apref_resample_affine_f32.  os > 1: SWarp's OVERSAMPLING (resample_all.sh:112, 339) - `affines` belongs to the
 * os-times finer grid (one per frame, or one per 16 x 64 tile of the OUTPUT grid), every output pixel is
 * (float)(sum * (1.0 / os^2)) of its os x os samples accumulated in float64 in row-major order: the same numbers as
 * apref_block_mean_f32 of the fine resample (the GPU's one-pass apgpu_resample_oversampled_f32 restated). */
static int resample_any(const float *frames, long n_frames, long h_in, long w_in, const uint8_t *mask,
                        const double *affines, int per_tile, int conserve_flux, const float *fscale, const float *lut,
                        int n_phases, long os, float *out, uint8_t *weight_out, long h_out, long w_out)
{
    if (!frames || !affines || !lut || !out || n_phases < 2 || (n_phases & (n_phases - 1)) || os < 1 || os > 16) return -1;
    int log2p = 0;
    while ((1 << log2p) < n_phases) log2p++;
    const int sh = 32 - log2p;
    const double inv = 1.0 / (double)(os * os);
#pragma omp parallel for collapse(2) schedule(static)
    for (long f = 0; f < n_frames; f++)
        for (long y = 0; y < h_out; y++) {
            const float *src = frames + f * h_in * w_in;
            const float fs0 = fscale ? fscale[f] : 1.0f;
            const long tiles_x = (w_out + 63) / 64, tiles_y = (h_out + 15) / 16;
            for (long x = 0; x < w_out; x++) {
                /* one transform per frame, or one per 16 x 64 output tile (piecewise-affine registration) */
                const double *A = affines + 6 * (per_tile ? (f * tiles_y + y / 16) * tiles_x + x / 64 : f);
                /* FSCALASTRO_TYPE VARIABLE: (fine) output pixel area in input pixels */
                const float fs = conserve_flux ? (float)((double)fs0 * fabs(fma(A[0], A[4], -(A[1] * A[3])))) : fs0;
                int64_t F[6];
                float res = NAN;
                if (resample_tile_fx(A, (x / 64) * 64, (y / 16) * 16, w_out, h_out, os, F)) {
                    if (os == 1) {
                        res = resample_sample_fx(src, mask, h_in, w_in, F, x, y, lut, sh, fs);
                    } else {
                        double acc = 0.0;            /* a NaN sample makes the sum, and the pixel, NaN */
                        for (long a = 0; a < os; a++)
                            for (long b = 0; b < os; b++)
                                acc += (double)resample_sample_fx(src, mask, h_in, w_in, F, x * os + b, y * os + a, lut, sh, fs);
                        res = (float)(acc * inv);
                    }
                }
                out[(f * h_out + y) * w_out + x] = res;
                if (weight_out) weight_out[(f * h_out + y) * w_out + x] = (res == res) ? 1 : 0;   /* weight plane: out is defined */
            }
        }
    return 0;
}

int apref_resample_affine_f32(const float *frames, long n_frames, long h_in, long w_in, const uint8_t *mask,
                              const double *affines, int per_tile, int conserve_flux, const float *fscale, const float *lut,
                              int n_phases,
                              float *out, uint8_t *weight_out, long h_out, long w_out)
{
    return resample_any(frames, n_frames, h_in, w_in, mask, affines, per_tile, conserve_flux, fscale, lut, n_phases, 1, out, weight_out,
                        h_out, w_out);
}

int apref_resample_oversampled_f32(const float *frames, long n_frames, long h_in, long w_in, const uint8_t *mask,
                                   const double *fine_affines, int per_tile, int conserve_flux, const float *fscale, const float *lut,
                                   int n_phases, int oversampling, float *out, uint8_t *weight_out, long h_out, long w_out)
{
    return resample_any(frames, n_frames, h_in, w_in, mask, fine_affines, per_tile, conserve_flux, fscale, lut, n_phases, oversampling, out,
                        weight_out, h_out, w_out);
}
