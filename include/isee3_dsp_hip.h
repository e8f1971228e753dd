/* isee3_dsp_hip.h -- C-ABI of libisee3dsp_hip.so: the data-parallel kernels of the pmdemod and
 * symdemod pipe stages on MI355X (gfx950).
 *
 * The reference keeps these computations inline in two main() functions (pmdemod.c:204-368,
 * symdemod.c:202-335); there is no library interface to mirror, so the boundary is drawn where the
 * reference's own loops are: one call per loop nest, sequential control (buffer sliding, the
 * scount += halfclock recurrences, Quinn interpolation, lock state) stays in the C host stages
 * (isee3-decoder_amd/cli/{pmdemod,symdemod}_core.c).  Plain pointers and sizes only.
 *
 * All functions return 0 on success, -1 on error (isee3dsp_last_error() says why).
 */
#ifndef ISEE3_DSP_HIP_H
#define ISEE3_DSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *isee3dsp_last_error(void);
/* default device of the create / alloc calls of the PROCESS (any thread; atomic), as v224hip_set_device */
int isee3dsp_set_device(int dev);
/* Which stream the handles created by the calling thread from now on use: 0 one of their own each (default), 1 ONE
 * stream per device shared by all such handles, 2 the null stream.  The in-process chain uses 1 for pmdemod and 2 for
 * symdemod: together with its two Viterbi decoders that is four busy streams on the GPU's four compute pipes. */
void isee3dsp_share_stream(int mode);

/* device / pinned host memory for C callers that keep streams resident in HBM (used by libisee3chain.so) */
void *isee3dsp_dev_alloc(size_t bytes);
void  isee3dsp_dev_free(void *d);
int   isee3dsp_h2d(void *d_dst, const void *h_src, size_t bytes);
int   isee3dsp_d2h(void *h_dst, const void *d_src, size_t bytes);

/* ------------------------------------------------------------------ symdemod (symdemod.c) ---- */
/* One handle owns: a device copy of the current sample window and its exact int64 prefix sum. */
void *symd_create(int max_samples);
void  symd_destroy(void *h);

/* Load n int16 samples (host pointer, or device pointer with is_dev = 1) and build
 * P[k] = sum(samples[0..k)) in int64.  Every integrate-and-dump sum of symdemod.c:227-235 /
 * :283-293 is then a difference of two P entries -- identical integers. */
int symd_load(void *h, const int16_t *samples, int n, int is_dev);

/* The same in pieces, for a window buffer that LIVES in device memory (symdemod.c:96-125: a two-window buffer that is
 * slid with memmove and topped up by read()).  The handle's buffer holds max_samples samples, zero after create /
 * symd_store_reset:
 *   symd_store_slide(h, slide, nsamples)  == memmove(buf, buf + slide, (nsamples - slide) samples); the rest of the
 *                                            buffer keeps its old contents, exactly like the reference's
 *   symd_store_put(h, at, src, n, dev)    == copy n samples to buf[at..at+n) from host or device memory (with a device
 *                                            source the call returns when the copy is done: src may be recycled)
 *   symd_store_scan(h, n)                 == prefix sums over buf[0..n), after which timesearch / demod work on it
 * symd_load(h, s, n, dev) = symd_store_put(h, 0, s, n, dev) + symd_store_scan(h, n). */
int symd_store_reset(void *h);
int symd_store_slide(void *h, int slide, int nsamples);
int symd_store_put(void *h, int at, const int16_t *src, int n, int src_is_dev);
int symd_store_scan(void *h, int n);

/* timesearch (symdemod.c:260-335) for offsets t = 0..noff-1 counted from base sample `lo`:
 *   sw[0] = 0, sw[k+1] = reference switchpoints[k]  (2*symbolclocks*nsymbols + 1 entries, host)
 *   energies[t] = sum over symbols, IN SYMBOL ORDER in double, of (long long)sym*sym
 * The host picks the first maximum (strict '>', symdemod.c:327). */
int symd_timesearch(void *h, int lo, const int *sw, int symbolclocks, int nsymbols, int noff,
                    double *energies);

/* trial_demod (symdemod.c:202-256).  edges[0] = first sample index, edges[k] the successive
 * nearbyint(scount) boundaries (2*symbolclocks*nsymbols + 1 entries, host, indices into the
 * loaded window).  energy_sum (may be NULL) = sequential double sum of sym^2;
 * out (may be NULL; host, or device when out_is_dev) gets the 8-bit symbols when gain != 0:
 * (unsigned char)clip(gain*integrator + 128, 0, 255)  (symdemod.c:240-251). */
int symd_demod(void *h, const int *edges, int symbolclocks, int nsymbols, double gain,
               uint8_t *out, int out_is_dev, double *energy_sum);

/* One whole window of symdemod.c:127-131,190-192 (no -t) with ONE synchronisation: timing search over offsets
 * firstsample + first_off + t (t < noff; sw as symd_timesearch), its first maximum, and trial_demod with gain
 * 100 / sqrt(maxenergy) -- the host SPECULATES the boundary tables: edges holds nspec tables of 2*symbolclocks*nsymbols + 1
 * absolute boundaries, table j being trial_demod's recurrence (symdemod.c:212-236) started at firstsample + spec_lo + j.
 * Returns 0: out[nsymbols], *symphase (the timing adjustment that won) and *maxenergy (per symbol) are what the
 * step-by-step calls give; 1: not decidable this way (adjustment outside the tables, sums outside the exactly representable
 * range, a differently rounded gain) -- nothing was produced, take symd_timesearch / symd_demod for this window; -1: error. */
int symd_window(void *h, int firstsample, const int *sw, int symbolclocks, int nsymbols, int first_off, int noff,
                const int *edges, int spec_lo, int nspec, uint8_t *out, int *symphase, double *maxenergy);

/* ------------------------------------------------------------------ pmdemod (pmdemod.c) ------ */
typedef struct {
  int    peak;            /* argmax |X|^2 over [firstbin,lastbin), last maximum wins (pmdemod.c:288-298) */
  double maxenergy;
  double peak_re, peak_im, next_re, next_im, prev_re, prev_im;   /* bins for Quinn-2 (:299-318) */
} pmd_peak;

typedef struct {
  double dc_re, dc_im;    /* mean of the spun-down block (pmdemod.c:332-336) */
  double amplitude;       /* |dc| */
  double diffsumsq;       /* mean((Re - amplitude)^2) after rotation (:341-348) */
} pmd_mix;

void *pmd_create(int fftsize);          /* fftsize = 2^n, 2^4 .. 2^24 */
void  pmd_destroy(void *h);

/* De-chirp LO (pmdemod.c:232-244): lophase_ri = fftsize complex doubles holding the reference's
 * lophase sequence, produced on the host by that same sequential recurrence (it restarts every
 * block, so it is computed once); its rounding walk (~i^1.5 * 1e-16) cannot be reproduced by a
 * closed form.  NULL = off. */
int pmd_set_dechirp(void *h, const double *lophase_ri);
/* pmdemod.c:204-230 (+ :237 with a de-chirp table): int16 (I,Q) pairs -> double complex block;
 * flip swaps I and Q.  iq is a host pointer, or a device pointer when is_dev. */
int pmd_load(void *h, const int16_t *iq, int is_dev, int flip);
/* pmdemod.c:253-298: forward unnormalised FFT (input preserved) + windowed peak search */
int pmd_fft_peak(void *h, int firstbin, int lastbin, pmd_peak *out);
/* pmdemod.c:321-368: spin down by cstep_rad = 2*pi*carrier_freq/samprate per sample (the closed
 * form of the reference's carrier recurrence, csrc/carrier_params.c), average,
 * rotate carrier onto the real axis, noise variance, quantise Im * sqrt(1/2) to int16.
 * out16 / pre (optional pre-quantisation doubles) are host pointers, or device if out_is_dev. */
int pmd_mix_quantise(void *h, double cstep_rad, pmd_mix *res, int16_t *out16, double *pre, int out_is_dev);
/* The same two calls in an asynchronous form: *_begin enqueues the work on the handle's stream and records an event,
 * *_end waits for that event alone and reads the results (pmd_fft_peak == begin + end; pmd_mix_quantise likewise).  Two
 * handles on one stream can thus overlap one block's transform with the host's work on the previous block -- and with its
 * spin-down passes, which read only the int16 block, never the spectrum.  pmd_mix_begin returns 1 (nothing enqueued) for
 * a block the stepped-carrier kernels do not take (under 1 024 samples, unaligned buffers): use pmd_mix_quantise then.
 * The int16 block announced by pmd_load must stay valid until pmd_mix_end has returned. */
int pmd_fft_peak_begin(void *h, int firstbin, int lastbin);
int pmd_fft_peak_end(void *h, pmd_peak *out);
int pmd_mix_begin(void *h, double cstep_rad, int16_t *out16, double *pre, int out_is_dev);
int pmd_mix_end(void *h, pmd_mix *res);
/* How pmd_fft_peak finds the peak (transforms of 2^12 points and more): a SINGLE-precision transform names the bins whose
 * energy lies within 2^-10 of the maximum (normally one), and those bins and their neighbours are then evaluated from the
 * int16 block in double precision, as direct sums -- the values returned are at least as accurate as a double transform's,
 * at under half its memory traffic.  When single precision cannot name the bin for certain (a second bin within 2^-10
 * of the maximum, an all-zero block) pmd_fft_peak_end runs the double transform instead and its spectrum decides, as
 * before.  ISEE3DSP_FFT_F64=1: always the double transform.
 * pmd_last_peak_path: 0 double transform, 1 search transform + exact bins, 2 search transform, then fallen back. */
int pmd_last_peak_path(void *h);
/* test hook: copy the spectrum (fftsize complex doubles) to host (runs the double transform if the peak search did not) */
int pmd_get_spectrum(void *h, double *out_ri);

/* ------------------------------------------------------------------ icesync.c correlator ---- */
/* The FFT sync-vector correlator of the legacy one-shot program (icesync.c:55-208): the cross-correlation
 * of one frame of baseband with the Manchester-coded tail + sync symbols, through transforms of corr_size points
 * (icesync.c:102-103 hard-codes 2^20).  The vector itself (encoder + Manchester coding, :55-97) is built by the host:
 * isee3-decoder_amd/cli/icesync_core.c. */
#define ISYNC_FAIL (-1234567890)                   /* icesync.c:31 SYNC_FAIL */
void *isync_create(int corr_size);                 /* power of two, 2^12 .. 2^24 */
void  isync_destroy(void *h);
/* icesync.c:122-135: zero-pad vec[0..synclen) to corr_size, transform, conjugate */
int   isync_set_vector(void *h, const double *vec, int synclen);
/* icesync.c:139-208: nsamples int16 baseband samples (host, or device with is_dev) zero-padded to corr_size, transformed,
 * multiplied with the vector's transform, transformed back; *peakindex = index of the first maximum > 0 of the result in
 * [low, min(high, corr_size)), folded to corr_size - index above corr_size / 2 -- or ISYNC_FAIL when every sample is zero or
 * nothing in the window is positive.  *maxpeak (optional) = the value there; result (optional, host, corr_size doubles). */
int   isync_search(void *h, const int16_t *samples, int nsamples, int is_dev, int low, int high,
                   int *peakindex, double *maxpeak, double *result);

#ifdef __cplusplus
}
#endif
#endif
