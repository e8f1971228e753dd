/* pmdemod_core.h -- host logic of the pmdemod pipe stage (reference pmdemod.c), engine-agnostic.
 * Stays on the host: options (:85-115), FFT size (:129-131), block reading (:204-230), search-window
 * selection from the lock state (:255-285), Quinn's second estimator on three bins (:43-46,:299-318),
 * C/N0 and lock bookkeeping (:349-354), status lines.  The per-sample loops run in the engine
 * (libisee3dsp_hip.so: pmd_load / pmd_fft_peak / pmd_mix_quantise). */
#ifndef PMDEMOD_CORE_H
#define PMDEMOD_CORE_H
#include <stdint.h>
#include <stdio.h>

typedef struct {
  double samprate;        /* -r, 250000 */
  double binsize;         /* -b, 4 */
  double search_freq;     /* -S */
  double search_width;    /* -W */
  double doppler_rate;    /* -D, Hz/s */
  double cn0_threshold;   /* -t, 21 dB-Hz */
  int    flip;            /* -f */
  int    quiet;           /* -q */
  const char *file;       /* optional input file */
  const char *argv0;
} pmdemod_opts;

typedef struct { int peak; double maxenergy, peak_re, peak_im, next_re, next_im, prev_re, prev_im; } pmdemod_peak;
typedef struct { double dc_re, dc_im, amplitude, diffsumsq; } pmdemod_mix;

typedef struct {
  void *(*create)(int fftsize);
  int   (*set_dechirp)(void *h, const double *lophase_ri);
  int   (*load)(void *h, const int16_t *iq, int flip);
  int   (*fft_peak)(void *h, int firstbin, int lastbin, pmdemod_peak *out);
  int   (*mix)(void *h, double cstep, pmdemod_mix *res, int16_t *out16);
  void  (*destroy)(void *h);
  /* optional (needed by pmdemod_run_io when its source / sink hand out device memory): the same two calls on blocks
     that are already, or shall stay, in device memory */
  int   (*load_dev)(void *h, const int16_t *d_iq, int flip);
  int   (*mix_dev)(void *h, double cstep, pmdemod_mix *res, int16_t *d_out16);
  /* optional (all four or none): fft_peak and mix split into enqueue / collect (include/isee3_dsp_hip.h: pmd_*_begin / _end).
     With them -- and blocks that do not depend on each other (-W 0) -- pmdemod_run_io alternates TWO handles so that block
     k+1's transform is in the engine's queue before block k's peak is waited for, and a block is handed on one iteration
     later.  mix_begin: 0 enqueued, 1 this block takes mix() / mix_dev(), < 0 error. */
  int   (*fft_peak_begin)(void *h, int firstbin, int lastbin);
  int   (*fft_peak_end)(void *h, pmdemod_peak *out);
  int   (*mix_begin)(void *h, double cstep, int16_t *out16, int out_is_dev);
  int   (*mix_end)(void *h, pmdemod_mix *res);
} pmdemod_engine;

/* where blocks come from / go to when they are not a FILE: views, so a capture in memory (host or device) is read in
 * place and a consumer in the same process takes the baseband where the engine wrote it */
typedef struct {
  /* a view of the next N (I,Q) pairs; 1 = *blk valid until the next call, 0 = end of input (a partial block is dropped,
     pmdemod.c:206-216), < 0 = error */
  int (*next)(void *ctx, int N, const int16_t **blk, int *is_dev);
  void *ctx;
} pmdemod_source;
typedef struct {
  /* room for the next block's N int16 samples (host or device memory), then commit() publishes it */
  int16_t *(*acquire)(void *ctx, int N, int *is_dev);
  int (*commit)(void *ctx, int16_t *buf, int N);
  void *ctx;
} pmdemod_sink;

typedef struct { int peak; double carrier_freq, cn0; } pmdemod_block_report;

void pmdemod_default_opts(pmdemod_opts *o);
/* returns 0, or the reference's exit code (1) for an unknown option */
int  pmdemod_parse_args(pmdemod_opts *o, int argc, char **argv, FILE *err);
/* returns the process exit code of pmdemod.c (0, 1, 2); report/nreport optional per-block log */
int  pmdemod_run(const pmdemod_opts *o, const pmdemod_engine *e, FILE *in, FILE *out, FILE *err,
                 pmdemod_block_report *report, int report_cap, int *nreport);
/* the same stage between a block source and a block sink (in_for_stat: optional FILE whose size goes into the
 * "demodulating ..." status line, as pmdemod.c:178-203 prints for a regular file) */
int  pmdemod_run_io(const pmdemod_opts *o, const pmdemod_engine *e, const pmdemod_source *src, const pmdemod_sink *dst,
                    FILE *in_for_stat, FILE *err, pmdemod_block_report *report, int report_cap, int *nreport);
#endif
