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
} pmdemod_engine;

typedef struct { int peak; double carrier_freq, cn0; } pmdemod_block_report;

void pmdemod_default_opts(pmdemod_opts *o);
/* returns 0, or the reference's exit code (1) for an unknown option */
int  pmdemod_parse_args(pmdemod_opts *o, int argc, char **argv, FILE *err);
/* returns the process exit code of pmdemod.c (0, 1, 2); report/nreport optional per-block log */
int  pmdemod_run(const pmdemod_opts *o, const pmdemod_engine *e, FILE *in, FILE *out, FILE *err,
                 pmdemod_block_report *report, int report_cap, int *nreport);
#endif
