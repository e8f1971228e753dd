/* symdemod_core.h -- host logic of the symdemod pipe stage (reference symdemod.c), engine-agnostic.
 * Sequential control stays here: option handling (:51-85), the sliding two-window buffer (:89-125),
 * the scount += halfclock boundary recurrences (:216-236, :268-292), first-maximum selection
 * (:327-331), the optional clock hill-climb (:133-174).  The loop nests over samples run in the
 * engine (libisee3dsp_hip.so: symd_load / symd_timesearch / symd_demod). */
#ifndef SYMDEMOD_CORE_H
#define SYMDEMOD_CORE_H
#include <stdint.h>
#include <stdio.h>

typedef struct {
  int    samprate;       /* -r, default 250000 */
  double symrate;        /* -c, default 1024.545058 */
  int    symbolclocks;   /* -C, default 1 */
  double window;         /* -w, default 1.0 s */
  int    clocktrack;     /* -t */
  int    quiet;          /* -q */
  const char *argv0;
} symdemod_opts;

typedef struct {
  void *(*create)(int max_samples);
  int   (*load)(void *h, const int16_t *samples, int n);
  int   (*timesearch)(void *h, int lo, const int *sw, int symbolclocks, int nsymbols, int noff, double *energies);
  int   (*demod)(void *h, const int *edges, int symbolclocks, int nsymbols, double gain, uint8_t *out, double *energy_sum);
  void  (*destroy)(void *h);
  /* optional (all three or none; needed by symdemod_run_blk): the two-window sample buffer of symdemod.c:96-125 kept
     inside the engine (in HBM), zero-filled at create:
       store_slide == memmove(buf, buf + slide, nsamples - slide), the rest of the buffer unchanged
       store_put   == copy n samples (host or device memory) to buf[at..)
       store_scan  == make buf[0..n) the window that timesearch / demod work on (what load() does after its copy) */
  int   (*store_slide)(void *h, int slide, int nsamples);
  int   (*store_put)(void *h, int at, const int16_t *src, int n, int src_is_dev);
  int   (*store_scan)(void *h, int n);
  /* optional: a whole window (search, first maximum, final demodulation) behind ONE synchronisation, the boundary tables
     of the final demodulation speculated by the caller for the timing adjustments spec_lo .. spec_lo + nspec - 1
     (include/isee3_dsp_hip.h: symd_window).  0 done, 1 take the step-by-step calls for this window, -1 error. */
  int   (*window)(void *h, int firstsample, const int *sw, int symbolclocks, int nsymbols, int first_off, int noff,
                  const int *edges, int spec_lo, int nspec, uint8_t *out, int *symphase, double *maxenergy);
} symdemod_engine;

void symdemod_default_opts(symdemod_opts *o);
int  symdemod_parse_args(symdemod_opts *o, int argc, char **argv);
/* the -c argument rule (symdemod.c:67-77) on its own: pure, no getopt -- for callers inside a library */
void symdemod_set_symrate(symdemod_opts *o, const char *arg);
int  symdemod_run(const symdemod_opts *o, const symdemod_engine *e, int fd_in, FILE *out, FILE *err);
/* the same stage reading through rd(ctx, buf, nbytes) (read(2) semantics: > 0 bytes, 0 at end of input) instead
 * of a file descriptor -- the in-process chain hands blocks over in memory */
typedef long (*symdemod_reader)(void *ctx, void *buf, unsigned long nbytes);
int  symdemod_run_rd(const symdemod_opts *o, const symdemod_engine *e, symdemod_reader rd, void *rctx, FILE *out, FILE *err);
/* the same stage fed with VIEWS of sample blocks that may lie in device memory: next(ctx, &blk, &is_dev, max) hands out
 * up to max samples (> 0; *blk stays valid until the next call) or 0 at end of input; the samples go straight into
 * the engine's store, never through host memory */
typedef long (*symdemod_block_reader)(void *ctx, const int16_t **blk, int *is_dev, long max_samples);
int  symdemod_run_blk(const symdemod_opts *o, const symdemod_engine *e, symdemod_block_reader next, void *rctx, FILE *out, FILE *err);
#endif
