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
#endif
