/* symdemod_core.c -- see symdemod_core.h.  Plain C11 (build with -ffp-contract=off). */
#define _GNU_SOURCE
#include "symdemod_core.h"
#include <fenv.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "timefmt.h"

#define NOMINALCLOCK 1024.0          /* symdemod.c:17 */
#define ACTUALCLOCK  1024.545058     /* symdemod.c:18 */

void symdemod_default_opts(symdemod_opts *o) {
  memset(o, 0, sizeof *o);
  o->samprate = 250000; o->symrate = ACTUALCLOCK; o->symbolclocks = 1; o->window = 1.0;
  o->argv0 = "symdemod";
}

/* the -c rule, symdemod.c:67-77: no decimal point = a nominal rate, scaled to the measured spacecraft clock */
void symdemod_set_symrate(symdemod_opts *o, const char *arg) {
  if (!strchr(arg, '.')) o->symrate = atof(arg) * ACTUALCLOCK / NOMINALCLOCK;
  else o->symrate = atof(arg);
  if (o->symrate < 1000) o->symbolclocks = (int)rint(NOMINALCLOCK / o->symrate);
}

int symdemod_parse_args(symdemod_opts *o, int argc, char **argv) {
  int c;
  symdemod_default_opts(o);
  if (argc > 0) o->argv0 = argv[0];
  optind = 1;
  while ((c = getopt(argc, argv, "w:c:r:qtC:")) != -1) {
    switch (c) {
    case 't': o->clocktrack = 1; break;
    case 'w': o->window = atof(optarg); break;
    case 'q': o->quiet = 1; break;
    case 'c': symdemod_set_symrate(o, optarg); break;
    case 'r': o->samprate = atoi(optarg); break;
    case 'C': o->symbolclocks = atoi(optarg); break;
    default: break;
    }
  }
  return 0;
}

typedef struct {
  const symdemod_engine *e; void *h;
  int symbolclocks;
  int *idx; int idx_cap;
  double *energies; int en_cap;
} ctx_t;

static int need_idx(ctx_t *c, int n) {
  if (c->idx_cap >= n) return 0;
  free(c->idx);
  c->idx = malloc(sizeof(int) * (size_t)n);
  c->idx_cap = c->idx ? n : 0;
  return c->idx ? 0 : -1;
}

/* trial_demod, symdemod.c:202-256: absolute boundaries from the running scount */
static int trial(ctx_t *c, int firstsample, double symbolsamples, int nsymbols, double gain, uint8_t *out,
                 double *energy_per_symbol) {
  int ne = 2 * c->symbolclocks * nsymbols + 1;
  if (need_idx(c, ne)) return -1;
  double halfclock = (0.5 / c->symbolclocks) * symbolsamples;
  double scount = firstsample + halfclock;
  c->idx[0] = firstsample;
  for (int k = 1; k < ne; k++) { c->idx[k] = (int)nearbyint(scount); scount += halfclock; }
  double esum = 0;
  if (c->e->demod(c->h, c->idx, c->symbolclocks, nsymbols, gain, out, &esum) != 0) return -1;
  if (energy_per_symbol) *energy_per_symbol = esum / nsymbols;
  return 0;
}

/* timesearch, symdemod.c:260-335: relative switch points; offsets use the GLOBAL Symbolsamples */
static int timesearch(ctx_t *c, int *symphase, int firstsample, double symbolsamples, double symbolsamples_global,
                      int nsymbols, double *maxenergy_per_symbol) {
  int nsw = 2 * c->symbolclocks * nsymbols + 1;
  if (need_idx(c, nsw)) return -1;
  double halfclock = (0.5 / c->symbolclocks) * symbolsamples, scount = halfclock;
  c->idx[0] = 0;
  for (int k = 1; k < nsw; k++) { c->idx[k] = (int)nearbyint(scount); scount += halfclock; }
  int first_off = (int)(-symbolsamples_global / 2), noff = 0;
  for (int o = first_off; o < symbolsamples_global / 2; o++) noff++;
  if (c->en_cap < noff) {
    free(c->energies);
    c->energies = malloc(sizeof(double) * (size_t)noff);
    c->en_cap = c->energies ? noff : 0;
    if (!c->energies) return -1;
  }
  if (c->e->timesearch(c->h, firstsample + first_off, c->idx, c->symbolclocks, nsymbols, noff, c->energies) != 0)
    return -1;
  double best = c->energies[0];
  int bi = 0;
  for (int t = 1; t < noff; t++) if (c->energies[t] > best) { best = c->energies[t]; bi = t; }   /* first max */
  *symphase = first_off + bi;
  *maxenergy_per_symbol = best / nsymbols;
  return 0;
}

/* A whole window through engine->window (symdemod_core.h): the relative switch points of the search, and the absolute
 * boundaries of the final demodulation for the timing adjustments -SPEC_HALF .. +SPEC_HALF -- each table the reference's own
 * recurrence (symdemod.c:212-236) started at firstsample + adjustment, as trial() forms it.  The adjustment is a sample
 * or two in a tracking loop; anything else comes back as 1 and the caller takes the step-by-step path. */
#define SPEC_HALF 2
static int window_fused(ctx_t *c, int firstsample, double symbolsamples, int nsymbols, uint8_t *out, int *symphase,
                        double *maxenergy_per_symbol) {
  const int nsw = 2 * c->symbolclocks * nsymbols + 1, nspec = 2 * SPEC_HALF + 1;
  if (need_idx(c, nsw * (1 + nspec))) return -1;
  const double halfclock = (0.5 / c->symbolclocks) * symbolsamples;
  int *sw = c->idx, *tab = c->idx + nsw;
  double scount = halfclock;
  sw[0] = 0;
  for (int k = 1; k < nsw; k++) { sw[k] = (int)nearbyint(scount); scount += halfclock; }
  int first_off = (int)(-symbolsamples / 2), noff = 0;
  for (int o = first_off; o < symbolsamples / 2; o++) noff++;
  for (int j = 0; j < nspec; j++) {
    const int fs = firstsample + j - SPEC_HALF;
    int *e = tab + (size_t)j * nsw;
    scount = fs + halfclock;
    e[0] = fs;
    for (int k = 1; k < nsw; k++) { e[k] = (int)nearbyint(scount); scount += halfclock; }
  }
  return c->e->window(c->h, firstsample, sw, c->symbolclocks, nsymbols, first_off, noff, tab, -SPEC_HALF, nspec, out, symphase,
                      maxenergy_per_symbol);
}

static long fd_reader(void *ctx, void *buf, unsigned long nbytes) { return (long)read(*(int *)ctx, buf, nbytes); }
int symdemod_run(const symdemod_opts *o, const symdemod_engine *e, int fd_in, FILE *out, FILE *err) {
  return symdemod_run_rd(o, e, fd_reader, &fd_in, out, err);
}

static int run(const symdemod_opts *o, const symdemod_engine *e, symdemod_reader rd, symdemod_block_reader next, void *rctx,
               FILE *out, FILE *err);
int symdemod_run_rd(const symdemod_opts *o, const symdemod_engine *e, symdemod_reader rd, void *rctx, FILE *out, FILE *err) {
  return run(o, e, rd, NULL, rctx, out, err);
}
int symdemod_run_blk(const symdemod_opts *o, const symdemod_engine *e, symdemod_block_reader next, void *rctx, FILE *out, FILE *err) {
  if (!e->store_slide || !e->store_put || !e->store_scan) return -1;
  return run(o, e, NULL, next, rctx, out, err);
}

/* one window loop for both forms: `store` = the sample buffer lives in the engine (views from next()), else in
 * samples[] here (bytes from rd()) and goes to the engine whole, once per window */
static int run(const symdemod_opts *o, const symdemod_engine *e, symdemod_reader rd, symdemod_block_reader next, void *rctx,
               FILE *out, FILE *err) {
  const int store = next != NULL;
  fesetround(FE_TONEAREST);                          /* symdemod.c:48 */
  int Samprate = o->samprate;
  double Symrate = o->symrate, window = o->window;
  if (!o->quiet)
    fprintf(err, "%s: sample rate %'d Hz; estimation window %.3lf sec; clocks/symbol %d; symbol rate %.3lf Hz; tracking %s\n",
            o->argv0, Samprate, window, o->symbolclocks, Symrate, o->clocktrack ? "on" : "off");
  double Symbolsamples = Samprate / Symrate;
  int fullwater = (int)(window * 2.0 * Samprate);
  int nsymbols = (int)(window * Symrate);
  int firstsample = (int)(Symbolsamples / 2);
  int nsamples = 0, rc = -1;
  long long total_samples = 0, total_symbols = 0;
  int slack = (int)Symbolsamples + 64;
  int16_t *samples = store ? NULL : calloc((size_t)fullwater + (size_t)slack, sizeof *samples);  /* ref: malloc */
  uint8_t *obuf = malloc((size_t)(window * 1.5 * Samprate / 2) + 4096);
  ctx_t c = { e, NULL, o->symbolclocks, NULL, 0, NULL, 0 };
  c.h = e->create(fullwater + slack);
  if ((!store && !samples) || !obuf || !c.h) goto done;

  for (;;) {
    if (firstsample >= window * Samprate) {          /* purge old samples, keep 2 symbols of slop */
      int slide = (int)(firstsample - 2 * Symbolsamples);
      if (slide > nsamples) slide = nsamples;
      if (store) { if (e->store_slide(c.h, slide, nsamples) != 0) goto done; }
      else memmove(samples, samples + slide, sizeof(*samples) * (size_t)(nsamples - slide));
      nsamples -= slide; firstsample -= slide; total_samples += slide;
    }
    while (nsamples < fullwater) {
      if (store) {
        const int16_t *blk = NULL; int is_dev = 0;
        long cnt = next(rctx, &blk, &is_dev, fullwater - nsamples);
        if (cnt <= 0) break;
        if (e->store_put(c.h, nsamples, blk, (int)cnt, is_dev) != 0) goto done;
        nsamples += (int)cnt;
      } else {
        long cnt = rd(rctx, samples + nsamples, sizeof(*samples) * (unsigned long)(fullwater - nsamples));
        if (cnt <= 0) break;
        nsamples += (int)(cnt / (long)sizeof(*samples));
      }
    }
    if (nsamples < window * Samprate) break;

    /* the whole buffer goes to the engine, stale tail included: the reference's search can run a
       little past nsamples near end of input and reads whatever the buffer holds there */
    if (store ? e->store_scan(c.h, fullwater + slack) != 0 : e->load(c.h, samples, fullwater + slack) != 0) goto done;

    int symphase = 0, fused = 0;
    double maxenergy = 0;
    if (e->window && !o->clocktrack) {               /* search + first maximum + final demodulation in one engine call */
      const int wr = window_fused(&c, firstsample, Symbolsamples, nsymbols, obuf, &symphase, &maxenergy);
      if (wr < 0) goto done;
      fused = wr == 0;
    }
    if (!fused && timesearch(&c, &symphase, firstsample, Symbolsamples, Symbolsamples, nsymbols, &maxenergy)) goto done;
    firstsample += symphase;

    if (o->clocktrack) {                             /* symdemod.c:133-174 */
      double clock_incr = 0.5 * Symbolsamples / (window * Samprate), en;
      int phase_incr = 1;
      for (int nochange = 0; nochange < 2;) {
        if (trial(&c, firstsample, Symbolsamples + clock_incr, nsymbols, 0., NULL, &en)) goto done;
        if (en > maxenergy) { maxenergy = en; Symbolsamples += clock_incr; Symrate = Samprate / Symbolsamples; nochange = 0; }
        else {
          if (trial(&c, firstsample, Symbolsamples - clock_incr, nsymbols, 0., NULL, &en)) goto done;
          if (en > maxenergy) { maxenergy = en; Symbolsamples -= clock_incr; Symrate = Samprate / Symbolsamples;
                                clock_incr = -clock_incr; nochange = 0; }
          else nochange++;
        }
        if (trial(&c, firstsample + phase_incr, Symbolsamples, nsymbols, 0., NULL, &en)) goto done;
        if (en > maxenergy) { maxenergy = en; firstsample += phase_incr; nochange = 0; }
        else {
          if (trial(&c, firstsample - phase_incr, Symbolsamples, nsymbols, 0., NULL, &en)) goto done;
          if (en > maxenergy) { maxenergy = en; firstsample += phase_incr;   /* sic, symdemod.c:164-166 */
                                phase_incr = -phase_incr; nochange = 0; }
          else nochange++;
        }
      }
      nsymbols = (int)(window * Symrate);
    }
    if (!o->quiet)
      fprintf(err, "%s: sample %'lld (%'.3lf sec, %s) symbol %'lld: clock %'.4lf Hz; %'.4lf samp/sym; timing adj %+d samples; energy %.3lf dB\n",
              o->argv0, firstsample + total_samples, (double)(firstsample + total_samples) / Samprate,
              isee3_format_hms((double)(firstsample + total_samples) / Samprate), total_symbols, Symrate,
              Symbolsamples, symphase, 10 * log10(maxenergy));

    double gain = 100. / sqrt(maxenergy);            /* symdemod.c:190 */
    if (!fused && trial(&c, firstsample, Symbolsamples, nsymbols, gain, obuf, NULL)) goto done;
    fwrite(obuf, 1, (size_t)nsymbols, out);
    firstsample = (int)(firstsample + nsymbols * Symbolsamples);
    total_symbols += nsymbols;
    fflush(out);
  }
  rc = 0;
done:
  if (c.h) e->destroy(c.h);
  free(c.idx); free(c.energies); free(samples); free(obuf);
  return rc;
}
