/* symdemod_oracle.c -- CPU restatement of the symdemod pipe stage (TEST INFRASTRUCTURE ONLY).
 *
 * Follows symdemod.c: option defaults :51-55, -c scaling :67-77, buffer management :89-125,
 * timesearch :260-335, optional clock hill-climb :133-174, final demod :188-193,
 * trial_demod :202-256.  Pinned against oracle/_ref/symdemod_ref (the reference compiled
 * unmodified) through tests/golden/symdemod_*.
 *
 * Integer sums are exact; every floating-point recurrence the reference runs sequentially
 * (scount += halfclock, energy += ...) is run in the same order here, in strict IEEE double
 * (build with -ffp-contract=off; rounding mode FE_TONEAREST as symdemod.c:48).
 *
 * timesearch is restated with an int64 prefix sum: the reference's incremental 3-sample update
 * (:312-322) yields, for every offset, exactly the integral over the shifted switch points, so
 * evaluating that integral from prefix differences gives identical integers.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define NOMINALCLOCK 1024.0         /* symdemod.c:17 */
#define ACTUALCLOCK  1024.545058    /* symdemod.c:18 */

void orc_symdemod_default(orc_symdemod_cfg *c) {
  memset(c, 0, sizeof *c);
  c->samprate = 250000; c->symrate = ACTUALCLOCK; c->symbolclocks = 1;
  c->window = 1.0; c->clocktrack = 0;
}

void orc_symdemod_set_c(orc_symdemod_cfg *c, const char *optarg) {
  if (!strchr(optarg, '.')) c->symrate = atof(optarg) * ACTUALCLOCK / NOMINALCLOCK;
  else c->symrate = atof(optarg);
  if (c->symrate < 1000) c->symbolclocks = (int)rint(NOMINALCLOCK / c->symrate);
}

double orc_trial_demod(const int16_t *samples, int firstsample, double symbolsamples,
                       int symbolclocks, int nsymbols, double gain, uint8_t *out) {
  double energy = 0;
  int ind = firstsample;
  double halfclock = (0.5 / symbolclocks) * symbolsamples;
  double scount = ind + halfclock;
  int edge = (int)nearbyint(scount);
  for (int i = 0; i < nsymbols; i++) {
    long acc = 0;
    for (int j = 0; j < symbolclocks; j++) {
      for (; ind < edge; ind++) acc -= samples[ind];
      scount += halfclock; edge = (int)nearbyint(scount);
      for (; ind < edge; ind++) acc += samples[ind];
      scount += halfclock; edge = (int)nearbyint(scount);
    }
    if (gain != 0) {
      double scaled = gain * acc + 128;
      if (scaled > 255) scaled = 255; else if (scaled < 0) scaled = 0;
      if (out) out[i] = (unsigned char)scaled;
    }
    energy += (long long)acc * acc;
  }
  return energy / nsymbols;
}

double orc_timesearch(int *symphase, const int16_t *samples, int firstsample,
                      double symbolsamples, double symbolsamples_global,
                      int symbolclocks, int nsymbols) {
  int nsw = nsymbols * 2 * symbolclocks;
  int *sw = malloc(sizeof(int) * (size_t)(nsw + 1));
  double halfclock = (0.5 / symbolclocks) * symbolsamples, scount = halfclock;
  sw[0] = 0;                                   /* sw[k+1] = reference switchpoints[k] */
  for (int k = 0; k < nsw; k++) { sw[k + 1] = (int)nearbyint(scount); scount += halfclock; }

  int first_off = (int)(-symbolsamples_global / 2);     /* symdemod.c:273 uses the GLOBAL */
  int lo = firstsample + first_off;                      /* lowest sample index touched */
  int noff = 0;
  for (int o = first_off; o < symbolsamples_global / 2; o++) noff++;
  int span = sw[nsw] + noff;                              /* prefix length needed */
  long long *P = malloc(sizeof(long long) * (size_t)(span + 1));
  P[0] = 0;
  for (int n = 0; n < span; n++) P[n + 1] = P[n] + samples[lo + n];

  double maxenergy = 0;
  for (int o = first_off, t = 0; o < symbolsamples_global / 2; o++, t++) {
    double energy = 0;
    for (int i = 0; i < nsymbols; i++) {
      long long sym = 0;
      for (int j = 0; j < symbolclocks; j++) {
        int k = 2 * (i * symbolclocks + j);
        sym += -(P[t + sw[k + 1]] - P[t + sw[k]]) + (P[t + sw[k + 2]] - P[t + sw[k + 1]]);
      }
      energy += sym * sym;                       /* (long long) product, double accumulate */
    }
    if (t == 0 || energy > maxenergy) { maxenergy = energy; *symphase = o; }
  }
  free(P); free(sw);
  return maxenergy / nsymbols;
}

size_t orc_symdemod(const orc_symdemod_cfg *c, const int16_t *in, size_t nin,
                    uint8_t *out, size_t outcap,
                    int *symphase_log, double *energy_log, int logcap, int *nwindows) {
  double Symbolsamples = c->samprate / c->symrate, Symrate = c->symrate;
  int Samprate = c->samprate;
  double window = c->window;
  int fullwater = (int)(window * 2.0 * Samprate);
  int16_t *samples = calloc((size_t)fullwater + 4, sizeof *samples);  /* ref: malloc, see note */
  int nsymbols = (int)(window * Symrate);
  int firstsample = (int)(Symbolsamples / 2);
  int nsamples = 0, nw = 0;
  size_t inpos = 0, nout = 0;

  for (;;) {
    if (firstsample >= window * Samprate) {
      int slide = (int)(firstsample - 2 * Symbolsamples);
      if (slide > nsamples) slide = nsamples;
      memmove(samples, samples + slide, sizeof(*samples) * (size_t)(nsamples - slide));
      nsamples -= slide; firstsample -= slide;
    }
    if (nsamples < fullwater) {
      size_t want = (size_t)(fullwater - nsamples), have = nin - inpos;
      size_t take = want < have ? want : have;
      memcpy(samples + nsamples, in + inpos, take * sizeof *samples);
      nsamples += (int)take; inpos += take;
    }
    if (nsamples < window * Samprate) break;

    int symphase = 0;
    double maxenergy = orc_timesearch(&symphase, samples, firstsample, Symbolsamples,
                                      Symbolsamples, c->symbolclocks, nsymbols);
    firstsample += symphase;

    if (c->clocktrack) {                          /* symdemod.c:133-174 */
      double clock_incr = 0.5 * Symbolsamples / (window * Samprate);
      int phase_incr = 1;
      for (int nochange = 0; nochange < 2;) {
        double e;
        if ((e = orc_trial_demod(samples, firstsample, Symbolsamples + clock_incr,
                                 c->symbolclocks, nsymbols, 0., NULL)) > maxenergy) {
          maxenergy = e; Symbolsamples += clock_incr; Symrate = Samprate / Symbolsamples; nochange = 0;
        } else if ((e = orc_trial_demod(samples, firstsample, Symbolsamples - clock_incr,
                                        c->symbolclocks, nsymbols, 0., NULL)) > maxenergy) {
          maxenergy = e; Symbolsamples -= clock_incr; Symrate = Samprate / Symbolsamples;
          clock_incr = -clock_incr; nochange = 0;
        } else nochange++;
        if ((e = orc_trial_demod(samples, firstsample + phase_incr, Symbolsamples,
                                 c->symbolclocks, nsymbols, 0., NULL)) > maxenergy) {
          maxenergy = e; firstsample += phase_incr; nochange = 0;
        } else if ((e = orc_trial_demod(samples, firstsample - phase_incr, Symbolsamples,
                                        c->symbolclocks, nsymbols, 0., NULL)) > maxenergy) {
          maxenergy = e; firstsample += phase_incr;   /* sic: symdemod.c:164-166 adds, not subtracts */
          phase_incr = -phase_incr; nochange = 0;
        } else nochange++;
      }
      nsymbols = (int)(window * Symrate);
    }
    if (nw < logcap) {
      if (symphase_log) symphase_log[nw] = symphase;
      if (energy_log) energy_log[nw] = maxenergy;
    }
    nw++;

    double gain = 100. / sqrt(maxenergy);
    if (nout + (size_t)nsymbols > outcap) break;
    orc_trial_demod(samples, firstsample, Symbolsamples, c->symbolclocks, nsymbols, gain, out + nout);
    nout += (size_t)nsymbols;
    firstsample = (int)(firstsample + nsymbols * Symbolsamples);
  }
  free(samples);
  if (nwindows) *nwindows = nw;
  return nout;
}
