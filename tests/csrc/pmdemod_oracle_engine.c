/* Test harness: the PRODUCT's pmdemod host logic (cli/pmdemod_core.c) on a CPU engine built from the
 * oracle's FFT and the reference's own per-sample recurrences.  TEST INFRASTRUCTURE ONLY. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../isee3-decoder_amd/cli/pmdemod_core.h"
#include "../../oracle/oracle.h"

typedef struct { int N; double *buf, *spec, *lo; } cpu_t;
static void *c_create(int N) { cpu_t *c = calloc(1, sizeof *c); c->N = N; c->buf = malloc(16 * (size_t)N); c->spec = malloc(16 * (size_t)N); return c; }
static int c_dechirp(void *h, const double *t) { cpu_t *c = h; c->lo = malloc(16 * (size_t)c->N); memcpy(c->lo, t, 16 * (size_t)c->N); return 0; }
static int c_load(void *h, const int16_t *iq, int flip) {
  cpu_t *c = h;
  for (int i = 0; i < c->N; i++) {
    double x = flip ? iq[2 * i + 1] : iq[2 * i], y = flip ? iq[2 * i] : iq[2 * i + 1];
    if (c->lo) { double pr = c->lo[2 * i], pi = -c->lo[2 * i + 1]; double nx = x * pr - y * pi, ny = x * pi + y * pr; x = nx; y = ny; }
    c->buf[2 * i] = x; c->buf[2 * i + 1] = y;
  }
  return 0;
}
static int c_peak(void *h, int a, int b, pmdemod_peak *o) {
  cpu_t *c = h; int N = c->N;
  orc_fft_forward(c->buf, c->spec, N);
  int peak = -1; double mx = 0;
  for (int i = a; i < b; i++) { double e = c->spec[2 * i] * c->spec[2 * i] + c->spec[2 * i + 1] * c->spec[2 * i + 1]; if (e >= mx) { mx = e; peak = i; } }
  o->peak = peak; o->maxenergy = mx;
  if (peak >= 0) {
    int nx = (peak + 1) % N, pv = (N + peak - 1) % N;
    o->peak_re = c->spec[2 * peak]; o->peak_im = c->spec[2 * peak + 1];
    o->next_re = c->spec[2 * nx]; o->next_im = c->spec[2 * nx + 1];
    o->prev_re = c->spec[2 * pv]; o->prev_im = c->spec[2 * pv + 1];
  }
  return 0;
}
static int c_mix(void *h, double cstep, pmdemod_mix *r, int16_t *out16) {
  cpu_t *c = h; int N = c->N;
  double sr = cos(cstep), si = -sin(cstep), cr = 1, ci = 0, dcr = 0, dci = 0;
  for (int i = 0; i < N; i++) {
    double x = c->buf[2 * i], y = c->buf[2 * i + 1], nx = x * cr - y * ci, ny = x * ci + y * cr;
    c->buf[2 * i] = nx; c->buf[2 * i + 1] = ny; dcr += nx; dci += ny;
    double ncr = cr * sr - ci * si, nci = cr * si + ci * sr; cr = ncr; ci = nci;
  }
  dcr /= N; dci /= N;
  double amp = hypot(dcr, dci), ur = dcr / amp, ui = -dci / amp, ds = 0;
  for (int i = 0; i < N; i++) {
    double x = c->buf[2 * i], y = c->buf[2 * i + 1], nx = x * ur - y * ui, ny = x * ui + y * ur;
    ds += (nx - amp) * (nx - amp);
    out16[i] = (short)(ny * M_SQRT1_2);
  }
  r->dc_re = dcr; r->dc_im = dci; r->amplitude = amp; r->diffsumsq = ds / N;
  return 0;
}
static void c_destroy(void *h) { cpu_t *c = h; free(c->buf); free(c->spec); free(c->lo); free(c); }

/* the enqueue / collect halves (pmdemod_core.h): "enqueue" computes at once and parks the result in a side table keyed by
 * the handle, "collect" hands it out; every call leaves a letter on stderr (B / E peak begin / end, M / N mix begin / end,
 * S = a block sent back to the synchronous mix) so that the test can read the order the core issued them in.
 * PMD_ASYNC_SYNCMIX=n: every n-th mix_begin answers 1. */
static struct { void *h; pmdemod_peak pk; pmdemod_mix mx; } g_park[2];
static int park_of(void *h) { for (int i = 0; i < 2; i++) if (g_park[i].h == h || !g_park[i].h) { g_park[i].h = h; return i; } return 0; }
static long g_mixes;
static int c_peak_begin(void *h, int a, int b) { fprintf(stderr, "B%d ", park_of(h)); return c_peak(h, a, b, &g_park[park_of(h)].pk); }
static int c_peak_end(void *h, pmdemod_peak *o) { fprintf(stderr, "E%d ", park_of(h)); *o = g_park[park_of(h)].pk; return 0; }
static int c_mix_begin(void *h, double cstep, int16_t *out16, int dev) {
  (void)dev;
  const int every = getenv("PMD_ASYNC_SYNCMIX") ? atoi(getenv("PMD_ASYNC_SYNCMIX")) : 0;
  if (every && ++g_mixes % every == 0) { fprintf(stderr, "S%d ", park_of(h)); return 1; }
  fprintf(stderr, "M%d ", park_of(h));
  return c_mix(h, cstep, &g_park[park_of(h)].mx, out16);
}
static int c_mix_end(void *h, pmdemod_mix *r) { fprintf(stderr, "N%d ", park_of(h)); *r = g_park[park_of(h)].mx; return 0; }

int main(int argc, char **argv) {
  pmdemod_opts o;
  int rc = pmdemod_parse_args(&o, argc, argv, stderr);
  if (rc) return rc;
  pmdemod_engine e = { c_create, c_dechirp, c_load, c_peak, c_mix, c_destroy };
  if (getenv("PMD_ASYNC") && atoi(getenv("PMD_ASYNC"))) {
    e.fft_peak_begin = c_peak_begin; e.fft_peak_end = c_peak_end; e.mix_begin = c_mix_begin; e.mix_end = c_mix_end;
  }
  pmdemod_block_report rep[64]; int n = 0;
  rc = pmdemod_run(&o, &e, stdin, stdout, stderr, rep, 64, &n);
  fprintf(stderr, "\n");
  for (int i = 0; i < n; i++) fprintf(stderr, "REPORT %d %.17g %.17g\n", rep[i].peak, rep[i].carrier_freq, rep[i].cn0);
  return rc;
}
