/* Test harness: the PRODUCT's symdemod host logic (cli/symdemod_core.c) on a plain CPU engine, so the
 * buffer / boundary / selection logic can be checked against the reference's symdemod output
 * without a GPU.  TEST INFRASTRUCTURE ONLY. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../isee3-decoder_amd/cli/symdemod_core.h"

typedef struct { int16_t *s; int n, cap; } cpu_t;
static void *c_create(int cap) { cpu_t *c = calloc(1, sizeof *c); c->s = calloc((size_t)cap, sizeof(int16_t)); c->cap = cap; return c; }
static int c_load(void *h, const int16_t *s, int n) { cpu_t *c = h; memcpy(c->s, s, sizeof(int16_t) * (size_t)n); c->n = n; return 0; }
static long long seg(const cpu_t *c, int a, int b) { long long v = 0; for (int i = a; i < b; i++) v += c->s[i]; return v; }
static int c_ts(void *h, int lo, const int *sw, int sc, int ns, int noff, double *en) {
  cpu_t *c = h;
  for (int t = 0; t < noff; t++) {
    double e = 0; int k = 0;
    for (int i = 0; i < ns; i++) {
      long long sym = 0;
      for (int j = 0; j < sc; j++, k += 2)
        sym += -seg(c, lo + t + sw[k], lo + t + sw[k + 1]) + seg(c, lo + t + sw[k + 1], lo + t + sw[k + 2]);
      e += sym * sym;
    }
    en[t] = e;
  }
  return 0;
}
static int c_demod(void *h, const int *ed, int sc, int ns, double gain, uint8_t *out, double *esum) {
  cpu_t *c = h; double e = 0; int k = 0;
  for (int i = 0; i < ns; i++) {
    long long v = 0;
    for (int j = 0; j < sc; j++, k += 2) v += -seg(c, ed[k], ed[k + 1]) + seg(c, ed[k + 1], ed[k + 2]);
    if (gain != 0 && out) { double s = gain * v + 128; if (s > 255) s = 255; else if (s < 0) s = 0; out[i] = (unsigned char)s; }
    e += v * v;
  }
  if (esum) *esum = e;
  return 0;
}
static void c_destroy(void *h) { cpu_t *c = h; free(c->s); free(c); }
/* the engine-resident window buffer (symdemod_run_blk): plain memmove / memcpy on the CPU engine's own copy */
static int c_slide(void *h, int slide, int n) { cpu_t *c = h; memmove(c->s, c->s + slide, sizeof(int16_t) * (size_t)(n - slide)); return 0; }
static int c_put(void *h, int at, const int16_t *src, int n, int dev) { cpu_t *c = h; (void)dev; if (at + n > c->cap) return -1; memcpy(c->s + at, src, sizeof(int16_t) * (size_t)n); return 0; }
static int c_scan(void *h, int n) { cpu_t *c = h; c->n = n; return 0; }
/* the fused window call (symdemod_core.h `window`): search, first maximum, demodulation from the speculated table.
 * SYMD_WINDOW_MISS=n: every n-th call answers 1 ("take the step-by-step calls"), as the product does for an adjustment
 * outside its tables; calls and misses are reported on stderr at exit. */
#include <math.h>
static long g_win_calls, g_win_done;
static void win_report(void) { fprintf(stderr, "WINDOW calls=%ld done=%ld\n", g_win_calls, g_win_done); }
static int c_window(void *h, int fs, const int *sw, int sc, int ns, int fo, int noff, const int *ed, int lo, int nspec,
                    uint8_t *out, int *ph, double *me) {
  cpu_t *c = h;
  const int nsw = 2 * sc * ns + 1, miss_every = getenv("SYMD_WINDOW_MISS") ? atoi(getenv("SYMD_WINDOW_MISS")) : 0;
  if (g_win_calls++ == 0) atexit(win_report);
  if (miss_every && g_win_calls % miss_every == 0) return 1;
  if (fs + fo < 0 || fs + fo + noff - 1 + sw[nsw - 1] > c->n) return 1;
  double *en = malloc(sizeof(double) * (size_t)noff);
  c_ts(h, fs + fo, sw, sc, ns, noff, en);
  int bi = 0;
  for (int t = 1; t < noff; t++) if (en[t] > en[bi]) bi = t;
  const int j = fo + bi - lo;
  const double best = en[bi];
  free(en);
  if (j < 0 || j >= nspec) return 1;
  const int *e = ed + (size_t)j * nsw;
  if (e[0] < 0 || e[nsw - 1] > c->n) return 1;
  *ph = fo + bi; *me = best / ns;
  c_demod(h, e, sc, ns, 100. / sqrt(*me), out, NULL);
  g_win_done++;
  return 0;
}
/* block views of stdin in pieces of an awkward size */
static int16_t g_blk[7919];
static long blk_next(void *ctx, const int16_t **blk, int *is_dev, long max) {
  (void)ctx;
  long want = max < 7919 ? max : 7919, got = 0;
  while (got < want * 2) { long k = (long)fread((char *)g_blk + got, 1, (size_t)(want * 2 - got), stdin); if (k <= 0) break; got += k; }
  *blk = g_blk; *is_dev = 0;
  return got / 2;
}

int main(int argc, char **argv) {
  symdemod_opts o;
  symdemod_parse_args(&o, argc, argv);
  const int win = getenv("SYMD_WINDOW") && atoi(getenv("SYMD_WINDOW"));
  if (getenv("SYMD_STORE") && atoi(getenv("SYMD_STORE"))) {
    symdemod_engine e = { c_create, c_load, c_ts, c_demod, c_destroy, c_slide, c_put, c_scan, win ? c_window : NULL };
    return symdemod_run_blk(&o, &e, blk_next, NULL, stdout, stderr) ? 2 : 0;
  }
  symdemod_engine e = { c_create, c_load, c_ts, c_demod, c_destroy, NULL, NULL, NULL, win ? c_window : NULL };
  return symdemod_run(&o, &e, 0, stdout, stderr) ? 2 : 0;
}
