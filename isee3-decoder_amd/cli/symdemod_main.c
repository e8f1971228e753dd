/* symdemod -- Manchester integrate-and-dump symbol demodulator pipe stage on MI355X.
 * Drop-in for reference symdemod.c: options -w s -c Hz -r Hz -q -t -C n; int16 baseband on stdin,
 * uint8 offset-128 soft symbols on stdout (byte-identical).  Kernels: libisee3dsp_hip.so. */
#include <locale.h>
#include <stdio.h>
#include <stdlib.h>
#include "symdemod_core.h"
#include "../../include/isee3_dsp_hip.h"

static void *eng_create(int n) { return symd_create(n); }
static int eng_load(void *h, const int16_t *s, int n) { return symd_load(h, s, n, 0); }
static int eng_ts(void *h, int lo, const int *sw, int sc, int ns, int noff, double *en) {
  return symd_timesearch(h, lo, sw, sc, ns, noff, en);
}
static int eng_demod(void *h, const int *edges, int sc, int ns, double gain, uint8_t *out, double *esum) {
  return symd_demod(h, edges, sc, ns, gain, out, 0, esum);
}
static void eng_destroy(void *h) { symd_destroy(h); }
static int eng_window(void *h, int fs, const int *sw, int sc, int ns, int fo, int noff, const int *ed, int lo, int nspec,
                      uint8_t *out, int *ph, double *me) {
  return symd_window(h, fs, sw, sc, ns, fo, noff, ed, lo, nspec, out, ph, me);
}

int main(int argc, char **argv) {
  symdemod_opts o;
  const char *lang = getenv("LANG");
  setlocale(LC_ALL, lang ? lang : "en_US.utf8");
  symdemod_parse_args(&o, argc, argv);
  symdemod_engine e = { eng_create, eng_load, eng_ts, eng_demod, eng_destroy, NULL, NULL, NULL, eng_window };
  if (getenv("SYMDEMOD_STEPWISE") && atoi(getenv("SYMDEMOD_STEPWISE"))) e.window = NULL;     /* the two-round-trip path */
  if (symdemod_run(&o, &e, 0, stdout, stderr) != 0) {
    fprintf(stderr, "%s: engine failed: %s\n", o.argv0, isee3dsp_last_error());
    return 2;
  }
  return 0;
}
