/* pmdemod -- PM carrier search / spin-down pipe stage on MI355X.
 * Drop-in for reference pmdemod.c: options -S Hz -W Hz -D Hz/s -t dB -q -b Hz -f -r Hz [file];
 * int16 LE (I,Q) pairs in, int16 baseband out.  Kernels: libisee3dsp_hip.so (double precision). */
#include <errno.h>
#include <locale.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include "pmdemod_core.h"
#include "../../include/isee3_dsp_hip.h"

static void *eng_create(int n) { return pmd_create(n); }
static int eng_dechirp(void *h, const double *t) { return pmd_set_dechirp(h, t); }
static int eng_load(void *h, const int16_t *iq, int flip) { return pmd_load(h, iq, 0, flip); }
static int eng_peak(void *h, int a, int b, pmdemod_peak *out) {
  pmd_peak p;
  if (pmd_fft_peak(h, a, b, &p) != 0) return -1;
  out->peak = p.peak; out->maxenergy = p.maxenergy; out->peak_re = p.peak_re; out->peak_im = p.peak_im;
  out->next_re = p.next_re; out->next_im = p.next_im; out->prev_re = p.prev_re; out->prev_im = p.prev_im;
  return 0;
}
static int eng_mix(void *h, double cstep, pmdemod_mix *r, int16_t *out16) {
  pmd_mix m;
  if (pmd_mix_quantise(h, cstep, &m, out16, NULL, 0) != 0) return -1;
  r->dc_re = m.dc_re; r->dc_im = m.dc_im; r->amplitude = m.amplitude; r->diffsumsq = m.diffsumsq;
  return 0;
}
static void eng_destroy(void *h) { pmd_destroy(h); }

int main(int argc, char **argv) {
  pmdemod_opts o;
  FILE *in = stdin;
  struct stat sb;
  const char *lang = getenv("LANG");
  setlocale(LC_ALL, lang ? lang : "en_US.utf8");
  int rc = pmdemod_parse_args(&o, argc, argv, stderr);
  if (rc) return rc;
  if (o.file && (in = fopen(o.file, "r")) == NULL) {
    fprintf(stderr, "%s: Can't read %s; %s\n", o.argv0, o.file, strerror(errno));
    return 1;
  }
  if (isatty(fileno(in))) { fprintf(stderr, "%s: Can't read from terminal\n", o.argv0); return 1; }
  if (fstat(fileno(in), &sb) == -1) { fprintf(stderr, "%s: fstat of input failed: %s\n", o.argv0, strerror(errno)); return 1; }
  if (S_ISDIR(sb.st_mode)) { fprintf(stderr, "%s: Can't read a directory\n", o.argv0); return 1; }
  pmdemod_engine e = { eng_create, eng_dechirp, eng_load, eng_peak, eng_mix, eng_destroy };
  rc = pmdemod_run(&o, &e, in, stdout, stderr, NULL, 0, NULL);
  if (rc == 2) fprintf(stderr, "%s: engine: %s\n", o.argv0, isee3dsp_last_error());
  if (in != stdin) fclose(in);
  return rc;
}
