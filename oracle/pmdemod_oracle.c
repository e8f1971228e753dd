/* pmdemod_oracle.c -- CPU restatement of the pmdemod pipe stage (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED at the FFT boundary: pmdemod.c needs FFTW3 (pmdemod.c:17,161,253; reference
 * Makefile:66, version unpinned, not vendored), which this image lacks, so the reference stage
 * cannot be compiled here and the reference holds no fixture for it.  FFTW's published contract
 * for fftw_plan_dft_1d(N,in,out,FFTW_FORWARD,..): out[k] = sum_j in[j] exp(-2 pi i jk/N),
 * unnormalised, out-of-place with the input preserved.  Any correct double FFT agrees with it to
 * ~1e-15 relative, far inside the 1e-9 tolerance; tests cross-check orc_fft_forward against
 * numpy.fft.  Everything around the FFT follows pmdemod.c line by line:
 *   FFT size :129-131, load/flip :204-230, de-chirp :232-244, search window :255-285,
 *   peak (">=": last maximum wins) :288-298, Quinn-2 :43-46,299-318, spin-down recurrence
 *   :321-336, rotate + C/N0 :337-354, quantise :360-368.
 * Complex products are written out as C99 does for finite operands:
 *   (a+jb)(c+jd) = (ac-bd) + j(ad+bc); build with -ffp-contract=off.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

void orc_pmdemod_default(orc_pmdemod_cfg *c) {
  memset(c, 0, sizeof *c);
  c->samprate = 250000; c->binsize = 4; c->cn0_threshold = 21;   /* pmdemod.c:76-83 */
}

int orc_pmdemod_fftsize(const orc_pmdemod_cfg *c) {
  int l = (int)nearbyint(log2(c->samprate / c->binsize));
  return 1 << l;
}

/* iterative radix-2 DIT, twiddles from direct sin/cos of each angle (no recurrence) */
void orc_fft_forward(const double *in, double *out, int n) {
  int lg = 0; while ((1 << lg) < n) lg++;
  for (int i = 0; i < n; i++) {
    unsigned r = 0;
    for (int b = 0; b < lg; b++) r |= ((unsigned)(i >> b) & 1u) << (lg - 1 - b);
    out[2 * r] = in[2 * i]; out[2 * r + 1] = in[2 * i + 1];
  }
  double *tw = malloc(sizeof(double) * (size_t)n);       /* n/2 complex twiddles */
  for (int k = 0; k < n / 2; k++) {
    double a = -2.0 * M_PI * k / n;
    tw[2 * k] = cos(a); tw[2 * k + 1] = sin(a);
  }
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1, stride = n / len;
    for (int base = 0; base < n; base += len)
      for (int k = 0; k < half; k++) {
        double wr = tw[2 * k * stride], wi = tw[2 * k * stride + 1];
        double *u = out + 2 * (base + k), *v = out + 2 * (base + k + half);
        double tr = v[0] * wr - v[1] * wi, ti = v[0] * wi + v[1] * wr;
        v[0] = u[0] - tr; v[1] = u[1] - ti;
        u[0] += tr; u[1] += ti;
      }
  }
  free(tw);
}

static double tau(double x) {                              /* pmdemod.c:43-46 */
  return 0.25 * log(3 * x * x + 6 * x + 1)
       - sqrt(6.) / 24 * log((x + 1 - sqrt(2 / 3.)) / (x + 1 + sqrt(2 / 3.)));
}

size_t orc_pmdemod(const orc_pmdemod_cfg *c, const int16_t *iq, size_t nsamp,
                   int16_t *out, double *pre, orc_pmdemod_blk *blk, int blkcap, int *nblk) {
  double Samprate = c->samprate, Search_width = fabs(c->search_width);
  double Carrier_search_freq = c->search_freq;
  if (Search_width > Samprate / 2) Search_width = Samprate / 2;
  int N = orc_pmdemod_fftsize(c);
  double Binsize = Samprate / N;
  double cn0 = -999;
  double drate = c->doppler_rate * 2 * M_PI / (Samprate * Samprate);
  double acc_r = cos(drate), acc_i = sin(drate);          /* loaccel, pmdemod.c:144 */
  double *buf = malloc(sizeof(double) * 2 * (size_t)N);
  double *spec = malloc(sizeof(double) * 2 * (size_t)N);
  size_t nb = nsamp / (size_t)N, produced = 0;

  for (size_t b = 0; b < nb; b++) {
    const int16_t *s = iq + 2 * b * (size_t)N;
    for (int i = 0; i < N; i++) {
      if (!c->flip) { buf[2 * i] = s[2 * i]; buf[2 * i + 1] = s[2 * i + 1]; }
      else          { buf[2 * i] = s[2 * i + 1]; buf[2 * i + 1] = s[2 * i]; }
    }
    if (c->doppler_rate != 0) {                            /* pmdemod.c:232-244 */
      double pr = 1, pi = 0, fr = 1, fi = 0;
      for (int i = 0; i < N; i++) {
        double x = buf[2 * i], y = buf[2 * i + 1];
        buf[2 * i]     = x * pr - y * (-pi);               /* * conj(lophase) */
        buf[2 * i + 1] = x * (-pi) + y * pr;
        double nfr = fr * acc_r - fi * acc_i, nfi = fr * acc_i + fi * acc_r;
        fr = nfr; fi = nfi;
        double npr = pr * fr - pi * fi, npi = pr * fi + pi * fr;
        pr = npr; pi = npi;
      }
    }
    orc_fft_forward(buf, spec, N);

    int firstbin, lastbin;
    if (Search_width != 0 && cn0 > c->cn0_threshold) {     /* pmdemod.c:257-272 */
      if (Carrier_search_freq - Search_width <= -Samprate / 2) firstbin = 0;
      else { firstbin = (int)((Carrier_search_freq - Search_width) / Binsize); if (firstbin < 0) firstbin += N; }
      if (Carrier_search_freq + Search_width >= Samprate / 2) lastbin = N / 2 - 1;
      else { lastbin = (int)((Carrier_search_freq + Search_width) / Binsize); if (lastbin < 0) lastbin += N; }
    } else { firstbin = 0; lastbin = N; }
    if (firstbin > lastbin) { int t = firstbin; firstbin = lastbin; lastbin = t; }

    int peak = -1; double maxenergy = 0;
    for (int i = firstbin; i < lastbin; i++) {
      double e = spec[2 * i] * spec[2 * i] + spec[2 * i + 1] * spec[2 * i + 1];
      if (e >= maxenergy) { maxenergy = e; peak = i; }
    }
    if (peak < 0) break;                                   /* reference asserts */
    int next = (peak + 1) % N, prev = (N + peak - 1) % N;
    double ap = (spec[2 * next] * spec[2 * peak] + spec[2 * next + 1] * spec[2 * peak + 1]) / maxenergy;
    double dp = -ap / (1 - ap);
    double am = (spec[2 * prev] * spec[2 * peak] + spec[2 * prev + 1] * spec[2 * peak + 1]) / maxenergy;
    double dm = am / (1 - am);
    double d = (dp + dm) / 2 + tau(dp * dp) - tau(dm * dm);
    double carrier_freq = Binsize * (peak + d);
    if (carrier_freq > Samprate / 2) carrier_freq -= Samprate;

    double cstep = 2 * M_PI * carrier_freq / Samprate;
    double sr = cos(cstep), si = -sin(cstep);              /* cpstep */
    double cr = 1, ci = 0, dcr = 0, dci = 0;
    for (int i = 0; i < N; i++) {                          /* pmdemod.c:332-335 */
      double x = buf[2 * i], y = buf[2 * i + 1];
      double nx = x * cr - y * ci, ny = x * ci + y * cr;
      buf[2 * i] = nx; buf[2 * i + 1] = ny;
      dcr += nx; dci += ny;
      double ncr = cr * sr - ci * si, nci = cr * si + ci * sr;
      cr = ncr; ci = nci;
    }
    dcr /= N; dci /= N;
    double amp = hypot(dcr, dci);                          /* cabs */
    double ur = dcr / amp, ui = -dci / amp;                /* conj(dc)/amp */
    double diffsumsq = 0;
    for (int i = 0; i < N; i++) {
      double x = buf[2 * i], y = buf[2 * i + 1];
      double nx = x * ur - y * ui, ny = x * ui + y * ur;
      buf[2 * i] = nx; buf[2 * i + 1] = ny;
      diffsumsq += (nx - amp) * (nx - amp);
    }
    diffsumsq /= N;
    cn0 = 10 * log10(Samprate * amp * amp / (2 * diffsumsq));
    if (cn0 > c->cn0_threshold) Carrier_search_freq = carrier_freq;

    for (int i = 0; i < N; i++) {
      double v = buf[2 * i + 1] * M_SQRT1_2;
      if (pre) pre[b * (size_t)N + i] = v;
      out[b * (size_t)N + i] = (short)v;
    }
    if (blk && (int)b < blkcap) {
      blk[b].peak = peak; blk[b].carrier_freq = carrier_freq; blk[b].cn0 = cn0; blk[b].amplitude = amp;
    }
    produced += (size_t)N;
  }
  free(buf); free(spec);
  if (nblk) *nblk = (int)nb;
  return produced;
}
