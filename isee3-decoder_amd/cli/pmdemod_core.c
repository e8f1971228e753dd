/* pmdemod_core.c -- see pmdemod_core.h.  Plain C11 (build with -ffp-contract=off). */
#define _GNU_SOURCE
#include "pmdemod_core.h"
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include "timefmt.h"

void pmdemod_default_opts(pmdemod_opts *o) {
  memset(o, 0, sizeof *o);
  o->samprate = 250000; o->binsize = 4; o->cn0_threshold = 21;   /* pmdemod.c:76-83 */
  o->argv0 = "pmdemod";
}

int pmdemod_parse_args(pmdemod_opts *o, int argc, char **argv, FILE *err) {
  int c;
  pmdemod_default_opts(o);
  if (argc > 0) o->argv0 = argv[0];
  optind = 1; opterr = 0;
  while ((c = getopt(argc, argv, "S:W:D:r:fb:qt:")) != -1) {
    switch (c) {
    case 'S': o->search_freq = atof(optarg); break;
    case 'W': o->search_width = atof(optarg); break;
    case 'D': o->doppler_rate = atof(optarg); break;
    case 't': o->cn0_threshold = atof(optarg); break;
    case 'q': o->quiet = 1; break;
    case 'b': o->binsize = atof(optarg); break;
    case 'r': o->samprate = atof(optarg); break;
    case 'f': o->flip++; break;
    default:
      fprintf(err, "%s: unknown option %c\n", o->argv0, c);
      return 1;                                      /* pmdemod.c:111-113 */
    }
  }
  if (argc > optind) o->file = argv[optind];
  return 0;
}

static double tau(double x) {                        /* Quinn's second estimator, pmdemod.c:43-46 */
  return 0.25 * log(3 * x * x + 6 * x + 1)
       - sqrt(6.) / 24 * log((x + 1 - sqrt(2 / 3.)) / (x + 1 + sqrt(2 / 3.)));
}

/* FILE-based source and sink: the stand-alone pipe stage */
typedef struct { FILE *f; int16_t *buf; } file_src;
static int file_next(void *p, int N, const int16_t **blk, int *is_dev) {
  file_src *s = p;
  if (!s->buf && !(s->buf = malloc(sizeof(int16_t) * 2 * (size_t)N))) return -1;
  if (fread(s->buf, 4, (size_t)N, s->f) < (size_t)N) return 0;
  *blk = s->buf; *is_dev = 0;
  return 1;
}
typedef struct { FILE *f; int16_t *buf; } file_dst;
static int16_t *file_acquire(void *p, int N, int *is_dev) {
  file_dst *d = p;
  if (!d->buf) d->buf = malloc(sizeof(int16_t) * (size_t)N);
  *is_dev = 0;
  return d->buf;
}
static int file_commit(void *p, int16_t *buf, int N) {
  file_dst *d = p;
  fwrite(buf, sizeof(int16_t), (size_t)N, d->f);
  fflush(d->f);
  return 0;
}

int pmdemod_run(const pmdemod_opts *o, const pmdemod_engine *e, FILE *in, FILE *out, FILE *err,
                pmdemod_block_report *report, int report_cap, int *nreport) {
  file_src fs = { in, NULL };
  file_dst fd = { out, NULL };
  pmdemod_source src = { file_next, &fs };
  pmdemod_sink dst = { file_acquire, file_commit, &fd };
  int rc = pmdemod_run_io(o, e, &src, &dst, in, err, report, report_cap, nreport);
  free(fs.buf); free(fd.buf);
  return rc;
}

int pmdemod_run_io(const pmdemod_opts *o, const pmdemod_engine *e, const pmdemod_source *src, const pmdemod_sink *dst,
                   FILE *in, FILE *err, pmdemod_block_report *report, int report_cap, int *nreport) {
  double Samprate = o->samprate, Search_width = o->search_width, Carrier_search_freq = o->search_freq;
  double cn0 = -999;
  int exitcode = 0, nb = 0;
  void *h = NULL;
  double *lo = NULL;
  long long total_samples = 0;

  if (nreport) *nreport = 0;
  if (fabs(Carrier_search_freq) > Samprate / 2) {
    fprintf(err, "%s: Carrier frequency of %'.1lf Hz is outside Nyquist bandwidth at %'.1lf Hz sample rate. Must be between +/- %'.1lf Hz\n",
            o->argv0, Carrier_search_freq, Samprate, Samprate / 2);
    return 1;
  }
  if (Search_width < 0) Search_width = fabs(Search_width);
  if (Search_width > Samprate / 2) {
    fprintf(err, "%s: Search width > 1/2 Nyquist rate; reduced to +/- %'.1lf Hz\n", o->argv0, Samprate / 2);
    Search_width = Samprate / 2;
  }
  int lfftsize = (int)nearbyint(log2(Samprate / o->binsize));
  int N = 1 << lfftsize;
  double Binsize = Samprate / N;
  if (!o->quiet)
    fprintf(err, "%s: FFT bin size %'.4lf Hz; Start carrier %'.4lf Hz; Doppler %'.6lf Hz/s; Search range +/-%'.1lf Hz\n",
            o->argv0, Samprate / N, Carrier_search_freq, o->doppler_rate, Search_width);
  if (o->flip && !o->quiet) fprintf(err, "%s: I & Q samples swapped (spectrum inverted)\n", o->argv0);

  h = e->create(N);
  if (!h) {
    fprintf(err, "%s: cannot set up a %d-point FFT\n", o->argv0, N);
    exitcode = 2;
    goto done;
  }
  if (o->doppler_rate != 0) {
    /* pmdemod.c:140-145,232-244: the LO restarts every block, so its phase sequence is a fixed
       table; it is produced with the reference's own sequential recurrence (its rounding walk is
       what the reference computes) and handed to the engine once */
    double drate = o->doppler_rate * 2 * M_PI / (Samprate * Samprate);
    double complex loaccel = cos(drate) + _Complex_I * sin(drate);
    double complex lofreq = 1, lophase = 1;
    lo = malloc(sizeof(double) * 2 * (size_t)N);
    if (!lo) { exitcode = 2; goto done; }
    for (int i = 0; i < N; i++) {
      lo[2 * i] = creal(lophase); lo[2 * i + 1] = cimag(lophase);
      lofreq *= loaccel;
      lophase *= lofreq;
    }
    if (e->set_dechirp(h, lo) != 0) { exitcode = 2; goto done; }
  }
  {
    struct stat sb;
    if (!o->quiet && in && fstat(fileno(in), &sb) == 0 && S_ISREG(sb.st_mode)) {
      long long nsamples = sb.st_size / 4;
      fprintf(err, "%s: demodulating %'lld bytes; %'lld samples; %'.2lf sec @ %'.1lf Hz\n", o->argv0,
              (long long)sb.st_size, nsamples, nsamples / Samprate, Samprate);
    }
  }

  if (Search_width == 0 && e->fft_peak_begin && e->fft_peak_end && e->mix_begin && e->mix_end &&
      !(getenv("PMDEMOD_SERIAL") && atoi(getenv("PMDEMOD_SERIAL")))) {
    /* Independent blocks (full-band search, pmdemod.c:273-276: nothing of a block's result enters the next one's search),
       two handles A / B used in turn.  Engine queue: FFT(0) FFT(1) mix(0) FFT(2) mix(1) ... -- block k+1's transform is
       enqueued before block k's peak is waited for, so the engine works while the host forms Quinn's estimate and the
       carrier parameters; block k-1 (sums, status line, commit) is finished in iteration k.  Same calls, same values,
       same order of output as the loop below. */
    void *hh[2] = { h, e->create(N) };
    struct { int active, async_mix, peak; double carrier_freq; int16_t *out16; pmdemod_mix mx; } b[2];
    memset(b, 0, sizeof b);
    int k = 0, more = 1, bad = hh[1] == NULL || (lo && e->set_dechirp(hh[1], lo) != 0);
    const int16_t *iq = NULL; int in_dev = 0;
    if (!bad) {
      int got = src->next(src->ctx, N, &iq, &in_dev);
      if (got == 0) more = 0;
      else if (got < 0 || (in_dev && !e->load_dev) || (in_dev ? e->load_dev(hh[0], iq, o->flip ? 1 : 0) : e->load(hh[0], iq, o->flip ? 1 : 0)) != 0 ||
               e->fft_peak_begin(hh[0], 0, N) != 0) bad = 1;
    }
    while (!bad && more) {
      const int cur = k & 1, nxt = cur ^ 1;
      int have_next = 0;
      {                                               /* (1) the next block's transform goes into the queue */
        int got = src->next(src->ctx, N, &iq, &in_dev);
        if (got < 0 || (got > 0 && in_dev && !e->load_dev)) { bad = 1; break; }
        if (got > 0) {
          if ((in_dev ? e->load_dev(hh[nxt], iq, o->flip ? 1 : 0) : e->load(hh[nxt], iq, o->flip ? 1 : 0)) != 0 ||
              e->fft_peak_begin(hh[nxt], 0, N) != 0) { bad = 1; break; }
          have_next = 1;
        }
      }
      pmdemod_peak pk;                                /* (2) this block's peak, Quinn's estimate (pmdemod.c:299-318) */
      if (e->fft_peak_end(hh[cur], &pk) != 0 || pk.peak < 0) { bad = 1; break; }
      double ap = (pk.next_re * pk.peak_re + pk.next_im * pk.peak_im) / pk.maxenergy;
      double dp = -ap / (1 - ap);
      double am = (pk.prev_re * pk.peak_re + pk.prev_im * pk.peak_im) / pk.maxenergy;
      double dm = am / (1 - am);
      double d = (dp + dm) / 2 + tau(dp * dp) - tau(dm * dm);
      double carrier_freq = Binsize * (pk.peak + d);
      if (carrier_freq > Samprate / 2) carrier_freq -= Samprate;
      double cstep = 2 * M_PI * carrier_freq / Samprate;
      for (int pass = 0; pass < 2 && !bad; pass++) {  /* (3) the previous block leaves, (4) this block's spin-down is enqueued; at the end: (3) for this block too */
        const int j = pass == 0 ? nxt : cur;
        if (pass == 1) {
          int out_dev = 0;
          int16_t *out16 = dst->acquire(dst->ctx, N, &out_dev);
          if (!out16 || (out_dev && !e->mix_dev)) { bad = 1; break; }
          int r = e->mix_begin(hh[cur], cstep, out16, out_dev);
          if (r < 0 || (r == 1 && (out_dev ? e->mix_dev(hh[cur], cstep, &b[cur].mx, out16) : e->mix(hh[cur], cstep, &b[cur].mx, out16)) != 0)) { bad = 1; break; }
          b[cur].active = 1; b[cur].async_mix = r == 0; b[cur].peak = pk.peak; b[cur].carrier_freq = carrier_freq; b[cur].out16 = out16;
          if (have_next) break;                       /* it is finished in the next iteration */
        }
        if (!b[j].active) continue;
        if (b[j].async_mix && e->mix_end(hh[j], &b[j].mx) != 0) { bad = 1; break; }
        cn0 = 10 * log10(Samprate * b[j].mx.amplitude * b[j].mx.amplitude / (2 * b[j].mx.diffsumsq));
        if (!o->quiet)
          fprintf(err, "%s: sample %'lld (%'.3lf sec, %s); carrier %'.1lf Hz; C/No = %'.2lf dB%s\n", o->argv0,
                  total_samples, total_samples / Samprate, isee3_format_hms(total_samples / Samprate), b[j].carrier_freq,
                  cn0, cn0 >= o->cn0_threshold ? " locked" : "");
        if (report && nb < report_cap) { report[nb].peak = b[j].peak; report[nb].carrier_freq = b[j].carrier_freq; report[nb].cn0 = cn0; }
        nb++;
        if (dst->commit(dst->ctx, b[j].out16, N) != 0) { bad = 1; break; }
        total_samples += N;
        b[j].active = 0;
      }
      more = have_next;
      k++;
    }
    if (bad) exitcode = 2;
    if (hh[1]) e->destroy(hh[1]);
    goto done;
  }

  for (;;) {
    /* a whole block or nothing: the remainder of the input is dropped (pmdemod.c:206-216) */
    const int16_t *iq = NULL; int in_dev = 0;
    int got = src->next(src->ctx, N, &iq, &in_dev);
    if (got == 0) break;
    if (got < 0 || (in_dev && !e->load_dev)) { exitcode = 2; break; }
    if ((in_dev ? e->load_dev(h, iq, o->flip ? 1 : 0) : e->load(h, iq, o->flip ? 1 : 0)) != 0) { exitcode = 2; break; }

    int firstbin, lastbin;
    if (Search_width != 0 && cn0 > o->cn0_threshold) {       /* locked: search near the last carrier */
      if (Carrier_search_freq - Search_width <= -Samprate / 2) firstbin = 0;
      else { firstbin = (int)((Carrier_search_freq - Search_width) / Binsize); if (firstbin < 0) firstbin += N; }
      if (Carrier_search_freq + Search_width >= Samprate / 2) lastbin = N / 2 - 1;
      else { lastbin = (int)((Carrier_search_freq + Search_width) / Binsize); if (lastbin < 0) lastbin += N; }
    } else { firstbin = 0; lastbin = N; }
    if (firstbin > lastbin) { int t = firstbin; firstbin = lastbin; lastbin = t; }

    pmdemod_peak pk;
    if (e->fft_peak(h, firstbin, lastbin, &pk) != 0 || pk.peak < 0) { exitcode = 2; break; }
    double ap = (pk.next_re * pk.peak_re + pk.next_im * pk.peak_im) / pk.maxenergy;
    double dp = -ap / (1 - ap);
    double am = (pk.prev_re * pk.peak_re + pk.prev_im * pk.peak_im) / pk.maxenergy;
    double dm = am / (1 - am);
    double d = (dp + dm) / 2 + tau(dp * dp) - tau(dm * dm);
    double carrier_freq = Binsize * (pk.peak + d);
    if (carrier_freq > Samprate / 2) carrier_freq -= Samprate;

    double cstep = 2 * M_PI * carrier_freq / Samprate;
    pmdemod_mix mx;
    int out_dev = 0;
    int16_t *out16 = dst->acquire(dst->ctx, N, &out_dev);
    if (!out16 || (out_dev && !e->mix_dev)) { exitcode = 2; break; }
    if ((out_dev ? e->mix_dev(h, cstep, &mx, out16) : e->mix(h, cstep, &mx, out16)) != 0) { exitcode = 2; break; }
    cn0 = 10 * log10(Samprate * mx.amplitude * mx.amplitude / (2 * mx.diffsumsq));
    if (cn0 > o->cn0_threshold) Carrier_search_freq = carrier_freq;
    if (!o->quiet)
      fprintf(err, "%s: sample %'lld (%'.3lf sec, %s); carrier %'.1lf Hz; C/No = %'.2lf dB%s\n", o->argv0,
              total_samples, total_samples / Samprate, isee3_format_hms(total_samples / Samprate), carrier_freq,
              cn0, cn0 >= o->cn0_threshold ? " locked" : "");
    if (report && nb < report_cap) { report[nb].peak = pk.peak; report[nb].carrier_freq = carrier_freq; report[nb].cn0 = cn0; }
    nb++;
    if (dst->commit(dst->ctx, out16, N) != 0) { exitcode = 2; break; }
    total_samples += N;
  }
done:
  if (nreport) *nreport = nb;
  if (h) e->destroy(h);
  free(lo);
  return exitcode;
}
