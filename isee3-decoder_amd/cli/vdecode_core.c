/* vdecode_core.c -- see vdecode_core.h.  Plain C11, no GPU code in here. */
#define _GNU_SOURCE
#include "vdecode_core.h"
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "isee3_code.h"

#define RING 4096                        /* vdecode.c:20 SYMBOLBUFSIZE */

/* vdecode.c:27-30: expected sign of the last 34 encoded tail+sync symbols */
static const signed char sync_sign[34] = {
  -1, 1, 1, 1, 1, 1, 1, -1, 1, -1, 1, 1, 1, 1, -1, -1, 1,
   1, -1, -1, 1, 1, -1, 1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1 };

void vdecode_default_opts(vdecode_opts *o) {
  memset(o, 0, sizeof *o);
  o->decode_delay = 200; o->status_interval = 1024; o->argv0 = "vdecode";
}

int vdecode_parse_args(vdecode_opts *o, int argc, char **argv) {
  int c;
  vdecode_default_opts(o);
  if (argc > 0) o->argv0 = argv[0];
  optind = 1;
  while ((c = getopt(argc, argv, "d:pi:qF")) != -1) {
    switch (c) {
    case 'F': o->dontflip = 1; break;
    case 'q': o->quiet = 1; break;
    case 'p': o->start_phase = 1; break;
    case 'i': o->status_interval = atoi(optarg); break;
    case 'd': o->decode_delay = atoi(optarg); break;
    default: break;                       /* the reference ignores unknown options */
    }
  }
  return 0;
}

typedef struct { size_t at_pair; } flip_event;

int vdecode_run(const vdecode_opts *o, const vdecode_engine *e, int fd_in, FILE *out, FILE *err,
                vdecode_result *res) {
  enum { INBLK = 1 << 16 };
  int delay = o->decode_delay;
  if (delay < 24) {
    fprintf(err, "%s: decoder delay too small, using 200\n", o->argv0);
    delay = 200;
  } else if (delay > 1024) {
    fprintf(err, "%s: Warning; excessive decode delay; 1MB/bit needed\n", o->argv0);
  }
  int startup = delay;
  unsigned char ring[RING], pair[2] = { 0, 0 };
  for (int i = 0; i < RING; i += 2) { ring[i] = ISEE3_G1FLIP ? 255 : 0; ring[i + 1] = ISEE3_G2FLIP ? 255 : 0; }
  int pos = o->start_phase ? 1 : 0;           /* low bit = decoder symbol phase */
  int sync_count = 0, peak_in = -1000000, peak_out = -1000000;
  unsigned long long reenc = 0, symerrs = 0, bits = 0, symerrs_total = 0, bits_out = 0;
  int flips = 0, rc = -1;

  unsigned char *inbuf = malloc(INBLK), *syms = malloc(INBLK), *hard = malloc(INBLK / 2 + 1);
  unsigned char *dec = malloc(INBLK / 2 + 1);
  char *obuf = malloc(INBLK / 2 + 1);
  flip_event *fl = malloc(sizeof(flip_event) * (INBLK / ISEE3_FRAMESYMBOLS + 2));
  void *vd = e->create(delay + 1 + e->ring_extra);
  if (!inbuf || !syms || !hard || !dec || !obuf || !fl || !vd) goto done;
  e->init(vd, 0);

  for (;;) {
    ssize_t got = read(fd_in, inbuf, INBLK);
    if (got <= 0) break;
    /* pass 1: pairing + phase tracking (depends on the input only) */
    size_t np = 0, nfl = 0;
    for (ssize_t n = 0; n < got; n++) {
      unsigned char c = inbuf[n];
      ring[pos] = c; pair[pos % 2] = c;
      if (!o->dontflip) {
        int sum = 0;
        for (int k = 0; k < 34; k++) sum += sync_sign[k] * ((int)ring[(RING + pos + k - 33) % RING] - 128);
        if ((pos % 2) == 0) { if (sum > peak_out) peak_out = sum; }
        else {
          if (sum > peak_in) peak_in = sum;
          if (++sync_count >= ISEE3_FRAMESYMBOLS) {
            sync_count = 0;
            if (peak_out > peak_in) {           /* other phase had the stronger sync: flip */
              fl[nfl++].at_pair = np;
              if ((pos % 2) == 0) pos++; else pos--;
            }
            peak_in = peak_out = -1000000;
          }
        }
      }
      if ((pos % 2) == 1) {
        syms[2 * np] = pair[0]; syms[2 * np + 1] = pair[1];
        /* hard slices of the symbols the re-encoder will be compared with (vdecode.c:174-177) */
        /* same index expressions as the reference, made safe for delays beyond ~2000 where
           the reference's own expression goes negative */
        int back = 2 * (delay + ISEE3_K - 2);
        unsigned h1 = ring[(((pos - back - 1) % RING) + RING) % RING] > 128;
        unsigned h2 = ring[(((pos - back) % RING) + RING) % RING] > 128;
        hard[np] = (unsigned char)(h1 | (h2 << 1));
        np++;
      }
      pos = (pos + 1) % RING;
    }
    /* the engine: np trellis steps, one traceback each */
    if (np && e->stream_decode(vd, syms, (int)np, delay, dec) != 0) goto done;
    /* pass 2: start-up suppression, output, re-encode statistics, status lines */
    size_t no = 0, f = 0;
    for (size_t j = 0; j < np; j++) {
      while (f < nfl && fl[f].at_pair == j) {
        flips++; f++;
        if (!o->quiet) fprintf(err, "%s: flipping phase\n", o->argv0);
      }
      if (startup == 0) {
        unsigned bit = dec[j] & 1u;
        obuf[no++] = bit ? '1' : '0';
        reenc = (reenc << 1) | bit;
      } else startup--;
      int s1 = ISEE3_G1FLIP ^ isee3_parity(reenc & ISEE3_POLY1);
      int s2 = ISEE3_G2FLIP ^ isee3_parity(reenc & ISEE3_POLY2);
      if (startup == 0) {
        unsigned add = (unsigned)(s1 ^ (hard[j] & 1)) + (unsigned)(s2 ^ ((hard[j] >> 1) & 1));
        symerrs += add; symerrs_total += add;
      }
      if (!o->quiet && o->status_interval != 0 && (++bits % (unsigned long long)o->status_interval) == 0) {
        fprintf(err, "%s: bits %'llu; symerrs %'llu/%'d %'.3lg%%\n", o->argv0, bits, symerrs,
                2 * o->status_interval, 100. * symerrs / (2. * o->status_interval));
        symerrs = 0;
      }
    }
    while (f < nfl) { flips++; f++; if (!o->quiet) fprintf(err, "%s: flipping phase\n", o->argv0); }
    if (no) { fwrite(obuf, 1, no, out); fflush(out); }
    bits_out += no;
  }
  rc = 0;
done:
  if (vd) e->destroy(vd);
  free(inbuf); free(syms); free(hard); free(dec); free(obuf); free(fl);
  if (res) { res->bits_out = bits_out; res->symerrs_total = symerrs_total; res->flips = flips; }
  return rc;
}
