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

/* pass 1 state (depends on the input only) and pass 2 state (consumes decoder output) */
typedef struct {
  unsigned char ring[RING], pair[2];
  int pos, sync_count, peak_in, peak_out;
} p1_state;
typedef struct {
  int startup, flips;
  unsigned long long reenc, symerrs, bits, symerrs_total, bits_out;
} p2_state;

/* pairing + phase tracking over one input block (vdecode.c:104-141): appends pairs to syms/hard from index np on,
 * flip events to fl; returns the new pair count */
static size_t pass1(const vdecode_opts *o, p1_state *st, const unsigned char *in, size_t got, int delay,
                    unsigned char *syms, unsigned char *hard, size_t np, flip_event *fl, size_t *nfl) {
  unsigned char *ring = st->ring;
  int pos = st->pos;
  for (size_t n = 0; n < got; n++) {
    unsigned char c = in[n];
    ring[pos] = c; st->pair[pos % 2] = c;
    if (!o->dontflip) {
      int sum = 0;
      for (int k = 0; k < 34; k++) sum += sync_sign[k] * ((int)ring[(RING + pos + k - 33) % RING] - 128);
      if ((pos % 2) == 0) { if (sum > st->peak_out) st->peak_out = sum; }
      else {
        if (sum > st->peak_in) st->peak_in = sum;
        if (++st->sync_count >= ISEE3_FRAMESYMBOLS) {
          st->sync_count = 0;
          if (st->peak_out > st->peak_in) {           /* other phase had the stronger sync: flip */
            fl[(*nfl)++].at_pair = np;
            if ((pos % 2) == 0) pos++; else pos--;
          }
          st->peak_in = st->peak_out = -1000000;
        }
      }
    }
    if ((pos % 2) == 1) {
      syms[2 * np] = st->pair[0]; syms[2 * np + 1] = st->pair[1];
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
  st->pos = pos;
  return np;
}

/* start-up suppression, output, re-encode statistics, status lines for pairs [j0, j1) (vdecode.c:151-184).
 * -1 when the engine handed back something that is not a bit once start-up is over, -2 when the output cannot be written
 * (a closed pipe, a memory stream that is full) */
static int pass2(const vdecode_opts *o, p2_state *st, const unsigned char *dec, const unsigned char *hard,
                  size_t j0, size_t j1, const flip_event *fl, size_t nfl, size_t *f, char *obuf, FILE *out, FILE *err) {
  size_t no = 0;
  for (size_t j = j0; j < j1; j++) {
    while (*f < nfl && fl[*f].at_pair == j) {
      st->flips++; (*f)++;
      if (!o->quiet) fprintf(err, "%s: flipping phase\n", o->argv0);
    }
    if (st->startup == 0) {
      if (dec[j] > 1u) { fprintf(err, "%s: decoder engine returned no bit for trellis step %zu\n", o->argv0, j); return -1; }
      unsigned bit = dec[j];
      obuf[no++] = bit ? '1' : '0';
      st->reenc = (st->reenc << 1) | bit;
    } else st->startup--;
    int s1 = ISEE3_G1FLIP ^ isee3_parity(st->reenc & ISEE3_POLY1);
    int s2 = ISEE3_G2FLIP ^ isee3_parity(st->reenc & ISEE3_POLY2);
    if (st->startup == 0) {
      unsigned add = (unsigned)(s1 ^ (hard[j] & 1)) + (unsigned)(s2 ^ ((hard[j] >> 1) & 1));
      st->symerrs += add; st->symerrs_total += add;
    }
    if (!o->quiet && o->status_interval != 0 && (++st->bits % (unsigned long long)o->status_interval) == 0) {
      fprintf(err, "%s: bits %'llu; symerrs %'llu/%'d %'.3lg%%\n", o->argv0, st->bits, st->symerrs,
              2 * o->status_interval, 100. * st->symerrs / (2. * o->status_interval));
      st->symerrs = 0;
    }
  }
  if (no && (fwrite(obuf, 1, no, out) != no || fflush(out) != 0)) {
    fprintf(err, "%s: short write on the output\n", o->argv0);
    return -2;
  }
  st->bits_out += no;
  return 0;
}

int vdecode_run(const vdecode_opts *o, const vdecode_engine *e, int fd_in, FILE *out, FILE *err,
                vdecode_result *res) {
  enum { INBLK = 1 << 18 };      /* read() hands over what the producer has ready, up to this much: long blocks can be shared between two decoders */
  int delay = o->decode_delay;
  if (delay < 24) {
    fprintf(err, "%s: decoder delay too small, using 200\n", o->argv0);
    delay = 200;
  } else if (delay > 1024) {
    fprintf(err, "%s: Warning; excessive decode delay; 1MB/bit needed\n", o->argv0);
  }
  p1_state s1; p2_state s2;
  memset(&s1, 0, sizeof s1); memset(&s2, 0, sizeof s2);
  for (int i = 0; i < RING; i += 2) { s1.ring[i] = ISEE3_G1FLIP ? 255 : 0; s1.ring[i + 1] = ISEE3_G2FLIP ? 255 : 0; }
  s1.pos = o->start_phase ? 1 : 0;            /* low bit = decoder symbol phase */
  s1.peak_in = s1.peak_out = -1000000;
  s2.startup = delay;
  int rc = -1;
  /* whole-input mode: the engine can decode one long stream faster than many short blocks (two decoders on
   * consecutive parts), so when the caller says the input is finite and latency does not matter, pass 1 runs over ALL
   * of it first.  Same pairs, same decoder output, same pass 2: stdout is byte-identical either way. */
  const int prog = o->whole_input && e->progressive_feed != NULL && e->progressive_end != NULL;
  const int whole = o->whole_input && (prog || e->stream_decode_whole != NULL);
  size_t cap = INBLK / 2 + 1, flcap = INBLK / ISEE3_FRAMESYMBOLS + 2;
  unsigned char *inbuf = malloc(INBLK), *syms = malloc(2 * cap), *hard = malloc(cap), *dec = malloc(cap);
  char *obuf = malloc(whole ? 1 : cap);
  flip_event *fl = malloc(sizeof(flip_event) * flcap);
  void *vd = e->create(delay + 1 + e->ring_extra);
  if (!inbuf || !syms || !hard || !dec || !obuf || !fl || !vd) goto done;
  e->init(vd, 0);

  if (!whole) {
    for (;;) {
      unsigned long want = e->read_limit ? e->read_limit(vd) : 0;
      if (want == 0 || want > INBLK) want = INBLK;
      if (want < 2) want = 2;
      ssize_t got = read(fd_in, inbuf, want);
      if (got <= 0) break;
      size_t nfl = 0, f = 0;
      size_t np = pass1(o, &s1, inbuf, (size_t)got, delay, syms, hard, 0, fl, &nfl);
      /* the engine: np trellis steps, one traceback each */
      if (np && e->stream_decode(vd, syms, (int)np, delay, dec) != 0) goto done;
      if ((rc = pass2(o, &s2, dec, hard, 0, np, fl, nfl, &f, obuf, out, err)) != 0) goto done;
      rc = -1;
      while (f < nfl) { s2.flips++; f++; if (!o->quiet) fprintf(err, "%s: flipping phase\n", o->argv0); }
    }
  } else {
    size_t np = 0, nfl = 0, f = 0;
    for (;;) {
      ssize_t got = read(fd_in, inbuf, INBLK);
      if (got <= 0) break;
      if (np + (size_t)got / 2 + 2 > cap) {
        cap = 2 * cap + (size_t)got;
        unsigned char *a = realloc(syms, 2 * cap), *b = realloc(hard, cap);
        if (a) syms = a;
        if (b) hard = b;
        if (!a || !b) goto done;
      }
      if (nfl + (size_t)got / ISEE3_FRAMESYMBOLS + 2 > flcap) {
        flcap = 2 * flcap + (size_t)got / ISEE3_FRAMESYMBOLS + 2;
        flip_event *c = realloc(fl, sizeof(flip_event) * flcap);
        if (!c) goto done;
        fl = c;
      }
      const size_t np0 = np;
      np = pass1(o, &s1, inbuf, (size_t)got, delay, syms, hard, np, fl, &nfl);
      if (prog && np > np0 && e->progressive_feed(vd, syms + 2 * np0, (int)(np - np0), delay) != 0) goto done;
    }
    free(dec); dec = malloc(np + 1);
    free(obuf); obuf = malloc(np + 1);
    if (!dec || !obuf) goto done;
    if (np && (prog ? e->progressive_end(vd, (long long)np, delay, dec) : e->stream_decode_whole(vd, syms, (long long)np, delay, dec)) != 0) goto done;
    if ((rc = pass2(o, &s2, dec, hard, 0, np, fl, nfl, &f, obuf, out, err)) != 0) goto done;
    rc = -1;
    while (f < nfl) { s2.flips++; f++; if (!o->quiet) fprintf(err, "%s: flipping phase\n", o->argv0); }
  }
  rc = 0;
done:
  if (vd) e->destroy(vd);
  free(inbuf); free(syms); free(hard); free(dec); free(obuf); free(fl);
  if (res) { res->bits_out = s2.bits_out; res->symerrs_total = s2.symerrs_total; res->flips = s2.flips; }
  return rc;
}
