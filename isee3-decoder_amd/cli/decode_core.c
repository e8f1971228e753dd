/* decode_core.c -- see decode_core.h */
#define _GNU_SOURCE
#include "decode_core.h"
#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "timefmt.h"

/* expected sign of the 34 sync symbols (decode.c:37-40; the same vector as vdecode.c:27-30) */
static const signed char sync_sign[DECODE_SYNCBITS] = {
  -1, 1, 1, 1, 1, 1, 1, -1, 1, -1, 1, 1, 1, 1, -1, -1, 1,
  1, -1, -1, 1, 1, -1, 1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1
};

void decode_default_opts(decode_opts *o) {
  memset(o, 0, sizeof *o);
  o->symrate = 1024; o->fano_scale = 8; o->fano_maxcycles = 100; o->fano_delta = (int)(4 * o->fano_scale);
  o->viterbi_enabled = 1; o->fano_enabled = 1; o->argv0 = "decode";
}

void decode_parse_args(decode_opts *o, int argc, char **argv) {
  int c;
  decode_default_opts(o);
  if (argc > 0) o->argv0 = argv[0];
  optind = 1;
  while ((c = getopt(argc, argv, "nFVvr:s:m:d:p")) != EOF) {
    switch (c) {
    case 'p': o->persistent = 1; break;
    case 'n': o->no_bad_frames = 1; break;
    case 'F': o->viterbi_enabled = 0; break;
    case 'V': o->fano_enabled = 0; break;
    case 'v': o->verbose++; break;
    case 'r': o->symrate = atof(optarg); break;
    case 's': o->fano_scale = atof(optarg); break;
    case 'm': o->fano_maxcycles = (unsigned long)atol(optarg); break;
    case 'd': o->fano_delta = atoi(optarg); break;
    default:
      printf("usage: %s [-F] [-V] [-v] [-r symrate] [-s fano_scale] [-m fano_maxcycles] [-d fano_delta]\n", o->argv0);
    }
  }
}

/* ---- input: one growing buffer holding symbols [origin, origin + n); positions are absolute ---- */
typedef struct { unsigned char *p; long long origin; size_t n, cap; int fd, eof; } inbuf;
static int in_need(inbuf *b, long long upto) {   /* make symbols [origin, upto) available; 0 if the input ends first */
  while (b->origin + (long long)b->n < upto && !b->eof) {
    if (b->cap - b->n < (1u << 16)) {
      size_t ncap = b->cap ? 2 * b->cap : (1u << 20);
      unsigned char *q = realloc(b->p, ncap);
      if (!q) return 0;
      b->p = q; b->cap = ncap;
    }
    ssize_t got = read(b->fd, b->p + b->n, b->cap - b->n);
    if (got <= 0) b->eof = 1; else b->n += (size_t)got;
  }
  return b->origin + (long long)b->n >= upto;
}
static const unsigned char *in_at(const inbuf *b, long long pos) { return b->p + (pos - b->origin); }
static void in_drop(inbuf *b, long long before) {  /* forget symbols below `before` once that is worth a memmove */
  long long k = before - b->origin;
  if (k < (4 << 20)) return;
  memmove(b->p, b->p + k, b->n - (size_t)k);
  b->n -= (size_t)k; b->origin = before;
}

#define CACHE 16
typedef struct { long long pos[CACHE]; unsigned char data[CACHE][DECODE_FRAMEBITS / 8]; int n; } fcache;

int decode_run(const decode_opts *o, const decode_engine *e, int fd_in, FILE *out, FILE *err, decode_result *res) {
  decode_result r = {0, 0, 0, 0, 0};
  inbuf in = {NULL, 0, 0, 0, fd_in, 0};
  fcache fc; fc.n = 0;
  void *ctx = NULL;
  int rc = 0;
  (void)err;
  fprintf(out, "%s: Fano %s; Viterbi %s\n", o->argv0, o->fano_enabled ? "enabled" : "disabled",
          o->viterbi_enabled ? "enabled" : "disabled");
  if (o->no_bad_frames) fprintf(out, "%s: Not displaying bad frames\n", o->argv0);
  if (!o->fano_enabled && !o->viterbi_enabled) {
    fprintf(out, "%s: Specify only one of -F or -V\n", o->argv0);
    return 1;
  }
  if (o->fano_enabled) {
    fprintf(out, "%s: the Fano sequential decoder (CPU, fano.c) is not part of this build; run with -V\n", o->argv0);
    return 2;
  }
  if ((ctx = e->create()) == NULL) {
    fprintf(out, "%s: Fano decoding disabled but cannot alloc %d * 1 MB + 2 * 16 MB = %.1lf GB RAM for Viterbi decoder!\n"
                 "Try -F for Fano only or the default of Fano with Viterbi fallback to allocate memory only when the Viterbi decoder is actually used.\n",
            o->argv0, DECODE_FRAMEBITS, (DECODE_FRAMEBITS + 32) / 1024.);
    return 2;
  }
  long long base = 0;                 /* absolute symbol index of the reference's symbols[0] (total_symbols) */
  long long frames = 1;
  int lock = 0, sync_start;
  for (;;) {
    /* decode.c:152-161: one frame + one sync length must be buffered */
    if (!in_need(&in, base + DECODE_FRAMESYMBOLS + DECODE_SYNCBITS)) break;
    sync_start = 0;
    if (!lock) {                      /* decode.c:162-192 */
      int record = -1000000;
      const unsigned char *s = in_at(&in, base);
      for (int i = 0; i < DECODE_FRAMESYMBOLS; i++) {
        int sum = 0;
        for (int k = 0; k < DECODE_SYNCBITS; k++) sum += sync_sign[k] * ((int)s[i + k] - 128);
        if (sum > record) { record = sum; sync_start = i; }
      }
      if (!in_need(&in, base + sync_start + DECODE_FRAMESYMBOLS + DECODE_SYNCBITS)) break;
    }
    /* the frame the reference decodes now: symbols[sync_start + SYNCBITS ...] */
    const long long fpos = base + sync_start + DECODE_SYNCBITS;
    const unsigned char *data = NULL;
    for (int c = 0; c < fc.n; c++) if (fc.pos[c] == fpos) { data = fc.data[c]; break; }
    if (!data) {
      /* not speculated (first frame, or lock was lost and the sync moved, or the cache ran dry): decode it now,
       * together with the frames that follow at the frame spacing as far as the input already read reaches --
       * many while the stream is in lock, one more while it is searching */
      const unsigned char *fr[CACHE];
      const int want = lock ? CACHE : 2;
      int nb = 0;
      r.wasted += fc.n;               /* whatever is still cached was never asked for */
      for (int b = 0; b < want; b++) {
        long long p = fpos + (long long)b * DECODE_FRAMESYMBOLS;
        if (p + DECODE_FRAMESYMBOLS > in.origin + (long long)in.n) break;
        fc.pos[nb] = p; fr[nb] = in_at(&in, p); nb++;
      }
      if (nb == 0) { rc = -1; break; }          /* cannot happen: the current frame is buffered */
      if (e->decode_frames(ctx, fr, nb, &fc.data[0][0]) != 0) { rc = -1; break; }
      fc.n = nb; r.batches++; r.decoded += nb;
      data = fc.data[0];
    }
    /* decode.c:238-249: lock = the frame ends with the sync word */
    unsigned long long lastword = 0;
    for (int i = 123; i < 128; i++) lastword = (lastword << 8) | data[i];
    lock = lastword == DECODE_SYNCWORD;
    if (lock) r.good++;
    if (lock || !o->no_bad_frames) {            /* decode.c:251-268 */
      unsigned long long start_symbol = (unsigned long long)fpos;
      fprintf(out, "Frame %'llu at symbol %'llu (%s) with %s %s\n", (unsigned long long)frames, start_symbol,
              isee3_format_hms((double)start_symbol / o->symrate), "Viterbi", !lock ? "(bad)" : "");
      for (int i = 0; i < DECODE_FRAMEBITS / 8; i++) {
        fprintf(out, "%02x", data[i]);
        fputc((i % 16) == 15 ? '\n' : ' ', out);
      }
      fputc('\n', out);
      fflush(out);
    }
    frames++; r.frames++;
    /* consumed entries leave the cache */
    {
      int w = 0;
      for (int c = 0; c < fc.n; c++)
        if (fc.pos[c] > fpos) { if (w != c) { fc.pos[w] = fc.pos[c]; memcpy(fc.data[w], fc.data[c], sizeof fc.data[0]); } w++; }
      fc.n = w;
    }
    base += sync_start + DECODE_FRAMESYMBOLS;   /* decode.c:270-282 */
    in_drop(&in, base);
  }
  r.wasted += fc.n;
  e->destroy(ctx);
  free(in.p);
  if (res) *res = r;
  return rc;
}
