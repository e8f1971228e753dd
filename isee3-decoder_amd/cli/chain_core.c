/* chain_core.c -- pmdemod | symdemod | vdecode in ONE process (libisee3chain.so, bin/isee3chain): the three reference pipe stages
 * (pmdemod.c, symdemod.c, vdecode.c) run as threads of one program connected by pipe(2), so they
 * share one HIP context and one GPU instead of paying three process start-ups.  Each stage is
 * exactly the code of the stand-alone binary (the cli/ *_core.c files); stdin = int16 I,Q pairs, stdout = ASCII
 * '0'/'1'.  Options: the union of the stages' options that matter for a chain:
 *     -r Hz  sample rate (pmdemod -r, symdemod -r)        -b Hz  FFT bin size (pmdemod -b)
 *     -c Hz  symbol rate (symdemod -c)                    -d n   decode delay (vdecode -d)
 *     -W Hz / -S Hz / -f  pmdemod search width / start / flip          -v  keep the stages' status lines
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <locale.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "pmdemod_core.h"
#include "symdemod_core.h"
#include "vdecode_core.h"
#include "../../include/isee3_dsp_hip.h"
#include "../../include/viterbi224_hip.h"
#include "../../include/isee3_chain.h"

/* ---- engines (same bindings as the stand-alone mains), except that the handles of a finished call are kept for the
 * next one: creating and destroying the pmdemod / symdemod / Viterbi objects costs 3.5 + 3.5 + 8.5 ms of a 52 ms
 * capture.  isee3_chain_release() frees what is kept. ---- */
#define H_POOL 4
typedef struct { void *h; int n; } pooled;
static pthread_mutex_t g_hpool_mu = PTHREAD_MUTEX_INITIALIZER;
static pooled g_pm_pool[H_POOL], g_sy_pool[H_POOL];
static void *pool_take(pooled *pool, int n) {
  void *h = NULL;
  pthread_mutex_lock(&g_hpool_mu);
  for (int i = 0; i < H_POOL; i++) if (pool[i].h && pool[i].n == n) { h = pool[i].h; pool[i].h = NULL; break; }
  pthread_mutex_unlock(&g_hpool_mu);
  return h;
}
static int pool_give(pooled *pool, void *h, int n) {
  int kept = 0;
  pthread_mutex_lock(&g_hpool_mu);
  for (int i = 0; i < H_POOL; i++) if (!pool[i].h) { pool[i].h = h; pool[i].n = n; kept = 1; break; }
  pthread_mutex_unlock(&g_hpool_mu);
  return kept;
}
typedef struct { void *h; int n; } dsp_ctx;       /* remembers the size the handle was made for */
static void *pm_create(int n) {
  dsp_ctx *c = malloc(sizeof *c);
  if (!c) return NULL;
  c->n = n; c->h = pool_take(g_pm_pool, n);
  if (c->h) pmd_set_dechirp(c->h, NULL);            /* a kept handle may carry the last call's de-chirp table */
  else c->h = pmd_create(n);
  if (!c->h) { free(c); return NULL; }
  return c;
}
#define PMH(x) (((dsp_ctx *)(x))->h)
static int pm_dechirp(void *h, const double *t) { return pmd_set_dechirp(PMH(h), t); }
static int pm_load(void *h, const int16_t *iq, int flip) { return pmd_load(PMH(h), iq, 0, flip); }
static int pm_peak(void *h, int a, int b, pmdemod_peak *o) {
  pmd_peak p;
  if (pmd_fft_peak(PMH(h), a, b, &p) != 0) return -1;
  o->peak = p.peak; o->maxenergy = p.maxenergy; o->peak_re = p.peak_re; o->peak_im = p.peak_im;
  o->next_re = p.next_re; o->next_im = p.next_im; o->prev_re = p.prev_re; o->prev_im = p.prev_im;
  return 0;
}
static int pm_mix(void *h, double cstep, pmdemod_mix *r, int16_t *out16) {
  pmd_mix m;
  if (pmd_mix_quantise(PMH(h), cstep, &m, out16, NULL, 0) != 0) return -1;
  r->dc_re = m.dc_re; r->dc_im = m.dc_im; r->amplitude = m.amplitude; r->diffsumsq = m.diffsumsq;
  return 0;
}
static void pm_destroy(void *p) {
  dsp_ctx *c = p;
  if (!c) return;
  if (!pool_give(g_pm_pool, c->h, c->n)) pmd_destroy(c->h);
  free(c);
}

static void *sy_create(int n) {
  dsp_ctx *c = malloc(sizeof *c);
  if (!c) return NULL;
  c->n = n; c->h = pool_take(g_sy_pool, n);
  if (!c->h) c->h = symd_create(n);
  if (!c->h) { free(c); return NULL; }
  return c;
}
static int sy_load(void *h, const int16_t *s, int n) { return symd_load(PMH(h), s, n, 0); }
static int sy_ts(void *h, int lo, const int *sw, int sc, int ns, int noff, double *en) { return symd_timesearch(PMH(h), lo, sw, sc, ns, noff, en); }
static int sy_demod(void *h, const int *e, int sc, int ns, double g, uint8_t *o, double *es) { return symd_demod(PMH(h), e, sc, ns, g, o, 0, es); }
static void sy_destroy(void *p) {
  dsp_ctx *c = p;
  if (!c) return;
  if (!pool_give(g_sy_pool, c->h, c->n)) symd_destroy(c->h);
  free(c);
}

/* stream chunk of the decoders (bits): read from the environment ONCE, before any chain thread exists */
static int g_chunk = 1020;
static pthread_once_t g_chunk_once = PTHREAD_ONCE_INIT;
static void chunk_from_env(void) {
  const char *e = getenv("V224HIP_CHUNK");
  int c = e ? atoi(e) : 1020;
  g_chunk = c < 8 ? 1020 : c;
}
/* vdecode engine: one decoder for block-wise streaming; for one long stream (whole-input mode) a second decoder joins
 * and the stream is decoded in two halves at once, verified at the seam (v224hip_stream_decode_split) */
typedef struct { void *d[2]; int len; } vd_ctx;
/* Viterbi decoders (2.2 GiB decision ring, placement probe) are kept between calls too: a few, so that concurrent
 * chains each find one. */
#define VD_POOL 4
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;
static vd_ctx *g_pool[VD_POOL];
static void vd_free(vd_ctx *c) {
  if (!c) return;
  for (int i = 0; i < 2; i++) if (c->d[i]) delete_viterbi224(c->d[i]);
  free(c);
}
static void vd_destroy(void *p) {
  vd_ctx *c = p;
  if (!c) return;
  pthread_mutex_lock(&g_pool_mu);
  for (int i = 0; i < VD_POOL; i++) if (!g_pool[i]) { g_pool[i] = c; c = NULL; break; }
  pthread_mutex_unlock(&g_pool_mu);
  vd_free(c);                                  /* pool full */
}
void isee3_chain_release(void) {
  pthread_mutex_lock(&g_pool_mu);
  for (int i = 0; i < VD_POOL; i++) { vd_free(g_pool[i]); g_pool[i] = NULL; }
  pthread_mutex_unlock(&g_pool_mu);
  pthread_mutex_lock(&g_hpool_mu);
  for (int i = 0; i < H_POOL; i++) {
    if (g_pm_pool[i].h) { pmd_destroy(g_pm_pool[i].h); g_pm_pool[i].h = NULL; }
    if (g_sy_pool[i].h) { symd_destroy(g_sy_pool[i].h); g_sy_pool[i].h = NULL; }
  }
  pthread_mutex_unlock(&g_hpool_mu);
}
static void *vd_create(int len) {
  vd_ctx *c = NULL;
  pthread_mutex_lock(&g_pool_mu);
  for (int i = 0; i < VD_POOL; i++)
    if (g_pool[i] && g_pool[i]->len == len && v224hip_stream_chunk(g_pool[i]->d[0]) == g_chunk) { c = g_pool[i]; g_pool[i] = NULL; break; }
  pthread_mutex_unlock(&g_pool_mu);
  if (c) return c;                             /* vdecode_run calls init() next: a used decoder is as good as new */
  c = calloc(1, sizeof *c);
  if (!c) return NULL;
  c->len = len;
  c->d[0] = create_viterbi224(len);
  if (!c->d[0]) { free(c); return NULL; }
  v224hip_set_option(c->d[0], "chunk", g_chunk);
  return c;
}
static int vd_init(void *h, int s) { return init_viterbi224(((vd_ctx *)h)->d[0], s); }
static int vd_stream(void *h, const unsigned char *s, int n, int d, unsigned char *o) {
  return v224hip_stream_decode(((vd_ctx *)h)->d[0], s, n, d, o);
}
#define VD_SPLIT_WARM (4 * 1020)
static int vd_whole(void *h, const unsigned char *s, long long n, int d, unsigned char *o) {
  vd_ctx *c = h;
  if (n > 0x7fffffff / 2) return -1;
  if (n < 6 * VD_SPLIT_WARM) return v224hip_stream_decode(c->d[0], s, (int)n, d, o);      /* too short to gain */
  if (!c->d[1]) {
    c->d[1] = create_viterbi224(c->len);
    if (!c->d[1]) return v224hip_stream_decode(c->d[0], s, (int)n, d, o);
    v224hip_set_option(c->d[1], "chunk", g_chunk);
  }
  unsigned char *ds = v224hip_dev_alloc(2 * (size_t)n), *dout = v224hip_dev_alloc((size_t)n);
  int rc = -1, redone = 0;
  if (ds && dout && v224hip_h2d(ds, s, 2 * (size_t)n) == 0 &&
      v224hip_stream_decode_split(c->d, 2, ds, (int)n, d, dout, VD_SPLIT_WARM, &redone) == 0 &&
      v224hip_d2h(o, dout, (size_t)n) == 0) rc = 0;
  if (getenv("V224HIP_VERBOSE")) fprintf(stderr, "isee3chain/vdecode: %lld bits on two decoders, %d part(s) decoded again\n", n, redone);
  v224hip_dev_free(ds); v224hip_dev_free(dout);
  return rc;
}

/* ---- in-memory channel pmdemod -> symdemod: a byte ring with read(2) semantics on one side and a stdio stream
 * (fopencookie) on the other; a pipe costs two kernel copies and a syscall per 64 KiB of a 2 B/sample stream ---- */
typedef struct {
  pthread_mutex_t mu; pthread_cond_t can_read, can_write;
  unsigned char *buf; size_t cap, head, count; int closed, reader_gone;
} chan;
static chan *chan_new(size_t cap) {
  chan *c = calloc(1, sizeof *c);
  if (!c) return NULL;
  c->buf = malloc(cap); c->cap = cap;
  if (!c->buf) { free(c); return NULL; }
  pthread_mutex_init(&c->mu, NULL); pthread_cond_init(&c->can_read, NULL); pthread_cond_init(&c->can_write, NULL);
  return c;
}
static void chan_free(chan *c) { if (c) { free(c->buf); free(c); } }
static ssize_t chan_write(void *p, const char *b, size_t n) {
  chan *c = p; size_t done = 0;
  pthread_mutex_lock(&c->mu);
  while (done < n) {
    while (c->count == c->cap && !c->reader_gone) pthread_cond_wait(&c->can_write, &c->mu);
    if (c->reader_gone) { pthread_mutex_unlock(&c->mu); return 0; }      /* like EPIPE: the stream goes into error */
    size_t tail = (c->head + c->count) % c->cap, room = c->cap - c->count;
    size_t k = n - done < room ? n - done : room;
    if (k > c->cap - tail) k = c->cap - tail;
    memcpy(c->buf + tail, b + done, k);
    c->count += k; done += k;
    pthread_cond_signal(&c->can_read);
  }
  pthread_mutex_unlock(&c->mu);
  return (ssize_t)n;
}
static int chan_close(void *p) {
  chan *c = p;
  pthread_mutex_lock(&c->mu); c->closed = 1; pthread_cond_broadcast(&c->can_read); pthread_mutex_unlock(&c->mu);
  return 0;
}
static long chan_read(void *p, void *b, unsigned long n) {       /* whole int16 samples only */
  chan *c = p;
  if (n < 2) return 0;
  pthread_mutex_lock(&c->mu);
  while (c->count < 2 && !c->closed) pthread_cond_wait(&c->can_read, &c->mu);
  size_t k = c->count & ~(size_t)1;
  if (k > (n & ~1ul)) k = n & ~1ul;
  if (k > c->cap - c->head) k = (c->cap - c->head) & ~(size_t)1;
  if (k == 0 && c->count >= 2) {                                   /* a sample straddles the wrap: hand it over alone */
    unsigned char *o = b; o[0] = c->buf[c->head]; o[1] = c->buf[(c->head + 1) % c->cap]; k = 2;
  } else memcpy(b, c->buf + c->head, k);
  c->head = (c->head + k) % c->cap; c->count -= k;
  pthread_cond_signal(&c->can_write);
  pthread_mutex_unlock(&c->mu);
  return (long)k;
}

typedef struct { pmdemod_opts o; FILE *in, *out; int rc; } pm_arg;
typedef struct { symdemod_opts o; chan *in; FILE *out; int rc; } sy_arg;
typedef struct { vdecode_opts o; int fd_in; FILE *out; int rc; } vd_arg;

static void *pm_thread(void *p) {
  pm_arg *a = p;
  pmdemod_engine e = { pm_create, pm_dechirp, pm_load, pm_peak, pm_mix, pm_destroy };
  a->rc = pmdemod_run(&a->o, &e, a->in, a->out, stderr, NULL, 0, NULL);
  fclose(a->out);
  return NULL;
}
static void *sy_thread(void *p) {
  sy_arg *a = p;
  symdemod_engine e = { sy_create, sy_load, sy_ts, sy_demod, sy_destroy };
  a->rc = symdemod_run_rd(&a->o, &e, chan_read, a->in, a->out, stderr);
  fclose(a->out);
  pthread_mutex_lock(&a->in->mu); a->in->reader_gone = 1; pthread_cond_broadcast(&a->in->can_write); pthread_mutex_unlock(&a->in->mu);
  return NULL;
}
static void *vd_thread(void *p) {
  vd_arg *a = p;
  vdecode_result r;
  vdecode_engine e = { vd_create, vd_init, vd_stream, vd_destroy, 2 * g_chunk, vd_whole };
  a->rc = vdecode_run(&a->o, &e, a->fd_in, a->out, stderr, &r);
  fflush(a->out);
  close(a->fd_in);
  return NULL;
}


static __thread char g_chain_err[256];
const char *isee3_chain_last_error(void) { return g_chain_err; }

void isee3_chain_default_opts(isee3_chain_opts *o) {
  memset(o, 0, sizeof *o);
  o->samprate = 250000; o->binsize = 4; o->decode_delay = 200;
}

static int chain_run(const isee3_chain_opts *co, FILE *in, FILE *out, int finite_input) {
  pm_arg pa; sy_arg sa; vd_arg va;
  int p2[2];
  chan *c1 = chan_new((size_t)64 << 20);
  cookie_io_functions_t cio = { NULL, chan_write, NULL, chan_close };
  pmdemod_default_opts(&pa.o); symdemod_default_opts(&sa.o); vdecode_default_opts(&va.o);
  pa.o.argv0 = "isee3chain/pmdemod"; sa.o.argv0 = "isee3chain/symdemod"; va.o.argv0 = "isee3chain/vdecode";
  pa.o.samprate = co->samprate; sa.o.samprate = (int)co->samprate;
  pa.o.binsize = co->binsize; pa.o.search_freq = co->search_freq; pa.o.search_width = co->search_width; pa.o.flip = co->flip;
  if (co->symrate) symdemod_set_symrate(&sa.o, co->symrate);   /* symdemod -c semantics; no getopt in a library that runs concurrent chains */
  va.o.decode_delay = co->decode_delay;
  /* a capture in memory is finite and nobody waits for early bits: let vdecode see the whole symbol stream at once.
   * ISEE3_CHAIN_WHOLE=0 / 1 overrides. */
  (void)finite_input;   /* measured: at 30 k bits per capture the second decoder and the lost overlap with pmdemod/symdemod cost more than the split saves (72 vs 57 ms) */
  va.o.whole_input = getenv("ISEE3_CHAIN_WHOLE") ? atoi(getenv("ISEE3_CHAIN_WHOLE")) : 0;
  pa.o.quiet = sa.o.quiet = va.o.quiet = !co->verbose;
  pthread_once(&g_chunk_once, chunk_from_env);
  if (!c1 || pipe(p2)) { snprintf(g_chain_err, sizeof g_chain_err, "pipe() / channel allocation failed"); chan_free(c1); return 2; }
#ifdef F_SETPIPE_SZ
  fcntl(p2[1], F_SETPIPE_SZ, 1 << 20);
#endif
  pa.in = in; pa.out = fopencookie(c1, "w", cio);
  sa.in = c1; sa.out = fdopen(p2[1], "w");
  va.fd_in = p2[0]; va.out = out;
  pthread_t t1, t2, t3;
  pthread_create(&t1, NULL, pm_thread, &pa);
  pthread_create(&t2, NULL, sy_thread, &sa);
  pthread_create(&t3, NULL, vd_thread, &va);
  pthread_join(t1, NULL); pthread_join(t2, NULL); pthread_join(t3, NULL);
  chan_free(c1);
  if (pa.rc || sa.rc || va.rc) {
    snprintf(g_chain_err, sizeof g_chain_err, "stage failed (pmdemod %d, symdemod %d, vdecode %d): %.80s / %.80s", pa.rc, sa.rc,
             va.rc, isee3dsp_last_error(), v224hip_last_error());
    return 2;
  }
  return 0;
}

int isee3_chain_run_fd(const isee3_chain_opts *o, int fd_in, int fd_out) {
  FILE *in = fdopen(dup(fd_in), "r"), *out = fdopen(dup(fd_out), "w");
  if (!in || !out) { snprintf(g_chain_err, sizeof g_chain_err, "fdopen failed"); return 2; }
  int rc = chain_run(o, in, out, 0);
  fclose(in); fclose(out);
  return rc;
}

int isee3_chain_run_mem(const isee3_chain_opts *o, const int16_t *iq, size_t nsamples, char *out, size_t cap, size_t *nout) {
  FILE *in = fmemopen((void *)iq, nsamples * 4, "r");       /* the capture is read where it lies */
  FILE *mo = fmemopen(out, cap, "w");
  if (!in || !mo) { snprintf(g_chain_err, sizeof g_chain_err, "fmemopen failed"); return 2; }
  setvbuf(in, NULL, _IOFBF, 1 << 20);                /* (unbuffered, glibc reads a memory stream byte by byte) */
  setvbuf(mo, NULL, _IONBF, 0);                       /* write straight into the caller's buffer */
  int rc = chain_run(o, in, mo, 1);
  long pos = ftell(mo);
  fclose(mo); fclose(in);
  if (nout) *nout = pos > 0 ? (size_t)pos : 0;
  return rc;
}
