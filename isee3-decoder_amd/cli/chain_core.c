/* chain_core.c -- pmdemod | symdemod | vdecode in ONE process (libisee3chain.so, bin/isee3chain): the three reference pipe stages
 * (pmdemod.c, symdemod.c, vdecode.c) run as threads of one program, so they share one HIP context and one GPU
 * instead of paying three process start-ups.  Each stage is exactly the code of the stand-alone binary (the
 * cli/ *_core.c files).  The sample streams stay in HBM: pmdemod writes each baseband block into a device slot of a
 * small ring, symdemod copies it (device to device) into its device-resident two-window buffer; a capture that is
 * already in device memory is read in place.  Only the soft symbols (1 kB per second of signal) and the decoded bits
 * cross to the host.  stdin = int16 I,Q pairs, stdout = ASCII '0'/'1'.
 * Options: the union of the stages' options that matter for a chain:
 *     -r Hz  sample rate (pmdemod -r, symdemod -r)        -b Hz  FFT bin size (pmdemod -b)
 *     -c Hz  symbol rate (symdemod -c)                    -d n   decode delay (vdecode -d)
 *     -W Hz / -S Hz / -f  pmdemod search width / start / flip          -v  keep the stages' status lines
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <locale.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
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
  isee3dsp_share_stream(1);                          /* one front-end stream for pmdemod and symdemod (isee3_dsp_hip.h: why) */
  c->n = n; c->h = pool_take(g_pm_pool, n);
  if (c->h) pmd_set_dechirp(c->h, NULL);            /* a kept handle may carry the last call's de-chirp table */
  else c->h = pmd_create(n);
  if (!c->h) { free(c); return NULL; }
  return c;
}
#define PMH(x) (((dsp_ctx *)(x))->h)
static int pm_dechirp(void *h, const double *t) { return pmd_set_dechirp(PMH(h), t); }
/* time spent inside engine calls, per stage (they include the waits for the GPU): isee3_chain_last_stage_ms() */
static __thread double t_stage_ms;
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
#define TIMED(expr) do { double t0_ = now_ms(); int r_ = (expr); t_stage_ms += now_ms() - t0_; return r_; } while (0)
static int pm_load(void *h, const int16_t *iq, int flip) { TIMED(pmd_load(PMH(h), iq, 0, flip)); }
static int pm_load_dev(void *h, const int16_t *iq, int flip) { TIMED(pmd_load(PMH(h), iq, 1, flip)); }
static int pm_peak(void *h, int a, int b, pmdemod_peak *o) {
  pmd_peak p;
  double t0 = now_ms();
  int rc = pmd_fft_peak(PMH(h), a, b, &p);
  t_stage_ms += now_ms() - t0;
  if (rc != 0) return -1;
  o->peak = p.peak; o->maxenergy = p.maxenergy; o->peak_re = p.peak_re; o->peak_im = p.peak_im;
  o->next_re = p.next_re; o->next_im = p.next_im; o->prev_re = p.prev_re; o->prev_im = p.prev_im;
  return 0;
}
static int pm_mix_any(void *h, double cstep, pmdemod_mix *r, int16_t *out16, int out_is_dev) {
  pmd_mix m;
  double t0 = now_ms();
  int rc = pmd_mix_quantise(PMH(h), cstep, &m, out16, NULL, out_is_dev);
  t_stage_ms += now_ms() - t0;
  if (rc != 0) return -1;
  r->dc_re = m.dc_re; r->dc_im = m.dc_im; r->amplitude = m.amplitude; r->diffsumsq = m.diffsumsq;
  return 0;
}
static int pm_mix(void *h, double cstep, pmdemod_mix *r, int16_t *out16) { return pm_mix_any(h, cstep, r, out16, 0); }
static int pm_mix_dev(void *h, double cstep, pmdemod_mix *r, int16_t *d_out16) { return pm_mix_any(h, cstep, r, d_out16, 1); }
/* the asynchronous halves (pmdemod_core.h): block k+1's transform is in the front-end stream before block k's peak is awaited */
static int pm_peak_begin(void *h, int a, int b) { TIMED(pmd_fft_peak_begin(PMH(h), a, b)); }
static int pm_peak_end(void *h, pmdemod_peak *o) {
  pmd_peak p;
  double t0 = now_ms();
  int rc = pmd_fft_peak_end(PMH(h), &p);
  t_stage_ms += now_ms() - t0;
  if (rc != 0) return -1;
  o->peak = p.peak; o->maxenergy = p.maxenergy; o->peak_re = p.peak_re; o->peak_im = p.peak_im;
  o->next_re = p.next_re; o->next_im = p.next_im; o->prev_re = p.prev_re; o->prev_im = p.prev_im;
  return 0;
}
static int pm_mix_begin(void *h, double cstep, int16_t *out16, int out_is_dev) { TIMED(pmd_mix_begin(PMH(h), cstep, out16, NULL, out_is_dev)); }
static int pm_mix_end(void *h, pmdemod_mix *r) {
  pmd_mix m;
  double t0 = now_ms();
  int rc = pmd_mix_end(PMH(h), &m);
  t_stage_ms += now_ms() - t0;
  if (rc != 0) return -1;
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
  /* symdemod's kernels go to the NULL stream, pmdemod's to the front-end stream: with the two decoders that makes four busy
   * streams on hardware queues 0 (the null stream's, always there) .. 3 = one per compute pipe, whatever the process created
   * before (isee3_dsp_hip.h, isee3dsp_share_stream).  ISEE3_CHAIN_SY_NULL=0: both stages on the one front-end stream. */
  { const char *e = getenv("ISEE3_CHAIN_SY_NULL"); isee3dsp_share_stream(e && atoi(e) == 0 ? 1 : 2); }
  c->n = n; c->h = pool_take(g_sy_pool, n);
  if (c->h) symd_store_reset(c->h);                 /* a kept handle: its window buffer starts out zero again */
  else c->h = symd_create(n);
  if (!c->h) { free(c); return NULL; }
  return c;
}
static int sy_load(void *h, const int16_t *s, int n) { TIMED(symd_load(PMH(h), s, n, 0)); }
static int sy_ts(void *h, int lo, const int *sw, int sc, int ns, int noff, double *en) { TIMED(symd_timesearch(PMH(h), lo, sw, sc, ns, noff, en)); }
static int sy_demod(void *h, const int *e, int sc, int ns, double g, uint8_t *o, double *es) { TIMED(symd_demod(PMH(h), e, sc, ns, g, o, 0, es)); }
static int sy_slide(void *h, int slide, int n) { TIMED(symd_store_slide(PMH(h), slide, n)); }
static int sy_put(void *h, int at, const int16_t *src, int n, int dev) { TIMED(symd_store_put(PMH(h), at, src, n, dev)); }
static int sy_scan(void *h, int n) { TIMED(symd_store_scan(PMH(h), n)); }
static int sy_window(void *h, int fs, const int *sw, int sc, int ns, int fo, int noff, const int *ed, int lo, int nspec,
                     uint8_t *out, int *ph, double *me) { TIMED(symd_window(PMH(h), fs, sw, sc, ns, fo, noff, ed, lo, nspec, out, ph, me)); }
static void sy_destroy(void *p) {
  dsp_ctx *c = p;
  if (!c) return;
  if (!pool_give(g_sy_pool, c->h, c->n)) symd_destroy(c->h);
  free(c);
}

/* stream chunk of the decoders (bits): read from the environment ONCE, before any chain thread exists */
#define CHAIN_CHUNK 1020            /* 68 passes of 15 */
static int g_chunk = CHAIN_CHUNK;
static pthread_once_t g_chunk_once = PTHREAD_ONCE_INIT;
static void chunk_from_env(void) {
  const char *e = getenv("V224HIP_CHUNK");
  int c = e ? atoi(e) : CHAIN_CHUNK;
  g_chunk = c < 8 ? CHAIN_CHUNK : c;
}
/* vdecode engine: one decoder for block-wise streaming; for one long stream (whole-input mode) a second decoder joins
 * and the stream is decoded in two halves at once, verified at the seam (v224hip_stream_decode_split) */
typedef struct { void *d[2]; int len, holder; volatile int *front_done; long long expected; void *prog; } vd_ctx;   /* holder: the decoder that carries the stream's state */
static __thread volatile int *t_front_done;      /* set by the chain for its vdecode thread: 1 once symdemod has finished */
static __thread long long t_expected_bits;       /* ditto: decoded bits the capture should give (places the cut of the progressive decode) */
/* Viterbi decoders (2.2 GiB decision ring, placement probe) are kept between calls too: a few, so that concurrent
 * chains each find one. */
#define VD_POOL 4
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;
static vd_ctx *g_pool[VD_POOL];
static void vd_free(vd_ctx *c) {
  if (!c) return;
  if (c->prog) { v224hip_progressive_abort(c->prog); c->prog = NULL; }
  for (int i = 0; i < 2; i++) if (c->d[i]) delete_viterbi224(c->d[i]);
  free(c);
}
static void vd_destroy(void *p) {
  vd_ctx *c = p;
  if (!c) return;
  if (c->prog) { v224hip_progressive_abort(c->prog); c->prog = NULL; }     /* a run that failed half-way */
  pthread_mutex_lock(&g_pool_mu);
  for (int i = 0; i < VD_POOL; i++) if (!g_pool[i]) { g_pool[i] = c; c = NULL; break; }
  pthread_mutex_unlock(&g_pool_mu);
  vd_free(c);                                  /* pool full */
}
static void slot_pool_release(void);
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
  slot_pool_release();
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
static int vd_init(void *h, int s) {
  vd_ctx *c = h;
  c->holder = 0; c->front_done = t_front_done; c->expected = t_expected_bits;
  return init_viterbi224(c->d[0], s);
}
#define VD_SPLIT_WARM (4 * 1020)
/* warm-up of a decoder that joins inside a block: one chunk of seam window (>= the decode delay) + 1 020 bits for its fresh
 * start to be forgotten.  Measured (profiles/r02h_metric_convergence.txt): all 2^23 path metrics agree with those of a
 * decoder that followed the stream from its beginning after <= 255 steps at Eb/N0 3 dB, <= 390 at 1.5 dB, <= 765 at 0 dB;
 * on pure noise only after ~3 000 -- there the seam check fails and the block is finished by the first decoder alone. */
#define VD_SHARE_WARM (3 * CHAIN_CHUNK)      /* 1 020 + 2 040 */
/* one block of the stream.  The stages in front run ahead (the Viterbi is the slowest), so blocks get long, and a long
 * block can be shared between two decoders (v224hip_stream_decode_shared: the second one starts fresh inside the block,
 * verified at the seam): same bits, two launch chains on the GPU instead of one.  But two 1024-thread workgroups per CU
 * are ALL the wave slots a CU has: while pmdemod / symdemod still have kernels to run, a second decoder starves them
 * (measured: front-end 13 -> 50 ms, chain 40 -> 54 ms).  So a block is shared only once the front end has finished --
 * typically the last, longest block.  ISEE3_CHAIN_SHARE=0 never shares, =2 shares whenever a block is long enough. */
static int vd_stream_any(vd_ctx *c, const unsigned char *s, int n, int d, unsigned char *o) {
  const char *e = getenv("ISEE3_CHAIN_SHARE");
  const int mode = e ? atoi(e) : 1;
  const int may = mode == 2 || (mode == 1 && c->front_done && *c->front_done);
  if (may && 5 * n >= 11 * VD_SHARE_WARM && !c->d[1]) {
    c->d[1] = create_viterbi224(c->len);
    if (c->d[1]) v224hip_set_option(c->d[1], "chunk", g_chunk);
  }
  if (!may || !c->d[1]) return v224hip_stream_decode(c->d[c->holder], s, n, d, o);
  return v224hip_stream_decode_shared(c->d, 2, &c->holder, s, n, d, o, VD_SHARE_WARM);
}
static int vd_stream(void *h, const unsigned char *s, int n, int d, unsigned char *o) { TIMED(vd_stream_any(h, s, n, d, o)); }
/* while the front end is still producing, take the symbols in blocks of at most eight windows: the moment it finishes
 * is then noticed within a few ms, and everything that is left goes to two decoders as ONE long block */
static unsigned long vd_read_limit(void *h) {
  vd_ctx *c = h;
  const char *e = getenv("ISEE3_CHAIN_SHARE");
  if (e && atoi(e) == 2) return 0ul;
  return (c->front_done && !*c->front_done) ? 8192ul : 0ul;
}
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

/* Progressive mode (the default when the capture's length is known): the whole symbol stream goes to
 * v224hip_progressive_* as it arrives.  Decoder 0 works from the first symbol on; decoder 1 joins at the planned cut as soon
 * as the symbols reach it -- ONE warm-up per capture instead of one per shared block, and both decoders busy for the second
 * half of the front end's run as well (measured, scratch/starve.py: a second busy decoder slows pmdemod / symdemod by 25 %,
 * not more, once every busy stream has its own compute pipe). */
#define VD_PROG_WARM (2 * g_chunk)           /* seam window (one chunk >= the decode delay of 200) + one chunk of forgetting */
static int vd_prog_feed_any(vd_ctx *c, const unsigned char *s, int n, int d) {
  if (!c->prog) {
    if (c->expected >= 4 * VD_SHARE_WARM && !c->d[1]) {
      c->d[1] = create_viterbi224(c->len);
      if (c->d[1]) v224hip_set_option(c->d[1], "chunk", g_chunk);
    }
    c->prog = v224hip_progressive_begin(c->d, c->d[1] ? 2 : 1, c->expected, d, VD_PROG_WARM);
    if (!c->prog) return -1;
  }
  return v224hip_progressive_feed(c->prog, s, n);
}
static int vd_prog_end_any(vd_ctx *c, long long n, int d, unsigned char *o) {
  long long got = 0;
  int redone = 0;
  void *p = c->prog;
  (void)d;
  c->prog = NULL;
  if (!p) return n == 0 ? 0 : -1;
  if (v224hip_progressive_end(p, o, n, &got, &redone) != 0 || got != n) return -1;
  if (getenv("V224HIP_VERBOSE")) fprintf(stderr, "isee3chain/vdecode: %lld bits, progressive on %d decoder(s), expected %lld, second part decoded again: %d\n",
                                         n, c->d[1] ? 2 : 1, c->expected, redone);
  return 0;
}
static int vd_prog_feed(void *h, const unsigned char *s, int n, int d) { TIMED(vd_prog_feed_any(h, s, n, d)); }
static int vd_prog_end(void *h, long long n, int d, unsigned char *o) { TIMED(vd_prog_end_any(h, n, d, o)); }

/* ---- block channel pmdemod -> symdemod: a ring of device slots, each one baseband block (N int16).  pmdemod acquires
 * a free slot, lets the engine write the block into it, commits it; symdemod takes views of committed slots in order
 * and copies them device-to-device into its window buffer.  No sample passes through host memory. ---- */
#define NSLOT 4
typedef struct {
  pthread_mutex_t mu; pthread_cond_t cv;
  int16_t *slot[NSLOT]; int N;
  int q[NSLOT], qhead, qcount;            /* committed slots, FIFO */
  int isfree[NSLOT];
  int closed, reader_gone;
  int cur, cur_pos;                       /* slot being consumed by the reader (-1: none) */
} blkchan;

/* slots of finished calls are kept (device allocations of up to 16 MiB each), keyed by block size */
#define SLOT_POOL 8
static pthread_mutex_t g_slot_mu = PTHREAD_MUTEX_INITIALIZER;
static struct { int16_t *p; int N; } g_slot_pool[SLOT_POOL];
static int16_t *slot_take(int N) {
  int16_t *p = NULL;
  pthread_mutex_lock(&g_slot_mu);
  for (int i = 0; i < SLOT_POOL; i++) if (g_slot_pool[i].p && g_slot_pool[i].N == N) { p = g_slot_pool[i].p; g_slot_pool[i].p = NULL; break; }
  pthread_mutex_unlock(&g_slot_mu);
  return p ? p : isee3dsp_dev_alloc(sizeof(int16_t) * (size_t)N);
}
static void slot_give(int16_t *p, int N) {
  if (!p) return;
  pthread_mutex_lock(&g_slot_mu);
  for (int i = 0; i < SLOT_POOL; i++) if (!g_slot_pool[i].p) { g_slot_pool[i].p = p; g_slot_pool[i].N = N; p = NULL; break; }
  pthread_mutex_unlock(&g_slot_mu);
  if (p) isee3dsp_dev_free(p);
}
static void slot_pool_release(void) {
  pthread_mutex_lock(&g_slot_mu);
  for (int i = 0; i < SLOT_POOL; i++) if (g_slot_pool[i].p) { isee3dsp_dev_free(g_slot_pool[i].p); g_slot_pool[i].p = NULL; }
  pthread_mutex_unlock(&g_slot_mu);
}

static void blk_init(blkchan *c) {
  memset(c, 0, sizeof *c);
  pthread_mutex_init(&c->mu, NULL); pthread_cond_init(&c->cv, NULL);
  c->cur = -1;
}
static void blk_free(blkchan *c) { for (int i = 0; i < NSLOT; i++) { slot_give(c->slot[i], c->N); c->slot[i] = NULL; } }
/* pmdemod side */
static int16_t *blk_acquire(void *p, int N, int *is_dev) {
  blkchan *c = p;
  int16_t *r = NULL;
  pthread_mutex_lock(&c->mu);
  if (c->N == 0) {                                   /* first block: now the size is known */
    c->N = N;
    for (int i = 0; i < NSLOT; i++) { c->slot[i] = slot_take(N); c->isfree[i] = c->slot[i] != NULL; }
  }
  for (;;) {
    int k = -1;
    for (int i = 0; i < NSLOT; i++) if (c->isfree[i]) { k = i; break; }
    if (k >= 0) { c->isfree[k] = 0; r = c->slot[k]; break; }
    if (!c->slot[0] || !c->slot[1]) break;           /* allocation failed */
    if (c->reader_gone) { r = c->slot[0]; break; }   /* nobody listens any more: any slot will do */
    pthread_cond_wait(&c->cv, &c->mu);
  }
  pthread_mutex_unlock(&c->mu);
  *is_dev = 1;
  return r;
}
static int blk_commit(void *p, int16_t *buf, int N) {
  blkchan *c = p;
  (void)N;
  pthread_mutex_lock(&c->mu);
  int k = -1;
  for (int i = 0; i < NSLOT; i++) if (c->slot[i] == buf) k = i;
  if (k >= 0) {
    if (c->reader_gone) c->isfree[k] = 1;
    else { c->q[(c->qhead + c->qcount) % NSLOT] = k; c->qcount++; }
  }
  pthread_cond_broadcast(&c->cv);
  pthread_mutex_unlock(&c->mu);
  return k >= 0 ? 0 : -1;
}
static void blk_close(blkchan *c) {
  pthread_mutex_lock(&c->mu); c->closed = 1; pthread_cond_broadcast(&c->cv); pthread_mutex_unlock(&c->mu);
}
/* symdemod side: a view of up to max samples; the previous view's slot is released first (its copy is done: the
 * engine's store_put returns after a device-to-device copy has finished) */
static long blk_next(void *p, const int16_t **blk, int *is_dev, long max) {
  blkchan *c = p;
  long n = 0;
  pthread_mutex_lock(&c->mu);
  if (c->cur >= 0 && c->cur_pos >= c->N) { c->isfree[c->cur] = 1; c->cur = -1; pthread_cond_broadcast(&c->cv); }
  while (c->cur < 0) {
    if (c->qcount > 0) { c->cur = c->q[c->qhead]; c->qhead = (c->qhead + 1) % NSLOT; c->qcount--; c->cur_pos = 0; break; }
    if (c->closed) break;
    pthread_cond_wait(&c->cv, &c->mu);
  }
  if (c->cur >= 0) {
    n = c->N - c->cur_pos < max ? c->N - c->cur_pos : max;
    *blk = c->slot[c->cur] + c->cur_pos; *is_dev = 1;
    c->cur_pos += (int)n;
  }
  pthread_mutex_unlock(&c->mu);
  return n;
}
static void blk_reader_gone(blkchan *c) {
  pthread_mutex_lock(&c->mu);
  c->reader_gone = 1;
  if (c->cur >= 0) { c->isfree[c->cur] = 1; c->cur = -1; }
  while (c->qcount > 0) { c->isfree[c->q[c->qhead]] = 1; c->qhead = (c->qhead + 1) % NSLOT; c->qcount--; }
  pthread_cond_broadcast(&c->cv);
  pthread_mutex_unlock(&c->mu);
}

/* ---- where pmdemod's blocks come from: a capture in memory (host or device: views, no copy) or a FILE ---- */
typedef struct { const int16_t *iq; size_t nsamples, pos; int is_dev; FILE *f; int16_t *buf; } iq_src;
static int iq_next(void *p, int N, const int16_t **blk, int *is_dev) {
  iq_src *s = p;
  if (s->f) {
    if (!s->buf && !(s->buf = malloc(sizeof(int16_t) * 2 * (size_t)N))) return -1;
    if (fread(s->buf, 4, (size_t)N, s->f) < (size_t)N) return 0;
    *blk = s->buf; *is_dev = 0;
    return 1;
  }
  if (s->pos + (size_t)N > s->nsamples) return 0;      /* a partial block is dropped (pmdemod.c:206-216) */
  *blk = s->iq + 2 * s->pos; *is_dev = s->is_dev;
  s->pos += (size_t)N;
  return 1;
}

typedef struct { pmdemod_opts o; iq_src src; blkchan *out; int rc; double ms; } pm_arg;
typedef struct { symdemod_opts o; blkchan *in; FILE *out; int rc; double ms; volatile int *done; double t0, t_done; } sy_arg;
typedef struct { vdecode_opts o; int fd_in; FILE *out; int rc; double ms; volatile int *front_done; long long expected_bits; int progressive; } vd_arg;

static void *pm_thread(void *p) {
  pm_arg *a = p;
  pmdemod_engine e = { pm_create, pm_dechirp, pm_load, pm_peak, pm_mix, pm_destroy, pm_load_dev, pm_mix_dev, NULL, NULL, NULL, NULL };
  /* Two pmdemod handles in turn, block k+1's transform enqueued before block k's peak is awaited (pmdemod_core.h): on for a
   * capture that lies in device memory, off for one in host memory (there the next block's staged H2D copy queues in front
   * of this block's spin-down: 7 % slower); ISEE3_CHAIN_PM_PIPELINE=0 / 1 overrides.  Measured with six launches per block
   * (profiles/r03ar_*): pmdemod's engine time 17.8 -> 14-15 ms at 10 MS/s, chain 14.51-14.65 -> 14.72-14.82 Gsamples/s;
   * 5.6 -> 2.3 ms at 250 kS/s, chain the same (its front end is paced by symdemod's windows).  (With eleven launches per
   * block and the double transform the chain had gained nothing at either rate: profiles/r03z_*.) */
  int pipeline = a->src.is_dev && !a->src.f;
  if (getenv("ISEE3_CHAIN_PM_PIPELINE")) pipeline = atoi(getenv("ISEE3_CHAIN_PM_PIPELINE")) != 0;
  if (pipeline) {
    e.fft_peak_begin = pm_peak_begin; e.fft_peak_end = pm_peak_end; e.mix_begin = pm_mix_begin; e.mix_end = pm_mix_end;
  }
  pmdemod_source src = { iq_next, &a->src };
  pmdemod_sink dst = { blk_acquire, blk_commit, a->out };
  t_stage_ms = 0;
  a->rc = pmdemod_run_io(&a->o, &e, &src, &dst, a->src.f, stderr, NULL, 0, NULL);
  a->ms = t_stage_ms;
  blk_close(a->out);
  return NULL;
}
static void *sy_thread(void *p) {
  sy_arg *a = p;
  symdemod_engine e = { sy_create, sy_load, sy_ts, sy_demod, sy_destroy, sy_slide, sy_put, sy_scan, sy_window };
  if (getenv("SYMDEMOD_STEPWISE") && atoi(getenv("SYMDEMOD_STEPWISE"))) e.window = NULL;
  /* this thread writes into the pipe the Viterbi stage reads: should that stage ever go away first, the write must fail
   * with EPIPE (a stage error) and not raise SIGPIPE, whose default action would end the HOST program */
  sigset_t sp;
  sigemptyset(&sp); sigaddset(&sp, SIGPIPE);
  pthread_sigmask(SIG_BLOCK, &sp, NULL);
  t_stage_ms = 0;
  a->rc = symdemod_run_blk(&a->o, &e, blk_next, a->in, a->out, stderr);
  a->ms = t_stage_ms;
  a->t_done = now_ms() - a->t0;
  __atomic_store_n(a->done, 1, __ATOMIC_RELEASE);    /* before the pipe closes: what vdecode reads from now on is all there is */
  fclose(a->out);
  blk_reader_gone(a->in);
  return NULL;
}
static void *vd_thread(void *p) {
  vd_arg *a = p;
  vdecode_result r;
  vdecode_engine e = { vd_create, vd_init, vd_stream, vd_destroy, 2 * g_chunk, vd_whole, vd_read_limit, NULL, NULL };
  if (a->progressive) { e.progressive_feed = vd_prog_feed; e.progressive_end = vd_prog_end; }
  t_stage_ms = 0;
  t_front_done = a->front_done;
  t_expected_bits = a->expected_bits;
  a->rc = vdecode_run(&a->o, &e, a->fd_in, a->out, stderr, &r);
  a->ms = t_stage_ms;
  fflush(a->out);
  if (a->rc != 0) {                 /* a stage that failed mid-stream: let symdemod finish writing (it stops at its own end of input) */
    char sink[4096];
    while (read(a->fd_in, sink, sizeof sink) > 0) {}
  }
  close(a->fd_in);
  return NULL;
}


static __thread char g_chain_err[256];
static __thread double g_stage_ms[3];
const char *isee3_chain_last_error(void) { return g_chain_err; }
void isee3_chain_last_stage_ms(double ms[3]) { ms[0] = g_stage_ms[0]; ms[1] = g_stage_ms[1]; ms[2] = g_stage_ms[2]; }

void isee3_chain_default_opts(isee3_chain_opts *o) {
  memset(o, 0, sizeof *o);
  o->samprate = 250000; o->binsize = 4; o->decode_delay = 200;
}

static int chain_run(const isee3_chain_opts *co, const iq_src *src, FILE *out) {
  pm_arg pa; sy_arg sa; vd_arg va;
  int p2[2];
  blkchan c1;
  pmdemod_default_opts(&pa.o); symdemod_default_opts(&sa.o); vdecode_default_opts(&va.o);
  pa.o.argv0 = "isee3chain/pmdemod"; sa.o.argv0 = "isee3chain/symdemod"; va.o.argv0 = "isee3chain/vdecode";
  pa.o.samprate = co->samprate; sa.o.samprate = (int)co->samprate;
  pa.o.binsize = co->binsize; pa.o.search_freq = co->search_freq; pa.o.search_width = co->search_width; pa.o.flip = co->flip;
  if (co->symrate) symdemod_set_symrate(&sa.o, co->symrate);   /* symdemod -c semantics; no getopt in a library that runs concurrent chains */
  va.o.decode_delay = co->decode_delay;
  /* how the Viterbi stage takes its symbols.  "progressive": one stream, fed as it arrives, second decoder joining at a
   * cut placed from the expected number of bits (v224hip_progressive_*).  "block": block by block as the reference does,
   * long blocks shared once the front end has finished (bits leave as they are decoded: the form for pipes).  "whole":
   * wait for all symbols, then split (ISEE3_CHAIN_WHOLE=1; measured 72 vs 57 ms in round 1).  Default: progressive whenever
   * the capture's length is known.  Where the Viterbi decoder is the slower part (250 kS/s) the second decoder joins; where
   * the front end is (10 MS/s) one decoder keeps up with the symbols and the cut is never placed -- and that one decoder
   * still does better than block mode (48 s of 10 MS/s: 40.2 against 45.8 ms): it advances in whole chunks, asynchronously,
   * where a block of whatever the pipe held costs remainder passes, two switches of the metric order and a wait. */
  {
    const char *m = getenv("ISEE3_CHAIN_MODE");
    const double secs = src->iq ? (double)src->nsamples / co->samprate : 0;
    va.expected_bits = (long long)(secs * sa.o.symrate / 2);
    va.progressive = m ? !strcmp(m, "progressive") : va.expected_bits > 0;
    va.o.whole_input = va.progressive || (m && !strcmp(m, "whole")) || (getenv("ISEE3_CHAIN_WHOLE") && atoi(getenv("ISEE3_CHAIN_WHOLE")));
  }
  pa.o.quiet = sa.o.quiet = va.o.quiet = !co->verbose;
  pthread_once(&g_chunk_once, chunk_from_env);
  if (pipe(p2)) { snprintf(g_chain_err, sizeof g_chain_err, "pipe() failed"); return 2; }
#ifdef F_SETPIPE_SZ
  fcntl(p2[1], F_SETPIPE_SZ, 1 << 20);
#endif
  blk_init(&c1);
  volatile int front_done = 0;
  pa.src = *src; pa.out = &c1;
  sa.in = &c1; sa.out = fdopen(p2[1], "w"); sa.done = &front_done;
  va.fd_in = p2[0]; va.out = out; va.front_done = &front_done;
  pthread_t t1, t2, t3;
  sa.t0 = now_ms(); sa.t_done = 0;
  pthread_create(&t1, NULL, pm_thread, &pa);
  pthread_create(&t2, NULL, sy_thread, &sa);
  pthread_create(&t3, NULL, vd_thread, &va);
  pthread_join(t1, NULL); pthread_join(t2, NULL); pthread_join(t3, NULL);
  blk_free(&c1);
  free(pa.src.buf);
  g_stage_ms[0] = pa.ms; g_stage_ms[1] = sa.ms; g_stage_ms[2] = va.ms;
  if (getenv("V224HIP_VERBOSE"))
    fprintf(stderr, "isee3chain: last symbol out of symdemod %.2f ms after the start, all stages done after %.2f ms\n", sa.t_done, now_ms() - sa.t0);
  if (!pa.rc && !sa.rc && va.rc == -2) {
    snprintf(g_chain_err, sizeof g_chain_err, "vdecode: the decoded bits could not be written (output buffer too small, or the output was closed)");
    return 2;
  }
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
  setvbuf(in, NULL, _IOFBF, 1 << 20);
  iq_src src = { NULL, 0, 0, 0, in, NULL };
  int rc = chain_run(o, &src, out);
  fclose(in); fclose(out);
  return rc;
}

static int run_memory(const isee3_chain_opts *o, const int16_t *iq, size_t nsamples, int is_dev, char *out, size_t cap, size_t *nout) {
  FILE *mo = fmemopen(out, cap, "w");
  if (!mo) { snprintf(g_chain_err, sizeof g_chain_err, "fmemopen failed"); return 2; }
  setvbuf(mo, NULL, _IONBF, 0);                       /* write straight into the caller's buffer */
  iq_src src = { iq, nsamples, 0, is_dev, NULL, NULL };     /* the capture is read where it lies */
  int rc = chain_run(o, &src, mo);
  long pos = ftell(mo);
  fclose(mo);
  if (nout) *nout = pos > 0 ? (size_t)pos : 0;
  return rc;
}
int isee3_chain_run_mem(const isee3_chain_opts *o, const int16_t *iq, size_t nsamples, char *out, size_t cap, size_t *nout) {
  return run_memory(o, iq, nsamples, 0, out, cap, nout);
}
int isee3_chain_run_dev(const isee3_chain_opts *o, const int16_t *d_iq, size_t nsamples, char *out, size_t cap, size_t *nout) {
  return run_memory(o, d_iq, nsamples, 1, out, cap, nout);
}
