/* vdecode -- ISEE-3/ICE Viterbi decoder pipe stage on MI355X.
 * Drop-in for reference vdecode.c: same options (-d n -p -i n -q -F), uint8 offset-128 soft symbols
 * on stdin, ASCII '0'/'1' on stdout (byte-identical), status on stderr.  The trellis runs in
 * libviterbi224_hip.so (hand-written HIP, gfx950); there is no CPU fallback. */
#include <locale.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include "vdecode_core.h"
#include "../../include/viterbi224_hip.h"

static int g_chunk = 1020;      /* 68 passes of 15 steps (engine LDS15); any value works */
/* one decoder for block-wise streaming; when ALL input is known up front (stdin is a regular file) a second decoder
 * joins and the stream is decoded in two halves at once, verified at the seam (v224hip_stream_decode_split) */
typedef struct { void *d[2]; int len, holder; } eng_ctx;     /* holder: the decoder that carries the stream's state */
static void eng_destroy(void *p) {
  eng_ctx *c = p;
  if (!c) return;
  for (int i = 0; i < 2; i++) if (c->d[i]) delete_viterbi224(c->d[i]);
  free(c);
}
static void *eng_create(int len) {
  eng_ctx *c = calloc(1, sizeof *c);
  if (!c) return NULL;
  c->len = len;
  c->d[0] = create_viterbi224(len);
  if (!c->d[0]) { free(c); return NULL; }
  v224hip_set_option(c->d[0], "chunk", g_chunk);
  return c;
}
static int eng_init(void *h, int s) { eng_ctx *c = h; c->holder = 0; return init_viterbi224(c->d[0], s); }
#define SPLIT_WARM (14 * 1020)
#define SHARE_WARM (3 * 1020)       /* one chunk of seam window + 2 040 bits of forgetting (chain_core.c: why) */
/* `... | vdecode`: blocks arrive as the pipe delivers them.  With VDECODE_SHARE=1 a long one (the producer ran ahead) is
 * shared between two decoders, verified at the seam (v224hip_stream_decode_shared).  Off by default: two decoders take
 * every wave slot of every CU, and a pmdemod / symdemod feeding this pipe from the same GPU would starve. */
static int g_share;
static int eng_stream(void *h, const unsigned char *syms, int nbits, int delay, unsigned char *out) {
  eng_ctx *c = h;
  if (g_share && 5 * nbits >= 11 * SHARE_WARM && !c->d[1]) {
    c->d[1] = create_viterbi224(c->len);
    if (c->d[1]) v224hip_set_option(c->d[1], "chunk", g_chunk);
  }
  if (!c->d[1]) return v224hip_stream_decode(c->d[c->holder], syms, nbits, delay, out);
  return v224hip_stream_decode_shared(c->d, 2, &c->holder, syms, nbits, delay, out, SHARE_WARM);
}
static int eng_whole(void *h, const unsigned char *s, long long n, int d, unsigned char *o) {
  eng_ctx *c = h;
  if (n > 0x7fffffff / 2) return -1;
  if (n < 6 * SPLIT_WARM) return v224hip_stream_decode(c->d[0], s, (int)n, d, o);        /* too short to gain */
  if (!c->d[1]) {
    c->d[1] = create_viterbi224(c->len);
    if (!c->d[1]) return v224hip_stream_decode(c->d[0], s, (int)n, d, o);
    v224hip_set_option(c->d[1], "chunk", g_chunk);
  }
  unsigned char *ds = v224hip_dev_alloc(2 * (size_t)n), *dout = v224hip_dev_alloc((size_t)n);
  int rc = -1, redone = 0;
  if (ds && dout && v224hip_h2d(ds, s, 2 * (size_t)n) == 0 &&
      v224hip_stream_decode_split(c->d, 2, ds, (int)n, d, dout, SPLIT_WARM, &redone) == 0 &&
      v224hip_d2h(o, dout, (size_t)n) == 0) rc = 0;
  v224hip_dev_free(ds); v224hip_dev_free(dout);
  return rc;
}

int main(int argc, char **argv) {
  vdecode_opts o;
  vdecode_result r;
  const char *lang = getenv("LANG");
  setlocale(LC_ALL, lang ? lang : "en_US.utf8");        /* vdecode.c:60-63 */
  vdecode_parse_args(&o, argc, argv);
  int chunk = getenv("V224HIP_CHUNK") ? atoi(getenv("V224HIP_CHUNK")) : 1020;
  if (chunk < 1) chunk = 1020;
  g_chunk = chunk;
  g_share = getenv("VDECODE_SHARE") && atoi(getenv("VDECODE_SHARE"));
  vdecode_engine e = { eng_create, eng_init, eng_stream, eng_destroy, 2 * chunk, eng_whole };
  {
    struct stat sb;                                        /* a file on stdin: all input is there, nobody waits for early bits */
    const char *w = getenv("VDECODE_WHOLE");
    o.whole_input = w ? atoi(w) : (fstat(0, &sb) == 0 && S_ISREG(sb.st_mode));
  }
  if (vdecode_run(&o, &e, 0, stdout, stderr, &r) != 0) {
    fprintf(stderr, "%s: decoder engine failed: %s\n", o.argv0, v224hip_last_error());
    return 2;
  }
  return 0;                                              /* vdecode.c:188 */
}
