/* vdecode -- ISEE-3/ICE Viterbi decoder pipe stage on MI355X.
 * Drop-in for reference vdecode.c: same options (-d n -p -i n -q -F), uint8 offset-128 soft symbols
 * on stdin, ASCII '0'/'1' on stdout (byte-identical), status on stderr.  The trellis runs in
 * libviterbi224_hip.so (hand-written HIP, gfx950); there is no CPU fallback. */
#include <locale.h>
#include <stdio.h>
#include <stdlib.h>
#include "vdecode_core.h"
#include "../../include/viterbi224_hip.h"

static int g_chunk = 1020;      /* 68 passes of 15 steps (engine LDS15); any value works */
static void *eng_create(int len) {
  void *h = create_viterbi224(len);
  if (h) v224hip_set_option(h, "chunk", g_chunk);
  return h;
}
static int eng_init(void *h, int s) { return init_viterbi224(h, s); }
static int eng_stream(void *h, const unsigned char *syms, int nbits, int delay, unsigned char *out) {
  return v224hip_stream_decode(h, syms, nbits, delay, out);
}
static void eng_destroy(void *h) { delete_viterbi224(h); }

int main(int argc, char **argv) {
  vdecode_opts o;
  vdecode_result r;
  const char *lang = getenv("LANG");
  setlocale(LC_ALL, lang ? lang : "en_US.utf8");        /* vdecode.c:60-63 */
  vdecode_parse_args(&o, argc, argv);
  int chunk = getenv("V224HIP_CHUNK") ? atoi(getenv("V224HIP_CHUNK")) : 1020;
  if (chunk < 1) chunk = 1020;
  g_chunk = chunk;
  vdecode_engine e = { eng_create, eng_init, eng_stream, eng_destroy, 2 * chunk };
  if (vdecode_run(&o, &e, 0, stdout, stderr, &r) != 0) {
    fprintf(stderr, "%s: decoder engine failed: %s\n", o.argv0, v224hip_last_error());
    return 2;
  }
  return 0;                                              /* vdecode.c:188 */
}
