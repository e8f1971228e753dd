/* decode -- ISEE-3/ICE framed Viterbi decoder stage on MI355X (Viterbi mode of reference decode.c: run as
 * `decode -V`).  uint8 soft symbols on stdin, the reference's frame dump on stdout (byte-identical).  Frames are
 * decoded in batches by v224hip_decode_frames() on three decoder objects (libviterbi224_hip.so, gfx950); there is no
 * CPU fallback, and the Fano sequential decoder of the reference's default mode is not part of this build. */
#include <locale.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "decode_core.h"
#include "../../include/viterbi224.h"
#include "../../include/viterbi224_hip.h"

#define NDEC 3            /* two fill the GPU; the third covers the others' init / traceback gaps (0.85 -> 0.80 ms per frame) */
typedef struct { void *dec[NDEC]; unsigned char *buf; } hipctx;
static void hip_destroy(void *c) {
  hipctx *h = c;
  if (!h) return;
  for (int i = 0; i < NDEC; i++) if (h->dec[i]) delete_viterbi224(h->dec[i]);
  free(h->buf); free(h);
}
static void *hip_create(void) {
  hipctx *h = calloc(1, sizeof *h);
  if (!h) return NULL;
  h->buf = malloc((size_t)16 * DECODE_FRAMESYMBOLS);
  /* decode.c:139 creates 1 024 rows; two frames padded to whole 15-step passes let v224hip_decode_frames run a frame's
     traceback under the next frame's passes (the reference's call pattern through init / update / chainback is unchanged) */
  for (int i = 0; i < NDEC; i++) h->dec[i] = create_viterbi224(2 * ((DECODE_FRAMEBITS + 14) / 15 * 15));
  for (int i = 0; i < NDEC; i++) if (!h->buf || !h->dec[i]) { hip_destroy(h); return NULL; }
  return h;
}
static int hip_frames(void *c, const unsigned char *const *frames, int n, unsigned char *out) {
  hipctx *h = c;
  if (n > 16) return -1;
  for (int f = 0; f < n; f++) memcpy(h->buf + (size_t)f * DECODE_FRAMESYMBOLS, frames[f], DECODE_FRAMESYMBOLS);
  return v224hip_decode_frames(h->dec, n < NDEC ? n : NDEC, h->buf, n, DECODE_FRAMEBITS,
                               (int)(DECODE_SYNCWORD & 0xffffff), (unsigned)(DECODE_SYNCWORD & 0xffffff), out);
}

int main(int argc, char **argv) {
  decode_opts o;
  decode_result r;
  const char *lang = getenv("LANG");
  setlocale(LC_ALL, lang ? lang : "en_US.utf8");          /* decode.c:56-59 */
  decode_parse_args(&o, argc, argv);
  decode_engine e = { hip_create, hip_frames, hip_destroy };
  int rc = decode_run(&o, &e, 0, stdout, stderr, &r);
  if (rc < 0) { fprintf(stderr, "%s: decoder engine failed: %s\n", o.argv0, v224hip_last_error()); return 2; }
  if (o.verbose) fprintf(stderr, "%s: %lld frames (%lld good), %lld batches, %lld frames decoded, %lld speculated in vain\n",
                         o.argv0, r.frames, r.good, r.batches, r.decoded, r.wasted);
  return rc;
}
