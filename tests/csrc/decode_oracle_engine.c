/* Test harness: the PRODUCT's framed-decode host logic (isee3-decoder_amd/cli/decode_core.c: frame sync, lock,
 * speculative batches, frame dump) driven by the CPU oracle instead of the HIP library, so it can be checked
 * against the reference's `decode -V` output on a machine without a GPU.  TEST INFRASTRUCTURE ONLY. */
#include <locale.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../isee3-decoder_amd/cli/decode_core.h"
#include "../../oracle/oracle.h"

static void *eng_create(void) { return orc_v224_create(DECODE_FRAMEBITS, ORC_V224_FAST); }
static int eng_frames(void *h, const unsigned char *const *frames, int n, unsigned char *out) {
  for (int f = 0; f < n; f++) {
    orc_v224_init(h, (int)(DECODE_SYNCWORD & 0xffffff));
    orc_v224_update(h, frames[f], DECODE_FRAMEBITS);
    orc_v224_chainback(h, out + 128 * f, DECODE_FRAMEBITS, (unsigned)(DECODE_SYNCWORD & 0xffffff));
  }
  return 0;
}
static void eng_destroy(void *h) { orc_v224_delete(h); }

int main(int argc, char **argv) {
  decode_opts o;
  decode_result r = {0, 0, 0, 0, 0};
  const char *lang = getenv("LANG");
  setlocale(LC_ALL, lang ? lang : "en_US.utf8");
  decode_parse_args(&o, argc, argv);
  decode_engine e = { eng_create, eng_frames, eng_destroy };
  int rc = decode_run(&o, &e, 0, stdout, stderr, &r);
  fprintf(stderr, "RESULT frames=%lld good=%lld batches=%lld decoded=%lld wasted=%lld\n", r.frames, r.good, r.batches,
          r.decoded, r.wasted);
  return rc < 0 ? 3 : rc;
}
