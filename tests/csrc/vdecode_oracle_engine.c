/* Test harness: the PRODUCT's vdecode host logic (isee3-decoder_amd/cli/vdecode_core.c) driven by
 * the CPU oracle instead of the HIP library, so the pairing / phase-flip / start-up / statistics
 * logic can be checked against the reference's vdecode output on a machine without a GPU.
 * TEST INFRASTRUCTURE ONLY. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../isee3-decoder_amd/cli/vdecode_core.h"
#include "../../oracle/oracle.h"

static void *eng_create(int len) { return orc_v224_create(len, ORC_V224_FAST); }
static int eng_init(void *h, int s) { return orc_v224_init(h, s); }
static unsigned long long steps;
static int eng_stream(void *h, const unsigned char *syms, int nbits, int delay, unsigned char *out) {
  for (int i = 0; i < nbits; i++) {
    orc_v224_update(h, syms + 2 * i, 1);
    steps++;
    out[i] = steps >= (unsigned long long)delay ? (unsigned char)orc_v224_decodebit(h, delay, 0) : 0xff;
    /* fault injection: an engine that hands back "no bit" for one trellis step after start-up */
    if (getenv("VDECODE_TEST_BADBIT") && steps == strtoull(getenv("VDECODE_TEST_BADBIT"), NULL, 10)) out[i] = 0xff;
  }
  return 0;
}
static int eng_whole(void *h, const unsigned char *syms, long long nbits, int delay, unsigned char *out) {
  return eng_stream(h, syms, (int)nbits, delay, out);
}
static void eng_destroy(void *h) { orc_v224_delete(h); }
static unsigned long eng_limit(void *h) { (void)h; return getenv("VDECODE_TEST_LIMIT") ? strtoul(getenv("VDECODE_TEST_LIMIT"), NULL, 10) : 0; }

/* progressive pair (VDECODE_WHOLE=2): decode each piece when it is announced, hand everything out at the end */
static unsigned char *pbuf; static long long pn, pcap, pcalls;
static int eng_feed(void *h, const unsigned char *syms, int nbits, int delay) {
  if (pn + nbits > pcap) { pcap = 2 * (pn + nbits) + 64; pbuf = realloc(pbuf, (size_t)pcap); if (!pbuf) return -1; }
  pcalls++;
  int rc = eng_stream(h, syms, nbits, delay, pbuf + pn);
  pn += nbits;
  return rc;
}
static int eng_end(void *h, long long nbits, int delay, unsigned char *out) {
  (void)h; (void)delay;
  if (nbits != pn) return -1;
  memcpy(out, pbuf, (size_t)nbits);
  fprintf(stderr, "PROGRESSIVE feeds=%lld\n", pcalls);
  return 0;
}

int main(int argc, char **argv) {
  vdecode_opts o;
  vdecode_result r;
  vdecode_parse_args(&o, argc, argv);
  vdecode_engine e = { eng_create, eng_init, eng_stream, eng_destroy, 0, eng_whole, eng_limit, NULL, NULL };
  o.whole_input = getenv("VDECODE_WHOLE") && atoi(getenv("VDECODE_WHOLE"));
  if (o.whole_input && atoi(getenv("VDECODE_WHOLE")) == 2) { e.progressive_feed = eng_feed; e.progressive_end = eng_end; }
  /* VDECODE_TEST_OUTCAP=n: the output is a memory stream of n bytes, as in isee3_chain_run_mem with a short buffer */
  FILE *out = stdout;
  char *mem = NULL;
  size_t cap = getenv("VDECODE_TEST_OUTCAP") ? strtoul(getenv("VDECODE_TEST_OUTCAP"), NULL, 10) : 0;
  if (cap) { mem = malloc(cap); out = fmemopen(mem, cap, "w"); setvbuf(out, NULL, _IONBF, 0); }
  int rc = vdecode_run(&o, &e, 0, out, stderr, &r);
  if (cap) { long n = ftell(out); fclose(out); fwrite(mem, 1, n > 0 ? (size_t)n : 0, stdout); free(mem); }
  fprintf(stderr, "RESULT rc=%d bits=%llu symerrs=%llu flips=%d\n", rc, r.bits_out, r.symerrs_total, r.flips);
  return rc == -2 ? 3 : rc ? 2 : 0;
}
