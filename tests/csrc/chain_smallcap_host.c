/* Test host (C, so that SIGPIPE keeps its default action -- Python ignores it): isee3_chain_run_mem with an output
 * buffer that is too small.  The Viterbi stage fails on its first short write while symdemod is still producing; the
 * library must come back with rc 2 and a message that says what happened, and this process must still be alive to
 * print it.  argv: samprate binsize cap < int16 IQ on stdin.  TEST INFRASTRUCTURE ONLY. */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../include/isee3_chain.h"

int main(int argc, char **argv) {
  if (argc < 4) return 64;
  size_t cap = strtoul(argv[3], NULL, 10), n = 0, have = 0, room = 1 << 20;
  int16_t *iq = malloc(room * sizeof *iq);
  for (;;) {
    size_t got = fread(iq + have, sizeof *iq, room - have, stdin);
    have += got;
    if (got == 0) break;
    if (have == room) { room *= 2; iq = realloc(iq, room * sizeof *iq); }
  }
  isee3_chain_opts o;
  isee3_chain_default_opts(&o);
  o.samprate = atof(argv[1]); o.binsize = atof(argv[2]); o.symrate = "1024";
  char *out = malloc(cap ? cap : 1);
  int rc = isee3_chain_run_mem(&o, iq, have / 2, out, cap, &n);
  printf("rc=%d n=%zu err=%s\n", rc, n, rc ? isee3_chain_last_error() : "");
  isee3_chain_release();
  return 0;
}
