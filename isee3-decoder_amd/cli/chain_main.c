/* isee3chain -- pmdemod | symdemod | vdecode in one process (see chain_core.c / include/isee3_chain.h).
 * stdin = int16 I,Q pairs, stdout = ASCII '0'/'1'.  Options:
 *     -r Hz  sample rate (pmdemod -r, symdemod -r)        -b Hz  FFT bin size (pmdemod -b)
 *     -c Hz  symbol rate (symdemod -c)                    -d n   decode delay (vdecode -d)
 *     -W Hz / -S Hz / -f  pmdemod search width / start / flip          -v  keep the stages' status lines */
#include <locale.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include "../../include/isee3_chain.h"

int main(int argc, char **argv) {
  isee3_chain_opts o;
  int c;
  const char *lang = getenv("LANG");
  setlocale(LC_ALL, lang ? lang : "en_US.utf8");
  isee3_chain_default_opts(&o);
  while ((c = getopt(argc, argv, "r:b:c:d:W:S:fv")) != -1) {
    switch (c) {
    case 'r': o.samprate = atof(optarg); break;
    case 'b': o.binsize = atof(optarg); break;
    case 'c': o.symrate = optarg; break;
    case 'd': o.decode_delay = atoi(optarg); break;
    case 'W': o.search_width = atof(optarg); break;
    case 'S': o.search_freq = atof(optarg); break;
    case 'f': o.flip = 1; break;
    case 'v': o.verbose = 1; break;
    default: fprintf(stderr, "usage: isee3chain [-r Hz] [-b Hz] [-c Hz] [-d n] [-W Hz] [-S Hz] [-f] [-v] < iq > bits\n"); return 1;
    }
  }
  int rc = isee3_chain_run_fd(&o, 0, 1);
  if (rc) fprintf(stderr, "isee3chain: %s\n", isee3_chain_last_error());
  return rc;
}
