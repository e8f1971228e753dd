/* vdecode_oracle.c -- the vdecode pipe stage as a buffer-to-buffer function (TEST INFRASTRUCTURE
 * ONLY).  Follows vdecode.c:38-189 step for step on top of the oracle's port-semantics decoder:
 * symbol ring pre-fill :55-59, delay clamp :86-91, create(delay+1)/init(0) :94-96, 34-tap sync
 * correlator and per-frame phase decision :110-141, one-bit update + decodebit(delay,0) with the
 * first `delay` outputs suppressed :142-158, re-encode symbol-error tally :159-177.
 * Pinned against oracle/_ref/vdecode_port_ref via tests/golden/vdecode_*.
 */
#include <string.h>
#include "oracle.h"

#define RING 4096                 /* vdecode.c:20 */
#define FRAMESYMBOLS 2048         /* vdecode.c:15 */

static const int8_t sync_sign[34] = {   /* vdecode.c:27-30: +1 where the expected symbol is 1 */
  -1, 1, 1, 1, 1, 1, 1, -1, 1, -1, 1, 1, 1, 1, -1, -1, 1,
   1, -1, -1, 1, 1, -1, 1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1 };

size_t orc_vdecode(const uint8_t *syms, size_t nsyms, int decode_delay, int start_phase,
                   int dontflip, int mode, char *out, orc_vdecode_stats *st) {
  uint8_t ring[RING], pair[2] = { 0, 0 };
  for (int i = 0; i < RING; i += 2) { ring[i] = ORC_G1FLIP ? 255 : 0; ring[i + 1] = ORC_G2FLIP ? 255 : 0; }
  if (decode_delay < 24) decode_delay = 200;
  int startup = decode_delay, pos = start_phase ? 1 : 0;
  int sync_count = 0, peak_in = -1000000, peak_out = -1000000;
  uint64_t reenc = 0, symerrs = 0;
  size_t nout = 0; int flips = 0;
  void *vd = orc_v224_create(decode_delay + 1, mode);
  if (!vd) return 0;
  orc_v224_init(vd, 0);

  for (size_t n = 0; n < nsyms; n++) {
    uint8_t c = syms[n];
    ring[pos] = c; pair[pos % 2] = c;
    if (!dontflip) {
      int sum = 0;
      for (int k = 0; k < 34; k++) sum += sync_sign[k] * ((int)ring[(RING + pos + k - 33) % RING] - 128);
      if ((pos % 2) == 0) { if (sum > peak_out) peak_out = sum; }
      else {
        if (sum > peak_in) peak_in = sum;
        if (++sync_count >= FRAMESYMBOLS) {
          sync_count = 0;
          if (peak_out > peak_in) { flips++; if ((pos % 2) == 0) pos++; else pos--; }
          peak_in = peak_out = -1000000;
        }
      }
    }
    if ((pos % 2) == 1) {
      orc_v224_update(vd, pair, 1);
      if (startup == 0) {
        int bit = orc_v224_decodebit(vd, decode_delay, 0);
        out[nout++] = bit ? '1' : '0';
        reenc = (reenc << 1) | (unsigned)bit;
      } else startup--;
      int s1 = ORC_G1FLIP ^ __builtin_parityll(reenc & ORC_POLY1);
      int s2 = ORC_G2FLIP ^ __builtin_parityll(reenc & ORC_POLY2);
      if (startup == 0)
        symerrs += (unsigned)(s1 ^ (ring[(RING + pos - 2 * (decode_delay + ORC_K - 2) - 1) % RING] > 128))
                 + (unsigned)(s2 ^ (ring[(RING + pos - 2 * (decode_delay + ORC_K - 2)) % RING] > 128));
    }
    pos = (pos + 1) % RING;
  }
  orc_v224_delete(vd);
  if (st) { st->bits = nout; st->symerrs = symerrs; st->flips = flips; }
  return nout;
}
