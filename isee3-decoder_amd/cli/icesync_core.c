/* icesync_core.c -- see include/isee3_icesync.h.  Plain C11. */
#include "../../include/isee3_icesync.h"
#include <stdlib.h>
#include <string.h>
#include "isee3_code.h"
#include "../../include/isee3_dsp_hip.h"

struct icesync_corr { void *h; double symbolsamples, framesamples; int synclen, size; };

int icesync_sync_vector(double symbolsamples, double *vec, int cap) {
  /* the encoder of encode.c:17-35 on {12 fc 81 9f be 00 00 00 00 00}, MSB first, starting state 0 */
  static const unsigned char data[10] = { 0x12, 0xfc, 0x81, 0x9f, 0xbe, 0, 0, 0, 0, 0 };
  unsigned char symbols[2 * 8 * 10];
  unsigned long long enc = 0;
  int n = 0;
  for (int b = 0; b < 10; b++)
    for (int i = 7; i >= 0; i--) {
      enc = (enc << 1) | ((data[b] >> i) & 1u);
      symbols[n++] = (unsigned char)(ISEE3_G1FLIP ^ isee3_parity(enc & ISEE3_POLY1));
      symbols[n++] = (unsigned char)(ISEE3_G2FLIP ^ isee3_parity(enc & ISEE3_POLY2));
    }
  int synclen = (int)(ICESYNC_SYNCBITS * symbolsamples + 1);          /* icesync.c:77 */
  if (synclen > cap) return -1;
  memset(vec, 0, sizeof(double) * (size_t)cap);
  int ind = 0;
  for (int k = 0; k < ICESYNC_SYNCBITS; k++) {                         /* icesync.c:88-97 */
    const int s = symbols[k + 80 - ICESYNC_SYNCBITS];
    for (; ind < (k + 0.5) * symbolsamples; ind++) vec[ind] = s ? -1 : 1;
    for (; ind < (k + 1) * symbolsamples; ind++) vec[ind] = s ? 1 : -1;
  }
  return synclen;
}

icesync_corr *icesync_corr_create(double samprate, double symrate, int corr_size_log2) {
  icesync_corr *c = calloc(1, sizeof *c);
  double *vec = NULL;
  if (!c) return NULL;
  c->symbolsamples = samprate / symrate;                               /* icesync.c:255 */
  c->framesamples = c->symbolsamples * 2 * ICESYNC_FRAMEBITS;          /* :256 */
  c->size = 1 << (corr_size_log2 > 0 ? corr_size_log2 : 20);           /* :102-103 */
  int cap = (int)(ICESYNC_SYNCBITS * c->symbolsamples + 2);
  if (cap > c->size || c->framesamples > c->size) goto fail;
  vec = malloc(sizeof(double) * (size_t)cap);
  if (!vec) goto fail;
  c->synclen = icesync_sync_vector(c->symbolsamples, vec, cap);
  c->h = isync_create(c->size);
  if (c->synclen < 0 || !c->h || isync_set_vector(c->h, vec, c->synclen) != 0) goto fail;
  free(vec);
  return c;
fail:
  free(vec);
  icesync_corr_destroy(c);
  return NULL;
}

int icesync_corr_search(icesync_corr *c, const short *samples, int low, int high, double *maxpeak) {
  int n = 0, peak = ICESYNC_FAIL;
  while (n < c->framesamples) n++;                                     /* the loop bound of icesync.c:153 */
  if (isync_search(c->h, samples, n, 0, low, high, &peak, maxpeak, NULL) != 0) return ICESYNC_FAIL;
  return peak;
}
double icesync_corr_framesamples(const icesync_corr *c) { return c->framesamples; }
int icesync_corr_synclen(const icesync_corr *c) { return c->synclen; }
void icesync_corr_destroy(icesync_corr *c) {
  if (!c) return;
  if (c->h) isync_destroy(c->h);
  free(c);
}
