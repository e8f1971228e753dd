/* icesync_oracle.c -- CPU restatement of the reference's FFT sync-vector correlator, icesync.c:55-208
 * (SURVEY 8 f4).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED: icesync.c needs FFTW3 (icesync.c:17, reference Makefile:54), which the image lacks, so the
 * reference program cannot be built here and there is no fixture of its output.  The restatement follows
 *   generate_sync   :55-97    sync word through the encoder, last 34 symbols, Manchester +-1 vector
 *   correlator set-up :99-135  zero-padded vector, forward transform, conjugate
 *   fft_sync_search :139-208  load + zero-pad, all-zero check, transform, multiply, inverse (unnormalised), first
 *                             maximum > 0 in [low, high), fold indices above size/2
 * with FFTW's r2c / c2r replaced by full complex transforms of real data through orc_fft_forward (same sums; FFTW's
 * c2r is the unnormalised inverse sum_k X[k] e^{+2 pi i k n / N} of a Hermitian spectrum).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define ICE_FRAMEBITS 1024        /* icesync.c:26 */
#define ICE_SYNCBITS  34          /* icesync.c:27 */

/* icesync.c:55-97.  symbolsamples = the GLOBAL Symbolsamples the reference reads (its symrate argument only feeds a
 * printf).  Returns Synclen; vec (cap doubles) gets the vector, zero beyond what the loops write. */
int orc_icesync_sync_vector(double symbolsamples, double *vec, int cap) {
  uint8_t data[10] = { 0x12, 0xfc, 0x81, 0x9f, 0xbe, 0, 0, 0, 0, 0 };
  uint8_t symbols[2 * 8 * 10];
  orc_encode(symbols, data, 10, 0);
  int synclen = (int)(ICE_SYNCBITS * symbolsamples + 1);
  if (synclen > cap) return -1;
  memset(vec, 0, sizeof(double) * (size_t)cap);
  int ind = 0;
  for (int k = 0; k < ICE_SYNCBITS; k++) {
    for (; ind < (k + 0.5) * symbolsamples; ind++) vec[ind] = symbols[k + 80 - ICE_SYNCBITS] ? -1 : 1;
    for (; ind < (k + 1) * symbolsamples; ind++) vec[ind] = symbols[k + 80 - ICE_SYNCBITS] ? 1 : -1;
  }
  return synclen;
}

/* icesync.c:139-208; result (optional, corr_size doubles) gets Corr_result */
int orc_icesync_search(const double *vec, int synclen, int corr_size, const int16_t *samples, double framesamples,
                       int low, int high, double *maxpeak_out, double *result) {
  size_t n = (size_t)corr_size;
  double *a = calloc(2 * n, sizeof(double)), *A = malloc(sizeof(double) * 2 * n);
  double *b = calloc(2 * n, sizeof(double)), *B = malloc(sizeof(double) * 2 * n);
  int rc = ORC_SYNC_FAIL, nonzero = 1, i;
  if (!a || !A || !b || !B) goto done;
  for (i = 0; i < synclen && i < corr_size; i++) a[2 * i] = vec[i];
  orc_fft_forward(a, A, corr_size);                          /* Corr_ff1; conj below (icesync.c:133-134) */
  for (i = 0; i < framesamples && i < corr_size; i++) {      /* :151-156 */
    b[2 * i] = samples[i];
    if (samples[i] != 0) nonzero = 0;
  }
  if (nonzero) goto done;                                    /* :157-158 */
  orc_fft_forward(b, B, corr_size);                          /* Corr_ff2 */
  for (size_t k = 0; k < n; k++) {                           /* :166-167: D *= conj(V); then conj for the inverse */
    double dr = B[2 * k], di = B[2 * k + 1], vr = A[2 * k], vi = -A[2 * k + 1];
    a[2 * k] = dr * vr - di * vi;
    a[2 * k + 1] = -(dr * vi + di * vr);
  }
  orc_fft_forward(a, A, corr_size);                          /* Corr_ffr: r[n] = Re(FFT(conj Y)[n]) */
  if (result) for (size_t k = 0; k < n; k++) result[k] = A[2 * k];
  {
    int peakindex = -1; double maxpeak = 0;                  /* :188-199 */
    if (high > corr_size) high = corr_size;
    for (i = low; i < high; i++) if (A[2 * i] > maxpeak) { maxpeak = A[2 * i]; peakindex = i; }
    if (maxpeak_out) *maxpeak_out = maxpeak;
    if (maxpeak == 0) goto done;                             /* :200-203 */
    if (peakindex > corr_size / 2) peakindex = corr_size - peakindex;   /* :204-205 */
    rc = peakindex;
  }
done:
  free(a); free(A); free(b); free(B);
  return rc;
}
