/* isee3_icesync.h -- C-ABI (libisee3chain.so) of the host side of the FFT sync-vector correlator of the reference's legacy one-shot program
 * (icesync.c:55-208).  The transforms, the product and the peak search run in libisee3dsp_hip.so
 * (isync_*); here: the sync vector (icesync.c:55-97) and the bookkeeping of generate_sync / fft_sync_search.  The rest of
 * icesync.c (frame loop, symbol integration, framed Viterbi) duplicates symdemod / decode and is not rebuilt. */
#ifndef ISEE3_ICESYNC_H
#define ISEE3_ICESYNC_H
#ifdef __cplusplus
extern "C" {
#endif

#define ICESYNC_FRAMEBITS 1024          /* icesync.c:26 */
#define ICESYNC_SYNCBITS  34            /* icesync.c:27 */
#define ICESYNC_FAIL (-1234567890)      /* icesync.c:31 SYNC_FAIL */

/* icesync.c:55-97: the 40 constant bits that end every minor frame (0x12fc819fbe) go through the encoder; the last 34
 * symbols are Manchester coded at `symbolsamples` samples per symbol (symbol 1: first half -1, second half +1).
 * Returns Synclen = (int)(34 * symbolsamples + 1), or -1 if vec (cap doubles) is too small; the rest of vec is zeroed. */
int icesync_sync_vector(double symbolsamples, double *vec, int cap);

typedef struct icesync_corr icesync_corr;
/* generate_sync + the correlator set-up (icesync.c:99-135): Symbolsamples = samprate / symrate, Framesamples =
 * Symbolsamples * 2 * 1024 (:255-256), transform size 2^20 (:102-103; corr_size_log2 = 0) or 2^corr_size_log2 */
icesync_corr *icesync_corr_create(double samprate, double symrate, int corr_size_log2);
/* fft_sync_search (icesync.c:139-208, without the plot file): samples = at least Framesamples int16 values in host memory */
int icesync_corr_search(icesync_corr *c, const short *samples, int low, int high, double *maxpeak);
double icesync_corr_framesamples(const icesync_corr *c);
int    icesync_corr_synclen(const icesync_corr *c);
void   icesync_corr_destroy(icesync_corr *c);

#ifdef __cplusplus
}
#endif
#endif
