/* decode_core.h -- host logic of the framed `decode` stage (Viterbi mode), independent of the decoder engine.
 *
 * Mirrors reference decode.c:42-289 with Fano disabled (-V): refill logic :152-161,184-192, the 34-tap frame
 * sync correlator over one frame :162-181 (first maximum wins), init(0x819fbe) / update(1024) / chainback per
 * frame :219-222, sync-word check :238-249, frame dump :251-268, purge :270-282.
 *
 * The reference decodes one frame, looks at its last five bytes (lock), and only then knows where the next frame
 * starts.  Here the state machine is the same, but a frame that it asks for is fetched from a small cache filled
 * by BATCHES: when a frame at symbol P is needed and not cached, frames at P, P+2048, P+4096, ... (as far as the
 * input read so far reaches) are decoded together -- the positions the machine will ask for next if lock holds.
 * A speculated frame that is never asked for is discarded; the batch size adapts (1 after a miss, doubling up
 * to 16).  Results and control flow are those of the reference: stdout is byte-identical. */
#ifndef DECODE_CORE_H
#define DECODE_CORE_H
#include <stdio.h>

#define DECODE_FRAMEBITS     1024                    /* decode.c:21 */
#define DECODE_FRAMESYMBOLS  (2 * DECODE_FRAMEBITS)  /* decode.c:22 */
#define DECODE_SYNCBITS      34                      /* decode.c:23 */
#define DECODE_SYNCWORD      0x12fc819fbeULL         /* decode.c:24 */

typedef struct {
  int verbose, viterbi_enabled, fano_enabled, no_bad_frames, persistent;
  double symrate;                  /* -r, default 1024 */
  double fano_scale;               /* -s, -m, -d: parsed like the reference, unused without Fano */
  unsigned long fano_maxcycles;
  int fano_delta;
  const char *argv0;
} decode_opts;

/* decoder engine: the product binds this to v224hip_decode_frames() on two decoder objects */
typedef struct {
  void *(*create)(void);
  /* frames[f] points at the 2048 symbols of frame f; out gets 128 bytes per frame: the result of
   * init(0x819fbe); update(frames[f], 1024); chainback(out + 128 f, 1024, 0x819fbe)            */
  int   (*decode_frames)(void *ctx, const unsigned char *const *frames, int nframes, unsigned char *out);
  void  (*destroy)(void *ctx);
} decode_engine;

typedef struct { long long frames, good, batches, decoded, wasted; } decode_result;

void decode_default_opts(decode_opts *o);
/* getopt loop of decode.c:71-104 (prints the reference's usage line on an unknown option) */
void decode_parse_args(decode_opts *o, int argc, char **argv);
/* 0 on success, 1 / 2 for the reference's exit codes (:112-115, :139-143), -1 on an engine failure */
int  decode_run(const decode_opts *o, const decode_engine *e, int fd_in, FILE *out, FILE *err, decode_result *res);
#endif
