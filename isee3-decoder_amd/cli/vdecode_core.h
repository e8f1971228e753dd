/* vdecode_core.h -- host logic of the vdecode pipe stage, independent of the decoder engine.
 *
 * Mirrors reference vdecode.c:38-189 (options :67-85, delay clamp :86-91, symbol pairing and the
 * 34-tap phase correlator :104-141, start-up suppression :151-158, re-encode symbol-error tally
 * :159-184).  The reference runs ONE trellis step and ONE 200-step traceback per input pair; here
 * the same pairs are collected per input block and handed to the engine in one call
 * (stream_decode), which is defined to return exactly what that per-pair loop would. */
#ifndef VDECODE_CORE_H
#define VDECODE_CORE_H
#include <stddef.h>
#include <stdio.h>

typedef struct {
  int decode_delay;      /* -d, default 200; < 24 => 200 with a warning (vdecode.c:86-88) */
  int start_phase;       /* -p */
  int status_interval;   /* -i, default 1024 */
  int quiet;             /* -q */
  int dontflip;          /* -F */
  int whole_input;       /* not a reference option: read ALL input before decoding (finite input, latency irrelevant) */
  const char *argv0;
} vdecode_opts;

/* decoder engine: the product binds these to libviterbi224_hip.so */
typedef struct {
  void *(*create)(int len);
  int   (*init)(void *h, int starting_state);
  /* out[i] = decodebit(delay,0) after the i-th of nbits single steps (0xff if history < delay) */
  int   (*stream_decode)(void *h, const unsigned char *syms, int nbits, int delay, unsigned char *out);
  void  (*destroy)(void *h);
  int   ring_extra;      /* rows the engine needs beyond decode_delay */
  /* optional (may be NULL): the same contract as stream_decode for one long stream right after init */
  int   (*stream_decode_whole)(void *h, const unsigned char *syms, long long nbits, int delay, unsigned char *out);
  /* optional (may be NULL): how many input bytes the next read() may take at most (0 = no preference).  An engine that
     wants to look at its surroundings between blocks keeps them short for a while. */
  unsigned long (*read_limit)(void *h);
  /* optional pair (both or neither; whole-input mode only): the one long stream is announced while it is being read --
     feed() gets the symbols of the next nbits trellis steps as soon as pass 1 has paired them and may start decoding, end()
     returns all nbits outputs under the contract of stream_decode_whole */
  int   (*progressive_feed)(void *h, const unsigned char *syms, int nbits, int delay);
  int   (*progressive_end)(void *h, long long nbits, int delay, unsigned char *out);
} vdecode_engine;

typedef struct {
  unsigned long long bits_out, symerrs_total;
  int flips;
} vdecode_result;

void vdecode_default_opts(vdecode_opts *o);
/* parse argv like vdecode.c:67-85; returns 0, or -1 on allocation trouble */
int  vdecode_parse_args(vdecode_opts *o, int argc, char **argv);
/* run the stage: read symbols from fd_in until EOF, write '0'/'1' to out, status to err.  0, -1 (engine / allocation
 * failure) or -2 (the output could not be written: closed pipe, full memory stream) */
int  vdecode_run(const vdecode_opts *o, const vdecode_engine *e, int fd_in, FILE *out, FILE *err,
                 vdecode_result *res);
#endif
