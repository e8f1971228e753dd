/* viterbi224_hip.h -- extensions of libviterbi224_hip.so beyond the reference's nine-function
 * API (include/viterbi224.h).  Everything here is plain C-ABI: pointers, sizes, ints.
 *
 * Why they exist: the reference's own streaming caller runs ONE trellis step per call and a
 * 200-step traceback per decoded bit (vdecode.c:145,152).  A GPU wants the same work handed over
 * in blocks; these entry points do exactly what a loop of update_viterbi224_blk(p,s,1) +
 * decodebit_viterbi224(p,delay,0) would do, for a whole block, with identical results.
 */
#ifndef VITERBI224_HIP_H
#define VITERBI224_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "viterbi224.h"

#ifdef __cplusplus
extern "C" {
#endif

/* engines (option "engine"): */
#define V224HIP_ENGINE_SIMPLE 0  /* one trellis step per launch, decisions in the port's bit order */
#define V224HIP_ENGINE_FUSED  1  /* register-resident radix-2^k passes, permuted decision layout  */
#define V224HIP_ENGINE_LDS    2  /* LDS-staged two-level passes, 8 trellis steps per launch        */
#define V224HIP_ENGINE_LDS15  3  /* LDS-staged four-level passes, 15 steps per launch, metrics kept
                                    in a tile-major order between launches                        */

/* Number of HIP devices visible / select the device used by subsequent create calls of the PROCESS, whichever thread
 * makes them (default: device 0, or $V224HIP_DEVICE).  The setting is one atomic per process -- the design is one process
 * per GPU, and libisee3chain.so creates its decoders in worker threads that must see the host's choice.  Every other
 * entry point selects the device its handle was created on, so handles may be used from any thread (one thread at a
 * time per handle; different handles concurrently).  -1 on error. */
int v224hip_device_count(void);
int v224hip_set_device(int dev);

/* create_viterbi224 with explicit engine (-1 = default / $V224HIP_ENGINE) and fused stage count
 * (0 = default / $V224HIP_K).  NULL on failure; v224hip_last_error() says why. */
void *v224hip_create(int len, int engine, int k);
const char *v224hip_last_error(void);

/* update_viterbi224_blk with the 2*nbits symbols already resident in device memory. */
int v224hip_update_dev(void *p, const uint8_t *d_syms, int nbits);

/* Streaming block decode == for each of nbits: update(1 bit); out[i] = decodebit(delay, 0).
 * out[i] is 0/1, or 0xff while fewer than `delay` steps have run since init (vdecode.c:151-158
 * suppresses exactly those).  Needs len >= delay + v224hip_stream_chunk(p) (twice the chunk with V224HIP_TB_STREAM=1).
 * `syms`/`out` are host buffers in the first form, device buffers in the _dev form. */
int v224hip_stream_decode(void *p, const uint8_t *syms, int nbits, int delay, uint8_t *out);
int v224hip_stream_decode_dev(void *p, const uint8_t *d_syms, int nbits, int delay, uint8_t *d_out);
int v224hip_stream_chunk(void *p);                 /* bits per internal chunk (option "chunk") */

/* ONE stream decoded by several decoders at the same time, with the result of a single decoder -- verified.
 * The stream is cut into ndec consecutive parts (multiples of the stream chunk); decoder j > 0 starts `warm_bits`
 * before its part from a fresh init(0).  1020 bits (one chunk) before the part begins, its path metrics minus their
 * minimum are compared, for all 2^23 states, with those of decoder j-1 at the same trellis step.  If they are equal,
 * every later decision of the two is equal (the recursion port.c:168-181 only ever compares metrics), so decoder j's
 * output from its part on is what decoder j-1 would have gone on to produce.  If they are not, the rest of the stream
 * is decoded again by decoder j-1 from where it stands; *nfallback = number of parts redone (0 in practice with a
 * warm-up of a few thousand bits).  Device buffers; all decoders on one device, same chunk, len >= delay + 2*chunk.
 * out[] is exactly what init(0) + v224hip_stream_decode_dev() of one decoder writes. */
int v224hip_stream_decode_split(void *const *decoders, int ndec, const uint8_t *d_syms, int nbits, int delay,
                                uint8_t *d_out, int warm_bits, int *nfallback);

/* Streaming form of the same sharing, host buffers: the next block of a stream that decoders[*holder] is in the middle
 * of.  A block of at least 2.2 warm-ups is shared between the holder and one other decoder (which starts fresh inside
 * the block; seam verified as above, redone by the holder's side if it fails) and *holder moves to the decoder that
 * stands at the end of the block; a shorter block just continues on the holder.  Feeding a stream block by block
 * through this call gives exactly the output of v224hip_stream_decode() on one decoder.  Start with init_viterbi224 on
 * decoders[0] and *holder = 0. */
int v224hip_stream_decode_shared(void *const *decoders, int ndec, int *holder, const uint8_t *syms, int nbits,
                                 int delay, uint8_t *out, int warm_bits);

/* The same for a stream that is still ARRIVING (host buffers), with ONE warm-up for the whole stream.  begin() takes the
 * expected length; feed() appends the symbols of the next nbits trellis steps and enqueues whatever can run now.  feed() never
 * waits for DECODE work; it does wait for its own upload: a blocking hipMemcpy from the caller's (pageable) buffer, which
 * HIP orders on the legacy null stream -- behind anything the host process has queued there (in libisee3chain.so:
 * symdemod's short kernels, isee3dsp_share_stream(2)).  That is deliberate: an asynchronous copy would have to ride on the
 * decoders' own streams (a bubble in a 17 us launch chain per feed, and decoder 1 waiting on an event behind decoder 0's
 * backlog) or on a fifth busy hardware queue ("One stream per decoder"); decoder 0 decodes from the first symbol on, decoder 1 joins from a fresh init `warm_bits` before
 * the cut and runs to the end.  The cut is placed when decoder 1 can start: x = (expected + warm + what decoder 0 has
 * finished by then) / 2, so that both finish together -- the middle when all symbols are there at once, later when they
 * trickle in, never when one decoder keeps up with them.  end() finishes, verifies the seam as above (decoder 0 decodes
 * the second part again if the check fails: *redone = 1), writes all bits -- exactly what one decoder's
 * v224hip_stream_decode() of the whole stream writes -- and releases the handle (also on failure).  A wrong expectation
 * only misplaces the cut.  warm_bits is raised to the seam window (delay rounded up to chunks) + one chunk.  Both decoders:
 * same device and chunk, len >= delay + chunk; they are (re-)initialised by these
 * calls.  NULL / -1 on failure (v224hip_last_error()); abort() drops a handle without finishing. */
void *v224hip_progressive_begin(void *const *decoders, int ndec, long long expected_bits, int delay, int warm_bits);
int   v224hip_progressive_feed(void *h, const uint8_t *syms, int nbits);
int   v224hip_progressive_end(void *h, uint8_t *out, long long cap, long long *nbits_out, int *redone);
void  v224hip_progressive_abort(void *h);

/* A batch of independent frames, each decoded as vtest224.c:116-118 / decode.c:220-222 do it:
 *   init_viterbi224(d, startstate); update_viterbi224_blk(d, syms + f*2*framebits, framebits);
 *   chainback_viterbi224(d, out + f*((framebits+7)/8), framebits, endstate);
 * Frame f runs on decoders[f % ndec] (each created with len >= framebits, all on one device); with ndec = 2 the
 * frames overlap on the GPU.  Host buffers.  0, or -1 (v224hip_last_error()). */
int v224hip_decode_frames(void *const *decoders, int ndec, const uint8_t *syms, int nframes, int framebits,
                          int startstate, unsigned int endstate, uint8_t *out);

/* Generic option setter: "chunk" (bits per stream chunk), "profile" (N > 0: bracket every Nth run of back-to-back
 * ACS launches -- one stream chunk or one update call -- with a HIP event pair on the decoder's
 * stream).  -1 on unknown key / bad value. */
int v224hip_set_option(void *p, const char *key, long value);

/* Counters, read and reset.  "chainback_redone": chainback_viterbi224 / v224hip_decode_frames walk a frame of 512 ..
 * 81 920 bits in 16 pieces at once, each piece checked against the one above it; this counts the pieces that failed the
 * check and were walked again from the true state (the output is the serial walk's either way).  Plain reads (inspection):
 * "dp" = ring index of the next decision row to be written (port.c:24 `dp`), "steps" = trellis steps since init.
 * -1 on unknown key. */
long v224hip_get_counter(void *p, const char *key);

/* Block until all enqueued work of this decoder has finished. */
int v224hip_sync(void *p);

/* ACS launch statistics since the last reset (needs option "profile" > 0): number of ACS launches
 * inside the timed runs, the summed device time of those runs in ms (hipEventElapsedTime), and the
 * trellis steps they covered. */
int v224hip_acs_stats(void *p, unsigned long long *launches, double *total_ms,
                      unsigned long long *steps, int reset);

/* Test/inspection: decision row `row` of the ring converted to the port's layout (bit s of the
 * 1 MiB row = decision for new state s, port.c:13,183), and the current path metrics minus their
 * minimum as uint32[2^23].  Host buffers. */
int v224hip_export_row(void *p, int row, uint8_t *out_1MiB);
int v224hip_export_metrics(void *p, uint32_t *out_8M);

/* Device memory helpers so that a C (non-HIP) caller can keep buffers resident in HBM. */
void *v224hip_dev_alloc(size_t bytes);
void  v224hip_dev_free(void *d);
int   v224hip_h2d(void *d_dst, const void *h_src, size_t bytes);
int   v224hip_d2h(void *h_dst, const void *d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
