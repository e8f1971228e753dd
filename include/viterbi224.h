/* viterbi224.h -- the K=24 r=1/2 Viterbi decoder API of the ISEE-3 receive chain,
 * implemented by libviterbi224_hip.so (hand-written HIP for MI355X / gfx950).
 *
 * Drop-in boundary.  These nine entry points are exactly the ones the reference declares in
 * /root/reference/viterbi224.h:8-16 and selects at LINK time by swapping viterbi224_port.o /
 * viterbi224_sse2.o (reference Makefile:27-32,43-48).  Linking a caller (vdecode.c, decode.c,
 * vtest224.c, hybridtest.c, bitsync.c, icesync.c) against -lviterbi224_hip instead of one of
 * those objects is the whole integration; see INTEGRATION.md.
 *
 * Semantics are those of viterbi224_port.c (the parity target):
 *   - survivor ties go to the predecessor with MSB set     (port.c:178-179, ">= 0")
 *   - init: every metric 1000, the start state 0           (port.c:40-46)
 *   - decisions are kept for `len` trellis steps in a ring (port.c:58,187-188)
 * Path metrics and decisions live in HBM; `syms` and `data` are caller (host) buffers borrowed
 * for the duration of the call.  Calls that return data (chainback, decodebit, decodeword,
 * min/max_metric) synchronise with the device; update only enqueues work.
 *
 * Exactness bound: decisions equal the port's for any stream of up to 8 421 502 bits after
 * init (no u32 metric of the port can have wrapped by then: (2^32-1000)/510).  Past that the
 * port itself misorders wrapped and unwrapped metrics for a few steps; this library keeps
 * the mathematically consistent (modular) ordering.
 */
#ifndef VITERBI224_H
#define VITERBI224_H

#ifdef __cplusplus
extern "C" {
#endif

/* viterbi224.h:8  -- (re)start a frame: all metrics 1000, metric[starting_state & 0x7fffff] = 0,
 *                    decision ring rewound.  -1 if p == NULL, else 0.            (port.c:34-48)  */
int init_viterbi224(void *p, int starting_state);

/* viterbi224.h:9  -- new decoder with `len` rows (1 MiB of HBM each) of decision history;
 *                    NULL on failure (no device, out of memory).                 (port.c:51-68)  */
void *create_viterbi224(int len);

/* viterbi224.h:10 -- trace back `nbits` steps from `endstate`, rows taken as n % len, write
 *                    nbits/8 bytes MSB-first into data.  -1 if p == NULL.        (port.c:72-101) */
int chainback_viterbi224(void *p, unsigned char *data, unsigned int nbits, unsigned int endstate);

/* viterbi224.h:11                                                                 (port.c:146-153) */
void delete_viterbi224(void *p);

/* viterbi224.h:12 -- run `nbits` trellis steps on 2*nbits offset-128 soft symbols (255 = strong 1).
 *                    Port return convention: -1 if p == NULL, else 0.            (port.c:159-195) */
int update_viterbi224_blk(void *p, const unsigned char *syms, int nbits);

/* viterbi224.h:13-14 -- largest / smallest current path metric, in the port's (never
 *                    renormalised) scale.  Only viterbi224_sse2.c implements them in the
 *                    reference (sse2.c:82-109); sole caller icesync.c:370-371.                   */
int max_metric_viterbi224(void *p);
int min_metric_viterbi224(void *p);

/* viterbi224.h:15 -- walk `delay` rows back from the newest and return the last decision bit
 *                    read; endstate < 0 = start from the best state (first minimum).
 *                    -1 if p == NULL or delay <= 0.                              (port.c:104-143) */
int decodebit_viterbi224(void *p, int delay, int endstate);

/* viterbi224.h:16 -- as decodebit but returns the `delay` most recent decisions walked, newest
 *                    first read ending in bit 63 (sse2.c:206-243).                               */
unsigned long long decodeword_viterbi224(void *p, int delay, int endstate);

#ifdef __cplusplus
}
#endif
#endif /* VITERBI224_H */
