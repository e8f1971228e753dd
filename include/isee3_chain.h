/* isee3_chain.h -- the whole receive chain (pmdemod | symdemod | vdecode, reference README.txt:6-9) as ONE
 * call on buffers in memory: libisee3chain.so.  The three stages are exactly the C cores of the
 * stand-alone pipe stages (isee3-decoder_amd/cli/ *_core.c), run as threads inside the calling process, so an already
 * initialised HIP context is reused from call to call; the sample streams between the stages stay in device memory. */
#ifndef ISEE3_CHAIN_H
#define ISEE3_CHAIN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double samprate;        /* pmdemod -r / symdemod -r, default 250000 */
  double binsize;         /* pmdemod -b, default 4 */
  double search_freq;     /* pmdemod -S */
  double search_width;    /* pmdemod -W */
  int    flip;            /* pmdemod -f */
  const char *symrate;    /* symdemod -c argument as text ("1024" = nominal, scaled; "1024.5" = exact); NULL = default */
  int    decode_delay;    /* vdecode -d, default 200 */
  int    verbose;         /* keep the stages' stderr status lines */
} isee3_chain_opts;

void isee3_chain_default_opts(isee3_chain_opts *o);

/* iq: nsamples (I,Q) int16 pairs; out: ASCII '0'/'1', at most cap bytes; *nout = bytes produced.
 * Returns 0, or 2 when a stage failed (isee3_chain_last_error() says which). */
int isee3_chain_run_mem(const isee3_chain_opts *o, const int16_t *iq, size_t nsamples, char *out, size_t cap, size_t *nout);
/* the same with the capture ALREADY IN DEVICE MEMORY (d_iq: device pointer, e.g. from isee3dsp_dev_alloc): nothing but
 * soft symbols and decoded bits crosses to the host */
int isee3_chain_run_dev(const isee3_chain_opts *o, const int16_t *d_iq, size_t nsamples, char *out, size_t cap, size_t *nout);
/* wall-clock ms the calling thread's last run spent inside engine calls of the pmdemod, symdemod and vdecode stage
 * (they include waiting for the GPU; the stages run concurrently, so the three overlap) */
void isee3_chain_last_stage_ms(double ms[3]);
/* same on file descriptors: reads int16 IQ from fd_in until EOF, writes bits to fd_out */
int isee3_chain_run_fd(const isee3_chain_opts *o, int fd_in, int fd_out);
const char *isee3_chain_last_error(void);
/* The library keeps the Viterbi decoder objects of finished calls (2.2 GiB of HBM each, at most four) for the next
 * call; this frees them. */
void isee3_chain_release(void);

#ifdef __cplusplus
}
#endif
#endif
