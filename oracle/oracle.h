/* oracle.h -- CPU restatement of the ISEE-3 receive chain hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or
 * executed by the product path (isee3-decoder_amd/, include/, the CLI stages):
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker.
 *
 * Pinning: the reference ships no golden vectors (its harnesses are seeded from
 * time(): vtest224.c:57-58).  The Viterbi, vdecode and symdemod restatements are
 * pinned against outputs of the reference itself, compiled from
 * /root/reference by oracle/Makefile into oracle/_ref/ and captured as
 * tests/golden/ fixtures by tests/golden/make_golden.py.
 * pmdemod: PARITY UNPINNED at the FFTW3 boundary (fftw3 absent from the image,
 * reference Makefile:66) -- the restatement is cross-checked against numpy's
 * FFT only.
 */
#ifndef ISEE3_ORACLE_H
#define ISEE3_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- code constants: code.h:54-63 (MCQLI24) ---- */
#define ORC_K        24
#define ORC_POLY1    073665667u
#define ORC_POLY2    073665665u
#define ORC_G1FLIP   0
#define ORC_G2FLIP   1
#define ORC_NSTATES  (1u << (ORC_K - 1))      /* 2^23 */
#define ORC_ROWBYTES (ORC_NSTATES / 8)        /* 1 MiB, viterbi224_port.c:13 */

/* ---- Viterbi, port semantics (viterbi224_port.c) ---- */
enum { ORC_V224_LITERAL = 0,   /* u32 metrics, 64-bit compare: line-for-line semantics */
       ORC_V224_FAST    = 1 }; /* u16 modular metrics, SSE2 + OpenMP; same decisions while
                                  no u32 metric has wrapped (>= 8.4e6 bits after init) */

void *orc_v224_create(int len, int mode);                       /* port.c:51-68  */
int   orc_v224_init(void *p, int starting_state);               /* port.c:34-48  */
int   orc_v224_update(void *p, const uint8_t *syms, int nbits); /* port.c:159-195 */
int   orc_v224_chainback(void *p, uint8_t *data, unsigned nbits, unsigned endstate); /* port.c:72-101 */
int   orc_v224_decodebit(void *p, int delay, int endstate);     /* port.c:104-143 */
unsigned long long orc_v224_decodeword(void *p, int delay, int endstate);   /* sse2.c:206-243, on the port's decisions */
void  orc_v224_delete(void *p);                                 /* port.c:146-153 */

/* test introspection */
const uint8_t *orc_v224_row(void *p, int row);          /* 1 MiB decision row, port bit order */
int      orc_v224_dp(void *p);                          /* index of next row to be written */
uint32_t orc_v224_metric_rel(void *p, uint32_t state);  /* metric[state] - min(metric) */
uint32_t orc_v224_spread(void *p);                      /* max - min of current metrics */
uint32_t orc_v224_metric_abs(void *p, int want_max);   /* LITERAL engine only */
int      orc_v224_set_metrics(void *p, const uint32_t *m_8M);   /* resume from given path metrics (any common offset); dp = 0 */
int      orc_v224_get_metrics(void *p, uint32_t *out_8M);       /* current metrics minus their minimum */
uint64_t orc_fnv1a(const void *buf, size_t n);          /* 64-bit FNV-1a, for row hashes */

/* ---- encoder (encode.c:17-35) ---- */
uint64_t orc_encode(uint8_t *symbols, const uint8_t *data, unsigned nbytes, uint64_t encstate);

/* ---- deterministic generators (this repo's own; reference has none that are seeded) ---- */
uint64_t orc_splitmix64(uint64_t *state);
void orc_gen_uniform_bytes(uint64_t seed, uint8_t *out, size_t n);
/* continuous (untailed) coded stream through an 8-bit AWGN channel.
 * noise sigma follows vtest224.c:93-95; noise_blocks_pct % of 1024-symbol blocks are pure noise. */
void orc_gen_coded_stream(uint64_t seed, size_t nbits, double ebn0_db, double amplitude,
                          int noise_blocks_pct, uint8_t *syms /*2*nbits*/, uint8_t *bits /*nbits, 0/1*/);
/* tailed frame as vtest224.c:100-112: framebits-24 random bits + 24 zeros, start state 0 */
void orc_gen_coded_frame(uint64_t seed, int framebits, double ebn0_db, double amplitude,
                         uint8_t *syms /*2*framebits*/, uint8_t *data /*framebits/8*/);
/* PM-modulated int16 IQ capture: returns number of samples written */
size_t orc_gen_iq(uint64_t seed, double samprate, double seconds, double fc_hz, double beta,
                  double symrate, double amp, double cn0_dbhz, int16_t *iq /*2 per sample*/,
                  uint8_t *sentbits, size_t sentbits_cap, size_t *nsent);
/* Manchester baseband int16 (symdemod input) without the PM/carrier stage */
size_t orc_gen_baseband(uint64_t seed, double samprate, double seconds, double symrate,
                        double amp, double noise_sigma, int16_t *out,
                        uint8_t *sentbits, size_t sentbits_cap, size_t *nsent);

/* ---- vdecode stage (vdecode.c:38-189) as a buffer-to-buffer function ---- */
typedef struct {
  uint64_t bits;        /* decoded bits emitted */
  uint64_t symerrs;     /* total re-encode symbol errors (vdecode.c:174-177), never reset here */
  int      flips;       /* phase flips performed */
} orc_vdecode_stats;
size_t orc_vdecode(const uint8_t *syms, size_t nsyms, int decode_delay, int start_phase,
                   int dontflip, int mode, char *out /* cap >= nsyms/2+1 */, orc_vdecode_stats *st);

/* ---- symdemod stage (symdemod.c) ---- */
typedef struct {
  double samprate_unused;
  int    samprate;       /* -r */
  double symrate;        /* after -c scaling */
  int    symbolclocks;   /* -C */
  double window;         /* -w */
  int    clocktrack;     /* -t */
} orc_symdemod_cfg;
void   orc_symdemod_default(orc_symdemod_cfg *c);                          /* symdemod.c:51-55 */
void   orc_symdemod_set_c(orc_symdemod_cfg *c, const char *optarg);        /* symdemod.c:67-77 */
double orc_timesearch(int *symphase, const int16_t *samples, int firstsample,
                      double symbolsamples_param, double symbolsamples_global,
                      int symbolclocks, int nsymbols);                      /* symdemod.c:260-335 */
double orc_trial_demod(const int16_t *samples, int firstsample, double symbolsamples,
                       int symbolclocks, int nsymbols, double gain, uint8_t *out); /* symdemod.c:202-256 */
/* whole stage; per-window symphase / energy log optional */
size_t orc_symdemod(const orc_symdemod_cfg *c, const int16_t *in, size_t nin,
                    uint8_t *out, size_t outcap,
                    int *symphase_log, double *energy_log, int logcap, int *nwindows);

/* ---- pmdemod stage (pmdemod.c) ---- */
typedef struct {
  double samprate;        /* -r, default 250000 */
  double binsize;         /* -b, default 4 (pmdemod.c:81) */
  double search_freq;     /* -S */
  double search_width;    /* -W */
  double doppler_rate;    /* -D */
  double cn0_threshold;   /* -t, default 21 */
  int    flip;            /* -f */
} orc_pmdemod_cfg;
typedef struct {
  int    peak;
  double carrier_freq;
  double cn0;
  double amplitude;
} orc_pmdemod_blk;
void   orc_pmdemod_default(orc_pmdemod_cfg *c);
int    orc_pmdemod_fftsize(const orc_pmdemod_cfg *c);                       /* pmdemod.c:129-131 */
/* processes floor(nsamp/N) blocks; out int16 (N per block); pre = optional pre-quantisation
 * doubles (imag*sqrt(1/2), N per block) for the 1e-9 check; blk = per-block report */
size_t orc_pmdemod(const orc_pmdemod_cfg *c, const int16_t *iq, size_t nsamp,
                   int16_t *out, double *pre, orc_pmdemod_blk *blk, int blkcap, int *nblk);
/* forward unnormalised DFT, out-of-place, N power of two (FFTW_FORWARD semantics, pmdemod.c:161) */
void   orc_fft_forward(const double *in_ri, double *out_ri, int n);

#ifdef __cplusplus
}
#endif
/* ---- icesync.c:55-208 FFT sync-vector correlator (icesync_oracle.c; PARITY UNPINNED: FFTW3 absent) ---- */
#define ORC_SYNC_FAIL (-1234567890)                          /* icesync.c:31 */
int orc_icesync_sync_vector(double symbolsamples, double *vec, int cap);
int orc_icesync_search(const double *vec, int synclen, int corr_size, const int16_t *samples, double framesamples,
                       int low, int high, double *maxpeak, double *result);

#endif
