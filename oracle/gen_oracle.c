/* gen_oracle.c -- deterministic synthetic inputs (TEST INFRASTRUCTURE ONLY, see oracle.h).
 *
 * The reference has no seeded generator (vtest224.c:57-58 seeds from time(), gensine.c emits an
 * unmodulated tone), so these are this repo's own.  What they imitate:
 *   - coded frames through an 8-bit AWGN channel: vtest224.c:93-112 (Gain 24, noise from Eb/N0)
 *   - sync word / frame length: framer.c:12,18 (1024-bit frames ending in 0x12fc819fbe)
 *   - Manchester sense: symdemod.c:227-235 (first half subtracted, second half added)
 *   - PM modulation index 1.1 rad: pmdemod.c:83
 * Noise is integer Irwin-Hall (no libm) so the 8-bit symbol and baseband streams are bit-identical on
 * every IEEE machine; only orc_gen_iq goes through libm sin/cos (its fixtures commit their inputs).
 */
#include <math.h>
#include <string.h>
#include "oracle.h"

uint64_t orc_splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

static inline double u01(uint64_t *s) {           /* (0,1] */
  return ((orc_splitmix64(s) >> 11) + 1) * (1.0 / 9007199254740992.0);
}

/* Gaussian-ish deviate without libm: Irwin-Hall sum of sixteen 16-bit uniforms (integer exact),
   scaled to unit variance by one IEEE multiply.  Tails stop at +-6.9 sigma, ample for test signals,
   and the stream is bit-identical on every IEEE machine. */
typedef struct { uint64_t s; int have; double spare; } gauss_t;
static double gauss(gauss_t *g) {
  int64_t sum = 0;
  for (int k = 0; k < 4; k++) {
    uint64_t z = orc_splitmix64(&g->s);
    sum += (int64_t)(z & 0xffff) + (int64_t)((z >> 16) & 0xffff)
         + (int64_t)((z >> 32) & 0xffff) + (int64_t)(z >> 48);
  }
  /* mean 16*32767.5 = 524280; variance 16*(65536^2-1)/12 => sigma = 75674.4545 */
  return (double)(2 * sum - 1048560) * (0.5 / 75674.45447441297);
}

void orc_gen_uniform_bytes(uint64_t seed, uint8_t *out, size_t n) {
  uint64_t s = seed;
  for (size_t i = 0; i < n; i += 8) {
    uint64_t z = orc_splitmix64(&s);
    for (size_t k = 0; k < 8 && i + k < n; k++) out[i + k] = (uint8_t)(z >> (8 * k));
  }
}

static inline uint8_t chan8(int sym, double amp, double sigma, gauss_t *g) {
  double x = 128.0 + amp * (2 * sym - 1) + sigma * gauss(g);
  long q = lrint(x);
  return (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
}

static double sigma_from_ebn0(double ebn0_db, double amp) {
  double esn0 = ebn0_db + 10 * log10(0.5);            /* vtest224.c:93 */
  return amp * M_SQRT1_2 / pow(10., 0.05 * esn0);     /* vtest224.c:95 */
}

void orc_gen_coded_stream(uint64_t seed, size_t nbits, double ebn0_db, double amplitude,
                          int noise_blocks_pct, uint8_t *syms, uint8_t *bits) {
  uint64_t sd = seed, sb = seed ^ 0xa5a5a5a5deadbeefull;
  gauss_t g = { seed * 0x2545f4914f6cdd1dull + 7, 0, 0 };
  double sigma = sigma_from_ebn0(ebn0_db, amplitude);
  uint64_t enc = 0, word = 0;
  int noisy = 0;
  for (size_t n = 0; n < nbits; n++) {
    if ((n & 63) == 0) word = orc_splitmix64(&sd);
    if ((n & 511) == 0)                                /* 1024-symbol block boundary */
      noisy = noise_blocks_pct > 0 && (int)(orc_splitmix64(&sb) % 100) < noise_blocks_pct;
    unsigned b = (word >> (n & 63)) & 1;
    if (bits) bits[n] = (uint8_t)b;
    enc = (enc << 1) | b;
    int y0 = ORC_G1FLIP ^ __builtin_parityll(enc & ORC_POLY1);
    int y1 = ORC_G2FLIP ^ __builtin_parityll(enc & ORC_POLY2);
    if (noisy) {
      syms[2 * n]     = chan8(0, 0.0, sigma, &g);
      syms[2 * n + 1] = chan8(0, 0.0, sigma, &g);
    } else {
      syms[2 * n]     = chan8(y0, amplitude, sigma, &g);
      syms[2 * n + 1] = chan8(y1, amplitude, sigma, &g);
    }
  }
}

void orc_gen_coded_frame(uint64_t seed, int framebits, double ebn0_db, double amplitude,
                         uint8_t *syms, uint8_t *data) {
  uint64_t s = seed;
  gauss_t g = { seed * 0x2545f4914f6cdd1dull + 11, 0, 0 };
  double sigma = sigma_from_ebn0(ebn0_db, amplitude);
  int nbytes = framebits / 8, tail = (framebits - ORC_K) / 8;
  orc_gen_uniform_bytes(s, data, (size_t)tail);
  memset(data + tail, 0, (size_t)(nbytes - tail));
  orc_encode(syms, data, (unsigned)nbytes, 0);
  for (int i = 0; i < 2 * framebits; i++) syms[i] = chan8(syms[i], amplitude, sigma, &g);
}

/* ---- telemetry bit source: 1024-bit frames, last 40 bits = sync word (framer.c:18,69) ---- */
typedef struct { uint64_t s, word; unsigned pos; } tlm_t;
static unsigned tlm_bit(tlm_t *t) {
  const uint64_t SYNC = 0x12fc819fbeull;
  unsigned inframe = t->pos & 1023, b;
  if (inframe >= 1024 - 40) b = (SYNC >> (1023 - inframe)) & 1;
  else {
    if ((inframe & 63) == 0) t->word = orc_splitmix64(&t->s);
    b = (t->word >> (inframe & 63)) & 1;
  }
  t->pos++;
  return b;
}

/* symbol source: continuous r=1/2 encoding of the telemetry bits */
typedef struct { tlm_t t; uint64_t enc; int half; unsigned y[2];
                 uint8_t *sent; size_t cap, n; } symsrc_t;
static unsigned next_symbol(symsrc_t *q) {
  if (q->half == 0) {
    unsigned b = tlm_bit(&q->t);
    if (q->sent && q->n < q->cap) q->sent[q->n] = (uint8_t)b;
    q->n++;
    q->enc = (q->enc << 1) | b;
    q->y[0] = ORC_G1FLIP ^ __builtin_parityll(q->enc & ORC_POLY1);
    q->y[1] = ORC_G2FLIP ^ __builtin_parityll(q->enc & ORC_POLY2);
  }
  unsigned y = q->y[q->half];
  q->half ^= 1;
  return y;
}

static inline int16_t sat16(double x) {
  long q = lrint(x);
  return (int16_t)(q < -32767 ? -32767 : q > 32767 ? 32767 : q);
}

size_t orc_gen_baseband(uint64_t seed, double samprate, double seconds, double symrate,
                        double amp, double noise_sigma, int16_t *out,
                        uint8_t *sentbits, size_t cap, size_t *nsent) {
  symsrc_t q = { { seed, 0, 0 }, 0, 0, { 0, 0 }, sentbits, cap, 0 };
  gauss_t g = { seed * 0x2545f4914f6cdd1dull + 13, 0, 0 };
  size_t n = (size_t)(samprate * seconds);
  double ph = 1.0, step = symrate / samprate;   /* ph >= 1 forces a fresh symbol first */
  unsigned y = 0;
  for (size_t i = 0; i < n; i++) {
    if (ph >= 1.0) { ph -= 1.0; y = next_symbol(&q); }
    double m = ((ph < 0.5) ? -1.0 : 1.0) * (y ? 1.0 : -1.0);   /* symdemod.c:227-235 */
    out[i] = sat16(amp * m + noise_sigma * gauss(&g));
    ph += step;
  }
  if (nsent) *nsent = q.n;
  return n;
}

size_t orc_gen_iq(uint64_t seed, double samprate, double seconds, double fc_hz, double beta,
                  double symrate, double amp, double cn0_dbhz, int16_t *iq,
                  uint8_t *sentbits, size_t cap, size_t *nsent) {
  symsrc_t q = { { seed, 0, 0 }, 0, 0, { 0, 0 }, sentbits, cap, 0 };
  gauss_t g = { seed * 0x2545f4914f6cdd1dull + 17, 0, 0 };
  size_t n = (size_t)(samprate * seconds);
  /* pmdemod.c:351: cn0 = 10log10(fs*A^2/(2*var_real))  =>  var_real = fs*A^2/(2*cn0) */
  double sigma = amp * sqrt(samprate / (2.0 * pow(10., cn0_dbhz / 10.)));
  uint64_t ps = seed ^ 0x0123456789abcdefull;
  double phi0 = 2.0 * M_PI * u01(&ps);
  double ph = 1.0, step = symrate / samprate;
  double w = 2.0 * M_PI * fc_hz / samprate;
  unsigned y = 0;
  for (size_t i = 0; i < n; i++) {
    if (ph >= 1.0) { ph -= 1.0; y = next_symbol(&q); }
    double m = ((ph < 0.5) ? -1.0 : 1.0) * (y ? 1.0 : -1.0);
    double th = fmod(w * (double)i, 2.0 * M_PI) + phi0 + beta * m;
    iq[2 * i]     = sat16(amp * cos(th) + sigma * gauss(&g));
    iq[2 * i + 1] = sat16(amp * sin(th) + sigma * gauss(&g));
    ph += step;
  }
  if (nsent) *nsent = q.n;
  return n;
}
