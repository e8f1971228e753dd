/* v224_oracle.c -- CPU restatement of the K=24 r=1/2 Viterbi decoder, PORT semantics.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity target: viterbi224_port.c.
 *
 * Two interchangeable engines behind one handle:
 *   ORC_V224_LITERAL  u32 path metrics, never renormalised, survivor test done in
 *                     64-bit signed arithmetic exactly as port.c:171-181 does on an
 *                     LP64 host ("unsigned long" temporaries, "(signed long)(m0-m1) >= 0").
 *   ORC_V224_FAST     u16 modular metrics; survivor test = sign of the 16-bit
 *                     difference.  Identical decisions as long as (a) the metric spread
 *                     stays < 2^15 (it is < 12 731 + 510: 23 steps reach any state) and
 *                     (b) no u32 metric of the literal engine has wrapped, which cannot
 *                     happen before (2^32-1000)/510 = 8 421 502 bits after init.
 *
 * Trellis (port.c:168-181): butterfly i in [0,2^22) reads old[i], old[i+2^22] and writes
 * new[2i], new[2i+1]; decision bit s of a row = 1 when the survivor into NEW state s came
 * from the predecessor with MSB set; ties go to that predecessor (">= 0").
 *
 * Branch metric (port.c:62-65,170): BT0[i] = parity(2i & POLY1) ? 255 : 0 (G1FLIP = 0),
 * BT1[i] = !parity(2i & POLY2) ? 255 : 0 (G2FLIP = 1); bm = (BT0^s0) + (BT1^s1).
 * Because POLY1 ^ POLY2 == 2, parity(2i&POLY2) = parity(2i&POLY1) ^ (i&1), so with
 * p = parity(2i & POLY1):
 *      i even: bm = p ? 510 - c0 : c0,   c0 = s0 + 255 - s1
 *      i odd : bm = p ? 510 - c1 : c1,   c1 = s0 + s1
 * No 2x4M-entry table is needed.
 */
#include <stdlib.h>
#include <string.h>
#include <emmintrin.h>
#include "oracle.h"

#define NST   ORC_NSTATES
#define NBF   (NST / 2)          /* butterflies per bit */
#define MASK  (NST - 1)

typedef struct {
  int mode, len;
  int dp;                         /* next row to write */
  uint32_t *m32[2];               /* literal engine */
  uint16_t *m16[2];               /* fast engine */
  int cur;                        /* index of "old" buffer */
  uint8_t *rows;                  /* len * 1 MiB */
} orc_v224;

static inline int par32(uint32_t x) { return __builtin_parity(x); }

void *orc_v224_create(int len, int mode) {
  if (len <= 0) return NULL;
  orc_v224 *v = calloc(1, sizeof *v);
  if (!v) return NULL;
  v->mode = mode; v->len = len;
  v->rows = calloc((size_t)len, ORC_ROWBYTES);   /* the reference mallocs (port.c:58): rows never written are undefined there, zero here and in the product */
  int ok = v->rows != NULL;
  for (int b = 0; b < 2 && ok; b++) {
    if (mode == ORC_V224_LITERAL) ok = (v->m32[b] = malloc(sizeof(uint32_t) * NST)) != NULL;
    else ok = posix_memalign((void **)&v->m16[b], 64, sizeof(uint16_t) * NST) == 0;
  }
  if (!ok) { orc_v224_delete(v); return NULL; }
  orc_v224_init(v, 0);
  return v;
}

void orc_v224_delete(void *p) {
  orc_v224 *v = p;
  if (!v) return;
  for (int b = 0; b < 2; b++) { free(v->m32[b]); free(v->m16[b]); }
  free(v->rows);
  free(v);
}

int orc_v224_init(void *p, int starting_state) {
  orc_v224 *v = p;
  if (!v) return -1;
  v->cur = 0; v->dp = 0;
  uint32_t s = (uint32_t)starting_state & MASK;
  if (v->mode == ORC_V224_LITERAL) {
    for (uint32_t i = 0; i < NST; i++) v->m32[0][i] = 1000;
    v->m32[0][s] = 0;
  } else {
    for (uint32_t i = 0; i < NST; i++) v->m16[0][i] = 1000;
    v->m16[0][s] = 0;
  }
  return 0;
}

/* one trellis step, literal engine */
static void step_literal(orc_v224 *v, unsigned s0, unsigned s1, uint8_t *row) {
  const uint32_t *old = v->m32[v->cur];
  uint32_t *nw = v->m32[v->cur ^ 1];
  const uint64_t c[2] = { (uint64_t)s0 + 255 - s1, (uint64_t)s0 + s1 };
  memset(row, 0, ORC_ROWBYTES);
  for (uint32_t i = 0; i < NBF; i++) {
    uint64_t bm = c[i & 1];
    if (par32((2 * i) & ORC_POLY1)) bm = 510 - bm;
    uint64_t a0 = (uint64_t)old[i] + bm;               /* into 2i   from MSB-clear pred */
    uint64_t a1 = (uint64_t)old[i + NBF] + (510 - bm); /* into 2i   from MSB-set pred   */
    uint64_t b0 = (uint64_t)old[i] + (510 - bm);       /* into 2i+1 from MSB-clear pred */
    uint64_t b1 = (uint64_t)old[i + NBF] + bm;         /* into 2i+1 from MSB-set pred   */
    unsigned d0 = (int64_t)(a0 - a1) >= 0;
    unsigned d1 = (int64_t)(b0 - b1) >= 0;
    nw[2 * i]     = (uint32_t)(d0 ? a1 : a0);
    nw[2 * i + 1] = (uint32_t)(d1 ? b1 : b0);
    row[i >> 2] |= (uint8_t)((d0 | (d1 << 1)) << ((2 * i) & 7));
  }
  v->cur ^= 1;
}

/* one trellis step, fast engine: 8 butterflies per iteration, modular u16 */
static void step_fast(orc_v224 *v, unsigned s0, unsigned s1, uint8_t *row) {
  const uint16_t *old = v->m16[v->cur];
  uint16_t *nw = v->m16[v->cur ^ 1];
  const uint16_t c0 = (uint16_t)(s0 + 255 - s1), c1 = (uint16_t)(s0 + s1);
  /* lane n of an aligned group of 8 butterflies: c alternates with n&1, and the group's
     parity is p(base) ^ p(n) because base and n occupy disjoint bits. */
  const __m128i cv  = _mm_set_epi16(c1, c0, c1, c0, c1, c0, c1, c0);
  const __m128i cvn = _mm_sub_epi16(_mm_set1_epi16(510), cv);
  uint16_t pn[8];
  for (int n = 0; n < 8; n++) pn[n] = par32((2u * n) & ORC_POLY1) ? 0xffff : 0;
  const __m128i pnv = _mm_loadu_si128((const __m128i *)pn);
  const __m128i k510 = _mm_set1_epi16(510);
  uint16_t *row16 = (uint16_t *)row;

#pragma omp parallel for schedule(static)
  for (uint32_t g = 0; g < NBF / 8; g++) {
    uint32_t i = g * 8;
    __m128i pm = _mm_xor_si128(pnv, _mm_set1_epi16(par32((2 * i) & ORC_POLY1) ? -1 : 0));
    __m128i bm  = _mm_or_si128(_mm_andnot_si128(pm, cv), _mm_and_si128(pm, cvn));
    __m128i bmn = _mm_sub_epi16(k510, bm);
    __m128i oi = _mm_load_si128((const __m128i *)(old + i));
    __m128i oj = _mm_load_si128((const __m128i *)(old + i + NBF));
    __m128i a0 = _mm_add_epi16(oi, bm),  a1 = _mm_add_epi16(oj, bmn);
    __m128i b0 = _mm_add_epi16(oi, bmn), b1 = _mm_add_epi16(oj, bm);
    /* pick0 = 0xffff where (int16)(a0-a1) < 0, i.e. the MSB-clear predecessor survives */
    __m128i pickA = _mm_cmplt_epi16(_mm_sub_epi16(a0, a1), _mm_setzero_si128());
    __m128i pickB = _mm_cmplt_epi16(_mm_sub_epi16(b0, b1), _mm_setzero_si128());
    __m128i na = _mm_or_si128(_mm_and_si128(pickA, a0), _mm_andnot_si128(pickA, a1));
    __m128i nb = _mm_or_si128(_mm_and_si128(pickB, b0), _mm_andnot_si128(pickB, b1));
    _mm_store_si128((__m128i *)(nw + 2 * i),     _mm_unpacklo_epi16(na, nb));
    _mm_store_si128((__m128i *)(nw + 2 * i + 8), _mm_unpackhi_epi16(na, nb));
    /* decisions: interleave (d0,d1) per butterfly -> 16 bits for new states 2i..2i+15 */
    __m128i dA = _mm_packs_epi16(pickA, _mm_setzero_si128());   /* bytes: 0xff where pred i */
    __m128i dB = _mm_packs_epi16(pickB, _mm_setzero_si128());
    unsigned notd = (unsigned)_mm_movemask_epi8(_mm_unpacklo_epi8(dA, dB));
    row16[g] = (uint16_t)~notd;
  }
  v->cur ^= 1;
}

int orc_v224_update(void *p, const uint8_t *syms, int nbits) {
  orc_v224 *v = p;
  if (!v) return -1;
  while (nbits-- > 0) {
    uint8_t *row = v->rows + (size_t)v->dp * ORC_ROWBYTES;
    if (v->mode == ORC_V224_LITERAL) step_literal(v, syms[0], syms[1], row);
    else step_fast(v, syms[0], syms[1], row);
    syms += 2;
    if (++v->dp >= v->len) v->dp = 0;
  }
  return 0;
}

static inline unsigned rowbit(const orc_v224 *v, int row, uint32_t state) {
  return (v->rows[(size_t)row * ORC_ROWBYTES + (state >> 3)] >> (state & 7)) & 1;
}

int orc_v224_chainback(void *p, uint8_t *data, unsigned nbits, unsigned endstate) {
  orc_v224 *v = p;
  if (!v) return -1;
  uint32_t st = endstate & MASK;
  unsigned acc = 0;
  /* port.c:86-98: row index is (n mod len), i.e. the block is assumed to start at row 0 */
  for (unsigned n = nbits; n-- > 0;) {
    acc = ((st & 1) << 7) | (acc >> 1);
    if ((n & 7) == 0) data[n >> 3] = (uint8_t)acc;
    unsigned b = rowbit(v, (int)(n % (unsigned)v->len), st);
    st = (b << (ORC_K - 2)) | (st >> 1);
  }
  return 0;
}

int orc_v224_decodebit(void *p, int delay, int endstate) {
  orc_v224 *v = p;
  if (!v) return -1;
  uint32_t st;
  if (endstate < 0) {           /* port.c:113-122: strict <, first minimum wins */
    st = 0;
    if (v->mode == ORC_V224_LITERAL) {
      const uint32_t *m = v->m32[v->cur];
      uint32_t best = m[0];
      for (uint32_t i = 1; i < NST; i++) if (m[i] < best) { best = m[i]; st = i; }
    } else {
      const uint16_t *m = v->m16[v->cur];
      uint16_t ref = m[0]; int16_t best = 0;
      for (uint32_t i = 1; i < NST; i++) {
        int16_t r = (int16_t)(m[i] - ref);
        if (r < best) { best = r; st = i; }
      }
    }
  } else st = (uint32_t)endstate;   /* NOT masked in the port; callers pass 0 */
  int row = v->dp, bit = -1;
  while (delay-- > 0) {
    if (--row < 0) row = v->len - 1;
    /* the port indexes c[endstate>>3] with the unmasked value; states >= 2^23 would read
       outside the row there.  The restatement masks (defined behaviour only). */
    bit = (int)rowbit(v, row, st & MASK);
    st = ((uint32_t)bit << (ORC_K - 2)) | ((st & MASK) >> 1);
  }
  return bit;
}

/* viterbi224_sse2.c:206-243 (the port has no decodeword): best state by strict < from state 0 when end < 0, else
 * end & 0xffffff; walk `delay` rows back from dp with ring wrap; result = bit << 63 | result >> 1 per step.
 * The reference does not mask bit 23 of `end` (it would index past the decision row); here it is masked. */
unsigned long long orc_v224_decodeword(void *p, int delay, int endstate) {
  orc_v224 *v = p;
  if (!v) return 0;
  uint32_t st;
  if (endstate < 0) {
    st = 0;
    if (v->mode == ORC_V224_LITERAL) {
      const uint32_t *m = v->m32[v->cur];
      uint32_t best = m[0];
      for (uint32_t i = 1; i < NST; i++) if (m[i] < best) { best = m[i]; st = i; }
    } else {
      const uint16_t *m = v->m16[v->cur];
      uint16_t ref = m[0]; int16_t best = 0;
      for (uint32_t i = 1; i < NST; i++) {
        int16_t r = (int16_t)(m[i] - ref);
        if (r < best) { best = r; st = i; }
      }
    }
  } else st = (uint32_t)endstate & 0xffffffu & MASK;
  unsigned long long result = 0;
  int row = v->dp;
  while (delay-- > 0) {
    if (--row < 0) row = v->len - 1;
    unsigned bit = rowbit(v, row, st);
    st = (bit << (ORC_K - 2)) | (st >> 1);
    result = ((unsigned long long)bit << 63) | (result >> 1);
  }
  return result;
}

const uint8_t *orc_v224_row(void *p, int row) {
  orc_v224 *v = p;
  return v->rows + (size_t)row * ORC_ROWBYTES;
}
int orc_v224_dp(void *p) { return ((orc_v224 *)p)->dp; }

static void minmax(const orc_v224 *v, uint32_t *mn, uint32_t *mx, uint32_t *ref) {
  if (v->mode == ORC_V224_LITERAL) {
    const uint32_t *m = v->m32[v->cur];
    /* modular min/max relative to m[0] so that it stays meaningful across a u32 wrap */
    int32_t lo = 0, hi = 0;
    for (uint32_t i = 1; i < NST; i++) {
      int32_t r = (int32_t)(m[i] - m[0]);
      if (r < lo) lo = r;
      if (r > hi) hi = r;
    }
    *ref = m[0]; *mn = (uint32_t)lo; *mx = (uint32_t)hi;
  } else {
    const uint16_t *m = v->m16[v->cur];
    int32_t lo = 0, hi = 0;
    for (uint32_t i = 1; i < NST; i++) {
      int32_t r = (int16_t)(m[i] - m[0]);
      if (r < lo) lo = r;
      if (r > hi) hi = r;
    }
    *ref = m[0]; *mn = (uint32_t)lo; *mx = (uint32_t)hi;
  }
}

uint32_t orc_v224_spread(void *p) {
  uint32_t mn, mx, ref; minmax(p, &mn, &mx, &ref);
  return mx - mn;
}

uint32_t orc_v224_metric_rel(void *p, uint32_t state) {
  orc_v224 *v = p;
  uint32_t mn, mx, ref; minmax(v, &mn, &mx, &ref);
  state &= MASK;
  if (v->mode == ORC_V224_LITERAL)
    return (uint32_t)((int32_t)(v->m32[v->cur][state] - ref) - (int32_t)mn);
  return (uint32_t)((int32_t)(int16_t)(v->m16[v->cur][state] - (uint16_t)ref) - (int32_t)mn);
}

/* absolute smallest (want_max = 0) / largest (1) u32 metric of the LITERAL engine: the port's own
   never-renormalised scale (what sse2.c:82-109 reports modulo its bias and renormals) */
uint32_t orc_v224_metric_abs(void *p, int want_max) {
  orc_v224 *v = p;
  if (v->mode != ORC_V224_LITERAL) return 0xffffffffu;
  const uint32_t *m = v->m32[v->cur];
  uint32_t r = m[0];
  for (uint32_t i = 1; i < NST; i++)
    if (want_max ? m[i] > r : m[i] < r) r = m[i];
  return r;
}

/* Resume from a given set of path metrics: load 2^23 metrics (state order, any common offset: only their differences
   enter port.c:171-181) as the current ones and rewind the ring (dp = 0).  The trellis recursion has no other state, so
   everything decoded from here on is what a decoder that reached these metrics by itself would decode.  Used to pin the
   product deep inside a 10^7-symbol stream, where running the oracle from the start would take an hour. */
int orc_v224_set_metrics(void *p, const uint32_t *m) {
  orc_v224 *v = p;
  if (!v || !m) return -1;
  v->dp = 0;
  if (v->mode == ORC_V224_LITERAL) memcpy(v->m32[v->cur], m, sizeof(uint32_t) * NST);
  else for (uint32_t i = 0; i < NST; i++) v->m16[v->cur][i] = (uint16_t)m[i];
  return 0;
}

/* the current path metrics minus their minimum (same form as the product's v224hip_export_metrics) */
int orc_v224_get_metrics(void *p, uint32_t *out) {
  orc_v224 *v = p;
  if (!v || !out) return -1;
  uint32_t mn, mx, ref; minmax(v, &mn, &mx, &ref);
  for (uint32_t i = 0; i < NST; i++)
    out[i] = v->mode == ORC_V224_LITERAL ? (uint32_t)((int32_t)(v->m32[v->cur][i] - ref) - (int32_t)mn)
                                         : (uint32_t)((int32_t)(int16_t)(v->m16[v->cur][i] - (uint16_t)ref) - (int32_t)mn);
  return 0;
}

uint64_t orc_fnv1a(const void *buf, size_t n) {
  const uint8_t *b = buf;
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
  return h;
}

/* ---- encoder: encode.c:17-35.  MSB-first data, symbol 0 from POLY1, symbol 1 from POLY2 ---- */
uint64_t orc_encode(uint8_t *symbols, const uint8_t *data, unsigned nbytes, uint64_t encstate) {
  for (unsigned n = 0; n < nbytes; n++)
    for (int b = 7; b >= 0; b--) {
      encstate = (encstate << 1) | ((data[n] >> b) & 1u);
      *symbols++ = (uint8_t)(ORC_G1FLIP ^ __builtin_parityll(encstate & ORC_POLY1));
      *symbols++ = (uint8_t)(ORC_G2FLIP ^ __builtin_parityll(encstate & ORC_POLY2));
    }
  return encstate & ((1ull << ORC_K) - 1);
}
