/*
 * Synthetic mixed-entropy corpus used by the tests and bench.py (not part of the product).
 *
 * Block b is generated independently from its index; class = b % 8:
 *   0 uniform random bytes            4 text-like (as 3)
 *   1 period-32 ramp  (i & 0x1f)      5 LZ-synthetic: random literal runs + back-references
 *   2 single-byte run (0x41)          6 LZ-synthetic (as 5)
 *   3 text-like: skewed draws from a  7 4 KiB segments alternating class 0 / class 3
 *     4096-word vocabulary
 * All arithmetic is integer so the bytes are identical on every machine.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static uint64_t sm64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

#define VOCAB 4096
static uint8_t vocab_bytes[VOCAB][12];
static uint8_t vocab_len[VOCAB];
static int vocab_ready = 0;

static void vocab_init(void) {
  if (vocab_ready) return;
  uint64_t s = 12345;
  for (int w = 0; w < VOCAB; w++) {
    uint64_t r = sm64(&s);
    int len = 2 + (int)(r % 9);
    vocab_len[w] = (uint8_t)len;
    for (int k = 0; k < len; k++) {
      r = sm64(&s);
      vocab_bytes[w][k] = (uint8_t)('a' + (r % 26));
    }
  }
  vocab_ready = 1;
}

static void gen_random(uint8_t *out, size_t n, uint64_t seed) {
  uint64_t s = seed;
  size_t k = 0;
  while (k + 8 <= n) {
    uint64_t r = sm64(&s);
    memcpy(out + k, &r, 8);
    k += 8;
  }
  if (k < n) {
    uint64_t r = sm64(&s);
    memcpy(out + k, &r, n - k);
  }
}

static void gen_text(uint8_t *out, size_t n, uint64_t seed) {
  vocab_init();
  uint64_t s = seed;
  size_t pos = 0;
  while (pos < n) {
    uint64_t u = sm64(&s) >> 32;
    uint64_t t = (u * u) >> 32;
    t = (t * u) >> 32;               /* u^3: heavy skew towards low ranks */
    uint32_t rank = (uint32_t)((t * VOCAB) >> 32);
    size_t len = vocab_len[rank];
    if (len > n - pos) len = n - pos;
    memcpy(out + pos, vocab_bytes[rank], len);
    pos += len;
    if (pos < n) out[pos++] = ' ';
  }
}

static void gen_lz(uint8_t *out, size_t n, uint64_t seed) {
  uint64_t s = seed;
  size_t pos = 0;
  while (pos < n) {
    /* literal run, geometric with mean ~24 */
    size_t lit = 1;
    for (;;) {
      uint64_t r = sm64(&s);
      int stop = 0;
      for (int k = 0; k < 4 && !stop; k++) {
        if (((r >> (16 * k)) & 0xFFFF) < 2730) stop = 1;
        else lit++;
      }
      if (stop || lit > 400) break;
    }
    if (lit > n - pos) lit = n - pos;
    for (size_t k = 0; k < lit; k += 8) {
      uint64_t r = sm64(&s);
      size_t m = lit - k < 8 ? lit - k : 8;
      memcpy(out + pos + k, &r, m);
    }
    pos += lit;
    if (pos >= n) break;
    uint64_t w = sm64(&s);
    size_t mlen = 4 + (size_t)(w & 0xFF);
    if (mlen > n - pos) mlen = n - pos;
    unsigned bits = 1 + (unsigned)((w >> 8) % 16);
    size_t off = 1 + (size_t)((w >> 16) & ((1u << bits) - 1));
    if (off > pos) off = pos;
    for (size_t k = 0; k < mlen; k++) out[pos + k] = out[pos + k - off];
    pos += mlen;
  }
}

void corpus_block(uint64_t index, uint64_t corpus_seed, uint8_t *out, size_t n) {
  uint64_t seed = 0x9E3779B97F4A7C15ull ^ index ^ (corpus_seed << 32);
  switch (index % 8) {
    case 0: gen_random(out, n, seed); break;
    case 1: for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(i & 0x1f); break;
    case 2: memset(out, 0x41, n); break;
    case 3: case 4: gen_text(out, n, seed); break;
    case 5: case 6: gen_lz(out, n, seed); break;
    default: {
      for (size_t o = 0; o < n; o += 4096) {
        size_t m = n - o < 4096 ? n - o : 4096;
        if ((o / 4096) % 2 == 0) gen_random(out + o, m, seed + o);
        else gen_text(out + o, m, seed + o);
      }
    }
  }
}

/* blocks [first, first+count) of `block` bytes each, written consecutively */
void corpus_fill(uint64_t first, uint64_t count, uint64_t corpus_seed, size_t block, uint8_t *out) {
  vocab_init();
  for (uint64_t b = 0; b < count; b++) corpus_block(first + b, corpus_seed, out + b * block, block);
}
