/*
 * lz4ref — CPU oracle (TEST INFRASTRUCTURE ONLY; see lz4ref.h).
 *
 * Scalar restatement of dart_lz4's block/frame hot path.  Each function cites
 * the reference lines (path:line under lib/src) whose behaviour it follows.
 */
#include "lz4ref.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ status */

const char *lz4ref_status_type(int s) {
  if (s == LZ4REF_OK) return "OK";
  if (s >= 1 && s < 10) return "Lz4FormatException";
  if (s >= 10 && s < 30) return "Lz4CorruptDataException";
  if (s == LZ4REF_LIMIT) return "Lz4OutputLimitException";
  if (s >= 40 && s < 50) return "Lz4UnsupportedFeatureException";
  if (s == LZ4REF_DICT_NOT_FOUND) return "Lz4Exception";
  if (s == LZ4REF_RANGE) return "RangeError";
  return "BackendError";
}

const char *lz4ref_status_message(int s) {
  switch (s) {
    case LZ4REF_OK: return "";
    case LZ4REF_FORMAT_EOF: return "Unexpected end of input";
    case LZ4REF_FORMAT_MAGIC: return "Invalid LZ4 frame magic number";
    case LZ4REF_FORMAT_FLG_RESERVED: return "Reserved FLG bit is set";
    case LZ4REF_FORMAT_BD_RESERVED: return "Reserved BD bits are set";
    case LZ4REF_FORMAT_BLOCK_MAX: return "Invalid block maximum size";
    case LZ4REF_FORMAT_INT_OVERFLOW: return "Integer overflow";
    case LZ4REF_FORMAT_CONTENT_SIZE_ARG: return "contentSize does not match src length";
    case LZ4REF_FORMAT_CONTENT_SIZE_STREAM: return "contentSize does not match stream length";
    case LZ4REF_FORMAT_SIZE_HEADER: return "Input too short for size header";
    case LZ4REF_CORRUPT_LITERAL: return "Literal length exceeds output size";
    case LZ4REF_CORRUPT_MATCH: return "Match length exceeds output size";
    case LZ4REF_CORRUPT_TRAILING: return "Trailing bytes after end of block";
    case LZ4REF_CORRUPT_DISTANCE: return "Invalid match distance";
    case LZ4REF_CORRUPT_HEADER_CK: return "Header checksum mismatch";
    case LZ4REF_CORRUPT_BLOCK_SIZE: return "Block size exceeds maximum";
    case LZ4REF_CORRUPT_BLOCK_CK: return "Block checksum mismatch";
    case LZ4REF_CORRUPT_CONTENT_CK: return "Content checksum mismatch";
    case LZ4REF_CORRUPT_CONTENT_SIZE: return "Content size mismatch";
    case LZ4REF_CORRUPT_LEGACY_SIZE: return "Invalid legacy block size";
    case LZ4REF_CORRUPT_LEGACY_NOT_FULL: return "Legacy block is not full";
    case LZ4REF_LIMIT: return "Output limit exceeded";
    case LZ4REF_UNSUPPORTED_VERSION: return "Unsupported LZ4 frame version";
    case LZ4REF_UNSUPPORTED_DICT:
      return "Dictionary ID present but no dictionary resolver provided";
    case LZ4REF_UNSUPPORTED_CONTENT_SIZE: return "Content size exceeds supported integer range";
    case LZ4REF_DICT_NOT_FOUND: return "Dictionary not found for ID: ";
    case LZ4REF_RANGE: return "Invalid argument";
    case LZ4REF_DST_TOO_SMALL: return "destination buffer too small";
    default: return "unknown";
  }
}

/* ------------------------------------------------------- byte sink (writer)
 * Semantics of ByteWriter (internal/byte_writer.dart:5-154) that are visible
 * on the path: append, maxLength -> 'Output limit exceeded' (:128-132),
 * copyMatch (:93-121).  The growth policy is replaced by a caller buffer of
 * physical capacity `cap`. */
typedef struct {
  uint8_t *buf;
  size_t len;
  size_t cap;      /* physical capacity of buf */
  int64_t max_len; /* ByteWriter maxLength, <0 = none */
} sink_t;

static int sink_reserve(sink_t *w, size_t add) {
  if (w->max_len >= 0 && (uint64_t)w->len + add > (uint64_t)w->max_len)
    return LZ4REF_LIMIT;
  if (w->len + add > w->cap) return LZ4REF_DST_TOO_SMALL;
  return LZ4REF_OK;
}

static int sink_put(sink_t *w, const uint8_t *p, size_t n) {
  int st = sink_reserve(w, n);
  if (st) return st;
  if (n) memcpy(w->buf + w->len, p, n);
  w->len += n;
  return LZ4REF_OK;
}

static int sink_u8(sink_t *w, uint32_t v) {
  uint8_t b = (uint8_t)v;
  return sink_put(w, &b, 1);
}

static int sink_u32le(sink_t *w, uint32_t v) {
  uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
  return sink_put(w, b, 4);
}

/* byte_writer.dart:93-121 — distance validated before capacity. */
static int sink_copy_match(sink_t *w, size_t distance, size_t match_len) {
  if (distance == 0 || distance > w->len) return LZ4REF_CORRUPT_DISTANCE;
  if (match_len == 0) return LZ4REF_OK;
  int st = sink_reserve(w, match_len);
  if (st) return st;
  uint8_t *d = w->buf + w->len;
  const uint8_t *s = d - distance;
  for (size_t k = 0; k < match_len; k++) d[k] = s[k]; /* byte-serial forward copy */
  w->len += match_len;
  return LZ4REF_OK;
}

/* --------------------------------------------------------------- byte source
 * ByteReader (internal/byte_reader.dart): every short read is FORMAT_EOF. */
typedef struct {
  const uint8_t *p;
  size_t len;
  size_t off;
} src_t;

static size_t src_remaining(const src_t *r) { return r->len - r->off; }
static int src_eof(const src_t *r) { return r->off >= r->len; }

static int src_u8(src_t *r, uint32_t *v) {
  if (src_remaining(r) < 1) return LZ4REF_FORMAT_EOF;
  *v = r->p[r->off++];
  return LZ4REF_OK;
}
static int src_u16le(src_t *r, uint32_t *v) {
  if (src_remaining(r) < 2) return LZ4REF_FORMAT_EOF;
  *v = (uint32_t)r->p[r->off] | ((uint32_t)r->p[r->off + 1] << 8);
  r->off += 2;
  return LZ4REF_OK;
}
static int src_u32le(src_t *r, uint32_t *v) {
  if (src_remaining(r) < 4) return LZ4REF_FORMAT_EOF;
  const uint8_t *q = r->p + r->off;
  *v = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
  r->off += 4;
  return LZ4REF_OK;
}
static int src_view(src_t *r, uint64_t count, const uint8_t **v) {
  if (count > src_remaining(r)) return LZ4REF_FORMAT_EOF;
  *v = r->p + r->off;
  r->off += (size_t)count;
  return LZ4REF_OK;
}

/* ------------------------------------------------------------------- xxh32
 * xxhash/xxh32.dart: primes :3-7, one-shot :166-240, streaming :21-164,
 * round :242-247.  _mul32 (:9-19) is a plain multiply modulo 2^32. */
#define P1 0x9E3779B1u
#define P2 0x85EBCA77u
#define P3 0xC2B2AE3Du
#define P4 0x27D4EB2Fu
#define P5 0x165667B1u

static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t rd32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint32_t xxh_round(uint32_t acc, uint32_t x) {
  acc += x * P2;
  acc = rotl32(acc, 13);
  return acc * P1;
}
static uint32_t xxh_tail(uint32_t h, const uint8_t *p, size_t len) {
  size_t k = 0;
  while (k + 4 <= len) { /* :221-225 */
    h += rd32(p + k) * P3;
    h = rotl32(h, 17) * P4;
    k += 4;
  }
  while (k < len) { /* :227-231 */
    h += (uint32_t)p[k] * P5;
    h = rotl32(h, 11) * P1;
    k++;
  }
  h ^= h >> 15; /* :233-237 */
  h *= P2;
  h ^= h >> 13;
  h *= P3;
  h ^= h >> 16;
  return h;
}

uint32_t lz4ref_xxh32(const uint8_t *p, size_t len, uint32_t seed) {
  size_t k = 0;
  uint32_t h;
  if (len >= 16) {
    uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    size_t limit = len - 16;
    while (k <= limit) {
      v1 = xxh_round(v1, rd32(p + k));
      v2 = xxh_round(v2, rd32(p + k + 4));
      v3 = xxh_round(v3, rd32(p + k + 8));
      v4 = xxh_round(v4, rd32(p + k + 12));
      k += 16;
    }
    h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
  } else {
    h = seed + P5;
  }
  h += (uint32_t)len;
  return xxh_tail(h, p + k, len - k);
}

void lz4ref_xxh32_init(lz4ref_xxh32_state *s, uint32_t seed) { /* :33-41 */
  s->seed = seed;
  s->v1 = seed + P1 + P2;
  s->v2 = seed + P2;
  s->v3 = seed;
  s->v4 = seed - P1;
  s->total_len = 0;
  s->mem_size = 0;
  memset(s->mem, 0, sizeof s->mem);
}

void lz4ref_xxh32_update(lz4ref_xxh32_state *s, const uint8_t *p, size_t len) { /* :43-125 */
  if (len == 0) return;
  s->total_len += len;
  if (s->mem_size + len < 16) {
    memcpy(s->mem + s->mem_size, p, len);
    s->mem_size += (uint32_t)len;
    return;
  }
  if (s->mem_size != 0) {
    size_t fill = 16 - s->mem_size;
    memcpy(s->mem + s->mem_size, p, fill);
    p += fill;
    len -= fill;
    s->mem_size = 0;
    s->v1 = xxh_round(s->v1, rd32(s->mem));
    s->v2 = xxh_round(s->v2, rd32(s->mem + 4));
    s->v3 = xxh_round(s->v3, rd32(s->mem + 8));
    s->v4 = xxh_round(s->v4, rd32(s->mem + 12));
  }
  while (len >= 16) {
    s->v1 = xxh_round(s->v1, rd32(p));
    s->v2 = xxh_round(s->v2, rd32(p + 4));
    s->v3 = xxh_round(s->v3, rd32(p + 8));
    s->v4 = xxh_round(s->v4, rd32(p + 12));
    p += 16;
    len -= 16;
  }
  if (len) {
    memcpy(s->mem, p, len);
    s->mem_size = (uint32_t)len;
  }
}

uint32_t lz4ref_xxh32_digest(const lz4ref_xxh32_state *s) { /* :127-163 */
  uint32_t h;
  if (s->total_len >= 16)
    h = rotl32(s->v1, 1) + rotl32(s->v2, 7) + rotl32(s->v3, 12) + rotl32(s->v4, 18);
  else
    h = s->seed + P5;
  h += (uint32_t)s->total_len;
  return xxh_tail(h, s->mem, s->mem_size);
}

/* ------------------------------------------------------ sequence emission
 * block/lz4_block_encoder.dart:172-223 (identical copy in the HC encoder,
 * hc/lz4_hc_block_encoder.dart:212-267). */
static int emit_length(sink_t *w, size_t length) { /* :216-223 */
  int st;
  while (length >= 255) {
    if ((st = sink_u8(w, 255))) return st;
    length -= 255;
  }
  return sink_u8(w, (uint32_t)length);
}

static int emit_sequence(sink_t *w, const uint8_t *in, size_t lit_start, size_t lit_len,
                         size_t distance, size_t match_len) { /* :172-200 */
  int st;
  size_t m4 = match_len - 4;
  uint32_t tl = lit_len < 15 ? (uint32_t)lit_len : 15;
  uint32_t tm = m4 < 15 ? (uint32_t)m4 : 15;
  if ((st = sink_u8(w, (tl << 4) | tm))) return st;
  if (lit_len >= 15 && (st = emit_length(w, lit_len - 15))) return st;
  if (lit_len && (st = sink_put(w, in + lit_start, lit_len))) return st;
  if ((st = sink_u8(w, (uint32_t)(distance & 0xff)))) return st;
  if ((st = sink_u8(w, (uint32_t)((distance >> 8) & 0xff)))) return st;
  if (m4 >= 15 && (st = emit_length(w, m4 - 15))) return st;
  return LZ4REF_OK;
}

static int emit_last_literals(sink_t *w, const uint8_t *in, size_t start, size_t length) { /* :202-214 */
  int st;
  uint32_t tl = length < 15 ? (uint32_t)length : 15;
  if ((st = sink_u8(w, tl << 4))) return st;
  if (length >= 15 && (st = emit_length(w, length - 15))) return st;
  if (length && (st = sink_put(w, in + start, length))) return st;
  return LZ4REF_OK;
}

size_t lz4ref_compress_bound(size_t n) { return n + n / 255 + 16; }

static uint32_t hash16(uint32_t seq) { return (seq * 2654435761u) >> 16; } /* :167-170 */

/* dict‖src work buffer (:50-72): dictionary trimmed to its last 65535 bytes. */
static const uint8_t *make_input(const uint8_t *src, size_t n, const uint8_t *dict,
                                 size_t dict_len, size_t *dict_used, uint8_t **owned) {
  *owned = NULL;
  if (dict == NULL || dict_len == 0) {
    *dict_used = 0;
    return src;
  }
  if (dict_len > 65535) {
    dict += dict_len - 65535;
    dict_len = 65535;
  }
  uint8_t *buf = (uint8_t *)malloc(dict_len + n);
  if (!buf) return NULL;
  memcpy(buf, dict, dict_len);
  memcpy(buf + dict_len, src, n);
  *dict_used = dict_len;
  *owned = buf;
  return buf;
}

/* ------------------------------------------------------ fast block encoder
 * block/lz4_block_encoder.dart:35-157. */
/* End-of-block limits.  The reference searches up to the last four bytes and lets a match
 * run to the very end of the block (plimit = elimit = total).  The optional spec-conformant
 * mode (not in the reference; lz4 block format "end of block" rules) stops searching 12 bytes
 * before the end, keeps the last 5 bytes literal and encodes blocks under 13 bytes as
 * literals only, so that any LZ4 decoder accepts the block. */
static void end_limits(int conformant, size_t total, size_t n, size_t *plimit, size_t *elimit) {
  *plimit = total;
  *elimit = total;
  if (conformant) {
    *plimit = n >= 13 ? total - 8 : 0;
    *elimit = n >= 5 ? total - 5 : 0;
  }
}

static int fast_encode_to_sink_ex(sink_t *w, const uint8_t *src, size_t n, const uint8_t *dict,
                                  size_t dict_len_in, int64_t acceleration, int conformant) {
  if (acceleration < 1) return LZ4REF_RANGE; /* :41-43, before the empty check */
  if (n == 0) return LZ4REF_OK;              /* :45-48 */

  size_t dict_len;
  uint8_t *owned;
  const uint8_t *in = make_input(src, n, dict, dict_len_in, &dict_len, &owned);
  if (!in) return LZ4REF_DST_TOO_SMALL;
  int st = LZ4REF_OK;
  int32_t *table = NULL;

  if (n < 4) { /* :74-78 */
    st = emit_last_literals(w, in, dict_len, n);
    goto done;
  }
  table = (int32_t *)malloc(65536 * sizeof(int32_t));
  if (!table) { st = LZ4REF_DST_TOO_SMALL; goto done; }
  for (int k = 0; k < 65536; k++) table[k] = -1; /* :80-81 */

  if (dict_len != 0) /* :83-88 */
    for (size_t pos = 0; pos + 4 <= dict_len; pos++) table[hash16(rd32(in + pos))] = (int32_t)pos;

  {
    size_t total = dict_len + n;
    size_t anchor = dict_len, i = dict_len;
    size_t plimit, elimit;
    end_limits(conformant, total, n, &plimit, &elimit);
    /* Dart's >> on a non-negative int yields 0 for shift counts >= 64 */
    unsigned sh = acceleration > 62 ? 63 : (unsigned)acceleration;
    while (i + 4 <= plimit) { /* :95 */
      uint32_t seq = rd32(in + i);
      uint32_t h = hash16(seq);
      int32_t ref = table[h];
      table[h] = (int32_t)i; /* :100 unconditional */
      if (ref >= 0 && i - (size_t)ref <= 0xFFFF && rd32(in + ref) == seq) { /* :102-103 */
        size_t ms = i, rs = (size_t)ref;
        while (ms > anchor && rs > 0 && in[ms - 1] == in[rs - 1]) { ms--; rs--; } /* :107-113 */
        size_t ml = 4;
        while (ms + ml < elimit && in[ms + ml] == in[rs + ml]) ml++; /* :117-121 */
        if ((st = emit_sequence(w, in, anchor, ms - anchor, ms - rs, ml))) goto done;
        i = ms + ml;
        anchor = i;
        if (i + 4 > plimit) break; /* :135-137 */
        for (size_t j = i - ml + 1; j + 4 <= i; j++) /* :139-145 */
          table[hash16(rd32(in + j))] = (int32_t)j;
        continue;
      }
      i += 1 + (size_t)(((uint64_t)(i - anchor)) >> sh); /* :150 */
    }
    if (total - anchor != 0) /* :153-156 */
      st = emit_last_literals(w, in, anchor, total - anchor);
  }
done:
  free(table);
  free(owned);
  return st;
}

static int fast_encode_to_sink(sink_t *w, const uint8_t *src, size_t n, const uint8_t *dict,
                               size_t dict_len_in, int64_t acceleration) {
  return fast_encode_to_sink_ex(w, src, n, dict, dict_len_in, acceleration, 0);
}

int lz4ref_block_compress_fast(const uint8_t *src, size_t n, const uint8_t *dict, size_t dict_len,
                               int64_t acceleration, uint8_t *dst, size_t cap, size_t *out_len) {
  sink_t w = {dst, 0, cap, -1};
  int st = fast_encode_to_sink(&w, src, n, dict, dict_len, acceleration);
  if (out_len) *out_len = w.len;
  return st;
}

/* -------------------------------------------------------- HC block encoder
 * hc/lz4_hc_block_encoder.dart:49-197; options hc/lz4_hc_options.dart:10-16. */
static int hc_encode_to_sink_ex(sink_t *w, const uint8_t *src, size_t n, const uint8_t *dict,
                                size_t dict_len_in, int64_t max_depth, int conformant) {
  if (max_depth < 1) return LZ4REF_RANGE; /* Lz4HcOptions ctor */
  if (n == 0) return LZ4REF_OK;           /* :57-60 */

  size_t dict_len;
  uint8_t *owned;
  const uint8_t *in = make_input(src, n, dict, dict_len_in, &dict_len, &owned);
  if (!in) return LZ4REF_DST_TOO_SMALL;
  int st = LZ4REF_OK;
  int32_t *head = NULL, *chain = NULL;

  if (n < 4) { /* :86-90 */
    st = emit_last_literals(w, in, dict_len, n);
    goto done;
  }
  {
    size_t total = dict_len + n;
    head = (int32_t *)malloc(65536 * sizeof(int32_t));
    chain = (int32_t *)malloc(total * sizeof(int32_t));
    if (!head || !chain) { st = LZ4REF_DST_TOO_SMALL; goto done; }
    for (int k = 0; k < 65536; k++) head[k] = -1;       /* :92-93 */
    for (size_t k = 0; k < total; k++) chain[k] = -1;    /* :94, :34 */

#define HC_INSERT(pos, refvar)                       \
  do {                                               \
    uint32_t h_ = hash16(rd32(in + (pos)));          \
    (refvar) = head[h_];                             \
    chain[(pos)] = (refvar);                         \
    head[h_] = (int32_t)(pos);                       \
  } while (0) /* :96-103 */

    int32_t tmp;
    if (dict_len != 0)
      for (size_t pos = 0; pos + 4 <= dict_len; pos++) HC_INSERT(pos, tmp); /* :105-109 */

    size_t anchor = dict_len, i = dict_len;
    size_t plimit, elimit;
    end_limits(conformant, total, n, &plimit, &elimit);
    while (i + 4 <= plimit) { /* :116 */
      int32_t cand;
      HC_INSERT(i, cand);
      size_t best_len = 0;
      int32_t best_ref = -1;
      int64_t depth = 0;
      uint32_t seq_i = rd32(in + i);
      while (cand >= 0 && depth < max_depth) { /* :123 */
        if (i - (size_t)cand > 0xFFFF) { /* :125-129: far candidates still count depth */
          cand = chain[cand];
          depth++;
          continue;
        }
        if (rd32(in + cand) == seq_i) { /* :131 */
          size_t ml = 4;
          while (i + ml < elimit && in[i + ml] == in[(size_t)cand + ml]) ml++;
          if (ml > best_len) { /* strict: nearest wins ties */
            best_len = ml;
            best_ref = cand;
            if (i + best_len == elimit) break; /* :140-142 */
          }
        }
        cand = chain[cand];
        depth++;
      }
      if (best_len >= 4) { /* :150 */
        size_t ms = i, rs = (size_t)best_ref, al = best_len;
        while (ms > anchor && rs > 0 && in[ms - 1] == in[rs - 1]) { ms--; rs--; al++; } /* :155-161 */
        if ((st = emit_sequence(w, in, anchor, ms - anchor, ms - rs, al))) goto done;
        i = ms + al;
        anchor = i;
        if (i + 4 > plimit) break; /* :176-178 */
        for (size_t j = i - al + 1; j + 4 <= i; j++) HC_INSERT(j, tmp); /* :180-185 */
        continue;
      }
      i++; /* :190 */
    }
    if (total - anchor != 0) st = emit_last_literals(w, in, anchor, total - anchor); /* :193-196 */
  }
done:
  free(head);
  free(chain);
  free(owned);
  return st;
}

static int hc_encode_to_sink(sink_t *w, const uint8_t *src, size_t n, const uint8_t *dict,
                             size_t dict_len_in, int64_t max_depth) {
  return hc_encode_to_sink_ex(w, src, n, dict, dict_len_in, max_depth, 0);
}

int lz4ref_block_compress_ex(const uint8_t *src, size_t n, const uint8_t *dict, size_t dict_len, int level,
                             int64_t param, int conformant, uint8_t *dst, size_t cap, size_t *out_len) {
  sink_t w = {dst, 0, cap, -1};
  int st = level == 0 ? fast_encode_to_sink_ex(&w, src, n, dict, dict_len, param, conformant)
                      : hc_encode_to_sink_ex(&w, src, n, dict, dict_len, param, conformant);
  if (out_len) *out_len = w.len;
  return st;
}

int lz4ref_block_compress_hc(const uint8_t *src, size_t n, const uint8_t *dict, size_t dict_len,
                             int64_t max_search_depth, uint8_t *dst, size_t cap, size_t *out_len) {
  sink_t w = {dst, 0, cap, -1};
  int st = hc_encode_to_sink(&w, src, n, dict, dict_len, max_search_depth);
  if (out_len) *out_len = w.len;
  return st;
}

/* ------------------------------------------------------------ block decoder */

static int read_ext_length(src_t *r, uint64_t *total) { /* lz4_block_decoder.dart:121-130 */
  uint64_t t = 0;
  for (;;) {
    uint32_t b;
    int st = src_u8(r, &b);
    if (st) return st;
    t += b;
    if (b != 255) break;
  }
  *total = t;
  return LZ4REF_OK;
}

/* strict: lz4_block_decoder.dart:7-77 */
int lz4ref_block_decompress(const uint8_t *src, size_t c, uint8_t *dst, int64_t size) {
  if (size < 0) return LZ4REF_RANGE; /* :11-13 */
  src_t r = {src, c, 0};
  sink_t w = {dst, 0, (size_t)size, size};
  int st;
  while (w.len < (size_t)size) { /* :21 */
    if (src_eof(&r)) return LZ4REF_FORMAT_EOF; /* :22-24 */
    uint32_t token;
    if ((st = src_u8(&r, &token))) return st;
    uint64_t lit = token >> 4;
    if (lit == 15) {
      uint64_t e;
      if ((st = read_ext_length(&r, &e))) return st;
      lit += e;
    }
    if (lit > (uint64_t)size - w.len) return LZ4REF_CORRUPT_LITERAL; /* :33-35 */
    if (lit != 0) {
      const uint8_t *v;
      if ((st = src_view(&r, lit, &v))) return st;
      if ((st = sink_put(&w, v, (size_t)lit))) return st;
    }
    if (w.len == (size_t)size) { /* :42-48 */
      if (!src_eof(&r)) return LZ4REF_CORRUPT_TRAILING;
      return LZ4REF_OK;
    }
    if (src_remaining(&r) < 2) return LZ4REF_FORMAT_EOF; /* :50-52 */
    uint32_t distance;
    if ((st = src_u16le(&r, &distance))) return st;
    uint64_t ml = (token & 15) + 4;
    if ((token & 15) == 15) {
      uint64_t e;
      if ((st = read_ext_length(&r, &e))) return st;
      ml += e;
    }
    if (ml > (uint64_t)size - w.len) return LZ4REF_CORRUPT_MATCH; /* :61-63 */
    st = sink_copy_match(&w, distance, (size_t)ml);
    if (st == LZ4REF_LIMIT) return LZ4REF_CORRUPT_MATCH; /* :65-69 */
    if (st) return st;
  }
  if (!src_eof(&r)) return LZ4REF_CORRUPT_TRAILING; /* :72-74 */
  return LZ4REF_OK;
}

/* lenient: lz4_block_decoder.dart:79-119, into a writer that may hold a prefix */
static int decode_into_sink(const uint8_t *src, size_t c, sink_t *w) {
  src_t r = {src, c, 0};
  int st;
  for (;;) {
    if (src_eof(&r)) return LZ4REF_OK; /* :86-88 */
    uint32_t token;
    if ((st = src_u8(&r, &token))) return st;
    uint64_t lit = token >> 4;
    if (lit == 15) {
      uint64_t e;
      if ((st = read_ext_length(&r, &e))) return st;
      lit += e;
    }
    if (lit != 0) { /* :97-100: reader truncation is raised before the writer cap */
      const uint8_t *v;
      if ((st = src_view(&r, lit, &v))) return st;
      if (lit > (uint64_t)SIZE_MAX) return LZ4REF_LIMIT;
      if ((st = sink_put(w, v, (size_t)lit))) return st;
    }
    if (src_eof(&r)) return LZ4REF_OK;                   /* :102-104 */
    if (src_remaining(&r) < 2) return LZ4REF_FORMAT_EOF; /* :106-108 */
    uint32_t distance;
    if ((st = src_u16le(&r, &distance))) return st;
    uint64_t ml = (token & 15) + 4;
    if ((token & 15) == 15) {
      uint64_t e;
      if ((st = read_ext_length(&r, &e))) return st;
      ml += e;
    }
    if (distance == 0 || distance > w->len) return LZ4REF_CORRUPT_DISTANCE;
    if (w->max_len >= 0 && (uint64_t)w->len + ml > (uint64_t)w->max_len) return LZ4REF_LIMIT;
    if ((st = sink_copy_match(w, distance, (size_t)ml))) return st; /* :117 */
  }
}

int lz4ref_block_decompress_into(const uint8_t *src, size_t c, uint8_t *dst, size_t prefix_len,
                                 size_t max_len, size_t *out_len) {
  sink_t w = {dst, prefix_len, max_len, (int64_t)max_len};
  int st = decode_into_sink(src, c, &w);
  if (out_len) *out_len = w.len;
  return st;
}

/* ----------------------------------------------------------- frame encoder */

void lz4ref_frame_options_default(lz4ref_frame_options *o) { /* lz4_frame_options.dart:78-88 */
  memset(o, 0, sizeof *o);
  o->block_size_id = 7;
  o->block_independence = 1;
  o->acceleration = 1;
  o->hc_max_search_depth = 64;
}

static size_t block_max_bytes(int id) { /* lz4_frame_options.dart:117-128 */
  switch (id) {
    case 4: return 64u * 1024;
    case 5: return 256u * 1024;
    case 6: return 1024u * 1024;
    case 7: return 4u * 1024 * 1024;
    default: return 0;
  }
}

size_t lz4ref_frame_bound(size_t n, const lz4ref_frame_options *o) {
  size_t bmax = block_max_bytes(o->block_size_id);
  size_t nblocks = bmax ? (n + bmax - 1) / bmax : 0;
  /* while a block is being compressed in place it may transiently reach
   * compress_bound(block) before being rewound to the raw form */
  return 19 + nblocks * 8 + n + 8 + lz4ref_compress_bound(bmax < n ? bmax : n);
}

/* lz4_frame_encoder.dart:105-274 */
int lz4ref_frame_encode(const uint8_t *src, size_t n, const lz4ref_frame_options *o,
                        const uint8_t *dict, size_t dict_len, uint8_t *dst, size_t cap,
                        size_t *out_len) {
  int st;
  if (o->acceleration < 1) return LZ4REF_RANGE; /* options ctor :89-91 */
  size_t bmax = block_max_bytes(o->block_size_id);
  if (!bmax) return LZ4REF_RANGE;
  sink_t w = {dst, 0, cap, -1};

  uint32_t flg = (1u << 6) | ((o->block_independence ? 1u : 0u) << 5) |
                 ((o->block_checksum ? 1u : 0u) << 4) | ((o->has_content_size ? 1u : 0u) << 3) |
                 ((o->content_checksum ? 1u : 0u) << 2) | (o->has_dict_id ? 1u : 0u); /* :112-117 */
  uint32_t bd = ((uint32_t)o->block_size_id & 7u) << 4;                                  /* :119 */
  if ((st = sink_u32le(&w, 0x184D2204u))) return st;
  if ((st = sink_u8(&w, flg))) return st;
  if ((st = sink_u8(&w, bd))) return st;
  if (o->has_content_size) { /* :128-133 */
    if ((st = sink_u32le(&w, (uint32_t)(o->content_size & 0xFFFFFFFFu)))) return st;
    if ((st = sink_u32le(&w, (uint32_t)(o->content_size >> 32)))) return st;
  }
  if (o->has_dict_id && (st = sink_u32le(&w, o->dict_id))) return st; /* :135-138 */
  uint32_t hc = (lz4ref_xxh32(w.buf + 4, w.len - 4, 0) >> 8) & 0xff;  /* :140-144 */
  if ((st = sink_u8(&w, hc))) return st;

  lz4ref_xxh32_state content;
  lz4ref_xxh32_init(&content, 0);

  const size_t window = 64 * 1024;
  uint8_t *combined = NULL;
  size_t offset = 0;
  while (offset < n) { /* :154 */
    size_t end = offset + bmax;
    if (end > n) end = n;
    const uint8_t *chunk = src + offset;
    size_t chunk_len = end - offset;
    if (o->content_checksum) lz4ref_xxh32_update(&content, chunk, chunk_len); /* :162-164 */

    const uint8_t *bdict = NULL;
    size_t bdict_len = 0;
    if (o->block_independence) { /* :167-169 */
      bdict = dict;
      bdict_len = dict ? dict_len : 0;
    } else if (offset == 0) { /* :176-177 */
      bdict = dict;
      bdict_len = dict ? dict_len : 0;
    } else if (offset >= window) { /* :178-180 */
      bdict = src + offset - window;
      bdict_len = window;
    } else { /* :181-206 mixed history */
      size_t dl = dict ? dict_len : 0;
      if (dl == 0) {
        bdict = src;
        bdict_len = offset;
      } else {
        size_t needed = window - offset;
        size_t take = dl > needed ? needed : dl;
        free(combined);
        combined = (uint8_t *)malloc(take + offset);
        if (!combined) return LZ4REF_DST_TOO_SMALL;
        memcpy(combined, dict + dl - take, take);
        memcpy(combined + take, src, offset);
        bdict = combined;
        bdict_len = take + offset;
      }
    }

    size_t block_start = w.len;
    if ((st = sink_u32le(&w, 0))) goto fail; /* placeholder :210-211 */
    size_t payload_start = w.len;
    if (o->compression == 0)
      st = fast_encode_to_sink_ex(&w, chunk, chunk_len, bdict, bdict_len, o->acceleration,
                                  o->spec_conformant);
    else
      st = hc_encode_to_sink_ex(&w, chunk, chunk_len, bdict, bdict_len, o->hc_max_search_depth,
                                o->spec_conformant);
    if (st) goto fail;

    size_t clen = w.len - payload_start;
    if (clen < chunk_len) { /* :235-247 */
      uint32_t v = (uint32_t)clen;
      w.buf[block_start] = (uint8_t)v;
      w.buf[block_start + 1] = (uint8_t)(v >> 8);
      w.buf[block_start + 2] = (uint8_t)(v >> 16);
      w.buf[block_start + 3] = (uint8_t)(v >> 24);
      if (o->block_checksum &&
          (st = sink_u32le(&w, lz4ref_xxh32(w.buf + payload_start, clen, 0))))
        goto fail;
    } else { /* :248-258 */
      w.len = block_start;
      if ((st = sink_u32le(&w, 0x80000000u | (uint32_t)chunk_len))) goto fail;
      if ((st = sink_put(&w, chunk, chunk_len))) goto fail;
      if (o->block_checksum && (st = sink_u32le(&w, lz4ref_xxh32(chunk, chunk_len, 0)))) goto fail;
    }
    offset = end;
  }
  free(combined);
  combined = NULL;

  if ((st = sink_u32le(&w, 0))) return st; /* :263 */
  if (o->content_checksum && (st = sink_u32le(&w, lz4ref_xxh32_digest(&content)))) return st;
  if (o->has_content_size && (uint64_t)n != o->content_size) /* :269-271, after all work */
    return LZ4REF_FORMAT_CONTENT_SIZE_ARG;
  if (out_len) *out_len = w.len;
  return LZ4REF_OK;
fail:
  free(combined);
  return st;
}

/* ----------------------------------------------------------- frame decoder
 * lz4_frame_decoder.dart:27-398. */
typedef struct {
  src_t r;
  sink_t out;              /* _out, maxLength = maxOutputBytes */
  int64_t max_output;      /* <0 = null */
  lz4ref_dict_resolver resolver;
  void *user;
  int have_dict;           /* _dict != null; persists across frames (:34,235) */
  const uint8_t *dict;
  size_t dict_len;
  uint32_t err_dict_id;
} fdec_t;

static int is_legacy_boundary(uint32_t magic) { /* :150-155 */
  if (magic == 0x184D2204u || magic == 0x184C2102u) return 1;
  return (magic & 0xFFFFFFF0u) == 0x184D2A50u;
}

static uint32_t peek_u32(const src_t *r) { return rd32(r->p + r->off); }

static int decode_legacy_frame(fdec_t *d) { /* :86-136 */
  const size_t legacy_max = 8u * 1024 * 1024;
  size_t frame_start = d->out.len;
  uint8_t *tmp = (uint8_t *)malloc(legacy_max);
  if (!tmp) return LZ4REF_DST_TOO_SMALL;
  int st = LZ4REF_OK;
  for (;;) {
    if (src_remaining(&d->r) == 0) break;
    if (src_remaining(&d->r) < 4) { st = LZ4REF_FORMAT_EOF; break; }
    if (is_legacy_boundary(peek_u32(&d->r))) break;
    uint32_t csize;
    if ((st = src_u32le(&d->r, &csize))) break;
    if (csize == 0) { st = LZ4REF_CORRUPT_LEGACY_SIZE; break; }
    const uint8_t *payload;
    if ((st = src_view(&d->r, csize, &payload))) break;
    sink_t t = {tmp, 0, legacy_max, (int64_t)legacy_max};
    if ((st = decode_into_sink(payload, csize, &t))) break;
    if ((st = sink_put(&d->out, tmp, t.len))) break;
    if (src_remaining(&d->r) == 0) break;
    if (src_remaining(&d->r) < 4) { st = LZ4REF_FORMAT_EOF; break; }
    if (!is_legacy_boundary(peek_u32(&d->r)) && t.len != legacy_max) {
      st = LZ4REF_CORRUPT_LEGACY_NOT_FULL;
      break;
    }
  }
  free(tmp);
  (void)frame_start;
  return st;
}

static int decode_standard_frame(fdec_t *d) { /* :157-324 */
  int st;
  src_t *r = &d->r;
  size_t frame_start = d->out.len; /* frameOut is a view of _out past this point */
  size_t desc_start = r->off;
  if (src_remaining(r) < 3) return LZ4REF_FORMAT_EOF; /* :168-170 */
  uint32_t flg = 0, bd = 0;
  src_u8(r, &flg);
  src_u8(r, &bd);
  if (((flg >> 6) & 3) != 1) return LZ4REF_UNSUPPORTED_VERSION; /* :175-179 */
  int indep = (flg >> 5) & 1, block_ck = (flg >> 4) & 1, has_size = (flg >> 3) & 1;
  int content_ck = (flg >> 2) & 1, has_dict = flg & 1;
  if ((flg >> 1) & 1) return LZ4REF_FORMAT_FLG_RESERVED; /* :188-190 */
  if (bd & 0x8F) return LZ4REF_FORMAT_BD_RESERVED;       /* :192-194 */
  size_t bmax = block_max_bytes((bd >> 4) & 7);
  if (!bmax) return LZ4REF_FORMAT_BLOCK_MAX; /* :196-197, :385-398 */
  uint64_t content_size = 0;
  if (has_size) { /* :199-205 */
    if (src_remaining(r) < 8) return LZ4REF_FORMAT_EOF;
    uint32_t lo, hi;
    src_u32le(r, &lo);
    src_u32le(r, &hi);
    content_size = ((uint64_t)hi << 32) | lo;
    if (content_size >> 63) return LZ4REF_FORMAT_INT_OVERFLOW; /* byte_reader.dart:74-76 */
  }
  uint32_t dict_id = 0;
  if (has_dict) { /* :207-213 */
    if (src_remaining(r) < 4) return LZ4REF_FORMAT_EOF;
    src_u32le(r, &dict_id);
  }
  size_t desc_end = r->off;
  uint32_t hc;
  if ((st = src_u8(r, &hc))) return st; /* :217 */
  if (hc != ((lz4ref_xxh32(r->p + desc_start, desc_end - desc_start, 0) >> 8) & 0xff))
    return LZ4REF_CORRUPT_HEADER_CK; /* :218-223 */
  if (has_dict) { /* :225-236 */
    if (!d->resolver) return LZ4REF_UNSUPPORTED_DICT;
    const uint8_t *dp = NULL;
    size_t dl = 0;
    if (!d->resolver(d->user, dict_id, &dp, &dl)) {
      d->err_dict_id = dict_id;
      return LZ4REF_DICT_NOT_FOUND;
    }
    d->have_dict = 1;
    d->dict = dp;
    d->dict_len = dl;
  }

  const size_t window = 64 * 1024;
  uint8_t *tmp = (uint8_t *)malloc(window + bmax);
  if (!tmp) return LZ4REF_DST_TOO_SMALL;
  st = LZ4REF_OK;
  for (;;) { /* :238-304 */
    if (src_remaining(r) < 4) { st = LZ4REF_FORMAT_EOF; break; }
    uint32_t raw;
    src_u32le(r, &raw);
    if (raw == 0) break;
    int stored = (raw & 0x80000000u) != 0;
    uint32_t bsize = raw & 0x7FFFFFFFu;
    if (bsize > bmax) { st = LZ4REF_CORRUPT_BLOCK_SIZE; break; } /* :251-253 */
    const uint8_t *payload;
    if ((st = src_view(r, bsize, &payload))) break; /* :255 */
    if (block_ck) {                                  /* :257-266 */
      uint32_t expected;
      if ((st = src_u32le(r, &expected))) break;
      if (lz4ref_xxh32(payload, bsize, 0) != expected) { st = LZ4REF_CORRUPT_BLOCK_CK; break; }
    }
    if (stored) { /* :268-271 */
      if ((st = sink_put(&d->out, payload, bsize))) break;
      continue;
    }
    if (indep) {
      size_t eff = 0;
      if (d->have_dict) { /* :275-290 — decoder keeps the last 65536 dictionary bytes */
        eff = d->dict_len > window ? window : d->dict_len;
        memcpy(tmp, d->dict + d->dict_len - eff, eff);
      }
      sink_t t = {tmp, eff, eff + bmax, (int64_t)(eff + bmax)};
      if ((st = decode_into_sink(payload, bsize, &t))) break;
      if ((st = sink_put(&d->out, tmp + eff, t.len - eff))) break;
    } else { /* _decodeDependentBlock :326-382 */
      size_t out_len = d->out.len - frame_start;
      size_t hist = out_len < window ? out_len : window;
      size_t prefix = 0;
      if (hist < window && d->have_dict) {
        size_t needed = window - hist;
        size_t cl = d->dict_len < needed ? d->dict_len : needed;
        memcpy(tmp, d->dict + d->dict_len - cl, cl);
        prefix = cl;
      }
      memcpy(tmp + prefix, d->out.buf + d->out.len - hist, hist);
      prefix += hist;
      sink_t t = {tmp, prefix, prefix + bmax, (int64_t)(prefix + bmax)};
      if ((st = decode_into_sink(payload, bsize, &t))) break;
      if (t.len - prefix > bmax) { st = LZ4REF_CORRUPT_BLOCK_SIZE; break; }
      if ((st = sink_put(&d->out, tmp + prefix, t.len - prefix))) break;
    }
  }
  free(tmp);
  if (st) return st;

  if (content_ck) { /* :306-315 */
    uint32_t expected;
    if ((st = src_u32le(r, &expected))) return st;
    if (lz4ref_xxh32(d->out.buf + frame_start, d->out.len - frame_start, 0) != expected)
      return LZ4REF_CORRUPT_CONTENT_CK;
  }
  if (has_size && (uint64_t)(d->out.len - frame_start) != content_size) /* :317-321 */
    return LZ4REF_CORRUPT_CONTENT_SIZE;
  return LZ4REF_OK;
}

int lz4ref_frame_decode(const uint8_t *src, size_t len, int64_t max_output_bytes,
                        lz4ref_dict_resolver resolver, void *user, uint8_t *dst, size_t cap,
                        size_t *out_len, uint32_t *err_dict_id) {
  fdec_t d;
  memset(&d, 0, sizeof d);
  d.r.p = src;
  d.r.len = len;
  d.out.buf = dst;
  d.out.cap = cap;
  d.out.max_len = max_output_bytes < 0 ? -1 : max_output_bytes;
  d.max_output = max_output_bytes;
  d.resolver = resolver;
  d.user = user;
  int st = LZ4REF_OK;
  while (!src_eof(&d.r)) { /* :45-50 */
    uint32_t magic;
    if (src_remaining(&d.r) < 4) { st = LZ4REF_FORMAT_EOF; break; } /* :53-55 */
    src_u32le(&d.r, &magic);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) { /* :59-62, :78-84 */
      uint32_t size;
      if ((st = src_u32le(&d.r, &size))) break;
      const uint8_t *skip;
      if ((st = src_view(&d.r, size, &skip))) break;
      continue;
    }
    if (magic == 0x184C2102u) { /* :64-68 */
      if ((st = decode_legacy_frame(&d))) break;
      continue;
    }
    if (magic != 0x184D2204u) { st = LZ4REF_FORMAT_MAGIC; break; } /* :70-72 */
    if ((st = decode_standard_frame(&d))) break;
  }
  if (out_len) *out_len = d.out.len;
  if (err_dict_id) *err_dict_id = d.err_dict_id;
  return st;
}

/* ===================================================== host-only frame formats */

static void put32(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

/* frame/lz4_frame_info.dart:89-199 */
int lz4ref_frame_info(const uint8_t *src, size_t len, lz4ref_frame_info_t *out) {
  memset(out, 0, sizeof *out);
  out->block_independence = 1; /* ctor defaults :56-61 */
  src_t r = {src, len, 0};
  if (src_remaining(&r) < 4) return LZ4REF_FORMAT_EOF; /* :92-94 */
  uint32_t magic;
  src_u32le(&r, &magic);
  out->magic = magic;
  if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) { /* :99-111 */
    if (src_remaining(&r) < 4) return LZ4REF_FORMAT_EOF;
    uint32_t size;
    src_u32le(&r, &size);
    out->is_skippable = 1;
    out->skippable_size = size;
    out->header_size = 8;
    return LZ4REF_OK;
  }
  if (magic == 0x184C2102u) { /* :114-121 */
    out->is_legacy = 1;
    out->header_size = 4;
    return LZ4REF_OK;
  }
  if (magic != 0x184D2204u) return LZ4REF_FORMAT_MAGIC; /* :124-126 */
  size_t desc_start = r.off;
  if (src_remaining(&r) < 3) return LZ4REF_FORMAT_EOF; /* :130-132 */
  uint32_t flg = 0, bd = 0;
  src_u8(&r, &flg);
  src_u8(&r, &bd);
  if (((flg >> 6) & 3) != 1) return LZ4REF_UNSUPPORTED_VERSION; /* :137-140 */
  int indep = (flg >> 5) & 1, block_ck = (flg >> 4) & 1, has_size = (flg >> 3) & 1;
  int content_ck = (flg >> 2) & 1, has_dict = flg & 1;
  if ((flg >> 1) & 1) return LZ4REF_FORMAT_FLG_RESERVED; /* :149-151 */
  if (bd & 0x8F) return LZ4REF_FORMAT_BD_RESERVED;       /* :153-155 */
  size_t bmax = block_max_bytes((bd >> 4) & 7);
  if (!bmax) return LZ4REF_FORMAT_BLOCK_MAX; /* :157-158, :201-214 */
  uint64_t content_size = 0;
  if (has_size) { /* :160-166 */
    if (src_remaining(&r) < 8) return LZ4REF_FORMAT_EOF;
    uint32_t lo, hi;
    src_u32le(&r, &lo);
    src_u32le(&r, &hi);
    content_size = ((uint64_t)hi << 32) | lo;
    if (content_size >> 63) return LZ4REF_FORMAT_INT_OVERFLOW; /* byte_reader.dart:74-76 */
  }
  uint32_t dict_id = 0;
  if (has_dict) { /* :168-174 */
    if (src_remaining(&r) < 4) return LZ4REF_FORMAT_EOF;
    src_u32le(&r, &dict_id);
  }
  size_t desc_end = r.off;
  uint32_t hc;
  int st = src_u8(&r, &hc); /* :178 */
  if (st) return st;
  if (hc != ((lz4ref_xxh32(src + desc_start, desc_end - desc_start, 0) >> 8) & 0xff))
    return LZ4REF_CORRUPT_HEADER_CK; /* :181-185 */
  out->block_independence = indep;
  out->block_checksum = block_ck;
  out->content_checksum = content_ck;
  out->has_content_size = has_size;
  out->content_size = content_size;
  out->has_dict_id = has_dict;
  out->dict_id = dict_id;
  out->block_max_size = (uint32_t)bmax;
  out->header_size = (uint32_t)r.off;
  return LZ4REF_OK;
}

/* frame/lz4_frame_encoder.dart:27-46 */
int lz4ref_skippable_frame_encode(const uint8_t *data, size_t n, int index, uint8_t *dst, size_t cap,
                                  size_t *out_len) {
  if (index < 0 || index > 15) return LZ4REF_RANGE;         /* :28-30 */
  if ((uint64_t)n > 0xFFFFFFFFull) return LZ4REF_RANGE;     /* :31-33 */
  sink_t w = {dst, 0, cap, -1};
  int st;
  if ((st = sink_u32le(&w, 0x184D2A50u + (uint32_t)index))) return st;
  if ((st = sink_u32le(&w, (uint32_t)n))) return st;
  if ((st = sink_put(&w, data, n))) return st;
  if (out_len) *out_len = w.len;
  return LZ4REF_OK;
}

size_t lz4ref_legacy_frame_bound(size_t n) {
  const size_t bm = 8u * 1024 * 1024;
  size_t nb = (n + bm - 1) / bm;
  return 4 + nb * (4 + 16 + bm / 255 + 1) + n;
}

/* frame/lz4_frame_encoder.dart:55-89 */
int lz4ref_legacy_frame_encode(const uint8_t *src, size_t n, int64_t acceleration, uint8_t *dst,
                               size_t cap, size_t *out_len) {
  const size_t bm = 8u * 1024 * 1024;
  sink_t w = {dst, 0, cap, -1};
  int st;
  if ((st = sink_u32le(&w, 0x184C2102u))) return st; /* :59 */
  size_t offset = 0;
  while (offset < n) { /* :62 */
    size_t end = offset + bm;
    if (end > n) end = n;
    size_t size_at = w.len;
    if ((st = sink_u32le(&w, 0))) return st;
    size_t start = w.len;
    /* lz4BlockCompress(chunk, acceleration): no dictionary; both branches of :75-83
     * write the compressed form */
    if ((st = fast_encode_to_sink(&w, src + offset, end - offset, NULL, 0, acceleration))) return st;
    put32(w.buf + size_at, (uint32_t)(w.len - start));
    offset = end;
  }
  if (out_len) *out_len = w.len;
  return LZ4REF_OK;
}

/* block/lz4_sized_block.dart:15-31 */
int lz4ref_compress_with_size(const uint8_t *src, size_t n, int level, int64_t param, uint8_t *dst,
                              size_t cap, size_t *out_len) {
  sink_t w = {dst, 0, cap, -1};
  int st;
  if ((st = sink_u32le(&w, (uint32_t)n))) return st; /* setUint32 truncates to 32 bits :28 */
  st = level == 0 ? fast_encode_to_sink(&w, src, n, NULL, 0, param)
                  : hc_encode_to_sink(&w, src, n, NULL, 0, param);
  if (st) return st;
  if (out_len) *out_len = w.len;
  return LZ4REF_OK;
}

/* block/lz4_sized_block.dart:39-55 */
int lz4ref_decompress_with_size(const uint8_t *src, size_t len, uint8_t *dst, size_t cap,
                                size_t *out_len) {
  if (len < 4) return LZ4REF_FORMAT_SIZE_HEADER; /* :40-42 */
  uint32_t size = rd32(src);
  if (size > cap) return LZ4REF_DST_TOO_SMALL;
  int st = lz4ref_block_decompress(src + 4, len - 4, dst, (int64_t)size);
  if (st) return st;
  if (out_len) *out_len = size;
  return LZ4REF_OK;
}

/* ============================================================ stream encoder
 * frame/lz4_frame_stream_encoder.dart:25-236 */
struct lz4ref_stream_encoder {
  lz4ref_frame_options o;
  size_t bmax;
  uint8_t *dict;      /* dictionary argument (NULL = null) */
  size_t dict_len;
  int have_dict;
  uint64_t total_in;
  lz4ref_xxh32_state hasher;
  uint8_t *buf;       /* BytesBuilder contents */
  size_t buffered, buf_cap;
  uint8_t history[64 * 1024];
  size_t history_len;
};

lz4ref_stream_encoder *lz4ref_stream_encoder_new(const lz4ref_frame_options *o, const uint8_t *dict,
                                                 size_t dict_len, int *status) {
  *status = LZ4REF_OK;
  if (o->acceleration < 1 || !block_max_bytes(o->block_size_id) ||
      (o->compression == 1 && o->hc_max_search_depth < 1)) { /* options ctor */
    *status = LZ4REF_RANGE;
    return NULL;
  }
  lz4ref_stream_encoder *e = (lz4ref_stream_encoder *)calloc(1, sizeof *e);
  if (!e) { *status = LZ4REF_DST_TOO_SMALL; return NULL; }
  e->o = *o;
  e->bmax = block_max_bytes(o->block_size_id);
  if (dict) {
    e->have_dict = 1;
    e->dict = (uint8_t *)malloc(dict_len ? dict_len : 1);
    memcpy(e->dict, dict, dict_len);
    e->dict_len = dict_len;
  }
  lz4ref_xxh32_init(&e->hasher, 0);
  const size_t window = 64 * 1024;
  if (e->have_dict && !o->block_independence) { /* :64-73 */
    if (dict_len > window) {
      memcpy(e->history, dict + dict_len - window, window);
      e->history_len = window;
    } else {
      memcpy(e->history, dict, dict_len);
      e->history_len = dict_len;
    }
  }
  return e;
}

void lz4ref_stream_encoder_free(lz4ref_stream_encoder *e) {
  if (!e) return;
  free(e->dict);
  free(e->buf);
  free(e);
}

/* _encodeHeader :238-266 */
int lz4ref_stream_encoder_header(lz4ref_stream_encoder *e, uint8_t *dst, size_t cap, size_t *out_len) {
  const lz4ref_frame_options *o = &e->o;
  sink_t w = {dst, 0, cap, -1};
  int st;
  uint32_t flg = (1u << 6) | ((o->block_independence ? 1u : 0u) << 5) |
                 ((o->block_checksum ? 1u : 0u) << 4) | ((o->has_content_size ? 1u : 0u) << 3) |
                 ((o->content_checksum ? 1u : 0u) << 2) | (o->has_dict_id ? 1u : 0u); /* :32-37 */
  uint32_t bd = ((uint32_t)o->block_size_id & 7u) << 4;
  if ((st = sink_u32le(&w, 0x184D2204u))) return st;
  if ((st = sink_u8(&w, flg))) return st;
  if ((st = sink_u8(&w, bd))) return st;
  if (o->has_content_size) {
    if ((st = sink_u32le(&w, (uint32_t)(o->content_size & 0xFFFFFFFFu)))) return st;
    if ((st = sink_u32le(&w, (uint32_t)(o->content_size >> 32)))) return st;
  }
  if (o->has_dict_id && (st = sink_u32le(&w, o->dict_id))) return st;
  uint32_t hc = (lz4ref_xxh32(w.buf + 4, w.len - 4, 0) >> 8) & 0xff;
  if ((st = sink_u8(&w, hc))) return st;
  *out_len = w.len;
  return LZ4REF_OK;
}

static void senc_append_history(lz4ref_stream_encoder *e, const uint8_t *bytes, size_t n) { /* :75-96 */
  const size_t window = 64 * 1024;
  if (n >= window) {
    memcpy(e->history, bytes + n - window, window);
    e->history_len = window;
    return;
  }
  size_t required = e->history_len + n;
  if (required <= window) {
    memcpy(e->history + e->history_len, bytes, n);
    e->history_len = required;
    return;
  }
  size_t drop = required - window;
  size_t keep = e->history_len - drop;
  memmove(e->history, e->history + drop, keep);
  e->history_len = keep;
  memcpy(e->history + e->history_len, bytes, n);
  e->history_len += n;
}

/* encodeBlock :108-156, appended to w as one piece */
static int senc_encode_block(lz4ref_stream_encoder *e, sink_t *w, const uint8_t *chunk, size_t n) {
  const lz4ref_frame_options *o = &e->o;
  const uint8_t *d = NULL;
  size_t dl = 0;
  if (o->block_independence) { /* :187-188 */
    if (e->have_dict) { d = e->dict; dl = e->dict_len; }
  } else if (e->history_len != 0) { /* :190-192 */
    d = e->history;
    dl = e->history_len;
  }
  int st;
  size_t start = w->len;
  if ((st = sink_u32le(w, 0))) return st;
  size_t payload = w->len;
  st = o->compression == 0
           ? fast_encode_to_sink_ex(w, chunk, n, d, dl, o->acceleration, o->spec_conformant)
           : hc_encode_to_sink_ex(w, chunk, n, d, dl, o->hc_max_search_depth, o->spec_conformant);
  if (st) return st;
  size_t clen = w->len - payload;
  if (clen < n) { /* :136-144 */
    put32(w->buf + start, (uint32_t)clen);
    if (o->block_checksum && (st = sink_u32le(w, lz4ref_xxh32(w->buf + payload, clen, 0)))) return st;
  } else { /* :145-155 */
    w->len = start;
    if ((st = sink_u32le(w, 0x80000000u | (uint32_t)n))) return st;
    if ((st = sink_put(w, chunk, n))) return st;
    if (o->block_checksum && (st = sink_u32le(w, lz4ref_xxh32(chunk, n, 0)))) return st;
  }
  return LZ4REF_OK;
}

static int piece_done(sink_t *w, size_t *mark, uint32_t *lens, size_t cap, size_t *np) {
  if (*np >= cap) return LZ4REF_DST_TOO_SMALL;
  lens[(*np)++] = (uint32_t)(w->len - *mark);
  *mark = w->len;
  return LZ4REF_OK;
}

/* body of `await for (final chunk in input)` :158-201 */
int lz4ref_stream_encoder_add(lz4ref_stream_encoder *e, const uint8_t *chunk, size_t n, uint8_t *dst,
                              size_t cap, size_t *out_len, uint32_t *piece_lens, size_t piece_cap,
                              size_t *npieces) {
  *out_len = 0;
  *npieces = 0;
  if (n == 0) return LZ4REF_OK; /* :160-162 */
  e->total_in += n;
  if (e->o.has_content_size && e->total_in > e->o.content_size) /* :165-167 */
    return LZ4REF_FORMAT_CONTENT_SIZE_STREAM;
  if (e->o.content_checksum) lz4ref_xxh32_update(&e->hasher, chunk, n); /* :168-170 */
  if (e->buffered + n > e->buf_cap) { /* buf.add(bytes) :172 */
    size_t nc = (e->buffered + n) * 2 + 64;
    e->buf = (uint8_t *)realloc(e->buf, nc);
    e->buf_cap = nc;
  }
  memcpy(e->buf + e->buffered, chunk, n);
  e->buffered += n;
  sink_t w = {dst, 0, cap, -1};
  size_t mark = 0;
  int st = LZ4REF_OK;
  size_t taken = 0;
  while (e->buffered - taken >= e->bmax) { /* :175-200 */
    const uint8_t *block = e->buf + taken;
    if ((st = senc_encode_block(e, &w, block, e->bmax))) break;
    if ((st = piece_done(&w, &mark, piece_lens, piece_cap, npieces))) break;
    if (!e->o.block_independence) senc_append_history(e, block, e->bmax);
    taken += e->bmax;
  }
  memmove(e->buf, e->buf + taken, e->buffered - taken);
  e->buffered -= taken;
  *out_len = mark;
  return st;
}

/* after the loop :203-234 */
int lz4ref_stream_encoder_close(lz4ref_stream_encoder *e, uint8_t *dst, size_t cap, size_t *out_len,
                                uint32_t *piece_lens, size_t piece_cap, size_t *npieces) {
  *out_len = 0;
  *npieces = 0;
  sink_t w = {dst, 0, cap, -1};
  size_t mark = 0;
  int st;
  if (e->buffered != 0) { /* :203-222 */
    if ((st = senc_encode_block(e, &w, e->buf, e->buffered))) return st;
    if ((st = piece_done(&w, &mark, piece_lens, piece_cap, npieces))) return st;
    if (!e->o.block_independence) senc_append_history(e, e->buf, e->buffered);
    e->buffered = 0;
  }
  *out_len = mark;
  if (e->o.has_content_size && e->total_in != e->o.content_size) /* :224-226 */
    return LZ4REF_FORMAT_CONTENT_SIZE_STREAM;
  if ((st = sink_u32le(&w, 0))) return st; /* :228 */
  if ((st = piece_done(&w, &mark, piece_lens, piece_cap, npieces))) return st;
  if (e->o.content_checksum) { /* :230-234 */
    if ((st = sink_u32le(&w, lz4ref_xxh32_digest(&e->hasher)))) return st;
    if ((st = piece_done(&w, &mark, piece_lens, piece_cap, npieces))) return st;
  }
  *out_len = mark;
  return LZ4REF_OK;
}

/* ============================================================ stream decoder
 * frame/lz4_frame_stream_decoder.dart:62-599 */
enum {
  SD_MAGIC, SD_SKIP_SIZE, SD_SKIP_SKIP, SD_DESCRIPTOR, SD_HEADER_CK, SD_BLOCK_SIZE, SD_BLOCK_DATA,
  SD_BLOCK_CK, SD_CONTENT_CK, SD_LEGACY_SIZE, SD_LEGACY_DATA, SD_LEGACY_AFTER
};

struct lz4ref_stream_decoder {
  /* _ChunkBuffer :601-711 */
  uint8_t *buf;
  size_t start, end, cap;
  int64_t max_output; /* <0 = null */
  lz4ref_dict_resolver resolver;
  void *user;
  int have_resolver;
  int finished;
  int state;
  uint64_t skippable_remaining;
  uint8_t descriptor[16];
  size_t descriptor_len;
  int indep, block_ck, has_size, content_ck, has_dict_id;
  size_t bmax;
  uint64_t content_size;
  uint32_t dict_id;
  int have_dict; /* _dict != null */
  const uint8_t *dict;
  size_t dict_len;
  uint32_t cur_size;
  int cur_raw;
  uint8_t *cur_data; /* _currentBlockData */
  uint8_t history[64 * 1024];
  size_t history_len;
  lz4ref_xxh32_state hasher;
  int hashing;
  uint64_t frame_produced, total_produced;
  uint32_t legacy_csize;
  uint8_t *legacy_pending;
  int legacy_has_pending;
  size_t legacy_pending_len;
  uint8_t *yielded; /* block handed to the caller */
  uint8_t *work;    /* prefix + block scratch */
  size_t work_cap;
};

lz4ref_stream_decoder *lz4ref_stream_decoder_new(int64_t max_output_bytes, lz4ref_dict_resolver resolver,
                                                 void *user, int have_resolver) {
  lz4ref_stream_decoder *d = (lz4ref_stream_decoder *)calloc(1, sizeof *d);
  if (!d) return NULL;
  d->max_output = max_output_bytes < 0 ? -1 : max_output_bytes;
  d->resolver = resolver;
  d->user = user;
  d->have_resolver = have_resolver;
  d->indep = 1;
  d->state = SD_MAGIC;
  return d;
}

void lz4ref_stream_decoder_free(lz4ref_stream_decoder *d) {
  if (!d) return;
  free(d->buf);
  free(d->cur_data);
  free(d->legacy_pending);
  free(d->yielded);
  free(d->work);
  free(d);
}

static size_t sd_len(const lz4ref_stream_decoder *d) { return d->end - d->start; }
static int sd_has(const lz4ref_stream_decoder *d, uint64_t n) { return sd_len(d) >= n; }
static uint32_t sd_peek32(const lz4ref_stream_decoder *d) { return rd32(d->buf + d->start); }
static uint32_t sd_read32(lz4ref_stream_decoder *d) {
  uint32_t v = rd32(d->buf + d->start);
  d->start += 4;
  return v;
}

int lz4ref_stream_decoder_add(lz4ref_stream_decoder *d, const uint8_t *chunk, size_t n) { /* :615-622 */
  if (n == 0) return LZ4REF_OK;
  size_t remaining = sd_len(d);
  if (d->start != 0) {
    memmove(d->buf, d->buf + d->start, remaining);
    d->start = 0;
    d->end = remaining;
  }
  if (d->end + n > d->cap) {
    size_t nc = d->cap ? d->cap : 1024;
    while (nc < d->end + n) nc *= 2;
    d->buf = (uint8_t *)realloc(d->buf, nc);
    if (!d->buf) return LZ4REF_DST_TOO_SMALL;
    d->cap = nc;
  }
  memcpy(d->buf + d->end, chunk, n);
  d->end += n;
  return LZ4REF_OK;
}

static void sd_reset_frame(lz4ref_stream_decoder *d) { /* :568-598 */
  d->skippable_remaining = 0;
  d->descriptor_len = 0;
  d->indep = 1;
  d->block_ck = d->has_size = d->content_ck = d->has_dict_id = 0;
  d->bmax = 0;
  d->content_size = 0;
  d->dict_id = 0;
  d->have_dict = 0;
  d->dict = NULL;
  d->dict_len = 0;
  d->cur_size = 0;
  d->cur_raw = 0;
  free(d->cur_data);
  d->cur_data = NULL;
  d->history_len = 0;
  d->hashing = 0;
  d->frame_produced = 0;
  d->legacy_csize = 0;
  free(d->legacy_pending);
  d->legacy_pending = NULL;
  d->legacy_has_pending = 0;
  d->legacy_pending_len = 0;
}

static void sd_append_history(lz4ref_stream_decoder *d, const uint8_t *bytes, size_t n) { /* :534-559 */
  const size_t window = 64 * 1024;
  if (n >= window) {
    memcpy(d->history, bytes + n - window, window);
    d->history_len = window;
    return;
  }
  size_t required = d->history_len + n;
  if (required <= window) {
    memcpy(d->history + d->history_len, bytes, n);
    d->history_len = required;
    return;
  }
  size_t drop = required - window;
  size_t keep = d->history_len - drop;
  memmove(d->history, d->history + drop, keep);
  d->history_len -= drop;
  memcpy(d->history + d->history_len, bytes, n);
  d->history_len += n;
}

static int sd_finish_frame(lz4ref_stream_decoder *d) { /* :561-566 */
  if (d->has_size && d->frame_produced != d->content_size) return LZ4REF_CORRUPT_CONTENT_SIZE;
  return LZ4REF_OK;
}

static int sd_work(lz4ref_stream_decoder *d, size_t need) {
  if (need <= d->work_cap) return LZ4REF_OK;
  free(d->work);
  d->work = (uint8_t *)malloc(need);
  d->work_cap = d->work ? need : 0;
  return d->work ? LZ4REF_OK : LZ4REF_DST_TOO_SMALL;
}

static int sd_yield(lz4ref_stream_decoder *d, const uint8_t *p, size_t n, const uint8_t **out,
                    size_t *out_len) {
  free(d->yielded);
  d->yielded = (uint8_t *)malloc(n ? n : 1);
  if (!d->yielded) return LZ4REF_DST_TOO_SMALL;
  memcpy(d->yielded, p, n);
  *out = d->yielded;
  *out_len = n;
  return LZ4REF_OK;
}

/* _decodeCurrentBlockAndUpdateState :428-532; *n_out = 0 stands for the empty list */
static int sd_decode_current(lz4ref_stream_decoder *d, const uint8_t **out, size_t *n_out) {
  const size_t window = 64 * 1024;
  const uint8_t *produced;
  size_t produced_len;
  int st;
  if (d->cur_raw) { /* :432-433 */
    produced = d->cur_data;
    produced_len = d->cur_size;
  } else if (d->indep) { /* :435-457 */
    size_t eff = 0;
    if (d->have_dict) eff = d->dict_len > window ? window : d->dict_len;
    if ((st = sd_work(d, eff + d->bmax + 1))) return st;
    if (eff) memcpy(d->work, d->dict + d->dict_len - eff, eff);
    sink_t t = {d->work, eff, eff + d->bmax, (int64_t)(eff + d->bmax)};
    if ((st = decode_into_sink(d->cur_data, d->cur_size, &t))) return st;
    produced = d->work + eff;
    produced_len = t.len - eff;
  } else { /* :458-506 */
    size_t hl = d->history_len;
    size_t prefix = 0;
    if ((st = sd_work(d, window + d->bmax + 1))) return st;
    if (hl < window && d->have_dict) { /* :466-491 */
      size_t needed = window - hl;
      size_t cl = d->dict_len < needed ? d->dict_len : needed;
      memcpy(d->work, d->dict + d->dict_len - cl, cl);
      prefix = cl;
    }
    if (hl) memcpy(d->work + prefix, d->history, hl);
    prefix += hl;
    sink_t t = {d->work, prefix, prefix + d->bmax, (int64_t)(prefix + d->bmax)};
    if ((st = decode_into_sink(d->cur_data, d->cur_size, &t))) return st;
    if (t.len - prefix > d->bmax) return LZ4REF_CORRUPT_BLOCK_SIZE;
    produced = d->work + prefix;
    produced_len = t.len - prefix;
  }
  d->state = SD_BLOCK_SIZE; /* :510-511 */
  *n_out = 0;
  if (produced_len == 0) { /* :513-515 */
    free(d->cur_data);
    d->cur_data = NULL;
    return LZ4REF_OK;
  }
  if (d->max_output >= 0 && d->total_produced + produced_len > (uint64_t)d->max_output) /* :517-520 */
    return LZ4REF_LIMIT;
  d->total_produced += produced_len;
  d->frame_produced += produced_len;
  if (d->hashing) lz4ref_xxh32_update(&d->hasher, produced, produced_len); /* :525 */
  if (!d->indep) sd_append_history(d, produced, produced_len);            /* :527-529 */
  st = sd_yield(d, produced, produced_len, out, n_out);
  free(d->cur_data);
  d->cur_data = NULL;
  return st;
}

/* decodeNext :113-413 */
int lz4ref_stream_decoder_next(lz4ref_stream_decoder *d, int *has_output, const uint8_t **out,
                               size_t *out_len, uint32_t *err_dict_id) {
  const size_t legacy_max = 8u * 1024 * 1024;
  int st;
  *has_output = 0;
  *out = NULL;
  *out_len = 0;
  for (;;) {
    switch (d->state) {
      case SD_MAGIC: { /* :116-135 */
        if (!sd_has(d, 4)) return LZ4REF_OK;
        uint32_t magic = sd_read32(d);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) { d->state = SD_SKIP_SIZE; continue; }
        if (magic == 0x184C2102u) { sd_reset_frame(d); d->state = SD_LEGACY_SIZE; continue; }
        if (magic != 0x184D2204u) return LZ4REF_FORMAT_MAGIC;
        sd_reset_frame(d);
        d->state = SD_DESCRIPTOR;
        continue;
      }
      case SD_SKIP_SIZE: /* :137-143 */
        if (!sd_has(d, 4)) return LZ4REF_OK;
        d->skippable_remaining = sd_read32(d);
        d->state = SD_SKIP_SKIP;
        continue;
      case SD_SKIP_SKIP: { /* :145-156 */
        if (d->skippable_remaining == 0) { d->state = SD_MAGIC; continue; }
        size_t avail = sd_len(d);
        size_t skipped = avail < d->skippable_remaining ? avail : (size_t)d->skippable_remaining;
        d->start += skipped;
        d->skippable_remaining -= skipped;
        if (d->skippable_remaining != 0) return LZ4REF_OK;
        d->state = SD_MAGIC;
        continue;
      }
      case SD_DESCRIPTOR: { /* :158-223 */
        if (!sd_has(d, 2)) return LZ4REF_OK;
        uint32_t flg0 = d->buf[d->start];
        size_t required = 2 + (((flg0 >> 3) & 1) ? 8 : 0) + ((flg0 & 1) ? 4 : 0);
        if (!sd_has(d, required)) return LZ4REF_OK;
        uint32_t flg = d->buf[d->start++], bd = d->buf[d->start++];
        d->descriptor[0] = (uint8_t)flg;
        d->descriptor[1] = (uint8_t)bd;
        d->descriptor_len = 2;
        if (((flg >> 6) & 3) != 1) return LZ4REF_UNSUPPORTED_VERSION;
        d->indep = (flg >> 5) & 1;
        d->block_ck = (flg >> 4) & 1;
        d->has_size = (flg >> 3) & 1;
        d->content_ck = (flg >> 2) & 1;
        d->has_dict_id = flg & 1;
        if ((flg >> 1) & 1) return LZ4REF_FORMAT_FLG_RESERVED;
        if (bd & 0x8F) return LZ4REF_FORMAT_BD_RESERVED;
        d->bmax = block_max_bytes((bd >> 4) & 7);
        if (!d->bmax) return LZ4REF_FORMAT_BLOCK_MAX;
        if (d->has_size) { /* :202-214 */
          memcpy(d->descriptor + d->descriptor_len, d->buf + d->start, 8);
          d->descriptor_len += 8;
          uint32_t lo = sd_read32(d), hi = sd_read32(d);
          uint64_t val = ((uint64_t)hi << 32) + lo;
          if (val >> 63) return LZ4REF_UNSUPPORTED_CONTENT_SIZE; /* val < 0 */
          d->content_size = val;
        }
        if (d->has_dict_id) { /* :216-220 */
          memcpy(d->descriptor + d->descriptor_len, d->buf + d->start, 4);
          d->descriptor_len += 4;
          d->dict_id = sd_read32(d);
        }
        d->state = SD_HEADER_CK;
        continue;
      }
      case SD_HEADER_CK: { /* :225-253 */
        if (!sd_has(d, 1)) return LZ4REF_OK;
        uint32_t hc = d->buf[d->start++];
        if (hc != ((lz4ref_xxh32(d->descriptor, d->descriptor_len, 0) >> 8) & 0xff))
          return LZ4REF_CORRUPT_HEADER_CK;
        if (d->has_dict_id) {
          if (!d->have_resolver) return LZ4REF_UNSUPPORTED_DICT;
          const uint8_t *dp = NULL;
          size_t dl = 0;
          if (!d->resolver(d->user, d->dict_id, &dp, &dl)) {
            if (err_dict_id) *err_dict_id = d->dict_id;
            return LZ4REF_DICT_NOT_FOUND;
          }
          d->have_dict = 1;
          d->dict = dp;
          d->dict_len = dl;
        }
        d->hashing = d->content_ck;
        if (d->hashing) lz4ref_xxh32_init(&d->hasher, 0);
        d->state = SD_BLOCK_SIZE;
        continue;
      }
      case SD_BLOCK_SIZE: { /* :255-278 */
        if (!sd_has(d, 4)) return LZ4REF_OK;
        uint32_t raw = sd_read32(d);
        if (raw == 0) {
          if (d->content_ck) { d->state = SD_CONTENT_CK; continue; }
          if ((st = sd_finish_frame(d))) return st;
          d->state = SD_MAGIC;
          continue;
        }
        d->cur_raw = (raw & 0x80000000u) != 0;
        d->cur_size = raw & 0x7FFFFFFFu;
        if (d->cur_size > d->bmax) return LZ4REF_CORRUPT_BLOCK_SIZE;
        d->state = SD_BLOCK_DATA;
        continue;
      }
      case SD_BLOCK_DATA: { /* :280-295 */
        if (!sd_has(d, d->cur_size)) return LZ4REF_OK;
        free(d->cur_data);
        d->cur_data = (uint8_t *)malloc(d->cur_size ? d->cur_size : 1);
        if (!d->cur_data) return LZ4REF_DST_TOO_SMALL;
        memcpy(d->cur_data, d->buf + d->start, d->cur_size);
        d->start += d->cur_size;
        if (d->block_ck) { d->state = SD_BLOCK_CK; continue; }
        if ((st = sd_decode_current(d, out, out_len))) return st;
        if (*out_len) { *has_output = 1; return LZ4REF_OK; }
        continue;
      }
      case SD_BLOCK_CK: { /* :297-312 */
        if (!sd_has(d, 4)) return LZ4REF_OK;
        uint32_t expected = sd_read32(d);
        if (lz4ref_xxh32(d->cur_data, d->cur_size, 0) != expected) return LZ4REF_CORRUPT_BLOCK_CK;
        if ((st = sd_decode_current(d, out, out_len))) return st;
        if (*out_len) { *has_output = 1; return LZ4REF_OK; }
        continue;
      }
      case SD_CONTENT_CK: { /* :314-326 */
        if (!sd_has(d, 4)) return LZ4REF_OK;
        uint32_t expected = sd_read32(d);
        uint32_t actual = d->hashing ? lz4ref_xxh32_digest(&d->hasher) : 0;
        if (actual != expected) return LZ4REF_CORRUPT_CONTENT_CK;
        if ((st = sd_finish_frame(d))) return st;
        d->state = SD_MAGIC;
        continue;
      }
      case SD_LEGACY_SIZE: { /* :328-354 */
        if (d->legacy_has_pending) { d->state = SD_LEGACY_AFTER; continue; }
        if (!sd_has(d, 4)) {
          if (d->finished) { d->state = SD_MAGIC; continue; }
          return LZ4REF_OK;
        }
        if (is_legacy_boundary(sd_peek32(d))) { d->state = SD_MAGIC; continue; }
        uint32_t csize = sd_read32(d);
        if (csize == 0) return LZ4REF_CORRUPT_LEGACY_SIZE;
        d->legacy_csize = csize;
        d->state = SD_LEGACY_DATA;
        continue;
      }
      case SD_LEGACY_DATA: { /* :356-376 */
        if (!sd_has(d, d->legacy_csize)) return LZ4REF_OK;
        const uint8_t *block = d->buf + d->start;
        d->start += d->legacy_csize;
        free(d->legacy_pending);
        d->legacy_pending = (uint8_t *)malloc(legacy_max);
        if (!d->legacy_pending) return LZ4REF_DST_TOO_SMALL;
        sink_t t = {d->legacy_pending, 0, legacy_max, (int64_t)legacy_max};
        if ((st = decode_into_sink(block, d->legacy_csize, &t))) return st;
        if (d->max_output >= 0 && d->total_produced + t.len > (uint64_t)d->max_output) return LZ4REF_LIMIT;
        d->total_produced += t.len;
        d->legacy_has_pending = 1;
        d->legacy_pending_len = t.len;
        d->state = SD_LEGACY_AFTER;
        continue;
      }
      case SD_LEGACY_AFTER: { /* :378-410 */
        if (!d->legacy_has_pending) { d->state = SD_LEGACY_SIZE; continue; }
        int next_state;
        if (!sd_has(d, 4)) {
          if (!d->finished) return LZ4REF_OK;
          next_state = SD_MAGIC;
        } else if (is_legacy_boundary(sd_peek32(d))) {
          next_state = SD_MAGIC;
        } else {
          if (d->legacy_pending_len != legacy_max) return LZ4REF_CORRUPT_LEGACY_NOT_FULL;
          next_state = SD_LEGACY_SIZE;
        }
        if ((st = sd_yield(d, d->legacy_pending, d->legacy_pending_len, out, out_len))) return st;
        free(d->legacy_pending);
        d->legacy_pending = NULL;
        d->legacy_has_pending = 0;
        d->legacy_pending_len = 0;
        d->state = next_state;
        *has_output = 1; /* the pending list is yielded even when empty */
        return LZ4REF_OK;
      }
    }
  }
}

int lz4ref_stream_decoder_finish(lz4ref_stream_decoder *d) { /* :415-426 */
  d->finished = 1;
  if (sd_len(d) == 0 && d->state == SD_MAGIC) return LZ4REF_OK;
  if (d->state == SD_LEGACY_AFTER && d->legacy_has_pending) {
    if (sd_len(d) == 0) return LZ4REF_OK;
  }
  return LZ4REF_FORMAT_EOF;
}

/* Not a reference function: tells whether a compressed block obeys the LZ4 block format's
 * end-of-block rules (last five bytes literals, last match starts at least twelve bytes before the
 * end).  The reference's encoders do not (lib/src/block/lz4_block_encoder.dart:95,118: the search
 * runs to totalLength-4 and matches extend to the last byte); the checker counts such blocks. */
int lz4ref_block_conformant(const uint8_t *src, size_t c) {
  size_t ip = 0, nmatch = 0, trailing = 0;
  uint64_t op = 0, last_start = 0;
  while (ip < c) {
    unsigned tok = src[ip++];
    size_t ll = tok >> 4;
    if (ll == 15) {
      while (ip < c) {
        unsigned x = src[ip++];
        ll += x;
        if (x != 255) break;
      }
    }
    ip += ll;
    op += ll;
    trailing = ll;
    if (ip + 2 > c) break;
    ip += 2;
    size_t ml = (tok & 15u) + 4u;
    if ((tok & 15u) == 15u) {
      while (ip < c) {
        unsigned x = src[ip++];
        ml += x;
        if (x != 255) break;
      }
    }
    last_start = op;
    op += ml;
    nmatch++;
    trailing = 0;
  }
  return !(nmatch && (trailing < 5 || last_start + 12 > op));
}

/* Harness helpers (no reference counterpart): the per-block calls above over a run of equal-sized blocks, so that
 * a host thread spends its time in the codec and not in the caller's per-call overhead.  Block b reads
 * src + b * block and writes comp + b * comp_stride; sizes[b] receives / gives its compressed length. */
int lz4ref_blocks_compress(const uint8_t *src, size_t nblocks, size_t block, const uint8_t *dict, size_t dict_len,
                           int level, int64_t param, uint8_t *comp, size_t comp_stride, int64_t *sizes) {
  for (size_t b = 0; b < nblocks; b++) {
    size_t n = 0;
    int st = lz4ref_block_compress_ex(src + b * block, block, dict, dict_len, level, param, 0, comp + b * comp_stride,
                                      comp_stride, &n);
    if (st) return st;
    sizes[b] = (int64_t)n;
  }
  return LZ4REF_OK;
}

/* dict_len == 0: strict decode of every block to exactly `block` bytes (lz4BlockDecompress).  Otherwise the way the
 * frame decoder takes a block behind a dictionary (lz4_frame_decoder.dart:273-290): lenient decode behind the
 * prefix in `tmp` (dict_len + block bytes, prefix already in place), then the block's bytes are copied out. */
int lz4ref_blocks_decompress(const uint8_t *comp, size_t comp_stride, const int64_t *sizes, size_t nblocks, size_t block,
                             uint8_t *tmp, size_t dict_len, uint8_t *out) {
  for (size_t b = 0; b < nblocks; b++) {
    if (!dict_len) {
      int st = lz4ref_block_decompress(comp + b * comp_stride, (size_t)sizes[b], out + b * block, (int64_t)block);
      if (st) return st;
    } else {
      size_t n = 0;
      int st = lz4ref_block_decompress_into(comp + b * comp_stride, (size_t)sizes[b], tmp, dict_len, dict_len + block, &n);
      if (st) return st;
      if (n != dict_len + block) return LZ4REF_CORRUPT_CONTENT_SIZE;
      memcpy(out + b * block, tmp + dict_len, block);
    }
  }
  return LZ4REF_OK;
}
