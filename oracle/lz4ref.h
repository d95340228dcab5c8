/*
 * lz4ref — CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * A scalar C restatement of the dart_lz4 1.0.0 block/frame hot path, written
 * from the reference's behaviour (citations are path:line under the reference
 * checkout, i.e. lib/src/...).  It exists to CHECK the HIP product path; the
 * product never links, loads or calls anything in this directory.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity pinning: the reference cannot be executed in the build container (no
 * Dart SDK), so this restatement is pinned by the reference's own offline
 * vectors: the block-decoder byte vectors, the xxHash32 sanity values, the
 * seven lz4-CLI golden frames, the frame error matrix, and the encoder size
 * known-answers from benchmark/baseline/macos.txt.  Encoder *byte* parity has
 * no golden vector in the reference; it rests on those size known-answers plus
 * hand traces.
 */
#ifndef LZ4REF_H
#define LZ4REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes: one per distinct (exception type, message) pair thrown on the
 * path.  lib/src/internal/lz4_exception.dart:2-35 gives the taxonomy. */
enum {
  LZ4REF_OK = 0,
  /* Lz4FormatException */
  LZ4REF_FORMAT_EOF = 1,            /* 'Unexpected end of input'   byte_reader.dart:104 */
  LZ4REF_FORMAT_MAGIC = 2,          /* 'Invalid LZ4 frame magic number' lz4_frame_decoder.dart:71 */
  LZ4REF_FORMAT_FLG_RESERVED = 3,   /* 'Reserved FLG bit is set'   :189 */
  LZ4REF_FORMAT_BD_RESERVED = 4,    /* 'Reserved BD bits are set'  :193 */
  LZ4REF_FORMAT_BLOCK_MAX = 5,      /* 'Invalid block maximum size' :396 */
  LZ4REF_FORMAT_INT_OVERFLOW = 6,   /* 'Integer overflow'          byte_reader.dart:75 */
  LZ4REF_FORMAT_CONTENT_SIZE_ARG = 7, /* 'contentSize does not match src length' lz4_frame_encoder.dart:270 */
  LZ4REF_FORMAT_CONTENT_SIZE_STREAM = 8, /* 'contentSize does not match stream length' lz4_frame_stream_encoder.dart:166,225 */
  LZ4REF_FORMAT_SIZE_HEADER = 9,    /* 'Input too short for size header' block/lz4_sized_block.dart:41 */
  /* Lz4CorruptDataException */
  LZ4REF_CORRUPT_LITERAL = 10,      /* 'Literal length exceeds output size' lz4_block_decoder.dart:34 */
  LZ4REF_CORRUPT_MATCH = 11,        /* 'Match length exceeds output size'   :62 */
  LZ4REF_CORRUPT_TRAILING = 12,     /* 'Trailing bytes after end of block'  :44,73 */
  LZ4REF_CORRUPT_DISTANCE = 13,     /* 'Invalid match distance'    byte_writer.dart:98 */
  LZ4REF_CORRUPT_HEADER_CK = 14,    /* 'Header checksum mismatch'  lz4_frame_decoder.dart:222 */
  LZ4REF_CORRUPT_BLOCK_SIZE = 15,   /* 'Block size exceeds maximum' :252,377 */
  LZ4REF_CORRUPT_BLOCK_CK = 16,     /* 'Block checksum mismatch'   :264 */
  LZ4REF_CORRUPT_CONTENT_CK = 17,   /* 'Content checksum mismatch' :313 */
  LZ4REF_CORRUPT_CONTENT_SIZE = 18, /* 'Content size mismatch'     :319 */
  LZ4REF_CORRUPT_LEGACY_SIZE = 19,  /* 'Invalid legacy block size' :112 */
  LZ4REF_CORRUPT_LEGACY_NOT_FULL = 20, /* 'Legacy block is not full' :131 */
  /* Lz4OutputLimitException */
  LZ4REF_LIMIT = 30,                /* 'Output limit exceeded'     byte_writer.dart:131 */
  /* Lz4UnsupportedFeatureException */
  LZ4REF_UNSUPPORTED_VERSION = 40,  /* 'Unsupported LZ4 frame version' lz4_frame_decoder.dart:177 */
  LZ4REF_UNSUPPORTED_DICT = 41,     /* 'Dictionary ID present but no dictionary resolver provided' :228 */
  LZ4REF_UNSUPPORTED_CONTENT_SIZE = 42, /* 'Content size exceeds supported integer range' lz4_frame_stream_decoder.dart:210 */
  /* base Lz4Exception */
  LZ4REF_DICT_NOT_FOUND = 50,       /* 'Dictionary not found for ID: <id>' :233 */
  /* Dart RangeError (argument validation) */
  LZ4REF_RANGE = 60,
  /* not a reference error: caller's destination buffer too small */
  LZ4REF_DST_TOO_SMALL = 71
};

const char *lz4ref_status_type(int status);    /* Dart exception class name */
const char *lz4ref_status_message(int status); /* Dart message (sans dict id) */

size_t lz4ref_compress_bound(size_t n);

int lz4ref_block_compress_fast(const uint8_t *src, size_t n,
                               const uint8_t *dict, size_t dict_len,
                               int64_t acceleration,
                               uint8_t *dst, size_t cap, size_t *out_len);

int lz4ref_block_compress_hc(const uint8_t *src, size_t n,
                             const uint8_t *dict, size_t dict_len,
                             int64_t max_search_depth,
                             uint8_t *dst, size_t cap, size_t *out_len);

/* level 0 fast (param = acceleration) / 1 hc (param = maxSearchDepth); conformant != 0 applies
 * the lz4 block format's end-of-block rules (no counterpart in the reference, which ends
 * blocks inside matches): no match starts in the last 12 bytes, the last 5 bytes are
 * literals, blocks under 13 bytes are literals only. */
int lz4ref_block_compress_ex(const uint8_t *src, size_t n, const uint8_t *dict, size_t dict_len,
                             int level, int64_t param, int conformant,
                             uint8_t *dst, size_t cap, size_t *out_len);

/* strict: lz4BlockDecompress (lz4_block_decoder.dart:7-77) */
int lz4ref_block_decompress(const uint8_t *src, size_t c,
                            uint8_t *dst, int64_t decompressed_size);

/* lenient: lz4BlockDecompressInto (:79-119). dst[0..prefix_len) is the
 * pre-filled history; max_len is the ByteWriter maxLength (counted from dst[0]). */
int lz4ref_block_decompress_into(const uint8_t *src, size_t c,
                                 uint8_t *dst, size_t prefix_len, size_t max_len,
                                 size_t *out_len);

uint32_t lz4ref_xxh32(const uint8_t *p, size_t len, uint32_t seed);

typedef struct {
  uint32_t seed, v1, v2, v3, v4;
  uint64_t total_len;
  uint8_t mem[16];
  uint32_t mem_size;
} lz4ref_xxh32_state;

void lz4ref_xxh32_init(lz4ref_xxh32_state *s, uint32_t seed);
void lz4ref_xxh32_update(lz4ref_xxh32_state *s, const uint8_t *p, size_t len);
uint32_t lz4ref_xxh32_digest(const lz4ref_xxh32_state *s);

/* Mirrors Lz4FrameOptions (lz4_frame_options.dart:38-101). */
typedef struct {
  int32_t block_size_id;      /* 4..7  (k64KB,k256KB,k1MB,k4MB) */
  int32_t block_independence; /* bool */
  int32_t block_checksum;     /* bool */
  int32_t content_checksum;   /* bool */
  int32_t has_content_size;   /* bool */
  int32_t has_dict_id;        /* bool */
  int32_t compression;        /* 0 fast, 1 hc */
  int32_t acceleration;
  int32_t hc_max_search_depth;
  uint32_t dict_id;
  uint64_t content_size;
  int32_t spec_conformant;    /* not in the reference: blocks obey the lz4 end-of-block rules */
  int32_t reserved;
} lz4ref_frame_options;

void lz4ref_frame_options_default(lz4ref_frame_options *o);
size_t lz4ref_frame_bound(size_t n, const lz4ref_frame_options *o);

int lz4ref_frame_encode(const uint8_t *src, size_t n,
                        const lz4ref_frame_options *o,
                        const uint8_t *dict, size_t dict_len,
                        uint8_t *dst, size_t cap, size_t *out_len);

/* resolver: returns 1 and sets *dict / *dict_len when found, 0 for "null". */
typedef int (*lz4ref_dict_resolver)(void *user, uint32_t dict_id,
                                    const uint8_t **dict, size_t *dict_len);

/* lz4FrameDecodeBytes (lz4_frame_decoder.dart:15-50): all concatenated frames.
 * max_output_bytes < 0 means "null".  *err_dict_id receives the id for
 * LZ4REF_DICT_NOT_FOUND. */
int lz4ref_frame_decode(const uint8_t *src, size_t len,
                        int64_t max_output_bytes,
                        lz4ref_dict_resolver resolver, void *user,
                        uint8_t *dst, size_t cap, size_t *out_len,
                        uint32_t *err_dict_id);

/* ---- host-only formats ---- */

/* Lz4FrameInfo (frame/lz4_frame_info.dart:13-63); has_* stand for nullable fields. */
typedef struct {
  uint32_t magic;
  int32_t is_skippable, is_legacy;
  int32_t block_independence, block_checksum, content_checksum;
  int32_t has_content_size, has_dict_id;
  uint32_t skippable_size, dict_id, block_max_size, header_size;
  uint64_t content_size;
} lz4ref_frame_info_t;

int lz4ref_frame_info(const uint8_t *src, size_t len, lz4ref_frame_info_t *out); /* :89-199 */

/* lz4SkippableFrameEncode (frame/lz4_frame_encoder.dart:27-46) */
int lz4ref_skippable_frame_encode(const uint8_t *data, size_t n, int index,
                                  uint8_t *dst, size_t cap, size_t *out_len);
/* lz4LegacyFrameEncode (:55-89) */
size_t lz4ref_legacy_frame_bound(size_t n);
int lz4ref_legacy_frame_encode(const uint8_t *src, size_t n, int64_t acceleration,
                               uint8_t *dst, size_t cap, size_t *out_len);
/* lz4CompressWithSize / lz4DecompressWithSize (block/lz4_sized_block.dart:15-55);
 * level 0 = fast (param = acceleration), 1 = hc (param = maxSearchDepth, default options). */
int lz4ref_compress_with_size(const uint8_t *src, size_t n, int level, int64_t param,
                              uint8_t *dst, size_t cap, size_t *out_len);
int lz4ref_decompress_with_size(const uint8_t *src, size_t len, uint8_t *dst, size_t cap,
                                size_t *out_len);

/* ---- streaming transformers ----
 * Encoder: frame/lz4_frame_stream_encoder.dart:25-236.  Every call appends the
 * pieces the transformer would yield to dst and records each piece's length.
 * When a call fails, *out_len / *npieces cover what was yielded before the throw. */
typedef struct lz4ref_stream_encoder lz4ref_stream_encoder;
lz4ref_stream_encoder *lz4ref_stream_encoder_new(const lz4ref_frame_options *o,
                                                 const uint8_t *dict, size_t dict_len, int *status);
int lz4ref_stream_encoder_header(lz4ref_stream_encoder *e, uint8_t *dst, size_t cap, size_t *out_len);
int lz4ref_stream_encoder_add(lz4ref_stream_encoder *e, const uint8_t *chunk, size_t n,
                              uint8_t *dst, size_t cap, size_t *out_len,
                              uint32_t *piece_lens, size_t piece_cap, size_t *npieces);
int lz4ref_stream_encoder_close(lz4ref_stream_encoder *e, uint8_t *dst, size_t cap, size_t *out_len,
                                uint32_t *piece_lens, size_t piece_cap, size_t *npieces);
void lz4ref_stream_encoder_free(lz4ref_stream_encoder *e);

/* Decoder: frame/lz4_frame_stream_decoder.dart:62-599.  next() is decodeNext():
 * *has_output = 0 stands for a null return; the block stays valid until the next call. */
typedef struct lz4ref_stream_decoder lz4ref_stream_decoder;
lz4ref_stream_decoder *lz4ref_stream_decoder_new(int64_t max_output_bytes,
                                                 lz4ref_dict_resolver resolver, void *user,
                                                 int have_resolver);
int lz4ref_stream_decoder_add(lz4ref_stream_decoder *d, const uint8_t *chunk, size_t n);
int lz4ref_stream_decoder_next(lz4ref_stream_decoder *d, int *has_output, const uint8_t **out,
                               size_t *out_len, uint32_t *err_dict_id);
int lz4ref_stream_decoder_finish(lz4ref_stream_decoder *d);
void lz4ref_stream_decoder_free(lz4ref_stream_decoder *d);

/* Checker only (no reference counterpart): 1 when the compressed block obeys the LZ4 block format's
 * end-of-block rules, 0 when it ends the way the reference's encoders allow (lz4_block_encoder.dart:95,118). */
int lz4ref_block_conformant(const uint8_t *src, size_t c);

/* Harness helpers (no reference counterpart): the per-block calls over a run of equal-sized blocks in one call
 * (level 0 fast / 1 hc as in lz4ref_block_compress_ex; decode strict without a dictionary, lenient behind the
 * dictionary prefix held in tmp[0..dict_len) otherwise). */
int lz4ref_blocks_compress(const uint8_t *src, size_t nblocks, size_t block, const uint8_t *dict, size_t dict_len,
                           int level, int64_t param, uint8_t *comp, size_t comp_stride, int64_t *sizes);
int lz4ref_blocks_decompress(const uint8_t *comp, size_t comp_stride, const int64_t *sizes, size_t nblocks,
                             size_t block, uint8_t *tmp, size_t dict_len, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
