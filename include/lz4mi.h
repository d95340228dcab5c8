/*
 * lz4mi — MI355X (gfx950) implementation of the dart_lz4 block/frame hot path.
 *
 * C ABI: plain pointers and sizes, caller-owned buffers, int32 status codes,
 * no exceptions and no aborts across the boundary.  A dart:ffi shim (or any
 * other FFI) binds exactly these entry points; each comment names the
 * reference interface (path:line in the dart_lz4 checkout) it stands in for.
 *
 * Status codes map 1:1 to the Dart exception class + message the reference
 * throws, so the host-language shim can rebuild the identical exception.
 *
 * Threading: every call takes an explicit context; a context owns one HIP
 * stream and its device scratch, and is used by one thread at a time.
 * Different contexts may be used concurrently (the reference keeps its scratch
 * in isolate-local globals, lib/src/block/lz4_block_encoder.dart:9-11).
 */
#ifndef LZ4MI_H
#define LZ4MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status
 * lib/src/internal/lz4_exception.dart:2-35 (taxonomy); message sources:
 * lib/src/block/lz4_block_decoder.dart:23,34,44,51,62,73,107;
 * lib/src/internal/byte_writer.dart:98,131,146; byte_reader.dart:75,104;
 * lib/src/frame/lz4_frame_decoder.dart:54,71,112,131,177,189,193,222,228,233,
 * 252,264,313,319,396; lib/src/frame/lz4_frame_encoder.dart:270;
 * lib/src/frame/lz4_frame_stream_encoder.dart:166,225;
 * lib/src/frame/lz4_frame_stream_decoder.dart:210,425; lib/src/block/lz4_sized_block.dart:41. */
enum {
  LZ4MI_OK = 0,
  LZ4MI_FORMAT_EOF = 1,              /* Lz4FormatException 'Unexpected end of input' */
  LZ4MI_FORMAT_MAGIC = 2,            /* 'Invalid LZ4 frame magic number' */
  LZ4MI_FORMAT_FLG_RESERVED = 3,     /* 'Reserved FLG bit is set' */
  LZ4MI_FORMAT_BD_RESERVED = 4,      /* 'Reserved BD bits are set' */
  LZ4MI_FORMAT_BLOCK_MAX = 5,        /* 'Invalid block maximum size' */
  LZ4MI_FORMAT_INT_OVERFLOW = 6,     /* 'Integer overflow' */
  LZ4MI_FORMAT_CONTENT_SIZE_ARG = 7, /* 'contentSize does not match src length' */
  LZ4MI_FORMAT_CONTENT_SIZE_STREAM = 8, /* 'contentSize does not match stream length' (stream encoder) */
  LZ4MI_FORMAT_SIZE_HEADER = 9,      /* 'Input too short for size header' (lz4DecompressWithSize) */
  LZ4MI_CORRUPT_LITERAL = 10,        /* Lz4CorruptDataException 'Literal length exceeds output size' */
  LZ4MI_CORRUPT_MATCH = 11,          /* 'Match length exceeds output size' */
  LZ4MI_CORRUPT_TRAILING = 12,       /* 'Trailing bytes after end of block' */
  LZ4MI_CORRUPT_DISTANCE = 13,       /* 'Invalid match distance' */
  LZ4MI_CORRUPT_HEADER_CK = 14,      /* 'Header checksum mismatch' */
  LZ4MI_CORRUPT_BLOCK_SIZE = 15,     /* 'Block size exceeds maximum' */
  LZ4MI_CORRUPT_BLOCK_CK = 16,       /* 'Block checksum mismatch' */
  LZ4MI_CORRUPT_CONTENT_CK = 17,     /* 'Content checksum mismatch' */
  LZ4MI_CORRUPT_CONTENT_SIZE = 18,   /* 'Content size mismatch' */
  LZ4MI_CORRUPT_LEGACY_SIZE = 19,    /* 'Invalid legacy block size' */
  LZ4MI_CORRUPT_LEGACY_NOT_FULL = 20,/* 'Legacy block is not full' */
  LZ4MI_LIMIT = 30,                  /* Lz4OutputLimitException 'Output limit exceeded' */
  LZ4MI_UNSUPPORTED_VERSION = 40,    /* Lz4UnsupportedFeatureException 'Unsupported LZ4 frame version' */
  LZ4MI_UNSUPPORTED_DICT = 41,       /* '... no dictionary resolver provided' */
  LZ4MI_UNSUPPORTED_CONTENT_SIZE = 42, /* 'Content size exceeds supported integer range' (stream decoder) */
  LZ4MI_DICT_NOT_FOUND = 50,         /* Lz4Exception 'Dictionary not found for ID: <id>' */
  LZ4MI_RANGE = 60,                  /* Dart RangeError (argument validation) */
  /* not reference errors */
  LZ4MI_BACKEND = 70,                /* HIP runtime failure; see lz4mi_last_backend_error */
  LZ4MI_DST_TOO_SMALL = 71,          /* caller's destination buffer cannot hold the result */
  LZ4MI_NEED_DICT = 72,              /* lz4mi_frame_decode_next: call again with the dictionary */
  LZ4MI_STREAM_NEED_INPUT = 74,      /* lz4mi_frame_decoder_next: nothing more can be decoded from the bytes fed so far */
  LZ4MI_UNSUPPORTED_HERE = 73        /* feature outside this library's path (dependent blocks through the shard calls, blocks of 2 GiB and more) */
};

/* Dart exception class name / message for a status (static strings). */
const char *lz4mi_status_type(int32_t status);
const char *lz4mi_status_message(int32_t status);

/* ----------------------------------------------------------------- context */
typedef struct lz4mi_ctx lz4mi_ctx;

/* device: HIP device ordinal.  stream: a hipStream_t to run on, or NULL for a
 * stream owned by the context.  Fails with LZ4MI_BACKEND when no GPU is usable. */
int32_t lz4mi_ctx_create(int32_t device, void *stream, lz4mi_ctx **out);
void lz4mi_ctx_destroy(lz4mi_ctx *ctx);
int32_t lz4mi_ctx_synchronize(lz4mi_ctx *ctx);
/* Device memory the context holds at the moment, in bytes: every scratch buffer it has allocated
 * (hash tables, staging, per-block bookkeeping, search index).  No counterpart in the reference;
 * lets a caller size its own allocations and watch that a call allocated nothing. */
uint64_t lz4mi_ctx_device_bytes(lz4mi_ctx *ctx);
/* ctx == NULL: reason of the last failed lz4mi_ctx_create */
const char *lz4mi_last_backend_error(lz4mi_ctx *ctx);
/* nonzero when built with the HIP kernels for gfx950 (always, in this build) */
int32_t lz4mi_version(void);

/* worst case of the fast/HC block encoders: n + n/255 + 16
 * (all-literal block, lib/src/block/lz4_block_encoder.dart:202-223). */
size_t lz4mi_compress_bound(size_t n);

/* ----------------------------------------------- single block, host buffers
 * These four are what the façade functions call.
 *
 * lz4Compress(level: fast)  lib/dart_lz4.dart:60-72 ->
 *   lz4BlockCompress        lib/src/block/lz4_block_encoder.dart:24-157
 * acceleration < 1 -> LZ4MI_RANGE even when n == 0 (:41-48).  dict may be NULL. */
int32_t lz4mi_block_compress_fast(lz4mi_ctx *ctx, const uint8_t *src, size_t n,
                                  const uint8_t *dict, size_t dict_len, int64_t acceleration,
                                  uint8_t *dst, size_t cap, size_t *out_len);

/* lz4Compress(level: hc) -> lz4HcBlockCompress  lib/src/hc/lz4_hc_block_encoder.dart:38-197;
 * max_search_depth < 1 -> LZ4MI_RANGE (lib/src/hc/lz4_hc_options.dart:13-15).
 * The search index takes 20 bytes of device scratch per input byte (dictionary included): a block
 * whose index does not fit the device's free memory (with 288 GB, blocks of several GiB would not)
 * returns LZ4MI_UNSUPPORTED_HERE with the sizes in lz4mi_last_backend_error(ctx). */
int32_t lz4mi_block_compress_hc(lz4mi_ctx *ctx, const uint8_t *src, size_t n,
                                const uint8_t *dict, size_t dict_len, int64_t max_search_depth,
                                uint8_t *dst, size_t cap, size_t *out_len);

/* lz4Decompress  lib/dart_lz4.dart:80-85 -> lz4BlockDecompress
 * lib/src/block/lz4_block_decoder.dart:7-77 (strict: exact size, trailing-byte check).
 * dst must hold decompressed_size bytes. */
int32_t lz4mi_block_decompress(lz4mi_ctx *ctx, const uint8_t *src, size_t c,
                               uint8_t *dst, int64_t decompressed_size);

/* lz4BlockDecompressInto  lib/src/block/lz4_block_decoder.dart:79-119 with a
 * ByteWriter(maxLength: max_len) already holding dst[0..prefix_len)
 * (lib/src/frame/lz4_frame_decoder.dart:284-286).  *out_len = final writer length. */
int32_t lz4mi_block_decompress_into(lz4mi_ctx *ctx, const uint8_t *src, size_t c,
                                    uint8_t *dst, size_t prefix_len, size_t max_len,
                                    size_t *out_len);

/* xxh32  lib/src/xxhash/xxh32.dart:166-240 (one-shot, host buffer).  One xxh32 stream is a chain of dependent
 * rounds: inputs below 4 MiB go to the device kernel, larger ones are hashed where they are, by the calling thread
 * (several times a GPU wavefront's rate, and no copy); LZ4MI_HASH_ENGINE=gpu keeps every size on the device. */
int32_t lz4mi_xxh32(lz4mi_ctx *ctx, const uint8_t *data, size_t len, uint32_t seed, uint32_t *out);

/* Xxh32 (streaming)  lib/src/xxhash/xxh32.dart:21-164.  Plain data owned by the caller:
 * v1..v4, the <16 carry bytes and the running length, exactly the reference's fields.
 * update() runs the whole-stripe rounds (:83-118) on the GPU; the carry bookkeeping
 * (:58-81,:120-124) and the digest's tail + avalanche over at most 15 carried bytes
 * (:127-163) are host arithmetic on the struct (an update that completes no stripe does
 * not touch the device and accepts ctx == NULL). */
typedef struct {
  uint32_t seed;
  uint32_t v[4];
  uint32_t mem_size;
  uint64_t total_len;
  uint8_t mem[16];
} lz4mi_xxh32_state;

void lz4mi_xxh32_state_init(lz4mi_xxh32_state *s, uint32_t seed);
int32_t lz4mi_xxh32_state_update(lz4mi_ctx *ctx, lz4mi_xxh32_state *s, const uint8_t *data, size_t len);
uint32_t lz4mi_xxh32_state_digest(const lz4mi_xxh32_state *s);

/* ------------------------------------------------ batches, device-resident
 * The per-block loop bodies of the frame encoder/decoder
 * (lib/src/frame/lz4_frame_encoder.dart:154-261, lz4_frame_decoder.dart:238-304)
 * as one launch over many independent blocks.  All pointers prefixed d_ are
 * device memory; work is enqueued on the context's stream and NOT synchronised.
 *
 * Block b reads  d_src + (d_src_off ? d_src_off[b] : b * src_stride), d_src_len[b] bytes
 * (d_src_len NULL: every block src_len_uniform bytes, the last one shortened to
 * src_total), and writes at d_dst + (d_dst_off ? d_dst_off[b] : b * dst_stride).
 */
typedef struct {
  const uint8_t *d_src;
  const uint64_t *d_src_off; /* optional */
  const uint32_t *d_src_len; /* optional */
  uint64_t src_stride;
  uint32_t src_len_uniform;
  uint64_t src_total;        /* used with src_len_uniform: total bytes over all blocks */
  uint8_t *d_dst;
  const uint64_t *d_dst_off; /* optional */
  uint64_t dst_stride;
  const uint32_t *d_dst_cap; /* optional per-block capacity / expected size */
  uint32_t dst_cap_uniform;
  const uint8_t *d_dict;     /* optional shared dictionary / history prefix */
  uint32_t dict_len;
  uint32_t nblocks;
  uint32_t *d_out_len;       /* per block: bytes produced (excluding the prefix) */
  int32_t *d_status;         /* per block: LZ4MI_* */
} lz4mi_batch;

/* lz4BlockCompressToWriter per block (lib/src/block/lz4_block_encoder.dart:35-157).
 * Every block needs capacity >= lz4mi_compress_bound(len) or gets LZ4MI_DST_TOO_SMALL. */
int32_t lz4mi_compress_fast_batch(lz4mi_ctx *ctx, const lz4mi_batch *b, int64_t acceleration);

/* lz4HcBlockCompressToWriter per block (lib/src/hc/lz4_hc_block_encoder.dart:49-197). */
int32_t lz4mi_compress_hc_batch(lz4mi_ctx *ctx, const lz4mi_batch *b, int64_t max_search_depth);

/* strict != 0: lz4BlockDecompress, capacity = exact decompressed size.
 * strict == 0: lz4BlockDecompressInto, capacity = ByteWriter maxLength minus the
 * prefix; the shared d_dict (last 65536 bytes used) is the pre-filled prefix. */
int32_t lz4mi_decompress_batch(lz4mi_ctx *ctx, const lz4mi_batch *b, int32_t strict);

/* lz4mi_decompress_batch for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arguments, same results; the launch's queue words and its block order list
 * live in a static device array of the library (8 MiB + 256 B per GPU, held until the process
 * ends) instead of context buffers, so the call performs no hipMalloc/hipFree.
 * At most LZ4MI_STATIC_MAX_BLOCKS blocks per call: more -> LZ4MI_UNSUPPORTED_HERE with the
 * reason in lz4mi_last_backend_error(ctx), nothing launched.  All contexts on one GPU share
 * that array.  Calls on one stream are ordered by the stream; a call that arrives on another
 * stream than the array's last user makes its stream wait for that user's launch first (an
 * event recorded behind every launch, hipStreamWaitEvent, no allocation), so calls from
 * several contexts or threads take turns instead of racing.  What stays the caller's: a
 * stream under graph capture must be the only user of this entry point on its GPU while the
 * graph is captured and replayed (a replay records no event). */
#define LZ4MI_STATIC_MAX_BLOCKS (1u << 20)
int32_t lz4mi_decompress_batch_static(lz4mi_ctx *ctx, const lz4mi_batch *b, int32_t strict);

/* How a decode launch spreads its blocks.  No counterpart in the reference (a tuning knob of this
 * library; results never depend on it).  0 (default): by launch size - launches that leave wavefront
 * slots empty give every block a workgroup of two wavefronts (one finds the sequences and places the
 * literals of the next batch while the other copies the matches of the current one: a block is done in
 * 0.6 of the time), larger launches give every block one wavefront (twice as many blocks in flight).
 * 1: always one wavefront per block.  2: always two.
 * lz4mi_ctx_set_decode_form sets it for one context (form -1: follow the process default again) and returns
 * the context's previous setting (-2 for a bad argument).  lz4mi_set_decode_form sets the process default
 * that contexts without a setting of their own follow; it returns the previous default, or -1 for a value
 * out of range.  The environment variable LZ4MI_DEC_FORM=single|pair sets the start value of the default. */
int32_t lz4mi_ctx_set_decode_form(lz4mi_ctx *ctx, int32_t form);
int32_t lz4mi_set_decode_form(int32_t form);

/* xxh32 of each block's source bytes (seed 0): block checksums
 * (lib/src/frame/lz4_frame_encoder.dart:242-257, lz4_frame_decoder.dart:257-266).
 * Only d_src/d_src_off/d_src_len/src_* /nblocks are read; d_hash receives nblocks words. */
int32_t lz4mi_xxh32_batch(lz4mi_ctx *ctx, const lz4mi_batch *b, uint32_t *d_hash);

/* Single-stream xxh32 over device memory (content checksum,
 * lib/src/frame/lz4_frame_encoder.dart:148,163,266; lz4_frame_decoder.dart:311).  The word is written to
 * d_out in stream order.  One xxh32 stream is a chain of dependent rounds that a host core runs several times
 * faster than a GPU wavefront, so inputs of 4 MiB and more are copied to pinned host memory in pieces (on a side
 * stream) and hashed by the calling thread: the call then returns when the hash is done.  LZ4MI_HASH_ENGINE=gpu
 * in the environment keeps every size on the device and the call asynchronous (needed under stream capture). */
int32_t lz4mi_xxh32_device(lz4mi_ctx *ctx, const uint8_t *d_data, uint64_t len, uint32_t seed,
                           uint32_t *d_out);

/* ------------------------------------------------------------------- frames
 * Mirrors Lz4FrameOptions (lib/src/frame/lz4_frame_options.dart:38-101);
 * block_size_id is Lz4FrameBlockSize.bdId 4..7 (:104-115). */
typedef struct {
  int32_t block_size_id;      /* 4=64KiB 5=256KiB 6=1MiB 7=4MiB (default 7) */
  int32_t block_independence; /* default 1 */
  int32_t block_checksum;
  int32_t content_checksum;
  int32_t has_content_size;
  int32_t has_dict_id;
  int32_t compression;        /* 0 fast, 1 hc  (Lz4FrameCompression) */
  int32_t acceleration;       /* default 1 */
  int32_t hc_max_search_depth;/* default 64 */
  uint32_t dict_id;
  uint64_t content_size;
  /* Not a reference option (default 0).  Nonzero: every block obeys the lz4 block format's
   * end-of-block rules (no match starts in the last 12 bytes, the last 5 bytes are literals,
   * blocks under 13 bytes are literals only), so the frame decodes with any LZ4 decoder.
   * The reference's encoders end blocks inside matches, which the lz4 CLI rejects; with this
   * flag the output is no longer byte-identical to the reference's. */
  int32_t spec_conformant;
  int32_t reserved;
} lz4mi_frame_options;

void lz4mi_frame_options_default(lz4mi_frame_options *o);

/* Upper bound of lz4mi_frame_encode's output for n input bytes. */
size_t lz4mi_frame_bound(size_t n, const lz4mi_frame_options *o);

/* lz4FrameEncodeWithOptions  lib/dart_lz4.dart:110-117 ->
 * lz4FrameEncodeBytesWithOptions lib/src/frame/lz4_frame_encoder.dart:105-274.
 * Host buffers.  A contentSize that differs from n is reported
 * (LZ4MI_FORMAT_CONTENT_SIZE_ARG) after the frame has been produced (:269-271). */
int32_t lz4mi_frame_encode(lz4mi_ctx *ctx, const uint8_t *src, size_t n,
                           const lz4mi_frame_options *o, const uint8_t *dict, size_t dict_len,
                           uint8_t *dst, size_t cap, size_t *out_len);

/* Same, device-resident source, dictionary and destination (independent or linked blocks);
 * cap >= lz4mi_frame_bound(n, o) or LZ4MI_DST_TOO_SMALL.  *out_len (host) = frame bytes.
 * The call synchronises the context's stream once (it reads the frame size back).
 * A contentSize that differs from n: the frame is produced, *out_len is set, and the call
 * returns LZ4MI_FORMAT_CONTENT_SIZE_ARG, as lz4mi_frame_encode does (:269-271). */
int32_t lz4mi_frame_encode_device(lz4mi_ctx *ctx, const uint8_t *d_src, uint64_t n,
                                  const lz4mi_frame_options *o, const uint8_t *d_dict,
                                  uint32_t dict_len, uint8_t *d_dst, uint64_t cap,
                                  uint64_t *out_len);

/* Multi-GPU frame assembly, step 1: compress this rank's blocks (device
 * resident) and report each block's record size in the frame:
 * 4 + stored_len (+4 with block checksums), where stored_len follows the
 * compressed-iff-smaller rule (lib/src/frame/lz4_frame_encoder.dart:235).
 * The compressed payloads stay in the context's scratch until step 2, and stored blocks
 * stay where they are in d_src: the caller keeps d_src alive and makes NO other call on this
 * context between step 1 and step 2 (any other entry point reuses that scratch and
 * invalidates the shard: step 2 then returns LZ4MI_RANGE). */
int32_t lz4mi_frame_shard_compress(lz4mi_ctx *ctx, const uint8_t *d_src, uint64_t n,
                                   const lz4mi_frame_options *o, const uint8_t *d_dict,
                                   uint32_t dict_len, uint32_t *d_record_sizes,
                                   uint32_t *nblocks_out);

/* Step 2: after the record sizes of all ranks have been all-gathered and
 * prefix-summed by the caller (the exchange is the caller's: RCCL all-gather from
 * torch.distributed, MPI, or a host gather; the library links no collective library),
 * write this rank's block records into d_dst (the rank's window of the frame).
 * d_record_offsets[b] = offset of block b's record relative to d_dst.
 * d_src, n and o must be step 1's (LZ4MI_RANGE otherwise); a record that would end behind
 * cap fails the call with LZ4MI_DST_TOO_SMALL before anything is written. */
int32_t lz4mi_frame_shard_write(lz4mi_ctx *ctx, const uint8_t *d_src, uint64_t n,
                                const lz4mi_frame_options *o, const uint64_t *d_record_offsets,
                                uint8_t *d_dst, uint64_t cap);

/* Frame header bytes (magic..HC) and trailer (end mark [+ content checksum]).
 * lib/src/frame/lz4_frame_encoder.dart:110-144 and :263-267. Host buffers. */
int32_t lz4mi_frame_header(const lz4mi_frame_options *o, uint8_t *dst, size_t cap, size_t *out_len);

/* One frame or skippable frame per call, so the caller's loop can run the
 * Lz4DictionaryResolver callback between header parse and block decode
 * (_decodeNextFrameOrSkippable lib/src/frame/lz4_frame_decoder.dart:52-76,
 * resolver :225-236).  Host buffers.
 *
 * max_out < 0: no limit ("maxOutputBytes: null"); otherwise the remaining
 * budget for this frame (:158-164).  dict/dict_len: the dictionary to use
 * (dict == NULL and have_dict == 0: none).  If the frame carries a dictId and
 * dict_state is 0 ("not resolved yet"), returns LZ4MI_NEED_DICT with
 * *dict_id_needed set and *consumed = 0; the caller resolves and calls again
 * with dict_state 1 (resolved; dict may be empty), 2 (no resolver supplied)
 * or 3 (resolver returned null). */
int32_t lz4mi_frame_decode_next(lz4mi_ctx *ctx, const uint8_t *src, size_t len, size_t *consumed,
                                const uint8_t *dict, size_t dict_len, int32_t dict_state,
                                int32_t have_prev_dict, int64_t max_out, uint8_t *dst, size_t cap,
                                size_t *out_len, uint32_t *dict_id_needed);

/* Capacity that lz4mi_frame_decode_next needs at most for the next frame: parses the header and
 * walks the block size words on the host (no GPU work).  Sum over the block records of: the
 * payload size for a stored block; min(blockMaxSize, 255 * payload size + 32) for a compressed
 * one (a payload byte yields at most 255 output bytes).  The contentSize field is not trusted
 * (a frame may lie; the reference reports that only after decoding, :317-321).  A caller with an
 * output limit passes min(bound, limit + 1) as cap. */
int32_t lz4mi_frame_decode_bound(const uint8_t *src, size_t len, size_t *bound);

/* ------------------------------------------------- sharded decode of one frame
 * The frame decoder's block loop (lib/src/frame/lz4_frame_decoder.dart:238-304) split by
 * contiguous block ranges over several GPUs.  lz4mi_frame_info is declared below.
 *
 * Step 1 (host, no GPU work): the walk over the size words (:243-255, structure only) of the
 * standard frame at src.  record_off[k] = offset in src of record k's size word; at most
 * max_records are stored (*nrecords is always the real count; more than max_records ->
 * LZ4MI_DST_TOO_SMALL; record_off == NULL just counts).  *end_off = offset behind the end
 * mark and content checksum.  Structural failures are the reference's: LZ4MI_FORMAT_EOF,
 * LZ4MI_CORRUPT_BLOCK_SIZE, header errors. */
struct lz4mi_frame_info_s;
int32_t lz4mi_frame_block_table(const uint8_t *src, size_t len, struct lz4mi_frame_info_s *info,
                                uint64_t *record_off, size_t max_records, size_t *nrecords,
                                size_t *end_off);

/* Step 2 (device): one rank decodes its records.  d_records[0..records_len) holds them (the
 * rank's window of the frame), d_record_off[b] is record b's offset in that window.  Block b's
 * bytes go to d_dst + b * blockMaxSize; d_out_len[b] / d_status[b] report what :251-296 would:
 * size word over blockMaxSize -> LZ4MI_CORRUPT_BLOCK_SIZE, record cut short ->
 * LZ4MI_FORMAT_EOF, block checksum (o->block_checksum) -> LZ4MI_CORRUPT_BLOCK_CK, then the
 * block decoder's status with the temporary writer's blockMaxSize cap (LZ4MI_LIMIT).  The
 * first failing block in frame order is the error the reference throws; the caller reduces
 * over ranks.  Only o->block_size_id, block_checksum and block_independence (must be 1) are
 * read.  Enqueued on the context's stream, not synchronised. */
int32_t lz4mi_frame_shard_decode(lz4mi_ctx *ctx, const uint8_t *d_records, uint64_t records_len,
                                 const uint64_t *d_record_off, uint32_t nblocks,
                                 const lz4mi_frame_options *o, const uint8_t *d_dict,
                                 uint32_t dict_len, uint8_t *d_dst, uint32_t *d_out_len,
                                 int32_t *d_status);

/* Step 3 (device): the reference appends whatever each block yields (:291-296); blocks that do
 * not fill their slot are packed here.  Copies d_len[b] bytes from d_slots + b * slot_stride to
 * d_dst + d_dst_off[b] (offsets from the all-gathered decoded sizes, prefix-summed by the
 * caller).  A block that would end behind cap fails the call with LZ4MI_DST_TOO_SMALL before
 * anything is written (one stream synchronisation). */
int32_t lz4mi_blocks_gather(lz4mi_ctx *ctx, const uint8_t *d_slots, uint64_t slot_stride,
                            const uint32_t *d_len, const uint64_t *d_dst_off, uint32_t nblocks,
                            uint8_t *d_dst, uint64_t cap);

/* d_flags[b] = 1 when compressed block b (d_src / d_src_off / d_src_len / src_stride of the
 * batch) breaks the LZ4 block format's end-of-block rules (last five bytes literals, last
 * match starts at least twelve bytes before the end), 0 otherwise.  The reference's encoders
 * produce such blocks (lib/src/block/lz4_block_encoder.dart:95,118: the search runs to the last
 * four bytes and matches extend to the last byte) and its own decoder accepts them; liblz4
 * and the lz4 CLI do not.  This reports how many, it changes nothing. */
int32_t lz4mi_blocks_conformance(lz4mi_ctx *ctx, const lz4mi_batch *b, uint32_t *d_flags);

/* ---------------------------------------------------- host-only frame formats
 * lz4FrameInfo  lib/src/frame/lz4_frame_info.dart:89-199: magic + descriptor only, no
 * payload is touched (no GPU work).  Fields mirror Lz4FrameInfo (:13-63); the has_* flags
 * stand for Dart's nullable fields. */
typedef struct lz4mi_frame_info_s {
  uint32_t magic;
  int32_t is_skippable;
  int32_t is_legacy;
  int32_t block_independence; /* 1 unless a standard frame says otherwise (:56) */
  int32_t block_checksum;
  int32_t content_checksum;
  int32_t has_content_size;
  int32_t has_dict_id;
  uint32_t skippable_size;    /* valid when is_skippable */
  uint32_t dict_id;
  uint32_t block_max_size;    /* 0 for skippable and legacy frames */
  uint32_t header_size;       /* 8 skippable, 4 legacy, magic..HC otherwise */
  uint64_t content_size;
} lz4mi_frame_info;

int32_t lz4mi_frame_info_parse(const uint8_t *src, size_t len, lz4mi_frame_info *out);

/* lz4SkippableFrameEncode  lib/src/frame/lz4_frame_encoder.dart:27-46.
 * index outside 0..15 or n > 2^32-1 -> LZ4MI_RANGE.  Output is 8 + n bytes. */
int32_t lz4mi_skippable_frame_encode(const uint8_t *data, size_t n, int32_t index, uint8_t *dst,
                                     size_t cap, size_t *out_len);

/* lz4LegacyFrameEncode  lib/src/frame/lz4_frame_encoder.dart:55-89: magic 0x184C2102, 8 MiB
 * blocks, each always stored compressed.  All blocks of the input are encoded in one launch
 * (one wavefront per block).  An empty input yields the bare magic without looking at
 * acceleration, as the reference does. */
size_t lz4mi_legacy_frame_bound(size_t n);
int32_t lz4mi_legacy_frame_encode(lz4mi_ctx *ctx, const uint8_t *src, size_t n, int64_t acceleration,
                                  uint8_t *dst, size_t cap, size_t *out_len);

/* ------------------------------------------------------------ stream encoder
 * lz4FrameEncoderTransformerWithOptions  lib/src/frame/lz4_frame_stream_encoder.dart:25-236.
 * The transformer's Dart side keeps only the chunk loop: it yields lz4mi_frame_header()'s
 * bytes first (:42-47), then whatever each update()/finish() call returns.  The buffered
 * input and the 64 KiB history of linked blocks stay in device memory; one update encodes
 * every block the chunk completes in a single launch.
 *
 * create: options are validated like Lz4FrameOptions (LZ4MI_RANGE).  On any other failure
 * *out is still set and must be destroyed.
 * update: a chunk that pushes the total past options.contentSize fails with
 * LZ4MI_FORMAT_CONTENT_SIZE_STREAM before anything is buffered (:165-167).
 * finish: last short block, then the contentSize check (:224-226; on that failure *out_len
 * bytes of dst hold the block that was yielded before the throw), end mark and content
 * checksum.  dst capacity for either call: lz4mi_frame_encoder_bound(e, n). */
typedef struct lz4mi_frame_encoder lz4mi_frame_encoder;
int32_t lz4mi_frame_encoder_create(lz4mi_ctx *ctx, const lz4mi_frame_options *o, const uint8_t *dict,
                                   size_t dict_len, lz4mi_frame_encoder **out);
size_t lz4mi_frame_encoder_bound(const lz4mi_frame_encoder *e, size_t n);
int32_t lz4mi_frame_encoder_update(lz4mi_frame_encoder *e, const uint8_t *chunk, size_t n,
                                   uint8_t *dst, size_t cap, size_t *out_len);
int32_t lz4mi_frame_encoder_finish(lz4mi_frame_encoder *e, uint8_t *dst, size_t cap, size_t *out_len);
void lz4mi_frame_encoder_destroy(lz4mi_frame_encoder *e);

/* ------------------------------------------------------------ stream decoder
 * _Lz4FrameStreamDecoder behind lz4FrameDecoderTransformer
 * lib/src/frame/lz4_frame_stream_decoder.dart:15-45 (driver loop), :62-599 (state machine).
 *
 *   feed(chunk)            = decoder.add(chunk)                         (:109-111)
 *   next(dst, cap, ...)    = decoder.decodeNext()                       (:113-413)
 *   finish()               = decoder.finish()                           (:415-426)
 *
 * next returns LZ4MI_OK with *out_len > 0 bytes of output (every complete block buffered so
 * far that fits into cap, decoded in one launch), LZ4MI_STREAM_NEED_INPUT where decodeNext
 * returns null, LZ4MI_NEED_DICT when a frame names a dictionary (*dict_id_needed; answer with
 * lz4mi_frame_decoder_set_dictionary: 1 resolved, 2 no resolver supplied, 3 resolver
 * returned null, then call next again), LZ4MI_DST_TOO_SMALL with *out_len = the capacity one
 * step needs (nothing consumed; call again with a larger dst), or the failure.  Output that
 * precedes a failing block is delivered first; the failure is returned by the call after.
 * max_output_bytes < 0: no limit (maxOutputBytes: null). */
typedef struct lz4mi_frame_decoder lz4mi_frame_decoder;
int32_t lz4mi_frame_decoder_create(lz4mi_ctx *ctx, int64_t max_output_bytes, lz4mi_frame_decoder **out);
int32_t lz4mi_frame_decoder_feed(lz4mi_frame_decoder *d, const uint8_t *chunk, size_t n);
int32_t lz4mi_frame_decoder_next(lz4mi_frame_decoder *d, uint8_t *dst, size_t cap, size_t *out_len,
                                 uint32_t *dict_id_needed);
int32_t lz4mi_frame_decoder_set_dictionary(lz4mi_frame_decoder *d, const uint8_t *dict, size_t n,
                                           int32_t answer);
int32_t lz4mi_frame_decoder_finish(lz4mi_frame_decoder *d);
void lz4mi_frame_decoder_destroy(lz4mi_frame_decoder *d);

#ifdef __cplusplus
}
#endif
#endif
