// dart:ffi binding of liblz4mi.so behind dart_lz4's public functions.
//
// Drop-in for lib/dart_lz4.dart:60-220: the functions below keep the
// reference's signatures; each copies its Uint8List arguments into native
// memory, calls the C ABI (include/lz4mi.h) and rebuilds either the result
// list or the reference's exception from the status code.  The stream
// transformers keep their async* chunk loops here and hand every chunk to the
// library's stream encoder / decoder objects.
//
// Not executed in the build image (no Dart SDK there); kept thin on purpose.
// Uses only dart:ffi from the SDK: malloc/free are looked up in the process
// (the package has no runtime dependencies, pubspec.yaml:10).
import 'dart:async';
import 'dart:ffi';
import 'dart:typed_data';

import 'package:dart_lz4/dart_lz4.dart'
    show
        Lz4CompressionLevel,
        Lz4CorruptDataException,
        Lz4DictionaryResolver,
        Lz4Exception,
        Lz4FormatException,
        Lz4FrameCompression,
        Lz4FrameInfo,
        Lz4FrameOptions,
        Lz4HcOptions,
        Lz4OutputLimitException,
        Lz4UnsupportedFeatureException;

final DynamicLibrary _lib = DynamicLibrary.open('liblz4mi.so');
final DynamicLibrary _proc = DynamicLibrary.process();

final Pointer<Uint8> Function(int) _malloc = _proc
    .lookupFunction<Pointer<Uint8> Function(IntPtr), Pointer<Uint8> Function(int)>('malloc');
final void Function(Pointer<Uint8>) _free =
    _proc.lookupFunction<Void Function(Pointer<Uint8>), void Function(Pointer<Uint8>)>('free');

typedef _Ctx = Pointer<Void>;

final int Function(int, Pointer<Void>, Pointer<_Ctx>) _ctxCreate = _lib.lookupFunction<
    Int32 Function(Int32, Pointer<Void>, Pointer<_Ctx>),
    int Function(int, Pointer<Void>, Pointer<_Ctx>)>('lz4mi_ctx_create');
final Pointer<Uint8> Function(int) _statusType = _lib
    .lookupFunction<Pointer<Uint8> Function(Int32), Pointer<Uint8> Function(int)>('lz4mi_status_type');
final Pointer<Uint8> Function(int) _statusMessage = _lib
    .lookupFunction<Pointer<Uint8> Function(Int32), Pointer<Uint8> Function(int)>('lz4mi_status_message');
final int Function(int) _compressBound =
    _lib.lookupFunction<IntPtr Function(IntPtr), int Function(int)>('lz4mi_compress_bound');

typedef _CompressC = Int32 Function(_Ctx, Pointer<Uint8>, IntPtr, Pointer<Uint8>, IntPtr, Int64,
    Pointer<Uint8>, IntPtr, Pointer<IntPtr>);
typedef _CompressD = int Function(
    _Ctx, Pointer<Uint8>, int, Pointer<Uint8>, int, int, Pointer<Uint8>, int, Pointer<IntPtr>);
final _CompressD _compressFast = _lib.lookupFunction<_CompressC, _CompressD>('lz4mi_block_compress_fast');
final _CompressD _compressHc = _lib.lookupFunction<_CompressC, _CompressD>('lz4mi_block_compress_hc');

final int Function(_Ctx, Pointer<Uint8>, int, Pointer<Uint8>, int) _decompress = _lib.lookupFunction<
    Int32 Function(_Ctx, Pointer<Uint8>, IntPtr, Pointer<Uint8>, Int64),
    int Function(_Ctx, Pointer<Uint8>, int, Pointer<Uint8>, int)>('lz4mi_block_decompress');

// Mirrors lz4mi_frame_options (include/lz4mi.h).
final class _FrameOptions extends Struct {
  @Int32() external int blockSizeId;
  @Int32() external int blockIndependence;
  @Int32() external int blockChecksum;
  @Int32() external int contentChecksum;
  @Int32() external int hasContentSize;
  @Int32() external int hasDictId;
  @Int32() external int compression;
  @Int32() external int acceleration;
  @Int32() external int hcMaxSearchDepth;
  @Uint32() external int dictId;
  @Uint64() external int contentSize;
  @Int32() external int specConformant; // not a reference option; always 0 behind the reference's API
  @Int32() external int reserved;
}

final int Function(int, Pointer<_FrameOptions>) _frameBound = _lib.lookupFunction<
    IntPtr Function(IntPtr, Pointer<_FrameOptions>),
    int Function(int, Pointer<_FrameOptions>)>('lz4mi_frame_bound');
final int Function(_Ctx, Pointer<Uint8>, int, Pointer<_FrameOptions>, Pointer<Uint8>, int, Pointer<Uint8>,
    int, Pointer<IntPtr>) _frameEncode = _lib.lookupFunction<
    Int32 Function(_Ctx, Pointer<Uint8>, IntPtr, Pointer<_FrameOptions>, Pointer<Uint8>, IntPtr,
        Pointer<Uint8>, IntPtr, Pointer<IntPtr>),
    int Function(_Ctx, Pointer<Uint8>, int, Pointer<_FrameOptions>, Pointer<Uint8>, int, Pointer<Uint8>, int,
        Pointer<IntPtr>)>('lz4mi_frame_encode');
final int Function(Pointer<Uint8>, int, Pointer<IntPtr>) _frameDecodeBound = _lib.lookupFunction<
    Int32 Function(Pointer<Uint8>, IntPtr, Pointer<IntPtr>),
    int Function(Pointer<Uint8>, int, Pointer<IntPtr>)>('lz4mi_frame_decode_bound');
final int Function(_Ctx, Pointer<Uint8>, int, Pointer<IntPtr>, Pointer<Uint8>, int, int, int, int,
    Pointer<Uint8>, int, Pointer<IntPtr>, Pointer<Uint32>) _frameDecodeNext = _lib.lookupFunction<
    Int32 Function(_Ctx, Pointer<Uint8>, IntPtr, Pointer<IntPtr>, Pointer<Uint8>, IntPtr, Int32, Int32,
        Int64, Pointer<Uint8>, IntPtr, Pointer<IntPtr>, Pointer<Uint32>),
    int Function(_Ctx, Pointer<Uint8>, int, Pointer<IntPtr>, Pointer<Uint8>, int, int, int, int,
        Pointer<Uint8>, int, Pointer<IntPtr>, Pointer<Uint32>)>('lz4mi_frame_decode_next');

_Ctx? _ctxCache; // one context per isolate (the reference's scratch is isolate-local too)

_Ctx _ctx() {
  final cached = _ctxCache;
  if (cached != null) return cached;
  final out = _malloc(8).cast<_Ctx>();
  final st = _ctxCreate(0, nullptr, out);
  final ctx = out.value;
  _free(out.cast());
  if (st != 0) throw StateError('lz4mi: no usable GPU (status $st)');
  return _ctxCache = ctx;
}

String _cstr(Pointer<Uint8> p) {
  var n = 0;
  while (p[n] != 0) n++;
  return String.fromCharCodes(p.asTypedList(n));
}

Never _throwStatus(int st, {int? dictId, String? arg, Object? value}) {
  if (st == 60) throw RangeError.value(value ?? 0, arg);
  final msg = _cstr(_statusMessage(st)) + (st == 50 && dictId != null ? '$dictId' : '');
  switch (_cstr(_statusType(st))) {
    case 'Lz4FormatException':
      throw Lz4FormatException(msg);
    case 'Lz4CorruptDataException':
      throw Lz4CorruptDataException(msg);
    case 'Lz4OutputLimitException':
      throw Lz4OutputLimitException(msg);
    case 'Lz4UnsupportedFeatureException':
      throw Lz4UnsupportedFeatureException(msg);
    case 'Lz4Exception':
      throw Lz4Exception(msg);
    default:
      throw StateError('lz4mi backend: $msg ($st)');
  }
}

Pointer<Uint8> _toNative(Uint8List bytes) {
  final p = _malloc(bytes.isEmpty ? 1 : bytes.length);
  p.asTypedList(bytes.length).setAll(0, bytes);
  return p;
}

/// lib/dart_lz4.dart:60-72
Uint8List lz4Compress(
  Uint8List src, {
  Lz4CompressionLevel level = Lz4CompressionLevel.fast,
  int acceleration = 1,
  Lz4HcOptions? hcOptions,
}) {
  if (level == Lz4CompressionLevel.fast && acceleration < 1) {
    throw RangeError.value(acceleration, 'acceleration');
  }
  final cap = _compressBound(src.length) + 16;
  final pSrc = _toNative(src), pDst = _malloc(cap), pLen = _malloc(8).cast<IntPtr>();
  try {
    final st = level == Lz4CompressionLevel.fast
        ? _compressFast(_ctx(), pSrc, src.length, nullptr, 0, acceleration, pDst, cap, pLen)
        : _compressHc(_ctx(), pSrc, src.length, nullptr, 0, (hcOptions ?? Lz4HcOptions()).maxSearchDepth,
            pDst, cap, pLen);
    if (st != 0) _throwStatus(st, arg: 'acceleration', value: acceleration);
    return Uint8List.fromList(pDst.asTypedList(pLen.value));
  } finally {
    _free(pSrc);
    _free(pDst);
    _free(pLen.cast());
  }
}

/// lib/dart_lz4.dart:80-85
Uint8List lz4Decompress(Uint8List src, {required int decompressedSize}) {
  if (decompressedSize < 0) throw RangeError.value(decompressedSize, 'decompressedSize');
  final pSrc = _toNative(src), pDst = _malloc(decompressedSize == 0 ? 1 : decompressedSize);
  try {
    final st = _decompress(_ctx(), pSrc, src.length, pDst, decompressedSize);
    if (st != 0) _throwStatus(st);
    return Uint8List.fromList(pDst.asTypedList(decompressedSize));
  } finally {
    _free(pSrc);
    _free(pDst);
  }
}

/// lib/dart_lz4.dart:110-117
Uint8List lz4FrameEncodeWithOptions(Uint8List src, {required Lz4FrameOptions options, Uint8List? dictionary}) {
  final o = _malloc(sizeOf<_FrameOptions>()).cast<_FrameOptions>();
  o.ref
    ..blockSizeId = 4 + options.blockSize.index
    ..blockIndependence = options.blockIndependence ? 1 : 0
    ..blockChecksum = options.blockChecksum ? 1 : 0
    ..contentChecksum = options.contentChecksum ? 1 : 0
    ..hasContentSize = options.contentSize != null ? 1 : 0
    ..contentSize = options.contentSize ?? 0
    ..hasDictId = options.dictId != null ? 1 : 0
    ..dictId = options.dictId ?? 0
    ..compression = options.compression == Lz4FrameCompression.fast ? 0 : 1
    ..acceleration = options.acceleration > 0x7fffffff ? 0x7fffffff : options.acceleration
    ..hcMaxSearchDepth = (options.hcOptions ?? Lz4HcOptions()).maxSearchDepth
    ..specConformant = 0
    ..reserved = 0;
  final cap = _frameBound(src.length, o);
  final pSrc = _toNative(src), pDst = _malloc(cap), pLen = _malloc(8).cast<IntPtr>();
  final pDict = dictionary == null ? nullptr.cast<Uint8>() : _toNative(dictionary);
  try {
    final st = _frameEncode(_ctx(), pSrc, src.length, o, pDict, dictionary?.length ?? 0, pDst, cap, pLen);
    if (st != 0) _throwStatus(st);
    return Uint8List.fromList(pDst.asTypedList(pLen.value));
  } finally {
    _free(pSrc);
    _free(pDst);
    _free(pLen.cast());
    _free(o.cast());
    if (dictionary != null) _free(pDict);
  }
}

/// lib/dart_lz4.dart:126-136.  The loop over concatenated frames and the resolver callback
/// stay on the Dart side (lib/src/frame/lz4_frame_decoder.dart:45-76, :225-236).
Uint8List lz4FrameDecode(Uint8List src, {int? maxOutputBytes, Lz4DictionaryResolver? dictionaryResolver}) {
  final pSrc = _toNative(src);
  final pWords = _malloc(32);
  final pConsumed = pWords.cast<IntPtr>(), pProduced = (pWords + 8).cast<IntPtr>();
  final pBound = (pWords + 16).cast<IntPtr>(), pNeed = (pWords + 24).cast<Uint32>();
  final out = BytesBuilder(copy: true);
  Uint8List? dict; // persists across frames like _Lz4FrameDecoder._dict
  var pos = 0;
  try {
    while (pos < src.length) {
      _frameDecodeBound(pSrc + pos, src.length - pos, pBound);
      final remaining = maxOutputBytes == null ? -1 : maxOutputBytes - out.length;
      final cap = remaining < 0 || pBound.value < remaining + 1 ? pBound.value : remaining + 1;
      final pDst = _malloc(cap == 0 ? 1 : cap);
      Uint8List? frameDict = dict;
      var state = 0, st = 0;
      try {
        while (true) {
          final pDict = frameDict == null ? nullptr.cast<Uint8>() : _toNative(frameDict);
          st = _frameDecodeNext(_ctx(), pSrc + pos, src.length - pos, pConsumed, pDict, frameDict?.length ?? 0,
              state, dict != null ? 1 : 0, remaining, pDst, cap, pProduced, pNeed);
          if (frameDict != null) _free(pDict);
          if (st == 72 && state == 0) {
            if (dictionaryResolver == null) {
              state = 2;
            } else {
              final d = dictionaryResolver(pNeed.value);
              if (d == null) {
                state = 3;
              } else {
                frameDict = d;
                state = 1;
              }
            }
            continue;
          }
          break;
        }
        if (st != 0) _throwStatus(st, dictId: pNeed.value);
        if (state == 1) dict = frameDict;
        out.add(pDst.asTypedList(pProduced.value));
        pos += pConsumed.value;
      } finally {
        _free(pDst);
      }
    }
    return out.takeBytes();
  } finally {
    _free(pSrc);
    _free(pWords);
  }
}

// ------------------------------------------------------------------ format edges

Pointer<_FrameOptions> _nativeOptions(Lz4FrameOptions options) {
  final o = _malloc(sizeOf<_FrameOptions>()).cast<_FrameOptions>();
  o.ref
    ..blockSizeId = 4 + options.blockSize.index
    ..blockIndependence = options.blockIndependence ? 1 : 0
    ..blockChecksum = options.blockChecksum ? 1 : 0
    ..contentChecksum = options.contentChecksum ? 1 : 0
    ..hasContentSize = options.contentSize != null ? 1 : 0
    ..contentSize = options.contentSize ?? 0
    ..hasDictId = options.dictId != null ? 1 : 0
    ..dictId = options.dictId ?? 0
    ..compression = options.compression == Lz4FrameCompression.fast ? 0 : 1
    ..acceleration = options.acceleration > 0x7fffffff ? 0x7fffffff : options.acceleration
    ..hcMaxSearchDepth = (options.hcOptions ?? Lz4HcOptions()).maxSearchDepth
    ..specConformant = 0
    ..reserved = 0;
  return o;
}

// Mirrors lz4mi_frame_info (include/lz4mi.h).
final class _FrameInfo extends Struct {
  @Uint32() external int magic;
  @Int32() external int isSkippable;
  @Int32() external int isLegacy;
  @Int32() external int blockIndependence;
  @Int32() external int blockChecksum;
  @Int32() external int contentChecksum;
  @Int32() external int hasContentSize;
  @Int32() external int hasDictId;
  @Uint32() external int skippableSize;
  @Uint32() external int dictId;
  @Uint32() external int blockMaxSize;
  @Uint32() external int headerSize;
  @Uint64() external int contentSize;
}

final int Function(Pointer<Uint8>, int, Pointer<_FrameInfo>) _frameInfoParse = _lib.lookupFunction<
    Int32 Function(Pointer<Uint8>, IntPtr, Pointer<_FrameInfo>),
    int Function(Pointer<Uint8>, int, Pointer<_FrameInfo>)>('lz4mi_frame_info_parse');
final int Function(Pointer<Uint8>, int, int, Pointer<Uint8>, int, Pointer<IntPtr>) _skippableEncode =
    _lib.lookupFunction<Int32 Function(Pointer<Uint8>, IntPtr, Int32, Pointer<Uint8>, IntPtr, Pointer<IntPtr>),
        int Function(Pointer<Uint8>, int, int, Pointer<Uint8>, int, Pointer<IntPtr>)>(
        'lz4mi_skippable_frame_encode');
final int Function(int) _legacyBound =
    _lib.lookupFunction<IntPtr Function(IntPtr), int Function(int)>('lz4mi_legacy_frame_bound');
final int Function(_Ctx, Pointer<Uint8>, int, int, Pointer<Uint8>, int, Pointer<IntPtr>) _legacyEncode =
    _lib.lookupFunction<Int32 Function(_Ctx, Pointer<Uint8>, IntPtr, Int64, Pointer<Uint8>, IntPtr, Pointer<IntPtr>),
        int Function(_Ctx, Pointer<Uint8>, int, int, Pointer<Uint8>, int, Pointer<IntPtr>)>(
        'lz4mi_legacy_frame_encode');

/// lib/src/frame/lz4_frame_info.dart:89-199.  Lz4FrameInfo's constructor is private to its
/// library, so the maintainer's copy of this function lives next to the class and fills it
/// from the struct; the fields map one to one (has* = nullable).
_FrameInfo lz4FrameInfoNative(Uint8List src, Pointer<_FrameInfo> out) {
  final pSrc = _toNative(src);
  try {
    final st = _frameInfoParse(pSrc, src.length, out);
    if (st != 0) _throwStatus(st);
    return out.ref;
  } finally {
    _free(pSrc);
  }
}

/// lib/dart_lz4.dart:207-209 -> lib/src/frame/lz4_frame_encoder.dart:27-46
Uint8List lz4SkippableEncode(Uint8List data, {int index = 0}) {
  if (index < 0 || index > 15) throw RangeError.range(index, 0, 15, 'index');
  final cap = 8 + data.length;
  final pSrc = _toNative(data), pDst = _malloc(cap), pLen = _malloc(8).cast<IntPtr>();
  try {
    final st = _skippableEncode(pSrc, data.length, index, pDst, cap, pLen);
    if (st != 0) _throwStatus(st, arg: 'data.length', value: data.length);
    return Uint8List.fromList(pDst.asTypedList(pLen.value));
  } finally {
    _free(pSrc);
    _free(pDst);
    _free(pLen.cast());
  }
}

/// lib/dart_lz4.dart:218-220 -> lib/src/frame/lz4_frame_encoder.dart:55-89
Uint8List lz4LegacyEncode(Uint8List src, {int acceleration = 1}) {
  final cap = _legacyBound(src.length);
  final pSrc = _toNative(src), pDst = _malloc(cap), pLen = _malloc(8).cast<IntPtr>();
  try {
    final st = _legacyEncode(_ctx(), pSrc, src.length, acceleration, pDst, cap, pLen);
    if (st != 0) _throwStatus(st, arg: 'acceleration', value: acceleration);
    return Uint8List.fromList(pDst.asTypedList(pLen.value));
  } finally {
    _free(pSrc);
    _free(pDst);
    _free(pLen.cast());
  }
}

/// lib/src/block/lz4_sized_block.dart:15-31
Uint8List lz4CompressWithSize(Uint8List src,
    {Lz4CompressionLevel level = Lz4CompressionLevel.fast, int acceleration = 1}) {
  final compressed = lz4Compress(src, level: level, acceleration: acceleration);
  final out = Uint8List(4 + compressed.length);
  ByteData.view(out.buffer).setUint32(0, src.length, Endian.little);
  out.setRange(4, out.length, compressed);
  return out;
}

/// lib/src/block/lz4_sized_block.dart:39-55
Uint8List lz4DecompressWithSize(Uint8List src) {
  if (src.length < 4) throw Lz4FormatException(_cstr(_statusMessage(9)));
  final size = ByteData.view(src.buffer, src.offsetInBytes, src.length).getUint32(0, Endian.little);
  return lz4Decompress(Uint8List.sublistView(src, 4), decompressedSize: size);
}

// -------------------------------------------------------------- stream transformers

typedef _Obj = Pointer<Void>;

final int Function(Pointer<_FrameOptions>, Pointer<Uint8>, int, Pointer<IntPtr>) _frameHeader =
    _lib.lookupFunction<Int32 Function(Pointer<_FrameOptions>, Pointer<Uint8>, IntPtr, Pointer<IntPtr>),
        int Function(Pointer<_FrameOptions>, Pointer<Uint8>, int, Pointer<IntPtr>)>('lz4mi_frame_header');
final int Function(_Ctx, Pointer<_FrameOptions>, Pointer<Uint8>, int, Pointer<_Obj>) _encCreate =
    _lib.lookupFunction<Int32 Function(_Ctx, Pointer<_FrameOptions>, Pointer<Uint8>, IntPtr, Pointer<_Obj>),
        int Function(_Ctx, Pointer<_FrameOptions>, Pointer<Uint8>, int, Pointer<_Obj>)>(
        'lz4mi_frame_encoder_create');
final int Function(_Obj, int) _encBound =
    _lib.lookupFunction<IntPtr Function(_Obj, IntPtr), int Function(_Obj, int)>('lz4mi_frame_encoder_bound');
final int Function(_Obj, Pointer<Uint8>, int, Pointer<Uint8>, int, Pointer<IntPtr>) _encUpdate =
    _lib.lookupFunction<Int32 Function(_Obj, Pointer<Uint8>, IntPtr, Pointer<Uint8>, IntPtr, Pointer<IntPtr>),
        int Function(_Obj, Pointer<Uint8>, int, Pointer<Uint8>, int, Pointer<IntPtr>)>(
        'lz4mi_frame_encoder_update');
final int Function(_Obj, Pointer<Uint8>, int, Pointer<IntPtr>) _encFinish = _lib.lookupFunction<
    Int32 Function(_Obj, Pointer<Uint8>, IntPtr, Pointer<IntPtr>),
    int Function(_Obj, Pointer<Uint8>, int, Pointer<IntPtr>)>('lz4mi_frame_encoder_finish');
final void Function(_Obj) _encDestroy =
    _lib.lookupFunction<Void Function(_Obj), void Function(_Obj)>('lz4mi_frame_encoder_destroy');

final int Function(_Ctx, int, Pointer<_Obj>) _decCreate = _lib.lookupFunction<
    Int32 Function(_Ctx, Int64, Pointer<_Obj>), int Function(_Ctx, int, Pointer<_Obj>)>('lz4mi_frame_decoder_create');
final int Function(_Obj, Pointer<Uint8>, int) _decFeed = _lib.lookupFunction<
    Int32 Function(_Obj, Pointer<Uint8>, IntPtr), int Function(_Obj, Pointer<Uint8>, int)>('lz4mi_frame_decoder_feed');
final int Function(_Obj, Pointer<Uint8>, int, Pointer<IntPtr>, Pointer<Uint32>) _decNext = _lib.lookupFunction<
    Int32 Function(_Obj, Pointer<Uint8>, IntPtr, Pointer<IntPtr>, Pointer<Uint32>),
    int Function(_Obj, Pointer<Uint8>, int, Pointer<IntPtr>, Pointer<Uint32>)>('lz4mi_frame_decoder_next');
final int Function(_Obj, Pointer<Uint8>, int, int) _decSetDict = _lib.lookupFunction<
    Int32 Function(_Obj, Pointer<Uint8>, IntPtr, Int32),
    int Function(_Obj, Pointer<Uint8>, int, int)>('lz4mi_frame_decoder_set_dictionary');
final int Function(_Obj) _decFinish =
    _lib.lookupFunction<Int32 Function(_Obj), int Function(_Obj)>('lz4mi_frame_decoder_finish');
final void Function(_Obj) _decDestroy =
    _lib.lookupFunction<Void Function(_Obj), void Function(_Obj)>('lz4mi_frame_decoder_destroy');

/// Splits concatenated block records into one list per record, as the reference yields them.
Iterable<Uint8List> _records(Uint8List all, bool blockChecksum) sync* {
  final view = ByteData.sublistView(all);
  var pos = 0;
  while (pos < all.length) {
    final n = 4 + (view.getUint32(pos, Endian.little) & 0x7fffffff) + (blockChecksum ? 4 : 0);
    yield Uint8List.sublistView(all, pos, pos + n);
    pos += n;
  }
}

/// lib/dart_lz4.dart:179-185 -> lib/src/frame/lz4_frame_stream_encoder.dart:25-236
StreamTransformer<List<int>, List<int>> lz4FrameEncoderWithOptions(
    {required Lz4FrameOptions options, Uint8List? dictionary}) {
  return StreamTransformer.fromBind((input) async* {
    final o = _nativeOptions(options);
    final pLen = _malloc(8).cast<IntPtr>();
    final pObj = _malloc(8).cast<_Obj>();
    final pHdr = _malloc(32);
    final pDict = dictionary == null ? nullptr.cast<Uint8>() : _toNative(dictionary);
    var enc = nullptr.cast<Void>();
    try {
      var st = _frameHeader(o, pHdr, 32, pLen);
      if (st != 0) _throwStatus(st);
      yield Uint8List.fromList(pHdr.asTypedList(pLen.value));
      st = _encCreate(_ctx(), o, pDict, dictionary?.length ?? 0, pObj);
      enc = pObj.value;
      if (st != 0) _throwStatus(st);
      await for (final chunk in input) {
        final bytes = chunk is Uint8List ? chunk : Uint8List.fromList(chunk);
        if (bytes.isEmpty) continue;
        final cap = _encBound(enc, bytes.length);
        final pSrc = _toNative(bytes), pDst = _malloc(cap);
        try {
          st = _encUpdate(enc, pSrc, bytes.length, pDst, cap, pLen);
          if (st != 0) _throwStatus(st);
          if (pLen.value != 0) {
            yield* Stream.fromIterable(
                _records(Uint8List.fromList(pDst.asTypedList(pLen.value)), options.blockChecksum));
          }
        } finally {
          _free(pSrc);
          _free(pDst);
        }
      }
      final cap = _encBound(enc, 0);
      final pDst = _malloc(cap);
      try {
        st = _encFinish(enc, pDst, cap, pLen);
        final tail = Uint8List.fromList(pDst.asTypedList(pLen.value));
        if (st != 0) {
          yield* Stream.fromIterable(_records(tail, options.blockChecksum));
          _throwStatus(st);
        }
        final trailer = 4 + (options.contentChecksum ? 4 : 0);
        final body = tail.length - trailer;
        yield* Stream.fromIterable(_records(Uint8List.sublistView(tail, 0, body), options.blockChecksum));
        yield Uint8List.sublistView(tail, body, body + 4);
        if (options.contentChecksum) yield Uint8List.sublistView(tail, body + 4);
      } finally {
        _free(pDst);
      }
    } finally {
      if (enc != nullptr) _encDestroy(enc);
      if (dictionary != null) _free(pDict);
      _free(pHdr);
      _free(pObj.cast());
      _free(pLen.cast());
      _free(o.cast());
    }
  });
}

/// lib/dart_lz4.dart:166-172
StreamTransformer<List<int>, List<int>> lz4FrameEncoder({int acceleration = 1, Uint8List? dictionary}) {
  return lz4FrameEncoderWithOptions(options: Lz4FrameOptions(acceleration: acceleration), dictionary: dictionary);
}

/// lib/dart_lz4.dart:147-155 -> lib/src/frame/lz4_frame_stream_decoder.dart:15-45
StreamTransformer<List<int>, List<int>> lz4FrameDecoder(
    {int? maxOutputBytes, Lz4DictionaryResolver? dictionaryResolver}) {
  return StreamTransformer.fromBind((input) async* {
    final pObj = _malloc(8).cast<_Obj>();
    final pWords = _malloc(16);
    final pLen = pWords.cast<IntPtr>(), pNeed = (pWords + 8).cast<Uint32>();
    var st = _decCreate(_ctx(), maxOutputBytes ?? -1, pObj);
    final dec = pObj.value;
    var cap = 32 << 20;
    var pDst = _malloc(cap);

    Iterable<Uint8List> drain() sync* {
      while (true) {
        final st = _decNext(dec, pDst, cap, pLen, pNeed);
        if (st == 0) {
          yield Uint8List.fromList(pDst.asTypedList(pLen.value));
        } else if (st == 74) {
          return; // decodeNext() == null
        } else if (st == 71) {
          _free(pDst);
          cap = pLen.value > cap * 2 ? pLen.value : cap * 2;
          pDst = _malloc(cap);
        } else if (st == 72) {
          final d = dictionaryResolver == null ? null : dictionaryResolver(pNeed.value);
          if (d == null) {
            _decSetDict(dec, nullptr, 0, dictionaryResolver == null ? 2 : 3);
          } else {
            final pDict = _toNative(d);
            _decSetDict(dec, pDict, d.length, 1);
            _free(pDict);
          }
        } else {
          _throwStatus(st, dictId: pNeed.value);
        }
      }
    }

    try {
      if (st != 0) _throwStatus(st);
      await for (final chunk in input) {
        final bytes = chunk is Uint8List ? chunk : Uint8List.fromList(chunk);
        final pSrc = _toNative(bytes);
        st = _decFeed(dec, pSrc, bytes.length);
        _free(pSrc);
        if (st != 0) _throwStatus(st);
        yield* Stream.fromIterable(drain());
      }
      st = _decFinish(dec);
      if (st != 0) _throwStatus(st, dictId: pNeed.value);
      yield* Stream.fromIterable(drain());
    } finally {
      if (dec != nullptr) _decDestroy(dec);
      _free(pDst);
      _free(pObj.cast());
      _free(pWords);
    }
  });
}
