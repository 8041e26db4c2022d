/*
 * kompressor_hip.h -- C ABI of libkompressor_hip.so, the MI355X (gfx950) backend
 * that plugs in underneath Kompressor's SliceTransform / ZstdCompressor /
 * ZstdDecompressor API.
 *
 * Part 1 mirrors, one to one, what the reference's JNI layer
 * (kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp) calls in
 * libzstd, with the same return-code convention (size_t, errors are
 * (size_t)-code with libzstd 1.5.7's code numbering), so that Wrapper.cpp can
 * be re-pointed at this library without touching the Kotlin side
 * (INTEGRATION.md shows the binding).
 *
 * Part 2 is the batched device-pointer API (no reference equivalent: the
 * reference compresses one slice per JNI call); it is what part 1 runs with
 * n = 1 and what bench.py measures.
 *
 * Plain pointers and sizes only.  Thread rules follow the reference
 * (SURVEY.md section 8b): a context is used by one thread at a time; different
 * contexts may be used concurrently; free may come from any thread.
 */
#ifndef KOMPRESSOR_HIP_H
#define KOMPRESSOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMP_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* Part 1: streaming-compatible single-slice API                             */
/* ------------------------------------------------------------------------ */
typedef struct kmp_zstd_cctx kmp_zstd_cctx;
typedef struct kmp_zstd_dctx kmp_zstd_dctx;

/* end_op values == ZSTD_EndDirective (Wrapper.cpp:112 passes e_end / e_continue) */
#define KMP_ZSTD_e_continue 0
#define KMP_ZSTD_e_flush    1
#define KMP_ZSTD_e_end      2

/* parameter ids == ZSTD_cParameter (kompressor-zstd--nativelib/src/commonMain/kotlin/
 * com/ensody/kompressor/zstd/ZstdParameter.kt:3-24; only 100 is ever set,
 * ZstdCompressor.jvm.kt:21) */
#define KMP_ZSTD_c_compressionLevel 100

/* replaces ZSTD_createCCtx            (Wrapper.cpp:10-17)  */
KMP_API kmp_zstd_cctx* kmp_zstd_create_cctx(void);
/* replaces ZSTD_freeCCtx              (Wrapper.cpp:19-27)  */
KMP_API size_t kmp_zstd_free_cctx(kmp_zstd_cctx* cctx);
/* replaces ZSTD_CCtx_setParameter     (Wrapper.cpp:29-39). Levels -131072 .. -1 and 1 .. 10 (0 = default = 3) run on the GPU where
 * kmp_zstd_compress_batch_level serves them (dictionaries: level 3 only; levels 5 .. 10: one closing call of at most 128 KiB -- 9 and 10
 * above 16 KiB; what is not served surfaces when the stream closes); levels 11 and up return (size_t)-40 "Unsupported parameter". */
KMP_API size_t kmp_zstd_cctx_set_parameter(kmp_zstd_cctx* cctx, int param, int value);
/* replaces ZSTD_CCtx_loadDictionary   (Wrapper.cpp:41-56). Served: dictionaries of 8 .. 130 560 bytes for slices <= 128 KiB, raw content
 * or zstd's own format (magic EC30A437: what `zstd --train` / ZDICT_trainFromBuffer write -- its Huffman and FSE tables, repeat offsets
 * and ID are loaded as libzstd's ZSTD_loadCEntropy does; a damaged header surfaces at the first frame as "Dictionary is corrupted");
 * a dictionary outside that size range returns (size_t)-40. */
KMP_API size_t kmp_zstd_cctx_load_dictionary(kmp_zstd_cctx* cctx, const void* dict, size_t dict_size);
/* replaces ZSTD_compressStream2       (Wrapper.cpp:75-121, call at :112).
 * Same buffer semantics as ZSTD_inBuffer / ZSTD_outBuffer: `*_size` is the
 * end-exclusive index, `*_pos` the cursor, both into the whole array
 * (Wrapper.cpp:101-110).  Returns 0 when the frame is completely flushed,
 * >0 = bytes still to flush, or an error code.
 * The frame is the one libzstd 1.5.7 writes under the same calls, which above 128 KiB depends on them: when the call
 * that ends the input (e_end) finds libzstd's staging buffer empty and room for ZSTD_compressBound of what it brought,
 * the caller's memory is compressed in place (ZSTD_compress2's frame); otherwise -- the reference's driver, whose output
 * slices hold max(8192, n / 10) bytes -- the input is staged in chunks of 128 KiB (kmp_zstd_compress_batch_reference),
 * and data that arrived with e_continue makes it a streaming frame (no content size).  Input is collected until e_end;
 * streams up to 1 GiB at every served level (beyond a level's window -- 512 KiB at level 1 and the negative levels, 1 MiB
 * at level 2, 2 MiB at levels 3 and 4 -- the window slides as libzstd's does). */
KMP_API size_t kmp_zstd_compress_stream(kmp_zstd_cctx* cctx,
                                        void* dst, size_t dst_size, size_t* dst_pos,
                                        const void* src, size_t src_size, size_t* src_pos,
                                        int end_op);

/* replaces ZSTD_createDCtx            (Wrapper.cpp:123-130) */
KMP_API kmp_zstd_dctx* kmp_zstd_create_dctx(void);
/* replaces ZSTD_freeDCtx              (Wrapper.cpp:132-140) */
KMP_API size_t kmp_zstd_free_dctx(kmp_zstd_dctx* dctx);
/* replaces ZSTD_DCtx_loadDictionary   (Wrapper.cpp:58-73): raw-content dictionaries and dictionaries in zstd's own format (their tables
 * stand behind a frame's first block: tree-less literals and "repeat" sequence tables refer to them; a frame that names another
 * dictionary's ID fails with "Dictionary mismatch"); a damaged header of that format returns (size_t)-64 as libzstd's DDict creation does */
KMP_API size_t kmp_zstd_dctx_load_dictionary(kmp_zstd_dctx* dctx, const void* dict, size_t dict_size);
/* replaces ZSTD_decompressStream      (Wrapper.cpp:142-187, call at :178).
 * Returns 0 when a frame is completely decoded and flushed, otherwise a
 * hint (>0) or an error code.  A frame is decoded when all of it has arrived; content up to 1 GiB. */
KMP_API size_t kmp_zstd_decompress_stream(kmp_zstd_dctx* dctx,
                                          void* dst, size_t dst_size, size_t* dst_pos,
                                          const void* src, size_t src_size, size_t* src_pos);

/* zlib twin of the above (kompressor-zlib--nativelib/src/jvmCommonMain/jni/Wrapper.cpp):
 * createCompressor :10-26 (deflateInit2), freeCompressor :28-38 (deflateEnd), compressStream :40-82
 * (deflate, call at :73).  Same cursor semantics; the return value is zlib's: Z_OK 0, Z_STREAM_END 1,
 * Z_BUF_ERROR -5 are benign for the Kotlin side (ZlibCompressor.jvm.kt:49-56).  The GPU path implements
 * levels 1 .. 9 (-1 = 6: zlib's deflate_fast for 1 .. 3, deflate_slow for 4 .. 9, each byte-identical to zlib's output),
 * windowBits as deflateInit2 takes it: -15 .. -9 (ZlibFormat.Raw), 9 .. 15 (ZlibFormat.Zlib: CMF / FLG header by window and
 * level -- 78 01 / 5E / 9C / DA with a 32 KiB window -- + Adler-32; 8 is taken as 9, as zlib does) or 25 .. 31 (ZlibFormat.Gzip:
 * 10-byte header, CRC-32 + ISIZE), memLevel 1 .. 9, strategy 0, streams up to 1 GiB (the stream is compressed when the caller
 * finishes it); other settings (level 0, windowBits 8 without the zlib wrapper -- an error in zlib too --, other strategies)
 * make create return NULL.  (A decompressor created with a smaller window than the stream's still decodes it: the window
 * declared to inflateInit2 is not enforced.) */
typedef struct kmp_zlib_cstream kmp_zlib_cstream;
typedef struct kmp_zlib_dstream kmp_zlib_dstream;
KMP_API kmp_zlib_cstream* kmp_zlib_create_compressor(int level, int window_bits, int mem_level, int strategy);
KMP_API int kmp_zlib_free_compressor(kmp_zlib_cstream* stream);
KMP_API int kmp_zlib_compress_stream(kmp_zlib_cstream* stream,
                                     void* dst, size_t dst_size, size_t* dst_pos,
                                     const void* src, size_t src_size, size_t* src_pos,
                                     int finish);
/* createDecompressor :84-98 (inflateInit2), freeDecompressor :100-110 (inflateEnd), decompressStream :112-153
 * (inflate, call at :144).  windowBits -15..-8 = raw, 8..15 = zlib wrapper (Adler-32 verified), 24..31 = gzip
 * (CRC-32 and length verified), 40..47 = zlib or gzip by the header (ZlibFormat.AutoDetectZlibGzip).  The stream is decoded when the
 * caller passes finish (the one-shot driver always does); any block types, any compression level. */
KMP_API kmp_zlib_dstream* kmp_zlib_create_decompressor(int window_bits);
KMP_API int kmp_zlib_free_decompressor(kmp_zlib_dstream* stream);
KMP_API int kmp_zlib_decompress_stream(kmp_zlib_dstream* stream,
                                       void* dst, size_t dst_size, size_t* dst_pos,
                                       const void* src, size_t src_size, size_t* src_pos,
                                       int finish);

/* replace ZSTD_isError / ZSTD_getErrorName (Wrapper.cpp:189-196) */
KMP_API unsigned kmp_zstd_is_error(size_t code);
KMP_API const char* kmp_zstd_get_error_name(size_t code);

/* ZSTD_compressBound */
KMP_API size_t kmp_zstd_compress_bound(size_t src_size);

/* ------------------------------------------------------------------------ */
/* Part 2: batched device API                                                */
/* ------------------------------------------------------------------------ */
typedef struct kmp_batch_ctx kmp_batch_ctx;

#define KMP_OK             0
#define KMP_ERR_HIP       (-1)   /* a HIP call failed: kmp_last_error() has the text */
#define KMP_ERR_ARG       (-2)
#define KMP_ERR_CAPACITY  (-3)   /* n or slice size beyond what the context was created for */
#define KMP_ERR_KERNEL    (-4)   /* a kernel guard tripped (never expected) */

/* Per-slice lengths are device data, so what the host cannot check before it launches is reported afterwards: the
 * slice gets d_out_len[i] = 0 (a frame or stream is never empty) and the context's status word collects these bits. */
#define KMP_STATUS_SLICE_TOO_LARGE 1u   /* a d_in_len[i] above the context's max_slice_bytes (DEFLATE: above 64 KiB) */
#define KMP_STATUS_KERNEL_GUARD    2u   /* a parser's loop guard tripped (never expected) */
#define KMP_STATUS_LEVEL_SIZE      4u   /* the level is another strategy at a slice's size (level 4: 128 - 256 KiB; levels 9, 10: 8 bytes .. 16 KiB): out_len 0 */

#define KMP_MAX_SLICE_BYTES (128u * 1024u)        /* one block per frame: the batched fast path */
#define KMP_MAX_BIG_SLICE_BYTES (1u << 30)       /* frames of several blocks (context created with max_slice_bytes above
                                                  * 128 KiB); beyond the level-3 window (2 MiB) it slides as libzstd's does */

/* Workspace for up to max_slices slices of up to max_slice_bytes each on HIP
 * device `device`.  team_lanes: lanes of a wave that cooperate on one slice in
 * the match kernel (4, 8, 16, 32 or 64; 0 = default). */
KMP_API int kmp_batch_create(kmp_batch_ctx** out, int device, uint32_t max_slices,
                             uint32_t max_slice_bytes, int team_lanes);
/* ... with options.  struct_bytes = sizeof(kmp_batch_options) (versioning).  team_lanes: as above.
 * table_span_gib: the address span, in GiB, over which a context with 4 GiB or more of level-3 team tables lays out its
 * workspace arena (-1 = the environment's KMP_TABLE_SPAN_GIB, else 100; 0 = packed).  The parser's table traffic runs 11 % faster
 * over 72 GiB or more of span than inside a few dozen (DESIGN.md section 5a'); the gaps are memory the context holds and does not
 * use, so the span is bounded by half of the device's free memory at creation, and kmp_batch_memory says what was taken.
 * table_retry: 1 = when the arena measures slow whatever the layout and the device has room for a second one, try one more and
 * keep the faster (the other is freed before the call returns: the one transient allocation creation can make); 0 = never
 * (-1 = KMP_TABLE_RETRY, else 0). */
typedef struct kmp_batch_options { uint32_t struct_bytes; int team_lanes; int table_span_gib; int table_retry; } kmp_batch_options;
KMP_API int kmp_batch_create_ex(kmp_batch_ctx** out, int device, uint32_t max_slices, uint32_t max_slice_bytes,
                                const kmp_batch_options* opts);
/* Device memory the context holds right now, by part, in bytes (sets allocated on a first use count once they exist): arena =
 * the one allocation of a large context (arena_used = its parts without the span's gaps), workspace = the separate allocations of
 * a small one plus the per-team / per-slice words, other_tables = the flat tables of levels 1 / 2 / the dictionary parser, level
 * 4's sets and a dictionary's tables, block_chain = the per-slice state of contexts for slices above 128 KiB, decode_staging =
 * what the pre-decoders leave (shared by the zstd decoder and inflate), deflate_workspace = link / match / symbol arrays. */
typedef struct kmp_batch_memory_info { uint32_t struct_bytes; uint32_t reserved; size_t arena, arena_used, workspace, other_tables, block_chain, decode_staging, deflate_workspace, total; } kmp_batch_memory_info;
KMP_API int kmp_batch_memory(kmp_batch_ctx* ctx, kmp_batch_memory_info* info);
KMP_API void kmp_batch_destroy(kmp_batch_ctx* ctx);
/* Status bits (KMP_STATUS_*) raised by the batches run on this context since the last call; waits for `hip_stream`,
 * then clears them.  Returns KMP_OK, KMP_ERR_CAPACITY (a slice was too large) or KMP_ERR_KERNEL; *bits may be NULL.
 * A context runs one batch at a time: a batch queued on another stream waits for the previous one's last kernel. */
KMP_API int kmp_batch_status(kmp_batch_ctx* ctx, uint32_t* bits, void* hip_stream);

/* zstd level-3 frames for n independent slices.  All pointers are device
 * pointers; slice i is d_src[d_in_off[i] .. +d_in_len[i]); its frame goes to
 * d_dst + d_out_off[i] (room for kmp_zstd_compress_bound(len) + 8 bytes) and its
 * size to d_out_len[i].  Asynchronous on `hip_stream` (a hipStream_t, may be 0).
 * The frames are the ones ZSTD_compress2 writes into a buffer of ZSTD_compressBound bytes.  Up to 128 KiB (one block)
 * that is also what the reference's ZstdCompressor(3).transform(ByteArray) returns; above, see
 * kmp_zstd_compress_batch_reference. */
KMP_API int kmp_zstd_compress_batch(kmp_batch_ctx* ctx,
                                    const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                    uint32_t n,
                                    void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                    void* hip_stream);

/* The same batch in `pieces` (1 .. 8) parts that run side by side, part p on hip_streams[p]: what a caller whose slices arrive over
 * PCIe (or from another rank) uses to overlap the copies with the kernels.  Part p covers the slices kmp_batch_piece_range(n, pieces, p)
 * names; its kernels are queued behind whatever the caller has queued on hip_streams[p] (the copy that brings those slices in) and the
 * caller queues what takes the part's frames away behind them (kmp_compact_batch on the part's range, the copy out).  The parts own
 * disjoint parts of the context's team slots and workspace, so a later part joins the earlier ones on the device instead of waiting for
 * them: the device fills up as the batch arrives.  d_in_off / d_in_len / d_out_off must be complete when the call is made (they are
 * small); frames are bit for bit kmp_zstd_compress_batch's.  Level 3, slices up to 128 KiB.  The next batch on the context, on whatever
 * stream, waits for all parts. */
KMP_API void kmp_batch_piece_range(uint32_t n, uint32_t pieces, uint32_t piece, uint32_t* first, uint32_t* count);
KMP_API int kmp_zstd_compress_batch_pieces(kmp_batch_ctx* ctx,
                                           const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                           uint32_t n,
                                           void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                           uint32_t pieces, void* const* hip_streams);

/* Streaming frames: what libzstd writes when a slice arrives through ZSTD_e_continue calls (finish = false:
 * SliceTransformRawSource.kt:32-55, BaseSliceTransformContentEncoder.kt:23-54) and is closed with ZSTD_e_end -- the size
 * is unknown when the frame starts: window 2^21, no content size in the header, the input taken in chunks of 128 KiB.
 * empty_end != 0: the closing call brought no data (then a stream that stops on a chunk boundary ends with an empty
 * block).  Any length the context holds (beyond 2 MiB + 128 KiB libzstd's staging buffer wraps: the lap before becomes
 * an older segment and the blocks are parsed by its extDict variant, restated here); the context must have been created
 * with max_slice_bytes above 128 KiB.
 * kmp_zstd_compress_stream produces these frames by itself when data arrived with KMP_ZSTD_e_continue. */
KMP_API int kmp_zstd_compress_batch_stream(kmp_batch_ctx* ctx,
                                           const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                           uint32_t n,
                                           void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                           int empty_end, void* hip_stream);
/* ... at level 3 (or 0: the call above), 4, 2, 1 or a negative level -- the reference's Ktor ZstdContentEncoder streams at
 * level 1 (kompressor-zstd-ktor ZstdContentEncoder.kt:11).  Levels 1 and below have a window of 2^19, level 2 of 2^20:
 * a longer stream is parsed as libzstd parses it once its staging buffer (window + 128 KiB) has wrapped -- blocks that
 * still reach into the lap before by its extDict variant of the "fast" parser (ZSTD_compressBlock_fast_extDict_generic),
 * restated in zstd_match_fast.h.  Any length the context holds. */
KMP_API int kmp_zstd_compress_batch_stream_level(kmp_batch_ctx* ctx,
                                                 const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                                 uint32_t n,
                                                 void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                                 int empty_end, int level, void* hip_stream);
/* The frames Kompressor's ZstdCompressor(level).transform(ByteArray) returns, for slices of any size the context holds.
 * Its driver (SliceTransform.kt:33-56) hands ZSTD_compressStream2 (Wrapper.cpp:112) output slices of max(8192, n / 10)
 * bytes: from 128 KiB + 1 on that is less than ZSTD_compressBound(n), so libzstd does not compress the array in place
 * but stages it in chunks of 128 KiB -- the block pre-splitter sees one chunk at a time, and beyond the window + 128 KiB
 * the staging buffer wraps and the window slides.  Frames differ from kmp_zstd_compress_batch's
 * wherever the pre-splitter cuts; up to 128 KiB they are the same.  Levels -131072 .. -1 and 1 .. 4 (0 = 3), any size the
 * context holds.  out_chunk: size of the caller's output slices if it is not the reference's (0 = max(8192, n / 10)). */
KMP_API int kmp_zstd_compress_batch_reference(kmp_batch_ctx* ctx,
                                              const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                              uint32_t n,
                                              void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                              int level, uint32_t out_chunk, void* hip_stream);
/* One-shot frames at another compression level: 1 and 2 (libzstd's one-table "fast" strategy) for slices of any size the
 * context holds (one block up to 128 KiB, frames of several blocks above; the window follows the slice size as
 * ZSTD_getCParams(level, n, 0) sets it, and a slice beyond the level's largest window is parsed with the window sliding);
 * 3 (or 0) = kmp_zstd_compress_batch.  Frames are the ones libzstd 1.5.7 writes at that level.
 * Negative levels (-131072 .. -1: libzstd's "fast" strategy on row 0 of its parameter tables, a step of 1 - level, literals
 * left uncompressed) are served like level 1.
 * Level 4 is served where libzstd runs it as "double-fast": slices above 16 KiB up to 128 KiB (window <= 17, chain 17, hash 17,
 * minimum match 4; ZSTD_getCParams(4, n, 0)), slices above 256 KiB (window <= 21, chain 18, hash 18, minimum match 5: contexts created
 * for slices above 128 KiB, per-slice tables of 2 MiB allocated by the first such batch) and streams of any size (kmp_zstd_compress_batch_stream_level).  Its tables (1 MiB per team: 64 GiB beside a 65 536-slice context, less when the device has less room; KMP_L4_TEAMS caps it) are allocated by
 * the first level-4 batch of a context.  Level 4 up to 16 KiB is strategy "greedy": those slices of a batch go through the kernels of levels
 * 5 .. 10; a slice of more than 128 KiB up to 256 KiB (greedy again, several blocks) is refused like an oversized one (out_len 0,
 * KMP_STATUS_LEVEL_SIZE).
 * Levels 5 .. 10 are libzstd's strategies "greedy" (5; 4 up to 16 KiB), "lazy" (6; 5 up to 16 KiB) and "lazy2" (7 .. 10; 6 .. 8 up to
 * 16 KiB) over its row-based match finder (windows above 2^14: the finder a library built for 128-bit vectors uses, as the reference's JNI
 * library is) or its hash chains (slices up to 16 KiB): served for slices of one block (<= 128 KiB, context created for such slices).  At
 * levels 9 and 10 a slice of 8 bytes .. 16 KiB is strategy "btlazy2": refused as above.  The first batch at these levels allocates their
 * workspace (16 bytes per position of up to 16 384 slices). */
KMP_API int kmp_zstd_compress_batch_level(kmp_batch_ctx* ctx,
                                          const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                          uint32_t n,
                                          void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                          int level, void* hip_stream);
/* same with a dictionary shared by all n slices (what ZstdCompressor(level, dictionary) does per slice:
 * ZSTD_CCtx_loadDictionary, Wrapper.cpp:41-56, then the one-shot compress): frames are the ones libzstd 1.5.7 writes --
 * its CDict is built here on the host once per dictionary; slices up to 16 KiB are parsed against the attached CDict,
 * larger ones against its copied tables.  h_dict is HOST memory, 8 .. 130 560 bytes; slices <= 128 KiB.
 * The dictionary is raw content, or -- when it starts with the magic EC30A437 -- in zstd's own format (ZDICT / `zstd --train`): then
 * matches are searched in its content part, the block is coded against its entropy tables (literals with its Huffman table when that
 * is cheaper or the input is small, sequence tables it marks complete reused below 1 000 sequences), a frame starts with its repeat
 * offsets and names its ID.  KMP_ERR_ARG when the magic is there and the header behind it is damaged. */
KMP_API int kmp_zstd_compress_batch_dict(kmp_batch_ctx* ctx,
                                         const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                         uint32_t n,
                                         void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                         const void* h_dict, uint32_t dict_size,
                                         void* hip_stream);

/* Inverse: n zstd frames -> n slices.  Frame i is d_src[d_in_off[i] .. +d_in_len[i]);
 * its content goes to d_dst + d_out_off[i] (capacity d_out_cap[i]); d_out_len[i]
 * receives the decoded size and d_status[i] 0 or a libzstd error code (20 =
 * corruption, 70 = destination too small, 72 = source size wrong / truncated, 10 = no zstd magic,
 * 14 = unsupported frame parameter ...).  An entry may hold several frames back to back, with skippable frames
 * between them: their contents are concatenated, as ZSTD_decompress does; an empty entry decodes to nothing.
 * Nothing outside [d_in_off[i], +d_in_len[i]) is read and nothing outside [d_out_off[i], +d_out_cap[i]) is
 * written, whatever the bytes of the entry are (tests/fuzz_decoders.py). */
KMP_API int kmp_zstd_decompress_batch(kmp_batch_ctx* ctx,
                                      const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                      uint32_t n,
                                      void* d_dst, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                      uint32_t* d_out_len, uint32_t* d_status,
                                      void* hip_stream);
/* same with a dictionary shared by all n frames (reference: ZstdDecompressor(dictionary), Wrapper.cpp:58-73, test ZstdTest.kt:49-65).
 * Raw content: d_dict[0 .. dict_size) is the history before the first byte of every frame.  zstd's own format (magic EC30A437): the
 * content part is that history, and the tables, repeat offsets and ID in front of it are what ZSTD_loadDEntropy makes of them (the
 * dictionary's first 2 KiB are read back to the host once per dictionary).  status 32 ("Dictionary mismatch") for a frame whose header
 * names another ID -- with a raw-content dictionary or none: any ID. */
KMP_API int kmp_zstd_decompress_batch_dict(kmp_batch_ctx* ctx,
                                           const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                           uint32_t n,
                                           void* d_dst, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                           uint32_t* d_out_len, uint32_t* d_status,
                                           const void* d_dict, uint32_t dict_size,
                                           void* hip_stream);

/* Raw DEFLATE (RFC 1951) streams as zlib level 6 / windowBits 15 / memLevel 8 / strategy 0 writes
 * them: the batched form of deflateInit2(6, Z_DEFLATED, -15, 8, 0) + deflate(Z_FINISH)
 * (reference: kompressor-zlib--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:20,73; Kotlin
 * ZlibCompressor(ZlibFormat.Raw, 6)).  Slices up to the context's max_slice_bytes (64 KiB at least; above 64 KiB zlib's
 * 32 KiB window slides, and a context made for such slices takes them through its kernels in 64 KiB spans: the call then waits
 * once on hip_stream, to read the batch's longest slice back, before it enqueues the rest); stream i goes to
 * d_dst + d_out_off[i] (room for kmp_deflate_bound(len), which covers all three formats: stored blocks of 5 + n
 * bytes per 65 535, the 18 bytes of a gzip header and trailer), its size to d_out_len[i]. */
KMP_API size_t kmp_deflate_bound(size_t src_size);
KMP_API int kmp_deflate_compress_batch(kmp_batch_ctx* ctx,
                                       const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                       uint32_t n,
                                       void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                       void* hip_stream);
/* same, with the RFC 1950 zlib wrapper (what ZlibFormat.Zlib at the default level produces) */
KMP_API int kmp_zlib_compress_batch(kmp_batch_ctx* ctx,
                                    const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                    uint32_t n,
                                    void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                    void* hip_stream);
/* same, with the RFC 1952 gzip wrapper (ZlibFormat.Gzip -> windowBits + 16, ZlibFormat.kt:41-44): the 10-byte header
 * zlib writes on Linux (no name, MTIME 0, XFL 0, OS 3), CRC-32 and ISIZE trailer */
KMP_API int kmp_gzip_compress_batch(kmp_batch_ctx* ctx,
                                    const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                    uint32_t n,
                                    void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                    void* hip_stream);
/* The three above at another of zlib's levels (1 .. 9, -1 = 6; ZlibCompressor(level = ...),
 * ZlibCompressor.jvm.kt:19-30 -> deflateInit2, Wrapper.cpp:20).  4 .. 9 (deflate_slow): the level's good_length / max_lazy /
 * nice_length / max_chain (4: 4 4 16 16, 5: 8 16 32 32, 6: 8 16 128 128, 7: 8 32 128 256, 8: 32 128 258 1024,
 * 9: 32 258 258 4096) drive the same kernels.  1 .. 3 (deflate_fast; 1: 4 4 8 4, 2: 4 5 16 8, 3: 4 6 32 32): one kernel that
 * parses and keeps its hash chains a lane per slice.  The zlib header's level flags (78 01 / 78 5E / 78 9C / 78 DA) and gzip's
 * XFL (4 at level 1, 2 at level 9) follow.  format: 0 raw, 1 zlib, 2 gzip.  Level 0 (stored) is not served: KMP_ERR_ARG. */
KMP_API int kmp_deflate_compress_batch_level(kmp_batch_ctx* ctx,
                                             const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                             uint32_t n,
                                             void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                             int format, int level, void* hip_stream);
/* ... and with the other two settings the Kotlin constructor passes to deflateInit2 (ZlibCompressor(format, compressionLevel,
 * windowBits, memLevel): ZlibCompressor.jvm.kt:7-17, ZlibFormat.kt:32-57; Wrapper.cpp:20): window_bits 9 .. 15 -- the window is
 * 1 << window_bits bytes, matches reach back (1 << window_bits) - 262, the zlib header's CMF byte says so; 8 is taken as 9 with
 * the zlib wrapper and refused without it, as zlib does -- and mem_level 1 .. 9: the hash has mem_level + 7 bits and a block is
 * closed after (1 << (mem_level + 6)) - 1 symbols.  Here window_bits is the plain number and `format` names the wrapper (the
 * sign / + 16 encoding of deflateInit2 is the streaming entry point's, kmp_zlib_create_compressor).  Stream i needs room for
 * kmp_deflate_bound_params(len, window_bits, mem_level): zlib's deflateBound for settings other than the default ones (an eighth
 * more than the data).  KMP_ERR_ARG for anything else. */
KMP_API size_t kmp_deflate_bound_params(size_t src_size, int window_bits, int mem_level);
KMP_API int kmp_deflate_compress_batch_params(kmp_batch_ctx* ctx,
                                              const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                              uint32_t n,
                                              void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                              int format, int level, int window_bits, int mem_level, void* hip_stream);
/* inflate: n streams -> d_dst + d_out_off[i] (capacity d_out_cap[i]); format 0 = raw deflate, 1 = zlib,
 * 2 = gzip, 3 = zlib or gzip decided per stream by its first bytes (ZlibFormat.AutoDetectZlibGzip,
 * ZlibFormat.kt:52-55); checksums are verified;
 * d_status[i] = 0, -3 (Z_DATA_ERROR) or -5 (Z_BUF_ERROR: capacity too small or input truncated).
 * Runs as two kernels when the context's staging is there (allocated on the first call, shared with the zstd decoder:
 * streams whose output fits the context's max_slice_bytes are decoded a lane per stream and executed a wave per stream;
 * the others -- and every stream with stored blocks or an error -- by one wave per stream); any n. */
KMP_API int kmp_inflate_batch(kmp_batch_ctx* ctx,
                              const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                              void* d_dst, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                              uint32_t* d_out_len, int32_t* d_status, int format, void* hip_stream);
/* ms4[0..3] = k_deflate_chains, k_deflate_best, k_deflate_parse, k_deflate_encode of the last batch */
KMP_API int kmp_deflate_last_kernel_ms(kmp_batch_ctx* ctx, float* ms4);

/* Dense packing helper: copies frame i from d_src + d_in_off[i] (d_len[i] bytes) to
 * d_dst + d_out_off[i], where d_out_off is the exclusive prefix sum of d_len that
 * this call also computes (d_out_off has n+1 entries; the last is the total). */
KMP_API int kmp_compact_batch(kmp_batch_ctx* ctx, const void* d_src, const uint64_t* d_in_off,
                              const uint32_t* d_len, uint32_t n, void* d_dst, uint64_t* d_out_off,
                              void* hip_stream);

/* Per-kernel device time of the last batch call, measured with HIP events on the
 * launch stream (0 = off, 1 = on).  kmp_batch_last_kernel_ms synchronises. */
KMP_API int kmp_batch_set_profiling(kmp_batch_ctx* ctx, int on);
/* which: 0 = zstd_match, 1 = zstd_entropy (mean over the launches of the last compress batch), 2 = zstd_decode */
KMP_API int kmp_batch_last_kernel_ms(kmp_batch_ctx* ctx, int which, float* ms);
/* kmp_zstd_compress_batch cuts a large batch into chunks so that k_zstd_entropy of one chunk runs beside
 * k_zstd_match of the next (second HIP stream inside the context; the caller's stream still sees the whole
 * batch finished).  Returns how many launches of each kernel the last batch used. */
KMP_API int kmp_batch_last_chunks(kmp_batch_ctx* ctx);
/* A context for slices above 128 KiB compresses block by block (libzstd decides each block's size from the bytes
 * produced so far): block rounds of the last batch. */
KMP_API int kmp_batch_last_rounds(kmp_batch_ctx* ctx);
/* diagnostic (no reference counterpart): milliseconds of blocks x 256 threads x iters random 4-byte read + write pairs over the
 * device region [d_region, d_region + bytes) -- the access pattern of the level-3 parser's tables.  The rate differs between
 * regions of one device's HBM (tools/region_probe.py). */
KMP_API int kmp_debug_probe_region(void* d_region, size_t bytes, uint32_t blocks, uint32_t iters, float* ms, void* hip_stream);

/* The batch for callers that hold HOST memory (the JVM): slices h_src[in_off[i] .. + in_len[i]) of at most 128 KiB in, frames at
 * h_dst[out_off[i] ..) out (out_cap[i] bytes of room: kmp_zstd_compress_bound(in_len[i]) always suffices), out_len[i] = frame
 * size (0 with KMP_ERR_CAPACITY when the room was too small).  Levels -131072 .. 4 (0 = 3): what kmp_zstd_compress_batch_level serves
 * for one-block slices -- the frames are what ZstdCompressor(level) returns for each slice on its own; a level-4 batch's slices of
 * 16 KiB or less (that level's "greedy" size class) come back with out_len 0 and the call returns KMP_ERR_CAPACITY after every other
 * slice has been compressed.  out_len is written for every slice whatever the return value.
 * A small batch goes through pinned staging and one device batch.  A large level-3 batch (more than 2 * KMP_HOST_BATCH_SLICES slices)
 * goes through the pipelined bulk compressor: eight pieces that run side by side on the device (kmp_zstd_compress_batch_pieces), each
 * piece's copy in, kernels and copy out overlapping the others'.  Memory the caller has made page-stable -- kmp_host_register, or its
 * own pinned allocation -- is read and written by the device directly; pageable memory goes through pinned staging filled and emptied
 * by KMP_HOST_BULK_WORKERS host threads.  jni/zstd/BatchWrapper.cpp binds the calls for direct ByteBuffers.  Reference counterpart:
 * none -- the reference compresses one slice per call (ZstdWrapper.kt:35-46); this is the call a maintainer adds to hand the GPU a batch.
 * kmp_zstd_compress_stream itself coalesces the closing calls of concurrent contexts into such batches (KMP_COALESCE=0: off;
 * KMP_COALESCE_US: the gather window, default 150 microseconds; KMP_COALESCE_MAX: slices per batch, default 256). */
KMP_API int kmp_zstd_compress_host_batch(int device, int level, const void* h_src, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                         void* h_dst, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len);
/* frames in, content out (out_cap[i] = room, at most 128 KiB); status[i] = libzstd's error number for entry i, 0 = fine */
KMP_API int kmp_zstd_decompress_host_batch(int device, const void* h_src, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                           void* h_dst, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len, uint32_t* status);
/* Page-stable caller memory: pins [ptr, ptr + bytes) and maps it for the device (hipHostRegister), so that the host-batch calls read
 * slices from it and write frames into it over PCIe without a staging copy -- what the reference's JNI layer borrows and releases
 * around every call (GetByteArrayElements / ReleaseByteArrayElements, Wrapper.cpp:92-118), borrowed once for a buffer that outlives
 * many calls: a direct ByteBuffer a Kotlin caller reuses.  Unregister before the memory is freed. */
KMP_API int kmp_host_register(void* ptr, size_t bytes);
KMP_API int kmp_host_unregister(void* ptr);
/* Gives back what the host-batch calls and the coalescer of the streaming entry points hold on `device` (-1: every device): pinned
 * staging, device buffers, batch contexts.  The next call that needs them makes them again.  Not while such a call is running. */
KMP_API int kmp_host_engines_release(int device);

/* diagnostic (no reference counterpart): random 4-byte loads per second, and load + store-into-the-same-word pairs per second,
 * over the context's level-3 team tables where they lie, measured once when the context was created (both 0 for a context
 * whose tables are below 4 GiB).  bench.py prices the parser's measured memory requests with them. */
KMP_API int kmp_batch_table_rates(kmp_batch_ctx* ctx, float* reads_per_s, float* pairs_per_s);

/* diagnostic: the per-slice parser records (32 bytes each: sequences, literal bytes, trailing literals, long-length info, status and
 * two spare words -- with KMP_MATCH_FLAGS bit 8 the device's 100 MHz clock when the slice was finished) of the last one-block
 * zstd batch, copied to host memory after the device has gone idle */
KMP_API int kmp_debug_copy_meta(kmp_batch_ctx* ctx, void* h_dst, uint32_t n);

KMP_API const char* kmp_last_error(void);
KMP_API const char* kmp_version(void);

#ifdef __cplusplus
}
#endif
#endif
