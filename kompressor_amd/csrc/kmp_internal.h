// kmp_internal.h -- what the translation units of libkompressor_hip.so share: the batch context, the error helpers, the
// knob macros and the few host functions one unit calls in another.  Nothing here is part of the C ABI
// (include/kompressor_hip.h); everything is built with hidden visibility.
//
//   kmp_batch.hip    zstd kernels, the batch context, kmp_zstd_{compress,decompress}_batch*, kmp_compact_batch
//   kmp_deflate.hip  DEFLATE / inflate kernels, kmp_deflate_* / kmp_zlib_* / kmp_gzip_compress_batch, kmp_inflate_batch
//   kmp_stream.hip   the streaming-compatible single-slice API (one function per JNI export of the reference) and the
//                    host-memory batch (kmp_coalesce.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "../../include/kompressor_hip.h"

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

struct KSeq; struct KSliceMeta; struct KFrameState; struct KdBest; struct KdSliceMeta; struct KdBlockInfo; struct KPreBlk; struct KPreLit;

// --------------------------------------------------------------------------
// errors
// --------------------------------------------------------------------------
extern thread_local std::string g_last_error;
int hip_fail(hipError_t e, const char* what);
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)
#define KMP_TRY(x) do { int r_ = (x); if (r_ != KMP_OK) return r_; } while (0)

// --------------------------------------------------------------------------
// knobs
// --------------------------------------------------------------------------
// The product reads a short, documented list of environment variables (INTEGRATION.md "Environment"): env_u32.  Everything
// that only ever served an experiment or an ablation is a KMP_KNOB: a compile-time constant in the product build (the
// variable's name does not even reach the binary) and an environment variable in the ablation build
// (-DKMP_ABLATIONS: libkompressor_hip_abl.so, which the tests of those paths load).
u32 env_u32(const char* name, u32 dflt);
#ifdef KMP_ABLATIONS
#define KMP_KNOB(name, dflt) env_u32(name, (u32)(dflt))
#else
#define KMP_KNOB(name, dflt) ((u32)(dflt))
#endif

// --------------------------------------------------------------------------
// batch context
// --------------------------------------------------------------------------
enum { KMP_MAX_CHUNKS = 4, KMP_MAX_PIECES = 8 };
struct kmp_batch_ctx {
    int device; u32 max_slices, max_slice_bytes; int G; int team_fixed; int table_retry = 0; u32 match_blocks, match_blocks_l3, nteams, l3_team_slots;
    u32 seq_cap, lit_cap, scratch_words;
    KSeq* seqs; u8* lits; KSliceMeta* meta; u32* scratch; u32* tables; u32* team_epoch; u32* counter;
    int profiling; hipEvent_t ev[14]; int ev_valid[7];
    // zstd compress pipeline: entropy coding of chunk i (second stream) runs beside the match kernel of chunk i+1
    // level 3, batches of more than half the team slots: one launch of each kernel or two chunks?  Tried once each on the
    // context's first two such batches (whole-step HIP events), then the faster stays -- the parse kernel's time differs
    // by 17 % between runs of the same box, and which setting wins depends on it.
    u32* tables_flat; u32* team_epoch_flat;     // levels 1 / 2 and the dictionary parser: one piece (they wait for latency and are slower over spread tables), allocated on first use when the level-3 tables are spread
    u32* tseg[4]; u32 tseg_n;                   // the team tables in four pieces spread over the context's arena (or tseg_n == 1: tables alone)
    u32* tables4 = nullptr; u32* epoch4 = nullptr; u32 teams4 = 0;      // level 4's table set (1 MiB per team), allocated by the first level-4 batch
    u32* big_tables4 = nullptr;                 // ... and on the block-chain path: 2 MiB per slice (2^18 + 2^18 entries)
    u32 table_layout;                           // which of the arena's layouts holds the table pieces (1 .. 8; 0: no arena)
    u8* arena; size_t arena_bytes;              // one allocation that holds seqs / lits / meta / scratch and the table pieces (else null: separate allocations)
    float place_ms; u32 place_tried;            // the team tables' placement: probe time of the region kept, candidates tried
    float table_reads_per_s, table_pairs_per_s; // random loads / load + store pairs per second over this context's team tables (k_table_probe at creation; 0 = not measured)
    int tune_state; int tune_pending; u32 tune_pick; float tune_ms[2]; hipEvent_t tune_ev[2];
    hipStream_t st2; hipEvent_t evm[KMP_MAX_CHUNKS][2], eve[KMP_MAX_CHUNKS][2], ev_join, ev_last_match; int have_last_match; u32 last_chunks;
    hipEvent_t ev_pre[KMP_MAX_CHUNKS + 1];      // decoder: [0] where the caller's stream stands, [1 + i] piece i pre-decoded
    // raw-deflate workspace, allocated on first use, for dfl_chunk slices at a time
    u32 dfl_chunk; u16* dfl_link; KdBest* dfl_best; u32* dfl_syms; KdSliceMeta* dfl_meta; u32* dfl_wr; u32* dfl_order;      // two halves of dfl_chunk slices each
    u32* dfl_fsyms; KdSliceMeta* dfl_fmeta; KdBlockInfo* dfl_fblocks; int dfl_ftried;          // levels 1 .. 3: symbols / blocks of 4 * dfl_chunk slices (one piece)
    hipStream_t dfl_sort_st[2]; hipEvent_t dfl_sorted[2][2], dfl_parsed[2][2]; int dfl_seg_sync;   // ... a sort stream per workspace half; span arrays in two copies (parity of the segment): sorted / parsed events
    u16* dfl_rank; u32* dfl_state; u32* dfl_maxlen;                                             // slices above 64 KiB (deflate_lazy.h, segments): the sort's ranks, the parse's state between segments, the batch's longest slice
    u32 dfl_pos_cap, dfl_blk_cap; KdBlockInfo* dfl_blocks;                                      // positions / blocks per slice in them
    hipEvent_t dfl_searched[2], dfl_done[2]; int dfl_events;
    // frames of several blocks (max_slice_bytes above 128 KiB): per-slice state carried between the block rounds
    // raw-content dictionary of the last kmp_zstd_compress_batch_dict call: device copy + CDict tables (built on the host)
    u8* d_dict; u32* d_dictL; u32* d_dictS; u32 dict_size; u64 dict_hash; u32 cdW, cdH, cdC, cdM;
    struct KDictDPrior* d_dprior; const void* ddict_ptr; u32 ddict_size; u64 ddict_hash; u32 ddict_off, ddict_id, ddict_rep[3];      // ... for the decoder (the caller's dictionary lies in device memory: its head is read back once per dictionary)
    u32* lz_srt; u32* lz_wr; u32* lz_order; u32 lz_pos_cap, lz_chunk;       // zstd levels 5 .. 10 (and level 4's slices up to 16 KiB): the sorted positions' records (KLazyRec, 16 bytes), where each position stands (one piece's worth)
    struct KDictPrior* d_prior; u32 dict_content; u32 dict_rep[2];      // a formatted dictionary: its tables on the device, the size of its content part, its repeat offsets
    int big; int big_G; KFrameState* fstate; u32* hufct; u32* big_tables; u32* remaining; u32* big_counters; u32 last_rounds;
    u32 cus;                                   // compute units of the device
    // decoder: sequences decoded ahead of k_zstd_decode (allocated on first use; pre_tried: do not try again)
    u64* pre_stage; KPreBlk* pre_blk; u32* pre_nblk; u32 pre_seq_cap, pre_blk_cap; int pre_tried;
    u32 pre_slices;                             // entries the staging areas hold (a larger batch is decoded in pieces)
    u32* pre_sort;                              // per staged entry: key, slot -> entry map; then 256 bucket counters
    u8* pre_lits; KPreLit* pre_lit; u32* pre_nlit; u32 pre_lit_cap;
    u32* len_ok; u32* d_status;                // sanitised slice lengths of the running batch; status word (KMP_STATUS_*)
    // one batch at a time per context: a batch queued on another stream waits for the previous one's last kernel
    hipEvent_t ev_done; int have_done;
    // ... or its pieces did, each on a stream of its own (kmp_zstd_compress_batch_pieces): the next batch waits for all of them
    hipEvent_t ev_piece[KMP_MAX_PIECES]; u32 pieces_pending;
    // experiment switches, read from the environment once, when the context is created
    struct { u32 chunks, match_flags, entropy_pad, first_permille, fast_first_permille, entropy_flags, decode_flags, decode_pad, big_rounds, big_spw,
                 dfl_chunk, dfl_chain_waves, dfl_serial, dfl_flags, decode_pre, decode_sort, decode_pieces, decode_stage_slices, inflate_pre, inflate_pieces, autotune, match_v2, fuse; } knob;
};

// --------------------------------------------------------------------------
// host functions shared between the translation units
// --------------------------------------------------------------------------
// A batch begins: it waits for the previous batch of this context (whatever stream that ran on), and its kernels get
// the sanitised lengths (k_len_guard).  A batch ends: oversized slices lose their frames, the event is recorded.
int batch_wait_previous(kmp_batch_ctx* c, hipStream_t st);
int batch_begin(kmp_batch_ctx* c, hipStream_t st, const u32* d_in_len, u32 n, u32 cap);
int batch_end(kmp_batch_ctx* c, hipStream_t st, const u32* d_in_len, u32 n, u32 cap, u32* d_out_len, const KSliceMeta* meta);
// a context whose arena is packed (the engines of the host-memory batch: kmp_coalesce.h)
int batch_create_packed(kmp_batch_ctx** out, int device, uint32_t max_slices, uint32_t max_slice_bytes);
// staging of the pre-decode kernels (shared by the zstd decoder and inflate), allocated on first use
u32 env_pre_min_batch();
void ensure_pre_staging(kmp_batch_ctx* c);
// lane slots of the lane-per-entry kernels in order of size: key / rank / permutation (the zstd pre-decoders' counting sort;
// len_shift = 0: keyed by the frames' sequence counts, else by entry bytes >> len_shift)
// ... with the keys (0 .. 255) and their histogram already there: bucket starts + permutation, largest keys first
int size_sort_keys(kmp_batch_ctx* c, hipStream_t st, u32 m, u32* key, u32* hist, u32* perm);
int size_sort(kmp_batch_ctx* c, hipStream_t st, const u8* src, const u64* in_off, const u32* in_len, u32 m, u32* key, u32* hist, u32* perm, u32 len_shift);
// a level-3 batch in pieces, each on a stream of its own (kmp_zstd_compress_batch_pieces = begin + every piece + end)
int pieces_begin(kmp_batch_ctx* c, u32 pieces, void* const* hip_streams);
int piece_enqueue(kmp_batch_ctx* c, u32 p, u32 pieces, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                  void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, hipStream_t st);
void pieces_end(kmp_batch_ctx* c, u32 pieces);
// frames out of the strided device layout straight into registered host memory (device-visible address h_dst_dev)
int scatter_frames(kmp_batch_ctx* c, hipStream_t st, const u8* d_src, const u64* d_in_off, u32* d_len, u32 n, u8* h_dst_dev, const u64* d_h_off, const u32* d_h_cap, u32* d_status);
// frames of several blocks (kmp_batch.hip); stream: KFrameArgs.stream
int zstd_compress_big(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                      uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, hipStream_t st, u32 stream, u32 strategy, u32 tail_direct = 0, u32 fast_step0 = 0, bool level4 = false);
int dict_header_state(const unsigned char* dict, size_t dict_size, int for_decoder);       // 1: a well-formed dictionary in zstd's own format, 0: raw content, -1: the magic with a damaged header
int inflate_batch_impl(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                       void* d_dst, const uint64_t* d_out_off, const uint32_t* d_out_cap, uint32_t* d_out_len, int32_t* d_status,
                       int format, int window_bits, void* hip_stream);
int deflate_batch_impl(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                       uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, u32 format, void* hip_stream, int level = 6,
                       int window_bits = 15, int mem_level = 8);
