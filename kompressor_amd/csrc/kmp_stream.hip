// kmp_stream.hip -- include/kompressor_hip.h part 1: the streaming-compatible single-slice API, one function per JNI export of
// the reference (kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:10-196, kompressor-zlib--nativelib/.../jni/Wrapper.cpp:10-153),
// and the batch for callers that hold host memory (kmp_coalesce.h).  Host code only: everything on the device goes through the
// batched API of kmp_batch.hip / kmp_deflate.hip.
#include "kmp_internal.h"
#include <mutex>
#include <new>
#include <vector>

// (constants of the kernels' headers this unit needs, without their device code)
#define KX_BLOCK_MAX (128u * 1024u)
#define KX_MAX_DICT (128u * 1024u - 512u)
#define KD_MAX_SLICE (1u << 30)

#include "kmp_coalesce.h"

// --------------------------------------------------------------------------
// streaming-compatible single-slice API (mirrors libzstd's calling convention)
// --------------------------------------------------------------------------
#define KERRC(code) ((size_t)0 - (size_t)(code))
enum { ZE_GENERIC = 1, ZE_prefix_unknown = 10, ZE_frameParameter_unsupported = 14, ZE_corruption_detected = 20,
       ZE_dictionary_corrupted = 30, ZE_parameter_unsupported = 40, ZE_parameter_outOfBound = 42, ZE_stage_wrong = 60, ZE_memory_allocation = 64,
       ZE_dstSize_tooSmall = 70, ZE_srcSize_wrong = 72, ZE_maxCode = 120 };

extern "C" unsigned kmp_zstd_is_error(size_t code) { return code > KERRC(ZE_maxCode); }
extern "C" const char* kmp_zstd_get_error_name(size_t code)
{
    if (!kmp_zstd_is_error(code)) return "No error detected";
    switch ((int)(0 - code)) {
    case 1: return "Error (generic)";
    case 10: return "Unknown frame descriptor";
    case 12: return "Version not supported";
    case 14: return "Unsupported frame parameter";
    case 16: return "Frame requires too much memory for decoding";
    case 20: return "Data corruption detected";
    case 22: return "Restored data doesn't match checksum";
    case 24: return "Header of Literals' block doesn't respect format specification";
    case 30: return "Dictionary is corrupted";
    case 32: return "Dictionary mismatch";
    case 34: return "Cannot create Dictionary from provided samples";
    case 40: return "Unsupported parameter";
    case 41: return "Unsupported combination of parameters";
    case 42: return "Parameter is out of bound";
    case 44: return "tableLog requires too much memory : unsupported";
    case 46: return "Unsupported max Symbol Value : too large";
    case 48: return "Specified maxSymbolValue is too small";
    case 49: return "This mode cannot generate an uncompressed block";
    case 50: return "pledged buffer stability condition is not respected";
    case 60: return "Operation not authorized at current processing stage";
    case 62: return "Context should be init first";
    case 64: return "Allocation error : not enough memory";
    case 66: return "workSpace buffer is not large enough";
    case 70: return "Destination buffer is too small";
    case 72: return "Src size is incorrect";
    case 74: return "Operation on NULL destination buffer";
    case 80: return "Operation made no progress over multiple calls, due to output buffer being full";
    case 82: return "Operation made no progress over multiple calls, due to input being empty";
    case 100: return "Frame index is too large";
    case 102: return "An I/O error occurred when reading/seeking";
    case 104: return "Destination buffer is wrong";
    case 105: return "Source buffer is wrong";
    case 106: return "Block-level external sequence producer returned an error code";
    case 107: return "External sequences are not valid";
    default: return "Unspecified error code";
    }
}

// device staging shared by the two stream contexts
struct stream_dev {
    kmp_batch_ctx* batch; u8* d_in; u8* d_out; u64* d_off; u32* d_len; size_t in_cap, out_cap; u32 tier;
};
// the staging buffers live on the device the context was first used on: later calls may come from a thread whose
// current device is another one (the reference frees contexts on a cleaner thread, Cleaner.jvm.kt:23-36)
static bool stream_dev_select(const stream_dev& s) { return !s.batch || hipSetDevice(s.batch->device) == hipSuccess; }
static void stream_dev_free(stream_dev& s);
// staging for one slice / frame of at most `bytes` on either side: the 128 KiB tier first, the 2 MiB tier
// (frames of several blocks) when a larger one shows up
// (level 1 above 128 KiB wants a context of exactly its 512 KiB window: `exact` = that tier)
static size_t stream_dev_init(stream_dev& s, size_t bytes = 0, u32 exact = 0)
{
    if (s.batch && exact && s.tier == exact) return 0;
    if (s.batch && !exact && bytes + 1024 <= s.in_cap) return 0;
    if (s.batch) stream_dev_free(s);
    // 128 KiB, 2 MiB, then the next power of two that holds the slice
    u32 tier = exact ? exact : (bytes <= KMP_MAX_SLICE_BYTES) ? KMP_MAX_SLICE_BYTES : (2u << 20);
    while (!exact && (size_t)tier < bytes && tier < KMP_MAX_BIG_SLICE_BYTES) tier <<= 1;
    s.tier = tier;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return KERRC(ZE_GENERIC);          // the caller's current device, as a libzstd context lives where its caller runs
    if (kmp_batch_create(&s.batch, dev, 1, tier, 8) != KMP_OK) return KERRC(ZE_memory_allocation);
    s.in_cap = tier + (tier >> 7) + 1024; s.out_cap = tier + (tier >> 7) + 1024;
    if (hipMalloc((void**)&s.d_in, s.in_cap) != hipSuccess || hipMalloc((void**)&s.d_out, s.out_cap) != hipSuccess ||
        hipMalloc((void**)&s.d_off, 64) != hipSuccess || hipMalloc((void**)&s.d_len, 64) != hipSuccess) return KERRC(ZE_memory_allocation);
    return 0;
}
static void stream_dev_free(stream_dev& s)
{
    if (s.batch) { (void)hipSetDevice(s.batch->device); kmp_batch_destroy(s.batch); (void)hipFree(s.d_in); (void)hipFree(s.d_out); (void)hipFree(s.d_off); (void)hipFree(s.d_len); }
    memset(&s, 0, sizeof(s));
}

struct kmp_zstd_cctx {
    int level; std::vector<u8> in; std::vector<u8> out; size_t out_pos; int stage;   // 0 = collecting, 1 = flushing
    stream_dev dev;
    std::vector<u8> dict;                       // raw-content dictionary (ZSTD_CCtx_loadDictionary keeps a copy too)
    size_t fed_continue;                        // bytes that arrived with ZSTD_e_continue: > 0 makes it a streaming frame
    int end_was_empty;                          // the closing calls brought no data
};

extern "C" kmp_zstd_cctx* kmp_zstd_create_cctx(void)
{
    kmp_zstd_cctx* c = new (std::nothrow) kmp_zstd_cctx();
    if (!c) return nullptr;
    c->level = 3; c->out_pos = 0; c->stage = 0; memset(&c->dev, 0, sizeof(c->dev)); c->fed_continue = 0; c->end_was_empty = 0;
    return c;
}
extern "C" size_t kmp_zstd_free_cctx(kmp_zstd_cctx* c) { if (c) { stream_dev_free(c->dev); delete c; } return 0; }
extern "C" size_t kmp_zstd_cctx_set_parameter(kmp_zstd_cctx* c, int param, int value)
{
    if (!c) return KERRC(ZE_GENERIC);
    if (param != KMP_ZSTD_c_compressionLevel) return KERRC(ZE_parameter_unsupported);
    if (c->stage != 0 || !c->in.empty()) return KERRC(ZE_stage_wrong);
    if (value == 0) value = 3;
    if (value < -131072 || value > 10) return KERRC(ZE_parameter_unsupported);      // (4 .. 10: what arrives in one closing call of a size class the level is served in, decided when the stream closes)
    c->level = value;
    return 0;
}
extern "C" size_t kmp_zstd_cctx_load_dictionary(kmp_zstd_cctx* c, const void* dict, size_t dict_size)
{
    if (!c) return KERRC(ZE_GENERIC);
    if (c->stage != 0 || !c->in.empty()) return KERRC(ZE_stage_wrong);
    c->dict.clear();
    if (dict == nullptr || dict_size == 0) return 0;
    if (dict_size < 8 || dict_size > KX_MAX_DICT) return KERRC(ZE_parameter_unsupported);
    // (a dictionary in zstd's own format -- magic EC30A437 -- is loaded with its tables by the batch call; libzstd too only copies the bytes
    // here and meets a damaged header when the first frame starts: "Dictionary is corrupted" then)
    c->dict.assign((const u8*)dict, (const u8*)dict + dict_size);
    return 0;
}

// first_room: room in the output slice of the call that closed the stream; end_avail: the bytes that call brought
static size_t run_single_compress(kmp_zstd_cctx* c, size_t first_room, size_t end_avail)
{
    size_t const n = c->in.size();
    if (n > KMP_MAX_BIG_SLICE_BYTES) return KERRC(ZE_srcSize_wrong);
    bool const streaming = c->fed_continue > 0;          // data arrived with finish = false: libzstd did not know the size
    // libzstd compresses the caller's memory in place when its staging buffer is empty and the output slice has room for
    // ZSTD_compressBound of what the call brought; otherwise it stages the input in chunks of 128 KiB (kmp_zstd_compress_batch_reference)
    bool const in_place = first_room >= kmp_zstd_compress_bound(end_avail);
    u32 tail_direct = 0;
    if (streaming && in_place && end_avail != 0) {
        // libzstd's staging buffer: the window of a stream of unknown size + one block (level 3 / 4: 2 MiB, level 2: 1 MiB, level 1 and
        // the negative levels: 512 KiB)
        size_t const lap = ((c->level == 3 || c->level == 4) ? 17u : c->level == 2 ? 9u : 5u) * (size_t)KX_BLOCK_MAX;
        if ((n - end_avail) % lap == 0) tail_direct = (u32)end_avail;
    }
    if (streaming && !c->dict.empty()) return KERRC(ZE_parameter_unsupported);
    // level 4: what arrives in one closing call, up to 128 KiB (its greedy and double-fast rows) or above 256 KiB; the rest: CPU library
    if (c->level == 4 && (!c->dict.empty() || (!streaming && !(n <= 131072u || n > 262144u)))) return KERRC(ZE_parameter_unsupported);
    // levels 5 .. 10 (greedy / lazy / lazy2): one closing call of at most 128 KiB; 9 and 10 up to 16 KiB are "btlazy2": CPU library
    if (c->level >= 5 && (!c->dict.empty() || streaming || n > KMP_MAX_SLICE_BYTES || (c->level >= 9 && n <= 16384u))) return KERRC(ZE_parameter_unsupported);
    if (c->level < 0 && !c->dict.empty()) return KERRC(ZE_parameter_unsupported);
    // the plain case -- level 3, no dictionary, the whole slice at once, one block -- joins whatever other contexts are
    // closing right now: one batch for all of them (kmp_coalesce.h); the frame is the one this context would get alone
    if (!streaming && c->level == 3 && c->dict.empty() && n <= KMP_MAX_SLICE_BYTES && coalesce_enabled()) {
        int dev = 0;
        if (c->dev.batch) dev = c->dev.batch->device; else if (hipGetDevice(&dev) != hipSuccess) return KERRC(ZE_GENERIC);
        int const rc = coalesced_compress(dev, c->in.data(), (u32)n, &c->out);
        if (rc == KMP_OK) return 0;
        (void)hipGetLastError();                    // fall through: compress alone
    }
    bool const l1big = c->level != 3 && c->level != 4 && (streaming || n > KMP_MAX_SLICE_BYTES);      // level 1 / 2 / negative, frame of several blocks / stream (any length: beyond the level's window it slides as libzstd's does)
    if (!stream_dev_select(c->dev)) return KERRC(ZE_GENERIC);
    { size_t const e = stream_dev_init(c->dev, streaming && n <= KMP_MAX_SLICE_BYTES ? KMP_MAX_SLICE_BYTES + 1 : n); if (e) return e; }
    stream_dev& s = c->dev;
    u64 offs[2] = { 0, 0 }; u32 len = (u32)n, olen = 0;
    if (n && hipMemcpy(s.d_in, c->in.data(), n, hipMemcpyHostToDevice) != hipSuccess) return KERRC(ZE_GENERIC);
    if (hipMemcpy(s.d_off, offs, sizeof(offs), hipMemcpyHostToDevice) != hipSuccess) return KERRC(ZE_GENERIC);
    if (hipMemcpy(s.d_len, &len, 4, hipMemcpyHostToDevice) != hipSuccess) return KERRC(ZE_GENERIC);
    if (streaming) {
        if (tail_direct) {
            bool const neg = c->level < 0;
            u32 const strategy = (c->level == 3 || c->level == 4) ? 0u : neg ? 1u : (u32)c->level;
            if (zstd_compress_big(s.batch, s.d_in, s.d_off, s.d_len, 1, s.d_out, s.d_off + 1, s.d_len + 1, nullptr, 1u, strategy, tail_direct, neg ? (u32)(1 - c->level) : 0u, c->level == 4) != KMP_OK) return KERRC(ZE_GENERIC);
        } else
        if (kmp_zstd_compress_batch_stream_level(s.batch, s.d_in, s.d_off, s.d_len, 1, s.d_out, s.d_off + 1, s.d_len + 1, c->end_was_empty, c->level, nullptr) != KMP_OK) return KERRC(ZE_GENERIC);
    } else
    if (n > KMP_MAX_SLICE_BYTES && !in_place && c->dict.empty() && (c->level == 3 || c->level == 4 || l1big)) {
        // the reference's one-shot driver above 128 KiB: staged input
        if (kmp_zstd_compress_batch_reference(s.batch, s.d_in, s.d_off, s.d_len, 1, s.d_out, s.d_off + 1, s.d_len + 1, c->level, (u32)first_room, nullptr) != KMP_OK) return KERRC(ZE_GENERIC);
    } else
    if (c->level != 3) {
        if (!c->dict.empty()) return KERRC(ZE_parameter_unsupported);   // levels 1 / 2: no dictionary
        if (kmp_zstd_compress_batch_level(s.batch, s.d_in, s.d_off, s.d_len, 1, s.d_out, s.d_off + 1, s.d_len + 1, c->level, nullptr) != KMP_OK) return KERRC(ZE_GENERIC);
    } else
    if (!c->dict.empty()) {
        if (n > KMP_MAX_SLICE_BYTES) return KERRC(ZE_srcSize_wrong);          // frames of several blocks with a dictionary: CPU library
        if (kmp_zstd_compress_batch_dict(s.batch, s.d_in, s.d_off, s.d_len, 1, s.d_out, s.d_off + 1, s.d_len + 1,
                                         c->dict.data(), (u32)c->dict.size(), nullptr) != KMP_OK) {
            return dict_header_state(c->dict.data(), c->dict.size(), 0) < 0 ? KERRC(ZE_dictionary_corrupted) : KERRC(ZE_GENERIC);
        }
    } else
    if (kmp_zstd_compress_batch(s.batch, s.d_in, s.d_off, s.d_len, 1, s.d_out, s.d_off + 1, s.d_len + 1, nullptr) != KMP_OK) return KERRC(ZE_GENERIC);
    if (hipMemcpy(&olen, s.d_len + 1, 4, hipMemcpyDeviceToHost) != hipSuccess) return KERRC(ZE_GENERIC);
    if (olen == 0 || olen > s.out_cap) return KERRC(ZE_GENERIC);
    c->out.resize(olen);
    if (hipMemcpy(c->out.data(), s.d_out, olen, hipMemcpyDeviceToHost) != hipSuccess) return KERRC(ZE_GENERIC);
    return 0;
}

extern "C" size_t kmp_zstd_compress_stream(kmp_zstd_cctx* c, void* dst, size_t dst_size, size_t* dst_pos,
                                           const void* src, size_t src_size, size_t* src_pos, int end_op)
{
    if (!c || !dst_pos || !src_pos) return KERRC(ZE_GENERIC);
    if (*dst_pos > dst_size) return KERRC(ZE_dstSize_tooSmall);
    if (*src_pos > src_size) return KERRC(ZE_srcSize_wrong);
    if ((unsigned)end_op > 2u) return KERRC(ZE_parameter_outOfBound);
    if (c->stage == 0) {
        // one-shot semantics (finish=true from the first call, SliceTransform.kt:33-45): the whole
        // slice arrives before the frame can be produced, so input is collected until e_end
        size_t const avail = src_size - *src_pos;
        if (avail) { const u8* p = (const u8*)src + *src_pos; c->in.insert(c->in.end(), p, p + avail); *src_pos = src_size; }
        if (c->in.size() > KMP_MAX_BIG_SLICE_BYTES) return KERRC(ZE_srcSize_wrong);
        if (end_op != KMP_ZSTD_e_end) { c->fed_continue += avail; return 0; }
        c->end_was_empty = avail == 0;
        size_t const e = run_single_compress(c, dst_size - *dst_pos, avail);
        if (e) return e;
        c->stage = 1; c->out_pos = 0;
    } else if (*src_pos != src_size) {
        return KERRC(ZE_stage_wrong);      // new input while a finished frame is still being flushed
    }
    {
        size_t const room = dst_size - *dst_pos, left = c->out.size() - c->out_pos;
        size_t const k = room < left ? room : left;
        if (k) { memcpy((u8*)dst + *dst_pos, c->out.data() + c->out_pos, k); *dst_pos += k; c->out_pos += k; }
        size_t const remaining = c->out.size() - c->out_pos;
        if (remaining == 0) { c->stage = 0; c->in.clear(); c->out.clear(); c->out_pos = 0; c->fed_continue = 0; c->end_was_empty = 0; }
        return remaining;
    }
}

// ---- zlib-compatible one-shot compressor (raw deflate, level 6) ---------------------------------
struct kmp_zlib_cstream { int level, window_bits, mem_level, strategy; std::vector<u8> in; std::vector<u8> out; size_t out_pos; int stage; stream_dev dev; };

extern "C" kmp_zlib_cstream* kmp_zlib_create_compressor(int level, int window_bits, int mem_level, int strategy)
{
    // what deflateInit2 accepts (zlib deflate.c deflateInit2_): levels 1 .. 9 (-1 = default = 6; 0, stored blocks only, is not served),
    // windowBits -15 .. -9 raw, 9 .. 15 zlib wrapper (8 is taken as 9), 25 .. 31 (+ 16) gzip wrapper -- 8 without the zlib wrapper is an
    // error there too --, memLevel 1 .. 9, strategy 0 (the only one the reference passes: ZlibCompressor.jvm.kt:24)
    if (level == -1) level = 6;
    int const wrap = window_bits < 0 ? 0 : window_bits > 15 ? 2 : 1;
    int const wb = window_bits < 0 ? -window_bits : window_bits > 15 ? window_bits - 16 : window_bits;
    if (level < 1 || level > 9 || wb < 8 || wb > 15 || (wb == 8 && wrap != 1) || mem_level < 1 || mem_level > 9 || strategy != 0) return nullptr;
    kmp_zlib_cstream* z = new (std::nothrow) kmp_zlib_cstream();
    if (!z) return nullptr;
    z->level = level; z->window_bits = window_bits; z->mem_level = mem_level; z->strategy = strategy; z->out_pos = 0; z->stage = 0;
    memset(&z->dev, 0, sizeof(z->dev));
    return z;
}
extern "C" int kmp_zlib_free_compressor(kmp_zlib_cstream* z) { if (z) { stream_dev_free(z->dev); delete z; } return 0; }

extern "C" int kmp_zlib_compress_stream(kmp_zlib_cstream* z, void* dst, size_t dst_size, size_t* dst_pos,
                                        const void* src, size_t src_size, size_t* src_pos, int finish)
{
    enum { Z_OK_ = 0, Z_STREAM_END_ = 1, Z_STREAM_ERROR_ = -2, Z_DATA_ERROR_ = -3, Z_MEM_ERROR_ = -4, Z_BUF_ERROR_ = -5 };
    if (!z || !dst_pos || !src_pos || *dst_pos > dst_size || *src_pos > src_size) return Z_STREAM_ERROR_;
    if (z->stage == 0) {
        size_t const avail = src_size - *src_pos;
        if (avail) { const u8* p = (const u8*)src + *src_pos; z->in.insert(z->in.end(), p, p + avail); *src_pos = src_size; }
        if (z->in.size() > KD_MAX_SLICE) return Z_MEM_ERROR_;          // streams above 1 GiB are not served
        if (!finish) return avail ? Z_OK_ : Z_BUF_ERROR_;
        int const wrap = z->window_bits < 0 ? 0 : z->window_bits > 15 ? 2 : 1;
        int const wb = z->window_bits < 0 ? -z->window_bits : z->window_bits > 15 ? z->window_bits - 16 : z->window_bits;
        // (settings other than the default ones can expand the data by an eighth: deflateBound's other branch)
        size_t const room = (wb == 15 && z->mem_level == 8) ? z->in.size() : kmp_deflate_bound_params(z->in.size(), wb, z->mem_level);
        if (!stream_dev_select(z->dev) || stream_dev_init(z->dev, room)) return Z_MEM_ERROR_;
        stream_dev& s = z->dev;
        if (kmp_deflate_bound_params(z->in.size(), wb, z->mem_level) > s.out_cap) return Z_MEM_ERROR_;      // (the largest tier holds 1 GiB and an eighth more only for less)
        u64 offs[2] = { 0, 0 }; u32 len = (u32)z->in.size(), olen = 0;
        if (len && hipMemcpy(s.d_in, z->in.data(), len, hipMemcpyHostToDevice) != hipSuccess) return Z_MEM_ERROR_;
        if (hipMemcpy(s.d_off, offs, sizeof(offs), hipMemcpyHostToDevice) != hipSuccess) return Z_MEM_ERROR_;
        if (hipMemcpy(s.d_len, &len, 4, hipMemcpyHostToDevice) != hipSuccess) return Z_MEM_ERROR_;
        if (deflate_batch_impl(s.batch, s.d_in, s.d_off, s.d_len, 1, s.d_out, s.d_off + 1, s.d_len + 1, (u32)wrap, nullptr, z->level, wb, z->mem_level) != KMP_OK) return Z_MEM_ERROR_;
        if (hipMemcpy(&olen, s.d_len + 1, 4, hipMemcpyDeviceToHost) != hipSuccess) return Z_MEM_ERROR_;
        if (olen == 0 || olen > s.out_cap) return Z_DATA_ERROR_;
        z->out.resize(olen);
        if (hipMemcpy(z->out.data(), s.d_out, olen, hipMemcpyDeviceToHost) != hipSuccess) return Z_MEM_ERROR_;
        z->stage = 1; z->out_pos = 0;
    }
    size_t const room = dst_size - *dst_pos, left = z->out.size() - z->out_pos;
    size_t const k = room < left ? room : left;
    if (k) { memcpy((u8*)dst + *dst_pos, z->out.data() + z->out_pos, k); *dst_pos += k; z->out_pos += k; }
    if (z->out_pos == z->out.size()) { z->stage = 2; return Z_STREAM_END_; }
    return k ? Z_OK_ : Z_BUF_ERROR_;
}

struct kmp_zlib_dstream { int window_bits; std::vector<u8> in; std::vector<u8> out; size_t out_pos; int stage; kmp_batch_ctx* batch; };

extern "C" kmp_zlib_dstream* kmp_zlib_create_decompressor(int window_bits)
{
    // inflateInit2 semantics: -15..-8 raw, 8..15 zlib wrapper, 24..31 gzip, 40..47 zlib or gzip by the header
    bool const raw = window_bits <= -8 && window_bits >= -15, zl = window_bits >= 8 && window_bits <= 15;
    bool const gz = window_bits >= 24 && window_bits <= 31, any = window_bits >= 40 && window_bits <= 47;
    if (!raw && !zl && !gz && !any) return nullptr;
    kmp_zlib_dstream* z = new (std::nothrow) kmp_zlib_dstream();
    if (!z) return nullptr;
    z->window_bits = window_bits; z->out_pos = 0; z->stage = 0; z->batch = nullptr;
    return z;
}
extern "C" int kmp_zlib_free_decompressor(kmp_zlib_dstream* z) { if (z) { if (z->batch) kmp_batch_destroy(z->batch); delete z; } return 0; }

extern "C" int kmp_zlib_decompress_stream(kmp_zlib_dstream* z, void* dst, size_t dst_size, size_t* dst_pos,
                                          const void* src, size_t src_size, size_t* src_pos, int finish)
{
    enum { Z_OK_ = 0, Z_STREAM_END_ = 1, Z_STREAM_ERROR_ = -2, Z_DATA_ERROR_ = -3, Z_MEM_ERROR_ = -4, Z_BUF_ERROR_ = -5 };
    if (!z || !dst_pos || !src_pos || *dst_pos > dst_size || *src_pos > src_size) return Z_STREAM_ERROR_;
    if (z->stage == 0) {
        size_t const avail = src_size - *src_pos;
        if (avail) { const u8* p = (const u8*)src + *src_pos; z->in.insert(z->in.end(), p, p + avail); *src_pos = src_size; }
        if (!finish) return avail ? Z_OK_ : Z_BUF_ERROR_;           // the stream is decoded when the caller finishes it
        if (!z->batch) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || kmp_batch_create(&z->batch, dev, 1, 65536, 8) != KMP_OK) return Z_MEM_ERROR_;
        }
        size_t const n = z->in.size();
        u8* d_in = nullptr; u8* d_out = nullptr; u64* d_off = nullptr; u32* d_len = nullptr; int rc = Z_MEM_ERROR_;
        if (hipMalloc((void**)&d_in, n + 64) == hipSuccess && hipMalloc((void**)&d_off, 64) == hipSuccess && hipMalloc((void**)&d_len, 64) == hipSuccess) {
            // the decoded size is not known in advance: grow the capacity until the stream fits
            for (size_t cap = 256u << 10; cap <= (256u << 20); cap <<= 2) {
                if (d_out) { (void)hipFree(d_out); d_out = nullptr; }
                if (hipMalloc((void**)&d_out, cap + 64) != hipSuccess) break;
                u64 offs[2] = { 0, 0 }; u32 lens[4] = { (u32)n, (u32)cap, 0, 0 };
                if ((n && hipMemcpy(d_in, z->in.data(), n, hipMemcpyHostToDevice) != hipSuccess) ||
                    hipMemcpy(d_off, offs, sizeof(offs), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(d_len, lens, sizeof(lens), hipMemcpyHostToDevice) != hipSuccess) break;
                // (the window declared for a zlib stream bounds what its header may name: inflate.c, "invalid window size")
                int const zw = z->window_bits >= 40 ? z->window_bits - 32 : (z->window_bits >= 8 && z->window_bits <= 15) ? z->window_bits : 0;
                if (inflate_batch_impl(z->batch, d_in, d_off, d_len, 1, d_out, d_off + 1, d_len + 1, d_len + 2, (int32_t*)(d_len + 3),
                                       z->window_bits < 0 ? 0 : (z->window_bits <= 15 ? 1 : (z->window_bits <= 31 ? 2 : 3)), zw, nullptr) != KMP_OK) break;
                if (hipMemcpy(lens, d_len, sizeof(lens), hipMemcpyDeviceToHost) != hipSuccess) break;
                int const st = (int)lens[3];
                if (st == Z_BUF_ERROR_) continue;                    // output did not fit: next capacity
                if (st != 0) { rc = st; break; }
                z->out.resize(lens[2]);
                if (lens[2] && hipMemcpy(z->out.data(), d_out, lens[2], hipMemcpyDeviceToHost) != hipSuccess) break;
                rc = Z_OK_;
                break;
            }
        }
        (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_off); (void)hipFree(d_len);
        if (rc != Z_OK_) return rc == Z_MEM_ERROR_ ? Z_MEM_ERROR_ : Z_DATA_ERROR_;
        z->stage = 1; z->out_pos = 0;
    }
    size_t const room = dst_size - *dst_pos, left = z->out.size() - z->out_pos;
    size_t const k = room < left ? room : left;
    if (k) { memcpy((u8*)dst + *dst_pos, z->out.data() + z->out_pos, k); *dst_pos += k; z->out_pos += k; }
    if (z->out_pos == z->out.size()) { z->stage = 2; return Z_STREAM_END_; }
    return k ? Z_OK_ : Z_BUF_ERROR_;
}

struct kmp_zstd_dctx {
    std::vector<u8> in; std::vector<u8> out; size_t out_pos; int stage;   // 0 = collecting a frame, 1 = flushing
    stream_dev dev; u32* d_status;
    std::vector<u8> dict; u8* d_dict;                  // raw-content dictionary (ZSTD_DCtx_loadDictionary), host copy + device copy
};

extern "C" kmp_zstd_dctx* kmp_zstd_create_dctx(void)
{
    kmp_zstd_dctx* d = new (std::nothrow) kmp_zstd_dctx();
    if (!d) return nullptr;
    d->out_pos = 0; d->stage = 0; memset(&d->dev, 0, sizeof(d->dev)); d->d_status = nullptr; d->d_dict = nullptr;
    return d;
}
extern "C" size_t kmp_zstd_free_dctx(kmp_zstd_dctx* d) { if (d) { stream_dev_free(d->dev); if (d->d_dict) (void)hipFree(d->d_dict); if (d->d_status) (void)hipFree(d->d_status); delete d; } return 0; }
extern "C" size_t kmp_zstd_dctx_load_dictionary(kmp_zstd_dctx* d, const void* dict, size_t dict_size)
{
    if (!d) return KERRC(ZE_GENERIC);
    // raw-content dictionary: its bytes are the history before every frame decoded by this context (Wrapper.cpp:58-73)
    if (d->stage != 0 || !d->in.empty()) return KERRC(ZE_stage_wrong);
    if (d->d_dict) { (void)hipFree(d->d_dict); d->d_dict = nullptr; }
    d->dict.clear();
    if (dict == nullptr || dict_size == 0) return 0;
    if (dict_size > (8u << 20)) return KERRC(ZE_memory_allocation);
    // (a dictionary in zstd's own format is parsed by the batch call; libzstd builds its DDict here and reports a damaged header as a
    // failed allocation: ZSTD_DCtx_loadDictionary -> ZSTD_createDDict_advanced returns NULL)
    if (dict_header_state((const u8*)dict, dict_size, 1) < 0) return KERRC(ZE_memory_allocation);
    d->dict.assign((const u8*)dict, (const u8*)dict + dict_size);
    if (hipMalloc((void**)&d->d_dict, dict_size + 64) != hipSuccess) { d->d_dict = nullptr; d->dict.clear(); return KERRC(ZE_memory_allocation); }
    if (hipMemcpy(d->d_dict, d->dict.data(), dict_size, hipMemcpyHostToDevice) != hipSuccess) return KERRC(ZE_GENERIC);
    return 0;
}

// Size of the complete frame at p (n bytes available): 0 = need more input, KERRC(..) = malformed
static size_t frame_total_size(const u8* p, size_t n, size_t* contentSize)
{
    if (n < 5) return 0;
    if ((p[0] & 0xF0) == 0x50 && p[1] == 0x2A && p[2] == 0x4D && p[3] == 0x18) {       // skippable frame: magic, size, payload
        if (n < 8) return 0;
        u32 sz; memcpy(&sz, p + 4, 4);
        *contentSize = 0;
        return n < 8 + (size_t)sz ? 0 : 8 + (size_t)sz;
    }
    if (p[0] != 0x28 || p[1] != 0xB5 || p[2] != 0x2F || p[3] != 0xFD) return KERRC(ZE_prefix_unknown);
    u32 const fhd = p[4]; u32 const dictID = fhd & 3, checksum = (fhd >> 2) & 1, single = (fhd >> 5) & 1, fcsId = fhd >> 6;
    if (fhd & 0x08) return KERRC(ZE_frameParameter_unsupported);
    size_t pos = 5 + (single ? 0 : 1);
    static const u32 didSize[4] = { 0, 1, 2, 4 };
    pos += didSize[dictID];
    u32 const fcsSize = fcsId == 0 ? single : (fcsId == 1 ? 2 : fcsId == 2 ? 4 : 8);
    if (n < pos + fcsSize) return 0;
    u64 cs = (u64)-1;
    if (fcsSize == 1) cs = p[pos]; else if (fcsSize == 2) cs = (u64)(p[pos] | (p[pos + 1] << 8)) + 256;
    else if (fcsSize == 4) { u32 v; memcpy(&v, p + pos, 4); cs = v; } else if (fcsSize == 8) memcpy(&cs, p + pos, 8);
    *contentSize = (size_t)cs;
    pos += fcsSize;
    for (;;) {
        if (n < pos + 3) return 0;
        u32 const bh = p[pos] | (p[pos + 1] << 8) | (p[pos + 2] << 16);
        u32 const last = bh & 1, type = (bh >> 1) & 3, bsz = bh >> 3;
        if (type == 3) return KERRC(ZE_corruption_detected);
        pos += 3 + (type == 1 ? 1 : bsz);
        if (last) break;
    }
    pos += checksum ? 4 : 0;
    if (n < pos) return 0;
    return pos;
}

extern "C" size_t kmp_zstd_decompress_stream(kmp_zstd_dctx* d, void* dst, size_t dst_size, size_t* dst_pos,
                                             const void* src, size_t src_size, size_t* src_pos)
{
    if (!d || !dst_pos || !src_pos) return KERRC(ZE_GENERIC);
    if (*dst_pos > dst_size) return KERRC(ZE_dstSize_tooSmall);
    if (*src_pos > src_size) return KERRC(ZE_srcSize_wrong);
    if (d->stage == 0) {
        // take input until one whole frame is buffered
        size_t content = (size_t)-1;
        // the largest frame this path stages: 1 GiB of content, whose frame is at most that + 1/128 + block headers
        size_t const frame_max = (size_t)KMP_MAX_BIG_SLICE_BYTES + (KMP_MAX_BIG_SLICE_BYTES >> 7) + 1024;
        {   // take what is offered (never more than one largest frame beyond what is buffered), then hand back what lies beyond the frame's end
            size_t avail = src_size - *src_pos;
            if (d->in.size() >= frame_max) return KERRC(ZE_frameParameter_unsupported);      // still no complete frame: larger than served here
            if (avail > frame_max - d->in.size()) avail = frame_max - d->in.size();
            const u8* p = (const u8*)src + *src_pos;
            d->in.insert(d->in.end(), p, p + avail); *src_pos += avail;
        }
        size_t const total = frame_total_size(d->in.data(), d->in.size(), &content);
        if (kmp_zstd_is_error(total)) return total;
        // a declared content size beyond what is served is refused as soon as the header is in
        if (content != (size_t)-1 && content > KMP_MAX_BIG_SLICE_BYTES) return KERRC(ZE_frameParameter_unsupported);
        if (total && total < d->in.size()) { *src_pos -= d->in.size() - total; d->in.resize(total); }
        if (total == 0) return d->in.size() >= frame_max ? KERRC(ZE_frameParameter_unsupported) : 3;      // hint: more input expected
        // the plain case -- no dictionary, content size in the header, at most 128 KiB either way -- joins whatever other contexts
        // are decoding right now: one batch for all of them (kmp_coalesce.h)
        if (d->dict.empty() && content != (size_t)-1 && content <= KMP_MAX_SLICE_BYTES && total <= KMP_MAX_SLICE_BYTES && coalesce_enabled()) {
            int dev = 0;
            if (d->dev.batch) dev = d->dev.batch->device; else if (hipGetDevice(&dev) != hipSuccess) return KERRC(ZE_GENERIC);
            d->out.resize(content ? content : 1);
            u32 olen = 0, st = 0;
            int const rc = coalesced_decompress(dev, d->in.data(), (u32)total, d->out.data(), (u32)content, &olen, &st);
            if (rc == KMP_OK) {
                if (st) { d->out.clear(); return KERRC(st); }
                d->out.resize(olen);
                d->in.clear(); d->stage = 1; d->out_pos = 0;
                goto flush_output;
            }
            (void)hipGetLastError(); d->out.clear();                 // fall through: decode alone
        }
        if (!d->d_status && hipMalloc((void**)&d->d_status, 64) != hipSuccess) return KERRC(ZE_memory_allocation);
        u32 res[2] = { 0, 0 };
        // content size in the header: staged for exactly that; none (streaming frames): for 4 x the frame (2 MiB at least),
        // and again for 4 x as much while the decoder answers "destination too small", up to the 1 GiB served here
        size_t want = content != (size_t)-1 ? (content > d->in.size() ? content : d->in.size())
                                            : (4 * total > (size_t)(2u << 20) ? 4 * total : (size_t)(2u << 20));
        for (;;) {
            if (want > KMP_MAX_BIG_SLICE_BYTES) want = KMP_MAX_BIG_SLICE_BYTES;
            if (!stream_dev_select(d->dev)) return KERRC(ZE_GENERIC);
            { size_t const e = stream_dev_init(d->dev, want > total ? want : total); if (e) return e; }
            stream_dev& s = d->dev;
            if (total > s.in_cap) return KERRC(ZE_frameParameter_unsupported);
            u64 offs[2] = { 0, 0 }; u32 lens[2] = { (u32)total, (u32)s.out_cap };
            if (hipMemcpy(s.d_in, d->in.data(), total, hipMemcpyHostToDevice) != hipSuccess) return KERRC(ZE_GENERIC);
            if (hipMemcpy(s.d_off, offs, sizeof(offs), hipMemcpyHostToDevice) != hipSuccess) return KERRC(ZE_GENERIC);
            if (hipMemcpy(s.d_len, lens, sizeof(lens), hipMemcpyHostToDevice) != hipSuccess) return KERRC(ZE_GENERIC);
            if (kmp_zstd_decompress_batch_dict(s.batch, s.d_in, s.d_off, s.d_len, 1, s.d_out, s.d_off + 1, s.d_len + 1,
                                               d->d_status, d->d_status + 1, d->d_dict, (u32)d->dict.size(), nullptr) != KMP_OK) return KERRC(ZE_GENERIC);
            if (hipMemcpy(res, d->d_status, 8, hipMemcpyDeviceToHost) != hipSuccess) return KERRC(ZE_GENERIC);
            if (res[1] == (u32)ZE_dstSize_tooSmall && content == (size_t)-1 && want < KMP_MAX_BIG_SLICE_BYTES) { want *= 4; continue; }
            break;
        }
        stream_dev& s = d->dev;
        if (res[1]) return KERRC(res[1]);
        d->out.resize(res[0]);
        if (res[0] && hipMemcpy(d->out.data(), s.d_out, res[0], hipMemcpyDeviceToHost) != hipSuccess) return KERRC(ZE_GENERIC);
        d->in.clear(); d->stage = 1; d->out_pos = 0;
    }
flush_output:
    {
        size_t const room = dst_size - *dst_pos, left = d->out.size() - d->out_pos;
        size_t const k = room < left ? room : left;
        if (k) { memcpy((u8*)dst + *dst_pos, d->out.data() + d->out_pos, k); *dst_pos += k; d->out_pos += k; }
        size_t const remaining = d->out.size() - d->out_pos;
        if (remaining == 0) { d->stage = 0; d->out.clear(); d->out_pos = 0; }
        return remaining;
    }
}
