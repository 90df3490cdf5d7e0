/*
 * claxon_hip.h -- C ABI of the MI355X (gfx950) batched FLAC frame decoder.
 *
 * This is the drop-in boundary for ONE path of ruuda/claxon: per-subframe
 * decode (src/subframe.rs) plus stereo decorrelation (src/frame.rs), executed
 * for a whole batch of independent frames by hand-written HIP kernels.
 * Every entry point below names the reference interface it replaces
 * (file:line into the claxon v0.4.3 tree).  Plain pointers and sizes only;
 * no exceptions, panics or C++/torch types cross this boundary.
 *
 * Threading: a clx_ctx / clx_batch is NOT thread safe (the reference takes
 * `&mut self` everywhere, frame.rs:667); distinct contexts (one per GPU / per
 * HIP stream) may be used concurrently.
 *
 * Memory: "device" pointers are HIP device pointers on the context's GPU.
 * A device arena must be 16-byte aligned and its ALLOCATION must cover
 * arena_len + 16 bytes rounded up to a multiple of 16 (the kernels read the
 * bitstream in aligned 8/16-byte granules; bytes past arena_len are never
 * interpreted).
 */
#ifndef CLAXON_HIP_H
#define CLAXON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLX_VERSION_MAJOR 0
#define CLX_VERSION_MINOR 2
#define CLX_VERSION_PATCH 0

/* ------------------------------------------------------------------------
 * Error convention.  Replaces `claxon::Error` (error.rs:18-32):
 *   IoError(io::Error) | FormatError(&'static str) | Unsupported(&'static str)
 * Equality in the reference is variant + string (error.rs:34-45) and its tests
 * compare strings (tests/testsamples.rs:412), so the strings are API: each
 * clx_msg maps 1:1 to the reference's message (clx_message()).
 * ---------------------------------------------------------------------- */
typedef enum clx_status {
    CLX_OK          = 0,
    CLX_IO_ERROR    = 1,  /* Error::IoError(UnexpectedEof): bits ran out mid-frame (input.rs:139-142, 242) */
    CLX_FORMAT_ERROR= 2,  /* Error::FormatError(msg) */
    CLX_UNSUPPORTED = 3,  /* Error::Unsupported(msg) */
    CLX_END_OF_STREAM = 4,/* Ok(None): EOF before the two sync bytes (frame.rs:140-143) */
    CLX_API_ERROR   = 5   /* bad argument / HIP failure; no reference analogue */
} clx_status;

typedef enum clx_msg {
    CLX_MSG_NONE = 0,
    CLX_MSG_UNEXPECTED_EOF,              /* io::ErrorKind::UnexpectedEof */
    /* subframe.rs */
    CLX_MSG_SUBFRAME_HEADER_INVALID,     /* subframe.rs:32  */
    CLX_MSG_SUBFRAME_HEADER_RESERVED,    /* subframe.rs:47,55 */
    CLX_MSG_WASTED_BITS_EXCEED_31,       /* subframe.rs:83  */
    CLX_MSG_NO_NON_WASTED_BITS,          /* subframe.rs:199 */
    CLX_MSG_RESIDUAL_RESERVED,           /* subframe.rs:245 */
    CLX_MSG_INVALID_PARTITION_ORDER,     /* subframe.rs:263 */
    CLX_MSG_INVALID_RESIDUAL,            /* subframe.rs:276 */
    CLX_MSG_FIXED_ORDER_GT_BLOCK,        /* subframe.rs:500 */
    CLX_MSG_LPC_ORDER_GT_BLOCK,          /* subframe.rs:663 */
    CLX_MSG_QLP_PRECISION_INVALID,       /* subframe.rs:674 */
    CLX_MSG_UNENCODED_BINARY,            /* subframe.rs:318,366 (Unsupported) */
    CLX_MSG_NEGATIVE_QLP_SHIFT,          /* subframe.rs:688-690 (Unsupported) */
    /* frame.rs */
    CLX_MSG_FRAME_CRC_MISMATCH,          /* frame.rs:761 */
    CLX_MSG_FRAME_HEADER_CRC_MISMATCH,   /* frame.rs:300 */
    CLX_MSG_FRAME_SYNC_MISSING,          /* frame.rs:148 */
    CLX_MSG_FRAME_HEADER_RESERVED,       /* frame.rs:157,177,224,236,241 */
    CLX_MSG_FRAME_HEADER_INVALID,        /* frame.rs:210 */
    CLX_MSG_FRAME_NUMBER_TOO_LARGE,      /* frame.rs:255 */
    CLX_MSG_BLOCK_SIZE_EXCEEDS_65535,    /* frame.rs:273 */
    CLX_MSG_INVALID_VARINT,              /* frame.rs:82,98 */
    CLX_MSG_NO_BPS_IN_HEADER,            /* frame.rs:691 (Unsupported) */
    /* lib.rs / metadata.rs (stream open; host only) */
    CLX_MSG_INVALID_STREAM_HEADER,       /* lib.rs:200 */
    CLX_MSG_ID3_HEADER,                  /* lib.rs:198 */
    CLX_MSG_STREAMINFO_MISSING,          /* lib.rs:247 */
    CLX_MSG_SECOND_STREAMINFO,           /* lib.rs:268 */
    CLX_MSG_STREAMINFO_LENGTH,           /* metadata.rs:272 */
    CLX_MSG_INVALID_METADATA_BLOCK_TYPE, /* metadata.rs:305 */
    CLX_MSG_MIN_BLOCK_GT_MAX_BLOCK,      /* metadata.rs:361 */
    CLX_MSG_BLOCK_SIZE_LT_16,            /* metadata.rs:364 */
    CLX_MSG_MIN_FRAME_GT_MAX_FRAME,      /* metadata.rs:367 */
    CLX_MSG_INVALID_SAMPLE_RATE,         /* metadata.rs:373 */
    CLX_MSG_APPLICATION_BLOCK_TOO_SHORT, /* metadata.rs:527 */
    CLX_MSG_APPLICATION_BLOCK_TOO_LARGE, /* metadata.rs:534 (Unsupported) */
    /* VORBIS_COMMENT (FLAC tags), metadata.rs:402-513 */
    CLX_MSG_VC_TOO_SHORT,                /* metadata.rs:406 */
    CLX_MSG_VC_TOO_LARGE,                /* metadata.rs:423 (Unsupported) */
    CLX_MSG_VC_VENDOR_TOO_LONG,          /* metadata.rs:431 */
    CLX_MSG_VC_TOO_MANY_ENTRIES,         /* metadata.rs:448 */
    CLX_MSG_VC_COMMENT_TOO_LONG,         /* metadata.rs:462 */
    CLX_MSG_VC_NAME_INVALID_BYTE,        /* metadata.rs:491 */
    CLX_MSG_VC_NO_EQUALS,                /* metadata.rs:498 */
    CLX_MSG_VC_EXCESS_DATA,              /* metadata.rs:503 */
    CLX_MSG_VC_WRONG_COUNT,              /* metadata.rs:507 */
    CLX_MSG_VC_NOT_UTF8,                 /* error.rs:92 */
    CLX_MSG_SECOND_VORBIS_COMMENT,       /* lib.rs:258 */
    CLX_MSG_COUNT
} clx_msg;

/* The reference's exact message string for `msg` ("" for CLX_MSG_NONE). */
const char* clx_message(uint32_t msg);
/* The status (error variant) the reference attaches to `msg`. */
int clx_message_status(uint32_t msg);
/* (major<<16)|(minor<<8)|patch */
uint32_t clx_version(void);

/* ------------------------------------------------------------------------
 * Frame headers (host).  Replaces `read_frame_header_or_eof` (frame.rs:131-316)
 * and `read_var_length_int` (frame.rs:64-105), incl. the CRC-8 check
 * (crc.rs:62-93) -- byte-aligned, ~6-16 bytes per frame, not accelerated.
 * ---------------------------------------------------------------------- */
enum { CLX_CH_INDEPENDENT = 0, CLX_CH_LEFT_SIDE = 1, CLX_CH_RIGHT_SIDE = 2, CLX_CH_MID_SIDE = 3 };

typedef struct clx_frame_header {
    uint64_t time;               /* first sample number: block_size*frame_number or sample number (frame.rs:771-774) */
    uint32_t sample_rate;        /* 0 = "get from streaminfo" (frame.rs:193) */
    uint32_t frame_or_sample_lo; /* low 32 bits of the coded number */
    uint16_t block_size;
    uint16_t header_bytes;       /* bytes consumed incl. the CRC-8 */
    uint8_t  n_channels;
    uint8_t  channel_assignment; /* CLX_CH_* */
    uint8_t  bps;                /* 0 = "get from streaminfo" -> Unsupported at decode (frame.rs:687-692) */
    uint8_t  variable_blocking;
} clx_frame_header;

/* Parse one frame header from `p[0..avail)`.  Returns a clx_status; on error
 * *msg holds the clx_msg.  CLX_END_OF_STREAM iff fewer than 2 bytes are
 * available (frame.rs:140-143).  `check_crc`=0 mirrors cfg(fuzzing). */
int clx_parse_frame_header(const uint8_t* p, size_t avail, int check_crc,
                           clx_frame_header* out, uint32_t* msg);

/* CRC helpers (crc.rs:89-113): CRC-8 poly 0x07, CRC-16 poly 0x8005, init 0, MSB first. */
uint8_t  clx_crc8(const uint8_t* p, size_t n);
uint16_t clx_crc16(const uint8_t* p, size_t n);

/* ------------------------------------------------------------------------
 * Batch decode (device).  Replaces, for n frames at once, the body of
 * `FrameReader::read_next_or_eof` between header parse and footer
 * (frame.rs:699-750): `subframe::decode` per channel on one bit cursor
 * (subframe.rs:184-228, called from frame.rs:708,715,716,725,726,734,735)
 * and `decode_left_side/right_side/mid_side` (frame.rs:319-389).
 * ---------------------------------------------------------------------- */
typedef struct clx_frame_desc {
    uint64_t byte_off;      /* frame start (sync code) in the arena */
    uint32_t max_bytes;     /* bytes readable from byte_off (rest of stream, or packet length) */
    uint16_t header_bytes;  /* first subframe starts at byte_off+header_bytes, bit 0 */
    uint16_t block_size;    /* 1..65535 */
    uint8_t  n_channels;    /* 1..8 */
    uint8_t  channel_assignment; /* CLX_CH_* */
    uint8_t  bps;           /* header bps (side channels decode at bps+1 on device) */
    uint8_t  reserved[5];
} clx_frame_desc;           /* 24 bytes */

typedef struct clx_frame_result {
    int32_t  status;        /* clx_status */
    uint32_t msg;           /* clx_msg */
    uint64_t end_bit;       /* bit offset (from byte_off) just past the last subframe;
                               the CRC-16 sits at byte ceil(end_bit/8) (frame.rs:744-754) */
} clx_frame_result;         /* 16 bytes */

typedef struct clx_ctx   clx_ctx;
typedef struct clx_batch clx_batch;

/* Create a context bound to HIP device `device`.  Fails (CLX_API_ERROR) when no
 * gfx950 device / HIP runtime is usable: there is NO CPU fallback. */
int  clx_create(int device, clx_ctx** out);
void clx_destroy(clx_ctx* ctx);
/* Human-readable text of the last CLX_API_ERROR on this context. */
const char* clx_last_error(const clx_ctx* ctx);

/* Flags for clx_decode_frames */
enum {
    CLX_ARENA_ON_DEVICE = 1u << 0,   /* arena is a device pointer (else host; copied H2D) */
    CLX_OUT_ON_DEVICE   = 1u << 1,   /* out is a device pointer (else host; copied D2H)   */
    CLX_VERIFY_CRC16    = 1u << 2,   /* also verify each frame's CRC-16 footer on device
                                        (frame.rs:752-763); mismatch -> CLX_MSG_FRAME_CRC_MISMATCH.
                                        Without it the footer is still read, as the reference does
                                        under cfg(fuzzing) (frame.rs:754): a frame whose two footer
                                        bytes lie beyond max_bytes fails with CLX_MSG_UNEXPECTED_EOF */
    /* Kernel path.  Default (neither bit): chosen from the batch shape.
     * WAVES: one wavefront per frame, wave-parallel Rice decode (lowest latency for a few frames).
     * LANES: one lane per subframe, lane-serial fused decode (highest throughput for many frames;
     *        needs arena_len < 4 GiB: an explicit CLX_PATH_LANES fails beyond, the default falls back to WAVES). */
    CLX_PATH_WAVES      = 1u << 3,
    CLX_PATH_LANES      = 1u << 4,
    CLX_PCM_ON_DEVICE   = 1u << 5,   /* clx_interleave: `pcm` is a device pointer (else host; copied D2H) */
    /* Build of the lane path's decode kernel (with CLX_PATH_LANES; default: by batch size): the fused one-wave
     * kernel (throughput), or the two-wave split kernel (latency). */
    CLX_LANES_FUSED     = 1u << 6,
    CLX_LANES_SPLIT     = 1u << 7,
    /* Build of the wave path's predictor kernel (default: by batch size): the multi-wave latency build
     * (clx_k_predict) or the one-wave throughput build (clx_k_predict_1w / _1w_hi). */
    CLX_K2_LATENCY      = 1u << 8,
    CLX_K2_THROUGHPUT   = 1u << 9,
    /* The fused lane build runs clx_k_lean first (the 16-bit tier: waves of <= 16-bit FIXED / LPC subframes of at most 12 taps in
     * aligned rows), then clx_k_lean24 (the split tier: <= 24 bits, <= 32 taps) when the batch holds frames of more than 16 bits,
     * and the general kernels on the groups those leave.  This flag leaves the tiers out: every group goes through the general
     * kernels (test and comparison target). */
    CLX_LANES_GENERAL   = 1u << 10,
    /* Waves composed by content (fused lane build; round 4).  A wave of 64 subframes runs the predictor build of its HIGHEST order,
     * the masked form of its turns when ONE lane holds a constant or verbatim subframe, the generic stereo form unless ALL its
     * pairs are mid/side -- so behind the scan a small kernel (clx_k_compose) re-deals the frames of a window (up to 16 384 stereo
     * frames of one block size) to the lanes by class: predictor order <= 4 / <= 8 / <= 12 / more, constant or verbatim subframes,
     * channel assignment.  Which frame a lane decodes changes, nothing else: every frame still goes to its own place in `out`.
     * Default: on for windows whose descriptors differ in their channel assignment (streams as encoders write them), off where
     * every descriptor is the same (synthetic batches of one shape).  These flags force it on / off. */
    CLX_COMPOSE         = 1u << 11,
    CLX_NO_COMPOSE      = 1u << 12,
    /* Narrow output straight from the decode (planned batches: clx_batch_create + clx_batch_run / clx_batch_submit; round 5).  The
     * batch's `d_out` buffers then hold channel-interleaved little-endian 16-bit PCM -- what lib.rs:473-520 (FlacSamples) walks and
     * examples/decode.rs:48-62 writes to a .wav -- instead of planar i32: `d_out` points to int16_t, indexed by the SAME sample
     * offsets (frame i's block starts at int16 index out_sample_offsets[i]; sample t of channel c at + t * n_channels + c), each
     * sample's low 16 bits as clx_batch_interleave(.., 2) gives them.  Every frame must have at most 16 bits per sample.  The
     * lean decode kernel writes a stereo frame's 32 samples as one 128-byte line from the tiles it stages anyway (half the bytes
     * through the write path), and a mono frame's as 64 bytes (round 6); whatever it leaves -- more channels, odd block sizes, waves that
     * give up -- the general kernels decode into staging rows of their workgroup's own and narrow row by row (one allocation per
     * internal stream, sized by what the descriptors say is left: round 5's planar scratch per run in flight is gone).  Failed frames'
     * bytes are unspecified, as the planar output's are.  Always the lane kernels, fused build. */
    CLX_OUT_PCM16       = 1u << 13,
    /* Pipelined submissions of the fused lane build, another launch form (round 6; off by default: measured slower, DESIGN.md
     * section 4.4): a merged launch's scan waves and 16-bit-tier decode waves as TICKETS taken off a counter by one grid of waves
     * that stay resident (clx_k_pool) instead of two kernels of one workgroup per wave (clx_k_scan, clx_k_lean).  Bit-exact like the
     * default.  (Batches whose waves are composed by content keep the two kernels anyway.) */
    CLX_POOL            = 1u << 14,
    /* The same as CLX_OUT_PCM16 with packed little-endian 24-bit samples (3 bytes each; round 6): `d_out` points to bytes, frame i's
     * block starts at byte 3 * out_sample_offsets[i], sample t of channel c at + 3 * (t * n_channels + c): what clx_batch_interleave(.., 3)
     * gives.  Every frame must have at most 24 bits per sample.  The split tier (clx_k_lean24, which takes the batch's 16-bit frames too
     * in this mode) writes a stereo frame's 32 sample pairs as twelve 16-byte pieces from the tiles it stages anyway (blocks that start on 16 bytes: out_sample_offsets[i] a multiple of 16); mono and
     * multi-channel frames, odd block sizes and waves that give up go through the general kernels' staging rows. */
    CLX_OUT_PCM24       = 1u << 15
};

/* One-shot convenience: plan + run + fetch results.  `out` is planar i32
 * (channel c of frame i at out[out_sample_offsets[i] + c*block_size ...),
 * frame.rs:409-410,477-481).  `results[i]` receives frame i's status. */
int clx_decode_frames(clx_ctx* ctx, const uint8_t* arena, size_t arena_len,
                      const clx_frame_desc* frames, size_t n,
                      int32_t* out, const uint64_t* out_sample_offsets,
                      clx_frame_result* results, uint32_t flags);

/* The same with several contexts -- one per GPU (or several on one GPU): the batch is cut into contiguous frame ranges of
 * near-equal algorithmic weight, context c decodes range c on a host thread of its own and receives only that range's slice
 * of the arena; nothing is exchanged between the devices (frames are independent: frame.rs:667-779 touches only its own bytes
 * and buffer; the reference's counterpart is one FrameReader per thread).  Host buffers only; frames in increasing,
 * non-overlapping output order (else the whole batch goes to ctxs[0]). */
int clx_decode_frames_multi(clx_ctx* const* ctxs, size_t n_ctx, const uint8_t* arena, size_t arena_len,
                            const clx_frame_desc* frames, size_t n, int32_t* out, const uint64_t* out_sample_offsets,
                            clx_frame_result* results, uint32_t flags);

/* Host-to-host decode as a pipeline (what a caller without device-resident data uses): the batch is cut into chunks of
 * frames; one chunk's compressed bytes travel to the device while the previous chunk is decoded and the one before that
 * returns its PCM.  Device buffers and plans live in the context and are reused from call to call.
 *   sample_bytes == 0: `out` is planar int32_t, as clx_decode_frames
 *   sample_bytes 1..4: `out` receives the narrow stage's channel-interleaved little-endian PCM (clx_batch_interleave), frame i
 *                      at byte out_sample_offsets[i] * sample_bytes -- what callers of the reference write out
 *                      (examples/decode.rs:48-62), half the bytes over the link for 16-bit audio
 *   out == NULL      : nothing is copied back but the results (decode throughput with the upload included)
 * Samples of failed frames read as zeros.  Frames in increasing, non-overlapping output order.  Buffers from clx_host_alloc
 * (pinned) make the copies asynchronous at link speed; any host memory works. */
int   clx_decode_frames_stream(clx_ctx* ctx, const uint8_t* arena, size_t arena_len, const clx_frame_desc* frames, size_t n,
                               void* out, uint32_t sample_bytes, const uint64_t* out_sample_offsets,
                               clx_frame_result* results, uint32_t flags);
/* Frames per chunk of clx_decode_frames_stream on this context; 0 (the default) = a third of the batch, 256 .. 8192: a chunk's
 * decode lasts at least as long as the predictor kernel's serial chain, so few large chunks beat many small ones. */
void  clx_set_stream_chunk(clx_ctx* ctx, size_t frames_per_chunk);
void* clx_host_alloc(size_t bytes);      /* pinned host memory (hipHostMalloc); NULL on failure */
void  clx_host_free(void* p);

/* One-shot interleave / narrow stage (see clx_batch_interleave).  `planar` follows CLX_OUT_ON_DEVICE, `pcm`
 * CLX_PCM_ON_DEVICE; `results` (may be NULL) marks frames to skip (status != CLX_OK). */
int clx_interleave(clx_ctx* ctx, const int32_t* planar, const clx_frame_desc* frames, size_t n,
                   const uint64_t* out_sample_offsets, const clx_frame_result* results,
                   void* pcm, uint32_t sample_bytes, uint32_t flags);

/* Config-2 entry: n independent byte-aligned SUBFRAMES (no frame header):
 * subframe i starts at arena[byte_offs[i]], decoded at bps[i] into
 * out[out_sample_offsets[i] .. +block_size[i]).  = `subframe::decode` (subframe.rs:184). */
int clx_decode_subframes(clx_ctx* ctx, const uint8_t* arena, size_t arena_len,
                         const uint64_t* byte_offs, const uint16_t* block_sizes,
                         const uint8_t* bps, size_t n,
                         int32_t* out, const uint64_t* out_sample_offsets,
                         clx_frame_result* results, uint32_t flags);

/* Planned batch: descriptors uploaded once, then run any number of times on
 * device-resident data (what bench.py times).  `stream` is a hipStream_t
 * (NULL = the context's own stream). */
int  clx_batch_create(clx_ctx* ctx, const clx_frame_desc* frames, size_t n,
                      const uint64_t* out_sample_offsets, uint32_t flags, clx_batch** out);
int  clx_batch_run(clx_batch* b, const uint8_t* d_arena, size_t arena_len,
                   int32_t* d_out, void* stream);
/* Pipelined submission: the same work and the same results as clx_batch_run, with several submissions in flight (the reference
 * has no counterpart: one FrameReader decodes one frame at a time, frame.rs:667).  Which kernels a batch's submissions use is
 * chosen for throughput (clx_batch_submit_lanes):
 *   - usually the lane kernels, fused build.  One run of those is a serial chain per subframe on a fraction of the machine, and
 *     the machine runs only a handful of kernels from different queues side by side -- so consecutive submissions are MERGED:
 *     they wait until a few of them are there (or until somebody flushes / asks for results) and go out as ONE grid whose second
 *     dimension is the submission, on two internal streams in turn (the scan stage of one launch overlaps the decode stage of
 *     another).  Every submission has its own scratch buffers and results.  No environment variable is involved: two internal
 *     streams fit HIP's default number of hardware queues.
 *   - the wave kernels, four in flight on internal streams of their own: only when CLX_PATH_WAVES asks for them or the arena is
 *     4 GiB or more (the merged lane launches are ahead at every batch size measured).
 * A submission starts no earlier than everything queued on `stream` when it (or a later one merged with it) was submitted.  Give
 * the submissions in flight different `d_out` buffers, i.e. rotate over clx_batch_submit_depth(b) of them (re-using a buffer is
 * legal: the submission then goes out after the earlier one that writes it).  Submissions may stay pending until
 * clx_batch_flush: work enqueued on `stream` after it sees every submission finished; clx_batch_results and
 * clx_batch_interleave flush by themselves; clx_batch_results returns the LAST submission's results.
 * One batch, one caller stream at a time: a batch's submissions and runs come in on ONE `stream` until a clx_batch_flush /
 * clx_batch_results on that stream (submissions that arrive on another stream are not merged with pending ones, but what is
 * already in flight is ordered against the stream it came in on only; distinct batches and contexts are independent).
 * Output buffers are told apart by their BASE address: two submissions with the same `d_out` are ordered (the later one goes out
 * behind the earlier one), two whose buffers overlap but start at different addresses are NOT -- hand over buffers that are either
 * identical or disjoint.
 * Pending submissions are not forgotten: re-planning or destroying a batch launches what is still pending first.  A merged launch
 * that cannot be made (a HIP error) drops ITS submissions: the call that triggered the launch returns CLX_API_ERROR, and so does,
 * once, the next clx_batch_flush / clx_batch_results / clx_batch_interleave of the batch (clx_last_error says which).  A batch
 * that still holds pending submissions should be flushed or destroyed BEFORE its context: clx_batch_destroy then drains the device
 * instead of ordering the launch behind the (possibly gone) stream the submissions came in on. */
#ifndef CLX_SUBMIT_DEPTH
#define CLX_SUBMIT_DEPTH 24     /* the most submissions any batch keeps in flight */
#endif
int  clx_batch_submit(clx_batch* b, const uint8_t* d_arena, size_t arena_len,
                      int32_t* d_out, void* stream);
/* How many submissions THIS batch keeps in flight, i.e. how many output buffers to rotate over: 4 for the wave kernels, 24 for
 * the lane kernels (two merged launches of twelve), 1 where a submission is a plain run. */
int  clx_batch_submit_depth(const clx_batch* b);
int  clx_batch_submit_lanes(const clx_batch* b);      /* 1: its pipelined submissions run the fused lane kernels */
int  clx_batch_submit_merge(const clx_batch* b);      /* how many consecutive submissions go out as one launch (1: none are merged) */
int  clx_batch_flush(clx_batch* b, void* stream);
/* Blocks until the last run finished, then copies the per-frame results to host. */
int  clx_batch_results(clx_batch* b, clx_frame_result* results);
/* Interleave / narrow output stage on the planned frames (what callers of the reference do next: FlacSamples,
 * lib.rs:473-520; Block::stereo_samples -> i16 WAV, examples/decode.rs:48-62).  Frame i's planar samples
 * d_planar[off_i + c*bs + s] become little-endian two's-complement PCM of `sample_bytes` (1..4) bytes at byte
 * (off_i + s*channels + c) * sample_bytes of d_pcm -- channel-interleaved, the order the STREAMINFO MD5 is defined
 * over (metadata.rs:52-53).  Frames whose last run failed are skipped.  Async on `stream`, after clx_batch_run. */
int  clx_batch_interleave(clx_batch* b, const int32_t* d_planar, void* d_pcm, uint32_t sample_bytes, void* stream);
/* Number of predictor slots (subframes incl. alignment padding) in the plan. */
uint64_t clx_batch_slots(const clx_batch* b);
/* Per-kernel HIP-event timing: kernels are numbered in launch order (clx_batch_kernel_name gives the name; NULL past the last
 * one).  enable = 1: every clx_batch_run / clx_batch_submit is a plain run with an event in front of each kernel (the LAST run's
 * durations are kept); enable = 2: pipelined submissions go out as usual and the events bracket the kernels of each MERGED launch
 * of the lane kernels (the LAST launch's durations are kept: submit, clx_batch_flush, synchronise, read); 0: off. */
int  clx_batch_set_profiling(clx_batch* b, int enable);
int  clx_batch_kernel_ms(clx_batch* b, int kernel, float* ms);
const char* clx_batch_kernel_name(const clx_batch* b, int kernel);
void clx_batch_destroy(clx_batch* b);

/* ------------------------------------------------------------------------
 * Stream API (host C++ classes claxon::FlacReader / FrameReader / Block live
 * in claxon_amd/csrc/host/claxon.hpp; these are their C handles so that any
 * FFI can reach them).  Replaces FlacReader::{new,open,streaminfo,blocks}
 * (lib.rs:217-458) and FrameReader::read_next_or_eof (frame.rs:667) for a
 * stream held in memory; frames are indexed on the host and decoded on the
 * device in batches.
 * ---------------------------------------------------------------------- */
typedef struct clx_streaminfo {          /* metadata.rs:24-54 */
    uint16_t min_block_size, max_block_size;
    uint32_t min_frame_size, max_frame_size;   /* 0 = unknown (None) */
    uint32_t sample_rate, channels, bits_per_sample;
    uint64_t samples;                          /* 0 = unknown (None) */
    uint8_t  md5sum[16];
} clx_streaminfo;

typedef struct clx_block_info {          /* frame.rs:402-411 (Block) */
    uint64_t time;
    uint32_t block_size;
    uint32_t channels;
} clx_block_info;

typedef struct clx_reader clx_reader;

/* Parse the `fLaC` marker + metadata blocks of an in-memory stream
 * (lib.rs:186-205, 230-307; metadata.rs:214-400).  *audio_offset = first frame byte. */
int clx_read_stream_header(const uint8_t* data, size_t len, clx_streaminfo* info,
                           size_t* audio_offset, uint32_t* msg);

/* FLAC tags (VORBIS_COMMENT block; FlacReader::vendor / tags / get_tag, lib.rs:321-360, metadata.rs:73-212). */
typedef struct clx_tags clx_tags;
enum {
    CLX_OPT_METADATA_ONLY        = 1u << 0,   /* FlacReaderOptions::metadata_only (lib.rs:131): stop once the wanted metadata is in */
    CLX_OPT_NO_VORBIS_COMMENT    = 1u << 1    /* FlacReaderOptions::read_vorbis_comment = false (lib.rs:141) */
};
/* FlacReader::new_ext (lib.rs:230-307) on an in-memory stream: clx_read_stream_header plus the tags.  *tags (may be
 * NULL on return: the stream has no Vorbis comment block, or it was not asked for) is owned by the caller. */
int clx_read_stream_header_ext(const uint8_t* data, size_t len, uint32_t options, clx_streaminfo* info,
                               size_t* audio_offset, clx_tags** tags, uint32_t* msg);
const char* clx_tags_vendor(const clx_tags* t, size_t* len);              /* the vendor string (UTF-8, not NUL-safe: use *len) */
size_t      clx_tags_count(const clx_tags* t);
/* i-th "NAME=value" pair in stream order; pointers stay valid until clx_tags_free */
int         clx_tags_get(const clx_tags* t, size_t i, const char** name, size_t* name_len, const char** value, size_t* value_len);
/* value of the `occurrence`-th tag whose name equals `name` ASCII-case-insensitively (metadata::GetTag, metadata.rs:197-211);
 * NULL when there is none */
const char* clx_tags_lookup(const clx_tags* t, const char* name, size_t occurrence, size_t* value_len);
void        clx_tags_free(clx_tags* t);
/* tags of an open reader (NULL if the stream has none); owned by the reader */
const clx_tags* clx_reader_tags(const clx_reader* r);

/* One metadata block, as `metadata::read_metadata_block` (metadata.rs:261-319) returns it: for streams embedded in a
 * container (examples/decode_ogg.rs:32-41, 85-103: Ogg packets hold metadata blocks with their header;
 * examples/decode_mp4.rs: the "FLAC specific box" holds type and raw data).  `kind` is the variant of the reference's
 * `MetadataBlock` enum (metadata.rs:104-131): seek tables, cue sheets and pictures are read as Padding (the reference's
 * TODOs at metadata.rs:287, 296, 301), unknown types as Reserved. */
enum { CLX_BLOCK_STREAMINFO = 0, CLX_BLOCK_PADDING = 1, CLX_BLOCK_APPLICATION = 2, CLX_BLOCK_VORBIS_COMMENT = 4, CLX_BLOCK_RESERVED = 126 };
typedef struct clx_metadata_block {
    uint32_t kind;                       /* CLX_BLOCK_* */
    uint32_t length;                     /* Padding { length } (metadata.rs:108-111); the block's length for every kind */
    clx_streaminfo streaminfo;           /* StreamInfo(..) */
    uint32_t application_id;             /* Application { id, data } (metadata.rs:113-118) */
    const uint8_t* application_data;     /* points into the caller's buffer */
    size_t application_len;
    clx_tags* tags;                      /* VorbisComment(..); owned by the caller: clx_tags_free */
} clx_metadata_block;
/* read_metadata_block (metadata.rs:261): `data[0..len)` stands right behind the block header, whose fields the caller
 * passes.  *consumed = bytes read on success.  Errors and messages as the reference (too short a buffer: CLX_IO_ERROR). */
int clx_read_metadata_block(const uint8_t* data, size_t len, uint8_t block_type, uint32_t length,
                            clx_metadata_block* out, size_t* consumed, uint32_t* msg);
/* read_metadata_block_with_header (metadata.rs:244): header (last-block flag + type, 24-bit length; metadata.rs:214-231)
 * and body.  *is_last receives the header's flag (MetadataBlockReader stops after it, metadata.rs:573-578). */
int clx_read_metadata_block_with_header(const uint8_t* data, size_t len, clx_metadata_block* out, int* is_last,
                                        size_t* consumed, uint32_t* msg);
/* Container packets -> frame descriptors.  What the reference's container examples do per packet -- FrameReader::new over
 * the packet's bytes, then read_next_or_eof (examples/decode_ogg.rs:105-114, decode_mp4.rs:143-152) -- for n packets at
 * once: packet i = arena[offs[i] .. offs[i] + lens[i]) holds one frame; its header is parsed (clx_parse_frame_header) into
 * descs[i] / headers[i] (either may be NULL) with max_bytes = lens[i].  results[i] (may be NULL) receives the header's
 * status and message; packets shorter than two bytes give CLX_END_OF_STREAM (frame.rs:140-143; the examples skip empty
 * packets).  Returns CLX_OK when every packet has a valid header, else the first failing packet's status. */
int clx_describe_packets(const uint8_t* arena, size_t arena_len, const uint64_t* offs, const uint32_t* lens, size_t n,
                         int check_crc, clx_frame_desc* descs, clx_frame_header* headers, clx_frame_result* results);

int  clx_reader_open(clx_ctx* ctx, const char* path, clx_reader** out, uint32_t* msg);       /* lib.rs:455 */
int  clx_reader_new(clx_ctx* ctx, const uint8_t* data, size_t len, clx_reader** out, uint32_t* msg); /* lib.rs:217 */
int  clx_reader_streaminfo(const clx_reader* r, clx_streaminfo* out);                        /* lib.rs:312 */
/* read_next_or_eof: decodes (in device batches, lazily) and hands out the next
 * block.  `buffer` (capacity `cap` i32) receives channels*block_size planar
 * samples.  CLX_END_OF_STREAM at the end; errors as the reference. */
int  clx_reader_next_block(clx_reader* r, int32_t* buffer, size_t cap,
                           clx_block_info* info, uint32_t* msg);                             /* frame.rs:667 */
void clx_reader_close(clx_reader* r);

/* Host-side frame indexer for a contiguous stream: locates frame starts by
 * sync code + CRC-8-valid header, confirmed by the previous frame's CRC-16
 * (frame.rs:131-316 grammar; the reference has no resync, frame.rs:601-602).
 * Writes up to `cap` descriptors/headers; returns the number found in *n_found.
 * Stops at the first position where the chain cannot be continued and reports
 * that byte offset in *stop_off. */
int clx_index_frames(const uint8_t* data, size_t len, size_t start_off,
                     clx_frame_desc* descs, clx_frame_header* headers, size_t cap,
                     size_t* n_found, size_t* stop_off);

/* The same indexer with the byte work on the GPU (sync-code scan + CRC-8 of every candidate header, CRC-16 of every
 * byte between candidates); identical outputs.  `data` is a host pointer, or with CLX_ARENA_ON_DEVICE a 16-byte
 * aligned device pointer whose allocation is padded like a decode arena (>= round16(len) + 32 bytes). */
int clx_index_frames_device(clx_ctx* ctx, const uint8_t* data, size_t len, size_t start_off,
                            clx_frame_desc* descs, clx_frame_header* headers, size_t cap,
                            size_t* n_found, size_t* stop_off, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif /* CLAXON_HIP_H */
