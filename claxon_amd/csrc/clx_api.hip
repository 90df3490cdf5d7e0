// clx_api.hip -- host side of the C ABI declared in include/claxon_hip.h.
//
// Owns what the reference's host code owns around the hot path: the byte-aligned frame header
// (frame.rs:131-316), the stream header / STREAMINFO (lib.rs:186-307, metadata.rs:214-400), the
// error convention (error.rs) and the Block / FrameReader / FlacReader surface -- and plans and
// launches the HIP kernels for everything between header and footer.  There is no CPU decode path
// in this library: without a usable HIP device every decode entry point fails with CLX_API_ERROR.
#include "clx_kernels.hip"
#include "clx_lanes.hip"
#include "clx_lean.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <memory>
#include <vector>
#include <unordered_map>
#include <thread>

#include "clx_plan.h"
#include "host/claxon.hpp"

// ------------------------------------------------------------------------------------------------
// messages (strings are the reference's, error.rs / call sites cited in claxon_hip.h)
// ------------------------------------------------------------------------------------------------
namespace {
struct MsgInfo { int status; const char* text; };
const MsgInfo kMsgs[CLX_MSG_COUNT] = {
    /* NONE */ { CLX_OK, "" },
    /* UNEXPECTED_EOF */ { CLX_IO_ERROR, "unexpected eof" },
    { CLX_FORMAT_ERROR, "invalid subframe header" },
    { CLX_FORMAT_ERROR, "invalid subframe header, encountered reserved value" },
    { CLX_FORMAT_ERROR, "wasted bits per sample must not exceed 31" },
    { CLX_FORMAT_ERROR, "subframe has no non-wasted bits" },
    { CLX_FORMAT_ERROR, "invalid residual, encountered reserved value" },
    { CLX_FORMAT_ERROR, "invalid partition order" },
    { CLX_FORMAT_ERROR, "invalid residual" },
    { CLX_FORMAT_ERROR, "invalid fixed subframe, order is larger than block size" },
    { CLX_FORMAT_ERROR, "invalid LPC subframe, lpc order is larger than block size" },
    { CLX_FORMAT_ERROR, "invalid subframe, qlp precision value invalid" },
    { CLX_UNSUPPORTED, "unencoded binary is not yet implemented" },
    { CLX_UNSUPPORTED, "a negative quantized linear predictor coefficient shift is not supported, please file a bug." },
    { CLX_FORMAT_ERROR, "frame CRC mismatch" },
    { CLX_FORMAT_ERROR, "frame header CRC mismatch" },
    { CLX_FORMAT_ERROR, "frame sync code missing" },
    { CLX_FORMAT_ERROR, "invalid frame header, encountered reserved value" },
    { CLX_FORMAT_ERROR, "invalid frame header" },
    { CLX_FORMAT_ERROR, "invalid frame header, frame number too large" },
    { CLX_FORMAT_ERROR, "invalid block size, exceeds 65535" },
    { CLX_FORMAT_ERROR, "invalid variable-length integer" },
    { CLX_UNSUPPORTED, "header without bits per sample info" },
    { CLX_FORMAT_ERROR, "invalid stream header" },
    { CLX_FORMAT_ERROR, "stream starts with ID3 header rather than FLAC header" },
    { CLX_FORMAT_ERROR, "streaminfo block missing" },
    { CLX_FORMAT_ERROR, "encountered second streaminfo block" },
    { CLX_FORMAT_ERROR, "invalid streaminfo metadata block length" },
    { CLX_FORMAT_ERROR, "invalid metadata block type" },
    { CLX_FORMAT_ERROR, "inconsistent bounds, min block size > max block size" },
    { CLX_FORMAT_ERROR, "invalid block size, must be at least 16" },
    { CLX_FORMAT_ERROR, "inconsistent bounds, min frame size > max frame size" },
    { CLX_FORMAT_ERROR, "invalid sample rate" },
    { CLX_FORMAT_ERROR, "application block length must be at least 4 bytes" },
    { CLX_UNSUPPORTED, "application blocks larger than 10 MiB are not supported" },
    { CLX_FORMAT_ERROR, "Vorbis comment block is too short" },
    { CLX_UNSUPPORTED, "Vorbis comment blocks larger than 10 MiB are not supported" },
    { CLX_FORMAT_ERROR, "vendor string too long" },
    { CLX_FORMAT_ERROR, "too many entries for Vorbis comment block" },
    { CLX_FORMAT_ERROR, "Vorbis comment too long for Vorbis comment block" },
    { CLX_FORMAT_ERROR, "Vorbis comment field name contains invalid byte" },
    { CLX_FORMAT_ERROR, "Vorbis comment does not contain '='" },
    { CLX_FORMAT_ERROR, "Vorbis comment block has excess data" },
    { CLX_FORMAT_ERROR, "Vorbis comment block contains wrong number of entries" },
    { CLX_FORMAT_ERROR, "Vorbis comment or vendor string is not valid UTF-8" },
    { CLX_FORMAT_ERROR, "encountered second Vorbis comment block" },
};

// CRC tables generated from the polynomials (crc.rs:61,69): x^8+x^2+x+1 and x^16+x^15+x^2+1.
struct CrcTables {
    uint8_t t8[256];
    uint16_t t16[256];
    CrcTables() {
        for (int i = 0; i < 256; ++i) {
            uint8_t a = (uint8_t)i;
            uint16_t b = (uint16_t)(i << 8);
            for (int k = 0; k < 8; ++k) {
                a = (uint8_t)((a & 0x80) ? (a << 1) ^ 0x07 : a << 1);
                b = (uint16_t)((b & 0x8000) ? (b << 1) ^ 0x8005 : b << 1);
            }
            t8[i] = a; t16[i] = b;
        }
    }
};
const CrcTables& crc_tables() { static const CrcTables t; return t; }
}  // namespace

extern "C" const char* clx_message(uint32_t msg) { return msg < CLX_MSG_COUNT ? kMsgs[msg].text : "unknown"; }
extern "C" int clx_message_status(uint32_t msg) { return msg < CLX_MSG_COUNT ? kMsgs[msg].status : CLX_API_ERROR; }
extern "C" uint32_t clx_version(void) { return (CLX_VERSION_MAJOR << 16) | (CLX_VERSION_MINOR << 8) | CLX_VERSION_PATCH; }

extern "C" uint8_t clx_crc8(const uint8_t* p, size_t n) {
    const CrcTables& t = crc_tables();
    uint8_t s = 0;
    for (size_t i = 0; i < n; ++i) s = t.t8[s ^ p[i]];
    return s;
}
extern "C" uint16_t clx_crc16(const uint8_t* p, size_t n) {
    const CrcTables& t = crc_tables();
    uint16_t s = 0;
    for (size_t i = 0; i < n; ++i) s = (uint16_t)((s << 8) ^ t.t16[(uint8_t)(s >> 8) ^ p[i]]);
    return s;
}

// ------------------------------------------------------------------------------------------------
// frame header (frame.rs:64-105, 131-316).  Byte-aligned, so a plain cursor does.
// ------------------------------------------------------------------------------------------------
namespace {
struct ByteCursor {
    const uint8_t* p; size_t n; size_t pos;
    bool u8(uint32_t* v) { if (pos >= n) return false; *v = p[pos++]; return true; }
    bool be16(uint32_t* v) { uint32_t a, b; if (!u8(&a) || !u8(&b)) return false; *v = (a << 8) | b; return true; }
};
inline int fail(uint32_t* msg, int status, uint32_t m) { if (msg) *msg = m; return status; }
}  // namespace

extern "C" int clx_parse_frame_header(const uint8_t* p, size_t avail, int check_crc,
                                      clx_frame_header* out, uint32_t* msg) {
    if (msg) *msg = CLX_MSG_NONE;
    if (!out || (!p && avail)) return fail(msg, CLX_API_ERROR, CLX_MSG_NONE);
    std::memset(out, 0, sizeof *out);
    ByteCursor c{ p, avail, 0 };
    uint32_t srb;
    if (!c.be16(&srb)) return CLX_END_OF_STREAM;                      // Ok(None), frame.rs:140-143
    if ((srb & 0xfffcu) != 0xfff8u) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_FRAME_SYNC_MISSING);
    if (srb & 2u) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_FRAME_HEADER_RESERVED);
    out->variable_blocking = (uint8_t)(srb & 1u);

    uint32_t bs_sr;
    if (!c.u8(&bs_sr)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    uint32_t block_size = 0; bool bs8 = false, bs16 = false;
    const uint32_t bn = bs_sr >> 4;
    if (bn == 0) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_FRAME_HEADER_RESERVED);
    else if (bn == 1) block_size = 192;
    else if (bn <= 5) block_size = 576u << (bn - 2);
    else if (bn == 6) bs8 = true;
    else if (bn == 7) bs16 = true;
    else block_size = 256u << (bn - 8);

    uint32_t sample_rate = 0; bool sr8 = false, sr16 = false, sr16x10 = false;
    static const uint32_t kRates[12] = { 0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000 };
    const uint32_t sn = bs_sr & 15u;
    if (sn < 12) sample_rate = kRates[sn];
    else if (sn == 12) sr8 = true;
    else if (sn == 13) sr16 = true;
    else if (sn == 14) sr16x10 = true;
    else return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_FRAME_HEADER_INVALID);

    uint32_t cbr;
    if (!c.u8(&cbr)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    const uint32_t ca = cbr >> 4;
    if (ca < 8) { out->channel_assignment = CLX_CH_INDEPENDENT; out->n_channels = (uint8_t)(ca + 1); }
    else if (ca == 8) { out->channel_assignment = CLX_CH_LEFT_SIDE; out->n_channels = 2; }
    else if (ca == 9) { out->channel_assignment = CLX_CH_RIGHT_SIDE; out->n_channels = 2; }
    else if (ca == 10) { out->channel_assignment = CLX_CH_MID_SIDE; out->n_channels = 2; }
    else return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_FRAME_HEADER_RESERVED);
    static const int kBps[8] = { 0, 8, 12, -1, 16, 20, 24, -1 };
    const int bps = kBps[(cbr & 0x0eu) >> 1];
    if (bps < 0) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_FRAME_HEADER_RESERVED);
    out->bps = (uint8_t)bps;
    if (cbr & 1u) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_FRAME_HEADER_RESERVED);

    // read_var_length_int, frame.rs:64-105
    uint32_t first;
    if (!c.u8(&first)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    uint32_t extra = 0, mask_data = 0x7f, mask_mark = 0x80;
    while (first & mask_mark) { ++extra; mask_data >>= 1; mask_mark >>= 1; }
    if (extra > 0) {
        if (extra == 1) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_INVALID_VARINT);
        --extra;
    }
    uint64_t number = (uint64_t)(first & mask_data) << (6 * extra);
    for (int i = (int)extra - 1; i >= 0; --i) {
        uint32_t b;
        if (!c.u8(&b)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
        if ((b & 0xc0u) != 0x80u) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_INVALID_VARINT);
        number |= (uint64_t)(b & 0x3fu) << (6 * i);
    }
    if (!out->variable_blocking && number > 0x7fffffffull)
        return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_FRAME_NUMBER_TOO_LARGE);

    if (bs8) { uint32_t b; if (!c.u8(&b)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); block_size = b + 1; }
    if (bs16) {
        uint32_t b; if (!c.be16(&b)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
        if (b == 0xffffu) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_BLOCK_SIZE_EXCEEDS_65535);
        block_size = b + 1;
    }
    if (sr8) { uint32_t b; if (!c.u8(&b)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); sample_rate = b; }
    if (sr16) { uint32_t b; if (!c.be16(&b)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); sample_rate = b; }
    if (sr16x10) { uint32_t b; if (!c.be16(&b)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); sample_rate = b * 10; }

    const uint8_t computed = clx_crc8(p, c.pos);
    uint32_t presumed;
    if (!c.u8(&presumed)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    if (check_crc && computed != presumed) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_FRAME_HEADER_CRC_MISMATCH);

    out->block_size = (uint16_t)block_size;
    out->sample_rate = sample_rate;
    out->header_bytes = (uint16_t)c.pos;
    out->frame_or_sample_lo = (uint32_t)number;
    out->time = out->variable_blocking ? number : (uint64_t)block_size * (uint64_t)(uint32_t)number;   // frame.rs:771-774
    return CLX_OK;
}

// ------------------------------------------------------------------------------------------------
// context / batch
// ------------------------------------------------------------------------------------------------
// one of the three chunks in flight of clx_decode_frames_stream: a stream, a re-plannable batch and the chunk's device buffers
struct clx_stream_slot {
    clx_batch* b = nullptr;
    hipStream_t st = nullptr;
    uint8_t* d_arena = nullptr; size_t arena_cap = 0;
    int32_t* d_out = nullptr;   size_t out_cap = 0;
    uint8_t* d_pcm = nullptr;   size_t pcm_cap = 0;
    clx_frame_result* h_res = nullptr; size_t res_cap = 0;      // pinned: results come back without stalling the stream
    size_t lo = 0, hi = 0;                                      // frames of the chunk whose results are pending (hi > lo)
};

#ifndef CLX_SUBMIT_MERGE
#define CLX_SUBMIT_MERGE 12            // runs per merged launch of the fused lane kernels (<= CLX_MAX_MERGE).  12 x 2 is the best shape with the
                                       // runtime's default of 4 hardware queues (profiles/r03_merge_sweep.txt); the library sets no environment
#endif
#ifndef CLX_SUBMIT_STREAMS
#define CLX_SUBMIT_STREAMS 2           // internal streams the merged launches rotate over (MERGE * STREAMS <= CLX_SUBMIT_DEPTH)
#endif
static_assert(CLX_SUBMIT_MERGE <= CLX_MAX_MERGE && CLX_SUBMIT_STREAMS * CLX_SUBMIT_MERGE <= CLX_SUBMIT_DEPTH && CLX_SUBMIT_STREAMS <= 6, "merge width");

struct clx_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string last_error;
    clx_stream_slot slots[3];
    size_t stream_chunk = 0;          // frames per chunk of clx_decode_frames_stream (0: the default rule)
};

// K2 build by batch size (groups of 64 predictor slots) unless CLX_K2_LATENCY / CLX_K2_THROUGHPUT force one
#define CLX_K2_LATENCY_GROUPS 512u

struct clx_batch {
    clx_ctx* ctx = nullptr;
    int device = 0;
    clx_path_choice choice = { false, true };          // for one run at a time (clx_batch_run)
    clx_path_choice choice_submit = { false, true };   // for pipelined submissions (clx_batch_submit)
    bool all_narrow_aligned = false;                   // every frame: bps <= 16, rows 16-byte aligned and a multiple of 4 samples long
    clx_dev_frame* h_up = nullptr; size_t up_cap = 0;      // pinned staging of the uploaded plan
    hipEvent_t ev_up = nullptr; bool up_in_flight = false; // recorded behind the staging's H2D copy: the staging is rewritten only after it
    size_t cap[17] = {};             // bytes allocated for d_frames, d_sfd, d_results, d_dump, d_slot_frame, d_multi, d_sf_start, d_errkey, d_endbits, d_taken, d_crc_part, d_crc_todo
    size_t n = 0;
    uint64_t n_slots = 0;
    uint32_t flags = 0;
    std::vector<clx_frame_desc> h_descs;
    std::vector<clx_dev_frame> h_frames;
    clx_dev_frame* d_frames = nullptr;
    clx_sf_desc* d_sfd = nullptr;
    clx_frame_result* d_results = nullptr;
    int32_t* d_dump = nullptr;       // 128 bytes per lane for out-of-row stores (the wave kernels and clx_k_lanes use 64 of them)
    // lane path
    bool lanes = false;              // clx_batch_run uses the lane kernels
    bool lanes_planned = false;      // their plan data (d_slot_frame, d_multi, scratch) exists (plan_lanes_data)
    bool lanes_submit = false;       // pipelined submissions take the lane kernels (unless a flag says otherwise)
    uint32_t* d_slot_frame = nullptr;
    uint32_t* d_multi = nullptr;
    size_t n_multi = 0;
    bool any_bps_le16 = false, any_bps_gt16 = false;     // which of clx_k_lean / clx_k_lean24 can find work at all
    uint64_t out_len = 0;            // samples the planar output spans (CLX_OUT_PCM16: the size of a flight's planar scratch)
    unsigned general_grid = 0;       // workgroups per run of the general lane kernels behind the tiers (clx_plan_general_grid)
    uint64_t general_sure = 0;       // groups that the descriptors say the tiers leave
    // narrow output (CLX_OUT_PCM16 / _PCM24): the general kernels decode a group into staging rows of their workgroup's own and narrow
    // them themselves: `stage_groups` workgroups per run at most, rows of `stage_stride` samples; one allocation per stream the
    // kernels are launched on (launches of one stream follow each other), made when the first such launch goes out
    uint32_t stage_stride = 0; unsigned stage_groups = 0;
    int32_t* d_stage[6 + 1] = {}; size_t stage_cap[6 + 1] = {};      // (kMaxStreams of them and one more: plain runs on the caller's stream)
    uint32_t* d_most_left = nullptr; // the longest list of groups the tiers left in any run so far (clx_k_left), and where the host
    uint32_t* h_most_left = nullptr; // finds a copy of it (pinned; read without waiting: it sizes later launches)
    uint32_t* h_most_left_dev = nullptr;     // the device's address of that copy (clx_k_left writes it)
    uint32_t* d_sf_start = nullptr;
    uint32_t* d_errkey = nullptr;
    uint64_t* d_endbits = nullptr;
    uint32_t* d_taken = nullptr;     // per group of 64 slots: the generation number of the run in which clx_k_lean decoded it
    clx_crc_part* d_crc_part = nullptr;      // per slot: the lean kernels' lanes' shares of their frames' CRC-16 (tagged with the run's generation number)
    uint32_t* d_crc_todo = nullptr;  // per frame: clx_k_finalize -> clx_k_crc16_runs
    // waves composed by content (clx_k_compose): the plan's windows, and flight 0's own slot maps and content classes
    uint32_t* d_first_slot = nullptr;        // per frame: the plan's slot of its first subframe (what clx_dev_frame::first_slot says)
    clx_window* d_windows = nullptr; size_t n_windows = 0;
    uint32_t* d_slot_frame_run = nullptr; uint32_t* d_first_slot_run = nullptr; uint32_t* d_fkey = nullptr;
    bool profiling = false, profile_merged = false;
    enum { kMaxKernels = 12 };
    hipEvent_t ev[kMaxKernels + 1] = {};
    const char* kname[kMaxKernels] = {};
    int n_kernels = 0;
    bool ev_valid = false;
    int ev_runs = 1;                  // runs in the launch the events belong to (1 for a plain run)
    hipStream_t last_stream = nullptr;
    size_t planned_arena_len = 0;
    // Pipelined submissions (clx_batch_submit): up to kDepth submissions in flight, each a whole run (Rice stage, predictor stage,
    // CRC) on a stream of its own with its own descriptors and results -- the Rice stages of two submissions share the machine
    // (neither ends in a partly filled round of waves) and their predictor stages, serial chains, run side by side.
    enum { kDepth = CLX_SUBMIT_DEPTH, kDepthWaves = 4, kDepthLanes = CLX_SUBMIT_DEPTH };
    struct Flight {
        hipStream_t stream = nullptr;
        clx_sf_desc* d_sfd = nullptr;            // flight 0 uses the batch's own buffers
        clx_frame_result* d_results = nullptr;
        uint32_t* d_sf_start = nullptr; uint32_t* d_errkey = nullptr; uint64_t* d_endbits = nullptr;   // lane kernels' scratch
        uint32_t* d_taken = nullptr; uint32_t gen = 0;     // groups clx_k_lean took (marked with the run's generation number, never cleared)
        clx_crc_part* d_crc_part = nullptr; uint32_t* d_crc_todo = nullptr;
        uint32_t* d_slot_frame = nullptr; uint32_t* d_first_slot = nullptr; uint32_t* d_fkey = nullptr;   // the run's slot maps (its own when waves are composed)
        hipEvent_t ev_in = nullptr, ev_done = nullptr;
        hipEvent_t ev_rice = nullptr, ev_side = nullptr;   // Rice stage done | the submission's kernels on side_stream done
        bool side_pending = false, side_recorded = false;
        bool pending = false;                    // submitted, nobody has been made to wait for it yet
        bool sfd_stale = true;                   // d_sfd holds something other than a previous run's descriptors
        bool scratch_stale = false;              // a launch that used the lane kernels' scratch failed half way: clx_k_finalize may not have left it cleared
        const int32_t* out = nullptr;            // where the pending submission writes
    } flight[kDepth];
    hipStream_t side_stream = nullptr;                 // CRC and left-over predictor kernels of the submissions in flight (launch_waves)
    // Lane path, fused build: consecutive submissions are MERGED into one launch (grid.y = the runs; kernels take a clx_runs table).
    // The machine runs only a handful of kernels from different queues side by side however many queues there are (measured:
    // about six), and one run of these kernels is a serial chain per subframe on a fraction of the machine -- so filling it takes one
    // grid that holds many runs, not many streams.  Submissions wait in `pend` until `merge` of them are there (kMerge, fewer for very
    // large batches: batch_plan; or until somebody asks for results / flushes); merged launches rotate over two internal streams,
    // so that the scan stage of one overlaps the decode stage of the other.  Flights are only the runs' scratch buffers here.
    enum { kMerge = CLX_SUBMIT_MERGE, kStreams = CLX_SUBMIT_STREAMS, kMaxStreams = 6 };
    bool merge_tuned = false;
    int merge = kMerge, n_streams = kStreams;          // (builds with -DCLX_TUNING let CLX_TUNE_MERGE / CLX_TUNE_STREAMS override them)
    struct Pending { const uint8_t* arena; size_t arena_len; int32_t* out; int flight; };
    std::vector<Pending> pend;
    bool launch_failed = false;                        // a merged launch was dropped; reported once more by the next flush / results
    hipStream_t pend_stream = nullptr;                 // the caller's stream the pending submissions came in on
    hipStream_t mstream[kMaxStreams] = {};
    clx_pool_state* d_pool[kMaxStreams] = {};          // clx_k_pool's ticket counter of each stream's launch (zeroed in front of every launch)
    unsigned pool_waves = 0;                           // waves of clx_k_pool the device holds at once (0: not asked yet)
    // An event behind each of a stream's last kEvRing launches: a later launch on ANOTHER stream that re-uses an output buffer or a
    // scratch set waits for exactly the launch that used it last -- not for whatever that stream has been given since (a region of
    // 20 steps goes out as 12 + 8: waiting for the other stream's LATEST launch would run the two one after the other).
    enum { kEvRing = 4 };
    hipEvent_t m_in[kMaxStreams] = {}, m_done[kMaxStreams][kEvRing] = {};
    uint64_t m_count[kMaxStreams] = {};                // launches made on each stream (launch c's event sits in slot (c - 1) % kEvRing)
    bool m_unwaited[kMaxStreams] = {};                 // nobody has been made to wait for the stream's latest launch yet
    struct LaunchRef { int stream; uint64_t count; };
    std::unordered_map<const int32_t*, LaunchRef> out_writer;   // output buffer -> the merged launch that wrote it last
    LaunchRef flight_launch[CLX_SUBMIT_DEPTH] = {};    // the merged launch that used a flight's scratch last (count 0: none)
    uint64_t n_merged = 0;
    uint64_t n_submitted = 0;
    int first_merge = 0;               // runs in the first merged launch behind a wait (0: `merge`); tune_prio: internal stream priorities (measurement builds)
    int tune_prio = 0;
    uint64_t launches_since_wait = 0;
    int last_slot = -1;              // flight of the most recent pipelined submission (-1: the last run was a plain clx_batch_run)
};

namespace {
bool hip_ok(clx_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    if (ctx) { ctx->last_error = std::string(what) + ": " + hipGetErrorString(e); }
    return false;
}
#define HIP_TRY(ctx, call) do { if (!hip_ok((ctx), (call), #call)) return CLX_API_ERROR; } while (0)
}  // namespace

extern "C" void clx_batch_destroy(clx_batch* b);

extern "C" int clx_create(int device, clx_ctx** out) {
    if (!out) return CLX_API_ERROR;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CLX_API_ERROR;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CLX_API_ERROR;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return CLX_API_ERROR;   // the code object is gfx950-only
    clx_ctx* c = new (std::nothrow) clx_ctx();
    if (!c) return CLX_API_ERROR;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c; return CLX_API_ERROR;
    }
    *out = c;
    return CLX_OK;
}

extern "C" void clx_destroy(clx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto& sl : ctx->slots) {
        if (sl.st) (void)hipStreamSynchronize(sl.st);
        if (sl.b) clx_batch_destroy(sl.b);
        if (sl.d_arena) (void)hipFree(sl.d_arena);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.d_pcm) (void)hipFree(sl.d_pcm);
        if (sl.h_res) (void)hipHostFree(sl.h_res);
        if (sl.st) (void)hipStreamDestroy(sl.st);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char* clx_last_error(const clx_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

namespace { int launch_pending(clx_batch* b, bool inputs_ready = false); }
extern "C" void clx_batch_destroy(clx_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);      // (by value: a batch destroyed after its context must not look into it)
    // (accepted submissions that still wait for company go out before the streams are drained below -- the buffers they write
    //  are the caller's)
    if (!b->pend.empty()) {
        clx_ctx scratch_ctx;                   // (the batch's own context may be gone: errors of this launch have nowhere to go)
        scratch_ctx.device = b->device;
        b->ctx = &scratch_ctx;
        // (and with it the stream the submissions came in on, when that was the context's own: the device is drained instead of
        //  an event recorded there -- whatever was queued in front of the submissions has then run)
        (void)hipDeviceSynchronize();
        (void)launch_pending(b, true);
        b->ctx = nullptr;
    }
    if (b->d_frames) (void)hipFree(b->d_frames);
    if (b->d_sfd) (void)hipFree(b->d_sfd);
    if (b->d_results) (void)hipFree(b->d_results);
    if (b->d_dump) (void)hipFree(b->d_dump);
    if (b->d_slot_frame) (void)hipFree(b->d_slot_frame);
    if (b->d_multi) (void)hipFree(b->d_multi);
    if (b->d_sf_start) (void)hipFree(b->d_sf_start);
    if (b->d_errkey) (void)hipFree(b->d_errkey);
    if (b->d_endbits) (void)hipFree(b->d_endbits);
    if (b->d_taken) (void)hipFree(b->d_taken);
    if (b->d_most_left) (void)hipFree(b->d_most_left);
    if (b->h_most_left) (void)hipHostFree(b->h_most_left);
    if (b->d_crc_part) (void)hipFree(b->d_crc_part);
    if (b->d_crc_todo) (void)hipFree(b->d_crc_todo);
    if (b->d_first_slot) (void)hipFree(b->d_first_slot);
    if (b->d_windows) (void)hipFree(b->d_windows);
    if (b->d_slot_frame_run) (void)hipFree(b->d_slot_frame_run);
    if (b->d_first_slot_run) (void)hipFree(b->d_first_slot_run);
    if (b->d_fkey) (void)hipFree(b->d_fkey);
    for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
    if (b->side_stream) { (void)hipStreamSynchronize(b->side_stream); (void)hipStreamDestroy(b->side_stream); }
    for (int k = 0; k < clx_batch::kMaxStreams; ++k) {
        if (b->mstream[k]) { (void)hipStreamSynchronize(b->mstream[k]); (void)hipStreamDestroy(b->mstream[k]); }
        if (b->d_pool[k]) (void)hipFree(b->d_pool[k]);
        if (b->d_stage[k]) (void)hipFree(b->d_stage[k]);
        if (b->m_in[k]) (void)hipEventDestroy(b->m_in[k]);
        for (auto& e : b->m_done[k]) if (e) (void)hipEventDestroy(e);
    }
    for (int i = 0; i < clx_batch::kDepth; ++i) {
        clx_batch::Flight& F = b->flight[i];
        if (F.stream) { (void)hipStreamSynchronize(F.stream); (void)hipStreamDestroy(F.stream); }
        if (F.ev_in) (void)hipEventDestroy(F.ev_in);
        if (F.ev_done) (void)hipEventDestroy(F.ev_done);
        if (F.ev_rice) (void)hipEventDestroy(F.ev_rice);
        if (F.ev_side) (void)hipEventDestroy(F.ev_side);
        if (i != 0 && F.d_sfd) (void)hipFree(F.d_sfd);
        if (i != 0 && F.d_results) (void)hipFree(F.d_results);
        if (i != 0 && F.d_sf_start) (void)hipFree(F.d_sf_start);
        if (i != 0 && F.d_errkey) (void)hipFree(F.d_errkey);
        if (i != 0 && F.d_endbits) (void)hipFree(F.d_endbits);
        if (i != 0 && F.d_taken) (void)hipFree(F.d_taken);
        if (i != 0 && F.d_crc_part) (void)hipFree(F.d_crc_part);
        if (i != 0 && F.d_crc_todo) (void)hipFree(F.d_crc_todo);
        if (i != 0 && F.d_fkey) { (void)hipFree(F.d_slot_frame); (void)hipFree(F.d_first_slot); (void)hipFree(F.d_fkey); }      // (its own maps)
    }
    if (b->d_stage[clx_batch::kMaxStreams]) (void)hipFree(b->d_stage[clx_batch::kMaxStreams]);
    if (b->ev_up) { (void)hipEventSynchronize(b->ev_up); (void)hipEventDestroy(b->ev_up); }
    if (b->h_up) (void)hipHostFree(b->h_up);
    delete b;
}

namespace {
// device buffer of at least `need` bytes: kept when large enough, else replaced (contents are never carried over)
template <typename T> bool grow(clx_ctx* ctx, T** p, size_t* cap, size_t need, const char* what) {
    if (*p && *cap >= need) return true;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const size_t want = need + need / 4;                                   // head room: a stream of similar chunks settles at once
    if (!hip_ok(ctx, hipMalloc((void**)p, want ? want : 16), what)) return false;
    *cap = want;
    return true;
}
// the marks of the groups of 64 slots (clx_k_lean's "taken in run gen"), and behind them the list clx_k_left makes of the groups the
// tiers left: a count and up to one entry per group
size_t taken_bytes(uint64_t n_slots) { return (size_t)(2u * ((n_slots + 63) / 64) + 2u) * sizeof(uint32_t); }
// The lane kernels' plan data for the batch as planned: slot -> frame map, the multi-channel frames, the scratch of flight 0.
int plan_lanes_data(clx_batch* b) {
    clx_ctx* ctx = b->ctx;
    if (b->lanes_planned) return CLX_OK;
    const size_t n = b->n, ns = b->n_slots ? (size_t)b->n_slots : 1, nf = n ? n : 1;
    std::vector<uint32_t> slot_frame(ns), multi(nf), first_slot(nf, 0u);
    b->n_multi = clx_plan_lanes(b->h_frames.data(), n, b->n_slots, slot_frame.data(), multi.data());
    for (size_t i = 0; i < n; ++i) first_slot[i] = b->h_frames[i].first_slot;
    // windows whose waves are composed by content (fused build with the lean tiers in front; CLX_COMPOSE / CLX_NO_COMPOSE force it)
    std::vector<clx_window> windows(nf);
    const int cmode = (b->flags & (CLX_NO_COMPOSE | CLX_LANES_GENERAL | CLX_LANES_SPLIT)) ? -1 : (b->flags & CLX_COMPOSE) ? 1 : 0;
    b->n_windows = clx_plan_windows(b->h_frames.data(), n, cmode, windows.data());
    b->any_bps_le16 = b->any_bps_gt16 = false;
    for (size_t i = 0; i < n; ++i) { if (b->h_frames[i].bps <= 16u) b->any_bps_le16 = true; else b->any_bps_gt16 = true; }
    b->general_grid = clx_plan_general_grid(b->h_frames.data(), slot_frame.data(), b->n_slots, b->flags, &b->general_sure);
    {   // narrow output: the general kernels' staging rows (clx_lanes_group) -- as long as the batch's largest block, and per run as many
        // groups' worth as the descriptors say are left for certain, twice, and sixteen: the workgroups loop over what is left
        uint32_t bs_max = 1;
        for (size_t i = 0; i < n; ++i) bs_max = std::max<uint32_t>(bs_max, b->h_frames[i].block_size);
        b->stage_stride = (bs_max + 3u) & ~3u;
        const uint64_t groups = (b->n_slots + 63) / 64;
        b->stage_groups = (unsigned)std::min<uint64_t>(std::max<uint64_t>(groups, 1), 2 * b->general_sure + 16);
    }
    if (!b->d_most_left && !hip_ok(ctx, hipMalloc((void**)&b->d_most_left, sizeof(uint32_t)), "hipMalloc most_left")) return CLX_API_ERROR;
    if (!b->h_most_left && !hip_ok(ctx, hipHostMalloc((void**)&b->h_most_left, sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc most_left")) return CLX_API_ERROR;
    *b->h_most_left = 0u;
    if (hipHostGetDevicePointer((void**)&b->h_most_left_dev, b->h_most_left, 0) != hipSuccess) { (void)hipGetLastError(); b->h_most_left_dev = nullptr; }      // (no hint then: full grids)
    if (!hip_ok(ctx, hipMemset(b->d_most_left, 0, sizeof(uint32_t)), "memset most_left")) return CLX_API_ERROR;
    if (!grow(ctx, &b->d_slot_frame, &b->cap[4], ns * sizeof(uint32_t), "hipMalloc slot_frame") ||
        !grow(ctx, &b->d_multi, &b->cap[5], nf * sizeof(uint32_t), "hipMalloc multi") ||
        !grow(ctx, &b->d_sf_start, &b->cap[6], ns * sizeof(uint32_t), "hipMalloc sf_start") ||
        !grow(ctx, &b->d_errkey, &b->cap[7], nf * sizeof(uint32_t), "hipMalloc errkey") ||
        !grow(ctx, &b->d_endbits, &b->cap[8], nf * sizeof(uint64_t), "hipMalloc endbits") ||
        !grow(ctx, &b->d_taken, &b->cap[9], taken_bytes(ns), "hipMalloc taken") ||
        !grow(ctx, &b->d_crc_part, &b->cap[10], ns * sizeof(clx_crc_part), "hipMalloc crc_part") ||
        !grow(ctx, &b->d_crc_todo, &b->cap[11], nf * sizeof(uint32_t), "hipMalloc crc_todo") ||
        !hip_ok(ctx, hipMemset(b->d_taken, 0, taken_bytes(ns)), "memset taken") ||
        !hip_ok(ctx, hipMemset(b->d_crc_part, 0, ns * sizeof(clx_crc_part)), "memset crc_part") ||
        !hip_ok(ctx, hipMemset(b->d_sf_start, 0xff, ns * sizeof(uint32_t)), "memset sf_start") ||
        !hip_ok(ctx, hipMemset(b->d_errkey, 0xff, nf * sizeof(uint32_t)), "memset errkey") ||
        !hip_ok(ctx, hipMemcpy(b->d_slot_frame, slot_frame.data(), ns * sizeof(uint32_t), hipMemcpyHostToDevice), "H2D slot_frame") ||
        !hip_ok(ctx, hipMemcpy(b->d_multi, multi.data(), nf * sizeof(uint32_t), hipMemcpyHostToDevice), "H2D multi")) return CLX_API_ERROR;
    if (!grow(ctx, &b->d_first_slot, &b->cap[12], nf * sizeof(uint32_t), "hipMalloc first_slot") ||
        !hip_ok(ctx, hipMemcpy(b->d_first_slot, first_slot.data(), nf * sizeof(uint32_t), hipMemcpyHostToDevice), "H2D first_slot")) return CLX_API_ERROR;
    if (b->n_windows) {
        if (!grow(ctx, &b->d_windows, &b->cap[13], b->n_windows * sizeof(clx_window), "hipMalloc windows") ||
            !grow(ctx, &b->d_slot_frame_run, &b->cap[14], ns * sizeof(uint32_t), "hipMalloc slot_frame (run)") ||
            !grow(ctx, &b->d_first_slot_run, &b->cap[15], nf * sizeof(uint32_t), "hipMalloc first_slot (run)") ||
            !grow(ctx, &b->d_fkey, &b->cap[16], nf * sizeof(uint32_t), "hipMalloc fkey") ||
            !hip_ok(ctx, hipMemcpy(b->d_windows, windows.data(), b->n_windows * sizeof(clx_window), hipMemcpyHostToDevice), "H2D windows") ||
            !hip_ok(ctx, hipMemcpy(b->d_slot_frame_run, slot_frame.data(), ns * sizeof(uint32_t), hipMemcpyHostToDevice), "H2D slot_frame (run)") ||
            !hip_ok(ctx, hipMemcpy(b->d_first_slot_run, first_slot.data(), nf * sizeof(uint32_t), hipMemcpyHostToDevice), "H2D first_slot (run)") ||
            !hip_ok(ctx, hipMemset(b->d_fkey, 0, nf * sizeof(uint32_t)), "memset fkey")) return CLX_API_ERROR;
    }
    // (the fills above go through the null stream, which the batch's non-blocking streams do not wait for)
    if (!hip_ok(ctx, hipStreamSynchronize(nullptr), "hipStreamSynchronize")) return CLX_API_ERROR;
    b->lanes_planned = true;
    return CLX_OK;
}
// (Re)plan `b` for a list of frames: host-side planning, kernel selection, device buffers (reused when they are large enough).
int batch_plan_(clx_batch* b, const clx_frame_desc* frames, size_t n, const uint64_t* out_sample_offsets, uint32_t flags);
int launch_pending(clx_batch* b, bool inputs_ready);
int batch_plan(clx_batch* b, const clx_frame_desc* frames, size_t n, const uint64_t* out_sample_offsets, uint32_t flags) {
    const int st = batch_plan_(b, frames, n, out_sample_offsets, flags);
    if (st != CLX_OK) { b->n = 0; b->n_slots = 0; b->n_multi = 0; b->planned_arena_len = (size_t)-1; }      // a failed plan leaves an empty batch, not a half-updated one
    if (!b->merge_tuned) {
        // runs per merged launch: twelve while that stays below ~24 000 waves of 64 subframes, fewer for larger batches (125 000 stereo
        // frames per run: 6 x 2 in flight 2.35 ms per run, 12 x 2 3.40, 1 x 2 2.64 -- profiles/r03_merge_sweep.txt)
        const uint64_t groups = (b->n_slots + 63) / 64;
        const uint64_t m = groups ? 24576ull / groups : (uint64_t)clx_batch::kMerge;
        b->merge = (int)std::min<uint64_t>(std::max<uint64_t>(m, 1), (uint64_t)clx_batch::kMerge);
    }
    return st;
}
int batch_plan_(clx_batch* b, const clx_frame_desc* frames, size_t n, const uint64_t* out_sample_offsets, uint32_t flags) {
    clx_ctx* ctx = b->ctx;
    if (!b->pend.empty()) {
        // submissions that were accepted (CLX_OK) but still wait for company: they go out under the plan they were made for, and
        // are waited for here -- their scratch is released below
        if (launch_pending(b) != CLX_OK) { b->launch_failed = false; return CLX_API_ERROR; }      // (reported right here)
        for (int k = 0; k < clx_batch::kMaxStreams; ++k)
            if (b->mstream[k] && !hip_ok(ctx, hipStreamSynchronize(b->mstream[k]), "hipStreamSynchronize")) return CLX_API_ERROR;
    }
    b->launch_failed = false;                  // (a new plan: nothing of it has been dropped)
    if (flags & (CLX_OUT_PCM16 | CLX_OUT_PCM24)) {
        // narrow output straight from the decode: <= 16-bit (24-bit) frames, the lane kernels' fused build with the tiers in front
        const uint32_t wide = (flags & CLX_OUT_PCM24) ? 24u : 16u;
        if ((flags & CLX_OUT_PCM16) && (flags & CLX_OUT_PCM24)) { ctx->last_error = "CLX_OUT_PCM16 and CLX_OUT_PCM24 exclude each other"; return CLX_API_ERROR; }
        if (flags & (CLX_PATH_WAVES | CLX_LANES_SPLIT | CLX_LANES_GENERAL)) { ctx->last_error = "CLX_OUT_PCM16 / CLX_OUT_PCM24 run the fused lane kernels with the lean tier: not with CLX_PATH_WAVES / CLX_LANES_SPLIT / CLX_LANES_GENERAL"; return CLX_API_ERROR; }
        for (size_t i = 0; i < n; ++i)
            if (frames[i].bps > wide) { ctx->last_error = std::string(wide == 16u ? "CLX_OUT_PCM16" : "CLX_OUT_PCM24") + ": frame " + std::to_string(i) + " has more than " + std::to_string(wide) + " bits per sample"; return CLX_API_ERROR; }
        flags |= CLX_PATH_LANES | CLX_LANES_FUSED;
    }
    b->n = n; b->flags = flags;
    b->out_len = 0;
    for (size_t i = 0; i < n; ++i) b->out_len = std::max<uint64_t>(b->out_len, out_sample_offsets[i] + (uint64_t)frames[i].n_channels * frames[i].block_size);
    b->h_descs.assign(frames, frames + n);
    b->h_frames.resize(n ? n : 1);
    uint64_t slot = 0;
    const long bad = clx_plan_frames(frames, n, out_sample_offsets, b->h_frames.data(), &slot);
    if (bad >= 0) { ctx->last_error = "invalid clx_frame_desc at index " + std::to_string(bad); return CLX_API_ERROR; }
    if (slot > 0xfffffff0ull) { ctx->last_error = "too many subframes in one batch"; return CLX_API_ERROR; }
    b->n_slots = slot;
    const size_t nf = n ? n : 1, ns = slot ? (size_t)slot : 1;
    if (!grow(ctx, &b->d_frames, &b->cap[0], nf * sizeof(clx_dev_frame), "hipMalloc frames") ||
        !grow(ctx, &b->d_sfd, &b->cap[1], ns * sizeof(clx_sf_desc), "hipMalloc sfdesc") ||
        !grow(ctx, &b->d_results, &b->cap[2], nf * sizeof(clx_frame_result), "hipMalloc results")) return CLX_API_ERROR;
    if (!b->ev[0]) for (auto& e : b->ev) if (!hip_ok(ctx, hipEventCreate(&e), "hipEventCreate")) return CLX_API_ERROR;
    // path: explicit flag, else by the batch's shape and content (clx_select_path, clx_plan.h)
    {
        uint64_t samples = 0, wide = 0, bytes = 0; bool all_mono = true, lengths_known = true, na = true;
        for (size_t i = 0; i < n; ++i) {
            const uint64_t sm = (uint64_t)frames[i].n_channels * frames[i].block_size;
            na = na && frames[i].bps <= 16 && (frames[i].block_size & 3u) == 0u && (out_sample_offsets[i] & 3ull) == 0ull;
            samples += sm;
            if (frames[i].bps > 16) wide += sm;
            if (frames[i].max_bytes >= (1u << 24)) lengths_known = false; else bytes += frames[i].max_bytes;
            all_mono = all_mono && frames[i].n_channels == 1;
        }
        b->all_narrow_aligned = na;
        b->choice = clx_select_path(slot, samples, lengths_known ? bytes : 0, 4 * wide >= samples && samples != 0, all_mono);
        b->choice_submit = clx_select_path(slot, samples, lengths_known ? bytes : 0, 4 * wide >= samples && samples != 0, all_mono, true);
    }
    b->lanes = (flags & CLX_PATH_LANES) ? true : (flags & CLX_PATH_WAVES) ? false : b->choice.lanes;
    {   // where stores that fall outside a row go (K2 and D2 keep their store instructions unconditional)
        const size_t lanes64 = ((ns + 127) / 128) * 128;
        if (!grow(ctx, &b->d_dump, &b->cap[3], lanes64 * 32 * sizeof(int32_t), "hipMalloc dump")) return CLX_API_ERROR;      // (128 bytes per lane: clx_k_lean's)
    }
    // (the lane kernels' plan data: now when a run uses them, else with the first pipelined submission -- plan_lanes_data: one-shot
    //  decodes of a few frames, which run the wave kernels, do not pay for it)
    b->lanes_submit = (flags & CLX_PATH_WAVES) == 0 && b->choice_submit.lanes;
    b->lanes_planned = false;
    if (b->lanes && plan_lanes_data(b) != CLX_OK) return CLX_API_ERROR;
    // (a re-planned batch starts over: nothing of the previous plan may be in flight -- the caller's contract)
    for (int i = 0; i < clx_batch::kDepth; ++i) {
        clx_batch::Flight& F = b->flight[i];
        if (i != 0 && F.d_sfd) (void)hipFree(F.d_sfd);
        if (i != 0 && F.d_results) (void)hipFree(F.d_results);
        if (i != 0 && F.d_sf_start) (void)hipFree(F.d_sf_start);
        if (i != 0 && F.d_errkey) (void)hipFree(F.d_errkey);
        if (i != 0 && F.d_endbits) (void)hipFree(F.d_endbits);
        if (i != 0 && F.d_taken) (void)hipFree(F.d_taken);
        if (i != 0 && F.d_crc_part) (void)hipFree(F.d_crc_part);
        if (i != 0 && F.d_crc_todo) (void)hipFree(F.d_crc_todo);
        if (i != 0 && F.d_fkey) { (void)hipFree(F.d_slot_frame); (void)hipFree(F.d_first_slot); (void)hipFree(F.d_fkey); }
        F.d_taken = nullptr; F.gen = 0; F.d_crc_part = nullptr; F.d_crc_todo = nullptr; F.d_slot_frame = nullptr; F.d_first_slot = nullptr; F.d_fkey = nullptr;
        if (i == 0) F.d_results = nullptr;
        F.d_sfd = nullptr; F.d_results = nullptr; F.d_sf_start = nullptr; F.d_errkey = nullptr; F.d_endbits = nullptr; F.pending = false; F.side_pending = false; F.sfd_stale = true; F.scratch_stale = false; F.out = nullptr;   // (side_recorded stays: the event is still there)
    }
    b->last_slot = -1;
    b->planned_arena_len = (size_t)-1;
    b->ev_valid = false;
    b->pend.clear();
    for (int k = 0; k < clx_batch::kMaxStreams; ++k) b->m_unwaited[k] = false;
    b->out_writer.clear();
    for (auto& f : b->flight_launch) f = clx_batch::LaunchRef{ 0, 0 };
    return CLX_OK;
}
}  // namespace

extern "C" int clx_batch_create(clx_ctx* ctx, const clx_frame_desc* frames, size_t n,
                                const uint64_t* out_sample_offsets, uint32_t flags, clx_batch** out) {
    if (!ctx || !out || (n && (!frames || !out_sample_offsets))) return CLX_API_ERROR;
    *out = nullptr;
    if (n > 0xfffffff0ull) { ctx->last_error = "too many frames in one batch"; return CLX_API_ERROR; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    clx_batch* b = new (std::nothrow) clx_batch();
    if (!b) return CLX_API_ERROR;
    b->ctx = ctx; b->device = ctx->device;
#ifdef CLX_TUNING
    {   // tuning knobs of the measurement builds (-DCLX_TUNING: tools/gpu_ab_sat.sh, tools/merge_sweep.sh); the product build reads
        // no environment variable: runs per merged launch, internal streams
        const char* em = std::getenv("CLX_TUNE_MERGE"); const char* es = std::getenv("CLX_TUNE_STREAMS");
        const int m = em ? std::atoi(em) : 0, st = es ? std::atoi(es) : 0;
        if (m >= 1 && m <= CLX_MAX_MERGE) { b->merge = m; b->merge_tuned = true; }
        if (st >= 1 && st <= clx_batch::kMaxStreams) b->n_streams = st;
        if (b->merge * b->n_streams > clx_batch::kDepthLanes) b->n_streams = std::max(1, clx_batch::kDepthLanes / b->merge);
        const char* ef = std::getenv("CLX_TUNE_FIRST"); const char* ep = std::getenv("CLX_TUNE_PRIO");
        if (ef && std::atoi(ef) >= 1 && std::atoi(ef) <= CLX_MAX_MERGE) b->first_merge = std::atoi(ef);
        if (ep) b->tune_prio = std::atoi(ep);
    }
#endif
    if (batch_plan(b, frames, n, out_sample_offsets, flags) != CLX_OK) { clx_batch_destroy(b); return CLX_API_ERROR; }
    *out = b;
    return CLX_OK;
}

extern "C" uint64_t clx_batch_slots(const clx_batch* b) { return b ? b->n_slots : 0; }

extern "C" int clx_batch_set_profiling(clx_batch* b, int enable) {
    if (!b) return CLX_API_ERROR;
    b->profiling = enable == 1;            // 1: every submission is a plain run with an event in front of each kernel
    b->profile_merged = enable == 2;       // 2: pipelined submissions as usual, events around the kernels of each MERGED launch
    b->ev_valid = false;
    return CLX_OK;
}

namespace {
// K3: four frames per workgroup at a time; beyond 8 workgroups per CU the waves loop
unsigned crc_grid(size_t n) { return (unsigned)std::min<size_t>((n + 3) / 4, 2048); }
// the same behind the lane kernels, per run of a merged launch: 256 workgroups (1 024 waves) per run fill the machine when every frame
// is theirs, and are few enough to find room at once when none is (the lean kernels' lanes gather the CRC of what they decode)
unsigned crc_grid_runs(size_t n) { return (unsigned)std::min<size_t>((n + 3) / 4, 256); }
// clamp every frame's readable span against the arena and upload the plan (once per arena size)
int upload_plan(clx_batch* b, size_t arena_len, hipStream_t stream) {
    clx_ctx* ctx = b->ctx;
    if (b->planned_arena_len == arena_len) return CLX_OK;
    // staged in pinned memory that lives with the batch: the copy is asynchronous and nothing has to be waited for here (the
    // next re-plan of this batch comes after the caller has waited for its stream: clx_batch_run's contract)
    if (b->up_cap < b->n) {
        if (b->up_in_flight) { HIP_TRY(ctx, hipEventSynchronize(b->ev_up)); b->up_in_flight = false; }
        if (b->h_up) (void)hipHostFree(b->h_up);
        b->h_up = nullptr; b->up_cap = 0;
        HIP_TRY(ctx, hipHostMalloc((void**)&b->h_up, (b->n + b->n / 4 + 1) * sizeof(clx_dev_frame), hipHostMallocDefault));
        b->up_cap = b->n + b->n / 4 + 1;
    }
    // (a copy out of the staging may still be in flight -- on whichever stream the previous upload went to)
    if (b->up_in_flight) { HIP_TRY(ctx, hipEventSynchronize(b->ev_up)); b->up_in_flight = false; }
    std::memcpy(b->h_up, b->h_frames.data(), b->n * sizeof(clx_dev_frame));
    clx_plan_limits(b->h_descs.data(), b->n, arena_len, b->h_up);
    HIP_TRY(ctx, hipMemcpyAsync(b->d_frames, b->h_up, b->n * sizeof(clx_dev_frame), hipMemcpyHostToDevice, stream));
    if (!b->ev_up) HIP_TRY(ctx, hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(b->ev_up, stream));
    b->up_in_flight = true;
    b->planned_arena_len = arena_len;
    return CLX_OK;
}
// the lane kernels address the arena with 32-bit offsets: an explicit CLX_PATH_LANES fails beyond 4 GiB, a batch that only
// defaulted to them runs the wave kernels instead (their buffers exist for every batch).  -1: error (message set)
int use_lanes(clx_batch* b, size_t arena_len) {
    if (!b->lanes) return 0;
    if ((uint64_t)arena_len + 32ull >= (1ull << 32)) {
        if (b->flags & CLX_PATH_LANES) { b->ctx->last_error = "CLX_PATH_LANES needs arena_len < 4 GiB"; return -1; }
        return 0;
    }
    return 1;
}
// How many waves clx_k_pool is launched with: as many as the device can hold at once, or as there are tickets.  The kernel's
// registers allow three waves per SIMD (twelve per CU), its LDS ten per CU; waves beyond what fits wait for one that leaves and find
// the counter run out.  (The runtime's occupancy query prices LDS for a 64 KiB CU -- four waves -- and is not asked.)
unsigned pool_waves(clx_batch* b) {
    if (b->pool_waves) return b->pool_waves;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device) != hipSuccess || cus < 1) cus = 256;
    (void)hipGetLastError();
    b->pool_waves = 12u * (unsigned)cus;
    return b->pool_waves;
}
// The lane kernels of `n_runs` runs of the batch in one launch each (grid.y = the run): scan (where the later channels of
// multi-channel frames start), the decode (the lean 16-bit tier, then the general kernels on the groups it left -- or the two-wave
// build), the per-frame results, the CRC.  The runs' scratch (sf_start, errkey) is expected cleared to 0xff: the host does that when
// it allocates it, clx_k_finalize leaves it so behind every run.
// pool (merged launches; null: never) with CLX_POOL: the scan and the 16-bit tier as tickets of ONE kernel of resident waves
// (clx_k_pool) when the batch's waves are not composed by content -- `pool` is the launch's stream's ticket counter, zeroed here in
// front of the kernel.
template <typename Mark>
bool launch_lanes(clx_batch* b, const clx_runs& runs, unsigned n_runs, bool split, hipStream_t stream, Mark&& mark, clx_pool_state* pool = nullptr,
                  int stage_slot = clx_batch::kMaxStreams) {
    const unsigned groups = (unsigned)((b->n_slots + 63) / 64);
    const bool composed = runs.r[0].fkey != nullptr && b->n_windows && b->n_multi;
    const bool pooled = pool != nullptr && (b->flags & CLX_POOL) && !(b->flags & CLX_OUT_PCM24) && !split && !composed && runs.r[0].taken != nullptr && b->any_bps_le16 && !(b->flags & CLX_LANES_GENERAL);
#ifdef CLX_POOL_SCAN_APART      // (measurement builds: the scan as a kernel of its own in front of a pool of decode tickets only)
    if (pooled && b->n_multi) {
        if (!mark("clx_k_scan")) return false;
        hipLaunchKernelGGL(clx_k_scan, dim3((unsigned)((b->n_multi + 63) / 64), n_runs), dim3(64), 0, stream, runs,
                           (const clx_dev_frame*)b->d_frames, (const uint32_t*)b->d_multi, (uint32_t)b->n_multi);
    }
    const uint32_t pool_multi = 0u;
#else
    const uint32_t pool_multi = (uint32_t)b->n_multi;
#endif
    if (pooled) {
        if (!mark("clx_k_pool")) return false;
        if (hipMemsetAsync(pool, 0, sizeof(clx_pool_state), stream) != hipSuccess) return false;
        clx_pool_args A;
        A.runs = runs; A.frames = b->d_frames; A.multi = b->d_multi; A.ps = pool; A.order = nullptr; A.dump_all = b->d_dump;
        A.n_runs = n_runs; A.n_slots = (uint32_t)b->n_slots; A.n_multi = pool_multi; A.pad = 0u;
        const uint64_t tickets = (uint64_t)n_runs * ((uint64_t)groups + (pool_multi + 63u) / 64u);
        hipLaunchKernelGGL(clx_k_pool, dim3((unsigned)std::min<uint64_t>(tickets, pool_waves(b))), dim3(64), 0, stream, A);
    }
    else if (b->n_multi) {
        if (!mark("clx_k_scan")) return false;
        if (b->flags & CLX_LANES_GENERAL)         // (the round-2 build of the scan, with the round-2 decode kernels: the comparison target)
            hipLaunchKernelGGL(clx_k_scan_general, dim3((unsigned)((b->n_multi + 63) / 64), n_runs), dim3(64), 0, stream, runs,
                               (const clx_dev_frame*)b->d_frames, (const uint32_t*)b->d_multi, (uint32_t)b->n_multi);
        else
            hipLaunchKernelGGL(clx_k_scan, dim3((unsigned)((b->n_multi + 63) / 64), n_runs), dim3(64), 0, stream, runs,
                               (const clx_dev_frame*)b->d_frames, (const uint32_t*)b->d_multi, (uint32_t)b->n_multi);
    }
    if (!split) {
        // waves composed by content: the windows' frames dealt to the lanes by class (the scan has left every frame's class)
        if (composed) {
            if (!mark("clx_k_compose")) return false;
            hipLaunchKernelGGL(clx_k_compose, dim3((unsigned)b->n_windows, n_runs), dim3(CLX_COMPOSE_THREADS), 0, stream, runs, (const clx_window*)b->d_windows);
        }
        // the 16-bit tier first: it marks the groups it decodes with the run's generation number, the general kernels skip them
        const bool p24 = (b->flags & CLX_OUT_PCM24) != 0;     // (packed 24-bit output is the split tier's, for the batch's 16-bit frames too)
        if (runs.r[0].taken != nullptr && b->any_bps_le16 && !pooled && !p24) {
            if (!mark("clx_k_lean")) return false;
            // CLX_LEAN_LDS_PAD (measurement builds only): extra dynamic LDS per wave, i.e. fewer decode waves per CU -- the knob behind
            // profiles/r05_occupancy_sweep.txt
#ifdef CLX_TUNING
            static const unsigned lean_pad = [] { const char* e = std::getenv("CLX_LEAN_LDS_PAD"); return e ? (unsigned)std::strtoul(e, nullptr, 10) : 0u; }();
#else
            const unsigned lean_pad = 0u;
#endif
            hipLaunchKernelGGL(clx_k_lean, dim3(groups, n_runs), dim3(64), lean_pad, stream, runs,
                               (const clx_dev_frame*)b->d_frames, (uint32_t)b->n_slots, b->d_dump);
        }
        // the split tier for audio of more than 16 bits (launched when the batch holds such frames: it also takes <= 16-bit groups of
        // more than 12 taps that share the batch, which otherwise stay with clx_k_lanes_hi)
        if (runs.r[0].taken != nullptr && (b->any_bps_gt16 || p24)) {
            if (!mark("clx_k_lean24")) return false;
            hipLaunchKernelGGL(clx_k_lean24, dim3(groups, n_runs), dim3(64), 0, stream, runs,
                               (const clx_dev_frame*)b->d_frames, (uint32_t)b->n_slots, b->d_dump);
        }
        // the general kernels: every group when no tier ran in front; else the groups the tiers left, listed by clx_k_left, through a
        // grid sized for what the descriptors say will be left (clx_plan_general_grid: usually a fraction of the groups, and nothing to do)
        // (one run by itself keeps the full grid: the workgroups that find nothing wait for nobody there, and a batch that is run
        //  once has no earlier run to learn from; merged launches size the grid by the plan and by the longest list seen so far)
        unsigned ggrid = groups;
        const bool listed = runs.r[0].taken != nullptr;
        if (listed) {
            if (!mark("clx_k_left")) return false;
            hipLaunchKernelGGL(clx_k_left, dim3((groups + 255) / 256, n_runs), dim3(256), 0, stream, runs, (uint32_t)groups, b->d_most_left, b->h_most_left_dev);
            if (n_runs > 1 && b->general_grid) {
                const unsigned seen = b->h_most_left ? *(volatile uint32_t*)b->h_most_left : 0u;
                ggrid = std::min(groups, std::max(b->general_grid, 2u * seen + 16u));
            }
        }
        // narrow output: every workgroup of the general kernels decodes into 64 staging rows of its own and narrows them itself
        // (clx_lanes_group) -- the stream's staging, grown here when this launch needs more of it than any before
        clx_runs gruns = runs;
        if (b->flags & (CLX_OUT_PCM16 | CLX_OUT_PCM24)) {
            ggrid = std::min(ggrid, std::max(b->stage_groups, 1u));
            const size_t per_run = (size_t)ggrid * 64u * b->stage_stride, need = per_run * n_runs * sizeof(int32_t);
            if (b->stage_cap[stage_slot] < need) {
                if (b->d_stage[stage_slot]) { (void)hipFree(b->d_stage[stage_slot]); b->d_stage[stage_slot] = nullptr; b->stage_cap[stage_slot] = 0; }      // (waits for what uses it)
                if (!hip_ok(b->ctx, hipMalloc((void**)&b->d_stage[stage_slot], need), "hipMalloc staging rows (narrow output)")) return false;
                b->stage_cap[stage_slot] = need;
            }
            for (unsigned r = 0; r < n_runs; ++r) {
                gruns.r[r].planar = b->d_stage[stage_slot] + per_run * r;
                gruns.r[r].flags |= CLX_RUN_STAGE_BITS(b->stage_stride);
            }
        }
        if (!mark("clx_k_lanes")) return false;             // (+ clx_k_lanes_hi, its order > 12 twin)
        hipLaunchKernelGGL(clx_k_lanes, dim3(ggrid, n_runs), dim3(64), 0, stream, gruns,
                           (const clx_dev_frame*)b->d_frames, (uint32_t)b->n_slots, b->d_dump);
        hipLaunchKernelGGL(clx_k_lanes_hi, dim3(ggrid, n_runs), dim3(64), 0, stream, gruns,
                           (const clx_dev_frame*)b->d_frames, (uint32_t)b->n_slots, b->d_dump);
    }
    else {
        if (n_runs != 1) return false;
        const clx_run& R = runs.r[0];
        if (!mark("clx_k_lanes2")) return false;
        hipLaunchKernelGGL(clx_k_lanes2, dim3((unsigned)((b->n_slots + 127) / 128)), dim3(256), 0, stream, R.arena, R.alloc_len,
                           (const clx_dev_frame*)b->d_frames, (const uint32_t*)b->d_slot_frame, (uint32_t)b->n_slots,
                           (const uint32_t*)R.sf_start, R.out, R.errkey, R.end_bits, b->d_dump);
    }
    if (!mark("clx_k_finalize")) return false;
    hipLaunchKernelGGL(clx_k_finalize, dim3((unsigned)((b->n + 255) / 256), n_runs), dim3(256), 0, stream, runs,
                       (const clx_dev_frame*)b->d_frames, (uint32_t)b->n, (uint32_t)b->n_slots);
    if (b->flags & CLX_VERIFY_CRC16) {
        if (!mark("clx_k_crc16")) return false;
        hipLaunchKernelGGL(clx_k_crc16_runs, dim3(n_runs > 1 ? crc_grid_runs(b->n) : crc_grid(b->n), n_runs), dim3(256), 0, stream, runs,
                           (const clx_dev_frame*)b->d_frames, (uint32_t)b->n);
    }
    return true;
}
// the run that decodes (arena, arena_len) into `out` with a flight's scratch
clx_run make_run(const clx_batch* b, const clx_batch::Flight& F, const uint8_t* d_arena, size_t arena_len, int32_t* d_out, bool lean) {
    clx_run R;
    R.arena = d_arena; R.alloc_len = (((uint64_t)arena_len + 15ull) & ~15ull) + 16ull;      // claxon_hip.h: the allocation covers this
    R.out = d_out; R.sf_start = F.d_sf_start; R.errkey = F.d_errkey; R.end_bits = F.d_endbits;
    R.taken = (lean && !(b->flags & CLX_LANES_GENERAL)) ? F.d_taken : nullptr;
    R.results = F.d_results; R.gen = F.gen;
    R.crc_part = F.d_crc_part; R.crc_todo = F.d_crc_todo;
    R.planar = nullptr;                        // (narrow output: launch_lanes hands the general kernels their staging rows)
    // the run's slot maps: its own when its waves are composed by content (clx_k_compose rewrites the windows' parts), else the plan's
    const bool composed = lean && b->n_windows != 0 && F.d_fkey != nullptr;
    R.slot_frame = composed ? F.d_slot_frame : b->d_slot_frame;
    R.first_slot = composed ? F.d_first_slot : b->d_first_slot;
    R.fkey = composed ? F.d_fkey : nullptr;
    R.flags = ((b->flags & CLX_VERIFY_CRC16) ? CLX_RUN_CRC : 0u) | ((b->flags & CLX_OUT_PCM16) ? CLX_RUN_PCM16 : 0u) | ((b->flags & CLX_OUT_PCM24) ? CLX_RUN_PCM24 : 0u);
    return R;
}
void launch_stage1_waves(clx_batch* b, const uint8_t* d_arena, uint64_t alloc_len, int32_t* d_out, clx_sf_desc* d_sfd, clx_frame_result* d_results,
                         hipStream_t stream) {
    // CLX_K1_LDS_PAD: extra dynamic LDS per workgroup, i.e. fewer K1 waves per CU -- the measurement knob behind DESIGN.md's
    // "lower occupancy is strictly worse" (32 -> 16 waves per CU: 0.28 -> 0.39 ms); not used otherwise
#ifdef CLX_TUNING
    static const unsigned k1_pad = [] { const char* e = std::getenv("CLX_K1_LDS_PAD"); return e ? (unsigned)std::strtoul(e, nullptr, 10) : 0u; }();
#else
    const unsigned k1_pad = 0u;
#endif
    hipLaunchKernelGGL(clx_k_residual, dim3((unsigned)b->n), dim3(64), k1_pad, stream,
                       d_arena, alloc_len, (const clx_dev_frame*)b->d_frames, (uint32_t)b->n, d_out, d_sfd, d_results);
}
// K2 (+ K3) behind it: the two-wave (latency) build while the groups of 64 rows are few, the one-wave (throughput) build beyond
bool k2_latency_build(const clx_batch* b) {
    const unsigned groups = (unsigned)((b->n_slots + 63) / 64);
    return (b->flags & CLX_K2_LATENCY) ? true : (b->flags & CLX_K2_THROUGHPUT) ? false : groups <= CLX_K2_LATENCY_GROUPS;
}
// `side` (pipelined submissions): a second stream for what needs the Rice stage's output only and is small -- the CRC kernel
// (the end bits) and the predictor kernels for the groups clx_k_predict16 leaves (usually none: the kernels find nothing to do,
// and in the submission's own stream each of them would still hold up what is queued behind it).  They start behind `ev_rice`
// and run beside the predictor stage; `ev_side` says when they are done.
template <typename Mark>
bool launch_waves(clx_batch* b, const uint8_t* d_arena, uint64_t alloc_len, int32_t* d_out, clx_sf_desc* d_sfd, clx_frame_result* d_results,
                  hipStream_t stream, Mark&& mark, bool k2_latency, hipStream_t side = nullptr, hipEvent_t ev_rice = nullptr,
                  hipEvent_t ev_side = nullptr) {
    if (!mark("clx_k_residual")) return false;
    launch_stage1_waves(b, d_arena, alloc_len, d_out, d_sfd, d_results, stream);
    hipStream_t rest = stream;                     // where the CRC kernel and the left-over predictor kernels go
    if (side != nullptr) {
        if (hipEventRecord(ev_rice, stream) != hipSuccess || hipStreamWaitEvent(side, ev_rice, 0) != hipSuccess) return false;
        rest = side;
    }
    const unsigned groups = (unsigned)((b->n_slots + 63) / 64);
    if (k2_latency) {
        // The groups of 64 rows that are all aligned, of 16-bit audio and of at most 8 taps go to clx_k_predict16: workgroups of
        // four waves with half the registers and a quarter of the LDS of the general kernel's, so that the predictor stages of
        // several submissions in flight are resident side by side.  The rest go to the general kernel -- unless the plan says
        // that only a predictor order above 8 can put a group there (every frame 16-bit and aligned): then to the one-wave
        // kernels, whose workgroups find room on a busy machine at once.
        if (!mark("clx_k_predict16")) return false;
        hipLaunchKernelGGL(clx_k_predict16, dim3(groups), dim3(256), 0, stream, d_out,
                           (const clx_sf_desc*)d_sfd, (uint32_t)b->n_slots, b->d_dump);
        if (b->all_narrow_aligned) {
            if (!mark("clx_k_predict_1w")) return false;
            hipLaunchKernelGGL(clx_k_predict_1w, dim3(groups), dim3(64), 0, rest, d_out, (const clx_sf_desc*)d_sfd, (uint32_t)b->n_slots, b->d_dump, 1u);
            if (!mark("clx_k_predict_1w_hi")) return false;
            hipLaunchKernelGGL(clx_k_predict_1w_hi, dim3(groups), dim3(64), 0, rest, d_out, (const clx_sf_desc*)d_sfd, (uint32_t)b->n_slots, b->d_dump, 1u);
        } else {
            if (!mark("clx_k_predict")) return false;
            hipLaunchKernelGGL(clx_k_predict, dim3((groups + 1) / 2), dim3(512), 0, rest, d_out,
                               (const clx_sf_desc*)d_sfd, (uint32_t)b->n_slots, b->d_dump);
        }
    } else {
        if (!mark("clx_k_predict_1w")) return false;
        hipLaunchKernelGGL(clx_k_predict_1w, dim3(groups), dim3(64), 0, stream, d_out,
                           (const clx_sf_desc*)d_sfd, (uint32_t)b->n_slots, b->d_dump, 0u);
        if (!mark("clx_k_predict_1w_hi")) return false;                     // groups with a predictor order above 12
        hipLaunchKernelGGL(clx_k_predict_1w_hi, dim3(groups), dim3(64), 0, stream, d_out,
                           (const clx_sf_desc*)d_sfd, (uint32_t)b->n_slots, b->d_dump, 0u);
    }
    if (b->flags & CLX_VERIFY_CRC16) {
        if (!mark("clx_k_crc16")) return false;
        hipLaunchKernelGGL(clx_k_crc16, dim3(crc_grid(b->n)), dim3(256), 0, rest, d_arena,
                           (const clx_dev_frame*)b->d_frames, (uint32_t)b->n, d_results);
    }
    if (side != nullptr && hipEventRecord(ev_side, side) != hipSuccess) return false;
    return true;
}
// Launch the pending submissions of the fused lane path as ONE grid (grid.y = the runs) on one of the two internal streams.
// A launch that cannot be made drops ITS submissions (they had been accepted with CLX_OK): the call that triggered it fails, the
// flights' scratch is marked for clearing, and the batch remembers the failure (launch_failed) until clx_batch_flush /
// clx_batch_results / clx_batch_interleave has reported it once -- a caller that only looks at the flush learns of it there.  Nothing is
// recorded about the launch (who wrote which output buffer last, which launch used which flight) before it and its event are out.
// inputs_ready: the device has been synchronised since the submissions came in -- nothing is recorded on the stream they came in on
// (clx_batch_destroy: that stream may have been the context's own, and the context may be gone).
int launch_pending(clx_batch* b, bool inputs_ready) {
    clx_ctx* ctx = b->ctx;
    if (b->pend.empty()) return CLX_OK;
    const int k = (int)(b->n_merged % (uint64_t)b->n_streams);
    bool went_out = false;                     // some kernel of this launch may have been queued
    auto fail = [&](const char* what, hipError_t e) -> int {
        if (went_out) for (const auto& P : b->pend) b->flight[P.flight].scratch_stale = true;       // (may never reach clx_k_finalize)
        b->pend.clear();
        b->launch_failed = true;
        ctx->last_error = std::string("a merged launch of the lane kernels failed (") + what + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string()) +
                          "); its submissions were dropped";
        return CLX_API_ERROR;
    };
#define LP_TRY(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(#expr, e_); } while (0)
    if (!b->mstream[k]) {
        // the stream, its events and its ticket counter into locals first: the batch gets them only when ALL of them exist (a stream
        // without its events would make every later launch on it fail)
        hipStream_t st = nullptr; hipEvent_t e_in = nullptr, e_done[clx_batch::kEvRing] = {}; clx_pool_state* pool = nullptr;
        hipError_t e0 = hipSuccess;
        if (b->tune_prio) {            // (measurement builds: the even streams above the odd ones, or the other way round)
            int lo_p = 0, hi_p = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);       // (least, greatest priority: greatest is the smaller number)
            const bool high = ((k & 1) == 0) == (b->tune_prio == 1);
            e0 = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, high ? hi_p : lo_p);
        } else e0 = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e0 == hipSuccess) e0 = hipMalloc((void**)&pool, sizeof(clx_pool_state));
        if (e0 == hipSuccess) e0 = hipEventCreateWithFlags(&e_in, hipEventDisableTiming);
        for (auto& e : e_done) if (e0 == hipSuccess) e0 = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (e0 != hipSuccess) {
            for (auto& e : e_done) if (e) (void)hipEventDestroy(e);
            if (e_in) (void)hipEventDestroy(e_in);
            if (pool) (void)hipFree(pool);
            if (st) (void)hipStreamDestroy(st);
            return fail("creating an internal stream", e0);
        }
        b->mstream[k] = st; b->m_in[k] = e_in; b->d_pool[k] = pool;
        for (int j = 0; j < clx_batch::kEvRing; ++j) b->m_done[k][j] = e_done[j];
    }
    hipStream_t ms = b->mstream[k];
    // behind everything queued so far on the stream the submissions came in on (their inputs)
    if (!inputs_ready) {
        LP_TRY(hipEventRecord(b->m_in[k], b->pend_stream));
        LP_TRY(hipStreamWaitEvent(ms, b->m_in[k], 0));
    }
    // (this stream's earlier launches are ahead of this one in the stream) a launch on ANOTHER stream must be waited for when it was
    // the last one to write one of this launch's output buffers -- however many launches ago -- or to use a scratch set this launch
    // uses (only after partial launches: flights are handed out in rotation, `merge` at a time, and merge * n_streams of them make
    // a full round).  (Round 5 tried the scan AHEAD of its launch, on a stream of its own with a third set of flights:
    // profiles/r05_scan_ahead_dropped.txt -- worse.)
    {
        uint64_t need[clx_batch::kMaxStreams] = {};      // per stream: the latest of its launches this one depends on
        for (const auto& P : b->pend) {
            const clx_batch::LaunchRef fl = b->flight_launch[P.flight];
            if (fl.count && fl.stream != k) need[fl.stream] = std::max(need[fl.stream], fl.count);
            const auto it = b->out_writer.find(P.out);
            if (it != b->out_writer.end() && it->second.stream != k) need[it->second.stream] = std::max(need[it->second.stream], it->second.count);
        }
        for (int j = 0; j < clx_batch::kMaxStreams; ++j) {
            if (!need[j] || !b->m_count[j]) continue;        // (references are made of launches that went out: need <= m_count)
            // (an event that has been recorded again since stands for a later launch of the same stream: still correct)
            const uint64_t c = (need[j] <= b->m_count[j] && b->m_count[j] - need[j] < (uint64_t)clx_batch::kEvRing) ? need[j] : b->m_count[j];
            LP_TRY(hipStreamWaitEvent(ms, b->m_done[j][(c - 1) % clx_batch::kEvRing], 0));
        }
    }
    // (the staging upload of the plan went out on whichever caller stream made that submission)
    if (b->up_in_flight) LP_TRY(hipStreamWaitEvent(ms, b->ev_up, 0));
    if (b->out_writer.size() > 4096u) {        // (a caller that never re-uses a buffer: forget the old ones behind a full ordering point)
        for (int a = 0; a < clx_batch::kMaxStreams; ++a)
            for (int j = 0; j < clx_batch::kMaxStreams; ++j)
                if (a != j && b->mstream[a] && b->m_count[j])
                    LP_TRY(hipStreamWaitEvent(b->mstream[a], b->m_done[j][(b->m_count[j] - 1) % clx_batch::kEvRing], 0));
        b->out_writer.clear();
    }
    clx_runs runs;
    std::memset(&runs, 0, sizeof runs);
    unsigned n_runs = 0;
    went_out = true;                           // (from here on the flights' scratch may have been touched)
    for (const auto& P : b->pend) {
        clx_batch::Flight& F = b->flight[P.flight];
        if (++F.gen == 0u) {       // (the generation number wrapped: nothing stale may look current)
            LP_TRY(hipMemsetAsync(F.d_taken, 0, (size_t)((b->n_slots + 63) / 64) * sizeof(uint32_t), ms));
            LP_TRY(hipMemsetAsync(F.d_crc_part, 0, (size_t)std::max<uint64_t>(b->n_slots, 1) * sizeof(clx_crc_part), ms));
            F.gen = 1u;
        }
        if (F.scratch_stale) {     // (clx_k_finalize leaves the scratch cleared behind every run that gets that far)
            LP_TRY(hipMemsetAsync(F.d_sf_start, 0xff, (size_t)std::max<uint64_t>(b->n_slots, 1) * sizeof(uint32_t), ms));
            LP_TRY(hipMemsetAsync(F.d_errkey, 0xff, std::max<size_t>(b->n, 1) * sizeof(uint32_t), ms));
            LP_TRY(hipMemsetAsync(F.d_taken + (b->n_slots + 63) / 64, 0, sizeof(uint32_t), ms));       // (the list of groups left: its count)
            F.scratch_stale = false;
        }
        runs.r[n_runs++] = make_run(b, F, P.arena, P.arena_len, P.out, true);
    }
    int nk = 0;
    auto mark = [&](const char* name) -> bool {          // (clx_batch_set_profiling(b, 2): an event in front of each kernel, one behind the last)
        if (!b->profile_merged) return true;
        if (nk > (name ? clx_batch::kMaxKernels - 1 : clx_batch::kMaxKernels)) { ctx->last_error = "more kernels in one launch than profiling marks"; return false; }
        if (name) b->kname[nk] = name;
        return hip_ok(ctx, hipEventRecord(b->ev[nk++], ms), "hipEventRecord");
    };
    if (!(launch_lanes(b, runs, n_runs, false, ms, mark, b->d_pool[k], k) && hipGetLastError() == hipSuccess)) return fail("kernel launch", hipSuccess);
    if (b->profile_merged) { if (!mark(nullptr)) return fail("hipEventRecord", hipSuccess); b->n_kernels = nk - 1; b->ev_valid = true; b->ev_runs = (int)n_runs; }
    LP_TRY(hipEventRecord(b->m_done[k][b->m_count[k] % clx_batch::kEvRing], ms));
#undef LP_TRY
    // the launch and its event are out: now it is the last writer of its output buffers and the last user of its flights
    ++b->m_count[k];
    for (const auto& P : b->pend) {
        b->out_writer[P.out] = clx_batch::LaunchRef{ k, b->m_count[k] };
        b->flight_launch[P.flight] = clx_batch::LaunchRef{ k, b->m_count[k] };
    }
    b->m_unwaited[k] = true;
    b->pend.clear();
    ++b->n_merged;
    ++b->launches_since_wait;
    return CLX_OK;
}
// make `stream` wait for every pipelined submission that nobody has waited for yet (what is pending is launched first)
int wait_flights(clx_batch* b, hipStream_t stream) {
    if (launch_pending(b) != CLX_OK) { b->launch_failed = false; return CLX_API_ERROR; }       // (reported right here)
    if (b->launch_failed) {        // an earlier launch was dropped and only a submit call has said so: the flush says it too, once
        b->launch_failed = false;
        b->ctx->last_error = "an earlier merged launch of this batch failed; its submissions were dropped";
        return CLX_API_ERROR;
    }
    b->launches_since_wait = 0;
    for (int k = 0; k < clx_batch::kMaxStreams; ++k)
        if (b->m_unwaited[k]) { HIP_TRY(b->ctx, hipStreamWaitEvent(stream, b->m_done[k][(b->m_count[k] - 1) % clx_batch::kEvRing], 0)); b->m_unwaited[k] = false; }
    for (auto& F : b->flight)
        if (F.pending) {
            HIP_TRY(b->ctx, hipStreamWaitEvent(stream, F.ev_done, 0));
            if (F.side_pending) HIP_TRY(b->ctx, hipStreamWaitEvent(stream, F.ev_side, 0));
            F.pending = false; F.side_pending = false;
        }
    return CLX_OK;
}
}  // namespace

extern "C" int clx_batch_run(clx_batch* b, const uint8_t* d_arena, size_t arena_len, int32_t* d_out, void* stream_) {
    if (!b || !b->ctx) return CLX_API_ERROR;
    clx_ctx* ctx = b->ctx;
    if (b->n == 0) return CLX_OK;
    if (!d_arena || !d_out) { ctx->last_error = "null device pointer"; return CLX_API_ERROR; }
    if (((uintptr_t)d_arena & 15u) != 0) { ctx->last_error = "device arena must be 16-byte aligned"; return CLX_API_ERROR; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;
    if (wait_flights(b, stream) != CLX_OK) return CLX_API_ERROR;       // pipelined submissions still in flight touch the same buffers
    b->last_slot = -1;
    b->last_stream = stream;
    if (upload_plan(b, arena_len, stream) != CLX_OK) return CLX_API_ERROR;
    const uint64_t alloc_len = (((uint64_t)arena_len + 15ull) & ~15ull) + 16ull;   // claxon_hip.h: the allocation covers this

    int nk = 0;
    auto mark = [&](const char* name) -> bool {          // event before each kernel (+ one after the last)
        if (!b->profiling) return true;
        if (nk > (name ? clx_batch::kMaxKernels - 1 : clx_batch::kMaxKernels)) { ctx->last_error = "more kernels in one run than profiling marks"; return false; }
        if (name) b->kname[nk] = name;
        return hip_ok(ctx, hipEventRecord(b->ev[nk++], stream), "hipEventRecord");
    };
    const int lanes = use_lanes(b, arena_len);
    if (lanes < 0) return CLX_API_ERROR;
    if (lanes) {
        // the two-wave (latency) build while its workgroups get a CU each (0.50 ms against 1.00 ms at 20k subframes), the fused
        // single-wave (throughput) build beyond (1.28 against 1.34 ms at 48k subframes); CLX_LANES_FUSED / CLX_LANES_SPLIT force one
        const bool split = (b->flags & CLX_LANES_SPLIT) ? true : (b->flags & CLX_LANES_FUSED) ? false : b->choice.lanes_split;
        clx_batch::Flight& F0 = b->flight[0];
        F0.d_results = b->d_results; F0.d_sf_start = b->d_sf_start; F0.d_errkey = b->d_errkey; F0.d_endbits = b->d_endbits; F0.d_taken = b->d_taken;
        F0.d_crc_part = b->d_crc_part; F0.d_crc_todo = b->d_crc_todo;
        F0.d_slot_frame = b->d_slot_frame_run; F0.d_first_slot = b->d_first_slot_run; F0.d_fkey = b->d_fkey;
        if (++F0.gen == 0u) {
            HIP_TRY(ctx, hipMemsetAsync(b->d_taken, 0, (size_t)((b->n_slots + 63) / 64) * sizeof(uint32_t), stream));
            HIP_TRY(ctx, hipMemsetAsync(b->d_crc_part, 0, (size_t)std::max<uint64_t>(b->n_slots, 1) * sizeof(clx_crc_part), stream));
            F0.gen = 1u;
        }
        if (F0.scratch_stale) {
            HIP_TRY(ctx, hipMemsetAsync(b->d_sf_start, 0xff, (size_t)std::max<uint64_t>(b->n_slots, 1) * sizeof(uint32_t), stream));
            HIP_TRY(ctx, hipMemsetAsync(b->d_errkey, 0xff, std::max<size_t>(b->n, 1) * sizeof(uint32_t), stream));
            HIP_TRY(ctx, hipMemsetAsync(b->d_taken + (b->n_slots + 63) / 64, 0, sizeof(uint32_t), stream));
            F0.scratch_stale = false;
        }
        clx_runs runs;
        std::memset(&runs, 0, sizeof runs);
        runs.r[0] = make_run(b, F0, d_arena, arena_len, d_out, !split);
        if (!launch_lanes(b, runs, 1u, split, stream, mark) || hipGetLastError() != hipSuccess) {
            F0.scratch_stale = true;
            ctx->last_error = "a launch of the lane kernels failed";
            return CLX_API_ERROR;
        }
    } else {
        // (K1 writes every slot of every frame on every run; the slots that only pad a stereo pair to an even index are cleared once)
        clx_batch::Flight& F0 = b->flight[0];
        if (F0.sfd_stale) { HIP_TRY(ctx, hipMemsetAsync(b->d_sfd, 0, (size_t)std::max<uint64_t>(b->n_slots, 1) * sizeof(clx_sf_desc), stream)); F0.sfd_stale = false; }
        if (!launch_waves(b, d_arena, alloc_len, d_out, b->d_sfd, b->d_results, stream, mark, k2_latency_build(b))) return CLX_API_ERROR;
    }
    if (b->profiling) { if (!mark(nullptr)) return CLX_API_ERROR; b->n_kernels = nk - 1; b->ev_valid = true; }
    HIP_TRY(ctx, hipGetLastError());
    return CLX_OK;
}

// Pipelined submission (wave path): the same work as clx_batch_run, on one of kDepth internal streams in rotation, so that up to
// kDepth submissions are in flight.  One run alone leaves the machine half idle twice: the Rice stage's last round of waves (10 000
// one-wave workgroups on 8 192 wave slots) runs at a fraction of the occupancy it needs, and the predictor stage is one serial
// chain per subframe on a fraction of the SIMDs.  With several submissions in flight the Rice stages share the machine and the
// predictor stages -- small workgroups (clx_k_predict16) -- are resident side by side; the CRC kernel and the predictor kernels
// for what clx_k_predict16 leaves go to one more stream (launch_waves).  Measured, 10 000 config-3 frames per step: one at a time
// 0.40 ms; four in flight 0.30 ms with a hardware queue per stream (GPU_MAX_HW_QUEUES=8 in the environment; HIP's default of
// 4 makes the internal streams share two: 0.33 ms), CRC-16 included at no extra time.
// Each flight has its own descriptors and results; the caller keeps the OUTPUTS of submissions in flight apart (one whose d_out
// is still being written by an earlier one waits for it -- correct, not overlapped).  `stream` is where the caller's inputs come
// from: the submission starts after everything queued on it so far.  clx_batch_flush makes `stream` wait for everything submitted;
// clx_batch_results does so itself.
namespace {
// which kernels a pipelined submission uses, and how many submissions it keeps in flight
bool submit_wants_lanes(const clx_batch* b) {
    return (b->flags & CLX_PATH_LANES) ? true : (b->flags & CLX_PATH_WAVES) ? false : b->lanes_submit;
}
}  // namespace

extern "C" int clx_batch_submit_lanes(const clx_batch* b) { return b && !b->profiling && submit_wants_lanes(b) && !(b->flags & CLX_LANES_SPLIT) ? 1 : 0; }

extern "C" int clx_batch_submit_depth(const clx_batch* b) {
    if (!b) return 1;
    if (b->profiling) return 1;
    if (submit_wants_lanes(b)) return (b->flags & CLX_LANES_SPLIT) ? 1 : b->merge * b->n_streams;      // (<= kDepthLanes)
    return (b->flags & CLX_K2_THROUGHPUT) ? 1 : clx_batch::kDepthWaves;
}

extern "C" int clx_batch_submit_merge(const clx_batch* b) { return (b && clx_batch_submit_lanes(b)) ? b->merge : 1; }

extern "C" int clx_batch_submit(clx_batch* b, const uint8_t* d_arena, size_t arena_len, int32_t* d_out, void* stream_) {
    if (!b || !b->ctx) return CLX_API_ERROR;
    clx_ctx* ctx = b->ctx;
    if (b->n == 0) return CLX_OK;
    if (!d_arena || !d_out) { ctx->last_error = "null device pointer"; return CLX_API_ERROR; }
    if (((uintptr_t)d_arena & 15u) != 0) { ctx->last_error = "device arena must be 16-byte aligned"; return CLX_API_ERROR; }
    // Which kernels (clx_select_path, `pipelined`): the wave kernels with the multi-wave predictor build, four submissions in
    // flight; or the lane kernels, fused build, twenty-four in flight (two merged launches of twelve) -- a run of those is one serial chain per subframe on a fraction of
    // the machine's registers, and a dozen of them side by side fill it.  The two-wave lane build and the one-wave predictor build
    // gain nothing from company (measured, tools/bench_configs.py): forced by flag they are plain runs.
    bool want_lanes = submit_wants_lanes(b);
    if (want_lanes && (uint64_t)arena_len + 32ull >= (1ull << 32)) {      // (the lane kernels address the arena with 32 bits)
        if (b->flags & CLX_PATH_LANES) { ctx->last_error = "CLX_PATH_LANES needs arena_len < 4 GiB"; return CLX_API_ERROR; }
        want_lanes = false;
    }
    // (a batch that only defaulted to the lane kernels and cannot use them on this arena runs the wave kernels, with THEIR depth)
    const int depth = (!want_lanes && submit_wants_lanes(b)) ? ((b->flags & CLX_K2_THROUGHPUT) ? 1 : (int)clx_batch::kDepthWaves) : clx_batch_submit_depth(b);
    if (depth <= 1) return clx_batch_run(b, d_arena, arena_len, d_out, stream_);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;
    const int slot = (int)(b->n_submitted % (uint64_t)depth);
    clx_batch::Flight& F = b->flight[slot];
    const size_t ns = b->n_slots ? (size_t)b->n_slots : 1, nf = b->n ? b->n : 1;
    if (!F.d_results) {
        if (slot == 0) F.d_results = b->d_results;
        else HIP_TRY(ctx, hipMalloc((void**)&F.d_results, nf * sizeof(clx_frame_result)));
    }
    if (want_lanes) {
        // ---- fused lane kernels: the submission joins the pending ones; kMerge of them go out as one launch (launch_pending)
        if (plan_lanes_data(b) != CLX_OK) return CLX_API_ERROR;       // (with the batch's first pipelined submission when no run needed it)
        if (!F.d_sf_start) {
            if (slot == 0) { F.d_sf_start = b->d_sf_start; F.d_errkey = b->d_errkey; F.d_endbits = b->d_endbits; F.d_taken = b->d_taken;
                             F.d_crc_part = b->d_crc_part; F.d_crc_todo = b->d_crc_todo;
                             F.d_slot_frame = b->d_slot_frame_run; F.d_first_slot = b->d_first_slot_run; F.d_fkey = b->d_fkey; }
            else {
                // everything into locals first: the flight gets its scratch only when ALL of it exists and is filled (d_sf_start is the
                // "allocated" sentinel) -- a failure half way frees what there is and leaves the flight as it was
                uint32_t *sl = nullptr, *fs = nullptr, *fk = nullptr, *todo = nullptr, *sfs = nullptr, *ek = nullptr, *tk = nullptr;
                clx_crc_part* part = nullptr; uint64_t* eb = nullptr;
                const auto all = [&]() -> bool {
                    if (b->n_windows) {      // (its own slot maps, starting as the plan's: clx_k_compose rewrites the windows' parts run by run)
                        if (!hip_ok(ctx, hipMalloc((void**)&sl, ns * sizeof(uint32_t)), "hipMalloc slot map") ||
                            !hip_ok(ctx, hipMalloc((void**)&fs, nf * sizeof(uint32_t)), "hipMalloc first slots") ||
                            !hip_ok(ctx, hipMalloc((void**)&fk, nf * sizeof(uint32_t)), "hipMalloc content classes") ||
                            !hip_ok(ctx, hipMemcpy(sl, b->d_slot_frame, ns * sizeof(uint32_t), hipMemcpyDeviceToDevice), "hipMemcpy slot map") ||
                            !hip_ok(ctx, hipMemcpy(fs, b->d_first_slot, nf * sizeof(uint32_t), hipMemcpyDeviceToDevice), "hipMemcpy first slots") ||
                            !hip_ok(ctx, hipMemset(fk, 0, nf * sizeof(uint32_t)), "hipMemset content classes")) return false;
                    }
                    return hip_ok(ctx, hipMalloc((void**)&part, ns * sizeof(clx_crc_part)), "hipMalloc crc parts") &&
                           hip_ok(ctx, hipMalloc((void**)&todo, nf * sizeof(uint32_t)), "hipMalloc crc todo") &&
                           hip_ok(ctx, hipMemset(part, 0, ns * sizeof(clx_crc_part)), "hipMemset crc parts") &&
                           hip_ok(ctx, hipMalloc((void**)&sfs, ns * sizeof(uint32_t)), "hipMalloc sf_start") &&
                           hip_ok(ctx, hipMalloc((void**)&ek, nf * sizeof(uint32_t)), "hipMalloc errkey") &&
                           hip_ok(ctx, hipMalloc((void**)&eb, nf * sizeof(uint64_t)), "hipMalloc end bits") &&
                           hip_ok(ctx, hipMalloc((void**)&tk, taken_bytes(ns)), "hipMalloc taken") &&
                           // (scratch starts cleared; clx_k_finalize leaves it cleared behind every run)
                           hip_ok(ctx, hipMemset(sfs, 0xff, ns * sizeof(uint32_t)), "hipMemset sf_start") &&
                           hip_ok(ctx, hipMemset(ek, 0xff, nf * sizeof(uint32_t)), "hipMemset errkey") &&
                           hip_ok(ctx, hipMemset(tk, 0, taken_bytes(ns)), "hipMemset taken") &&
                           hip_ok(ctx, hipStreamSynchronize(nullptr), "hipStreamSynchronize");       // (the fills went through the null stream; the launches' streams do not wait for it)
                };
                if (!all()) {
                    const std::string why = ctx->last_error;
                    for (void* q : { (void*)sl, (void*)fs, (void*)fk, (void*)part, (void*)todo, (void*)sfs, (void*)ek, (void*)eb, (void*)tk }) if (q) (void)hipFree(q);
                    ctx->last_error = why;
                    return CLX_API_ERROR;
                }
                F.d_slot_frame = sl; F.d_first_slot = fs; F.d_fkey = fk; F.d_crc_part = part; F.d_crc_todo = todo;
                F.d_errkey = ek; F.d_endbits = eb; F.d_taken = tk;
                F.d_sf_start = sfs;                      // (last: the sentinel)
            }
        }
        // what cannot share a launch with the pending submissions goes after them: another caller stream (the launch waits for
        // ONE stream's inputs), an output buffer one of them writes, a plan that has to be uploaded again (the arena's length)
        bool apart = !b->pend.empty() && b->pend_stream != stream;
        for (const auto& P : b->pend) if (P.out == d_out) apart = true;
        if (b->planned_arena_len != arena_len && b->planned_arena_len != (size_t)-1) {
            if (wait_flights(b, stream) != CLX_OK) return CLX_API_ERROR;       // (the submissions in flight read the plan)
        } else if (apart && launch_pending(b) != CLX_OK) return CLX_API_ERROR;
        if (upload_plan(b, arena_len, stream) != CLX_OK) return CLX_API_ERROR;
        b->pend.push_back(clx_batch::Pending{ d_arena, arena_len, d_out, slot });
        b->pend_stream = stream;
        b->last_slot = slot;
        b->last_stream = stream;
        ++b->n_submitted;
        b->ev_valid = false;
        const int want = (b->first_merge > 0 && b->launches_since_wait == 0) ? std::min(b->first_merge, b->merge) : b->merge;
        if ((int)b->pend.size() >= want && launch_pending(b) != CLX_OK) return CLX_API_ERROR;
        return CLX_OK;
    }
    // ---- wave kernels: a whole run on the flight's own stream
    if (!F.stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&F.stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&F.ev_in, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&F.ev_done, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&F.ev_rice, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&F.ev_side, hipEventDisableTiming));
    }
    if (!b->side_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&b->side_stream, hipStreamNonBlocking));
    if (!F.d_sfd) {
        if (slot == 0) F.d_sfd = b->d_sfd;
        else { HIP_TRY(ctx, hipMalloc((void**)&F.d_sfd, ns * sizeof(clx_sf_desc))); F.sfd_stale = true; }
    }
    // a plan that has to be uploaded again (the arena's length changed) is read by the submissions in flight
    if (b->planned_arena_len != arena_len && b->planned_arena_len != (size_t)-1 && wait_flights(b, stream) != CLX_OK) return CLX_API_ERROR;
    if (upload_plan(b, arena_len, stream) != CLX_OK) return CLX_API_ERROR;
    const uint64_t alloc_len = (((uint64_t)arena_len + 15ull) & ~15ull) + 16ull;
    HIP_TRY(ctx, hipEventRecord(F.ev_in, stream));
    HIP_TRY(ctx, hipStreamWaitEvent(F.stream, F.ev_in, 0));
    // (the flight's previous submission is ahead of this one in the same stream) an output buffer that another flight is still
    // writing cannot take new residuals yet
    for (auto& G : b->flight)
        if (&G != &F && G.pending && G.out == d_out) {
            HIP_TRY(ctx, hipStreamWaitEvent(F.stream, G.ev_done, 0));
            if (G.side_pending) HIP_TRY(ctx, hipStreamWaitEvent(F.stream, G.ev_side, 0));      // (its side-stream kernels write there too)
        }
    // (this flight's previous side-stream kernels used the descriptors and results that are about to be overwritten)
    if (F.side_recorded) HIP_TRY(ctx, hipStreamWaitEvent(F.stream, F.ev_side, 0));
    const auto no_mark = [](const char*) { return true; };
    if (F.sfd_stale) { HIP_TRY(ctx, hipMemsetAsync(F.d_sfd, 0, ns * sizeof(clx_sf_desc), F.stream)); F.sfd_stale = false; }
    if (!launch_waves(b, d_arena, alloc_len, d_out, F.d_sfd, F.d_results, F.stream, no_mark, true, b->side_stream, F.ev_rice, F.ev_side)) return CLX_API_ERROR;
    F.side_pending = true; F.side_recorded = true;
    HIP_TRY(ctx, hipEventRecord(F.ev_done, F.stream));
    F.pending = true;
    F.out = d_out;
    b->last_slot = slot;
    b->last_stream = stream;
    ++b->n_submitted;
    b->ev_valid = false;
    HIP_TRY(ctx, hipGetLastError());
    return CLX_OK;
}

extern "C" int clx_batch_flush(clx_batch* b, void* stream_) {
    if (!b || !b->ctx) return CLX_API_ERROR;
    clx_ctx* ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : (b->last_stream ? b->last_stream : ctx->stream);
    return wait_flights(b, stream);
}

extern "C" int clx_batch_interleave(clx_batch* b, const int32_t* d_planar, void* d_pcm, uint32_t sample_bytes, void* stream_) {
    if (!b || !b->ctx) return CLX_API_ERROR;
    clx_ctx* ctx = b->ctx;
    if (b->n == 0) return CLX_OK;
    if (!d_planar || !d_pcm || sample_bytes < 1u || sample_bytes > 4u) { ctx->last_error = "clx_batch_interleave: bad argument"; return CLX_API_ERROR; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : (b->last_stream ? b->last_stream : ctx->stream);
    if (wait_flights(b, stream) != CLX_OK) return CLX_API_ERROR;       // pipelined submissions still writing d_planar / their results (no-op when none)
    hipLaunchKernelGGL(clx_k_interleave, dim3((unsigned)b->n), dim3(256), 0, stream, d_planar,
                       (const clx_dev_frame*)b->d_frames, (const clx_frame_result*)(b->last_slot > 0 ? b->flight[b->last_slot].d_results : b->d_results), (uint32_t)b->n,
                       (uint8_t*)d_pcm, sample_bytes);
    HIP_TRY(ctx, hipGetLastError());
    return CLX_OK;
}

extern "C" int clx_batch_results(clx_batch* b, clx_frame_result* results) {
    if (!b || !b->ctx || (b->n && !results)) return CLX_API_ERROR;
    clx_ctx* ctx = b->ctx;
    if (b->n == 0) return CLX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = b->last_stream ? b->last_stream : ctx->stream;
    if (clx_batch_flush(b, stream) != CLX_OK) return CLX_API_ERROR;       // (pipelined submissions: the predictor stage too)
    const clx_frame_result* src = b->last_slot > 0 ? b->flight[b->last_slot].d_results : b->d_results;
    HIP_TRY(ctx, hipMemcpyAsync(results, src, b->n * sizeof(clx_frame_result), hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    return CLX_OK;
}

extern "C" int clx_batch_kernel_ms(clx_batch* b, int kernel, float* ms) {
    if (!b || !ms || kernel < 0 || !b->ev_valid || kernel >= b->n_kernels) return CLX_API_ERROR;
    clx_ctx* ctx = b->ctx;
    HIP_TRY(ctx, hipEventSynchronize(b->ev[b->n_kernels]));
    HIP_TRY(ctx, hipEventElapsedTime(ms, b->ev[kernel], b->ev[kernel + 1]));
    return CLX_OK;
}

extern "C" const char* clx_batch_kernel_name(const clx_batch* b, int kernel) {
    if (!b || !b->ev_valid || kernel < 0 || kernel >= b->n_kernels) return nullptr;
    return b->kname[kernel];
}

extern "C" int clx_decode_frames(clx_ctx* ctx, const uint8_t* arena, size_t arena_len,
                                 const clx_frame_desc* frames, size_t n,
                                 int32_t* out, const uint64_t* out_sample_offsets,
                                 clx_frame_result* results, uint32_t flags) {
    if (!ctx) return CLX_API_ERROR;
    if (n == 0) return CLX_OK;
    if (!arena || !frames || !out || !out_sample_offsets || !results) { ctx->last_error = "null argument"; return CLX_API_ERROR; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    clx_batch* b = nullptr;
    int st = clx_batch_create(ctx, frames, n, out_sample_offsets, flags, &b);
    if (st != CLX_OK) return st;
    uint8_t* d_arena = nullptr; int32_t* d_out = nullptr;
    uint64_t out_len = 0;
    for (size_t i = 0; i < n; ++i)
        out_len = std::max<uint64_t>(out_len, out_sample_offsets[i] + (uint64_t)frames[i].n_channels * frames[i].block_size);
    auto cleanup = [&]() {
        if (!(flags & CLX_ARENA_ON_DEVICE) && d_arena) (void)hipFree(d_arena);
        if (!(flags & CLX_OUT_ON_DEVICE) && d_out) (void)hipFree(d_out);
        clx_batch_destroy(b);
    };
    if (flags & CLX_ARENA_ON_DEVICE) d_arena = const_cast<uint8_t*>(arena);
    else {
        const size_t alloc = ((arena_len + 15) & ~(size_t)15) + 32;
        if (!hip_ok(ctx, hipMalloc((void**)&d_arena, alloc), "hipMalloc arena")) { cleanup(); return CLX_API_ERROR; }
        if (!hip_ok(ctx, hipMemsetAsync(d_arena + (alloc - 32), 0, 32, ctx->stream), "memset") ||
            !hip_ok(ctx, hipMemcpyAsync(d_arena, arena, arena_len, hipMemcpyHostToDevice, ctx->stream), "H2D arena")) { cleanup(); return CLX_API_ERROR; }
    }
    if (flags & CLX_OUT_ON_DEVICE) d_out = out;
    else {
        // the whole range comes back to the caller: what no frame covers (and what a failed frame leaves) reads as zeros
        if (!hip_ok(ctx, hipMalloc((void**)&d_out, std::max<uint64_t>(out_len, 1) * sizeof(int32_t)), "hipMalloc out") ||
            !hip_ok(ctx, hipMemsetAsync(d_out, 0, std::max<uint64_t>(out_len, 1) * sizeof(int32_t), ctx->stream), "memset out")) { cleanup(); return CLX_API_ERROR; }
    }
    st = clx_batch_run(b, d_arena, arena_len, d_out, ctx->stream);
    if (st == CLX_OK && !(flags & CLX_OUT_ON_DEVICE))
        hipLaunchKernelGGL(clx_k_clear_failed, dim3((unsigned)n), dim3(256), 0, ctx->stream, d_out, (const clx_dev_frame*)b->d_frames,
                           (const clx_frame_result*)b->d_results, (uint32_t)n);
    if (st == CLX_OK) st = clx_batch_results(b, results);
    if (st == CLX_OK && !(flags & CLX_OUT_ON_DEVICE)) {
        // only blocks of successfully decoded frames are observable (frame.rs:667: Err drops the buffer)
        if (!hip_ok(ctx, hipMemcpyAsync(out, d_out, out_len * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream), "D2H out") ||
            !hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) st = CLX_API_ERROR;
    }
    cleanup();
    return st;
}

// Several contexts (one per GPU, or several on one GPU) decode one batch: SURVEY section 8(e).  Frames are independent
// (frame.rs:667-779 touches only its own bytes and buffer), so the batch is cut into contiguous ranges of near-equal algorithmic
// weight (compressed bytes + 4 B per decoded sample), every range is decoded by one context on a host thread of its own --
// only its slice of the arena goes to that device -- and nothing is exchanged between them.  Host buffers only.
extern "C" int clx_decode_frames_multi(clx_ctx* const* ctxs, size_t n_ctx, const uint8_t* arena, size_t arena_len,
                                       const clx_frame_desc* frames, size_t n, int32_t* out, const uint64_t* out_sample_offsets,
                                       clx_frame_result* results, uint32_t flags) {
    if (!ctxs || n_ctx == 0) return CLX_API_ERROR;
    for (size_t c = 0; c < n_ctx; ++c) if (!ctxs[c]) return CLX_API_ERROR;
    if (n == 0) return CLX_OK;
    if (flags & (CLX_ARENA_ON_DEVICE | CLX_OUT_ON_DEVICE)) { ctxs[0]->last_error = "clx_decode_frames_multi takes host buffers"; return CLX_API_ERROR; }
    if (!arena || !frames || !out || !out_sample_offsets || !results) { ctxs[0]->last_error = "null argument"; return CLX_API_ERROR; }
    // a context's share must own a contiguous piece of `out`: frames in increasing, non-overlapping output order
    for (size_t i = 1; i < n; ++i)
        if (out_sample_offsets[i] < out_sample_offsets[i - 1] + (uint64_t)frames[i - 1].n_channels * frames[i - 1].block_size) n_ctx = 1;
    std::vector<double> cum(n + 1, 0.0);
    for (size_t i = 0; i < n; ++i) {
        const uint64_t avail = frames[i].byte_off < arena_len ? (uint64_t)arena_len - frames[i].byte_off : 0;
        cum[i + 1] = cum[i] + (double)std::min<uint64_t>(frames[i].max_bytes, avail) + 4.0 * frames[i].n_channels * frames[i].block_size;
    }
    std::vector<size_t> cut(n_ctx + 1, n);
    cut[0] = 0;
    for (size_t c = 1; c < n_ctx; ++c) {
        const double target = cum[n] * (double)c / (double)n_ctx;
        size_t i = (size_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        cut[c] = std::min(std::max(i, cut[c - 1]), n);
    }
    std::vector<int> status(n_ctx, CLX_OK);
    auto work = [&](size_t c) {
        const size_t lo = cut[c], hi = cut[c + 1];
        if (hi <= lo) return;
        // this share's slice of the arena (16-byte aligned start) and of the output
        uint64_t a0 = UINT64_MAX, a1 = 0;
        for (size_t i = lo; i < hi; ++i) {
            const uint64_t off = std::min<uint64_t>(frames[i].byte_off, arena_len);
            a0 = std::min(a0, off);
            a1 = std::max(a1, std::min<uint64_t>(off + frames[i].max_bytes, arena_len));
        }
        a0 &= ~15ull;
        std::vector<clx_frame_desc> d(frames + lo, frames + hi);
        std::vector<uint64_t> offs(hi - lo);
        const uint64_t o0 = out_sample_offsets[lo];
        for (size_t i = lo; i < hi; ++i) {
            d[i - lo].byte_off = frames[i].byte_off >= a0 ? frames[i].byte_off - a0 : 0;     // (a frame past the arena keeps failing the same way)
            if (frames[i].byte_off >= arena_len) d[i - lo].byte_off = a1 - a0;
            offs[i - lo] = out_sample_offsets[i] - o0;
        }
        status[c] = clx_decode_frames(ctxs[c], arena + a0, (size_t)(a1 - a0), d.data(), hi - lo, out + o0, offs.data(), results + lo, flags);
    };
    std::vector<std::thread> th;
    for (size_t c = 1; c < n_ctx; ++c) th.emplace_back(work, c);
    work(0);
    for (auto& t : th) t.join();
    for (size_t c = 0; c < n_ctx; ++c) if (status[c] != CLX_OK) { if (c) ctxs[0]->last_error = ctxs[c]->last_error; return status[c]; }
    return CLX_OK;
}

// ------------------------------------------------------------------------------------------------
// Host-to-host decode as a pipeline: the batch is cut into chunks of frames; chunk c's compressed bytes go to the device
// while chunk c-1 is decoded and chunk c-2's PCM comes back -- three slots, a stream each, device buffers and plans kept in the
// context and reused from call to call.  Output either planar i32 (Block layout) or, with sample_bytes != 0, the narrow stage's
// channel-interleaved little-endian PCM (what callers of the reference write out: lib.rs:473-520, examples/decode.rs:48-62),
// which halves the bytes that cross the link for 16-bit audio.  Pinned host buffers (clx_host_alloc) let the copies run
// asynchronously at link speed; pageable ones work, staged by the runtime.
// ------------------------------------------------------------------------------------------------
extern "C" void* clx_host_alloc(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
extern "C" void clx_host_free(void* p) { if (p) (void)hipHostFree(p); }

extern "C" void clx_set_stream_chunk(clx_ctx* ctx, size_t frames_per_chunk) { if (ctx) ctx->stream_chunk = frames_per_chunk; }

extern "C" int clx_decode_frames_stream(clx_ctx* ctx, const uint8_t* arena, size_t arena_len, const clx_frame_desc* frames, size_t n,
                                        void* out, uint32_t sample_bytes, const uint64_t* out_sample_offsets,
                                        clx_frame_result* results, uint32_t flags) {
    if (!ctx) return CLX_API_ERROR;
    if (n == 0) return CLX_OK;
    if (!arena || !frames || !out_sample_offsets || !results || sample_bytes > 4u) { ctx->last_error = "clx_decode_frames_stream: bad argument"; return CLX_API_ERROR; }
    if (flags & (CLX_ARENA_ON_DEVICE | CLX_OUT_ON_DEVICE)) { ctx->last_error = "clx_decode_frames_stream takes host buffers"; return CLX_API_ERROR; }
    for (size_t i = 1; i < n; ++i)
        if (out_sample_offsets[i] < out_sample_offsets[i - 1] + (uint64_t)frames[i - 1].n_channels * frames[i - 1].block_size) {
            ctx->last_error = "clx_decode_frames_stream: frames must be in increasing, non-overlapping output order"; return CLX_API_ERROR;
        }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // Few, large chunks: a chunk's decode lasts at least as long as the predictor kernel's serial chain (0.19 ms however few
    // frames it has), so small chunks cost more than their share of overlap brings (10 000 frames, upload only: 16 chunks 2.50 ms,
    // 4 chunks 1.73, 3 chunks 1.44, 2 chunks 1.45).
    // Round 4: a fourth, smaller chunk at the end (42 / 33 / 19 / 6 %: the call waits for the last chunk's decode, which nothing
    // overlaps) was measured and is worse -- 1.78 against 1.44 ms: a chunk costs ~0.2 ms whatever its size.  Three chunks stay; the
    // last one is the smallest (40 / 40 / 20 %).
    const size_t per = ctx->stream_chunk ? ctx->stream_chunk : std::min<size_t>(std::max<size_t>((n + 2) / 3, 256), 8192);
    std::vector<size_t> cuts;            // chunk c = frames [cuts[c], cuts[c + 1])
    cuts.push_back(0);
    if (!ctx->stream_chunk && n >= 4096 && n <= 3 * 8192) {
        const double share[2] = { 0.40, 0.80 };
        for (double sh : share) cuts.push_back(std::max<size_t>(cuts.back() + 1, (size_t)(sh * (double)n) & ~(size_t)31));
        cuts.push_back(n);
    } else {
        for (size_t lo = per; lo < n; lo += per) cuts.push_back(lo);
        cuts.push_back(n);
    }
    auto harvest = [&](clx_stream_slot& S) -> bool {                                       // results of the slot's finished chunk
        if (S.hi <= S.lo) return true;
        if (!hip_ok(ctx, hipStreamSynchronize(S.st), "sync")) return false;
        std::memcpy(results + S.lo, S.h_res, (S.hi - S.lo) * sizeof(clx_frame_result));
        S.lo = S.hi = 0;
        return true;
    };
    int st = CLX_OK;
    for (size_t c = 0; c + 1 < cuts.size() && st == CLX_OK; ++c) {
        const size_t lo = cuts[c], hi = cuts[c + 1], nc = hi - lo;
        clx_stream_slot& S = ctx->slots[c % 3];
        if (!S.st && !hip_ok(ctx, hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking), "hipStreamCreate")) { st = CLX_API_ERROR; break; }
        if (!harvest(S)) { st = CLX_API_ERROR; break; }                                  // (also: the slot's buffers are free again)
        if (!S.b) {
            S.b = new (std::nothrow) clx_batch();
            if (!S.b) { st = CLX_API_ERROR; break; }
            S.b->ctx = ctx; S.b->device = ctx->device;
        }
        // the chunk's slice of the arena (16-byte aligned start) and of the output
        uint64_t a0 = UINT64_MAX, a1 = 0;
        for (size_t i = lo; i < hi; ++i) {
            const uint64_t off = std::min<uint64_t>(frames[i].byte_off, arena_len);
            a0 = std::min(a0, off);
            a1 = std::max(a1, std::min<uint64_t>(off + frames[i].max_bytes, arena_len));
        }
        a0 &= ~15ull;
        const size_t span = (size_t)(a1 - a0);
        std::vector<clx_frame_desc> d(frames + lo, frames + hi);
        std::vector<uint64_t> offs(nc);
        const uint64_t o0 = out_sample_offsets[lo];
        const uint64_t o1 = out_sample_offsets[hi - 1] + (uint64_t)frames[hi - 1].n_channels * frames[hi - 1].block_size;
        for (size_t i = lo; i < hi; ++i) {
            d[i - lo].byte_off = frames[i].byte_off >= arena_len ? a1 - a0 : frames[i].byte_off - a0;
            offs[i - lo] = out_sample_offsets[i] - o0;
        }
        if (batch_plan(S.b, d.data(), nc, offs.data(), flags) != CLX_OK) { st = CLX_API_ERROR; break; }
        const size_t arena_alloc = ((span + 15) & ~(size_t)15) + 48, out_n = (size_t)(o1 - o0);      // (>= 48: the tail cleared below lies inside it also when no frame has a readable byte)
        if (!grow(ctx, &S.d_arena, &S.arena_cap, arena_alloc, "hipMalloc arena") ||
            !grow(ctx, &S.d_out, &S.out_cap, std::max<size_t>(out_n, 1) * sizeof(int32_t), "hipMalloc out") ||
            (sample_bytes && !grow(ctx, &S.d_pcm, &S.pcm_cap, std::max<size_t>(out_n, 1) * sample_bytes, "hipMalloc pcm"))) { st = CLX_API_ERROR; break; }
        if (S.res_cap < nc) {
            if (S.h_res) (void)hipHostFree(S.h_res);
            S.h_res = nullptr; S.res_cap = 0;
            if (!hip_ok(ctx, hipHostMalloc((void**)&S.h_res, (nc + nc / 4) * sizeof(clx_frame_result), hipHostMallocDefault), "hipHostMalloc results")) { st = CLX_API_ERROR; break; }
            S.res_cap = nc + nc / 4;
        }
        // what no frame covers, and what a failed frame leaves, comes back as zeros: the planar output's failed frames are cleared
        // behind the decode (clx_k_clear_failed), so the buffer itself only needs clearing when the frames leave gaps in it or the
        // narrow stage (which skips failed frames) writes the bytes that go back -- and not at all when nothing goes back
        bool gaps = false;
        for (size_t i = lo + 1; i < hi && !gaps; ++i)
            gaps = out_sample_offsets[i] != out_sample_offsets[i - 1] + (uint64_t)frames[i - 1].n_channels * frames[i - 1].block_size;
        const bool clear_out = out != nullptr && (sample_bytes != 0u || gaps);
        const bool ok =
            hip_ok(ctx, hipMemsetAsync(S.d_arena + (arena_alloc - 48), 0, 48, S.st), "memset") &&
            hip_ok(ctx, hipMemcpyAsync(S.d_arena, arena + a0, span, hipMemcpyHostToDevice, S.st), "H2D arena") &&
            (!clear_out || hip_ok(ctx, hipMemsetAsync(sample_bytes ? (void*)S.d_pcm : (void*)S.d_out, 0, out_n * (sample_bytes ? sample_bytes : 4u), S.st), "memset out"));
        if (!ok) { st = CLX_API_ERROR; break; }
        st = clx_batch_run(S.b, S.d_arena, span, S.d_out, S.st);
        if (st != CLX_OK) break;
        if (sample_bytes) {
            st = clx_batch_interleave(S.b, S.d_out, S.d_pcm, sample_bytes, S.st);
            if (st != CLX_OK) break;
            if (out && !hip_ok(ctx, hipMemcpyAsync((uint8_t*)out + o0 * sample_bytes, S.d_pcm, out_n * sample_bytes, hipMemcpyDeviceToHost, S.st), "D2H pcm")) { st = CLX_API_ERROR; break; }
        } else if (out) {
            hipLaunchKernelGGL(clx_k_clear_failed, dim3((unsigned)nc), dim3(256), 0, S.st, S.d_out, (const clx_dev_frame*)S.b->d_frames,
                               (const clx_frame_result*)S.b->d_results, (uint32_t)nc);
            if (!hip_ok(ctx, hipMemcpyAsync((int32_t*)out + o0, S.d_out, out_n * sizeof(int32_t), hipMemcpyDeviceToHost, S.st), "D2H out")) { st = CLX_API_ERROR; break; }
        }
        if (!hip_ok(ctx, hipMemcpyAsync(S.h_res, S.b->d_results, nc * sizeof(clx_frame_result), hipMemcpyDeviceToHost, S.st), "D2H results")) { st = CLX_API_ERROR; break; }
        S.lo = lo; S.hi = hi;
    }
    for (auto& S : ctx->slots) if (S.st && !harvest(S)) st = CLX_API_ERROR;     // (also after an error: nothing stays in flight)
    return st;
}

extern "C" int clx_interleave(clx_ctx* ctx, const int32_t* planar, const clx_frame_desc* frames, size_t n,
                              const uint64_t* out_sample_offsets, const clx_frame_result* results,
                              void* pcm, uint32_t sample_bytes, uint32_t flags) {
    if (!ctx) return CLX_API_ERROR;
    if (n == 0) return CLX_OK;
    if (!planar || !frames || !out_sample_offsets || !pcm || sample_bytes < 1u || sample_bytes > 4u) { ctx->last_error = "clx_interleave: bad argument"; return CLX_API_ERROR; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<clx_dev_frame> dev(n);
    uint64_t n_slots = 0;
    const long bad = clx_plan_frames(frames, n, out_sample_offsets, dev.data(), &n_slots);
    if (bad >= 0) { ctx->last_error = "clx_interleave: invalid frame descriptor"; return CLX_API_ERROR; }
    uint64_t len = 0;
    for (size_t i = 0; i < n; ++i)
        len = std::max<uint64_t>(len, out_sample_offsets[i] + (uint64_t)frames[i].n_channels * frames[i].block_size);
    clx_dev_frame* d_frames = nullptr; clx_frame_result* d_res = nullptr; int32_t* d_planar = nullptr; uint8_t* d_pcm = nullptr;
    auto cleanup = [&]() {
        if (d_frames) (void)hipFree(d_frames);
        if (d_res) (void)hipFree(d_res);
        if (!(flags & CLX_OUT_ON_DEVICE) && d_planar) (void)hipFree(d_planar);
        if (!(flags & CLX_PCM_ON_DEVICE) && d_pcm) (void)hipFree(d_pcm);
    };
    bool ok = hip_ok(ctx, hipMalloc((void**)&d_frames, n * sizeof(clx_dev_frame)), "hipMalloc frames") &&
              hip_ok(ctx, hipMemcpyAsync(d_frames, dev.data(), n * sizeof(clx_dev_frame), hipMemcpyHostToDevice, ctx->stream), "H2D frames");
    if (ok && results) ok = hip_ok(ctx, hipMalloc((void**)&d_res, n * sizeof(clx_frame_result)), "hipMalloc results") &&
                            hip_ok(ctx, hipMemcpyAsync(d_res, results, n * sizeof(clx_frame_result), hipMemcpyHostToDevice, ctx->stream), "H2D results");
    if (ok) {
        if (flags & CLX_OUT_ON_DEVICE) d_planar = const_cast<int32_t*>(planar);
        else ok = hip_ok(ctx, hipMalloc((void**)&d_planar, std::max<uint64_t>(len, 1) * 4), "hipMalloc planar") &&
                  hip_ok(ctx, hipMemcpyAsync(d_planar, planar, len * 4, hipMemcpyHostToDevice, ctx->stream), "H2D planar");
    }
    if (ok) {
        if (flags & CLX_PCM_ON_DEVICE) d_pcm = (uint8_t*)pcm;
        else ok = hip_ok(ctx, hipMalloc((void**)&d_pcm, std::max<uint64_t>(len, 1) * sample_bytes), "hipMalloc pcm") &&
                  // bytes of skipped frames come back as the caller left them
                  hip_ok(ctx, hipMemcpyAsync(d_pcm, pcm, len * sample_bytes, hipMemcpyHostToDevice, ctx->stream), "H2D pcm");
    }
    if (ok) {
        hipLaunchKernelGGL(clx_k_interleave, dim3((unsigned)n), dim3(256), 0, ctx->stream, d_planar, d_frames, d_res, (uint32_t)n, d_pcm, sample_bytes);
        ok = hip_ok(ctx, hipGetLastError(), "clx_k_interleave");
    }
    if (ok && !(flags & CLX_PCM_ON_DEVICE))
        ok = hip_ok(ctx, hipMemcpyAsync(pcm, d_pcm, len * sample_bytes, hipMemcpyDeviceToHost, ctx->stream), "D2H pcm");
    if (ok) ok = hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync");
    cleanup();
    return ok ? CLX_OK : CLX_API_ERROR;
}

extern "C" int clx_decode_subframes(clx_ctx* ctx, const uint8_t* arena, size_t arena_len,
                                    const uint64_t* byte_offs, const uint16_t* block_sizes,
                                    const uint8_t* bps, size_t n,
                                    int32_t* out, const uint64_t* out_sample_offsets,
                                    clx_frame_result* results, uint32_t flags) {
    if (!ctx) return CLX_API_ERROR;
    if (n == 0) return CLX_OK;
    if (!byte_offs || !block_sizes || !bps) { ctx->last_error = "null argument"; return CLX_API_ERROR; }
    std::vector<clx_frame_desc> descs(n);
    for (size_t i = 0; i < n; ++i) {
        clx_frame_desc& d = descs[i];
        std::memset(&d, 0, sizeof d);
        d.byte_off = byte_offs[i];
        d.max_bytes = 0xffffffffu;
        d.header_bytes = 0;
        d.block_size = block_sizes[i];
        d.n_channels = 1;
        d.channel_assignment = CLX_CH_INDEPENDENT;
        d.bps = bps[i];
        d.reserved[0] = 1;                // bare subframe: no CRC-16 footer
    }
    return clx_decode_frames(ctx, arena, arena_len, descs.data(), n, out, out_sample_offsets, results,
                             flags & ~(uint32_t)CLX_VERIFY_CRC16);
}

// ------------------------------------------------------------------------------------------------
// stream header + STREAMINFO (lib.rs:186-205, 230-307; metadata.rs:214-400).  Other metadata blocks
// are skipped by length (VORBIS_COMMENT parsing is outside the hot-path scope).
// ------------------------------------------------------------------------------------------------
struct clx_tags {
    std::string vendor;
    std::vector<std::pair<std::string, size_t>> comments;     // "NAME=value" and the index of '=' (metadata.rs:97)
};

namespace {
// String::from_utf8 (metadata.rs:441, 495): well-formed UTF-8 only -- no overlong forms, no surrogates, nothing past U+10FFFF
bool valid_utf8(const uint8_t* p, size_t n) {
    size_t i = 0;
    while (i < n) {
        const uint8_t b = p[i];
        if (b < 0x80) { ++i; continue; }
        size_t need; uint32_t cp, min;
        if ((b & 0xe0) == 0xc0) { need = 1; cp = b & 0x1f; min = 0x80; }
        else if ((b & 0xf0) == 0xe0) { need = 2; cp = b & 0x0f; min = 0x800; }
        else if ((b & 0xf8) == 0xf0) { need = 3; cp = b & 0x07; min = 0x10000; }
        else return false;
        if (i + need >= n) return false;                     // truncated sequence
        for (size_t k = 1; k <= need; ++k) {
            if ((p[i + k] & 0xc0) != 0x80) return false;
            cp = (cp << 6) | (p[i + k] & 0x3f);
        }
        if (cp < min || cp > 0x10ffff || (cp >= 0xd800 && cp <= 0xdfff)) return false;
        i += need + 1;
    }
    return true;
}

// read_vorbis_comment_block, metadata.rs:402-513.  `c` stands right behind the block header.
int read_vorbis_comment(ByteCursor& c, uint32_t length, clx_tags& out, uint32_t* msg) {
    if (length < 8) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_VC_TOO_SHORT);
    if (length > 10u * 1024 * 1024) return fail(msg, CLX_UNSUPPORTED, CLX_MSG_VC_TOO_LARGE);
    auto le32 = [&](uint32_t* v) { uint32_t r = 0; for (int i = 0; i < 4; ++i) { uint32_t b; if (!c.u8(&b)) return false; r |= b << (8 * i); } *v = r; return true; };
    auto take = [&](size_t n, const uint8_t** p) { if (n > c.n - c.pos) return false; *p = c.p + c.pos; c.pos += n; return true; };
    uint32_t vendor_len;
    if (!le32(&vendor_len)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    if (vendor_len > length - 8) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_VC_VENDOR_TOO_LONG);
    const uint8_t* bytes;
    if (!take(vendor_len, &bytes)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    if (!valid_utf8(bytes, vendor_len)) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_VC_NOT_UTF8);
    out.vendor.assign((const char*)bytes, vendor_len);
    uint32_t comments_len;
    if (!le32(&comments_len)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    if (comments_len >= length / 4) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_VC_TOO_MANY_ENTRIES);
    uint32_t bytes_left = length - 8 - vendor_len;
    while (bytes_left >= 4 && out.comments.size() < comments_len) {
        uint32_t clen;
        if (!le32(&clen)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
        bytes_left -= 4;
        if (clen > bytes_left) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_VC_COMMENT_TOO_LONG);
        if (clen == 0) { comments_len -= 1; continue; }          // empty comments occur in the wild: skipped (metadata.rs:465-472)
        if (!take(clen, &bytes)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
        bytes_left -= clen;
        size_t sep = clen;
        for (size_t i = 0; i < clen; ++i) if (bytes[i] == '=') { sep = i; break; }
        if (sep == clen) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_VC_NO_EQUALS);
        for (size_t i = 0; i < sep; ++i) if (bytes[i] < 0x20 || bytes[i] > 0x7d) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_VC_NAME_INVALID_BYTE);
        if (!valid_utf8(bytes, clen)) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_VC_NOT_UTF8);
        out.comments.emplace_back(std::string((const char*)bytes, clen), sep);
    }
    if (bytes_left != 0) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_VC_EXCESS_DATA);
    if (out.comments.size() != comments_len) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_VC_WRONG_COUNT);
    return CLX_OK;
}
}  // namespace

namespace {
// read_streaminfo_block (metadata.rs:321-400): the 34 bytes behind the block header, then the reference's sanity checks
int read_streaminfo_body(ByteCursor& c, clx_streaminfo* out, uint32_t* msg) {
    auto be = [&](int nbytes, uint64_t* v) { uint64_t r = 0; for (int i = 0; i < nbytes; ++i) { uint32_t b; if (!c.u8(&b)) return false; r = (r << 8) | b; } *v = r; return true; };
    clx_streaminfo s; std::memset(&s, 0, sizeof s);
    uint64_t v, sr_msb, sr_lsb, bps_ns, ns_lsb;
    if (!be(2, &v)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    s.min_block_size = (uint16_t)v;
    if (!be(2, &v)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    s.max_block_size = (uint16_t)v;
    if (!be(3, &v)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    s.min_frame_size = (uint32_t)v;
    if (!be(3, &v)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    s.max_frame_size = (uint32_t)v;
    if (!be(2, &sr_msb) || !be(1, &sr_lsb) || !be(1, &bps_ns) || !be(4, &ns_lsb)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    s.sample_rate = (uint32_t)((sr_msb << 4) | (sr_lsb >> 4));
    s.channels = (uint32_t)(((sr_lsb >> 1) & 7) + 1);
    s.bits_per_sample = (uint32_t)((((sr_lsb & 1) << 4) | (bps_ns >> 4)) + 1);
    s.samples = ((bps_ns & 15) << 32) | ns_lsb;
    if (c.pos + 16 > c.n) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    std::memcpy(s.md5sum, c.p + c.pos, 16); c.pos += 16;
    if (s.min_block_size > s.max_block_size) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_MIN_BLOCK_GT_MAX_BLOCK);
    if (s.min_block_size < 16) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_BLOCK_SIZE_LT_16);
    if (s.min_frame_size > s.max_frame_size && s.max_frame_size != 0) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_MIN_FRAME_GT_MAX_FRAME);
    if (s.sample_rate == 0 || s.sample_rate > 655350) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_INVALID_SAMPLE_RATE);
    *out = s;
    return CLX_OK;
}
}  // namespace

extern "C" int clx_read_metadata_block(const uint8_t* d, size_t len, uint8_t block_type, uint32_t length,
                                       clx_metadata_block* out, size_t* consumed, uint32_t* msg) {
    if (msg) *msg = CLX_MSG_NONE;
    if (consumed) *consumed = 0;
    if (!out || (!d && len)) return CLX_API_ERROR;
    std::memset(out, 0, sizeof *out);
    out->length = length;
    ByteCursor c{ d, len, 0 };
    auto skip = [&](uint32_t n) { if (n > c.n - c.pos) return false; c.pos += n; return true; };     // ReadBytes::skip, input.rs:269-277
    int st = CLX_OK;
    switch (block_type) {
    case 0:                                                                                    // metadata.rs:266-274
        if (length != 34) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_STREAMINFO_LENGTH);
        out->kind = CLX_BLOCK_STREAMINFO;
        st = read_streaminfo_body(c, &out->streaminfo, msg);
        break;
    case 2: {                                                                                  // read_application_block, metadata.rs:525-551
        if (length < 4) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_APPLICATION_BLOCK_TOO_SHORT);
        if (length > 10u * 1024 * 1024) return fail(msg, CLX_UNSUPPORTED, CLX_MSG_APPLICATION_BLOCK_TOO_LARGE);
        uint32_t id = 0;
        for (int i = 0; i < 4; ++i) { uint32_t b; if (!c.u8(&b)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); id = (id << 8) | b; }
        out->kind = CLX_BLOCK_APPLICATION;
        out->application_id = id;
        out->application_data = d + c.pos;
        out->application_len = length - 4u;
        if (!skip(length - 4u)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
        break;
    }
    case 4: {                                                                                  // metadata.rs:291-294
        std::unique_ptr<clx_tags> t(new clx_tags());
        st = read_vorbis_comment(c, length, *t, msg);
        if (st == CLX_OK) { out->kind = CLX_BLOCK_VORBIS_COMMENT; out->tags = t.release(); }
        break;
    }
    case 127:                                                                                  // metadata.rs:303-306
        return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_INVALID_METADATA_BLOCK_TYPE);
    case 1: case 3: case 5: case 6:                                                            // padding; seek table, cue sheet, picture: "pretend it is padding"
        out->kind = CLX_BLOCK_PADDING;
        if (!skip(length)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
        break;
    default:                                                                                   // metadata.rs:307-316
        out->kind = CLX_BLOCK_RESERVED;
        if (!skip(length)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
        break;
    }
    if (st != CLX_OK) return st;
    if (consumed) *consumed = c.pos;
    return CLX_OK;
}

extern "C" int clx_read_metadata_block_with_header(const uint8_t* d, size_t len, clx_metadata_block* out, int* is_last,
                                                   size_t* consumed, uint32_t* msg) {
    if (msg) *msg = CLX_MSG_NONE;
    if (consumed) *consumed = 0;
    if (is_last) *is_last = 0;
    if (!out || (!d && len)) return CLX_API_ERROR;
    if (len < 4) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);                       // read_metadata_block_header, metadata.rs:214-231
    const uint32_t length = ((uint32_t)d[1] << 16) | ((uint32_t)d[2] << 8) | d[3];
    size_t used = 0;
    const int st = clx_read_metadata_block(d + 4, len - 4, (uint8_t)(d[0] & 0x7fu), length, out, &used, msg);
    if (st != CLX_OK) return st;
    if (is_last) *is_last = (d[0] >> 7) & 1;
    if (consumed) *consumed = used + 4;
    return CLX_OK;
}

extern "C" int clx_describe_packets(const uint8_t* arena, size_t arena_len, const uint64_t* offs, const uint32_t* lens, size_t n,
                                    int check_crc, clx_frame_desc* descs, clx_frame_header* headers, clx_frame_result* results) {
    if (n == 0) return CLX_OK;
    if (!arena || !offs || !lens) return CLX_API_ERROR;
    int first_bad = CLX_OK;
    for (size_t i = 0; i < n; ++i) {
        if (offs[i] > arena_len || lens[i] > arena_len - offs[i]) return CLX_API_ERROR;
        clx_frame_header h;
        uint32_t m = CLX_MSG_NONE;
        const int st = clx_parse_frame_header(arena + offs[i], lens[i], check_crc, &h, &m);
        if (results) { results[i].status = st; results[i].msg = m; results[i].end_bit = 0; }
        if (headers) headers[i] = h;
        if (descs) {
            clx_frame_desc& d = descs[i];
            std::memset(&d, 0, sizeof d);
            d.byte_off = offs[i];
            d.max_bytes = lens[i];
            if (st == CLX_OK) {
                d.header_bytes = h.header_bytes; d.block_size = h.block_size; d.n_channels = h.n_channels;
                d.channel_assignment = h.channel_assignment; d.bps = h.bps;
            }
        }
        if (st != CLX_OK && first_bad == CLX_OK) first_bad = st;
    }
    return first_bad;
}

extern "C" int clx_read_stream_header_ext(const uint8_t* d, size_t len, uint32_t options, clx_streaminfo* info,
                                          size_t* audio_offset, clx_tags** tags_out, uint32_t* msg) {
    if (msg) *msg = CLX_MSG_NONE;
    if (tags_out) *tags_out = nullptr;
    if (!info || !audio_offset || (!d && len)) return CLX_API_ERROR;
    ByteCursor c{ d, len, 0 };
    auto be = [&](int nbytes, uint64_t* v) { uint64_t r = 0; for (int i = 0; i < nbytes; ++i) { uint32_t b; if (!c.u8(&b)) return false; r = (r << 8) | b; } *v = r; return true; };
    uint64_t magic;
    if (!be(4, &magic)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    if (magic != 0x664c6143ull)
        return fail(msg, CLX_FORMAT_ERROR, (magic & 0xffffff00ull) == 0x49443300ull ? CLX_MSG_ID3_HEADER : CLX_MSG_INVALID_STREAM_HEADER);
    const bool metadata_only = (options & CLX_OPT_METADATA_ONLY) != 0;
    bool want_vc = (options & CLX_OPT_NO_VORBIS_COMMENT) == 0;     // opts_current.read_vorbis_comment (lib.rs:232, 265)
    std::unique_ptr<clx_tags> vc;
    bool first = true;
    for (;;) {
        uint32_t hb; uint64_t length;
        if (!c.u8(&hb) || !be(3, &length)) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
        const bool is_last = (hb >> 7) == 1;
        const uint32_t type = hb & 0x7fu;
        if (type == 0) {
            if (length != 34) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_STREAMINFO_LENGTH);
            clx_streaminfo s;
            const int sst = read_streaminfo_body(c, &s, msg);
            if (sst != CLX_OK) return sst;
            if (!first) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_SECOND_STREAMINFO);
            *info = s;
        } else if (type == 4) {
            // the block is always parsed (metadata.rs:291-294), so its errors surface whatever the options say
            std::unique_ptr<clx_tags> t(new clx_tags());
            const int st = read_vorbis_comment(c, (uint32_t)length, *t, msg);
            if (st != CLX_OK) return st;
            if (first) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_STREAMINFO_MISSING);
            if (vc) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_SECOND_VORBIS_COMMENT);       // lib.rs:257-259
            vc = std::move(t);
            want_vc = false;
        } else {
            if (type == 127) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_INVALID_METADATA_BLOCK_TYPE);
            if (type == 2) {
                if (length < 4) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_APPLICATION_BLOCK_TOO_SHORT);
                if (length > 10u * 1024 * 1024) return fail(msg, CLX_UNSUPPORTED, CLX_MSG_APPLICATION_BLOCK_TOO_LARGE);
            }
            if (c.pos + length > len) return fail(msg, CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
            c.pos += (size_t)length;
            if (first) return fail(msg, CLX_FORMAT_ERROR, CLX_MSG_STREAMINFO_MISSING);
        }
        // the streaminfo block is read on its own (lib.rs:243-248); the early-out is only evaluated after a LATER block
        // (lib.rs:273-277), so even a metadata-only reader that wants no tags reads the block after the streaminfo
        const bool was_first = first;
        first = false;
        if (is_last) break;
        if (!was_first && metadata_only && !want_vc) break;
    }
    if (options & CLX_OPT_NO_VORBIS_COMMENT) vc.reset();                                      // lib.rs:283-285
    if (tags_out) *tags_out = vc.release();
    *audio_offset = c.pos;
    return CLX_OK;
}

extern "C" int clx_read_stream_header(const uint8_t* d, size_t len, clx_streaminfo* info, size_t* audio_offset, uint32_t* msg) {
    return clx_read_stream_header_ext(d, len, 0, info, audio_offset, nullptr, msg);
}

extern "C" const char* clx_tags_vendor(const clx_tags* t, size_t* len) {
    if (!t) { if (len) *len = 0; return nullptr; }
    if (len) *len = t->vendor.size();
    return t->vendor.c_str();
}
extern "C" size_t clx_tags_count(const clx_tags* t) { return t ? t->comments.size() : 0; }
extern "C" int clx_tags_get(const clx_tags* t, size_t i, const char** name, size_t* name_len, const char** value, size_t* value_len) {
    if (!t || i >= t->comments.size()) return CLX_API_ERROR;
    const std::string& s = t->comments[i].first;
    const size_t sep = t->comments[i].second;
    if (name) *name = s.data();
    if (name_len) *name_len = sep;
    if (value) *value = s.data() + sep + 1;
    if (value_len) *value_len = s.size() - sep - 1;
    return CLX_OK;
}
extern "C" const char* clx_tags_lookup(const clx_tags* t, const char* name, size_t occurrence, size_t* value_len) {
    if (!t || !name) return nullptr;
    const size_t nlen = std::strlen(name);
    auto lower = [](unsigned char ch) { return (ch >= 'A' && ch <= 'Z') ? (unsigned char)(ch + 32) : ch; };    // eq_ignore_ascii_case
    for (const auto& c : t->comments) {
        if (c.second != nlen) continue;
        bool eq = true;
        for (size_t i = 0; i < nlen && eq; ++i) eq = lower((unsigned char)c.first[i]) == lower((unsigned char)name[i]);
        if (!eq) continue;
        if (occurrence == 0) { if (value_len) *value_len = c.first.size() - c.second - 1; return c.first.data() + c.second + 1; }
        --occurrence;
    }
    return nullptr;
}
extern "C" void clx_tags_free(clx_tags* t) { delete t; }

// ------------------------------------------------------------------------------------------------
// frame indexer (host): sync code + CRC-8-valid header, chain confirmed by the previous frame's CRC-16
// ------------------------------------------------------------------------------------------------
extern "C" int clx_index_frames(const uint8_t* data, size_t len, size_t start_off,
                                clx_frame_desc* descs, clx_frame_header* headers, size_t cap,
                                size_t* n_found, size_t* stop_off) {
    if (!n_found || !stop_off || (!data && len) || (cap && !descs)) return CLX_API_ERROR;
    *n_found = 0; *stop_off = start_off;
    const CrcTables& t = crc_tables();
    size_t pos = start_off;
    while (*n_found < cap && pos + 2 <= len) {
        clx_frame_header h; uint32_t m;
        if (clx_parse_frame_header(data + pos, len - pos, 1, &h, &m) != CLX_OK) break;
        // find the end: the next CRC-8-valid header whose preceding two bytes are this frame's CRC-16,
        // or the end of the stream
        uint16_t crc = 0;
        size_t q = pos;
        size_t found_end = 0;
        const size_t min_end = pos + h.header_bytes + 2;
        for (;;) {
            // crc covers [pos, q)
            if (q + 2 >= min_end && q + 2 <= len) {
                const uint16_t footer = (uint16_t)((data[q] << 8) | data[q + 1]);
                if (footer == crc) {
                    const size_t e = q + 2;
                    if (e == len) { found_end = e; break; }
                    if (e + 2 <= len && data[e] == 0xff && (data[e + 1] & 0xfe) == 0xf8) {
                        clx_frame_header nh; uint32_t nm;
                        if (clx_parse_frame_header(data + e, len - e, 1, &nh, &nm) == CLX_OK) { found_end = e; break; }
                    }
                }
            }
            if (q >= len) break;
            crc = (uint16_t)((crc << 8) ^ t.t16[(uint8_t)(crc >> 8) ^ data[q]]);
            ++q;
        }
        if (!found_end) break;
        clx_frame_desc& d = descs[*n_found];
        std::memset(&d, 0, sizeof d);
        d.byte_off = pos;
        d.max_bytes = (uint32_t)std::min<size_t>(len - pos, 0xffffffffu);    // rest of the stream, as the reference's reader sees it
        d.header_bytes = h.header_bytes;
        d.block_size = h.block_size;
        d.n_channels = h.n_channels;
        d.channel_assignment = h.channel_assignment;
        d.bps = h.bps;
        if (headers) headers[*n_found] = h;
        ++*n_found;
        pos = found_end;
    }
    *stop_off = pos;
    return CLX_OK;
}

// Device frame indexer: same contract and same answer as clx_index_frames; the byte scan and the
// CRC-16 of every byte run on the GPU (K5-K7 in clx_kernels.hip), the chain walk over the few candidates here.
// `scan_end` <= len bounds the bytes that are looked at: with scan_end < len the stream goes on behind the window, so a frame
// may only end at a later candidate inside it (never at the window's edge), and descriptors still say "readable to the end
// of the stream".  Frames are appended to `descs` / `hdrs` (at most `cap`).
static int index_frames_device_impl(clx_ctx* ctx, const uint8_t* data, size_t len, size_t scan_end, size_t start_off,
                                    std::vector<clx_frame_desc>& descs, std::vector<clx_frame_header>* hdrs, size_t cap,
                                    size_t* stop_off, uint32_t flags) {
    *stop_off = start_off;
    if (scan_end > len) scan_end = len;
    if (start_off + 2 > scan_end || cap == 0) return CLX_OK;
    const bool to_eos = scan_end == len;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint8_t* d_data = nullptr; uint64_t* d_cand = nullptr; uint32_t* d_count = nullptr; uint8_t* d_hdr = nullptr;
    uint64_t* d_pos = nullptr; uint16_t* d_crc = nullptr;
    auto cleanup = [&]() {
        if (!(flags & CLX_ARENA_ON_DEVICE) && d_data) (void)hipFree(d_data);
        if (d_cand) (void)hipFree(d_cand);
        if (d_count) (void)hipFree(d_count);
        if (d_hdr) (void)hipFree(d_hdr);
        if (d_pos) (void)hipFree(d_pos);
        if (d_crc) (void)hipFree(d_crc);
    };
    auto fail_api = [&](const char* what) { if (what) ctx->last_error = what; cleanup(); return (int)CLX_API_ERROR; };
    if (flags & CLX_ARENA_ON_DEVICE) d_data = const_cast<uint8_t*>(data);      // (16-byte aligned, padded like a decode arena)
    else {
        const size_t alloc = ((scan_end + 15) & ~(size_t)15) + 32;
        if (!hip_ok(ctx, hipMalloc((void**)&d_data, alloc), "hipMalloc stream") ||
            !hip_ok(ctx, hipMemsetAsync(d_data + (alloc - 48), 0, 48, st), "memset") ||
            !hip_ok(ctx, hipMemcpyAsync(d_data, data, scan_end, hipMemcpyHostToDevice, st), "H2D stream")) return fail_api(nullptr);
    }
    // ---- K5: candidates (a frame is at least 10 bytes long, and candidates need a CRC-8 match on top: a list of one entry
    //      per 16 scanned bytes cannot overflow on real streams; adversarial input is caught by the count check)
    const uint64_t scan0 = (uint64_t)start_off & ~15ull;
    const uint32_t cand_cap = (uint32_t)std::min<uint64_t>((scan_end - scan0) / 16 + 4096, 1u << 26);
    if (!hip_ok(ctx, hipMalloc((void**)&d_cand, (size_t)cand_cap * 8), "hipMalloc candidates") ||
        !hip_ok(ctx, hipMalloc((void**)&d_count, 4), "hipMalloc count") ||
        !hip_ok(ctx, hipMemsetAsync(d_count, 0, 4, st), "memset")) return fail_api(nullptr);
    const uint64_t n_threads = (scan_end - scan0 + 15) / 16;
    hipLaunchKernelGGL(clx_k_find_headers, dim3((unsigned)((n_threads + 255) / 256)), dim3(256), 0, st, d_data, (uint64_t)scan_end,
                       (uint64_t)start_off, d_cand, cand_cap, d_count);
    uint32_t count = 0;
    if (!hip_ok(ctx, hipMemcpyAsync(&count, d_count, 4, hipMemcpyDeviceToHost, st), "D2H count") ||
        !hip_ok(ctx, hipStreamSynchronize(st), "sync")) return fail_api(nullptr);
    if (count > cand_cap) return fail_api("clx_index_frames_device: candidate list overflow");
    std::vector<uint64_t> cand(count);
    if (count && (!hip_ok(ctx, hipMemcpyAsync(cand.data(), d_cand, (size_t)count * 8, hipMemcpyDeviceToHost, st), "D2H candidates") ||
                  !hip_ok(ctx, hipStreamSynchronize(st), "sync"))) return fail_api(nullptr);
    std::sort(cand.begin(), cand.end());
    // ---- K7 + the authoritative host parse of every candidate
    std::vector<uint64_t> pos; std::vector<clx_frame_header> hdr;
    if (count) {
        std::vector<uint8_t> bytes((size_t)count * 20);
        if (!hip_ok(ctx, hipMalloc((void**)&d_hdr, bytes.size()), "hipMalloc headers") ||
            !hip_ok(ctx, hipMemcpyAsync(d_cand, cand.data(), (size_t)count * 8, hipMemcpyHostToDevice, st), "H2D candidates")) return fail_api(nullptr);
        hipLaunchKernelGGL(clx_k_gather_headers, dim3((count + 255) / 256), dim3(256), 0, st, d_data, (uint64_t)scan_end, d_cand, count, d_hdr);
        if (!hip_ok(ctx, hipMemcpyAsync(bytes.data(), d_hdr, bytes.size(), hipMemcpyDeviceToHost, st), "D2H headers") ||
            !hip_ok(ctx, hipStreamSynchronize(st), "sync")) return fail_api(nullptr);
        for (uint32_t i = 0; i < count; ++i) {
            clx_frame_header h; uint32_t m;
            const size_t avail = std::min<size_t>(20, scan_end - (size_t)cand[i]);
            if (clx_parse_frame_header(bytes.data() + (size_t)i * 20, avail, 1, &h, &m) == CLX_OK) { pos.push_back(cand[i]); hdr.push_back(h); }
        }
    }
    if (pos.empty() || pos[0] != start_off) { cleanup(); return CLX_OK; }          // no frame starts at start_off
    // ---- K6: CRC-16 of the span between consecutive candidates (the last one runs to the end of the scanned bytes)
    const uint32_t m = (uint32_t)pos.size();
    pos.push_back(scan_end);
    std::vector<uint16_t> crc(m);
    if (!hip_ok(ctx, hipMalloc((void**)&d_pos, (size_t)(m + 1) * 8), "hipMalloc pos") ||
        !hip_ok(ctx, hipMalloc((void**)&d_crc, (size_t)m * 2), "hipMalloc crc") ||
        !hip_ok(ctx, hipMemcpyAsync(d_pos, pos.data(), (size_t)(m + 1) * 8, hipMemcpyHostToDevice, st), "H2D pos")) return fail_api(nullptr);
    hipLaunchKernelGGL(clx_k_span_crc16, dim3(m), dim3(64), 0, st, d_data, d_pos, m, d_crc);
    if (!hip_ok(ctx, hipMemcpyAsync(crc.data(), d_crc, (size_t)m * 2, hipMemcpyDeviceToHost, st), "D2H crc") ||
        !hip_ok(ctx, hipStreamSynchronize(st), "sync")) return fail_api(nullptr);
    cleanup();
    // ---- chain: frame i ends at the first later candidate (or the end of the stream) e >= its header + 2 with
    //      crc16([pos_i, e)) == 0, i.e. whose two preceding bytes are the frame's CRC-16
    const uint32_t last_end = to_eos ? m : m - 1;            // the window's edge is not a frame end unless it is the stream's
    uint32_t cur = 0;
    size_t n_found = 0;
    while (n_found < cap && cur < m) {
        uint32_t acc = 0, end = 0;
        for (uint32_t j = cur + 1; j <= last_end; ++j) {
            acc = clx_gf_mulmod(acc, clx_xpow8_64(pos[j] - pos[j - 1])) ^ crc[j - 1];
            if (acc == 0u && pos[j] >= pos[cur] + hdr[cur].header_bytes + 2u) { end = j; break; }
        }
        if (!end) break;
        clx_frame_desc d;
        std::memset(&d, 0, sizeof d);
        d.byte_off = pos[cur];
        d.max_bytes = (uint32_t)std::min<uint64_t>(len - pos[cur], 0xffffffffu);
        d.header_bytes = hdr[cur].header_bytes;
        d.block_size = hdr[cur].block_size;
        d.n_channels = hdr[cur].n_channels;
        d.channel_assignment = hdr[cur].channel_assignment;
        d.bps = hdr[cur].bps;
        descs.push_back(d);
        if (hdrs) hdrs->push_back(hdr[cur]);
        ++n_found;
        cur = end;
    }
    *stop_off = (size_t)pos[cur];
    return CLX_OK;
}

// Device frame indexer: same contract and same answer as clx_index_frames; the byte scan and the
// CRC-16 of every byte run on the GPU (K5-K7 in clx_kernels.hip), the chain walk over the few candidates here.
extern "C" int clx_index_frames_device(clx_ctx* ctx, const uint8_t* data, size_t len, size_t start_off,
                                       clx_frame_desc* descs, clx_frame_header* headers, size_t cap,
                                       size_t* n_found, size_t* stop_off, uint32_t flags) {
    if (!ctx) return CLX_API_ERROR;
    if (!n_found || !stop_off || (!data && len) || (cap && !descs)) { ctx->last_error = "clx_index_frames_device: bad argument"; return CLX_API_ERROR; }
    *n_found = 0; *stop_off = start_off;
    std::vector<clx_frame_desc> d; std::vector<clx_frame_header> h;
    const int st = index_frames_device_impl(ctx, data, len, len, start_off, d, headers ? &h : nullptr, cap, stop_off, flags);
    if (st != CLX_OK) return st;
    std::copy(d.begin(), d.end(), descs);
    if (headers) std::copy(h.begin(), h.end(), headers);
    *n_found = d.size();
    return CLX_OK;
}

// ------------------------------------------------------------------------------------------------
// claxon::FrameReader / FlacReader / Block (host/claxon.hpp) -- the reference's API surface
// ------------------------------------------------------------------------------------------------
namespace claxon {

Error Error::from(int status, uint32_t msg) {
    Error e;
    e.kind = status == CLX_IO_ERROR ? ErrorKind::IoError : status == CLX_FORMAT_ERROR ? ErrorKind::FormatError
           : status == CLX_UNSUPPORTED ? ErrorKind::Unsupported : ErrorKind::Api;
    e.status = status; e.msg = msg; e.text = clx_message(msg);
    return e;
}

struct FrameReader::Impl {
    clx_ctx* ctx = nullptr;
    std::vector<uint8_t> data;        // the stream (host)
    uint8_t* d_arena = nullptr;       // same bytes on the device, uploaded once
    size_t pos = 0;                   // next undecoded byte
    size_t batch_frames = 4096;
    // frame index of the rest of the stream (speculative: confirmed frame by frame against the decode), and how far it is used
    std::vector<clx_frame_desc> idx_descs; std::vector<clx_frame_header> idx_hdrs; size_t idx_next = 0;
    struct Pending { clx_frame_header hdr; int status; uint32_t msg; std::vector<int32_t> samples; };
    std::vector<Pending> queue;
    size_t qhead = 0;
    bool failed = false; int fail_status = 0; uint32_t fail_msg = 0;
    int device = 0;                   // (kept by value: the destructor must not look into a context that may be gone)
    // one re-plannable batch, one output buffer on the device and one pinned on the host, reused by every fill of the queue
    clx_batch* batch = nullptr;
    int32_t* d_out = nullptr; size_t out_cap = 0;
    int32_t* h_out = nullptr; size_t h_out_cap = 0;
    ~Impl() {
        (void)hipSetDevice(device);
        if (batch) clx_batch_destroy(batch);
        if (d_out) (void)hipFree(d_out);
        if (h_out) (void)hipHostFree(h_out);
        if (d_arena) (void)hipFree(d_arena);
    }
};

FrameReader::FrameReader(clx_ctx* ctx, const uint8_t* data, size_t len) : impl_(new Impl()) {
    impl_->ctx = ctx;
    impl_->device = ctx ? ctx->device : 0;
    impl_->data.assign(data, data + len);
}
FrameReader::FrameReader(FrameReader&& o) noexcept : impl_(o.impl_) { o.impl_ = nullptr; }
FrameReader::~FrameReader() { delete impl_; }
size_t FrameReader::position() const { return impl_->pos; }
std::pair<std::vector<uint8_t>, size_t> FrameReader::into_inner() && {
    std::pair<std::vector<uint8_t>, size_t> r(std::move(impl_->data), impl_->pos);
    delete impl_;
    impl_ = nullptr;
    return r;
}
void FrameReader::set_batch_frames(size_t n) { impl_->batch_frames = n ? n : 1; }

// Decode the next batch of frames starting at impl_->pos into the queue.
static int fill_queue(FrameReader::Impl& I) {
    I.queue.clear(); I.qhead = 0;
    const uint8_t* data = I.data.data();
    const size_t len = I.data.size();
    // header of the frame at pos decides what the reference would do first (frame.rs:674-692)
    clx_frame_header h0; uint32_t m0 = 0;
    int st0 = clx_parse_frame_header(data + I.pos, len - I.pos, 1, &h0, &m0);
    if (st0 != CLX_OK) {
        FrameReader::Impl::Pending p{}; p.status = st0; p.msg = m0;
        I.queue.push_back(std::move(p));
        return CLX_OK;
    }
    auto upload = [&]() -> bool {
        if (I.d_arena) return true;
        if (hipSetDevice(I.ctx->device) != hipSuccess) return false;
        const size_t alloc = ((len + 15) & ~(size_t)15) + 32;
        if (hipMalloc((void**)&I.d_arena, alloc) != hipSuccess) return false;
        return hipMemset(I.d_arena, 0, alloc) == hipSuccess && hipMemcpy(I.d_arena, data, len, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (I.idx_next >= I.idx_descs.size() || I.idx_descs[I.idx_next].byte_off != I.pos) {
        // (re)index from here: long remainders on the device in one go, short ones on the host a batch at a time
        I.idx_next = 0;
        size_t nf = 0, stop = 0;
        if (len - I.pos >= (256u << 10)) {
            // a window of the stream at a time: the index costs memory in proportion to the frames found, and a stream whose
            // chain keeps breaking pays for one window per break, not for the whole remainder
            if (!upload()) return CLX_API_ERROR;
            const size_t window = (size_t)64 << 20;
            const size_t scan_end = len - I.pos > window + (window >> 2) ? I.pos + window : len;
            I.idx_descs.clear(); I.idx_hdrs.clear();
            const int st = index_frames_device_impl(I.ctx, I.d_arena, len, scan_end, I.pos, I.idx_descs, &I.idx_hdrs, (size_t)-1, &stop,
                                                    CLX_ARENA_ON_DEVICE);
            if (st != CLX_OK) return st;
            nf = I.idx_descs.size();
        } else {
            I.idx_descs.resize(I.batch_frames); I.idx_hdrs.resize(I.batch_frames);
            clx_index_frames(data, len, I.pos, I.idx_descs.data(), I.idx_hdrs.data(), I.batch_frames, &nf, &stop);
        }
        I.idx_descs.resize(nf); I.idx_hdrs.resize(nf);
    }
    size_t n = std::min(I.batch_frames, I.idx_descs.size() - I.idx_next);
    std::vector<clx_frame_desc> descs(I.idx_descs.begin() + (ptrdiff_t)I.idx_next, I.idx_descs.begin() + (ptrdiff_t)(I.idx_next + n));
    std::vector<clx_frame_header> hdrs(I.idx_hdrs.begin() + (ptrdiff_t)I.idx_next, I.idx_hdrs.begin() + (ptrdiff_t)(I.idx_next + n));
    I.idx_next += n;
    if (n == 0) { descs.resize(1); hdrs.resize(1); }
    if (n == 0) {
        // the chain could not be confirmed from here: decode this one frame against the rest of the stream
        n = 1; hdrs[0] = h0;
        clx_frame_desc& d = descs[0];
        std::memset(&d, 0, sizeof d);
        d.byte_off = I.pos; d.max_bytes = (uint32_t)std::min<size_t>(len - I.pos, 0xffffffffu);
        d.header_bytes = h0.header_bytes; d.block_size = h0.block_size; d.n_channels = h0.n_channels;
        d.channel_assignment = h0.channel_assignment; d.bps = h0.bps;
    }
    // frames whose header carries no bps are Unsupported before any subframe is read (frame.rs:687-692)
    size_t usable = n;
    for (size_t i = 0; i < n; ++i) if (descs[i].bps == 0) { usable = i; break; }
    std::vector<clx_frame_result> results(usable);
    std::vector<uint64_t> offs(usable);
    uint64_t total = 0;
    for (size_t i = 0; i < usable; ++i) { offs[i] = total; total += (uint64_t)descs[i].n_channels * descs[i].block_size; }
    const int32_t* host_out = I.h_out;
    if (usable) {
        clx_ctx* ctx = I.ctx;
        if (hipSetDevice(ctx->device) != hipSuccess) return CLX_API_ERROR;
        if (!upload()) return CLX_API_ERROR;
        if (!I.batch) {
            I.batch = new (std::nothrow) clx_batch();
            if (!I.batch) return CLX_API_ERROR;
            I.batch->ctx = ctx; I.batch->device = ctx->device;
        }
        if (batch_plan(I.batch, descs.data(), usable, offs.data(), CLX_VERIFY_CRC16) != CLX_OK) return CLX_API_ERROR;
        if (!grow(ctx, &I.d_out, &I.out_cap, std::max<uint64_t>(total, 1) * sizeof(int32_t), "hipMalloc out")) return CLX_API_ERROR;
        if (I.h_out_cap < total) {
            if (I.h_out) (void)hipHostFree(I.h_out);
            I.h_out = nullptr; I.h_out_cap = 0;
            if (!hip_ok(ctx, hipHostMalloc((void**)&I.h_out, (size_t)(total + total / 4 + 1) * sizeof(int32_t), hipHostMallocDefault), "hipHostMalloc out")) return CLX_API_ERROR;
            I.h_out_cap = (size_t)(total + total / 4 + 1);
        }
        host_out = I.h_out;
        int st = clx_batch_run(I.batch, I.d_arena, len, I.d_out, ctx->stream);
        if (st == CLX_OK && !hip_ok(ctx, hipMemcpyAsync(I.h_out, I.d_out, total * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream), "D2H out")) st = CLX_API_ERROR;
        if (st == CLX_OK) st = clx_batch_results(I.batch, results.data());        // (waits for the stream: the samples are there too)
        if (st != CLX_OK) return st;
    }
    size_t expect = I.pos;
    for (size_t i = 0; i < n; ++i) {
        FrameReader::Impl::Pending p{};
        p.hdr = hdrs[i];
        if (descs[i].byte_off != expect) break;             // the speculative index diverged from the true chain
        if (i >= usable) { p.status = CLX_UNSUPPORTED; p.msg = CLX_MSG_NO_BPS_IN_HEADER; I.queue.push_back(std::move(p)); break; }
        p.status = results[i].status; p.msg = results[i].msg;
        if (p.status == CLX_OK) {
            const size_t cnt = (size_t)descs[i].n_channels * descs[i].block_size;
            p.samples.assign(host_out + offs[i], host_out + offs[i] + cnt);
            expect = (size_t)descs[i].byte_off + (size_t)((results[i].end_bit + 7) / 8) + 2;
            I.queue.push_back(std::move(p));
        } else { I.queue.push_back(std::move(p)); break; }
    }
    I.pos = expect;
    return CLX_OK;
}

FrameResult FrameReader::read_next_or_eof(std::vector<int32_t> buffer) {
    Impl& I = *impl_;
    FrameResult r;
    if (I.failed) { r.error = Error::from(I.fail_status, I.fail_msg); r.is_err = true; return r; }
    if (I.qhead >= I.queue.size()) {
        int st = fill_queue(I);
        if (st != CLX_OK) { r.is_err = true; r.error = Error::from(CLX_API_ERROR, 0); r.error.text = clx_last_error(I.ctx); return r; }
    }
    Impl::Pending& p = I.queue[I.qhead++];
    if (p.status == CLX_END_OF_STREAM) { --I.qhead; r.has_block = false; return r; }      // Ok(None), stays at EOF
    if (p.status != CLX_OK) {
        I.failed = true; I.fail_status = p.status; I.fail_msg = p.msg;
        r.is_err = true; r.error = Error::from(p.status, p.msg); return r;
    }
    // ensure_buffer_len (frame.rs:616-637) then overwrite every sample
    buffer.resize(p.samples.size());
    std::copy(p.samples.begin(), p.samples.end(), buffer.begin());
    r.has_block = true;
    r.block = Block(p.hdr.time, p.hdr.block_size, std::move(buffer));
    return r;
}

struct FlacReader::Impl {
    clx_streaminfo info{};
    FrameReader* frames = nullptr;
    clx_tags* tags = nullptr;
    ~Impl() { delete frames; clx_tags_free(tags); }
};

FlacReader::FlacReader() : impl_(new Impl()) {}
FlacReader::FlacReader(FlacReader&& o) noexcept : impl_(o.impl_) { o.impl_ = nullptr; }
FlacReader::~FlacReader() { delete impl_; }

Result<FlacReader> FlacReader::create_ext(clx_ctx* ctx, const uint8_t* data, size_t len, FlacReaderOptions opts) {
    Result<FlacReader> r;
    clx_streaminfo si; size_t off = 0; uint32_t msg = 0;
    clx_tags* tags = nullptr;
    const uint32_t o = (opts.metadata_only ? (uint32_t)CLX_OPT_METADATA_ONLY : 0u) | (opts.read_vorbis_comment ? 0u : (uint32_t)CLX_OPT_NO_VORBIS_COMMENT);
    int st = clx_read_stream_header_ext(data, len, o, &si, &off, &tags, &msg);
    if (st != CLX_OK) { r.is_err = true; r.error = Error::from(st, msg); return r; }
    r.value.impl_->info = si;
    r.value.impl_->tags = tags;
    if (!opts.metadata_only) r.value.impl_->frames = new FrameReader(ctx, data + off, len - off);
    return r;
}
Result<FlacReader> FlacReader::create(clx_ctx* ctx, const uint8_t* data, size_t len) { return create_ext(ctx, data, len, FlacReaderOptions()); }

Result<FlacReader> FlacReader::open_ext(clx_ctx* ctx, const char* path, FlacReaderOptions opts) {
    Result<FlacReader> r;
    FILE* f = std::fopen(path, "rb");
    if (!f) { r.is_err = true; r.error = Error::from(CLX_IO_ERROR, CLX_MSG_NONE); r.error.text = "cannot open file"; return r; }
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + got);
    std::fclose(f);
    return create_ext(ctx, data.data(), data.size(), opts);
}
Result<FlacReader> FlacReader::open(clx_ctx* ctx, const char* path) { return open_ext(ctx, path, FlacReaderOptions()); }

const clx_streaminfo& FlacReader::streaminfo() const { return impl_->info; }
const clx_tags* FlacReader::raw_tags() const { return impl_->tags; }
bool FlacReader::vendor(std::string* out) const {
    if (!impl_->tags) return false;
    size_t n = 0; const char* v = clx_tags_vendor(impl_->tags, &n);
    if (out) out->assign(v, n);
    return true;
}
std::vector<std::pair<std::string, std::string>> FlacReader::tags() const {
    std::vector<std::pair<std::string, std::string>> out;
    for (size_t i = 0; i < clx_tags_count(impl_->tags); ++i) {
        const char *n, *v; size_t nl, vl;
        clx_tags_get(impl_->tags, i, &n, &nl, &v, &vl);
        out.emplace_back(std::string(n, nl), std::string(v, vl));
    }
    return out;
}
std::vector<std::string> FlacReader::get_tag(const char* name) const {
    std::vector<std::string> out;
    for (size_t k = 0;; ++k) {
        size_t vl = 0; const char* v = clx_tags_lookup(impl_->tags, name, k, &vl);
        if (!v) break;
        out.emplace_back(v, vl);
    }
    return out;
}
FrameReader& FlacReader::blocks() {
    // lib.rs:367-373 panics with this message
    if (!impl_->frames) throw std::logic_error("FlacReaderOptions::metadata_only must be false to be able to use blocks()");
    return *impl_->frames;
}

}  // namespace claxon

// ------------------------------------------------------------------------------------------------
// C handles over the C++ reader
// ------------------------------------------------------------------------------------------------
struct clx_reader { claxon::FlacReader reader; std::vector<int32_t> recycle; claxon::FrameResult pending; bool have_pending = false; };
extern "C" const clx_tags* clx_reader_tags(const clx_reader* r) { return r ? r->reader.raw_tags() : nullptr; }

extern "C" int clx_reader_new(clx_ctx* ctx, const uint8_t* data, size_t len, clx_reader** out, uint32_t* msg) {
    if (msg) *msg = CLX_MSG_NONE;
    if (!ctx || !out || (!data && len)) return CLX_API_ERROR;
    *out = nullptr;
    auto r = claxon::FlacReader::create(ctx, data, len);
    if (r.is_err) { if (msg) *msg = r.error.msg; return r.error.status; }
    *out = new clx_reader{ std::move(r.value), {}, {}, false };
    return CLX_OK;
}

extern "C" int clx_reader_open(clx_ctx* ctx, const char* path, clx_reader** out, uint32_t* msg) {
    if (msg) *msg = CLX_MSG_NONE;
    if (!ctx || !out || !path) return CLX_API_ERROR;
    *out = nullptr;
    auto r = claxon::FlacReader::open(ctx, path);
    if (r.is_err) { if (msg) *msg = r.error.msg; return r.error.status; }
    *out = new clx_reader{ std::move(r.value), {}, {}, false };
    return CLX_OK;
}

extern "C" int clx_reader_streaminfo(const clx_reader* r, clx_streaminfo* out) {
    if (!r || !out) return CLX_API_ERROR;
    *out = r->reader.streaminfo();
    return CLX_OK;
}

extern "C" int clx_reader_next_block(clx_reader* r, int32_t* buffer, size_t cap, clx_block_info* info, uint32_t* msg) {
    if (msg) *msg = CLX_MSG_NONE;
    if (!r || !info) return CLX_API_ERROR;
    // a block that does not fit stays pending: the call can be repeated with a larger buffer (info says how large)
    if (!r->have_pending) {
        r->pending = r->reader.blocks().read_next_or_eof(std::move(r->recycle));
        r->recycle.clear();
        r->have_pending = true;
    }
    claxon::FrameResult& fr = r->pending;
    if (fr.is_err) { if (msg) *msg = fr.error.msg; return fr.error.status; }
    if (!fr.has_block) return CLX_END_OF_STREAM;
    info->time = fr.block.time(); info->block_size = fr.block.duration(); info->channels = fr.block.channels();
    if ((size_t)fr.block.len() > cap || (!buffer && fr.block.len())) return CLX_API_ERROR;
    r->have_pending = false;
    std::vector<int32_t> buf = fr.block.into_buffer();
    std::memcpy(buffer, buf.data(), buf.size() * sizeof(int32_t));
    r->recycle = std::move(buf);
    return CLX_OK;
}

extern "C" void clx_reader_close(clx_reader* r) { delete r; }

#ifdef CLX_TIMELINE
// debug aid, see clx_device.h / tools/timeline.py: kernel 0 = residual, 1 = predict, 2 = scan, 3 = lanes
extern "C" int clx_debug_timeline_reset(void) {
    const uint32_t z[4] = { 0u, 0u, 0u, 0u };
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(clx_timeline_n), z, sizeof z, 0, hipMemcpyHostToDevice);
}
extern "C" int clx_debug_timeline_count(int kernel, uint32_t* n) {
    return (int)hipMemcpyFromSymbol(n, HIP_SYMBOL(clx_timeline_n), sizeof(uint32_t), (size_t)kernel * sizeof(uint32_t), hipMemcpyDeviceToHost);
}
extern "C" int clx_debug_timeline(int kernel, void* host, size_t n_waves) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(clx_timeline_buf), n_waves * 14 * sizeof(uint64_t),
                                    (size_t)kernel * CLX_TL_WAVES * 14 * sizeof(uint64_t), hipMemcpyDeviceToHost);
}
#endif
