// claxon.hpp -- C++ host mirror of the reference's API surface for the frame path:
//   claxon::Error / Result      <- error.rs:18-32, 97
//   claxon::Block               <- frame.rs:402-529
//   claxon::FrameReader         <- frame.rs:603-609, 650-785
//   claxon::decode_packets      <- the per-packet loop of examples/decode_ogg.rs / decode_mp4.rs, as one batch
//   claxon::FlacReader          <- lib.rs:93-97, 207-471
//   claxon::FlacSamples         <- lib.rs:169-178, 473-520   (FlacIntoSamples <- lib.rs:181-184, 417, 522-560)
//   claxon::MetadataBlock, read_metadata_block[_with_header], MetadataBlockReader <- metadata.rs:104-131, 244-319, 553-603
// Same names, argument meaning and error behaviour; decoding itself happens on the GPU through
// the C ABI in include/claxon_hip.h (frames are indexed on the host and decoded in batches).
#ifndef CLAXON_HPP
#define CLAXON_HPP

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/claxon_hip.h"

namespace claxon {

enum class ErrorKind { IoError, FormatError, Unsupported, Api };

// error.rs:18-32.  Equality is variant + message (error.rs:34-45); IoError never compares equal.
struct Error {
    ErrorKind kind = ErrorKind::Api;
    int status = CLX_API_ERROR;
    uint32_t msg = CLX_MSG_NONE;
    std::string text;
    static Error from(int status, uint32_t msg);
    bool operator==(const Error& o) const {
        if (kind == ErrorKind::IoError || o.kind == ErrorKind::IoError) return false;
        return kind == o.kind && text == o.text;
    }
};

template <typename T>
struct Result {
    bool is_err = false;
    Error error;
    T value;
    bool is_ok() const { return !is_err; }
};

// An iterator over the stereo sample pairs in a block (frame.rs:548-580).
class StereoSamples {
public:
    StereoSamples(const int32_t* buf, uint32_t duration) : buf_(buf), duration_(duration), cur_(0) {}
    bool next(std::pair<int32_t, int32_t>* out) {
        if (cur_ == duration_) return false;
        *out = { buf_[cur_], buf_[cur_ + duration_] };
        ++cur_;
        return true;
    }
private:
    const int32_t* buf_; uint32_t duration_, cur_;
};

// A block of raw audio samples: planar i32, channel c at [c*block_size, (c+1)*block_size) (frame.rs:402-411).
class Block {
public:
    Block() : first_sample_number_(0), block_size_(0), channels_(0) {}
    Block(uint64_t time, uint32_t bs, std::vector<int32_t> buffer)                      // frame.rs:414-421
        : first_sample_number_(time), block_size_(bs), channels_(bs ? (uint32_t)(buffer.size() / bs) : 0), buffer_(std::move(buffer)) {}
    static Block empty() { return Block(); }                                            // frame.rs:424-431
    uint64_t time() const { return first_sample_number_; }                              // frame.rs:438
    uint32_t len() const { return block_size_ * channels_; }                            // frame.rs:446
    uint32_t duration() const { return block_size_; }                                   // frame.rs:459
    uint32_t channels() const { return channels_; }                                     // frame.rs:467
    // frame.rs:477-481; throws std::out_of_range where the reference panics
    const int32_t* channel(uint32_t ch) const {
        if (ch >= channels_) throw std::out_of_range("Block::channel");
        return buffer_.data() + (size_t)ch * block_size_;
    }
    int32_t sample(uint32_t ch, uint32_t s) const { return buffer_.at((size_t)ch * block_size_ + s); }   // frame.rs:494-497
    std::vector<int32_t> into_buffer() { return std::move(buffer_); }                   // frame.rs:503-505
    StereoSamples stereo_samples() const {                                              // frame.rs:516-528
        if (channels_ != 2) throw std::logic_error("stereo_samples() must only be called for blocks with two channels.");
        return StereoSamples(buffer_.data(), block_size_);
    }
private:
    uint64_t first_sample_number_;
    uint32_t block_size_, channels_;
    std::vector<int32_t> buffer_;
};

// FrameResult = Result<Option<Block>> (frame.rs:609)
struct FrameResult {
    bool is_err = false;
    Error error;
    bool has_block = false;
    Block block;
};

// Reads frames from an in-memory stream positioned at a frame header (frame.rs:603-605).
class FrameReader {
public:
    struct Impl;
    FrameReader(clx_ctx* ctx, const uint8_t* data, size_t len);                         // FrameReader::new, frame.rs:652
    FrameReader(FrameReader&&) noexcept;
    FrameReader(const FrameReader&) = delete;
    ~FrameReader();
    // Decodes the next frame (frame.rs:667).  The buffer is moved into the returned block and can be
    // recovered with Block::into_buffer().  Frames are decoded on the device in batches ahead of the caller.
    FrameResult read_next_or_eof(std::vector<int32_t> buffer);
    size_t position() const;              // bytes consumed from the stream handed to the constructor
    // Destroys the frame reader and returns the underlying reader (frame.rs:782): here the stream's bytes and
    // position() -- the end of the last frame decoded, which lies ahead of the last block handed out while blocks of the
    // current device batch are still queued; equal to it once read_next_or_eof has returned the end of the stream.
    std::pair<std::vector<uint8_t>, size_t> into_inner() &&;
    void set_batch_frames(size_t n);      // how many frames to index + decode per device batch
private:
    Impl* impl_;
};

// One frame per packet, a batch of packets at once: what a container demuxer's loop does with the reference --
// `FrameReader::new(Cursor::new(&packet.data)).read_next_or_eof(buffer)` per packet (examples/decode_ogg.rs:105-114,
// decode_mp4.rs:143-152) -- with the headers parsed on the host and all frames decoded on the device in one launch
// (the batch entry next to read_next_or_eof).  Packet i is arena[offs[i] .. offs[i] + lens[i]).
// Result i is what read_next_or_eof would return for it: a block, Ok(None) for a packet too short to hold a sync code
// (the examples skip empty packets), or the reference's error.
inline std::vector<FrameResult> decode_packets(clx_ctx* ctx, const uint8_t* arena, size_t arena_len, const uint64_t* offs,
                                               const uint32_t* lens, size_t n, bool check_crc = true) {
    std::vector<FrameResult> out(n);
    std::vector<clx_frame_desc> descs(n);
    std::vector<clx_frame_header> hdrs(n);
    std::vector<clx_frame_result> hres(n);
    if (n == 0) return out;
    if (clx_describe_packets(arena, arena_len, offs, lens, n, check_crc ? 1 : 0, descs.data(), hdrs.data(), hres.data()) == CLX_API_ERROR) {
        for (auto& r : out) { r.is_err = true; r.error = Error::from(CLX_API_ERROR, CLX_MSG_NONE); }
        return out;
    }
    std::vector<size_t> which;                    // packets that go to the device
    std::vector<clx_frame_desc> d2;
    std::vector<uint64_t> out_offs;
    uint64_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        if (hres[i].status == CLX_END_OF_STREAM) continue;                                     // Ok(None), frame.rs:140-143
        if (hres[i].status != CLX_OK) { out[i].is_err = true; out[i].error = Error::from(hres[i].status, hres[i].msg); continue; }
        if (hdrs[i].bps == 0) { out[i].is_err = true; out[i].error = Error::from(CLX_UNSUPPORTED, CLX_MSG_NO_BPS_IN_HEADER); continue; }   // frame.rs:687-692
        which.push_back(i); d2.push_back(descs[i]); out_offs.push_back(total);
        total += (uint64_t)hdrs[i].block_size * hdrs[i].n_channels;
    }
    if (which.empty()) return out;
    std::vector<int32_t> pcm((size_t)total);
    std::vector<clx_frame_result> res(which.size());
    const int st = clx_decode_frames(ctx, arena, arena_len, d2.data(), d2.size(), pcm.data(), out_offs.data(), res.data(),
                                     check_crc ? CLX_VERIFY_CRC16 : 0u);
    for (size_t j = 0; j < which.size(); ++j) {
        FrameResult& r = out[which[j]];
        const clx_frame_header& h = hdrs[which[j]];
        if (st != CLX_OK) { r.is_err = true; r.error = Error::from(CLX_API_ERROR, CLX_MSG_NONE); r.error.text = clx_last_error(ctx); continue; }
        if (res[j].status != CLX_OK) { r.is_err = true; r.error = Error::from(res[j].status, res[j].msg); continue; }
        const size_t len = (size_t)h.block_size * h.n_channels;
        r.has_block = true;
        r.block = Block(h.time, h.block_size, std::vector<int32_t>(pcm.begin() + (ptrdiff_t)out_offs[j], pcm.begin() + (ptrdiff_t)(out_offs[j] + len)));
    }
    return out;
}

class FlacReader;

// An iterator that yields interleaved samples (lib.rs:169-178, 473-520).
class FlacSamples {
public:
    explicit FlacSamples(FrameReader& fr) : fr_(fr) {}
    // returns false at the end of the stream; *err set when a read failed (then iteration ends)
    bool next(int32_t* sample, Error* err, bool* failed) {
        *failed = false;
        if (has_failed_) return false;
        ++channel_;
        if (channel_ >= block_.channels()) {
            channel_ = 0;
            ++sample_;
            if (sample_ >= block_.duration()) {
                sample_ = 0;
                Block cur = std::move(block_);
                block_ = Block::empty();
                FrameResult r = fr_.read_next_or_eof(cur.into_buffer());
                if (r.is_err) { has_failed_ = true; *failed = true; *err = r.error; return true; }
                if (!r.has_block) return false;
                block_ = std::move(r.block);
            }
        }
        *sample = block_.sample(channel_, sample_);
        return true;
    }
private:
    FrameReader& fr_;
    Block block_;
    uint32_t sample_ = 0, channel_ = 0;
    bool has_failed_ = false;
};

// A metadata block of the stream (metadata.rs:104-131).  CueSheet / Picture / SeekTable are never produced: the reference
// reads those blocks as Padding (its TODOs at metadata.rs:287-305).
struct MetadataBlock {
    enum class Kind { StreamInfo, Padding, Application, SeekTable, VorbisComment, CueSheet, Picture, Reserved };
    Kind kind = Kind::Reserved;
    clx_streaminfo streaminfo{};                                     // StreamInfo(..)
    uint32_t length = 0;                                             // Padding { length }
    uint32_t id = 0;                                                 // Application { id, data }
    std::vector<uint8_t> data;
    std::string vendor;                                              // VorbisComment { vendor, comments }: "NAME=value" split at '='
    std::vector<std::pair<std::string, std::string>> comments;
};

namespace detail {
inline Result<MetadataBlock> metadata_block_result(int st, uint32_t msg, clx_metadata_block& b) {
    Result<MetadataBlock> r;
    if (st != CLX_OK) { r.is_err = true; r.error = Error::from(st, msg); return r; }
    MetadataBlock& m = r.value;
    m.length = b.length;
    switch (b.kind) {
    case CLX_BLOCK_STREAMINFO: m.kind = MetadataBlock::Kind::StreamInfo; m.streaminfo = b.streaminfo; break;
    case CLX_BLOCK_PADDING: m.kind = MetadataBlock::Kind::Padding; break;
    case CLX_BLOCK_APPLICATION:
        m.kind = MetadataBlock::Kind::Application; m.id = b.application_id;
        m.data.assign(b.application_data, b.application_data + b.application_len);
        break;
    case CLX_BLOCK_VORBIS_COMMENT: {
        m.kind = MetadataBlock::Kind::VorbisComment;
        size_t n = 0;
        const char* v = clx_tags_vendor(b.tags, &n);
        m.vendor.assign(v ? v : "", n);
        for (size_t i = 0; i < clx_tags_count(b.tags); ++i) {
            const char *name, *value; size_t ln, lv;
            if (clx_tags_get(b.tags, i, &name, &ln, &value, &lv) == CLX_OK) m.comments.emplace_back(std::string(name, ln), std::string(value, lv));
        }
        clx_tags_free(b.tags);
        break;
    }
    default: m.kind = MetadataBlock::Kind::Reserved; break;
    }
    return r;
}
}  // namespace detail

// Read a single metadata block of the given type and length from `data` (metadata.rs:261): for streams embedded in a
// container, e.g. the MP4 "FLAC specific box".  *consumed (may be null) receives the bytes read.
inline Result<MetadataBlock> read_metadata_block(const uint8_t* data, size_t len, uint8_t block_type, uint32_t length, size_t* consumed = nullptr) {
    clx_metadata_block b; uint32_t msg = 0; size_t used = 0;
    const int st = clx_read_metadata_block(data, len, block_type, length, &b, &used, &msg);
    if (consumed) *consumed = used;
    return detail::metadata_block_result(st, msg, b);
}
// Read a single metadata block header and body (metadata.rs:244): e.g. an Ogg packet of the FLAC mapping.
inline Result<MetadataBlock> read_metadata_block_with_header(const uint8_t* data, size_t len, size_t* consumed = nullptr, bool* is_last = nullptr) {
    clx_metadata_block b; uint32_t msg = 0; size_t used = 0; int last = 0;
    const int st = clx_read_metadata_block_with_header(data, len, &b, &last, &used, &msg);
    if (consumed) *consumed = used;
    if (is_last) *is_last = last != 0;
    return detail::metadata_block_result(st, msg, b);
}

// Reads metadata blocks from a stream positioned at a block header and yields them one by one (metadata.rs:553-603):
// at least one element; after the block flagged as last, or after an error, next() returns false.
class MetadataBlockReader {
public:
    MetadataBlockReader(const uint8_t* data, size_t len) : data_(data), len_(len) {}                 // metadata.rs:567
    bool next(Result<MetadataBlock>* out) {                                                        // metadata.rs:582-597
        if (done_) return false;
        size_t used = 0; bool last = false;
        *out = read_metadata_block_with_header(data_ + pos_, len_ - pos_, &used, &last);
        pos_ += used;
        done_ = out->is_err || last;
        return true;
    }
    size_t position() const { return pos_; }
private:
    const uint8_t* data_; size_t len_, pos_ = 0; bool done_ = false;
};

// Controls what FlacReader reads when it is constructed (lib.rs:100-140).
struct FlacReaderOptions {
    bool metadata_only = false;        // stop after the metadata blocks; blocks() / samples() are unavailable (lib.rs:113)
    bool read_vorbis_comment = true;   // parse the VORBIS_COMMENT block (lib.rs:125)
};

// A FLAC decoder over a stream held in memory (lib.rs:93-97).
class FlacReader {
public:
    struct Impl;
    FlacReader(FlacReader&&) noexcept;
    FlacReader(const FlacReader&) = delete;
    ~FlacReader();
    static Result<FlacReader> open(clx_ctx* ctx, const char* path);                      // lib.rs:455
    static Result<FlacReader> create(clx_ctx* ctx, const uint8_t* data, size_t len);     // FlacReader::new, lib.rs:217
    static Result<FlacReader> open_ext(clx_ctx* ctx, const char* path, FlacReaderOptions opts);                       // lib.rs:462
    static Result<FlacReader> create_ext(clx_ctx* ctx, const uint8_t* data, size_t len, FlacReaderOptions opts);      // FlacReader::new_ext, lib.rs:227
    const clx_streaminfo& streaminfo() const;                                            // lib.rs:312
    bool vendor(std::string* out) const;                                                 // lib.rs:321 (false: no Vorbis comment block)
    std::vector<std::pair<std::string, std::string>> tags() const;                       // lib.rs:335 (name, value) in stream order
    std::vector<std::string> get_tag(const char* name) const;                            // lib.rs:356 ASCII-case-insensitive, all matches
    const clx_tags* raw_tags() const;
    // lib.rs:367 / 396; both throw std::logic_error where the reference panics: the reader was made with metadata_only
    FrameReader& blocks();
    FlacSamples samples() { return FlacSamples(blocks()); }
    FlacReader();
    friend class FlacIntoSamples;
private:
    friend struct Result<FlacReader>;
    Impl* impl_;
};

// An iterator that yields samples and owns its reader (FlacReader::into_samples, lib.rs:181-184, 417-433): for
// callers that want to hand the iterator on without keeping the reader alive themselves.
class FlacIntoSamples {
public:
    explicit FlacIntoSamples(FlacReader&& reader) : reader_(std::move(reader)), samples_(reader_.blocks()) {}
    FlacIntoSamples(const FlacIntoSamples&) = delete;
    bool next(int32_t* sample, Error* err, bool* failed) { return samples_.next(sample, err, failed); }
    FlacReader& reader() { return reader_; }
private:
    FlacReader reader_;
    FlacSamples samples_;
};
inline FlacIntoSamples into_samples(FlacReader&& reader) { return FlacIntoSamples(std::move(reader)); }

}  // namespace claxon

#endif
