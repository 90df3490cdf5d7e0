// claxon.hpp -- C++ host mirror of the reference's API surface for the frame path:
//   claxon::Error / Result      <- error.rs:18-32, 97
//   claxon::Block               <- frame.rs:402-529
//   claxon::FrameReader         <- frame.rs:603-609, 650-785
//   claxon::FlacReader          <- lib.rs:93-97, 207-471
//   claxon::FlacSamples         <- lib.rs:169-178, 473-520
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
    void set_batch_frames(size_t n);      // how many frames to index + decode per device batch
private:
    Impl* impl_;
};

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
private:
    friend struct Result<FlacReader>;
    Impl* impl_;
};

}  // namespace claxon

#endif
