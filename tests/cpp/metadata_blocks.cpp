// The metadata side of the C++ mirror (claxon.hpp: MetadataBlock, read_metadata_block_with_header, MetadataBlockReader)
// on the streams handed over as arguments: prints one line per block; tests/test_metadata_blocks.py compares the lines
// with what the C ABI and the oracle say.  Needs no GPU (metadata is parsed on the host).
//   usage: metadata_blocks <file.flac>...
#include <cinttypes>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../claxon_amd/csrc/host/claxon.hpp"

static const char* kind_name(claxon::MetadataBlock::Kind k) {
    using K = claxon::MetadataBlock::Kind;
    switch (k) {
    case K::StreamInfo: return "StreamInfo"; case K::Padding: return "Padding"; case K::Application: return "Application";
    case K::VorbisComment: return "VorbisComment"; case K::Reserved: return "Reserved"; default: return "Other";
    }
}

// tests/testsamples.rs:404-426 (verify_limits_on_vendor_string, verify_limits_on_vorbis_comment_block) against
// FlacReader::open: both files fail while the metadata is read, before anything would be decoded -- no device needed.
static int verify_limits(const std::string& dir) {
    int bad = 0;
    {
        auto r = claxon::FlacReader::open(nullptr, (dir + "/large_vendor_string.flac").c_str());
        claxon::Error want; want.kind = claxon::ErrorKind::FormatError; want.text = "vendor string too long";
        const bool ok = !r.is_ok() && r.error == want;
        std::printf("verify_limits_on_vendor_string %s (%s)\n", ok ? "ok" : "FAILED", r.error.text.c_str());
        bad += !ok;
    }
    {
        auto r = claxon::FlacReader::open(nullptr, (dir + "/large_vorbis_comment_block.flac").c_str());
        const bool ok = !r.is_ok() && r.error.kind == claxon::ErrorKind::Unsupported;
        std::printf("verify_limits_on_vorbis_comment_block %s (%s)\n", ok ? "ok" : "FAILED", r.error.text.c_str());
        bad += !ok;
    }
    return bad;
}

// verify_block_sample / verify_block_stereo_samples_iterator (frame.rs:531-543, 582-597) on claxon::Block.  (The
// reference's second test builds a two-channel block over a longer buffer; here the channel count follows from the
// buffer, so the block holds the six samples the iterator visits.)
static int verify_block() {
    int bad = 0;
    claxon::Block b(0, 5, std::vector<int32_t>{ 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47 });
    bad += !(b.sample(0, 2) == 5 && b.sample(1, 3) == 23 && b.sample(2, 4) == 47);
    bad += !(b.len() == 15 && b.duration() == 5 && b.channels() == 3 && b.time() == 0 && b.channel(1)[0] == 13);
    bool threw = false;
    try { (void)b.stereo_samples(); } catch (const std::logic_error&) { threw = true; }      // frame.rs:517-519 panics
    bad += !threw;
    claxon::Block s(0, 3, std::vector<int32_t>{ 2, 3, 5, 7, 11, 13 });
    claxon::StereoSamples it = s.stereo_samples();
    std::pair<int32_t, int32_t> p;
    const std::pair<int32_t, int32_t> want[3] = { { 2, 7 }, { 3, 11 }, { 5, 13 } };
    for (int i = 0; i < 3; ++i) bad += !(it.next(&p) && p == want[i]);
    bad += it.next(&p) ? 1 : 0;
    claxon::Block e = claxon::Block::empty();
    bad += !(e.len() == 0 && e.channels() == 0 && e.duration() == 0);
    std::vector<int32_t> back = std::move(s).into_buffer();
    bad += !(back.size() == 6 && back[5] == 13);
    std::printf(bad ? "block accessors FAILED (%d)\n" : "block accessors ok\n", bad);
    return bad;
}

int main(int argc, char** argv) {
    if (argc == 3 && std::string(argv[1]) == "--limits") return verify_limits(argv[2]) ? 4 : 0;
    if (argc == 2 && std::string(argv[1]) == "--block") return verify_block() ? 5 : 0;
    for (int a = 1; a < argc; ++a) {
        std::FILE* f = std::fopen(argv[a], "rb");
        if (!f) { std::printf("%s: cannot open\n", argv[a]); return 2; }
        std::vector<uint8_t> d;
        uint8_t buf[65536]; size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        std::fclose(f);
        if (d.size() < 4) { std::printf("%s: too short\n", argv[a]); continue; }
        claxon::MetadataBlockReader reader(d.data() + 4, d.size() - 4);          // behind the `fLaC` marker (lib.rs:186-205)
        claxon::Result<claxon::MetadataBlock> r;
        int i = 0;
        while (reader.next(&r)) {
            if (r.is_err) { std::printf("%s block=%d error status=%d text=%s\n", argv[a], i, r.error.status, r.error.text.c_str()); break; }
            const claxon::MetadataBlock& m = r.value;
            std::printf("%s block=%d kind=%s length=%" PRIu32, argv[a], i, kind_name(m.kind), m.length);
            if (m.kind == claxon::MetadataBlock::Kind::StreamInfo)
                std::printf(" sample_rate=%" PRIu32 " channels=%" PRIu32 " bits_per_sample=%" PRIu32 " samples=%" PRIu64,
                            m.streaminfo.sample_rate, m.streaminfo.channels, m.streaminfo.bits_per_sample, (uint64_t)m.streaminfo.samples);
            if (m.kind == claxon::MetadataBlock::Kind::Application) std::printf(" id=%08" PRIx32 " data_len=%zu", m.id, m.data.size());
            if (m.kind == claxon::MetadataBlock::Kind::VorbisComment) std::printf(" comments=%zu vendor_len=%zu", m.comments.size(), m.vendor.size());
            std::printf("\n");
            ++i;
        }
        std::printf("%s end=%zu\n", argv[a], reader.position() + 4);
        // FrameReader::new / into_inner (frame.rs:652, 782) need no device: nothing is decoded before the first read
        claxon::FrameReader fr(nullptr, d.data(), d.size());
        const size_t at = fr.position();
        std::pair<std::vector<uint8_t>, size_t> inner = std::move(fr).into_inner();
        if (at != 0 || inner.second != 0 || inner.first != d) { std::printf("%s into_inner mismatch\n", argv[a]); return 3; }
    }
    return 0;
}
