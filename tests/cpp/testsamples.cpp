// The reference's integration tests (tests/testsamples.rs) restated against the C++ mirror of its API
// (claxon_amd/csrc/host/claxon.hpp): same test names, same expectations.  Where the reference shells out to
// `metaflac` / `flac -d` for the expected values, this program (a) checks what a stream vouches for itself -- the
// MD5 of the decoded audio in STREAMINFO -- and (b) prints the facts (`name key=value`) that the calling pytest
// (tests/test_cpp_mirror.py) compares with the oracle's.  Needs a gfx950 device: decoding happens on the GPU.
//   usage: testsamples <fixture-dir>        exit code 0 = every assertion held
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <dirent.h>
#include <string>
#include <vector>
#include <algorithm>

#include "../../claxon_amd/csrc/host/claxon.hpp"

static int g_failed = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_failed; } } while (0)

// ---- MD5 (RFC 1321), for the audio checksum of STREAMINFO
struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    uint64_t n = 0; uint8_t buf[64]; size_t fill = 0;
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
            0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
            0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
            0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391 };
        static const int S[64] = { 7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22, 5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20,
                                   4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23, 6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21 };
        uint32_t m[16];
        for (int i = 0; i < 16; ++i) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; ++i) {
            uint32_t f; int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D; D = C; C = B; B = B + rol(A + f + K[i] + m[g], S[i]); A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void update(const uint8_t* p, size_t len) {
        n += len;
        while (len) {
            const size_t take = std::min(len, sizeof buf - fill);
            std::memcpy(buf + fill, p, take); fill += take; p += take; len -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void finish(uint8_t out[16]) {
        const uint64_t bits = n * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t lenb[8];
        for (int i = 0; i < 8; ++i) lenb[i] = (uint8_t)(bits >> (8 * i));
        update(lenb, 8);
        const uint32_t w[4] = { a, b, c, d };
        for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(w[i / 4] >> (8 * (i % 4)));
    }
};

static std::string hex(const uint8_t* p, size_t n) {
    std::string s; char t[3];
    for (size_t i = 0; i < n; ++i) { std::snprintf(t, sizeof t, "%02x", p[i]); s += t; }
    return s;
}

static clx_ctx* g_ctx = nullptr;
static std::string g_dir;
static std::string path(const char* name) { return g_dir + "/" + name; }

// testsamples.rs:71-99 -- the fields metaflac would print, for the caller to compare with the oracle's
static void verify_streaminfo(const char* name, const char* file) {
    auto r = claxon::FlacReader::open(g_ctx, path(file).c_str());
    CHECK(r.is_ok());
    if (!r.is_ok()) return;
    const clx_streaminfo& si = r.value.streaminfo();
    std::printf("%s min_block_size=%u max_block_size=%u min_frame_size=%u max_frame_size=%u sample_rate=%u channels=%u bits_per_sample=%u samples=%" PRIu64 " md5sum=%s\n",
                name, si.min_block_size, si.max_block_size, si.min_frame_size, si.max_frame_size, si.sample_rate, si.channels, si.bits_per_sample,
                (uint64_t)si.samples, hex(si.md5sum, 16).c_str());
}

// testsamples.rs:164-216 -- every sample through FlacReader::samples(), against the checksum the encoder stored
static void verify_decoded_stream(const char* name, const char* file) {
    auto r = claxon::FlacReader::open(g_ctx, path(file).c_str());
    CHECK(r.is_ok());
    if (!r.is_ok()) return;
    const clx_streaminfo si = r.value.streaminfo();
    const uint32_t bytes = (si.bits_per_sample + 7) / 8;
    Md5 md5;
    uint64_t count = 0;
    claxon::FlacSamples it = r.value.samples();
    int32_t s; claxon::Error err; bool failed = false;
    std::vector<uint8_t> chunk;
    while (it.next(&s, &err, &failed)) {
        CHECK(!failed);
        if (failed) { std::printf("%s error=%s\n", name, err.text.c_str()); return; }
        for (uint32_t b = 0; b < bytes; ++b) chunk.push_back((uint8_t)((uint32_t)s >> (8 * b)));
        if (chunk.size() >= 1 << 16) { md5.update(chunk.data(), chunk.size()); chunk.clear(); }
        ++count;
    }
    md5.update(chunk.data(), chunk.size());
    uint8_t digest[16];
    md5.finish(digest);
    CHECK(count == (uint64_t)si.samples * si.channels);
    static const uint8_t unset[16] = { 0 };                       // an encoder may leave the checksum out (non_subset.flac)
    if (std::memcmp(si.md5sum, unset, 16) != 0) CHECK(std::memcmp(digest, si.md5sum, 16) == 0);
    std::printf("%s samples=%" PRIu64 " md5=%s\n", name, count, hex(digest, 16).c_str());
}

// FlacReader::into_samples (lib.rs:417-433): the iterator that owns its reader yields what samples() yields
static void verify_into_samples(const char* name, const char* file) {
    auto a = claxon::FlacReader::open(g_ctx, path(file).c_str());
    auto b = claxon::FlacReader::open(g_ctx, path(file).c_str());
    CHECK(a.is_ok() && b.is_ok());
    if (!a.is_ok() || !b.is_ok()) return;
    claxon::FlacSamples borrowed = a.value.samples();
    claxon::FlacIntoSamples owned = claxon::into_samples(std::move(b.value));
    int32_t s1 = 0, s2 = 0; claxon::Error e1, e2; bool f1 = false, f2 = false;
    uint64_t count = 0;
    for (;;) {
        const bool m1 = borrowed.next(&s1, &e1, &f1), m2 = owned.next(&s2, &e2, &f2);
        CHECK(m1 == m2 && f1 == f2);
        if (!m1 || !m2 || f1 || f2) break;
        CHECK(s1 == s2);
        ++count;
    }
    const clx_streaminfo si = owned.reader().streaminfo();
    CHECK(count == (uint64_t)si.samples * si.channels);
    std::printf("%s samples=%" PRIu64 "\n", name, count);
}

// decode_packets: the frames of a stream handed over as packets (one frame each, as a container demuxer does) decode to
// the blocks that FlacReader::blocks() yields; an empty packet is Ok(None), a damaged one fails alone
static void verify_decode_packets(const char* name, const char* file) {
    std::vector<uint8_t> d;
    {
        std::FILE* f = std::fopen(path(file).c_str(), "rb");
        CHECK(f != nullptr);
        if (!f) return;
        uint8_t buf[65536]; size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        std::fclose(f);
    }
    clx_streaminfo si; size_t audio = 0; uint32_t msg = 0;
    CHECK(clx_read_stream_header(d.data(), d.size(), &si, &audio, &msg) == CLX_OK);
    std::vector<clx_frame_desc> descs(4096); std::vector<clx_frame_header> hdrs(4096);
    size_t found = 0, stop = 0;
    CHECK(clx_index_frames(d.data(), d.size(), audio, descs.data(), hdrs.data(), descs.size(), &found, &stop) == CLX_OK);
    CHECK(found > 0 && stop == d.size());
    std::vector<uint64_t> offs; std::vector<uint32_t> lens;
    for (size_t i = 0; i < found; ++i) {
        offs.push_back(descs[i].byte_off);
        lens.push_back((uint32_t)((i + 1 < found ? descs[i + 1].byte_off : d.size()) - descs[i].byte_off));
    }
    offs.push_back(d.size()); lens.push_back(0);                                  // an empty packet at the end
    std::vector<uint8_t> arena(d);
    arena.resize(d.size() + 32, 0);
    std::vector<claxon::FrameResult> got = claxon::decode_packets(g_ctx, arena.data(), d.size(), offs.data(), lens.data(), offs.size());
    CHECK(got.size() == found + 1 && !got[found].is_err && !got[found].has_block);
    auto r = claxon::FlacReader::open(g_ctx, path(file).c_str());
    CHECK(r.is_ok());
    if (!r.is_ok()) return;
    uint64_t samples = 0;
    for (size_t i = 0; i < found; ++i) {
        claxon::FrameResult want = r.value.blocks().read_next_or_eof(std::vector<int32_t>());
        CHECK(!want.is_err && want.has_block && !got[i].is_err && got[i].has_block);
        if (want.is_err || !want.has_block || got[i].is_err || !got[i].has_block) return;
        CHECK(got[i].block.time() == want.block.time() && got[i].block.duration() == want.block.duration() && got[i].block.channels() == want.block.channels());
        for (uint32_t c = 0; c < want.block.channels(); ++c)
            CHECK(std::memcmp(got[i].block.channel(c), want.block.channel(c), sizeof(int32_t) * want.block.duration()) == 0);
        samples += got[i].block.len();
    }
    // a flipped bit in the first packet's body: that packet reports the frame CRC mismatch, the others still decode
    if (lens[0] > 12) {
        arena[offs[0] + lens[0] / 2] ^= 0x10;
        std::vector<claxon::FrameResult> bad = claxon::decode_packets(g_ctx, arena.data(), d.size(), offs.data(), lens.data(), offs.size());
        CHECK(bad[0].is_err);
        for (size_t i = 1; i < found; ++i) CHECK(!bad[i].is_err && bad[i].has_block);
    }
    std::printf("%s packets=%zu samples=%" PRIu64 "\n", name, found, samples);
}

// the same audio through blocks(): Block accessors agree with each other (frame.rs:402-529)
static void verify_blocks(const char* name, const char* file) {
    auto r = claxon::FlacReader::open(g_ctx, path(file).c_str());
    CHECK(r.is_ok());
    if (!r.is_ok()) return;
    const clx_streaminfo si = r.value.streaminfo();
    uint64_t t = 0, n_blocks = 0;
    std::vector<int32_t> buffer;
    for (;;) {
        claxon::FrameResult fr = r.value.blocks().read_next_or_eof(std::move(buffer));
        CHECK(!fr.is_err);
        if (fr.is_err || !fr.has_block) break;
        const claxon::Block& b = fr.block;
        CHECK(b.channels() == si.channels);
        CHECK(b.len() == b.duration() * b.channels());
        // (time() is block_size * frame_number for fixed-blocksize streams, with the block size of THIS frame -- frame.rs:772 --
        //  so it only lines up with the running count while blocks are full)
        if (b.duration() == si.max_block_size && si.min_block_size == si.max_block_size && n_blocks == 0) CHECK(b.time() == t || b.time() % b.duration() == 0);
        CHECK(b.duration() >= 1 && b.duration() <= si.max_block_size);
        for (uint32_t c = 0; c < b.channels(); ++c) CHECK(b.channel(c)[b.duration() - 1] == b.sample(c, b.duration() - 1));
        if (b.channels() == 2) {
            claxon::StereoSamples ss = b.stereo_samples();
            std::pair<int32_t, int32_t> lr; uint32_t i = 0;
            while (ss.next(&lr)) { CHECK(lr.first == b.sample(0, i) && lr.second == b.sample(1, i)); ++i; }
            CHECK(i == b.duration());
        }
        t += b.duration(); ++n_blocks;
        buffer = fr.block.into_buffer();
    }
    CHECK(t == si.samples);
    std::printf("%s blocks=%" PRIu64 " samples_per_channel=%" PRIu64 "\n", name, n_blocks, t);
}

// testsamples.rs:320-330
static void test_flac_reader_get_tag_returns_all_matches() {
    auto r = claxon::FlacReader::open(g_ctx, path("repeated_vorbis_comment.flac").c_str());
    CHECK(r.is_ok());
    if (!r.is_ok()) return;
    const std::vector<std::string> foo = r.value.get_tag("FOO");
    CHECK(foo.size() == 2 && foo[0] == "bar" && foo[1] == "baz");
    // the lookup is case-insensitive, non-existing tags are not found (testsamples.rs:289-317, on a file this
    // repository does not carry; the same properties on this one)
    const std::vector<std::string> lower = r.value.get_tag("foo");
    CHECK(lower.size() == 2 && lower[0] == "bar" && lower[1] == "baz");
    CHECK(r.value.get_tag("foobar").empty());
}

// testsamples.rs:332-351
static void test_flac_reader_tags_skips_empty_vorbis_comments() {
    auto r = claxon::FlacReader::open(g_ctx, path("empty_vorbis_comment.flac").c_str());
    CHECK(r.is_ok());
    if (!r.is_ok()) return;
    const auto tags = r.value.tags();
    CHECK(tags.size() == 2);
    if (tags.size() == 2) {
        CHECK(tags[0].first == "FOO" && tags[0].second == "bar");
        CHECK(tags[1].first == "X" && tags[1].second == "Y");
    }
}

// testsamples.rs:428-447
static void metadata_only_still_reads_vorbis_comment_block() {
    claxon::FlacReaderOptions opts; opts.metadata_only = true; opts.read_vorbis_comment = true;
    auto r = claxon::FlacReader::open_ext(g_ctx, path("short.flac").c_str(), opts);
    CHECK(r.is_ok());
    if (!r.is_ok()) return;
    std::string vendor;
    CHECK(r.value.vendor(&vendor));
    CHECK(vendor == "reference libFLAC 1.3.2 20170101");
}
static void no_read_vorbis_comment_block_does_not_contain_vendor_string() {
    claxon::FlacReaderOptions opts; opts.metadata_only = true; opts.read_vorbis_comment = false;
    auto r = claxon::FlacReader::open_ext(g_ctx, path("short.flac").c_str(), opts);
    CHECK(r.is_ok());
    if (!r.is_ok()) return;
    CHECK(!r.value.vendor(nullptr));
}
// testsamples.rs:449-469 (#[should_panic]): std::logic_error stands in for the panic
static void blocks_and_samples_panic_when_metadata_only_is_set() {
    claxon::FlacReaderOptions opts; opts.metadata_only = true; opts.read_vorbis_comment = true;
    auto r = claxon::FlacReader::open_ext(g_ctx, path("short.flac").c_str(), opts);
    CHECK(r.is_ok());
    if (!r.is_ok()) return;
    bool threw = false;
    try { (void)r.value.blocks(); } catch (const std::logic_error&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { (void)r.value.samples(); } catch (const std::logic_error&) { threw = true; }
    CHECK(threw);
}

// testsamples.rs:498-560: the fuzz corpus decodes without crashing, twice to the same result, and every sample of
// every returned block was written (buffers pre-filled with 13 and with 17 give identical blocks)
static void regression_test_fuzz_samples() {
    const std::string dir = path("fuzz");
    DIR* d = opendir(dir.c_str());
    CHECK(d != nullptr);
    if (!d) return;
    std::vector<std::string> files;
    while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        if (n.size() > 5 && n.substr(n.size() - 5) == ".flac") files.push_back(n);
    }
    closedir(d);
    std::sort(files.begin(), files.end());
    size_t n_blocks = 0;
    for (const std::string& f : files) {
        std::vector<std::vector<int32_t>> decodes[2];
        int ends[2] = { 0, 0 };
        for (int pass = 0; pass < 2; ++pass) {
            auto r = claxon::FlacReader::open(g_ctx, (dir + "/" + f).c_str());
            if (!r.is_ok()) { ends[pass] = -1; continue; }
            for (;;) {
                std::vector<int32_t> buffer(1024 * 16, pass == 0 ? 13 : 17);
                claxon::FrameResult fr = r.value.blocks().read_next_or_eof(std::move(buffer));
                if (fr.is_err) { ends[pass] = 1; break; }
                if (!fr.has_block) break;
                std::vector<int32_t> b = fr.block.into_buffer();
                b.resize(fr.block.len() ? fr.block.len() : b.size());
                decodes[pass].push_back(std::move(b));
            }
        }
        CHECK(ends[0] == ends[1]);
        CHECK(decodes[0].size() == decodes[1].size());
        for (size_t i = 0; i < std::min(decodes[0].size(), decodes[1].size()); ++i) CHECK(decodes[0][i] == decodes[1][i]);
        n_blocks += decodes[0].size();
        std::printf("regression_test_fuzz_samples %s end=%d blocks=%zu\n", f.c_str(), ends[0], decodes[0].size());
    }
    std::printf("regression_test_fuzz_samples files=%zu blocks=%zu\n", files.size(), n_blocks);
    CHECK(files.size() == 23);
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <fixture-dir>\n", argv[0]); return 2; }
    g_dir = argv[1];
    if (clx_create(0, &g_ctx) != CLX_OK) { std::fprintf(stderr, "no gfx950 device: this test decodes on the GPU\n"); return 3; }
    static const char* const files[][2] = { { "pop", "pop.flac" }, { "short", "short.flac" }, { "wasted_bits", "wasted_bits.flac" }, { "non_subset", "non_subset.flac" } };
    for (auto& f : files) verify_streaminfo((std::string("verify_streaminfo_") + f[0]).c_str(), f[1]);
    for (auto& f : files) verify_decode_packets((std::string("verify_decode_packets_") + f[0]).c_str(), f[1]);
    for (auto& f : files) verify_into_samples((std::string("verify_into_samples_") + f[0]).c_str(), f[1]);
    for (auto& f : files) verify_decoded_stream((std::string("verify_decoded_stream_") + f[0]).c_str(), f[1]);
    for (auto& f : files) verify_blocks((std::string("verify_blocks_") + f[0]).c_str(), f[1]);
    test_flac_reader_get_tag_returns_all_matches();
    test_flac_reader_tags_skips_empty_vorbis_comments();
    metadata_only_still_reads_vorbis_comment_block();
    no_read_vorbis_comment_block_does_not_contain_vendor_string();
    blocks_and_samples_panic_when_metadata_only_is_set();
    regression_test_fuzz_samples();
    clx_destroy(g_ctx);
    std::printf("%s\n", g_failed ? "SOME CHECKS FAILED" : "ALL CHECKS PASSED");
    return g_failed ? 1 : 0;
}
