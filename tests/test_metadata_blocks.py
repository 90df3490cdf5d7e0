"""f4: single metadata blocks for streams embedded in containers -- metadata::read_metadata_block /
read_metadata_block_with_header / MetadataBlockReader (metadata.rs:244-319, 553-603) -- and container packets -> frame
descriptors (what examples/decode_ogg.rs and decode_mp4.rs do per packet).  The product's host parser against the oracle
(C restatement) on every block of the reference's fixtures, on every truncation of them, on crafted blocks of every type
and on random headers; the Ogg mapping's first packet as decode_ogg.rs:65-103 reads it."""
import os
import struct

import numpy as np
import pytest

import claxon_amd as cx
from conftest import FIXTURES
from test_metadata import block, streaminfo_block, vc_body

NAMES = ("pop.flac", "short.flac", "wasted_bits.flac", "non_subset.flac", "repeated_vorbis_comment.flac", "empty_vorbis_comment.flac")


def same(o, p):
    assert (o["status"], o["msg"]) == (p["status"], p["msg"]), (o, p)
    if o["status"] != cx.OK:
        return
    for key in ("kind", "length", "consumed", "is_last", "app_id", "app_data", "vendor", "tags"):
        assert o.get(key) == p.get(key), key
    if o["kind"] == cx.BLOCK_STREAMINFO:
        a, b = o["streaminfo"], p["streaminfo"]
        for f in ("min_block_size", "max_block_size", "min_frame_size", "max_frame_size", "sample_rate", "channels", "bits_per_sample", "samples"):
            assert getattr(a, f) == getattr(b, f), f
        assert bytes(a.md5sum) == bytes(b.md5sum)


def both(oracle, data, *a):
    o, p = oracle.read_metadata_block(data, *a), cx.read_metadata_block(data, *a)
    same(o, p)
    return p


def walk_blocks(oracle, data):
    """MetadataBlockReader (metadata.rs:553-603): blocks with header until the last-block flag."""
    pos, out = 4, []
    while True:
        b = both(oracle, data[pos:])
        out.append(b)
        if b["status"] != cx.OK or b["is_last"]:
            return out, pos + b.get("consumed", 0)
        pos += b["consumed"]


@pytest.mark.parametrize("name", NAMES)
def test_fixture_blocks(oracle, name):
    data = open(os.path.join(FIXTURES, name), "rb").read()
    blocks, end = walk_blocks(oracle, data)
    assert all(b["status"] == cx.OK for b in blocks) and blocks[0]["kind"] == cx.BLOCK_STREAMINFO
    st, msg, si, off, vendor, tags = cx.read_stream_header_ext(data)          # FlacReader::new_ext walks the same blocks
    assert st == cx.OK and off == end
    assert bytes(blocks[0]["streaminfo"].md5sum) == bytes(si.md5sum) and blocks[0]["streaminfo"].samples == si.samples
    vcs = [b for b in blocks if b["kind"] == cx.BLOCK_VORBIS_COMMENT]
    assert [(b["vendor"], b["tags"]) for b in vcs] == ([(vendor, tags)] if vendor is not None else [])
    # every truncation of the metadata section fails the same way on both sides
    for cut in range(4, end, max(1, (end - 4) // 97)):
        walk_blocks(oracle, data[:cut])


def test_every_block_type(oracle):
    si = streaminfo_block()[4:]
    vc = vc_body(b"claxon", [b"TITLE=x", b"ARTIST=\xc3\xa9"])
    for btype in list(range(0, 9)) + [64, 126, 127]:
        for body in (si, vc, b"", b"\x01\x02\x03", b"ABCD", b"ABCDpayload", bytes(40)):
            for length in {len(body), 0, 3, 4, 34, len(body) + 1}:
                p = both(oracle, body, btype, length)
                if p["status"] == cx.OK:
                    assert p["consumed"] == length
    # what the reference's variants carry
    p = both(oracle, b"ABCDpayload", 2, 11)
    assert (p["kind"], p["app_id"], p["app_data"]) == (cx.BLOCK_APPLICATION, 0x41424344, b"payload")
    for t in (1, 3, 5, 6):                              # seek table, cue sheet, picture: "pretend it is padding" (metadata.rs:287-305)
        assert both(oracle, bytes(10), t, 10)["kind"] == cx.BLOCK_PADDING
    assert both(oracle, bytes(10), 9, 10)["kind"] == cx.BLOCK_RESERVED
    p = both(oracle, bytes(10), 127, 10)
    assert cx.message(p["msg"]) == "invalid metadata block type"
    p = both(oracle, si, 0, 33)
    assert cx.message(p["msg"]) == "invalid streaminfo metadata block length"
    p = both(oracle, b"ABC", 2, 3)
    assert cx.message(p["msg"]) == "application block length must be at least 4 bytes"
    p = both(oracle, b"ABCD", 2, 10 * 1024 * 1024 + 1)
    assert p["status"] == cx.UNSUPPORTED and cx.message(p["msg"]) == "application blocks larger than 10 MiB are not supported"
    assert both(oracle, bytes(5), 1, 6)["status"] == cx.IO_ERROR                       # skip past the end (input.rs:269-277)
    p = both(oracle, vc, 4, len(vc))
    assert p["vendor"] == b"claxon" and p["tags"] == [(b"TITLE", b"x"), (b"ARTIST", b"\xc3\xa9")]


def test_random_headers(oracle):
    rng = np.random.default_rng(20260926)
    base = streaminfo_block()[4:] + vc_body(b"v", [b"A=b"]) + bytes(64)
    for _ in range(3000):
        body = bytearray(base[:int(rng.integers(0, len(base)))])
        for _ in range(int(rng.integers(0, 4))):
            if body:
                body[int(rng.integers(0, len(body)))] = int(rng.integers(0, 256))
        hdr = bytes([int(rng.integers(0, 256))]) + int(rng.choice([0, 3, 4, 8, 34, len(body), int(rng.integers(0, 200))])).to_bytes(3, "big")
        both(oracle, hdr + bytes(body))
        both(oracle, (hdr + bytes(body))[:int(rng.integers(0, 8))])


def test_ogg_mapping_first_packet(oracle):
    """examples/decode_ogg.rs:65-103: 7 bytes of magic and version, the big-endian count of header packets, `fLaC`, then
    the streaminfo block with its header; the packets that follow hold one metadata block each (decode_ogg.rs:36-41)."""
    data = open(os.path.join(FIXTURES, "repeated_vorbis_comment.flac"), "rb").read()
    blocks, end = walk_blocks(oracle, data)
    raw, pos = [], 4
    for b in blocks:
        raw.append(data[pos:pos + b["consumed"]]); pos += b["consumed"]
    first = b"\x7fFLAC\x01\x00" + struct.pack(">H", len(raw) - 1) + b"fLaC" + raw[0]
    assert struct.unpack(">H", first[7:9])[0] == len(raw) - 1
    p = both(oracle, first[13:])
    st, msg, si, off, vendor, tags = cx.read_stream_header_ext(data)
    assert p["kind"] == cx.BLOCK_STREAMINFO and bytes(p["streaminfo"].md5sum) == bytes(si.md5sum)
    assert p["streaminfo"].sample_rate == si.sample_rate and p["streaminfo"].channels == si.channels
    got = [both(oracle, pkt) for pkt in raw[1:]]
    assert [g["kind"] for g in got] == [b["kind"] for b in blocks[1:]]
    assert [g["tags"] for g in got if g["kind"] == cx.BLOCK_VORBIS_COMMENT] == [tags]


def test_packets_to_descriptors(oracle):
    """One frame per packet (decode_ogg.rs:105-114): descriptors from clx_describe_packets equal the ones the stream
    indexer finds in the same bytes, packet lengths become max_bytes, bad packets report what the frame reader would."""
    import synth
    w = synth.small_mixed(24)                        # frames of every subframe type / bit depth / channel count, one per packet
    data = w.arena[:w.arena_len]
    offs, lens = np.asarray(w.offs, dtype=np.uint64), np.asarray(w.lens, dtype=np.uint32)
    descs, hdrs = cx.descs_from_offsets(data, offs, lens)
    d2, h2, res = cx.describe_packets(data, offs, lens)
    assert np.all(res["status"] == cx.OK)
    for f in ("byte_off", "max_bytes", "header_bytes", "block_size", "n_channels", "channel_assignment", "bps"):
        assert np.array_equal(d2[f], descs[f]), f
    assert np.array_equal(d2["max_bytes"], lens) and np.array_equal(h2, hdrs)
    # the oracle's frame reader over each packet on its own (FrameReader::new(Cursor(packet)).read_next_or_eof) agrees
    for i in range(offs.size):
        info, samples = oracle.frame_decode(data[int(offs[i]):int(offs[i]) + int(lens[i])], True)
        assert info.status == 0 and info.block_size == d2["block_size"][i] and info.channels == d2["n_channels"][i]
    # an empty packet, a one-byte packet, a packet that does not start with the sync code, a damaged header
    bad = np.concatenate([data[int(offs[0]):int(offs[0]) + 16], np.array([0x12, 0x34, 0, 0], dtype=np.uint8)]).copy()
    bad[3] ^= 0x10
    d3, h3, r3 = cx.describe_packets(bad, [0, 0, 16, 0], [0, 1, 4, 16])
    assert list(r3["status"][:2]) == [cx.END_OF_STREAM, cx.END_OF_STREAM]
    assert cx.message(int(r3["msg"][2])) == "frame sync code missing" or r3["status"][2] != cx.OK
    assert r3["status"][3] == cx.FORMAT_ERROR
    with pytest.raises(cx.ClaxonError):
        cx.describe_packets(bad, [10], [100])


def test_cpp_metadata_block_reader(oracle):
    """claxon.hpp's MetadataBlockReader / read_metadata_block_with_header (the C++ mirror of metadata.rs:244-319, 553-603)
    walks the fixtures' metadata exactly as the C ABI and the oracle do; a stream cut inside a block ends with the
    reference's IoError."""
    import subprocess
    import sys
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    cx.build()
    exe = g.build_cpp_metadata_test()
    kinds = {cx.BLOCK_STREAMINFO: "StreamInfo", cx.BLOCK_PADDING: "Padding", cx.BLOCK_APPLICATION: "Application",
             cx.BLOCK_VORBIS_COMMENT: "VorbisComment", cx.BLOCK_RESERVED: "Reserved"}
    with tempfile.TemporaryDirectory() as tmp:
        paths = [os.path.join(FIXTURES, n) for n in NAMES]
        # a crafted stream with an application block, padding, a picture (read as padding) and a reserved type
        crafted = b"fLaC" + streaminfo_block() + block(2, b"ABCDxyz") + block(1, bytes(5)) + block(6, bytes(9)) + block(9, b"??", last=True)
        cut = open(paths[4], "rb").read()[:60]
        for name, blob in (("crafted.flac", crafted), ("cut.flac", cut)):
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(blob)
            paths.append(os.path.join(tmp, name))
        r = subprocess.run([exe] + paths, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.splitlines()
        for path in paths:
            data = open(path, "rb").read()
            mine = [l[len(path) + 1:] for l in lines if l.startswith(path + " ")]
            blocks, end = walk_blocks(oracle, data)
            want = []
            for i, b in enumerate(blocks):
                if b["status"] != cx.OK:
                    want.append("block=%d error status=%d text=%s" % (i, b["status"], cx.message(b["msg"])))
                    break
                line = "block=%d kind=%s length=%d" % (i, kinds[b["kind"]], b["length"])
                if b["kind"] == cx.BLOCK_STREAMINFO:
                    si = b["streaminfo"]
                    line += " sample_rate=%d channels=%d bits_per_sample=%d samples=%d" % (si.sample_rate, si.channels, si.bits_per_sample, si.samples)
                if b["kind"] == cx.BLOCK_APPLICATION:
                    line += " id=%08x data_len=%d" % (b["app_id"], len(b["app_data"]))
                if b["kind"] == cx.BLOCK_VORBIS_COMMENT:
                    line += " comments=%d vendor_len=%d" % (len(b["tags"]), len(b["vendor"]))
                want.append(line)
            assert mine[:-1] == want, (path, mine, want)
            assert mine[-1].startswith("end=")
            if all(b["status"] == cx.OK for b in blocks):
                assert mine[-1] == "end=%d" % end
        assert any("kind=Application" in l and "id=41424344 data_len=3" in l for l in lines)
        assert any("cut.flac" in l and "error status=%d" % cx.IO_ERROR in l for l in lines)


def test_cpp_reader_limits_match_reference_tests():
    """verify_limits_on_vendor_string / verify_limits_on_vorbis_comment_block (tests/testsamples.rs:404-426) against the
    C++ mirror's FlacReader::open -- the files fail in the metadata, so the check runs without a device."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    cx.build()
    exe = g.build_cpp_metadata_test()
    r = subprocess.run([exe, "--limits", FIXTURES], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verify_limits_on_vendor_string ok (vendor string too long)" in r.stdout
    assert "verify_limits_on_vorbis_comment_block ok (Vorbis comment blocks larger than 10 MiB are not supported)" in r.stdout
