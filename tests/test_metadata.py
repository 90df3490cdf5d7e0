"""f4: FLAC tags (VORBIS_COMMENT, metadata.rs:402-513) and FlacReaderOptions (lib.rs:123-166, 230-307).
Golden expectations from the reference's own tests (tests/testsamples.rs:320-349): repeated_vorbis_comment.flac holds
FOO=bar and FOO=baz, empty_vorbis_comment.flac holds FOO=bar, an empty comment (skipped) and X=Y.  Everything else:
oracle (C restatement) against the product's host parser on crafted blocks that hit every error path."""
import os
import struct

import numpy as np
import pytest

import claxon_amd as cx
from claxon_msgs import MSG
from conftest import FIXTURES


def fixture(name):
    return np.frombuffer(open(os.path.join(FIXTURES, name), "rb").read(), dtype=np.uint8)


def streaminfo_block(last=False):
    si = bytearray(34)
    si[0:2] = (4096).to_bytes(2, "big"); si[2:4] = (4096).to_bytes(2, "big")
    si[10:14] = ((44100 << 12) | (1 << 9) | (15 << 4)).to_bytes(4, "big")
    return bytes([0x80 if last else 0x00, 0, 0, 34]) + bytes(si)


def vc_body(vendor, comments, count=None, tail=b""):
    b = struct.pack("<I", len(vendor)) + vendor + struct.pack("<I", len(comments) if count is None else count)
    for c in comments:
        b += struct.pack("<I", len(c)) + c
    return b + tail


def block(btype, body, last=False, length=None):
    n = len(body) if length is None else length
    return bytes([(0x80 if last else 0) | btype]) + n.to_bytes(3, "big") + body


def stream(*blocks):
    return np.frombuffer(b"fLaC" + b"".join(blocks) + b"\xff\xf8", dtype=np.uint8)


def both(oracle, data, **opt):
    o = oracle.stream_open_ext(data, **opt)
    p = cx.read_stream_header_ext(data, **opt)
    assert (o[0], o[1]) == (p[0], p[1]), (o[:2], p[:2])
    if o[0] == cx.OK:
        assert o[3] == p[3] and o[4] == p[4] and o[5] == p[5]
        assert bytes(o[2].md5sum) == bytes(p[2].md5sum) and o[2].sample_rate == p[2].sample_rate
    return p


def test_limits_fixtures_match_reference_tests(oracle):
    """tests/testsamples.rs:404-426: a vendor string that would not fit in its block is a FormatError with this very
    message; a Vorbis comment block that claims 16 MB is Unsupported (no allocation is attempted).  Both files have no
    STREAMINFO in front: the block is read first, its error wins (lib.rs:244-248)."""
    st, msg, *_ = both(oracle, fixture("large_vendor_string.flac"))
    assert st == cx.FORMAT_ERROR and cx.message(msg) == "vendor string too long"                     # testsamples.rs:412
    st, msg, *_ = both(oracle, fixture("large_vorbis_comment_block.flac"))
    assert st == cx.UNSUPPORTED and cx.message(msg) == "Vorbis comment blocks larger than 10 MiB are not supported"   # testsamples.rs:423


def test_fixture_tags_match_reference_tests(oracle):
    st, msg, si, off, vendor, tags = both(oracle, fixture("repeated_vorbis_comment.flac"))
    assert st == cx.OK and (b"FOO", b"bar") in tags and (b"FOO", b"baz") in tags
    assert [v for n, v in tags if n == b"FOO"] == [b"bar", b"baz"]                   # testsamples.rs:320-328
    st, msg, si, off, vendor, tags = both(oracle, fixture("empty_vorbis_comment.flac"))
    assert st == cx.OK and tags == [(b"FOO", b"bar"), (b"X", b"Y")]                  # testsamples.rs:331-349
    for name in ("pop.flac", "short.flac", "wasted_bits.flac", "non_subset.flac"):
        st, msg, si, off, vendor, tags = both(oracle, fixture(name))
        assert st == cx.OK
    # options (lib.rs:123-166): tags not wanted -> none; metadata only -> same tags
    assert both(oracle, fixture("repeated_vorbis_comment.flac"), read_vorbis_comment=False)[4:] == (None, [])
    assert both(oracle, fixture("repeated_vorbis_comment.flac"), metadata_only=True)[5] == [(b"FOO", b"bar"), (b"FOO", b"baz")]
    both(oracle, fixture("repeated_vorbis_comment.flac"), metadata_only=True, read_vorbis_comment=False)


def test_vorbis_comment_error_paths(oracle):
    si = streaminfo_block()
    ok = vc_body(b"vendor \xc3\xa9", [b"ARTIST=Queen", b"artist=Bowie", b"Title=Under Pressure \xe2\x99\xab", b"EMPTY="])
    cases = [
        (stream(si, block(4, ok, last=True)), cx.OK, "NONE"),
        (stream(si, block(4, b"\0" * 7, last=True)), cx.FORMAT_ERROR, "VC_TOO_SHORT"),
        (stream(si, block(4, b"", last=True, length=10 * 1024 * 1024 + 1)), cx.UNSUPPORTED, "VC_TOO_LARGE"),
        (stream(si, block(4, struct.pack("<I", 100) + b"abcd" + b"\0" * 4, last=True)), cx.FORMAT_ERROR, "VC_VENDOR_TOO_LONG"),
        (stream(si, block(4, vc_body(b"v", [], count=3), last=True)), cx.FORMAT_ERROR, "VC_TOO_MANY_ENTRIES"),
        (stream(si, block(4, vc_body(b"v", [], count=1, tail=struct.pack("<I", 99) + b"A=b" + b"\0" * 9), last=True)), cx.FORMAT_ERROR, "VC_COMMENT_TOO_LONG"),
        (stream(si, block(4, vc_body(b"v", [b"A\x1f=b", b"C=d", b"E=f", b"G=h"]), last=True)), cx.FORMAT_ERROR, "VC_NAME_INVALID_BYTE"),
        (stream(si, block(4, vc_body(b"v", [b"A~=b", b"C=d", b"E=f", b"G=h"]), last=True)), cx.FORMAT_ERROR, "VC_NAME_INVALID_BYTE"),
        (stream(si, block(4, vc_body(b"v", [b"noequals", b"C=d", b"E=f", b"G=h"]), last=True)), cx.FORMAT_ERROR, "VC_NO_EQUALS"),
        (stream(si, block(4, vc_body(b"v", [b"A=b", b"C=d", b"E=f"], count=2), last=True)), cx.FORMAT_ERROR, "VC_EXCESS_DATA"),
        (stream(si, block(4, vc_body(b"v", [b"A=b"], count=2, tail=b"\0\0"), last=True)), cx.FORMAT_ERROR, "VC_EXCESS_DATA"),
        (stream(si, block(4, vc_body(b"vendor", [b"A=b", b"C=d"], count=4, tail=b""), last=True)), cx.FORMAT_ERROR, "VC_WRONG_COUNT"),
        (stream(si, block(4, vc_body(b"\xff\xfe", [b"A=b", b"C=d", b"E=f"]), last=True)), cx.FORMAT_ERROR, "VC_NOT_UTF8"),
        (stream(si, block(4, vc_body(b"v", [b"A=\xc0\xaf", b"C=d", b"E=f", b"G=h"]), last=True)), cx.FORMAT_ERROR, "VC_NOT_UTF8"),     # overlong
        (stream(si, block(4, vc_body(b"v", [b"A=\xed\xa0\x80", b"C=d", b"E=f", b"G=h"]), last=True)), cx.FORMAT_ERROR, "VC_NOT_UTF8"),  # surrogate
        (stream(si, block(4, vc_body(b"v", [b"A=\xf4\x90\x80\x80", b"C=d", b"E=f"]), last=True)), cx.FORMAT_ERROR, "VC_NOT_UTF8"),      # > U+10FFFF
        (stream(si, block(4, vc_body(b"v", [b"A=\xe2\x82", b"C=d", b"E=f", b"G=h"]), last=True)), cx.FORMAT_ERROR, "VC_NOT_UTF8"),      # truncated
        (stream(si, block(4, ok), block(4, ok, last=True)), cx.FORMAT_ERROR, "SECOND_VORBIS_COMMENT"),
        (stream(block(4, ok), si), cx.FORMAT_ERROR, "STREAMINFO_MISSING"),
        (stream(si, block(4, ok, last=True, length=len(ok) + 40)), cx.FORMAT_ERROR, "VC_EXCESS_DATA"),
        (stream(si, block(4, ok, last=True))[:-12], cx.IO_ERROR, "UNEXPECTED_EOF"),                             # the stream ends inside a comment
        (stream(si, block(4, ok, last=True))[:4 + 38 + 4 + 6], cx.IO_ERROR, "UNEXPECTED_EOF"),                  # ... inside the vendor string
        (stream(si, block(4, vc_body(b"v", [b"", b"A=b", b"", b"C=d"]) , last=True)), cx.OK, "NONE"),                 # empty comments are skipped
        (stream(si, block(1, b"\0" * 10), block(4, ok), block(3, b"\1" * 18), block(6, b"pic", last=True)), cx.OK, "NONE"),
    ]
    for i, (data, st, name) in enumerate(cases):
        got = both(oracle, data)
        assert (got[0], got[1]) == (st, MSG["CLX_MSG_" + name]), (i, name, got[:2], cx.message(got[1]))
    st, msg, si_, off, vendor, tags = both(oracle, cases[0][0])
    assert vendor == "vendor é".encode() and tags[2] == (b"Title", "Under Pressure ♫".encode()) and tags[3] == (b"EMPTY", b"")
    # early-out (lib.rs:273-277): metadata only and no tags wanted -> the block AFTER the streaminfo is still read, then stop:
    # the broken Vorbis comment behind the padding is never looked at
    bad = stream(si, block(1, b"\0" * 4), block(4, b"\0" * 7, last=True))
    assert both(oracle, bad)[:2] == (cx.FORMAT_ERROR, MSG["CLX_MSG_VC_TOO_SHORT"])
    assert both(oracle, bad, metadata_only=True, read_vorbis_comment=False)[0] == cx.OK
    assert both(oracle, bad, metadata_only=True)[0] == cx.FORMAT_ERROR
    bad2 = stream(si, block(4, b"\0" * 7, last=True))
    assert both(oracle, bad2, metadata_only=True, read_vorbis_comment=False)[0] == cx.FORMAT_ERROR


def test_fuzz_corpus_open_parity(oracle):
    import glob
    files = sorted(glob.glob(os.path.join(FIXTURES, "fuzz", "*.flac")))
    assert len(files) == 23
    for p in files:
        both(oracle, np.frombuffer(open(p, "rb").read(), dtype=np.uint8))


@pytest.mark.gpu
def test_gpu_reader_tags():
    ctx = cx.Context(0, wait_s=120)
    rd = cx.FlacReader.open(ctx, os.path.join(FIXTURES, "repeated_vorbis_comment.flac"))
    assert rd.get_tag("FOO") == ["bar", "baz"] and rd.get_tag("foo") == ["bar", "baz"] and rd.get_tag("foobar") == []
    assert rd.vendor() is not None and ("FOO", "bar") in rd.tags()
    assert len(list(rd.blocks())) > 0
    rd = cx.FlacReader.open(ctx, os.path.join(FIXTURES, "empty_vorbis_comment.flac"))
    assert rd.tags() == [("FOO", "bar"), ("X", "Y")]
    rd = cx.FlacReader.open(ctx, os.path.join(FIXTURES, "short.flac"))
    assert isinstance(rd.tags(), list)
