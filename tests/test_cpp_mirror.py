"""The C++ mirror of the reference's API (claxon_amd/csrc/host/claxon.hpp: FlacReader, FrameReader, Block, FlacSamples,
FlacReaderOptions) exercised by a C++ program that restates the reference's integration tests (tests/cpp/testsamples.cpp
<- tests/testsamples.rs).  The program checks what is self-evident (tags, vendor strings, the audio MD5 stored in
STREAMINFO, block accessors); the facts it prints are compared with the oracle's here."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import FIXTURES  # noqa: E402


def _exe():
    import __graft_entry__ as g
    import claxon_amd
    claxon_amd.build()
    return g.build_cpp_tests()


def test_cpp_mirror_builds_and_refuses_to_run_without_a_gpu():
    """Header + library are enough to build against (no torch, no python); without a device the program stops at
    clx_create: there is no CPU decode path to fall back to."""
    exe = _exe()
    assert os.path.exists(exe)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, FIXTURES], capture_output=True, text=True, timeout=120)
        assert r.returncode == 3 and "no gfx950 device" in r.stderr


def _oracle_md5(oracle, data, info):
    import hashlib
    import numpy as np
    si, blocks, st, _ = oracle.decode_stream(data)
    assert st == 0
    nbytes = (int(info.bits_per_sample) + 7) // 8
    h = hashlib.md5()
    for fi, samples in blocks:                        # planar [channels * block_size] -> interleaved little-endian
        planar = np.asarray(samples, dtype="<i4").reshape(int(info.channels), -1)
        inter = np.ascontiguousarray(planar.T)
        h.update(inter.view(np.uint8).reshape(-1, 4)[:, :nbytes].tobytes())
    return h.hexdigest()


@pytest.mark.gpu
def test_cpp_testsamples(oracle):
    exe = _exe()
    r = subprocess.run([exe, FIXTURES], capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL CHECKS PASSED" in r.stdout and "FAILED" not in r.stdout
    facts = {}
    for line in r.stdout.splitlines():
        name, _, rest = line.partition(" ")
        facts.setdefault(name, []).append(dict(kv.split("=", 1) for kv in rest.split() if "=" in kv))
    # verify_streaminfo_* (testsamples.rs:88-99): every STREAMINFO field equals the oracle's reading of the same file
    for short in ("pop", "short", "wasted_bits", "non_subset"):
        data = open(os.path.join(FIXTURES, short + ".flac"), "rb").read()
        st, _, info, _ = oracle.stream_open(data)
        assert st == 0
        got = facts["verify_streaminfo_" + short][0]
        for key in ("min_block_size", "max_block_size", "min_frame_size", "max_frame_size", "sample_rate", "channels", "bits_per_sample", "samples"):
            assert int(got[key]) == int(getattr(info, key)), (short, key)
        assert got["md5sum"] == bytes(info.md5sum).hex()
        # verify_decoded_stream_* (testsamples.rs:164-216): all samples came out, and they hash to the stored checksum
        dec = facts["verify_decoded_stream_" + short][0]
        assert int(dec["samples"]) == int(info.samples) * int(info.channels)
        if any(bytes(info.md5sum)):
            assert dec["md5"] == bytes(info.md5sum).hex()
        else:       # no checksum stored (non_subset.flac): hash the oracle's decode of the same stream instead
            assert dec["md5"] == _oracle_md5(oracle, data, info)
        assert int(facts["verify_blocks_" + short][0]["samples_per_channel"]) == int(info.samples)
    # regression_test_fuzz_samples (testsamples.rs:498): each file ends the way the oracle says it ends
    seen = 0
    for line in r.stdout.splitlines():
        m = re.match(r"regression_test_fuzz_samples (\S+\.flac) end=(-?\d+) blocks=(\d+)", line)
        if not m:
            continue
        seen += 1
        si, blocks, st, _ = oracle.decode_stream(open(os.path.join(FIXTURES, "fuzz", m.group(1)), "rb").read())
        want_end = -1 if si is None else (1 if st != 0 else 0)
        assert (int(m.group(2)), int(m.group(3))) == (want_end, len(blocks)), m.group(1)
    assert seen == 23
