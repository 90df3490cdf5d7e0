"""The decode lanes' CRC-16 arithmetic (claxon_amd/csrc/clx_crct.h) is host-compilable: tests/cpp/crct_check.cpp checks it against the
byte-wise CRC-16 (crc.rs:109-112) on random frames split into shares at random granule boundaries.  Test infrastructure; no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trinomial_crc_matches_bytewise_crc16(tmp_path):
    exe = str(tmp_path / "crct_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "crct_check.cpp")])
    out = subprocess.run([exe, "4000"], capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout.decode() + out.stderr.decode()
    assert out.stdout.decode().startswith("ok ")
