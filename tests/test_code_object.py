"""The gfx950 code object inside libclaxon_hip.so, read on the CPU (no GPU needed): the hot kernels' register files, spills and LDS
are what DESIGN.md section 4.2 says they are.  Round 5 ended with clx_k_lean spilling three vector registers to scratch after its
last kernel change and nobody looked (weakness 5): this is the look."""
import os
import re
import subprocess
import tempfile

import pytest

import claxon_amd as cx

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_notes():
    """{kernel name: {field: int}} from the AMDGPU metadata note of the library's gfx950 code object."""
    if not os.path.exists(cx.LIB_PATH):
        cx.build()
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("no LLVM binary tools under %s" % LLVM)
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "co.elf")
        subprocess.check_call([tools[0], "-O", "binary", "--only-section=.hip_fatbin", cx.LIB_PATH, fat])
        targets = subprocess.check_output([tools[1], "--list", "--type=o", "--input=" + fat], text=True).split()
        gfx = [t for t in targets if t.endswith("gfx950")]
        assert len(gfx) == 1, targets                      # (one device target: no second architecture, no generic fallback)
        subprocess.check_call([tools[1], "--unbundle", "--type=o", "--input=" + fat, "--targets=" + gfx[0], "--output=" + co])
        notes = subprocess.check_output([tools[2], "--notes", co], text=True)
    out = {}
    for block in re.split(r"\n  - ", notes):
        name = re.search(r"\.name:\s+(clx_k_\w+)\n", block)
        if not name:
            continue
        out[name.group(1)] = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)(?=\n|$)", block)}
    return out


@pytest.fixture(scope="module")
def notes():
    return kernel_notes()


def test_hot_kernels_stay_off_scratch(notes):
    """The decode kernels of the timed steps: no vector register spilled, no private segment (every wave of a kernel with one sets up
    scratch), the register files that give 3 (clx_k_lean) / 2 (clx_k_lean24) waves per SIMD, the LDS that gives ten waves per CU."""
    for name, vgprs in (("clx_k_lean", 168), ("clx_k_lean24", 256)):
        k = notes[name]
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (name, k)
        assert k["vgpr_count"] <= vgprs and k["agpr_count"] == 0, (name, k)
        assert k["group_segment_fixed_size"] == 15360, (name, k)
        assert k["wavefront_size"] == 64 and k["max_flat_workgroup_size"] == 64, (name, k)
    k = notes["clx_k_scan"]
    assert k["vgpr_spill_count"] == 0 and k["sgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
    assert k["vgpr_count"] <= 96 and k["group_segment_fixed_size"] == 7168, k        # (five scan waves per SIMD by registers)


def test_every_kernel_is_wave64_and_within_its_budget(notes):
    """Every kernel of the library: wave64 code, LDS that lets at least three workgroups share a CU's 160 KiB (one of clx_k_predict's: eight
    waves and a ring of twelve tiles), no dynamic stack (no indirect call, no recursion)."""
    assert len(notes) >= 20, sorted(notes)
    for name, k in notes.items():
        assert k["wavefront_size"] == 64, name
        assert k.get("uses_dynamic_stack", 0) == 0, name
        assert k["group_segment_fixed_size"] <= (96 if name == "clx_k_predict" else 50) * 1024, (name, k["group_segment_fixed_size"])
    # the kernels of the wave path (frozen design: DESIGN.md 4.1) keep their occupancy too
    assert notes["clx_k_residual"]["vgpr_count"] <= 64 and notes["clx_k_residual"]["private_segment_fixed_size"] == 0
