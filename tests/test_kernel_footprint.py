"""What the second-stage kernels OCCUPY is part of the design (rc_l2.hip): next to the following batch's reduce kernel a CU
has 32 KB of LDS and 104 registers per SIMD to spare, and a workgroup that needs more of either runs IN PLACE of a reduce workgroup.  The
compiler also raises a kernel's register claim to 129 once its static LDS limits it to three waves per SIMD.  This test reads the kernel
descriptors out of the built library (no GPU needed) and pins the footprints those measurements led to."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "pyrecode_amd", "librecode_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _descriptors():
    """kernel name (mangled) -> (static LDS bytes, claimed VGPRs) for every gfx950 kernel of the library"""
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(LIB) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or ROCm LLVM tools not present")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([tools[0], "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
        assert starts, "no offload bundle in the library"
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):   # one bundle per translation unit
            bun, elf = os.path.join(d, "b%d.bin" % n), os.path.join(d, "b%d.elf" % n)
            open(bun, "wb").write(data[a:b])
            subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + bun, "--output=" + elf],
                           check=True, stderr=subprocess.DEVNULL)
            image = open(elf, "rb").read()
            sec = subprocess.run([tools[2], "-S", "-W", elf], check=True, capture_output=True, text=True).stdout
            m = re.search(r"\]\s+\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
            if not m:
                continue
            ro_addr, ro_off = int(m.group(1), 16), int(m.group(2), 16)
            syms = subprocess.run([tools[2], "-s", "-W", elf], check=True, capture_output=True, text=True).stdout
            for line in syms.splitlines():
                f = line.split()
                if len(f) >= 8 and f[-1].endswith(".kd") and f[3] == "OBJECT":
                    off = int(f[1], 16) - ro_addr + ro_off
                    kd = image[off:off + 64]
                    lds = struct.unpack_from("<I", kd, 0)[0]
                    rsrc1 = struct.unpack_from("<I", kd, 48)[0]
                    out[f[-1][:-3]] = (lds, ((rsrc1 & 0x3F) + 1) * 8)     # gfx90a+: VGPRs are claimed in granules of 8
    return out


def _find(desc, needle):
    hits = {k: v for k, v in desc.items() if needle in k}
    assert hits, "no kernel named *%s* in the library" % needle
    return hits


def test_level2_kernels_fit_next_to_the_reduce_kernel():
    desc = _descriptors()
    for name in ("k_l2_dir", "k_l2_link", "k_l2_stats", "k_l2_emit"):
        for k, (lds, vgprs) in _find(desc, name).items():
            assert lds <= 2600, "%s: %d bytes of static LDS (at most 2.6 KB: rc_l2.hip)" % (k, lds)
            assert vgprs <= 64, "%s claims %d registers (at most 64: two of its waves fit a SIMD's 104 spare ones)" % (k, vgprs)


def test_gather_takes_no_lds_and_at_most_64_registers():
    desc = _descriptors()
    for k, (lds, vgprs) in _find(desc, "k_gather").items():
        assert lds == 0 and vgprs <= 64, "%s: %d bytes of LDS, %d registers" % (k, lds, vgprs)


def test_steady_state_reduce_kernels_keep_their_occupancy():
    """three-wave workgroups: five of them (31.9 KB each) share a CU's 160 KB and their waves claim at most 128 registers (four per SIMD)"""
    desc = _descriptors()
    seen = 0
    for k, (lds, vgprs) in _find(desc, "k_reduce_tilesILi3ELi4ELb1ELb1ELb1E").items():   # RWAVES = 3, BZ = 4, aligned, explicit loads, level 1
        seen += 1
        assert 5 * lds <= 160 * 1024, "%s: %d bytes of LDS - five workgroups no longer share a CU" % (k, lds)
        assert vgprs <= 128, "%s claims %d registers - four waves no longer share a SIMD" % (k, vgprs)
    assert seen >= 8
