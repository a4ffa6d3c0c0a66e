"""Development: the gfx950 ISA of one kernel out of a built object / library, with a static instruction census.
usage: python tools/kernel_isa.py <file.o | librecode_hip.so> <regex on the mangled name> [out.s]
(static counts say nothing about trip counts; they are for comparing two instantiations and for finding scratch traffic)"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path, d):
    fat = os.path.join(d, "fat.bin")
    subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):
        bun, elf = os.path.join(d, "b%d.bin" % n), os.path.join(d, "b%d.elf" % n)
        open(bun, "wb").write(data[a:b])
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + bun, "--output=" + elf],
                           stderr=subprocess.DEVNULL)
        if r.returncode == 0 and os.path.exists(elf):
            yield elf


def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else None
    with tempfile.TemporaryDirectory() as d:
        for elf in code_objects(path, d):
            syms = subprocess.run([LLVM + "/llvm-readelf", "-s", "-W", elf], check=True, capture_output=True, text=True).stdout
            names = [f.split()[-1] for f in syms.splitlines() if " FUNC " in f and pat.search(f.split()[-1])]
            if not names:
                continue
            dis = subprocess.run([LLVM + "/llvm-objdump", "-d", elf], check=True, capture_output=True, text=True).stdout
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", elf], check=True, capture_output=True, text=True).stdout
            for name in names:
                m = re.search(r"^[0-9a-f]+ <%s>:\n(.*?)(?=^[0-9a-f]+ <|\Z)" % re.escape(name), dis, re.S | re.M)
                body = m.group(1)
                ins = [l.split("//")[0].strip() for l in body.splitlines() if l.strip() and not l.strip().startswith("<")]
                ops = [i.split()[0] for i in ins if i]
                cls = lambda p: sum(1 for o in ops if o.startswith(p))
                meta = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(name), notes, re.S)
                mt = meta.group(1) if meta else ""
                g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, mt) or [None, "?"])[1]
                print("%s\n  instructions %d: v_ %d  s_ %d  ds_ %d  global_ %d  scratch_ %d  branches %d  s_waitcnt %d | vgpr %s spill %s sgpr %s sgpr-spill %s scratch %s B lds %s"
                      % (name, len(ops), cls("v_"), cls("s_"), cls("ds_"), cls("global_"), cls("scratch_"), sum(1 for o in ops if "branch" in o), cls("s_waitcnt"),
                         g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
                if out:
                    open(out, "w").write(body)


main()
