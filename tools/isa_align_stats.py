"""Dev tool (round 5): how many 8-byte instructions of a graph's assembled kernels straddle a 64-byte line?  (tools/ubench/valu_align: a lone
wave pays about 8 cycles for each; an s_nop costs it 4.)  Works without a GPU: specialises into a temporary directory and disassembles.
usage: isa_align_stats.py workload [workload ...]"""
import os, re, subprocess, sys, tempfile, glob, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
for name in sys.argv[1:]:
    t = workloads.get(name)
    with tempfile.TemporaryDirectory() as d:
        f = fd.compile_table(t, specialize="isa", cache_dir=d, flags=capi.FDG_SPEC_KEEP_SOURCE)
        for src in glob.glob(d + "/*.s"):
            co = src[:-2] + ".o"
            subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", co], check=True)
            out = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True).stdout
            kern = None; stats = collections.OrderedDict()
            for ln in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\w+)>:", ln)
                if m: kern = m.group(1); stats[kern] = [0, 0, 0, 0, 0]; continue
                m = re.match(r"^\s+(\S+).*//\s*([0-9A-F]+):\s*((?:[0-9A-F]{8}\s*)+)$", ln)
                if not m or kern is None: continue
                addr = int(m.group(2), 16); size = 4 * len(m.group(3).split())
                s = stats[kern]; s[0] += 1
                if size >= 8:
                    s[1] += 1
                    if addr % 8: s[2] += 1
                    if addr // 64 != (addr + size - 1) // 64: s[3] += 1
                if m.group(1) == "s_nop": s[4] += 1
            for k, s in stats.items():
                if s[0] > 50: print(f"{name:28s} {k:24s} {s[0]:7d} instructions, {s[1]:7d} of 8 bytes: {s[2]:6d} displaced by 4, {s[3]:6d} straddle a 64-byte line; {s[4]} s_nop")
