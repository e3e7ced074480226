"""VGPR / scratch / LDS figures of every kernel of one .hip file (hipcc -save-temps, gfx950): python tools/kernel_regs.py FILE [filter]"""
import os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-result",
                "-c", src, "-o", os.path.join(tmp, "x.o"), "-save-temps"], cwd=tmp, check=True, capture_output=True)
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
s = open(os.path.join(tmp, asm)).read()
for b in s.split(".amdhsa_kernel ")[1:]:
    name = b.split("\n")[0]
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in d:
        continue
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\d+)", b) or [None, "?"])[1]
    print("%-90s vgpr %s agpr-off %s scratch %s lds %s" % (d[:90], g("next_free_vgpr"), g("accum_offset"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
