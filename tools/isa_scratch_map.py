"""Where the scratch accesses of a kernel sit: for every scratch_load / scratch_store of the kernels whose demangled name contains
FILTER, the loop nest of its basic block (the compiler's "in Loop: Header=... Depth=..." comments).  Scratch at depth <= 1 of the fused
persistent kernel is once per TR iteration; depth >= 2 is inside the trip loop.
usage: python tools/isa_scratch_map.py FILE.hip FILTER"""
import os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1]); flt = sys.argv[2]
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-result",
                "-c", src, "-o", os.path.join(tmp, "x.o"), "-save-temps"], cwd=tmp, check=True, capture_output=True)
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
lines = open(os.path.join(tmp, asm)).read().split("\n")
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if not m:
        i += 1; continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    j = i + 1
    while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
        j += 1
    if flt in name:
        depth, hdr, cnt = 0, "", {}
        for k in range(i, j):
            l = lines[k]
            if re.match(r"^\.LBB\d+_\d+:", l):
                depth, hdr = 0, ""
                for q in range(k, min(k + 6, j)):
                    mm = re.search(r"(?:in Loop|Loop Header|Inner Loop Header|Parent Loop).*?(BB\d+_\d+).*?Depth=(\d+)", lines[q])
                    if mm and "Parent" not in lines[q]:
                        depth, hdr = int(mm.group(2)), mm.group(1); break
                    mm = re.search(r"=>This (?:Inner )?Loop Header: Depth=(\d+)", lines[q])
                    if mm:
                        depth, hdr = int(mm.group(1)), "self"; break
            if "scratch_" in l:
                key = (depth, "load" if "scratch_load" in l else "store")
                cnt[key] = cnt.get(key, 0) + 1
        print(name[:100]); print("   ", ", ".join("depth %d %s: %d" % (d, t, c) for (d, t), c in sorted(cnt.items())) or "no scratch")
    i = j
