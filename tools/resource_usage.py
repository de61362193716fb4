#!/usr/bin/env python
"""Kernel resource usage of every __global__ function, as the compiler reports it:
   python tools/resource_usage.py > table.txt
Compiles each translation unit of __graft_entry__.HIP_OBJECTS for the device only with -Rpass-analysis=kernel-resource-usage (the flags of
__graft_entry__.build_hip_lib otherwise) and prints name, file, SGPRs, VGPRs, scratch bytes per lane, occupancy (waves per SIMD), static LDS."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
rows = []
for obj, (src, fl, _) in g.HIP_OBJECTS.items():
    if obj == "api.o":
        continue
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", *fl, "-c", src, "-o", "/dev/null"]
    err = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True).stderr
    cur = None
    for line in err.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"\(.*$", "", name).replace("pcp::", "").replace("(anonymous namespace)::", "")
            cur = {"name": name, "file": os.path.basename(src) + (" TU" + fl[0][-1] if fl else "")}
            rows.append(cur)
            continue
        for key, pat in (("sgpr", r"SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
print(f"{'kernel':<58}{'file':<24}{'SGPR':>5}{'VGPR':>6}{'AGPR':>6}{'scratch':>9}{'occ':>5}{'sLDS':>7}")
print("-" * 120)
for r in rows:
    print(f"{r['name'][:57]:<58}{r['file']:<24}{r.get('sgpr', 0):>5}{r.get('vgpr', 0):>6}{r.get('agpr', 0):>6}{r.get('scratch', 0):>9}{r.get('occ', 0):>5}{r.get('lds', 0):>7}")
