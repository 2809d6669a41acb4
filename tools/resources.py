#!/usr/bin/env python3
"""Table of registers, spill bytes, occupancy and LDS of every kernel in one HIP source file (no GPU needed).

usage: python3 tools/resources.py [wavefront.hip|kernels.hip] [filter]
"""
import re, subprocess, sys, os
src = sys.argv[1] if len(sys.argv) > 1 else "wavefront.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bidirectional-path-tracer_amd", "csrc")
flags = "-std=c++17 -O3 -fPIC -ffp-contract=off --offload-arch=gfx950".split()
out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     cwd=here, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: \s*(.+?): (.+?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("bdpt::", "")
        name = re.sub(r"\(.*", "", name)
        cur = {"name": name}; rows.append(cur)
    elif cur is not None: cur[k] = v
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spill':>6s} {'occ':>4s} {'LDS':>6s}")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('SGPRs','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>6s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>6s}")
