#!/usr/bin/env python3
"""developer: static look at the pooled traversal loop of k_spcbpt in the device assembly -- instruction mix, scratch (spill)
accesses and SGPR-spill v_readlane/v_writelane inside the loop that holds the pool cursor's ds_add_rtn.
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only kernels.hip -o k.s;  python3 tools/loop_asm.py k.s [kernel-substring]"""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_spcbptILb0ELb1ELb1"
L = open(path).read().split("\n")
start = next(i for i, l in enumerate(L) if l.startswith("_Z") and want in l and l.split(":")[0].endswith("E") and ":" in l)
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
K = L[start:end]
print("kernel lines", len(K), "static scratch ops", sum("scratch_" in l for l in K))
# every basic block carries "in Loop: Header=BBx_y Depth=d" for its innermost loop: group the lines by that header
cur = None
groups = collections.defaultdict(list)
depth = {}
for i, l in enumerate(K):
    m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
    if m:
        cur = m.group(1); depth[cur] = int(m.group(2))
    elif l.startswith(".LBB") and "in Loop" not in l or l.startswith("; %bb") and "in Loop" not in l:
        cur = None
    if cur:
        groups[cur].append(i)
cand = [h for h in groups if depth[h] >= 2 and any("ds_add_rtn_u32" in K[i] for i in groups[h])]
print("depth>=2 loops with a ds_add_rtn:", {h: len(groups[h]) for h in cand})
hdr = max(cand, key=lambda h: len(groups[h]))
lo, hi = min(groups[hdr]), max(groups[hdr])
body = [l for l in (K[i] for i in groups[hdr]) if l.startswith("\t") and not l.strip().startswith((";", "."))]
c = collections.Counter(l.split()[0] for l in body)
print("loop", hdr, "lines", lo, hi, "instructions", len(body))
print("scratch", sum(v for k, v in c.items() if k.startswith("scratch")), "readlane/writelane", sum(v for k, v in c.items() if "lane_b32" in k),
      "global_load", sum(v for k, v in c.items() if k.startswith("global_load")), "ds", sum(v for k, v in c.items() if k.startswith("ds_")))
for l in (K[i] for i in groups[hdr]):
    if "scratch_" in l:
        print("   ", l.strip()[:100])
