"""developer: dynamic profile of the eye megakernel by source line from a rocprofv3 PC-sampling run (tools/pc_sample.sh).
usage: python3 tools/pc_hist.py <rocprofv3 output dir> <library built with -gline-tables-only>
Reads the *pc_sampling*.csv the profiler wrote, keeps the samples that fall inside k_spcbpt<false, true, true, false>, maps their
code-object offsets to file:line through `llvm-objdump -d -l` of the library's gfx950 code object, and prints the shares by file,
by 10-line bucket and by instruction class at the sampled PC (a wave stalled at s_waitcnt is sampled there: the share of samples
on s_waitcnt is the share of wave-time spent waiting)."""
import collections, csv, glob, os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
out_dir, lib = sys.argv[1], sys.argv[2]
kernel_sym = sys.argv[3] if len(sys.argv) > 3 else "_ZN3spc8k_spcbptILb0ELb1ELb1ELb0EEEvNS_7KParamsE"
files = [f for f in glob.glob(os.path.join(out_dir, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
print("pc sampling files:", [(f, os.path.getsize(f)) for f in files])
if not files:
    sys.exit("no pc_sampling csv")
d = tempfile.mkdtemp(prefix="pcs", dir="/tmp")
shutil.copy(lib, os.path.join(d, "lib.so"))
subprocess.run([LLVM + "/llvm-objdump", "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
co = [x for x in sorted(os.listdir(d)) if "gfx950" in x][0]
dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "-l", os.path.join(d, co)], stdout=subprocess.PIPE, text=True).stdout.splitlines()
addr_line, addr_ins, cur, sym, in_k = {}, {}, None, None, False
k_lo = k_hi = None
for l in dis:
    m = re.match(r"^([0-9a-f]+) <([^>]+)>:", l)
    if m:
        sym = m.group(2); in_k = sym == kernel_sym
        if in_k: k_lo = int(m.group(1), 16)
        continue
    m = re.match(r"^; (\S+):(\d+)", l)
    if m:
        cur = (os.path.basename(m.group(1)), int(m.group(2))); continue
    m = re.match(r"^\s+([a-z_0-9]+)\s.*//\s*([0-9A-Fa-f]+):", l)
    if m and in_k:
        a = int(m.group(2), 16)
        addr_line[a] = cur; addr_ins[a] = m.group(1); k_hi = a
print("kernel", kernel_sym, "address range", hex(k_lo or 0), hex(k_hi or 0), "instructions", len(addr_ins))
hist = collections.Counter(); total = 0; inside = 0
hdr = None
for f in files:
    with open(f, newline="") as fh:
        rd = csv.reader(fh)
        hdr = next(rd)
        lo = [h.lower() for h in hdr]
        ci = next((i for i, h in enumerate(lo) if "offset" in h), None)
        if ci is None:
            print("header without an offset column:", hdr); continue
        for row in rd:
            total += 1
            try:
                a = int(row[ci], 0) if not row[ci].isdigit() else int(row[ci])
            except ValueError:
                continue
            if a in addr_ins:
                inside += 1; hist[a] += 1
print("header:", hdr)
print("samples", total, "inside the kernel", inside)
if not inside:
    sys.exit(0)
by_file, by_bucket, by_ins = collections.Counter(), collections.Counter(), collections.Counter()
for a, c in hist.items():
    fl = addr_line.get(a) or ("?", 0)
    by_file[fl[0]] += c; by_bucket[(fl[0], fl[1] // 10 * 10)] += c
    i = addr_ins[a]
    cls = ("s_waitcnt" if i.startswith("s_waitcnt") else "vmem" if i.startswith(("global_", "flat_", "scratch_", "buffer_")) else "lds" if i.startswith("ds_") else
           "valu" if i.startswith("v_") else "salu/branch")
    by_ins[cls] += c
print("by instruction class at the sampled PC:", {k: round(v / inside, 4) for k, v in by_ins.most_common()})
print("by file:", {k: round(v / inside, 4) for k, v in by_file.most_common()})
print("by 10-line bucket (share of samples):")
for (fn, ln), c in by_bucket.most_common(90):
    print(f"  {c / inside:7.4f}  {fn}:{ln}")
# waits by bucket: where the waiting is
wb = collections.Counter()
for a, c in hist.items():
    if addr_ins[a].startswith("s_waitcnt"):
        fl = addr_line.get(a) or ("?", 0); wb[(fl[0], fl[1] // 10 * 10)] += c
print("s_waitcnt samples by bucket:")
for (fn, ln), c in wb.most_common(40):
    print(f"  {c / inside:7.4f}  {fn}:{ln}")
