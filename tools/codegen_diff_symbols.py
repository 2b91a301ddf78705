"""developer: per-kernel comparison of the gfx950 code in two sets of object files (sources moved between translation units: the split of
kernels.hip in round 6).  usage: python3 tools/codegen_diff_symbols.py before.o [before2.o ...] -- after.o [after2.o ...]
Every symbol of the 'before' set must exist in the 'after' set with the same instruction stream (addresses and branch targets aside)."""
import os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
def symbols(objs):
    out = {}
    for o in objs:
        d = tempfile.mkdtemp(prefix="cgs", dir="/tmp")
        shutil.copy(o, os.path.join(d, "x.o"))
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", "x.o"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [x for x in os.listdir(d) if "gfx950" in x]
        if not co: continue
        dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(d, co[0])], stdout=subprocess.PIPE, text=True).stdout
        cur = None
        for l in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
            if m: cur = m.group(1); out[cur] = []; continue
            if cur and re.match(r"^\s+[a-z_0-9]+", l):
                l = re.sub(r"//.*", "", l).strip()
                l = re.sub(r"<[^>]*>", "<>", l)            # branch targets by symbol + offset
                l = re.sub(r"\b(s_getpc|s_add_u32|s_addc_u32)\b.*", r"\1 ...", l)   # pc-relative address arithmetic
                out[cur].append(l)
        for k in out:   # trailing padding belongs to the object's layout, not to the function
            while out[k] and out[k][-1].split()[0] in ("s_code_end", "s_nop"): out[k].pop()
        shutil.rmtree(d)
    return out
i = sys.argv.index("--")
a, b = symbols(sys.argv[1:i]), symbols(sys.argv[i + 1:])
bad = 0
for k in sorted(a):
    if k not in b: print("MISSING", k); bad += 1
    elif a[k] != b[k]:
        n = sum(1 for x, y in zip(a[k], b[k]) if x != y) + abs(len(a[k]) - len(b[k]))
        print(f"DIFFERENT {k}: {len(a[k])} -> {len(b[k])} instructions, {n} lines differ"); bad += 1
print(f"{len(a)} kernels / functions compared, {bad} differ")
sys.exit(1 if bad else 0)
