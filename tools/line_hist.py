# developer: instructions of the bench kernel per source line region (build with `make EXTRA=-gline-tables-only`), largest first
# usage: python3 tools/line_hist.py <libspcbpt_hip.so built with line tables>
import re,collections,subprocess,tempfile,shutil,os,sys
LLVM="/opt/rocm/lib/llvm/bin"
d=tempfile.mkdtemp(prefix="cg",dir="/tmp")
lib=shutil.copy(sys.argv[1],d)
subprocess.run([LLVM+"/llvm-objdump","--offloading",lib],cwd=d,stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL)
f=[x for x in sorted(os.listdir(d)) if "gfx950" in x][0]
out=subprocess.run([LLVM+"/llvm-objdump","-d","-l","--no-show-raw-insn",os.path.join(d,f)],stdout=subprocess.PIPE,text=True).stdout
lines=out.splitlines()
start=[i for i,l in enumerate(lines) if "<_ZN3spc8k_spcbptILb0ELb1ELb1ELb0EEEvNS_7KParamsE>:" in l][0]
cur=None; hist=collections.Counter(); fhist=collections.Counter()
for l in lines[start+1:]:
    if re.match(r'^[0-9a-f]+ <',l): break
    m=re.match(r'^; (\S+):(\d+)',l)
    if m: cur=(os.path.basename(m.group(1)),int(m.group(2))); continue
    if re.match(r'^\s+[a-z_0-9]+\s',l) and cur: hist[cur]+=1; fhist[cur[0]]+=1
print(fhist)
# bucket by file and 20-line ranges
b=collections.Counter()
for (fn,ln),c in hist.items(): b[(fn,ln//10*10)]+=c
for (fn,ln),c in sorted(b.items(), key=lambda x:-x[1])[:70]: print(f"{c:5d} {fn}:{ln}")
