# developer: instruction mix of the pooled traversal loop of the bench kernel in one or more builds of libspcbpt_hip.so
# usage: python3 tools/loop_mix.py <lib.so> [<lib.so> ...]  ->  total instructions, loop size, v_readlane / scratch accesses / SALU / VALU inside the loop
import re,collections,subprocess,tempfile,shutil,os,sys
LLVM="/opt/rocm/lib/llvm/bin"
def analyse(libpath):
    d=tempfile.mkdtemp(prefix="cg",dir="/tmp")
    lib=shutil.copy(libpath,d)
    subprocess.run([LLVM+"/llvm-objdump","--offloading",lib],cwd=d,stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL)
    f=[x for x in sorted(os.listdir(d)) if "gfx950" in x][0]
    out=subprocess.run([LLVM+"/llvm-objdump","-d","--no-show-raw-insn",os.path.join(d,f)],stdout=subprocess.PIPE,text=True).stdout
    lines=out.splitlines()
    start=[i for i,l in enumerate(lines) if "<_ZN3spc8k_spcbptILb0ELb1ELb1ELb0EEEvNS_7KParamsE>:" in l][0]
    ins=[];base=None
    for l in lines[start+1:]:
        if re.match(r'^[0-9a-f]+ <',l): break
        m=re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*(.*)$',l)
        if m:
            a=int(m.group(3),16)
            if base is None: base=a
            t=re.search(r'<[^>]*\+0x([0-9a-f]+)>',m.group(4))
            ins.append((a,m.group(1),m.group(2),int(t.group(1),16)+base if t else None))
    idx={a:i for i,(a,_,_,_) in enumerate(ins)}
    for i,(a,op,args,tgt) in enumerate(ins):
        if tgt is not None and tgt<=a and tgt in idx:
            s=idx[tgt]; ops=[ins[k][1] for k in range(s,i+1)]
            if 'ds_add_rtn_u32' in ops and 'v_pk_fma_f32' in ops and i-s>800:
                c=collections.Counter(ops)
                return dict(total=len(ins),loop=i-s+1,readlane=c['v_readlane_b32'],writelane=c['v_writelane_b32'],scratch=sum(v for k,v in c.items() if k.startswith('scratch')),salu=sum(v for k,v in c.items() if k.startswith('s_')),valu=sum(v for k,v in c.items() if k.startswith('v_')))
for p in sys.argv[1:]: print(p, analyse(p))
