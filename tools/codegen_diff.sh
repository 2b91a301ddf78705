#!/bin/bash
# developer: are two builds of kernels.o the SAME device code?  (used when sources are moved around: the header split of round 6)
# usage: tools/codegen_diff.sh before.o after.o  -- compares the gfx950 disassembly (addresses and the embedded source hash ignored)
LLVM=/opt/rocm/lib/llvm/bin
W=$(mktemp -d)
for k in 1 2; do
  f=${!k}; mkdir -p $W/$k; cp "$f" $W/$k/x.o
  (cd $W/$k && $LLVM/llvm-objdump --offloading x.o > /dev/null 2>&1; co=$(ls | grep -i gfx950 | head -1); $LLVM/llvm-objdump -d "$co" | sed 's|//.*||' | grep -v "file format" > dis.txt; wc -l dis.txt)
done
if cmp -s $W/1/dis.txt $W/2/dis.txt; then echo "IDENTICAL device code"; else echo "DIFFERENT device code"; diff $W/1/dis.txt $W/2/dis.txt | head -20; fi
rm -rf $W
