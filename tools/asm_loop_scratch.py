"""Kernels with scratch traffic INSIDE a loop, from hipcc --cuda-device-only -S assembly files: a scratch reload
next to memory stores drains the whole in-order vmcnt queue, so these are the spills that cost.

    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 --cuda-device-only -S riemann_amd/csrc/rmh_kernels_big.hip -o /tmp/k.s
    python tools/asm_loop_scratch.py /tmp/k.s ...
"""
import re,sys,subprocess
def dem(n):
    return subprocess.run(["c++filt",n],capture_output=True,text=True).stdout.strip()
for f in sys.argv[1:]:
    s=open(f).read()
    for m in re.finditer(r"^(_ZN3rmh\S+):[^\n]*\n", s, re.M):
        n=m.group(1); i=m.end(); j=s.find(".Lfunc_end", i)
        if j<0: continue
        body=s[i:j].split("\n")
        inl=0; ins=0; cur=False; mf=0
        for l in body:
            t=l.strip()
            if l.startswith('.LBB'):
                cur = ('Loop' in l)
            elif t.startswith('; %bb'):
                cur = ('Loop' in l)
            if 'scratch_load' in l and cur: inl+=1
            if 'scratch_store' in l and cur: ins+=1
            if 'v_mfma' in l: mf+=1
        if inl:
            print("%3d in-loop reloads %3d in-loop spills mfma %4d  %s" % (inl, ins, mf, dem(n)[:140]))
