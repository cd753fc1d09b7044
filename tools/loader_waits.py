"""list s_waitcnt vmcnt(N), N < 8, between the first and the last LDS-DMA instruction of every kernel of a compiled engine (.s from
hipcc -save-temps): a compiler-inserted wait inside a loader wave's loop drains the DMA queue the compiler does not know about.
usage: hipcc ... -save-temps=obj engine.hip; python tools/loader_waits.py [file.s]   (spans of S = 5 instances include consumer code: look at the context)"""
import re,sys
# for every ring kernel: list s_waitcnt vmcnt(N) with small N that sit between the first and the last LDS-DMA instruction of the kernel (the loader's code)
src=open(sys.argv[1] if len(sys.argv) > 1 else 'engine-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
name=None; body=[]
def report(name, body):
    idx=[i for i,l in enumerate(body) if 'global_load_lds_dwordx4' in l and ' nt' in l]
    if not idx: return
    lo,hi=idx[0],idx[-1]
    bad=[(i,body[i].strip()) for i in range(lo,hi) if re.search(r's_waitcnt vmcnt\((\d+)\)',body[i]) and int(re.search(r'vmcnt\((\d+)\)',body[i]).group(1))<8 and 'ASMSTART' not in body[i-1]]
    print(name[:70], 'dma', len(idx), 'small vmcnt waits inside loader span:', len(bad), [b[0]-lo for b in bad][:10])
for l in src:
    m=re.match(r'^(_ZN5rwkvk\d+k_[a-z_]+I[^:]*):',l)
    if m:
        if name: report(name, body)
        name=m.group(1); body=[]
    elif name: body.append(l)
report(name, body)
