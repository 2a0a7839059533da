"""The vmcnt operands of the s_waitcnt instructions inside every innermost loop that issues window loads
(global_load_dwordx4), per kernel: a software-pipelined loop must not drain its loads (vmcnt(0)) in steady state.

    hipcc --offload-arch=gfx950 -O3 ... -S --offload-device-only -o /tmp/lib.s gipuma_amd/csrc/gipuma_hip.hip
    python scripts/audit_waits.py /tmp/lib.s [kernel-name-substring ...]
"""
import re,sys,collections
src=open(sys.argv[1]).read().split('\n')
cur=None; kernels={}
for i,l in enumerate(src):
    m=re.match(r'^(_ZN2pm\w+):',l)
    if m: cur=m.group(1); kernels[cur]=[i,None]
    if cur and 's_endpgm' in l and kernels[cur][1] is None: kernels[cur][1]=i
want=sys.argv[2:] or ['push_kernelILi15','push_kernelILi25','sweep_cols_kernelILi15ELb1ELi1','sweep_cols_kernelILi25ELb1ELi1','sweep_group_kernelILi15ELi1','init_cols_kernelILi15ELb1ELi1','sweep_kernelILi15ELb1ELb1ELb1ELi4','sweep_kernelILi15ELb1ELb1ELb1ELi1','push_kernel_c4','sweep_cols_kernelILi15ELb1ELi4']
for name,(st,en) in kernels.items():
    if not any(w in name for w in want): continue
    k=src[st:en]
    lab={}
    for i,l in enumerate(k):
        m=re.match(r'^(\.LBB\d+_\d+):',l)
        if m: lab[m.group(1)]=i
    loops=[]
    for i,l in enumerate(k):
        m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.search(r's_branch\s+(\.LBB\d+_\d+)',l)
        if m and m.group(1) in lab and lab[m.group(1)]<i: loops.append((lab[m.group(1)],i))
    inner=[(a,b) for (a,b) in loops if not any((c>=a and d<=b and (c,d)!=(a,b)) for (c,d) in loops)]
    print(name[8:60])
    for a,b in inner:
        body=k[a:b+1]
        ng=sum('global_load_dwordx4' in x for x in body)
        if ng>=1 and b-a>60:
            w=[re.search(r'vmcnt\((\d+)\)',x).group(1) for x in body if 's_waitcnt' in x and 'vmcnt' in x]
            va=sum(re.match(r'\s+v_',x) is not None for x in body)
            print('   loop',a,b,'loads',ng,'valu',va,'div' if any('v_div_scale' in x for x in body) else 'fast','vmcnt',w)
