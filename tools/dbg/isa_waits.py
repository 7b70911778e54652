import re,sys,collections
lines=open(sys.argv[1]).read().split('\n')
infl=[]; hist=collections.Counter(); kinds=collections.Counter(); sites=[]
for i,l in enumerate(lines):
    t=l.strip().split()
    if not t: continue
    op=t[0]
    if op.startswith(('global_load','scratch_load','flat_load','buffer_load')):
        infl.append((i,op))
    elif op=='s_waitcnt' and 'vmcnt' in l:
        m=re.search(r'vmcnt\((\d+)\)',l); n=int(m.group(1))
        waited=infl[:len(infl)-n] if n<len(infl) else []
        if waited:
            k=len(infl)
            hist[min(k,9)]+=1
            if k<=2:
                kk=','.join(sorted(set(o.split('_')[0] for _,o in infl)))
                kinds[kk]+=1; sites.append((i,kk,k))
        infl=infl[len(infl)-n:] if n<len(infl) else infl if n>=len(infl) else []
    elif op.startswith(('s_cbranch','s_branch','s_endpgm','s_swappc','s_setpc')):
        pass
print('loads in flight at a vmcnt wait (static):',sorted(hist.items()))
print('kinds for <=2 in flight:',kinds)
open(sys.argv[1]+'.sites','w').write('\n'.join(f'{i} {k} {n}' for i,k,n in sites))
