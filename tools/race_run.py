import sys, time
sys.path.insert(0, '.')
import numpy as np
import tidehunter_amd as TH
lib = sys.argv[2] if len(sys.argv) > 2 else None
th=TH.TideHunter(device=0, lib_path=lib)
B=int(sys.argv[1]) if len(sys.argv)>1 else 4096
seqs=[TH.synth_read(th.lib,1,i,1000,10,0.1) for i in range(B)]
T=sum(len(s) for s in seqs)
ref=None; fails=0; diffs=0; refst=None
for it in range(8):
    th.upload(seqs)
    try:
        th.run(); r=th.collect(); ok=True
    except RuntimeError as e:
        fails+=1; print('run',it,'FAILED',e); ok=False; r=None
    rec=th.debug_stage(6, 32*(8*B+4096), np.int32).copy().reshape(-1,8); par=th.debug_stage(9, 8*T+1024, np.int32).copy()
    canon=np.array([v for x in rec for v in [x[0],x[1],x[2],x[3],x[4],x[6]]+list(par[x[5]:x[5]+x[6]])], dtype=np.int64)
    recs=[(int(x[0]),int(x[1]),int(x[2]),int(x[3]),int(x[4]),tuple(int(v) for v in par[x[5]:x[5]+x[6]])) for x in rec]
    if it==0: ref_recs=recs
    else:
        bad=[(a,b) for a,b in zip(ref_recs,recs) if a!=b]
        if bad: print('run',it,'first differing chain record: ref',bad[0][0],'now',bad[0][1], 'count',len(bad))
    st=[th.debug_stage(k, 4*T, np.int32).copy() for k in (4,5)]+[canon, th.debug_stage(7, 4*T, np.int32).copy()]
    if refst is None: refst=st
    else:
        for name,a,b in zip(('score','from','chains+par','cells'),refst,st):
            if a.shape!=b.shape or not np.array_equal(a,b):
                w=np.nonzero(a!=b)[0] if a.shape==b.shape else []
                print('run',it,'STAGE',name,'differs', a.shape,b.shape, 'n',len(w), 'first idx', w[:6], a[w[:6]] if len(w) else '', b[w[:6]] if len(w) else '')
    if ok:
        if ref is None: ref=r
        elif r!=ref:
            diffs+=1; bad=[i for i,(a,b) in enumerate(zip(ref,r)) if a!=b]; print('run',it,'DIFFERS in',len(bad),'reads, first',bad[:5])
print('fails',fails,'diffs',diffs)
