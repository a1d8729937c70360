import numpy as np, itertools, random
n=12
GROUPS=[[l for l in range(64) if (l>>5)==g5 and (((l>>2)^(l>>3)^(l>>4))&1)==par] for g5 in (0,1) for par in (0,1)]
holesets=list(itertools.combinations(range(n),4))
# precompute for each holeset and group the deposited p for each lane
dep={}
for hs in holesets:
    free=[p for p in range(n) if p not in hs]
    arr=[]
    for grp in GROUPS:
        ps=[]
        for t in grp:
            p=0
            for k in range(6):
                if (t>>k)&1: p|=1<<free[k]
            ps.append(p)
        arr.append(ps)
    dep[hs]=np.array(arr)   # [4][16]
allp=np.stack([dep[hs] for hs in holesets])   # [495][4][16]
def cost(codes):
    v=np.zeros_like(allp)
    for i,c in enumerate(codes):
        v ^= ((allp>>(4+i))&1)*c
    slots=(allp^v)&15
    # max multiplicity per (holeset, group)
    tot=0
    oh=np.zeros(slots.shape[:2]+(16,),int)
    for s in range(16): oh[:,:,s]=(slots==s).sum(axis=2)
    return oh.max(axis=2).sum(axis=1)/4.0     # per holeset factor
print("none", cost([0,0,0,0]).mean())
best=(9,None)
random.seed(0)
for codes in itertools.product(range(1,16),repeat=4):
    pass
# exhaustive is 50625 evals of a vectorised cost (~2 ms each): fine
res=[]
for codes in itertools.product(range(1,16),repeat=4):
    c=cost(codes).mean()
    res.append((c,codes))
res.sort()
print(res[:10])
