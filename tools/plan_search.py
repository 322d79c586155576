"""Candidate stage orders for the mixed-radix z-row plans (fft_radix.h: OFDFT_ZGPLAN): points per lane E and the share of the lanes
that hold a grid point in the entry pattern (first stage) and the exit pattern (last stage) -- the fused z kernels do their pointwise
physics in those patterns.  Build-time aid: no GPU.  usage: python tools/plan_search.py"""
import itertools, math
RAD=[2,3,4,5,6,8,9,10,12,15,16]
CUR={24:(8,[4,3,2]),48:(16,[4,4,3]),60:(16,[4,5,3]),72:(32,[4,6,3]),80:(32,[4,4,5]),96:(32,[4,4,3,2]),120:(32,[4,5,6]),125:(32,[5,5,5]),135:(32,[3,9,5]),144:(64,[4,4,3,3]),160:(32,[4,8,5]),192:(64,[4,4,4,3]),240:(64,[4,4,3,5])}
def info(M,P,R):
    nbf=[M//r for r in R]; nb=[-(-n//P) for n in nbf]; slots=[r*b for r,b in zip(R,nb)]
    E=max(slots)
    eff_in=M/(P*slots[0]); eff_out=M/(P*slots[-1])
    return E,slots[0],slots[-1],eff_in,eff_out
for M,(P0,R0) in CUR.items():
    c=info(M,P0,R0)
    print('M=%d current P=%d %s: E=%d slots in/out %d/%d eff in/out %.2f/%.2f'%(M,P0,R0,*c))
    res=[]
    for n in (2,3,4):
        for R in itertools.product(RAD,repeat=n):
            if math.prod(R)!=M: continue
            for P in (8,16,32,64):
                if P>M: continue
                E,si,so,ei,eo=info(M,P,list(R))
                if E>max(c[0],6): continue
                res.append((-(min(ei,eo)), E, n, -eo, P, R))
    res.sort()
    for r in res[:4]: print('     cand P=%d %s E=%d stages=%d eff min %.2f out %.2f'%(r[4],r[5],r[1],r[2],-r[0],-r[3]))
