"""Lane-level emulation of the block substitution in trsm64_mfma_kernel (MFMA operand maps, no data shuffles)."""
import numpy as np, scipy.linalg as sla
def mfma(a,b,c):
    # a,b: (64,), c: (64,4)
    A=np.zeros((16,4)); B=np.zeros((4,16))
    for lane in range(64):
        l15,l4=lane&15,lane>>4
        A[l15,l4]=a[lane]; B[l4,l15]=b[lane]
    D=A@B
    d=c.copy()
    for lane in range(64):
        l15,l4=lane&15,lane>>4
        for r in range(4): d[lane,r]+=D[l4+4*r,l15]
    return d
rng=np.random.default_rng(0)
for LDL in (False,True):
    Lf=np.tril(rng.standard_normal((64,64)))+8*np.eye(64)
    if LDL:
        d=rng.uniform(0.5,2,64)*rng.choice([-1,1],64)
        Lu=np.tril(Lf,-1)/8+np.eye(64)
        Dblk=np.tril(Lu,-1)+np.diag(d); Luse=Lu
    else:
        Dblk=Lf.copy(); Luse=Lf
    inv16=np.zeros((4,16,16))
    for b in range(4): inv16[b]=sla.solve_triangular(Luse[16*b:16*b+16,16*b:16*b+16],np.eye(16),lower=True)
    Bm=rng.standard_normal((16,64))  # strip: 16 rows x 64 cols
    lanes=np.arange(64); l15=lanes&15; l4=lanes>>4
    Ln={}
    for cb in range(1,4):
        for ib in range(cb):
            Ln[(cb,ib)]=[ -Dblk[16*cb+l15, 16*ib+4*s+l4] for s in range(4)]
    Iv=[[inv16[cb][l15,4*s+l4] for s in range(4)] for cb in range(4)]
    X=[np.zeros((64,4)) for _ in range(4)]
    for cb in range(4):
        for r in range(4): X[cb][:,r]=Bm[l15,16*cb+l4+4*r]
    for cb in range(4):
        t=X[cb].copy()
        for ib in range(cb):
            for s in range(4): t=mfma(Ln[(cb,ib)][s], X[ib][:,s], t)
        x=np.zeros((64,4))
        for s in range(4): x=mfma(Iv[cb][s], t[:,s], x)
        X[cb]=x
    out=np.zeros((16,64))
    for cb in range(4):
        for r in range(4): out[l15,16*cb+l4+4*r]=X[cb][:,r]
    ref=sla.solve_triangular(Luse,Bm.T,lower=True).T
    print("LDL",LDL,"err",np.abs(out-ref).max())
