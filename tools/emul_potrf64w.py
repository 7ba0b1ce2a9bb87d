"""Lane-level emulation of potrf64w_kernel (csrc/factor.hip): the v_mfma_f64_16x16x4 operand/accumulator lane maps, the
4x4 pivot broadcast, the masked rank-4 updates and the 16x16 inverse by block substitution, checked against numpy for
Cholesky and LDL^T with NaN in the strict upper triangle (written before the first GPU run of the kernel)."""
import numpy as np, scipy.linalg as sla
LANES=np.arange(64); L15=LANES&15; L4=LANES>>4
def mfma(a,b,c):
    A=np.zeros((16,4)); B=np.zeros((4,16))
    A[L15,L4]=a; B[L4,L15]=b
    D=A@B
    d=c.copy()
    for r in range(4): d[:,r]+=D[L4+4*r,L15]
    return d
def piv4(p, LDL, tol=0.0):
    # p: dict (r,k)->value lower 4x4. returns c (Cf), s, dg
    c={}; s=[0]*4; dg=[0]*4
    if LDL:
        def piv(d):
            zero = not (abs(d)>tol)
            return 1.0/(1.0 if zero else d), (0.0 if zero else d)
        s[0],dg[0]=piv(p[0,0]); c[1,0]=p[1,0]; c[2,0]=p[2,0]; c[3,0]=p[3,0]
        x10,x20,x30=p[1,0]*s[0],p[2,0]*s[0],p[3,0]*s[0]
        s[1],dg[1]=piv(p[1,1]-x10*c[1,0]); c[2,1]=p[2,1]-x20*c[1,0]; c[3,1]=p[3,1]-x30*c[1,0]
        x21,x31=c[2,1]*s[1],c[3,1]*s[1]
        s[2],dg[2]=piv(p[2,2]-x20*c[2,0]-x21*c[2,1]); c[3,2]=p[3,2]-x30*c[2,0]-x31*c[2,1]
        x32=c[3,2]*s[2]
        s[3],dg[3]=piv(p[3,3]-x30*c[3,0]-x31*c[3,1]-x32*c[3,2])
    else:
        def piv(t): sc=1/np.sqrt(t); return sc, t*sc
        s[0],dg[0]=piv(p[0,0]); c[1,0]=p[1,0]*s[0]; c[2,0]=p[2,0]*s[0]; c[3,0]=p[3,0]*s[0]
        s[1],dg[1]=piv(p[1,1]-c[1,0]**2); c[2,1]=(p[2,1]-c[2,0]*c[1,0])*s[1]; c[3,1]=(p[3,1]-c[3,0]*c[1,0])*s[1]
        s[2],dg[2]=piv(p[2,2]-c[2,0]**2-c[2,1]**2); c[3,2]=(p[3,2]-c[3,0]*c[2,0]-c[3,1]*c[2,1])*s[2]
        s[3],dg[3]=piv(p[3,3]-c[3,0]**2-c[3,1]**2-c[3,2]**2)
    return c,s,dg
def inv44(c,s,LDL):
    # inverse of L44: Cholesky: L[r][k]=c[r,k], L[k][k]=1/s[k].  LDL: unit lower, l[r][k]=c[r,k]*s[k]
    y=np.zeros((4,4))
    l=lambda r,k: (c[r,k]*s[k] if LDL else c[r,k])
    rd=lambda r: (1.0 if LDL else s[r])
    for q in range(4):
        y[q,q]=rd(q)
        for r in range(q+1,4):
            acc=0.0
            for m in range(q,r): acc+=l(r,m)*y[m,q]
            y[r,q]=-acc*rd(r)
    return y
def potrf64(A, LDL):
    # Lt[cb][b]: (64,4) regs
    Lt=[[None]*4 for _ in range(4)]
    for cb in range(4):
        for b in range(cb+1):
            v=np.zeros((64,4))
            for r in range(4):
                v[:,r]=A[16*cb+L15, 16*b+L4+4*r]
                if cb==b: v[:,r]=np.where(L15>=L4+4*r, v[:,r], 0.0)
            Lt[cb][b]=v
    dvec=np.zeros(64); dinv=np.zeros(64); inv16=np.zeros((4,16,16))
    for b in range(4):
        aop_inv=[None]*4
        for tt in range(4):
            t=4*b+tt
            D=Lt[b][b]
            p={}
            for jj in range(4):
                for kk in range(jj+1):
                    p[jj,kk]=D[(4*tt+jj)+16*kk, tt]   # readlane
            c,s,dg=piv4(p,LDL)
            y=inv44(c,s,LDL)
            # per-lane A operand of the 4x4 inverse: lane(l15=i,l4=k): y[i-4tt][k] if i in group
            ii=L15-4*tt
            aop=np.where((ii>=0)&(ii<4), y[np.clip(ii,0,3), L4], 0.0)
            aop_inv[tt]=aop
            ssel=np.array(s)[L4]
            X=[None]*4; V=[None]*4
            for cb in range(b,4):
                out=mfma(aop, Lt[cb][b][:,tt], np.zeros((64,4)))
                v=out[:,tt]
                x=v*ssel if LDL else v
                if cb==b:
                    # rows j< 4tt -> 0 ; rows in pivot group: exact values
                    jj=L15-4*tt
                    inpiv=(jj>=0)&(jj<4)
                    lval=np.zeros(64); vval=np.zeros(64)
                    for l in range(64):
                        j=jj[l]; k=L4[l]
                        if 0<=j<4:
                            if j==k: lval[l]=dg[k]; vval[l]= dg[k] if LDL else dg[k]
                            elif j>k:
                                lval[l]= c[j,k]*s[k] if LDL else c[j,k]
                                vval[l]= c[j,k]
                    x=np.where(jj<0,0.0,np.where(inpiv,lval,x))
                    v=np.where(jj<0,0.0,np.where(inpiv,vval,v))
                X[cb]=x; V[cb]=v
                Lt[cb][b][:,tt]=x
            # updates
            for cb1 in range(b,4):
                a_op = X[cb1]
                if cb1==b: a_op=np.where(L15>=4*tt+4, a_op, 0.0)
                for cb2 in range(cb1,4):
                    b_op = V[cb2] if LDL else X[cb2]
                    if cb2==b and LDL:
                        # diag entries of pivot rows hold d (not v): only junk upper outputs, fine
                        pass
                    Lt[cb2][cb1]=mfma(-a_op, b_op, Lt[cb2][cb1])
            for k in range(4):
                dvec[t*1+0+ (0)]=dvec[t*1] # noop
            for k in range(4):
                dvec[4*t+k]=dg[k]; dinv[4*t+k]= s[k] if LDL else 1.0
        # inv16 of block b
        Lb=Lt[b][b]
        T=np.zeros((64,4))
        for r in range(4): T[:,r]=np.where(L15==L4+4*r,1.0,0.0)  # identity in acc layout: Y[i][j], i=l4+4r, j=l15
        Y=np.zeros((64,4))
        for p_ in range(4):
            out=mfma(aop_inv[p_], T[:,p_], np.zeros((64,4)))
            Y[:,p_]=out[:,p_]
            if p_<3:
                # T -= L[:, group p] * Y[p]: Aop lane(l15=i,l4=k) = L16[i][4p+k] = Lb reg p
                T=mfma(-Lb[:,p_], Y[:,p_], T)
        for r in range(4):
            inv16[b][L4+4*r, L15]=Y[:,r]
    Lout=np.zeros((64,64))
    for cb in range(4):
        for b in range(cb+1):
            for r in range(4):
                vals=Lt[cb][b][:,r]
                rows=16*cb+L15; cols=16*b+L4+4*r
                m = rows>=cols
                Lout[rows[m],cols[m]]=vals[m]
    return Lout,dvec,dinv,inv16
rng=np.random.default_rng(3)
for LDL in (False,True):
    R=rng.standard_normal((64,64)); A=R@R.T+64*np.eye(64)
    if LDL:
        # quasi-definite-ish indefinite with stable no-pivot LDL: make some diagonal negative dominant
        A=np.diag(rng.choice([-1,1],64)*rng.uniform(50,100,64))+0.5*(R+R.T)
    Ain=A.copy(); Ain[np.triu_indices(64,1)]=np.nan
    Lo,dvec,dinv,inv16=potrf64(np.nan_to_num(Ain,nan=1e300) if False else Ain, LDL)
    if LDL:
        Lu=np.tril(Lo,-1)+np.eye(64); D=np.diag(Lo).copy()
        err=np.abs(Lu@np.diag(D)@Lu.T-A).max(); print("LDL recon err",err, "dvec==diag",np.abs(dvec-D).max(), "dinv",np.abs(dinv-1/D).max())
        for b in range(4):
            blk=Lu[16*b:16*b+16,16*b:16*b+16]; print(" inv16 err",np.abs(inv16[b]@blk-np.eye(16)).max())
    else:
        err=np.abs(Lo@Lo.T-A).max(); print("Chol recon err",err, np.abs(Lo-np.linalg.cholesky(A)).max())
        for b in range(4):
            blk=Lo[16*b:16*b+16,16*b:16*b+16]; print(" inv16 err",np.abs(inv16[b]@blk-np.eye(16)).max())
