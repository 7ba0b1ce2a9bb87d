"""Numerical check behind trsm64_mfma_kernel: blocked Cholesky of the C3 condensed KKT matrix with the triangular solve done
by block substitution with explicit 16x16 inverses vs row-wise substitution (backward error and factor probes).
usage: python tools/emul_block_trsm.py [case118|case1354pegase]"""
import sys, time, numpy as np
sys.path.insert(0,'.')
from madnlp_jl_amd.problems import opf_shaped
from oracle import sparse_condensed as osc, kernels as okern
from oracle.lapack_cpu import LapackCPUSolver, CHOLESKY
import scipy.linalg as sla
case = sys.argv[1] if len(sys.argv)>1 else "case118"
P=opf_shaped(case, du=1e-8)
k=osc.SparseCondensedKKTSystem(P.n,P.m,P.jac_I,P.jac_J,P.hess_I,P.hess_J,P.ind_ineq,P.ind_lb,P.ind_ub,lambda A: LapackCPUSolver(A,CHOLESKY))
for f in ("reg","l_diag","u_diag","l_lower","u_lower","du_diag"): getattr(k,f)[:]=getattr(P,f)
k.jac[:]=P.jac; k.hess[:]=P.hess
k.compress_jacobian(); k.compress_hessian(); okern.set_aug_diagonal(k); k.build_kkt()
Kd=k.aug_com.to_dense(); K=Kd+np.tril(Kd,-1).T
N=K.shape[0]; Np=(N+63)//64*64
A=np.eye(Np); A[:N,:N]=K
print("N",N,"diag range",np.diag(K).min(),np.diag(K).max())
def trsm16(B, L, use_inv):
    # X = B L^-T, L 64x64 lower; block substitution with 16-blocks
    X=np.zeros_like(B)
    for cb in range(4):
        T=B[:,16*cb:16*cb+16].copy()
        for ib in range(cb):
            T-=X[:,16*ib:16*ib+16]@L[16*cb:16*cb+16,16*ib:16*ib+16].T
        Lb=L[16*cb:16*cb+16,16*cb:16*cb+16]
        if use_inv:
            inv=sla.solve_triangular(Lb,np.eye(16),lower=True)
            X[:,16*cb:16*cb+16]=T@inv.T
        else:
            X[:,16*cb:16*cb+16]=sla.solve_triangular(Lb,T.T,lower=True).T
    return X
def chol(A,use_inv):
    A=A.copy(); n=A.shape[0]
    for j in range(0,n,64):
        L=np.linalg.cholesky(A[j:j+64,j:j+64]); A[j:j+64,j:j+64]=L
        if j+64<n:
            X=trsm16(A[j+64:,j:j+64],L,use_inv); A[j+64:,j:j+64]=X
            A[j+64:,j+64:]-=X@X.T
    return np.tril(A)
b=np.random.default_rng(0).standard_normal(Np)
for ui in (False,True):
    t=time.time(); L=chol(A,ui)
    y=sla.solve_triangular(L,b,lower=True); x=sla.solve_triangular(L.T,y,lower=False)
    nrm=np.abs(A).sum(1).max()
    res=np.abs(A@x-b).max()/(nrm*np.abs(x).max()+np.abs(b).max())
    # factorization backward error on probes
    V=np.random.default_rng(1).standard_normal((Np,3))
    fe=np.abs(L@(L.T@V)-A@V).max()/(nrm*np.abs(V).max())
    print("use_inv",ui,"solve bwd err",res,"factor probe err",fe,"time",time.time()-t)
c,low=sla.cho_factor(A,lower=True); x=sla.cho_solve((c,low),b)
print("lapack bwd", np.abs(A@x-b).max()/(nrm*np.abs(x).max()+np.abs(b).max()))
