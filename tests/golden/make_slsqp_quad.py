"""Generates tests/golden/slsqp_quad_N8.npz: a short quadcopter hop (N=8) solved by scipy SLSQP with autograd derivatives of oracle/nlp_ref_quad.py
(third-party cross-check of the quadcopter oracle optimum; ~2 min).  Run from the repo root: python tests/golden/make_slsqp_quad.py"""
import sys, time, numpy as np, torch
R=__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path[:0]=[R, R+'/oracle']
import oracle_quad as Q
from nlp_ref_quad import QuadNLP
from scipy.optimize import minimize, Bounds
N=8; Ts=round(0.25*80/N*100)/100
x0=Q.X0.copy(); xF=Q.X0.copy(); xF[:3]=[1.6,1.8,2.6]     # a short hop in front of the first wall (reachable in 8 steps)
xWS=Q.warm_start(x0,xF,N)
r=Q.quadcopter_signed_dist(x0,xF,N,Ts,Q.EGO_R,Q.OB_CLAMPED,xWS,1.0)
print('oracle',r['exitflag'],r['iters'],r['obj'],r['t'],flush=True)
nlp=QuadNLP(x0,xF,N,Ts,Q.EGO_R,Q.OB_CLAMPED)
L=Q.layout(N)
# start: the oracle's own starting point construction is internal; use warm start + dual ws via a 0-iteration oracle call
oo=Q.default_opts(); oo.max_iter=0
r0=Q.quadcopter_signed_dist(x0,xF,N,Ts,Q.EGO_R,Q.OB_CLAMPED,xWS,1.0,opts=oo)
v0=np.concatenate([r0['xp'].T[1:].reshape(-1), r0['up'].T.reshape(-1), [r0['t']], r0['lp'].T.reshape(-1), r0['slack'].T.reshape(-1), np.zeros(5*(N+1))])
c0=nlp.c(torch.tensor(v0)).numpy(); 
# row slack = row value
cob=c0[12*N+12:].reshape(N+1,5,2); v0[nlp.iso]=np.maximum(cob[:,:,1].reshape(-1),1e-2)
f=lambda w: nlp.f(torch.tensor(w)).item()
g=lambda w: torch.autograd.functional.jacobian(nlp.f, torch.tensor(w)).numpy()
c=lambda w: nlp.c(torch.tensor(w)).numpy()
J=lambda w: torch.autograd.functional.jacobian(nlp.c, torch.tensor(w)).numpy()
lb=np.where(np.isfinite(nlp.lb), nlp.lb, -np.inf); ub=np.where(np.isfinite(nlp.ub), nlp.ub, np.inf)
v0=np.clip(v0, lb+1e-3, np.where(np.isfinite(ub),ub-1e-3,np.inf))
t0=time.time()
res=minimize(f, v0, jac=g, method='SLSQP', bounds=Bounds(lb,ub), constraints=[dict(type='eq', fun=c, jac=J)], options=dict(maxiter=400, ftol=1e-12))
print('SLSQP',res.status,res.message,res.nit,res.fun,'cviol',np.abs(c(res.x)).max(),'time',time.time()-t0,flush=True)
x,u,t,lam,s,so=nlp.unpack(torch.tensor(res.x))
print('t',float(t),'x err',np.abs(x.numpy().T-r['xp']).max(),'u err',np.abs(u.numpy().T-r['up']).max())
np.savez(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'slsqp_quad_N8.npz'),N=N,Ts=Ts,x0=x0,xF=xF,xWS=xWS,status=res.status,nit=res.nit,obj=res.fun,cviol=np.abs(c(res.x)).max(),xp=x.numpy().T,up=u.numpy().T,t=float(t))
