import sys, numpy as np
sys.path.insert(0,'/root/repo')
import obca_amd as OA
from obca_amd import scenarios as S
N,B,sh=80,64,8
bt=S.make_batch(S.BACKWARDS,B,N); xWS=bt['xWS'].copy(); xWS[:,0,:]=bt['x0']
b=OA.Batch(OA.Context(0),B,N)
b.upload(bt['x0'],bt['xF'],bt['Ts'],bt['L'],bt['ego'],bt['XYbounds'],bt['vOb'],bt['A'],bt['b'],xWS[:,:,0],xWS[:,:,1],xWS[:,:,2],0,xWS,bt['uWS'])
for mu0,bp in ((0.1,1e-2),(1e-2,1e-2),(1e-3,1e-3),(1e-4,1e-4),(1e-3,1e-2),(1e-2,1e-3)):
    b.upload(bt['x0'],bt['xF'],bt['Ts'],bt['L'],bt['ego'],bt['XYbounds'],bt['vOb'],bt['A'],bt['b'],xWS[:,:,0],xWS[:,:,1],xWS[:,:,2],0,xWS,bt['uWS'])
    b.solve(); o1=b.download()
    o=OA.default_opts(); o.mu_init=mu0; o.bound_push=bp; o.bound_frac=bp
    b.shift_warm_start(sh); b.solve(o); o2=b.download()
    print(mu0,bp,'cold iters %.1f'%o1['iters'].mean(),'warm iters %.1f max %d conv %.2f'%(o2['iters'].mean(),o2['iters'].max(),(o2['exitflag']==1).mean()), 'kernel ms', b.kernel_ms()[0])
