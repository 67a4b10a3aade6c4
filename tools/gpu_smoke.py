import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
import obca_amd as OA
from obca_amd import scenarios as S
import oracle as O
B=int(sys.argv[1]) if len(sys.argv)>1 else 8; N=80
bt=S.make_batch(S.BACKWARDS,B,N)
xWS=bt['xWS'].copy(); xWS[:,0,:]=bt['x0']
ctx=OA.Context(0); print(ctx.name())
# dualws parity
ls,ns,ds=OA.dualmult_ws_batch(N,bt['vOb'],bt['A'],bt['b'],xWS[:,:,0],xWS[:,:,1],xWS[:,:,2],bt['ego'])
lo,no,do=O.dualmult_ws(N,bt['vOb'],bt['A'],bt['b'],xWS[0,:,0],xWS[0,:,1],xWS[0,:,2],bt['ego'])
print('dualws diff', np.abs(ls[0]-lo).max(), np.abs(ns[0]-no).max(), np.abs(ds[0]-do).max())
b=OA.Batch(ctx,B,N)
b.upload(bt['x0'],bt['xF'],bt['Ts'],bt['L'],bt['ego'],bt['XYbounds'],bt['vOb'],bt['A'],bt['b'],xWS[:,:,0],xWS[:,:,1],xWS[:,:,2],0,xWS,bt['uWS'])
for rep in range(3):
    t0=time.perf_counter(); b.solve(); t1=time.perf_counter()
    print('solve wall %.4f s'%(t1-t0), 'kernel ms', b.kernel_ms(), 'solves/s %.1f'%(B/(t1-t0)))
out=b.download()
print('exitflags', out['exitflag'].sum(), '/', B, 'iters mean', out['iters'].mean(), 'max', out['iters'].max())
for i in range(min(B,6)):
    r=O.parking_signed_dist(bt['x0'][i],bt['xF'][i],N,bt['Ts'][i],bt['L'],bt['ego'],bt['XYbounds'],bt['vOb'],bt['A'],bt['b'],xWS[i,:,0],xWS[i,:,1],xWS[i,:,2],0,xWS[i],bt['uWS'][i])
    print(i,'gpu',out['exitflag'][i],out['iters'][i],round(out['obj'][i],6),'oracle',r['exitflag'],r['iters'],round(r['obj'],6),'dx %.2e du %.2e'%(np.abs(out['xp'][i]-r['xp']).max(),np.abs(out['up'][i]-r['up']).max()))
print('scratch MB', b.scratch_bytes()/1e6)
