"""Shows the chain-stream calibration of the Cholesky schedule (development aid): python tools/calib_show.py"""
import torch, time, sys, os
sys.path.insert(0, os.getcwd())
from superscreen_amd import kernels
n=1024
S=torch.eye(n,dtype=torch.float64,device="cuda")*4
kernels.chol_factor(S,n); torch.cuda.synchronize()
us,g=kernels.chol_chain_stream_costs()
print("default stream:", [round(c,1) for c in us], g)
S=torch.eye(n,dtype=torch.float64,device="cuda")*4
torch.cuda.synchronize(); t=time.perf_counter()
s2=torch.cuda.Stream()
with torch.cuda.stream(s2):
    kernels.chol_factor(S,n); torch.cuda.synchronize()
print("recalibrating call ms", round(1e3*(time.perf_counter()-t),1))
us,g=kernels.chol_chain_stream_costs()
print("torch side stream:", [round(c,1) for c in us], g)
