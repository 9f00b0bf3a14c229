import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, ctypes as C
from mujoco_rl_manipulate_unknown_objects_amd import engine
L=engine.lib(); n=40
rng=np.random.default_rng(0)
A=np.zeros((n,13,13),np.float32); b=rng.normal(size=(n,13)).astype(np.float32)
for i in range(n):
    G=rng.normal(size=(13,20)); A[i]=(G@G.T+np.eye(13)).astype(np.float32)
At=torch.from_numpy(A).cuda(); bt=torch.from_numpy(b).cuda(); xt=torch.zeros(n,13,device="cuda")
L.grip_test_chol.argtypes=[C.c_void_p]*3+[C.c_int,C.c_void_p]
assert L.grip_test_chol(At.data_ptr(), bt.data_ptr(), xt.data_ptr(), n, None)==0
torch.cuda.synchronize()
ref=np.linalg.solve(A.astype(np.float64), b.astype(np.float64)[...,None])[...,0]
print("chol solve max rel err", np.abs(xt.cpu().numpy()-ref).max()/np.abs(ref).max())
