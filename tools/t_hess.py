import sys, os; sys.path.insert(0,'/root/repo'); os.environ["GRIP_DEBUG_H"]="1"
import numpy as np, torch, ctypes as C
from mujoco_rl_manipulate_unknown_objects_amd import engine
from oracle import orc
sys.argv=[sys.argv[0]]
exec(open('/root/repo/tools/gpu_check.py').read().split("# ---- oracle forward on the same")[0])
for i in [11,30,49]:
    s = orc.Sim(m)
    s.qpos[:] = qpos[i].astype(np.float32); s.qvel[:] = qvel[i].astype(np.float32)
    s.ctrl[:] = ctrl[i].astype(np.float32); s.qacc_warmstart[:] = warm[i].astype(np.float32)
    s.d.xfrc[1][2] = 0.438 * 9.81
    s.forward()
    H=dbg["M"][i]
    print("env",i,"ncon",s.d.ncon,[ (c.g1,c.g2) for c in s.contacts()])
    print(" H sym err", np.abs(H-H.T).max(), " diag", np.round(np.diag(H),2))
    print(" M diag", np.round(np.diag(s.M),3))
    # bottom-zone full quadratic Hessian as an upper bound reference
    J=s.efc_J[:s.d.nefc]; D=s.efc_D[:s.d.nefc]
    Hq=s.M+J.T@np.diag(D)@J
    print(" Hquad diag", np.round(np.diag(Hq),2))
    print(" H-M offdiag block norm (0:7,7:13)", np.abs(H[:7,7:]).max(), "ref", np.abs(Hq[:7,7:]).max())

def cone(jar, D0, impratio, fs, ft):
    mu=fs/np.sqrt(impratio); D1=D0*impratio; D3=D1*ft*ft/(fs*fs)
    S=np.array([mu,fs,fs,ft]); U=S*jar; N=U[0]; T=np.linalg.norm(U[1:])
    if N>=mu*T or (T<=0 and N>=0): return 0.0, np.zeros(4), np.zeros((4,4))
    if mu*N+T<=0 or (T<=0 and N<0):
        D=np.array([D0,D1,D1,D3]); return 0.5*(D*jar*jar).sum(), D*jar, np.diag(D)
    kap=D0/mu**2; s1=1/np.sqrt(1+mu*mu); dist=(mu*T-N)*s1
    nU=np.r_[-s1, mu*s1*U[1:]/T]
    H=np.outer(nU,nU); c2=dist*mu*s1/T
    t=U[1:]/T; H[1:,1:]+=c2*(np.eye(3)-np.outer(t,t))
    return 0.5*kap*dist**2, kap*S*dist*nU, kap*np.outer(S,S)*H
i=49
s = orc.Sim(m)
s.qpos[:] = qpos[i].astype(np.float32); s.qvel[:] = qvel[i].astype(np.float32)
s.ctrl[:] = ctrl[i].astype(np.float32); s.qacc_warmstart[:] = warm[i].astype(np.float32)
s.d.xfrc[1][2] = 0.438 * 9.81
s.forward()
J=s.efc_J[:s.d.nefc].copy(); aref=s.efc_aref[:s.d.nefc].copy(); D=s.efc_D[:s.d.nefc].copy()
def total(x):
    jar=J@x-aref; c=0.5*(x-s.qacc_smooth)@s.M@(x-s.qacc_smooth); H=s.M.copy(); g=s.M@(x-s.qacc_smooth)
    for ci,con in enumerate(s.contacts()):
        a=con.efc_adr
        cc,gg,HH=cone(jar[a:a+4], D[a], 10.0, con.friction[0], con.friction[1])
        c+=cc; g+=J[a:a+4].T@gg; H+=J[a:a+4].T@HH@J[a:a+4]
    return c,g,H
cw,gw,Hw=total(np.array(s.qacc_warmstart)); cs_,gs,Hs=total(np.array(s.qacc_smooth))
print("cost warm",cw,"cost smooth",cs_)
Href = Hw if cw<cs_ else Hs
Hg=dbg["M"][i]
print("max |H_gpu - H_ref| / max|H_ref|", np.abs(Hg-Href).max()/np.abs(Href).max())
print(np.round(Hg[:4,:4],3)); print(np.round(Href[:4,:4],3))
print("contact order gpu:", dbg["con"][i,:2,7:9])


print("gpu per-stage gauss/ccost:", np.round(dbg["bias"][i][:12].reshape(6,2),3))
x0=np.array(s.qacc_warmstart)
xx=x0.copy()
for it in range(3):
    c,g,H=total(xx); ga=0.5*(xx-s.qacc_smooth)@s.M@(xx-s.qacc_smooth); print("ref stage",it+2,"gauss",round(ga,3),"ccost",round(c-ga,3),"|g|",round(np.linalg.norm(g),3))
    p=-np.linalg.solve(H,g); lo,hi=0.0,4.0
    for _ in range(60):
        mid=0.5*(lo+hi); cm,gm,_=total(xx+mid*p)
        if gm@p<0: lo=mid
        else: hi=mid
    xx=xx+0.5*(lo+hi)*p
    if it==0: x1=xx.copy()
print("x stage3 gpu", np.round(dbg["xpos"][i].reshape(-1)[:13],3)); print("x1 ref      ", np.round(x1,3))
ga=lambda x: 0.5*(x-s.qacc_smooth)@s.M@(x-s.qacc_smooth)
xg=dbg["xpos"][i].reshape(-1)[:13].astype(float); print("ref cost fn at gpu x1: gauss", ga(xg), "ccost", total(xg)[0]-ga(xg))


Hd=dbg["M"][i].reshape(-1)
print("lane0 LS [alpha dp hp g0 g1]:"); print(np.round(Hd[0:25].reshape(5,5),4))
print("lane5 LS [alpha dp hp g0 g1]:"); print(np.round(Hd[25:50].reshape(5,5),4))
