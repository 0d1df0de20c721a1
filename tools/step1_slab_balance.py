"""Step-1 work per z-slab under the cluster culling (host-side estimate with the kernel's skip rule on 8x8x16 tiles): equal-plane slabs are not equal work
when the kernel decays over a small part of the grid.  Feeds the imbalance column of DESIGN.md section 5.   python tools/step1_slab_balance.py"""
import sys,os,numpy as np
sys.path.insert(0,os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm=shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
from scipy.spatial import cKDTree
def part1by2(x):
    x=x.astype(np.uint64)&0x1fffff
    x=(x|(x<<32))&0x1f00000000ffff
    x=(x|(x<<16))&0x1f0000ff0000ff
    x=(x|(x<<8))&0x100f00f00f00f00f
    x=(x|(x<<4))&0x10c30c30c30c30c3
    x=(x|(x<<2))&0x1249249249249249
    return x
for path,hc,prec in (("data/SprayBottle.pc",6,32),("data/rocker.obj",5,32),("data/bunny_small.obj",4,64)):
    pre=HostSolver(path).preprocess(hCoef=float(hc))
    pos=np.asarray(pre["pos"]).reshape(-1,3); wn=np.asarray(pre["wnormal"]).reshape(-1,3)
    n=pre["n"]; cell=pre["cell"]; lam=pre["lam"]; b0=np.asarray(pre["bbox_min"])
    S=len(pos)
    q=np.clip(((pos-pos.min(0))/(pos.max(0)-pos.min(0)+1e-30)*1023).astype(np.int64),0,1023)
    key=part1by2(q[:,0])|(part1by2(q[:,1])<<1)|(part1by2(q[:,2])<<2)
    o=np.argsort(key,kind='stable'); pos=pos[o]; wn=wn[o]
    ncl=(S+63)//64
    cc=np.zeros((ncl,3)); rad=np.zeros(ncl); lw=np.zeros(ncl)
    w=np.linalg.norm(wn,axis=1)
    for c in range(ncl):
        p=pos[c*64:(c+1)*64]; cc[c]=p.mean(0); rad[c]=np.linalg.norm(p-cc[c],axis=1).max(); lw[c]=np.log(w[c*64:(c+1)*64].max()+1e-300)
    eps=6e-8 if prec==32 else 1.1e-16
    skip_base=np.log(64*S/eps)
    tz=16  # culling unit: 8x8x16
    tx=n//8; nzt=n//tz
    tree=cKDTree(pos)
    work=np.zeros(nzt)
    rt=np.sqrt(3.5**2*2+7.5**2)*cell
    ii,jj=np.meshgrid(np.arange(tx),np.arange(tx),indexing='ij')
    for kz in range(nzt):
        cen=np.stack([(ii*8+3.5)*cell+b0[0],(jj*8+3.5)*cell+b0[1],np.full(ii.shape,(kz*tz+7.5)*cell+b0[2])],-1).reshape(-1,3)
        dmin,idx=tree.query(cen)
        ln_anear=np.log(w[idx]+1e-300)
        r_hi=dmin+rt
        # distance tile centre -> cluster centres
        d=np.linalg.norm(cen[:,None,:]-cc[None,:,:],axis=2) if len(cen)*ncl<6e7 else None
        if d is None:
            tot=0
            for a in range(0,len(cen),4096):
                dd=np.linalg.norm(cen[a:a+4096,None,:]-cc[None,:,:],axis=2)
                gap=dd-rt-rad[None,:]-r_hi[a:a+4096,None]
                tot+=(gap<=(skip_base+lw[None,:]-ln_anear[a:a+4096,None])/lam).sum()
            work[kz]=tot
        else:
            gap=d-rt-rad[None,:]-r_hi[:,None]
            work[kz]=(gap<=(skip_base+lw[None,:]-ln_anear[:,None])/lam).sum()
    frac=work.sum()/(nzt*tx*tx*ncl)
    print(path,"n",n,"clusters",ncl,"kept fraction %.3f"%frac)
    for P in (2,4,8):
        per=work.reshape(P,-1).sum(1)
        print("  P=%d equal-plane slabs: max/mean work = %.3f"%(P,per.max()/per.mean()), np.round(per/per.mean(),2))
