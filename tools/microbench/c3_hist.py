import sys, math, numpy as np, torch
sys.path.insert(0, "ml-hugs_amd")
from diff_gaussian_rasterization import GaussianRasterizationSettings, _debug_forward_state
from hugs_amd import synthetic as syn
P=int(sys.argv[1]) if len(sys.argv) > 1 else 110210; rng=np.random.default_rng(5); q=rng.standard_normal((P,4))
m={"xyz":(rng.standard_normal((P,3))*np.array([0.22,0.55,0.14])).astype(np.float32),"scales":(0.035/math.sqrt(P/6890.0)*np.exp(0.3*rng.standard_normal((P,3)))).astype(np.float32),
   "rotq":(q/np.linalg.norm(q,axis=1,keepdims=True)*rng.uniform(0.8,1.2,(P,1))).astype(np.float32),"shs":(0.3*rng.standard_normal((P,16,3))).astype(np.float32),"opacity":rng.uniform(0.05,1.0,(P,1)).astype(np.float32)}
cam=syn.rotating_camera(3,10,dist=5.0,fov=0.4,img_size=512)
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().cuda()
st=GaussianRasterizationSettings(512,512,math.tan(cam["fovx"]/2),math.tan(cam["fovy"]/2),torch.ones(3).cuda(),1.0,dev(cam["world_view_transform"]),dev(cam["full_proj_transform"]),0,dev(cam["camera_center"]),False,False)
_,_,s=_debug_forward_state(dev(m["xyz"]),dev(m["opacity"]),st,shs=dev(m["shs"]),scales=dev(m["scales"]),rotations=dev(m["rotq"]))
r=s["ranges"].cpu().numpy(); n=r[:,1]-r[:,0]; nc=s["n_contrib"].cpu().numpy()
print(f"C3: N={s['N']} tiles={len(n)} nonempty={(n>0).sum()} mean(nonempty)={n[n>0].mean():.0f} max={n.max()} <=256:{(n<=256).sum()} <=512:{((n>256)&(n<=512)).sum()} <=1024:{((n>512)&(n<=1024)).sum()} <=2048:{((n>1024)&(n<=2048)).sum()} >2048:{(n>2048).sum()}; n_contrib mean {nc[nc>0].mean():.0f} max {nc.max()}")
lc=(nc & 0x0FFFFFFF).reshape(512,512); depth=lc.reshape(32,16,32,16).max(axis=(1,3)).reshape(-1)  # deepest list position any pixel of the tile composited
d=depth[n>0]; print(f"    deepest composited position per non-empty tile: mean {d.mean():.0f} p90 {np.percentile(d,90):.0f} max {d.max()}; sum {d.sum()} of N")
