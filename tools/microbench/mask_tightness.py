import sys, math, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'ml-hugs_amd'); sys.path.insert(0,'tests')
from diff_gaussian_rasterization import GaussianRasterizationSettings, _debug_forward_state
from hugs_amd import synthetic as syn
dev = torch.device('cuda:0')
P,H,W,D = 200_000,1080,1920,3
cam = syn.pinhole_camera(H,W); g = syn.scene_gaussians(P, cam, seed=0, sigma_px=4.0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
s = GaussianRasterizationSettings(H,W,math.tan(cam['fovx']/2),math.tan(cam['fovy']/2),torch.ones(3,device=dev),1.0,d(cam['world_view_transform']),d(cam['full_proj_transform']),D,d(cam['camera_center']),False,False)
color, radii, st = _debug_forward_state(d(g['means3D']), d(g['opacities']), s, shs=d(g['shs']), scales=d(g['scales']), rotations=d(g['rotations']))
vals = st['values'].cpu().numpy(); masks = st['quad_masks'].cpu().numpy(); rng = st['ranges'].cpu().numpy(); sp = st['splats'].cpu().numpy()
r = np.random.default_rng(0); tiles = r.choice(len(rng), 400, replace=False)
need=0; setb=0; ent=0
for t in tiles:
    a,b = rng[t]
    if b<=a: continue
    ids = vals[a:b]; m = masks[a:b]
    tx,ty = t%120, t//120
    xs = (tx*16+np.arange(16))[None,None,:].astype(np.float64); ys=(ty*16+np.arange(16))[None,:,None].astype(np.float64)
    rec = sp[ids].astype(np.float64)
    dx = rec[:,0][:,None,None]-xs; dy = rec[:,1][:,None,None]-ys
    power = rec[:,2][:,None,None]*dx*dx + rec[:,3][:,None,None]*dx*dy + rec[:,4][:,None,None]*dy*dy
    hit = (power<=0)&(np.minimum(0.99, rec[:,5][:,None,None]*np.exp(power))>=1/255)
    q = hit.reshape(len(ids),2,8,2,8).any(axis=(2,4))  # [n, qy, qx]
    needm = (q[:,0,0]*1 + q[:,0,1]*2 + q[:,1,0]*4 + q[:,1,1]*8).astype(np.int64)
    assert np.all((needm & ~m.astype(np.int64))==0)
    need += sum(bin(x).count('1') for x in needm); setb += sum(bin(int(x)).count('1') for x in m); ent += len(ids)
print('entries',ent,'needed bits',need,'set bits',setb,'ratio',setb/need,'set frac',setb/(4*ent))
