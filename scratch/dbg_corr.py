import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import corr as ocorr
from vipe_amd.slam.networks import CorrBlock
dev=torch.device('cuda:0')
rng = np.random.default_rng(2)
E, C, h, w = 3, 128, 16, 24
f1 = torch.from_numpy(rng.normal(0, 1, (1, E, C, h, w)).astype(np.float16)).to(dev)
f2 = torch.from_numpy(rng.normal(0, 1, (1, E, C, h, w)).astype(np.float16)).to(dev)
cb = CorrBlock(f1, f2)
coords = np.stack([rng.uniform(-2, w + 1, (1, E, h, w)), rng.uniform(-2, h + 1, (1, E, h, w))], -1).astype(np.float32)
out = cb(torch.from_numpy(coords).to(dev)).cpu().numpy()
ref = ocorr.corr_lookup([lv.cpu().numpy() for lv in cb.corr_pyramid], coords, 3)
bad = np.argwhere(out.view(np.uint16)!=ref.view(np.uint16))
print(len(bad), out.size)
for b in bad[:10]:
    b=tuple(b); print(b, out[b], ref[b], coords[0,b[1],b[3],b[4]], 'level', b[2]//49, 'a,b', (b[2]%49)//7, (b[2]%49)%7)
for i,l in enumerate(cb.corr_pyramid): print(i, l.shape, l.dtype, l.is_contiguous(), float(l.float().abs().max()))
b=tuple(bad[0])
lv=cb.corr_pyramid[3].cpu().numpy()
slab=lv[b[1],b[3],b[4]]
print('slab',slab, slab.view(np.uint16))
x0=np.float32(coords[0,b[1],b[3],b[4],0])/np.float32(8); y0=np.float32(coords[0,b[1],b[3],b[4],1])/np.float32(8)
dx=np.float32(x0-np.floor(x0)); dy=np.float32(y0-np.floor(y0))
print('x0,y0',repr(x0),repr(y0),'dx,dy',repr(dx),repr(dy), 'w11 f32', repr(np.float32(dx*dy)), 'w11 f16', repr(np.float16(np.float32(dx*dy))))
for s in slab.ravel():
    print(float(s), repr(np.float16(np.float32(s)*np.float32(np.float16(np.float32(dx*dy))))))
