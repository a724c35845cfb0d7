import sys, contextlib, io, time
sys.path.insert(0, '/root/repo')
import torch
from grl_amd import train_engine as TE, engine
from grl_amd.reid import models
from grl_amd.synthetic import synth_state_dict, synth_clips
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
cnn.load_state_dict(synth_state_dict(cnn, seed=0)); cnn = cnn.to(dev).train()
cl = synth_clips(32, 4, seed=0).to(dev)
orig = engine.gemm
def nokb(*a, **kw):
    if kw.get('kblock') and a[3] > 256 and not TE._in_backward[0]: kw['kblock'] = False
    return orig(*a, **kw)
for name, fn in (('kblock', orig), ('no kblock (M > 256)', nokb), ('kblock', orig), ('no kblock (M > 256)', nokb)):
    engine.gemm = fn
    def step():
        xu, xc = cnn(cl); cnn.zero_grad(set_to_none=True); (xu.sum() + xc.sum()).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize()
    print('%-22s CNN forward + backward %.2f ms' % (name, (time.perf_counter() - t0) / 8 * 1e3))
engine.gemm = orig
