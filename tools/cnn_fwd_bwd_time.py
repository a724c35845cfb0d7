import contextlib, io, os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from grl_amd import train_engine as TE
from grl_amd.reid import models
from grl_amd.synthetic import synth_state_dict, synth_clips
dev = torch.device('cuda:0')
math = sys.argv[1]
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
cnn.load_state_dict(synth_state_dict(cnn, seed=0)); cnn = cnn.to(dev).train()
cl = synth_clips(32, 4, seed=0).to(dev)
q1, q2 = torch.randn(32, 2048, device=dev), torch.randn(32, 4, 2048, device=dev)
TE.set_math(math)
def step():
    xu, xc = cnn(cl); cnn.zero_grad(set_to_none=True); ((xu * q1).sum() + (xc * q2).sum()).backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print('%s CNN fwd+bwd %.2f ms' % (math, (time.perf_counter() - t0) / 10 * 1e3))
