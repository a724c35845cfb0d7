import contextlib, io, os, sys
sys.path.insert(0, '/root/repo')
import torch
from grl_amd import train_engine as TE, engine
from grl_amd.reid import models
from grl_amd.synthetic import synth_state_dict, synth_clips_structured
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
cnn.load_state_dict(synth_state_dict(cnn, seed=0, profile='conditioned'))
cnn = cnn.to(dev).train()
clips = synth_clips_structured(8, 4, seed=3).to(dev)
r1, r2 = torch.randn(8, 2048, device=dev), torch.randn(8, 4, 2048, device=dev)
def fb(bwd=True):
    xu, xc = cnn(clips)
    if bwd:
        cnn.zero_grad(set_to_none=True)
        ((xu * r1).sum() + (xc * r2).sum()).backward()
for cfg in sys.argv[1:]:
    trl, wg, bwd = cfg.split(',')
    engine.TRL_STREAMS = trl == '1'; TE.WGRAD_STREAM = wg == '1'
    for _ in range(2): fb(bwd == '1')
    torch.cuda.synchronize()
    cnn.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            fb(bwd == '1')
        print(cfg, 'capture OK'); g.replay(); torch.cuda.synchronize()
    except Exception as e:
        import traceback; traceback.print_exc(limit=12); print(cfg, 'FAILED')
        torch.cuda.synchronize()
