#!/usr/bin/env python
"""Which torch operators does ONE SEQTrainer step still launch, and from which line of this repo?  A TorchDispatchMode
sees every aten call; each is attributed to the innermost frame under grl_amd/ or bench.py.  (Everything that is not a
grl_* launch -- fills, copies, cats, adds -- is a small kernel in the middle of the chain.)
   python tools/train_torch_ops.py [math]"""
import collections, contextlib, io, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from grl_amd import train_engine as TE
from grl_amd.reid import models
from grl_amd.reid.train import SEQTrainer
from grl_amd.reid.loss import OIMLoss, PairLoss
from grl_amd.synthetic import synth_clips, synth_state_dict

math = sys.argv[1] if len(sys.argv) > 1 else 'f32'
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
cnn.load_state_dict(synth_state_dict(cnn, seed=0))
cnn, siam, siamv = cnn.to(dev).train(), siam.to(dev).train(), siamv.to(dev).train()
tr = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev),
                OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), None)
opt = torch.optim.SGD(tr._all_params(), lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True, fused=True)
clips = synth_clips(32, 4, seed=0).to(dev)
pids = (torch.arange(32, device=dev) // 2 * 7) % 625
TE.set_math(math)

NO_KERNEL = ('aten.empty', 'aten.view', 'aten.detach', 'aten.select', 'aten.slice', 'aten.as_strided', 'aten._unsafe_view',
             'aten.alias', 'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.t.', 'aten.transpose', 'aten.permute',
             'aten.empty_like', 'aten.empty_strided', 'aten.record_stream', 'aten._reshape_alias', 'aten.split', 'aten.unbind',
             'aten.new_empty', 'aten.lift_fresh', 'aten.is_pinned', 'aten._local_scalar_dense')


class Tracer(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(NO_KERNEL):
            site = '?'
            for fr in reversed(traceback.extract_stack(limit=40)):
                if (ROOT in fr.filename) and 'tools/' not in fr.filename:
                    site = '%s:%d %s' % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                    break
            self.sites[(name.split('.default')[0], site)] += 1
        return func(*args, **(kwargs or {}))


def step():
    loss, _, _, _ = tr._forward([clips], pids, 0, 0)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t = Tracer()
with t:
    step()
torch.cuda.synchronize()
print('kernel-launching aten calls in one %s step: %d' % (math, sum(t.sites.values())))
for (name, site), n in t.sites.most_common(60):
    print('  %4d  %-28s %s' % (n, name, site))
