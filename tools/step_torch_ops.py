#!/usr/bin/env python
"""Which torch (aten) operators does ONE full SEQTrainer step still launch, and from where?  (Everything that is not a
grl_* launch: fills, copies, cats, adds ... each a small kernel in the middle of the chain.)
   python tools/step_torch_ops.py [math]"""
import os, sys, contextlib, io, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
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
def step():
    loss, _, _, _ = tr._forward([clips], pids, 0, 0)
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.name.startswith('aten::') and e.cpu_parent is not None or e.name.startswith('aten::')]
top = collections.Counter()
where = collections.defaultdict(collections.Counter)
for e in prof.events():
    if not e.name.startswith('aten::'):
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith('aten::'):
        continue                                   # count outermost aten calls only
    top[e.name] += 1
    frame = next((s for s in (e.stack or []) if '/root/repo/' in s or 'grl_amd' in s or 'bench' in s), (e.stack or ['?'])[0] if e.stack else '?')
    where[e.name][frame.strip()[:110]] += 1
print('outermost aten ops in one step (%s):' % math)
for k, v in top.most_common(30):
    print('  %-28s %4d   %s' % (k, v, '; '.join('%s x%d' % kv for kv in where[k].most_common(3))))
mem = [e for e in prof.events() if 'Memcpy' in e.name]
print('device memcpys:', collections.Counter(e.name for e in mem))
