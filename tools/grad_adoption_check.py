import sys, contextlib, io, collections
sys.path.insert(0, '/root/repo')
import torch
from grl_amd import train_engine as TE
from grl_amd.reid import models
from grl_amd.synthetic import synth_state_dict, synth_clips
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
cnn.load_state_dict(synth_state_dict(cnn, seed=0)); cnn = cnn.to(dev).train()
cl = synth_clips(8, 4, seed=0).to(dev)
for _ in range(2):
    xu, xc = cnn(cl); cnn.zero_grad(set_to_none=True); (xu.sum() + xc.sum()).backward()
st = collections.Counter()
for p in cnn.parameters():
    if p.grad is not None:
        st[p.grad.untyped_storage().data_ptr()] += 1
print('params with grad', sum(st.values()), 'distinct storages', len(st), 'largest group', max(st.values()))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    xu, xc = cnn(cl); cnn.zero_grad(set_to_none=True); (xu.sum() + xc.sum()).backward()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if 'copy' in e.key.lower() or 'Memcpy' in e.key or 'clone' in e.key.lower() or 'contiguous' in e.key.lower() or 'fill' in e.key.lower() or 'zero' in e.key.lower() or 'add' in e.key.lower() or 'cat' in e.key.lower()]
for e in sorted(rows, key=lambda e: -e.count)[:25]:
    print('%-60s %5d' % (e.key[:60], e.count))
