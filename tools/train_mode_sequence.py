import sys, importlib.util, torch, time
sys.path.insert(0, '/root/repo')
spec = importlib.util.spec_from_file_location('b', '/root/repo/bench.py'); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from grl_amd import engine
from grl_amd.synthetic import synth_clips
dev = torch.device('cuda:0')
what = sys.argv[1]
cnn, siam, sd, ssd = b.build_models(dev)
clips = synth_clips(32, 4, seed=0).to(dev)
if 'alt' in what:
    for mode in ('bf16x3', 'bf16', 'bf16s'):
        with engine.math_mode(mode):
            for _ in range(5): engine.extract_features(cnn, siam, clips)
    torch.cuda.synchronize()
if 'c3' in what:
    c3 = synth_clips(64, 8, seed=0).to(dev)
    with engine.math_mode('bf16s'):
        for _ in range(5): engine.extract_features(cnn, siam, c3)
    torch.cuda.synchronize(); del c3
if 'tb' in what:
    b.train_block(dev, 0, 1, None, 'nccl')
    if 'empty' in what:
        torch.cuda.empty_cache()
for m in (sys.argv[2:] or ('f32', 'mixed', 'bf16x3', 'bf16s')):
    if 'sleep' in what:
        torch.cuda.synchronize(); time.sleep(3.0)
    if 'gc' in what:
        import gc; gc.collect(); torch.cuda.empty_cache()
    print('   reserved %.1f GB allocated %.1f GB' % (torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9), flush=True)
    print(what, m, round(b.train_step_ms(dev, m), 2), flush=True)
