"""Checker for tests/golden/trainer_step_cond_b8t4.npz (make_golden.py:trainer_step_golden -- ONE step of the
reference's own SEQTrainer: `_forward` as shipped + loss.backward()): used by the CPU test that pins the oracle's
restatement of trainer.py:107-170 and by the `-m gpu` test that pins grl_amd's SEQTrainer on HIP."""
import numpy as np
import torch

from train_cond_check import sign_pattern


def luts(g, n_classes=625):
    rg = np.random.Generator(np.random.PCG64(int(g['lut_seed'])))

    def unit(a):
        return (a / np.linalg.norm(a, axis=-1, keepdims=True)).astype(np.float32)
    return torch.from_numpy(unit(rg.standard_normal((n_classes, 2048)))), torch.from_numpy(unit(rg.standard_normal((n_classes, 2048))))


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def check_grads(g, prefix, grads, tol, label, cnn_model=False, max_outliers=8):
    """relative L2 over the stored samples, whole-tensor norm and +-1 projections."""
    errs = {}
    for k in [str(k) for k in g[prefix + '.keys']]:
        t = grads.get(k)
        assert t is not None, 'no gradient for %s.%s' % (prefix, k)
        f = t.detach().reshape(-1).double()
        idx = torch.linspace(0, f.numel() - 1, min(256, f.numel())).long().to(f.device)
        val = g['%s.%s.val' % (prefix, k)].astype(np.float64)
        e = np.linalg.norm(f[idx].cpu().numpy() - val) / max(np.linalg.norm(val), 1e-300)
        n_ref = float(g['%s.%s.norm' % (prefix, k)])
        e_norm = abs(float(f.norm()) - n_ref) / n_ref
        proj = np.array([float((f * sign_pattern(f.numel(), sd, f.device)).sum()) for sd in range(4)])
        e_proj = np.abs(proj - g['%s.%s.proj' % (prefix, k)]).max() / n_ref
        errs[k] = max(e, e_norm, e_proj / 4)
    v = np.array(sorted(errs.values()))
    print('%s %s gradients: %d tensors, error median %.1e p90 %.1e max %.1e' % (label, prefix, len(v), np.median(v), v[int(0.9 * len(v))], v[-1]))
    if not cnn_model:
        bad = {k: e for k, e in errs.items() if e > tol}
        assert not bad, bad
        return errs
    # The CNN's gradients under the real 5-term loss go through 50 train-mode ReLU layers.  Round 4: the fixture carries
    # the reference's own float64 run of the SAME step, so every tensor is held to the model of train_cond_check.check():
    # `tol` (1e-3) wherever the reference's own fp32 run is within tol / 2 of its float64 run (`ref_l2err`), else 2.5x
    # that; every >= 2-D weight has to meet it; at most `max_outliers` 1-D vectors may miss it (single ReLU-flip events,
    # bounded by 1e-2); median <= tol / 2 and p90 <= tol over all tensors.
    assert ('%s.%s.ref_l2err' % (prefix, next(iter(errs)))) in g.files, 'fixture without the float64 companion run: regenerate it'
    bad, outliers = {}, []
    for k, e in errs.items():
        ref_err = float(g['%s.%s.ref_l2err' % (prefix, k)])
        tk = tol if ref_err <= tol / 2 else 2.5 * ref_err
        f = grads[k].detach().reshape(-1).double()
        idx = torch.linspace(0, f.numel() - 1, min(256, f.numel())).long().to(f.device)
        v64 = g['%s.%s.f64' % (prefix, k)]
        e64 = np.linalg.norm(f[idx].cpu().numpy() - v64) / max(np.linalg.norm(v64), 1e-300)
        if e <= tk and e64 <= 3 * ref_err + tol:
            continue
        if grads[k].dim() == 1 and e <= 1e-2:
            outliers.append(k)
        else:
            bad[k] = (e, tk, e64, ref_err)
    print('%s %s: %d tensors held to 1e-3 / 2.5x the reference\'s own fp32-vs-float64 error, %d flip outliers among the 1-D tensors%s'
          % (label, prefix, len(errs), len(outliers), (': ' + ', '.join(outliers)) if outliers else ''))
    assert not bad, bad
    assert len(outliers) <= max_outliers, outliers
    assert np.median(v) <= tol / 2 and v[int(0.9 * len(v))] <= tol, (np.median(v), v[int(0.9 * len(v))])
    return errs
