"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not a product path.

CPU restatement (plain functional PyTorch-CPU fp32 ops + numpy) of the GRL
hot path, written from the math of the reference and citing the file:line it
follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; ``grl_amd`` never does.

Pinning: the reference ships no tests or golden vectors.
This restatement is pinned against outputs of the reference itself, imported
in the build container with stub modules by ``tests/golden/make_golden.py``;
the resulting vectors are committed under ``tests/golden/`` and checked by
``tests/test_oracle_golden.py``.  ``oim_*`` below restates reid/loss/oim.py, whose
legacy autograd Function cannot be *applied* on torch >= 1.5; its forward/backward
bodies still execute as plain functions on a stub ``self`` (make_golden.py:oim_golden),
so OIM -- logits, loss, input gradient, LUT after the sequential update, and the order
of the two same-LUT updates of one training step -- is pinned like everything else
(tests/golden/oim.npz).

All functions take a flat ``state`` dict with the reference's state_dict keys
(e.g. ``backbone.base.4.0.conv1.weight``).
"""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-5          # nn.BatchNorm default
MOM = 0.1


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
def _bn(state, p, x, train):
    """nn.BatchNorm{1,2}d forward; train mode uses batch statistics and
    updates running stats in ``state`` in place (momentum 0.1, unbiased var
    in the running estimate), eval mode uses running stats."""
    if train and (p + '.num_batches_tracked') in state:
        state[p + '.num_batches_tracked'] += 1
    return F.batch_norm(x, state[p + '.running_mean'], state[p + '.running_var'],
                        state[p + '.weight'], state[p + '.bias'],
                        training=train, momentum=MOM, eps=EPS)


def _bottleneck(state, p, x, stride, train):
    """reid/models/resnets1.py:73-93."""
    out = F.relu(_bn(state, p + '.bn1', F.conv2d(x, state[p + '.conv1.weight']), train))
    out = F.conv2d(out, state[p + '.conv2.weight'], stride=stride, padding=1)
    out = F.relu(_bn(state, p + '.bn2', out, train))
    out = _bn(state, p + '.bn3', F.conv2d(out, state[p + '.conv3.weight']), train)
    if (p + '.downsample.0.weight') in state:
        res = F.conv2d(x, state[p + '.downsample.0.weight'], stride=stride)
        res = _bn(state, p + '.downsample.1', res, train)
    else:
        res = x
    return F.relu(out + res)


_LAYERS = ((4, 3, 1), (5, 4, 2), (6, 6, 2), (7, 3, 1))   # (seq idx, blocks, stride)


def trunk_forward(state, x, train=False, taps=None, prefix='backbone.base'):
    """ResNet-50 trunk, layer4 stride 1 (resnets1.py:101-109, basebranch.py:27-36,54).
    x [N,3,256,128] -> [N,2048,16,8]."""
    x = F.conv2d(x, state[prefix + '.0.weight'], stride=2, padding=3)
    x = F.relu(_bn(state, prefix + '.1', x, train))
    if taps is not None:
        taps['stem'] = x
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    if taps is not None:
        taps['pool'] = x
    for li, (idx, blocks, stride) in enumerate(_LAYERS):
        for b in range(blocks):
            x = _bottleneck(state, '%s.%d.%d' % (prefix, idx, b), x,
                            stride if b == 0 else 1, train)
        if taps is not None:
            taps['layer%d' % (li + 1)] = x
    return x


def gce_forward(state, x, b, t, train=False, taps=None, prefix='backbone'):
    """Global-guided correlation estimation (basebranch.py:56-68).
    x [b*t,2048,16,8] -> (x_uncorr, x_corr, corr_map)."""
    n, c, h, w = x.shape
    x_glo = x.view(b, t, c, h, w).mean(dim=-1).mean(dim=-1).mean(dim=1)
    g = F.linear(x_glo, state[prefix + '.glo_fc.0.weight'], state[prefix + '.glo_fc.0.bias'])
    g = F.relu(_bn(state, prefix + '.glo_fc.1', g, train))
    glo = g.view(b, 1, 1024, 1, 1).expand(b, t, 1024, h, w).reshape(b * t, 1024, h, w)
    y = torch.cat((x, glo), dim=1)
    y = _bn(state, prefix + '.corr_atte.1', F.conv2d(y, state[prefix + '.corr_atte.0.weight']), train)
    y = F.conv2d(y, state[prefix + '.corr_atte.2.weight'])
    y = F.relu(_bn(state, prefix + '.corr_atte.3', y, train))
    y = _bn(state, prefix + '.corr_atte.6', F.conv2d(y, state[prefix + '.corr_atte.5.weight']), train)
    corr_map = torch.sigmoid(y).view(b * t, 1, h, w)
    if taps is not None:
        taps['x_glo'] = x_glo
        taps['glo'] = g
        taps['corr_map'] = corr_map
    return x * (1 - corr_map), x * corr_map, corr_map


def _memo_block(state, p, x1, x2, train):
    """grl_model.py:67-85 (all-1x1 bottleneck on x1+x2, residual, ReLU)."""
    x = x1 + x2
    out = F.relu(_bn(state, p + '.bn1', F.conv2d(x, state[p + '.conv1.weight']), train))
    out = F.relu(_bn(state, p + '.bn2', F.conv2d(out, state[p + '.conv2.weight']), train))
    out = _bn(state, p + '.bn3', F.conv2d(out, state[p + '.conv3.weight']), train)
    return F.relu(out + x)


def _trl_dir(state, p, f1, f2, mlp, memo_blk, memo, xc, xu, train, taps, tag):
    f_a = F.relu(F.conv2d(memo, state['%s.%s.0.weight' % (p, f1)], state['%s.%s.0.bias' % (p, f1)]))
    f_b = F.relu(F.conv2d(xc, state['%s.%s.0.weight' % (p, f2)], state['%s.%s.0.bias' % (p, f2)]))
    d = (f_a - f_b).pow(2).mean(dim=-1).mean(dim=-1)
    hid = F.relu(F.linear(d, state['%s.%s.0.weight' % (p, mlp)]))
    c = torch.sigmoid(F.linear(hid, state['%s.%s.2.weight' % (p, mlp)]))
    bsz, ch = c.shape
    x_temp = xc * c.view(bsz, ch, 1, 1) + xc
    step = x_temp.mean(dim=-1).mean(dim=-1)
    memo = _memo_block(state, '%s.%s' % (p, memo_blk), memo, xu, train)
    if taps is not None:
        taps.setdefault(tag + '_catte', []).append(c)
        taps.setdefault(tag + '_memo', []).append(memo)
    return step, memo


def trl_forward(state, x_uncorr, x_corr, train=False, taps=None,
                prefix='temporal_learning_block'):
    """Temporal reciprocal learning (grl_model.py:131-180).
    x_* [b,t,2048,16,8] -> (f_uncorr [b,2048], f_corr [b,t,2048]).
    Module-call order inside one time step is forward direction first, then
    backward direction -- it matters in train mode because each BN's running
    statistics are updated once per call (T calls per forward)."""
    b, t = x_corr.shape[:2]
    memo_f = x_uncorr.mean(dim=1)
    memo_b = x_uncorr.mean(dim=1)
    steps_f, steps_b = [], []
    for i in range(t):
        s, memo_f = _trl_dir(state, prefix, 'forward_f1', 'forward_f2',
                             'channel_atte_foreward_corr', 'uncorr_memo_forward',
                             memo_f, x_corr[:, i], x_uncorr[:, i], train, taps, 'fwd')
        steps_f.append(s)
        j = t - 1 - i
        s, memo_b = _trl_dir(state, prefix, 'backward_f1', 'backward_f2',
                             'channel_atte_backward_corr', 'uncorr_memo_backward',
                             memo_b, x_corr[:, j], x_uncorr[:, j], train, taps, 'bwd')
        steps_b.append(s)
    f_fwd = torch.stack(steps_f, dim=1)
    f_bwd = torch.stack(steps_b[::-1], dim=1)
    f_corr = f_fwd + f_bwd
    f_uncorr = memo_f.mean(dim=-1).mean(dim=-1) + memo_b.mean(dim=-1).mean(dim=-1)
    return f_uncorr, f_corr


def grl_forward(state, inputs, train=False, taps=None):
    """ResNet50_GRL_Model.forward (grl_model.py:211-228).
    inputs [b,t,3,256,128] -> (x_uncorr [b,2048], x_corr [b,t,2048])."""
    b, t, c, h, w = inputs.shape
    x = trunk_forward(state, inputs.reshape(b * t, c, h, w), train, taps)
    xu, xc, _ = gce_forward(state, x, b, t, train, taps)
    xu = xu.view(b, t, *xu.shape[1:])
    xc = xc.view(b, t, *xc.shape[1:])
    f_uncorr, f_corr = trl_forward(state, xu, xc, train, taps)
    if taps is not None:
        taps['f_uncorr'] = f_uncorr
        taps['f_corr'] = f_corr
    xc = _bn(state, 'corr_bn', f_corr.reshape(b * t, 2048), train).view(b, t, 2048)
    xc = F.normalize(xc, p=2, dim=2)
    xu = F.normalize(_bn(state, 'uncorr_bn', f_uncorr, train), p=2, dim=1)
    return xu, xc


# ----------------------------------------------------------------------------
# Siamese heads
# ----------------------------------------------------------------------------
def self_attention(sstate, x, train=False):
    """Siamese.self_attention (Siamese.py:79-106): Q,K = L2(BN1d(Linear(x)));
    softmax over the last dim of Q K^T (T x T); sum_T(W V) with V = x; L2."""
    b, t, d = x.shape
    flat = x.reshape(b * t, d)
    q = _bn(sstate, 'featQ_bn', F.linear(flat, sstate['featQ.weight'], sstate['featQ.bias']), train)
    q = (q / q.norm(2, 1, keepdim=True)).view(b, t, -1)
    k = _bn(sstate, 'featK_bn', F.linear(flat, sstate['featK.weight'], sstate['featK.bias']), train)
    k = (k / k.norm(2, 1, keepdim=True)).view(b, t, -1)
    w = torch.softmax(torch.matmul(q, k.transpose(-1, -2)), dim=-1)
    pooled = torch.matmul(w, x).sum(1)
    return pooled / pooled.norm(2, 1, keepdim=True)


def _verify_head(sstate, probe, gallery, train):
    """(p_i - g_j)^2 -> BN1d -> Linear(2048,2)  (Siamese.py:127-140)."""
    nb = probe.shape[0]
    diff = (probe.unsqueeze(1) - gallery.unsqueeze(0)).pow(2).reshape(nb * nb, -1)
    diff = _bn(sstate, 'classifierBN', diff, train)
    cls = F.linear(diff, sstate['classifierlinear.weight'], sstate['classifierlinear.bias'])
    return cls.view(nb, nb, -1)


def siamese_forward(sstate, x, train=False):
    """Siamese.forward (Siamese.py:108-142). x [B,T,D], B even, rows are
    interleaved (probe, gallery) pairs."""
    if x.shape[0] % 2:
        raise RuntimeError("the batch size should be even number!")
    bsz, t = x.shape[:2]
    x = x.view(bsz // 2, 2, t, -1)
    probe = self_attention(sstate, x[:, 0].contiguous(), train)
    gallery = self_attention(sstate, x[:, 1].contiguous(), train)
    return _verify_head(sstate, probe, gallery, train), torch.cat((probe, gallery))


def siamese_video_forward(sstate, x, train=False):
    """Siamese_video.forward (Siamese_video.py:158-184). x [B,D]."""
    bsz = x.shape[0]
    x = x.reshape(bsz // 2, 2, -1)
    probe, gallery = x[:, 0], x[:, 1]
    return _verify_head(sstate, probe, gallery, train), torch.cat((probe, gallery))


def extract_features(state, sstate, clips):
    """Eval-mode clip feature (attevaluator.py:100-112): one 6144-d row per
    clip = cat(x_uncorr, self_attention(x_corr), mean_T(x_corr))."""
    with torch.no_grad():
        xu, xc = grl_forward(state, clips, train=False)
        pooled = self_attention(sstate, xc, train=False)
        return torch.cat((xu, pooled, xc.mean(dim=1)), dim=1)


# ----------------------------------------------------------------------------
# evaluator math
# ----------------------------------------------------------------------------
def cosin_dist(qf, gf):
    """attevaluator.py:44-46."""
    return -torch.mm(qf, gf.t())


def pairwise_distance(x, y):
    """attevaluator.py:33-41: sqrt(clamp(|x|^2 + |y|^2 - 2 x.y, 1e-12))."""
    m, n = x.shape[0], y.shape[0]
    d = x.pow(2).sum(1, keepdim=True).expand(m, n) + y.pow(2).sum(1, keepdim=True).expand(n, m).t()
    d = d - 2.0 * torch.mm(x, y.t())
    return d.clamp(min=1e-12).sqrt()


def fma_chain_dot(q, g, order=None, kblock=False):
    """Bit-exact model of the HIP GEMM's accumulation ('GEMM
    numerics'): every output element is ONE fp32 accumulator updated by a
    k-ordered chain of fused multiply-adds, acc = fma(q[k], g[k], acc), which
    is what v_mfma_f32_32x32x2_f32 computes.  numpy has no fp32 fma, so the
    chain is evaluated in float64 with a rounding to fp32 after every step:
    the exact product of two fp32 values (48 bits) plus an fp32 addend is
    exactly representable whenever the exponents are within 2^5 of each other
    and otherwise correctly rounded by fp64 first -- double rounding can differ
    from a true fma in rare half-ulp cases, so the C oracle (oracle/ref_c) is
    the authority and this is its slow cross-check."""
    q = np.asarray(q, np.float32)
    g = np.asarray(g, np.float32)
    k = q.shape[1]
    order = np.arange(k) if order is None else np.asarray(order)
    acc = np.zeros((q.shape[0], g.shape[0]), np.float32)
    tot = np.zeros_like(acc)
    for n, kk in enumerate(order):
        acc = (q[:, kk:kk + 1].astype(np.float64) * g[None, :, kk].astype(np.float64)
               + acc.astype(np.float64)).astype(np.float32)
        if kblock and (n + 1) % 512 == 0 and n + 1 < k:   # segment boundary (gemm_f32.hip SEG_STAGES)
            tot = tot + acc
            acc = np.zeros_like(acc)
    return acc + tot if (kblock and k > 512) else acc


def evaluate(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=100):
    """CMC / mAP (eva_functions.py:134-184): argsort each row, drop gallery
    entries with the query's pid AND camid, first-hit CMC and average
    precision.  Returns (cmc[max_rank] float32, mAP, indices)."""
    num_q, num_g = distmat.shape
    max_rank = min(max_rank, num_g)
    indices = np.argsort(distmat, axis=1)
    matches = (g_pids[indices] == q_pids[:, None]).astype(np.int32)
    all_cmc, all_ap = [], []
    for qi in range(num_q):
        order = indices[qi]
        keep = ~((g_pids[order] == q_pids[qi]) & (g_camids[order] == q_camids[qi]))
        hits = matches[qi][keep]
        if not hits.any():
            continue
        cmc = hits.cumsum()
        cmc[cmc > 1] = 1
        all_cmc.append(cmc[:max_rank])
        cum = hits.cumsum() / (np.arange(hits.size) + 1.0)
        all_ap.append((cum * hits).sum() / hits.sum())
    assert all_cmc, "Error: all query identities do not appear in gallery"
    cmc = np.asarray(all_cmc).astype(np.float32).sum(0) / float(len(all_cmc))
    return cmc, float(np.mean(all_ap)), indices


# ----------------------------------------------------------------------------
# losses (train step)
# ----------------------------------------------------------------------------
class _OIMFn(torch.autograd.Function):
    """reid/loss/oim.py:8-27 as a static Function (pinned to the reference's own forward /
    backward bodies by tests/golden/oim.npz)."""

    @staticmethod
    def forward(ctx, inputs, targets, lut, momentum):
        ctx.save_for_backward(inputs, targets)
        ctx.lut, ctx.momentum = lut, momentum
        return inputs.mm(lut.t())

    @staticmethod
    def backward(ctx, grad_out):
        inputs, targets = ctx.saved_tensors
        grad_in = grad_out.mm(ctx.lut)
        for x, y in zip(inputs, targets):          # sequential, per sample
            ctx.lut[y] = ctx.momentum * ctx.lut[y] + (1. - ctx.momentum) * x
            ctx.lut[y] /= ctx.lut[y].norm()
        return grad_in, None, None, None


def oim_loss(inputs, targets, lut, scalar=30.0, momentum=0.5):
    """OIMLoss.forward (oim.py:46-53)."""
    logits = _OIMFn.apply(inputs, targets, lut, momentum) * scalar
    return F.cross_entropy(logits, targets), logits


def triplet_soft_batch_hard(feat, ids):
    """TripletLoss('soft', batch_hard=True) (triplet.py:16-76, cdist :78-90)."""
    diff = feat.unsqueeze(1) - feat.unsqueeze(0)
    dist = (diff.pow(2).sum(2) + 1e-12).sqrt()
    same = ids.unsqueeze(1).eq(ids.unsqueeze(0))
    eye = torch.eye(feat.shape[0], dtype=torch.bool)
    pos = same & ~eye
    max_pos = (dist * pos.float()).max(1)[0]
    min_neg = (dist + 1e5 * same.float()).min(1)[0]
    return torch.log(1 + torch.exp(max_pos - min_neg))


def pair_loss(score, tar_probe, tar_gallery):
    """PairLoss.forward (pairloss.py:18-45): BCE(score, pid_i == pid_j),
    plus top-1 precision of the (1-s, s) pseudo-logits."""
    n = score.shape[0]
    mask = tar_probe.unsqueeze(0).expand(n, n).eq(tar_gallery.unsqueeze(1).expand(n, n))
    labels = mask.reshape(-1).float()
    s = score.contiguous().view(-1)
    loss = F.binary_cross_entropy(s, labels)
    pred = (s.detach() > 0.5).float()      # argmax over (1-s, s); ties -> class 0
    prec = (pred == labels).float().mean()
    return loss, prec


def trainer_forward(state, sstate, vstate, clips, pids, lut_c, lut_u, scalar=30.0, momentum=0.5):
    """SEQTrainer._forward (reid/train/trainer.py:107-170): frame-level id loss on x_corr and clip-level id loss on
    the Siamese-pooled rows (ONE criterion, one LUT: :126,:138), 20 x pair verification (:143-149), batch-hard soft
    triplet (:141), id loss of the uncorrelated branch through Siamese_video (:151-152); the uncorrelated
    verification loss is evaluated upstream (:157-162) but never added (:165-168).  Returns
    (all_loss, (x_uncorr, x_corr), (siamese_out, siamese_video_out)).  The LUTs are updated in place by the backward
    in autograd order (oim.py:24-26)."""
    b, t = clips.shape[:2]
    xu, xc = grl_forward(state, clips, train=True)
    l_frame, _ = oim_loss(xc.reshape(b * t, -1), pids.repeat_interleave(t), lut_c, scalar, momentum)
    tp, tg = pids[0::2], pids[1::2]
    target = torch.cat((tp, tg))
    cls, sout = siamese_forward(sstate, xc, train=True)
    l_vid, _ = oim_loss(sout, target, lut_c, scalar, momentum)
    l_tri = triplet_soft_batch_hard(sout, target).mean()
    prob = F.softmax(cls.view(-1, 2), dim=-1).view(cls.shape[0], cls.shape[1], 2)[:, :, 1]
    l_ver, _ = pair_loss(prob, tp, tg)
    _, vout = siamese_video_forward(vstate, xu, train=True)
    l_unc, _ = oim_loss(vout, target, lut_u, scalar, momentum)
    return l_unc + (l_frame + l_vid + l_ver * 20 + l_tri), (xu, xc), (sout, vout)


# ----------------------------------------------------------------------------
# training input transforms (flip / erase / ToTensor / Normalize)
# ----------------------------------------------------------------------------
def augment_apply(u8, params, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """numpy restatement of RandomHorizontalFlip + RandomSizedEarser + ToTensor + Normalize
    (seqtransforms.py:92-216) for GIVEN random decisions: u8 [n,T,3,H,W] uint8, params [n, 1+8T]
    int (flip, then per frame erase, left, top, w, h, R, G, B -- the patch is pasted at (left, top)
    of the flipped frame and clipped by it, as PIL's paste does)."""
    u8 = np.asarray(u8)
    n, T, _, H, W = u8.shape
    out = np.empty(u8.shape, np.float32)
    for i in range(n):
        p = [int(v) for v in params[i]]
        for t in range(T):
            fr = u8[i, t, :, :, ::-1].copy() if p[0] else u8[i, t].copy()
            e, x0, y0, w, h, r, g, b = p[1 + 8 * t:9 + 8 * t]
            if e:
                for c, col in enumerate((r, g, b)):
                    fr[c, y0:min(y0 + h, H), x0:min(x0 + w, W)] = col
            f = fr.astype(np.float32) / np.float32(255)
            for c in range(3):
                out[i, t, c] = (f[c] - np.float32(mean[c])) / np.float32(std[c])
    return out
