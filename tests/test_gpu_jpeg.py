"""Frame decode on the device (grl_jpeg_decode_batch, grl_amd/csrc/jpeg.hip) through the C ABI: bit-identical to Pillow's
`Image.open(f).convert('RGB')` (/root/reference/reid/data/video_loader.py:124-141), to the committed fixture and to the C
oracle; batches across workgroup boundaries; the loader -> prefetcher -> extract_features path on real files."""
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu_models(synth_models):
    assert torch.cuda.is_available()
    cnn, siam, siamv = synth_models
    dev = torch.device('cuda:0')
    return cnn.to(dev).eval(), siam.to(dev).eval(), siamv.to(dev).eval()


def _frame(h, w, rng, grey=False):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([128 + 90 * np.sin(xx / rng.uniform(5, 40) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(6, 60) + rng.uniform(0, 6))
                     for _ in range(3)], -1)
    img = np.clip(base + rng.normal(0, 22, (h, w, 3)), 0, 255).astype(np.uint8)
    return img[..., 0] if grey else img


def _encode(img, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format='JPEG', **kw)
    return buf.getvalue()


def _pil_chw(data):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(data)).convert('RGB')).transpose(2, 0, 1))


def test_device_decode_matches_the_committed_pillow_fixture(golden):
    from grl_amd.reid.data.jpeg import decode_jpeg_batch
    from oracle.ref_c import jpeg_decode
    g = golden('jpeg_frames.npz')
    for nm in [k[5:] for k in g.files if k.startswith('jpeg.')]:
        data = g['jpeg.' + nm].tobytes()
        got = decode_jpeg_batch([data], 'cuda')
        want = g['rgb.' + nm].transpose(2, 0, 1)
        assert got.shape == (1,) + want.shape and got.dtype == torch.uint8
        assert np.array_equal(got[0].cpu().numpy(), want), nm
        assert np.array_equal(got[0].cpu().numpy(), jpeg_decode(data).transpose(2, 0, 1)), nm


@pytest.mark.parametrize('h,w', [(256, 128), (17, 33), (250, 130), (1, 1), (31, 2), (64, 48)])
@pytest.mark.parametrize('sub', [0, 1, 2])
def test_device_decode_is_bit_identical_to_pillow(h, w, sub):
    """sizes on and off the MCU grid, 4:4:4 / 4:2:2 / 4:2:0; per batch: qualities 30..100, default and optimised Huffman
    tables (per-frame tables in LDS), restart intervals -- frames of ONE geometry in one launch."""
    from grl_amd.reid.data.jpeg import decode_jpeg_batch
    rng = np.random.default_rng(h * 1000 + w + sub)
    streams = []
    for q, kw in ((30, {}), (75, dict(optimize=True)), (90, {}), (100, {}), (60, dict(restart_marker_blocks=2)), (85, dict(restart_marker_rows=1))):
        try:
            streams.append(_encode(_frame(h, w, rng), quality=q, subsampling=sub, **kw))
        except OSError:
            continue
    streams.append(_encode(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), quality=95, subsampling=sub))     # white noise
    got = decode_jpeg_batch(streams, 'cuda').cpu().numpy()
    for i, s in enumerate(streams):
        assert np.array_equal(got[i], _pil_chw(s)), (h, w, sub, i)


def test_device_decode_batches_across_workgroups_and_grey():
    """130 frames of MARS geometry = three 64-frame entropy workgroups (the last with 2 live lanes), every frame with its
    own content and quality; grey frames; the [B, T] batch shape of a clip tensor."""
    from grl_amd.reid.data.jpeg import JpegBatch, decode_jpeg_batch
    rng = np.random.default_rng(7)
    streams = [_encode(_frame(256, 128, rng), quality=int(rng.integers(40, 98))) for _ in range(130)]
    got = decode_jpeg_batch(JpegBatch(streams, (65, 2)), 'cuda')
    assert got.shape == (65, 2, 3, 256, 128)
    flat = got.view(130, 3, 256, 128).cpu().numpy()
    for i in (0, 1, 63, 64, 65, 127, 128, 129):
        assert np.array_equal(flat[i], _pil_chw(streams[i])), i
    again = decode_jpeg_batch(JpegBatch(streams, (65, 2)), 'cuda')
    assert torch.equal(got, again)
    rgb = [_encode(_frame(48, 40, rng), keep_rgb=True, quality=q) for q in (60, 92)]          # Adobe transform 0: no colour conversion
    gr = decode_jpeg_batch(rgb, 'cuda').cpu().numpy()
    for i, s in enumerate(rgb):
        assert b'Adobe' in s and np.array_equal(gr[i], _pil_chw(s))
    grey = [_encode(_frame(48, 40, rng, grey=True), quality=q) for q in (50, 90)]
    gg = decode_jpeg_batch(grey, 'cuda').cpu().numpy()
    for i, s in enumerate(grey):
        assert np.array_equal(gg[i], _pil_chw(s))


def test_device_decode_rejects_mixed_geometry_and_has_no_host_fallback():
    from grl_amd import _lib
    from grl_amd.reid.data.jpeg import JpegUnsupported, decode_jpeg_batch
    rng = np.random.default_rng(3)
    a, b = _encode(_frame(32, 32, rng), quality=80), _encode(_frame(48, 32, rng), quality=80)
    with pytest.raises(_lib.GrlHipError, match='another geometry'):
        decode_jpeg_batch([a, b], 'cuda')
    with pytest.raises(JpegUnsupported):
        decode_jpeg_batch([_encode(_frame(32, 32, rng), progressive=True)], 'cuda')
    with pytest.raises(_lib.GrlHipError, match='MI355X only'):
        decode_jpeg_batch([a], 'cpu')


def test_loader_to_features_with_device_decode(tmp_path, gpu_models):
    """The reference's loader path on real files: tracklets of JPEG frames -> RawVideoDataset(decode='device') (workers
    only read the files) -> jpeg_collate -> engine.DevicePrefetcher (decode on the prefetch stream) ->
    engine.extract_features; the features equal the ones of the host-decoded (Pillow) uint8 clips bit for bit."""
    from PIL import Image
    from torch.utils.data import DataLoader
    from grl_amd import engine
    from grl_amd.reid.data import RawVideoDataset
    from grl_amd.reid.data.jpeg import jpeg_collate
    cnn, siam, _ = gpu_models
    rng = np.random.default_rng(11)
    tracklets = []
    for tr in range(4):
        paths = []
        for fi in range(6):
            p = os.path.join(tmp_path, 't%d_f%d.jpg' % (tr, fi))
            Image.fromarray(_frame(256, 128, rng)).save(p, format='JPEG', quality=92)
            paths.append(p)
        tracklets.append((paths, tr, tr % 2))
    dev_ds = RawVideoDataset(tracklets, seq_len=4, sample='rrs_test', decode='device')
    host_ds = RawVideoDataset(tracklets, seq_len=4, sample='rrs_test', decode='host')
    dev_loader = DataLoader(dev_ds, batch_size=2, collate_fn=jpeg_collate, num_workers=0)
    host_loader = DataLoader(host_ds, batch_size=2, num_workers=0)
    feats = {}
    for name, loader in (('device', dev_loader), ('host', host_loader)):
        rows = []
        for clips, pids, cams in engine.DevicePrefetcher(loader, 'cuda'):
            assert clips.dtype == torch.uint8 and clips.shape == (2, 4, 3, 256, 128)
            rows.append(engine.extract_features(cnn, siam, clips))
        feats[name] = torch.cat(rows)
    assert feats['device'].shape == (4, 6144)
    assert torch.equal(feats['device'], feats['host'])


@pytest.mark.parametrize('h,w,q', [(128, 64, 90), (144, 56, 85), (321, 150, 95)])
def test_device_decode_then_rect_scale_equals_pillow_on_other_frame_sizes(h, w, q):
    """Datasets whose frames are not 256 x 128 (iLIDS-VID / PRID 2011: 128 x 64; detector crops: anything): the reference
    opens the file and resizes the PIL image (video_loader.py:124-141 -> seqtransforms.py:30-47 RectScale, BILINEAR).
    Device: grl_jpeg_decode_batch -> engine.rect_scale_u8, both on uint8 -- the composition equals Pillow's bit for bit."""
    from PIL import Image
    from grl_amd import engine
    from grl_amd.reid.data.jpeg import JpegBatch, decode_jpeg_batch
    rng = np.random.default_rng(h * 7 + w)
    streams = [_encode(_frame(h, w, rng), quality=q) for _ in range(6)]
    want = np.stack([np.asarray(Image.open(io.BytesIO(s)).convert('RGB').resize((128, 256), Image.BILINEAR)).transpose(2, 0, 1)
                     for s in streams])
    got = engine.rect_scale_u8(decode_jpeg_batch(JpegBatch(streams, (2, 3)), 'cuda'))
    assert got.shape == (2, 3, 3, 256, 128) and got.dtype == torch.uint8
    assert np.array_equal(got.cpu().numpy().reshape(6, 3, 256, 128), want)


def test_mixed_frame_sizes_decode_per_geometry_and_rect_scale(tmp_path, gpu_models):
    """DukeMTMC-VideoReID's crops differ in size from tracklet to tracklet (duke.py:124-145): decode_jpeg_batch(size=)
    decodes per geometry group and RectScales on the device -- every frame equals Pillow's open + convert + resize
    (video_loader.py:124-141, seqtransforms.py:30-47); the loader path (decode='device', nothing else to set) gives the
    features of the host path with host_rect_scale=True bit for bit.  Without `size` a mixed batch stays an error."""
    from PIL import Image
    from torch.utils.data import DataLoader
    from grl_amd import engine, _lib
    from grl_amd.reid.data import RawVideoDataset
    from grl_amd.reid.data.jpeg import JpegBatch, decode_jpeg_batch, jpeg_collate
    rng = np.random.default_rng(23)
    sizes = [(256, 128), (171, 74), (256, 128), (300, 131), (171, 74), (96, 40)]
    streams = [_encode(_frame(h, w, rng, grey=(k == 3)), quality=88, subsampling=0 if k == 5 else 2) if k != 3 else
               _encode(_frame(h, w, rng, grey=True), quality=88) for k, (h, w) in enumerate(sizes)]
    want = np.stack([np.asarray(Image.open(io.BytesIO(s)).convert('RGB').resize((128, 256), Image.BILINEAR)).transpose(2, 0, 1)
                     if hw != (256, 128) else np.asarray(Image.open(io.BytesIO(s)).convert('RGB')).transpose(2, 0, 1)
                     for s, hw in zip(streams, sizes)])
    got = decode_jpeg_batch(JpegBatch(streams, (3, 2)), 'cuda', size=(256, 128))
    assert got.shape == (3, 2, 3, 256, 128)
    assert np.array_equal(got.cpu().numpy().reshape(6, 3, 256, 128), want)
    with pytest.raises(_lib.GrlHipError, match='another geometry'):
        decode_jpeg_batch(JpegBatch(streams, (3, 2)), 'cuda')
    # the loader path
    cnn, siam, _ = gpu_models
    tracklets = []
    for tr, (h, w) in enumerate([(256, 128), (210, 90), (140, 61), (256, 128)]):
        paths = []
        for fi in range(5):
            p = os.path.join(tmp_path, 'm%d_f%d.jpg' % (tr, fi))
            Image.fromarray(_frame(h, w, rng)).save(p, format='JPEG', quality=90)
            paths.append(p)
        tracklets.append((paths, tr, tr % 2))
    dev_loader = DataLoader(RawVideoDataset(tracklets, seq_len=4, sample='rrs_test', decode='device'), batch_size=2,
                            collate_fn=jpeg_collate, num_workers=0)
    host_loader = DataLoader(RawVideoDataset(tracklets, seq_len=4, sample='rrs_test', decode='host', host_rect_scale=True),
                             batch_size=2, num_workers=0)
    feats = {}
    for name, loader in (('device', dev_loader), ('host', host_loader)):
        rows = []
        for clips, pids, cams in engine.DevicePrefetcher(loader, 'cuda'):
            assert clips.dtype == torch.uint8 and clips.shape == (2, 4, 3, 256, 128)
            rows.append(engine.extract_features(cnn, siam, clips))
        feats[name] = torch.cat(rows)
    assert torch.equal(feats['device'], feats['host'])


def test_device_decode_with_per_frame_huffman_tables():
    """More than eight distinct Huffman table sets in one batch (every frame encoded with optimised tables): the
    look-ahead tables are per frame and read through the L2 instead of LDS; mixed batches (default + optimised) use the
    shared sets.  Same pixels as Pillow either way."""
    from grl_amd.reid.data.jpeg import JpegBatch, decode_jpeg_batch
    rng = np.random.default_rng(21)
    many = [_encode(_frame(64, 48, rng), quality=int(rng.integers(30, 99)), optimize=True) for _ in range(70)]
    _, fr = JpegBatch(many, (70,)).pack()
    assert [f.tabset for f in fr] == list(range(70))
    got = decode_jpeg_batch(many, 'cuda').cpu().numpy()
    for i, s in enumerate(many):
        assert np.array_equal(got[i], _pil_chw(s)), i
    mixed = [_encode(_frame(64, 48, rng), quality=80), _encode(_frame(64, 48, rng), quality=80, optimize=True),
             _encode(_frame(64, 48, rng), quality=50), _encode(_frame(64, 48, rng), quality=95, optimize=True)]
    got = decode_jpeg_batch(mixed, 'cuda').cpu().numpy()
    for i, s in enumerate(mixed):
        assert np.array_equal(got[i], _pil_chw(s)), i


def test_device_decode_of_damaged_streams_equals_the_oracle():
    """Truncated / corrupted / marker-riddled scans (tests/test_jpeg_cpu.py:_damaged_streams): no hang, no crash, and the
    pixels the oracle produces for the same bytes (zero bits past the data, as libjpeg feeds them)."""
    from test_jpeg_cpu import _damaged_streams
    from grl_amd import _lib
    from grl_amd.reid.data.jpeg import decode_jpeg_batch
    from oracle.ref_c import jpeg_decode
    n = 0
    for s in _damaged_streams(np.random.default_rng(9), 60):
        try:
            want = jpeg_decode(s)
        except ValueError:
            continue
        try:
            got = decode_jpeg_batch([s], 'cuda')[0].cpu().numpy()
        except _lib.GrlHipError:
            continue                                  # (a header the product parser refuses: fine, loudly)
        assert np.array_equal(got, want.transpose(2, 0, 1)), len(s)
        n += 1
    torch.cuda.synchronize()
    assert n > 40


def test_dense_mode_evaluator_with_device_decode(tmp_path, gpu_models):
    """test_all.py's dense mode (attevaluator.py:68-98: every clip of a tracklet, features averaged) fed from JPEG bytes:
    RawVideoDataset(sample='dense', decode='device') -> jpeg_collate -> ATTEvaluator(only_eval=True).extract_feature equals
    the host-decoded run bit for bit."""
    from PIL import Image
    from torch.utils.data import DataLoader
    from grl_amd.reid.data import RawVideoDataset
    from grl_amd.reid.data.jpeg import jpeg_collate
    from grl_amd.reid.evaluator import ATTEvaluator
    cnn, siam, _ = gpu_models
    rng = np.random.default_rng(13)
    tracklets = []
    for tr in range(3):
        paths = []
        for fi in range(9 + tr):
            p = os.path.join(tmp_path, 'd%d_f%d.jpg' % (tr, fi))
            Image.fromarray(_frame(256, 128, rng)).save(p, format='JPEG', quality=88)
            paths.append(p)
        tracklets.append((paths, tr, tr % 2))
    ev = ATTEvaluator(cnn, siam, only_eval=True)
    feats = {}
    for mode, kw in (('device', dict(collate_fn=jpeg_collate)), ('host', {})):
        ds = RawVideoDataset(tracklets, seq_len=4, sample='dense', decode=mode)
        f, pids, cams = ev.extract_feature(DataLoader(ds, batch_size=1, num_workers=0, **kw))
        feats[mode] = f
        assert list(pids) == [0, 1, 2] and f.shape == (3, 6144)
    assert torch.equal(feats['device'], feats['host'])


def test_workgroup_per_frame_and_lane_per_frame_decoders_agree():
    """grl_jpeg_parallel_mode: the self-synchronising workgroup-per-frame entropy decoder (default) and the one-lane-per-frame
    decoder give the same pixels -- on a MARS-geometry batch with per-frame qualities, on tiny and odd-sized frames (one
    subsequence, partial MCUs), on optimised tables, and on damaged streams."""
    from test_jpeg_cpu import _damaged_streams
    from grl_amd import _lib
    from grl_amd.reid.data.jpeg import decode_jpeg_batch
    lib = _lib.load()
    rng = np.random.default_rng(31)
    batches = [[_encode(_frame(256, 128, rng), quality=int(rng.integers(30, 99))) for _ in range(70)],
               [_encode(_frame(17, 33, rng), quality=q, subsampling=1, optimize=True) for q in (40, 90)],
               [_encode(_frame(1, 1, rng), quality=50)],
               [_encode(rng.integers(0, 256, (64, 48, 3), dtype=np.uint8), quality=100, subsampling=0) for _ in range(3)]]
    for s in _damaged_streams(np.random.default_rng(17), 30):
        batches.append([s])
    n = 0
    for streams in batches:
        outs = []
        for mode in (1, 0):
            was = lib.grl_jpeg_parallel_mode(mode)
            try:
                outs.append(decode_jpeg_batch(streams, 'cuda').cpu())
            except _lib.GrlHipError:
                outs.append(None)
            finally:
                lib.grl_jpeg_parallel_mode(was)
        if outs[0] is None or outs[1] is None:
            assert outs[0] is None and outs[1] is None
            continue
        assert torch.equal(outs[0], outs[1]), len(streams[0])
        n += 1
    torch.cuda.synchronize()
    assert n > 20


def test_trainer_input_from_jpeg_bytes_with_device_augmentation(tmp_path):
    """The TRAINING input path with nothing but file reads on the host: tracklets of JPEG files ->
    RawVideoDataset(decode='device', augment=True) -> jpeg_collate -> engine.DevicePrefetcher (device decode) ->
    SEQTrainer._parse_data (flip / erase / ToTensor / Normalize on the device) gives exactly the float clips of the
    host-decoded (Pillow) path with the same augmentation draws."""
    import random
    from PIL import Image
    from torch.utils.data import DataLoader
    from grl_amd import engine
    from grl_amd.reid.data import RawVideoDataset
    from grl_amd.reid.data.jpeg import jpeg_collate
    from grl_amd.reid.train.trainer import SEQTrainer
    rng = np.random.default_rng(19)
    tracklets = []
    for tr in range(4):
        paths = []
        for fi in range(5):
            p = os.path.join(tmp_path, 'a%d_f%d.jpg' % (tr, fi))
            Image.fromarray(_frame(256, 128, rng)).save(p, format='JPEG', quality=90)
            paths.append(p)
        tracklets.append((paths, tr // 2, tr % 2))
    tr_obj = SEQTrainer.__new__(SEQTrainer)
    tr_obj.device = torch.device('cuda', 0)
    outs = {}
    for mode, kw in (('device', dict(collate_fn=jpeg_collate)), ('host', {})):
        random.seed(5); np.random.seed(5)                      # the same frame picks and augmentation draws for both runs
        ds = RawVideoDataset(tracklets, seq_len=4, sample='rrs_train', augment=True, decode=mode)
        batch = next(iter(engine.DevicePrefetcher(DataLoader(ds, batch_size=4, num_workers=0, **kw), 'cuda')))
        assert len(batch) == 4 and batch[0].dtype == torch.uint8 and batch[0].shape == (4, 4, 3, 256, 128)
        (imgs,), pids = tr_obj._parse_data(batch)
        outs[mode] = (imgs.clone(), pids.clone(), batch[3].clone())
    assert torch.equal(outs['device'][2], outs['host'][2]) and torch.equal(outs['device'][1], outs['host'][1])
    assert outs['device'][0].dtype == torch.float32 and torch.equal(outs['device'][0], outs['host'][0])
