"""Frame decode, CPU half: the C oracle (oracle/ref_c/jpeg_baseline.c) against Pillow itself --
`Image.open(f).convert('RGB')`, the call of /root/reference/reid/data/video_loader.py:124-141 -- and against the
committed fixture; the product library's HOST header parser (grl_jpeg_parse: no GPU call) against the same streams."""
import ctypes as C
import io

import numpy as np
import pytest

from oracle.ref_c import jpeg_decode


def _frame(h, w, rng, grey=False):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([128 + 90 * np.sin(xx / rng.uniform(5, 40) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(6, 60) + rng.uniform(0, 6))
                     for _ in range(3)], -1)
    img = np.clip(base + rng.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)
    return img[..., 0] if grey else img


def _encode(img, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format='JPEG', **kw)
    return buf.getvalue()


def _pil(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))


def test_jpeg_oracle_matches_the_committed_pillow_fixture(golden):
    g = golden('jpeg_frames.npz')
    names = [k[5:] for k in g.files if k.startswith('jpeg.')]
    assert len(names) >= 6
    for nm in names:
        got = jpeg_decode(g['jpeg.' + nm].tobytes())
        assert np.array_equal(got, g['rgb.' + nm]), nm


def test_jpeg_oracle_is_bit_identical_to_pillow_on_a_sweep():
    """Sizes on and off the MCU grid (down to 1 x 1), 4:4:4 / 4:2:2 / 4:2:0, qualities 5..100 (quality 100 = quantiser 1:
    saturating blocks), optimised Huffman tables, restart intervals, grey, saturated colours, white noise."""
    pytest.importorskip('PIL')
    rng = np.random.default_rng(0)
    n = 0
    for (h, w) in [(256, 128), (16, 16), (8, 8), (17, 33), (250, 130), (1, 1), (3, 5), (100, 37), (31, 2), (5, 3), (64, 48)]:
        for sub in (0, 1, 2):
            for q, opt in ((30, False), (75, True), (90, False), (100, True)):
                try:
                    data = _encode(_frame(h, w, rng), quality=q, subsampling=sub, optimize=opt)
                except OSError:                  # Pillow's encoder buffer is too small for some tiny optimised frames
                    continue
                assert np.array_equal(jpeg_decode(data), _pil(data)), (h, w, sub, q, opt)
                n += 1
    for sub in (0, 1, 2):
        for q in (5, 50, 100):
            noise = rng.integers(0, 256, (96, 80, 3), dtype=np.uint8)
            sat = np.zeros((64, 96, 3), np.uint8)
            sat[:, :48] = (255, 0, 0); sat[:32, 48:] = (0, 0, 255); sat[32:, 48:] = (255, 255, 255)
            for img, kw in ((noise, {}), (sat, {}), (noise, dict(restart_marker_blocks=1)), (noise, dict(restart_marker_rows=1)),
                            (noise[:40, :24], dict(restart_marker_blocks=5))):
                data = _encode(np.ascontiguousarray(img), quality=q, subsampling=sub, **kw)
                assert np.array_equal(jpeg_decode(data), _pil(data)), (sub, q, kw)
                n += 1
    for (h, w) in [(256, 128), (17, 33)]:
        data = _encode(_frame(h, w, rng, grey=True), quality=85)
        assert np.array_equal(jpeg_decode(data), _pil(data))
        n += 1
    data = _encode(rng.integers(0, 256, (48, 40, 3), dtype=np.uint8), qtables=[[1] * 64, [3] * 64])
    assert np.array_equal(jpeg_decode(data), _pil(data))
    for kw in (dict(keep_rgb=True), dict(keep_rgb=True, quality=95), dict(keep_rgb=True, subsampling=0)):
        data = _encode(_frame(40, 24, rng), **kw)            # Adobe marker, transform 0: the components ARE R, G, B
        assert b'Adobe' in data and np.array_equal(jpeg_decode(data), _pil(data)), kw
        n += 1
    assert n > 150


def test_jpeg_oracle_and_parser_refuse_what_is_out_of_scope(golden):
    pytest.importorskip('PIL')
    from grl_amd import _lib
    from grl_amd.reid.data.jpeg import JpegBatch, JpegUnsupported
    rng = np.random.default_rng(1)
    prog = _encode(rng.integers(0, 256, (32, 32, 3), dtype=np.uint8), progressive=True)
    with pytest.raises(ValueError, match=r'\(-2\)'):
        jpeg_decode(prog)
    with pytest.raises(ValueError, match=r'\(-1\)'):
        jpeg_decode(b'\x00\x01 not a jpeg')
    with pytest.raises(JpegUnsupported, match='progressive'):
        JpegBatch([prog], (1,)).pack()
    with pytest.raises(_lib.GrlHipError, match='SOI'):
        JpegBatch([b'\x00\x01 not a jpeg at all'], (1,)).pack()
    with pytest.raises(_lib.GrlHipError):                                   # truncated inside the headers
        JpegBatch([g_bytes(golden)[:200]], (1,)).pack()
    from PIL import Image                                                   # iLIDS-VID / PRID frames are PNG files
    png = io.BytesIO()
    Image.fromarray(np.zeros((16, 8, 3), np.uint8)).save(png, format='PNG')
    with pytest.raises(JpegUnsupported, match="PNG.*decode='host'"):
        JpegBatch([g_bytes(golden), png.getvalue()], (2,)).pack()


def g_bytes(golden):
    return golden('jpeg_frames.npz')['jpeg.mars_420_q90'].tobytes()


def test_host_header_parser_fills_the_frame_descriptor(golden):
    """grl_jpeg_parse (host code of libgrl_hip.so, no GPU call): geometry, sampling, table selectors, restart interval,
    quantisation tables in natural order and canonical Huffman tables for every fixture stream; offsets are relative to
    the batch buffer."""
    from grl_amd import _lib
    from grl_amd.reid.data.jpeg import JpegBatch
    g = golden('jpeg_frames.npz')
    names = [k[5:] for k in g.files if k.startswith('jpeg.')]
    streams = [g['jpeg.' + nm].tobytes() for nm in names]
    frames = []
    for s in streams:                       # one geometry per pack() is a decode-time rule; parsing is per stream
        frames.append(JpegBatch([s], (1,)).pack()[1][0])
    for nm, s, f in zip(names, streams, frames):
        h, w = g['rgb.' + nm].shape[:2]
        assert (f.height, f.width) == (h, w), nm
        assert s[f.scan_off - 2 - (6 + 2 * f.ncomp):f.scan_off][:2] == b'\xff\xda'          # the scan starts right behind SOS
        assert f.scan_off + f.scan_len == len(s)
        assert f.ncomp == (1 if nm.startswith('grey') else 3)
        want = {'mars': (2, 2), 'small_444': (1, 1), 'odd_422': (2, 1), 'odd_420': (2, 2), 'grey': (1, 1)}
        assert (f.hmax, f.vmax) == next(v for k, v in want.items() if nm.startswith(k)), nm
        assert f.restart_interval == (3 if nm.endswith('rst') else 0)
        assert all(1 <= f.q[f.tq[c]][k] <= 255 for c in range(f.ncomp) for k in range(64))
        for t in ([0, 2] if f.ncomp == 1 else [0, 1, 2, 3]):
            mc = list(f.maxcode[t])
            assert mc[17] == 0x7fffffff and any(m >= 0 for m in mc[1:17])
    # offsets inside a batch buffer
    two = JpegBatch([streams[0], streams[1]], (2,))
    host, fr = two.pack()
    assert fr[1].scan_off == len(streams[0]) + frames[1].scan_off and host.numel() % 8 == 0
    assert bytes(host[:len(streams[0])].numpy().tobytes()) == streams[0]
    lib = _lib.load()
    assert lib.grl_jpeg_workspace_bytes(C.byref(fr[0]), 2) >= 2 * (768 * 128 + 256 * 128 * 3 // 2)
    assert lib.grl_jpeg_decode_batch(None, None, fr, 2, None, None, 0, None) == -1             # argument checks need no GPU


def test_host_header_parser_survives_mutated_and_truncated_headers(golden):
    """grl_jpeg_parse on ~6000 damaged HEADERS (random byte edits before the scan, every truncation length of one stream):
    it returns a GRL code -- never reads outside the buffer, never accepts a descriptor the kernels' launch rules do not
    cover.  Every stream ends at an inaccessible guard page, so a read past its end is a fault, not luck."""
    from grl_amd import _lib
    lib = _lib.load()
    g = golden('jpeg_frames.npz')
    streams = [g[k].tobytes() for k in g.files if k.startswith('jpeg.')]
    rng = np.random.default_rng(11)
    fr = _lib.GrlJpegFrame()
    codes = {}

    # a mapping whose last page is inaccessible: a stream is placed so that it ENDS at the guard page
    import mmap
    page = mmap.PAGESIZE
    room = 8 * page
    region = mmap.mmap(-1, room + page)
    base = C.addressof(C.c_char.from_buffer(region))
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    assert libc.mprotect(base + room, page, 0) == 0, C.get_errno()

    def parse(b):
        assert len(b) <= room
        at = room - len(b)
        region[at:room] = b
        rc = lib.grl_jpeg_parse(C.cast(base + at, C.POINTER(C.c_uint8)), len(b), 0, C.byref(fr))
        codes[rc] = codes.get(rc, 0) + 1
        assert rc in (0, _lib.GRL_EINVAL, _lib.GRL_EUNSUPPORTED), rc
        if rc == 0:
            assert fr.scan_off <= len(b) and fr.scan_off + fr.scan_len == len(b)
            assert fr.ncomp in (1, 3) and fr.width and fr.height and fr.hmax in (1, 2) and fr.vmax in (1, 2)
            for c in range(fr.ncomp):
                assert 1 <= fr.hs[c] <= 2 and 1 <= fr.vs[c] <= 2 and fr.tq[c] <= 3 and fr.td[c] <= 1 and fr.ta[c] <= 1
        return rc

    for s in streams:
        assert parse(s) == 0
        head = fr.scan_off
        for _ in range(600):
            b = bytearray(s)
            for _ in range(int(rng.integers(1, 5))):
                b[int(rng.integers(2, head))] = int(rng.integers(0, 256))
            parse(bytes(b))
    s = streams[0]
    assert parse(s) == 0
    for cut in range(4, fr.scan_off + 8):
        parse(s[:cut])
    assert codes.get(0, 0) > 100 and codes.get(_lib.GRL_EINVAL, 0) > 100 and codes.get(_lib.GRL_EUNSUPPORTED, 0) > 10, codes


def test_device_entropy_core_on_the_cpu_matches_the_oracle_coefficients(tmp_path):
    """grl_amd/csrc/jpeg_core.h is the per-lane logic of jpeg_entropy_kernel and compiles as plain C++
    (tests/jpeg_core_host.cpp, g++): the quantised coefficients it leaves equal the oracle's for every stream of a sweep
    (look-ahead tables + long-code fallback, the refill paths at all four alignments of the scan inside the batch buffer,
    0xFF00 stuffing, restart markers, optimised tables) -- through the general reader and through the clean reader fed by
    the unstuffing rule the pre-pass kernel applies."""
    import os
    import subprocess
    pytest.importorskip('PIL')
    from grl_amd.reid.data.jpeg import JpegBatch
    from oracle.ref_c import jpeg_coefficients
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(tmp_path, 'libgjhost.so')
    subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', os.path.join(root, 'tests', 'jpeg_core_host.cpp'), '-o', so])
    lib = C.CDLL(so)
    for fn in (lib.gj_host_decode, lib.gj_host_decode_clean):       # the general reader / the clean reader behind the unstuffing rule
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(0)
    streams = []
    for (h, w) in [(256, 128), (17, 33), (1, 1), (31, 2), (64, 48), (100, 77)]:
        for sub in (0, 1, 2):
            for q, kw in ((30, {}), (75, dict(optimize=True)), (100, {}), (60, dict(restart_marker_blocks=2)),
                          (85, dict(restart_marker_rows=1))):
                try:
                    streams.append(_encode(_frame(h, w, rng), quality=q, subsampling=sub, **kw))
                except OSError:
                    pass
            streams.append(_encode(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), quality=100, subsampling=sub))
    assert len(streams) > 90
    for s in streams:
        ref = jpeg_coefficients(s)
        for lead in (0, 1, 2, 3):
            host, fr = JpegBatch([s], (1,)).pack()
            buf = np.ascontiguousarray(np.concatenate([np.full(lead, 0xFF, np.uint8), host.numpy(), np.zeros(8, np.uint8)]))
            f = fr[0]
            f.scan_off += lead
            for fn in ((lib.gj_host_decode,) if f.restart_interval else (lib.gj_host_decode, lib.gj_host_decode_clean)):
                out = np.full_like(ref, 7)              # (every block is written whole: no zero-fill needed)
                fn(buf.ctypes.data, len(buf) - 8, C.addressof(f), out.ctypes.data)
                assert np.array_equal(out, ref), (len(s), lead, fn)


def test_table_sets_are_shared_or_per_frame():
    """grl_jpeg_assign_tables: frames with Pillow's default tables share ONE set; every optimised frame has its own; more
    than eight distinct sets switch the batch to per-frame tables."""
    pytest.importorskip('PIL')
    from grl_amd.reid.data.jpeg import JpegBatch
    rng = np.random.default_rng(2)
    same = [_encode(_frame(32, 32, rng), quality=q) for q in (40, 60, 80, 95)]
    _, fr = JpegBatch(same, (4,)).pack()
    assert [f.tabset for f in fr] == [0, 0, 0, 0]
    mixed = same[:2] + [_encode(_frame(32, 32, rng), quality=70, optimize=True) for _ in range(3)]
    _, fr = JpegBatch(mixed, (5,)).pack()
    assert [f.tabset for f in fr][:2] == [0, 0] and len({f.tabset for f in fr}) == 4
    many = [_encode(_frame(32, 32, rng), quality=70, optimize=True) for _ in range(12)]
    _, fr = JpegBatch(many, (12,)).pack()
    assert [f.tabset for f in fr] == list(range(12))


def _damaged_streams(rng, count):
    """truncated inside the scan / bytes overwritten / a stray 0xFF xx inserted: what a decoder must survive"""
    from grl_amd.reid.data.jpeg import JpegBatch
    out = []
    for it in range(count):
        h, w = int(rng.integers(8, 80)), int(rng.integers(8, 80))
        s = bytearray(_encode(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), quality=int(rng.integers(20, 100)),
                              subsampling=int(rng.integers(0, 3))))
        so = JpegBatch([bytes(s)], (1,)).pack()[1][0].scan_off
        kind = it % 3
        if kind == 0:
            s = s[:so + int(rng.integers(0, len(s) - so))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 6))):
                s[so + int(rng.integers(0, len(s) - so - 2))] = int(rng.integers(0, 256))
        else:
            p = so + int(rng.integers(0, len(s) - so - 2))
            s[p:p] = bytes([0xFF, int(rng.choice([0x00, 0xFF, 0xD0, 0xD9, 0x55]))])
        out.append(bytes(s))
    return out


def test_damaged_streams_decode_the_same_way_everywhere(tmp_path):
    """Truncated, corrupted and marker-riddled scans: the device core (both readers, CPU build) leaves the coefficients
    the oracle leaves -- zero bits past the data, symbol 0 for an impossible code, decoding stops feeding at a marker
    (libjpeg's documented behaviour for damaged data).  No crash, no read outside the scan (the build in /tmp is also run
    under ASan / UBSan by hand; here it is the plain build)."""
    import os
    import subprocess
    pytest.importorskip('PIL')
    from grl_amd.reid.data.jpeg import JpegBatch
    from oracle.ref_c import jpeg_coefficients
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(tmp_path, 'libgjhost.so')
    subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', os.path.join(root, 'tests', 'jpeg_core_host.cpp'), '-o', so])
    lib = C.CDLL(so)
    for fn in (lib.gj_host_decode, lib.gj_host_decode_clean):
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    n = 0
    for s in _damaged_streams(np.random.default_rng(5), 150):
        try:
            ref = jpeg_coefficients(s)
        except ValueError:
            continue
        host, fr = JpegBatch([s], (1,)).pack()
        buf = np.ascontiguousarray(np.concatenate([host.numpy(), np.zeros(8, np.uint8)]))
        for fn in (lib.gj_host_decode, lib.gj_host_decode_clean):
            out = np.full_like(ref, 7)
            fn(buf.ctypes.data, len(buf) - 8, C.addressof(fr[0]), out.ctypes.data)
            assert np.array_equal(out, ref), len(s)
            n += 1
    assert n > 200


def test_read_file_refuses_a_truncated_frame_like_pillow(tmp_path):
    """video_loader.py:124-141 raises OSError on a truncated file (Pillow does not load truncated images unless told to);
    the device loader's read_file raises too, and accepts what Pillow accepts: bytes behind the EOI marker, an embedded
    thumbnail with its own EOI in front of the main image."""
    import os
    from PIL import Image
    from grl_amd.reid.data.jpeg import read_file
    rng = np.random.default_rng(2)
    base = _encode(_frame(64, 32, rng), quality=85)
    thumb = _encode(_frame(16, 8, rng), quality=60)
    app1 = b'\xff\xe1' + (len(thumb) + 8).to_bytes(2, 'big') + b'Exif\0\0' + thumb
    with_thumb = base[:2] + app1 + base[2:]
    cases = {'whole': (base, True), 'tail': (base + b'\x00\x01garbage' * 9, True), 'thumb': (with_thumb, True),
             'no_eoi': (base[:-2], False), 'cut': (base[:len(base) * 2 // 3], False),
             'thumb_cut': (with_thumb[:len(with_thumb) - len(base) // 3], False)}
    for name, (data, ok) in cases.items():
        path = os.path.join(tmp_path, name + '.jpg')
        with open(path, 'wb') as fh:
            fh.write(data)
        try:
            Image.open(path).convert('RGB')
            pil_ok = True
        except OSError:
            pil_ok = False
        assert pil_ok == ok, name
        if ok:
            assert read_file(path) == data
        else:
            with pytest.raises(OSError, match='truncated'):
                read_file(path)


def test_compressed_batches_cross_loader_worker_processes(tmp_path):
    """INTEGRATION.md's loader: RawVideoDataset(decode='device') under DataLoader(num_workers=2, collate_fn=jpeg_collate)
    -- the workers read files, JpegBatch objects are pickled back, the headers parse in the parent (no GPU call)."""
    import os
    from PIL import Image
    from torch.utils.data import DataLoader
    from grl_amd.reid.data import RawVideoDataset
    from grl_amd.reid.data.jpeg import JpegBatch, jpeg_collate
    rng = np.random.default_rng(0)
    tracklets = []
    for t in range(6):
        paths = []
        for f in range(5):
            p = os.path.join(tmp_path, 't%d_%d.jpg' % (t, f))
            Image.fromarray(_frame(64, 32, rng)).save(p, format='JPEG', quality=80)
            paths.append(p)
        tracklets.append((paths, t, t % 2))
    loader = DataLoader(RawVideoDataset(tracklets, seq_len=4, sample='rrs_test', decode='device'), batch_size=2,
                        collate_fn=jpeg_collate, num_workers=2)
    seen = []
    for jb, pids, cams in loader:
        assert isinstance(jb, JpegBatch) and jb.shape == (2, 4)
        _, fr = jb.pack()
        assert all((f.width, f.height) == (32, 64) for f in fr)
        seen += pids.tolist()
    assert seen == list(range(6))


def test_compressed_batches_shard_at_pair_granularity():
    """dist.PairShardedBatches slices a global batch along the clip dimension; a JpegBatch slices the same way (the rank
    then uploads and decodes only its own pairs)."""
    pytest.importorskip('PIL')
    import torch
    from grl_amd import dist as D
    from grl_amd.reid.data.jpeg import JpegBatch
    rng = np.random.default_rng(4)
    streams = [_encode(_frame(16, 16, rng), quality=70) + bytes([i]) for i in range(8 * 2)]      # 8 clips x 2 frames, tagged
    jb = JpegBatch(streams, (8, 2))
    assert len(jb) == 8 and jb[2:6].shape == (4, 2) and jb[2:6].streams == streams[4:12]
    with pytest.raises(TypeError):
        jb[0]
    pids = torch.arange(8) // 2
    cams = torch.arange(8) % 2
    got = []
    for r in range(2):
        for b, p, c in D.PairShardedBatches([(jb, pids, cams)], rank=r, world=2):
            assert isinstance(b, JpegBatch) and b.shape == (4, 2) and len(p) == 4 and int(p[0]) == int(p[1])
            got += b.streams
    assert got == streams


def test_parallel_entropy_decoder_emulated_on_the_cpu_matches_the_oracle(tmp_path):
    """jpeg_par.h (the workgroup-per-frame form: self-synchronising subsequences, block-count scan, DC prefix sums) with its
    lanes emulated one after the other (tests/jpeg_core_host.cpp: gj_host_decode_par): the oracle's coefficients for intact
    and damaged streams, for the production geometry (256 lanes, 1024-bit subsequences) and for geometries that stress the
    synchronisation (32-bit subsequences, 1 / 7 / 64 lanes)."""
    import os
    import subprocess
    pytest.importorskip('PIL')
    from grl_amd.reid.data.jpeg import JpegBatch
    from oracle.ref_c import jpeg_coefficients
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(tmp_path, 'libgjhost.so')
    subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', os.path.join(root, 'tests', 'jpeg_core_host.cpp'), '-o', so])
    lib = C.CDLL(so)
    lib.gj_host_decode_par.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.gj_host_decode_par.restype = C.c_int
    rng = np.random.default_rng(0)
    streams = []
    for (h, w) in [(256, 128), (17, 33), (1, 1), (31, 2), (64, 48), (100, 77)]:
        for sub in (0, 1, 2):
            for q, kw in ((30, {}), (75, dict(optimize=True)), (100, {})):
                try:
                    streams.append(_encode(_frame(h, w, rng), quality=q, subsampling=sub, **kw))
                except OSError:
                    pass
    streams.append(_encode(_frame(48, 40, rng, grey=True), quality=80))
    streams += _damaged_streams(np.random.default_rng(5), 60)
    n, rounds = 0, []
    for s in streams:
        try:
            ref = jpeg_coefficients(s)
            host, fr = JpegBatch([s], (1,)).pack()
        except Exception:                      # noqa: BLE001 (a damaged header either side refuses)
            continue
        if fr[0].restart_interval:
            continue
        buf = np.ascontiguousarray(np.concatenate([host.numpy(), np.zeros(8, np.uint8)]))
        for lanes, minb in ((256, 1024), (256, 64), (7, 32), (1, 32), (64, 32)):
            out = np.full_like(ref, 7)
            r = lib.gj_host_decode_par(buf.ctypes.data, len(buf) - 8, C.addressof(fr[0]), out.ctypes.data, lanes, minb)
            assert np.array_equal(out, ref), (len(s), lanes, minb, r)
            if (lanes, minb) == (256, 1024) and len(s) > 8000:
                rounds.append(r)
            n += 1
    assert n > 400 and rounds and max(rounds) < 256           # (4-6 rounds for quality-90 MARS frames; quality 100 needs up to ~40)
