"""Frame decode on the device: the counterpart of `Image.open(img_path).convert('RGB')` in the reference's loaders
(/root/reference/reid/data/video_loader.py:124-141, :91-96, :108-113) through grl_jpeg_parse / grl_jpeg_decode_batch
(include/grl_hip.h, grl_amd/csrc/jpeg.hip) -- bit-identical to Pillow for baseline JPEGs (4:4:4 / 4:2:2 / 4:2:0 / grey).

A loader worker only READS the files; a batch crosses PCIe as compressed bytes (MARS: ~6-13 KB per 98 KB frame) and is
decoded on the prefetch stream next to the previous batch's compute.  There is no host fallback in here: a stream the
device decoder does not cover raises ``JpegUnsupported`` (the caller may then decide to decode that file with Pillow in
its own worker -- ``RawVideoDataset(decode='host')`` is that path).
"""
import ctypes as C

import numpy as np
import torch

from grl_amd import _lib
from grl_amd._lib import GrlHipError, GrlJpegFrame

__all__ = ['JpegUnsupported', 'JpegBatch', 'decode_jpeg_batch', 'jpeg_collate', 'read_file']


class JpegUnsupported(GrlHipError):
    """A valid JPEG outside the device decoder's scope (progressive, arithmetic, CMYK, 12-bit, 4:4:0 ...)."""


def read_file(path):
    """The bytes of one frame file.  A file whose scan is not closed by an EOI marker raises OSError like the reference's
    `Image.open(path).convert('RGB')` does ("image file is truncated": Pillow runs libjpeg with a suspending source,
    which never reaches the end of such an image) -- the device decoder itself follows libjpeg's rule for damaged data
    (zero bits past the end) and would return a partly grey frame without a word."""
    with open(path, 'rb') as fh:
        data = fh.read()
    if data[:2] == b'\xff\xd8' and data.rfind(b'\xff\xd9') < data.rfind(b'\xff\xda'):
        raise OSError('image file is truncated (no EOI marker behind the scan): %s' % path)
    return data


class JpegBatch(object):
    """Compressed frames of one batch: ``streams`` (bytes objects, row-major over ``shape``, e.g. (B, T))."""

    def __init__(self, streams, shape):
        self.streams, self.shape = list(streams), tuple(shape)
        n = 1
        for s in self.shape:
            n *= s
        if n != len(self.streams):
            raise ValueError('JpegBatch: %d streams for shape %s' % (len(self.streams), self.shape))

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        """A slice of the FIRST dimension (clips): what dist.PairShardedBatches takes from a global batch -- with compressed
        frames a rank then uploads and decodes only its own pair shard."""
        if not isinstance(key, slice):
            raise TypeError('JpegBatch supports slicing along the clip dimension only')
        per = 1
        for d in self.shape[1:]:
            per *= d
        idx = range(self.shape[0])[key]
        streams = [s for i in idx for s in self.streams[i * per:(i + 1) * per]]
        return JpegBatch(streams, (len(idx),) + self.shape[1:])

    def pack(self, into=None):
        """-> (uint8 host tensor of the concatenated streams padded to a multiple of 8 bytes, (GrlJpegFrame * n) parsed).
        ``into``: a (pinned) uint8 host tensor to build the byte buffer in (a view of it is returned)."""
        lib = _lib.load()
        n = len(self.streams)
        raw = b''.join(self.streams)
        size = (len(raw) + 15) // 8 * 8
        host = torch.empty(size, dtype=torch.uint8) if into is None or into.numel() < size else into[:size]
        buf = host.numpy()
        buf[:len(raw)] = np.frombuffer(raw, np.uint8)
        buf[len(raw):] = 0
        offs = np.zeros(n + 1, np.int64)
        np.cumsum([len(s) for s in self.streams], out=offs[1:])
        frames = (GrlJpegFrame * n)()
        bad = C.c_int(-1)
        rc = lib.grl_jpeg_parse_batch(buf.ctypes.data, offs.ctypes.data, n, frames, C.addressof(bad))   # headers + table sets: one call
        if rc:
            msg = lib.grl_last_error().decode('utf-8', 'replace')
            if 0 <= bad.value < n and self.streams[bad.value][:8] == b'\x89PNG\r\n\x1a\n':
                # iLIDS-VID / PRID 2011 ship PNG frames (ilidsvidsequence.py:113): inflate is not on the device
                raise JpegUnsupported("frame %d is a PNG file: the device decoder covers baseline JPEG (MARS, DukeMTMC-"
                                      "VideoReID); load this dataset with RawVideoDataset(decode='host')" % bad.value)
            raise (JpegUnsupported if rc == _lib.GRL_EUNSUPPORTED else GrlHipError)('frame %d: %s' % (bad.value, msg))
        return host, frames


class _PinnedRing(object):
    """A few reusable pinned staging buffers.  `tensor.pin_memory()` per batch allocates pinned host memory again and
    again (the caching host allocator cannot hand a block back while its copy is in flight) and hipHostMalloc stalls the
    device: the eval step fed from JPEG bytes ran decode and compute back to back (24.8 ms = 14.5 + 10.3) instead of side
    by side.  (Found later on the tensor path: `pin_memory()` / `copy_` also fan the copy out over torch's OpenMP pool, whose
    spin-wait takes the cores from the issue thread -- engine.DevicePrefetcher; the ring's single-threaded copy avoids both.)
    A slot is reused only after the event recorded behind its last copy has completed."""

    def __init__(self, slots=4):
        self.bufs, self.events, self.i = [None] * slots, [None] * slots, 0

    def get(self, nbytes):
        i = self.i
        self.i = (i + 1) % len(self.bufs)
        if self.events[i] is not None:
            self.events[i].synchronize()
        b = self.bufs[i]
        if b is None or b.numel() < nbytes:
            b = self.bufs[i] = torch.empty(max(int(nbytes * 1.25), 1 << 16), dtype=torch.uint8, pin_memory=True)
        return i, b

    def mark(self, i):
        ev = torch.cuda.Event()
        ev.record()
        self.events[i] = ev


_rings = {}


def _geometry(f):
    return (int(f.width), int(f.height), int(f.ncomp), int(f.hmax), int(f.vmax))


def decode_jpeg_batch(batch, device, size=None):
    """JpegBatch (or a list of bytes) -> uint8 tensor ``batch.shape + (3, H, W)`` on ``device`` (planar RGB, what
    np.asarray(Image.open(f).convert('RGB')).transpose(2, 0, 1) holds for every frame).  Work is enqueued on torch's
    current stream.

    ``size`` = None: all frames must share one geometry (size, components, sampling) -- MARS.  ``size`` = (H, W): a
    batch whose frames differ in geometry (DukeMTMC-VideoReID's crops) is decoded per geometry group and every group is
    brought to H x W with engine.rect_scale_u8 -- the reference's `RectScale` on the opened image (seqtransforms.py:30-47),
    bit-identical to PIL's BILINEAR -- so the result is ``batch.shape + (3, H, W)``; a batch of ONE geometry is returned
    at its native size (the caller's RectScale is then the same single device pass)."""
    if not isinstance(batch, JpegBatch):
        batch = JpegBatch(batch, (len(batch),))
    device = torch.device(device)
    if device.type != 'cuda':
        raise GrlHipError('decode_jpeg_batch executes on MI355X only (got device %s); grl_amd has no CPU path' % device)
    lib = _lib.load()
    n = len(batch.streams)
    key = device.index if device.index is not None else torch.cuda.current_device()
    ring_b, ring_f = _rings.setdefault(key, (_PinnedRing(), _PinnedRing()))
    ib, pinned = ring_b.get(sum(len(s) for s in batch.streams) + 16)
    host, frames = batch.pack(into=pinned)
    if size is not None:
        groups = {}
        for i in range(n):
            groups.setdefault(_geometry(frames[i]), []).append(i)
        if len(groups) > 1:
            from grl_amd import engine
            ring_b.mark(ib)                       # (the slot was not used for a copy; hand it back in order)
            H, W = int(size[0]), int(size[1])
            out = torch.empty((n, 3, H, W), dtype=torch.uint8, device=device)
            for idx in groups.values():
                part = decode_jpeg_batch(JpegBatch([batch.streams[i] for i in idx], (len(idx),)), device)
                out[torch.tensor(idx, device=device)] = engine.rect_scale_u8(part, H, W)
            return out.view(batch.shape + (3, H, W))
    f0 = frames[0]
    H, W = int(f0.height), int(f0.width)
    dbytes = host.to(device, non_blocking=True)
    ring_b.mark(ib)
    fsize = C.sizeof(GrlJpegFrame) * n
    jf, fpin = ring_f.get(fsize)
    C.memmove(fpin.data_ptr(), frames, fsize)
    dframes = fpin[:fsize].to(device, non_blocking=True)
    ring_f.mark(jf)
    need = int(lib.grl_jpeg_workspace_bytes(frames, n))
    ws = torch.empty(need, dtype=torch.uint8, device=device)
    out = torch.empty(batch.shape + (3, H, W), dtype=torch.uint8, device=device)
    rc = lib.grl_jpeg_decode_batch(dbytes.data_ptr(), dframes.data_ptr(), frames, n, out.data_ptr(), ws.data_ptr(), need,
                                   _lib.stream())
    _lib.check(rc, 'grl_jpeg_decode_batch')
    # (dbytes / dframes / ws were allocated under the stream the kernels run on: the caching allocator orders their reuse)
    return out


def jpeg_collate(items):
    """collate_fn for loaders whose dataset yields (list of T byte strings, pid, camid[, params]) -- or, in dense mode,
    (list of n_clips lists of T byte strings, pid, camid) --:
    -> (JpegBatch of shape (B, T) / (B, n_clips, T), pids, camids[, params]); engine.DevicePrefetcher decodes the batch."""
    clips = [it[0] for it in items]
    dense = len(clips[0]) > 0 and isinstance(clips[0][0], (list, tuple))
    if dense:
        n = len(clips[0])
        if any(len(c) != n for c in clips):
            raise ValueError('dense tracklets of one batch must have the same number of clips (the reference uses batch size 1)')
        t = len(clips[0][0])
        streams = [s for c in clips for row in c for s in row]
        shape = (len(clips), n, t)
    else:
        t = len(clips[0])
        streams = [s for c in clips for s in c]
        shape = (len(clips), t)
    out = [JpegBatch(streams, shape), torch.as_tensor([it[1] for it in items]), torch.as_tensor([it[2] for it in items])]
    for k in range(3, len(items[0])):
        out.append(torch.stack([torch.as_tensor(it[k]) for it in items]))
    return tuple(out)
