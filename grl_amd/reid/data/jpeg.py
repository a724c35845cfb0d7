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
    with open(path, 'rb') as fh:
        return fh.read()


class JpegBatch(object):
    """Compressed frames of one batch: ``streams`` (bytes objects, row-major over ``shape``, e.g. (B, T))."""

    def __init__(self, streams, shape):
        self.streams, self.shape = list(streams), tuple(shape)
        n = 1
        for s in self.shape:
            n *= s
        if n != len(self.streams):
            raise ValueError('JpegBatch: %d streams for shape %s' % (len(self.streams), self.shape))

    def pack(self):
        """-> (uint8 host tensor of the concatenated streams padded to a multiple of 8 bytes, (GrlJpegFrame * n) parsed)"""
        lib = _lib.load()
        n = len(self.streams)
        raw = b''.join(self.streams)
        buf = np.zeros((len(raw) + 15) // 8 * 8, np.uint8)
        buf[:len(raw)] = np.frombuffer(raw, np.uint8)
        offs = np.zeros(n + 1, np.int64)
        np.cumsum([len(s) for s in self.streams], out=offs[1:])
        frames = (GrlJpegFrame * n)()
        bad = C.c_int(-1)
        rc = lib.grl_jpeg_parse_batch(buf.ctypes.data, offs.ctypes.data, n, frames, C.addressof(bad))   # headers + table sets: one call
        if rc:
            msg = lib.grl_last_error().decode('utf-8', 'replace')
            raise (JpegUnsupported if rc == _lib.GRL_EUNSUPPORTED else GrlHipError)('frame %d: %s' % (bad.value, msg))
        return torch.from_numpy(buf), frames


_ws_cache = {}


def decode_jpeg_batch(batch, device, stream=None):
    """JpegBatch (or a list of bytes) -> uint8 tensor ``batch.shape + (3, H, W)`` on ``device`` (planar RGB, what
    np.asarray(Image.open(f).convert('RGB')).transpose(2, 0, 1) holds for every frame).  All frames must share one
    geometry (size, components, sampling).  Work is enqueued on torch's current stream."""
    if not isinstance(batch, JpegBatch):
        batch = JpegBatch(batch, (len(batch),))
    device = torch.device(device)
    if device.type != 'cuda':
        raise GrlHipError('decode_jpeg_batch executes on MI355X only (got device %s); grl_amd has no CPU path' % device)
    lib = _lib.load()
    host, frames = batch.pack()
    n = len(batch.streams)
    f0 = frames[0]
    H, W = int(f0.height), int(f0.width)
    host = host.pin_memory()
    dbytes = host.to(device, non_blocking=True)
    fbytes = torch.frombuffer(bytearray(bytes(frames)), dtype=torch.uint8).pin_memory()
    dframes = fbytes.to(device, non_blocking=True)
    need = int(lib.grl_jpeg_workspace_bytes(C.byref(f0), n))
    ws = torch.empty(need, dtype=torch.uint8, device=device)
    out = torch.empty(batch.shape + (3, H, W), dtype=torch.uint8, device=device)
    rc = lib.grl_jpeg_decode_batch(dbytes.data_ptr(), dframes.data_ptr(), frames, n, out.data_ptr(), ws.data_ptr(), need,
                                   _lib.stream())
    _lib.check(rc, 'grl_jpeg_decode_batch')
    # the pinned staging buffers must outlive the asynchronous copies: tie them to the output
    out._grl_keepalive = (host, fbytes, dbytes, dframes, ws)
    return out


def jpeg_collate(items):
    """collate_fn for loaders whose dataset yields (list of T byte strings, pid, camid[, params]):
    -> (JpegBatch of shape (B, T), pids, camids[, params]) -- the batch is decoded by engine.DevicePrefetcher."""
    clips = [it[0] for it in items]
    t = len(clips[0])
    streams = [s for c in clips for s in c]
    out = [JpegBatch(streams, (len(clips), t)), torch.as_tensor([it[1] for it in items]), torch.as_tensor([it[2] for it in items])]
    for k in range(3, len(items[0])):
        out.append(torch.stack([torch.as_tensor(it[k]) for it in items]))
    return tuple(out)
