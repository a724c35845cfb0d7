"""`Logger(path)`: assign it to sys.stdout and every print also lands in a log file
(the call site is mars_train.py:56-66; upstream counterpart utils/logging.py)."""
import os
import sys


class Logger(object):
    """Tee: forwards write/flush to the stream that was sys.stdout at construction and, when a
    path is given, to that file (its directory is created).  Usable as a context manager."""

    def __init__(self, fpath=None):
        self.console = sys.stdout
        self.file = None
        if fpath:
            folder = os.path.dirname(fpath)
            if folder:
                os.makedirs(folder, exist_ok=True)
            self.file = open(fpath, 'w')

    def _sinks(self):
        return [s for s in (self.console, self.file) if s is not None]

    def write(self, msg):
        for s in self._sinks():
            s.write(msg)

    def flush(self):
        for s in self._sinks():
            s.flush()
        if self.file is not None:
            os.fsync(self.file.fileno())       # a killed training run keeps its log

    def close(self):
        f, self.file = self.file, None
        if f is not None:
            f.close()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
