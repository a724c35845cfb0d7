"""Failure visibility for the ranks of ONE node (bench.py's N > 1 blocks): no torch import, no GPU call.

The ranks of a launch tell each other through two files in the temp directory: `.err` = some rank's block raised or
timed out (its text is the reason; the FIRST failure's text wins and is what every rank reports), `.out` = rank 0
has printed the line.  A watchdog thread per rank polls them, so a healthy rank that sits in a collective whose peer
has failed leaves within a second instead of waiting for the time limit -- but only after rank 0 has printed.

Round 6: nothing is deleted when a guard starts.  Rank 0 used to remove `.err` / `.out` in
start(); a peer that failed BEFORE rank 0 got there lost its reason and rank 0 reported the time-out text instead.
Uniqueness comes from the key (rendezvous port + elastic run id + per-launch nonce + tag); a file that is older
than the launcher process itself belongs to an earlier launch and is ignored; the reason is published atomically
(temp file + hard link: the name appears with its full text or not at all, and only the first writer creates it).
"""
import contextlib
import os
import sys
import tempfile
import threading
import time

EXIT_TRAIN_BLOCK_FAILED = 3


def _launcher_start_time():
    """Creation time of the parent (torch.distributed.run / the test harness): every file of THIS launch is younger."""
    try:
        import psutil
        return psutil.Process(os.getppid()).create_time()
    except Exception:                                   # noqa: BLE001 (no psutil / parent gone: no staleness filter)
        return 0.0


class TrainBlockGuard(object):
    """N > 1: a guarded block is the only part of the bench line with collectives in it.  The headline must not
    depend on them, and a hung or failed all-reduce step must be VISIBLE in the launcher's exit code: rank 0 prints
    the line with `{"error": ...}` and every rank then leaves with EXIT_TRAIN_BLOCK_FAILED (never 0, never a re-exec:
    these processes have touched the GPU; a retry, if any, is a fresh `bench.py --gpus N` from the GPU-less parent)."""

    def __init__(self, rank, world, limit, emit_error_line, tag='train', directory=None, key=None, exit_fn=None):
        self.rank, self.world, self.limit, self.emit_error_line = rank, world, limit, emit_error_line
        # (GRL_BENCH_NONCE: set per launch by bench.launch_ranks; bare torchrun launches fall back to the launcher's
        #  pid, which its ranks share)
        if key is None:
            key = 'grl_bench_%s_%s_%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'x'),
                                             os.environ.get('GRL_BENCH_NONCE') or os.getppid(), tag)
        directory = directory or tempfile.gettempdir()
        self.err = os.path.join(directory, key + '.err')
        self.out = os.path.join(directory, key + '.out')
        self.not_before = _launcher_start_time() - 1.0
        self.done = threading.Event()
        self.lock = threading.Lock()
        self.thread = None
        self.exit_fn = exit_fn or os._exit

    # -- files ---------------------------------------------------------------------------------------------------
    def _fresh(self, path):
        try:
            return os.stat(path).st_mtime >= self.not_before
        except OSError:
            return False

    def _publish(self, reason):
        """Create `.err` with `reason` unless a fresh one exists; return the text that stands (the first failure's)."""
        if os.path.exists(self.err) and not self._fresh(self.err):
            with contextlib.suppress(OSError):          # a leftover of an earlier launch with the same key
                os.remove(self.err)
        tmp = '%s.%d.%d' % (self.err, os.getpid(), self.rank)
        try:
            with open(tmp, 'w') as fh:
                fh.write(reason)
            try:
                os.link(tmp, self.err)                  # atomic, fails if a peer was first
            except OSError:
                pass
        except OSError:
            pass
        finally:
            with contextlib.suppress(OSError):
                os.remove(tmp)
        try:
            first = open(self.err).read()
        except OSError:
            first = ''
        if first and first != reason:
            return '%s | then on rank %d: %s' % (first, self.rank, reason)
        return reason

    # -- protocol ------------------------------------------------------------------------------------------------
    def start(self):
        if self.world > 1:
            self.thread = threading.Thread(target=self._watch, daemon=True)
            self.thread.start()

    def _watch(self):
        t0 = time.time()
        while not self.done.wait(0.25):
            if self._fresh(self.err):
                with contextlib.suppress(OSError):
                    self.leave(open(self.err).read() or 'a rank failed')
            if time.time() - t0 > self.limit:
                self.leave("train block (RCCL all-reduce step) did not finish within %.0f s on rank %d" % (self.limit, self.rank))

    def leave(self, reason):
        """Publish the reason, let rank 0 print the line, exit non-zero.  Called from the watchdog thread (peer
        failed / time limit) or from the main thread (this rank's block raised)."""
        with self.lock:
            if self.done.is_set():
                return
            reason = self._publish(reason)
            if self.rank == 0:
                self.emit_error_line(reason)
                with contextlib.suppress(OSError):
                    open(self.out, 'w').close()
            else:
                t0 = time.time()
                while not self._fresh(self.out) and time.time() - t0 < 20:
                    time.sleep(0.1)
            sys.stderr.write('bench.py rank %d: %s -- exit %d\n' % (self.rank, reason, EXIT_TRAIN_BLOCK_FAILED))
            sys.stderr.flush()
            self.exit_fn(EXIT_TRAIN_BLOCK_FAILED)

    def finished(self):
        self.done.set()
