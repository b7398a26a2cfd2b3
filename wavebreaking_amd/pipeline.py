"""Several batches in flight on one GPU.

A context (``_hip.Context``) is one HIP stream driven by one host thread, which waits for the device twice per batch (the
summary of phase 1, the result tables of phase 2); most of its kernels are latency-bound per-(step, level) jobs with a
tail.  Two contexts on the same device, each fed by its own thread, fill those waits and tails with the other batch's
kernels (measured on MI355X, 0.25 degree, 1460 steps per batch: 5.6 ms per batch alone, 4.7 ms with two in flight).

Selection maximum.  The reference selects contours with ``exp_lon == exp_lon.max()`` over the WHOLE frame it is given
(streamer_index.py:106).  Here every batch is a frame of its own: batch ``k`` is selected against the maximum over
batch ``k`` on all ranks.  A record cut into batches therefore gives the single-call catalogue exactly when every batch
holds a contour of the largest extent -- always, when each block of time steps has a circumglobal contour (extent = the
width of the periodically extended grid, which no contour can exceed).  ``run(..., full_width=W)`` checks that: a batch
whose reduced maximum is smaller than ``W`` raises instead of silently selecting against a smaller maximum.

``BatchPipeline`` keeps ``depth`` contexts, each with a single worker thread (so a context is never entered by two
threads at once), and runs a sequence of batches through the two phases of the ABI.  Whatever must happen in program
order on ONE thread -- the all-reduce of the selection maximum between the phases and the gather after the second, in
a multi-GPU run -- is done by the caller's thread, batch after batch, so every rank issues its collectives in the same
order no matter how the workers interleave.
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

from wavebreaking_amd import _hip


class FullWidthNotReached(RuntimeError):
    """BatchPipeline.run_record: the record holds no contour of the full width it was selected against"""


class PipelineTimeout(RuntimeError):
    """a worker did not come back within WB_PIPE_TIMEOUT seconds (default 600): a batch is stuck on the device or in an
    exchange.  The waiting thread raises instead of waiting for ever; the stuck worker thread cannot be interrupted, so a
    program that catches this should end (bench.py exits non-zero)."""


def _timeout():
    t = float(os.environ.get("WB_PIPE_TIMEOUT", "600"))
    return t if t > 0 else None


def _result(fut, what):
    from concurrent.futures import TimeoutError as _FT
    try:
        return fut.result(timeout=_timeout())
    except _FT:
        raise PipelineTimeout("BatchPipeline: %s did not finish within %s s (WB_PIPE_TIMEOUT)" % (what, os.environ.get("WB_PIPE_TIMEOUT", "600")))


class BatchPipeline:
    def __init__(self, device=0, depth=2, library=None, stagger=None):
        self.contexts = [_hip.Context(device, library) for _ in range(max(1, int(depth)))]
        self._workers = [ThreadPoolExecutor(max_workers=1) for _ in self.contexts]
        # stagger (set_stagger): which of the two phases of the ABI are entered by one batch at a time.  Default 2: phase 1 (the HBM-bound scan + the trace kernels) is entered by one batch at a time.  Left free, two batches
        # that happen to fall into step -- the same kernel of both at the same time, each on half the device -- stay in step
        # (one run in ten, more often the longer the record: 300-320 k timesteps/s instead of 335-345 k on the bench batch); with
        # phase 1 exclusive they cannot: 338-345 k in every run (profiles/r04_two_batches_in_step.txt).  Phase 2 exclusive, or both, cost 5-8 %.
        if stagger is None:
            stagger = int(os.environ.get("WB_PIPE_STAGGER", "2"))
        self.default_stagger = int(stagger)
        self.set_stagger(self.default_stagger)
        self.offset_start = os.environ.get("WB_PIPE_OFFSET_START", "1") != "0"

    def set_stagger(self, mode):
        """0: the batches' phases overlap freely; 1: each phase is entered by one batch at a time -- one batch's scan + trace
        kernels then run beside the other's index kernels, never beside their twins (the scan keeps ~0.62 of the HBM peak
        instead of ~0.5, the whole job loses 5-8 %); 2 (the default) / 3: only phase 1 / only phase 2 exclusive"""
        mode = int(mode)
        self._locks = (threading.Lock() if mode in (1, 2) else None, threading.Lock() if mode in (1, 3) else None) if mode else None

    @property
    def depth(self):
        return len(self.contexts)

    def close(self):
        for w in self._workers:
            w.shutdown(wait=True)
        for c in self.contexts:
            c.close()

    def close_batch(self, k, batch):
        """free batch k, on the calling thread: a context's allocators have a lock of their own since round 6 (wb_free_batch may
        come from another thread than the one that drives the context), so the worker of context k % depth is not interrupted"""
        batch.close()

    def run_record(self, n, contours, reduce_max, indices, collect, full_width, before_reduce=None):
        """``n`` batches of ONE record, selected against ``full_width`` WITHOUT waiting for an exchange: no contour can exceed
        the width of the periodically extended grid, so if any batch of any rank holds a contour that spans it -- always, with
        a circumglobal dynamical tropopause -- that IS ``contours.exp_lon.max()`` over the whole frame (streamer_index.py:106)
        and every batch can go from phase 1 straight into phase 2 on its worker.  The assumption is verified by the ONE
        ``reduce_max`` of the record -- the largest local maximum of this rank's batches, after the last batch -- and
        ``FullWidthNotReached`` is raised if it fails (the caller then takes ``run``, which exchanges before every
        selection).  ``collect(k, result)`` on the caller's thread in batch order, as in ``run``.  ``before_reduce()`` is
        called after the last ``collect`` and before the exchange: a run whose gathers are issued by another thread
        (distributed.TableGatherer) waits for them there -- every rank must issue its collectives in ONE order, all gathers of
        the record, then the all-reduce."""
        import queue
        d = self.depth

        # The workers schedule THEMSELVES: worker j runs batches j, j + d, j + 2 d, ... one after the other -- phase 1, phase 2, the
        # next batch's phase 1 -- and hands every finished batch to this thread through a queue; nothing this thread does (reading
        # a batch's tables, waiting for its geometry, freeing it: `collect`) stands between two passes of a worker.  (Until round 6
        # this thread submitted batch k + d to the worker when it had received batch k, and the worker read the tables and freed
        # the batch before it returned: 0.47 ms of host work between a pass's last kernel and the next pass's first.)  A worker
        # runs at most `ahead` batches in front of what has been collected (their device scratch is alive until then).
        # The FIRST batch of worker j waits until worker j - 1 has finished phase 1 of its first batch, so that the contexts start
        # out of step -- one batch's HBM-bound scan beside the other's index kernels rather than beside its twin -- and, taking
        # equally long per batch, stay so (option offset_start, default on)
        started = [threading.Event() for _ in range(d)]
        stop = threading.Event()
        ahead = max(1, int(os.environ.get("WB_PIPE_AHEAD", "2")))
        room = [threading.Semaphore(ahead) for _ in range(d)]
        done = [queue.Queue() for _ in range(d)]

        l1, l2 = self._locks if self._locks is not None else (None, None)

        def both(ctx, k):
            if self.offset_start and 0 < k < d:
                if not started[k - 1].wait(_timeout()):
                    raise PipelineTimeout("BatchPipeline: batch %d waited for batch %d to enter phase 1 for more than WB_PIPE_TIMEOUT s" % (k, k - 1))
            try:
                if l1 is not None:                  # set_stagger: a phase is entered by one batch at a time
                    with l1:
                        b = contours(ctx, k)
                else:
                    b = contours(ctx, k)
            finally:
                if k < d:
                    started[k].set()                # (also when phase 1 raised: the next worker must not wait for ever)
            if l2 is not None:
                with l2:
                    return b.max_nux(), indices(ctx, k, b, full_width)
            return b.max_nux(), indices(ctx, k, b, full_width)

        def loop(j):
            for k in range(j, n, d):
                room[j].acquire()
                if stop.is_set():
                    return
                try:
                    done[j].put((k, both(self.contexts[j], k), None))
                except BaseException as e:          # handed to the caller's thread, which raises it in batch order
                    done[j].put((k, None, e))
                    return
        futs = [self._workers[j].submit(loop, j) for j in range(min(d, n))]
        out, local = [], 0
        try:
            for k in range(n):
                try:
                    kk, mr, err = done[k % d].get(timeout=_timeout())
                except queue.Empty:
                    raise PipelineTimeout("BatchPipeline: batch %d did not finish within %s s (WB_PIPE_TIMEOUT)" % (k, os.environ.get("WB_PIPE_TIMEOUT", "600")))
                if err is not None:
                    raise err
                assert kk == k
                m, r = mr
                local = max(local, int(m))
                out.append(collect(k, r))
                room[k % d].release()
        except BaseException:
            stop.set()                              # the workers end behind the batch they are in
            for sem in room:
                sem.release()
            raise
        for f in futs:
            _result(f, "a worker of the record")
        if before_reduce is not None:
            before_reduce()
        g = reduce_max(local)
        if g != full_width:
            raise FullWidthNotReached("no contour of this record spans the %d columns of the extended grid (largest: %d): the "
                                      "selection maximum has to be exchanged before the index phase" % (full_width, g))
        return out

    def run(self, n, contours, reduce_max, indices, collect, full_width=None):
        """``n`` batches.  Worker of context ``k % depth``:  ``b = contours(ctx, k)`` then, once the caller's thread has
        turned the batch's local maximum into ``g = reduce_max(b.max_nux())``, ``r = indices(ctx, k, b, g)`` (which also
        closes the batch).  Caller's thread, in batch order: ``reduce_max`` and ``collect(k, r)``.  Returns the list of
        ``collect`` results.  ``full_width``: the extent every batch's maximum must reach (see the module docstring)."""
        d = self.depth
        if self._locks is not None:
            l1, l2 = self._locks
            raw_contours, raw_indices = contours, indices

            def contours(ctx, k):
                if l1 is None:
                    return raw_contours(ctx, k)
                with l1:
                    return raw_contours(ctx, k)

            def indices(ctx, k, b, g):
                if l2 is None:
                    return raw_indices(ctx, k, b, g)
                with l2:
                    return raw_indices(ctx, k, b, g)
        c_fut, i_fut, out = {}, {}, []
        for k in range(min(d, n)):
            c_fut[k] = self._workers[k % d].submit(contours, self.contexts[k % d], k)
        for k in range(n):
            b = _result(c_fut.pop(k), "phase 1 of batch %d" % k)
            g = reduce_max(b.max_nux())
            if full_width is not None and g != full_width:
                raise RuntimeError("BatchPipeline: batch %d has no contour of extent %d (largest: %d): its selection would "
                                   "differ from a single call over the whole record" % (k, full_width, g))
            i_fut[k] = self._workers[k % d].submit(indices, self.contexts[k % d], k, b, g)
            if k + d < n:
                c_fut[k + d] = self._workers[k % d].submit(contours, self.contexts[k % d], k + d)
            if k >= d - 1:
                j = k - (d - 1)
                out.append(collect(j, _result(i_fut.pop(j), "phase 2 of batch %d" % j)))
        for j in sorted(i_fut):
            out.append(collect(j, _result(i_fut.pop(j), "phase 2 of batch %d" % j)))
        return out
