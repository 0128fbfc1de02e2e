"""Caller-side pipelining of a SEQUENCE of independent batches over a few HIP streams.

One voting call is a chain of kernels with very different limits: the mask scan is HBM-bound, the compaction is a
queue of latency-bound blocks, the inlier count is VALU-bound.  Issued back to back on one stream they run one after
the other; issued for consecutive batches on alternating streams, the scan and compaction of batch i+1 run under the
count kernel of batch i: +12-18 % images/s at B = 64 on one MI355X (``tools/two_stream.py``; ``bench.py`` reports it
as ``extra.two_stream_images_per_s``).  The library itself keeps no state and never creates streams,
so this is a few lines of stream bookkeeping on the caller's side -- ``StreamRing`` is those lines.

    ring = StreamRing(2)
    for mask, vertex in batches:                        # produced on the current stream
        kpt = ring.run(ransac_voting_layer_v3, mask, vertex, 512, inlier_thresh=0.99)
        results.append(kpt)                             # not yet ordered against the current stream
    ring.join()                                         # now the current stream sees every result

The bookkeeping costs ~10-15 us of host time per run() (an event, two stream switches), so it pays for calls that keep
the GPU busy longer than that -- B >= 32 at 480x640; a shard of 8 images (64 us per call) is host-bound either way
(``tools/ring_streams.py``: 224 -> 273 k images/s at B = 64 with two streams, three or four are not better; 92 -> 123 k at
B = 8 against 128 k for plain calls on one stream).

The reference has no counterpart (its evaluation loop is one image at a time on the default stream,
lib/evaluators/linemod/pvnet.py:166-186).
"""
import torch


class StreamRing:
    """Round-robin over ``n`` side streams.  ``run(fn, *args, **kw)`` enqueues ``fn`` on the next side stream, ordered
    after everything already enqueued on the CURRENT stream (so inputs produced there are ready) and returns what ``fn``
    returns; ``join()`` orders the current stream after all side streams and tells the caching allocator that the
    returned tensors are used on it."""

    def __init__(self, n=2, device=None):
        if n < 1:
            raise ValueError("StreamRing needs at least one stream")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n)]
        self._next = 0
        self._pending = []          # (stream, tensors returned by fn) since the last join()

    @staticmethod
    def _tensors(x):
        if isinstance(x, torch.Tensor):
            return [x]
        if isinstance(x, (tuple, list)):
            return [t for v in x for t in StreamRing._tensors(v)]
        if isinstance(x, dict):
            return [t for v in x.values() for t in StreamRing._tensors(v)]
        return []

    def run(self, fn, *args, **kw):
        side = self.streams[self._next]
        self._next = (self._next + 1) % len(self.streams)
        side.wait_stream(torch.cuda.current_stream(self.device))
        # the inputs were allocated on the current stream and are read on `side`: keep their memory until `side` is done
        for t in self._tensors(args) + self._tensors(kw):
            if t.is_cuda:
                t.record_stream(side)
        with torch.cuda.stream(side):
            out = fn(*args, **kw)
        self._pending.append((side, self._tensors(out)))
        return out

    def join(self):
        cur = torch.cuda.current_stream(self.device)
        for side, outs in self._pending:
            cur.wait_stream(side)
            for t in outs:
                if t.is_cuda:
                    t.record_stream(cur)
        self._pending = []
