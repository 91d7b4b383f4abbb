"""hipGraph-captured training step for the launch-bound models.

The VQ-VAE / VAE steps are ~80 launches of 5-20 us kernels each: the GPU waits for Python.  `GraphedTrainStep` captures
`training_step -> backward -> optimizer.step` once over a static input buffer and replays it per batch (the same idiom as the
sampler, src/runtime/sampler.py).  What makes the step capturable: every kernel takes its stream from torch's current stream; no
host sync anywhere in the step (losses are logged as device scalars, upstream gradients are read on the device); Adam's step
count and learning rate live in device memory (`FlatAdam(device_state=True)`); random draws inside the step use torch's
graph-safe device generator.  Not for WGAN-GP: the reference draws its noise from the CPU generator every step.
"""
from __future__ import annotations

import torch


def _new_graph():
    """A CUDAGraph that keeps its captured hipGraph_t readable (node_types) where this torch can."""
    try:
        return torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:
        return torch.cuda.CUDAGraph()


def node_types(g):
    """{node type: count} of a captured graph, None when it cannot be read.  The steps captured here are meant to be kernel nodes only: a
    memset node (hipMemsetAsync in a library call) was seen to take effect out of stream order in replays (round 5)."""
    try:
        import collections
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        names = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord"}
        raw = g.raw_cuda_graph()
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0:
            return None
        arr = (ctypes.c_void_p * n.value)()
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), arr, ctypes.byref(n)) != 0:
            return None
        c = collections.Counter()
        for i in range(n.value):
            t = ctypes.c_int(-1)
            hip.hipGraphNodeGetType(ctypes.c_void_p(arr[i]), ctypes.byref(t))
            c[names.get(t.value, str(t.value))] += 1
        return dict(c)
    except Exception:       # noqa: BLE001
        return None


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_batch, warmup: int = 3):
        self.model, self.opt = model, optimizer
        imgs = example_batch[0]
        self.x = imgs.clone()
        self.rest = tuple(example_batch[1:])
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                               # warm-up outside capture: workspaces, lazy state, autograd buffers
            for i in range(warmup):
                loss = model.training_step((self.x,) + self.rest, i)
                loss.backward()
                optimizer.step()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = _new_graph()
        with torch.cuda.graph(self.graph):
            self.loss = model.training_step((self.x,) + self.rest, 0)
            self.loss.backward()
            optimizer.step()
        self.warmup_steps = warmup

    def __call__(self, batch):
        self.x.copy_(batch[0], non_blocking=True)
        self.graph.replay()
        # the replayed Adam rewrote the flat parameter buffers behind Python's back: anything derived from them that is built
        # OUTSIDE the graph (an eager forward's bf16 weight copies, the sampler's time-bias table) must see a new version
        for n in getattr(self.opt, "nets", ()):
            n.mark_params_dirty()
        return self.loss


class SegmentedGraphedTrainStep:
    """The hipGraph step under data parallelism.  A collective cannot ride inside the captured step portably, and without it the
    whole step cannot be ONE graph: the capture is CUT at every gradient bucket instead.  `model.training_step_and_backward(batch)`
    (forward + tape-replay backward on this thread, no autograd) runs under stream capture; whenever the flat-gradient reducer finds
    a bucket final it calls `_cut(lo, hi)`: the current graph ends there, the next one begins.  The fused Adam is the last graph.
    Replay = graph 0, all-reduce(bucket 0) async on RCCL's stream, graph 1, all-reduce(bucket 1), ..., wait for the collectives,
    Adam graph: ~5 graph launches + ~4 collectives per step instead of ~600 kernel launches, with the same overlap of communication
    and backward as the eager step (bucket k is on the wire while graph k + 1 computes)."""

    def __init__(self, model, optimizer, reducer, example_batch):
        self.model, self.opt, self.red = model, optimizer, reducer
        self.x = example_batch[0].clone()
        self.rest = tuple(example_batch[1:])
        self.segments = []                          # (CUDAGraph, (lo, hi) or None)
        self.pool = torch.cuda.graph_pool_handle()
        self._cur = None
        optimizer.device_state = True
        dev = self.x.device
        self.stream = torch.cuda.Stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            reducer.begin()
            reducer.capture_sink = self._cut
            try:
                self._final = False
                self._begin()
                self.loss = model.training_step_and_backward((self.x,) + self.rest, 0)
                self._final = True                  # the last bucket ends the last graph of backward: no empty graph after it
                reducer.finish()                    # flushes that bucket into the sink (nothing to wait for)
                if self._cur is not None:
                    self._end(None)
            except BaseException:
                if self._cur is not None:           # leave the stream out of capture mode before the error travels on
                    try:
                        self._cur.capture_end()
                    except Exception:               # noqa: BLE001
                        pass
                    self._cur = None
                raise
            finally:
                reducer.capture_sink = None
            self.adam = torch.cuda.CUDAGraph()
            self.adam.capture_begin(pool=self.pool)
            optimizer.step()
            self.adam.capture_end()
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        self.logged = {}

    def _begin(self):
        self._cur = _new_graph()                                  # (the captured hipGraph_t stays readable: node_types)
        self._cur.capture_begin(pool=self.pool)

    @staticmethod
    def _node_count(g):
        """Nodes of a captured graph, None when it cannot be read."""
        nt = node_types(g)
        return None if nt is None else sum(nt.values())

    def _end(self, rng):
        # every captured segment that HOLDS something is kept and replayed, whatever it holds (library launches, torch fills / copies /
        # random draws): deciding "empty" from the library's launch counter would silently drop a segment of torch-only ops.  The one
        # segment that can be truly empty is the trailing one without a bucket (the last bucket was cut inside backward and finish()
        # had nothing left to flush): it is dropped only when the captured graph itself reports zero nodes.
        self._cur.capture_end()
        if rng is None and self._node_count(self._cur) == 0:
            self._cur = None
            return
        self.segments.append((self._cur, rng))
        self._cur = None

    def _cut(self, lo, hi):
        self._end((lo, hi))
        if not self._final:
            self._begin()

    def __call__(self, batch):
        self.x.copy_(batch[0], non_blocking=True)
        self.red.begin()
        for g, rng in self.segments:
            g.replay()
            if rng is not None:
                self.red.launched.append(rng)
                self.red.all_reduce_async(*rng)     # RCCL's stream waits for the replay just enqueued; the next segment does not wait for it
        self.red.wait_all()
        self.adam.replay()
        for n in getattr(self.opt, "nets", ()):
            n.mark_params_dirty()
        return self.loss
