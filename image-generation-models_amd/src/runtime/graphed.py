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
        self.graph = torch.cuda.CUDAGraph()
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
