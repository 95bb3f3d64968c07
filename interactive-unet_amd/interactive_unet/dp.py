"""Data-parallel plumbing of the native training step (one process per GPU, torch.distributed: backend "nccl" = RCCL
over xGMI on the GPU box, gloo in the CPU tests).  The reference trains on one device (trainer.py:61-62); this is the
scale-out of the same step: every rank runs the step on its own batches, the flat fp32 gradient is averaged.

* `broadcast_state`: rank 0's parameters and BatchNorm running statistics to every rank -- without it each rank would
  keep its own random initialisation (or its own view of a checkpoint that rank 0 deletes, trainer.py:41-43) and the
  ranks would apply the same averaged gradient to different weights.
* `GradBuckets`: the gradient all-reduce in three buckets.  The decoder + head gradients (the TAIL of the flat tensor:
  parameters are laid out encoder first) are complete after the decoder backward, so their all-reduce is started
  there and runs on the communication stream while the encoder backward computes; the bottom encoder level (47 % of
  the 3-D net's parameters: 2.65 M of 5.6 M) follows as soon as its two layers are done, and only the upper encoder
  levels (0.66 M parameters = 2.6 MB) are reduced at the end, in the open.  5.6 M parameters = 22.4 MB fp32 for the 3-D
  net: three large messages, not one per tensor.
"""
import torch.distributed as dist


def broadcast_state(flat, buffers, group, src=0):
    """In place: `flat` (the fp32 master parameters) and every tensor of `buffers` become rank `src`'s."""
    root = dist.get_global_rank(group, src) if group is not None and group is not dist.group.WORLD else src
    dist.broadcast(flat, src=root, group=group)
    for b in buffers:
        dist.broadcast(b, src=root, group=group)


class GradBuckets:
    def __init__(self, grad, split, group):
        """grad: flat gradient tensor; [split:] = the bucket that is complete first (decoder + head)."""
        self.grad, self.split, self.group = grad, int(split), group
        self.pending = []
        self.started = []          # [lo, hi) ranges whose all-reduce is in flight

    @property
    def world(self):
        return dist.get_world_size(self.group)

    def start_tail(self):
        """Called when the decoder + head gradients are final: starts their all-reduce asynchronously."""
        self.pending, self.started = [], []
        if self.split > 0:
            self.start(self.split, self.grad.numel())

    def start(self, lo, hi):
        """The gradients [lo, hi) are final: their all-reduce goes out now (every rank calls this at the same point of its backward,
        so the collectives are issued in one order everywhere)."""
        lo, hi = int(lo), int(hi)
        if hi > lo:
            self.pending.append(dist.all_reduce(self.grad[lo:hi], group=self.group, async_op=True))
            self.started.append((lo, hi))

    def finish(self):
        """Called after the whole backward: reduces what start_tail has not, waits for everything.  The gradient then
        holds the SUM over ranks (the optimiser divides by world)."""
        at = 0
        for lo, hi in sorted(self.started) + [(self.grad.numel(), self.grad.numel())]:
            if lo > at:
                self.pending.append(dist.all_reduce(self.grad[at:lo], group=self.group, async_op=True))
            at = max(at, hi)
        for work in self.pending:
            work.wait()
        self.pending, self.started = [], []
        return self.world
