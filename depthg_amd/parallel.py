"""Data-parallel glue for the correlation loss (new in the build; the reference trains on one device,
src/train_segmentation.py:683-714).

One process per GPU, batch-sharded: every rank evaluates the loss on its local shard (shard-local
`old_mean` and negatives, i.e. what DDP would do with the reference module - SURVEY.md section 8(e)).
The only exchange step is one all-reduce (sum, then * 1/world) of a single flat fp32 buffer holding the
trainable head gradients; torch.distributed's "nccl" backend is RCCL over xGMI on ROCm, "gloo" is used by
the CPU tests.  Messages are <= 3 MB, i.e. latency-bound: one bucket, one collective per step.
"""
from typing import Iterable, List, Optional, Sequence

import torch


def shard_range(global_batch: int, world: int, rank: int):
    """Contiguous [lo, hi) slice of the global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradBucket:
    """One flat fp32 buffer for all head gradients; `allreduce_mean_` averages it over the process group."""

    def __init__(self, numel: int, device, dist_module=None, group=None):
        self.flat = torch.zeros(int(numel), dtype=torch.float32, device=device)
        self.dist = dist_module
        self.group = group
        self._views: List[torch.Tensor] = []

    @classmethod
    def for_parameters(cls, params: Sequence[torch.Tensor], dist_module=None, group=None):
        params = [p for p in params if p.requires_grad]
        b = cls(sum(p.numel() for p in params), params[0].device, dist_module, group)
        off = 0
        for p in params:
            b._views.append(b.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        b._params = params
        return b

    def pack(self):
        """Copy p.grad of the registered parameters into the flat buffer (zeros for missing grads)."""
        for p, v in zip(self._params, self._views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)

    def unpack(self):
        for p, v in zip(self._params, self._views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)

    def fill_from(self, t: torch.Tensor):
        """Benchmark stand-in: fill the bucket from the leading elements of a gradient tensor."""
        src = t.reshape(-1)
        n = min(src.numel(), self.flat.numel())
        self.flat[:n].copy_(src[:n])

    def allreduce_mean_(self, even_if_alone: bool = False, async_op: bool = False):
        """Average the flat buffer over the group.  async_op (RCCL only): the collective is enqueued on the communication
        stream and the compute stream does NOT wait for it here; call `wait()` before the buffer is read or refilled - the
        exchange then overlaps whatever is launched in between (in training: the frozen ViT forward of the next step)."""
        if self.dist is None:
            return self.flat
        world = self.dist.get_world_size(self.group)
        if world > 1 or even_if_alone:
            if self.dist.get_backend(self.group) == "nccl":        # RCCL averages inside the collective: no extra kernel
                self._work = self.dist.all_reduce(self.flat, op=self.dist.ReduceOp.AVG, group=self.group, async_op=async_op)
            else:                                                   # gloo (CPU tests) has no AVG
                self.dist.all_reduce(self.flat, op=self.dist.ReduceOp.SUM, group=self.group)
                self.flat.mul_(1.0 / world)
        return self.flat

    def exchange_on(self, stream, source: Optional[torch.Tensor] = None, even_if_alone: bool = False):
        """The whole exchange on a side stream: `stream` waits for everything enqueued so far on the current stream, then (fills
        the bucket from `source` and) all-reduces it there.  The compute stream is never made to wait - no barrier packet in
        its queue; successive exchanges are ordered by `stream` itself, so the buffer is not refilled before the previous
        collective has read it.  Whoever consumes the averaged gradients (the optimiser step) calls `wait_exchange()` first."""
        if not self.flat.is_cuda:
            # CPU buckets (gloo: the tests of the step schedule): there are no streams, the exchange completes right here
            if source is not None:
                self.fill_from(source)
            self.allreduce_mean_(even_if_alone=even_if_alone)
            return
        cur = torch.cuda.current_stream()
        done = cur.record_event()
        if source is not None:
            source.record_stream(stream)           # the caching allocator must not hand the tensor out while `stream` reads it
        with torch.cuda.stream(stream):
            stream.wait_event(done)
            if source is not None:
                self.fill_from(source)
            self.allreduce_mean_(even_if_alone=even_if_alone)
            self._exchange_done = stream.record_event()

    def wait_exchange(self, host: bool = False):
        """Order the last `exchange_on` in front of what comes next.  host=False: the current stream waits for it (no host block).
        host=True: the HOST waits for it - no packet in the compute stream's queue.  A cross-stream wait in front of every replayed
        step costs that stream ~10 us although the event it names completed long before (round 5, one rank, `--force-dist`: 0.327 ->
        0.317 ms); the collective of two steps ago is done by the time the host, which runs ahead of the GPU, gets here, so the host
        wait only bounds the run-ahead to the two buckets."""
        ev = getattr(self, "_exchange_done", None)
        if ev is not None:
            if host:
                ev.synchronize()
            else:
                torch.cuda.current_stream().wait_event(ev)

    def wait(self):
        """Make the current stream wait for an outstanding async all-reduce (no host block)."""
        w = getattr(self, "_work", None)
        if w is not None:
            w.wait()
            self._work = None


class DoubleBufferedExchange:
    """The N > 1 step schedule (bench.py, and what a trainer around `UnsupervisedSegmenter.training_step(grad_sync=...)` does):
    two gradient buckets used alternately, so that step i + 1 fills the other buffer while step i's all-reduce is in flight on
    the side stream.  Per step, in this order:

        1. `wait_exchange` of bucket k = i mod 2  - the collective that read this buffer two steps ago (long done); the HOST waits
                                                    (`host_wait`, the default on GPU buckets): the refill must not overtake that
                                                    collective, and a wait packet in the compute stream's queue costs ~10 us per step
        2. `compute(k)`                           - the step's kernels; they end by filling bucket k (a hipGraph replay in bench.py)
        3. `exchange_on(comm)` of bucket k        - the all-reduce on the side stream, behind everything enqueued so far

    `warm()` runs the step's kernels WITHOUT the collective (clock warm-up: ranks may run different counts of it, and a
    collective there would pair up steps of different ranks); `drain()` makes every outstanding collective complete before the
    timed region ends.  Every rank must call `step()` the same number of times.  `trace` (when given a list) records the calls -
    the CPU tests drive this class over gloo with a stub `compute`."""

    def __init__(self, buckets, compute, comm_stream=None, even_if_alone=False, exchange=True, alternate=True, trace=None, host_wait=True):
        assert len(buckets) == 2
        self.buckets, self.compute, self.comm = buckets, compute, comm_stream
        self.even_if_alone, self.exchange, self.alternate, self.host_wait = even_if_alone, exchange, alternate, host_wait
        self.count, self.trace = 0, trace

    def _note(self, what, k):
        if self.trace is not None:
            self.trace.append((what, k, self.count))

    def step(self):
        k = (self.count & 1) if self.alternate else 0
        self._note("wait", k)
        self.buckets[k].wait_exchange(host=self.host_wait)
        self._note("compute", k)
        out = self.compute(k)
        if self.exchange:
            self._note("exchange", k)
            self.buckets[k].exchange_on(self.comm, even_if_alone=self.even_if_alone)
        self.count += 1
        return out

    def warm(self):
        """The step's kernels alone (bucket 0's buffer is overwritten; nothing is exchanged)."""
        self._note("warm", 0)
        self.buckets[0].wait_exchange()
        return self.compute(0)

    def drain(self):
        for k, b in enumerate(self.buckets):
            self._note("drain", k)
            b.wait_exchange()
