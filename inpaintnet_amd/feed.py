"""Input feed: host batches -> device batches without stalling the step.

Reference side of the contract:
  * MusicDataset.data_loaders (DatasetManager/music_dataset.py:177-221): three torch DataLoaders over
    TensorDataset(score (N,1,384), metadata (N,1,384,3)), batch_size, drop_last=True, train shuffled;
  * the trainers move every batch to the GPU synchronously from pageable memory and widen it to int64 there
    (MeasureVAE/vae_trainer.py:42-55, LatentRNN/latent_rnn_trainer.py:134-176, utils/helpers.py:17-26).

Here:
  BatchLoader  the DataLoader contract (len, shuffle, drop_last, the same draws from torch's global CPU generator as
               DataLoader + RandomSampler make, so a seeded run visits the same batches in the same order) over
               in-memory tensors, gathering each batch into a ring of PINNED staging buffers;
  DeviceFeed   wraps any iterable of host batches: takes this rank's shard of every batch, copies it to the GPU on
               a dedicated copy stream `depth` batches ahead of the consumer (int32 stays int32 on the wire: 4 bytes
               per token), and hands out device tensors ordered after the copy by an event -- no host sync.
The widening to int64 and the past/target/future split are HIP kernels (ops.tokens_to_long / ops.split_score),
called by the trainers' process_batch_data.
"""
from collections import deque

import torch

from . import dp


class BatchLoader:
    """len() / iteration semantics of torch.utils.data.DataLoader(TensorDataset(*tensors), batch_size, shuffle,
    drop_last) with num_workers=0.  A yielded batch stays valid until `ring - 1` further batches have been drawn."""

    def __init__(self, tensors, batch_size, shuffle=False, drop_last=True, pin_memory=None, ring=4):
        assert len(tensors) > 0 and all(t.shape[0] == tensors[0].shape[0] for t in tensors)
        self.tensors = [t.contiguous() for t in tensors]
        self.n = int(tensors[0].shape[0])
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self.drop_last = bool(drop_last)
        self.pin_memory = torch.cuda.is_available() if pin_memory is None else bool(pin_memory)
        self.ring = max(2, int(ring))
        self._staging = None

    def __len__(self):
        if self.drop_last:
            return self.n // self.batch_size
        return (self.n + self.batch_size - 1) // self.batch_size

    def _buffers(self):
        if self._staging is None:
            self._staging = [[torch.empty((self.batch_size,) + tuple(t.shape[1:]), dtype=t.dtype,
                                          pin_memory=self.pin_memory) for t in self.tensors]
                             for _ in range(self.ring)]
        return self._staging

    def __iter__(self):
        # the two draws DataLoader makes per epoch from the global CPU generator: the iterator's base seed, then (when
        # shuffling) RandomSampler's seed for its private permutation generator
        torch.empty((), dtype=torch.int64).random_()
        if self.shuffle:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            g = torch.Generator()
            g.manual_seed(seed)
            order = torch.randperm(self.n, generator=g)
        else:
            order = torch.arange(self.n)
        bufs = self._buffers()
        for i in range(len(self)):
            idx = order[i * self.batch_size:(i + 1) * self.batch_size]
            slot = bufs[i % self.ring]
            out = []
            for src, dst in zip(self.tensors, slot):
                view = dst[:idx.numel()]
                torch.index_select(src, 0, idx, out=view)
                out.append(view)
            yield tuple(out)


class DeviceFeed:
    """Iterate `loader` with the H2D copies running `depth` batches ahead on their own stream.

    fields: indices of the batch tuple that are needed on the device (others are passed through as None: the
    MeasureVAE / LatentRNN trainers never read the metadata tensor).  With torch.distributed initialised each rank
    copies only its contiguous shard of the global batch."""

    def __init__(self, loader, fields=(0,), device=None, depth=2):
        self.loader = loader
        self.fields = tuple(fields)
        self.device = device
        self.depth = max(1, int(depth))

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        dev = self.device or torch.device("cuda", torch.cuda.current_device())
        if torch.device(dev).type != "cuda":           # host-only plumbing (CPU tests of the epoch loop): shard, no copy
            for batch in self.loader:
                host = dp.shard_batch(tuple(batch))
                yield tuple(b if i in self.fields else None for i, b in enumerate(host))
            return
        copy_stream = torch.cuda.Stream(device=dev)
        it = iter(self.loader)
        inflight = deque()
        # A loader that gathers into a ring of staging buffers (BatchLoader) overwrites slot k % ring when batch k is
        # drawn: the copy of batch k - ring must have left the buffer by then.  The copies run far ahead of the
        # compute stream, so this host wait is almost never taken -- but it is what makes the reuse safe.
        ring = int(getattr(self.loader, "ring", 0))
        copies = deque()

        def issue():
            if ring and len(copies) == ring:
                copies.popleft().synchronize()
            try:
                batch = next(it)
            except StopIteration:
                return False
            host = dp.shard_batch(tuple(batch))
            with torch.cuda.stream(copy_stream):
                moved = tuple(b.to(dev, non_blocking=True) if (i in self.fields and torch.is_tensor(b)) else None
                              for i, b in enumerate(host))
                done = torch.cuda.Event()
                done.record(copy_stream)
            if ring:
                copies.append(done)
            inflight.append((moved, done, host))       # `host` keeps the source tensors alive until consumed
            return True

        for _ in range(self.depth):
            if not issue():
                break
        while inflight:
            moved, done, _host = inflight.popleft()
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(done)                       # device-side ordering only; the host runs ahead
            for t in moved:
                if t is not None:
                    t.record_stream(cur)               # allocated on the copy stream, consumed on the compute stream
            issue()
            yield moved
