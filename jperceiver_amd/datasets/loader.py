"""Batch feeding for the data-parallel train step: pinned host staging + asynchronous H2D on a dedicated copy stream,
double-buffered so that step k's upload (and device-side preprocessing) overlaps step k-1's compute.  Replaces the
reference's synchronous `data[k].cuda()` per tensor (mono/apis/trainer.py:20-27) behind a 24-worker PIL pipeline
(config workers_per_gpu, mono/datasets/loader/build_loader.py:18-55): at ~55 images/s per GPU that path is the bottleneck.

    loader = DeviceLoader(iterable_of_cpu_batches, device, preprocess=None, depth=2)
    for batch in loader: runner.train_iter(batch)        # tensors already on the GPU, ready on the current stream
"""
from __future__ import annotations

import queue
import threading

import torch


def collate(samples):
    """mmcv.parallel.collate for dicts of equally shaped tensors / arrays: stack along a new batch dimension."""
    out = {}
    for k in samples[0]:
        v0 = samples[0][k]
        if torch.is_tensor(v0):
            out[k] = torch.stack([s[k] for s in samples])
        else:
            out[k] = torch.stack([torch.as_tensor(s[k]) for s in samples])
    return out


def build_dataloader(dataset, imgs_per_gpu, workers_per_gpu, num_gpus=1, dist=True, **kwargs):
    """mono/datasets/loader/build_loader.py:18-55 with the same arguments and the same sampler choice: shuffled runs cut
    ONE epoch-seeded plan across the ranks (`DistributedGroupSampler`; `GroupSampler` for the single-process path),
    `shuffle=False` takes `DistributedSampler` / sequential order; batches of `imgs_per_gpu` (x `num_gpus` when not
    distributed), `drop_last=True`.  Collation is `collate` above (stack along a new batch axis)."""
    from torch.utils.data import DataLoader
    from .sampler import DistributedGroupSampler, DistributedSampler, GroupSampler
    shuffle = kwargs.pop("shuffle", True)
    if dist:
        sampler = DistributedGroupSampler(dataset, imgs_per_gpu) if shuffle else DistributedSampler(dataset, shuffle=False)
        batch_size, num_workers = imgs_per_gpu, workers_per_gpu
    else:
        sampler = GroupSampler(dataset, imgs_per_gpu) if shuffle else None
        batch_size, num_workers = num_gpus * imgs_per_gpu, num_gpus * workers_per_gpu
    kwargs.setdefault("collate_fn", collate)
    return DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers, pin_memory=False,
                      drop_last=True, **kwargs)


class DeviceLoader:
    def __init__(self, batches, device="cuda", preprocess=None, depth=2):
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.batches, self.dev, self.pre, self.depth = batches, dev, preprocess, max(1, depth)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.sampler = getattr(batches, "sampler", None)       # DistSamplerSeedHook's handle (runner.train_epoch)

    def __len__(self):
        return len(self.batches)

    def _stage(self, batch):
        """pin + enqueue the upload (and the device preprocessing) on the copy stream; -> (device batch, ready event)"""
        with torch.cuda.stream(self.copy_stream):
            dev_batch = {}
            for k, v in batch.items():
                t = torch.as_tensor(v)
                if not t.is_pinned():
                    t = t.contiguous().pin_memory()
                dev_batch[k] = t.to(self.dev, non_blocking=True)
            if self.pre is not None:
                dev_batch = self.pre(dev_batch)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return dev_batch, ev

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        stop = object()

        def producer():
            try:
                torch.cuda.set_device(self.dev)
                for b in self.batches:
                    q.put(self._stage(b))
            except BaseException as e:      # surface loader errors in the consumer
                q.put(e)
            q.put(stop)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        while True:
            item = q.get()
            if item is stop:
                break
            if isinstance(item, BaseException):
                raise item
            dev_batch, ev = item
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(ev)                              # no host sync: the compute stream waits on the copy
            for t in dev_batch.values():
                if torch.is_tensor(t):
                    t.record_stream(cur)
            yield dev_batch
        th.join()
