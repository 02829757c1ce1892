"""Asynchronous host->device batch prefetcher.

The reference copies each batch with a blocking ``tensor.to("cuda:k")`` inside the training step
(``example_trainer.py:70``) although the loader pins memory.  Here
the copy of batch i+1 is issued on a dedicated HIP copy stream from pinned memory with
``non_blocking=True`` while batch i computes; an event hands the tensors over to the compute
stream (``record_stream`` keeps the allocator honest).  On CPU it is a transparent pass-through.
"""
import torch

from ..utils.profiler import trace_range


class DevicePrefetcher(object):
    def __init__(self, loader, device, depth=2):
        self.loader = loader
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None

    def __len__(self):
        return len(self.loader)

    def _to_device(self, batch):
        if not self.cuda:
            return batch, None
        with trace_range("h2d"), torch.cuda.stream(self.stream):
            out = []
            for e in batch:
                if torch.is_tensor(e):
                    if not e.is_pinned() and e.device.type == "cpu":
                        e = e.pin_memory()
                    out.append(e.to(self.device, non_blocking=True))
                else:
                    out.append(e)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        queue = []
        try:
            for _ in range(self.depth):
                queue.append(self._to_device(next(it)))
        except StopIteration:
            pass
        while queue:
            batch, ev = queue.pop(0)
            try:
                queue.append(self._to_device(next(it)))
            except StopIteration:
                pass
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for e in batch:
                    if torch.is_tensor(e):
                        e.record_stream(torch.cuda.current_stream(self.device))
            yield batch
