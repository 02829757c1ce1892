"""Device-resident dataset cache: a loader that materialises a (small enough) dataset ONCE on the
GPU and afterwards serves shuffled, rank-sharded batches with a device-side gather — no per-sample
Python, no host->device copies in the steady state.

Why: the reference feeds its trainer with ``num_workers=0`` per-sample Python
(``trainer/trainer.py:209-217``), which caps a VGG16 step at a few hundred images/s while the
kernels here consume ~9,000 images/s per GPU.  With 288 GB of HBM a uint8 224x224 dataset of
1.9 M images fits on one GPU; the synthetic datasets of the benchmark configs fit trivially.  The
loader keeps the DataLoader surface the Trainer uses (``len``, iteration, ``.sampler`` with
``set_epoch``), so sharding and epoch reshuffling stay index-for-index those of ``ShardedSampler``.
"""
import torch


class DeviceCachedLoader(object):
    def __init__(self, dataset, batch_size, sampler, device, max_bytes=None, build_batch=256):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.sampler = sampler
        self.device = torch.device(device)
        self.max_bytes = max_bytes
        self.build_batch = int(build_batch)
        self._x = self._y = None

    def __len__(self):
        n = len(self.sampler)
        return (n + self.batch_size - 1) // self.batch_size

    @staticmethod
    def fits(dataset, device, max_bytes=None):
        """True when the whole dataset (as returned by ``dataset[0]``) fits the budget (default:
        a quarter of the device's free memory)."""
        if len(dataset) == 0:
            return False
        x0, _ = dataset[0]
        need = len(dataset) * x0.numel() * x0.element_size()
        if max_bytes is None:
            if torch.device(device).type != "cuda":
                return False
            free, _ = torch.cuda.mem_get_info(torch.device(device))
            max_bytes = free // 4
        return need <= max_bytes

    def _materialise(self):
        n = len(self.dataset)
        x0, _ = self.dataset[0]
        self._x = torch.empty((n,) + tuple(x0.shape), dtype=x0.dtype, device=self.device)
        self._y = torch.empty((n,), dtype=torch.long, device=self.device)
        for s in range(0, n, self.build_batch):
            e = min(n, s + self.build_batch)
            items = [self.dataset[i] for i in range(s, e)]
            self._x[s:e].copy_(torch.stack([it[0] for it in items]), non_blocking=False)
            self._y[s:e].copy_(torch.as_tensor([int(it[1]) for it in items]))

    def __iter__(self):
        if self._x is None:
            self._materialise()
        idx = torch.as_tensor(list(self.sampler), dtype=torch.long, device=self.device)
        for s in range(0, idx.numel(), self.batch_size):
            sel = idx[s:s + self.batch_size]
            yield [self._x.index_select(0, sel), self._y.index_select(0, sel)]
