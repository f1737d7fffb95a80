"""SmoothedValue / MetricLogger with the reference's API (util/metrics.py) -- observability only."""
from __future__ import annotations

import datetime
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist

from .dist import is_dist_avail_and_initialized


class SmoothedValue:
    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, num=1):
        self.deque.append(value)
        self.count += num
        self.total += value * num

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / max(self.count, 1)

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(float(v))

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join(f"{n}: {m}" for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        start = end = time.time()
        it_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        n = len(iterable) if hasattr(iterable, "__len__") else -1
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            it_time.update(time.time() - end)
            if print_freq and (i % print_freq == 0 or i == n - 1):
                eta = str(datetime.timedelta(seconds=int(it_time.global_avg * max(n - i, 0)))) if n > 0 else "?"
                mem = f"max mem: {torch.cuda.max_memory_allocated() / 2**20:.0f}" if torch.cuda.is_available() else ""
                print(self.delimiter.join([header, f"[{i}/{n}]", f"eta: {eta}", str(self), f"time: {it_time}",
                                           f"data: {data_time}", mem]))
            end = time.time()
        total = time.time() - start
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))}")
