"""Running meters for the training / evaluation loops -- observability only.

Same public surface as the loops of the reference expect (`util/metrics.py`): ``MetricLogger(delimiter)`` with
``update(**scalars)``, ``meters[name].global_avg``, ``log_every(iterable, print_freq, header)``,
``synchronize_between_processes()``, ``add_meter``; ``SmoothedValue`` with ``median / avg / global_avg / max / value``.
Implementation is plain Python (no tensors are created for the statistics).
"""
from __future__ import annotations

import collections
import datetime
import statistics
import time

import torch
import torch.distributed as dist

from .dist import is_dist_avail_and_initialized

_DEFAULT_FMT = "{median:.4f} ({global_avg:.4f})"


class SmoothedValue:
    """Windowed statistics over the last `window_size` updates plus the running global average."""

    def __init__(self, window_size: int = 20, fmt: str | None = None):
        self._window = collections.deque(maxlen=window_size)
        self._weighted_sum = 0.0
        self._weight = 0
        self.fmt = fmt if fmt is not None else _DEFAULT_FMT

    # the reference exposes these two as plain attributes
    @property
    def total(self) -> float:
        return self._weighted_sum

    @property
    def count(self) -> int:
        return self._weight

    def update(self, value, num: int = 1) -> None:
        value = float(value)
        self._window.append(value)
        self._weight += num
        self._weighted_sum += value * num

    def synchronize_between_processes(self) -> None:
        """Global average over all ranks (the window stays local)."""
        if not is_dist_avail_and_initialized():
            return
        where = "cuda" if dist.get_backend() == "nccl" else "cpu"
        packed = torch.tensor([float(self._weight), self._weighted_sum], dtype=torch.float64, device=where)
        dist.barrier()
        dist.all_reduce(packed)
        self._weight, self._weighted_sum = int(packed[0].item()), float(packed[1].item())

    @property
    def median(self) -> float:
        # lower median, like torch.median on the window
        return statistics.median_low(self._window) if self._window else float("nan")

    @property
    def avg(self) -> float:
        return sum(self._window) / len(self._window) if self._window else float("nan")

    @property
    def global_avg(self) -> float:
        return self._weighted_sum / self._weight if self._weight else 0.0

    @property
    def max(self) -> float:
        return max(self._window)

    @property
    def value(self) -> float:
        return self._window[-1]

    def __str__(self) -> str:
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


def _eta(seconds: float) -> str:
    return str(datetime.timedelta(seconds=int(seconds)))


class MetricLogger:
    def __init__(self, delimiter: str = "\t"):
        self.meters: dict = collections.defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **scalars) -> None:
        for name, v in scalars.items():
            self.meters[name].update(v.item() if isinstance(v, torch.Tensor) else v)

    def add_meter(self, name: str, meter: SmoothedValue) -> None:
        self.meters[name] = meter

    def __getattr__(self, name):
        meters = self.__dict__.get("meters")
        if meters is not None and name in meters:
            return meters[name]
        raise AttributeError(name)

    def __str__(self) -> str:
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())

    def synchronize_between_processes(self) -> None:
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def log_every(self, iterable, print_freq, header=None):
        """Yield the items of `iterable`, printing a progress line every `print_freq` items and at the end."""
        header = header or ""
        n_items = len(iterable) if hasattr(iterable, "__len__") else -1
        step_time = SmoothedValue(fmt="{avg:.4f}")
        wait_time = SmoothedValue(fmt="{avg:.4f}")
        t_begin = t_mark = time.time()
        for pos, item in enumerate(iterable):
            wait_time.update(time.time() - t_mark)
            yield item
            step_time.update(time.time() - t_mark)
            if print_freq and (pos % print_freq == 0 or pos == n_items - 1):
                fields = [header, f"[{pos}/{n_items}]",
                          "eta: " + (_eta(step_time.global_avg * (n_items - pos)) if n_items > 0 else "?"),
                          str(self), f"time: {step_time}", f"data: {wait_time}"]
                if torch.cuda.is_available():
                    fields.append(f"max mem: {torch.cuda.max_memory_allocated() / 2 ** 20:.0f}")
                print(self.delimiter.join(fields))
            t_mark = time.time()
        print(f"{header} Total time: {_eta(time.time() - t_begin)}")
