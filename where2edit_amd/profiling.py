"""HIP-event timing of individual C-ABI calls on the stream they are launched on (torch's current
stream).  bench.py uses it to price the dominant kernel inside the timed region; off by default."""
import torch

_active = None


class KernelTimer:
    """with KernelTimer() as t: ... ; t.summary() -> {name: (calls, total_ms, total_work)}"""

    def __init__(self):
        self.records = []  # (name, work, start_event, end_event)
        self.enabled = True  # the caller may switch spans off for some steps (a timed event record costs ~8 us of GPU time)

    def __enter__(self):
        global _active
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, work, a, b in self.records:
            calls, ms, tot = out.get(name, (0, 0.0, 0.0))
            out[name] = (calls + 1, ms + a.elapsed_time(b), tot + work)
        return out


class _Span:
    __slots__ = ("name", "work", "start")

    def __init__(self, name, work):
        self.name, self.work = name, work
        self.start = torch.cuda.Event(enable_timing=True)
        self.start.record()

    def end(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _active.records.append((self.name, self.work, self.start, e))


def span(name, work=0.0):
    """Returns an object whose .end() closes the span, or None when timing is off."""
    return _Span(name, work) if (_active is not None and _active.enabled) else None
