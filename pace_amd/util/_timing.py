"""Timer / NullTimer with the reference's API (util/pace/util/_timing.py:9-98) and a per-entry-point device-time collector.

``Timer`` accumulates wall-clock time of named operations (``start`` / ``stop`` / ``clock``, ``times``, ``hits``, ``reset``,
``enable`` / ``disable``).  Like the reference on a GPU (which synchronises the device around every region), a region is
closed only after the work launched inside it has finished -- but it waits for the CURRENT STREAM instead of the whole
device, so transfers and side streams of other regions keep running.

``KernelTimes`` is what the reference's ``TimingCollector.exec_info`` / ``StencilFactory.exec_report()`` are for GT4Py's
generated kernels (dsl/pace/dsl/stencil.py:103-163): one record per C entry point with the number of calls and the
device time between a pair of HIP events recorded on the launch stream around each call.
"""
import contextlib
from timeit import default_timer as _now


def _sync():
    try:
        import torch

        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.current_stream().synchronize()
    except Exception:  # noqa: BLE001  (timing must never break a run)
        pass


class Timer:
    """Class to accumulate timings for named operations."""

    def __init__(self):
        self._clock_starts = {}
        self._accumulated_time = {}
        self._hit_count = {}
        self._enabled = True

    def start(self, name: str):
        """Start timing a given named operation."""
        if self._enabled:
            _sync()
            if name in self._clock_starts:
                raise ValueError(f"clock already started for '{name}'")
            self._clock_starts[name] = _now()

    def stop(self, name: str):
        """Stop timing a given named operation, add the time elapsed to accumulated timing and increase the hit count."""
        if self._enabled:
            _sync()
            dt = _now() - self._clock_starts.pop(name)
            self._accumulated_time[name] = self._accumulated_time.get(name, 0.0) + dt
            self._hit_count[name] = self._hit_count.get(name, 0) + 1

    @contextlib.contextmanager
    def clock(self, name: str):
        """Context manager to produce timings of operations."""
        self.start(name)
        yield
        self.stop(name)

    @property
    def times(self):
        """accumulated timings for each operation name"""
        if len(self._clock_starts) > 0:
            import warnings

            warnings.warn("Retrieved times while clocks are still going, incomplete times are not included: "
                          f"{list(self._clock_starts.keys())}", RuntimeWarning)
        return self._accumulated_time.copy()

    @property
    def hits(self):
        """accumulated hit counts for each operation name"""
        return self._hit_count.copy()

    def reset(self):
        """Remove all accumulated timings."""
        self._accumulated_time.clear()
        self._hit_count.clear()

    def enable(self):
        """Enable the Timer."""
        self._enabled = True

    def disable(self):
        """Disable the Timer."""
        if len(self._clock_starts) > 0:
            raise RuntimeError(f"Cannot disable timer while clocks are still going: {list(self._clock_starts.keys())}")
        self._enabled = False

    @property
    def enabled(self) -> bool:
        """Indicates whether the timer is currently enabled."""
        return self._enabled


class NullTimer(Timer):
    """A Timer class which does not actually accumulate timings."""

    def __init__(self):
        super().__init__()
        self._enabled = False

    def enable(self):
        raise NotImplementedError("NullTimer cannot be enabled")


class KernelTimes:
    """Per-entry-point device times: ``Library.call`` brackets every call with two events on the call's launch stream while a
    collector is attached (``lib.timing = KernelTimes()``).  ``report()`` resolves the events and prints the table."""

    def __init__(self):
        self._pending = []  # (name, start event, end event)
        self.exec_info = {}  # name -> {"ncalls", "total_run_time" (s)}

    def bracket(self, name, stream_ptr=None):
        """Start event of one entry-point call, recorded on the stream the call launches on (`stream_ptr`: the raw HIP stream
        the entry point was handed -- its last argument -- or None for the current stream: d_sw's wind half runs on a side
        stream, and events on a stream that runs nothing would time nothing).  Returns an object whose .record() closes the
        bracket on the same stream."""
        import torch

        if not torch.cuda.is_available():
            return None
        stream = torch.cuda.ExternalStream(int(stream_ptr)) if stream_ptr else torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        self._pending.append((name, e0, e1))

        class _End:
            def record(self_inner):
                e1.record(stream)

        return _End()

    def resolve(self):
        import torch

        if self._pending:
            torch.cuda.synchronize()
        for name, e0, e1 in self._pending:
            rec = self.exec_info.setdefault(name, {"ncalls": 0, "total_run_time": 0.0})
            rec["ncalls"] += 1
            rec["total_run_time"] += e0.elapsed_time(e1) * 1e-3
        self._pending.clear()
        return self.exec_info

    def report(self, key: str = "total_run_time", name_width: int = 44, bar_width: int = 30) -> str:
        info = self.resolve()
        rows = sorted(((k, v[key], v["ncalls"]) for k, v in info.items()), key=lambda r: -r[1])
        if not rows:
            return "Total: 0"
        top = rows[0][1] or 1.0
        out = [f"Total: {sum(r[1] for r in rows):.3e}"]
        for name, val, n in rows:
            out.append(f"{name.rjust(name_width)} | {val:.3e} | {n:6d} | " + "#" * int(val / top * bar_width))
        return "\n".join(out)
