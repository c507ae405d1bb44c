"""Shader clock of the GPU while a measurement runs (boxes of the pool hold 2.0 - 2.15 GHz under the fused kernels and
differ by several per cent on the same binary): a host thread samples torch.cuda.clock_rate() every 50 ms.
    with ClockSampler(0) as clk: ...measure...        clk.median_mhz() -> float or None (no SMI library)"""
import statistics
import threading
import time


class ClockSampler:
    def __init__(self, device_index=0, period_s=0.05):
        self.dev, self.period, self.samples = device_index, period_s, []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import torch

        while not self._stop.is_set():
            try:
                self.samples.append((time.perf_counter(), float(torch.cuda.clock_rate(self.dev))))
            except Exception:  # noqa: BLE001
                return
            time.sleep(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join(timeout=2.0)

    def median_mhz(self, last_s=None):
        s = self.samples
        if last_s is not None and s:
            t_end = s[-1][0]
            s = [x for x in s if x[0] >= t_end - last_s]
        return round(statistics.median(c for _, c in s), 1) if s else None
