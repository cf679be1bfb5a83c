"""Read-only telemetry of the card HIP device <index> lives on: socket power, core clock, power cap (amdgpu hwmon files in sysfs).

sysfs shows every card of the host, so the right directory is found through the device's PCI address (hipDeviceGetPCIBusId).
Used by bench.py (`power` object of the JSON line) and tools/power_probe.py.  Nothing is written, no setting is changed."""
import ctypes
import glob
import os
import statistics
import threading
import time


def _pci_address(index: int):
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
            return buf.value.decode().lower()
    except OSError:
        pass
    return None


class Hwmon:
    def __init__(self, index: int = 0):
        self.dir = self.power_file = None
        addr = _pci_address(index)
        if not addr:
            return
        for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            if not os.path.realpath(os.path.join(hw, "device")).lower().endswith(addr):
                continue
            for f in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(hw, f)):
                    self.dir, self.power_file = hw, f
                    return

    @property
    def ok(self) -> bool:
        return self.dir is not None

    def _read(self, name, scale):
        try:
            return int(open(os.path.join(self.dir, name)).read()) / scale
        except (OSError, ValueError, TypeError):
            return None

    def cap_W(self):
        return self._read("power1_cap", 1e6) if self.ok else None

    def sample(self):
        """(socket power in W, core clock in MHz), None where a file is missing."""
        if not self.ok:
            return None, None
        return self._read(self.power_file, 1e6), self._read("freq1_input", 1e6)


class Watch:
    """with Watch(hw, settle_s) as w: <keep the card busy> ; w.summary() -> medians of what was sampled every 0.1 s after settle_s."""

    def __init__(self, hw: Hwmon, settle_s: float = 1.0, period_s: float = 0.1):
        self.hw, self.settle_s, self.period_s = hw, settle_s, period_s
        self.samples = []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        if self._stop.wait(self.settle_s):
            return
        while not self._stop.is_set():
            self.samples.append(self.hw.sample())
            self._stop.wait(self.period_s)

    def __enter__(self):
        if self.hw.ok:
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th.is_alive():
            self._th.join()

    def summary(self):
        w = [s[0] for s in self.samples if s[0]]
        f = [s[1] for s in self.samples if s[1]]
        return {"socket_W_median": statistics.median(w) if w else None, "socket_W_max": max(w) if w else None,
                "sclk_MHz_median": statistics.median(f) if f else None, "sclk_MHz_min": min(f) if f else None, "samples": len(self.samples)}
