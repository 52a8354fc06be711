"""Developer tool: sample socket power, the power cap, the shader clock and the energy counter of GPU 0 through librocm_smi64
(ctypes, in-process: no child process, nothing that touches the HIP runtime) while a kernel runs.

    with Sampler(period=0.02) as s:
        ... launches ...
    s.summary()  ->  {"power_w": mean, "power_max_w": ..., "cap_w": ..., "sclk_mhz": mean, "sclk_min_mhz": ..., "energy_j": ..., "samples": n}

Every field whose query the library refuses is None (an ordinary user may not be allowed every sysfs file)."""
import ctypes
import threading
import time


class _Freqs(ctypes.Structure):
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                ("frequency", ctypes.c_uint64 * 33)]


class Sampler:
    def __init__(self, device=0, period=0.02):
        self.dev, self.period = device, period
        self.samples = []            # (t, power_w or None, sclk_mhz or None)
        self.lib = None
        self.cap_w = None
        self._stop = threading.Event()
        self._e0 = self._e1 = None
        try:
            lib = ctypes.CDLL("librocm_smi64.so")
            if lib.rsmi_init(ctypes.c_uint64(0)) == 0:
                self.lib = lib
        except OSError:
            pass
        if self.lib:
            cap = ctypes.c_uint64(0)
            if self.lib.rsmi_dev_power_cap_get(self.dev, 0, ctypes.byref(cap)) == 0:
                self.cap_w = cap.value / 1e6

    def _power(self):
        p, typ = ctypes.c_uint64(0), ctypes.c_int(0)
        if self.lib.rsmi_dev_power_get(self.dev, ctypes.byref(p), ctypes.byref(typ)) == 0:
            return p.value / 1e6
        if self.lib.rsmi_dev_current_socket_power_get(self.dev, ctypes.byref(p)) == 0:
            return p.value / 1e6
        if self.lib.rsmi_dev_power_ave_get(self.dev, 0, ctypes.byref(p)) == 0:
            return p.value / 1e6
        return None

    def _sclk(self):
        f = _Freqs()
        if self.lib.rsmi_dev_gpu_clk_freq_get(self.dev, 0, ctypes.byref(f)) == 0 and f.current < 33:      # RSMI_CLK_TYPE_SYS
            return f.frequency[f.current] / 1e6
        return None

    def _energy(self):
        e, res, ts = ctypes.c_uint64(0), ctypes.c_float(0), ctypes.c_uint64(0)
        if self.lib.rsmi_dev_energy_count_get(self.dev, ctypes.byref(e), ctypes.byref(res), ctypes.byref(ts)) == 0:
            return e.value * res.value * 1e-6          # micro-joule units -> J
        return None

    def _run(self):
        while not self._stop.is_set():
            self.samples.append((time.perf_counter(), self._power(), self._sclk()))
            self._stop.wait(self.period)

    def __enter__(self):
        self.samples = []
        self._stop.clear()
        if self.lib:
            self._e0 = self._energy()
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        if self.lib:
            self._stop.set()
            self._t.join()
            self._e1 = self._energy()

    def summary(self):
        pw = [p for _, p, _ in self.samples if p is not None]
        ck = [c for _, _, c in self.samples if c is not None]
        return {
            "power_w": sum(pw) / len(pw) if pw else None, "power_max_w": max(pw) if pw else None, "cap_w": self.cap_w,
            "sclk_mhz": sum(ck) / len(ck) if ck else None, "sclk_min_mhz": min(ck) if ck else None,
            "energy_j": (self._e1 - self._e0) if (self._e0 is not None and self._e1 is not None) else None,
            "samples": len(self.samples),
        }


_once = None


def read_once(device=0):
    """one reading: socket power, shader clock and the temperature sensors the library exposes (edge, junction, memory), for log lines"""
    global _once
    if _once is None:
        _once = Sampler(device=device)
    s = _once
    if not s.lib:
        return {}
    out = {"w": s._power(), "sclk": s._sclk()}
    for name, sensor in (("edge", 0), ("junction", 1), ("mem", 2)):
        t = ctypes.c_int64(0)
        if s.lib.rsmi_dev_temp_metric_get(s.dev, sensor, 0, ctypes.byref(t)) == 0:      # RSMI_TEMP_CURRENT, millidegrees
            out[name + "_c"] = t.value / 1000.0
    return out


def fmt(s):
    def f(v, u, d=0):
        return ("%.*f %s" % (d, v, u)) if v is not None else "n/a"
    return "power %s (max %s, cap %s)  sclk %s (min %s)  energy %s  [%d samples]" % (
        f(s["power_w"], "W"), f(s["power_max_w"], "W"), f(s["cap_w"], "W"), f(s["sclk_mhz"], "MHz"), f(s["sclk_min_mhz"], "MHz"),
        f(s["energy_j"], "J", 1), s["samples"])


if __name__ == "__main__":
    with Sampler(period=0.05) as s:
        time.sleep(0.5)
    print(fmt(s.summary()))
