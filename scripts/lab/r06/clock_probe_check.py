"""What does d(s_memtime) / d(s_memrealtime) read over idle and busy intervals of different lengths?  (bench.py's ClockProbe)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from bench import ClockProbe  # noqa: E402
from librecommender_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
clk = ClockProbe(dev)
print("realtime counter MHz", round(clk.rt_hz / 1e6, 3))
out = torch.zeros(4, device=dev)


def busy(ms):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        ops._call("lr_mfma_f32_probe", 20000, 2, out.data_ptr(), ops._stream())
        torch.cuda.synchronize()


for name, fn in (("idle 50 ms", lambda: time.sleep(0.05)), ("idle 2 ms", lambda: time.sleep(0.002)), ("busy 100 ms", lambda: busy(100)),
                 ("busy 5 ms", lambda: busy(5)), ("busy 1 ms", lambda: busy(1)), ("busy 100 ms again", lambda: busy(100))):
    clk.mark(0)
    torch.cuda.synchronize()
    fn()
    clk.mark(1)
    b = clk.buf.cpu()
    ok = (b[0, :, 1] > 0) & (b[1, :, 1] > b[0, :, 1])
    r = ((b[1, :, 0] - b[0, :, 0]).double() / (b[1, :, 1] - b[0, :, 1]).double().clamp_min(1))[ok] * 100
    q = torch.quantile(r, torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], dtype=torch.float64)) if int(ok.sum()) else []
    print(f"{name:18s} MHz {clk.mhz()}   CUs seen twice {int(ok.sum())}, quantiles of d_sh / d_rt x 100: {[round(float(x)) for x in q]}")
