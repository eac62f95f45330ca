#!/usr/bin/env python
"""Socket power and shader clock while ONE kernel shape runs back to back for a few seconds (rocm-smi polled from a thread).
The GEMMs of the video tower are power-limited (tools/clock_probe.hip): at the cap, the wall time of a launch IS its energy.

    python tools/power_probe.py [seconds per shape]      (shapes: fc1 and fc2 forward of cfg #2)"""
import json
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
samples, stop = [], [False]


def poll():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = next(iter(d.values()))
            pw = next((float(v) for k, v in card.items() if "ower" in k and "W" in k), None)
            sclk = next((v for k, v in card.items() if k.startswith("sclk")), None)
            samples.append((time.time(), pw, sclk))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), None, repr(e)[:80]))
        time.sleep(0.2)


M = 8 * 2356
bf = torch.bfloat16
shapes = [("fc1", 3072, 768, dict(epilogue=L.EPI_BIAS_GELU)), ("fc2", 768, 3072, dict(epilogue=L.EPI_BIAS_RESID))]
th = threading.Thread(target=poll, daemon=True)
th.start()
for name, N, K, kw in shapes:
    A = torch.randn(M, K, device="cuda").to(bf)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, dtype=bf, device="cuda")
    if kw["epilogue"] == L.EPI_BIAS_RESID:
        kw["resid"] = torch.randn(M, N, device="cuda").to(bf)
    if kw["epilogue"] == L.EPI_BIAS_GELU:
        kw["aux"] = torch.empty(M, N, dtype=bf, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    while time.time() - t0 < SECS:
        for _ in range(200):
            H.gemm(A, W, M, N, K, out=out, bias=bias, **kw)
        n += 200
        torch.cuda.synchronize()
    en.record(); torch.cuda.synchronize()
    t1 = time.time()
    us = st.elapsed_time(en) * 1e3 / n
    mine = [s for s in samples if t0 + 0.5 <= s[0] <= t1]
    pws = [s[1] for s in mine if s[1] is not None]
    print(f"{name}: {us:7.1f} us/launch ({2*M*N*K/us/1e6:6.0f} TFLOP/s) over {n} launches; power W min/avg/max "
          f"{min(pws, default=0):.0f}/{sum(pws)/max(len(pws),1):.0f}/{max(pws, default=0):.0f}; sclk samples {[s[2] for s in mine][:6]}")
    time.sleep(1.0)
stop[0] = True
idle = [s for s in samples if s[1] is not None][-2:]
print("after:", idle)
