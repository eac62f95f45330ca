#!/usr/bin/env python
"""Bandwidth of the two AdamW update kernels alone on the chip: xp_adamw_step (one workgroup per 64 Ki-element chunk) against
xp_adamw_step_confined at several workgroup counts (= compute units held), on one 96 M-element tensor set with bf16 shadows
(30 bytes per element).  Needs tools/experiments/adamw_confined.diff applied to csrc/optim.hip and the entry point added to _lib.py
(measured, negative: profiles/r06w_adamw_confined_to_few_cus_negative.txt).  Usage: python tools/experiments/adamw_confined_probe.py [workgroups ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from xpretrain_amd import _lib as L  # noqa: E402

lib = L.lib()
wgs = [int(a) for a in sys.argv[1:]] or [8, 16, 24, 32, 48, 64, 96, 128, 256]
sizes = [768 * 3072] * 40
p = [torch.randn(n, device="cuda") * 0.1 for n in sizes]
m = [torch.zeros(n, device="cuda") for n in sizes]
v = [torch.zeros(n, device="cuda") for n in sizes]
g = [torch.randn(n, device="cuda") * 0.05 for n in sizes]
sh = [torch.zeros(n, device="cuda", dtype=torch.bfloat16) for n in sizes]
tab = (L.XpAdamTensor * len(sizes))()
cmap = []
for j, n in enumerate(sizes):
    tab[j].p, tab[j].m, tab[j].v, tab[j].shadow, tab[j].shadow_dtype, tab[j].numel = p[j].data_ptr(), m[j].data_ptr(), v[j].data_ptr(), sh[j].data_ptr(), L.XP_BF16, n
    cmap += [x for c in range(-(-n // L.XP_OPT_CHUNK)) for x in (j, c)]
table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda()
chunk_map = torch.tensor(cmap, dtype=torch.int32).cuda()
n_chunks = len(cmap) // 2
grads = (C.c_void_p * len(sizes))(*[t.data_ptr() for t in g])
grp = (C.c_uint8 * len(sizes))(*[0] * len(sizes))
groups = (L.XpAdamGroup * 1)(L.XpAdamGroup(1e-3, 0.9, 0.98, 1e-6, 0.05, 2e-3))
st = torch.cuda.current_stream().cuda_stream
args = (table.data_ptr(), chunk_map.data_ptr(), n_chunks, grads, grp, len(sizes), groups, 1, None, 0, 0.0)
gb = sum(sizes) * 30 / 1e9


def timed(f, reps=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


t = timed(lambda: L.check(lib.xp_adamw_step(*args, None, st), "xp_adamw_step"))
print(f"{gb:.2f} GB per pass, {n_chunks} chunks")
print(f"xp_adamw_step (one workgroup per chunk)      {t:8.3f} ms  {gb / t:7.2f} TB/s")
for w in wgs:
    t = timed(lambda: L.check(lib.xp_adamw_step_confined(*args, w, st), "xp_adamw_step_confined"))
    print(f"xp_adamw_step_confined, {w:3d} workgroups        {t:8.3f} ms  {gb / t:7.2f} TB/s  {gb / t / w * 1e3:6.1f} GB/s per CU")
