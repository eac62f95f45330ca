#!/usr/bin/env python
"""Population of the bf16 path's loss deviation over input draws (VERDICT r3 "next" #6).  CPU, oracle only.

For each BASELINE shape at batch 2 and each of N input seeds (weights fixed: tests/gpu_util.seeded_model), the oracle computes the loss
  fp32        -- the reference arithmetic (pinned to the reference itself to 1e-4 on the loss, tests/test_oracle_golden.py)
  bf16 all    -- bf16 storage at every point the HIP path stores bf16, residual stream included (the round-2 build)
  bf16+proxy  -- the same with the M proxy rows of the residual stream kept in fp32 (the round-3/4 build: fp32 side rows)
and prints |loss - fp32| per seed, then mean / max over the seeds.  One draw is one sample of a distribution: a single case moving from
6.9e-3 to 2.0e-2 between two builds (configs[3], round 2 -> round 3) is a statement about that draw only if the population did not move.

    python tools/loss_seed_study.py [cfg2,cfg3,cfg4] [n_seeds]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clipvip_oracle as O  # noqa: E402
from tests.gpu_util import seeded_model  # noqa: E402
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("rpe", os.path.join(ROOT, "tools", "residual_precision_experiment.py"))

SHAPES = {"cfg2": (12, 224, 12), "cfg3": (8, 448, 12), "cfg4": (32, 224, 12)}      # frames, res, temporal_size of the config
which = (sys.argv[1] if len(sys.argv) > 1 else "cfg2,cfg3,cfg4").split(",")
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6

MODE = {"m": "all"}


def layer(x, sd, pfx, heads, size, pad_mask):
    """encoder layer with selectable residual-stream rounding (as tools/residual_precision_experiment.py)"""
    R = O.ROUND
    video = size is not None

    def rnd(t):
        if MODE["m"] == "all":
            return R(t)
        if not video:
            return t                                  # the text tower's whole stream is a side buffer in the HIP path
        Mp = size[0]
        return torch.cat([t[:, :Mp], R(t[:, Mp:])], dim=1)
    h = O.layer_norm(x, sd[pfx + "layer_norm1.weight"], sd[pfx + "layer_norm1.bias"])
    x = rnd(x + O.attention_block(h, sd, pfx + "self_attn.", heads, size, pad_mask))
    h = O.layer_norm(x, sd[pfx + "layer_norm2.weight"], sd[pfx + "layer_norm2.bias"])
    h = O.linear(R(O.quick_gelu(O.linear(h, sd, pfx + "mlp.fc1"))), sd, pfx + "mlp.fc2")
    return rnd(x + h)


orig = O.encoder_layer
torch.set_num_threads(max(1, (os.cpu_count() or 8)))
for name in which:
    frames, res, tsize = SHAPES[name]
    cfgd = O.vit_b_config(16, res)
    model = seeded_model(cfgd, tsize)
    sd = {k: v.detach() for k, v in O.strip_prefix(model.state_dict()).items()}
    cfg = O.OracleCfg.from_hf_dict(cfgd, temporal_size=tsize)
    rows = []
    t0 = time.time()
    for seed in range(nseeds):
        video, ids, mask = O.synthetic_inputs(2, frames, res, 32, seed=4321 + 1000 * seed)
        out = {}
        for mode in ("fp32", "all", "proxy"):
            O.encoder_layer = orig if mode == "fp32" else layer
            MODE["m"] = mode
            O.ROUND.dtype = None if mode == "fp32" else torch.bfloat16
            try:
                with torch.no_grad():
                    loss, vis, txt = O.full_step(video, ids, mask, sd, cfg)
            finally:
                O.ROUND.dtype = None
                O.encoder_layer = orig
            out[mode] = (loss.item(), vis, txt)
        f = out["fp32"]
        row = [seed, f[0]]
        for mode in ("all", "proxy"):
            l, v, t = out[mode]
            row += [abs(l - f[0]), (v - f[1]).abs().max().item(), (t - f[2]).abs().max().item()]
        rows.append(row)
        print(f"{name} seed {4321 + 1000 * seed}: fp32 loss {f[0]:.5f} | bf16 all: |d loss| {row[2]:.2e} |d vis| {row[3]:.2e} |d txt| {row[4]:.2e} | "
              f"bf16 + fp32 proxy rows: |d loss| {row[5]:.2e} |d vis| {row[6]:.2e} |d txt| {row[7]:.2e}   ({time.time() - t0:.0f} s)", flush=True)
    for label, c in (("bf16 all", 2), ("bf16 + fp32 proxy rows", 5)):
        d = [r[c] for r in rows]
        rel = [r[c] / abs(r[1]) for r in rows]
        print(f"{name} {label:24s}: |d loss| mean {sum(d) / len(d):.2e}  max {max(d):.2e}  (relative: mean {100 * sum(rel) / len(rel):.2f} %  max {100 * max(rel):.2f} %)  "
              f"|d vis| max {max(r[c + 1] for r in rows):.2e}  |d txt| max {max(r[c + 2] for r in rows):.2e}", flush=True)
