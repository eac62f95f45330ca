#!/usr/bin/env python
"""Interleaved whole-step A/B on ONE box -- the measurement kernel choices are made on.

    python tools/instep_ab.py [--rounds 3] [--steps 20] [--out gpurun_out/ab.txt] VARIANT [VARIANT ...]

VARIANT = name[@tree][:ENV=value[,ENV=value...]][/--bench-flag=value ...]
          e.g.   default   one:XPRETRAIN_FWD_SPLIT=0   r2@_ab_r2   pf:XPRETRAIN_BENCH_FORCE_COLLECTIVES=1/--prefetch=1/--prefetch-dtype=uint8
          (tree: another checkout of this repository with its library built, relative to the repository root; default: this one)

Every round runs `python bench.py --no-cpu-baseline --steps N` once per variant, in the order given; the table lists pairs/s,
ms/step and the two forward-only probes per round, then the per-variant median.  Boxes of the pool differ by +-3 % on one build and
isolated kernel timings rank power-limited kernels differently from the step: only numbers from one invocation are comparable."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_variant(text):
    text, *flags = text.split("/")
    name, _, env = text.partition(":")
    name, _, tree = name.partition("@")
    envs = dict(kv.split("=", 1) for kv in env.split(",") if kv)
    return name, os.path.join(ROOT, tree) if tree else ROOT, envs, flags


def run(tree, envs, steps, flags=()):
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", str(steps), *flags], cwd=tree, env=dict(os.environ, **envs),
                         capture_output=True, text=True, timeout=900)
    line = next((l for l in out.stdout.splitlines() if l.startswith("{")), None)
    if line is None:
        raise RuntimeError(f"bench.py printed no JSON line in {tree} with {envs}:\n{out.stderr[-2000:]}")
    return json.loads(line)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    variants = [parse_variant(v) for v in a.variants]
    res = {name: [] for name, _, _, _ in variants}
    for _ in range(a.rounds):
        for name, tree, envs, flags in variants:
            res[name].append(run(tree, envs, a.steps, flags))
    keys = ("value", "ms_per_step", "vit_forward_ms", "vit_forward_train_mode_ms")
    lines = [f"bench.py --steps {a.steps} --no-cpu-baseline, one box, {a.rounds} interleaved rounds: pairs/s ms/step vit_fwd_ms vit_fwd_train_mode_ms"]
    w = max(len(v) for v in a.variants)
    for (name, _, _, _), text in zip(variants, a.variants):
        cells = [" ".join("-" if d.get(k) is None else f"{d[k]:.3f}" for k in keys) for d in res[name]]
        med = statistics.median(d["ms_per_step"] for d in res[name])
        lines.append(f"{text:{w}s}  " + " | ".join(cells) + f"  || median {med:.3f} ms/step")
    report = "\n".join(lines)
    print(report)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(report + "\n")


if __name__ == "__main__":
    main()
