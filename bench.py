#!/usr/bin/env python
"""Headline benchmark: CLIP-ViP video-text contrastive training step on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N>1: the script starts its own N ranks -- `python bench.py --gpus 8` re-executes itself under torch.distributed.run
     with one process per GPU on 127.0.0.1; under an external launcher -- python -m torch.distributed.run --nnodes=1
     --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ..., WORLD_SIZE set -- it just joins the group)

One "step" = clamp logit_scale -> VidCLIP.forward (ViT-B/16 video tower with video-proxy tokens + CLIP text
tower) -> packed all-gather of features (N>1) -> NCELearnableTempLoss -> backward -> bucketed gradient
all-reduce overlapped with backward -> global-norm clip 5.0 -> AdamW step, on BASELINE config #2 per GPU:
B=8 pairs, T=12 frames 224x224, 32 text tokens, bf16 compute / fp32 master weights, synthetic inputs resident
in HBM, random-init weights of the named architecture.  Weak scaling: per-GPU batch fixed.

Prints ONE JSON line on rank 0 (metric/value/..., plus `roofline` for the dominant kernel and `cpu_baseline`).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
SURVEY_REFERENCE_CPU = {"pairs_per_s": 0.229, "cores": 8, "what": "the UNMODIFIED reference (VidCLIP + NCELearnableTempLoss, fwd+bwd, "
                        "fp32, B=8 of this config) timed in the build container, BASELINE.md 2b"}


def pmc_traffic(variant):
    """HBM-side bytes per launch of a 256x256-family GEMM from the newest kept PMC summary (profiles/r*_pmc_gemm256.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/gemm256_probe.py; FETCH_SIZE x 2 is the microarch
    guide's gfx950 correction for wide reads, confirmed on the xp_cast calibration launch of the same run).  Returns
    (bytes | None, note): stale evidence -- a summary whose source stamp differs from the GEMM sources in the tree -- is
    refused, not printed."""
    import glob
    import importlib.util
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_gemm256.json")))
    if not files:
        return None, "no profiles/r*_pmc_gemm256.json"
    spec = importlib.util.spec_from_file_location("pmcj", os.path.join(ROOT, "tools", "pmc_gemm256_json.py"))
    pmcj = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pmcj)
    d = json.load(open(files[-1]))
    if d.get("source_stamp") != pmcj.source_stamp():
        return None, f"{os.path.basename(files[-1])} is stale (GEMM sources changed since it was measured)"
    k = d["kernels"].get(variant, {})
    if "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
        return None, f"{os.path.basename(files[-1])} has no {variant} row"
    cal = d["kernels"].get("xp_cast_calibration", {})
    note = f"{os.path.basename(files[-1])}: FETCH_SIZE {k['FETCH_SIZE']:.0f} KiB x2 + WRITE_SIZE {k['WRITE_SIZE']:.0f} KiB"
    if cal.get("FETCH_SIZE"):
        note += f"; calibration xp_cast (256 MiB read / 128 MiB written): FETCH_SIZE x2 = {2 * cal['FETCH_SIZE'] / 1024:.0f} MiB, WRITE_SIZE = {cal.get('WRITE_SIZE', 0) / 1024:.0f} MiB"
    return int((2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024), note


def pmc_vit_forward():
    """Aggregate MFMA-busy fraction of the video tower's training-mode forward from the newest kept PMC summary
    (profiles/r*_pmc_vit_forward.json: one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass over tools/fwd_only.py,
    summarised by tools/pmc_vit_forward.py).  (fraction | None, note)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_vit_forward.json")))
    if not files:
        return None, "no profiles/r*_pmc_vit_forward.json"
    d = json.load(open(files[-1]))
    return d.get("mfma_busy_frac"), f"{os.path.basename(files[-1])}: {d.get('note', '')}"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU (BASELINE cfg #2: 8)")
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--patch", type=int, default=16)
    ap.add_argument("--txt-len", type=int, default=32)
    ap.add_argument("--opt-overlap", type=int, default=3, help="K >= 0: AdamW updates the encoder layers >= K of both towers on a stream of "
                    "its own and the next forward waits for them in front of layer K (optimization.AdamW.overlap_next_forward; default 3: -0.1 ms per step, profiles/r06k_*); -1: off")
    ap.add_argument("--graph", type=int, default=0, help="1: capture the whole step in a HIP graph and replay it "
                    "(works; measured 23.3 vs 23.1 ms/step eager on MI355X -- the step is GPU-bound and replaying a "
                    "600-node multi-stream graph costs the host as much as the eager launches); 0 (default): eager")
    ap.add_argument("--prefetch", type=int, default=0, help="0 (default, the contract's line): the batch is resident in HBM; 1: every step "
                    "takes its batch from xpretrain_amd.utils.prefetch.PrefetchLoader over pinned host tensors, copied on the loader's OWN stream "
                    "while the previous step computes (the reference loop's stream environment, dataloader.py:95-157); 2: the same on the "
                    "text tower's stream, behind the tower's forward (no additional stream); 3: stream='auto' (2 when data-parallel, else 1).  "
                    "The line then carries `prefetch` and is NOT the contract's value.")
    ap.add_argument("--prefetch-dtype", default="fp32", choices=["fp32", "uint8"], help="host frames: normalised fp32 (57.8 MB per step at "
                    "cfg #2) or decoded uint8 (14.5 MB; normalisation inside the patch-GEMM loader)")
    ap.add_argument("--bucket-mb", type=float, default=64.0, help="gradient bucket size of the all-reduce (MB of fp32 gradients)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-run", action="store_true", help="for rocprofv3 runs (tools/gpu_run.sh profile): no CPU baseline, no isolated "
                    "roofline launches and no forward-only probes, so that every 888-workgroup launch of the NT GEMM kernel in the trace is one of "
                    "the in-step fc1 launches `roofline.kernel_ms` is the median of (tools/kernel_by_grid.py)")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, join the process group, print the world size "
                    "measured by a collective and exit (no GPU work; backend gloo when there is no GPU) -- the launcher self-test")
    ap.add_argument("--cpu-baseline-batch", type=int, default=8)
    return ap.parse_args()


from xpretrain_amd.workload import ModelArgs as Args  # noqa: E402  (the args object VidCLIP's constructor reads)


def _time(f, iters):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # torch's current stream = the
    s.record()                                                                            # stream the kernels launch on
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def time_dominant_kernels(dev, rows, D=768, Dff=3072, iters=30):
    """The largest forward kernel (fc1 GEMM + bias + quick_gelu, two outputs: 26 % of forward FLOPs) and the largest
    backward kernel (dW1 = dpre^T . h2, both operands k-strided, split-K into fp32 slabs + the deterministic reduce), timed
    with HIP events on the stream they are launched on.  Returns {name: (TFLOP/s, ms)}."""
    from xpretrain_amd import hip_ops as H, _lib as L
    from xpretrain_amd.functional import _wgrad
    bf = torch.bfloat16
    A = torch.randn(rows, D, device=dev).to(bf)
    W = (torch.randn(Dff, D, device=dev) * 0.02).to(bf)
    bias = torch.zeros(Dff, device=dev)
    out = torch.empty(rows, Dff, dtype=bf, device=dev)
    aux = torch.empty(rows, Dff, dtype=bf, device=dev)
    dpre = (torch.randn(rows, Dff, device=dev) * 1e-3).to(bf)
    flops = 2.0 * rows * D * Dff
    ms_f = _time(lambda: H.gemm(A, W, rows, Dff, D, out=out, epilogue=L.EPI_BIAS_GELU, bias=bias, aux=aux), iters)
    ms_b = _time(lambda: _wgrad(dpre, A, rows, Dff, D, slack=True), iters)
    return {"fwd": (flops / (ms_f * 1e-3) / 1e12, ms_f), "bwd": (flops / (ms_b * 1e-3) / 1e12, ms_b)}


def host_cpu():
    """(model name, sockets, physical cores) from /proc/cpuinfo"""
    model, phys, cores = "unknown", set(), set()
    try:
        pid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip(); phys.add(pid)
            elif line.startswith("core id"):
                cores.add((pid, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    return model, max(1, len(phys)), max(1, len(cores)) if cores else (os.cpu_count() or 1)


def effective_cpus():
    """What this process may actually use: (cpus in the affinity mask, cgroup CPU quota in cpus or None, first NUMA node's cpus in
    the mask).  A container that shows 128 cores in /proc/cpuinfo but is limited to a few by cpu.max / cpuset would otherwise be
    oversubscribed by one thread per visible core (the round-3 line: 64 threads, 33 s per step)."""
    aff = sorted(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    node0 = aff
    try:
        cl = open("/sys/devices/system/node/node0/cpulist").read().strip()
        ids = set()
        for part in cl.split(","):
            a, _, b = part.partition("-")
            ids.update(range(int(a), int(b or a) + 1))
        inter = [c for c in aff if c in ids]
        if inter:
            node0 = inter
    except (OSError, ValueError):
        pass
    return aff, quota, node0


def cpu_baseline(args):
    """The CPU oracle (port of the reference algorithm, oracle/clipvip_oracle.py -- /root/reference does not exist on the
    GPU box) on this host: one fwd + loss + bwd step at the FULL local batch of the config (B = 8), fp32,
    threads = the physical cores of one socket; a B = 2 pass first warms the allocator / thread pools; best of the timed
    passes that fit ~25 s.  A reported baseline, not a target."""
    from oracle import clipvip_oracle as O
    torch.manual_seed(1234)
    from xpretrain_amd.modeling import VidCLIP
    model_name, sockets, cores = host_cpu()
    aff, quota, node0 = effective_cpus()
    # threads: the physical cores of ONE socket / NUMA node that this process may really use (affinity mask, cgroup quota);
    # SMT siblings excluded (half of the node's logical cpus when the node lists more than the socket has cores)
    per_socket = max(1, cores // sockets)
    threads = min(per_socket, len(node0))
    if quota is not None:
        threads = max(1, min(threads, int(quota)))
    old_threads = torch.get_num_threads()
    old_aff = os.sched_getaffinity(0)
    try:
        os.sched_setaffinity(0, set(node0))            # one NUMA node: the weights (600 MB) are first-touched there
    except OSError:
        pass
    torch.set_num_threads(threads)
    cfgd = O.vit_b_config(args.patch, args.res)
    model = VidCLIP(Args(cfgd))
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in O.strip_prefix(model.state_dict()).items()}
    del model
    cfg = O.OracleCfg.from_hf_dict(cfgd)

    def one(Bc):
        video, ids, mask = O.synthetic_inputs(Bc, args.frames, args.res, args.txt_len)
        for v in sd.values():
            v.grad = None
        t0 = time.time()
        loss, _, _ = O.full_step(video, ids, mask, sd, cfg)
        loss.backward()
        return time.time() - t0
    one(2)
    Bc = args.cpu_baseline_batch
    times = [one(Bc)]
    if times[0] > 12.0 and Bc > 2:       # a slow host: bound the sample (the metric is per pair; B = 2 is BASELINE's smallest batch
        Bc = 2                           # with a non-trivial contrastive loss) so that two timed passes still fit the budget
        times = [one(Bc)]
    while len(times) < 2 or (sum(times) + times[-1] < 25.0 and len(times) < 3):
        times.append(one(Bc))
    torch.set_num_threads(old_threads)
    try:
        os.sched_setaffinity(0, old_aff)
    except OSError:
        pass
    dt = min(times)
    return {"value": round(Bc / dt, 4), "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"oracle/clipvip_oracle.py fp32, fwd+loss+bwd (no optimizer) at B={Bc} of the same T={args.frames}/{args.res}^2/"
                      f"Lt={args.txt_len} ViT-B/{args.patch} config after a B=2 warm-up pass; best of {len(times)} "
                      f"({', '.join(f'{t:.1f}' for t in times)} s); {threads} threads pinned to NUMA node 0 "
                      f"({len(node0)} of the {len(aff)} cpus in the affinity mask; cgroup quota "
                      f"{'none' if quota is None else f'{quota:.1f} cpus'}; /proc/cpuinfo: {sockets} x {cores // sockets}-core "
                      f"{model_name}); torch.get_num_threads() was {old_threads}; torch {torch.__version__}",
            "reference_measured_in_build_container": SURVEY_REFERENCE_CPU}


def workload_tag(a, W):
    """which BASELINE.json config the shape corresponds to"""
    shape = (a.patch, a.frames, a.res, a.txt_len)
    if shape == (16, 12, 224, 32):
        return "BASELINE configs[1]" if W == 1 else "BASELINE configs[2] (configs[1] per GPU)"
    if shape == (16, 8, 448, 32):
        return "BASELINE configs[3] shape"
    if shape == (16, 32, 224, 32):
        return "BASELINE configs[4] shape"
    return "custom shape"


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` (one rank per GPU; the reference is started as
    `horovodrun -np $NUM_GPUS python src/pretrain/run_pretrain.py`, CLIP-ViP/README.md:58).  Does not return."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def parse_rccl_log(text):
    """channels, the environment settings RCCL says it honoured, and (when the TUNING subsystem was on) algorithm / protocol per
    message size, from RCCL's NCCL_DEBUG=INFO lines; the raw lines it matched are kept"""
    import re
    out = {"coll_channels": None, "ring_channel_lines": 0, "env_honoured": {}, "tuning": [], "raw": []}
    algos = {0: "Tree", 1: "Ring", 2: "CollnetDirect", 3: "CollnetChain", 4: "NVLS", 5: "NVLSTree"}
    protos = {0: "LL", 1: "LL128", 2: "Simple"}
    seen = set()
    for line in text.splitlines():
        m = re.search(r"(\d+) coll channels", line)
        if m:
            out["coll_channels"] = int(m.group(1)); out["raw"].append(line.strip()[-160:])
        if re.search(r"NCCL INFO Channel \d+/\d+\s*:", line):
            out["ring_channel_lines"] += 1
        m = re.search(r"((?:NCCL|RCCL)_\w+) set by environment to (\S+?)\.?$", line.strip())
        if m:
            out["env_honoured"][m.group(1)] = m.group(2)
        m = re.search(r"(\w+): (\d+) Bytes -> Algo (\d+) proto (\d+)", line)
        if m:
            key = (m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)))
            if key not in seen and len(out["tuning"]) < 12:
                seen.add(key)
                out["tuning"].append({"op": key[0], "bytes": key[1], "algo": algos.get(key[2], key[2]), "proto": protos.get(key[3], key[3])})
    return out


def dp_diagnostics(step, reducer, sync, dev, rank, W, n_seen, cu_budget, steps=3):
    """Per-rank facts of a data-parallel run, gathered on rank 0: the ranks this rank saw, what RCCL chose, the
    gradient buckets, and the EXPOSED tail of the gradient exchange -- main-stream time inside reducer.synchronize() (everything the
    backward did not hide), HIP events, mean of `steps` extra steps after the timed region."""
    import glob
    exposed = []
    real_sync = reducer.synchronize

    def timed_sync():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); real_sync(); e1.record()
        exposed.append((e0, e1))
    reducer.synchronize = timed_sync
    try:
        for _ in range(steps):
            step()
        sync()
    finally:
        reducer.synchronize = real_sync
    tail = [e0.elapsed_time(e1) for e0, e1 in exposed]
    info = {"rank": rank, "n_ranks_seen": n_seen, "device": str(dev), "gemm_cu_budget": cu_budget,
            "buckets": len(reducer.buckets), "bucket_mb": [round(b["n"] * 4 / 2 ** 20, 1) for b in reducer.buckets],
            "exposed_grad_exchange_tail_ms": round(sum(tail) / max(len(tail), 1), 3),
            "rccl_env": {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_ALGO", "NCCL_PROTO")}}
    pat = os.environ.get("NCCL_DEBUG_FILE", "").replace("%h", "*").replace("%p", str(os.getpid()))
    text = ""
    for f in glob.glob(pat) if pat else []:
        try:
            text += open(f, errors="replace").read()
        except OSError:
            pass
    info["rccl"] = parse_rccl_log(text) if text else "no RCCL debug file (NCCL_DEBUG_FILE unset or unreadable)"
    if W > 1:
        box = [None] * W
        torch.distributed.all_gather_object(box, info)
        return box
    return [info]


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus)
    from xpretrain_amd import distributed as D
    if a.gpus > 1:
        # RCCL defaults for one 8 x MI355X node (set before the communicator exists; an outer environment wins).  Channels: every
        # channel is one RCCL workgroup that owns a CU while a bucket all-reduce runs (profiles/r04a_rccl_gfx950_kernel_footprint.txt);
        # 32 of them leave 224 CUs, which still hold every GEMM grid of the step in the same number of rounds (222-tile dX GEMMs, 3 x 224
        # >= 666, 4 x 224 >= 888) -- the split-K planning is told below.  Ring + Simple: the buckets are 64 MB, bandwidth-bound.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "32")
        os.environ.setdefault("NCCL_ALGO", "Ring")
        os.environ.setdefault("NCCL_PROTO", "Simple")
    if a.gpus > 1 or os.environ.get("XPRETRAIN_BENCH_FORCE_COLLECTIVES", "") in ("1", "gather", "reducer"):
        # what RCCL actually chose (channels, algorithm, protocol per message size) goes to a per-process file that dp_diagnostics()
        # parses after the timed region: the first multi-GPU run then shows whether the defaults above were honoured
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() in ("VERSION", "WARN"):      # (the image exports NCCL_DEBUG=VERSION)
            os.environ["NCCL_DEBUG"] = "INFO"
        # INIT + ENV only: both print while the communicator is built, nothing per collective (TUNING would write a line per
        # enqueued collective -- inside the timed steps)
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,ENV")
        os.environ.setdefault("NCCL_DEBUG_FILE", os.path.join(os.environ.get("TMPDIR", "/tmp"), "xp_bench_rccl.%h.%p.log"))
    local_rank = D.init_from_env()
    W, rank = D.world_size(), D.rank()
    if W != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={W} ranks")
    n_seen = D.ranks_seen()
    if a.launch_check:
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": a.gpus, "world_size": W, "n_ranks_seen": n_seen,
                              "backend": torch.distributed.get_backend() if W > 1 else None}), flush=True)
        if W > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # probe switch (tools/instep_ab.py): a ONE-rank RCCL group with the collectives forced on -- the feature gather and the bucket
    # all-reduces go through the process group's stream and events as in a data-parallel run (no bytes leave the GPU); marked in the line
    fmode = os.environ.get("XPRETRAIN_BENCH_FORCE_COLLECTIVES", "")          # "1" | "gather" | "reducer" (cost attribution)
    forced = W == 1 and fmode in ("1", "gather", "reducer")
    if forced:
        torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", world_size=1, rank=0)
        D.FORCE_COLLECTIVES = True
    cu_budget = D.reserve_cus_for_collectives()          # 256 in a 1-rank run

    from xpretrain_amd import workload as O         # model dimensions, synthetic inputs, FLOP model
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import NCELearnableTempLoss

    torch.manual_seed(1234)
    cfgd = O.vit_b_config(a.patch, a.res)
    model = VidCLIP(Args(cfgd))
    with torch.no_grad():
        model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.02)
    model.to(dev).train()
    D.broadcast_parameters(model)
    loss_fn = NCELearnableTempLoss()
    import xpretrain_amd.functional as XF
    # buckets aligned to the encoder layers: the native layer backward writes its gradients straight into bucket storage
    reducer = D.GradBucketReducer(model.parameters(), bucket_mb=a.bucket_mb, average=True,
                                  layout_groups=XF.layer_grad_groups(model), segments=D.tower_segments(model),
                                  wire_dtype={"fp32": None, "bf16": torch.bfloat16}[os.environ.get("XPRETRAIN_GRAD_WIRE", "fp32")])
    if forced and fmode == "gather":        # the gradient reducer stays out of it
        reducer.remove(); reducer._active = False
    use_graph = a.graph == 1
    if use_graph and a.prefetch:
        raise SystemExit("bench.py: --prefetch needs the eager step (a captured graph replays fixed input addresses)")
    # pretrain_vip_base_16.json:68-80: adamw, betas (0.9, 0.98), lr 5e-6, wd 0.05, lr_mul 1, cosine decay with 1 % warmup,
    # grad_norm 5.0; grouping = optimization/utils.py:124-154
    from xpretrain_amd.optimization import AdamW, get_lr_sched, build_e2e_optimizer_w_lr_mul
    LR, TOTAL_STEPS = 5e-6, 100000
    groups = build_e2e_optimizer_w_lr_mul(list(model.named_parameters()), LR, 0.05, lr_mul=1, lr_mul_prefix="")
    opt = AdamW([g for g in groups if g["params"]], lr=LR, betas=(0.9, 0.98))
    # single-process step only: beside a process group's streams the optimizer's own stream is one stream too many (beyond
    # four the runtime shares hardware queues) -- through a one-rank RCCL group it costs +0.08 ms instead of saving 0.10
    # (profiles/r06o_in_step_ab_optimizer_overlap_under_forced_collectives.txt)
    opt_overlap = a.opt_overlap if (a.opt_overlap >= 0 and a.graph != 1 and W == 1 and not forced) else None
    if opt_overlap is not None:
        opt.overlap_next_forward(model, opt_overlap)
    sched_step = [1000]      # start past the warmup so the synthetic loss moves
    video, ids, mask = O.synthetic_inputs(a.batch, a.frames, a.res, a.txt_len, seed=4321 + rank)
    batches = None
    if a.prefetch:
        # the reference loop's hand-over: a small pool of pinned host batches (a DataLoader with pin_memory=True would produce these),
        # walked for ever; PrefetchLoader copies batch i+1 while step i runs
        from xpretrain_amd.utils.prefetch import PrefetchLoader
        pool = []
        for i in range(2):
            v, i_, m_ = O.synthetic_inputs(a.batch, a.frames, a.res, a.txt_len, seed=4321 + rank + 100 * i)
            if a.prefetch_dtype == "uint8":
                v = (v * 58.0 + 122.0).clamp_(0, 255).to(torch.uint8)
            pool.append({"video": v.pin_memory(), "text_input_ids": i_.pin_memory(), "text_input_mask": m_.pin_memory()})

        def forever():
            while True:
                for b in pool:
                    yield b
        batches = iter(PrefetchLoader(forever(), stream={1: None, 2: "text", 3: "auto"}[a.prefetch]))
    video, ids, mask = video.to(dev), ids.to(dev), mask.to(dev)
    logit_scale = model.clipmodel.logit_scale
    params = [p for p in model.parameters()]

    def step():
        nonlocal video, ids, mask
        if batches is not None:
            b = next(batches)
            video, ids, mask = b["video"], b["text_input_ids"], b["text_input_mask"]
        with torch.no_grad():
            logit_scale.clamp_(0, math.log(200.0))                       # run_pretrain.py:335-340
        out = model(video, ids, mask)
        if forced and fmode == "reducer":    # the feature gather stays out of it
            vis, txt = out["vis_features"], out["text_features"]
        else:
            vis, txt = D.gather_features(out["vis_features"], out["text_features"])
        loss = loss_fn(vis, txt, logit_scale)
        loss.backward()
        reducer.synchronize()
        lr_t = get_lr_sched(sched_step[0], "cosine", LR, TOTAL_STEPS, warmup_ratio=0.01)   # run_video_retrieval.py:372-383
        for g in opt.param_groups:
            g["lr"] = lr_t
        sched_step[0] += 1
        opt.clip_and_step(5.0)                     # clip_grad_norm_(…, 5.0) + AdamW.step + bf16 weight copies, one pass
        reducer.zero_grad()
        return loss

    def sync():
        if W > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    cap_stream = torch.cuda.Stream(device=dev) if use_graph else None
    if use_graph:                      # warm up on the stream the graph will be captured on (AccumulateGrad nodes
        cap_stream.wait_stream(torch.cuda.current_stream())      # bind to the stream of their first use)
        with torch.cuda.stream(cap_stream):
            for _ in range(a.warmup):
                loss = step()
        torch.cuda.current_stream().wait_stream(cap_stream)
    else:
        for _ in range(a.warmup):
            loss = step()
    sync()
    if use_graph:
        # The step has no host-side data dependence (no .item(), static shapes, workspaces cached), so the ~600 kernel
        # launches are captured once into a HIP graph and replayed: no Python / ctypes / launch cost per kernel.
        eager_step = step
        graph = torch.cuda.CUDAGraph()
        static = {}
        loss = None
        with torch.cuda.graph(graph, stream=cap_stream):
            static["loss"] = eager_step()

        def step():                                                       # noqa: F811
            graph.replay()
            return static["loss"]
        for _ in range(2):
            loss = step()
        sync()
    c_proc0, c_main0 = time.process_time(), time.thread_time()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    t_enq = time.perf_counter() - t0          # WALL time until the last launch is queued: includes any back-pressure of the
    #                                           runtime's queues once the host runs ahead, so it is not a cost
    c_proc, c_main = time.process_time() - c_proc0, time.thread_time() - c_main0   # CPU seconds: all threads (forward thread +
    sync()                                    # autograd thread + runtime helpers) / the forward thread alone, before the sync
    dt = time.perf_counter() - t0
    if W > 1:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()
    dp_diag = dp_diagnostics(step, reducer, sync, dev, rank, W, n_seen, cu_budget) if (W > 1 or (forced and fmode != "gather")) else None

    # the dominant forward kernel timed WHERE IT RUNS: three more training steps with the library's GEMM timer armed for the fc1
    # shape (HIP events on the stream the kernel is launched on, around each of its 12 launches per step)
    # The shipped step runs the video tower's forward as TWO half-batch chains on two streams (functional.ForwardSplit): a launch
    # there shares the CUs with the other chain's kernels and its event-bracketed duration is not a per-kernel quantity.  The
    # roofline kernel is therefore timed in three steps run as ONE chain (XPRETRAIN_FWD_SPLIT=0: the full-batch launch alone on the
    # chip apart from the text tower's stream); the half-batch launch as it runs in the shipped step is reported beside it.
    import xpretrain_amd.functional as XF
    fc1_in_step_ms = fc1_two_chain_ms = None
    rows_ = a.batch * (4 + a.frames * (a.res // a.patch) ** 2)
    two_chains = XF.FWD_SPLIT and XF.LAYER_CALLS and a.batch % 2 == 0 and rows_ >= XF.FWD_SPLIT_MIN_ROWS

    def timed_gemm(M_, N_, K_, epi, aks, bks, split):
        """median HIP-event duration (ms) of the launches of ONE GEMM shape inside three training steps (xp_debug_gemm_timer brackets
        every matching xp_gemm call on the stream it is launched on -- the weight-gradient stream for the dW GEMMs)"""
        if rank == 0:
            L.check(L.lib().xp_debug_gemm_timer_arm(M_, N_, K_, epi, aks, bks, split, 64), "xp_debug_gemm_timer_arm")
        for _ in range(3):
            step()
        sync()
        if rank == 0:
            buf = (C.c_float * 64)()
            n_t = L.lib().xp_debug_gemm_timer_read(buf, 64)
            if n_t > 0:
                return statistics.median(buf[i] for i in range(min(n_t, 64)))
        return None
    import ctypes as C
    import statistics
    from xpretrain_amd import _lib as L
    if two_chains:
        fc1_two_chain_ms = timed_gemm(rows_ // 2, 3072, 768, L.EPI_BIAS_GELU, 0, 0, 1)
    # the dominant backward kernel where it runs: dW1 = dpre^T . h2 on the weight-gradient stream, beside the dX chain
    dw1_split = XF._split_for(3072, 768, rows_, torch.bfloat16, (0, 0, 0), True)      # (a "slack" launch of the layer's backward: csrc/layer.hip)
    dw1_in_step_ms = timed_gemm(3072, 768, rows_, L.EPI_NONE, 1, 1, dw1_split)
    saved = XF.FWD_SPLIT
    XF.FWD_SPLIT = False
    try:
        fc1_in_step_ms = timed_gemm(rows_, 3072, 768, L.EPI_BIAS_GELU, 0, 0, 1)
    finally:
        XF.FWD_SPLIT = saved

    def forward_probe():
        for _ in range(2):
            model.clipmodel.vision_model(pixel_values=video)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            model.clipmodel.vision_model(pixel_values=video)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / 5
    vit_fwd_ms = vit_fwd_train_ms = 0.0          # (--profile-run: not measured, printed as 0)
    if not a.profile_run:
        # ViT-forward-only time (north-star target: <= 3.4 ms at cfg #2), inference mode, weights cached
        with torch.no_grad():
            vit_fwd_ms = forward_probe()
        # the same pass as the training step runs it (activations and the MLP pre-activation kept, two half-batch chains)
        vit_fwd_train_ms = forward_probe()

    if rank == 0:
        f_vis, f_txt = O.flops_per_pair(a.frames, a.res, a.txt_len, a.patch)
        pairs_s = W * a.batch * a.steps / dt
        step_flops = 3.0 * (f_vis + f_txt) * a.batch                     # per GPU, fwd+bwd convention
        rows = a.batch * (4 + a.frames * (a.res // a.patch) ** 2)
        dom = time_dominant_kernels(dev, rows) if not a.profile_run else {"fwd": (0.0, 0.0), "bwd": (0.0, 0.0)}
        (k_tf_iso, k_ms_iso), (b_tf, b_ms) = dom["fwd"], dom["bwd"]
        # roofline of the dominant forward kernel: the in-step measurement (isolated launches of the same kernel are kept beside it:
        # back-to-back identical GEMMs run at a lower clock than the same kernel between the step's memory-bound neighbours)
        # Primary: the launch of the SHIPPED step -- the video tower's forward runs as two half-batch chains, so fc1 is a [rows/2] launch
        # (444 workgroups at cfg #2) that shares the chip with the other chain's kernels; secondary: the full-batch launch of a one-chain
        # step (the per-kernel quantity of rounds 1-5) and isolated launches.
        k1_ms = fc1_in_step_ms if fc1_in_step_ms else k_ms_iso
        k1_tf = 2.0 * rows * 768 * 3072 / (k1_ms * 1e-3) / 1e12
        shipped = two_chains and fc1_two_chain_ms
        k_rows = rows // 2 if shipped else rows
        k_ms = fc1_two_chain_ms if shipped else k1_ms
        k_tf = 2.0 * k_rows * 768 * 3072 / (k_ms * 1e-3) / 1e12
        mfma_busy, mfma_note = pmc_vit_forward()
        bw_ms = dw1_in_step_ms if dw1_in_step_ms else b_ms
        bw_tf = 2.0 * rows * 768 * 3072 / (bw_ms * 1e-3) / 1e12
        full = rows == 18848
        tr_f, note_f = pmc_traffic("NT_fc1_fwd") if full else (None, "PMC summary exists for the cfg #2 shape only")
        tr_b, note_b = pmc_traffic("SS_dw1") if full else (None, "PMC summary exists for the cfg #2 shape only")
        res = {
            "metric": "video-text pairs/sec", "value": round(pairs_s, 3), "unit": "pairs/s", "n_gpus": W,
            "n_ranks_seen": n_seen,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"CLIP-ViP ViT-B/{a.patch} video-text contrastive train step (fwd+loss+bwd+grad-sync+"
                                   f"clip+AdamW), {a.frames} frames {a.res}^2, {a.txt_len} text tokens, "
                                   f"local batch {a.batch}, " + workload_tag(a, W),
                       "global_batch": W * a.batch, "parallelism": f"dp{W}", "final_loss": round(final_loss, 4),
                       "gemm_cu_budget": cu_budget, "grad_wire": os.environ.get("XPRETRAIN_GRAD_WIRE", "fp32"),
                       "forced_one_rank_collectives": forced,
                       "prefetch": ({"loader": "xpretrain_amd.utils.prefetch.PrefetchLoader", "host_frames": a.prefetch_dtype,
                                     "copy_stream": {1: "own", 2: "text tower's", 3: "auto"}[a.prefetch]} if a.prefetch else None),
                       "video_forward_chains": 2 if two_chains else 1,
                       "optimizer_overlaps_next_forward_from_layer": opt_overlap,
                       "second_chain_stream": XF.second_chain_stream_mode() if two_chains else None,
                       "launch": "hipGraph replay of the captured step" if use_graph else "eager", "profile_run": bool(a.profile_run)},
            "step_tflops_per_gpu": round(step_flops / (dt / a.steps) / 1e12, 1),
            "step_frac_of_bf16_peak": round(step_flops / (dt / a.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "host_cpu_ms_per_step": round(c_proc / a.steps * 1e3, 3),              # CPU time of ALL host threads per step
            "host_main_thread_cpu_ms_per_step": round(c_main / a.steps * 1e3, 3),  # forward + optimizer thread alone
            "host_enqueue_wall_ms_per_step": round(t_enq / a.steps * 1e3, 3),      # wall incl. queue back-pressure; not a cost
            # the ViT forward as the training step runs it (activations and the MLP pre-activation kept): north_star's 0.40 target
            "vit_forward_train_mode_ms": round(vit_fwd_train_ms, 3),
            "vit_forward_frac_of_bf16_peak": round(f_vis * a.batch / (vit_fwd_train_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if vit_fwd_train_ms else 0.0,
            "vit_forward_ms": round(vit_fwd_ms, 3),
            "vit_forward_note": "vit_forward_ms = inference mode (torch.no_grad: no pre-activation kept; two half-batch chains like the training "
                                "pass since round 6); the fraction of peak is quoted on vit_forward_train_mode_ms, the kernels the benchmark step runs",
            # dominant forward kernel; algorithmic bytes = A + W + two bf16 outputs
            "roofline": {"bound": "mfma", "kernel": f"gemm256_kernel<NT> (256x256 tiles, the kernel the training step runs) fc1 +bias+quick_gelu, two bf16 outputs [{k_rows}x768]x[768x3072]",
                         "achieved": round(k_tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(k_tf / PEAK_BF16_TFLOPS, 4), "kernel_ms": round(k_ms, 4),
                         "kernel_ms_source": ("median of the kernel's launches inside 3 ordinary training steps after the timed region (HIP events on the "
                                              "launch stream, xp_debug_gemm_timer)" if (shipped or fc1_in_step_ms) else "30 isolated launches (HIP events)"),
                         # the shipped mode: a half-batch launch beside the OTHER chain's kernels -- a shared-chip duration, so `frac` is the
                         # rate of one of two concurrent chains, not of the chip; `one_chain` below is the per-kernel quantity of earlier rounds
                         "measured_in": "two-chain steps" if shipped else ("one-chain steps (XPRETRAIN_FWD_SPLIT=0)" if fc1_in_step_ms else "isolated launches"),
                         "workgroups": (k_rows + 255) // 256 * 12, "concurrent_chains": 2 if shipped else 1,
                         "rocprof_check": "profiles/r06zz_kernel_by_grid.txt: gemm256_kernel<false, false>, 444 workgroups (two-chain steps) / 888 (one-chain)",
                         "one_chain": {"kernel_ms": round(k1_ms, 4), "achieved": round(k1_tf, 1), "frac": round(k1_tf / PEAK_BF16_TFLOPS, 4),
                                       "measured_in": "one-chain steps (XPRETRAIN_FWD_SPLIT=0): the full-batch launch alone on the chip apart from the text tower" if fc1_in_step_ms else "isolated launches"},
                         "kernel_ms_isolated": round(k_ms_iso, 4), "frac_isolated": round(k_tf_iso / PEAK_BF16_TFLOPS, 4),
                         # every kernel of the video tower's training-mode forward: matrix-core busy cycles / (SIMDs x elapsed cycles)
                         "vit_forward_mfma_busy_frac": mfma_busy, "vit_forward_mfma_busy_source": mfma_note,
                         "traffic": tr_f, "traffic_unit": "bytes/launch (full-batch launch)", "traffic_source": note_f,
                         "algorithmic_bytes": (k_rows * 768 + 3072 * 768 + 2 * k_rows * 3072) * 2},
            # dominant backward kernel (incl. its split-K reduce); algorithmic bytes = dpre + h2 + fp32 dW
            # `frac` / `kernel_ms`: 30 isolated launches INCLUDING the split-K reduce (the quantity of rounds 1-4, comparable across rounds);
            # `in_step_gemm_only`: the GEMM kernel alone where it runs (the reduce -- 35 us in the step, profiles/r06z_kernel_by_grid.txt -- is
            # not inside the timer's bracket)
            "roofline_bwd": {"bound": "mfma", "kernel": f"gemm256_kernel<SS> dW1 = dpre^T.h2 [3072x{rows}]x[{rows}x768] split-K {dw1_split} into fp32 slabs",
                             "achieved": round(b_tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(b_tf / PEAK_BF16_TFLOPS, 4), "kernel_ms": round(b_ms, 4),
                             "kernel_ms_source": "30 isolated launches incl. the split-K reduce",
                             "in_step_gemm_only": None if not dw1_in_step_ms else {
                                 "kernel_ms": round(bw_ms, 4), "achieved": round(bw_tf, 1), "frac": round(bw_tf / PEAK_BF16_TFLOPS, 4),
                                 "source": "median of the GEMM kernel's launches inside 3 training steps (HIP events on the weight-gradient stream it is "
                                           "launched on, xp_debug_gemm_timer); it runs beside the dX chain of the main stream: a shared-chip duration",
                                 # the split-K planning fills at most 112 CUs here on purpose (csrc/gemm.hip::XP_SPLITK_FILL_SLACK): `frac` is a
                                 # whole-chip rate of a launch that holds `workgroups` of the 256 CUs
                                 "workgroups": 36 * dw1_split, "frac_of_held_cus": round(bw_tf / PEAK_BF16_TFLOPS * 256.0 / min(256, 36 * dw1_split), 4)},
                             "traffic": tr_b, "traffic_unit": "bytes/launch (GEMM kernel only)", "traffic_source": note_b,
                             "algorithmic_bytes": (rows * 3072 + rows * 768) * 2 + 3072 * 768 * 4},
        }
        if dp_diag is not None:                        # one entry per rank: what the data-parallel machinery did on it
            res["data_parallel"] = dp_diag
        if not a.no_cpu_baseline and not a.profile_run and W == 1:           # reported baseline, rank 0 of the single-GPU run only
            res["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(res), flush=True)
    if W > 1 or forced:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
