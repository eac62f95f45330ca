"""CPU: the generated-code lint that guards the inline-asm LDS transpose reads (tools/check_isa.py) must flag a use of the
destination registers before `s_waitcnt lgkmcnt(0)` and accept the waited form."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
check_isa = importlib.util.module_from_spec(spec)
spec.loader.exec_module(check_isa)

GOOD = """
kernel:
\t;;#ASMSTART
\tds_read_b64_tr_b16 v[8:9], v7 offset:0
\t;;#ASMEND
\t;;#ASMSTART
\tds_read_b64_tr_b16 v[10:11], v7 offset:2048
\t;;#ASMEND
\tv_add_u32_e32 v20, v21, v22
\ts_waitcnt lgkmcnt(0)
\tv_mfma_f32_16x16x32_bf16 v[2:5], v[8:11], v[8:11], v[2:5]
\ts_endpgm
"""
BAD = GOOD.replace("\tv_add_u32_e32 v20, v21, v22\n\ts_waitcnt lgkmcnt(0)\n", "\tv_mov_b32_e32 v30, v9\n\ts_waitcnt lgkmcnt(0)\n")
BUILTIN = """
kernel:
\tds_read_b64_tr_b16 v[8:9], v7
\tv_mov_b32_e32 v30, v9
\ts_endpgm
"""     # compiler-emitted read (no ASMSTART marker): hipcc tracks its own waits, not the lint's business


# hipcc rotates loops: the block with the wait is laid out BEFORE the reads and reached by a branch; what follows the loop in the
# file (here a use of v9 for something else) is not on any path from the reads without the wait
ROTATED = """
kernel:
\ts_branch .LBB0_2
.LBB0_1:
\ts_waitcnt lgkmcnt(0)
\tv_mfma_f32_16x16x32_bf16 v[2:5], v[8:11], v[8:11], v[2:5]
\ts_cbranch_scc1 .LBB0_3
.LBB0_2:
\t;;#ASMSTART
\tds_read_b64_tr_b16 v[8:9], v7 offset:0
\t;;#ASMEND
\ts_branch .LBB0_1
.LBB0_3:
\tv_mov_b32_e32 v9, v1
\ts_endpgm
"""
# one arm of a conditional branch reaches a use without the wait
BRANCH_BAD = """
kernel:
\t;;#ASMSTART
\tds_read_b64_tr_b16 v[10:11], v5 offset:0
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_2
\tv_add_f32 v1, v10, v2
.LBB0_2:
\ts_waitcnt lgkmcnt(0)
\tv_add_f32 v1, v10, v2
\ts_endpgm
"""


def _run(tmp_path, text):
    f = tmp_path / "k.s"
    f.write_text(text)
    return check_isa.check(str(f))


def test_lint_accepts_waited_reads(tmp_path):
    assert _run(tmp_path, GOOD) == 0
    assert _run(tmp_path, BUILTIN) == 0


def test_lint_flags_use_before_wait(tmp_path, capsys):
    assert _run(tmp_path, BAD) == 1
    assert "before s_waitcnt lgkmcnt(0)" in capsys.readouterr().out


def test_lint_follows_the_control_flow(tmp_path, capsys):
    assert _run(tmp_path, ROTATED) == 0
    assert _run(tmp_path, BRANCH_BAD) == 1
    assert ":7:" in capsys.readouterr().out


def test_register_parser():
    assert check_isa.regs_of("v_mfma_f32_16x16x32_bf16 v[2:5], v[8:11], v7, v[2:5]") == {2, 3, 4, 5, 7, 8, 9, 10, 11}


def test_packed_fp32_op_sel_lint(tmp_path):
    """lint (1): a packed fp32 instruction whose low half takes a high source dword is rejected in any file; op_sel_hi
    redirection (high half <- low dword) and unmodified packed ops are accepted (measured exact)."""
    ok = tmp_path / "ok.s"
    ok.write_text("k:\n\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                  "\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7]\n\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3] op_sel_hi:[0,1,1]\n\ts_endpgm\n")
    assert check_isa.check_pk(str(ok)) == 0
    for form in ("v_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]",
                 "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]",
                 "v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3] op_sel:[1,0,0]"):
        bad = tmp_path / "bad.s"
        bad.write_text(f"k:\n\t{form}\n\ts_endpgm\n")
        assert check_isa.check_pk(str(bad)) == 1, form


def test_built_kernels_pass_the_lints():
    """the .s files the build left behind (make runs the lint; this keeps the result visible in the CPU suite)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "xpretrain_amd", "csrc", "build", "*.s")))
    if not files:
        import pytest
        pytest.skip("no generated code in the tree (run __graft_entry__.build())")
    assert sum(check_isa.check_pk(f) for f in files if not f.endswith("probe.s")) == 0


def test_workload_module_matches_the_oracle_definitions():
    """bench.py and the tools build their model / inputs / FLOP count from xpretrain_amd.workload (product side); the oracle keeps its
    own statement of the same things (test side): same config dict, same tensors from the same seed, same FLOPs"""
    import torch
    from oracle import clipvip_oracle as O
    from xpretrain_amd import workload as Wk
    for patch, image in ((16, 224), (32, 224), (16, 448)):
        assert Wk.vit_b_config(patch, image) == O.vit_b_config(patch, image)
    for args in ((2, 2, 32, 16), (3, 4, 64, 32)):
        for a, b in zip(Wk.synthetic_inputs(*args, seed=7), O.synthetic_inputs(*args, seed=7)):
            assert a.dtype == b.dtype and torch.equal(a, b)
    for args in ((12, 224, 32, 16), (8, 448, 32, 16), (2, 224, 16, 32)):
        assert Wk.flops_per_pair(*args) == O.flops_per_pair(*args)


def test_kernel_by_grid_separates_the_shapes_of_one_kernel(tmp_path):
    """tools/kernel_by_grid.py: one row per (kernel, workgroup count) of a rocprofv3 kernel trace -- the 888-workgroup fc1 launches
    are read apart from the 444-workgroup half-batch launches of the same kernel"""
    import csv
    import importlib.util
    spec = importlib.util.spec_from_file_location("kbg", os.path.join(ROOT, "tools", "kernel_by_grid.py"))
    kbg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kbg)
    name = "void (anonymous namespace)::gemm256_kernel<false, false>(xpgemm::KParams)"
    rows = [dict(Kernel_Name=name, Start_Timestamp=0, End_Timestamp=118000 + 2000 * i, Workgroup_Size_X=512, Workgroup_Size_Y=1,
                 Workgroup_Size_Z=1, Grid_Size_X=888 * 512, Grid_Size_Y=1, Grid_Size_Z=1) for i in range(3)]
    rows += [dict(Kernel_Name=name, Start_Timestamp=0, End_Timestamp=96000, Workgroup_Size_X=512, Workgroup_Size_Y=1, Workgroup_Size_Z=1,
                  Grid_Size_X=444 * 512, Grid_Size_Y=1, Grid_Size_Z=1)]
    rows += [dict(Kernel_Name="ln_fwd", Start_Timestamp=0, End_Timestamp=13000, Workgroup_Size_X=256, Workgroup_Size_Y=1, Workgroup_Size_Z=1,
                  Grid_Size_X=256 * 100, Grid_Size_Y=1, Grid_Size_Z=1)]
    path = tmp_path / "trace.csv"
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    out = kbg.summarize(str(path), ("gemm256",))
    assert [(r[2], r[4], r[5]) for r in out] == [(888, 3, 120.0), (444, 1, 96.0)]


def test_bench_parses_what_rccl_reports_it_chose():
    """bench.py's data-parallel diagnostics: channels / algorithm / protocol out of RCCL's NCCL_DEBUG=INFO (INIT,TUNING) lines"""
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    log = "\n".join([
        "node:11:22 [3] NCCL INFO Channel 00/32 :    0   1   2   3   4   5   6   7",
        "node:11:22 [3] NCCL INFO Channel 01/32 :    0   1   2   3   4   5   6   7",
        "node:11:22 [3] NCCL INFO 32 coll channels, 0 collnet channels, 0 nvls channels, 32 p2p channels, 2 p2p channels per peer",
        "node:11:22 [3] NCCL INFO AllReduce: 67108864 Bytes -> Algo 1 proto 2 time 512.3",
        "node:11:22 [3] NCCL INFO AllReduce: 67108864 Bytes -> Algo 1 proto 2 time 512.3",
        "node:11:22 [3] NCCL INFO AllGather: 16384 Bytes -> Algo 1 proto 0 time 12.3",
        "node:11:22 [3] NCCL INFO NCCL_MAX_NCHANNELS set by environment to 32.",
        "node:11:22 [3] NCCL INFO NCCL_ALGO set by environment to Ring"])
    got = bench.parse_rccl_log(log)
    assert got["coll_channels"] == 32 and got["ring_channel_lines"] == 2
    assert got["tuning"] == [{"op": "AllReduce", "bytes": 67108864, "algo": "Ring", "proto": "Simple"},
                             {"op": "AllGather", "bytes": 16384, "algo": "Ring", "proto": "LL"}]
    assert got["env_honoured"] == {"NCCL_MAX_NCHANNELS": "32", "NCCL_ALGO": "Ring"}
    assert bench.parse_rccl_log("")["coll_channels"] is None
