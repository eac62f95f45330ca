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


def test_register_parser():
    assert check_isa.regs_of("v_mfma_f32_16x16x32_bf16 v[2:5], v[8:11], v7, v[2:5]") == {2, 3, 4, 5, 7, 8, 9, 10, 11}
