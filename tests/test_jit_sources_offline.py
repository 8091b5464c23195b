"""CPU-only: the GENERATORS of the run-time specialised kernels (csrc/expr_jit.hpp: expression tree, projection list behind a selection,
one-pass selection + projection, streaming aggregate under a predicate tree) produce sources that hipRTC compiles for gfx950 — with the
options the library passes — without spilling a register.  No GPU needed: tools/jit_offline builds representative programs by hand,
prints the sources and compiles each one (hipRTC compiles offline).  What the GPU tests then add is that the kernels compute the
right thing."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_kernels_compile_offline_without_spills(tmp_path):
    out = subprocess.run([os.path.join(ROOT, "tools", "jit_offline", "run.sh"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rows = {}
    for line in out.stdout.splitlines():
        m = re.match(r"(\S+) (\S+)s vgprs=(\d+) vgpr_spills=(\d+) scratch=(\d+)", line)
        if m:
            rows[m.group(1)] = (float(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)))
    assert set(rows) == {"nqe_jit_expr", "nqe_jit_expr_nulls", "nqe_jit_proj", "nqe_jit_selproj", "nqe_jit_agg", "nqe_jit_agg_u64key_i64val"}, out.stdout
    for name, (secs, vgprs, spills, scratch) in rows.items():
        assert spills == 0 and scratch == 0, (name, rows[name])
        assert 0 < vgprs <= 128, (name, vgprs)           # the aggregate kernel runs 1024-thread workgroups: 128 VGPRs at most
        assert secs < 30, (name, secs)                    # a tree shape costs about half a second of compilation
    # the aggregate kernel's LDS atomics are the native f64 ones (-munsafe-fp-atomics), not compare-and-swap loops
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", str(tmp_path / "nqe_jit_agg.co")], capture_output=True, text=True).stdout
    assert "ds_add_f64" in dis and "ds_min_f64" in dis and "ds_max_f64" in dis and "ds_cmpst" not in dis
