"""CPU: every script under tools/, examples/ and the repo root parses and byte-compiles (they run on the GPU box only; a syntax or
import-time name error there would cost a GPU call to find), and bench.py's host-only helpers behave."""
import glob
import os
import py_compile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scripts_byte_compile(tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "examples", "train", "*.py")) +
                   [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])
    assert len(files) >= 14
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)


def test_step_breakdown_tool_on_a_synthetic_trace(tmp_path):
    import csv
    import subprocess
    import sys
    rows, t = [], 0
    for step in range(8):
        for name, dur in (("void (anonymous namespace)::af_stats((anonymous namespace)::Tab)", 100_000), ("gemm_bf16_nt_256_kernel<true, 2>(GemmArgs)", 300_000),
                          ("gemm_bf16_nt_256_kernel<true, 2>(GemmArgs)", 300_000), ("swiglu_fwd_k(...)", 70_000)):
            rows.append({"Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + dur})
            t += dur + 2_000
        t += 30_000_000
    p = tmp_path / "trace.csv"
    with open(p, "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0]))
        wr.writeheader()
        wr.writerows(rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_breakdown.py"), str(p)], capture_output=True, text=True, check=True).stdout
    assert "gemm_bf16_nt_256_kernel<true, 2>,2.0,300.0" in out and "steady-state steps" in out
