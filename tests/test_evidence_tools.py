"""The evidence scripts whose output DESIGN.md quotes are code too: tools/kernel_gaps.py on a hand-made kernel trace."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_idle_time_report_of_a_kernel_trace(tmp_path):
    """three streams' kernels merged into one timeline: overlapping kernels are busy time once, a gap is counted only while NO kernel runs and
    is charged to the kernel that ended last before it and the one that starts after it; (anonymous namespace) and arguments are stripped"""
    d = tmp_path / "run" / "host"
    d.mkdir(parents=True)
    rows = [  # (start ns, end ns, name)
        (1_000_000, 2_000_000, "void ss::a_kernel<3>(int)"),
        (1_500_000, 2_500_000, "ss::(anonymous namespace)::b_kernel(ss::Args)"),        # overlaps a: busy until 2.5 ms
        (2_900_000, 3_000_000, "void ss::a_kernel<3>(int)"),                             # 400 us idle after b
        (3_000_000, 3_100_000, "ss::c_kernel()"),                                        # back to back: no gap
        (3_150_000, 3_200_000, "ss::c_kernel()"),                                        # 50 us
    ]
    with open(d / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for s, e, name in rows:
            w.writerow(["KERNEL_DISPATCH", name, s, e])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_gaps.py"), str(tmp_path / "run"), "100", "1.0"],
                         capture_output=True, text=True, env=dict(os.environ, GAPS_SEQUENCE="1"))
    assert out.returncode == 0, out.stderr
    text = out.stdout
    assert "window 2.20 ms: device busy 1.75 ms, idle 0.45 ms in 2 gaps" in text, text
    assert "gaps under 100 us: 1, 0.05 ms together" in text
    assert "ss::b_kernel" in text and "anonymous" not in text and "(int)" not in text
    # the 400 us gap is between b (which ended last) and a
    line = [ln for ln in text.splitlines() if "400.0 us" in ln]
    assert line and "after ss::b_kernel" in line[0] and "before void ss::a_kernel<3>" in line[0], text
    pair = [ln for ln in text.splitlines() if "-> void ss::a_kernel<3>" in ln]
    assert pair and pair[0].split()[0] == "0.400" and pair[0].split()[1] == "1", text
