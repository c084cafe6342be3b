"""Host-side check of the safegcd inversion (sandstorm_amd/csrc/inv252.h, host+device code):
20 000 inputs including 0, 1, p-1 against the field's own Fermat inversion (fp252.h)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_safegcd_matches_fermat(tmp_path):
    exe = str(tmp_path / "inv_safegcd_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "sandstorm_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", "inv_safegcd_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad = 0" in out.stdout
