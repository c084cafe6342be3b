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


def test_lazy_curve_formulas_match_plain_ones(tmp_path):
    """sandstorm_amd/csrc/ec252.h: the lazy mixed addition the Pedersen kernels run, against the plain 8 x 32
    formulas on chains over multiples of P1, the exceptional cases, and the multiplier at the lazy limb bounds"""
    exe = str(tmp_path / "ec_lazy_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "ec_lazy_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr


def test_fused_dot_product_on_the_host(tmp_path):
    """fl252.h FlWide (the DEEP kernel's and the generated constraint kernels' accumulation with one Montgomery reduction
    per <= 16 products) equals the sum of single products, at the limb bounds it is specified for (tests/cpp/fl_wide_test.cpp)"""
    exe = str(tmp_path / "fl_wide_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "fl_wide_test.cpp")])
    assert "FL_WIDE_OK" in subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout


def test_host_arithmetic_in_64_bit_limbs_is_the_32_bit_arithmetic(tmp_path):
    """csrc/fp252_host.h (what the coin's Pedersen chain and DEEP's polynomials are computed in on the host) against csrc/fp252.h:
    products, sums, differences and powers of 200 000 pairs with 0, 1, p - 1 and the values around p's middle limbs among them"""
    exe = str(tmp_path / "fp252_host_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "fp252_host_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "bad = 0" in out.stdout, out.stdout + out.stderr
