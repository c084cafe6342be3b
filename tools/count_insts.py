"""Static instruction counts of the device code inside object files (llvm-objdump of the gfx950 code object of each .o): vector ALU
instructions, v_mad_u64_u32 among them, scratch (spill) operations, LDS and global memory operations.  The generated constraint
kernels are straight-line code: their static counts are what a point executes (profiles/r04_quotient_algebra.txt).
usage: python tools/count_insts.py sandstorm_amd/_build/quotient_gen_starknet_p?.o ..."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def count(obj):
    tmp = tempfile.mkdtemp()
    try:
        dst = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, dst)
        subprocess.run([OBJDUMP, "--offloading", dst], capture_output=True, text=True)
        out = {}
        for f in os.listdir(tmp):
            if f.startswith(os.path.basename(obj) + ".") and "gfx950" in f:
                txt = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True).stdout
                out = {"valu": len(re.findall(r"^\s+v_", txt, re.M)), "mads": len(re.findall(r"v_mad_u64_u32", txt)),
                       "scratch": len(re.findall(r"^\s+scratch_", txt, re.M)), "ds": len(re.findall(r"^\s+ds_", txt, re.M)),
                       "global": len(re.findall(r"^\s+global_", txt, re.M))}
        return out
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":
    total = 0
    for obj in sys.argv[1:]:
        c = count(obj)
        if c:
            total += c["valu"]
            print("%-44s valu %6d mads %6d scratch %4d ds %5d global %4d" % (os.path.basename(obj), c["valu"], c["mads"], c["scratch"], c["ds"], c["global"]))
    if len(sys.argv) > 2:
        print("%-44s valu %6d" % ("total", total))
