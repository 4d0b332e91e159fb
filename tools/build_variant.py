"""A second library next to the product one, built with extra compiler flags (an A/B of a tunable that is a macro: -DTSC_SORTED_OCC=4, ...).
    python tools/build_variant.py NAME [-DMACRO=VALUE ...]   ->  tscode_amd/ab_libs/NAME.so   (git-ignored; travels to the GPU box)
Run with  TSCODE_AMD_LIB=$PWD/tscode_amd/ab_libs/NAME.so python bench.py ...   (bench.py then reports a binary digest that is not the sources':
traffic is withheld, as for any binary it cannot tie to a profile)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tscode_amd import build as b

name, extra = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(b.HERE, "ab_libs")
obj_dir = os.path.join(out_dir, "_obj_" + name)
os.makedirs(obj_dir, exist_ok=True)
procs = [subprocess.Popen([b._hipcc()] + b.CFLAGS + extra + ['-DTSC_CSRC_DIGEST="variant:%s"' % name, "-c", "-o", os.path.join(obj_dir, s + ".o"), os.path.join(b.CSRC, s)])
         for s in b.SOURCES]
if any(p.wait() != 0 for p in procs):
    sys.exit("compile failed")
out = os.path.join(out_dir, name + ".so")
subprocess.run([b._hipcc()] + b.LDFLAGS + ["-o", out] + [os.path.join(obj_dir, s + ".o") for s in b.SOURCES], check=True)
print(out)
