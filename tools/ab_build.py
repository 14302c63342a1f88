"""Build variants of liblidbox_hip.so that differ in the -D flags of ONE source, for same-process A/B timing
(tools/ab_gemm.py).  usage: python tools/ab_build.py <name> <source.hip> [-DFLAG=..]...  -> tools/ab/lib<name>.so"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidbox_amd import build as b

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build(verbose=False)
out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, name + "_" + src[:-4] + ".o")
subprocess.run([b.HIPCC] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj], check=True)
objs = [os.path.join(b.OBJDIR, f[:-4] + ".o") for f in b._sources() if f != src] + [obj]
lib = os.path.join(out_dir, "lib%s.so" % name)
subprocess.run([b.HIPCC, "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
