"""Build variants of liblidbox_hip.so that differ in the -D flags (or the text) of ONE source, for same-process A/B
timing (tools/ab_gemm.py, tools/ab_feat.py).
usage: python tools/ab_build.py <name> <source.hip> [--from other_source_file] [-DFLAG=..]...  -> tools/ab/lib<name>.so
--from compiles another file in place of csrc/<source.hip> (e.g. an older revision: git show HEAD~1:... > tools/ab/x.hip)."""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidbox_amd import build as b

name, src, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
path = os.path.join(b.CSRC, src)
if "--from" in rest:
    i = rest.index("--from")
    path = os.path.abspath(rest[i + 1])
    rest = rest[:i] + rest[i + 2:]
flags = rest
b.build(verbose=False)
out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("LIDBOX_AB_DIR", "ab"))
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, name + "_" + src[:-4] + ".o")
subprocess.run([b.HIPCC] + b.FLAGS + flags + ["-x", "hip", "-c", path, "-o", obj], check=True)
objs = [os.path.join(b.OBJDIR, f[:-4] + ".o") for f in b._sources() if f != src] + [obj]
lib = os.path.join(out_dir, "lib%s.so" % name)
subprocess.run([b.HIPCC, "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
