"""
In-tree build of liblidbox_hip.so (hipcc, gfx950 only).

    python -m lidbox_amd.build [--force]

Every csrc/*.hip is compiled to an object (in parallel, one hipcc per file) and linked into
lidbox_amd/csrc/liblidbox_hip.so.  The .so is git-ignored but travels to the GPU box with the
repo snapshot; nothing is JIT-compiled at import time.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "liblidbox_hip.so")
OBJDIR = os.path.join(CSRC, "build")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC,
         "-Wall", "-Wno-unused-function"] + os.environ.get("LIDBOX_HIPCC_FLAGS", "").split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(path):
    """hash of a source plus every header it can see (cheap: all *.h in csrc/ and include/)."""
    h = hashlib.sha1()
    deps = [path] + [os.path.join(d, f) for d in (CSRC, INCLUDE) for f in sorted(os.listdir(d))
                     if f.endswith(".h")]
    for d in deps:
        with open(d, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJDIR, src[:-4] + ".o")
    stamp_file = obj + ".stamp"
    stamp = _stamp(os.path.join(CSRC, src))
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("liblidbox_hip.so: %s (%d sources, %s)" % (LIB, len(srcs), "rebuilt" if rebuilt else "up to date"))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
