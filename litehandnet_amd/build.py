"""Build liblhn.so (HIP, gfx950 only) in-tree with hipcc.  `python -m litehandnet_amd.build`."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblhn.so")
NO_CONTRACT = {"k_heatmap.hip", "k_deploy.hip"}   # bit-exact decode/encode/BN-fold arithmetic: no fused multiply-add
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "lhn.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            extra = ["-ffp-contract=off"] if os.path.basename(src) in NO_CONTRACT else []
            cmd = [hipcc] + FLAGS + extra + ["-c", "-x", "hip", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
