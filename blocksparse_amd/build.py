"""Build the gfx950 HIP library in-tree: blocksparse_amd/libbsmm_hip.so (hipcc cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
OUT = os.path.join(_HERE, "libbsmm_hip.so")
SOURCES = ["bsmm_api.hip", "bst_api.hip", "bsmm_dist.hip"]
def _headers():
    """every header under csrc/ (a hand-kept list once missed the bench-path kernel: edit it and build() kept the old .so)"""
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + _headers()] + [os.path.join(INCLUDE, "bsmm.h"), os.path.join(INCLUDE, "bst.h"), os.path.join(INCLUDE, "bsmm_dist.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(out, extra_flags):
    """An experiment build of the whole library with extra compiler flags (-DU2_... switches) into `out` (see scripts/)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + INCLUDE, "-I" + CSRC] + list(extra_flags)
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + "\n".join([ln for ln in (r.stdout + r.stderr).splitlines() if "error" in ln][:10]))
    return out


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + INCLUDE, "-I" + CSRC]
    cmd += os.environ.get("BSMM_EXTRA_CXXFLAGS", "").split()
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl", "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    tmp = OUT + ".tmp"
    cmd[-1] = tmp
    if os.path.exists(OUT):
        os.remove(OUT)          # never leave a stale library behind a failed build
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        errs = [ln for ln in (r.stdout + r.stderr).splitlines() if "error" in ln][:10]
        raise RuntimeError("hipcc failed:\n" + "\n".join(errs))
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
