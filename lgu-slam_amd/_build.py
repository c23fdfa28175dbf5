"""Builds liblgu_corr.so — the C-ABI library of hand-written gfx950 HIP kernels.

`hipcc` cross-compiles for gfx950 without a GPU, so this runs in the build container;
the resulting in-tree .so is what travels to the GPU box.  No torch headers are
involved: the library's boundary is plain C (include/lgu_corr.h).
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(_HERE, "liblgu_corr.so")
SOURCES = ["capi.hip", "defcorr.hip", "defcorr_lean.hip", "defcorr_bwd.hip", "gaussmask.hip", "lowmem.hip", "lowmem_tile.hip", "lowmem_mfma.hip", "lowmem_coop.hip", "ba.hip", "ba_chol.hip", "offsets.hip", "offconv.hip"]
# -ffp-contract=off: keep the reference's fp32 evaluation order (no FMA contraction) so
# results are bit-comparable with the CPU oracle; these kernels are memory-bound.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wno-unused-value", "-fno-gpu-rdc"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 kernels cannot be built")
    return exe


def needs_build():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(_HERE, "..", "include", "lgu_corr.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip source for gfx950 and link liblgu_corr.so in-tree."""
    if not force and not needs_build():
        return SO_PATH
    hipcc = _hipcc()
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        extra = os.environ.get("LGU_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DCO_PF=3)
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        objs.append(obj)
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", SO_PATH] + objs
    subprocess.check_call(cmd)
    return SO_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
