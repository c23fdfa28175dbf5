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
SOURCES = ["capi.hip", "defcorr.hip", "defcorr_lean.hip", "defcorr_bwd.hip", "gaussmask.hip", "volbuild.hip", "lowmem.hip", "lowmem_tile.hip", "lowmem_mfma.hip", "lowmem_coop.hip", "ba.hip", "ba_chol.hip", "offsets.hip", "offconv.hip"]
# -ffp-contract=off: keep the reference's fp32 evaluation order (no FMA contraction) so
# results are bit-comparable with the CPU oracle; these kernels are memory-bound.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wno-unused-value", "-fno-gpu-rdc"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 kernels cannot be built")
    return exe


STAMP_PATH = SO_PATH + ".flags"   # the flags the in-tree library was built with (travels with it)


def _extra_flags():
    return os.environ.get("LGU_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DCO_PF=3)


def _flags_key(extra):
    return " ".join(FLAGS + list(extra))


def needs_build():
    if not os.path.exists(SO_PATH):
        return True
    try:
        with open(STAMP_PATH) as fh:
            if fh.read() != _flags_key(_extra_flags()):
                return True      # the library in the tree was built with other flags
    except OSError:
        return True
    t = os.path.getmtime(SO_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(_HERE, "..", "include", "lgu_corr.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, out=None, extra=None):
    """Compile every .hip source for gfx950 and link liblgu_corr.so in-tree.

    out / extra (experiments only): link the library somewhere else (e.g. build/ab/liblgu_x.so, loaded through
    LGU_LIB_PATH) with extra compiler flags; the flags end up in lgu_version().  The default library is only ever the
    default build: LGU_EXTRA_HIPCC_FLAGS without `out` is refused."""
    extra = _extra_flags() if extra is None else list(extra)
    if out is None:
        if extra:
            raise RuntimeError("LGU_EXTRA_HIPCC_FLAGS builds an experiment: pass out=<path> (and load it with LGU_LIB_PATH); "
                               "the in-tree library stays the default build")
        if not force and not needs_build():
            return SO_PATH
    so = SO_PATH if out is None else os.path.abspath(out)
    hipcc = _hipcc()
    objdir = os.path.join(_HERE, "build") if out is None else os.path.splitext(so)[0] + "_obj"
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + extra + (["-DLGU_BUILD_FLAGS=\"%s\"" % " ".join(extra)] if extra and src == "capi.hip" else []) + \
            ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, pr in procs:
        log, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, log.decode(errors="replace")))
        objs.append(obj)
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so] + objs
    subprocess.check_call(cmd)
    if out is None:
        with open(STAMP_PATH, "w") as fh:
            fh.write(_flags_key(extra))
    return so


if __name__ == "__main__":
    import sys
    # python _build.py [out.so [extra flags...]]
    if len(sys.argv) > 1:
        print(build(force=True, verbose=True, out=sys.argv[1], extra=sys.argv[2:]))
    else:
        print(build(force=True, verbose=True))
