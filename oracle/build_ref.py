#!/usr/bin/env python3
"""Builds the REFERENCE's own kernels for gfx950 into oracle/_ref/ (test infrastructure).

What it builds, from the sources where they lie under /root/reference:
  ref_defCorrSample.so  <- offersample_LGS/{droid.cpp, defCorrSample_kernel.cu,
                           corrSample_kernel.cu, gaussianAttn.cu, lowMem_defSample.cu}
                           (the whole `defCorrSample` extension, its own pybind11 binding)
  ref_altcorr.so        <- src/altcorr_kernel.cu + oracle/ref_altcorr_bind.cpp (own binding
                           of the two altcorr launchers; the reference binds them in
                           src/droid.cpp next to BA code that needs Eigen/lietorch — absent)

How: `torch.utils.cpp_extension.load`, i.e. the standard PyTorch-ROCm extension build that
the reference's own `CUDAExtension` setup.py would run on a ROCm machine: torch's bundled
hipify rewrites the CUDA headers/intrinsics to HIP, hipcc compiles for gfx950.  No stand-in
headers or stubs are written.  hipify needs writable copies, so the sources are copied to a
temporary directory OUTSIDE the repository and removed afterwards; only the two .so files
stay, under oracle/_ref/ (git-ignored, but shipped to the GPU box with the snapshot).
They are only ever loaded by oracle/gen_golden.py and tools/compare_ref.py on the GPU box.
"""
import os
import shutil
import sys
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

TARGETS = {
    "ref_defCorrSample": [os.path.join(REF, "offersample_LGS", n) for n in
                          ("droid.cpp", "defCorrSample_kernel.cu", "corrSample_kernel.cu", "gaussianAttn.cu",
                           "lowMem_defSample.cu")],
    "ref_altcorr": [os.path.join(REF, "src", "altcorr_kernel.cu"), os.path.join(HERE, "ref_altcorr_bind.cpp")],
}


def main():
    if not os.path.isdir(REF):
        print("build_ref: %s not present — nothing to do" % REF)
        return 0
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT, exist_ok=True)
    for name, srcs in TARGETS.items():
        so = os.path.join(OUT, name + ".so")
        if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
            print("build_ref: %s up to date" % so)
            continue
        tmp = tempfile.mkdtemp(prefix="lgu_ref_src_")
        bld = tempfile.mkdtemp(prefix="lgu_ref_bld_")
        try:
            local = []
            for s in srcs:
                shutil.copy(s, tmp)
                local.append(os.path.join(tmp, os.path.basename(s)))
            ce.load(name=name, sources=local, build_directory=bld, extra_cflags=["-O2"], extra_cuda_cflags=["-O2"],
                    verbose=False, is_python_module=False)
            shutil.copy(os.path.join(bld, name + ".so"), so)
            print("build_ref: built %s" % so)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
            shutil.rmtree(bld, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
