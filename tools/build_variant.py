#!/usr/bin/env python3
"""Builds an EXPERIMENT library build/ab/liblgu_<name>.so next to (never over) the in-tree default library: the named
sources are recompiled with the extra flags (or taken from another file / git revision), every other object is the default
build's (lgu-slam_amd/build/*.o, which must be current: run lgu_slam_amd.build() first).  The flags end up in
lgu_version().  Load a variant with LGU_LIB_PATH (tools/ab_lib_variants.sh, tools/ab_coop_variants.sh).

    python tools/build_variant.py <name> [--src lowmem_coop.hip=<file or git-rev>]... [-- <hipcc flags>...]
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgu_slam_amd  # noqa: E402

B = lgu_slam_amd._build


def main(argv):
    name = argv[0]
    flags, srcs = [], {}
    rest = argv[1:]
    if "--" in rest:
        k = rest.index("--")
        rest, flags = rest[:k], rest[k + 1:]
    while rest:
        assert rest[0] == "--src", rest
        f, where = rest[1].split("=", 1)
        srcs[f] = where
        rest = rest[2:]
    if not srcs:
        srcs = {"lowmem_coop.hip": None}
    out = os.path.join(ROOT, "build", "ab", "liblgu_%s.so" % name)
    objdir = os.path.join(ROOT, "build", "ab", name + "_obj")
    os.makedirs(objdir, exist_ok=True)
    label = " ".join(flags + ["%s@%s" % (f, w) for f, w in srcs.items() if w])
    recompiled = dict(srcs)
    recompiled.setdefault("capi.hip", None)
    objs = []
    tmpdir = tempfile.mkdtemp(prefix="lgu_variant_")
    try:
        for src in B.SOURCES:
            if src not in recompiled:
                o = os.path.join(os.path.dirname(B.SO_PATH), "build", src.replace(".hip", ".o"))
                assert os.path.exists(o), "default object missing: build the default library first"
                objs.append(o)
                continue
            path = os.path.join(B.CSRC, src)
            where = recompiled[src]
            if where:
                text = open(where).read() if os.path.exists(where) else subprocess.check_output(
                    ["git", "-C", ROOT, "show", "%s:lgu-slam_amd/csrc/%s" % (where, src)]).decode()
                path = os.path.join(tmpdir, src)
                open(path, "w").write(text)
            obj = os.path.join(objdir, src.replace(".hip", ".o"))
            cmd = [B._hipcc()] + B.FLAGS + (flags if src != "capi.hip" else []) + \
                (["-DLGU_BUILD_FLAGS=\"%s\"" % (label or name)] if src == "capi.hip" else []) + ["-I", B.CSRC, "-c", path, "-o", obj]
            subprocess.check_call(cmd)
            objs.append(obj)
    finally:
        for f in os.listdir(tmpdir):
            os.remove(os.path.join(tmpdir, f))
        os.rmdir(tmpdir)
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out] + objs)
    print(out)


if __name__ == "__main__":
    main(sys.argv[1:])
