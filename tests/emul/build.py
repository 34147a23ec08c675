"""Builds the test-only host emulations (never part of the product library)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def _build(src, out, cc, extra):
    srcp, outp = os.path.join(HERE, src), os.path.join(HERE, out)
    deps = [srcp, os.path.join(HERE, "..", "..", "vieo_slam_amd", "csrc", "quadtree.inl"),
            os.path.join(HERE, "..", "..", "vieo_slam_amd", "csrc", "sincosf_exact.h")]
    if not os.path.exists(outp) or any(os.path.getmtime(d) > os.path.getmtime(outp) for d in deps):
        subprocess.check_call([cc, "-O2", "-fPIC", "-shared", "-ffp-contract=off"] + extra +
                              ["-o", outp, srcp, "-lm"])
    return outp


def quadtree():
    return _build("quadtree_emul.cc", "libquadtree_emul.so", "g++", ["-std=c++17"])


def sincos():
    return _build("sincos_emul.c", "libsincos_emul.so", "gcc", [])
