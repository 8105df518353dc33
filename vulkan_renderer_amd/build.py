"""Builds the gfx950 shared library in-tree with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libvkrhip.so")
SOURCES = ["vkrhip.hip", "vkrhip_host.cpp", "vkrhip_shadow.cpp", "vkrhip_gltf.cpp"]
EXAMPLE_SRC = os.path.join(os.path.dirname(PKG_DIR), "examples", "scene_frame.c")
EXAMPLE_BIN = os.path.join(PKG_DIR, "scene_frame")  # C host of the library (examples/scene_frame.c)

# -ffp-contract=off: the reference fuses multiply-add only inside mat4_mul
# (lib/src/math/mat.h:1145-1163); hipcc's default would fuse everywhere and
# change result bits. Square roots must be correctly rounded (sqrtf on the host).
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-ffp-contract=off",
    # no automatic pairing of scalar float operations into packed instructions: the
    # register moves that form the pairs cost more than the halved operation count
    # saves (the fused kernel is 4 % faster without); the packed plane tests are
    # written out by hand
    "-fno-slp-vectorize",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared", "-std=c++17",
    "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build libvkrhip.so")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp"))]
    deps.append(os.path.join(os.path.dirname(PKG_DIR), "include", "vkr_hip_scene.h"))
    return any(os.path.getmtime(d) > built for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    extra = os.environ.get("VKRHIP_EXTRA_FLAGS", "").split()  # experiments only
    cmd = [hipcc()] + HIPCC_FLAGS + extra + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    build_example(verbose)
    return LIB_PATH


def build_example(verbose: bool = False) -> str:
    """The plain C host (C11, gcc): proves the header is C and the library links
    without anything else of this repository."""
    include = os.path.join(os.path.dirname(PKG_DIR), "include")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-I", include, "-o", EXAMPLE_BIN, EXAMPLE_SRC,
           "-L", PKG_DIR, "-lvkrhip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return EXAMPLE_BIN


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
