"""Build libfastmax_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The library is the product: there is no CPU or PyTorch fallback behind it.  ``build()`` is
what ``__graft_entry__.build()`` calls; the built .so travels to the GPU box with the tree.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfastmax_hip.so")
SOURCES = ["fastmax_api.hip", "fastmax_generic.hip", "fastmax_normalize.hip", "fastmax_rope.hip", "fastmax_mfma.hip", "fastmax_mfma_gen.hip", "fastmax_mfma_bf16.hip", "fastmax_mfma_split.hip", "fastmax_quad_mfma.hip",
    "fastmax_quad32_mfma.hip", "fastmax_quad_mfma_bwd.hip", "fastmax_quad32_bwd.hip", "fastmax_mfma_bwd_lin.hip", "fastmax_decode.hip", "nf4_lora.hip"]
HEADERS = ["fastmax_common.h", "fastmax_mfma_common.h", "fastmax_mfma32_common.h", os.path.join("..", "..", "include", "fastmax_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libfastmax_hip.so cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into fastmax_experiments_amd/libfastmax_hip.so."""
    if not force and not stale():
        return LIB
    extra = os.environ.get("FASTMAX_HIPCC_EXTRA", "").split()      # e.g. -DFASTMAX_QUAD32_ABLATION for tools/ablate builds
    cmd = [_hipcc()] + FLAGS + extra + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
