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
SOURCES = ["fastmax_api.hip", "fastmax_generic.hip", "fastmax_normalize.hip", "fastmax_rope.hip", "fastmax_ce.hip", "fastmax_mfma.hip", "fastmax_mfma_v2.hip", "fastmax_mfma_gen.hip", "fastmax_mfma_bf16.hip", "fastmax_mfma_d128_2p.hip", "fastmax_scan_d128_2p.hip", "fastmax_mfma_split.hip", "fastmax_quad_mfma.hip",
    "fastmax_quad32_mfma.hip", "fastmax_quad_mfma_bwd.hip", "fastmax_quad32_bwd.hip", "fastmax_mfma_bwd_lin.hip", "fastmax_decode.hip", "nf4_lora.hip", "nf4_gemm.hip", "lora_thin.hip"]
HEADERS = ["fastmax_common.h", "fastmax_mfma_common.h", "fastmax_mfma32_common.h", os.path.join("..", "..", "include", "fastmax_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc"]
# per-source flags.  The p=2 tile kernels interleave their per-score vector work with MFMAs: packed-f32 instructions
# (v_pk_fma_f32 / v_pk_add_f32, which the SLP vectoriser forms from adjacent scalar operations) cost 3-4x their issue slot
# beside matrix instructions on gfx950, so those files are built without it.
# The backward kernels that take the whole register file (one wave per SIMD) would otherwise get their MFMAs selected in the
# accumulator-file form, which moves every score tile between the two halves of the file (v_accvgpr_read / _write) on its way
# to and from the vector ALU: -amdgpu-mfma-vgpr-form keeps matrix results in the VGPRs the vector instructions read.
FILE_FLAGS = {"fastmax_quad32_mfma.hip": ["-fno-slp-vectorize"],
              "fastmax_quad32_bwd.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]}


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


def _compile_one(args):
    cc, src, obj, extra, verbose = args
    cmd = [cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"] + FILE_FLAGS.get(src, []) + extra + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return obj


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into fastmax_experiments_amd/libfastmax_hip.so.
    One object per source (compiled in parallel, only the stale ones), then one link."""
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    cc = _hipcc()
    extra = os.environ.get("FASTMAX_HIPCC_EXTRA", "").split()      # e.g. -DFASTMAX_QUAD32_ABLATION for ablation builds
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    tag = os.path.join(objdir, ".flags")
    flags_now = " ".join(extra)
    flags_changed = not os.path.exists(tag) or open(tag).read() != flags_now
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        src_time = max(os.path.getmtime(os.path.join(CSRC, src)), hdr_time)
        if force or flags_changed or not os.path.exists(obj) or os.path.getmtime(obj) < src_time:
            jobs.append((cc, src, obj, extra, verbose))
    workers = max(1, min(8, (os.cpu_count() or 2), len(jobs) or 1))
    with ThreadPoolExecutor(workers) as pool:
        list(pool.map(_compile_one, jobs))
    open(tag, "w").write(flags_now)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
