"""Build libcapdec_hip.so (gfx950) in-tree with hipcc.  `python -m capdec_amd.build [--force]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcapdec_hip.so")
SOURCES = ["capi_context.hip", "weights.hip", "gemm_dispatch.hip", "decode.hip", "train_step.hip", "train_mapper.hip", "train_ops.hip", "train_optim.hip", "clip.hip", "comm.hip", "gemm_f32.hip", "gemm_bf16x3.hip", "gemm_f16x2.hip", "gemm_h2w.hip", "gemm_pp.hip", "elementwise.hip", "attention.hip", "resnet.hip", "select.hip", "preprocess.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "config.h"), os.path.join(CSRC, "context.h"), os.path.join(CSRC, "train.h"), os.path.join(CSRC, "gemm_epilogue.h"), os.path.join(CSRC, "gemm_epilogue_w.h"), os.path.join(CSRC, "gemm_epilogue_lds.h"), os.path.join(CSRC, "bf16x3.h"), os.path.join(os.path.dirname(HERE), "include", "capdec.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# measurement variant (-DCAPDEC_MEASURE: ablations that compute wrong results on purpose, ring-depth / occupancy overrides,
# per-block phase stamps, the diverged-beam hook): tools/ and bench.py's untimed tail load it explicitly; the product path
# (capdec_amd._capi.load_library()) never does
MEASURE_LIB_PATH = os.path.join(LIB_DIR, "libcapdec_hip_measure.so")


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built on this machine")
    return exe


def source_hash() -> str:
    """sha256 over every translation unit and header the library is built from: compiled into the library
    (capdec_build_id) and re-computed by capdec_amd._capi.load_library, so a stale .so is refused instead of silently
    running old kernels"""
    import hashlib
    h = hashlib.sha256()
    for path in sorted([os.path.join(CSRC, s) for s in SOURCES] + HEADERS):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, measure: bool = False) -> str:
    """Compile every .hip translation unit to an object, link the shared library.
    Incremental: only stale objects are rebuilt.  measure=True builds libcapdec_hip_measure.so (-DCAPDEC_MEASURE)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj_measure" if measure else "obj")
    lib_path = MEASURE_LIB_PATH if measure else LIB_PATH
    flags = FLAGS + (["-DCAPDEC_MEASURE"] if measure else [])
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    objs, procs = [], []
    build_id = source_hash()
    id_file = os.path.join(obj_dir, "build_id.txt")
    id_changed = not os.path.exists(id_file) or open(id_file).read().strip() != build_id
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + HEADERS) or (src == "capi_context.hip" and id_changed):
            cmd = [hipcc, *flags, "-c", sp, "-o", op]
            if src == "capi_context.hip":
                cmd.insert(-4, f'-DCAPDEC_BUILD_ID="{build_id}"')
            if verbose:
                print("[capdec build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(lib_path, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", lib_path]
        if verbose:
            print("[capdec build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(id_file, "w") as f:
        f.write(build_id + "\n")
    return lib_path


if __name__ == "__main__":
    # both variants by default (a measurement library older than the sources is refused like a stale product library);
    # --product / --measure build one of them
    if "--measure" not in sys.argv:
        print(build(force="--force" in sys.argv))
    if "--product" not in sys.argv:
        print(build(force="--force" in sys.argv, measure=True))
