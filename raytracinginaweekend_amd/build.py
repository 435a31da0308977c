"""Builds libmirt.so (the HIP kernels + C ABI) in-tree for gfx950.

    python -m raytracinginaweekend_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is required: parity with the reference's
non-FMA SSE2 arithmetic (and with the CPU oracle) is bit-exact only without contraction.
-fno-slp-vectorize: on gfx950 v_pk_add_f32 / v_pk_mul_f32 issue at ~1/3 the rate of the plain forms
(tools/ubench/valu_rate.hip: 20 vs 60 T wave-lane-instr/s), so SLP-packing scalar fp32 math loses.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmirt.so")
VARIANT_LIB = os.path.join(LIBDIR, "libmirt_defer.so")
SOURCES = ["mirt.hip", "mirt_multi.hip"]
DEPS = ["mirt.hip", "mirt_multi.hip", "rt_kernels.h", "rt_device.h", "scene_host.h", os.path.join("..", "..", "include", "mirt.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical", "-ldl", "-pthread"]


def skip_variant() -> bool:
    """MIRT_SKIP_VARIANT=1: build the product library alone (the measured-and-lost second-stream variant is test material)."""
    return os.environ.get("MIRT_SKIP_VARIANT", "") not in ("", "0")


def stale() -> bool:
    if not os.path.exists(LIB) or (not skip_variant() and not os.path.exists(VARIANT_LIB)):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        build_host(False)
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    # the measured-and-lost variant with a second, material-sorted stream per wave (profiles/r02_material_stream.txt): built so
    # that its bit-identity with the product stays under test, never loaded by the product
    variant = [hipcc] + FLAGS + ["-DMIRT_DEFER_MD=1", "-DMIRT_LIST_WAVES=6"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", VARIANT_LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    procs = [subprocess.Popen(cmd)] + ([] if skip_variant() else [subprocess.Popen(variant)])
    if any([p.wait() != 0 for p in procs]):
        raise subprocess.CalledProcessError(1, cmd)
    build_host(verbose)
    return LIB


def build_host(verbose: bool = True) -> None:
    """The drop-in C++ host layer (g++, no GPU code); its structs embed mirt.h types, so it follows libmirt."""
    subprocess.run(["make", "-C", os.path.join(HERE, "host")] + ([] if verbose else ["-s"]), check=True)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
