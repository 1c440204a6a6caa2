"""Builds the HIP engine in-tree: pylbl_amd/liblbl_amd.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
shared library travels with the tree to the GPU box.
"""
from pathlib import Path
import os
import shutil
import subprocess

PACKAGE = Path(__file__).resolve().parent
SOURCE = PACKAGE / "csrc" / "engine.hip"
LIBRARY = PACKAGE / "liblbl_amd.so"
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared"]


def sources():
    return sorted((PACKAGE / "csrc").glob("*")) + [PACKAGE.parent / "include" / "lbl_amd.h"]


def is_stale():
    if not LIBRARY.exists():
        return True
    built = LIBRARY.stat().st_mtime
    return any(x.stat().st_mtime > built for x in sources())


def build(force=False, verbose=False):
    """Compiles csrc/engine.hip when the library is missing or older than its sources."""
    if not force and not is_stale():
        return LIBRARY
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the gfx950 engine.")
    command = [hipcc] + FLAGS + [str(SOURCE), "-o", str(LIBRARY), "-ldl"]
    if verbose:
        print(" ".join(command))
    subprocess.run(command, check=True)
    return LIBRARY


if __name__ == "__main__":
    build(force=True, verbose=True)
