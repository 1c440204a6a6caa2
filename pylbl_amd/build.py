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


VARIANTS = {
    # name: (extra flags, what it is for) -- built beside the shipped library, loaded only when
    # $PYLBL_AMD_LIBRARY names them (engine.library()).
    "ablate": (["-DLBL_ABLATE"], "engine option 'ablate' (scripts/ablate_*.sh): parts of the "
                                 "accumulate kernel switched off for timing, results wrong"),
    "asan": (["-O1", "-g", "-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan"],
             "host code under AddressSanitizer (scripts/checks/host_asan.sh)"),
    "tsan": (["-O1", "-g", "-Xarch_host", "-fsanitize=thread", "-shared-libsan"],
             "host code under ThreadSanitizer (scripts/checks/host_tsan.sh)"),
    "ubsan": (["-O1", "-g", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host",
               "-fno-sanitize=vptr,function", "-shared-libsan"],
              "host code under UndefinedBehaviorSanitizer (scripts/checks/host_ubsan.sh)"),
}


def build_variant(name, verbose=False):
    """pylbl_amd/liblbl_amd_<name>.so: the same translation unit with the variant's flags."""
    extra, _ = VARIANTS[name]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    target = PACKAGE / f"liblbl_amd_{name}.so"
    command = [hipcc] + FLAGS + extra + [str(SOURCE), "-o", str(target), "-ldl"]
    if verbose:
        print(" ".join(command))
    subprocess.run(command, check=True)
    return target


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:
        for variant in sys.argv[1:]:
            build_variant(variant, verbose=True)
    else:
        build(force=True, verbose=True)
