"""Build csrc/libnsc_hip.so (the C-ABI library of include/nsc.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU; the .so is built IN-TREE so it travels to the GPU box.
"""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libnsc_hip.so")
ARCH = "gfx950"

# -ffp-contract=off: the bit-exact stages round every float op exactly as written (FMAs only where
# spelled out).  Correctly rounded float32 divide/sqrt is hipcc's default and is stated explicitly.
HIPCC_FLAGS = [
    "-O3", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared", "-std=c++17",
    "-Wall", "-Wno-unused-function",
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    root = os.path.dirname(_HERE)
    return (sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc"))
            + glob.glob(os.path.join(root, "include", "*.h")))


FLAGS_FILE = LIB + ".flags"        # the flags the library in place was built with (a development build must not
                                   # pass for the product build just because it is newer than the sources)


def _flags():
    flags = list(HIPCC_FLAGS)
    if os.environ.get("NSC_DEV_BUILD") == "1":       # enables the NSC_TUNE_* development knobs
        flags.append("-DNSC_DEV_TUNING")
    for d in os.environ.get("NSC_DEV_DEFINES", "").split():      # development A/B builds
        flags.append("-D" + d)
    return flags


def is_stale():
    if not os.path.exists(LIB):
        return True
    try:
        if open(FLAGS_FILE).read() != " ".join(_flags()):
            return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _deps())


def build_hip(force=False, verbose=False):
    """Compile every .hip under csrc/ into libnsc_hip.so.  Returns the library path."""
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = _flags()
    cmd = [hipcc] + flags + ["-o", LIB] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    with open(FLAGS_FILE, "w") as f:
        f.write(" ".join(flags))
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
