"""Per-kernel resource table of a built libfsrl_hip.so: VGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, LDS bytes.

Reads the gfx950 code object inside the library's .hip_fatbin section (llvm-objcopy -> clang-offload-bundler --unbundle) and the
AMDGPU metadata note of that object (llvm-readelf --notes).  Pure host tooling: no GPU, nothing executes.

usage: python tools/sonotes.py [path/to/lib.so] [substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "fsrl_amd", "libfsrl_hip.so")
FIELDS = ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
          "max_flat_workgroup_size")


def demangle(names):
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True)
        return out.stdout.split("\n")[:len(names)]
    except Exception:
        return list(names)


def kernel_notes(lib=DEFAULT_LIB):
    """-> {demangled kernel name: {field: int}} for every kernel of the library's gfx950 code object"""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    # the note is YAML: kernel entries start at "  - .", their own keys sit at four spaces
    kernels, cur = [], None
    for line in txt.splitlines():
        if re.match(r"  - \.\w+:", line):
            cur = {}
            kernels.append(cur)
            line = "    " + line[4:]
        if cur is None:
            continue
        m = re.match(r"    \.(\w+):\s*(\S+)\s*$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "name":
            cur["name"] = val
        elif key in FIELDS:
            cur[key] = int(val)
    kernels = [k for k in kernels if "name" in k]
    # the metadata lists .name for arguments too: keep the symbol-derived kernel name
    names = demangle([k["name"] for k in kernels])
    return {n: {f: k.get(f, 0) for f in FIELDS} for n, k in zip(names, kernels)}


def main():
    args = sys.argv[1:]
    lib = DEFAULT_LIB
    if args and args[0].endswith(".so"):
        lib, args = args[0], args[1:]
    notes = kernel_notes(lib)
    print(f"{'kernel':100s} vgpr  vspill sspill scratch  lds   threads")
    for name in sorted(notes):
        if args and not any(a in name for a in args):
            continue
        k = notes[name]
        print(f"{name[:100]:100s} {k['vgpr_count']:4d} {k['vgpr_spill_count']:6d} {k['sgpr_spill_count']:6d} "
              f"{k['private_segment_fixed_size']:7d} {k['group_segment_fixed_size']:6d} {k['max_flat_workgroup_size']:5d}")


if __name__ == "__main__":
    main()
