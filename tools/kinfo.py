"""Register / scratch / LDS summary of the kernels in a -save-temps gfx950 assembly file.
usage: python tools/kinfo.py <file.s> [substring ...]   (substrings select kernels by demangled-ish name)"""
import re
import sys


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    name, info = None, {}
    for line in open(path, errors="replace"):
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            name, info = m.group(1), {}
            continue
        if name is None:
            continue
        m = re.match(r"\s*\.amdhsa_(group_segment_fixed_size|private_segment_fixed_size|next_free_vgpr|accum_offset)\s+(\S+)", line)
        if m:
            info[m.group(1)] = m.group(2)
        if ".end_amdhsa_kernel" in line:
            if not pats or any(p in name for p in pats):
                print(f"{name[:90]:90s} lds {info.get('group_segment_fixed_size')} scratch {info.get('private_segment_fixed_size')} "
                      f"vgpr {info.get('next_free_vgpr')} accum_offset {info.get('accum_offset')}")
            name = None


if __name__ == "__main__":
    main()
