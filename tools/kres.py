#!/usr/bin/env python3
"""Register / spill / occupancy table of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage,
device-only compile: seconds, no GPU).  usage: python tools/kres.py srwn_group.hip [name-filter] [-DFLAG ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    if not os.path.exists(src):
        src = os.path.join(ROOT, "sr-wavenet_amd", "csrc", src)
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    filt = [a for a in sys.argv[2:] if not a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-pass-failed",
           "-ffp-contract=on", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", src, "-o", "/dev/null"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-4000:])
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    # (c++filt does not know DF16b = __bf16: demangle with it spelled as a known type)
    dem = subprocess.run(["c++filt"], input="\n".join(r["name"].replace("DF16b", "Dh") for r in rows), capture_output=True,
                         text=True).stdout.replace("_Float16", "bf16").splitlines()
    print("%-110s %5s %5s %6s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "occ", "LDS"))
    for r, d in zip(rows, dem):
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        d = re.sub(r"\(.*\)$", "", d).replace("void ", "")
        if filt and not all(f in d for f in filt):
            continue
        print("%-110s %5d %5d %6d %6d %4d %7d" % (d[:110], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1),
                                                  r.get("SGPRs Spill", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))


if __name__ == "__main__":
    main()
