#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel of a HIP source, from hipcc's own remarks (no GPU needed):
    python tools/kernel_resources.py inpaintnet_amd/csrc/decode_b1.hip [more.hip ...] [--ref <git rev>]
--ref REV compiles the same files as of that revision next to the working tree and prints only the kernels whose numbers moved."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "inpaintnet_amd", "csrc")


def resources(src, incdir):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", "-I", incdir, "-I", os.path.join(REPO, "include")]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=incdir).stderr
    out, name = {}, None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
            out[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            out[name][m.group(1).split(" [")[0]] = int(m.group(2))
    return out


def fmt(r):
    return " ".join(f"{k}={v}" for k, v in r.items())


def main():
    args = sys.argv[1:]
    ref = None
    if "--ref" in args:
        i = args.index("--ref")
        ref = args[i + 1]
        del args[i:i + 2]
    for src in args:
        src = os.path.abspath(src)
        new = resources(src, CSRC)
        if ref is None:
            for k, v in new.items():
                print(f"{os.path.basename(src)}: {k}: {fmt(v)}")
            continue
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.check_call(f"git -C {REPO} archive {ref} inpaintnet_amd/csrc | tar -x -C {tmp}", shell=True)
            old = resources(os.path.join(tmp, "inpaintnet_amd", "csrc", os.path.basename(src)), os.path.join(tmp, "inpaintnet_amd", "csrc"))
        for k in sorted(set(new) | set(old)):
            if new.get(k) != old.get(k):
                print(f"{os.path.basename(src)}: {k}\n    {ref}: {fmt(old.get(k, {}))}\n    tree: {fmt(new.get(k, {}))}")
        print(f"{os.path.basename(src)}: {len(new)} kernels, the rest unchanged")


if __name__ == "__main__":
    main()
