#!/usr/bin/env python
"""Register / spill / occupancy / LDS table of every kernel of the gfx950 build, from the compiler's own report
(hipcc -O3 -Rpass-analysis=kernel-resource-usage; no GPU needed).
   python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ml-mdm_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FILT = "/usr/bin/c++filt"
KEYS = (("VGPRs", "VGPRs:"), ("AGPRs", "AGPRs:"), ("spill", "VGPRs Spill:"), ("scratch", "ScratchSize [bytes/lane]:"),
        ("occ", "Occupancy [waves/SIMD]:"), ("lds", "LDS Size [bytes/block]:"))


def demangle(names):
    # this c++filt predates the bf16 mangling (DF16b): demangle it as the other 16-bit builtin and rename afterwards
    out = subprocess.run([FILT], input="\n".join(n.replace("DF16b", "Dh") for n in names), capture_output=True, text=True).stdout.splitlines()
    res = []
    for n in out:
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*$", "", n)          # drop the parameter list
        n = n.replace("mdm::", "").replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16").replace("<half", "<bf16")
        res.append(n)
    return res


def main():
    print("Kernel resources of the gfx950 build (hipcc -O3 -Rpass-analysis=kernel-resource-usage, ROCm 7.2), %s." % (sys.argv[1] if len(sys.argv) > 1 else "HEAD"))
    print("VGPR + AGPR share the 512-entry file of a SIMD lane: waves / SIMD = floor(512 / (VGPRs + AGPRs)), capped by LDS and the block size.")
    print("%-100s %5s %5s %6s %7s %6s %9s" % ("kernel", "VGPR", "AGPR", "spill", "scratch", "occ", "LDS bytes"))
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        if p.returncode != 0:
            sys.exit(p.stderr[-2000:])
        kernels, cur = [], None
        for line in p.stderr.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                kernels.append(cur)
                continue
            if cur is None:
                continue
            for key, tag in KEYS:
                m = re.search(re.escape(tag) + r"\s*(\d+)", line)
                if m and "remark:     " + tag in line:
                    cur[key] = int(m.group(1))
        names = demangle([k["name"] for k in kernels])
        print("## %s (%d kernels)" % (os.path.basename(src), len(kernels)))
        for k, n in zip(kernels, names):
            print("%-100s %5d %5d %6d %7d %6d %9d" % (n[:100], k.get("VGPRs", -1), k.get("AGPRs", 0), k.get("spill", 0),
                                                     k.get("scratch", 0), k.get("occ", 0), k.get("lds", 0)))


if __name__ == "__main__":
    main()
