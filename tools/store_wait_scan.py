#!/usr/bin/env python
"""Static scan of the compiled kernels for waits that make a wave sit out the acknowledgement of its own stores.

gfx950 counts vector-memory loads AND stores in one counter (vmcnt) that retires in issue order, so `s_waitcnt vmcnt(N)`
for a load that was issued after a store also waits for that store to be acknowledged by L2 -- a full memory round trip
with nothing to show for it.  A loop of the form {load, wait, compute, store} pays that once per iteration (rounds 3-5: the
GEMM epilogue's store loop, the GroupNorm backward's residual add; round 6 removed them, -2.4 ms per train step).
For every kernel this walks the instruction stream in program order (branches ignored: a lower bound for loops), keeps the
queue of outstanding vector-memory operations, and reports each wait that retires at least one STORE together with a LOAD
issued after it -- i.e. a load-after-store the wave then blocks on.  CPU-only:
   python tools/store_wait_scan.py [file.hip ...]        # default: every source of the library; prints a table
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ml-mdm_amd", "csrc")
VM_LOAD = re.compile(r"^\s+(global_load|buffer_load|scratch_load|flat_load|global_atomic\w*\s+v)")
VM_STORE = re.compile(r"^\s+(global_store|buffer_store|scratch_store|flat_store|global_atomic)")
WAIT = re.compile(r"^\s+s_waitcnt\b.*vmcnt\((\d+)\)")


def scan(path):
    out = collections.OrderedDict()
    name, queue, hits = None, [], 0
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s+; @", line)
        if m:
            name, queue, hits = m.group(1), [], 0
            out[name] = [0, 0]          # [blocking waits, stores]
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        if VM_LOAD.match(line) and "lds" not in line.split(";")[0].split()[-1:]:
            queue.append("L")
        elif VM_STORE.match(line):
            queue.append("S")
            out[name][1] += 1
        else:
            w = WAIT.match(line)
            if w:
                n = int(w.group(1))
                retired = queue[:len(queue) - n] if n < len(queue) else []
                # a store retired together with a load issued after it
                if "S" in retired and "L" in retired[retired.index("S"):]:
                    out[name][0] += 1
                queue = queue[len(queue) - n:] if n < len(queue) else queue
    return out


def main():
    srcs = sys.argv[1:] or ["gemm_conv.hip", "norm.hip", "attention.hip", "elementwise.hip", "optim.hip", "diffusion_ops.hip"]
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for s in srcs:
            src = s if os.path.exists(s) else os.path.join(CSRC, s)
            o = os.path.join(td, os.path.basename(src) + ".s")
            procs.append((o, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                                               "-I" + CSRC, "-S", "--cuda-device-only", src, "-o", o], stderr=subprocess.DEVNULL)))
        for o, pr in procs:
            pr.wait()
            res = scan(o)
            bad = [(v[0], v[1], k) for k, v in res.items() if v[0]]
            print("%s: %d kernels, %d with a load-after-store wait" % (os.path.basename(o)[:-2], len(res), len(bad)))
            for n, st, k in sorted(bad, reverse=True)[:40]:
                dem = subprocess.run(["c++filt", k], stdout=subprocess.PIPE).stdout.decode().strip()
                print("   %3d blocking waits, %3d stores: %s" % (n, st, dem[:150]))


if __name__ == "__main__":
    main()
