#!/usr/bin/env python
"""Static check of the compiled kernels for the hazard found in round 3: an MFMA result read back (v_accvgpr_read, VALU,
LDS / global store) with too few wait states when the read sits on the TAKEN edge of a branch -- ROCm 7.2's hazard
recognizer covered the fall-through path only (conv3x3_direct_kernel<64, 64, 8, 32, 4>, HISTORY.md section 4.1e).
Walks every path of up to 10 wait states after each v_mfma (following branches) and reports reads of its destination
registers that come earlier than `MIN_WS` (the compiler itself leaves 11 in straight-line code).  CPU-only:
   python tools/mfma_hazard_scan.py            # compiles csrc/gemm_conv.hip and attention.hip to ISA, exits 1 on a finding
Also checks, on the same ISA, that no kernel spills more than SPILL_CAP vector registers, and that the bias-gradient dot
products of conv_wgrad_bl_kernel still sit behind scalar branches (round 5: as plain code hipcc if-converted them into every
wave's instruction stream -- all 64 v_dot2c + a v_cndmask per row -- which is what made one block in nine the tail of the
launch; a parity test does not see that either), and -- round 6 -- that no GEMM epilogue loads between the wide stores of its
chunked store loop (loads_between_wide_stores)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIN_WS = 6
SPILL_CAP = 24   # spilled VGPRs tolerated per kernel (today's worst: 15, a tap-selecting 8-wave GEMM tile)


def regs(tok):
    out = set()
    for m in re.finditer(r"\b([av])\[(\d+):(\d+)\]", tok):
        out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    for m in re.finditer(r"\b([av])(\d+)\b", tok):
        out.add((m.group(1), int(m.group(2))))
    return out


def analyze(path):
    txt = open(path).read()
    findings = []
    for km in re.finditer(r"^(\S+):\s*; @\1\n", txt, re.M):
        name, i = km.group(1), km.end()
        lines = [ln.strip() for ln in txt[i:txt.find(".Lfunc_end", i)].split("\n")]
        labels = {m.group(1): k for k, ln in enumerate(lines) for m in [re.match(r"^(\.LBB\w+):", ln)] if m}
        code = [(k, ln) for k, ln in enumerate(lines) if ln and not ln.startswith(";") and not ln.startswith(".")]
        for n, (_, ln) in enumerate(code):
            if not ln.startswith("v_mfma"):
                continue
            dst = regs(ln.split(None, 1)[1].split(",")[0])
            stack = [(n + 1, 0)]
            while stack:
                pos, ws = stack.pop()
                while pos < len(code) and ws < 10:
                    ll = code[pos][1]
                    op = ll.split()[0]
                    if op == "s_nop":
                        ws += int(ll.split()[1]) + 1
                    elif op.startswith("v_mfma"):
                        if regs(ll.split(None, 1)[1].split(",")[0]) & dst:
                            break                      # accumulates into / overwrites it: interlocked by hardware
                        ws += 4
                    else:
                        parts = [p.strip() for p in (ll.split(None, 1)[1] if " " in ll else "").split(",")]
                        stores = op.startswith(("global_store", "ds_write", "buffer_store", "scratch_store"))
                        used = set()
                        for t in (parts if stores else parts[1:]):
                            used |= regs(t)
                        if used & dst:
                            findings.append((name, ws, ln, ll))
                            break
                        if op.startswith("s_cbranch") or op == "s_branch":
                            tgt = ll.split()[-1]
                            if tgt in labels:
                                nxt = [nn for nn, (k2, _) in enumerate(code) if k2 > labels[tgt]]
                                if nxt:
                                    stack.append((nxt[0], ws + 1))
                            if op == "s_branch":
                                break
                        ws += 1
                    pos += 1
    return findings


def unguarded_dot2(path):
    """v_dot2c instructions of conv_wgrad_bl_kernel that are reached from an MFMA without a conditional branch in between
    (the guarded form is: MFMAs, s_cbranch around a block of four v_dot2c, label)"""
    txt = open(path).read()
    out = []
    for km in re.finditer(r"^(\S*conv_wgrad_bl_kernel\S*):\s*; @\1\n", txt, re.M):
        name, i = km.group(1), km.end()
        lines = [ln.strip() for ln in txt[i:txt.find(".Lfunc_end", i)].split("\n")]
        code = [ln for ln in lines if ln and not ln.startswith(";") and not ln.startswith(".")]
        n_dot = 0
        for k, ln in enumerate(code):
            if not ln.startswith("v_dot2c"):
                continue
            n_dot += 1
            j = k - 1
            while j >= 0 and code[j].startswith("v_dot2c"):
                j -= 1
            while j >= 0 and not (code[j].startswith("s_cbranch") or code[j].startswith("v_mfma")):
                j -= 1
            if j < 0 or code[j].startswith("v_mfma"):
                out.append((name, ln))
        if n_dot == 0:
            out.append((name, "no v_dot2c at all: the bias sums moved -- update this check"))
    return out


def loads_between_wide_stores(path):
    """Round 6: the chunked store loops of the GEMM epilogues (conv_gemm_bl_kernel / conv_gemm_kernel) and of the slab form of
    conv_wgrad_bl_kernel must not LOAD between their 8 / 16-byte stores.  gfx950 counts loads and stores in one in-order
    counter (vmcnt), so the wait for a load that follows a store is a wait for that store's acknowledgement -- rounds 3-5 had
    a vector load (a development knob read from a __device__ variable) and a join-point wait in front of every chunk, and the
    "additive store phase" they produced cost 2.1 ms per train step.  A parity test cannot see this.  Flags every vector load
    that sits between two wide global stores with a vmcnt wait behind it (the per-element paths use narrow stores and are
    not looked at; the arena form of the weight gradient loads one row AHEAD of its stores by design and is exempt)."""
    txt = open(path).read()
    out = []
    for km in re.finditer(r"^(\S*(?:conv_gemm_bl_kernel|conv_gemm_kernel)\S*):\s*; @\1\n", txt, re.M):
        name, i = km.group(1), km.end()
        code = [ln.strip() for ln in txt[i:txt.find(".Lfunc_end", i)].split("\n")]
        code = [ln for ln in code if ln and not ln.startswith(";") and not ln.startswith(".")]
        last_wide, pending_load = False, None
        for ln in code:
            if ln.startswith("global_store_dwordx4") or ln.startswith("global_store_dwordx2"):
                if last_wide and pending_load is not None:
                    out.append((name, pending_load))
                last_wide, pending_load = True, None
            elif ln.startswith("global_store") or ln.startswith("s_barrier") or ln.startswith("v_mfma"):
                last_wide, pending_load = False, None
            elif last_wide and (ln.startswith("global_load") or (ln.startswith("buffer_load") and " lds" not in ln)):
                pending_load = ln
    return out


def main():
    srcs = sys.argv[1:] or ["gemm_conv.hip", "attention.hip"]
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for s in srcs:
            src = s if os.path.exists(s) else os.path.join(ROOT, "ml-mdm_amd", "csrc", s)
            out = os.path.join(td, os.path.basename(src) + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                                   "-I" + os.path.join(ROOT, "ml-mdm_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", out])
            fs = analyze(out)
            early = [f for f in fs if f[1] < MIN_WS]
            print("%s: %d reads within 10 wait states of their MFMA, %d earlier than %d" % (os.path.basename(src), len(fs), len(early), MIN_WS))
            for (name, ws), cnt in sorted(collections.Counter((f[0], f[1]) for f in early).items()):
                print("   %d wait states, x%d: %s" % (ws, cnt, name[:100]))
            bad += len(early)
            # second static guard on the same ISA: spilled vector registers.  The hot kernels run at 0-15 (the 8-wave GEMM
            # tiles sit at their 256-register limit); a recompile that pushes any kernel past SPILL_CAP is a scratch-memory
            # round trip inside an MFMA loop -- the kind of regression a parity test does not see.
            meta = open(out).read()
            spills = [(int(m.group(2)), m.group(1)) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", meta)]
            worst = sorted((sp for sp in spills if sp[0] > 0), reverse=True)
            print("   kernels with spilled VGPRs: %d of %d, worst %s" % (len(worst), len(spills), ["%d %s" % (n, k[:60]) for n, k in worst[:3]]))
            over = [sp for sp in worst if sp[0] > SPILL_CAP]
            for n, k in over:
                print("   OVER THE SPILL CAP (%d): %d %s" % (SPILL_CAP, n, k))
            bad += len(over)
            if os.path.basename(src) == "gemm_conv.hip":
                ung = unguarded_dot2(out)
                print("   conv_wgrad_bl_kernel: %d bias-sum dot products not behind a scalar branch" % len(ung))
                for name, ln in ung[:4]:
                    print("      %s: %s" % (name[:60], ln))
                bad += len(ung)
                lw = loads_between_wide_stores(out)
                print("   GEMM epilogues: %d vector loads between two wide stores of a chunked store loop" % len(lw))
                for name, ln in lw[:4]:
                    print("      %s: %s" % (name[:60], ln))
                bad += len(lw)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
