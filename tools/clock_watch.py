#!/usr/bin/env python
"""Shader clock / socket power while the train step runs (rocm-smi polled from a side process):
   gpurun -- python tools/clock_watch.py [unet64|idle] [seconds]
Prints min / median / max of sclk (MHz) and power (W) over the polling window, idle first, then under load."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def poll(seconds):
    rows = []
    t_end = time.time() + seconds
    while time.time() < t_end:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            sclk = [v for k, v in card.items() if "sclk" in k.lower()]
            pw = [v for k, v in card.items() if "power" in k.lower() and "W" in k]
            rows.append((time.time(), sclk[0] if sclk else None, pw[0] if pw else None, card if not rows else None))
        except Exception as ex:   # noqa: BLE001
            rows.append((time.time(), "error %s" % ex, None, None))
        time.sleep(0.2)
    return rows


def summarise(tag, rows):
    import re
    clk = [float(re.sub(r"[^0-9.]", "", str(r[1]))) for r in rows if r[1] and not str(r[1]).startswith("error")]
    pw = [float(r[2]) for r in rows if r[2] not in (None, "N/A")]
    clk.sort(); pw.sort()
    med = lambda v: v[len(v) // 2] if v else None   # noqa: E731
    print("%s: %d samples  sclk MHz min/med/max %s/%s/%s   power W min/med/max %s/%s/%s" % (
        tag, len(rows), clk[0] if clk else None, med(clk), clk[-1] if clk else None, pw[0] if pw else None, med(pw), pw[-1] if pw else None))


def main():
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 15
    first = poll(2)
    print("first sample:", json.dumps(first[0][3])[:1500])
    summarise("idle", first)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "5", "--no-cpu-baseline", "--no-reference-loop",
                          "--no-nested", "--no-nested1024", "--no-roofline", "--no-sampling"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    time.sleep(25)   # import, build the model, warm up
    rows = poll(secs)
    summarise("train step (UNet-64, batch 64)", rows)
    out, _ = p.communicate(timeout=600)
    for line in out.splitlines():
        if line.startswith("{"):
            print("ms_per_step", json.loads(line)["ms_per_step"])


if __name__ == "__main__":
    main()
