"""bench.py's output contract, on the reduced `mini` architecture (development workload: seconds, not a reportable number):
one JSON line from rank 0 with the keys the driver reads, at N = 1 and -- both ranks on the one GPU, over RCCL with a real
peer (each rank claims its own NCCL_HOSTID: tests/test_distributed_gpu.py) -- at N = 2."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--workload", "mini", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-reference-loop", "--no-nested",
         "--no-nested1024", "--no-sampling"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _line(cmd, env):
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-2000:], out.stderr[-3000:])
    return json.loads(lines[0])


def _env(**kw):
    env = dict(os.environ, BENCH_SETTLE_STEPS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_one_rank_line_has_the_contract_keys():
    d = _line([sys.executable, "bench.py", "--gpus", "1"] + QUICK, _env())
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and "synthetic" in d["data"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-3      # N x steps/s, N = 1
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and r["bound"] in ("mfma", "hbm")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "cpu_baseline" not in d          # --no-cpu-baseline (the default run carries it: profiles/r06_bench_unet64_b64_line.json)
    assert len(d["binary"]["sha256"]) == 64


def test_two_ranks_on_one_gpu_over_rccl():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", "bench.py", "--gpus", "2", "--no-roofline"] + QUICK
    d = _line(cmd, _env(MDM_BENCH_DEVICE="0"))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    comm = d["config"]["comm"]
    assert comm["backend"] == "nccl" and comm["world_size"] == 2 and len(comm["bucket_timeline_ms"]) >= 1
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 2.0) < 2e-3                           # whole job: 2 x steps/s
