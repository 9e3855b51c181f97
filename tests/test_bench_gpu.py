"""bench.py keeps its contract: one JSON line with the driver's keys, the roofline and cpu_baseline objects,
and the same under a 2-rank launch (ranks share the box's GPU, gloo for the scalar reductions)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line(dev):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nodes", "300000", "--steps", "4",
                        "--warmup", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["unit"] == "edges/s" and d["dtype"] == "f32" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    nnz = d["config"]["stored_entries_per_gpu"]
    assert abs(d["value"] - nnz / (d["ms_per_step"] * 1e-3)) <= 0.02 * d["value"]     # value = units / time
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


def test_two_rank_line(dev):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MP_DIST_BACKEND="gloo", MP_SHARE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--nodes", "200000", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    # whole-job aggregate: both ranks' entries over the slowest rank's time
    per_rank = d["config"]["stored_entries_per_gpu"]
    assert d["value"] > 1.5 * per_rank / (d["ms_per_step"] * 1e-3) * 0.5
    assert abs(d["value"] - 2 * per_rank / (d["ms_per_step"] * 1e-3)) <= 0.05 * d["value"]


def test_two_ranks_without_a_launcher_one_line_with_the_step_object(dev):
    """the driver's command: `python bench.py --gpus 2 ...` and nothing around it.  bench.py starts its own two ranks
    (they share this box's GPU: gloo for the collectives) and prints ONE line that carries the weak-scaling aggregate
    AND the data-parallel training step with its gradient exchange"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(MP_DIST_BACKEND="gloo", MP_SHARE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nodes", "200000", "--steps", "3",
                        "--warmup", "1", "--centres", "64", "--step-steps", "3"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    rf = d["roofline"]
    assert rf["peak_all_gpus"] == 2 * 8000.0 and abs(rf["frac_all_gpus"] - rf["achieved_all_gpus"] / rf["peak_all_gpus"]) < 1e-9
    st = d["step"]
    assert st["n_gpus"] == 2 and st["world_size"] == 2 and st["collective_backend"] == "gloo" and st["collectives_executed"]
    for k in ("allreduce_ms", "allreduce_exposed_ms", "lpt_imbalance", "ms_per_step", "ms_per_step_no_exchange",
              "ms_per_step_fresh_batch"):
        assert k in st and st[k] is not None and st[k] >= 0, k
    assert 1.0 <= st["lpt_imbalance"] < 1.5
