"""bench.py contract on a GPU box: one JSON line with the required fields, and the multi-rank code path (two ranks on the one GPU
over gloo -- P3D_BENCH_REHEARSAL -- because a second GPU is not available to the tests)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline"}


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_single_rank_line():
    res = subprocess.run([sys.executable, "bench.py", "--nslices", "16", "--steps", "6", "--warmup", "1", "--cpu-seconds", "1"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _last_json(res.stdout)
    assert REQUIRED <= set(line) and line["n_gpus"] == 1 and line["steps"] == 6 and line["value"] > 0
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof) and roof["bound"] == "hbm"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    cpu = line["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["kind"] == "port"
    assert line["config"]["workload"] and "model" not in line["config"]


def test_two_ranks_rehearsal():
    env = dict(os.environ, P3D_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", "bench.py", "--gpus", "2", "--nslices", "16", "--steps", "6", "--warmup", "1"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _last_json(res.stdout)
    assert line["n_gpus"] == 2 and line["config"]["slices_per_gpu"] == 8 and line["value"] > 0 and line["cpu_baseline"] is None
    assert line["gather_ms"] > 0


@pytest.mark.parametrize("config,extra", [(0, []), (3, ["--nslices", "8", "--steps", "5"]), (4, ["--nil", "256", "--nxl", "128", "--nslices", "2", "--steps", "3"])])
def test_other_configurations_emit_a_line(config, extra):
    """bench.py --config i: every BASELINE configuration has a driver-runnable leg with its own roofline and CPU baseline (the
    wavelet and shearlet legs on reduced sizes here; their full-size parity is in test_gpu_wavelet / test_gpu_shearlet)."""
    res = subprocess.run([sys.executable, "bench.py", "--config", str(config), "--warmup", "1", "--cpu-seconds", "1", "--repeats", "2"] + extra,
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = _last_json(res.stdout)
    assert REQUIRED <= set(line) and line["value"] > 0 and line["repeats"]["n"] == 2
    assert line["roofline"]["frac"] > 0 and line["roofline"]["algorithmic_bytes_per_launch"] > 0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    assert line["steady_state_iterations_per_s"] > 0
