"""bench.py prints exactly one JSON line with the contract keys — at N = 1, and through the N > 1 code path (bucketing,
exchange, checks, max-over-ranks timing, the `exchange` object), which the test boxes can only run with several ranks on
ONE GPU and host collectives (`--backend gloo`; RCCL refuses two ranks on a device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"]


def _one_json_line(stdout):
    lines = [x for x in stdout.decode().splitlines() if x.strip()]
    assert len(lines) == 1, "stdout must carry exactly the JSON record:\n" + stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line(wm_lib):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "4000000", "--indices", "500000",
                        "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.5"], capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = _one_json_line(p.stdout)
    assert all(k in r for k in CONTRACT) and r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1
    assert r["unit"] == "GB/s" and r["higher_is_better"] is True and r["vs_baseline"] is None and r["dtype"] == "f32"
    assert "workload" in r["config"] and r["value"] > 0 and r["ms_per_step"] > 0
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["algorithmic_bytes_per_launch"] == 500000 * (8 + 512 + 512)
    cpu = r["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_line_over_host_collectives(wm_lib, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo",
           "--rows", "1000003", "--indices", "300000", "--steps", "3", "--warmup", "1"]
    p = subprocess.run(cmd, capture_output=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = _one_json_line(p.stdout)
    assert all(k in r for k in CONTRACT) and r["n_gpus"] == world and r["scaling"] == "weak"
    assert r["config"]["memory_type"] == "distributed" and r["config"]["rows_per_gpu"] == 1000003
    assert r["value"] > 0 and "cpu_baseline" not in r            # the host baseline is an N = 1 item
    assert r["exchange"]["bound"] == "xgmi" and "BRING-UP" in r["exchange"]["note"]
    assert r["roofline"]["bound"] == "hbm" and r["roofline"]["achieved"] > 0 and "local shard" in r["roofline"]["scope"]
    # whole-job aggregate: all ranks' lookups over the slowest rank's time
    assert abs(r["mlookups_per_s"] - world * 300000 / (r["ms_per_step"] * 1e-3) / 1e6) / r["mlookups_per_s"] < 0.02
