"""The N-rank path of bench.py end to end on the one GPU of the test box: `python -m torch.distributed.run --nproc-per-node 3
bench.py --gpus 3 --scaling strong` under PTMI_BENCH_REHEARSAL=1 (all ranks share cuda:0 and talk over gloo), as a fresh child
process, at a reduced sample count.  Rank 0's gathered image must equal the image ONE context renders, bit for bit: row stripes,
seeds from the global pixel index, the overlapped gather and the reassembly change nothing.  (Three ranks: a GPU box admits six
processes on its card, and the test runner and the launcher's agent are two of them; the 8-part shape itself -- 10-row stripes, 270 rows per part -- is
rehearsed in one process by tests/test_group.py and over gloo, without the card, by tests/test_parallel.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_strong_scaling_rehearsal_gathers_the_one_context_image():
    env = dict(os.environ, PTMI_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "3", "--scaling", "strong", "--spp", "6",
           "--steps", "2", "--warmup", "1", "--check-image", "--no-n1-reference"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 3 and out["scaling"] == "strong"
    assert out["config"]["width"] == 3840 and out["config"]["height"] == 2160
    assert out["gathered_image_equals_one_context"] is True
    # the line explains itself (bench.py, N > 1): the job checked its own image before anything was timed, says how many ranks and
    # DISTINCT devices the collective saw (one here: the rehearsal shares cuda:0), what every rank did and what the gather cost
    col = out["collective"]
    assert col["world_size"] == 3 and col["backend"] == "gloo"
    assert col["distinct_devices"] == 1 and len(col["devices"]) == 3
    assert col["gathered_image_equals_one_context_at_6_spp"] is True
    ranks = col["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1, 2]
    assert sum(r["rows"] for r in ranks) == 2160 and all(r["kernel_ms"] > 0 and r["live_bounces"] > 0 for r in ranks)
    assert col["imbalance"] >= 1.0
    assert col["render_only_ms_per_step"] > 0 and isinstance(col["gather_hidden_frac"], float)
    assert max(r["elapsed_ms_per_step"] for r in ranks) == pytest.approx(out["ms_per_step"], rel=1e-3)
    # ... and names the binary that produced it
    assert out["binary_build_id"] == out["code_id_now"]


def test_bench_weak_scaling_rehearsal_two_ranks():
    """`--scaling weak` (every rank renders 1920x1080 pixels of a 1920 x 2160 image: per-GPU work fixed, the mode whose N-series is ONE workload
    per GPU with the N = 1 line) through the same pre-check, gather and image comparison, two ranks on the one card over gloo."""
    env = dict(os.environ, PTMI_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "weak", "--spp", "4",
           "--steps", "2", "--warmup", "1", "--check-image"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["width"] == 1920 and out["config"]["height"] == 2160 and out["config"]["rows_per_gpu"] == 1080
    assert out["gathered_image_equals_one_context"] is True
    assert out["collective"]["gathered_image_equals_one_context_at_4_spp"] is True
    assert [r["rows"] for r in out["collective"]["per_rank"]] == [1080, 1080]
    # whole-job throughput: both ranks' pixels over the slowest rank's time
    assert out["value"] == pytest.approx(1920 * 2160 * 4 * 8 * 2 / (out["ms_per_step"] * 2 * 1e-3) / 1e6, rel=1e-3)
