"""bench.py end to end on the GPU box: the N = 1 line carries the forward_* keys next to the train-step metric, and the N > 1 code path
(torch.distributed.run, one process per rank, FlatGradSync collectives, max-over-ranks timing, rank-0 JSON) runs with two ranks that share
the one GPU of the test box over gloo -- so that the driver's first real multi-GPU run does not die on plumbing."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{") and l.rstrip().endswith("}")]
    assert len(lines) == 1, f"bench.py must print exactly ONE JSON line on stdout, got {len(lines)}:\n{out[-2000:]}"
    return json.loads(lines[0])


def test_bench_single_gpu_line_has_forward_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == 128 and line["unit"] == "images/sec" and line["value"] > 0
    assert line["roofline"] is not None and line["roofline"]["bound"] in ("mfma", "hbm") and 0 < line["roofline"]["frac"] < 1
    for k in ("forward_images_per_sec", "forward_ms", "forward_frac_of_bf16_peak"):
        assert k in line and line[k] is not None and line[k] > 0, k
    # the forward pass is (much) cheaper than the train step on the same batch
    assert line["forward_ms"] < line["ms_per_step"]
    assert "hipGraph replay" in line["forward_note"]          # (whole batch or --infer-parts graph branches: whichever the schedule probe measured faster -- forward_schedule_probe_ms)
    assert line["stage_errors"] == 0 and "forward_schedule_probe_ms" in line


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X: bench.py --gpus 2 over RCCL, one device per rank (BASELINE config 4)")
def test_bench_two_ranks_rccl():
    """The driver's N > 1 launch line on real devices: torch.distributed.run, backend nccl (= RCCL), FlatGradSync all-reduces over xGMI."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LMV_BENCH_SINGLE_DEVICE", None); env.pop("LMV_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:] + "\n" + r.stderr[-3000:])
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 256 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    assert line["value"] > 0 and line["value"] == line["value"]


def test_bench_two_ranks_on_one_device():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, LMV_BENCH_SINGLE_DEVICE="1", LMV_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:] + "\n" + r.stderr[-3000:])
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 256 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    assert line["value"] > 0 and line["value"] == line["value"] and line["ms_per_step"] > 0          # finite
    assert "cpu_baseline" not in line
