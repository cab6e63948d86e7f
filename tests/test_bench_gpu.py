"""GPU tests of the measurement contract: bench.py prints ONE JSON line with the agreed keys, in the default mode (three samples
in flight, launch-plan replay), in the eager mode, and with two ranks (control flow of the multi-GPU launch: the collective
decision about the recording, the closing all-gather, max over ranks) -- the two ranks share cuda:0 over gloo, which is what this
box allows; on a multi-GPU node the same code runs over RCCL."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--width", "640", "--height", "480", "--steps", "6", "--warmup", "2", "--samples", "3", "--roofline-steps", "4", "--verify-steps", "12",
         "--no-cpu-baseline", "--steady-seconds", "0.2", "--settle-seconds", "0.2"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "single_stream_eager", "steady_state"}


def _line(out: str) -> dict:
    lines = [ln for ln in out.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _check(line: dict, n_gpus: int) -> None:
    assert KEYS <= set(line), sorted(KEYS - set(line))
    assert line["n_gpus"] == n_gpus and line["steps"] == 6 and line["higher_is_better"] is True
    assert line["value"] > 0 and abs(line["value"] - n_gpus * 1e3 / line["ms_per_step"]) < 1e-2 * line["value"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["launches"] == 5 * r["steps_with_events"]  # five pmn_warp_correlate launches per depth map
    assert r["tap_bytes_per_step"] > r["alg_bytes_per_step"] and 0 < r["l1_frac"] < 1
    assert line["steady_state"]["steps"] >= line["steps"] and line["steady_state"]["value"] > 0
    c = line["config"]
    # the collectives saw every rank, the spread over ranks brackets the max-reduced figure, every rank reports its NUMA binding
    assert c["ranks_seen"] == n_gpus and len(c["numa"]) == n_gpus and all(isinstance(x, str) and x for x in c["numa"])
    assert c["ms_per_step_rank_min"] <= c["ms_per_step_rank_max"] and abs(c["ms_per_step_rank_max"] - line["ms_per_step"]) < 1e-3


def test_bench_line_default_and_eager():
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    for extra in ([], ["--eager"]):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, capture_output=True, text=True,
                           timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        line = _line(p.stdout)
        _check(line, 1)
        assert line["config"]["in_flight"] == (1 if extra else 3)
        # the timed mode's outputs are the eager forward's, bit for bit, on the runtime's default hardware queues
        assert "default" in line["config"]["hardware_queues"], line["config"]["hardware_queues"]
        ver = line["outputs_verified"]
        assert (ver is None) == bool(extra) and (extra or (ver["steps"] >= 6 and ver["steps_that_differ_from_the_eager_forward"] == 0)), ver
        # both input modes of the replay are on the line (in place = `value`; copied into the slot = rounds 2-4's `value`)
        other = line["value_other_input_mode"]
        assert (other is None) == bool(extra) and (extra or (other["value"] > 0 and "copied" in other["mode"]))
        assert ("launch-plan replay" in line["config"]["launch"]) == (not extra), line["config"]["launch"]
        assert line["warmup"] == line["config"]["untimed_steps_before_the_timed_region"] >= line["warmup_requested"] == 2


def test_bench_graph_mode():
    """--launch graph = rounds 2-5's replay form (HIP graphs), on the same default hardware queues, held to the same verification."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--launch", "graph"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _line(p.stdout)
    _check(line, 1)
    assert "default" in line["config"]["hardware_queues"] and "HIP-graph replay" in line["config"]["launch"]
    assert line["outputs_verified"]["steps_that_differ_from_the_eager_forward"] == 0


def test_bench_two_ranks_on_one_gpu():
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    port = str(29700 + os.getpid() % 1000)
    env = dict(os.environ, PMN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, cwd=ROOT) for r in range(2)]
    outs = [q.communicate(timeout=900) for q in procs]
    for q, (so, se) in zip(procs, outs):
        assert q.returncode == 0, se[-3000:]
    line = _line(outs[0][0])
    _check(line, 2)
    assert "{" not in outs[1][0]  # only rank 0 prints
    assert line["config"]["in_flight"] == 3 and line["scaling"] == "weak"
    assert line["config"]["backend"] == "gloo"


def test_bench_one_rank_over_rccl():
    """The product's collective backend, for real: launched the torch.distributed.run way with WORLD_SIZE=1, bench.py initialises
    backend "nccl" (= RCCL) with device_id=, and its barrier / all_reduce / closing all_gather_into_tensor run on device memory
    through librccl -- the same calls eight ranks make."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29800 + os.getpid() % 1000))
    env.pop("PMN_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _line(p.stdout)
    _check(line, 1)
    assert line["config"]["backend"] == "nccl"


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher (the form of the driver's N=1 command) spawns the two ranks itself and rank 0
    prints the single line; gloo because this box has one GPU."""
    assert torch.cuda.is_available(), "GPU tests selected but no ROCm device is visible"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["PMN_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _line(p.stdout)
    _check(line, 2)
    assert line["scaling"] == "weak"
