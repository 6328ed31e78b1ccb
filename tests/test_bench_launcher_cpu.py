"""`python bench.py --gpus N` must start N ranks itself (the driver's command line) and report n_gpus = N.
Runs the launcher, the rendezvous on 127.0.0.1, the barrier + max-over-ranks timing and the JSON assembly on CPU
(gloo, 2 ranks) through bench.py's model-free --selftest step."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--selftest", "--steps", "5", "--warmup", "1"] + extra,
                         capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    line = _run(["--gpus", "2", "--backend", "gloo"])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 5 and line["warmup"] == 1
    assert line["config"]["parallelism"] == "replicas" and line["scaling"] == "weak"
    assert abs(line["steps_per_s"] * line["ms_per_step"] / 1e3 - 2.0) < 1e-6      # whole-job rate = 2 ranks' steps / max time


def test_tp_child_job_reports_what_its_collectives_ran_on():
    """The replica headline's `tp_70b` block (configuration E tensor-parallel over the same N ranks, a child job with a
    timeout): N ranks really formed the group (`rccl_ranks` == N == `n_gpus`), the run says at top level which transport the
    collectives used (`allreduce_kind`, `xgmi_self_check`, `collectives_env`) -- what a first real multi-GPU run is read by."""
    line = _run(["--gpus", "2", "--backend", "gloo"])
    tp = line["tp_70b"]
    assert "error" not in tp, tp
    assert tp["n_gpus"] == 2 and tp["rccl_ranks"] == 2 and tp["scaling"] == "strong" and tp["config"]["parallelism"] == "tp2"
    assert tp["allreduce_kind"] == "gloo" and tp["collectives_env"] == "xgmi"
    assert tp["xgmi_self_check"].startswith("not running on the xGMI kernels")
    assert tp["allreduce"]["sum_ok"] is True


def test_gpus_1_stays_in_process():
    line = _run(["--gpus", "1"])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["config"]["parallelism"] == "single"


def test_world_size_mismatch_is_refused():
    e = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--selftest", "--gpus", "1"], capture_output=True,
                         text=True, timeout=120, env=e)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
