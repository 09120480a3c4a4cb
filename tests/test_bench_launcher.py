"""`python bench.py --gpus N` is the N-rank job: without a launcher around it the file starts its own ranks under
torch.distributed.run, one per GPU; with fewer GPUs than ranks it fails at once instead of printing a 1-GPU line for an N-GPU request.
Everything here runs on a box without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def gpu_count():
    import torch
    return torch.cuda.device_count()


def test_more_gpus_asked_for_than_present_fails_at_once_and_prints_no_line():
    n = gpu_count() + 2
    r = run(["--gpus", str(n), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert f"needs {n} GPUs, found {gpu_count()}" in r.stderr
    assert '"n_gpus"' not in r.stdout, "no measurement line for a request that cannot be served"


def test_launch_check_starts_one_rank_per_requested_gpu_without_touching_one():
    r = run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["launch_check"] == "ok" and line["n_ranks"] == 2
    ranks = sorted(line["ranks"], key=lambda e: e["rank"])
    assert [e["rank"] for e in ranks] == [0, 1] and [e["local_rank"] for e in ranks] == [0, 1]
    assert all(e["world_size"] == 2 and e["env"]["WORLD_SIZE"] == "2" and e["env"]["MASTER_ADDR"] == "127.0.0.1" for e in ranks)
    assert ranks[0]["pid"] != ranks[1]["pid"], "one process per GPU"


def test_a_launcher_with_another_world_size_than_gpus_is_refused():
    """WORLD_SIZE from a launcher that disagrees with --gpus: never an n_gpus line for another N than the one asked for"""
    r = run(["--gpus", "4", "--launch-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in r.stderr
    r = run(["--gpus", "1", "--launch-check"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr


def test_config4_takes_the_same_launcher():
    n = gpu_count() + 2
    r = run(["--gpus", str(n), "--config", "4", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and f"needs {n} GPUs" in r.stderr and '"n_gpus"' not in r.stdout


def test_summary_is_the_last_key_and_names_every_secondary_row():
    sys.path.insert(0, ROOT)
    import bench
    line = {"ms_per_step": 0.78, "roofline": {"frac": 0.6, "whole_path_frac": 0.5}, "cpu_baseline": {"value": 2700.0},
            "extra": {"config3_cell_errors": {"ms_per_step": 0.85}, "config4_n1": {"value": 2e5, "without_solve": {"frames_per_s": 6e5}},
                      "config5_extract_4096": {"captures_per_s": 45000.0}, "single_frame": {"ms_per_frame": 0.19}, "mode66": {"threshold_hbm_frac": 0.52},
                      "ingest": {"error": "x"}}}
    s = bench.summary(line)
    assert s["configs2_ms_per_step"] == 0.85 and s["configs3_n1_without_solve"] == 6e5 and s["configs4_captures_per_s"]["4096"] == 45000.0
    assert s["single_frame_ms"] == 0.19 and s["mode_k1_frac"]["66"] == 0.52 and s["mode_k1_frac"]["67"] is None
    assert len(json.dumps(s)) < 1500, "must fit the tail of the line the driver records"
