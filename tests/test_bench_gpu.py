"""GPU tests (-m gpu) of bench.py itself: the JSON contract of the default line, and the N > 1 code path (process
group, barrier, max-over-ranks time, frames summed over ranks) exercised on the ONE GPU a test box has."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pytestmark = pytest.mark.gpu


def run(cmd, extra_env=None, timeout=600, rc=0):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **(extra_env or {}))
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert (r.returncode == 0) == (rc == 0), r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout   # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def check_line(line, n_gpus, steps, warmup):
    assert line["unit"] == "frames/sec" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["n_gpus"] == n_gpus and line["steps"] == steps and line["warmup"] == warmup
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["fixed_warmup"] >= 200  # the untimed launches in front of --warmup are stated in the line
    assert abs(line["value"] - n_gpus * 1e3 / line["ms_per_step"]) < 0.02 * line["value"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # the kernel's own time (HIP events on its stream) can never exceed the wall clock per step
    assert rf["avg_launch_ms"] <= line["ms_per_step"] * 1.02
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / rf["avg_launch_ms"] / 1e6) < 0.01 * rf["achieved"]
    # both clocks are in the line: SURVEY 8d's formula on `value`, and what separates the two
    per_gpu_fps = line["value"] / n_gpus
    frames_per_launch = max(line["config"].get("frames_per_launch", 1), 1)
    assert abs(rf["frac_by_value"] - rf["algorithmic_bytes_per_launch"] / frames_per_launch * per_gpu_fps / 1e9 / rf["peak"]) < 2e-3
    assert rf["frac_by_value"] <= rf["frac"] * 1.02
    assert abs(rf["sync_overhead_us_per_step"] - 1e3 * (line["ms_per_step"] - rf["avg_launch_ms"])) < 0.5
    assert ("sync_overhead_note" in rf) == (steps < 200)


def check_per_rank(line, n_ranks):
    """the N > 1 line says what every rank did"""
    pr = line["per_rank"]
    assert len(pr["frames_per_sec"]) == len(pr["avg_launch_ms"]) == len(pr["roofline_frac"]) == len(pr["elapsed_s"]) == n_ranks
    assert pr["min"] == min(pr["frames_per_sec"]) and pr["max"] == max(pr["frames_per_sec"])
    assert 0 <= pr["slowest_rank"] < n_ranks and pr["frames_per_sec"][pr["slowest_rank"]] == pr["min"]
    assert pr["slowest_rank_avg_launch_ms"] == pr["avg_launch_ms"][pr["slowest_rank"]] > 0
    assert all(0 < f < 1 for f in pr["roofline_frac"])
    # value = all frames over the slowest rank's time: never more than the sum of the ranks' own rates
    assert line["value"] <= sum(pr["frames_per_sec"]) * 1.001
    assert abs(max(pr["elapsed_s"]) * 1e3 / line["steps"] - line["ms_per_step"]) < 0.02 * line["ms_per_step"]


def test_default_line_as_the_driver_runs_it():
    """python bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command): roofline, secondary, cpu_baseline present"""
    line = run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2"])
    check_line(line, 1, 20, 5)
    assert line["roofline"]["valu"]["frac"] > 0.2
    # config 2 alone and four such channels per launch (round 5), 720p50 (the reference's third format) fused / through the channel kernel /
    # four channels per launch, config 3
    # ... and file playback (round 5): a 1080p and a 720p yuv420p clip on a 1080p channel, 4 x 1080i on a 1080p channel
    assert [s["config"][:5] for s in line["secondary"]] == ["2: 1 ", "2 x 4", "720p5", "720p5", "720p5", "3: 1 ", "f1: 1", "f2: 1", "f3: 1"]
    for s in line["secondary"]:
        assert 0 < s["roofline"]["frac"] < 1
    batch, alone = line["secondary"][1], line["secondary"][0]
    assert batch["channels_per_launch"] == 4 and batch["ms_per_frame"] < 0.92 * alone["ms_per_frame"]  # the batch kernel pays
    assert alone["bytes_as_benched"] > alone["algorithmic_bytes"]  # config 2's bytes both ways
    ceil = line["roofline"]["ceiling"]  # north_star's 0.80 answered by a number
    assert 0.45 < ceil["hbm_frac_at_valu_issue_peak"] < 0.65 and ceil["hbm_frac_at_sustained_issue_rate"][0] < ceil["hbm_frac_at_sustained_issue_rate"][1] < ceil["hbm_frac_at_valu_issue_peak"]
    assert line["roofline"]["frac"] < ceil["hbm_frac_at_valu_issue_peak"]
    assert "shader_clock" in line["roofline"]
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert line["value"] > 100 * cb["value"]
    # roofline.traffic is measured in the run when rocprofv3 is there (two --pmc child passes), and says so
    import shutil
    rf = line["roofline"]
    if shutil.which("rocprofv3"):
        assert rf["traffic_source"].startswith("measured in this run"), rf.get("traffic_not_measured")
        assert 0.95 < rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 1.25, rf
        # every secondary roofline names what binds it, from counters taken in the run
        assert "secondary_counters_not_measured" not in line, line.get("secondary_counters_not_measured")
        for s in line["secondary"]:
            r2 = s["roofline"]
            assert r2["binding_resource"] in ("valu issue", "hbm") and 0 < r2["valu"]["frac"] < 1 and r2["traffic"] > 0.9 * s["algorithmic_bytes"], s
            unit_ms = [v for k, v in s.items() if k.startswith("ms_per_")][0]
            assert abs(r2["valu"]["achieved"] - r2["valu"]["instructions_per_unit"] / (unit_ms * 1e-3) / 1e12) < 0.02 * r2["valu"]["achieved"]
        assert rf["valu"]["source"].startswith("measured in this run")
    else:
        assert rf["traffic_source"].startswith("recorded")


def test_rank_path_under_torchrun_with_rccl_on_one_gpu():
    """one rank started the way the driver starts N: the RCCL process group, barrier and all-reduce run for real"""
    line = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29571", "bench.py", "--gpus", "1", "--steps", "30", "--warmup", "5", "--cpu-seconds", "0", "--no-secondary"],
               {"PH_BENCH_FORCE_DIST": "1", "PH_BENCH_ROUTE_HEIGHT": "540"})
    check_line(line, 1, 30, 5)
    check_per_rank(line, 1)
    # the distributed path carries BASELINE config 5 in the same line: 2 channels per rank, every fourth layer routed
    # through ph_route_* (RCCL; with one rank the peer is the own rank), checked by fingerprint
    rt = line["route"]
    assert rt.get("error") is None, rt
    assert rt["fingerprint_check"] == "ok" and rt["rccl_ranks_in_communicator"] == 1 and rt["ranks"] == 1
    assert rt["channels"] == 2 and rt["routes_crossing_ranks_per_rank"] == 2
    assert rt["bytes_per_hop"] == 3840 * 540 * 16 and rt["route_bytes_per_rank_per_step"] == 4 * rt["bytes_per_hop"]
    assert rt["hop_ms_unoverlapped"] > 0 and rt["frames_per_sec"] > 0 and rt["path"].startswith("ph_route")


def test_two_ranks_sum_their_frames():
    """two ranks on the one GPU (gloo): value is the whole job's rate, one line from rank 0, no cpu_baseline / secondary at N > 1"""
    line = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29572", "bench.py", "--gpus", "2", "--steps", "30", "--warmup", "5", "--width", "1920", "--height", "1080"],
               {"PH_BENCH_SHARE_GPU": "1"})
    check_line(line, 2, 30, 5)
    check_per_rank(line, 2)
    assert line["cpu_baseline"] is None and "secondary" not in line
    assert line["config"]["channels"] == 2


def test_a_stalled_route_does_not_cost_the_line():
    """config 5 abandoned by the watchdog (limit set to nothing): rank 0 still prints the headline line, the failure recorded in
    it - and the job ends with a non-zero exit code, so a hung RCCL group is not mistaken for a clean run"""
    line = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29573", "bench.py", "--gpus", "1", "--steps", "30", "--warmup", "5", "--cpu-seconds", "0", "--no-secondary"],
               {"PH_BENCH_FORCE_DIST": "1", "PH_BENCH_ROUTE_HEIGHT": "540", "PH_BENCH_ROUTE_TIMEOUT": "0.001"}, rc=3)
    check_line(line, 1, 30, 5)
    assert "abandoned by the watchdog" in line["route"]["error"]
