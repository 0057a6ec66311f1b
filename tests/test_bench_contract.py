"""``bench.py``'s own contract: one JSON line; N > 1 self-validating (``parity_vs_single``, ``rccl_ranks``, per-rank step
times); every failure ONE JSON error line + a non-zero exit code.  The N > 1 cases run the real ``bench.main()`` with the
ranks as processes sharing cuda:0 over host-staged gloo (``ANEMOI_AMD_BENCH_SHARE_GPU=1``): the code path the driver's
8-GPU run takes, minus RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _last_json(text: str) -> dict:
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_failure_is_one_json_error_line():
    """No GPU in the CPU suite's container: the run must end in a JSON error line naming the stage, exit code != 0 -- never
    a bare traceback on stdout, never a result line.  (On a GPU box the same command is the driver's default run.)"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU (the failure under test is the missing device)")
    res = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0", "--workload", "cfg1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, WORLD_SIZE="1", RANK="0"))
    assert res.returncode != 0
    out = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, res.stdout
    line = json.loads(out[0])
    assert "error" in line and "stage" in line and "metric" not in line


def test_bench_gpus_mismatch_refused():
    """``--gpus`` must be the launcher's WORLD_SIZE: a mismatch is one JSON error line, not a silently smaller run."""
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, WORLD_SIZE="2", RANK="0"))
    assert res.returncode != 0
    out = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, res.stdout
    line = json.loads(out[0])
    assert "error" in line and "WORLD_SIZE=2" in line["error"] and "metric" not in line


def _env_without_launcher(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bench_self_launch_relays_one_error_line_without_a_gpu():
    """``python bench.py --gpus 2`` with no launcher starts its two ranks itself (before touching the GPU) and relays ONE
    JSON line.  In the CPU container both ranks fail (no device): the parent prints one error line, exit code != 0."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU (the GPU boxes run test_bench_self_launch_on_a_shared_gpu)")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "cfg1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=_env_without_launcher())
    assert res.returncode != 0
    out = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, res.stdout
    line = json.loads(out[0])
    assert "error" in line and "stage" in line and "metric" not in line


def test_bench_watchdog_thread_prints_one_error_line_and_exits_4():
    """``bench.Stage`` alone, no GPU: a stage with a time limit that overruns ends the process with exit code 4 and ONE
    JSON error line naming the stage; stages without a limit (build, timed steps, cpu baseline) are never interrupted."""
    code = ("import os, sys, time; sys.path.insert(0, %r); import bench; "
            "os.environ['ANEMOI_AMD_BENCH_WATCHDOG_SCALE'] = '0.005'; s = bench.Stage(); s.watch(); "
            "s[0] = 'build'; time.sleep(2.5); s[0] = 'first all_reduce'; time.sleep(30); print('not reached')" % ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 4, (res.returncode, res.stderr[-1500:])
    out = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(out) == 1 and "not reached" not in res.stdout
    line = json.loads(out[0])
    assert line["stage"] == "first all_reduce" and "watchdog" in line["error"]


@pytest.mark.gpu
def test_bench_self_launch_on_a_shared_gpu():
    """The same command on a GPU box (the two ranks share cuda:0 over host-staged gloo): one result line, exit code 0, with
    the N > 1 self-checks and the per-exchange timing block."""
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg1",
                          "--dtype", "fp32", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200,
                         env=_env_without_launcher(ANEMOI_AMD_BENCH_SHARE_GPU="1"))
    assert res.returncode == 0, res.stderr[-3000:]
    out = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, res.stdout
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["parity_vs_single"]["finite"]
    assert line["run_to_run_identical"] is True  # (the last timed output against two more steps)
    ex = line["exchanges"]
    assert ex["alone_us"]["processor"]["median"] > 0 and ex["in_step"]["exchanges"] >= 4  # 4 blocks (+ the decoder's)
    assert 0.0 <= ex["in_step"]["exposed_fraction_of_step"]


@pytest.mark.gpu
def test_bench_parity_failure_prints_the_error_line_only():
    """A run whose output fails its comparison with the CPU oracle must not print a result line: with every parity bound scaled
    to 1e-6 of its value (test hook) a healthy config-2 run ends with exit code 3 and ONE JSON line that carries the error and
    the measured errors -- no "metric", no "value"."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", ANEMOI_AMD_BENCH_PARITY_BOUND_SCALE="1e-6")
    res = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--workload", "cfg2"], capture_output=True,
                         text=True, timeout=900, env=env)
    assert res.returncode == 3, (res.returncode, res.stderr[-2000:])
    out = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, res.stdout[-3000:]
    line = json.loads(out[0])
    assert "error" in line and line["stage"] == "parity" and "metric" not in line and "value" not in line
    assert line["parity"]["output_rel_err"] > 0 and line["parity_fp32"]["output_rel_err"] > 0


@pytest.mark.gpu
def test_bench_watchdog_ends_a_rank_whose_peer_never_arrives():
    """World 2 with only rank 0 started: process-group init can never complete.  The watchdog prints ONE JSON error line
    naming the stage and ends the process with exit code 4 -- bounded, no restart."""
    port = 29600 + (os.getpid() % 300)
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               ANEMOI_AMD_BENCH_SHARE_GPU="1", ANEMOI_AMD_BENCH_WATCHDOG_SCALE="0.05")  # init limit 300 s -> 15 s
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "cfg1",
                          "--dtype", "fp32", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 4, (res.returncode, res.stderr[-2000:])
    out = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, res.stdout
    line = json.loads(out[0])
    assert "watchdog" in line["error"] and line["stage"] == "init_process_group"


@pytest.mark.gpu
@pytest.mark.parametrize("world,workload,extra", [
    (2, "cfg1", ["--dtype", "fp32"]),      # (config 1's head size 4 has no bf16 folded edge kernel: f32 there)
    (3, "cfg1", ["--dtype", "fp32"]),
    (2, "cfg2", ["--rollout", "2"]),       # config 4's stepping (state sharded between lead times) at O96 size, bf16
    (4, "cfg2", []),
])
def test_bench_main_world_n_ranks_sharing_one_gpu(world, workload, extra):
    """``bench.py --gpus N`` end to end (ranks share cuda:0, gloo): the line carries ``parity_vs_single`` -- the partitioned
    forward against the unsharded one, checked BEFORE timing on every rank --, the rank count the process group really has,
    per-rank step times, the halo summary; exit code 0."""
    port = 29100 + (os.getpid() % 400)
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               ANEMOI_AMD_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, BENCH, "--gpus", str(world), "--steps", "3", "--warmup", "1", "--workload", workload,
           "--no-cpu-baseline", *extra]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert [p.returncode for p in procs] == [0] * world, [o[1][-1500:] for o in outs]
    line = _last_json(outs[0][0])
    assert all(not [ln for ln in o[0].splitlines() if ln.startswith("{")] for o in outs[1:])  # rank 0 alone prints
    assert line["n_gpus"] == world and line["rccl_ranks"] == world and line["scaling"] == "strong"
    assert line["run_to_run_identical"] is True
    pv = line["parity_vs_single"]
    assert pv["finite"] and pv["ranks_checked"] == world and pv["rows_checked"] > 0
    assert pv["max_rel_err"] <= pv["bound"], pv
    ranks = line["ms_per_step_ranks"]
    assert len(ranks["all"]) == world and ranks["min"] <= ranks["max"]
    assert abs(ranks["max"] - line["ms_per_step"]) < 1e-2 * line["ms_per_step"] + 1e-3  # value = MAX over ranks
    assert line["halo"]["own_mesh_rows"] > 0 and "DEBUG" in line["config"]["parallelism"]
    ex = line["exchanges"]  # per-exchange timing: every halo all-to-all-v alone, and start / exposed wait inside a step
    assert set(ex["alone_us"]) >= {"processor"} and ex["in_step"]["exchanges"] > 0


def test_bench_refuses_a_relaxed_parity_gate():
    """``ANEMOI_AMD_BENCH_PARITY_BOUND_SCALE`` exists for the test above and can only TIGHTEN the gates: a factor above 1
    ends the run before anything is measured -- a line printed under a relaxed gate would look like any other line."""
    res = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0", "--workload", "cfg1"], capture_output=True,
                         text=True, timeout=600, env=dict(os.environ, ANEMOI_AMD_BENCH_PARITY_BOUND_SCALE="10"))
    assert res.returncode != 0 and "only factors in (0, 1]" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_bench_secondary_legs_never_raise():
    """``secondary`` (round 6: the Transformer processor @ config 3 with the mesh attention's roofline, config 2, config 5
    inside the default line): a leg that fails -- here every leg, the CPU container has no device -- reports its error under
    its own name; the block itself never raises, so it can never cost the headline."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU (the GPU boxes run test_bench_secondary_block_on_the_gpu)")
    sys.path.insert(0, ROOT)
    import bench

    os.environ["ANEMOI_AMD_BENCH_SECONDARY"] = "cfg2"
    try:
        out = bench.secondary_block(torch.device("cuda", 0), "bf16")
    finally:
        del os.environ["ANEMOI_AMD_BENCH_SECONDARY"]
    assert set(out) == {"cfg2", "note"} and "error" in out["cfg2"]
    assert [leg[0] for leg in bench.SECONDARY_LEGS] == ["transformer_cfg3", "cfg2", "cfg5_gnn"]


@pytest.mark.gpu
def test_bench_secondary_block_on_the_gpu(monkeypatch):
    """The two O96-sized legs of ``secondary`` as the default line runs them (config 2 and config 5; the Transformer leg at
    config 3 is the same function on the larger graph and rides in the driver's own bench run): step time, value and the
    roofline objects of each leg -- the fused Linear's MFMA fraction for both, the edge kernel's HBM fraction for config 2."""
    import torch

    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    monkeypatch.setenv("ANEMOI_AMD_BENCH_SECONDARY", "cfg2,cfg5_gnn")
    out = bench.secondary_block(torch.device("cuda", 0), "bf16")
    assert set(out) == {"cfg2", "cfg5_gnn", "note"}
    for name in ("cfg2", "cfg5_gnn"):
        leg = out[name]
        assert "error" not in leg, leg
        assert 0.5 < leg["ms_per_step"] < 100.0 and leg["value"] > 0 and leg["steps"] == 20
        assert 0.0 < leg["roofline"]["frac"] < 1.0 and leg["roofline"]["bound"] == "mfma"
    assert 0.0 < out["cfg2"]["roofline_edge"]["frac"] < 1.0 and out["cfg2"]["roofline_edge"]["bound"] == "hbm"
    json.dumps(out)  # serialisable as it is


def test_host_threads_follow_the_cgroup_cpu_quota(tmp_path, monkeypatch):
    """``bench.host_threads`` (the oracle's thread count, ``cpu_baseline.cores``): the affinity mask, cut by the cgroup CPU
    quota when there is one -- the GPU boxes of this pool show 256 logical CPUs under a quota of 16."""
    import builtins

    sys.path.insert(0, ROOT)
    import bench

    real_open = builtins.open

    def fake(content):
        def _open(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                f = tmp_path / "cpu.max"
                f.write_text(content)
                return real_open(f, *a, **k)
            return real_open(path, *a, **k)
        return _open

    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert bench.host_threads() == 16
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert bench.host_threads() == 256
    monkeypatch.setattr(builtins, "open", fake("50000 100000\n"))
    assert bench.host_threads() == 1
