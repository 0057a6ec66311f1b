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
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, WORLD_SIZE="1", RANK="0"))
    assert res.returncode != 0 and "torch.distributed.run" in (res.stderr + res.stdout)


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
    pv = line["parity_vs_single"]
    assert pv["finite"] and pv["ranks_checked"] == world and pv["rows_checked"] > 0
    assert pv["max_rel_err"] <= pv["bound"], pv
    ranks = line["ms_per_step_ranks"]
    assert len(ranks["all"]) == world and ranks["min"] <= ranks["max"]
    assert abs(ranks["max"] - line["ms_per_step"]) < 1e-2 * line["ms_per_step"] + 1e-3  # value = MAX over ranks
    assert line["halo"]["own_mesh_rows"] > 0 and "DEBUG" in line["config"]["parallelism"]
