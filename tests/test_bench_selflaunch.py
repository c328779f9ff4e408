"""``python bench.py --gpus N`` must start N ranks by itself (the driver starts the N=1 run as a plain
``python3 bench.py --gpus 1``; the same form with N > 1 used to die on an assert before measuring anything).  Run here on the
CPU with the stub workload over gloo: launcher, rendezvous on 127.0.0.1, max-over-ranks timing, rank-0-only JSON."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"HRV_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"})
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def test_gpus2_without_a_launcher_self_launches_two_ranks_and_prints_one_json_line():
    r = _run(["--gpus", "2", "--workload", "stub", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["rccl_ranks"] == 0 and j["config"]["world_size"] == 2 and j["config"]["dist_backend"] == "gloo" and j["config"]["global_batch"] == 4
    assert j["config"]["parallelism"] == "dp2-allreduce"
    # whole-job throughput: images of both ranks / max-over-ranks time
    assert abs(j["value"] - j["config"]["global_batch"] * 1e3 / j["ms_per_step"]) < 0.02 * j["value"]


def test_gpus1_runs_in_process_without_a_process_group():
    r = _run(["--gpus", "1", "--workload", "stub", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["config"]["rccl_ranks"] == 0 and j["config"]["world_size"] == 1 and j["config"]["dist_backend"].startswith("none")


def test_launched_by_torch_distributed_run_it_does_not_relaunch():
    # the driver's N > 1 form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"HRV_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload",
                        "stub", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
