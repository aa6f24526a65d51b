"""bench.py's N > 1 plumbing end to end on CPU (gloo, world size 2): self-launch of the ranks, rendezvous on
127.0.0.1, env sharding by global index (muzero.py:170-178: worker i gets seed config.seed + i), the weight
broadcast from rank 0 (trainer.py:87-95 <-> self_play.py:37), the max-over-ranks timing reduction and the relay
of rank 0's single JSON line.  `--rehearse-cpu` replaces the search by a sleep: nothing here is a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(cmd, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env or {})
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [line for line in proc.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {proc.stdout!r}"
    return json.loads(lines[0])


def _check(out, world):
    assert out["n_gpus"] == world and out["collective_backend"] == "gloo" and out["rccl_ranks"] == 0
    assert out["weights_identical_after_refresh"] is True and out["weight_refreshes"] == 3
    per = out["envs_per_rank"]
    assert out["rank_seeds"] == [list(range(r * per, (r + 1) * per)) for r in range(world)]
    assert out["ms_per_step"] >= 1.0


def test_bench_self_launches_two_ranks():
    out = _run([sys.executable, BENCH, "--gpus", "2", "--rehearse-cpu", "--steps", "12", "--bcast-every", "5"])
    _check(out, 2)
    assert out["self_launched"] is True


def test_bench_under_torch_distributed_run():
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29631", BENCH, "--gpus", "2", "--rehearse-cpu",
                "--steps", "12", "--bcast-every", "5"])
    _check(out, 2)
    assert out["self_launched"] is False


def test_bench_without_gpus_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""                       # also on a GPU box: no device for this check
    env["CUDA_VISIBLE_DEVICES"] = ""
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2"], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0 and "needs MI355X GPUs" in proc.stderr


def test_bench_launcher_ends_when_a_rank_dies_before_the_rendezvous():
    """A rank that dies at start leaves rank 0 waiting in the rendezvous with its stdout open: the launcher must not
    block on that pipe -- it notices the failed rank, ends the others and exits non-zero in bounded time."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MZ_BENCH_FAULT_RANK"] = "1"
    t0 = time.time()
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-cpu", "--steps", "4"], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=120)
    assert proc.returncode != 0
    assert time.time() - t0 < 60
    assert not [line for line in proc.stdout.splitlines() if line.startswith("{")]      # no result line from a failed job
