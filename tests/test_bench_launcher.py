"""bench.py's multi-rank entry point (reference run.py:50-59, train.py:39-48): `python bench.py --gpus N` must start N
ranks itself, the line must carry the process group's world size, and a failing rank must fail the run.  Exercised without a GPU through
tests/bench_dry_run.py (bench.main with a CPU rank body: gloo, CPU tensors, the C-ABI emulator on a 64x64 miniature of the wiring)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "tests", "bench_dry_run.py")      # bench.py's launcher with a CPU rank body (no GPU here)


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def _last_json(stdout: str):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bare_command_spawns_its_ranks_and_reports_the_group_size():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["dry"] is True and line["value"] is None
    assert line["config"]["global_batch"] == 2 and line["config"]["parallelism"] == "dp2"
    assert line["config"]["replicas_equal_after_steps"] is True        # the exchange ran: both ranks hold the same weights after Adam
    assert line["config"]["bn_buffers_equal_after_sync"] is True       # train.sync_bn_buffers (the explicit collective every rank issues before rank 0 saves)


def test_driver_command_form_through_torch_distributed_run():
    port = str(36500 + os.getpid() % 2000)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", port, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 2


def test_world_size_mismatch_and_failing_rank_are_fatal():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="1", RANK="0"), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--fail-rank", "1"], env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
