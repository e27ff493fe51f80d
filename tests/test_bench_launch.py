"""bench.py's N-rank launch path (round-2 verdict, missing #1): `python bench.py --gpus N` -- the form the driver uses --
must start N ranks by itself (the reference starts one process per GPU: run/train_egoclip.py:39-45,128-134), and must
never print a line that claims N GPUs from fewer.  Runs here without a GPU: `--launch-dry-run` goes through the same
self re-exec under torch.distributed.run and the same rendezvous, then stops after one gloo all-reduce."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _json_lines(out):
    res = []
    for line in out.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            res.append(json.loads(line))
    return res


def test_bare_invocation_launches_n_ranks():
    r = _run(["--gpus", "2", "--launch-dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout           # ONE line, from rank 0
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_joined"] == 2 and lines[0]["world_size"] == 2 and lines[0]["ok"]


def test_three_ranks_join():
    r = _run(["--gpus", "3", "--launch-dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]["ranks_joined"] == 3


def test_refuses_more_gpus_than_visible():
    """No GPU in this container: `--gpus 2` must exit non-zero BEFORE launching anything and print no result line (on the
    1-GPU gpurun box the same check refuses --gpus 2; tests/test_gpu_dist.py)."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env_extra={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0
    assert "only 0 HIP device" in r.stderr, r.stderr[-2000:]
    assert not _json_lines(r.stdout)


def test_world_size_mismatch_is_an_error():
    """Launched by an external torchrun with a different rank count than --gpus: refuse, do not relabel."""
    r = _run(["--gpus", "4", "--launch-dry-run"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                             "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29777"})
    assert r.returncode != 0
    assert "must agree" in (r.stderr + r.stdout)


def test_external_launcher_form_still_works():
    """The documented driver form for N > 1: python -m torch.distributed.run ... bench.py --gpus N."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29778", BENCH, "--gpus", "2", "--launch-dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2
