"""`python bench.py --gpus N` (the way the driver calls it) must start its own N ranks, from a parent
that never touches the GPU, relay rank 0's single JSON line and fail when the launcher fails."""
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_launcher_command_is_torchrun_on_localhost():
    b = _bench()
    cmd = b.launcher_command(4, ["--gpus", "4", "--steps", "7"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")


class _FakePopen:
    def __init__(self, lines, rc):
        self.stdout, self.rc = io.StringIO("".join(lines)), rc

    def wait(self):
        return self.rc


def test_spawn_relays_one_json_line_and_the_exit_status(monkeypatch, capsys):
    b = _bench()
    seen = {}
    line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 2}) + "\n"

    def fake(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return _FakePopen(["RCCL version banner\n", line, "trailing noise\n"], 0)
    monkeypatch.setattr(b.subprocess, "Popen", fake)
    assert b.spawn_ranks(2, ["--gpus", "2"]) == 0
    out = capsys.readouterr()
    assert out.out == line                       # exactly the one line; everything else went to stderr
    assert "RCCL version banner" in out.err
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "--nproc-per-node=2" in seen["cmd"]
    monkeypatch.setattr(b.subprocess, "Popen", lambda *a, **k: _FakePopen([line], 5))
    assert b.spawn_ranks(2, []) == 5             # a failing rank fails the run
    monkeypatch.setattr(b.subprocess, "Popen", lambda *a, **k: _FakePopen([], 0))
    assert b.spawn_ranks(2, []) == 1             # no result line is a failure too


def test_parent_spawns_before_importing_torch():
    """The real thing on this CPU-only machine: the parent must reach the launcher without importing torch
    (so without any HIP call), the two ranks start, find no GPU and fail, and the parent reports that."""
    code = (
        "import sys, runpy\n"
        "sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0', '--backend', 'gloo', '--no-secondary']\n"
        "import subprocess\n"
        "real = subprocess.Popen\n"
        "def spy(*a, **k):\n"
        "    assert 'torch' not in sys.modules, 'parent imported torch before spawning'\n"
        "    print('SPAWNED', a[0][1:4], file=sys.stderr)\n"
        "    return real(*a, **k)\n"
        "subprocess.Popen = spy\n"
        "runpy.run_path(%r, run_name='__main__')\n" % os.path.join(ROOT, "bench.py"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert "SPAWNED ['-m', 'torch.distributed.run', '--nnodes=1']" in p.stderr
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0 and p.stdout.strip() == ""


import pytest


@pytest.mark.gpu
def test_two_rank_rehearsal_line_has_the_multi_gpu_keys():
    """`python bench.py --gpus 2 --backend gloo`: two ranks share the one GPU (RCCL needs one GPU per rank, so the
    in-library ncclAllReduce leg is skipped under gloo); the line must carry what a scaling run is judged by."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "C2",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["exchange"] == "torch/gloo"
    assert out["rccl_ranks"] == 0                      # gloo rehearsal: no RCCL communicator exists
    assert out["rank_ms_per_step"]["min"] <= out["rank_ms_per_step"]["max"] == pytest.approx(out["ms_per_step"])
    assert out["exchange_exposed_us"] is not None and out["exchange_exposed_us"] > 0
    assert out["perm_kind"] == "inline keyed bijection" and out["index_bytes"] == 0
    assert out["alg_bytes_per_transition"] == 584


@pytest.mark.gpu
def test_one_rank_rccl_line_runs_both_exchanges():
    """PPO_XP_FORCE_PG=1: the multi-rank control flow with ONE rank over RCCL - the headline under the caller-side
    all-reduce and TGT again under the library's own ncclAllReduce (`also.TGT_exchange_rccl`)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", PPO_XP_FORCE_PG="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1
    # both exchanges were timed on TGT; the line reports the faster one and keeps the other under "also"
    mine = out["config"]["exchange"]
    assert mine in ("torch/nccl", "rccl/nccl")
    other_key = "TGT_exchange_rccl" if mine == "torch/nccl" else "TGT_exchange_torch"
    other = out["also"][other_key]
    assert other["rccl_ranks"] == 1 and other["config"]["exchange"] == ("rccl/nccl" if mine == "torch/nccl" else "torch/nccl")
    assert other["exchange_exposed_us"] is not None and out["exchange_exposed_us"] is not None
    assert out["ms_per_step"] <= other["ms_per_step"]
    assert "roofline" in out and out["roofline"]["frac"] > 0.5
