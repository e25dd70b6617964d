"""bench.py host logic that needs no GPU: argument defaults (>= 100 timed / >= 20 warm-up steps) and the
self-launch of N ranks when `--gpus N` is given without a launcher."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_follow_the_measurement_plan():
    a = bench.parse([])
    assert a.gpus == 1 and a.steps >= 100 and a.warmup >= 20 and a.batch == 256 and a.z_dim == 32
    assert a.global_batch == 1024 and a.backend == "nccl"
    assert bench.parse(["--per-gpu-batch", "128"]).batch == 128 and bench.parse(["--batch", "64"]).batch == 64


def test_gpus_n_without_launcher_starts_n_ranks(monkeypatch):
    seen = {}

    class Res:
        returncode = 0
        stdout = 'rank noise\n{"metric": "x", "n_gpus": 4}\n'

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return Res()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and os.path.basename(cmd[-5]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
