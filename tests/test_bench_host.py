"""Host logic of bench.py that needs no GPU: the configs[4] flag and the self-launch of `--gpus N`."""
import os
import sys

import pytest

import bench


def test_config5_is_deepfm_with_one_100m_row_table():
    a = bench.parse_args(["--config5", "--gpus", "8"])
    assert a.model == "deepfm" and a.big_table_rows == 100_000_000 and a.batch == 4096 and a.gpus == 8
    a = bench.parse_args(["--config5", "--big-table-rows", "5000000"])
    assert a.big_table_rows == 5_000_000


def test_self_launch_refuses_more_gpus_than_present(monkeypatch):
    monkeypatch.delenv("RECALGO_DIST_BACKEND", raising=False)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(os, "execvpe", lambda *a, **k: pytest.fail("must not launch"))
    with pytest.raises(SystemExit) as e:
        bench.self_launch(bench.parse_args(["--gpus", "8"]))
    assert "refusing" in str(e.value)


def test_self_launch_command(monkeypatch):
    """One rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1, the user's own flags passed through."""
    seen = {}
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(os, "execvpe", lambda exe, cmd, env: seen.update(exe=exe, cmd=cmd, env=env))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--config5"])
    bench.self_launch(bench.parse_args(["--gpus", "4", "--steps", "7", "--config5"]))
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--config5"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
