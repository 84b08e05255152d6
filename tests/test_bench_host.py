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


def test_in_step_table_reads_the_newest_committed_kernel_stats(tmp_path, monkeypatch):
    """bench.in_step_table: the `in_step` object of the bench line is the committed rocprofv3 summary of the same command —
    calls per step from the optimizer launch's call count, kernels that do not run every step left out."""
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r01a_dcn_kernel_stats.md").write_text("| `old_kernel(x)` | 10 | 10 | 1 | 1 | 1 | 1 |\n")
    (prof / "r02b_dcn_kernel_stats.md").write_text(
        "source: x\n\n| kernel | calls | total_ns | avg_ns | min_ns | max_ns | % |\n|---|---:|---:|---:|---:|---:|---:|\n"
        "| `void (anonymous namespace)::dense_bwd_kernel<true, true>((anonymous namespace)::DgradArgs)` | 300 | 7500000 | 25000 | 1 | 2 | 50 |\n"
        "| `(anonymous namespace)::adam_tf1_step_kernel((anonymous namespace)::AdamStepArgs)` | 100 | 600000 | 6000 | 1 | 2 | 10 |\n"
        "| `void at::native::reduce_kernel<512>(int)` | 1 | 260000 | 260000 | 1 | 2 | 1 |\n"
        "\nper launch shape (kernel, grid size):\n\n| `(anonymous namespace)::adam_tf1_step_kernel(x)` | 512 | 100 | 6000 | 1 | 2 |\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    t = bench.in_step_table("dcn")
    assert t["source"] == "profiles/r02b_dcn_kernel_stats.md" and t["steps_profiled"] == 100
    assert [(k["kernel"], k["calls_per_step"], k["avg_us"]) for k in t["kernels"]] == [
        ("dense_bwd_kernel<true, true>", 3.0, 25.0), ("adam_tf1_step_kernel", 1.0, 6.0)]
    assert t["dispatches_per_step"] == 4.0 and t["kernel_us_per_step"] == 81.0
    assert bench.in_step_table("din") is None


def test_host_fed_leg_is_bounded_and_optional(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--no-host-fed"])
    assert bench.parse_args().no_host_fed
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    assert not bench.parse_args().no_host_fed
