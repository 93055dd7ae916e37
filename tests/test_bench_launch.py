"""`python bench.py --gpus N` is the command the driver's SCALE run is shaped like (distribute.py:16-18: one replica per GPU).
CPU: the launcher starts N ranks under torch.distributed.run as a child process, forwards its status, never re-execs; a rank
whose WORLD_SIZE disagrees with --gpus refuses to run.  GPU (one-device functional mode, gloo): the N = 2 line is real."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_gpus_flag_launches_ranks_as_a_child(monkeypatch):
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return Done()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the child's status is the launcher's status
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]      # same arguments for every rank
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_rank_refuses_a_world_size_that_is_not_gpus(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


@pytest.mark.gpu
def test_bench_gpus_2_one_device_functional_line(cuda):
    env = dict(os.environ, RNET_BENCH_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-infer", "--no-cpu-baseline", "--no-probe", "--no-exclusive"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                   # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64
    assert line["config"]["rccl_ranks"] == 2 and line["config"]["backend"] == "gloo"     # one-device mode: not RCCL
    assert line["config"]["comm"] == "torch"
    assert 0 < line["config"]["syncbn_messages"] <= 140
    assert line["value"] > 0 and line["scaling"] == "weak"
