"""bench.py typed as `python bench.py --gpus N` must start its own ranks and hand back THEIR exit code (the
driver's scaling run types exactly that).  Without a HIP device the ranks fail their device assertion: the
launcher must relay that failure (non-zero exit code, the ranks' message on stderr) instead of hanging or
reporting success."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU form of this test is in test_gpu_model.py")
def test_self_launch_relays_the_ranks_exit_code():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--same-device", "--steps", "1", "--warmup", "0", "--shapes", "1", "--no-roofline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert "needs a HIP device" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_launcher_command_is_the_contract_form(monkeypatch):
    """The child command is `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the same flags>`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gv_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class FakeProc:
        stdout = iter(['{"ok": 1}\n', "noise\n"])

        def wait(self):
            return 7

    def fake_popen(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        return FakeProc()

    monkeypatch.setattr(bench.subprocess, "Popen", fake_popen)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])

    class A:
        gpus = 4
    assert bench.launch_ranks(A()) == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
