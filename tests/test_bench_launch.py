"""`python bench.py --gpus N` starts its own ranks (one process per GPU — the reference's unit is one process per contig,
/root/reference README.md:73-76) without the parent touching a GPU: checked here on the CPU by intercepting the child
command."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_n_launches_one_rank_per_gpu(monkeypatch):
    import subprocess

    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    # a torch import in the parent would be harmless, a torch.cuda call would not: main() must leave before either
    monkeypatch.setitem(sys.modules, "torch", None)
    try:
        b.main()
        code = None
    except SystemExit as e:
        code = e.code
    assert code == 7  # the child's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_under_a_launcher_bench_does_not_launch_again(monkeypatch):
    import subprocess

    b = _bench()
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: (_ for _ in ()).throw(AssertionError("launched again")))
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setitem(sys.modules, "torch", None)  # the next statement after the launch decision is `import torch`
    try:
        b.main()
        raise AssertionError("expected the torch import to be reached")
    except ImportError:
        pass
