"""bench.py --gpus N without a launcher starts its own ranks (torch.distributed.run as a child process)
before anything touches the GPU.  On this CPU-only box the ranks must get as far as torch.cuda."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_command_plumbing():
    sys.path.insert(0, ROOT)
    import bench

    args = argparse.Namespace(gpus=4)
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2", "--rows", "1000"]
    cmd = bench.launcher_command(args, argv, 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv


def test_gpus2_without_launcher_reaches_torch_cuda():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "200000",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--settle-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    if r.returncode == 0:  # a box with two GPUs: the line must be there
        assert '"n_gpus": 2' in r.stdout
        return
    assert "needs torch.distributed.run" not in r.stderr
    low = r.stderr.lower()
    assert "cuda" in low or "hip" in low or "gpu" in low, r.stderr[-2000:]
