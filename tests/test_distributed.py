"""Multi-rank path: world_size 2 under gloo (CPU here, cuda:0 shared on the GPU box), and the same sweeps in a
group of one rank under RCCL with every collective issued."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(mode, tmp_path, port, nproc=2):
    out = tmp_path / "result.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(out)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert out.read_text() == "OK"


def test_sharded_sums_world2_cpu(tmp_path):
    _launch("cpu", tmp_path, 29531)


def test_sharded_engine_world2_cpu(tmp_path):
    _launch("cpu_engine", tmp_path, 29533)


@pytest.mark.gpu
def test_sharded_sweeps_world2_gpu(tmp_path):
    _launch("gpu", tmp_path, 29532)


@pytest.mark.gpu
def test_rccl_call_path_one_rank_gpu(tmp_path):
    _launch("rccl", tmp_path, 29534, nproc=1)
