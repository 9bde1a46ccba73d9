"""Two (and three) ranks, one process each, one MI355X: the partitioned layer / network (tests/rehearse_gloo2_gpu.py)
and bench.py's N = 2 command line with gloo carrying the collectives between the processes (RCCL cannot
hold two ranks on one card; RCCL's own call forms are covered at world size 1 by test_a1_*).  The
children start BEFORE this process touches the GPU (file name: right after test_a0 / test_a1).  A child
that cannot set up its process group or runs out of time is an environment matter and reported as a
skip; a child that ran its checks and found a mismatch fails the test."""
import json
import os
import signal
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _launch(port, script_args, timeout, extra_env=None, world=2):
    if torch.cuda.device_count() == 0:
        pytest.skip("no GPU")
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: run this file first / alone")
    env = dict(os.environ)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    # a session of its own: on a timeout the launcher AND its rank processes are ended (by the group
    # id of the process started here), none is left behind on the GPU
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        pytest.skip(f"{world}-rank gloo child did not finish in {timeout} s: {out[-300:]}")
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_on_one_gpu_match_the_single_gpu_layer_and_the_oracle_network(world):
    r = _launch(29689 + 2 * world, [os.path.join(ROOT, "tests", "rehearse_gloo2_gpu.py")], 280, world=world)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("ok", "FAILED"))]
    if r.returncode != 0 and not lines:
        pytest.skip("two-rank gloo group could not be created here: " + r.stderr[-400:])
    assert r.returncode == 0 and len(lines) == 13 and all(ln.startswith("ok") for ln in lines), \
        r.stdout[-3000:] + r.stderr[-2000:]


def test_bench_line_at_two_ranks():
    """The driver's N = 2 command (python -m torch.distributed.run ... bench.py --gpus 2) with the backend
    switched to gloo: the graph really is split over two processes, both exchange patterns are timed."""
    r = _launch(29693, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "arxiv", "--steps", "2",
                        "--warmup", "1", "--chunks", "3"], 280, {"GAT_DIST_BACKEND": "gloo"})
    if r.returncode != 0 and "init_process_group" in r.stderr and not r.stdout.strip():
        pytest.skip("two-rank gloo group could not be created here: " + r.stderr[-400:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "gloo" in line["config"]["parallelism"]
    part = line["config"]["partition"]
    assert len(part["nodes_per_rank"]) == 2 and min(part["nodes_per_rank"]) > 0
    assert sum(part["nodes_per_rank"]) == line["config"]["nodes"]
    assert sum(part["edges_per_rank"]) == line["config"]["edges"]
    assert max(part["edges_per_rank"]) <= 1.1 * line["config"]["edges"] / 2        # balanced by in-edges
    assert line["exchange_h"]["pieces"] == 3 and line["exchange_h"]["value"] > 0
    assert {"wait_gather_h", "wait_scatter_dh", "allreduce_params"} <= set(line["exchange_h"]["comm_ms"])
    assert line["exchange_h"]["gathered_table"]["received_bytes_per_rank_fwd"] > 0
    assert line["exchange_h_one_piece"]["value"] > 0
    assert line["cpu_baseline"]["value"] > 0
