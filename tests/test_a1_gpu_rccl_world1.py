"""RCCL itself on the GPU box: a one-rank "nccl" process group runs the partitioned layer and network
(tests/rehearse_rccl_world1.py) and bench.py's N>1 code path in child processes.  The children are
started BEFORE this process touches the GPU (as in test_a0_gpu_trials.py: the file name puts these tests
right after it, and they skip themselves when the GPU is already initialised)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _env(port):
    if torch.cuda.device_count() == 0:
        pytest.skip("no GPU")
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: run this file first / alone")
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_partitioned_layer_and_network_over_rccl():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rehearse_rccl_world1.py")],
                       env=_env(29677), capture_output=True, text=True, timeout=280)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("ok", "FAILED"))]
    if r.returncode != 0 and not lines:
        # the process group never came up (no check ran): an environment matter, reported as a skip so
        # that `pytest -x` goes on to the parity tests
        pytest.skip("one-rank nccl group could not be created here: " + r.stderr[-400:])
    assert r.returncode == 0 and len(lines) == 15 and all(ln.startswith("ok") for ln in lines), \
        r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("exchange", ["auto", "h"])
def test_bench_partitioned_path_over_rccl(exchange):
    """bench.py's N>1 branch (gat_amd.dist.bench_partitioned) on a one-rank RCCL group: the JSON line the
    driver would read at N = 2, 4, 8 comes out, with both exchange modes in it."""
    env = _env(29678 if exchange == "auto" else 29680)      # a port of its own per child: no wait for a closed socket
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", GAT_FORCE_PARTITIONED="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "arxiv",
                        "--steps", "2", "--warmup", "1", "--exchange", exchange]
                       + (["--chunks", "3"] if exchange == "h" else []),
                       env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    if r.returncode != 0 and "init_process_group" in r.stderr and not r.stdout.strip():
        pytest.skip("one-rank nccl group could not be created here: " + r.stderr[-400:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "nccl" in line["config"]["parallelism"]
    assert line["roofline"]["frac"] > 0 and "traffic" in line["roofline"]
    # the N > 1 line says where its time went and what it moved
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    part = line["config"]["partition"]
    assert part["padded_nodes_over_n"] >= 1.0 and sum(part["nodes_per_rank"]) == line["config"]["nodes"]
    assert line["gathered_table"]["bytes"] > 0 and line["gathered_table"]["rows_per_node"] >= 1.0
    assert "allreduce_params" in line["comm_ms"] and line["comm_ms_total"] >= 0
    if exchange == "auto":
        assert "allgather_a_src" in line["comm_ms"]
        assert line["exchange_h"]["value"] > 0 and "allgather_h" in line["exchange_h"]["comm_ms"]
    else:
        assert line["config"]["parallelism"].count("pieces") == 1 and line["pieces"] == 3
        assert {"wait_gather_h", "wait_scatter_dh", "allreduce_params"} <= set(line["comm_ms"])
        assert line["exchange_h_one_piece"]["value"] > 0
        assert "reduce_scatter_dh" in line["exchange_h_one_piece"]["comm_ms"]
