"""The N > 1 code path on REAL RCCL with one rank: an 8-GPU node is not ours to launch, but
init_process_group("nccl", device_id=...), all_gather_into_tensor / all_reduce on device tensors and the overlapped
exchange of gvcnn-tf_amd/sharding.py can run on one MI355X.  The work happens in a fresh child process (started as a
subprocess: never a re-exec of a process that has touched the GPU); this test asserts on its report."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_sharded_engines_on_a_one_rank_rccl_group_match_the_unsharded_ones():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_child.py")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL1 ")]
    assert line, r.stdout[-2000:]
    rep = json.loads(line[-1][6:])
    print(json.dumps(rep))
    assert rep["backend"] == "nccl" and rep["world"] == 1 and rep["forced"]
    assert rep["rccl_version"] and rep["rccl_version"][0].isdigit()          # the line carries the RCCL version
    assert rep["bare_all_gather_ok"] and rep["bare_all_reduce_ok"]
    # inference: the same data through the same kernels, plus collectives that move bytes unchanged — bit for bit
    for k in ("infer_collective_bitwise", "infer_direct_bitwise", "infer_overlap_bitwise", "infer_scores_exchange_bitwise"):
        assert rep[k], k
    # training: the sharded step takes other launches in places (BatchNorm sums leave the producing launch to be
    # all-reduced; the moving averages are updated from the gathered statistics): same scheme, same loss to fp32
    # rounding, gradients within 1e-4 of the largest gradient — bitwise where the launches coincide (reported)
    for mode in ("shapes", "views"):
        t = rep["train_" + mode]
        assert t["scheme_equal"], (mode, t)
        assert abs(t["loss_plain"] - t["loss_sharded"]) <= 1e-5 * max(1.0, abs(t["loss_plain"])), (mode, t)
        assert t["grad_worst_rel"] < 1e-4, (mode, t)
