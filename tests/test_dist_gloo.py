"""CPU-only, world_size 2 over gloo: the multi-GPU layer (sharding + the single final gather)."""
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from instrain_amd import dist as idist
    from instrain_amd._lib import SNV_DT, LD_DT
    rank, local, world = idist.init_from_env(backend="gloo")
    assert world == 2
    # each rank owns the scaffolds LPT gives it and "profiles" them (fake rows tagged by scaffold id)
    costs = [5, 1, 4, 2, 3]
    mine = idist.lpt_shards(costs, world)[rank]
    snv = np.zeros(sum(costs[i] for i in mine), dtype=SNV_DT)
    k = 0
    for i in mine:
        snv["gpos"][k:k + costs[i]] = 1000 * i + np.arange(costs[i])
        k += costs[i]
    ld = np.zeros(rank, dtype=LD_DT)          # ragged: rank 0 contributes an EMPTY table
    ld["gpos_a"] = 7
    out = idist.gather_tables({"snv": snv, "ld": ld}, dst=0)
    if rank == 0:
        got = np.sort(out["snv"]["gpos"])
        exp = np.sort(np.concatenate([1000 * i + np.arange(c) for i, c in enumerate(costs)]))
        assert (got == exp).all(), (got, exp)
        assert len(out["ld"]) == 1 and out["ld"]["gpos_a"][0] == 7
        print("GATHER_OK")
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()
""") % REPO


def test_gather_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER_OK" in r.stdout
