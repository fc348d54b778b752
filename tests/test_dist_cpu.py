"""N>1 path of bench.py on CPU: world_size-2 gloo run of the contig sharding + max-over-ranks timing logic."""
import os
import subprocess
import sys
import textwrap

from helpers import ROOT


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import bench
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        # every rank owns its own, disjoint, deterministic set of contigs (weak scaling, no data-path collective)
        seqs = bench.synth_contigs(3, 1000, 12345 + 1000 * rank)
        digest = torch.tensor([float(sum(s[:50]))  for s in seqs], dtype=torch.float64)
        allv = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(allv, digest)
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"max": t.item(), "distinct": len({tuple(v.tolist()) for v in allv})}))
        dist.barrier(); dist.destroy_process_group()
    """ % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    import json
    out = json.loads(line)
    assert out["max"] == 2.0 and out["distinct"] == 2
