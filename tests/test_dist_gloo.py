"""The N>1 path on CPU: world_size-2 gloo run of the sharding + timing-combination logic bench.py uses."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from .conftest import REPO


def test_shard_ranges_cover_without_overlap():
    from longreadselfcorrect_amd.dist import shard_range

    for n in (0, 1, 7, 100, 100001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(r, world, n) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_run(tmp_path):
    """Two processes (gloo): each generates only its own shard of reads; shards are disjoint and equal the
    single-process generation; (max time, sum bases) combine across ranks; no data-path collective is needed."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f'''
        import os, sys, json
        sys.path.insert(0, {str(REPO)!r})
        import numpy as np, torch, torch.distributed as dist
        from longreadselfcorrect_amd import Lrsc, dist as lrdist
        rank, world, _ = lrdist.env_rank()
        dist.init_process_group(backend="gloo")
        api = Lrsc()
        g = api.synth_genome(11, 30000)
        n = 25
        bases, off = api.synth_reads(12, g, n, 800, first_read=lrdist.weak_shard_first_read(rank, n))
        lrdist.barrier()
        t, b = lrdist.combine(1.0 + rank, float(off[-1]))
        first, last = lrdist.shard_range(rank, world, 101)
        json.dump({{"rank": rank, "world": world, "bases": int(off[-1]), "t": t, "b": b, "first": first, "last": last,
                   "crc": int(np.frombuffer(bases.tobytes(), dtype=np.uint8).astype(np.uint64).sum())}},
                  open(os.environ["OUT"] + f"/r{{rank}}.json", "w"))
        dist.destroy_process_group()
    '''))
    env = dict(os.environ, OUT=str(tmp_path), MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29513", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    res = [json.load(open(tmp_path / f"r{k}.json")) for k in range(2)]
    assert [x["world"] for x in res] == [2, 2]
    assert res[0]["t"] == res[1]["t"] == 2.0                          # max over ranks
    assert res[0]["b"] == res[1]["b"] == res[0]["bases"] + res[1]["bases"]   # sum over ranks
    assert (res[0]["first"], res[0]["last"], res[1]["first"], res[1]["last"]) == (0, 51, 51, 101)
    # the two shards together are exactly what one process generates for 2n reads
    from longreadselfcorrect_amd import Lrsc
    api = Lrsc()
    g = api.synth_genome(11, 30000)
    bases, off = api.synth_reads(12, g, 50, 800)
    assert int(off[25]) == res[0]["bases"] and int(off[50] - off[25]) == res[1]["bases"]
    assert int(bases[: int(off[25])].astype(np.uint64).sum()) == res[0]["crc"]
    assert int(bases[int(off[25]):].astype(np.uint64).sum()) == res[1]["crc"]
