"""CPU: the N>1 path (pair sharding + the single packed all-gather) with world_size 2 on gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mast3r_slam import dist as m3dist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = list(m3dist.shard_range(5, rank, world))
        g = torch.Generator().manual_seed(100 + rank)
        p = 3                                                    # equal per-rank batch (weak scaling)
        X = torch.randn(p, 4, 6, 3, generator=g)
        idx = torch.randint(0, 24, (p, 24), generator=g)
        valid = torch.rand(p, 24, 1, generator=g) > 0.5
        pose = torch.randn(p, 8, generator=g)
        out = m3dist.all_gather_results((X, idx, valid, pose))
        handle = m3dist.all_gather_results((X, idx, valid, pose), async_op=True)      # the overlapped form bench.py uses
        X.zero_()                                                                      # inputs may be overwritten at once
        out_async = handle.wait()
        ok = all(torch.equal(a, b) for a, b in zip(out, out_async))
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            Xr = torch.randn(p, 4, 6, 3, generator=gr); ir = torch.randint(0, 24, (p, 24), generator=gr)
            vr = torch.rand(p, 24, 1, generator=gr) > 0.5; pr = torch.randn(p, 8, generator=gr)
            sl = slice(r * p, (r + 1) * p)
            ok &= torch.equal(out[0][sl], Xr) and torch.equal(out[1][sl], ir)
            ok &= torch.equal(out[2][sl], vr) and torch.equal(out[3][sl], pr)
        ok &= out[1].dtype == torch.int64 and out[2].dtype == torch.bool and out[0].shape == (world * p, 4, 6, 3)
        # the per-step exchange of bench.py: buffers allocated once, index narrowed to int32 inside its copy, results as
        # strided views [world, ...] of ONE receive buffer; two steps through the same object
        X = torch.randn(p, 4, 6, 3, generator=torch.Generator().manual_seed(100 + rank))
        pg = m3dist.PackedGather((X, idx, valid, pose), dtypes=[torch.float32, torch.int32, torch.bool, torch.float32])
        base = pg.recv.data_ptr()
        for step in range(2):
            v = pg.post((X + step, idx, valid, pose)).wait()
            ok &= all(t.data_ptr() >= base and t.data_ptr() < base + pg.recv.numel() for t in v)       # views, not copies
            ok &= v[0].shape == (world, p, 4, 6, 3) and v[1].dtype == torch.int32 and pg.recv.data_ptr() == base
            for r in range(world):
                gr = torch.Generator().manual_seed(100 + r)
                Xr = torch.randn(p, 4, 6, 3, generator=gr); ir = torch.randint(0, 24, (p, 24), generator=gr)
                vr = torch.rand(p, 24, 1, generator=gr) > 0.5; pr = torch.randn(p, 8, generator=gr)
                ok &= torch.equal(v[0][r], Xr + step) and torch.equal(v[1][r], ir.to(torch.int32))
                ok &= torch.equal(v[2][r], vr) and torch.equal(v[3][r], pr)
        q.put((rank, ok, pairs))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_all_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == [0, 1, 2] and res[1][2] == [3, 4]       # block partition, remainder to low ranks


def test_shard_range_covers_everything_once():
    for total in (0, 1, 7, 64, 1524):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in m3dist.shard_range(total, r, world)]
            assert got == list(range(total))


def test_pack_unpack_roundtrip_single_rank():
    ts = (torch.randn(2, 5, 3), torch.arange(14, dtype=torch.int64).view(2, 7), torch.tensor([[True], [False]]),
          torch.randn(2, 8).to(torch.bfloat16))
    buf, meta = m3dist.pack(ts)
    assert buf.numel() % 16 == 0
    out = m3dist.unpack(buf[None], meta, 1)
    assert all(torch.equal(a, b) for a, b in zip(ts, out))
    assert all(o.data_ptr() >= buf.data_ptr() for o in out)                      # one rank: views of the buffer
    two = torch.stack([buf, buf])
    views = m3dist.unpack(two, meta, 2, fold=False)                              # rank axis kept: strided views, no copy
    assert all(v.shape == (2,) + tuple(t.shape) and torch.equal(v[1], t) for v, t in zip(views, ts))
    assert all(two.data_ptr() <= v.data_ptr() < two.data_ptr() + two.numel() for v in views)


def _worker_edges(rank, world, port, q):
    """Config-5 sharding on CPU: each rank evaluates the per-edge normal-equation blocks of ITS edges with the
    oracle (stand-in for the HIP block kernel), the blocks are all-gathered raggedly (7 edges over 2 ranks)."""
    import numpy as np
    from mast3r_slam import synthetic
    from oracle import gn_rays as OG
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Twc, Xs, Cs, ii, jj, idx, valid, Q = synthetic.gn_graph(4, 300, num_edges=7, seed=3)[:8]
        e = len(ii)
        t, qq, sc = Twc[:, :3].astype(np.float64), Twc[:, 3:7].astype(np.float64), Twc[:, 7].astype(np.float64)
        iu = np.triu_indices(7)

        def blocks(edges):
            rows = []
            for k in edges:
                Hjj, gj, n = OG.edge_blocks(t, qq, sc, Xs, Cs, int(ii[k]), int(jj[k]), idx[k], valid[k], Q[k])
                rows.append(np.concatenate([Hjj[iu], gj, [float(n)]]))
            return torch.from_numpy(np.asarray(rows, dtype=np.float64).reshape(len(rows), 36))
        got = m3dist.sharded_edge_blocks(blocks, e)
        full = blocks(range(e))
        q.put((rank, bool(torch.equal(got, full)) and got.shape == (e, 36) and got.dtype == torch.float64, e))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_edge_blocks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_edges, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res) and res[0][2] % 2 == 1      # an odd edge count: the shards are ragged


def _fake_match(model, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j):
    """Stand-in for mast3r_match_symmetric on CPU: deterministic per-edge results derived from the features, so that
    any rank computes the same thing for the same edge."""
    b, n = feat_i.shape[0], 12
    key = (feat_i[:, 0, 0] * 10 + feat_j[:, 0, 0]).long()                      # feat = keyframe id
    base = torch.arange(n)[None].repeat(b, 1)
    idx_i2j = (base + key[:, None]) % n
    idx_j2i = (base + 2 * key[:, None]) % n
    vj = ((base + key[:, None]) % 5 != 0)[..., None]
    vi = ((base + key[:, None]) % 7 != 0)[..., None]
    # edge (0,3) gets low descriptor confidence -> dropped by the match-fraction test (not consecutive)
    q = torch.where((feat_i[:, 0, 0] == 0) & (feat_j[:, 0, 0] == 3), 1.0, 3.0)[:, None, None].expand(b, n, 1)
    return idx_i2j, idx_j2i, vj, vi, q, q, q, q


def _graph_frames():
    from types import SimpleNamespace
    shape = torch.tensor([[16, 16]], dtype=torch.int32)
    return [SimpleNamespace(feat=torch.full((1, 1024), float(k)), pos=torch.zeros((1, 2), dtype=torch.long),
                            img_true_shape=shape, X_canon=torch.zeros(12, 3)) for k in range(6)]


def _edge_lists():
    ii = [0, 0, 1, 0, 1, 2, 2, 3, 3, 4]
    jj = [1, 2, 2, 3, 3, 3, 4, 4, 5, 5]
    return ii, jj


def _worker_factor_graph(rank, world, port, q):
    from types import SimpleNamespace
    from mast3r_slam.global_opt import FactorGraph
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ii, jj = _edge_lists()
        fg = FactorGraph(SimpleNamespace(device="cpu"), _graph_frames(), group=dist.group.WORLD, batch=2)
        ok = fg.add_factors(ii[:6], jj[:6], 0.1, _fake_match)
        ok &= fg.add_factors(ii[6:], jj[6:], 0.1, _fake_match)                 # a second call: ownership interleaves
        uniq = fg.get_unique_kf_idx()
        li, lj, lidx, lvalid, lQ, graph = fg._local_edges(uniq)
        q.put((rank, ok, fg.ii.tolist(), fg.jj.tolist(), fg.owner.tolist(), li.tolist(), lj.tolist(),
               lidx.tolist(), graph[0].tolist(), graph[1].tolist(), graph[2]))
    finally:
        dist.destroy_process_group()


def test_two_rank_edge_sharded_factor_graph():
    """FactorGraph(group=...) on CPU: each rank matches only its shard of every add_factors call, the keep decision is
    global, and the rank-ordered directed edge list every rank derives is the concatenation of the ranks' own lists."""
    from types import SimpleNamespace
    from mast3r_slam.global_opt import FactorGraph
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_factor_graph, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ii, jj = _edge_lists()
    ref = FactorGraph(SimpleNamespace(device="cpu"), _graph_frames())           # the same graph on one rank
    assert ref.add_factors(ii[:6], jj[:6], 0.1, _fake_match) and ref.add_factors(ii[6:], jj[6:], 0.1, _fake_match)
    assert (0, 3) not in zip(ref.ii.tolist(), ref.jj.tolist()) and ref.ii.numel() == 9       # the weak edge was dropped
    ri, rj, ridx, _, _, rgraph = ref._local_edges(ref.get_unique_kf_idx())
    assert rgraph is None
    for r in res:
        assert r[1] and r[2] == ref.ii.tolist() and r[3] == ref.jj.tolist()     # every rank knows the whole graph
        assert r[8] == res[0][5] + res[1][5] and r[9] == res[0][6] + res[1][6]  # rank-ordered global directed list
        assert r[10] == [len(res[0][5]), len(res[1][5])]
    assert res[0][4] == res[1][4] and set(res[0][4]) == {0, 1}
    # the union of the ranks' stored matches equals the single-rank graph's, edge by edge
    want = {(a, b): row for a, b, row in zip(ri.tolist(), rj.tolist(), ridx.tolist())}
    got = {}
    for r in res:
        got.update({(a, b): row for a, b, row in zip(r[5], r[6], r[7])})
    assert got == want
