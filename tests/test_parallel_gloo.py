"""CPU, world_size 2, 4 and 8 (gloo): the one-process-per-rank driver — CYCLIC rank ownership of the half-space rows, z-plane
ownership, the block-store chunk layout and the exchange between the Z and XY stages, in one piece and pipelined in plane
groups through the two-slot ring — reproduces the single-process oracle result, for R = 1 and with z-residue streaming."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, WMAP


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, R, outdir, group_bytes=None, groups=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import zdo
    from standin_engine import NumpyEngine
    from zeldovich_plt_amd.parallel import SlabPipeline, split_ranks
    pk = zdo.pk_from_file(WMAP, 720.0)
    cube = zdo.mode_cube(zdo.make_params(n, numblock=2), pk)
    # `groups` independent groups of ranks: group j runs the residue passes j, j + groups, ...; the rows / planes of a pass
    # are sharded over the ranks of a group only
    grp_id, grank, gsz = split_ranks(rank, world, groups)
    pgs = [dist.new_group(list(range(j * gsz, (j + 1) * gsz))) for j in range(groups)] if groups > 1 else [None]
    eng = NumpyEngine(cube, n, R, grank, gsz)
    kw = dict(group_bytes=group_bytes) if group_bytes else {}
    pipe = SlabPipeline(eng, n, world=gsz, dist=dist, device="cpu", chunk_bytes=3 * n * n * eng.record_size, rank_base=grp_id * gsz,
                        process_group=pgs[grp_id], **kw)
    if group_bytes:
        assert pipe.group_planes < eng.Zq  # several plane groups per pass: the pipelined form of the exchange
    got = {}

    def consume(zs, ring):
        v = ring.numpy().view(np.complex128)
        per = n * eng.na * n
        for i, z in enumerate(zs):
            got[int(z)] = v[i * per:(i + 1) * per].reshape(n, eng.na, n).copy()

    pipe.run(consume, pass_first=grp_id, pass_step=groups)
    np.save(os.path.join(outdir, "rank%d.npy" % rank), got, allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("R,world,group_bytes", [(1, 2, None), (2, 2, None), (1, 2, 3 * 16 * 16 * 32 * 2), (1, 4, 2 * 8192), (1, 8, None)])
def test_multi_rank_pipeline_matches_oracle(tmp_path, oracle, R, world, group_bytes):
    n = 16
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, R, str(tmp_path), group_bytes), nprocs=world, join=True)
    pk = oracle.pk_from_file(WMAP, 720.0)
    ref = oracle.run(oracle.make_params(n, numblock=2), pk, want_planes=True)["planes"]  # [z][a][y][x]
    seen = {}
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r), allow_pickle=True).item()
        # ownership: rank r finishes planes residue + R*(r*Zq + i)
        Zq = n // R // world
        expect = sorted(res + R * (r * Zq + i) for res in range(R) for i in range(Zq))
        assert sorted(d.keys()) == expect
        seen.update(d)
    assert sorted(seen.keys()) == list(range(n))
    scale = np.abs(ref).max()
    for z, plane in seen.items():  # plane: [y][a][x]
        assert np.abs(plane.transpose(1, 0, 2) - ref[z]).max() / scale < 1e-13


@pytest.mark.parametrize("R,world,groups", [(2, 2, 2), (4, 4, 2), (4, 4, 4), (2, 4, 2)])
def test_pass_groups_match_oracle(tmp_path, oracle, R, world, groups):
    """the ranks as `groups` independent groups: group j runs the residue passes j, j + groups, ... (nothing travels between
    groups; one rank per group: no exchange at all), inside a group the rows / planes are sharded with the exchange"""
    n = 16
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, R, str(tmp_path), None, groups), nprocs=world, join=True)
    pk = oracle.pk_from_file(WMAP, 720.0)
    ref = oracle.run(oracle.make_params(n, numblock=2), pk, want_planes=True)["planes"]
    seen = {}
    gsz = world // groups
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r), allow_pickle=True).item()
        j, gr = r // gsz, r % gsz
        Zq = n // R // gsz
        expect = sorted(res + R * (gr * Zq + i) for res in range(j, R, groups) for i in range(Zq))
        assert sorted(d.keys()) == expect
        assert not (set(d) & set(seen))
        seen.update(d)
    assert sorted(seen.keys()) == list(range(n))
    scale = np.abs(ref).max()
    for z, plane in seen.items():
        assert np.abs(plane.transpose(1, 0, 2) - ref[z]).max() / scale < 1e-13
