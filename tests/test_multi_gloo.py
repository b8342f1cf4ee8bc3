"""world_size-2 (and 3) gloo test of the multi-GPU exchange protocol on CPU: rank bookkeeping, the
rank-major padded all-gather layout and the scatter back into the full frame.  The per-rank pixels are
rendered by the CPU checker with the same partition; the assembled frame must equal the checker's
unpartitioned render bit for bit (seeds depend on pixel index and subframe only)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class NumpyPacker:
    """Test double with the DevicePacker interface: same pixel lists, numpy indexing instead of the HIP kernels."""

    def __init__(self, frame_by_kind, lists, rank, width):
        import torch

        self.torch, self.frames, self.lists, self.rank, self.width = torch, frame_by_kind, lists, rank, width
        self.padded = max(len(l) for l in lists)

    def owned_padded(self):
        return len(self.lists[self.rank]), self.padded

    def alloc(self, n, which):
        f = self.frames[which]
        return self.torch.zeros((n,) + f.shape[2:], dtype=self.torch.from_numpy(f[:1, :1]).dtype)

    def pack(self, which, dst):
        px = self.lists[self.rank]
        v = self.frames[which][(px >> 16).astype(np.int64), (px & 0xFFFF).astype(np.int64)]
        dst[: len(px)] = self.torch.from_numpy(np.ascontiguousarray(v))

    def unpack(self, which, src, into=None):
        a = src.numpy()
        dst = self.frames[which] if into is None else into
        for r, px in enumerate(self.lists):
            dst[(px >> 16).astype(np.int64), (px & 0xFFFF).astype(np.int64)] = a[r * self.padded : r * self.padded + len(px)]

    # the overlapped hand-off's entry points (synchronous here): display buffers are separate from the rendered ones
    def pack_async(self, which, dst, slot):
        self.pack(which, dst)

    def pack_wait(self, slot):
        pass

    def unpack_display(self, which, src):
        if not hasattr(self, "display"):
            self.display = {}
        if which not in self.display:
            self.display[which] = np.zeros_like(self.frames[which])
        self.unpack(which, src, self.display[which])


def _worker(rank, world, port, w, h, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from oracle import orc
    from optixpathtracer_amd import multigpu, scenes

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O = orc.Oracle("det")
        m = scenes.cornell_box()
        probe = scenes.constant_probe().BuildCDF()
        U, V, W = scenes.uvw_frame(**scenes.CORNELL_CAMERA, aspect=w / h)
        full = O.render(O.make_scene(m), O.make_probe(probe), (U, V, W), scenes.CORNELL_CAMERA["eye"], w, h, 2, nthreads=2)
        lists = multigpu.pixel_lists(w, h, world, 16, 8)
        assert sum(len(l) for l in lists) == w * h and len(np.unique(np.concatenate(lists))) == w * h
        # this rank "renders" only its own pixels
        mine = np.zeros((h, w), bool)
        px = lists[rank]
        mine[(px >> 16).astype(np.int64), (px & 0xFFFF).astype(np.int64)] = True
        accum = np.where(mine[..., None], full["accum"], np.float32(-7.0)).astype(np.float32)
        frame = np.where(mine, full["frame"], np.uint32(0)).astype(np.int32)
        frames = {0: accum, 1: frame}
        packer = NumpyPacker(frames, lists, rank, w)
        multigpu.exchange_frame(packer, 0, world, dist.all_gather_into_tensor)
        multigpu.exchange_frame(packer, 1, world, dist.all_gather_into_tensor)
        ok = np.array_equal(frames[0].view(np.uint32), full["accum"].view(np.uint32)) and np.array_equal(frames[1].view(np.uint32), full["frame"])
        # the overlapped hand-off (multigpu.HandOff): frame k is submitted, frame k-1 collected into the DISPLAY buffer while "frame k renders";
        # the rendered buffer keeps changing underneath (here: every frame adds k to this rank's own pixels)
        own = frame.copy()
        packer2 = NumpyPacker({1: own}, lists, rank, w)
        h_off = multigpu.HandOff(packer2, 1, world, dist.all_gather_into_tensor)
        want_prev = None
        for k in range(4):
            own[mine] = (full["frame"].astype(np.int64)[mine] + k).astype(np.int32)  # "render(k)" overwrites this rank's pixels
            if h_off.collect():
                ok = ok and np.array_equal(packer2.display[1], want_prev)
            h_off.submit()
            want_prev = (full["frame"].astype(np.int64) + k).astype(np.int32)
        ok = ok and h_off.collect() and np.array_equal(packer2.display[1], want_prev) and not h_off.collect()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, (72, 40)), (3, (50, 33))])
def test_exchange_protocol_gloo(world, size):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, size[0], size[1], q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(r, True) for r in range(world)]


def test_pixel_lists_cover_and_order():
    sys.path.insert(0, ROOT)
    from optixpathtracer_amd import multigpu

    for (w, h, world) in ((64, 32, 1), (70, 33, 4), (1920, 1080, 8)):
        lists = multigpu.pixel_lists(w, h, world, 64, 16)
        allp = np.concatenate(lists)
        assert len(allp) == w * h == len(np.unique(allp))
        if world == 8:  # interleaving balances the load
            sizes = np.array([len(l) for l in lists])
            assert sizes.max() / sizes.min() < 1.05
    l = multigpu.pixel_lists(16, 16, 1)[0]
    assert (l[:8] & 0xFFFF).tolist() == list(range(8)) and (l[:64] >> 16).max() == 7  # 8x8 block order


@pytest.mark.parametrize("n", [2, 3])
def test_bench_self_launcher_cpu(n):
    """`python3 bench.py --gpus N` without a launcher spawns its own N ranks (before importing torch), relays exactly one JSON line from
    rank 0 and exits with the children's status.  --launch-check keeps the ranks off the GPU: they join the gloo group and all-reduce."""
    import json
    import subprocess

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--launch-check"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert {k: d[k] for k in ("launch_check", "n_gpus", "n_ranks_seen", "backend", "rendered")} == {"launch_check": True, "n_gpus": n, "n_ranks_seen": n, "backend": "gloo", "rendered": False}
    # one record per rank came through the all-gather (round 5: the same gather carries the per-rank render check of --launch-render on a GPU box)
    assert [x["rank"] for x in d["ranks"]] == list(range(n)) and len({x["pid"] for x in d["ranks"]}) == n


def test_bench_self_launcher_reports_a_failed_rank():
    """A rank that cannot start (backend nccl without a GPU here; any failure on a GPU box) must end the launcher with a non-zero status
    instead of leaving the other ranks in a collective."""
    import subprocess

    import torch

    from conftest import ROOT

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl", "--launch-check"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0 and not res.stdout.strip()
