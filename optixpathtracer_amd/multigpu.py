"""Multi-GPU: one process per GPU, image-space sharding, one RCCL all-gather per displayed frame.

The reference is single-GPU (SURVEY.md §2 "Parallelism": none).  Pixels are independent and seeds depend
only on (pixel index, subframe) (deviceProgram.cu:357), so the image is cut into interleaved tiles
(the pattern of sutil/WorkDistribution.h:34-91) and every rank renders its own tiles with the scene, BVH
and probe replicated; progressive accumulation needs no communication.  The only exchange is the display
hand-off: each rank packs its pixels (pt_pack), ranks all-gather the packed arrays over RCCL/xGMI
(rank-major, padded to the largest share — exactly what all_gather_into_tensor produces) and scatter
them into the full frame (pt_unpack).  4K float4 = 133 MB in total, 16.6 MB per rank on 8 GPUs: a
single direct all-gather keeps all 7 xGMI links busy with one message each; no ring all-reduce of
zero-padded full frames (8x the bytes, per-link bound).
"""
from __future__ import annotations

import numpy as np


def pixel_lists(width: int, height: int, world: int, tile_w: int = 64, tile_h: int = 16):
    """Host mirror of the library's partition (pt_api.hip build_pixel_lists): per rank, the owned pixels as
    x | y << 16 in 8x8-block order; owner of a block = (x0/tile_w + y0/tile_h) % world."""
    assert tile_w % 8 == 0 and tile_h % 8 == 0 and world >= 1
    lists = [[] for _ in range(world)]
    for by in range((height + 7) // 8):
        for bx in range((width + 7) // 8):
            owner = ((bx * 8) // tile_w + (by * 8) // tile_h) % world
            ys, xs = np.meshgrid(np.arange(by * 8, by * 8 + 8), np.arange(bx * 8, bx * 8 + 8), indexing="ij")
            ok = (xs < width) & (ys < height)
            lists[owner].append((xs[ok].astype(np.uint32) | (ys[ok].astype(np.uint32) << 16)))
    return [np.concatenate(l) if l else np.zeros(0, np.uint32) for l in lists]


class DevicePacker:
    """pack/unpack through the library's kernels on torch CUDA tensors (device memory + RCCL plumbing)."""

    def __init__(self, renderer):
        import torch

        self.torch = torch
        self.r = renderer
        self._bufs = {}

    def owned_padded(self):
        return self.r.ownedPixels()

    def alloc(self, n, which, role=0):
        from .renderer import PT_BUF_FRAME

        key = (n, which == PT_BUF_FRAME, role)
        # one send and one receive buffer per (size, element type), reused every frame (sizes differ: padded vs padded * world)
        buf = self._bufs.get(key)
        if buf is None:
            if which == PT_BUF_FRAME:
                buf = self.torch.zeros(n, dtype=self.torch.int32, device="cuda")
            else:
                buf = self.torch.zeros((n, 4), dtype=self.torch.float32, device="cuda")
            self._bufs[key] = buf
            # torch zero-fills on its current stream; the library's streams are non-blocking (they do not wait for the null stream),
            # so the fill must be complete before a pack kernel may write the buffer
            self.torch.cuda.current_stream().synchronize()
        return buf

    def sync(self):
        """The all-gather runs on RCCL's stream and torch's current stream waits for it; pt_unpack runs on the library's own
        (non-blocking) stream, so the hand-over between the two is made explicit (include/pt_amd.h, STREAM CONTRACT): an event recorded
        on torch's stream behind the collective, which the context's stream waits for on the device — no host wait."""
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream())
        self.r.waitEvent(ev.cuda_event)
        self._ev = ev  # alive until the next hand-over

    def pack(self, which, dst):
        self.r.pack(which, dst.data_ptr())

    def unpack(self, which, src):
        self.r.unpack(which, src.data_ptr())

    # overlapped hand-off (pt_pack_async / pt_pack_wait / pt_unpack_display)
    def pack_async(self, which, dst, slot):
        self.r.packAsync(which, dst.data_ptr(), slot)

    def pack_wait(self, slot):
        self.r.packWait(slot)

    def unpack_display(self, which, src):
        self.r.unpackDisplay(which, src.data_ptr())


def exchange_frame(packer, which, world: int, all_gather_into_tensor):
    """The display hand-off: pack → all-gather → unpack.  `all_gather_into_tensor(dst, src)` is
    torch.distributed's (backend nccl == RCCL on ROCm; gloo in the CPU tests)."""
    owned, padded = packer.owned_padded()
    src = packer.alloc(padded, which)
    dst = packer.alloc(padded * world, which, 1) if isinstance(packer, DevicePacker) else packer.alloc(padded * world, which)
    packer.pack(which, src)  # returns after the pack kernel finished (pt_pack synchronises its stream)
    all_gather_into_tensor(dst, src)
    if hasattr(packer, "sync"):
        packer.sync()
    packer.unpack(which, dst)
    return dst


class HandOff:
    """The display hand-off overlapped with rendering (include/pt_amd.h, pt_pack_async ...).  With frames in flight the loop is

        for k in frames:  render(k);  h.collect();  h.submit()
        h.collect()       # the last frame

    submit() enqueues the pack of the frame just enqueued behind its last kernel (no host wait) into one of two send buffers;
    collect() takes the frame submitted before: waits for ITS pack only, all-gathers the strips (RCCL over xGMI), and scatters them
    into the display copy of the buffer — all while the next frame renders.  collect() is called before the next submit() so that the
    scatter is not queued behind the wait for the frame that is still rendering."""

    def __init__(self, packer, which, world: int, all_gather_into_tensor):
        self.packer, self.which, self.world, self.all_gather = packer, which, world, all_gather_into_tensor
        _, padded = packer.owned_padded()
        self.send = [packer.alloc(padded, which, 2 + s) if isinstance(packer, DevicePacker) else packer.alloc(padded, which) for s in (0, 1)]
        self.recv = packer.alloc(padded * world, which, 1) if isinstance(packer, DevicePacker) else packer.alloc(padded * world, which)
        self.n = 0
        self.pending = None

    def submit(self):
        slot = self.n & 1
        self.packer.pack_async(self.which, self.send[slot], slot)
        self.pending = slot
        self.n += 1

    def collect(self):
        """Returns True when a frame was handed over (it is then complete in the display buffer once the library's stream reaches it:
        pt_display_sync / pt_download_display wait for that)."""
        if self.pending is None:
            return False
        slot, self.pending = self.pending, None
        self.packer.pack_wait(slot)
        self.all_gather(self.recv, self.send[slot])
        if hasattr(self.packer, "sync"):
            self.packer.sync()
        self.packer.unpack_display(self.which, self.recv)
        return True
