"""RCCL plumbing check on one GPU (world size 1): the collectives bench.py uses, with its dtypes and shapes, and the display hand-off
(pack -> all_gather_into_tensor -> unpack) through the nccl backend."""
import os, sys
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.tensor([1.5, 2.0], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.all_reduce(t, op=dist.ReduceOp.SUM); dist.barrier()
from optixpathtracer_amd import scenes, renderer as R, multigpu
m = scenes.cornell_box(); probe = scenes.sky_probe(256, 128).BuildCDF()
r = R.SampleRenderer(m, device=0); r.setProbe(probe); r.setPartition(0, 1, 64, 16); r.resize((200, 120)); r.setCamera(R.make_camera(scenes.CORNELL_CAMERA, 200 / 120))
r.launchParams.samples_per_launch = 2; r.render()
before = (r.download(R.PT_BUF_FRAME).copy(), r.download(R.PT_BUF_ACCUM).copy())
packer = multigpu.DevicePacker(r)
for which in (R.PT_BUF_FRAME, R.PT_BUF_ACCUM):
    multigpu.exchange_frame(packer, which, 1, lambda dst, src: dist.all_gather_into_tensor(dst, src))
torch.cuda.synchronize()
assert np.array_equal(r.download(R.PT_BUF_FRAME), before[0]) and np.array_equal(r.download(R.PT_BUF_ACCUM).view(np.uint32), before[1].view(np.uint32))
dist.barrier(); dist.destroy_process_group()
print("nccl smoke ok", float(t[0]))
