"""The RCCL calls bench.py makes for --gpus > 1, exercised with a world of one rank on a 1-GPU box (init with device_id, int64
all-reduce of the packed totals on the device, barrier, MAX of the wall time).
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 scripts/check_rccl_single.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
vec = np.arange(10, dtype=np.int64) * (1 << 40)
t = torch.from_numpy(vec).to(dev)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
assert np.array_equal(t.cpu().numpy(), vec * dist.get_world_size())
w = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(w, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("RCCL world=%d: int64 SUM and float64 MAX all-reduce on %s ok" % (dist.get_world_size(), dev))
dist.destroy_process_group()
