"""One-rank nccl (= RCCL) group on cuda:0: the very torch.distributed calls of professad_amd.distributed on the kinds of tensors it
passes -- zero-copy views of engine-owned device memory (`_RawDeviceBuffer`), a side stream, async all-to-all + wait, the in-place
all-reduce of a slice of the context's device scalars.  A single rank moves no data between GPUs, but every call goes through
ProcessGroupNCCL (stream synchronisation, recordStream on foreign memory, dtype / contiguity checks).  Prints 'ok'."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from professad_amd import synth  # noqa: E402
from professad_amd.distributed import Comm, HipStages, _RawDeviceBuffer  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%s' % os.environ.get('MASTER_PORT', '29731'), world_size=1, rank=0)
comm = Comm()
assert not comm.active and dist.get_backend() == 'nccl'
# engine-owned memory: the context's 16 device scalars, wrapped exactly as HipStages wraps them
st = HipStages((16, 16, 32), dev, nranks=1, rank=0)
st.set_cell(torch.as_tensor(synth.cubic_cell(16)))
sc = st.device_scalars
sc[:] = torch.arange(16, dtype=torch.double, device=dev)
dist.all_reduce(sc[0:13], op=dist.ReduceOp.SUM)                          # Comm.all_reduce_dev under nccl
dist.all_reduce(sc[15:16], op=dist.ReduceOp.SUM)
assert torch.equal(sc.cpu(), torch.arange(16, dtype=torch.double))
# exchange buffers: raw uint8 views, async all-to-all issued on a side stream, the wait orders that stream after it
n = 1 << 20
a = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev)
b = torch.zeros(n, dtype=torch.uint8, device=dev)
send = torch.as_tensor(_RawDeviceBuffer(a.data_ptr(), n), device=dev)
recv = torch.as_tensor(_RawDeviceBuffer(b.data_ptr(), n), device=dev)
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    w = dist.all_to_all_single(recv, send, async_op=True)               # Comm.all_to_all under nccl
    w.wait()
    chk = (recv == send).all()
torch.cuda.current_stream(dev).wait_stream(side)
assert bool(chk) and torch.equal(a, b)
st.close()
dist.destroy_process_group()
print('ok')
