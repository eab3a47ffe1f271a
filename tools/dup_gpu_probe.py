#!/usr/bin/env python3
"""Can two RCCL ranks share ONE GPU (so that the N = 2 exchange could be rehearsed on a one-GPU box)?  No: ncclCommInitRank fails
with "Duplicate GPU detected : rank 0 and rank 1 both on CUDA device ..." (RCCL 2.26.6, round 3).  Kept as the evidence behind
DESIGN.md section 6: N > 1 over RCCL needs N devices; the N-rank protocol is covered on one GPU by the local transport instead.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/dup_gpu_probe.py"""
import os, sys, torch, torch.distributed as dist
rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda",0))
x=torch.ones(4,device="cuda")*(rank+1)
try:
    dist.all_reduce(x); torch.cuda.synchronize()
    print("rank",rank,"allreduce ok",x.tolist(),flush=True)
    y=torch.arange(8,device="cuda",dtype=torch.float32)+100*rank; z=torch.empty_like(y)
    dist.all_to_all_single(z,y); torch.cuda.synchronize()
    print("rank",rank,"a2a ok",z.tolist(),flush=True)
except Exception as e:
    print("rank",rank,"FAILED",repr(e)[:300],flush=True)
dist.destroy_process_group()
