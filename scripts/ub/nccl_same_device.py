"""Probe: can RCCL run two ranks on ONE GPU (so that hg_prove_sharded's all-reduce could execute with N > 1 on a 1-GPU box)?
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 scripts/ub/nccl_same_device.py
Result on the MI355X pool (RCCL 2.26.6 / 2.27.7, also with RCCL_ENABLE_MULTI_RANK_PER_GPU=1): ncclCommInitRank fails with
"Duplicate GPU detected : rank 0 and rank 1 both on CUDA device ..." - the library's collective with more than one rank needs more
than one device."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda") * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", rank, "allreduce ok", t.tolist(), flush=True)
except Exception as e:
    print("rank", rank, "FAILED", str(e)[:300], flush=True)
