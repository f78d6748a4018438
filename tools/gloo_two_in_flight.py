"""Does gloo complete two asynchronous all_reduce of device tensors that are in flight together?
(Rehearsal aid for bench.py's N > 1 path on a one-GPU box; not part of the product.)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/gloo_two_in_flight.py"""
import faulthandler, os, sys, time
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
faulthandler.dump_traceback_later(45, exit=True)          # a stall ends here, with every thread's stack on stderr
torch.cuda.set_device(0)
dist.init_process_group("gloo")
a = torch.full((1080, 1920, 4), rank + 1, dtype=torch.int32, device="cuda")
b = torch.full((1080, 1920, 4), 10 * (rank + 1), dtype=torch.int32, device="cuda")
for rep in range(4):
    wa = dist.all_reduce(a, async_op=True)
    wb = dist.all_reduce(b, async_op=True)
    wa.wait(); print(f"[rank {rank}] rep {rep} first done", file=sys.stderr, flush=True)
    wb.wait(); print(f"[rank {rank}] rep {rep} second done", file=sys.stderr, flush=True)
torch.cuda.synchronize()
print(f"[rank {rank}] ok", int(a[0, 0, 0]), int(b[0, 0, 0]), file=sys.stderr, flush=True)
dist.destroy_process_group()
