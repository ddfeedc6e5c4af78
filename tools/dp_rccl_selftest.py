import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
coach = bench.build_coach(256, 2, "cuda:0", True, "hip")
w = bench.synthetic_latents(coach.net.decoder, 2, 0)
for _ in range(3):
    d = coach.train_step(w)
dist.barrier(); torch.cuda.synchronize()
print("RCCL world-1 DP step ok, loss", float(d["loss"]), "bucket MB", coach.bucket.nbytes / 1e6)
dist.destroy_process_group()
