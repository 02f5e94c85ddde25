"""hipGraph capture / replay of the forward on three streams while an RCCL process group (and its watchdog thread) is
alive, then the all-gather of the result rows: the combination the multi-GPU bench runs.  One rank, one GPU."""
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
from selfmask_amd import MaskFormer, synthetic_state_dict, synthetic_images, GraphedForward, StreamRing
m = MaskFormer(n_queries=20, patch_size=16, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
m.load_state_dict(synthetic_state_dict(0, "soft", patch_size=16)); m = m.to("cuda:0")
x = torch.from_numpy(synthetic_images(1, (8, 3, 224, 224))).cuda()
ref = m(x)["objectness"].clone()
g = GraphedForward(m); ring = StreamRing(torch.device("cuda", 0), 3)
rows = torch.zeros(9, 8, device="cuda")
ring.fork()
for k in range(9):
    with ring.next():
        rows[k] = g(x)["objectness"][:, -1, 0, 0]
ring.join()
gathered = torch.empty(9, 8, device="cuda"); dist.all_gather_into_tensor(gathered, rows); torch.cuda.synchronize()
print("graph:", g.captures, g.replays, g.failed, "match:", bool(torch.equal(gathered[8], ref[:, -1, 0, 0])))
dist.barrier(); dist.destroy_process_group()
