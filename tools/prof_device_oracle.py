"""torch.profiler table of a few device-oracle iterations (where do its 50 ms go?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, torch.nn.functional as F
from oracle import immoco_oracle as orc
from device_oracle_sampler import slice_input
dev = torch.device("cuda", 0)
k, lines, gt = slice_input(1)
masks = orc.extract_movement_groups(lines, make_list=True)
model = orc.OracleIMMoCo(masks, device=dev)
kin = (k / k.abs().max() * 16000).to(dev)
opt = torch.optim.Adam([{"params": model.motion_inr.parameters(), "lr": 1e-2}, {"params": model.image_inr.parameters(), "lr": 1e-2}])
def it():
    opt.zero_grad()
    kf, ip = model()
    loss = F.mse_loss(torch.view_as_real(kf), torch.view_as_real(kin)) + orc.gradient_entropy_loss(ip) * 1e-2
    loss.backward()
    opt.step()
for _ in range(3): it()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): it()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
