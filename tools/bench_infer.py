"""Dev: throughput of the bf16 inference path (BASELINE.json config c2: FrostNet-Large, B = 256, 224x224)."""
import os, sys, time, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import frostnet as F
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(1882)
model = F.MODEL_REGISTRY["frostnet_large_1_0"]().cuda().eval()
x = torch.randn(B, 3, 224, 224, device="cuda").contiguous(memory_format=torch.channels_last)
for _ in range(3): model.hip_infer_bf16(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): model.hip_infer_bf16(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"c2 bf16 inference: batch {B}: {dt*1e3:.2f} ms/batch = {B/dt:.0f} img/s  (algorithmic 26.13 MB/img -> {B/dt*26.13e6/1e9:.0f} GB/s of 8000)")
