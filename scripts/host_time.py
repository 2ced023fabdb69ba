"""Host-side enqueue time vs GPU time of one training step (bs and size as bench.py): if enqueue ~ wall, the step is host-bound."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import recipe, yolov4_amd
from yolov4_amd.ddp import BucketedDDP
from yolov4_amd.yolo.model.yolov4 import YOLOv4
from yolov4_amd.yolo.model.yololoss import YOLOLoss
mode = sys.argv[1] if len(sys.argv) > 1 else 'f16x2'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
yolov4_amd.set_conv_mode(mode)
dev = torch.device('cuda:0')
m = YOLOv4(recipe.MODEL_CFG, device=dev)
sd = m.state_dict(); recipe.fill_state_dict_(sd, 1234); m.load_state_dict(sd)
m = m.to(dev).train()
ddp = BucketedDDP(m)
crit = YOLOLoss(recipe.MODEL_CFG, 0.7, device=dev)
x = torch.randn((B, 3, 608, 608)).to(dev)
labels = recipe.synth_labels(B, 608, 2000).to(dev)
def step():
    ddp.zero_grad()
    loss = crit(ddp(x), {'padded_labels': labels})
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{mode} B={B}: host enqueue {1e3 * (t1 - t0):.1f} ms, wall incl. drain {1e3 * (t2 - t0):.1f} ms', flush=True)
