"""End-to-end sanity of the training path as the reference's train() drives it (yolo/engine/build.py:37-69):
model -> YOLOLoss -> backward (BucketedDDP, in-place gradient slots) -> fused Adam, on one fixed synthetic batch.
The loss must stay finite and fall.  usage: python scripts/train_sanity.py [steps] [batch] [size]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, recipe
from yolov4_amd.yolo.model.yolov4 import YOLOv4
from yolov4_amd.yolo.model.yololoss import YOLOLoss
from yolov4_amd.yolo.optim.optimizers.build import build_optimizer
from yolov4_amd.ddp import BucketedDDP

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 416
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = dict(recipe.FULL_CFG)
cfg['OPTIMIZER'] = {'TYPE': 'ADAM', 'LR': '3e-4', 'NO_BIAS': True, 'NO_NORM': True}
m = YOLOv4(recipe.MODEL_CFG, device=dev).to(dev).train()          # the reference's own initialisation
ddp = BucketedDDP(m)
opt = build_optimizer(cfg, m)
crit = YOLOLoss(recipe.MODEL_CFG, 0.7, device=dev)
x = recipe.randn((B, 3, S, S), 80).to(dev)
labels = recipe.synth_labels(B, S, 81)
losses = []
for i in range(steps):
    ddp.zero_grad()
    loss = crit(ddp(x), {'padded_labels': labels})
    loss.backward()
    ddp.finish_backward()
    opt.step()
    losses.append(float(loss))
    if i % 5 == 0 or i == steps - 1:
        print(f'step {i:3d} loss {losses[-1]:.3f}', flush=True)
assert all(l == l and abs(l) < 1e12 for l in losses), 'non-finite loss'
assert losses[-1] < 0.97 * losses[0], (losses[0], losses[-1])
print('ok: loss', losses[0], '->', losses[-1])
