"""Per-layer difference between two conv arithmetic modes on the smoke() network (eval, calibrated BN):
usage: python scripts/mode_diff.py [modeA=f32] [modeB=f16x2] [size=64]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, recipe, yolov4_amd
from yolov4_amd.yolo.model.yolov4 import YOLOv4
from yolov4_amd.darknet.darknet import ConvBNAct
ma, mb = (sys.argv[1:3] + ['f32', 'f16x2'])[:2] if len(sys.argv) > 2 else ('f32', 'f16x2')
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device('cuda:0')
cfg = recipe.MODEL_CFG
model = YOLOv4(cfg, device=dev)
sd = {k: v.cpu() for k, v in model.state_dict().items()}
recipe.fill_state_dict_(sd, 7)
model.load_state_dict(sd)
model = model.to(dev)
yolov4_amd.set_conv_mode(ma)
recipe.calibrate_bn_(model, recipe.randn((8, 3, S, S), 3).to(dev))
model.eval()
x = recipe.randn((2, 3, S, S), 1).to(dev)
outs = {}
names = {m: n for n, m in model.named_modules() if isinstance(m, ConvBNAct)}
def hook(m, i, o):
    outs.setdefault(cur[0], {})[names[m]] = (o.detach().float().cpu().clone(), i[0].detach().float().cpu().clone())
for m in names:
    m.register_forward_hook(hook)
cur = [ma]
res = {}
for mode in (ma, mb):
    cur[0] = mode
    yolov4_amd.set_conv_mode(mode)
    with torch.no_grad():
        res[mode] = model(x).cpu()
print('final: max abs diff scores', (res[ma][..., 4:] - res[mb][..., 4:]).abs().max().item(),
      'boxes rel', ((res[ma][..., :4] - res[mb][..., :4]).abs().max() / res[ma][..., :4].abs().max()).item())
for n in outs[ma]:
    a, xa = outs[ma][n]; b, xb = outs[mb][n]
    d = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30)
    dx = (xa - xb).abs().max().item() / max(xa.abs().max().item(), 1e-30)
    flag = '  <<<' if d > 20 * max(dx, 1e-7) else ''
    print(f'{n:40s} out {tuple(a.shape)} rel diff {d:.2e} (input diff {dx:.2e}) amax {a.abs().max().item():.3g}{flag}')
