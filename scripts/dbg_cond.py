"""Is the gradient mismatch conditioning or a bug?  Same linear functional of the head logits,
three backends: CPU fp64 (truth), CPU fp32 (the reference's arithmetic), HIP fp32."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, recipe
from oracle import network as NW
from yolov4_amd.yolo.model.yolov4 import YOLOv4
S, B = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda:0')
sd = NW.empty_state_dict(); recipe.fill_state_dict_(sd, 1234)
x = recipe.randn((B, 3, S, S), 80)
G = [recipe.randn((B, 255, S // s, S // s), 900 + i) for i, s in enumerate((8, 16, 32))]
def run_cpu(dtype):
    net = NW.RefNet({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}, recipe.MODEL_CFG)
    lg = net.forward_train(x.to(dtype))
    torch.autograd.backward(lg, [g.to(dtype) for g in G])
    return {k: v.grad.double() for k, v in net.p.items() if v.grad is not None}, [t.detach().double() for t in lg]
g64, l64 = run_cpu(torch.float64)
g32, l32 = run_cpu(torch.float32)
m = YOLOv4(recipe.MODEL_CFG, device=dev); m.load_state_dict(sd); m = m.to(dev).train()
p = m.neck(*m.backbone(x.to(dev))); lg = m.head.logits(*p)
torch.autograd.backward(lg, [g.to(dev) for g in G])
gh = {k: v.grad.double().cpu() for k, v in m.named_parameters()}
lh = [t.detach().double().cpu() for t in lg]
for i in range(3):
    s = l64[i].abs().max()
    print(f'logits{i}: cpu32 err {float((l32[i]-l64[i]).abs().max()/s):.2e}  hip err {float((lh[i]-l64[i]).abs().max()/s):.2e}')
rows = []
for k in g64:
    n = g64[k].norm()
    rows.append((k, float((g32[k]-g64[k]).norm()/n), float((gh[k]-g64[k]).norm()/n)))
worst = sorted(rows, key=lambda r: -r[2])[:12]
print('worst HIP relative L2 errors vs fp64 (param, cpu32 err, hip err):')
for r in worst: print(f'  {r[0]:58s} {r[1]:.2e} {r[2]:.2e}')
a = np.array([r[1] for r in rows]); b = np.array([r[2] for r in rows])
print(f'median cpu32 {np.median(a):.2e} hip {np.median(b):.2e}; max cpu32 {a.max():.2e} hip {b.max():.2e}; ratio of medians {np.median(b)/np.median(a):.2f}')
