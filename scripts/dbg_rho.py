import torch, numpy as np, sys
sys.path.insert(0,'.')
from oracle import sched_ref, loss_ref
from oracle import backdoor_ref as BD
from baddiffusion_amd import ops
_, a, ac = sched_ref.make_tables()
B,S=6,32
u8 = torch.randint(0, 256, (B, S, S, 3), generator=torch.Generator().manual_seed(0), dtype=torch.uint8)
img = torch.stack([BD.image_u8_to_float(u) for u in u8])
g = BD.get_trigger("BOX_14", 3, S); y = BD.get_target("CORNER", g)
pois = torch.tensor([True, False, True, False, False, True])
eps = torch.randn(B,3,S,S, generator=torch.Generator().manual_seed(1)); t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(2))
Rr, x0 = BD.make_batch(img, pois, g, y)
xn_ref, tg_ref = loss_ref.q_sample(a, ac, x0, Rr, t, eps)
for images in (img.cuda(), u8.cuda()):
    xn, tg, Rg, x0g = ops.poison_qsample(images, pois.cuda(), g.cuda(), y.cuda(), eps.cuda(), t.cuda(), a.cuda(), ac.cuda(), want_batch=True)
    tgc = tg.permute(0,3,1,2).cpu()
    d = (tgc - tg_ref)
    print("per-row max |dtg|", d.abs().amax((1,2,3)).numpy(), " R diff", (Rg.cpu()-Rr).abs().amax((1,2,3)).numpy())
    for b in (0,2,5):
        m = Rr[b].abs() > 0.1
        rho_dev = ((tgc[b]-eps[b])[m] / Rr[b][m]).double().mean()
        rho_ref = ((tg_ref[b]-eps[b])[m] / Rr[b][m]).double().mean()
        print(b, int(t[b]), float(rho_dev), float(rho_ref))
xn2, tg2 = ops.qsample(x0.cuda(), Rr.cuda(), eps.cuda(), t.cuda(), a.cuda(), ac.cuda())
print("qsample kernel max diff", (tg2.permute(0,3,1,2).cpu()-tg_ref).abs().max().item())
