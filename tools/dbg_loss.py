import sys, torch
sys.path.insert(0, '/root/repo')
import nvdiffrecmc_amd.renderutils as ru
from oracle import renderutils_ref as rr
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
img = (torch.rand(2, 512, 512, 3, generator=g) * 3 - 0.5)
tgt = torch.rand(2, 512, 512, 3, generator=g) * 2
for special in (False, True):
    if special: img[0, 0, 0] = torch.tensor([70000.0, -2.0, 1.0])
    for loss, tm in (('l1', 'log_srgb'), ('l1', 'none'), ('relmse', 'none'), ('mse', 'none')):
        ref = rr.image_loss(img, tgt, loss, tm)
        out = ru.image_loss(img.to(dev), tgt.to(dev), loss=loss, tonemapper=tm)
        print(special, loss, tm, ref.item(), out.item())
