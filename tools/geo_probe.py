"""Geometry-unlocked iterations on a perturbed bob: loss and vertex error over 80 steps for a few position learning rates (round 4)."""
import sys, torch
sys.path.insert(0, '.')
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import scene as sc
dev = torch.device('cuda:0')
for lr_pos, pert in ((1e-4, 0.004), (3e-4, 0.004), (1e-4, 0.0)):
    st = DirectLightingStep('bob', 160, 4, view=[0, 2, 5], device=dev, tex_res=512, optimize_geometry=True, perturb_pos=pert, lr=0.01, lr_pos=lr_pos)
    with torch.no_grad():
        st.kd_tex.copy_(st.mesh['kd_tex']); st.ks_tex.copy_(st.mesh['ks'].view(1, 1, 3).expand_as(st.ks_tex)); st.light.base.copy_(sc.env_map('E1', 256).to(dev))
    for name in st.param_names[:-1]:
        st.set_lr_scale(name, 0.0)
    vt = st.mesh['v_pos']
    out = []
    for k in range(80):
        l = float(st.step().detach())
        if k % 8 == 0 or k == 79:
            out.append('%d: loss %.5f err %.4f' % (k, l, float((st.v_pos.detach() - vt).norm())))
    print(lr_pos, pert, ' | '.join(out))
