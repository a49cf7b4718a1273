#!/usr/bin/env python3
"""Pack the CC0 benchmark meshes into compact fixtures under assets/.

The bob and spot models (Keenan Crane's model repository, CC0 1.0 Universal -- see the
LICENSE.txt files next to them in the reference checkout) are the inputs BASELINE.json
names ("bob mesh", "spot_metal").  /root/reference does not exist on the GPU box, so the
geometry is converted once, here, into a small .npz (float32 positions, int32 triangles,
per-corner uv indices, a down-sampled linear-RGB kd texture).  Data only, no reference code.

    python tools/make_assets.py [/root/reference]
"""
import os, sys
import numpy as np

def load_obj(path):
    v, vt, f, ft = [], [], [], []
    with open(path) as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == 'v':
                v.append([float(x) for x in p[1:4]])
            elif p[0] == 'vt':
                vt.append([float(p[1]), float(p[2])])
            elif p[0] == 'f':
                idx = [q.split('/') for q in p[1:]]
                assert len(idx) == 3, "triangulated meshes only"
                f.append([int(q[0]) - 1 for q in idx])
                ft.append([int(q[1]) - 1 if len(q) > 1 and q[1] else 0 for q in idx])
    return (np.asarray(v, np.float32), np.asarray(vt, np.float32),
            np.asarray(f, np.int32), np.asarray(ft, np.int32))

def srgb_to_linear(x):
    return np.where(x <= 0.04045, x / 12.92, ((np.maximum(x, 0.04045) + 0.055) / 1.055) ** 2.4)

def load_tex(path, res):
    from PIL import Image
    img = Image.open(path).convert('RGB').resize((res, res), Image.BILINEAR)
    a = np.asarray(img, np.float32) / 255.0
    return srgb_to_linear(a).astype(np.float16)

def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'assets')
    os.makedirs(out, exist_ok=True)
    for name, obj, tex, ks in (('bob', 'data/bob/bob_tri.obj', 'data/bob/bob_diffuse.png', (0.0, 0.25, 0.0)),
                               ('spot', 'data/spot/spot.obj', 'data/spot/spot_texture.png', (0.0, 0.2, 1.0))):
        v, vt, f, ft = load_obj(os.path.join(ref, obj))
        kd = load_tex(os.path.join(ref, tex), 512)
        np.savez_compressed(os.path.join(out, name + '.npz'), v_pos=v, v_tex=vt, t_pos_idx=f, t_tex_idx=ft,
                            kd_tex=kd, ks=np.asarray(ks, np.float32))
        print(name, v.shape, f.shape, vt.shape, kd.shape)

if __name__ == '__main__':
    main()
