import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import limg_amd
from oracle.bind import Oracle, REC_DTYPE
orc = Oracle(); g = limg_amd.LimgHip(0)
def enc(d_img, alpha, ff):
    h, w = d_img.shape
    g.set_options(float_fast=ff)
    planes = g.alloc_planes_device(w, h)
    rec = torch.zeros((h // 8 * (w // 8), 16), dtype=torch.int32, device="cuda"); sh = torch.zeros(h // 8 * (w // 8), dtype=torch.int32, device="cuda")
    g.encode3d_device(d_img, alpha, planes, records=rec, shifts=sh); torch.cuda.synchronize()
    return planes, rec.cpu().numpy().view(REC_DTYPE).reshape(h // 8, w // 8), sh.cpu().numpy() & 0xFFFFFF
for kind, alpha, n in (("pn", True, 1024), ("rg", True, 1024), ("rga", True, 512), ("pn", False, 1024), ("rg", False, 1024), ("png", False, 0)):
    if kind == "png":
        import golden_util as gu
        img = gu.big_input("original_rgb", orc)
        img = np.ascontiguousarray(img[:616, :1024])
    else:
        img = orc.photo_noise(n, n, 1) if kind == "pn" else orc.random_gradient(n, n, 1, kind == "rg")
    d = torch.from_numpy(img.view(np.int32)).cuda()
    pe, re_, se = enc(d, alpha, False); pf, rf, sf = enc(d, alpha, True)
    nb = re_.size
    out = []
    for f in ("dirA_min", "dirA_max", "dirB_offset", "dirB_mag", "dirC_offset", "dirC_mag"):
        dd = np.abs(re_[f].astype(int) - rf[f].astype(int)).max(axis=-1)
        out.append("%s >2:%d max%d" % (f[3:], (dd > 2).sum(), dd.max()))
    pa = g.compare_device(d, pe["pDecoded"], alpha)[0]; pb = g.compare_device(d, pf["pDecoded"], alpha)[0]
    print(kind, alpha, nb, "|", " | ".join(out), "| psnr %.4f %.4f" % (pa, pb), "| shifts differ on", int((se != sf).sum()), flush=True)
