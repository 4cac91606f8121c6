"""Per-tile instance counts of a C4 frame (995,328 Gaussians, 1920x1080): which branch of the tile sort each tile takes."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import setup_training
from instantsplat_amd.gaussian_renderer import render
from instantsplat_amd import diff_gaussian_rasterization as dgr
dev = torch.device('cuda:0')
st = setup_training(syn_pointmap(12, 288, 288, 1920, 1080, seed=0), dev)
g = st.gaussians
dgr.keep_last_frame(True)
for v in (0, 5):
    cam = st.cameras[v]
    with torch.no_grad():
        render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))
    tiles, W, H = dgr._LAST_FRAME["tiles"], dgr._LAST_FRAME["W"], dgr._LAST_FRAME["H"]
    T = ((W + 15) // 16) * ((H + 15) // 16); al = lambda x: (x + 255) & ~255
    start = tiles[2 * al(T * 4): 2 * al(T * 4) + (T + 1) * 4].cpu().view(torch.int32).numpy().astype(np.int64)
    c = np.diff(start)
    print("view", v, "T", T, "R", int(start[-1]), "mean %.0f median %.0f max %d" % (c.mean(), np.median(c), c.max()))
    edges = [0, 1, 257, 513, 1025, 2049, 4097, 8193, 10**9]
    names = ["empty", "<=256 regs<1>", "<=512 regs<2>", "<=1024 regs<4>", "<=2048 regs<8>", "<=4096 long", "<=8192 long", ">8192 global"]
    for lo, hi, nm in zip(edges[:-1], edges[1:], names):
        m = (c >= lo) & (c < hi)
        print("   %-16s tiles %5d  instances %8d (%.1f%%)" % (nm, int(m.sum()), int(c[m].sum()), 100.0 * c[m].sum() / max(1, c.sum())))
