"""Does bench.py's VALU-issue model of k_composite_bwd count instructions correctly?  On FIXED frames (a C3 scene trained for a
while, then frozen) the backward of each view is run N times with the shipped kernel and once with its counting
instantiation; run this under `rocprofv3 --pmc SQ_INSTS_VALU` (tools/pmc.sh) and compare the mean of the shipped kernel's
launches with the model printed here — same frames, same instruction stream, no drift of the scene between the two numbers
(the live model was compared with a PMC pass of a different training state).
usage (GPU box):  bash tools/pmc.sh issue_model SQ_INSTS_VALU,SQ_INSTS_LDS python tools/validate_issue_model.py
Measurement helper, not product code."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import _lib
from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
from instantsplat_amd.gaussian_renderer import render
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, setup_training

dev = torch.device("cuda:0")
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev)
ra = RunAhead(st, window=10)
for _ in range(200):
    ra.step()
ra.flush()
if ra.trainer is not None:
    ra.trainer.close()
BinningPolicy.reset("exact")
g = st.gaussians
L = _lib.lib()


def fwd_bwd(cam):
    img = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))["render"]
    loss, _ = fused_l1_ssim_loss(img.unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)
    loss.backward()
    for p in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation, g.P):
        p.grad = None


N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(N):
    for cam in st.cameras:
        fwd_bwd(cam)
ctr = torch.zeros(16, dtype=torch.int64, device=dev)
_lib.check(L.mi355gs_profile_work_counters(_lib.ptr(ctr)), "work_counters")
for cam in st.cameras:
    fwd_bwd(cam)
torch.cuda.synchronize()
_lib.check(L.mi355gs_profile_work_counters(None), "work_counters")
steps, quads, quads_valid, lanes, reduced, waves = [float(x) / len(st.cameras) for x in ctr.tolist()[:6]]
tab = json.load(open(os.path.join(ROOT, "instantsplat_amd", "lib", "bwd_issue_model.json")))
INS, CYC = tab["INS"], tab["CYC"]
inits = max(steps - quads_valid / 4.0, 0.0)
ins = steps * INS["step"] + quads * INS["quad"] + quads_valid * INS["quad_valid"] + reduced * INS["reduce"] + inits * INS["init"] + waves * INS["wave"]
cyc = steps * CYC["step"] + quads * CYC["quad"] + quads_valid * CYC["quad_valid"] + reduced * CYC["reduce"] + inits * CYC["init"] + waves * CYC["wave"]
groups, hits, fsteps, fvalid, fblended, fwaves = [float(x) / len(st.cameras) for x in ctr.tolist()[8:14]]
ftab = json.load(open(os.path.join(ROOT, "instantsplat_amd", "lib", "fwd_issue_model.json")))
print(json.dumps({"kernel": "k_composite_fwd<4, false>", "frames": "C3 after 200 iterations, frozen; mean over the 3 views", "groups": groups, "hits": hits,
                  "walk_steps": fsteps, "valid_pairs": fvalid, "quadrant_waves": fwaves, "useful_lane_frac": fvalid / (64.0 * hits),
                  "model_valu_wave_instructions_per_launch": fsteps * ftab["INS"]["step"] + groups * ftab["INS"]["group"] + fwaves * ftab["INS"]["wave"],
                  "model_valu_issue_cycles_per_launch": fsteps * ftab["CYC"]["step"] + groups * ftab["CYC"]["group"] + fwaves * ftab["CYC"]["wave"],
                  "compare_with": "mean_SQ_INSTS_VALU of `k_composite_fwd<4; false>` in the PMC csv of this command"}))
print(json.dumps({"frames": "C3 after 200 iterations, frozen; mean over the 3 views", "steps": steps, "quadrant_bodies": quads,
                  "quadrant_bodies_with_valid_lanes": quads_valid, "reductions": reduced, "waves_with_work": waves,
                  "model_valu_wave_instructions_per_launch": ins, "model_valu_issue_cycles_per_launch": cyc,
                  "compare_with": "mean_SQ_INSTS_VALU of `k_composite_bwd<1, false>` in the PMC csv of this command"}))
