"""Ad-hoc: spread of the loss EMA between the synchronous autograd loop and the run-ahead one-call loop (tests/ops_util.py
check_run_ahead_equals_sync_loop), repeated in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantsplat_amd.arguments import OptimizationParams
from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, setup_training, train_iteration
dev = torch.device("cuda:0")
sc = syn_pointmap(3, 20, 20, 48, 48, seed=7)
names = ("_xyz", "_features_dc", "_opacity", "_scaling", "P")
mk = lambda: setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    out = []; params = []
    for mode in ("sync", "sync", "ahead", "ahead", "ahead_unfused"):
        st = mk()
        if mode == "sync":
            ema = 0.0
            for _ in range(23): ema = 0.4 * train_iteration(st) + 0.6 * ema
        else:
            ra = RunAhead(st, window=5, fused_step=(mode == "ahead"))
            for _ in range(23): ra.step()
            ema = ra.flush()
        out.append(ema)
        params.append({n: getattr(st.gaussians, n).detach().cpu().clone() for n in names})
        BinningPolicy.reset("exact")
    print(rep, " ".join("%.6f" % e for e in out), "| rel(sync,ahead) %.2e  rel(sync,sync) %.2e  rel(ahead,ahead) %.2e" % (
        abs(out[0] - out[2]) / out[0], abs(out[0] - out[1]) / out[0], abs(out[2] - out[3]) / out[0]))
    rel = lambda a, b: {n: float((a[n] - b[n]).norm() / (a[n].norm() + 1e-12)) for n in names}
    print("    params sync vs ahead:", " ".join("%s %.1e" % kv for kv in rel(params[0], params[2]).items()))
    print("    params sync vs sync :", " ".join("%s %.1e" % kv for kv in rel(params[0], params[1]).items()))
