"""Where do the 1000 iterations of `iters_per_sec_1k` spend their wall clock?  (bench.py's headline is the median 20-iteration block
of iterations 200..1000; the 1k figure is the whole run from iteration 1 and has read 20 % lower on some boxes.)  Per 50-iteration
stretch: wall time with a device synchronise at each end, for the reference loop and for the one-call step.
Measurement helper, not product code."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd.launch import pin_mode, pin_rank_to_cpu_slice
if pin_mode() != "off":
    pin_rank_to_cpu_slice(0, 1, device_of_rank=lambda r: 0, compact=pin_mode() == "compact")
from instantsplat_amd.arguments import OptimizationParams
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import release_trainer, setup_training, train_iteration
dev = torch.device("cuda:0")
scene = syn_pointmap(3, 256, 256, 512, 512, seed=0)
for name, kw in (("reference loop (train.py loss as written)", dict(fused_loss=False)), ("one-call step", dict(fused_step=True))):
    for rep in range(2):
        st = setup_training(scene, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks = []
        for i in range(1000):
            if i == 999:   # (the reference skips the optimiser on the last iteration)
                pass
            train_iteration(st, **kw)
            if i in (0, 1, 2, 3, 9) or (i + 1) % 50 == 0:
                torch.cuda.synchronize()
                marks.append((i + 1, time.perf_counter() - t0))
        release_trainer(st)
        if rep == 0:   # a second, finer look at the first 60 iterations: every iteration on its own clock, with the caching allocator's counters
            st2 = setup_training(scene, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True))
            torch.cuda.synchronize()
            slow = []
            for i in range(60):
                ms0 = torch.cuda.memory_stats(dev)
                t = time.perf_counter()
                train_iteration(st2, **kw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t
                ms1 = torch.cuda.memory_stats(dev)
                if dt > 1e-3:
                    slow.append(f"it {i + 1}: {dt * 1e3:.2f} ms, cudaMalloc calls +{ms1['num_device_alloc'] - ms0['num_device_alloc']}, "
                                f"reserved {ms0['reserved_bytes.all.current'] >> 20} -> {ms1['reserved_bytes.all.current'] >> 20} MiB, "
                                f"frees +{ms1['num_device_free'] - ms0['num_device_free']}")
            release_trainer(st2)
            print(f"{name}: iterations of the first 60 over 1 ms: " + ("; ".join(slow) or "none"), flush=True)
            del st2
        total = marks[-1][1]
        prev_i, prev_t, out = 0, 0.0, []
        for i, t in marks:
            out.append(f"{i}:{1e3 * (t - prev_t) / (i - prev_i):.3f}")
            prev_i, prev_t = i, t
        print(f"{name}, run {rep}: 1000 iterations {total * 1e3:.1f} ms = {1000 / total:.0f} it/s; ms per iteration up to iteration N: " + " ".join(out), flush=True)
        del st
