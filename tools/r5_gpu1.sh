# round 5, first GPU call: the GPU test tier, the default bench line, the kernel stats of the same command, and the render-only A/B
# with its two PMC passes.       usage (gpurun): bash tools/r5_gpu1.sh
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_gputest_tail_v1.txt; cat gpurun_out/r05_gputest_tail_v1.txt
timeout 600 python bench.py > gpurun_out/r05_bench_default_run_v1.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_default_run_v1.json"))
r = d["roofline"]
print("value %.0f | loops %s" % (d["value"], {k: round(v["iters_per_sec"]) for k, v in d["loops"].items()}))
print("bwd %.1f us frac %.3f | fwd %.1f us | render-only %.1f us frac %.3f | raster %.3f ms | fps %s" % (
    r["avg_kernel_ms"] * 1e3, r["frac"], r["composite_fwd"]["avg_kernel_ms"] * 1e3, r["render_only"]["avg_kernel_ms"] * 1e3,
    r["render_only"]["frac"], d["rasterize_ms_per_frame"], d["fps_reference_method"]))
PY
timeout 300 python tools/render_only_loop.py 2>/dev/null | grep "^{" > gpurun_out/r05_ab_render_only_forward.txt; cat gpurun_out/r05_ab_render_only_forward.txt
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc.sh render_only $c python tools/render_only_loop.py 10 > /dev/null 2>&1
  cp gpurun_out/pmc_render_only_$c.csv gpurun_out/r05_pmc_render_only_$c.csv; grep -E "kernel,|composite_fwd" gpurun_out/r05_pmc_render_only_$c.csv
done
bash tools/prof.sh r05_bench_c3 python bench.py --steps 200 --warmup 20 --cpu-iters 0 --no-long-run > /dev/null 2>&1
head -20 gpurun_out/r05_bench_c3_kernel_stats.csv | cut -c1-160
