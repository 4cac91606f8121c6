# round 5, second GPU call: GPU tier (whole), bench line, deterministic-mode cost + kernel stats.   usage (gpurun): bash tools/r5_gpu2.sh
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_gputest_full_v2.txt; tail -25 gpurun_out/r05_gputest_full_v2.txt
timeout 600 python bench.py > gpurun_out/r05_bench_default_run_v2.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_default_run_v2.json"))
r = d["roofline"]
print("value %.0f | loops %s" % (d["value"], {k: round(v["iters_per_sec"]) for k, v in d["loops"].items()}))
print("bwd %.1f us frac %.3f | fwd %.1f us | render-only %.1f us frac %.3f | raster %.3f ms | fps %s" % (
    r["avg_kernel_ms"] * 1e3, r["frac"], r["composite_fwd"]["avg_kernel_ms"] * 1e3, r["render_only"]["avg_kernel_ms"] * 1e3,
    r["render_only"]["frac"], d["rasterize_ms_per_frame"], d["fps_reference_method"]["fps"]))
PY
timeout 300 python tools/det_cost.py 2>/dev/null | grep "^{" > gpurun_out/r05_deterministic_mode_cost.txt; cat gpurun_out/r05_deterministic_mode_cost.txt
bash tools/prof.sh r05_det_c3 python tools/det_cost.py 100 > /dev/null 2>&1
grep -E "k_det|k_scan_block|k_scan_apply|k_composite_bwd|fillBuffer|Name" gpurun_out/r05_det_c3_kernel_stats.csv | cut -c1-220
