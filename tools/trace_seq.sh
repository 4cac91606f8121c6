cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o tr -- python bench.py --cpu-iters 0 --steps 60 --warmup 10 > gpurun_out/tr.log 2>&1 < /dev/null
f=$(find gpurun_out/tr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    n = r["Kernel_Name"]
    for key in ("k_scatter_lds", "k_count_tiles_lds", "k_sort_tiles_small", "k_composite_fwd", "k_composite_bwd", "k_adam_multi"):
        if key in n:
            out.append((key, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
import collections
seq = [f"{k[2:9]}:{d:.0f}" for k, d in out[:700]]
open("gpurun_out/tr_seq.txt", "w").write(" ".join(seq))
PY
rm -rf gpurun_out/tr
