"""Generates tests/golden/cfg_args_reference.json: the text the reference's train.py leaves in <model_path>/cfg_args
(train.py:245-246, `str(Namespace(**vars(args)))`) for the command line of its scripts, produced by the reference's OWN argument
classes (arguments/__init__.py, loaded from /root/reference) and the parser set-up of train.py:298-314 — and what the reference's own
`get_combined_args` (arguments/__init__.py:96-116) makes of it for render.py; and the lines the reference's `save_time`
(utils/sfm_utils.py:43-50) appends to <model_path>/train_time.txt.  Run in the build container only."""
import importlib.util, json, os, sys
from argparse import ArgumentParser, Namespace

REF = "/root/reference"
spec = importlib.util.spec_from_file_location("ref_arguments", os.path.join(REF, "arguments", "__init__.py"))
A = importlib.util.module_from_spec(spec); spec.loader.exec_module(A)

argv = ["-s", "/data/scene", "-m", "/out/scene_3_views", "-r", "1", "--n_views", "3", "--iterations", "1000", "--pp_optimizer", "--optim_pose"]
parser = ArgumentParser(description="Training script parameters")          # train.py:298-312
lp, op, pp = A.ModelParams(parser), A.OptimizationParams(parser), A.PipelineParams(parser)
parser.add_argument('--ip', type=str, default="127.0.0.1")
parser.add_argument('--port', type=int, default=6009)
parser.add_argument('--debug_from', type=int, default=-1)
parser.add_argument('--detect_anomaly', action='store_true', default=False)
parser.add_argument("--test_iterations", nargs="+", type=int, default=[])
parser.add_argument("--save_iterations", nargs="+", type=int, default=[])
parser.add_argument("--quiet", action="store_true")
parser.add_argument('--disable_viewer', action='store_true', default=True)
parser.add_argument("--checkpoint_iterations", nargs="+", type=int, default=[])
parser.add_argument("--start_checkpoint", type=str, default=None)
args = parser.parse_args(argv)
args.save_iterations.append(args.iterations)                                  # train.py:314
dataset = lp.extract(args)                                                    # ModelParams.extract makes source_path absolute ...
text = str(Namespace(**vars(args)))                                           # ... on the group; cfg_args gets vars(args) as parsed

# render.py:253-264: its own parser, then get_combined_args merges the command line over the cfg_args text
out_dir = "/tmp/_cfg_golden"; os.makedirs(out_dir, exist_ok=True)
open(os.path.join(out_dir, "cfg_args"), "w").write(text)
rp = ArgumentParser(description="Testing script parameters")
model = A.ModelParams(rp, sentinel=True); pipeline = A.PipelineParams(rp)
rp.add_argument("--iterations", default=-1, type=int)
old = sys.argv; sys.argv = ["render.py", "-m", out_dir]
try:
    merged = A.get_combined_args(rp)
finally:
    sys.argv = old
ds = model.extract(merged)
# utils/sfm_utils.py:43-50 `save_time` (the module itself needs cv2 / roma / open3d: the function definition is taken from the file)
import ast, tempfile
from pathlib import Path
src = open(os.path.join(REF, "utils", "sfm_utils.py")).read()
fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "save_time")
ns = {"Path": Path}
exec(compile(ast.Module(body=[fn], type_ignores=[]), "sfm_utils.py", "exec"), ns)
td = tempfile.mkdtemp()
calls = [["[2] train_joint_TrainTime", 83.74], ["[2] train_joint", 125.2], ["[4] render", 3.9]]
for name, sec in calls:
    ns["save_time"](os.path.join(td, "model"), name, sec)
train_time_text = open(os.path.join(td, "model", "train_time.txt")).read()

json.dump({"argv": argv, "cfg_args_text": text, "save_time_calls": calls, "train_time_txt": train_time_text, "train_dataset_source_path": dataset.source_path,
           "render_dataset": {k: getattr(ds, k) for k in ("sh_degree", "source_path", "images", "resolution", "white_background", "data_device", "eval", "n_views",
                                                          "init_scale_from_view_depth")},
           "render_iterations": merged.iterations},
          open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg_args_reference.json"), "w"), indent=1)
print(text)
