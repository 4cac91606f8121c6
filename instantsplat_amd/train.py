"""Joint Gaussian + camera-pose optimisation loop — the body of reference train.py:124-227 on the
HIP path, driven by a synthetic 3-view pointmap scene (no MASt3R / dataset offline).

Per iteration (same order as the reference): LR schedule -> (SH degree up every 1000) -> pop a random
view -> render(camera_pose=P[uid]) -> (1-l)*L1 + l*(1-SSIM) -> backward -> loss.item() ->
optimizer.step() unless it is the last iteration -> zero_grad(set_to_none=True).
"""
from __future__ import annotations

import ctypes
import gc
import math
import random
import time
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _lib
from .arguments import ModelParams, OptimizationParams, PipelineParams
from .optim import PerPointAdam
from .fused_ssim import fused_l1_ssim_loss, fused_ssim
from .diff_gaussian_rasterization import BinningPolicy, binning_hint
from .gaussian_renderer import render
from .pose_utils import quadmultiply
from .scene import GaussianModel, confidence_to_lr_modifiers
from .synthetic import PointmapScene


from .loss_utils import l1_loss   # reference utils/loss_utils.py:39-40 as one HIP node
from .lazy_loss import LazyScalar


def psnr(img1, img2):
    """reference utils/image_utils.py:17-19 (per-row-of-first-dim MSE)."""
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


@dataclass
class TrainState:
    gaussians: GaussianModel
    cameras: list
    gt_images: List[torch.Tensor]
    background: torch.Tensor
    opt: OptimizationParams
    pipe: PipelineParams
    iteration: int = 0
    viewpoint_stack: list = field(default_factory=list)
    rng: random.Random = field(default_factory=lambda: random.Random(0))
    last_loss: Optional[torch.Tensor] = None
    test_cameras: list = field(default_factory=list)   # scene.getTestCameras() (only an --eval init directory has any)
    _trainer: object = None
    _prepared: object = None   # the NEXT iteration, host half done and forward + backward enqueued: (saved host state, arguments,
                               # result slot, event, camera), see _fused_synced_iteration


def setup_training_from_init(scene, device, opt: OptimizationParams | None = None, pipe: PipelineParams | None = None,
                             model: ModelParams | None = None) -> TrainState:
    """The state reference train.py:90-118 builds from an init directory (`scene_io.load_init_scene`): Gaussians from the point
    cloud with the camera-based extent (scene/__init__.py:94-100), poses from the COLMAP extrinsics, the per-point or the plain
    optimizer, the loaded images as ground truth — no teacher, nothing synthetic."""
    opt = opt or OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)
    pipe = pipe or PipelineParams()
    model = model or ModelParams()
    dev = torch.device(device)
    bg = torch.tensor([1.0, 1.0, 1.0] if model.white_background else [0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
    g = GaussianModel(model.sh_degree)
    g.create_from_pcd(scene.points, scene.colors, scene.cameras_extent, dev, scale_gaussian=scene.scale_gaussian)
    cams = [c.to(dev) for c in scene.cameras]
    g.init_RT_seq(cams, dev)
    if opt.pp_optimizer:
        if scene.confidence_lr is None:   # train.py:95 np.load()s the file unconditionally; the per-point optimizer cannot do without
            raise FileNotFoundError("pp_optimizer needs sparse_<n>/0/confidence_dsp.npy")
        g.training_setup_pp(opt, scene.confidence_lr.to(dev))
    else:
        g.training_setup(opt)
    st = TrainState(g, cams, [c.original_image.to(dev).contiguous() for c in cams], bg, opt, pipe)
    st.test_cameras = [c.to(dev) for c in getattr(scene, "test_cameras", [])]
    st.rng = scene.rng   # the view sampling continues on the stream the camera shuffle drew from (one `random` module in the reference)
    return st


def setup_training(scene, device, opt: OptimizationParams | None = None, pipe: PipelineParams | None = None,
                   model: ModelParams | None = None) -> TrainState:
    """An init directory's scene (`scene_io.InitScene`): see setup_training_from_init.  A synthetic PointmapScene:
    Teacher = create_from_pcd(scene points) at the true poses -> ground-truth images.
    Student = teacher with perturbed positions / colours / poses (what MASt3R noise would look like)."""
    if not isinstance(scene, PointmapScene):
        return setup_training_from_init(scene, device, opt, pipe, model)
    opt = opt or OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)
    pipe = pipe or PipelineParams()
    model = model or ModelParams()
    dev = torch.device(device)
    bg = torch.tensor([1.0, 1.0, 1.0] if model.white_background else [0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
    g = torch.Generator().manual_seed(1234)

    teacher = GaussianModel(model.sh_degree)
    teacher.create_from_pcd(scene.points, scene.colors, scene.extent, dev)
    teacher.init_RT_seq(scene.cameras, dev)
    gts = []
    with torch.no_grad():
        for cam in scene.cameras:
            gts.append(render(cam, teacher, pipe, bg, camera_pose=teacher.get_RT(cam.uid))["render"].clamp(0, 1).detach())

    student = GaussianModel(model.sh_degree)
    noisy_pts = scene.points + 0.01 * torch.randn(scene.points.shape, generator=g)
    noisy_col = (scene.colors + 0.05 * torch.randn(scene.colors.shape, generator=g)).clamp(0, 1)
    student.create_from_pcd(noisy_pts, noisy_col, scene.extent, dev)
    student.init_RT_seq(scene.cameras, dev)
    with torch.no_grad():
        P = student.P.detach().clone()
        dq = scene.pose_noise_q.to(dev)
        P[:, :4] = quadmultiply(dq, P[:, :4])
        P[:, 4:] += scene.pose_noise_t.to(dev)
    student.P = P.requires_grad_(True)
    conf_lr = confidence_to_lr_modifiers(scene.confidence.to(dev), scale=(1.0, 100.0))
    if opt.pp_optimizer:
        student.training_setup_pp(opt, conf_lr)
    else:
        student.training_setup(opt)
    import copy
    cams = [copy.copy(c).to(dev) for c in scene.cameras]  # per-view constants live on the device, as in the reference
    return TrainState(student, cams, gts, bg, opt, pipe)


def hint_key(st, cam):
    """Key of a training view in BinningPolicy's process-wide tables: the view of THIS scene (two training states in one
    process — two scenes, or a state and its copy — must not read each other's instance counts)."""
    return ("train", id(st.gaussians), cam.uid)


def _pick_camera(st: TrainState):
    """reference train.py:152-157: pop a random view from the stack, refilling it when empty."""
    if not st.viewpoint_stack:
        st.viewpoint_stack = list(st.cameras)
    return st.viewpoint_stack.pop(st.rng.randint(0, len(st.viewpoint_stack) - 1))


def _forward_backward_step(st: TrainState, fused_loss: bool):
    """Body of reference train.py:140-211 without the host read-back of the loss."""
    st.iteration += 1
    it, g, opt = st.iteration, st.gaussians, st.opt
    g.update_learning_rate(it)
    if not opt.optim_pose:
        g.P.requires_grad_(False)
    if it % 1000 == 0:
        g.oneupSHdegree()
    cam = _pick_camera(st)
    pose = g.get_RT(cam.uid)
    bg = torch.rand(3, device=st.background.device) if opt.random_background else st.background
    with binning_hint(hint_key(st, cam), tag=it):
        pkg = render(cam, g, st.pipe, bg, camera_pose=pose)
    image = pkg["render"]
    gt = st.gt_images[cam.uid]
    if fused_loss is True:
        loss, _ = fused_l1_ssim_loss(image.unsqueeze(0), gt.unsqueeze(0), opt.lambda_dssim)
    else:
        # train.py:171-176 as written: l1_loss + fused_ssim + scalar arithmetic.  fused_loss=False: utils/loss_utils.l1_loss is this
        # package's drop-in (loss_utils.py: one HIP node); fused_loss="torch": the reference's own PyTorch expression for it
        Ll1 = l1_loss(image, gt) if fused_loss is False else torch.abs((image - gt)).mean()
        loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - fused_ssim(image.unsqueeze(0), gt.unsqueeze(0)))
    loss.backward()
    # (train.py:188 calls `.item()` on `loss` itself: with the loss lines as written that is a LazyScalar whose item() reads the
    # value from pinned host memory without waiting for the backward, lazy_loss.py — a detached copy would take the ordinary read)
    return loss if type(loss) is LazyScalar else loss.detach()


def _optimizer_step(st: TrainState):
    with torch.no_grad():
        if st.iteration < st.opt.iterations:
            st.gaussians.optimizer.step()
            st.gaussians.optimizer.zero_grad(set_to_none=True)


def _host_state(st: TrainState, tr):
    """Everything the host half of an iteration (FusedTrainer.prepare) changes"""
    return (st.iteration, list(st.viewpoint_stack), st.rng.getstate(), st.gaussians.active_sh_degree,
            [st.gaussians.optimizer.state[p]["step"] for p in tr.params], [g["lr"] for g in st.gaussians.optimizer.param_groups])


def _restore_host_state(st: TrainState, tr, saved):
    st.iteration, st.viewpoint_stack, st.gaussians.active_sh_degree = saved[0], list(saved[1]), saved[3]
    st.rng.setstate(saved[2])
    for p, s0 in zip(tr.params, saved[4]):
        st.gaussians.optimizer.state[p]["step"] = s0
    for g, lr in zip(st.gaussians.optimizer.param_groups, saved[5]):
        g["lr"] = lr


def cancel_prepared(st: TrainState):
    """The synced one-call loop runs the host half of iteration t + 1 while the device works on iteration t.  Whoever takes
    the state elsewhere (the autograd path, RunAhead, a checkpoint, a test reading st.iteration) first takes that half back."""
    pre = getattr(st, "_prepared", None)
    if pre is not None:
        st._prepared = None
        _restore_host_state(st, st._trainer, pre[0])


def release_trainer(st: TrainState):
    """Leave the synced one-call loop: take back the prepared half iteration and close the handle (one handle at a time writes
    the parameters, include/mi355gs.h)."""
    cancel_prepared(st)
    if getattr(st, "_trainer", None) is not None:
        st._trainer.close()
        st._trainer = None


WAIT_WITH_EVENT = False   # A/B switch: True = an event recorded behind every iteration's backward (the first form of the loop)
_WAIT = {"timeout_us": 50_000, "ema_us": None}   # spin limit of _wait_for_words and the moving average of the waits it is scaled from
_NOT_YET = -2   # as float32 a NaN with a payload no arithmetic produces; as a count impossible


def _wait_for_words(words: torch.Tensor, dev):
    """Until the kernels of an iteration have stored its loss and instance count into `words` (pinned, device-mapped host
    memory preset to _NOT_YET).  Polled rather than waited for with an event: an event is a marker packet on the queue between
    the iteration's last kernel and the next one (~7 us of idle device per iteration, tools/gap_analysis.py), and a stream
    synchronize would also wait for the NEXT iteration's forward + backward, which are already enqueued."""
    ext = _lib.compiled()
    if ext is not None:
        # the spin is bounded by a few times what an iteration has been taking (a frame of 1 M Gaussians at 1080p takes milliseconds,
        # a shared GPU more): past that the fallback — a stream synchronize, which also waits for iteration t + 1's forward and
        # backward — is the exception it is meant to be, not a per-iteration cost
        t0 = time.perf_counter()
        ext.wait_for_words(words, _NOT_YET, _device_token(dev), _WAIT["timeout_us"])
        dt_us = 1e6 * (time.perf_counter() - t0)
        _WAIT["ema_us"] = dt_us if _WAIT["ema_us"] is None else 0.9 * _WAIT["ema_us"] + 0.1 * dt_us
        _WAIT["timeout_us"] = int(min(max(50_000, 8 * _WAIT["ema_us"]), 5_000_000))
        return
    for spin in range(200_000):
        if int(words[0]) != _NOT_YET and int(words[1]) != _NOT_YET:
            return
    torch.cuda.current_stream(dev).synchronize()
    if int(words[0]) == _NOT_YET or int(words[1]) == _NOT_YET:
        raise RuntimeError("mi355gs: the step did not report its results")


_DEVICE_TOKEN = {}


def _device_token(dev):
    """a tensor that names the device (and through it the current stream) for the binding"""
    t = _DEVICE_TOKEN.get(dev)
    if t is None:
        t = _DEVICE_TOKEN[dev] = torch.empty(1, device=dev)
    return t


def _fused_synced_iteration(st: TrainState):
    """The reference's loop shape — loss read back on the host every iteration — on the one-call fused step, with the device
    never waiting for the host.  An iteration is enqueued in two parts: forward + backward (mi355gs_trainer_step with the
    optimizer deferred) and the optimizer launch (mi355gs_trainer_optimizer_step), which carries the sticky device-side commit
    gate: it writes nothing when the frame's instance count exceeded the buffers (and neither does any launch behind it), so
    nothing enqueued depends on the host having seen a count.  A call for iteration t
      1. enqueues iteration t's optimizer launch (its forward + backward were enqueued by the previous call — or are now),
      2. runs the host half of iteration t + 1 (LR schedule, view sampling, step counts: FusedTrainer.prepare) and enqueues its
         forward + backward — they read the parameters iteration t's update leaves, in stream order, and write only scratch,
      3. waits until iteration t's kernels have stored its loss and count into pinned host memory (polling: no event, no
         stream synchronize) and reads them.
    When the call returns, the state is iteration t's (parameters and moments after its update, in stream order); what is in
    flight for t + 1 has changed nothing but the handle's scratch, and `cancel_prepared` takes its host half back when the loop
    is left.  An iteration that overflowed is redone on the exact-sizing autograd path: the device discarded its update."""
    tr = getattr(st, "_trainer", None)
    if tr is None:
        st._prepared = None
        with torch.no_grad():
            for cam in st.cameras:  # exact instance counts of every view size the fixed buffers
                with binning_hint(hint_key(st, cam)):
                    render(cam, st.gaussians, st.pipe, st.background, camera_pose=st.gaussians.get_RT(cam.uid))
        need = max(BinningPolicy.known[hint_key(st, c)] for c in st.cameras)
        tr = st._trainer = FusedTrainer(st, int(BinningPolicy.slack * need) + BinningPolicy.pad)
        # loss and instance count are stored by the kernels that produce them straight into words of pinned, device-mapped
        # host memory (count as int32: a float32 detour would round counts above 2^24, reachable at 1 M Gaussians / 1080p, and
        # could hide an overflow of a few instances): the iteration's read-back is a poll of these words, not a copy.  Two pairs:
        # iteration t + 1 writes its own while iteration t's are still to be read.
        st._host_words = torch.zeros(4, dtype=torch.int32, pin_memory=(tr.dev.type == "cuda"))
        st._loss_slots = [st._host_words[0:1].view(torch.float32), st._host_words[2:3].view(torch.float32)]
        st._count_slots = [st._host_words[1:2], st._host_words[3:4]]
    cuda = tr.dev.type == "cuda"

    def enqueue_forward_backward(slot):
        saved = _host_state(st, tr)
        args = tr.prepare()
        st._host_words[2 * slot: 2 * slot + 2] = _NOT_YET      # the two kernels overwrite these; the wait below polls them
        cam = tr.launch(args, st._loss_slots[slot], st._count_slots[slot], defer_optimizer=True)
        ev = None
        if cuda and WAIT_WITH_EVENT:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(tr.dev))
        return saved, args, slot, ev, cam

    pre = getattr(st, "_prepared", None)
    st._prepared = None
    saved, args, slot, ev, cam = pre if pre is not None else enqueue_forward_backward(0)
    tr.apply_optimizer(gated=True)            # (no launch on the run's last iteration: prepare() left nothing pending)
    nxt = None
    if args["it"] < st.opt.iterations:        # (the run's last iteration has no successor)
        nxt = enqueue_forward_backward(slot ^ 1)
    if ev is not None:
        ev.synchronize()
    elif cuda:
        _wait_for_words(st._host_words[2 * slot: 2 * slot + 2], tr.dev)
    loss, r = float(st._loss_slots[slot][0]), int(st._count_slots[slot][0])
    BinningPolicy.known[hint_key(st, cam)] = int(r)
    if r > tr.capacity:   # dropped instances (the device left the update out): redo exactly, and grow the buffers for the next iterations
        _restore_host_state(st, tr, saved)
        tr.close()
        st._trainer = None
        return None
    st._prepared = nxt
    if r * 1.2 + 1024 > tr.capacity:
        cancel_prepared(st)
        tr.close()
        st._trainer = None
    return loss


def train_iteration(st: TrainState, fused_loss: bool = True, sync_loss: bool = True, fused_step: bool = False):
    """One pass of reference train.py:140-211, with the reference's per-iteration `loss.item()` (sync_loss=True).
    fused_step=True takes the one-call library step when the configuration allows it (same results)."""
    if fused_step and sync_loss and fused_loss is True and FusedTrainer.supported(st):
        out = _fused_synced_iteration(st)
        if out is not None:
            st.last_loss = out
            return out
    cancel_prepared(st)
    loss = _forward_backward_step(st, fused_loss)
    out = loss.item() if sync_loss else loss
    _optimizer_step(st)
    st.last_loss = out
    return out


class FusedTrainer:
    """Whole iteration in one library call (mi355gs_trainer_step): same kernels as the op-by-op path, no autograd
    graph, no temporaries, 11 launches.  Used by RunAhead whenever the configuration is the one the reference's
    scripts run (SH colours of any active degree, scale/rotation covariance, PerPointAdam with pose optimisation)."""

    ORDER = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")

    @staticmethod
    def supported(st: TrainState) -> bool:
        g, o, p = st.gaussians, st.opt, st.pipe
        if g.max_sh_degree != 3 or not o.optim_pose or p.debug or p.compute_cov3D_python or p.convert_SHs_python:
            return False
        if not isinstance(g.optimizer, PerPointAdam) or len(g.optimizer.param_groups) != 7:
            return False
        grp = g.optimizer.param_groups
        if any(x["weight_decay"] != 0 or x["betas"] != grp[0]["betas"] or x["eps"] != grp[0]["eps"] for x in grp):
            return False
        if [x["name"] for x in grp] != ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "pose"]:
            return False
        # The library gets raw pointers: everything it will index with the trainer's fixed (W, H, device) must really have
        # that shape, dtype, layout and device, or a mismatched view becomes an out-of-bounds read instead of an error.
        dev = g._xyz.device
        W, H = int(st.cameras[0].image_width), int(st.cameras[0].image_height)
        ok = lambda t, shape: (isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.is_contiguous() and t.device == dev
                               and tuple(t.shape) == shape)
        for cam in st.cameras:
            if int(cam.image_width) != W or int(cam.image_height) != H or not ok(cam.projection_matrix, (4, 4)):
                return False
            if cam.uid >= len(st.gt_images) or not ok(st.gt_images[cam.uid], (3, H, W)):
                return False
        return ok(st.background, (3,))

    def __init__(self, st: TrainState, capacity: int):
        self.st, self.capacity = st, int(capacity)
        g = st.gaussians
        L = _lib.lib()
        self.params = [getattr(g, n) for n in self.ORDER]
        dev = _lib.require_device(*[p.data for p in self.params])
        self.dev = dev
        for p in self.params:  # same lazy state creation as PerPointAdam.step
            s = g.optimizer.state[p]
            if len(s) == 0:
                s["step"], s["exp_avg"], s["exp_avg_sq"] = 0, torch.zeros_like(p), torch.zeros_like(p)
        P, V = g._xyz.shape[0], g.P.shape[0]
        cam = st.cameras[0]
        self.W, self.H = int(cam.image_width), int(cam.image_height)
        nbytes = L.mi355gs_trainer_workspace_bytes(P, self.W, self.H, V, self.capacity)
        self.workspace = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        PTR = ctypes.c_void_p * 7
        m = PTR(*[g.optimizer.state[p]["exp_avg"].data_ptr() for p in self.params])
        v = PTR(*[g.optimizer.state[p]["exp_avg_sq"].data_ptr() for p in self.params])
        pplr = g.optimizer.param_groups[0].get("per_point_lr")
        self._keep = (m, v, pplr)
        self.handle = L.mi355gs_trainer_create(P, self.W, self.H, V, self.capacity, *[_lib.ptr(p.data) for p in self.params], m, v,
                                               _lib.ptr(pplr), _lib.ptr(self.workspace))
        if not self.handle:
            raise RuntimeError("mi355gs_trainer_create failed")
        self.num_rendered = torch.zeros(1, dtype=torch.int32, device=dev)
        # Asynchronous verification: the step's instance count is stored by the tile-scan kernel STRAIGHT into a slot of this
        # ring of pinned host memory (device-mapped), so no device-to-host copy — a 4-5 us blit kernel on the training stream,
        # once per iteration — is enqueued for it.  A slot is reused after COUNT_RING steps; RunAhead verifies every `window` steps.
        self._count_ring = torch.zeros(self.COUNT_RING, dtype=torch.int32, pin_memory=(dev.type == "cuda"))
        self._count_next = 0

    def close(self):
        handle, self.handle = getattr(self, "handle", None), None
        if handle:
            try:
                _lib.lib().mi355gs_trainer_destroy(ctypes.c_void_p(handle))
            except Exception:  # interpreter shutdown: module globals may already be gone (the handle is host memory only)
                pass

    def gradients(self) -> dict:
        """Copies of the gradients the last `step` left in the workspace, keyed like the GaussianModel attributes
        (what `.grad` holds on the autograd path) — for tests and diagnostics."""
        L, base, out = _lib.lib(), self.workspace.data_ptr(), {}
        for k, (name, p) in enumerate(zip(self.ORDER, self.params)):
            addr = L.mi355gs_trainer_grad(ctypes.c_void_p(self.handle), k)
            off, nbytes = int(addr) - base, 4 * p.numel()
            assert addr and 0 <= off and off + nbytes <= self.workspace.numel(), (name, off)
            out[name] = self.workspace[off:off + nbytes].view(torch.float32).view(p.shape).clone()
        return out

    __del__ = close

    COUNT_RING = 256

    def prepare(self):
        """Host half of an iteration (reference train.py:140-157 + the optimizer's bookkeeping): advances the iteration counter,
        the LR schedule, the SH degree, the view sampling and the optimizer's step counts, and returns the ready-made arguments
        of the library call.  Nothing is enqueued: `launch()` does that — possibly later, while the host half of the NEXT
        iteration overlaps it (see _fused_synced_iteration)."""
        st = self.st
        st.iteration += 1
        it, g, opt = st.iteration, st.gaussians, st.opt
        g.update_learning_rate(it)
        if it % 1000 == 0:
            g.oneupSHdegree()
        cam = _pick_camera(st)
        bg = torch.rand(3, device=self.dev) if opt.random_background else st.background
        do_opt = it < opt.iterations
        grp = g.optimizer.param_groups
        steps = []
        for p in self.params:
            s = g.optimizer.state[p]
            if do_opt:
                s["step"] += 1
            steps.append(max(s["step"], 1))
        b1, b2 = grp[0]["betas"]
        F7, I7 = ctypes.c_float * 7, ctypes.c_int32 * 7
        return dict(it=it, cam=cam, bg=bg, do_opt=do_opt, sh=int(g.active_sh_degree), lr=F7(*[float(x["lr"]) for x in grp]), steps=I7(*steps),
                    b1=float(b1), b2=float(b2), eps=float(grp[0]["eps"]), lam=float(opt.lambda_dssim))

    def launch(self, a, loss_slot: torch.Tensor, count_out: torch.Tensor, defer_optimizer: bool = False):
        cam = a["cam"]
        with _lib.on_device(self.dev):
            _lib.check(_lib.lib().mi355gs_trainer_step(
                ctypes.c_void_p(self.handle), _lib.stream_ptr(self.dev), int(cam.uid), a["sh"], _lib.ptr(self.st.gt_images[cam.uid]),
                _lib.ptr(cam.projection_matrix), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), _lib.ptr(a["bg"]),
                a["lr"], a["steps"], a["b1"], a["b2"], a["eps"], a["lam"],
                1 if (a["do_opt"] and not defer_optimizer) else 0, _lib.ptr(loss_slot), _lib.ptr(count_out)), "trainer_step")
        self._pending_opt = (a["lr"], a["steps"], a["b1"], a["b2"], a["eps"]) if a["do_opt"] else None
        return cam

    def step(self, loss_slot: torch.Tensor, defer_optimizer: bool = False, verify_async: bool = True, record_event: bool = True,
             count_out: torch.Tensor | None = None):
        """One iteration of reference train.py:140-211; the loss lands in `loss_slot` (device float[1]).
        defer_optimizer: stop after backward; `apply_optimizer()` then commits the update (or the caller discards it).
        record_event=False: the caller synchronises with the stream itself before it polls the counts (RunAhead's read-back of
        the loss ring does), so no event is recorded behind the step.
        count_out (with verify_async=False): where the step's instance count goes instead of the handle's device word."""
        a = self.prepare()
        if verify_async:
            if len(BinningPolicy.pending) >= self.COUNT_RING:
                raise RuntimeError("more unverified frames than count slots: call BinningPolicy.poll() at least every "
                                   f"{self.COUNT_RING} steps")
            k = self._count_next
            self._count_next = (k + 1) % self.COUNT_RING
            count_out = self._count_ring[k:k + 1]
        elif count_out is None:
            count_out = self.num_rendered
        cam = self.launch(a, loss_slot, count_out, defer_optimizer)
        if not verify_async:
            return cam
        # asynchronous verification of the instance count (same bookkeeping as the bounded BinningPolicy): the count is in its
        # pinned slot once the event recorded behind the step has completed
        with binning_hint(hint_key(self.st, cam), tag=a["it"]):
            BinningPolicy.defer(count_out, self.capacity, self.dev, event=record_event)
        return cam

    def apply_optimizer(self, gated: bool = False):
        """The optimizer launch of the last `launch(..., defer_optimizer=True)`.  gated: the host has not seen that frame's
        instance count — the launch carries the device-side commit gate (include/mi355gs.h)."""
        if self._pending_opt is not None:
            lr, steps, b1, b2, eps = self._pending_opt
            with _lib.on_device(self.dev):
                _lib.check(_lib.lib().mi355gs_trainer_optimizer_step(ctypes.c_void_p(self.handle), _lib.stream_ptr(self.dev), lr, steps,
                                                                     b1, b2, eps, 1 if gated else 0), "trainer_optimizer_step")
            self._pending_opt = None


# ---- run-ahead variant: same arithmetic, no host synchronisation inside the iteration ------------------------
class RunAhead:
    """Drives `train_iteration` without per-iteration host syncs and with results identical to the synchronous loop.

    * The reference reads `loss.item()` every iteration only to maintain an EMA it displays every 10 iterations
      (train.py:188-191).  Here each loss is written to a device ring buffer and the EMA is evaluated from it when
      the window is read back (same values, one D2H copy per `window` iterations).
    * The rasterizer's instance buffers are sized from the last verified count of the same view ("bounded"
      BinningPolicy).  At each window boundary all counts of the window are verified; if any frame overflowed, the
      parameters, optimizer state, RNG and view stack are restored from the snapshot taken at the previous boundary
      and the window is replayed with exact sizing — so an overflow costs time, never correctness.
    * On the one-call step there is nothing to restore on the device: its commit gate is sticky (include/mi355gs.h) — the
      optimizer launch of an overflowed iteration and of every iteration enqueued behind it writes nothing, so parameters and
      moments ARE the ones the iteration before the overflow left.  The host only rewinds its own half (iteration counter,
      LR, view stack, RNG, step counts: a tuple kept per iteration of the window) to that iteration and redoes the rest of the
      window with exact sizing.  (The snapshot was 21 tensor copies — ~57 MB at the bench size — per window: 20 us per iteration.)
    """

    def __init__(self, st: TrainState, window: int = 10, fused_loss: bool = True, fused_step: bool = True):
        self.st, self.window, self.fused = st, window, fused_loss
        self.fused_step, self.trainer = fused_step, None
        cancel_prepared(st)
        dev = st.background.device
        self.ring = torch.zeros(window, dtype=torch.float32, device=dev)
        self.ema = 0.0
        self.n_in_window = 0
        self.replays = 0
        self.partial_replays = 0     # ... of which began inside a window (the iterations before the overflow stood)
        BinningPolicy.reset("bounded")
        self.snap, self._host_hist = None, []
        if fused_step and FusedTrainer.supported(st):
            self._make_trainer()
        if self.trainer is None:
            self._snapshot()

    def _make_trainer(self):
        """Instance capacity from an exact count of every training view (one un-timed forward each)."""
        st = self.st
        with torch.no_grad():
            for cam in st.cameras:
                if hint_key(st, cam) not in BinningPolicy.known:
                    mode, BinningPolicy.mode = BinningPolicy.mode, "exact"
                    with binning_hint(hint_key(st, cam)):
                        render(cam, st.gaussians, st.pipe, st.background, camera_pose=st.gaussians.get_RT(cam.uid))
                    BinningPolicy.mode = mode
        need = max(BinningPolicy.known[hint_key(st, c)] for c in st.cameras)
        if self.trainer is not None:
            self.trainer.close()
        self.trainer = FusedTrainer(st, int(BinningPolicy.slack * need) + BinningPolicy.pad)

    def _tensors(self):
        g = self.st.gaussians
        return [g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation, g.P]

    def _snapshot(self):
        st, g = self.st, self.st.gaussians
        opt_state = []
        for p in self._tensors():
            s = g.optimizer.state.get(p, {})
            opt_state.append(None if not s else (s["step"], s["exp_avg"].clone(), s["exp_avg_sq"].clone()))
        self.snap = dict(params=[p.detach().clone() for p in self._tensors()], opt=opt_state, iteration=st.iteration,
                         stack=list(st.viewpoint_stack), rng=st.rng.getstate(), sh=g.active_sh_degree, ema=self.ema,
                         lrs=[grp["lr"] for grp in g.optimizer.param_groups])

    def _restore(self):
        st, g, sn = self.st, self.st.gaussians, self.snap
        with torch.no_grad():
            for p, v, o in zip(self._tensors(), sn["params"], sn["opt"]):
                p.copy_(v)
                p.grad = None
                if o is None:
                    g.optimizer.state.pop(p, None)
                else:
                    s = g.optimizer.state[p]
                    s["step"] = o[0]
                    s["exp_avg"].copy_(o[1])
                    s["exp_avg_sq"].copy_(o[2])
        st.iteration, st.viewpoint_stack, g.active_sh_degree, self.ema = sn["iteration"], list(sn["stack"]), sn["sh"], sn["ema"]
        st.rng.setstate(sn["rng"])
        for grp, lr in zip(g.optimizer.param_groups, sn["lrs"]):
            grp["lr"] = lr

    def step(self):
        """One training iteration; returns the EMA loss at window boundaries (like the reference's progress bar), else None."""
        if self.trainer is not None and FusedTrainer.supported(self.st):
            if self.snap is not None and self.n_in_window == 0:
                self.snap = None       # (a window that began on the autograd path ended: the sticky gate takes over)
            self._host_hist.append(_host_state(self.st, self.trainer))
            # (flush() reads the loss ring back — a blocking copy on this stream — before it polls the counts: no events needed)
            self.trainer.step(self.ring[self.n_in_window:self.n_in_window + 1], record_event=False)
        else:
            if self.snap is None:      # the configuration left the one-call step's domain mid-window: this path needs a state to return to
                self.flush()
                self._snapshot()
            loss = _forward_backward_step(self.st, self.fused)
            # (a detached value: with the loss lines as written `loss` is a LazyScalar whose materialised tensor requires grad after
            # the backward — written as it is, the persistent ring would become a non-leaf and chain every iteration's graph onto itself)
            self.ring[self.n_in_window] = loss.detach()
            _optimizer_step(self.st)
        self.n_in_window += 1
        if self.n_in_window == self.window:
            return self.flush()
        return None

    def flush(self):
        """Verify the window (replaying it with exact sizing if a frame overflowed), fold its losses into the EMA."""
        n = self.n_in_window
        if n == 0:
            return self.ema
        losses = self.ring[:n].tolist()           # the only blocking read-back of the window
        overflowed = BinningPolicy.poll(block=True)
        if overflowed:
            self.replays += 1
            first = 0
            if len(self._host_hist) == n:
                # one-call steps only: everything from the first overflowed iteration on discarded itself on the device
                its = [h[0] + 1 for h in self._host_hist]            # the iteration number each step of the window ran as
                first = min(its.index(t) for t in overflowed if t in its) if any(t in its for t in overflowed) else 0
                _restore_host_state(self.st, self.trainer, self._host_hist[first])
                losses = losses[:first]
                self.partial_replays += first > 0
            else:
                self._restore()
                losses = []
            BinningPolicy.mode = "exact"
            for _ in range(n - first):
                l = _forward_backward_step(self.st, self.fused)
                losses.append(float(l.item()))
                _optimizer_step(self.st)
            BinningPolicy.mode = "bounded"
            if self.trainer is not None:
                # the replay rewrote parameters and moments behind the handle's back (it is their only writer otherwise,
                # include/mi355gs.h): start from a fresh handle, sized for the counts the replay has just verified
                self._make_trainer()
        for l in losses:
            self.ema = 0.4 * l + 0.6 * self.ema   # reference train.py:188
        self.st.last_loss = losses[-1]
        self.n_in_window = 0
        self._host_hist = []
        if self.trainer is not None:  # grow the fixed-capacity buffers before the scene outgrows them
            need = max(BinningPolicy.known.get(hint_key(self.st, c), 0) for c in self.st.cameras)
            if need * 1.2 + 1024 > self.trainer.capacity:
                self._make_trainer()
        if self.trainer is None:
            self._snapshot()
        return self.ema


@torch.no_grad()
def evaluate_psnr(st: TrainState) -> float:
    vals = []
    for cam in st.cameras:
        img = render(cam, st.gaussians, st.pipe, st.background, camera_pose=st.gaussians.get_RT(cam.uid))["render"].clamp(0, 1)
        vals.append(psnr(img, st.gt_images[cam.uid]).mean())
    return float(torch.stack(vals).mean())


@torch.no_grad()
def training_report(st: TrainState, iteration: int, testing_iterations=(), quiet: bool = False) -> dict:
    """reference train.py:253-295 (called at the run's last iteration, :218): at a testing iteration (or every 5000th) mean L1 and
    PSNR of the clamped renders over the test cameras and over the training cameras, printed in the reference's words.
    -> {"test": (l1, psnr), "train": (l1, psnr)} (an empty dict at any other iteration).  No tensorboard writer: the reference
    makes one only when the package is installed, and logs the same numbers to it.  Like the reference, the test cameras' poses
    come from `gaussians.get_RT_test`, which nothing in the reference ever fills (scene/gaussian_model.py:138-140 reads
    `test_P`): with --eval test cameras this raises the reference's own TypeError."""
    out = {}
    if not (iteration in set(int(i) for i in testing_iterations) or iteration % 5000 == 0):
        return out
    g = st.gaussians
    for name, cams in (("test", st.test_cameras), ("train", [st.cameras[i % len(st.cameras)] for i in range(len(st.cameras))])):
        if not cams:
            continue
        l1_test = psnr_test = 0.0
        for cam in cams:
            pose = g.get_RT(cam.uid) if name == "train" else g.get_RT_test(cam.uid)
            image = torch.clamp(render(cam, g, st.pipe, st.background, camera_pose=pose)["render"], 0.0, 1.0)
            gt_image = st.gt_images[cam.uid] if name == "train" else cam.original_image   # (a camera's own `original_image`, as loaded)
            gt = torch.clamp(gt_image.to(image.device), 0.0, 1.0)
            l1_test += l1_loss(image, gt).mean().double()
            psnr_test += psnr(image, gt).mean().double()
        out[name] = (float(l1_test / len(cams)), float(psnr_test / len(cams)))
        if not quiet:
            print("\n[ITER {}] Evaluating {}: L1 {} PSNR {}".format(iteration, name, out[name][0], out[name][1]))
    return out


def _save_outputs(st: TrainState, iteration: int, model_path: str, colmap_ids):
    """What the reference writes at a saving iteration (train.py:220-223, scene/__init__.py:97-99): the Gaussians as
    point_cloud/iteration_<it>/point_cloud.ply and the optimised poses as pose/ours_<it>/pose_optimized.npy."""
    import os
    from .io_formats import save_pose
    st.gaussians.save_ply(os.path.join(model_path, "point_cloud", f"iteration_{iteration}", "point_cloud.ply"))
    os.makedirs(os.path.join(model_path, "pose", f"ours_{iteration}"), exist_ok=True)
    save_pose(os.path.join(model_path, "pose", f"ours_{iteration}", "pose_optimized.npy"), st.gaussians.P, colmap_ids)


def training(scene, device, iterations: int = 1000, log_every: int = 0, run_ahead: bool = True,
             fused_loss: bool = True, model_path: str | None = None, saving_iterations=(), checkpoint_iterations=(),
             start_checkpoint: str | None = None, opt: OptimizationParams | None = None, n_views: int | None = None,
             model: ModelParams | None = None, after_setup=None, resolution=1, testing_iterations=()) -> dict:
    """Train one scene.  run_ahead=True (default): the fastest loop with the reference's results — the one-call step with every
    iteration's loss read back while the device already works on the next iteration (`train_iteration(fused_step=True)`,
    _fused_synced_iteration) for the configuration the reference's scripts run, the window-verified RunAhead driver on the
    autograd operators otherwise.  run_ahead=False: the reference's own loop shape on the drop-in operators (autograd,
    `loss.item()`, `optimizer.step()` per iteration).

    scene: a synthetic PointmapScene, a `scene_io.InitScene`, or the path of an init directory (`-s <source_path>` with
    `n_views` and `-r <resolution>`; loaded with `scene_io.load_init_scene`, which also leaves input.ply and cameras.json in
    `model_path` like the reference's Scene).  opt: the optimisation parameters (default: the scripts' `--pp_optimizer --optim_pose`); its
    `iterations` is overridden by the argument.  after_setup(state): hook between set-up and the first iteration (tests).

    testing_iterations: reference --test_iterations; `training_report` runs at the LAST iteration if it is one of them (train.py:218)
    and its numbers come back under "report".
    model_path / saving_iterations / checkpoint_iterations / start_checkpoint: the reference's outputs and resume
    (train.py:103-110,220-227): pose/ours_<it>/pose_org.npy before training, point_cloud.ply + pose_optimized.npy at every saving
    iteration, chkpnt<it>.pth = torch.save((gaussians.capture(), it)) at every checkpoint iteration, and a run started from
    such a file continues at its iteration with its optimizer state."""
    import os
    from .io_formats import save_pose, save_time
    import dataclasses
    opt = dataclasses.replace(opt, iterations=iterations) if opt is not None else OptimizationParams(iterations=iterations, pp_optimizer=True, optim_pose=True)
    if isinstance(scene, (str, os.PathLike)):
        from .scene_io import load_init_scene
        if n_views is None:
            raise ValueError("training(<source_path>) needs n_views (the sparse_<n_views> directory to read)")
        # the ModelParams fields that decide WHAT is read (reference arguments/__init__.py:47-62, scene/__init__.py:42-44) are
        # honoured here, so that the cfg_args written below describes the scene that was actually loaded
        mp_ = model or ModelParams()
        if resolution == 1 and mp_.resolution not in (None, 1, -1):
            resolution = mp_.resolution
        scene = load_init_scene(os.fspath(scene), n_views, images=(None if mp_.images in (None, "", "images") else mp_.images), eval=bool(mp_.eval),
                                resolution=resolution, device=device, model_path=model_path,
                                init_scale_from_view_depth=bool(mp_.init_scale_from_view_depth))
    if model_path:   # reference train.py:233-246 (prepare_output_and_logger): the run's arguments next to its outputs
        os.makedirs(model_path, exist_ok=True)
        from .arguments import cfg_args_text
        mp = dataclasses.replace(model or ModelParams(), model_path=os.fspath(model_path), source_path=getattr(scene, "source_path", None) or "",
                                 n_views=int(getattr(scene, "n_views", n_views) or 0), resolution=resolution)
        saves = sorted(set(int(i) for i in saving_iterations) | {int(iterations)})
        with open(os.path.join(model_path, "cfg_args"), "w") as f:   # a Namespace(...) text: what the reference's render.py evaluates
            f.write(cfg_args_text(mp, opt, PipelineParams(), save_iterations=saves, checkpoint_iterations=sorted(int(i) for i in checkpoint_iterations),
                                  start_checkpoint=start_checkpoint))
    st = setup_training(scene, device, opt=opt, model=model)
    if after_setup is not None:
        after_setup(st)
    first_iter = 0
    if start_checkpoint:
        model_params, first_iter = torch.load(start_checkpoint, map_location=device, weights_only=False)
        # (the reference's restore() falls back to plain Adam here; a run started with --pp_optimizer keeps its multiplier)
        st.gaussians.restore(model_params, opt, confidence_lr=getattr(st.gaussians, "per_point_lr", None))
        st.iteration = int(first_iter)
    colmap_ids = [int(c.colmap_id) for c in st.cameras]
    saving, checkpoints = set(int(i) for i in saving_iterations), set(int(i) for i in checkpoint_iterations)
    if (saving or checkpoints) and not model_path:
        raise ValueError("saving_iterations / checkpoint_iterations need a model_path")
    for it in sorted(saving):   # reference train.py:107-110: the initial poses, once per saving iteration
        os.makedirs(os.path.join(model_path, "pose", f"ours_{it}"), exist_ok=True)
        save_pose(os.path.join(model_path, "pose", f"ours_{it}", "pose_org.npy"), st.gaussians.P, colmap_ids)
    psnr0 = evaluate_psnr(st)
    is_cuda = torch.device(device).type == "cuda"
    # Host housekeeping before the loop: one FULL collection of Python's cyclic collector now, and everything that survives it
    # parked in the permanent generation for the duration of the loop.  A process that has imported torch tracks ~2 x 10^5
    # container objects and a full (generation-2) pass over them takes 70-90 ms — a quarter of a 1000-iteration run of this
    # scene (0.28 ms per iteration) whenever one happens to fall inside it, which is what made `iters_per_sec_1k` read 2650-2830
    # on some runs and 3500-3650 on others (profiles/r06_onek_probe_*.txt: a single stall of 87 ms at a varying iteration,
    # no allocator activity).  With the old objects frozen, the collections that do run inside the loop only walk what the loop
    # itself allocated.  Undone on the way out (the caller's process gets its collector back as it was).
    gc.collect()
    gc.freeze()
    try:   # (whatever happens in the loop, the caller gets its collector back)
        if is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = last = None
        one_call = bool(run_ahead and fused_loss and FusedTrainer.supported(st))
        ra = RunAhead(st, fused_loss=fused_loss) if run_ahead and not one_call else None
        ema = 0.0
        report = {}
        for i in range(int(first_iter), iterations):
            if one_call:
                last = train_iteration(st, fused_step=True)
                first = last if first is None else first
                ema = 0.4 * last + 0.6 * ema   # reference train.py:188
                if log_every and (i + 1) % log_every == 0:
                    print(f"[iter {i + 1}] ema loss {ema:.6f}")
            elif ra is not None:
                ema = ra.step()
                if first is None and ema is not None:
                    first = st.last_loss
                if log_every and ema is not None and (i + 1) % log_every == 0:
                    print(f"[iter {i + 1}] ema loss {ema:.6f}")
            else:
                last = train_iteration(st, fused_loss=fused_loss)
                first = last if first is None else first
                if log_every and (i + 1) % log_every == 0:
                    print(f"[iter {i + 1}] loss {last:.6f}")
            it = i + 1
            if it == iterations and model_path:   # reference train.py:213-217: the time of the loop itself, before the last save
                save_time(model_path, "[2] train_joint_TrainTime", time.perf_counter() - t0)
            if it == iterations and (it in set(int(i) for i in testing_iterations) or it % 5000 == 0):   # train.py:218 (training_report)
                cancel_prepared(st)
                if ra is not None:
                    ra.flush()
                report = training_report(st, it, testing_iterations, quiet=not log_every)
            if it in saving or it in checkpoints:
                cancel_prepared(st)
                if ra is not None:
                    ra.flush()   # the window up to here is verified (and replayed if a frame overflowed) before anything is written
                if it in saving:
                    _save_outputs(st, it, model_path, colmap_ids)
                if it in checkpoints:
                    torch.save((st.gaussians.capture(), it), os.path.join(model_path, f"chkpnt{it}.pth"))
        release_trainer(st)
        if ra is not None:
            ra.flush()
            last = st.last_loss
            BinningPolicy.reset("exact")
        if is_cuda:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        gc.unfreeze()
    # the loss pair of the last iteration (train.py:171-176 as written) holds that frame's image, its render graph and four
    # image-sized buffers until the next l1_loss call: none will come from this loop
    from . import lazy_loss
    lazy_loss.forget()
    if model_path:   # reference train.py:229-231
        save_time(model_path, "[2] train_joint", dt)
    n_done = max(iterations - int(first_iter), 1)
    return dict(seconds=dt, iters_per_sec=n_done / dt, first_loss=first, last_loss=last, psnr_before=psnr0,
                psnr_after=evaluate_psnr(st), state=st, report=report)
