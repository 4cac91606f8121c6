"""Drop-in for the `diff_gaussian_rasterization` package the reference imports at
gaussian_renderer/__init__.py:14-17 and calls at :60-78 (settings) and :126-135 (forward).

Same public names, argument meaning and error behaviour as that operator package
(GaussianRasterizationSettings with its 12 fields in order, GaussianRasterizer.forward returning
`(color[3,H,W], radii[P])`, `markVisible`); the compute is libmi355gs.so (HIP, gfx950).
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn as nn

from .. import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# Roofline accounting hook (bench.py): when enabled, the scratch of the most recent forward is kept
# reachable so mi355gs_raster_frame_stats can be asked for its instance counts.
_KEEP_LAST_FRAME = False
_LAST_FRAME = {}


def keep_last_frame(flag: bool):
    global _KEEP_LAST_FRAME
    _KEEP_LAST_FRAME = bool(flag)
    if not flag:
        _LAST_FRAME.clear()


def last_frame_stats():
    """(R, R_eff) of the most recent forward: instances binned / instances the composite kernels consumed."""
    if not _LAST_FRAME:
        raise RuntimeError("keep_last_frame(True) was not set before the forward")
    tiles, W, H = _LAST_FRAME["tiles"], _LAST_FRAME["W"], _LAST_FRAME["H"]
    stats = torch.zeros(2, dtype=torch.int64, device=tiles.device)
    with _lib.on_device(tiles.device):
        _lib.check(_lib.lib().mi355gs_raster_frame_stats(_lib.stream_ptr(tiles.device), W, H, _lib.ptr(tiles), _lib.ptr(stats)),
                   "raster_frame_stats")
    r, reff = stats.tolist()
    return int(r), int(reff)


def reference_instance_count(means_cam: torch.Tensor, projmatrix: torch.Tensor, radii: torch.Tensor, W: int, H: int) -> int:
    """Roofline accounting only: the number of (tile, Gaussian) instances the PUBLISHED operator's binning holds for this frame —
    its `num_rendered`: every tile of the 3-sigma square around the projected centre (recalled upstream `getRect`: truncating
    `(p - r) / 16` and `(p + r + 15) / 16`, clamped to the grid), summed over the Gaussians with radii > 0.  This library bins
    fewer (csrc/preprocess.hip intersects the square with the box around {alpha >= 1/255}); SURVEY 8(d) counts a kernel's
    algorithmic bytes per instance of the reference formulation.  means_cam: camera-frame centres [P,3]; projmatrix as handed to
    the operator; radii: the operator's output."""
    with torch.no_grad():
        m = means_cam.detach().float()
        hom = torch.cat([m, torch.ones_like(m[:, :1])], dim=1) @ projmatrix.to(m.device).float()
        pw = 1.0 / (hom[:, 3] + 0.0000001)
        px = ((hom[:, 0] * pw + 1.0) * W - 1.0) * 0.5
        py = ((hom[:, 1] * pw + 1.0) * H - 1.0) * 0.5
        r = radii.to(m.device).float()
        gx, gy = (W + 15) // 16, (H + 15) // 16
        x0 = ((px - r) / 16).trunc().clamp(0, gx); x1 = ((px + r + 15) / 16).trunc().clamp(0, gx)
        y0 = ((py - r) / 16).trunc().clamp(0, gy); y1 = ((py + r + 15) / 16).trunc().clamp(0, gy)
        area = ((x1 - x0) * (y1 - y0)).to(torch.int64)
        return int(area[radii.to(m.device) > 0].sum())


class _Pending:
    """One frame rendered without reading its instance count back (BinningPolicy mode "bounded")."""
    __slots__ = ("ev", "slot", "capacity", "key", "tag", "hold", "dev")

    def __init__(self, ev, slot, capacity, key, tag, hold=None, dev=None):
        # ev: a torch.cuda.Event recorded behind the frame's tile scan; None: already complete (CPU / emulated kernels);
        # "stream": no event was recorded — the caller orders the count by synchronising `dev`'s current stream itself
        # (RunAhead's window read-back), so only a BLOCKING poll may look at the slot.
        self.ev, self.slot, self.capacity, self.key, self.tag, self.hold, self.dev = ev, slot, capacity, key, tag, hold, dev


class BinningPolicy:
    """How a frame's instance buffer is sized.

    mode "exact" (default): read the instance count R back after the projection stage — one blocking 4-byte
        D2H copy per forward, exactly what the reference operator does internally — and allocate exactly.
    mode "bounded": no host synchronisation.  For a frame rendered under `binning_hint(key)` whose key has a
        known count, capacity = slack * R_known + pad; R of the new frame is copied to pinned memory
        asynchronously and examined later by `poll()`, which refreshes R_known and reports every frame whose R
        exceeded its capacity (those frames dropped instances and everything computed from them is invalid —
        instantsplat_amd.train rolls back to its last verified snapshot and replays them in exact mode).
        Frames without a hint, or with an unknown key, use the exact path.

    Process-wide state (one policy, one `known` table, one pending list): right for the deployment this package is
    built for — one process per GPU, one scene per process — and a trap for two scenes in one process, which must use
    distinct hint keys and poll together.
    """
    mode = "exact"
    slack = 1.5
    pad = 16384
    known = {}       # key -> last verified R
    pending = []     # _Pending
    current_key = None
    current_tag = None

    @classmethod
    def reset(cls, mode="exact"):
        for e in cls.pending:
            _release_slot(e)
        cls.mode, cls.known, cls.pending, cls.current_key, cls.current_tag = mode, {}, [], None, None

    @classmethod
    def deferred_capacity(cls):
        """Capacity for the frame about to be rendered if its count will NOT be read back (bounded mode, hinted key with a
        known count); None: the exact path."""
        key = cls.current_key
        if cls.mode == "bounded" and key is not None and key in cls.known:
            return int(cls.slack * cls.known[key]) + cls.pad
        return None

    @classmethod
    def defer(cls, slot, capacity, dev, event=True):
        """Queue the frame whose tile scan was just enqueued on `dev`'s current stream for later verification."""
        ev = None
        if dev is not None and dev.type == "cuda":
            if event:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
            else:
                ev = "stream"
        ring = _COUNT_RINGS.get(dev)
        k = ring.reserved.pop(slot.data_ptr(), None) if ring is not None else None
        cls.pending.append(_Pending(ev, slot, capacity, cls.current_key, cls.current_tag, None if k is None else (ring, k), dev))

    @classmethod
    def poll(cls, block: bool = False):
        """Process finished read-backs; returns the tags of frames that overflowed their capacity."""
        bad, keep, synced = [], [], set()
        for e in cls.pending:
            ev = e.ev
            if ev is None:
                pass
            elif ev == "stream":
                if not block:   # nothing orders this count yet: its slot may still hold an older frame's value
                    keep.append(e)
                    continue
                if e.dev not in synced:
                    torch.cuda.current_stream(e.dev).synchronize()
                    synced.add(e.dev)
            elif block:
                ev.synchronize()
            elif not ev.query():
                keep.append(e)
                continue
            r = int(e.slot[0])
            cls.known[e.key] = r
            if r > e.capacity:
                bad.append(e.tag)
            _release_slot(e)
        cls.pending = keep
        return bad


class binning_hint:
    """with binning_hint(key, tag): frames rendered inside may reuse the instance count last seen for `key`
    (e.g. the camera uid) when BinningPolicy.mode == "bounded"; `tag` (e.g. the iteration) labels overflow reports."""

    def __init__(self, key, tag=None):
        self.key, self.tag = key, tag

    def __enter__(self):
        self.prev = (BinningPolicy.current_key, BinningPolicy.current_tag)
        BinningPolicy.current_key, BinningPolicy.current_tag = self.key, self.tag

    def __exit__(self, *a):
        BinningPolicy.current_key, BinningPolicy.current_tag = self.prev


RENDER_ONLY_WHEN_NO_GRAD = True   # A/B switch of this (ctypes) binding; the compiled one has ext.render_only(bool)
_BACKWARD_FOLLOWS = [True]        # set by the callers of the Python nodes around apply(): inside forward() grad mode is off


class backward_follows:
    """with backward_follows(flag): the forward(s) inside know whether a backward can follow them — asked where the operator is
    CALLED (grad mode on and an input requiring a gradient).  flag False selects the render-only stage 2."""

    def __init__(self, *tensors):
        self.flag = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)

    def __enter__(self):
        self.prev = _BACKWARD_FOLLOWS[0]
        _BACKWARD_FOLLOWS[0] = self.flag

    def __exit__(self, *a):
        _BACKWARD_FOLLOWS[0] = self.prev


def size_and_render(L, stream, dev, P, W, H, num_rendered, bg, geom, tiles, color, debug):
    """Stage 2 of the forward for both bindings (operator-level and posed): size the instance buffers according to the
    BinningPolicy — the reference operator's own blocking 4-byte read-back, or a verified bound with no host sync — allocate
    them and enqueue binning + composite.  Returns (capacity handed to the library, binning scratch).  A frame no backward
    can follow (`backward_follows`) takes the render-only stage 2: a binning buffer of keys + lists only."""
    key = BinningPolicy.current_key
    on_gpu = dev.type == "cuda"   # then `num_rendered` is a slot of pinned host memory the tile-scan kernel stores into (count_slot)
    R = BinningPolicy.deferred_capacity()
    if R is not None:
        BinningPolicy.defer(num_rendered, R, dev)   # capacity from the last verified count, no host sync
    else:
        # the reference operator's own blocking read-back of the count — without its device-to-host copy: the kernel has
        # stored the value in host memory, the host only waits for the stream
        if on_gpu:
            torch.cuda.current_stream(dev).synchronize()
        R = int(num_rendered[0])
        if key is not None:
            BinningPolicy.known[key] = R
    if _BACKWARD_FOLLOWS[0] or not RENDER_ONLY_WHEN_NO_GRAD:
        binning = _empty_bytes(L.mi355gs_raster_binning_bytes(R, W, H), dev)
        _lib.check(L.mi355gs_raster_forward_render(stream, P, W, H, R, _lib.ptr(bg), _lib.ptr(geom), _lib.ptr(tiles),
                                                   _lib.ptr(binning), _lib.ptr(color), debug), "raster_forward_render")
    else:
        binning = _empty_bytes(L.mi355gs_raster_binning_bytes_render_only(R, W, H), dev)
        _lib.check(L.mi355gs_raster_forward_render_only(stream, P, W, H, R, _lib.ptr(bg), _lib.ptr(geom), _lib.ptr(tiles),
                                                        _lib.ptr(binning), _lib.ptr(color), debug), "raster_forward_render_only")
    return R, binning


def check_frame_buffers(L, binning, R, W, H):
    """The deterministic-backward mode is process-wide and enters the layout of `binning`: a frame must see the same mode in its
    backward as in its forward.  Switched on in between (a retained graph, a frame in flight, another thread) the backward would
    write per-instance rows past the end of a buffer laid out without them — refused here, on the host, from the sizes."""
    need = int(L.mi355gs_raster_binning_bytes(int(R), W, H))
    if R > 0 and binning.numel() < need:
        raise RuntimeError(f"mi355gs: this frame's binning buffer ({binning.numel()} bytes) is smaller than the backward's layout needs ({need}): "
                           "the deterministic-backward mode was switched on between the frame's forward and its backward")


def _empty_bytes(n: int, device) -> torch.Tensor:
    return torch.empty(max(int(n), 1), dtype=torch.uint8, device=device)


_SIZES = {}
_GRAD_SCRATCH = {}


def grad_scratch_bytes(L, P):
    n = _GRAD_SCRATCH.get(P)
    if n is None:
        if len(_GRAD_SCRATCH) > 64:
            _GRAD_SCRATCH.clear()
        n = _GRAD_SCRATCH[P] = int(L.mi355gs_raster_grad_scratch_bytes(P))
    return n


def set_deterministic(on: bool) -> bool:
    """Deterministic-backward mode (include/mi355gs.h, mi355gs_tune_deterministic): the backward sums every Gaussian's moments in a
    fixed order instead of with float atomics — bit-identical gradients, and with them bit-identical training runs, at the cost
    of a few launches and 52 B per instance per backward.  Process-wide; it enters the scratch sizes, so frames in flight must be
    finished first (call it between iterations) and a one-call trainer handle must be re-created (instantsplat_amd.train does
    that: `release_trainer`).  Also switched on by MI355GS_DETERMINISTIC=1 in the environment.  Returns the previous setting."""
    old = bool(_lib.lib().mi355gs_tune_deterministic(1 if on else 0))
    _GRAD_SCRATCH.clear()
    return old


def frame_buffers(L, P, W, H, dev):
    """Per-frame outputs and workspaces of the forward: (radii, color, geom, tiles, num_rendered).  None is pre-filled: the
    projection kernel writes every radius (0 for a culled Gaussian) and the tile scan writes the instance count, so the two
    memset launches of `torch.zeros` per frame are not needed; the workspace sizes are asked of the library once per shape.
    The count lives in host memory (count_slot)."""
    key = (P, W, H)
    sizes = _SIZES.get(key)
    if sizes is None:
        if len(_SIZES) > 64:
            _SIZES.clear()
        sizes = _SIZES[key] = (max(int(L.mi355gs_raster_geom_bytes(P)), 1), max(int(L.mi355gs_raster_tiles_bytes(W, H)), 1))
    i32, u8 = torch.int32, torch.uint8
    return (torch.empty(P, dtype=i32, device=dev), torch.empty(3, H, W, dtype=torch.float32, device=dev),
            torch.empty(sizes[0], dtype=u8, device=dev), torch.empty(sizes[1], dtype=u8, device=dev), count_slot(dev))


_COUNT_RINGS = {}
COUNT_RING = 256


class _CountRing:
    """Pinned (device-mapped) host words the tile-scan kernel stores a frame's instance count into — no device-to-host copy
    is enqueued to read it.  Two halves with different lifetimes:
      * words [0, COUNT_RING): frames whose count is read back at once (exact mode, inference) — handed out round-robin; the
        value is consumed before the forward returns, so reuse after COUNT_RING forwards is harmless;
      * words [COUNT_RING, 2 * COUNT_RING): frames queued in BinningPolicy.pending — a word stays reserved until poll() has
        read it, however many other forwards run in between."""

    def __init__(self):
        self.words = torch.zeros(2 * COUNT_RING, dtype=torch.int32, pin_memory=True)
        self.views = [self.words[k:k + 1] for k in range(2 * COUNT_RING)]   # made once: slicing a tensor costs ~2 us per frame
        self.values = self.words.numpy()                                    # the same memory, for reading a count without a tensor op
        self.index_of = {v.data_ptr(): k for k, v in enumerate(self.views)}
        self.next = 0
        self.free = list(range(2 * COUNT_RING - 1, COUNT_RING - 1, -1))
        self.reserved = {}   # address of a handed-out reserved word -> its index, until BinningPolicy.defer() takes it over


def _release_slot(entry):
    hold = entry.hold
    if hold is not None:
        ring, k = hold
        ring.free.append(k)
        entry.hold = None


def count_slot(dev):
    """Where the forward leaves the frame's instance count (see _CountRing); on the CPU (emulated kernels) a plain word."""
    if dev.type != "cuda":
        return torch.zeros(1, dtype=torch.int32, device=dev)
    ring = _COUNT_RINGS.get(dev)
    if ring is None:
        ring = _COUNT_RINGS[dev] = _CountRing()
    if BinningPolicy.deferred_capacity() is None:
        k = ring.next
        ring.next = (k + 1) % COUNT_RING
        return ring.views[k]
    if not ring.free:
        raise RuntimeError(f"more unverified frames than count slots: call BinningPolicy.poll() at least every {COUNT_RING} forwards")
    k = ring.free.pop()
    slot = ring.views[k]
    ring.reserved[slot.data_ptr()] = k
    return slot


def read_count(slot) -> int:
    """The value of a count word (after the frame's tile scan is known to have run)."""
    if _COUNT_RINGS:
        ptr = slot.data_ptr()
        for ring in _COUNT_RINGS.values():
            k = ring.index_of.get(ptr)
            if k is not None:
                return int(ring.values[k])
    return int(slot[0])


def _cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                sh_rest=None):
        s = raster_settings
        L = _lib.lib()
        opt = lambda t: None if (t is None or t.numel() == 0) else _lib.f32c(t)
        means3D = _lib.f32c(means3D)
        opac = _lib.f32c(opacities)
        sh_, col_, sc_, rot_, cov_ = opt(sh), opt(colors_precomp), opt(scales), opt(rotations), opt(cov3Ds_precomp)
        shr_ = opt(sh_rest)  # split SH storage: sh = DC [P,1,3], sh_rest = the other coefficients [P,M-1,3]
        bg, view, proj, campos = _lib.f32c(s.bg), _lib.f32c(s.viewmatrix), _lib.f32c(s.projmatrix), _lib.f32c(s.campos)
        dev = _lib.require_device(means3D, opac, sh_, shr_, col_, sc_, rot_, cov_, bg, view, proj, campos)
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        P = means3D.shape[0]
        H, W = int(s.image_height), int(s.image_width)
        M = (sh_.shape[1] + (shr_.shape[1] if shr_ is not None else 0)) if sh_ is not None else 0
        D = int(s.sh_degree)
        stream = _lib.stream_ptr(dev)
        debug = 1 if s.debug else 0

        radii, color, geom, tiles, num_rendered = frame_buffers(L, P, W, H, dev)

        def run():
            _lib.check(L.mi355gs_raster_forward_preprocess(
                stream, P, D, M, W, H, _lib.ptr(means3D), _lib.ptr(sh_), _lib.ptr(shr_), _lib.ptr(col_), _lib.ptr(opac), _lib.ptr(sc_),
                float(s.scale_modifier), _lib.ptr(rot_), _lib.ptr(cov_), _lib.ptr(view), _lib.ptr(proj), _lib.ptr(campos),
                float(s.tanfovx), float(s.tanfovy), int(bool(s.prefiltered)), _lib.ptr(radii), _lib.ptr(geom),
                _lib.ptr(tiles), _lib.ptr(num_rendered), None, None, debug), "raster_forward_preprocess")
            return size_and_render(L, stream, dev, P, W, H, num_rendered, bg, geom, tiles, color, debug)

        if s.debug:
            try:
                with _lib.on_device(dev):
                    R, binning = run()
            except Exception:
                torch.save(_cpu_deep_copy_tuple((means3D, sh_, col_, opac, sc_, rot_, cov_, tuple(s))), "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            with _lib.on_device(dev):
                R, binning = run()

        if _KEEP_LAST_FRAME:
            _LAST_FRAME.update(tiles=tiles, W=W, H=H, geom=geom, binning=binning, capacity=int(R), P=P)
        ctx.raster_settings = s
        ctx.num_rendered = R
        ctx.dims = (P, D, M, W, H)
        ctx.sh_rest = shr_
        ctx.save_for_backward(means3D, sh_ if sh_ is not None else torch.empty(0), col_ if col_ is not None else torch.empty(0),
                              opac, sc_ if sc_ is not None else torch.empty(0), rot_ if rot_ is not None else torch.empty(0),
                              cov_ if cov_ is not None else torch.empty(0), radii, geom, tiles, binning, bg, view, proj, campos, color)
        ctx.mark_non_differentiable(radii)
        ctx.opacity_shape = opacities.shape
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        s = ctx.raster_settings
        L = _lib.lib()
        P, D, M, W, H = ctx.dims
        (means3D, sh_, col_, opac, sc_, rot_, cov_, radii, geom, tiles, binning, bg, view, proj, campos, color) = ctx.saved_tensors
        opt = lambda t: None if t.numel() == 0 else t
        sh_, col_, sc_, rot_, cov_ = opt(sh_), opt(col_), opt(sc_), opt(rot_), opt(cov_)
        dev = means3D.device
        g = _lib.f32c(grad_out_color)
        _lib.require_device(g)
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        dL_dmeans3D, dL_dmeans2D, dL_dopac = new(P, 3), new(P, 3), new(P)
        shr_ = ctx.sh_rest
        dL_dsh = (new(P, M, 3) if shr_ is None else new(P, 1, 3)) if sh_ is not None else None
        dL_dshr = new(P, M - 1, 3) if shr_ is not None else None
        dL_dcol = new(P, 3)
        dL_dscales = new(P, 3) if cov_ is None else None
        dL_drot = new(P, 4) if cov_ is None else None
        dL_dcov = new(P, 6) if cov_ is not None else None
        scratch = _empty_bytes(grad_scratch_bytes(L, P), dev)
        stream = _lib.stream_ptr(dev)
        check_frame_buffers(L, binning, int(ctx.num_rendered), W, H)

        def run():
            _lib.check(L.mi355gs_raster_backward(
                stream, P, D, M, W, H, _lib.ptr(bg), _lib.ptr(means3D), _lib.ptr(sh_), _lib.ptr(shr_), _lib.ptr(col_), _lib.ptr(opac),
                _lib.ptr(sc_), float(s.scale_modifier), _lib.ptr(rot_), _lib.ptr(cov_), _lib.ptr(view), _lib.ptr(proj),
                _lib.ptr(campos), float(s.tanfovx), float(s.tanfovy), _lib.ptr(geom), _lib.ptr(tiles), _lib.ptr(binning),
                int(ctx.num_rendered), _lib.ptr(radii), _lib.ptr(color), _lib.ptr(g), _lib.ptr(scratch), _lib.ptr(dL_dmeans3D),
                _lib.ptr(dL_dmeans2D), _lib.ptr(dL_dsh), _lib.ptr(dL_dshr), _lib.ptr(dL_dcol), _lib.ptr(dL_dopac), _lib.ptr(dL_dscales),
                _lib.ptr(dL_drot), _lib.ptr(dL_dcov), 0, 1 if s.debug else 0), "raster_backward")

        if s.debug:
            try:
                with _lib.on_device(dev):
                    run()
            except Exception:
                torch.save(_cpu_deep_copy_tuple((means3D, sh_, col_, opac, sc_, rot_, cov_, radii, g, tuple(s))), "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            with _lib.on_device(dev):
                run()
        return (dL_dmeans3D, dL_dmeans2D, dL_dsh, dL_dcol if sh_ is None else None, dL_dopac.reshape(ctx.opacity_shape),
                dL_dscales, dL_drot, dL_dcov, None, dL_dshr)


_LAST_COUNT = {}   # (P, W, H, hint key) -> instance count of the last frame like this one (sizes the speculative stage 2)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        sh_rest=None):
    s = raster_settings
    ext = None if (s.debug or _KEEP_LAST_FRAME) else _lib.compiled()
    if ext is None:   # the ctypes / Python autograd.Function binding (also: the operator's debug mode with its snapshot dumps)
        with backward_follows(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, sh_rest):
            return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                             raster_settings, sh_rest)
    # the compiled node (csrc_torch/binding.cpp::RasterizeFn): the same three C-ABI calls, with size_and_render's bookkeeping here
    opt = lambda t: None if (t is None or t.numel() == 0) else t
    dev = means3D.device
    H, W = int(s.image_height), int(s.image_width)
    slot = count_slot(dev)
    cap = BinningPolicy.deferred_capacity()
    key = BinningPolicy.current_key
    ck = (means3D.shape[0], W, H, key)
    color, radii = ext.rasterize(means3D, means2D, opt(sh), opt(colors_precomp), opacities, opt(scales), opt(rotations), opt(cov3Ds_precomp),
                                 opt(sh_rest), s.bg, s.viewmatrix, s.projmatrix, s.campos, H, W, float(s.tanfovx), float(s.tanfovy),
                                 float(s.scale_modifier), int(s.sh_degree), bool(s.prefiltered), -1 if cap is None else cap,
                                 0 if cap is not None else _LAST_COUNT.get(ck, 0), slot)
    if cap is not None:
        BinningPolicy.defer(slot, cap, dev)
    else:
        r = int(slot[0])
        if len(_LAST_COUNT) > 256:
            _LAST_COUNT.clear()
        _LAST_COUNT[ck] = r
        if key is not None:
            BinningPolicy.known[key] = r
    return color, radii


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            s = self.raster_settings
            pos, view, proj = _lib.f32c(positions), _lib.f32c(s.viewmatrix), _lib.f32c(s.projmatrix)
            dev = _lib.require_device(pos, view, proj)
            present = torch.zeros(pos.shape[0], dtype=torch.uint8, device=dev)
            with _lib.on_device(dev):
                _lib.check(_lib.lib().mi355gs_raster_mark_visible(_lib.stream_ptr(dev), pos.shape[0], _lib.ptr(pos), _lib.ptr(view),
                                                                  _lib.ptr(proj), _lib.ptr(present)), "raster_mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, shs_rest=None):
        """Same arguments as the reference operator, plus `shs_rest`: with it, `shs` is the DC coefficient [P,1,3] and
        `shs_rest` the remaining ones [P,M-1,3] (GaussianModel's own storage), so no concatenated copy is needed."""
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        e = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, e if shs is None else shs, e if colors_precomp is None else colors_precomp,
                                   opacities, e if scales is None else scales, e if rotations is None else rotations,
                                   e if cov3D_precomp is None else cov3D_precomp, raster_settings, shs_rest)
