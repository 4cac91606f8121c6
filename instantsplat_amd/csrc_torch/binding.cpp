// Compiled PyTorch binding of the drop-in path: render()'s differentiable body, the training loss and the
// PerPointAdam step as C++ autograd nodes / functions over the C ABI of libmi355gs.so (include/mi355gs.h).
//
// Why it exists: the reference binds its operators through compiled torch extensions
// (reference gaussian_renderer/__init__.py:14-17: `from diff_gaussian_rasterization import ...` is a pybind module,
// train.py:39-43 `fused_ssim`).  The ctypes + Python `autograd.Function` binding of the same entry points
// (instantsplat_amd/fused.py, fused_ssim/__init__.py, optim.py) costs 100-190 us of host time per iteration — the
// autograd engine re-entering Python from its device thread, ~40-argument ctypes calls, ~25 tensor allocations made one
// Python call at a time — which the reference-shaped loop (two blocking read-backs per iteration) cannot hide.
// This module makes the same calls from C++.  It holds NO compute: every kernel is behind the C ABI, whose entry points
// it receives as addresses from instantsplat_amd/_lib.py (so the CPU test tier can hand it the emulated build).
//
// Built by __graft_entry__.build() -> instantsplat_amd/lib/_mi355gs_torch.so (instantsplat_amd/csrc_torch/build.py).
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // ROCm builds of PyTorch call their HIP devices "cuda"
#include <c10/hip/HIPStream.h>

#include <chrono>
#include <optional>
#include <map>
#include <mutex>

#include "../../include/mi355gs.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// ---- C ABI entry points (addresses handed over by _lib.py: whichever build of the library the package has loaded)
struct Abi {
  decltype(&mi355gs_raster_geom_bytes) geom_bytes = nullptr;
  decltype(&mi355gs_raster_tiles_bytes) tiles_bytes = nullptr;
  decltype(&mi355gs_raster_binning_bytes) binning_bytes = nullptr;
  decltype(&mi355gs_raster_grad_scratch_bytes) grad_scratch_bytes = nullptr;
  decltype(&mi355gs_raster_grad_gate_offset) grad_gate_offset = nullptr;
  decltype(&mi355gs_posed_forward_preprocess) posed_forward_preprocess = nullptr;
  decltype(&mi355gs_raster_forward_preprocess) forward_preprocess = nullptr;
  decltype(&mi355gs_raster_backward) raster_backward = nullptr;
  decltype(&mi355gs_raster_forward_render) forward_render = nullptr;
  decltype(&mi355gs_raster_binning_bytes_render_only) binning_bytes_render_only = nullptr;
  decltype(&mi355gs_raster_forward_render_only) forward_render_only = nullptr;
  decltype(&mi355gs_posed_backward) posed_backward = nullptr;
  decltype(&mi355gs_ssim_scratch_bytes) ssim_scratch_bytes = nullptr;
  decltype(&mi355gs_l1_ssim_loss_fused) l1_ssim_loss_fused = nullptr;
  decltype(&mi355gs_l1_scratch_bytes) l1_scratch_bytes = nullptr;
  decltype(&mi355gs_l1_loss_forward) l1_loss_forward = nullptr;
  decltype(&mi355gs_l1_loss_backward) l1_loss_backward = nullptr;
  decltype(&mi355gs_l1_ssim_pair_forward) pair_forward = nullptr;
  decltype(&mi355gs_l1_ssim_pair_backward) pair_backward = nullptr;
  decltype(&mi355gs_loss_program_eval) program_eval = nullptr;
  decltype(&mi355gs_loss_program_eval_grad) program_eval_grad = nullptr;
  decltype(&mi355gs_ssim_forward) ssim_forward = nullptr;
  decltype(&mi355gs_ssim_backward) ssim_backward = nullptr;
  decltype(&mi355gs_adam_multi_step) adam_multi_step = nullptr;
  decltype(&mi355gs_error_string) error_string = nullptr;
  bool bound = false;
  bool allow_cpu = false;   // only the CPU test tier (SIMT-emulated build of the kernels) runs on CPU tensors
} g_abi;

void bind_abi(const std::map<std::string, uintptr_t>& sym, bool allow_cpu_tensors) {
  g_abi.allow_cpu = allow_cpu_tensors;
  auto get = [&](const char* name) {
    auto it = sym.find(name);
    TORCH_CHECK(it != sym.end() && it->second, "mi355gs torch binding: entry point ", name, " was not provided");
    return it->second;
  };
#define GS_BIND(field, name) g_abi.field = reinterpret_cast<decltype(g_abi.field)>(get(#name))
  GS_BIND(geom_bytes, mi355gs_raster_geom_bytes);
  GS_BIND(tiles_bytes, mi355gs_raster_tiles_bytes);
  GS_BIND(binning_bytes, mi355gs_raster_binning_bytes);
  GS_BIND(grad_scratch_bytes, mi355gs_raster_grad_scratch_bytes);
  GS_BIND(grad_gate_offset, mi355gs_raster_grad_gate_offset);
  GS_BIND(posed_forward_preprocess, mi355gs_posed_forward_preprocess);
  GS_BIND(forward_preprocess, mi355gs_raster_forward_preprocess);
  GS_BIND(raster_backward, mi355gs_raster_backward);
  GS_BIND(forward_render, mi355gs_raster_forward_render);
  GS_BIND(binning_bytes_render_only, mi355gs_raster_binning_bytes_render_only);
  GS_BIND(forward_render_only, mi355gs_raster_forward_render_only);
  GS_BIND(posed_backward, mi355gs_posed_backward);
  GS_BIND(ssim_scratch_bytes, mi355gs_ssim_scratch_bytes);
  GS_BIND(l1_ssim_loss_fused, mi355gs_l1_ssim_loss_fused);
  GS_BIND(l1_scratch_bytes, mi355gs_l1_scratch_bytes);
  GS_BIND(l1_loss_forward, mi355gs_l1_loss_forward);
  GS_BIND(l1_loss_backward, mi355gs_l1_loss_backward);
  GS_BIND(pair_forward, mi355gs_l1_ssim_pair_forward);
  GS_BIND(pair_backward, mi355gs_l1_ssim_pair_backward);
  GS_BIND(program_eval, mi355gs_loss_program_eval);
  GS_BIND(program_eval_grad, mi355gs_loss_program_eval_grad);
  GS_BIND(ssim_forward, mi355gs_ssim_forward);
  GS_BIND(ssim_backward, mi355gs_ssim_backward);
  GS_BIND(adam_multi_step, mi355gs_adam_multi_step);
  GS_BIND(error_string, mi355gs_error_string);
#undef GS_BIND
  g_abi.bound = true;
}

void check(int code, const char* what) {
  TORCH_CHECK(code == 0, "mi355gs: ", what, " failed: ", g_abi.error_string ? g_abi.error_string(code) : "?", " (", code, ")");
}

// The launches inside the library go to the process's current device: make it the tensors' device for the call, and hand
// over torch's current stream on it (CPU tensors: the emulated build of the CPU test tier, no stream).
struct DeviceScope {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard;
  void* stream = nullptr;
  explicit DeviceScope(const Tensor& t) {
    if (t.is_cuda()) {
      guard.set_device(t.device());
      stream = (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
    }
  }
  void synchronize(const Tensor& t) const {
    if (t.is_cuda()) c10::hip::getCurrentHIPStream(t.device().index()).synchronize();
  }
};

// The frame's instance count is stored by the tile-scan kernel into a word of pinned, device-mapped host memory that the
// caller set to -1: poll it (the store needs no stream synchronisation to become visible: the memory is fine-grained host
// memory), and fall back to waiting for the stream if it does not show up within a couple of milliseconds.
void wait_for_count(const int32_t* count, const DeviceScope& dev, const Tensor& t) {
  if (!t.is_cuda()) return;   // emulated kernels run synchronously
  // the forward is called from Python: other Python threads (a data loader, a progress reporter) run while this one spins or
  // waits for the stream.  (The autograd engine's threads never get here, and they do not hold the GIL.)
  std::optional<py::gil_scoped_release> nogil;
  if (PyGILState_Check()) nogil.emplace();
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    if (*reinterpret_cast<const volatile int32_t*>(count) >= 0) return;
    if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
  }
  dev.synchronize(t);
  TORCH_CHECK(*reinterpret_cast<const volatile int32_t*>(count) >= 0, "mi355gs: the forward did not report its instance count");
}

// The same wait for a caller's own result words (instantsplat_amd/train.py: loss and instance count of an iteration, stored by
// the kernels that produce them into pinned host memory the caller preset to `sentinel`): spin until none of them holds the
// sentinel, GIL released; past `timeout_us` fall back to waiting for `like`'s current stream.  No event is recorded behind the
// producing kernels for this (an event is a marker packet on the queue: ~7 us between two kernels, tools/gap_analysis.py).
void wait_for_words(Tensor words, int64_t sentinel, Tensor like, int64_t timeout_us) {
  TORCH_CHECK(words.scalar_type() == at::kInt && !words.is_cuda() && words.is_contiguous(), "wait_for_words: an int32 host tensor");
  const volatile int32_t* w = words.data_ptr<int32_t>();
  const int64_t n = words.numel();
  auto all_there = [&]() {
    for (int64_t i = 0; i < n; ++i)
      if (w[i] == (int32_t)sentinel) return false;
    return true;
  };
  if (all_there()) return;
  {
    std::optional<py::gil_scoped_release> nogil;
    if (PyGILState_Check()) nogil.emplace();
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
      if (all_there()) return;
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(timeout_us)) break;
    }
    if (like.is_cuda()) {
      const DeviceScope dev(like);
      dev.synchronize(like);
    }
  }
  TORCH_CHECK(all_there(), "mi355gs: the step did not report its results");
}

Tensor f32c(const Tensor& t, const char* name, const Tensor& like) {
  TORCH_CHECK(t.is_cuda() || g_abi.allow_cpu, "instantsplat_amd operators run on the GPU only (got a CPU tensor; there is no CPU fallback)");
  TORCH_CHECK(t.scalar_type() == at::kFloat, "expected float32, got ", t.scalar_type(), " (", name, ")");
  TORCH_CHECK(t.device() == like.device(), "tensors on different devices: ", like.device(), " vs ", t.device(), " (", name, ")");
  return t.is_contiguous() ? t : t.contiguous();
}
float* fp(const Tensor& t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }

Tensor empty_bytes(size_t n, const Tensor& like) {
  return at::empty({(int64_t)(n > 0 ? n : 1)}, like.options().dtype(at::kByte));
}

// ---- host-time accounting of the nodes (diagnostics for tools/host_timeline.py: where does an iteration's host time go?)
struct HostClock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double* acc;
  explicit HostClock(double* a) : acc(a) {}
  ~HostClock() { *acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
};
double g_host_us[6] = {0, 0, 0, 0, 0, 0};   // render fwd, render bwd, loss fwd, loss bwd, adam step, render fwd's wait for the count

// ---- the gate flags of the most recent posed backward, for the optimizer (see adam_step).
// mi355gs_posed_backward leaves "gradient of group k has a non-zero element" flags behind its gradient records.  The
// optimizer may use them instead of re-reading all gradients only for tensors that are PROVABLY the ones that call wrote:
// same storage object (kept alive only weakly here), offset 0, never modified since (the version counter autograd's
// .grad shares with the tensor this node returned is where the node left it; an accumulation into an existing .grad bumps
// it, a clone has another storage).
struct GateRecord {
  Tensor scratch;  // keeps the flags alive (they live in the backward's gradient scratch)
  const float* gate = nullptr;
  std::vector<c10::weak_intrusive_ptr<c10::StorageImpl>> storage;   // the seven gradient tensors' storages, group order
  int64_t numel[7] = {0, 0, 0, 0, 0, 0, 0};
  int64_t version[7] = {0, 0, 0, 0, 0, 0, 0};   // as handed to autograd (zeros_like leaves 1, empty_like 0)
  bool valid = false;
};
std::mutex g_gate_mutex;
GateRecord g_gates;

// Does a backward of this call exist at all?  (The forward then owns the backward's accumulator buffer already and lets the
// projection kernel clear it on its way: no memset in front of the backward's first kernel.)
template <class... T> bool any_requires_grad(const T&... t) {
  if (!at::GradMode::is_enabled()) return false;
  bool any = false;
  ((any = any || (t.defined() && t.requires_grad())), ...);
  return any;
}
// ... asked where the operator is CALLED: inside a custom function's forward() grad mode is off, so the question would always be
// answered "no" there (round 4 shipped it that way for a while: the backward's memset stayed, profiles/r04_dropin_aten_ops.txt)
thread_local bool t_backward_follows = false;
bool g_backward_inline = false;   // A/B switch (backward_inline): loss_affine_backward runs the engine on the calling thread
bool g_render_only_when_no_grad = true;   // A/B switch (render_only): false = every forward runs the training instantiation of stage 2
bool g_forward_owns_scratch = true;   // A/B switch (forward_owns_scratch): false = the backward allocates and memsets, as before ABI v7
struct BackwardFollows {
  explicit BackwardFollows(bool v) { t_backward_follows = v; }
  ~BackwardFollows() { t_backward_follows = false; }
};

// The all-zero gradient of `f_rest` below its SH degree (what the reference's cat(f_dc, f_rest) backward produces: 35 MB of
// zeros per iteration at C3, filled by a kernel every backward).  ONE persistent zero buffer per parameter (device, shape, address of its memory) is handed
// to autograd instead, as a fresh alias each time: AccumulateGrad takes it over without a copy (use count 1, dense), the
// optimizer reads zeros, zero_grad(set_to_none=True) drops the alias.  (Keyed on the parameter's own memory: see zero_grad_like.)  Aliases share the buffer's version counter, so any
// in-place operation on such a `.grad` (accumulation over two backward passes, clipping, zero_grad(set_to_none=False), an
// all-reduce) is seen here as a changed version and the buffer is zeroed again before its next use.  Writes BEHIND the
// version counter (`.grad.data`, raw pointers) are the caller's to declare: mi355gs shared_zero_grad(false)
// (instantsplat_amd.optim.PerPointAdam.use_backward_gates = False switches it off together with the gate shortcut).
struct ZeroGrad { Tensor buf; uint32_t version = 0; };
std::mutex g_zero_mutex;
std::map<std::string, ZeroGrad> g_zero_pool;
bool g_share_zero_grad = true;
Tensor zero_grad_like(const Tensor& like) {
  if (!g_share_zero_grad) return at::zeros_like(like);
  // one buffer per PARAMETER (its memory's address), not per shape: two models of the same size in one process — teacher and
  // student, a deep copy, two scenes — must not find each other's writes in their `.grad`
  std::ostringstream key;
  key << like.device() << ":" << like.sizes() << "@" << like.data_ptr();
  std::lock_guard<std::mutex> lock(g_zero_mutex);
  if (g_zero_pool.size() > 8 && !g_zero_pool.count(key.str())) g_zero_pool.clear();   // (shapes come and go: a handful of scenes at most)
  ZeroGrad& z = g_zero_pool[key.str()];
  if (!z.buf.defined()) {
    z.buf = at::zeros_like(like, like.options().requires_grad(false), at::MemoryFormat::Contiguous);
    z.version = z.buf._version();
  } else if (z.buf._version() != z.version) {
    z.buf.zero_();
    z.version = z.buf._version();
  }
  return z.buf.detach();
}

// ------------------------------------------------------------------------------------------------
// GaussianModel.get_RT: row `idx` of the learnable pose table (reference scene/gaussian_model.py:134-136, `self.P[idx]`).
// Autograd's own backward of that selection fills a zero table and copies the seven values into it — two launches per
// iteration of the reference loop.  As a node of this binding the selection can be seen by the render node: its backward then
// has the pose-finishing kernel write the WHOLE table gradient (mi355gs_posed_backward, pose_rows / pose_row) and hands this
// node a 7-element alias of it, which is recognised here and widened back to the table without a launch.  Anything else
// arriving as the row's gradient (a sum with another consumer's gradient, a gradient from the op-by-op render path) takes the
// general path: zeros + copy, as autograd would.
// ------------------------------------------------------------------------------------------------
struct PoseRowSeen {   // forward side, per calling thread: the last row handed out
  c10::weak_intrusive_ptr<c10::TensorImpl> row{c10::intrusive_ptr<c10::TensorImpl>()};
  int64_t rows = 0, index = 0;
};
thread_local PoseRowSeen t_pose_row;
struct PoseTableWritten {   // backward side: the table gradient the render node's backward wrote last
  c10::weak_intrusive_ptr<c10::StorageImpl> storage{c10::intrusive_ptr<c10::StorageImpl>()};
  int64_t rows = 0, index = 0;
  bool valid = false;
};
PoseTableWritten g_pose_table;   // under g_gate_mutex

struct PoseRowFn : public torch::autograd::Function<PoseRowFn> {
  static Tensor forward(AutogradContext* ctx, Tensor table, int64_t index) {
    TORCH_CHECK(table.dim() == 2 && table.size(1) == 7, "pose_row: the pose table must be [views, 7]");
    const int64_t rows = table.size(0);
    if (index < 0) index += rows;
    TORCH_CHECK(index >= 0 && index < rows, "pose_row: index ", index, " is out of range for ", rows, " poses");
    ctx->saved_data["rows"] = rows;
    ctx->saved_data["index"] = index;
    Tensor row = table.select(0, index).detach();   // the same memory, no autograd view relation to keep consistent
    t_pose_row.row = c10::weak_intrusive_ptr<c10::TensorImpl>(row.getIntrusivePtr());
    t_pose_row.rows = table.is_contiguous() && table.scalar_type() == at::kFloat ? rows : 0;
    t_pose_row.index = index;
    return row;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grad_out) {
    const int64_t rows = ctx->saved_data["rows"].toInt(), index = ctx->saved_data["index"].toInt();
    const Tensor& g = grad_out[0];
    if (!g.defined()) return {Tensor(), Tensor()};
    {
      std::lock_guard<std::mutex> lock(g_gate_mutex);
      auto strong = g_pose_table.storage.lock();
      const bool mine = g_pose_table.valid && strong && strong.get() == g.storage().unsafeGetStorageImpl() && g_pose_table.rows == rows &&
                        g_pose_table.index == index && g.scalar_type() == at::kFloat && g.dim() == 1 && g.numel() == 7 && g.is_contiguous() &&
                        g.storage_offset() == 7 * index;
      g_pose_table.valid = false;
      if (mine) {
        Tensor full = at::empty({0}, g.options());
        full.set_(g.storage(), 0, {rows, 7}, {7, 1});
        // the optimizer's record of "gradients this backward wrote" (group 6 = the pose table): the table is what P.grad becomes
        if (g_gates.valid && g_gates.storage.size() == 7) {
          g_gates.storage[6] = full.storage().getWeakStorageImpl();
          g_gates.numel[6] = full.numel();
          g_gates.version[6] = (int64_t)full._version();
        }
        return {full, Tensor()};
      }
    }
    Tensor full = at::zeros({rows, 7}, g.options());
    full.select(0, index).copy_(g);
    return {full, Tensor()};
  }
};

Tensor pose_row(Tensor table, int64_t index) { return PoseRowFn::apply(table, index); }

// The deterministic-backward mode (mi355gs_tune_deterministic) is process-wide and enters the layout of `binning` and the size
// of the gradient scratch.  A frame must see the SAME mode in its backward as in its forward: switched on in between (a retained
// graph, a frame in flight, a second thread), the backward would write per-instance rows past the end of buffers laid out
// without them.  The sizes say so: checked on the host before anything is enqueued.
void check_frame_buffers(const Tensor& binning, const Tensor& scratch, int64_t R, int P, int W, int H) {
  const size_t need_b = g_abi.binning_bytes(R, W, H), need_s = g_abi.grad_scratch_bytes(P);
  TORCH_CHECK(!binning.defined() || (size_t)binning.numel() >= need_b || R <= 0,
              "mi355gs: this frame's binning buffer (", binning.numel(), " bytes) is smaller than the backward's layout needs (", need_b,
              "): the deterministic-backward mode was switched on between the frame's forward and its backward");
  TORCH_CHECK(!scratch.defined() || (size_t)scratch.numel() >= need_s,
              "mi355gs: this frame's gradient scratch (", scratch.numel(), " bytes) is smaller than the backward needs (", need_s,
              "): the deterministic-backward mode was switched on between the frame's forward and its backward");
}

// ------------------------------------------------------------------------------------------------
// render()'s differentiable body: raw GaussianModel tensors + the 7-vector camera pose in, image out
// (reference gaussian_renderer/__init__.py:81-135; the Python twin is instantsplat_amd/fused.py::_RenderPosed)
// ------------------------------------------------------------------------------------------------
struct RenderPosedFn : public torch::autograd::Function<RenderPosedFn> {
  static variable_list forward(AutogradContext* ctx, Tensor xyz_, Tensor rot_, Tensor scaling_, Tensor opl_, Tensor f_dc_,
                               Tensor f_rest_, Tensor pose_, Tensor means2D, Tensor bg_, Tensor view_, Tensor proj_, Tensor origin_,
                               int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier, int64_t D,
                               int64_t capacity, int64_t count_hint, Tensor count_slot) {
    TORCH_CHECK(g_abi.bound, "mi355gs torch binding: bind() has not been called");
    HostClock clock(&g_host_us[0]);
    (void)means2D;  // its VALUE is never read (the reference's viewspace_points dummy); it only receives a gradient
    const Tensor xyz = f32c(xyz_, "xyz", xyz_), rot = f32c(rot_, "rotation", xyz), scaling = f32c(scaling_, "scaling", xyz),
                 opl = f32c(opl_, "opacity", xyz), f_dc = f32c(f_dc_, "features_dc", xyz), f_rest = f32c(f_rest_, "features_rest", xyz),
                 pose = f32c(pose_, "camera_pose", xyz), bg = f32c(bg_, "bg", xyz), view = f32c(view_, "viewmatrix", xyz),
                 proj = f32c(proj_, "projmatrix", xyz), origin = f32c(origin_, "campos", xyz);
    TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "means3D must have dimensions (num_points, 3)");
    TORCH_CHECK(count_slot.scalar_type() == at::kInt && count_slot.numel() == 1, "count_slot must be one int32");
    const int P = (int)xyz.size(0);
    const DeviceScope dev(xyz);
    Tensor radii = at::empty({P}, xyz.options().dtype(at::kInt));
    Tensor visible = at::empty({P}, xyz.options().dtype(at::kBool));   // radii > 0, written by the projection kernel
    Tensor color = at::empty({3, H, W}, xyz.options());
    Tensor geom = empty_bytes(g_abi.geom_bytes(P), xyz), tiles = empty_bytes(g_abi.tiles_bytes((int)W, (int)H), xyz);
    // a backward will follow: its accumulator buffer is allocated now and cleared by the projection kernel on its way
    Tensor scratch;
    if (t_backward_follows && g_forward_owns_scratch) scratch = empty_bytes(g_abi.grad_scratch_bytes(P), xyz);
    int32_t* count = count_slot.data_ptr<int32_t>();  // pinned host memory the tile-scan kernel stores into (a CPU word under emulation)
    auto preprocess = [&]() {
      *reinterpret_cast<volatile int32_t*>(count) = -1;   // "not written yet" for wait_for_count
      check(g_abi.posed_forward_preprocess(dev.stream, P, (int)D, (int)W, (int)H, fp(xyz), fp(f_dc), fp(f_rest), fp(opl), fp(scaling),
                                           (float)scale_modifier, fp(rot), fp(pose), fp(view), fp(proj), fp(origin), (float)tanfovx,
                                           (float)tanfovy, radii.data_ptr<int32_t>(), geom.data_ptr(), tiles.data_ptr(), count,
                                           P > 0 ? reinterpret_cast<uint8_t*>(visible.data_ptr<bool>()) : nullptr,
                                           scratch.defined() ? scratch.data_ptr() : nullptr, 0),
            "posed_forward_preprocess");
    };
    Tensor binning;
    // no backward will follow (nothing requires a gradient, or grad mode is off where the operator was called): the render-only
    // stage 2 — a binning buffer of keys + lists only, no boundary records / hit masks / unit table left for a backward
    const bool train = t_backward_follows || !g_render_only_when_no_grad;
    auto stage2 = [&](int64_t cap) {
      if (train) {
        binning = empty_bytes(g_abi.binning_bytes(cap, (int)W, (int)H), xyz);
        check(g_abi.forward_render(dev.stream, P, (int)W, (int)H, cap, fp(bg), geom.data_ptr(), tiles.data_ptr(), binning.data_ptr(),
                                   fp(color), 0),
              "raster_forward_render");
      } else {
        binning = empty_bytes(g_abi.binning_bytes_render_only(cap, (int)W, (int)H), xyz);
        check(g_abi.forward_render_only(dev.stream, P, (int)W, (int)H, cap, fp(bg), geom.data_ptr(), tiles.data_ptr(), binning.data_ptr(),
                                        fp(color), 0),
              "raster_forward_render_only");
      }
    };
    preprocess();
    int64_t R = capacity;
    if (R >= 0) {
      stage2(R);   // the caller's bound: no host synchronisation at all (BinningPolicy "bounded"; it verifies the count later)
    } else {
      // The reference operator's own blocking read-back of the instance count (its forward sizes the sort buffers from it).
      // The blocking semantics are kept — the call returns knowing the exact count, and no instance was dropped — but the GPU
      // does not sit idle through the round trip: when a count of an earlier frame like this one is known (`count_hint`),
      // stage 2 is enqueued at once in buffers sized from it, and the host then merely waits for the count word, which the
      // tile-scan kernel stores straight into pinned host memory.  A frame that outgrew the guess is projected and rendered a
      // second time with exact buffers (identical result; rare: the guess is 1.5 x + 16384).
      const int64_t guess = count_hint > 0 ? count_hint + count_hint / 2 + 16384 : -1;
      if (guess > 0) stage2(guess);
      {
        HostClock wait_clock(&g_host_us[5]);
        wait_for_count(count, dev, xyz);
      }
      R = *reinterpret_cast<volatile int32_t*>(count);
      if (guess > 0 && R <= guess) {
        R = guess;                      // the capacity the frame's buffers were laid out for: the backward needs this number
      } else {
        if (guess > 0) preprocess();    // the overflowing stage 2 consumed the tile cursors: start the frame again
        if (guess > 0) wait_for_count(count, dev, xyz);
        stage2(R);
      }
    }
    ctx->saved_data["dims"] = std::vector<int64_t>{P, D, W, H, R};
    ctx->saved_data["scalars"] = std::vector<double>{tanfovx, tanfovy, scale_modifier};
    ctx->saved_data["scratch_is_clear"] = scratch.defined();
    {
      // is the pose the row get_RT just handed out on this thread?  Then the backward writes the table's gradient (PoseRowFn)
      auto seen = t_pose_row.row.lock();
      const bool is_row = seen && seen.get() == pose_.unsafeGetTensorImpl() && t_pose_row.rows > 0;
      ctx->saved_data["pose_table"] = std::vector<int64_t>{is_row ? t_pose_row.rows : 0, is_row ? t_pose_row.index : 0};
    }
    ctx->save_for_backward({xyz, rot, scaling, opl, f_dc, f_rest, pose, radii, geom, tiles, binning, bg, view, proj, origin, color,
                            scratch.defined() ? scratch : Tensor()});
    ctx->mark_non_differentiable({radii, visible});
    // (no zero tensors for the gradients of `radii` / `visible`: autograd otherwise fills one of each per backward — two launches)
    ctx->set_materialize_grads(false);
    return {color, radii, visible};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grad_out) {
    HostClock clock(&g_host_us[1]);
    const auto saved = ctx->get_saved_variables();
    const Tensor &xyz = saved[0], &rot = saved[1], &scaling = saved[2], &opl = saved[3], &f_dc = saved[4], &f_rest = saved[5],
                 &pose = saved[6], &radii = saved[7], &geom = saved[8], &tiles = saved[9], &binning = saved[10], &bg = saved[11],
                 &view = saved[12], &proj = saved[13], &origin = saved[14], &color = saved[15];
    const auto dims = ctx->saved_data["dims"].toIntVector();
    const auto sc = ctx->saved_data["scalars"].toDoubleVector();
    const int P = (int)dims[0], D = (int)dims[1], W = (int)dims[2], H = (int)dims[3];
    const int64_t R = dims[4];
    if (!grad_out[0].defined()) return variable_list(21);   // the image took no part in the loss: no gradient flows back
    const Tensor g = f32c(grad_out[0], "grad_color", xyz);
    const DeviceScope dev(xyz);
    Tensor d_xyz = at::empty_like(xyz), d_rot = at::empty_like(rot), d_scaling = at::empty_like(scaling), d_opl = at::empty_like(opl),
           d_fdc = at::empty_like(f_dc), d_m2d = at::empty_like(xyz);
    // below its SH degree f_rest gets the all-zero gradient cat(f_dc, f_rest) would give it (the optimizer then takes
    // PerPointAdam's zero-gradient step on it, as in the reference)
    Tensor d_frest = D == 0 ? zero_grad_like(f_rest) : at::empty_like(f_rest);
    const auto pose_table = ctx->saved_data["pose_table"].toIntVector();
    const int64_t pose_rows = pose_table[0], pose_index = pose_table[1];
    Tensor d_pose_store = at::empty({pose_rows > 0 ? pose_rows * 7 : 7}, xyz.options());   // the 7 values, or the whole table's gradient
    Tensor d_pose = d_pose_store;
    if (pose_rows > 0) {   // an alias of row `pose_index` (not a view: nothing keeps the table's tensor alive but its memory)
      d_pose = at::empty({0}, xyz.options());
      d_pose.set_(d_pose_store.storage(), 7 * pose_index, {7}, {1});
    }
    // the accumulator buffer the forward allocated and had cleared — once: a second backward of the same frame (retain_graph)
    // finds it used and lets the library clear it
    Tensor scratch = saved.size() > 16 ? saved[16] : Tensor();
    int scratch_is_clear = (scratch.defined() && ctx->saved_data["scratch_is_clear"].toBool()) ? 1 : 0;
    ctx->saved_data["scratch_is_clear"] = false;
    if (!scratch.defined()) scratch = empty_bytes(g_abi.grad_scratch_bytes(P), xyz);
    Tensor pose_scratch = at::empty({16 * (((int64_t)P + 255) / 256) + 32}, xyz.options());
    check_frame_buffers(binning, scratch, R, P, W, H);
    check(g_abi.posed_backward(dev.stream, P, D, W, H, fp(bg), fp(xyz), fp(f_dc), fp(f_rest), fp(opl), fp(scaling), (float)sc[2], fp(rot),
                               fp(pose), fp(view), fp(proj), fp(origin), (float)sc[0], (float)sc[1], geom.data_ptr(), tiles.data_ptr(),
                               binning.data_ptr(), R, radii.data_ptr<int32_t>(), fp(color), fp(g), scratch.data_ptr(), fp(pose_scratch),
                               fp(d_xyz), fp(d_m2d), fp(d_fdc), D ? fp(d_frest) : nullptr, fp(d_opl), fp(d_scaling), fp(d_rot),
                               fp(d_pose_store), (int)pose_rows, (int)pose_index, scratch_is_clear, 0),
          "posed_backward");
    {
      std::lock_guard<std::mutex> lock(g_gate_mutex);
      g_pose_table.valid = pose_rows > 0;
      if (pose_rows > 0) {
        g_pose_table.storage = d_pose_store.storage().getWeakStorageImpl();
        g_pose_table.rows = pose_rows; g_pose_table.index = pose_index;
      }
      const Tensor* outs[7] = {&d_xyz, &d_fdc, &d_frest, &d_opl, &d_scaling, &d_rot, &d_pose};   // the optimizer's group order
      g_gates.storage.clear();
      for (int k = 0; k < 7; ++k) {
        g_gates.storage.push_back(outs[k]->storage().getWeakStorageImpl());
        g_gates.numel[k] = outs[k]->numel();
        g_gates.version[k] = (int64_t)outs[k]->_version();
      }
      g_gates.scratch = scratch;
      g_gates.gate = reinterpret_cast<const float*>(static_cast<const char*>(scratch.data_ptr()) + g_abi.grad_gate_offset(P));
      g_gates.valid = true;
    }
    Tensor none;
    return {d_xyz, d_rot, d_scaling, d_opl, d_fdc, d_frest, d_pose, d_m2d, none, none, none, none,
            none, none, none, none, none, none, none, none, none};
  }
};

std::vector<Tensor> render_posed(Tensor xyz, Tensor rot, Tensor scaling, Tensor opl, Tensor f_dc, Tensor f_rest, Tensor pose,
                                 Tensor means2D, Tensor bg, Tensor view, Tensor proj, Tensor origin, int64_t H, int64_t W,
                                 double tanfovx, double tanfovy, double scale_modifier, int64_t D, int64_t capacity, int64_t count_hint,
                                 Tensor count_slot) {
  const BackwardFollows scope(any_requires_grad(xyz, rot, scaling, opl, f_dc, f_rest, pose, means2D));
  return RenderPosedFn::apply(xyz, rot, scaling, opl, f_dc, f_rest, pose, means2D, bg, view, proj, origin, H, W, tanfovx, tanfovy,
                              scale_modifier, D, capacity, count_hint, count_slot);
}

// ------------------------------------------------------------------------------------------------
// The operator itself: GaussianRasterizer.forward of the package the reference imports at gaussian_renderer/__init__.py:14-17
// and calls at :126-135 (Python twin: diff_gaussian_rasterization/__init__.py::_RasterizeGaussians).  Optional inputs are
// undefined tensors; gradients come back for (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
// cov3Ds_precomp, sh_rest) like the upstream operator's.
// ------------------------------------------------------------------------------------------------
struct RasterizeFn : public torch::autograd::Function<RasterizeFn> {
  using OptTensor = c10::optional<Tensor>;   // (an undefined Tensor cannot be an argument of a custom function: its metadata is recorded)
  static variable_list forward(AutogradContext* ctx, Tensor means3D_, Tensor means2D, OptTensor sh_, OptTensor colors_, Tensor opac_,
                               OptTensor scales_, OptTensor rot_, OptTensor cov_, OptTensor sh_rest_, Tensor bg_, Tensor view_, Tensor proj_,
                               Tensor campos_,
                               int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier, int64_t D, bool prefiltered,
                               int64_t capacity, int64_t count_hint, Tensor count_slot) {
    TORCH_CHECK(g_abi.bound, "mi355gs torch binding: bind() has not been called");
    HostClock clock(&g_host_us[0]);
    (void)means2D;
    auto opt = [&](const OptTensor& t, const char* name, const Tensor& like) {
      return (t.has_value() && t->defined() && t->numel()) ? f32c(*t, name, like) : Tensor();
    };
    const Tensor means3D = f32c(means3D_, "means3D", means3D_), opac = f32c(opac_, "opacities", means3D);
    const Tensor sh = opt(sh_, "sh", means3D), colors = opt(colors_, "colors_precomp", means3D), scales = opt(scales_, "scales", means3D),
                 rot = opt(rot_, "rotations", means3D), cov = opt(cov_, "cov3Ds_precomp", means3D), sh_rest = opt(sh_rest_, "sh_rest", means3D);
    const Tensor bg = f32c(bg_, "bg", means3D), view = f32c(view_, "viewmatrix", means3D), proj = f32c(proj_, "projmatrix", means3D),
                 campos = f32c(campos_, "campos", means3D);
    TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");
    TORCH_CHECK(count_slot.scalar_type() == at::kInt && count_slot.numel() == 1, "count_slot must be one int32");
    const int P = (int)means3D.size(0);
    const int M = sh.defined() ? (int)(sh.size(1) + (sh_rest.defined() ? sh_rest.size(1) : 0)) : 0;
    const DeviceScope dev(means3D);
    Tensor radii = at::empty({P}, means3D.options().dtype(at::kInt));
    Tensor color = at::empty({3, H, W}, means3D.options());
    Tensor geom = empty_bytes(g_abi.geom_bytes(P), means3D), tiles = empty_bytes(g_abi.tiles_bytes((int)W, (int)H), means3D);
    Tensor scratch;   // see RenderPosedFn::forward
    if (t_backward_follows && g_forward_owns_scratch) scratch = empty_bytes(g_abi.grad_scratch_bytes(P), means3D);
    int32_t* count = count_slot.data_ptr<int32_t>();
    auto preprocess = [&]() {
      *reinterpret_cast<volatile int32_t*>(count) = -1;
      check(g_abi.forward_preprocess(dev.stream, P, (int)D, M, (int)W, (int)H, fp(means3D), fp(sh), fp(sh_rest), fp(colors), fp(opac), fp(scales),
                                     (float)scale_modifier, fp(rot), fp(cov), fp(view), fp(proj), fp(campos), (float)tanfovx, (float)tanfovy,
                                     prefiltered ? 1 : 0, radii.data_ptr<int32_t>(), geom.data_ptr(), tiles.data_ptr(), count, nullptr,
                                     scratch.defined() ? scratch.data_ptr() : nullptr, 0),
            "raster_forward_preprocess");
    };
    Tensor binning;
    const bool train = t_backward_follows || !g_render_only_when_no_grad;   // see RenderPosedFn::forward
    auto stage2 = [&](int64_t cap) {
      if (train) {
        binning = empty_bytes(g_abi.binning_bytes(cap, (int)W, (int)H), means3D);
        check(g_abi.forward_render(dev.stream, P, (int)W, (int)H, cap, fp(bg), geom.data_ptr(), tiles.data_ptr(), binning.data_ptr(), fp(color), 0),
              "raster_forward_render");
      } else {
        binning = empty_bytes(g_abi.binning_bytes_render_only(cap, (int)W, (int)H), means3D);
        check(g_abi.forward_render_only(dev.stream, P, (int)W, (int)H, cap, fp(bg), geom.data_ptr(), tiles.data_ptr(), binning.data_ptr(),
                                        fp(color), 0),
              "raster_forward_render_only");
      }
    };
    preprocess();
    int64_t R = capacity;
    if (R >= 0) {
      stage2(R);
    } else {   // blocking count read-back with a speculative stage 2: see RenderPosedFn::forward
      const int64_t guess = count_hint > 0 ? count_hint + count_hint / 2 + 16384 : -1;
      if (guess > 0) stage2(guess);
      {
        HostClock wait_clock(&g_host_us[5]);
        wait_for_count(count, dev, means3D);
      }
      R = *reinterpret_cast<volatile int32_t*>(count);
      if (guess > 0 && R <= guess) {
        R = guess;
      } else {
        if (guess > 0) { preprocess(); wait_for_count(count, dev, means3D); }
        stage2(R);
      }
    }
    ctx->saved_data["dims"] = std::vector<int64_t>{P, D, M, W, H, R};
    ctx->saved_data["scalars"] = std::vector<double>{tanfovx, tanfovy, scale_modifier};
    ctx->saved_data["opacity_shape"] = opac_.sizes().vec();
    ctx->saved_data["scratch_is_clear"] = scratch.defined();
    ctx->save_for_backward({means3D, sh, colors, opac, scales, rot, cov, sh_rest, radii, geom, tiles, binning, bg, view, proj, campos, color,
                            scratch.defined() ? scratch : Tensor()});
    ctx->mark_non_differentiable({radii});
    ctx->set_materialize_grads(false);
    return {color, radii};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grad_out) {
    if (!grad_out[0].defined()) return variable_list(23);
    const auto saved = ctx->get_saved_variables();
    const Tensor &means3D = saved[0], &sh = saved[1], &colors = saved[2], &opac = saved[3], &scales = saved[4], &rot = saved[5], &cov = saved[6],
                 &sh_rest = saved[7], &radii = saved[8], &geom = saved[9], &tiles = saved[10], &binning = saved[11], &bg = saved[12],
                 &view = saved[13], &proj = saved[14], &campos = saved[15], &color = saved[16];
    const auto dims = ctx->saved_data["dims"].toIntVector();
    const auto sc = ctx->saved_data["scalars"].toDoubleVector();
    const int P = (int)dims[0], D = (int)dims[1], M = (int)dims[2], W = (int)dims[3], H = (int)dims[4];
    const int64_t R = dims[5];
    const Tensor g = f32c(grad_out[0], "grad_color", means3D);
    const DeviceScope dev(means3D);
    const auto o = means3D.options();
    Tensor d_means3D = at::empty({P, 3}, o), d_means2D = at::empty({P, 3}, o), d_opac = at::empty({P}, o), d_col = at::empty({P, 3}, o);
    Tensor d_sh, d_shr, d_scales, d_rot, d_cov;
    if (sh.defined()) d_sh = at::empty({P, sh_rest.defined() ? 1 : M, 3}, o);
    if (sh_rest.defined()) d_shr = at::empty({P, M - 1, 3}, o);
    if (cov.defined()) d_cov = at::empty({P, 6}, o); else { d_scales = at::empty({P, 3}, o); d_rot = at::empty({P, 4}, o); }
    Tensor scratch = saved.size() > 17 ? saved[17] : Tensor();
    const int scratch_is_clear = (scratch.defined() && ctx->saved_data["scratch_is_clear"].toBool()) ? 1 : 0;
    ctx->saved_data["scratch_is_clear"] = false;
    if (!scratch.defined()) scratch = empty_bytes(g_abi.grad_scratch_bytes(P), means3D);
    check_frame_buffers(binning, scratch, R, P, W, H);
    check(g_abi.raster_backward(dev.stream, P, D, M, W, H, fp(bg), fp(means3D), fp(sh), fp(sh_rest), fp(colors), fp(opac), fp(scales), (float)sc[2],
                                fp(rot), fp(cov), fp(view), fp(proj), fp(campos), (float)sc[0], (float)sc[1], geom.data_ptr(), tiles.data_ptr(),
                                binning.data_ptr(), R, radii.data_ptr<int32_t>(), fp(color), fp(g), scratch.data_ptr(), fp(d_means3D), fp(d_means2D),
                                fp(d_sh), fp(d_shr), fp(d_col), fp(d_opac), fp(d_scales), fp(d_rot), fp(d_cov), scratch_is_clear, 0),
          "raster_backward");
    Tensor none;
    const auto shape = ctx->saved_data["opacity_shape"].toIntVector();
    return {d_means3D, d_means2D, d_sh, sh.defined() ? none : d_col, d_opac.reshape(shape), d_scales, d_rot, d_cov, d_shr,
            none, none, none, none, none, none, none, none, none, none, none, none, none, none};
  }
};

std::vector<Tensor> rasterize(Tensor means3D, Tensor means2D, c10::optional<Tensor> sh, c10::optional<Tensor> colors, Tensor opac,
                              c10::optional<Tensor> scales, c10::optional<Tensor> rot, c10::optional<Tensor> cov, c10::optional<Tensor> sh_rest,
                              Tensor bg, Tensor view, Tensor proj, Tensor campos, int64_t H, int64_t W, double tanfovx, double tanfovy,
                              double scale_modifier, int64_t D, bool prefiltered, int64_t capacity, int64_t count_hint, Tensor count_slot) {
  const Tensor none;
  auto val = [&](const c10::optional<Tensor>& t) -> const Tensor& { return t.has_value() ? *t : none; };
  const BackwardFollows scope(any_requires_grad(means3D, means2D, opac, val(sh), val(colors), val(scales), val(rot), val(cov), val(sh_rest)));
  return RasterizeFn::apply(means3D, means2D, sh, colors, opac, scales, rot, cov, sh_rest, bg, view, proj, campos, H, W,
                            tanfovx, tanfovy, scale_modifier, D, prefiltered, capacity, count_hint, count_slot);
}

// ------------------------------------------------------------------------------------------------
// (1 - lambda) * L1 + lambda * (1 - SSIM) and its gradient from one pass over the images
// (reference train.py:171-176; the Python twin is fused_ssim/__init__.py::_FusedL1SSIM)
// ------------------------------------------------------------------------------------------------
struct L1SsimLossFn : public torch::autograd::Function<L1SsimLossFn> {
  static variable_list forward(AutogradContext* ctx, Tensor img1, Tensor img2, double lambda_dssim) {
    TORCH_CHECK(g_abi.bound, "mi355gs torch binding: bind() has not been called");
    HostClock clock(&g_host_us[2]);
    const Tensor a = f32c(img1, "img1", img1), b = f32c(img2, "img2", a);
    TORCH_CHECK(a.dim() == 4 && a.sizes() == b.sizes(), "fused_ssim expects two [B,C,H,W] tensors of equal shape");
    const int B = (int)a.size(0), C = (int)a.size(1), H = (int)a.size(2), W = (int)a.size(3);
    const DeviceScope dev(a);
    Tensor scratch = empty_bytes(g_abi.ssim_scratch_bytes(B, C, H, W), a);
    Tensor out = at::empty({2}, a.options());   // [ssim_mean, l1_mean]
    Tensor loss = at::empty({}, a.options());
    Tensor grad = at::empty_like(a);
    check(g_abi.l1_ssim_loss_fused(dev.stream, B, C, H, W, fp(a), fp(b), scratch.data_ptr(), (float)lambda_dssim, fp(out), fp(out) + 1,
                                   fp(loss), fp(grad)),
          "l1_ssim_loss_fused");
    ctx->save_for_backward({grad});
    ctx->mark_non_differentiable({out});
    ctx->set_materialize_grads(false);
    return {loss, out};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grad_out) {
    HostClock clock(&g_host_us[3]);
    Tensor none;
    if (!grad_out[0].defined()) return {none, none, none};
    const auto saved = ctx->get_saved_variables();
    return {saved[0] * grad_out[0], none, none};
  }
};

std::vector<Tensor> l1_ssim_loss(Tensor img1, Tensor img2, double lambda_dssim) { return L1SsimLossFn::apply(img1, img2, lambda_dssim); }

// ------------------------------------------------------------------------------------------------
// l1_loss(network_output, gt) = abs(a - b).mean() (reference utils/loss_utils.py:39-40, train.py:171; Python twin:
// loss_utils.py::_L1Loss): two launches instead of sub / abs / mean, one instead of their four backward kernels.  The gradient
// goes to network_output only (gt is data wherever the reference calls it).
// ------------------------------------------------------------------------------------------------
struct L1LossFn : public torch::autograd::Function<L1LossFn> {
  static Tensor forward(AutogradContext* ctx, Tensor a_, Tensor b_) {
    TORCH_CHECK(g_abi.bound, "mi355gs torch binding: bind() has not been called");
    const Tensor a = f32c(a_, "network_output", a_), b = f32c(b_, "gt", a);
    TORCH_CHECK(a.sizes() == b.sizes() && a.numel() > 0, "l1_loss expects two non-empty tensors of equal shape");
    const DeviceScope dev(a);
    Tensor scratch = empty_bytes(g_abi.l1_scratch_bytes(a.numel()), a);
    Tensor out = at::empty({}, a.options());
    check(g_abi.l1_loss_forward(dev.stream, a.numel(), fp(a), fp(b), scratch.data_ptr(), fp(out)), "l1_loss_forward");
    ctx->save_for_backward({a, b});
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grad_out) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &a = saved[0], &b = saved[1];
    const Tensor g = f32c(grad_out[0], "grad", a);
    const DeviceScope dev(a);
    Tensor d = at::empty_like(a);
    check(g_abi.l1_loss_backward(dev.stream, a.numel(), fp(a), fp(b), fp(g), fp(d)), "l1_loss_backward");
    return {d, Tensor()};
  }
};

Tensor l1_loss(Tensor a, Tensor b) { return L1LossFn::apply(a, b); }

// ------------------------------------------------------------------------------------------------
// The reference's loss expression AS WRITTEN (train.py:171-176): l1_loss(image, gt), fused_ssim(image[None], gt[None]) and four
// scalar operations between 0-dim tensors, then loss.backward().  instantsplat_amd/loss_utils.py serves that text with these two
// functions: loss_pair_forward when l1_loss is called (both means and d(ssim_mean)/dimg1 from ONE pass; no autograd node — the
// values are wrapped in lazy scalars on the Python side, which record the arithmetic), and loss_affine when the recorded
// expression is needed as a tensor (backward(), item(), anything else): one node on `image`, one launch forward (the program),
// one launch backward over the image.
// ------------------------------------------------------------------------------------------------
std::vector<Tensor> loss_pair_forward(Tensor img1, Tensor img2) {
  TORCH_CHECK(g_abi.bound, "mi355gs torch binding: bind() has not been called");
  HostClock clock(&g_host_us[2]);
  const Tensor a = f32c(img1, "img1", img1), b = f32c(img2, "img2", a);
  TORCH_CHECK((a.dim() == 3 || a.dim() == 4) && a.sizes() == b.sizes() && a.numel() > 0, "the loss pair expects two [C,H,W] or [B,C,H,W] tensors of equal shape");
  const int64_t o = a.dim() == 4 ? 1 : 0;
  const int B = o ? (int)a.size(0) : 1, C = (int)a.size(o), H = (int)a.size(o + 1), W = (int)a.size(o + 2);
  const DeviceScope dev(a);
  Tensor scratch = empty_bytes(g_abi.ssim_scratch_bytes(B, C, H, W), a);   // the means, as per-workgroup partial sums until a program_eval finishes them
  Tensor means = at::empty({2}, a.options());                               // [l1_mean, ssim_mean], filled by the first program_eval
  Tensor dmap = at::empty_like(a);
  check(g_abi.pair_forward(dev.stream, B, C, H, W, fp(a), fp(b), scratch.data_ptr(), fp(dmap)), "l1_ssim_pair_forward");
  return {means, scratch, dmap, a, b};
}

struct LossAffineFn : public torch::autograd::Function<LossAffineFn> {
  // image: the tensor the caller differentiates (what was handed to l1_loss); a / b / dmap / means / scratch: loss_pair_forward's results
  // host_slot (may be undefined): float32[2] of pinned host memory the program kernel also stores (value, ticket) into
  // unit_grad (may be undefined): the root gradient — a 1 — the caller is about to run the backward with (loss_affine_backward):
  //   the forward's launch then also writes the expression's gradient over the image, and the backward, finding exactly that
  //   tensor as its incoming gradient, hands it on without a launch
  static Tensor forward(AutogradContext* ctx, Tensor image, Tensor a, Tensor b, Tensor dmap, Tensor means, Tensor scratch,
                        std::vector<int64_t> ops, std::vector<double> consts, double c_l1, double c_ssim, Tensor host_slot, double ticket,
                        c10::optional<Tensor> unit_grad_) {
    const Tensor unit_grad = unit_grad_.has_value() ? *unit_grad_ : Tensor();   // (an undefined Tensor cannot be an argument of a custom function)
    TORCH_CHECK(g_abi.bound, "mi355gs torch binding: bind() has not been called");
    TORCH_CHECK(ops.size() == consts.size() && !ops.empty() && ops.size() <= MI355GS_LOSS_PROGRAM_MAX, "loss_affine: 1..16 operations");
    TORCH_CHECK(image.numel() == a.numel() && means.numel() == 2, "loss_affine: image and its contiguous copy differ in size");
    const int64_t o = a.dim() == 4 ? 1 : 0;
    const int B = o ? (int)a.size(0) : 1, C = (int)a.size(o), H = (int)a.size(o + 1), W = (int)a.size(o + 2);
    const DeviceScope dev(a);
    int32_t op32[MI355GS_LOSS_PROGRAM_MAX]; float k32[MI355GS_LOSS_PROGRAM_MAX];
    for (size_t i = 0; i < ops.size(); ++i) { op32[i] = (int32_t)ops[i]; k32[i] = (float)consts[i]; }
    Tensor out = at::empty({}, a.options());
    float* host_out = nullptr;
    if (host_slot.defined() && host_slot.numel() >= 2) {
      TORCH_CHECK(host_slot.scalar_type() == at::kFloat && !host_slot.is_cuda() && host_slot.is_contiguous(), "loss_affine: the host slot is float32[2] host memory");
      host_out = host_slot.data_ptr<float>();
    }
    Tensor d;
    const bool with_grad = unit_grad.defined() && unit_grad.numel() == 1;
    if (with_grad) {
      d = at::empty_like(a);
      check(g_abi.program_eval_grad(dev.stream, (int)ops.size(), op32, k32, B, C, H, W, scratch.data_ptr(), fp(means) + 1, fp(means), fp(out), host_out,
                                    (float)ticket, fp(a), fp(b), fp(dmap), (float)c_l1, (float)c_ssim, fp(d)),
            "loss_program_eval_grad");
    } else {
      check(g_abi.program_eval(dev.stream, (int)ops.size(), op32, k32, B, C, H, W, scratch.data_ptr(), fp(means) + 1, fp(means), fp(out), host_out,
                               (float)ticket),
            "loss_program_eval");
    }
    ctx->save_for_backward({a, b, dmap, with_grad ? d : Tensor()});
    ctx->saved_data["c"] = std::vector<double>{c_l1, c_ssim};
    ctx->saved_data["shape"] = image.sizes().vec();
    ctx->saved_data["unit_grad_ptr"] = with_grad ? (int64_t)(uintptr_t)unit_grad.data_ptr() : (int64_t)0;
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grad_out) {
    HostClock clock(&g_host_us[3]);
    const auto saved = ctx->get_saved_variables();
    const Tensor &a = saved[0], &b = saved[1], &dmap = saved[2];
    const auto c = ctx->saved_data["c"].toDoubleVector();
    Tensor none;
    const int64_t unit_ptr = ctx->saved_data["unit_grad_ptr"].toInt();
    if (unit_ptr != 0 && saved.size() > 3 && saved[3].defined() && (int64_t)(uintptr_t)grad_out[0].data_ptr() == unit_ptr) {
      // the incoming gradient IS the 1 the forward was told about: its launch has written this gradient already
      return {saved[3].view(ctx->saved_data["shape"].toIntVector()), none, none, none, none, none, none, none, none, none, none, none, none};
    }
    const Tensor g = f32c(grad_out[0].reshape({1}), "grad", a);
    const DeviceScope dev(a);
    Tensor d = at::empty_like(a);
    check(g_abi.pair_backward(dev.stream, a.numel(), fp(a), fp(b), fp(dmap), fp(g), (float)c[0], fp(g), (float)c[1], fp(d)), "l1_ssim_pair_backward");
    return {d.view(ctx->saved_data["shape"].toIntVector()), none, none, none, none, none, none, none, none, none, none, none, none};
  }
};

Tensor loss_affine(Tensor image, Tensor a, Tensor b, Tensor dmap, Tensor means, Tensor scratch, std::vector<int64_t> ops,
                   std::vector<double> consts, double c_l1, double c_ssim, Tensor host_slot, double ticket) {
  return LossAffineFn::apply(image, a, b, dmap, means, scratch, ops, consts, c_l1, c_ssim, host_slot, ticket, c10::optional<Tensor>());
}

// train.py:188 `loss.item()` for a value the program kernel has also stored to a pinned host slot: spin (GIL released) until the
// slot carries this materialisation's ticket.  NaN-boxed "not there" on timeout or when a later materialisation has taken the slot
// over — the caller then reads the tensor the ordinary way.
py::object wait_for_loss(Tensor host_slot, double ticket, int64_t timeout_us) {
  TORCH_CHECK(host_slot.scalar_type() == at::kFloat && !host_slot.is_cuda() && host_slot.numel() >= 2, "wait_for_loss: float32[2] host memory");
  const volatile float* w = host_slot.data_ptr<float>();
  const float want = (float)ticket;
  // Tickets count 1 .. 16,000,000 and wrap (instantsplat_amd/lazy_loss.py::_host_slot); a slot is handed out again after 64
  // materialisations.  A slot that already holds a LATER ticket than the one asked for has been taken over: the value will
  // never come — say so at once instead of spinning for the whole timeout (a caller that kept old losses and reads them late).
  auto taken_over = [want](float have) {
    if (have == 0.f || have == want) return false;          // 0: never written
    float ahead = have - want;
    if (ahead < 0.f) ahead += 16000000.f;
    return ahead < 8000000.f;
  };
  bool there = w[1] == want;
  if (!there && !taken_over(w[1])) {
    py::gil_scoped_release nogil;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
      const float have = w[1];
      if (have == want) { there = true; break; }
      if (taken_over(have)) break;
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(timeout_us)) break;
    }
  }
  if (!there) return py::none();
  const float v = w[0];
  if (w[1] != want) return py::none();   // (taken over between the two reads)
  return py::float_((double)v);
}

// loss.backward() of a recorded expression (train.py:177) in one call: the node is created and the engine run from here — no
// Python frames of torch.autograd.backward in between — with a cached 1 per device as the root gradient (autograd's own
// ones_like(loss) is a fill launch per iteration; the node only reads the value).  Returns the materialised tensor.
Tensor loss_affine_backward(Tensor image, Tensor a, Tensor b, Tensor dmap, Tensor means, Tensor scratch, std::vector<int64_t> ops,
                            std::vector<double> consts, double c_l1, double c_ssim, Tensor host_slot, double ticket) {
  static std::mutex mu;
  static std::map<std::string, std::pair<Tensor, uint32_t>> ones;   // per device: the 1 and its version counter as created
  Tensor one;
  {
    std::lock_guard<std::mutex> lock(mu);
    std::ostringstream key;
    key << a.device();
    auto& slot = ones[key.str()];
    if (!slot.first.defined() || slot.first._version() != slot.second) {   // (written to since: a fresh one)
      slot.first = at::ones({}, a.options().requires_grad(false));
      slot.second = slot.first._version();   // (at::ones fills in place: the counter does not start at 0)
    }
    one = slot.first;
  }
  // The engine normally hands the device's nodes to a worker thread and parks the caller: a wake-up and a hand-back per
  // backward() that sit on the stretch of the iteration where the host feeds the device (loss launch -> render backward launch).
  // With multithreading switched off for THIS call (thread-local autograd state — what torch.autograd.set_multithreading_enabled(False)
  // sets) the nodes run on the calling thread instead.
  struct InlineEngine {
    bool prev;
    bool on;
    explicit InlineEngine(bool enable) : prev(c10::AutogradState::get_tls_state().get_multithreading_enabled()), on(enable) {
      if (on) c10::AutogradState::get_tls_state().set_multithreading_enabled(false);
    }
    ~InlineEngine() { if (on) c10::AutogradState::get_tls_state().set_multithreading_enabled(prev); }
  };
  const bool will_run = at::GradMode::is_enabled() && image.requires_grad();
  Tensor out = LossAffineFn::apply(image, a, b, dmap, means, scratch, ops, consts, c_l1, c_ssim, host_slot, ticket, will_run ? c10::optional<Tensor>(one) : c10::optional<Tensor>());
  if (!out.requires_grad()) return out;
  py::gil_scoped_release nogil;   // the engine's worker threads take the GIL themselves for Python-defined nodes
  InlineEngine inline_engine(g_backward_inline);
  torch::autograd::backward({out}, {one}, /*retain_graph=*/false, /*create_graph=*/false);
  return out;
}

// ------------------------------------------------------------------------------------------------
// fused_ssim(img1, img2, padding, train): the operator the reference imports at train.py:39-43 and calls at :173
// (Python twin: fused_ssim/__init__.py::_FusedSSIM).  Gradient with respect to img1 only, as upstream.
// ------------------------------------------------------------------------------------------------
struct SsimFn : public torch::autograd::Function<SsimFn> {
  static Tensor forward(AutogradContext* ctx, Tensor img1, Tensor img2, bool train, bool valid) {
    TORCH_CHECK(g_abi.bound, "mi355gs torch binding: bind() has not been called");
    const Tensor a = f32c(img1, "img1", img1), b = f32c(img2, "img2", a);
    TORCH_CHECK(a.dim() == 4 && a.sizes() == b.sizes(), "fused_ssim expects two [B,C,H,W] tensors of equal shape");
    const int B = (int)a.size(0), C = (int)a.size(1), H = (int)a.size(2), W = (int)a.size(3);
    TORCH_CHECK(!valid || (H > 10 && W > 10), "fused_ssim(padding=\"valid\") needs images larger than the 11x11 window");
    const DeviceScope dev(a);
    Tensor dm1, dm2, dm3;
    if (train) { dm1 = at::empty_like(a); dm2 = at::empty_like(a); dm3 = at::empty_like(a); }
    Tensor scratch = empty_bytes(g_abi.ssim_scratch_bytes(B, C, H, W), a);
    Tensor out = at::empty({2}, a.options());   // [ssim_mean, l1_mean]
    check(g_abi.ssim_forward(dev.stream, B, C, H, W, fp(a), fp(b), fp(dm1), fp(dm2), fp(dm3), scratch.data_ptr(), fp(out),
                             valid ? nullptr : fp(out) + 1, valid ? 1 : 0),
          "ssim_forward");
    if (train) ctx->save_for_backward({a, b, dm1, dm2, dm3});
    ctx->saved_data["flags"] = std::vector<int64_t>{train ? 1 : 0, valid ? 1 : 0};
    return out.select(0, 0).clone();
  }
  static variable_list backward(AutogradContext* ctx, variable_list grad_out) {
    const auto flags = ctx->saved_data["flags"].toIntVector();
    TORCH_CHECK(flags[0] == 1, "fused_ssim was called with train=False; no gradient is available");
    const auto saved = ctx->get_saved_variables();
    const Tensor &a = saved[0], &b = saved[1];
    const int B = (int)a.size(0), C = (int)a.size(1), H = (int)a.size(2), W = (int)a.size(3);
    const DeviceScope dev(a);
    const Tensor scale = f32c(grad_out[0].reshape({1}), "grad", a);
    Tensor grad = at::empty_like(a);
    check(g_abi.ssim_backward(dev.stream, B, C, H, W, fp(a), fp(b), fp(saved[2]), fp(saved[3]), fp(saved[4]), fp(scale), nullptr, fp(grad),
                              (int)flags[1]),
          "ssim_backward");
    Tensor none;
    return {grad, none, none, none};
  }
};

Tensor fused_ssim(Tensor img1, Tensor img2, bool train, bool valid) { return SsimFn::apply(img1, img2, train, valid); }

// ------------------------------------------------------------------------------------------------
// PerPointAdam.step over a fixed set of tensors (reference scene/per_point_adam.py:34-100; Python twin: optim.py)
// ------------------------------------------------------------------------------------------------
struct AdamPlan {
  std::vector<Tensor> params, exp_avg, exp_avg_sq, pplr;  // pplr[t] undefined: no per-point multiplier
  std::vector<int64_t> numel;
  std::vector<int32_t> row;
  double beta1, beta2, eps;
  int64_t last_used_gates = 0;   // diagnostics: tensors of the last step that took a gate flag of the posed backward
  std::string last_gate_note;    // diagnostics: why a tensor of the last step did not qualify
  // the library's memory of gated-off tensors across this plan's steps (include/mi355gs.h, mi355gs_adam_multi_step `live`):
  // f_rest below its SH degree is skipped without reading its first moment again.  These steps are the only writers of the
  // moments the library knows of; a torch operation on a moment tensor shows in its version counter and resets the memory.
  Tensor live;
  uint32_t seq = 0;
  std::vector<int64_t> moment_version;
  // the Parameters themselves (set_owners): the steady-state step reads their `.grad` and clears it from here, so the optimizer's
  // Python between the loss read-back and the Adam launch — the GPU has nothing queued in that stretch — is a dozen lines
  std::vector<Tensor> owners;

  AdamPlan(std::vector<Tensor> p, std::vector<Tensor> m, std::vector<Tensor> v, std::vector<c10::optional<Tensor>> pp, double b1, double b2,
           double e)
      : params(std::move(p)), exp_avg(std::move(m)), exp_avg_sq(std::move(v)), beta1(b1), beta2(b2), eps(e) {
    const size_t n = params.size();
    TORCH_CHECK(n >= 1 && n <= 8 && exp_avg.size() == n && exp_avg_sq.size() == n && pp.size() == n, "AdamPlan: 1..8 tensors, equal-length lists");
    for (size_t t = 0; t < n; ++t) {
      const Tensor& q = params[t];
      TORCH_CHECK(q.is_cuda() || g_abi.allow_cpu, "instantsplat_amd operators run on the GPU only (got a CPU tensor; there is no CPU fallback)");
      TORCH_CHECK(q.scalar_type() == at::kFloat && q.is_contiguous(), "AdamPlan: parameters must be contiguous float32");
      for (const Tensor* s : {&exp_avg[t], &exp_avg_sq[t]})
        TORCH_CHECK(s->scalar_type() == at::kFloat && s->is_contiguous() && s->device() == q.device() && s->numel() == q.numel(),
                    "AdamPlan: moments must match their parameter");
      numel.push_back(q.numel());
      if (pp[t].has_value() && pp[t]->defined()) {
        const Tensor& l = *pp[t];
        TORCH_CHECK(l.scalar_type() == at::kFloat && l.is_contiguous() && l.device() == q.device() && q.dim() >= 1 && q.size(0) > 0 &&
                        l.numel() == q.size(0),
                    "AdamPlan: per_point_lr must hold one contiguous float32 per point");
        pplr.push_back(l);
        row.push_back((int32_t)(q.numel() / q.size(0)));
      } else {
        pplr.emplace_back();
        row.push_back(1);
      }
    }
  }

  void set_owners(std::vector<Tensor> o) {
    TORCH_CHECK(o.size() == params.size(), "AdamPlan.set_owners: one Parameter per tensor");
    owners = std::move(o);
  }
  // The step with the gradients read from the Parameters: false (nothing done) if a Parameter has no dense gradient or no
  // longer is the tensor this plan was built for — the caller then takes the general path.
  bool step_owned(const std::vector<double>& lr, const std::vector<int64_t>& step_counts) {
    const size_t n = params.size();
    if (owners.size() != n) return false;
    std::vector<Tensor> grads(n);
    for (size_t t = 0; t < n; ++t) {
      const Tensor& g = owners[t].grad();
      if (!g.defined() || g.is_sparse() || owners[t].data_ptr() != params[t].data_ptr() || g.numel() != numel[t]) return false;
      grads[t] = g;
    }
    this->step(grads, lr, step_counts);
    return true;
  }
  // optimizer.zero_grad(set_to_none=True) for the plan's Parameters
  void zero_owned() {
    for (Tensor& o : owners) o.mutable_grad().reset();
  }

  // grads[t]: the tensor's .grad; lr[t], step[t] (1-based, already incremented by the caller)
  void step(const std::vector<Tensor>& grads, const std::vector<double>& lr, const std::vector<int64_t>& step) {
    const size_t n = params.size();
    TORCH_CHECK(grads.size() == n && lr.size() == n && step.size() == n, "AdamPlan.step: list lengths");
    HostClock clock(&g_host_us[4]);
    const DeviceScope dev(params[0]);
    int64_t nm[8]; int32_t rw[8], st[8], gidx[8];
    float* pp[8]; const float* gg[8]; float* mm[8]; float* vv[8]; const float* ll[8]; float lrs[8];
    std::vector<Tensor> keep;
    int64_t n_flagged = 0;
    GateRecord rec;
    {
      std::lock_guard<std::mutex> lock(g_gate_mutex);
      rec = g_gates;
    }
    for (size_t t = 0; t < n; ++t) {
      Tensor g = grads[t];
      TORCH_CHECK(g.defined() && g.numel() == numel[t] && g.device() == params[t].device(), "AdamPlan.step: gradient ", t, " does not match its parameter");
      bool mine = false;
      if (rec.valid && g.scalar_type() == at::kFloat && g.is_contiguous() && g.storage_offset() == 0) {
        const c10::StorageImpl* impl = g.storage().unsafeGetStorageImpl();
        for (int k = 0; k < (int)rec.storage.size() && !mine; ++k) {
          auto strong = rec.storage[k].lock();
          if (strong && strong.get() == impl && rec.numel[k] == g.numel() && rec.version[k] == (int64_t)g._version()) { gidx[t] = k; mine = true; }
        }
      }
      if (!mine) {
        std::ostringstream why;
        why << "tensor " << t << ": record " << (rec.valid ? "valid" : "absent") << ", offset " << g.storage_offset() << ", version " << g._version()
            << ", storage " << (const void*)g.storage().unsafeGetStorageImpl() << ", numel " << g.numel();
        last_gate_note = why.str();
      }
      if (!mine) gidx[t] = -1;   // the library sums this tensor's squared gradient itself
      n_flagged += mine ? 1 : 0;
      if (g.scalar_type() != at::kFloat || !g.is_contiguous()) { g = g.to(at::kFloat).contiguous(); keep.push_back(g); }
      nm[t] = numel[t]; rw[t] = row[t]; st[t] = (int32_t)step[t]; lrs[t] = (float)lr[t];
      pp[t] = params[t].data_ptr<float>(); gg[t] = g.data_ptr<float>(); mm[t] = exp_avg[t].data_ptr<float>();
      vv[t] = exp_avg_sq[t].data_ptr<float>(); ll[t] = pplr[t].defined() ? pplr[t].data_ptr<float>() : nullptr;
    }
    const bool gated = n_flagged > 0;
    Tensor scratch;
    if (n_flagged < (int64_t)n) scratch = at::empty({8}, params[0].options());
    last_used_gates = n_flagged;
    bool moments_touched = !live.defined();
    if (moment_version.size() != 2 * n) moment_version.assign(2 * n, -1);
    for (size_t t = 0; t < n; ++t) {
      const int64_t v0 = (int64_t)exp_avg[t]._version(), v1 = (int64_t)exp_avg_sq[t]._version();
      moments_touched = moments_touched || v0 != moment_version[2 * t] || v1 != moment_version[2 * t + 1];
      moment_version[2 * t] = v0; moment_version[2 * t + 1] = v1;
    }
    if (!live.defined()) live = at::zeros({16}, params[0].options().dtype(at::kInt));
    else if (moments_touched) live.zero_();
    if (++seq == 0u) seq = 1u;
    check(g_abi.adam_multi_step(dev.stream, (int)n, nm, rw, pp, gg, mm, vv, ll, lrs, (float)beta1, (float)beta2, (float)eps, st,
                                scratch.defined() ? scratch.data_ptr<float>() : nullptr, gated ? rec.gate : nullptr, gated ? gidx : nullptr,
                                reinterpret_cast<uint32_t*>(live.data_ptr<int32_t>()), seq),
          "adam_multi_step");
  }
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled PyTorch binding of libmi355gs.so's drop-in operators (no compute of its own)";
  m.def("bind", &bind_abi, "hand over the C-ABI entry points (name -> address) of the loaded libmi355gs build; allow_cpu_tensors: test tier only");
  m.def("l1_loss", &l1_loss, "utils/loss_utils.py::l1_loss as one node (mi355gs_l1_loss_forward / _backward)");
  m.def("render_posed", &render_posed);
  m.def("forward_owns_scratch", [](bool on) { const bool was = g_forward_owns_scratch; g_forward_owns_scratch = on; return was; },
        "A/B switch: true (default) = a forward that a backward will follow allocates the backward's accumulators and has the projection kernel clear them");
  m.def("render_only", [](bool on) { const bool was = g_render_only_when_no_grad; g_render_only_when_no_grad = on; return was; },
        "A/B switch: true (default) = a forward no backward can follow takes the render-only stage 2 (mi355gs_raster_forward_render_only)");
  m.def("wait_for_words", &wait_for_words, "spin (GIL released) until no element of an int32 pinned host tensor holds the sentinel; stream wait after timeout_us");
  m.def("pose_row", &pose_row, "GaussianModel.get_RT: row `index` of the [views, 7] pose table as a node the render node's backward cooperates with");
  m.def("rasterize", &rasterize);
  m.def("l1_ssim_loss", &l1_ssim_loss);
  m.def("loss_pair_forward", &loss_pair_forward, "-> [means (l1, ssim: filled by the first loss_affine), partial sums, d(ssim_mean)/dimg1, img1, img2 (contiguous)]: one launch, no autograd node");
  m.def("backward_inline", [](bool on) { const bool was = g_backward_inline; g_backward_inline = on; return was; },
        "A/B switch: loss_affine_backward runs the autograd engine on the calling thread (thread-local multithreading off for the call)");
  m.def("wait_for_loss", &wait_for_loss, "the value the program kernel stored to a pinned host slot under this ticket, or None (timeout / slot taken over)");
  m.def("loss_affine_backward", &loss_affine_backward, "loss_affine + the engine run of loss.backward() in one call (root gradient: a cached 1)");
  m.def("loss_affine", &loss_affine, "the recorded scalar expression over (l1_mean, ssim_mean) as ONE node on `image`");
  m.def("fused_ssim", &fused_ssim);
  m.def("host_times_us", [](bool reset) {
    std::vector<double> v(g_host_us, g_host_us + 6);
    if (reset) for (double& x : g_host_us) x = 0;
    return v;
  }, "accumulated host microseconds: render fwd (incl. the count wait), render bwd, loss fwd, loss bwd, adam step, count wait");
  m.def("forget_gates", []() { std::lock_guard<std::mutex> lock(g_gate_mutex); g_gates = GateRecord(); });
  m.def("shared_zero_grad", [](bool on) {
    std::lock_guard<std::mutex> lock(g_zero_mutex);
    const bool old = g_share_zero_grad;
    g_share_zero_grad = on;
    if (!on) g_zero_pool.clear();
    return old;
  }, "f_rest's all-zero gradient below its SH degree: true (default) = aliases of one persistent zero buffer, false = a fresh zeros tensor per backward");
  py::class_<AdamPlan>(m, "AdamPlan")
      .def(py::init<std::vector<Tensor>, std::vector<Tensor>, std::vector<Tensor>, std::vector<c10::optional<Tensor>>, double, double, double>())
      .def("step", &AdamPlan::step)
      .def("set_owners", &AdamPlan::set_owners)
      .def("step_owned", &AdamPlan::step_owned)
      .def("zero_owned", &AdamPlan::zero_owned)
      .def_readonly("last_used_gates", &AdamPlan::last_used_gates)
      .def_readonly("last_gate_note", &AdamPlan::last_gate_note);
}
