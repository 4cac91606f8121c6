"""tests/golden/loss_utils_vectors.npz: the reference's own `utils/loss_utils.py` (l1_loss :39-40, l2_loss :42-43, ssim :55-85,
l1_loss_mask :17-23, ssim_loss_mask :25-37, gaussian / create_window :45-53, _ssim :65-85) run from /root/reference on seeded inputs (build container only): values, and the gradient of
1.7 * l1_loss with respect to its first argument (inputs with exact ties, where abs'(0) = 0 matters).
Run:  python tests/golden/make_golden_loss_utils.py"""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_loss_utils", "/root/reference/utils/loss_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

g = torch.Generator().manual_seed(123)
out = {}
for k, shape in enumerate(((3, 37, 53), (3, 64, 64), (1, 3, 40, 41), (5000,), (3, 128, 96))):
    a = torch.rand(shape, generator=g).requires_grad_(True)
    b = torch.rand(shape, generator=g)
    b.view(-1)[::7] = a.detach().view(-1)[::7]          # exact zeros of the difference
    v = ref.l1_loss(a, b)
    (v * 1.7).backward()
    out[f"l1_{k}_a"], out[f"l1_{k}_b"] = a.detach().numpy(), b.numpy()
    out[f"l1_{k}_value"], out[f"l1_{k}_grad"] = v.detach().numpy(), a.grad.numpy()
    out[f"l2_{k}_value"] = ref.l2_loss(a.detach(), b).numpy()
x, y = torch.rand(1, 3, 48, 40, generator=g), torch.rand(1, 3, 48, 40, generator=g)
mask = (torch.rand(1, 3, 48, 40, generator=g) > 0.4).float()
out.update(ssim_x=x.numpy(), ssim_y=y.numpy(), ssim_mask=mask.numpy(), ssim_11=ref.ssim(x, y).numpy(), ssim_7=ref.ssim(x, y, window_size=7).numpy(),
           ssim_11_per_image=ref.ssim(x, y, size_average=False).numpy(), ssim_3d=ref.ssim(x[0], y[0]).numpy(),
           l1_mask=ref.l1_loss_mask(x, y, mask).numpy(),
           # the rest of the module's names (render.py:30 imports ssim_loss_mask; :25-37, :45-53)
           ssim_mask_11=ref.ssim_loss_mask(x, y, mask).numpy(), ssim_mask_7_per_image=ref.ssim_loss_mask(x, y, mask, window_size=7, size_average=False).numpy(),
           gaussian_11=ref.gaussian(11, 1.5).numpy(), window_7_3=ref.create_window(7, 3).numpy(),
           ssim_core_7=ref._ssim(x, y, ref.create_window(7, 3), 7, 3, True).numpy(), names=np.array(sorted(n for n in dir(ref) if callable(getattr(ref, n)) and getattr(getattr(ref, n), "__module__", "") == "ref_loss_utils")))
np.savez_compressed(os.path.join(HERE, "loss_utils_vectors.npz"), **out)
print("wrote", len(out), "arrays")
