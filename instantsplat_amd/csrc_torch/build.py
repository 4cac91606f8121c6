"""Builds instantsplat_amd/lib/_mi355gs_torch.so from binding.cpp with g++ against the installed PyTorch (ROCm build).
Host code only — no kernels: the module calls the C ABI of libmi355gs.so through addresses handed over at run time.

    python instantsplat_amd/csrc_torch/build.py [--force]
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "lib", "_mi355gs_torch.so")
SRC = os.path.join(HERE, "binding.cpp")


def build(force: bool = False) -> str:
    deps = [SRC, os.path.join(HERE, "..", "..", "include", "mi355gs.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    lib_dir = ce.library_paths()[0]
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-DTORCH_EXTENSION_NAME=_mi355gs_torch", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    for inc in ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(rocm, "include")]:
        cmd += ["-isystem", inc]
    cmd += [SRC, "-o", OUT, f"-L{lib_dir}", f"-Wl,-rpath,{lib_dir}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_hip", "-ltorch_python"]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
