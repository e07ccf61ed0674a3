#!/usr/bin/env python3
"""Which HIP runtime sees the GPU on this box?  Each candidate is tried in its own torch-free
subprocess (pure ctypes): the system ROCm runtime libspng_mi355.so links against, and the runtime
bundled with the torch wheel."""
import os
import subprocess
import sys

CODE = r"""
import ctypes, sys, os
path = sys.argv[1]
lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
n = ctypes.c_int(-1)
lib.hipGetErrorString.restype = ctypes.c_char_p
e = lib.hipGetDeviceCount(ctypes.byref(n))
v = ctypes.c_int(0)
lib.hipRuntimeGetVersion(ctypes.byref(v))
print(path, "-> hipGetDeviceCount rc", e, lib.hipGetErrorString(e).decode(), "count", n.value, "runtime", v.value)
"""


def main():
    import importlib.util
    spec = importlib.util.find_spec("torch")
    tlib = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    for path in ("/opt/rocm/lib/libamdhip64.so", tlib):
        for env_extra in ({}, {"HSA_ENABLE_DEBUG": "0", "AMD_LOG_LEVEL": "1"}):
            env = dict(os.environ, **env_extra)
            r = subprocess.run([sys.executable, "-c", CODE, path], capture_output=True, text=True, env=env)
            print(r.stdout.strip(), "|", r.stderr.strip()[-600:].replace("\n", " / "))
    for cmd in ("ls -la /dev/kfd /dev/dri", "/opt/rocm/bin/rocminfo | head -30", "env | grep -E 'HSA|HIP|ROC|LD_LIB'",
                "ls /opt/amdgpu/share/libdrm 2>&1 | head", "cat /sys/module/amdgpu/version 2>&1"):
        r = subprocess.run(cmd, shell=True, capture_output=True, text=True)
        print("$", cmd, "\n", (r.stdout + r.stderr)[-1500:])


if __name__ == "__main__":
    main()
