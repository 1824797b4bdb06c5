"""Build node/phaneron_napi.node: the raw N-API addon over libphaneron_hip.so (gcc, no node-gyp).
Skipped (with a message) where the node headers are not installed."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "phaneron_napi.node")
INC = "/usr/include/node"


def build(force=False):
    if not os.path.exists(os.path.join(INC, "node_api.h")):
        print("node_api.h not found under %s: N-API addon not built" % INC)
        return None
    src = os.path.join(HERE, "ph_napi.c")
    lib_dir = os.path.normpath(os.path.join(HERE, "..", "phaneron_amd", "lib"))
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) > os.path.getmtime(src):
        return OUT
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=gnu99", "-Wall", "-DNAPI_VERSION=6", "-I", INC, src, "-o", OUT,
           "-L", lib_dir, "-lphaneron_hip", "-Wl,-rpath,$ORIGIN/../phaneron_amd/lib"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
    sys.exit(0)
