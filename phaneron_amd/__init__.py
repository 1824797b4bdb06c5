"""phaneron_amd - MI355X-native (gfx950) replacement for the per-pixel hot path of
Streampunk/phaneron: hand-written HIP kernels behind a C ABI (include/phaneron_hip.h).

  phaneron_amd.capi   ctypes binding of libphaneron_hip.so (tests, bench, Python callers)
  phaneron_amd.build  in-tree hipcc build
  node/               the N-API addon + JS operator layer that mirrors the reference's
                      src/process and src/clJobQueue.ts (the reference's own host language)

There is no CPU fallback: importing capi without the built library, or creating a context
without a HIP device, raises.
"""
__version__ = "0.1.0"
