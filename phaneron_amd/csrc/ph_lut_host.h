// ph_lut_host.h - host API of the LUT compressor (ph_lut.cpp)
#pragma once
#include <stdint.h>
#include <vector>

#include "ph_lut.h"

namespace ph {

struct LutHostInfo {
  uint32_t bytes = 0, shift = 0, first = 0, n_anchors = 0, delta_off = 0, hole = 0;  // bytes / delta_off: LDS footprint / address (hole included)
  float bias = 0;
};

// max LDS a single workgroup can hold on gfx950 is 160 KiB; tables must leave room for nothing else
constexpr uint32_t kLutMaxLdsBytes = 160 * 1024;

// Returns false when the table cannot be represented exactly within max_bytes.
bool lut_compress(const float *lut65536, uint32_t max_bytes, std::vector<uint32_t> &blob, LutHostInfo &info);
// The kernel-side view of a compressed table whose blob lives at `blob_dev`.
LutView lut_view(const LutHostInfo &info, const void *blob_dev);

}  // namespace ph
