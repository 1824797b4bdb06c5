// ph_program.h - device-free kernel selection for createProgram (ph_program.cpp)
#pragma once
#include <string>

namespace ph {

enum KernelId {
  K_PACK_READ,   // any pack format other than v210 (ProgramChoice::format)
  K_PACK_WRITE,
  K_V210_READ,
  K_V210_WRITE,
  K_YADIF,
  K_YADIF_PAIR,       // both send_field outputs in one pass (ph_yadif_pair)
  K_V210_YADIF_PAIR,  // ToRGBA of the window + both outputs, n layers (ph_v210_yadif_pair)
  K_V210_READ_BATCH,  // ToRGBA of n frames of one size and colour recipe in one launch (ph_v210_read_batch)
  K_CHAN_COMPOSE,     // a channel's frame straight from its v210 sources, n layers (ph_chan_compose_v210)
  K_COMPOSE_UP,       // [transform] x n -> combine_n -> v210 write for layers enlarged 2x or more (ph_compose_up_write_v210)
  K_COMPOSE_V210,     // [transform] x n (+ wipes) -> combine_n -> v210 write in one launch (ph_compose_wipe_write_v210)
  K_TRANSFORM,
  K_RESIZE,
  K_COMBINE,
  K_DISSOLVE,
  K_TWIPE,
  K_MIXER,
  K_WIPE,
  K_RGB_UNPACK,       // extension: a packed f32 RGB image (12 bytes per pixel) made the f32 RGBA image its buffer is declared as, in place (ph_image_unpack_rgb)
  K_FUSED_V210  // extension: v210 x N -> read, combine_N, write in one launch (ph_fused_v210_combine)
};

struct ProgramChoice {
  KernelId id;
  int n_layers;        // combine_N / fused_v210_combine_N
  int format;          // PH_FMT_* of a read / write program
  std::string kernel;  // "v210_read", "yuv420p_write", "combine_4", ...
  int how;             // PH_RESOLVED_*
};

// PH_OK, or PH_E_UNKNOWN_KERNEL / PH_E_INVALID with a message in err
int resolve_program(const char *kernel_src, const char *name, ProgramChoice &choice, std::string &err);

}  // namespace ph
