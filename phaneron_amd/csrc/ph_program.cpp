// ph_program.cpp - which precompiled gfx950 kernel a nodencl `createProgram(kernelSrc, {name})` call means.
// Device-free: pure string work, usable (and tested) without a GPU.
//
// The reference compiles OpenCL C text at run time (packer.ts:97-103, imageProcess.ts:69-72).  This library
// ships its kernels precompiled, so the text only SELECTS one.  Image kernels are named uniquely
// (yadif, transform, resize, combine_N, transition_dissolve, transition_wipe, mixer, wipe).  All seven pack
// formats call their kernels `read` / `write`; they are told apart, in this order, by
//   1. a "phaneron:<fmt>" tag instead of source text (this repo's own callers),
//   2. the fingerprint of the text (FNV-1a/64 over the non-whitespace bytes) against the seven sources of
//      reference v0.0.15 - exact, no guessing (refbuild/kernel_hashes.py prints the table in the build container),
//   3. the argument list of the named kernel, plus - only where two formats share an argument list
//      (yuv422p8 / yuv420p, rgba8 / bgra8) - one structural probe of that kernel's BODY, comments removed.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/phaneron_hip.h"
#include "ph_kernels.h"
#include "ph_program.h"

namespace ph {

namespace {

const char *const kFmtNames[7] = {"v210", "yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"};

struct KnownSource {
  unsigned long long fingerprint;
  int format;
};
const KnownSource kKnownSources[] = {
    {0x50381d4911dc3c85ull, PH_FMT_V210},       // src/process/v210.ts
    {0x2fe10e00781c536eull, PH_FMT_YUV422P10},  // src/process/yuv422p10.ts
    {0x63cef3e5dedd7ec0ull, PH_FMT_YUV422P8},   // src/process/yuv422p8.ts
    {0xe84729134a6e8e80ull, PH_FMT_YUV420P},    // src/process/yuv420p.ts
    {0x2de5f250398843f1ull, PH_FMT_NV12},       // src/process/nv12.ts
    {0x4cf02e14f968c0b4ull, PH_FMT_RGBA8},      // src/process/rgba8.ts
    {0xaef4a0044615ca5cull, PH_FMT_BGRA8},      // src/process/bgra8.ts
};

unsigned long long fingerprint(const char *s) {
  unsigned long long h = 0xCBF29CE484222325ull;
  for (; *s; ++s) {
    const unsigned char c = (unsigned char)*s;
    if (c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v') continue;
    h = (h ^ c) * 0x100000001B3ull;
  }
  return h;
}

// OpenCL C text without // and /* */ comments (string literals do not occur in kernels)
std::string strip_comments(const char *s) {
  std::string o;
  for (size_t i = 0; s[i];) {
    if (s[i] == '/' && s[i + 1] == '/') {
      while (s[i] && s[i] != '\n') ++i;
    } else if (s[i] == '/' && s[i + 1] == '*') {
      i += 2;
      while (s[i] && !(s[i] == '*' && s[i + 1] == '/')) ++i;
      if (s[i]) i += 2;
      o += ' ';
    } else {
      o += s[i++];
    }
  }
  return o;
}

bool is_ident(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; }

// does `hay` contain `word` as a whole identifier?
bool has_word(const std::string &hay, const char *word) {
  const size_t n = strlen(word);
  for (size_t p = hay.find(word); p != std::string::npos; p = hay.find(word, p + 1))
    if ((p == 0 || !is_ident(hay[p - 1])) && (p + n >= hay.size() || !is_ident(hay[p + n]))) return true;
  return false;
}

// `__kernel void <name>(<signature>) { <body> }` of comment-free text; false if the kernel is not there
bool find_kernel(const std::string &src, const char *name, std::string &signature, std::string &body) {
  const std::string pat = std::string("void ") + name;
  for (size_t k = src.find(pat); k != std::string::npos; k = src.find(pat, k + 1)) {
    size_t p = k + pat.size();
    while (p < src.size() && (src[p] == ' ' || src[p] == '\t' || src[p] == '\n' || src[p] == '\r')) ++p;
    if (p >= src.size() || src[p] != '(') continue;  // e.g. `void readX(` or a longer name
    const size_t close = src.find(')', p);
    if (close == std::string::npos) return false;
    signature = src.substr(p + 1, close - p - 1);
    const size_t open = src.find('{', close);
    if (open == std::string::npos) return false;
    int depth = 0;
    size_t e = open;
    for (; e < src.size(); ++e) {
      if (src[e] == '{') ++depth;
      if (src[e] == '}' && --depth == 0) break;
    }
    body = src.substr(open, e - open + 1);
    return true;
  }
  return false;
}

int format_by_signature(const char *src_raw, const char *name, bool is_read) {
  const std::string src = strip_comments(src_raw);
  std::string sig, body;
  if (!find_kernel(src, name, sig, body)) return -1;
  const bool col = has_word(sig, "colMatrix");
  if (has_word(sig, "uint4") && col) return PH_FMT_V210;                 // v210.ts:25-30,113-118
  if (has_word(sig, is_read ? "inputC" : "outputC")) return PH_FMT_NV12;  // nv12.ts:25-31
  if (has_word(sig, "ushort8")) return PH_FMT_YUV422P10;                 // yuv422p10.ts:25
  if (has_word(sig, is_read ? "inputU" : "outputU")) {
    // yuv422p8.ts:25 and yuv420p.ts:25 share an argument list; only the 4:2:0 kernels address line PAIRS
    // through a chroma offset of their own (yuv420p.ts:44-50,153-160)
    return has_word(body, is_read ? "inOffUV" : "outOffUV") ? PH_FMT_YUV420P : PH_FMT_YUV422P8;
  }
  if (has_word(sig, "uchar4") && !col) {
    // rgba8.ts:25 and bgra8.ts:25 share an argument list; the component order shows in the statement that fills
    // channel 0 on the far side of the table: read  `rgb.s0 = gammaLut[...(<px>.s0 | .s2 ...)]` (:30),
    // write `<px>.s0 = convert_uchar_sat_rte(rgb_f.s0 | .s2 ...)` (:71)
    const char *needle = is_read ? "gammaLut" : "convert_uchar";
    for (size_t a = body.find(".s0 ="); a != std::string::npos; a = body.find(".s0 =", a + 1)) {
      const size_t end = body.find(';', a);
      const std::string rhs = body.substr(a + 5, end == std::string::npos ? std::string::npos : end - a - 5);
      if (rhs.find(needle) == std::string::npos) continue;
      if (rhs.find(".s2") != std::string::npos) return PH_FMT_BGRA8;
      if (rhs.find(".s0") != std::string::npos) return PH_FMT_RGBA8;
      return -1;
    }
    return -1;
  }
  return -1;
}

int set(ProgramChoice &c, KernelId id, const std::string &kernel, int how) {
  c.id = id, c.kernel = kernel, c.how = how;
  return PH_OK;
}

}  // namespace

int resolve_program(const char *src, const char *name, ProgramChoice &c, std::string &err) {
  c = ProgramChoice{K_V210_READ, 0, PH_FMT_V210, "", PH_RESOLVED_BY_NAME};
  if (!name) {
    err = "createProgram: kernel name missing";
    return PH_E_INVALID;
  }
  const bool tagged = src && 0 == strncmp(src, "phaneron:", 9);
  if (0 == strcmp(name, "read") || 0 == strcmp(name, "write")) {
    const bool is_read = name[0] == 'r';
    int fmt = -1, how = PH_RESOLVED_BY_TAG;
    if (tagged) {
      for (int i = 0; i < 7; ++i)
        if (0 == strcmp(src + 9, kFmtNames[i])) fmt = i;
    } else if (src) {
      const unsigned long long fp = fingerprint(src);
      how = PH_RESOLVED_BY_TEXT;
      for (const KnownSource &k : kKnownSources)
        if (k.fingerprint == fp) fmt = k.format;
      if (fmt < 0) fmt = format_by_signature(src, name, is_read), how = PH_RESOLVED_BY_SIGNATURE;
    }
    if (fmt < 0) {
      err = std::string("cannot tell which pack format the '") + name + "' kernel belongs to";
      return PH_E_UNKNOWN_KERNEL;
    }
    c.format = fmt;
    if (fmt == PH_FMT_V210) return set(c, is_read ? K_V210_READ : K_V210_WRITE, is_read ? "v210_read" : "v210_write", how);
    return set(c, is_read ? K_PACK_READ : K_PACK_WRITE, std::string(kFmtNames[fmt]) + (is_read ? "_read" : "_write"), how);
  }
  const int how = tagged ? PH_RESOLVED_BY_TAG : PH_RESOLVED_BY_NAME;
  auto layers = [&](const char *prefix, int lo, KernelId id) {
    // the whole suffix must be the layer count: "combine_4x" is not combine_4
    const char *digits = name + strlen(prefix);
    char *end = nullptr;
    const long parsed = strtol(digits, &end, 10);
    if (end == digits || *end != '\0' || digits[0] < '0' || digits[0] > '9') {
      err = std::string("unknown kernel '") + name + "' (" + prefix + "<n> takes a plain layer count)";
      return (int)PH_E_UNKNOWN_KERNEL;
    }
    const int n = parsed > 1000 ? 1000 : (int)parsed;
    if (n < lo || n > kMaxLayers) {
      char b[96];
      snprintf(b, sizeof b, "%s%d: %d..%d layers are built", prefix, n, lo, kMaxLayers);
      err = b;
      return (int)PH_E_UNKNOWN_KERNEL;
    }
    c.n_layers = n;
    return set(c, id, name, how);
  };
  if (0 == strcmp(name, "yadif")) return set(c, K_YADIF, name, how);
  if (0 == strcmp(name, "yadif_pair")) return set(c, K_YADIF_PAIR, name, how);
  if (0 == strncmp(name, "v210_yadif_pair_", 16)) return layers("v210_yadif_pair_", 1, K_V210_YADIF_PAIR);
  if (0 == strncmp(name, "v210_read_batch_", 16)) return layers("v210_read_batch_", 1, K_V210_READ_BATCH);
  if (0 == strncmp(name, "compose_write_v210_", 19)) return layers("compose_write_v210_", 1, K_COMPOSE_V210);
  if (0 == strncmp(name, "chan_compose_v210_", 18)) return layers("chan_compose_v210_", 1, K_CHAN_COMPOSE);
  if (0 == strncmp(name, "compose_up_write_v210_", 22)) return layers("compose_up_write_v210_", 1, K_COMPOSE_UP);
  if (0 == strcmp(name, "transform")) return set(c, K_TRANSFORM, name, how);
  if (0 == strcmp(name, "resize")) return set(c, K_RESIZE, name, how);
  if (0 == strncmp(name, "combine_", 8)) return layers("combine_", 2, K_COMBINE);
  // not reference kernels: the headline chain / the field pipeline as one program
  if (0 == strncmp(name, "fused_v210_combine_", 19)) return layers("fused_v210_combine_", 1, K_FUSED_V210);
  if (0 == strcmp(name, "transition_dissolve")) return set(c, K_DISSOLVE, name, how);
  if (0 == strcmp(name, "transition_wipe")) return set(c, K_TWIPE, name, how);
  if (0 == strcmp(name, "mixer")) return set(c, K_MIXER, name, how);
  if (0 == strcmp(name, "wipe")) return set(c, K_WIPE, name, how);
  if (0 == strcmp(name, "rgb_unpack")) return set(c, K_RGB_UNPACK, name, how);
  err = std::string("unknown kernel '") + name + "'";
  return PH_E_UNKNOWN_KERNEL;
}

}  // namespace ph
