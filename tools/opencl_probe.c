/* opencl_probe.c - could the reference's OpenCL image kernels have run on this box at all?
 *
 * The reference samples with read_imagef(NORMALIZED | CLAMP | LINEAR) (src/process/transform.ts:25-28,57 and
 * resize.ts:25-28) and OpenCL leaves the filter's precision to the implementation.  This probe records what the
 * OpenCL runtime on the GPU box offers: every platform / device, CL_DEVICE_IMAGE_SUPPORT, and - if some device does
 * support images - the results of a three-line read_imagef(LINEAR) kernel next to the f32 evaluation this repository
 * pins the sampler to, so that a real implementation's filter can be compared bit for bit.
 *
 *   gcc -O1 tools/opencl_probe.c -o tools/opencl_probe -lOpenCL -lm && tools/opencl_probe > profiles/r03_opencl_probe.txt
 *
 * Own code; carries no text of the reference.  Exit code 0 whatever it finds: the output is the record. */
#define CL_TARGET_OPENCL_VERSION 120
#include <CL/cl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const char *kProbeSrc =
    "__constant sampler_t lin = CLK_NORMALIZED_COORDS_TRUE | CLK_ADDRESS_CLAMP | CLK_FILTER_LINEAR;\n"
    "__kernel void probe(__read_only image2d_t img, __global const float2 *pos, __global float4 *out) {\n"
    "  int i = get_global_id(0);\n"
    "  out[i] = read_imagef(img, lin, pos[i]);\n"
    "}\n";

static uint64_t rng_state = 0x5EED0003ull;
static uint32_t rnd(void) {  /* splitmix64 */
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}
static float unit(void) { return (float)(rnd() >> 8) * (1.0f / 16777216.0f); }

/* the evaluation DESIGN.md section 2 fixes: weights first, ((w00 t00 + w10 t10) + w01 t01) + w11 t11, no fma */
static void pinned_sample(const float *img, int w, int h, float s, float t, float out[4]) {
  const float u = s * (float)w, v = t * (float)h;
  const float fu = u - 0.5f, fv = v - 0.5f;
  const float flu = floorf(fu), flv = floorf(fv);
  const int i0 = (int)flu, j0 = (int)flv, i1 = i0 + 1, j1 = j0 + 1;
  const float a = fu - flu, b = fv - flv, oma = 1.0f - a, omb = 1.0f - b;
  const volatile float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
  for (int c = 0; c < 4; ++c) {
#define TEXEL(i, j) (((i) >= 0 && (i) < w && (j) >= 0 && (j) < h) ? img[((j) * w + (i)) * 4 + c] : 0.0f)
    volatile float p0 = w00 * TEXEL(i0, j0), p1 = w10 * TEXEL(i1, j0), p2 = w01 * TEXEL(i0, j1), p3 = w11 * TEXEL(i1, j1);
    volatile float acc = p0 + p1;
    acc = acc + p2;
    acc = acc + p3;
    out[c] = acc;
#undef TEXEL
  }
}

static long ulp_distance(float x, float y) {
  int32_t a, b;
  memcpy(&a, &x, 4), memcpy(&b, &y, 4);
  if (a < 0) a = (int32_t)0x80000000 - a;
  if (b < 0) b = (int32_t)0x80000000 - b;
  long d = (long)a - (long)b;
  return d < 0 ? -d : d;
}

static void run_image_probe(cl_platform_id plat, cl_device_id dev) {
  cl_int err;
  cl_context_properties props[] = {CL_CONTEXT_PLATFORM, (cl_context_properties)plat, 0};
  cl_context ctx = clCreateContext(props, 1, &dev, NULL, NULL, &err);
  if (err) { printf("    image probe: clCreateContext failed (%d)\n", err); return; }
  cl_command_queue q = clCreateCommandQueue(ctx, dev, 0, &err);
  if (err) { printf("    image probe: clCreateCommandQueue failed (%d)\n", err); return; }
  enum { W = 37, H = 23, N = 65536 };
  float *img = malloc(sizeof(float) * W * H * 4), *pos = malloc(sizeof(float) * 2 * N), *out = malloc(sizeof(float) * 4 * N);
  for (int i = 0; i < W * H * 4; ++i) img[i] = unit() * 1.25f - 0.125f;
  for (int i = 0; i < N; ++i) {  /* inside, on texel centres, on the border and outside */
    pos[2 * i] = unit() * 1.2f - 0.1f, pos[2 * i + 1] = unit() * 1.2f - 0.1f;
    if (i % 7 == 0) pos[2 * i] = ((float)(rnd() % (W + 2)) - 0.5f) / (float)W;
    if (i % 11 == 0) pos[2 * i + 1] = ((float)(rnd() % (H + 2)) - 0.5f) / (float)H;
  }
  cl_image_format fmt = {CL_RGBA, CL_FLOAT};
  cl_image_desc desc;
  memset(&desc, 0, sizeof desc);
  desc.image_type = CL_MEM_OBJECT_IMAGE2D, desc.image_width = W, desc.image_height = H;
  cl_mem im = clCreateImage(ctx, CL_MEM_READ_ONLY | CL_MEM_COPY_HOST_PTR, &fmt, &desc, img, &err);
  if (err) { printf("    image probe: clCreateImage(RGBA, FLOAT) failed (%d)\n", err); return; }
  cl_mem dpos = clCreateBuffer(ctx, CL_MEM_READ_ONLY | CL_MEM_COPY_HOST_PTR, sizeof(float) * 2 * N, pos, &err);
  cl_mem dout = clCreateBuffer(ctx, CL_MEM_WRITE_ONLY, sizeof(float) * 4 * N, NULL, &err);
  cl_program prog = clCreateProgramWithSource(ctx, 1, &kProbeSrc, NULL, &err);
  err = clBuildProgram(prog, 1, &dev, "-cl-std=CL1.2", NULL, NULL);
  if (err) {
    char log[4096] = {0};
    clGetProgramBuildInfo(prog, dev, CL_PROGRAM_BUILD_LOG, sizeof log - 1, log, NULL);
    printf("    image probe: clBuildProgram failed (%d): %s\n", err, log);
    return;
  }
  cl_kernel k = clCreateKernel(prog, "probe", &err);
  clSetKernelArg(k, 0, sizeof im, &im), clSetKernelArg(k, 1, sizeof dpos, &dpos), clSetKernelArg(k, 2, sizeof dout, &dout);
  size_t gws = N;
  err = clEnqueueNDRangeKernel(q, k, 1, NULL, &gws, NULL, 0, NULL, NULL);
  if (!err) err = clEnqueueReadBuffer(q, dout, CL_TRUE, 0, sizeof(float) * 4 * N, out, 0, NULL, NULL);
  if (err) { printf("    image probe: launch / read back failed (%d)\n", err); return; }
  long worst = 0, differing = 0;
  double worst_abs = 0;
  for (int i = 0; i < N; ++i) {
    float want[4];
    pinned_sample(img, W, H, pos[2 * i], pos[2 * i + 1], want);
    for (int c = 0; c < 4; ++c) {
      long d = ulp_distance(want[c], out[4 * i + c]);
      double ad = fabs((double)want[c] - (double)out[4 * i + c]);
      if (d) ++differing;
      if (d > worst) worst = d;
      if (ad > worst_abs) worst_abs = ad;
    }
  }
  printf("    image probe: read_imagef(LINEAR) on a %dx%d RGBA f32 image, %d positions: %ld of %d components differ from the pinned f32 "
         "evaluation, worst %ld ULP, worst absolute %.3g\n", W, H, N, differing, 4 * N, worst, worst_abs);
  for (int i = 0; i < 4; ++i)
    printf("      sample %d: pos (%a, %a) -> device (%a, %a, %a, %a)\n", i, pos[2 * i], pos[2 * i + 1], out[4 * i], out[4 * i + 1],
           out[4 * i + 2], out[4 * i + 3]);
}

int main(void) {
  cl_platform_id plats[8];
  cl_uint np = 0;
  cl_int err = clGetPlatformIDs(8, plats, &np);
  printf("opencl_probe: clGetPlatformIDs -> %d, %u platform(s)\n", err, np);
  int image_devices = 0, devices_total = 0;
  for (cl_uint p = 0; p < np && p < 8; ++p) {
    char name[256] = {0}, ver[256] = {0};
    clGetPlatformInfo(plats[p], CL_PLATFORM_NAME, sizeof name - 1, name, NULL);
    clGetPlatformInfo(plats[p], CL_PLATFORM_VERSION, sizeof ver - 1, ver, NULL);
    cl_device_id devs[16];
    cl_uint nd = 0;
    err = clGetDeviceIDs(plats[p], CL_DEVICE_TYPE_ALL, 16, devs, &nd);
    if (err == CL_DEVICE_NOT_FOUND) nd = 0;
    printf("platform %u: %s | %s | clGetDeviceIDs(ALL) -> %d, %u device(s)\n", p, name, ver, err, nd);
    for (cl_uint d = 0; d < nd && d < 16; ++d) {
      char dn[256] = {0}, dv[256] = {0};
      cl_device_type type = 0;
      cl_bool images = 0;
      size_t w2d = 0, h2d = 0;
      cl_uint samplers = 0;
      clGetDeviceInfo(devs[d], CL_DEVICE_NAME, sizeof dn - 1, dn, NULL);
      clGetDeviceInfo(devs[d], CL_DEVICE_VERSION, sizeof dv - 1, dv, NULL);
      clGetDeviceInfo(devs[d], CL_DEVICE_TYPE, sizeof type, &type, NULL);
      clGetDeviceInfo(devs[d], CL_DEVICE_IMAGE_SUPPORT, sizeof images, &images, NULL);
      clGetDeviceInfo(devs[d], CL_DEVICE_IMAGE2D_MAX_WIDTH, sizeof w2d, &w2d, NULL);
      clGetDeviceInfo(devs[d], CL_DEVICE_IMAGE2D_MAX_HEIGHT, sizeof h2d, &h2d, NULL);
      clGetDeviceInfo(devs[d], CL_DEVICE_MAX_SAMPLERS, sizeof samplers, &samplers, NULL);
      printf("  device %u: %s | %s | type %s%s%s | CL_DEVICE_IMAGE_SUPPORT = %u | IMAGE2D_MAX %zux%zu | MAX_SAMPLERS %u\n", d, dn, dv,
             (type & CL_DEVICE_TYPE_GPU) ? "GPU" : "", (type & CL_DEVICE_TYPE_CPU) ? "CPU" : "",
             (type & CL_DEVICE_TYPE_ACCELERATOR) ? "ACCELERATOR" : "", (unsigned)images, w2d, h2d, samplers);
      ++devices_total;
      if (images) {
        ++image_devices;
        run_image_probe(plats[p], devs[d]);
      }
    }
  }
  printf("summary: %d OpenCL device(s), %d with image support\n", devices_total, image_devices);
  if (!image_devices)
    printf("conclusion: no OpenCL device on this box can run the reference's image2d_t kernels (yadif, transform, resize, combine, "
           "transition, mixer, wipe); read_imagef(LINEAR) cannot be pinned to a reference run here.\n");
  return 0;
}
