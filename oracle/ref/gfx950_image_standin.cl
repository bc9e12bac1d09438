/* gfx950_image_standin.cl -- BUILDER-WRITTEN STAND-IN (test infrastructure; no reference code): the three image built-ins the reference's `logic`
 * kernel calls when it is built with USE_ENV_MAP -- read_imagef(image2d_t, sampler_t, float2), read_imagef(image2d_t, sampler_t, int2) and
 * get_image_dim(image2d_t) (src/env_map.cl:39-53, src/wf_logic.cl:97,232).  gfx950 has no image instructions (CL_DEVICE_IMAGE_SUPPORT = 0; AMD's own
 * read_imagef lowers to llvm.amdgcn.image.sample.*, which this chip cannot execute), so the 16 USE_ENV_MAP variants of `logic` could not run on the
 * MI355X at all (Pin 5).  With these three functions linked in FRONT of AMD's opencl.bc / ocml.bc / ockl.bc (oracle/ref/Makefile, GERULE)
 * everything else of that kernel -- alias sampling, envMapPdf, the MIS weights, sin / cos / acos / atan2 / normalize / length -- meets AMD's own
 * built-in library AS THE KERNEL; what stays on trust is the image filter below, a restatement of OpenCL 1.2 s8.2 (the same restatement as
 * ocl_builtins.c's x86 one, which tests/test_gpu_ocl_builtins.py checks against an independent float64 model).  By the task's rules a stand-in
 * upgrades no pin: these code objects are labelled "stand-in image built-ins" wherever their results are quoted.
 *
 * What the kernel receives where it expects its image2d_t (an 8-byte pointer in the constant address space on amdgcn): a buffer
 * {int width, height, 0, 0; float4 texels[width * height]} written by oracle/ref_gpu.py (upload_envmap).  The sampler argument is ignored: the
 * reference uses exactly one sampler per coordinate type -- float2 coordinates with samplerFloat = NORMALIZED | CLAMP_TO_EDGE | LINEAR, int2 with
 * samplerInt = unnormalised | CLAMP_TO_EDGE | NEAREST (src/env_map.cl:7,10). */
typedef struct { int width, height, pad0, pad1; } flx_img;

static float4 flx_texel(__constant flx_img *img, int i, int j)
{
    i = clamp(i, 0, img->width - 1);
    j = clamp(j, 0, img->height - 1);
    return ((__constant float4 *)(img + 1))[j * img->width + i];
}

int2 flx_get_image_dim(__constant flx_img *img) __asm("_Z13get_image_dim14ocl_image2d_ro");
int2 flx_get_image_dim(__constant flx_img *img) { return (int2)(img->width, img->height); }

/* s8.2: (u, v) = (s w, t h); i0 = floor(u - 0.5), a = frac(u - 0.5); T = (1-a)(1-b) T00 + a(1-b) T10 + (1-a) b T01 + a b T11, indices clamped to the edge */
float4 flx_read_imagef_f(__constant flx_img *img, __constant void *smp, float2 c) __asm("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_f");
float4 flx_read_imagef_f(__constant flx_img *img, __constant void *smp, float2 c)
{
    const float fu = c.x * (float)img->width - 0.5f, fv = c.y * (float)img->height - 0.5f;
    const float i0f = floor(fu), j0f = floor(fv);
    const float a = fu - i0f, b = fv - j0f;
    const int i0 = (int)i0f, j0 = (int)j0f;
    const float4 t00 = flx_texel(img, i0, j0), t10 = flx_texel(img, i0 + 1, j0), t01 = flx_texel(img, i0, j0 + 1), t11 = flx_texel(img, i0 + 1, j0 + 1);
    return (1.0f - a) * (1.0f - b) * t00 + a * (1.0f - b) * t10 + (1.0f - a) * b * t01 + a * b * t11;
}

float4 flx_read_imagef_i(__constant flx_img *img, __constant void *smp, int2 c) __asm("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_i");
float4 flx_read_imagef_i(__constant flx_img *img, __constant void *smp, int2 c) { return flx_texel(img, c.x, c.y); }
