/*
 * gpt.h — C ABI of the MI355X path-tracing integrator.
 *
 * This is the drop-in boundary for the reference's render API
 * (reference src/pathtracer.h:10-12):
 *
 *     void BeginRender(Scene& scene, unsigned width, unsigned height, float ep);
 *     void Render(Scene& scene, unsigned width, unsigned height, Camera* camera,
 *                 unsigned iter, bool reset, float3* output);
 *     void EndRender();
 *
 * and for the host-side steps that produce what BeginRender() uploads
 * (Scene::Init, reference src/scene.h:50-83; BVH::build, src/bvh.cpp:18-36;
 * Camera constructor/Lookat, src/camera.h:31-46,123-128; LoadScene,
 * src/parsescene.h:26).  Entry points take plain pointers and sizes; records
 * are the reference's own struct layouts (gpt_types.h).  Every function returns
 * GPT_OK or a negative error code and never aborts; gpt_last_error() returns a
 * thread-local message for the last failure.
 *
 * The C++ signatures above are provided on top of this ABI by
 * gpu_pathtracer_amd/csrc/pathtracer.h for callers that link as C++.
 */
#ifndef GPT_H
#define GPT_H

#include "gpt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* exported even when the library is built with -fvisibility=hidden */
#pragma GCC visibility push(default)

#define GPT_OK                 0
#define GPT_ERR_INVALID_ARG   -1
#define GPT_ERR_UNSUPPORTED   -2   /* integrator other than "pt" / "ao" / "vpt", non-triangle primitive, wide tree beyond 4 GB */
#define GPT_ERR_HIP           -3   /* a HIP runtime call failed (message has file:line) */
#define GPT_ERR_NO_DEVICE     -4
#define GPT_ERR_IO            -5
#define GPT_ERR_PARSE         -6

typedef struct gpt_ctx gpt_ctx;      /* one renderer: scene in HBM + film state */
typedef struct gpt_scene gpt_scene;  /* host-side Scene built by the loader */

const char *gpt_last_error(void);
const char *gpt_version(void);

/* ---- render API --------------------------------------------------------- */

/* BeginRender (src/pathtracer.cu:2568-2695): copy the scene out of the caller's
 * arrays, lay it out for the GPU, allocate the W*H*3 accumulator
 * (kernel_acc_image) and last-sample (kernel_color) planes, both zeroed.
 * `device` is the HIP device ordinal.  The caller keeps ownership of `scene`:
 * every array it points at (density grids of heterogeneous media included) is
 * read before gpt_begin returns and never afterwards.  Node arrays and triangle
 * records share one device allocation addressed with 32-bit offsets: a scene whose
 * two together exceed 4 GB (about 50 M triangles) is refused, GPT_ERR_UNSUPPORTED. */
int gpt_begin(const gpt_scene_desc *scene, uint32_t width, uint32_t height, float epsilon,
              int device, gpt_ctx **out);

/* Multi-GPU tile ownership: this context renders only the 8x8 pixel tiles t
 * (row-major tile index) with t % n_ranks == rank; other pixels stay untouched
 * (zero), so a sum-reduce of the accumulators over ranks equals the 1-GPU image
 * bit for bit.  Default rank 0 of 1. */
int gpt_set_tile_owner(gpt_ctx *ctx, int rank, int n_ranks);

/* The reference reads scene.integrator.{type, maxDepth | maxDist} on the host at every Render() call
 * (src/pathtracer.cu:2711-2715); gpt_begin takes them from the scene description and this call changes them
 * afterwards.  GPT_IT_PT and GPT_IT_VPT use max_depth, GPT_IT_AO uses max_dist.  GPT_IT_VPT (Volpath, :1025-1242)
 * renders homogeneous and density-grid media and material-less surfaces (matIdx -1) between them; Path and Ao refuse
 * a scene with such surfaces (GPT_ERR_UNSUPPORTED), as every call refuses the other integrator types. */
int gpt_set_integrator(gpt_ctx *ctx, int32_t integrator_type, int32_t max_depth, float max_dist);

/* Traversal order of the BVH:
 *   GPT_TRAVERSAL_REFERENCE (0)           the reference's order on its binary tree; results are the reference's bit for bit.
 *   GPT_TRAVERSAL_WIDE4 (2)               a 4-wide tree collapsed from the reference's, walked one lane per ray
 *                                         (include/gpt_wide_bvh.h): a quarter of the node visits; GPT_ERR_UNSUPPORTED for an empty
 *                                         scene or a tree deeper than 85 wide levels (the reference's own 64-entry stack ends at
 *                                         binary depth 64).
 *   GPT_TRAVERSAL_AUTO (-1)               back to gpt_begin's choice (include/gpt_traversal.h): the wide tree for every scene that
 *                                         does not fit LDS and has one, the reference order otherwise.
 * The wide order changes only the ORDER of the reference's box and triangle tests: it is bit-identical to the oracle in the same
 * mode and within 1e-4 relative RMS of the reference order (measured: identical films, or single pixels where two hits tie within
 * rounding - relative RMS <= 2e-7).  It always traverses from global memory.  gpt_begin selects it for every scene that does not fit
 * LDS (where it is 9 - 74 % faster) and the reference order for scenes that do; this call overrides the choice either way.  The first
 * selection of the wide order uploads the wide tree with a copy of the triangle records behind it (128 B per wide node + 48 B per
 * triangle, one allocation below 4 GB - GPT_ERR_UNSUPPORTED beyond) and the per-wave stack spill space. */
int gpt_set_traversal_order(gpt_ctx *ctx, int32_t order);

/* Renderer options, by name; none of them changes a result.  Nothing in the library is steered by environment
 * variables: what differs from the defaults was set through this call and can be read back.
 *   "lds_scene"        1 (default): a scene of <= 12 KB is staged in LDS by every workgroup; 0: always global memory
 *   "vpt_walk_kernel"  0 (default): Volpath picks its kernel from the scene; 1: always the one-ray-at-a-time kernel
 *   "max_batch"        iterations per path-kernel launch; default 256 x the number of ranks sharing the frame (a rank's sample
 *                      planes cover its own tiles only), always bounded by 16 GiB and by the free device memory
 *   "chunk_iters"      iterations per work item, 0 (default) = cost model
 * gpt_get_option also answers (read-only) "lds_scene_active", "walk_kernel_active", "traversal_order", "owned_tiles", "last_batch",
 * "sample_plane_bytes", "wide_fallback" (1: gpt_begin wanted the 4-wide walk and had no room for its tree; gpt_last_error() keeps why),
 * and the time spans the library measures with HIP events on its own stream (microseconds; reading one synchronises on its end):
 * "last_trace_us" (the last gpt_debug_trace), "last_reduce_us" (the last gpt_reduce_film: the ncclReduce alone), "last_tonemap_us" (the
 * last gpt_tonemap / gpt_tonemap_from), "output_kernel_us" (the accumulation kernel behind every path-kernel launch, summed since the last
 * gpt_kernel_time_reset - the companion of gpt_kernel_time, which covers the path kernel only). */
int gpt_set_option(gpt_ctx *ctx, const char *name, int64_t value);
int gpt_get_option(gpt_ctx *ctx, const char *name, int64_t *value);

/* Render (src/pathtracer.cu:2705-2750), batched: for iter = iter_first ..
 * iter_first+iter_count-1 add one sample per pixel, seeded by (pixel, iter),
 * into the accumulator, exactly as iter_count successive reference calls would
 * (iter_count = 1 is the reference call).  `reset` clears the accumulator first
 * (reference: reset on the first call after a camera move).  `camera` is read
 * on every call like the reference's per-call cudaMemcpy.  If
 * `out_tonemapped_dev` is not NULL it must be a DEVICE pointer to W*H*3 floats
 * (the reference's mapped GL buffer) and receives tonemap(acc / last_iter).
 * Asynchronous: returns after enqueueing on the context's stream.  That stream is a
 * blocking stream, i.e. ordered with the legacy default stream exactly like the
 * reference's default-stream launches: a hipMemcpy / default-stream kernel issued after
 * gpt_render sees the finished film.  Work on a caller's own NON-blocking stream must be
 * ordered with gpt_synchronize() first. */
int gpt_render(gpt_ctx *ctx, const gpt_camera *camera, uint32_t iter_first, uint32_t iter_count,
               int reset, float *out_tonemapped_dev);

/* Output pass alone (src/pathtracer.cu:2516-2531 without the accumulate):
 * out = tonemap(acc / iter).  Used on the root rank after the framebuffer
 * reduce.  `out_dev` is a device pointer. */
int gpt_tonemap(gpt_ctx *ctx, uint32_t iter, int filmic, float *out_dev);

/* ... the same pass reading any W*H*3 accumulator in device memory (e.g. a reduced frame) */
int gpt_tonemap_from(gpt_ctx *ctx, const float *acc_dev, uint32_t iter, int filmic, float *out_dev);

int gpt_synchronize(gpt_ctx *ctx);

/* ---- multi-GPU: one process per GPU, pixel tiles across ranks, ONE framebuffer reduce -------------------------------
 * The path shards embarrassingly (every pixel-sample is independent, src/pathtracer.cu:888): every rank holds the whole
 * scene and renders the 8x8 tiles t with t % n_ranks == rank; the frame is assembled by one RCCL sum-reduce of the fp32
 * W*H*3 accumulator over xGMI, issued on the renderer's own stream right behind the render.  Disjoint supports: the sum
 * adds zeros, so the root's frame is bit-identical to a 1-GPU render.  A caller shaped like src/main.cpp:134-144 becomes
 *     gpt_comm_unique_id(id) on rank 0, hand the 128 bytes to every rank (MPI, a file, torch.distributed ...);
 *     gpt_begin(...); gpt_comm_init(ctx, rank, n, id);                     // also sets the tile ownership
 *     per frame: gpt_render(ctx, cam, 1, spp, 1, NULL); gpt_reduce_film(ctx, 0);
 *                rank 0: gpt_tonemap_from(ctx, gpt_reduced_device_ptr(ctx), spp, filmic, output);
 * The reduce RECEIVES into a separate buffer on the root (gpt_reduced_device_ptr): the accumulator of every rank keeps
 * exactly its own tiles, so progressive rendering (reset = 0) followed by another reduce stays correct. */
int gpt_comm_unique_id(void *id128);                                         /* 128 bytes (ncclUniqueId) */
int gpt_comm_init(gpt_ctx *ctx, int rank, int n_ranks, const void *id128);
int gpt_reduce_film(gpt_ctx *ctx, int root);                                 /* asynchronous, on the context's stream */
float *gpt_reduced_device_ptr(gpt_ctx *ctx);                                 /* the root's whole frame (sum of samples) */
int gpt_read_reduced(gpt_ctx *ctx, float *host_rgb);                         /* synchronises */
int gpt_comm_destroy(gpt_ctx *ctx);                                          /* also done by gpt_end */

/* Film state.  Device pointers stay valid until gpt_end(); W*H*3 floats each,
 * row 0 = bottom (reference convention, src/imageio.cpp:66).  Pixel (x, y) lives at
 * float index 3 * (x + y * stride) with stride = 32 * (W / 32): the reference's pixel
 * index uses blockDim.x * gridDim.x of its (W/32, H/4) grid as the row stride
 * (src/pathtracer.cu:881-883, 2707-2709).  For the widths the reference is used with
 * (multiples of 32) that is plain row-major; other widths skew, here as there, and only
 * the first stride columns and 4 * (H / 4) rows are rendered. */
float *gpt_accum_device_ptr(gpt_ctx *ctx);   /* kernel_acc_image: running SUM of samples */
float *gpt_color_device_ptr(gpt_ctx *ctx);   /* kernel_color: last finite sample */
int gpt_read_accum(gpt_ctx *ctx, float *host_rgb);        /* synchronises */
int gpt_read_color(gpt_ctx *ctx, float *host_rgb);
int gpt_write_state(gpt_ctx *ctx, const float *host_acc, const float *host_color); /* resume */
int gpt_copy_to_host(gpt_ctx *ctx, const float *dev, float *host, size_t n_floats);

/* Use caller-owned DEVICE buffers (W*H*3 floats each, zero them first) as the
 * accumulator / last-sample planes instead of the ones gpt_begin allocated —
 * like the reference, where the caller owns the device `output` buffer
 * (src/main.cpp:136-137).  Lets a framework tensor (e.g. the send buffer of
 * the RCCL framebuffer reduce) be the film itself.  NULL keeps the current one. */
int gpt_bind_film(gpt_ctx *ctx, float *acc_dev, float *color_dev);

/* EndRender (src/pathtracer.cu:2697-2703); frees everything gpt_begin allocated. */
int gpt_end(gpt_ctx *ctx);

/* ---- measurement ---------------------------------------------------------- */

/* HIP-event time of the path kernel launches since the last reset, measured on
 * the stream the kernel runs on.  Synchronises. */
int gpt_kernel_time(gpt_ctx *ctx, uint32_t *launches, double *total_ms);
int gpt_kernel_time_reset(gpt_ctx *ctx);

/* Work counters of the NEXT gpt_render call (it runs the counting build of the
 * kernel): out6 = node visits, primitive tests, bounce iterations, shadow rays,
 * closest-hit rays, samples — the terms of SURVEY.md §8(d) B_alg. */
int gpt_enable_counters(gpt_ctx *ctx, int enable);
int gpt_read_counters(gpt_ctx *ctx, uint64_t out6[6]);
/* The 6 counters above plus SIMD-utilisation probes of the counting build: [6] wave-level node-loop
 * trips, [7] wave-level triangle-loop trips, [8]/[9] wave/lane bounce trips, [10]/[11] wave/lane
 * hit-shading trips, [12]/[13] wave/lane direct-light trips.  lanes/(64*waves) = lane utilisation. */
int gpt_read_probe_counters(gpt_ctx *ctx, uint64_t out16[16]);

/* Device-side evaluation of the elementary float operations the kernel relies
 * on, for parity tests against the CPU oracle: fn 0 sin, 1 cos, 2 tan, 3 atan,
 * 4 acos, 5 pow(x,y), 6 x/y, 7 sqrt, 8 1/sqrt, 9 exp, 10 log (the last two: src/medium.h:15,41-43,73,
 * src/common.h:81-86, src/wrap.h:158-160).  Host pointers. */
int gpt_debug_math(int device, int fn, const float *x, const float *y, float *out, int n);
/* SampleBSDF / Fr (src/pathtracer.cu:491-695, 698-826, with GetTexel :324-359) on the device, through the routines the render kernels
 * shade with (csrc/pt_bsdf.h: surface_prepare, then surface_respond or surface_scatter).  Case i: geom11[11 i ..] = in.xyz (the unit
 * vector back along the arriving ray), normal.xyz, dpdu.xyz, uv.xy; in3[3 i ..] = mode 0: the direction `out` Fr is asked about,
 * mode 1: the three draws `u` of SampleBSDF.  out7[7 i ..] = out.xyz, fr.xyz, pdf.  material->textureIdx is -1, or 0 and then
 * `texture` is that texture.  Host pointers. */
int gpt_debug_bsdf(int device, const gpt_material *material, const gpt_texture *texture, const float *geom11, const float *in3, int n,
                   int mode, float *out7);
/* The traversal operators alone (Intersect / IntersectP, src/pathtracer.cu:214-296; BBox::Intersect, src/bbox.h:77-96;
 * Triangle::Intersect, src/mesh.h:45-67) on the device, through the render kernel's own ray pools and traversal loops, in the
 * context's current traversal order and memory path.  Ray i = rays8[8 i ..] = {origin.xyz, direction.xyz, tmax, any_hit != 0},
 * tmin = the context's epsilon.  prim_out[i] = hit primitive (BVH order) or -1, tb_out[3 i ..] = {t, b1, b2}.  Host pointers. */
int gpt_debug_trace(gpt_ctx *ctx, const float *rays8, int n, int32_t *prim_out, float *tb_out);
/* Test hook: the NEXT allocation of a wide tree (gpt_begin / gpt_set_traversal_order) reports "out of device memory" without asking
 * the device - exercises gpt_begin's fall-back to the reference's traversal order (option "wide_fallback" then reads 1). */
int gpt_debug_fail_next_wide_alloc(int enable);
/* first n uniform draws of the (pixel, iter) stream, evaluated on the device */
int gpt_debug_rng(int device, uint32_t pixel, uint32_t iter, uint32_t *seed_out, float *u_out, int n);
/* the scene-file reader on its own (tests: against the rapidjson the reference vendors): -1 when the reader refuses the
 * document; for an array of numbers their count, the values (as the reader's doubles) in out[0..cap); 0 for other documents */
int gpt_debug_json_numbers(const char *text, double *out, int cap);

/* ---- host-side scene preparation (CPU; no GPU needed) ---------------------- */

/* BVH::build (src/bvh.cpp:18-173): binned-SAH build + preorder flatten.
 * prims_out holds n records, nodes_out at least 2*n.  root_box6 = min.xyz,max.xyz */
int gpt_bvh_build(const gpt_primitive *prims_in, int32_t n, gpt_primitive *prims_out,
                  gpt_bvh_node *nodes_out, int32_t *n_nodes_out, float root_box6[6]);

/* The split BVH north_star names (src/sbvh.h is an empty class in the reference; Stich et al., HPG 2009): object splits like the
 * builder above + spatial splits that DUPLICATE the primitives straddling the plane, in the same tree layout, so every
 * traversal (reference order, 4-wide) and gpt_begin take it as they take the reference's tree.
 * alpha: a spatial split is considered when the object split's children overlap by more than alpha x the root's surface area
 * (the paper's 1e-5).  prims_out / orig_out hold prims_cap records (n .. 2n in practice; duplication stops when the capacity is
 * used), orig_out[i] = input index of prims_out[i]; nodes_out holds nodes_cap (2 * prims_cap suffices). */
int gpt_sbvh_build(const gpt_primitive *prims_in, int32_t n, float alpha, gpt_primitive *prims_out, int32_t prims_cap,
                   int32_t *n_prims_out, int32_t *orig_out, gpt_bvh_node *nodes_out, int32_t nodes_cap, int32_t *n_nodes_out,
                   float root_box6[6]);

/* Scene::Init light power CDF (src/scene.h:65-82); cdf_out holds n_lights+2 floats */
int gpt_light_distribution(const gpt_area *lights, int32_t n_lights, const gpt_infinite *infinite,
                           float *cdf_out, int32_t *n_out);

/* Infinite::Init (src/infinite.h:61-63) */
int gpt_infinite_init(gpt_infinite *infinite, const float root_box6[6]);

/* Camera::Lookat + constructor (src/camera.h:31-46,123-128; called as main.cpp:268-270) */
int gpt_camera_init(gpt_camera *camera, const float position[3], const float lookat[3], const float up[3],
                    float res_x, float res_y, float distance, float fov_degrees, float aperture_radius,
                    float focal_distance, int filmic, int environment);

/* LoadScene + InitScene (src/parsescene.cpp:45-590, src/main.cpp:261-278):
 * parse a scene JSON, read its OBJ meshes, build the BVH and the light CDF. */
int gpt_scene_load(const char *json_path, gpt_scene **out);
/* ... with BVH::LoadOrBuildBVH's cache file <scene dir>/bvh.cache (src/bvh.cpp:189-218) when use_bvh_cache != 0.  The
 * reference always uses it and silently reuses a stale one; here it is opt-in and carries a content hash. */
int gpt_scene_load_cached(const char *json_path, int use_bvh_cache, gpt_scene **out);
/* ... with flags: GPT_LOAD_BVH_CACHE as above; GPT_LOAD_SBVH builds the tree with gpt_sbvh_build (alpha 1e-5, at most 2x the
 * primitives) instead of the reference's builder: same scene, same films up to exactly-equal-distance ties, fewer node visits;
 * GPT_LOAD_REFERENCE_BVH builds with the reference's builder (src/bvh.cpp:38-173) whatever tree it makes.  With neither of the two
 * (gpt_scene_load, gpt_scene_load_cached without a cache) the tree is the reference builder's UNLESS it has a leaf of more than 16
 * primitives - bvh.cpp:43 makes one leaf of any set whose box is thinner than 1e-4, however large - in which case it is the split
 * tree (config-3 stand-in: 2.5 -> 3.0 Gsamples/s); every scene the reference ships keeps the reference's tree. */
#define GPT_LOAD_BVH_CACHE 1
#define GPT_LOAD_SBVH 2
#define GPT_LOAD_REFERENCE_BVH 4
int gpt_scene_load_ex(const char *json_path, int flags, gpt_scene **out);
int gpt_scene_get_desc(const gpt_scene *scene, gpt_scene_desc *desc_out);
int gpt_scene_get_config(const gpt_scene *scene, int32_t *width, int32_t *height, float *epsilon,
                         gpt_camera *camera_out);
int gpt_scene_set_integrator(gpt_scene *scene, int32_t integrator_type, int32_t max_depth);
int gpt_scene_free(gpt_scene *scene);

/* ImageIO::SavePng (src/imageio.cpp:61-78): flip Y, clamp, truncate to 8 bit.
 * `rgb` is W*H*3 host floats, row 0 = bottom. */
int gpt_save_png(const char *path, int32_t width, int32_t height, const float *rgb);
/* ImageIO::SaveExr (src/imageio.cpp:104-161): linear radiance as OpenEXR, channels B,G,R stored as HALF */
int gpt_save_exr(const char *path, int32_t width, int32_t height, const float *rgb);
/* linear radiance as PFM (little-endian float32, bottom-up) */
int gpt_save_pfm(const char *path, int32_t width, int32_t height, const float *rgb);

/* The file decoders by themselves (the loader calls them for "diffuse": "<file>" and "infinite": "<file>").  A null
 * output buffer asks for the size only; `capacity` counts elements of the buffer's type.
 * gpt_decode_image8: what stb_image hands ImageIO::LoadTexture (src/imageio.cpp:13-14: flip on load, 0 = the file's own
 *                    channel count): PNG, JPEG, BMP, binary PNM or TGA -> `components` interleaved bytes per pixel, row 0 = bottom.
 * gpt_load_texture:  ImageIO::LoadTexture + Texture::Texture (src/imageio.cpp:11-59, src/texture.h:15-27): the texels as
 *                    the kernel samples them (1/255, powf(x, 2.2f) on r g b, truncated back to 8 bit).
 * gpt_load_exr:      ImageIO::LoadExr (src/imageio.cpp:80-102): float R,G,B per pixel, row 0 = top. */
int gpt_decode_image8(const char *path, int32_t *width, int32_t *height, int32_t *components, unsigned char *pixels, int64_t capacity);
int gpt_load_texture(const char *path, int32_t *width, int32_t *height, gpt_uchar4 *texels, int64_t capacity);
int gpt_load_exr(const char *path, int32_t *width, int32_t *height, float *rgb, int64_t capacity);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif

#endif /* GPT_H */
