/*
 * tbrm.h — C-ABI of the MI355X-native raymarch + illumination hot path.
 *
 * Drop-in boundary for the one hot path of tommybazar/TBRaymarcherPlugin:
 * the operator library URaymarchUtils (reference Source/Raymarcher/Public/Util/RaymarchUtils.h:33-93),
 * its render-thread drivers (Public/Rendering/LightingShaders.h:17-22) and the material entry point
 * PerformWindowedLitRaymarch (Shaders/Private/WindowedRaymarchMaterials.usf:36-96) preceded by
 * PerformRaymarchCubeSetup (Shaders/Private/RaymarchMaterialCommon.usf:23-69).
 *
 * Everything here is plain C: PODs, pointers, sizes. No HIP or torch types appear in signatures; a
 * "stream" is passed as an opaque void* (a hipStream_t) only in the *_device entry points.
 * Every struct below names the reference type it stands for. All calls on one tbrm_resources handle
 * are enqueued on that handle's HIP stream in FIFO order (the reference's game thread -> render thread
 * command queue, RaymarchUtils.cpp:63-66); tbrm_flush is the join (FlushRenderingCommands()).
 */
#ifndef TBRM_H
#define TBRM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TBRM_API __attribute__((visibility("default")))

/* ---- status codes (the reference has no return codes; RaymarchUtils.cpp:39-49 reports through
 *      `bool& LightAdded`, which every light entry point below keeps as an out-flag) ---- */
enum {
    TBRM_OK = 0,
    TBRM_ERR_INVALID_ARG = 1,   /* null handle / null pointer / bad enum */
    TBRM_ERR_NOT_INITIALIZED = 2, /* a resource the reference null-checks is missing -> light_added=false */
    TBRM_ERR_NO_DEVICE = 3,     /* no HIP device / HIP runtime error (message via tbrm_last_error) */
    TBRM_ERR_OUT_OF_MEMORY = 4,
    TBRM_ERR_UNSUPPORTED = 5,
    TBRM_ERR_AXES_DIFFER = 6    /* tbrm_slab_light_begin: a Change across major axes; run remove + add (nothing was enqueued) */
};

/* ---- pixel formats of UVolumeTexture-shaped buffers (VolumeInfo.cpp:96-119) ---- */
enum {
    TBRM_FMT_G8 = 0,        /* PF_G8        UNORM8  */
    TBRM_FMT_G16 = 1,       /* PF_G16       UNORM16 */
    TBRM_FMT_R32_FLOAT = 2  /* PF_R32_FLOAT float   */
};

/* Address mode of the *material's* data-volume sampler (the texture asset's sampler, never set by the
 * reference -> UVolumeTexture default TA_Wrap). The light volume is always sampled with
 * wrap (Material.Wrap_WorldGroupSettings, WindowedRaymarchMaterials.usf:30) and the propagation shaders
 * always sample the data volume with border addressing (LightingShaders.h:82-89). */
enum { TBRM_ADDRESS_WRAP = 0, TBRM_ADDRESS_CLAMP = 1 };

/* How sampler border colours are formed (engine behaviour outside the reference):
 * ENGINE_8BIT : read-buffer border = sRGB-8-bit round trip of clamp01(I*w)  (LightingShaderUtils.cpp:197-203),
 *               data-volume border = round(clamp01(C - W/2)*255)/255         (LightingShaders.h:82-85);
 * EXACT_FLOAT : both borders keep their float value (diagnostics). */
enum { TBRM_BORDER_ENGINE_8BIT = 0, TBRM_BORDER_EXACT_FLOAT = 1 };

/* ERaymarchMaterial (RaymarchVolume.h:24-29). Only Lit is on the hot path this round. */
enum { TBRM_MATERIAL_LIT = 0, TBRM_MATERIAL_INTENSITY = 1, TBRM_MATERIAL_OCTREE = 2 };

/* FVector / FQuat / FTransform — double precision in UE5. */
typedef struct tbrm_vec3d { double x, y, z; } tbrm_vec3d;
typedef struct tbrm_quatd { double x, y, z, w; } tbrm_quatd;
typedef struct tbrm_transform {
    tbrm_quatd rotation;    /* unit quaternion */
    tbrm_vec3d translation;
    tbrm_vec3d scale3d;
} tbrm_transform;

/* FDirLightParameters (RaymarchTypes.h:20-41). */
typedef struct tbrm_dir_light_params {
    tbrm_vec3d light_direction; /* world space, need not be normalised */
    float light_intensity;
    int32_t _pad;
} tbrm_dir_light_params;

/* FClippingPlaneParameters (RaymarchTypes.h:45-71): Center + the direction that is NOT clipped away. */
typedef struct tbrm_clipping_plane_params {
    tbrm_vec3d center;
    tbrm_vec3d direction;
} tbrm_clipping_plane_params;

/* FRaymarchWorldParameters (RaymarchTypes.h:136-153). */
typedef struct tbrm_world_params {
    tbrm_transform volume_transform;
    tbrm_clipping_plane_params clipping_plane;
} tbrm_world_params;

/* FWindowingParameters (VolumeTextureToolkit/Public/VolumeAsset/VolumeInfo.h:31-53). */
typedef struct tbrm_windowing_params {
    float center;        /* default 0.5 */
    float width;         /* default 1.0 */
    int32_t low_cutoff;  /* default true */
    int32_t high_cutoff; /* default true */
} tbrm_windowing_params;

/* What InitializeRaymarchResources (RaymarchVolume.cpp:821-920) decides from the data volume + flags. */
typedef struct tbrm_resources_desc {
    int32_t dim_x, dim_y, dim_z;              /* data volume size in voxels */
    int32_t data_format;                      /* TBRM_FMT_* */
    int32_t light_volume_32bit;               /* bLightVolume32Bit: PF_R32_FLOAT instead of PF_G8 (:857-861) */
    int32_t light_volume_half_resolution;     /* LightVolumeHalfResolution: ceil(dim/2) (:850-855) */
    int32_t device;                           /* HIP device ordinal */
    int32_t data_address_mode;                /* TBRM_ADDRESS_* for the raymarch material's data sampler */
    int32_t border_mode;                      /* TBRM_BORDER_* */
    int32_t _reserved;
} tbrm_resources_desc;

/* The view uniforms PerformRaymarchCubeSetup reads from the engine (ResolvedView.WorldCameraOrigin,
 * ViewToTranslatedWorld, CameraVector; RaymarchMaterialCommon.usf:26-48) restated as an explicit pinhole
 * camera. Pixel (px,py) looks along normalize(forward + sx*right + sy*up) with
 * sx = (2*(px+0.5)/width - 1)*tan_half_fov_x, sy = (1 - 2*(py+0.5)/height)*tan_half_fov_y. */
typedef struct tbrm_camera {
    tbrm_vec3d position;  /* world */
    tbrm_vec3d forward;   /* world, unit */
    tbrm_vec3d right;     /* world, unit */
    tbrm_vec3d up;        /* world, unit */
    double tan_half_fov_x;
    double tan_half_fov_y;
    int32_t width;        /* full framebuffer size in pixels */
    int32_t height;
} tbrm_camera;

/* The part of the framebuffer one call renders (image-tile sharding). Output row j of the
 * w x h result is framebuffer row  y0 + (j/8)*8*row_group_step + (j%8)  (row_group_step = 1: plain
 * rectangle; = N: every N-th group of 8 rows, the load-balanced interleave used across N GPUs). */
typedef struct tbrm_tile {
    int32_t x0, y0;
    int32_t w, h;
    int32_t row_group_step;
    int32_t _pad;
} tbrm_tile;

/* Per-call raymarch parameters = the material parameters ARaymarchVolume sets by name
 * (RaymarchMaterialParameters.h:13-23, RaymarchVolume.cpp:671-728). */
typedef struct tbrm_raymarch_params {
    float steps;              /* "Steps" (RaymarchingSteps, default 150; RaymarchVolume.h:188-189) */
    int32_t jitter_frame;     /* View.StateFrameIndexMod8 (0..7); < 0 disables JitterEntryPos */
    int32_t enable_skipping;  /* empty-space skipping on/off (results are identical either way) */
    int32_t _pad;
} tbrm_raymarch_params;

/* Host-side parameter block of ONE propagation axis pass — everything LightingShaders.cpp:91-131 computes
 * before its slice loop, already narrowed to what the shader binds (LightingShaders.h:100-160). Exposed so
 * the host math can be tested without a GPU. */
typedef struct tbrm_light_pass {
    int32_t face;              /* FCubeFace 0..5 = +X,-X,+Y,-Y,+Z,-Z (LightingShaderUtils.h:21-29) */
    int32_t axis;              /* face / 2 */
    float weight;              /* FaceWeight[i].second after the 0.99 snap / 1-w0 rule */
    float light_alpha;         /* GetLightAlpha = Intensity * weight: buffer clear value */
    float border_light;        /* read-buffer sampler border colour (see TBRM_BORDER_*) */
    float prev_pixel_offset[2];/* PrevPixelOffset (GetUVOffset) */
    float uvw_offset[3];       /* UVWOffset after renormalisation to 1/min(TD) */
    float step_size;           /* StepSize */
    int32_t td[3];             /* TransposedDimensions of the light volume */
    int32_t start, stop, dir;  /* GetLoopStartStopIndexes */
} tbrm_light_pass;

typedef struct tbrm_resources tbrm_resources; /* opaque: FBasicRaymarchRenderingResources (RaymarchTypes.h:87-129) */

/* ------------------------------------------------------------------------------------------------ */
/* library                                                                                           */
/* Bumped whenever an entry point changes its signature or meaning (3: round 3 — tbrm_change_dir_light has carried its
 * trailing gpu_sync argument since 2; 3 adds tbrm_abi_version itself and the error report of tbrm_flush; 4 adds
 * tbrm_path_counters, and every entry point that waits for the handle's stream now reports a failed sweep like tbrm_flush).
 * A host built against another number must not call into the library. */
#define TBRM_ABI_VERSION 5
TBRM_API int tbrm_abi_version(void);
TBRM_API const char* tbrm_version(void);
TBRM_API const char* tbrm_last_error(void);       /* thread-local message of the last failing call */
TBRM_API int tbrm_device_count(int* out_count);   /* TBRM_ERR_NO_DEVICE when the HIP runtime has none */

/* Process-wide tunables: A/B switches for measurements and parity tests (no counterpart in the reference; nothing a
 * host needs to call). Each starts from the environment variable TBRM_<NAME IN CAPITALS>, read once when the library is
 * loaded; no operator reads the environment. Names (default): force_slice_kernel (0), chunk_steps (0 = by fit),
 * occ_slices (0 = 128; the chunked chain's occlusion spans), sparse_occ (1), occ_list (1), light_cache_mb (-1 = an eighth of
 * the device's memory at most, 0 = off, else MiB), light_batching (1; 0 never, 2 always), share_grid (1), ray_wave_skip (the lit march takes the empty trips a whole wave shares in one go: 1 on, 0 off,
 * -1 = for data volumes of at least 384 voxels a side), ray_lanes (0 = by
 * load; 4 / 8), chain_fast_loop (1), chain_rect_planes (1), occ_overlap (2 = workgroups per CU of an occlusion launch that
 * runs beside a chunked chain; 0 = one after the other), light_sweep (1 = axis passes take the pipelined sweep kernel where
 * it applies; 2 = except the passes of a Change whose two lights pull opposite ways, which otherwise take two sweeps; 0 = the
 * chunked chain everywhere), sweep_prefetch (0 = 2 slices), stream_priority (of a handle's own stream, read by tbrm_resources_create: 0 = default, 1 = highest, -1 = lowest),
 * sweep_debug (diagnostics: bit 0 tiles do not
 * wait for each other, bits 3 / 4 skip buffer hazards — WRONG light volumes —; bit 1 prints per-tile time stamps at tbrm_flush,
 * bit 2 the host's time per operator phase, bit 5 leaves out the events behind tbrm_last_gpu_time_ms), occ_dual (1 = the two
 * axis passes of a light share one occlusion launch — their sampling positions are the same, LightingShaders.cpp:114-124 —,
 * 0 = one launch per pass), sweep_chain (4 = up to four consecutive sweep passes of an operator share ONE launch —
 * k_light_sweep_chain: the next pass's tiles take their tickets behind this pass's and start as its tiles retire, ordered brick
 * layer by brick layer through progress words; 1 = one launch per pass), sweep_timeout_ms (0 = a sweep tile waits 2 s of wall time for a neighbour's hand-off word before it
 * gives up and the handle reports the light volume undefined; < 0 = not at all: a test hook), sweep_epoch_preset (0; > 0: a
 * handle's first sweep launch continues from this 16-bit launch tag: a test hook for the tags' wrap-around), fast_window_div (1 = where the host can vouch for the window — finite, moderate centre / width, UNORM data — the kernels
 * compute the transfer-function position with three fmas instead of the IEEE division: the same bits, tbrm_selftest_window_division; 0 = always
 * divide), slab_sweep (0; 1 = a slab's share of a pass along z
 * runs as one sweep instead of the chunked chain: measured, a tie or a loss), gpu_timing (1 = operators record the HIP events behind
 * tbrm_last_gpu_time_ms; 0 = they do not, and tbrm_last_gpu_time_ms fails until an operator has run with it on again), ray_tables (1 = the
 * lit march reads the data taps' offsets out of LDS tables where a step is at most one texel; 0 = computes them per sample), ray_xcd_rows (1 = the lit march deals its 8 x 8
 * pixel blocks to the GPU's eight XCDs row by row — horizontal neighbours, which march through the same bricks, share an L2 —; n = in
 * bands of n rows; 0 = in launch order, i.e. round-robin block by block: 10 % slower at 512^3). Unknown name: TBRM_ERR_INVALID_ARG. */
TBRM_API int tbrm_set_tunable(const char* name, int32_t value);
TBRM_API int tbrm_get_tunable(const char* name, int32_t* value);

/* ------------------------------------------------------------------------------------------------ */
/* resources: ARaymarchVolume::InitializeRaymarchResources / FreeRaymarchResources                   */
/* (RaymarchVolume.cpp:821-949). The handle owns every device allocation (data volume copy, TF LUT,   */
/* light volume, 4 read/write buffers per axis, skipping metadata) and one HIP stream.                */
TBRM_API int tbrm_resources_create(const tbrm_resources_desc* desc, tbrm_resources** out);
TBRM_API int tbrm_resources_destroy(tbrm_resources* res);
/* Everything the whole-volume light operators of this handle will need, allocated NOW — the reference creates its buffers once, in
 * ARaymarchVolume::InitializeRaymarchResources (RaymarchVolume.cpp:821-920), never inside AddDirLightToSingleVolume: the second
 * stream and its events, the factor scratch buffers, hand-off records (previous-slice taps up to two texels from the pixel), block
 * lists and ordering events for n_lights lights, and one arena for the factor cache's entries (4 entries per light, within the
 * light_cache_mb budget, at most half of what the device has free beyond 8 GiB). Afterwards an operator allocates nothing, asks the device nothing and
 * never waits for a stream (tbrm_path_counters out[12], out[13] stand still); a cache entry that does not fit the arena evicts
 * entries nothing in flight reads, or the pass goes uncached. flags bit 0: the chunked-chain fallback's stores too (passes the
 * sweep declines). Optional: a handle nobody reserved takes the arena, the events and the block lists for n_lights = 4 inside its
 * first light operator and allocates scratch stores and hand-off records when the first pass that needs them runs (a few
 * allocations per scene: a host that holds many handles of a large volume does not pay 8 KiB per 16 x 16 x 8 block of every one
 * of them up front). May be called again with more lights (drains the streams, drops the cache). */
TBRM_API int tbrm_resources_reserve(tbrm_resources* res, int32_t n_lights, uint32_t flags);
TBRM_API int tbrm_resources_light_volume_dims(const tbrm_resources* res, int32_t out_dims[3]);
TBRM_API int tbrm_resources_is_initialized(const tbrm_resources* res); /* bIsInitialized: volume + TF present */

/* UVolumeTexture-shaped input: dense x-fastest array in desc.data_format (CreateVolumeTextureMip memcpy,
 * TextureUtilities.cpp:43-78). Host pointer variant copies over PCIe; device variant reads HBM.        */
TBRM_API int tbrm_upload_volume(tbrm_resources* res, const void* host_voxels, size_t n_bytes);
/* device variant: the voxels must be complete in HBM before the call (the library reads them on its own stream: sync
 * the producing stream first); the buffer may be released when the call returns. */
TBRM_API int tbrm_upload_volume_device(tbrm_resources* res, const void* device_voxels, size_t n_bytes);

/* Transfer function. tbrm_set_tf_lut takes the 256 x RGBA float samples ColorCurveToTexture would take from
 * the curve and stores them as FFloat16 (RaymarchUtils.cpp:143-174). tbrm_color_curve_to_lut evaluates
 * piecewise-linear colour-curve keys (UCurveLinearColor with RCIM_Linear keys, constant extrapolation)
 * at i/255. tbrm_make_default_tf_lut = MakeDefaultTFTexture (RaymarchUtils.cpp:113-141).                */
TBRM_API int tbrm_set_tf_lut(tbrm_resources* res, const float* rgba_256x4);
TBRM_API int tbrm_color_curve_to_lut(const float* key_times[4], const float* key_values[4],
                                     const int32_t n_keys[4], float* out_rgba_256x4);
TBRM_API int tbrm_make_default_tf_lut(float* out_rgba_256x4);
/* What tbrm_set_tf_lut stores: every sample rounded to FFloat16 and widened back (no GPU needed). */
TBRM_API int tbrm_host_bake_tf_lut(const float* rgba_256x4, float* out_rgba_256x4);

/* FBasicRaymarchRenderingResources::WindowingParameters. */
TBRM_API int tbrm_set_windowing(tbrm_resources* res, const tbrm_windowing_params* windowing);

/* ------------------------------------------------------------------------------------------------ */
/* illumination operators (URaymarchUtils, RaymarchUtils.h:33-49)                                     */

/* AddDirLightToSingleVolume(Resources, LightParameters, Added, WorldParameters, LightAdded, bGPUSync).
 * gpu_sync is accepted and ignored (the reference's bGPUSync branch is a no-op, RaymarchUtils.cpp:51-59;
 * this library always runs the real propagation). Zero light direction: TBRM_OK, nothing enqueued,
 * *light_added = 1 (LightingShaders.cpp:41-46). */
TBRM_API int tbrm_add_dir_light(tbrm_resources* res, const tbrm_dir_light_params* light, int added,
                                const tbrm_world_params* world, int* light_added, int gpu_sync);

/* Several AddDirLightToSingleVolume calls as one (what ARaymarchVolume::ResetAllLights issues after its clear,
 * RaymarchVolume.cpp:418-451). Axis passes of different lights that leave the same cube face share one slice loop
 * (the multi-light optimisation of the Sunden/Ropinski scheme that the reference lists as not done, Readme.md:186-187),
 * the rest run as in tbrm_add_dir_light. The per-voxel updates happen in a different order than light by light — the
 * same sum, but an UNORM8 light volume can differ by one code where an fp32 add lands on a rounding tie — so the order
 * is reported: `schedule` (nullable, room for 8 * n_lights ints) receives 4 ints per entry {light a, pass a, light b,
 * pass b} (b = -1 -1 for an unpaired pass), entry e's a before its b, entries in order; *n_entries their number. */
TBRM_API int tbrm_add_dir_lights(tbrm_resources* res, const tbrm_dir_light_params* lights, int32_t n_lights, int added,
                                 const tbrm_world_params* world, int32_t* schedule, int32_t* n_entries);

/* ChangeDirLightInSingleVolume(Resources, Old, New, WorldParameters, LightAdded, bGPUSync) (RaymarchUtils.h:39-41).
 * Falls back to remove + add when the two major axes differ (LightingShaders.cpp:192-198). gpu_sync is accepted and
 * ignored, as in tbrm_add_dir_light (the reference's ChangeDirLight body never reads it, RaymarchUtils.cpp:70-92). */
TBRM_API int tbrm_change_dir_light(tbrm_resources* res, const tbrm_dir_light_params* old_light,
                                   const tbrm_dir_light_params* new_light, const tbrm_world_params* world,
                                   int* light_added, int gpu_sync);

/* ---- slabs: one light operation spread over the GPUs of a node (BASELINE config 4) --------------
 * The reference has no multi-GPU path; this is the same AddDirLight / ChangeDirLight arithmetic, partitioned. Every
 * handle holds the whole data volume (it is read-only and 288 GB hold any volume the plugin loads); the LIGHT volume's
 * z range is dealt out in slabs and each handle computes, and owns, only its slab [z_begin, z_end). An axis pass is
 * cut into chunks of 16/8/4/2 slices; between chunks the host exchanges the propagated-light planes:
 *   - pass along x or y ("lateral", z is the row axis of the slice plane): every handle runs every chunk on its rows;
 *     after each chunk it needs halo_rows rows of its two z neighbours' planes (what a chunk's bilinear taps can reach);
 *   - pass along z: the slabs are a pipeline; a handle imports the planes of the handle before it in propagation order,
 *     runs its chunks and hands its final planes on.
 * Per voxel the arithmetic and its inputs are those of the unpartitioned operator, so the partitioned light volume is
 * bit-identical to it. The exchange itself (RCCL send/recv, or a device copy between handles of one process) is the
 * host's: tbraymarcherplugin_amd/slabs.py is the driver; INTEGRATION.md shows the call sequence.
 * Slab bounds and the light volume's depth must be multiples of 32. */
typedef struct tbrm_slab {
    int32_t z_begin, z_end;    /* owned light-volume slices */
} tbrm_slab;

typedef struct tbrm_slab_pass {
    int32_t axis, dir;         /* propagation axis 0/1/2 and direction +-1 (tbrm_light_pass) */
    int32_t lateral;           /* 1: pass along x or y; 0: pass along z (pipeline over the slabs) */
    int32_t streams;           /* 1: Add (the light), 2: Change (0 = the added light, 1 = the removed light) */
    int32_t plane_w, plane_h;  /* propagated-light plane: plane_h rows of plane_w floats (lateral: row = z) */
    int32_t chunk_slices;
    int32_t chunks_of_pass;    /* chunks of the whole pass */
    int32_t first_chunk;       /* this handle runs chunks [first_chunk, first_chunk + n_chunks) of them */
    int32_t n_chunks;
    int32_t halo_rows;         /* lateral: rows each z neighbour has to supply after every chunk (else 0) */
    int32_t plane_elem_bytes;  /* 4 (float planes of the chunk kernels), or the light volume's element size when a steep pass
                                  runs one slice per chunk on the reference's read / write buffers (chunk_slices == 1) */
} tbrm_slab_pass;

/* Takes the operation apart: removed == NULL: AddDirLight(light, added); else ChangeDirLight(removed -> light), which
 * returns TBRM_ERR_AXES_DIFFER when the major axes differ (run remove + add, as LightingShaders.cpp:192-198 does).
 * *n_passes = axis passes to run (0..2), in order. Nothing is enqueued; every pass is checked first, so an operation
 * either runs completely or not at all. A pass whose taps lie more than 16 texels from the pixel (12 for a Change) runs
 * one slice per chunk (tbrm_slab_pass: chunk_slices == 1, halo_rows == the taps' reach); TBRM_ERR_UNSUPPORTED only when
 * that reach exceeds the slab's own depth. */
TBRM_API int tbrm_slab_light_begin(tbrm_resources* res, const tbrm_dir_light_params* removed,
                                   const tbrm_dir_light_params* light, int added, const tbrm_world_params* world,
                                   const tbrm_slab* slab, int32_t* n_passes);
/* Plans pass `pass` (enqueues its per-pass set-up kernels) and describes it. */
TBRM_API int tbrm_slab_pass_begin(tbrm_resources* res, int32_t pass, tbrm_slab_pass* out);
/* Enqueues this handle's chunk `chunk` (0 .. n_chunks-1) on the handle's stream. */
TBRM_API int tbrm_slab_pass_chunk(tbrm_resources* res, int32_t chunk);
/* Device address of the plane of `stream` holding the state BEFORE chunk `boundary` (boundary == n_chunks: after the
 * last chunk): plane_w * plane_h elements of plane_elem_bytes, row-major, valid in this handle's rows (lateral) or
 * everywhere (along z). */
TBRM_API int tbrm_slab_pass_plane(tbrm_resources* res, int32_t boundary, int32_t stream, void** device_plane);

/* ---- slab-resident handles: a GPU that holds only its part of the two volumes -------------------------------------
 * tbrm_resources_create_slab(desc, owned, &res): desc describes the WHOLE volume; the handle owns light-volume slices
 * [owned.z_begin, owned.z_end) and allocates only the brick layers (8 slices) it can touch: of the light volume the owned
 * layers, one layer either side, and a copy of the layer that wrap addressing reaches across the volume's ends; of the
 * data volume the layers the occlusion of its rows and the raymarch of its samples read (tbrm_slab_resident_slices
 * reports both ranges: {first slice, end slice, first slice of the wrap copy or -1}). Such a handle runs tbrm_slab_*
 * (its own slab only) and tbrm_raymarch_lit_slab_device; the whole-volume operators return TBRM_ERR_UNSUPPORTED. After
 * light operations and before a frame, neighbouring handles exchange their boundary light-volume layers
 * (tbrm_slab_light_halo: the layer to send to, and the layer to receive from, the neighbour on that side; the
 * neighbours form a ring, because the light volume is sampled with wrap addressing). */
TBRM_API int tbrm_resources_create_slab(const tbrm_resources_desc* desc, const tbrm_slab* owned, tbrm_resources** out);
TBRM_API int tbrm_slab_resident_slices(const tbrm_resources* res, int32_t data[3], int32_t light[3]);
/* Dense x-fastest slices [z_begin, z_begin + z_count) of the data volume (whole brick layers; any resident layer, the
 * wrap copy's source included). */
TBRM_API int tbrm_upload_volume_slices(tbrm_resources* res, int32_t z_begin, int32_t z_count, const void* host_voxels,
                                       size_t n_bytes);
TBRM_API int tbrm_download_light_slices(tbrm_resources* res, int32_t z_begin, int32_t z_count, void* host_out, size_t n_bytes);
/* side 0: the neighbour towards z = 0, side 1: the other. *recv_layer is NULL when the handle owns the whole depth. */
TBRM_API int tbrm_slab_light_halo(tbrm_resources* res, int32_t side, void** send_layer, void** recv_layer, size_t* layer_bytes);

/* ClearResourceLightVolumes(Resources, ClearValue) (RaymarchUtils.cpp:104-111). */
TBRM_API int tbrm_clear_light_volume(tbrm_resources* res, float clear_value);

/* ------------------------------------------------------------------------------------------------ */
/* raymarch: PerformRaymarchCubeSetup + PerformWindowedLitRaymarch for every pixel of `tile`.         */
/* Output: premultiplied RGBA float, tile.w x tile.h x 4, row-major (what BLEND_AlphaComposite receives). */
/* scene_depth (optional, may be NULL): CalcSceneDepth per full-framebuffer pixel, world units.        */
TBRM_API int tbrm_raymarch_lit(tbrm_resources* res, const tbrm_camera* camera, const tbrm_tile* tile,
                               const tbrm_raymarch_params* params, const tbrm_world_params* world,
                               float* host_out_rgba);
TBRM_API int tbrm_raymarch_lit_device(tbrm_resources* res, const tbrm_camera* camera, const tbrm_tile* tile,
                                      const tbrm_raymarch_params* params, const tbrm_world_params* world,
                                      const float* device_scene_depth, float* device_out_rgba);

/* One stage of a frame marched slab by slab over several GPUs (the raymarch counterpart of tbrm_slab_*): the handle
 * accumulates, into the per-pixel LightEnergy state of the tile (device_state_rgba, in place, zeros before the first
 * stage), exactly those samples of every ray whose position lies in its light-volume slices [z_begin, z_end). Along a
 * ray the slabs come in order, so the stages run in that order: direction > 0 takes the rays that travel towards +z in
 * volume space (run the handles in ascending slab order), direction < 0 the others (descending order); the state of a
 * tile passes from handle to handle between stages (16 bytes per pixel). Every sample, its position and the 0.95 early
 * exit are those of the unpartitioned march: after the last stage of both sweeps the state IS tbrm_raymarch_lit_device's
 * frame, bit for bit. direction == 0: all rays (a single handle owning everything reproduces the plain march). */
TBRM_API int tbrm_raymarch_lit_slab_device(tbrm_resources* res, const tbrm_camera* camera, const tbrm_tile* tile,
                                           const tbrm_raymarch_params* params, const tbrm_world_params* world,
                                           const float* device_scene_depth, float* device_state_rgba,
                                           const tbrm_slab* slab, int direction);

/* The Intensity render mode (ERaymarchMaterial::Intensity, SwitchRenderer RaymarchVolume.cpp:786-800):
 * PerformRaymarchCubeSetup + PerformWindowedIntensityRaymarch (WindowedRaymarchMaterials.usf:187-242) — per pixel the
 * windowed intensity (clamped TF position, as RGB with alpha 1) of the first sample the clipping plane does not remove,
 * read with the material's clamp sampler; (0,0,0,0) when every sample is clipped. No transfer function, light volume or
 * early exit is involved; params->enable_skipping is ignored. Output layout and scene_depth as for tbrm_raymarch_lit. */
TBRM_API int tbrm_raymarch_intensity(tbrm_resources* res, const tbrm_camera* camera, const tbrm_tile* tile,
                                     const tbrm_raymarch_params* params, const tbrm_world_params* world,
                                     float* host_out_rgba);
TBRM_API int tbrm_raymarch_intensity_device(tbrm_resources* res, const tbrm_camera* camera, const tbrm_tile* tile,
                                            const tbrm_raymarch_params* params, const tbrm_world_params* world,
                                            const float* device_scene_depth, float* device_out_rgba);

/* The Octree render mode (ERaymarchMaterial::Octree; experimental in the reference: its march does no skipping yet).
 * tbrm_generate_octree = URaymarchUtils::GenerateOctree (RaymarchUtils.cpp:94-102) -> GenerateOctreeForVolume_RenderThread
 * (OctreeShaders.cpp:28-54) + GenerateOctreeShader.usf:28-107: a 4-mip UNORM16 max pyramid whose base level has the
 * volume's dimensions rounded up to powers of two (RaymarchVolume.cpp:873-877); mip 0 copies the volume (0 outside it).
 * tbrm_octree_mip_dims / tbrm_download_octree_mip expose a level (dense, x fastest, uint16 codes).
 * tbrm_raymarch_octree[_device] = the M_Octree_Raymarch material: PerformRaymarchCubeSetup + PerformWindowedRaymarchOctree
 * (WindowedRaymarchMaterials.usf:99-183): the unlit march over level `octree_mip` (OctreeVolumeMip), point sampled. */
TBRM_API int tbrm_generate_octree(tbrm_resources* res);
TBRM_API int tbrm_octree_mip_dims(const tbrm_resources* res, int mip, int32_t out_dims[3]);
TBRM_API int tbrm_download_octree_mip(tbrm_resources* res, int mip, uint16_t* host_out, size_t bytes);
TBRM_API int tbrm_raymarch_octree(tbrm_resources* res, const tbrm_camera* camera, const tbrm_tile* tile,
                                  const tbrm_raymarch_params* params, const tbrm_world_params* world, int octree_mip,
                                  float* host_out_rgba);
TBRM_API int tbrm_raymarch_octree_device(tbrm_resources* res, const tbrm_camera* camera, const tbrm_tile* tile,
                                         const tbrm_raymarch_params* params, const tbrm_world_params* world, int octree_mip,
                                         const float* device_scene_depth, float* device_out_rgba);

/* Nominal samples of one tile: sum over rays of floor(Steps*thickness) + [frac > 0]. Runs on
 * the GPU with the same cube-setup arithmetic as the raymarch; result is written to *out_samples.        */
TBRM_API int tbrm_count_nominal_samples(tbrm_resources* res, const tbrm_camera* camera, const tbrm_tile* tile,
                                        const tbrm_raymarch_params* params, const tbrm_world_params* world,
                                        uint64_t* out_samples);

/* ------------------------------------------------------------------------------------------------ */
/* readback / sync / interop                                                                          */
TBRM_API int tbrm_download_light_volume(tbrm_resources* res, void* host_out, size_t n_bytes);
TBRM_API int tbrm_upload_light_volume(tbrm_resources* res, const void* host_in, size_t n_bytes);
/* Device address of the light volume in the library's internal 8x8x8-bricked layout (DESIGN.md "Data layout");
 * meant for element-wise device-side combination across GPUs, not for indexing. */
TBRM_API int tbrm_light_volume_device_ptr(tbrm_resources* res, void** out_ptr, size_t* out_bytes);
/* Kernel launches since creation: out[0] = chunked propagation launches, out[1] = slice-per-launch propagation
 * launches (fallback path), out[2] = raymarch launches. Lets tests assert which kernel actually ran. */
/* Device self-test of the kernels' division-free window position (GetTransferFuncPosition, WindowedSampling.usf:14-17, for values
 * filtered out of UNORM data): compares it with the IEEE quotient for EVERY float in [0, 1] and returns the number of
 * mismatching bit patterns (0 expected); *out_fast_path = 0 when the library divides for this window anyway. */
TBRM_API int tbrm_selftest_window_division(int device, float center, float width, uint64_t* out_mismatches, int* out_fast_path);
/* Device self-test of the opacity correction's short form (1 - pow(1 - a, step), WindowedSampling.usf:35, AddDirLightShader.usf:111:
 * the kernels never use the power itself, so they skip what cannot show in 1 - r): compares it with 1 - pow for EVERY float
 * 1 - a in [0, 1] at the two given step sizes (finite, >= 0) and returns the number of mismatching bit patterns (0 expected). */
TBRM_API int tbrm_selftest_opacity_correction(int device, float step0, float step1, uint64_t* out_mismatches);
TBRM_API int tbrm_launch_counters(const tbrm_resources* res, uint64_t out[3]);
/* Of out[0] above, the launches of the pipelined sweep kernel (one per axis pass; the rest are chunks of the chained kernel). */
TBRM_API int tbrm_sweep_launches(const tbrm_resources* res, uint64_t* out);
/* Which path the light operators took since creation, per AXIS PASS and per launch (what a benchmark line or a test needs to
 * say which kernels it measured): out[0] axis passes run as a pipelined sweep, out[1] as the chunked chain, out[2] one slice
 * per launch; out[3] sweep launches (a two-way Change takes two per pass), out[4] chain launches, out[5] slice launches;
 * out[6] occlusion launches that served one pass, out[7] occlusion launches that served both passes of a light
 * (tunable occ_dual); out[8] stream-passes whose occlusion came from the factor cache; out[9] lit-raymarch launches;
 * out[10] sweep launches that propagated two lights' passes at once (tbrm_add_dir_lights); out[11] passes / dual launches whose
 * block lists (empty-block flags, work list, ranks) had to be computed — the others found them with the handle;
 * out[12] device-memory management calls (hipMalloc / hipHostMalloc / hipFree / hipMemGetInfo / event and stream creation) made
 * INSIDE light operators since creation, out[13] host-side waits for a stream made inside them — both stand still once the handle
 * is reserved (tbrm_resources_reserve) and the scene stays inside the reserved envelope; out[14] sweep launches that ran several
 * axis passes at once (k_light_sweep_chain: the next pass fills while the one before drains; tunable sweep_chain); out[15] reserved (0). */
#define TBRM_PATH_COUNTERS 16
TBRM_API int tbrm_path_counters(const tbrm_resources* res, uint64_t out[TBRM_PATH_COUNTERS]);
/* The factor cache of the light operators (no counterpart in the reference, invisible in the results). The expensive half
 * of an axis pass of the Add / Change shaders is its occlusion: the factors 1 - CurrentSample
 * (AddDirLightShader.usf:85-117, ChangeDirLightShader.usf:100-145) of every voxel, which depend on the volume, the transfer
 * function, the window, the clip plane and the light's direction — not on the light volume and not on the light's
 * intensity. A pass that samples the volume for a light which stays in the scene keeps its factors, block-compact: only the
 * 16 x 16 x 8 blocks that can be opaque at all, 8 KiB each (UNORM8 light volumes whose passes are whole brick layers: the
 * passes the pipelined sweep kernel takes; other passes cache nothing). Later operators on the same light then do not sample
 * the volume again: the removed side of a ChangeDirLight, a removal, a re-add after ClearResourceLightVolumes propagate from
 * the kept factors. Entries whose light has left the scene are reused first, then the least recently used; the budget is
 * the tunable light_cache_mb (MiB; 0 = off; the default -1 = an eighth of the device's memory, and never the last free
 * gigabytes). out[0] = stream-passes whose occlusion came from the cache, out[1] = stream-passes whose occlusion was
 * computed, out[2] = entries held, out[3] = their bytes. */
TBRM_API int tbrm_light_cache_stats(const tbrm_resources* res, uint64_t out[4]);
/* Gives the cache's HBM back (blocks until the handle's streams are idle). The next operators sample and keep again —
 * a host that wants the memory for good sets the tunable light_cache_mb to 0 first. */
TBRM_API int tbrm_light_cache_clear(tbrm_resources* res);
/* FlushRenderingCommands(). Also reports a light-propagation sweep that failed on the device (TBRM_ERR_NO_DEVICE with the
 * reason in tbrm_last_error: a tile gave up waiting for its neighbours — the device was reset or starved for seconds —; the
 * light volume is then undefined and has to be rebuilt: ClearResourceLightVolumes + the lights again). The condition is
 * sticky and not tbrm_flush's alone: tbrm_download_light_volume / _slices, tbrm_raymarch_lit (the host-buffer form),
 * tbrm_last_gpu_time_ms and every later light operator return the same error until tbrm_clear_light_volume or
 * tbrm_upload_light_volume defines the light volume again; meanwhile axis passes take the chained kernel. */
TBRM_API int tbrm_flush(tbrm_resources* res);
TBRM_API int tbrm_stream(tbrm_resources* res, void** out_hip_stream);
/* GPU time (ms) of the most recent operator call of each kind, measured with HIP events on the handle's
 * stream; blocks until that work is complete. kind: 0 = add/change/clear (illumination), 1 = raymarch.  */
TBRM_API int tbrm_last_gpu_time_ms(tbrm_resources* res, int kind, float* out_ms);

/* Self-test of the kernels' UNORM decode: writes decode(c) for every 8-bit code (256 floats) and every 16-bit code
 * (65536 floats) as evaluated ON THE DEVICE, so a test can compare them with IEEE c/255 and c/65535. */
TBRM_API int tbrm_selftest_unorm_decode(int device, float* out_u8_256, float* out_u16_65536);

/* Self-test of the kernels' UNORM8 store: out[i] = what in[i] reads back as after a round trip through a UNORM8
 * render target (D3D11: NaN -> 0, clamp to [0,1], trunc(x*255 + 0.5), load code/255), evaluated ON THE DEVICE with the
 * routine the propagation kernels use for WriteBuffer / the light volume (RaymarchVolume.cpp:857-866 picks PF_G8). */
TBRM_API int tbrm_selftest_unorm8_roundtrip(int device, const float* in, size_t n, float* out);

/* ------------------------------------------------------------------------------------------------ */
/* host parameter math, no GPU needed (LightingShaderUtils.cpp:29-265, LightingShaders.cpp:48-131)     */
/* Fills out[0..1]; *n_passes = number of axis passes an Add would run (it breaks on weight == 0).      */
TBRM_API int tbrm_host_light_passes(const tbrm_dir_light_params* light, const tbrm_world_params* world,
                                    const int32_t light_volume_dims[3], int border_mode,
                                    tbrm_light_pass out[2], int* n_passes);
/* The planner alone (no device): which kernel each axis pass of AddDirLightToSingleVolume(light) takes for a light volume of these
 * dimensions. Per pass k: out[4k] = 0 the pipelined sweep, 1 the chunked chain, 2 one slice per launch (-1: no such pass);
 * out[4k+1], out[4k+2] = the sweep's reach in texels along the plane's x / y (chain: the chunk length in out[4k+1]); out[4k+3] = why
 * the sweep declined: 1 previous-slice taps on both sides of the pixel, 2 a reach beyond 14 texels, 3 more hand-off words than the
 * hand-off wave carries, 4 a short ragged downward pass, 5 sweeps off, 6 more than 1024 slices. */
TBRM_API int tbrm_host_plan_light(const tbrm_dir_light_params* light, const tbrm_world_params* world, const int32_t light_volume_dims[3],
                                  int light_volume_32bit, int32_t out[8], int* n_passes);
/* GetLocalClippingParameters (LightingShaderUtils.cpp:205-220), narrowed to float[3] + float[3].        */
TBRM_API int tbrm_host_local_clipping(const tbrm_world_params* world, float out_center[3], float out_dir[3]);
/* Data-volume border colour of the propagation sampler (LightingShaders.h:82-85).                      */
TBRM_API float tbrm_host_data_border(const tbrm_windowing_params* windowing, int border_mode);
/* WorldToLocal of the cube mesh (row-vector 4x3: rows 0-2 = 3x3 incl. 1/scale, row 3 = translation).    */
TBRM_API int tbrm_host_world_to_local(const tbrm_transform* volume_transform, float out_m[12]);

#ifdef __cplusplus
}
#endif
#endif /* TBRM_H */
