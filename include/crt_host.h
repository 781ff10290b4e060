/* crt_host.h -- C linkage over the C++ host mirror (libcrt_host.so) so that non-C++ drivers
 * (the Python tests, bench.py) can run the reference's own call sequence
 * (Engine.cpp:56-80 start-up, EngineMain.cpp:11-17 per-frame) against
 * Renderer:: / ResourceManager:: / AssetManager_ / CPU_RayCast. One thin forwarding function per
 * mirrored C++ entry point; C++ callers should include the headers in clraytracer_amd/host/ directly.
 */
#ifndef CRT_HOST_H
#define CRT_HOST_H
#include <stddef.h>
#include <stdint.h>
#include "crt_types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Renderer::Initialize (Renderer.cpp:175). Returns 1 on success, 0 on failure (crth_last_error). */
int crth_initialize(int device, int width, int height);
/* Renderer::InitializeDevices: several GPUs in this process behind the same calls (crt_init_devices). */
int crth_initialize_devices(const int* devices, int numDevices, int width, int height);
/* Host-only session: importer, BVH build and CPU_RayCast work; crth_render fails loudly. */
int crth_initialize_host_only(int width, int height);
void crth_terminate(void);                                   /* Renderer::Terminate */
int crth_last_error(void);
void crth_clear_error(void);                 /* Renderer::ClearError: forget a reported Renderer error (e.g. a refused resize) */

void crth_prepare_meshes(void);                              /* ResourceManager::PrepareMeshes */
int crth_import_texture(const char* path);                   /* ResourceManager::ImportTexture */
int crth_import_texture_rgb8(const char* name, int width, int height, const unsigned char* rgb);
int crth_import_mesh(const char* path);                      /* ResourceManager::ImportMesh */
void crth_push_meshes(void);                                 /* ResourceManager::PushMeshesToGPU */
void crth_set_device_bvh_build(int enabled);                 /* ResourceManager::SetDeviceBVHBuild */
void crth_set_mesh_cache(int enabled);                       /* AssetManager_SetMeshCache: the `.clm` cache (AssetManager.cpp:291-381), on by default */
size_t crth_qlz_decompress(const unsigned char* src, size_t srcLen, unsigned char* dst, size_t dstCap); /* QuickLZ 1.5.0 level 1 */
size_t crth_qlz_store(const unsigned char* src, size_t size, unsigned char* dst);                        /* stored block, size + 9 bytes */
size_t crth_qlz_compress(const unsigned char* src, size_t size, unsigned char* dst);                     /* upstream's qlz_compress (level 1) byte for byte; dst: size + 400 bytes */
/* JPEG -> RGB8 as the reference's stbi_load(path, &w, &h, &channels, 3) (ResourceManager.cpp:193). Returns the number of
 * bytes written (width*height*3), or 0 on failure (*error, if given, names the reason). With dst == NULL only the header is
 * read: info[0..3] = width, height, components in the file (1 or 3), progressive; the return value is the size needed. */
size_t crth_jpeg_decode(const unsigned char* data, size_t size, unsigned char* dst, size_t dstCap, int info[4], const char** error);
void crth_set_asset_root(const char* dir);                   /* ResourceManager::SetAssetRoot: where relative texture paths of .mtl/.clm files resolve */
void crth_push_textures(void);                               /* ResourceManager::PushTexturesToGPU */
void crth_push_materials(void);                              /* ResourceManager::PushMaterialsToGPU */
int crth_create_material(int count);                         /* ResourceManager::CreateMaterial -> first handle, -1 on failure */
void crth_edit_material(int handle, const CrtMaterial* value); /* ResourceManager::EditMaterial(handle) = *value */

void crth_begin_instances(void);                             /* Renderer::BeginInstanceRegister */
unsigned crth_register_instance(int mesh, int material, const float matrix[16]); /* material 0xFFFF = DefaultMaterial */
void crth_end_instances(void);                               /* Renderer::EndInstanceRegister */
void crth_clear_instances(void);
void crth_set_mesh_matrix(unsigned instance, const float matrix[16]);
void crth_set_mesh_position(unsigned instance, const float position[3]);
void crth_set_instance_material(unsigned instance, int material);

void crth_set_camera(const float position[3], const float front[3]); /* Camera position/Front + RecalculateView */
void crth_get_camera(float invView[16], float invProj[16], float position[3]);
void crth_resize(int width, int height);                     /* Renderer::OnWindowResize */
void crth_set_postprocess(int enabled);
void crth_set_shadows(int enabled);                          /* Renderer::SetShadows (extension) */
void crth_set_fxaa(int enabled);                             /* Renderer::SetFXAA (extension) */
void crth_set_refraction(int enabled);                       /* Renderer::SetRefraction (extension) */
void crth_set_unorm8(int enabled);                           /* Renderer::SetUnorm8 (hazard H8) */
const unsigned char* crth_map_output_rgba8(void);            /* Renderer::MapOutputRGBA8 */
void crth_set_pipelined(int enabled);                        /* Renderer::SetPipelined (frames in flight) */
void crth_set_row_bands(int bandRows, int rank, int nRanks);
unsigned crth_render(float sunAngle);                        /* Renderer::Render: frame index, 0 on failure */
const float* crth_map_output(void);                          /* Renderer::MapOutput */
float crth_last_frame_ms(void);

/* CPU_RayCast (CPURayTrace.cpp:186) over n rays (xyz triples). */
void crth_cpu_raycast(const float* origins, const float* dirs, int n, CrtHitRecord* out, int nthreads);
/* CPU_RayCastSSE: the same with upstream's SSE instruction mix (_mm_dp_ps, approximate _mm_rcp_ps) -- the timing flavour. */
void crth_cpu_raycast_sse(const float* origins, const float* dirs, int n, CrtHitRecord* out, int nthreads);

/* read-only views of the host arenas (ResourceManager.cpp:49-55, Renderer.cpp:45-50) */
const CrtTri* crth_triangles(void);         size_t crth_num_triangles(void);
const CrtBVHNode* crth_nodes(void);         size_t crth_num_nodes(void);
const uint32_t* crth_roots(void);           int crth_num_meshes(void);
const CrtMaterial* crth_materials(void);    int crth_num_materials(void);
const CrtTexture* crth_textures(void);      int crth_num_textures(void);
const CrtRGB8* crth_texels(void);           size_t crth_texel_bytes(void);
const CrtMeshInstance* crth_instances(void); unsigned crth_num_instances(void);
void crth_mesh_info(int mesh, uint32_t out[4]); /* numTriangles, triangleStart, materialStart, numMaterials */

/* stand-alone pieces for known-answer tests */
uint32_t crth_build_bvh(CrtTri* tris, const uint32_t* meshTriCounts, int numMeshes, CrtBVHNode* nodes, uint32_t* roots);
uint16_t crth_float_to_half(float v);
float crth_half_to_float(uint16_t h);
void crth_inverse_transform(const float in[16], float out[16]);
void crth_inverse(const float in[16], float out[16]);
void crth_perspective_fov_rh(float fovRad, float width, float height, float zNear, float zFar, float out[16]);
void crth_look_at_rh(const float eye[3], const float front[3], const float up[3], float out[16]);
int crth_write_obj(const char* path, const float* positions, int numPositions, const float* uvs, int numUvs,
                   const float* normals, int numNormals, const int* faces, const int* faceMaterial, int numFaces,
                   const char* const* materialNames, int numMaterials);

/* A barrier for the ranks of ONE node in POSIX shared memory (no reference counterpart; host/ShmBarrier.cpp): bench.py brackets its timed
 * region with it so that a TCP / collective barrier's latency and exit skew are not counted as rendering time. `name` starts with '/';
 * one rank creates (create = 1) before the others open; crth_shm_barrier_wait returns 0, or -1 when not every rank arrived within
 * timeoutMs (the caller must give up: the barrier is then unusable); close unlinks the object on the creating rank. */
void* crth_shm_barrier_open(const char* name, int nRanks, int create);
int crth_shm_barrier_wait(void* handle, int timeoutMs);
void crth_shm_barrier_close(void* handle);

#ifdef __cplusplus
}
#endif
#endif
