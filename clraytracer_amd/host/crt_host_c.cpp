// crt_host_c.cpp -- C linkage forwarding to the C++ host mirror (see include/crt_host.h).
#include "../../include/crt_host.h"
#include "../../include/crt_api.h"
#include "AssetManager.hpp"
#include "CPURayTrace.hpp"
#include "JpegDecode.hpp"
#include "Renderer.hpp"
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace ResourceManager { size_t TexelBytesUsed(); int NumMaterials(); int NumTextures(); size_t NumTriangles(); size_t NumNodes(); }

namespace { bool hostOnly = false; int hostOnlyError = 0; }

extern "C" {

// (a new session starts without the last host-only session's refusal: crth_last_error used to keep reporting it)
int crth_initialize(int device, int width, int height) { hostOnly = false; hostOnlyError = 0; return Renderer::Initialize(device, width, height); }
int crth_initialize_devices(const int* devices, int numDevices, int width, int height) { hostOnly = false; hostOnlyError = 0; return Renderer::InitializeDevices(devices, numDevices, width, height); }

int crth_initialize_host_only(int width, int height)
{
    hostOnly = true; hostOnlyError = 0;
    ResourceManager::Initialize(false);
    g_NumMeshInstances = 0;
    Renderer::ClearAllInstances();
    Camera& cam = Renderer::EditCamera();
    cam = Camera();
    cam.RecalculateProjection(width, height);
    cam.RecalculateView();
    return 1;
}

void crth_terminate(void) { if (hostOnly) ResourceManager::Finalize(); else Renderer::Terminate(); hostOnly = false; }
int crth_last_error(void) { return hostOnlyError ? hostOnlyError : Renderer::LastError(); }
void crth_clear_error(void) { Renderer::ClearError(); }

void crth_prepare_meshes(void) { ResourceManager::PrepareMeshes(); }
int crth_import_texture(const char* path) { return ResourceManager::ImportTexture(path); }
int crth_import_texture_rgb8(const char* name, int w, int h, const unsigned char* rgb) { return ResourceManager::ImportTextureRGB8(name, w, h, rgb); }
int crth_import_mesh(const char* path) { return ResourceManager::ImportMesh(path); }
void crth_push_meshes(void) { ResourceManager::PushMeshesToGPU(); }
void crth_set_mesh_cache(int enabled) { AssetManager_SetMeshCache(enabled != 0); }
size_t crth_qlz_decompress(const unsigned char* src, size_t srcLen, unsigned char* dst, size_t dstCap) { return MeshCache_QlzDecompress(src, srcLen, dst, dstCap); }
size_t crth_jpeg_decode(const unsigned char* data, size_t size, unsigned char* dst, size_t dstCap, int info[4], const char** error)
{
    std::vector<unsigned char> rgb;
    JpegInfo ji;
    if (!JpegDecodeRGB8(data, size, &ji, rgb, error)) return 0;
    if (info) { info[0] = ji.width; info[1] = ji.height; info[2] = ji.components; info[3] = ji.progressive ? 1 : 0; }
    if (dst) { if (dstCap < rgb.size()) return 0; std::memcpy(dst, rgb.data(), rgb.size()); }
    return rgb.size();
}
void crth_set_asset_root(const char* dir) { ResourceManager::SetAssetRoot(dir); }
size_t crth_qlz_store(const unsigned char* src, size_t size, unsigned char* dst) { return MeshCache_QlzStore(src, size, dst); }
size_t crth_qlz_compress(const unsigned char* src, size_t size, unsigned char* dst) { return MeshCache_QlzCompress(src, size, dst); }
void crth_set_device_bvh_build(int enabled) { ResourceManager::SetDeviceBVHBuild(enabled != 0); }
void crth_push_textures(void) { ResourceManager::PushTexturesToGPU(); }
void crth_push_materials(void) { ResourceManager::PushMaterialsToGPU(); }
int crth_create_material(int count) { MaterialHandle h = 0; return ResourceManager::CreateMaterial(&h, count) ? (int)h : -1; }
void crth_edit_material(int handle, const CrtMaterial* value) { if (value && handle >= 0 && handle < CRT_MAX_MATERIALS) ResourceManager::EditMaterial((MaterialHandle)handle) = *value; }

void crth_begin_instances(void) { Renderer::BeginInstanceRegister(); }
unsigned crth_register_instance(int mesh, int material, const float matrix[16])
{
    Matrix4 m; std::memcpy(&m, matrix, 64);
    return Renderer::RegisterMeshInstance((MeshHandle)mesh, (MaterialHandle)material, m);
}
void crth_end_instances(void) { Renderer::EndInstanceRegister(); }
void crth_clear_instances(void) { Renderer::ClearAllInstances(); }
void crth_set_mesh_matrix(unsigned instance, const float matrix[16]) { Matrix4 m; std::memcpy(&m, matrix, 64); Renderer::SetMeshMatrix(instance, m); }
void crth_set_mesh_position(unsigned instance, const float p[3]) { Renderer::SetMeshPosition(instance, float3(p[0], p[1], p[2])); }
void crth_set_instance_material(unsigned instance, int material) { Renderer::SetMeshInstanceMaterial(instance, (MaterialHandle)material); }

void crth_set_camera(const float position[3], const float front[3])
{
    Camera& cam = Renderer::EditCamera();
    cam.position = Vector3f(position[0], position[1], position[2]);
    cam.Front = Vector3f(front[0], front[1], front[2]);
    cam.RecalculateView();
}
void crth_get_camera(float invView[16], float invProj[16], float position[3])
{
    const Camera& cam = Renderer::GetCamera();
    std::memcpy(invView, &cam.inverseView, 64);
    std::memcpy(invProj, &cam.inverseProjection, 64);
    position[0] = cam.position.x; position[1] = cam.position.y; position[2] = cam.position.z;
}
void crth_resize(int width, int height)
{
    if (hostOnly) { if (width >= 16 && height >= 16) Renderer::EditCamera().RecalculateProjection(width, height); }
    else Renderer::OnWindowResize(width, height);
}
void crth_set_postprocess(int enabled) { Renderer::SetPostProcess(enabled != 0); }
void crth_set_shadows(int enabled) { Renderer::SetShadows(enabled != 0); }
void crth_set_refraction(int enabled) { Renderer::SetRefraction(enabled != 0); }
void crth_set_fxaa(int enabled) { Renderer::SetFXAA(enabled != 0); }
void crth_set_unorm8(int enabled) { Renderer::SetUnorm8(enabled != 0); }
const unsigned char* crth_map_output_rgba8(void) { return hostOnly ? nullptr : Renderer::MapOutputRGBA8(); }
void crth_set_pipelined(int enabled) { Renderer::SetPipelined(enabled != 0); }
void crth_set_row_bands(int bandRows, int rank, int nRanks) { Renderer::SetRowBands(bandRows, rank, nRanks); }
unsigned crth_render(float sunAngle)
{
    if (hostOnly) { hostOnlyError = CRT_E_NOT_INITIALIZED; std::fprintf(stderr, "[crth] render requested in a host-only session: there is no CPU render path\n"); return 0; }
    return Renderer::Render(sunAngle);
}
const float* crth_map_output(void) { return hostOnly ? nullptr : Renderer::MapOutput(); }
float crth_last_frame_ms(void) { return hostOnly ? -1.0f : Renderer::LastFrameMs(); }

static void cpu_raycast_many(HitRecord (*cast)(RaySSE), const float* origins, const float* dirs, int n, CrtHitRecord* out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    // blocks of 4096 rays dealt round-robin: neighbouring rays stay on one thread (cache-friendly), load stays balanced
    auto work = [&](int t) {
        for (int base = t * 4096; base < n; base += nthreads * 4096) {
            const int end = base + 4096 < n ? base + 4096 : n;
            for (int k = base; k < end; ++k) {
                RaySSE r;
                r.origin[0] = origins[3 * k]; r.origin[1] = origins[3 * k + 1]; r.origin[2] = origins[3 * k + 2]; r.origin[3] = 1.0f;
                r.direction[0] = dirs[3 * k]; r.direction[1] = dirs[3 * k + 1]; r.direction[2] = dirs[3 * k + 2]; r.direction[3] = 0.0f;
                out[k] = cast(r);
            }
        }
    };
    if (nthreads == 1) { work(0); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
}
void crth_cpu_raycast(const float* origins, const float* dirs, int n, CrtHitRecord* out, int nthreads) { cpu_raycast_many(CPU_RayCast, origins, dirs, n, out, nthreads); }
void crth_cpu_raycast_sse(const float* origins, const float* dirs, int n, CrtHitRecord* out, int nthreads) { cpu_raycast_many(CPU_RayCastSSE, origins, dirs, n, out, nthreads); }

const CrtTri* crth_triangles(void) { return g_Triangles; }
size_t crth_num_triangles(void) { return ResourceManager::NumTriangles(); }
const CrtBVHNode* crth_nodes(void) { return g_BVHNodes; }
size_t crth_num_nodes(void) { return ResourceManager::NumNodes(); }
const uint32_t* crth_roots(void) { return g_BVHIndices; }
int crth_num_meshes(void) { return ResourceManager::GetNumMeshes(); }
const CrtMaterial* crth_materials(void) { return g_Materials; }
int crth_num_materials(void) { return ResourceManager::NumMaterials(); }
const CrtTexture* crth_textures(void) { return g_Textures; }
int crth_num_textures(void) { return ResourceManager::NumTextures(); }
const CrtRGB8* crth_texels(void) { return g_TexturePixels; }
size_t crth_texel_bytes(void) { return ResourceManager::TexelBytesUsed(); }
const CrtMeshInstance* crth_instances(void) { return g_MeshInstances; }
unsigned crth_num_instances(void) { return g_NumMeshInstances; }
void crth_mesh_info(int mesh, uint32_t out[4])
{
    MeshInfo mi = ResourceManager::GetMeshInfo((MeshHandle)mesh);
    out[0] = mi.numTriangles; out[1] = mi.triangleStart; out[2] = mi.materialStart; out[3] = mi.numMaterials;
}

uint32_t crth_build_bvh(CrtTri* tris, const uint32_t* meshTriCounts, int numMeshes, CrtBVHNode* nodes, uint32_t* roots)
{
    std::vector<MeshInfo> infos((size_t)numMeshes);
    uint32_t start = 0;
    for (int i = 0; i < numMeshes; ++i) { infos[i] = MeshInfo{ meshTriCounts[i], start, 0, 0, nullptr }; start += meshTriCounts[i]; }
    ResetBVHNodeCounter();
    uint32_t n = BuildBVH(tris, infos.data(), numMeshes, nodes, roots);
    ResetBVHNodeCounter();
    return n;
}
uint16_t crth_float_to_half(float v) { return crtmath::ConvertFloatToHalf(v); }
float crth_half_to_float(uint16_t h) { return crtmath::ConvertHalfToFloat(h); }
void crth_inverse_transform(const float in[16], float out[16]) { Matrix4 m; std::memcpy(&m, in, 64); m = Matrix4::InverseTransform(m); std::memcpy(out, &m, 64); }
void crth_inverse(const float in[16], float out[16]) { Matrix4 m; std::memcpy(&m, in, 64); m = Matrix4::Inverse(m); std::memcpy(out, &m, 64); }
void crth_perspective_fov_rh(float fov, float w, float h, float zn, float zf, float out[16]) { Matrix4 m = Matrix4::PerspectiveFovRH(fov, w, h, zn, zf); std::memcpy(out, &m, 64); }
void crth_look_at_rh(const float eye[3], const float front[3], const float up[3], float out[16])
{
    Matrix4 m = Matrix4::LookAtRH(Vector3f(eye[0], eye[1], eye[2]), Vector3f(front[0], front[1], front[2]), Vector3f(up[0], up[1], up[2]));
    std::memcpy(out, &m, 64);
}
int crth_write_obj(const char* path, const float* positions, int numPositions, const float* uvs, int numUvs,
                   const float* normals, int numNormals, const int* faces, const int* faceMaterial, int numFaces,
                   const char* const* materialNames, int numMaterials)
{
    return AssetManager_WriteObj(path, positions, numPositions, uvs, numUvs, normals, numNormals, faces, faceMaterial, numFaces, materialNames, numMaterials);
}

} // extern "C"
