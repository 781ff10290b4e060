// Renderer.cpp -- device init, instance table and the per-frame launch (reference: Renderer.cpp:1-394),
// headless and over the C-ABI of libcrt_hip.so instead of OpenCL + GL interop.
#include "Renderer.hpp"
#include "CPURayTrace.hpp"
#include "../../include/crt_api.h"
#include <cstdio>
#include <vector>

uint g_NumMeshInstances = 0u;
static Matrix4 m_MeshTransforms[Renderer::MaxNumInstances];
static MeshInstance m_MeshInstances[Renderer::MaxNumInstances];
Matrix4* g_MeshTransforms = m_MeshTransforms;
MeshInstance* g_MeshInstances = m_MeshInstances;

namespace {
    Camera camera;
    bool deviceReady = false;
    bool postProcess = true;   // upstream always runs PostProcess (Renderer.cpp:360-363)
    bool shadows = false, pipelined = false, unorm8 = false, refraction = false, fxaa = false;
    std::vector<unsigned char> hostFrame8;
    float timeSeconds = 0.0f;
    unsigned frameIndex = 0;
    int lastError = 0;
    std::vector<float> hostFrame;

    uint numRegisteredInstances = 0, lastRegisterInstanceIndex = 0;
    ushort removedInstances[50];
    ushort numRemovedInstances = 0;
    uint MinUpdatedInstanceIndex = 0xFFFFu, MaxUpdatedInstanceIndex = 0u;
    bool shouldUpdateInstances = false;
    bool hasRemovedInstances = false;

    bool check(int rc, const char* what)
    {
        if (rc == 0) return true;
        lastError = rc;
        std::fprintf(stderr, "[Renderer] %s failed: %s\n", what, crt_error_string(rc));
        return false;
    }
    void touch(MeshInstanceHandle h)
    {
        shouldUpdateInstances = true;
        MinUpdatedInstanceIndex = h < MinUpdatedInstanceIndex ? h : MinUpdatedInstanceIndex;
        MaxUpdatedInstanceIndex = (h + 1) > MaxUpdatedInstanceIndex ? (h + 1) : MaxUpdatedInstanceIndex;
    }
}

const Camera& Renderer::GetCamera() { return camera; }
Camera& Renderer::EditCamera() { return camera; }
void Renderer::SetPostProcess(bool enabled) { postProcess = enabled; }
void Renderer::SetShadows(bool enabled) { shadows = enabled; }
void Renderer::SetRefraction(bool enabled) { refraction = enabled; }
void Renderer::SetFXAA(bool enabled) { fxaa = enabled; }
void Renderer::SetUnorm8(bool enabled) { unorm8 = enabled; }
void Renderer::SetPipelined(bool enabled) { pipelined = enabled; }
void Renderer::SetTime(float seconds) { timeSeconds = seconds; }
int Renderer::LastError() { return lastError ? lastError : ResourceManager::LastError(); }
void Renderer::ClearError() { lastError = 0; }
float Renderer::LastFrameMs() { return deviceReady ? crt_last_kernel_ms(0) : -1.0f; }

static int initialize_common(int rc, int width, int height)
{
    lastError = 0;
    camera = Camera();
    camera.RecalculateProjection(width, height);
    camera.RecalculateView();
    if (!check(rc, "crt_init")) return 0;
    deviceReady = true;
    ResourceManager::Initialize(true);
    CPU_RayTraceInitialize();
    g_NumMeshInstances = 0; numRegisteredInstances = 0; lastRegisterInstanceIndex = 0;
    shouldUpdateInstances = false; MinUpdatedInstanceIndex = 0xFFFFu; MaxUpdatedInstanceIndex = 0u;
    frameIndex = 0;
    return 1;
}

int Renderer::Initialize(int device, int width, int height) { return initialize_common(crt_init(device, width, height), width, height); }

// Several GPUs of the node behind the same Renderer: replicated scene, the frame tiled in 16-row bands over the devices,
// Render() returns when the whole frame has been gathered on the first device (include/crt_api.h, crt_init_devices).
int Renderer::InitializeDevices(const int* devices, int numDevices, int width, int height)
{
    return initialize_common(crt_init_devices(devices, numDevices, width, height), width, height);
}

void Renderer::OnWindowResize(int width, int height)
{
    if (width < 16 || height < 16) return;
    if (deviceReady && !check(crt_resize(width, height), "crt_resize")) return;   // a failed resize leaves the old frame size (and projection) in place
    camera.RecalculateProjection(width, height);
}

void Renderer::SetRowBands(int bandRows, int rank, int nRanks)
{
    if (deviceReady) check(crt_set_row_bands(bandRows, rank, nRanks), "crt_set_row_bands");
}

void Renderer::BeginInstanceRegister() { numRegisteredInstances = 0; }

MeshInstanceHandle Renderer::RegisterMeshInstance(MeshHandle handle, MaterialHandle materialHandle, float3 position,
                                                  const Quaternion& rotation, const float3& scale)
{
    return RegisterMeshInstance(handle, materialHandle, Matrix4::PositionRotationScale(position, rotation, scale));
}

MeshInstanceHandle Renderer::RegisterMeshInstance(MeshHandle handle, MaterialHandle materialHandle, const Matrix4& matrix)
{
    if (g_NumMeshInstances >= MaxNumInstances) { // the reference exits here (Renderer.cpp:229)
        std::fprintf(stderr, "[Renderer] too many mesh instances (max %u)\n", MaxNumInstances);
        lastError = CRT_E_OUT_OF_RANGE;
        return ~0u;
    }
    if (materialHandle == ResourceManager::DefaultMaterial) materialHandle = ResourceManager::GetMeshInfo(handle).materialStart;
    MeshInstance& instance = g_MeshInstances[g_NumMeshInstances];
    g_MeshTransforms[g_NumMeshInstances++] = matrix;
    const Matrix4 inv = Matrix4::InverseTransform(matrix);
    std::memset(&instance, 0, sizeof instance);
    std::memcpy(&instance.inverseTransform, &inv, 64);
    instance.meshIndex = handle;
    instance.materialStart = materialHandle;
    return numRegisteredInstances++;
}

void Renderer::EndInstanceRegister()
{
    if (deviceReady && numRegisteredInstances)
        check(crt_upload_instances(g_MeshInstances + lastRegisterInstanceIndex, lastRegisterInstanceIndex, numRegisteredInstances), "crt_upload_instances");
    lastRegisterInstanceIndex += numRegisteredInstances;
    numRegisteredInstances = 0;
}

void Renderer::RemoveMeshInstance(MeshInstanceHandle handle)
{
    // recorded only; upstream never applies removals either (Renderer.cpp:322 "todo")
    if (numRemovedInstances < 50) removedInstances[numRemovedInstances++] = (ushort)handle;
    hasRemovedInstances = true;
}

void Renderer::SetMeshInstanceMaterial(MeshInstanceHandle instanceHandle, MaterialHandle materialHandle)
{
    if (instanceHandle >= g_NumMeshInstances) return;
    g_MeshInstances[instanceHandle].materialStart = materialHandle;
    touch(instanceHandle);
}

void Renderer::SetMeshPosition(MeshInstanceHandle instanceHandle, float3 position)
{
    if (instanceHandle >= g_NumMeshInstances) return;
    Matrix4& transform = g_MeshTransforms[instanceHandle];
    transform.m[3][0] = position.x; transform.m[3][1] = position.y; transform.m[3][2] = position.z;
    const Matrix4 inv = Matrix4::InverseTransform(transform);
    std::memcpy(&g_MeshInstances[instanceHandle].inverseTransform, &inv, 64);
    touch(instanceHandle);
}

void Renderer::SetMeshMatrix(MeshInstanceHandle instanceHandle, const Matrix4& matrix)
{
    if (instanceHandle >= g_NumMeshInstances) return;
    g_MeshTransforms[instanceHandle] = matrix;
    const Matrix4 inv = Matrix4::InverseTransform(matrix);
    std::memcpy(&g_MeshInstances[instanceHandle].inverseTransform, &inv, 64);
    touch(instanceHandle);
}

void Renderer::ClearAllInstances() { g_NumMeshInstances = 0; lastRegisterInstanceIndex = 0; numRegisteredInstances = 0; }

unsigned Renderer::Render(float sunAngle)
{
    if (!deviceReady) { lastError = CRT_E_NOT_INITIALIZED; std::fprintf(stderr, "[Renderer] Render without a device\n"); return 0; }
    if (shouldUpdateInstances) { // Renderer.cpp:312-320
        if (!check(crt_upload_instances(g_MeshInstances + MinUpdatedInstanceIndex, MinUpdatedInstanceIndex,
                                        MaxUpdatedInstanceIndex - MinUpdatedInstanceIndex), "crt_upload_instances")) return 0;
        MinUpdatedInstanceIndex = 0xFFFFu; MaxUpdatedInstanceIndex = 0u;
        shouldUpdateInstances = false;
    }
    CrtTraceArgs args;
    args.cameraPos[0] = camera.position.x; args.cameraPos[1] = camera.position.y; args.cameraPos[2] = camera.position.z;
    args.time = timeSeconds; args.numMeshes = g_NumMeshInstances; args.sunAngle = sunAngle;
    const int flags = (postProcess ? CRT_RENDER_POSTPROCESS : 0) | (shadows ? CRT_RENDER_SHADOWS : 0) | (refraction ? CRT_RENDER_REFRACTION : 0) | (fxaa ? CRT_RENDER_FXAA : 0) | (pipelined ? CRT_RENDER_ASYNC : 0) | (unorm8 ? CRT_RENDER_UNORM8 : 0);
    if (!check(crt_render(&args, &camera.inverseView.m[0][0], &camera.inverseProjection.m[0][0], flags), "crt_render")) return 0;
    return ++frameIndex;
}

const float* Renderer::MapOutput()
{
    if (!deviceReady) return nullptr;
    const size_t n = (size_t)camera.projWidth * (size_t)camera.projHeight * 4;
    hostFrame.resize(n);
    if (!check(crt_read_output(hostFrame.data(), n), "crt_read_output")) return nullptr;
    return hostFrame.data();
}

const unsigned char* Renderer::MapOutputRGBA8()
{
    if (!deviceReady) return nullptr;
    const size_t n = (size_t)camera.projWidth * (size_t)camera.projHeight * 4;
    hostFrame8.resize(n);
    if (!check(crt_read_output_rgba8(hostFrame8.data(), n), "crt_read_output_rgba8")) return nullptr;
    return hostFrame8.data();
}

void Renderer::Terminate()
{
    ResourceManager::Finalize();
    if (deviceReady) crt_shutdown();
    deviceReady = false;
}
