// Renderer.hpp -- host mirror of the reference's Renderer namespace (Renderer.hpp:1-46): device
// init, instance table, per-frame launch. GL/window parts are gone (headless); Render() returns a
// frame counter instead of a GL texture id and MapOutput() exposes the HDR float4 frame.
#pragma once
#include "ResourceManager.hpp"

typedef CrtMeshInstance MeshInstance;
typedef uint MeshInstanceHandle;

namespace Renderer
{
    constexpr uint MaxNumInstances = 401;

    // device: HIP ordinal; width/height: initial frame (the reference takes them from the window)
    int Initialize(int device = 0, int width = 1249, int height = 720);
    // several GPUs in this process (extension: upstream drives one device): same API afterwards, Render() delivers the
    // whole frame; `devices` lists HIP ordinals (a GPU may be named twice to rehearse the path on a one-GPU box)
    int InitializeDevices(const int* devices, int numDevices, int width = 1249, int height = 720);
    void Terminate();
    // one frame: RayGen -> Trace -> PostProcess -> wait (Renderer.cpp:305-375). Returns the frame
    // index (>0) or 0 on failure (see LastError()).
    unsigned Render(float sunAngle);
    void OnWindowResize(int width, int height);

    void BeginInstanceRegister();
    MeshInstanceHandle RegisterMeshInstance(MeshHandle handle, MaterialHandle material, const Matrix4& mat);
    MeshInstanceHandle RegisterMeshInstance(MeshHandle handle, MaterialHandle material, float3 position, const Quaternion& rotation, const float3& scale);
    void EndInstanceRegister();
    void RemoveMeshInstance(MeshInstanceHandle handle);
    void ClearAllInstances();

    void SetMeshInstanceMaterial(MeshInstanceHandle meshHandle, MaterialHandle materialHandle);
    void SetMeshPosition(MeshInstanceHandle handle, float3 position);
    void SetMeshMatrix(MeshInstanceHandle handle, const Matrix4& matrix);
    const Camera& GetCamera();

    // ---- additions for the headless build ----
    Camera& EditCamera();                 // set position/Front, then RecalculateView()
    void SetPostProcess(bool enabled);    // PostProcess is on upstream; parity is judged pre-post
    void SetUnorm8(bool enabled);         // quantise like upstream's RGBA8 render target (hazard H8); MapOutputRGBA8() returns the bytes
    const unsigned char* MapOutputRGBA8();
    void SetShadows(bool enabled);        // extension: the shadow ray upstream leaves as a TODO (kernel_main.cl:256-258); off by default
    void SetFXAA(bool enabled);           // extension: upstream's FXAA (kernel_main.cl:289-340) is dead code (call commented out, kernel_main.cl:349); runs it as the first PostProcess stage; off by default
    void SetRefraction(bool enabled);     // extension: upstream's README TODO "refraction / transculency": materials with MTL d < 1 transmit; off by default
    void SetPipelined(bool enabled);      // Render() returns without waiting (frames in flight); MapOutput()/uploads wait. Off by default (upstream clFinish()es)
    void SetTime(float seconds);          // TraceArgs.time (Window::GetTime upstream)
    void SetRowBands(int bandRows, int rank, int nRanks); // multi-GPU image tiling
    const float* MapOutput();             // host copy of the float4 frame (width*height*4), valid until next Render
    float LastFrameMs();                  // HIP-event time of the last frame's kernels
    int LastError();                      // first error since Initialize / ClearError (errors are returned as codes, not exit(0): DESIGN.md 8)
    void ClearError();
}

extern uint g_NumMeshInstances;
extern Matrix4* g_MeshTransforms;
extern MeshInstance* g_MeshInstances;
