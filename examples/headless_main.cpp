// headless_main.cpp -- the reference's start-up and frame loop (EngineMain.cpp:5-23, Engine.cpp:56-80) against the
// mirrored C++ API, without window, editor or GL: load a scene, register instances, render N frames on the MI355X,
// write the last frame as a binary PPM (V flipped like Editor.cpp:93 displays it).
//
//   crt_headless <skybox.ppm> <out.ppm> <width> <height> <frames> <camx> <camy> <camz> <frontx> <fronty> <frontz> <mesh.obj>...
//
// Every mesh is registered once with the identity transform and its own materials (ResourceManager::DefaultMaterial).
// Textures may be JPEG (as upstream's) or binary PPM. Environment: CRT_ASSET_ROOT = the folder that holds `Assets/` (where the
// texture paths inside upstream's .mtl/.clm files resolve, ResourceManager::SetAssetRoot); CRT_DEVICES = "0,1,2,3" renders
// on several GPUs from this one process (Renderer::InitializeDevices) -- the rest of the program does not change.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../clraytracer_amd/host/Renderer.hpp"

int main(int argc, char** argv)
{
    if (argc < 13) { std::fprintf(stderr, "usage: %s sky.ppm out.ppm W H frames cx cy cz fx fy fz mesh.obj...\n", argv[0]); return 2; }
    const char* sky = argv[1]; const char* outPath = argv[2];
    const int width = std::atoi(argv[3]), height = std::atoi(argv[4]), frames = std::atoi(argv[5]);
    const Vector3f camPos((float)std::atof(argv[6]), (float)std::atof(argv[7]), (float)std::atof(argv[8]));
    const Vector3f camFront((float)std::atof(argv[9]), (float)std::atof(argv[10]), (float)std::atof(argv[11]));

    std::vector<int> devices;
    if (const char* list = std::getenv("CRT_DEVICES")) for (const char* p = list; *p;) { devices.push_back(std::atoi(p)); p = std::strchr(p, ','); if (!p) break; ++p; }
    const int ok = devices.size() > 1 ? Renderer::InitializeDevices(devices.data(), (int)devices.size(), width, height)
                                      : Renderer::Initialize(devices.empty() ? 0 : devices[0], width, height);
    if (!ok) { std::fprintf(stderr, "Renderer::Initialize failed (%d)\n", Renderer::LastError()); return 1; }
    if (const char* root = std::getenv("CRT_ASSET_ROOT")) ResourceManager::SetAssetRoot(root);

    // Engine_Start (Engine.cpp:56-80)
    ResourceManager::PrepareMeshes();
    ResourceManager::ImportTexture(sky);                 // must be first: texture index 2 is the skybox
    std::vector<MeshHandle> meshes;
    for (int i = 12; i < argc; ++i) meshes.push_back(ResourceManager::ImportMesh(argv[i]));
    ResourceManager::PushMeshesToGPU();
    ResourceManager::PushTexturesToGPU();
    Renderer::BeginInstanceRegister();
    for (MeshHandle m : meshes) Renderer::RegisterMeshInstance(m, ResourceManager::DefaultMaterial, Matrix4::Identity());
    Renderer::EndInstanceRegister();
    if (Renderer::LastError()) { std::fprintf(stderr, "scene setup failed (%d)\n", Renderer::LastError()); return 1; }

    Camera& cam = Renderer::EditCamera();
    cam.position = camPos; cam.Front = camFront;
    cam.RecalculateView();

    // main loop (EngineMain.cpp:11-17)
    const float sunAngle = -1.96f;                       // Engine.cpp:18
    double ms = 0.0;
    for (int f = 0; f < frames; ++f) {
        if (!Renderer::Render(sunAngle)) { std::fprintf(stderr, "Render failed (%d)\n", Renderer::LastError()); return 1; }
        ms += Renderer::LastFrameMs();
    }
    const float* rgba = Renderer::MapOutput();
    if (!rgba) return 1;
    FILE* f = std::fopen(outPath, "wb");
    if (!f) return 1;
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    for (int y = height - 1; y >= 0; --y)
        for (int x = 0; x < width; ++x) {
            unsigned char px[3];
            for (int c = 0; c < 3; ++c) {
                float v = rgba[4 * ((size_t)y * width + x) + c];
                v = v != v ? 0.0f : (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));   // write_imagef to UNORM8 clamps (hazard H8)
                px[c] = (unsigned char)(v * 255.0f + 0.5f);
            }
            std::fwrite(px, 1, 3, f);
        }
    std::fclose(f);
    std::printf("%d frame(s), %.3f ms GPU time per frame, wrote %s\n", frames, frames ? ms / frames : 0.0, outPath);
    Renderer::Terminate();
    return 0;
}
