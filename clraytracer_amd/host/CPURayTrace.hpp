// CPURayTrace.hpp -- host mirror of CPURayTrace.hpp:1-18 (single-ray CPU pick ray).
#pragma once
#include "Renderer.hpp"

typedef CrtHitRecord HitRecord;
struct RaySSE { float origin[4]; float direction[4]; }; // two __m128 upstream (Vector4.hpp:63-67)

constexpr float RayacastMissDistance = 1e30f;

void CPU_RayTraceInitialize();
HitRecord CPU_RayCast(RaySSE ray);
// Timing flavour with upstream's instruction mix (_mm_dp_ps, 12-bit _mm_rcp_ps; CPURayTrace.cpp:42-128). `rcpps` is
// implementation-defined, so results may differ from CPU_RayCast in the last places; not used for parity.
HitRecord CPU_RayCastSSE(RaySSE ray);
