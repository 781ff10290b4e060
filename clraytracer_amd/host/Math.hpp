// Math.hpp -- the slice of the reference's Math/*.hpp that feeds the ray-trace path, in portable
// scalar C++ (the reference is SSE/MSVC-only). Operation order follows the SIMD originals lane by
// lane so results are bit-identical wherever the original is deterministic; the two places where
// the original uses hardware estimate instructions are pinned to IEEE (noted inline).
//
//   half conversions  <- Math/Math.hpp:154-201
//   Sin/Cos/FMod      <- Math/Math.hpp:33-38,92-112
//   Matrix4           <- Math/Matrix.hpp (Identity 169, LookAtRH 211, PerspectiveFovRH 237,
//                        InverseTransform 292, Inverse 327, Multiply 376, PositionRotationScale 433)
//   Camera            <- Math/Camera.hpp:7-136 (window/input-free subset)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include "../../include/crt_types.h"

typedef unsigned int uint;
typedef unsigned short ushort;
typedef crt_half half;

struct Vector2f { float x = 0, y = 0; };
struct Vector3f {
    float x = 0, y = 0, z = 0;
    Vector3f() {}
    Vector3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
typedef Vector3f float3;
struct Quaternion { float x = 0, y = 0, z = 0, w = 1; };

namespace crtmath {

inline uint32_t BitsOf(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float FloatOf(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// (int)x with defined behaviour for NaN / out-of-range input (UB upstream)
inline int TruncToInt(float x)
{
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int)x;
}

// Math.hpp:190-197: adds half an ulp before truncating (round-half-up), saturates to 0x7FFF
inline half ConvertFloatToHalf(float value)
{
    const uint32_t b = BitsOf(value) + 0x00001000u;
    const uint32_t e = (b & 0x7F800000u) >> 23;
    const uint32_t m = b & 0x007FFFFFu;
    uint32_t r = (b & 0x80000000u) >> 16;
    if (e > 112) r |= (((e - 112) << 10) & 0x7C00u) | (m >> 13);
    if (e < 113 && e > 101) r |= (((0x007FF000u + m) >> (125 - e)) + 1) >> 1;
    if (e > 143) r |= 0x7FFFu;
    return (half)r;
}

// Math.hpp:156-164
inline float ConvertHalfToFloat(half x)
{
    const uint32_t e = (x & 0x7C00u) >> 10;
    const uint32_t m = (uint32_t)(x & 0x03FFu) << 13;
    uint32_t a = (uint32_t)(x & 0x8000u) << 16;
    if (e != 0) a |= ((e + 112) << 23) | m;
    else if (m != 0) {
        const uint32_t v = BitsOf((float)m) >> 23;
        a |= ((v - 37) << 23) | ((m << (150 - v)) & 0x007FE000u);
    }
    return FloatOf(a);
}

constexpr float PI = 3.14159265358f;
constexpr float DegToRad = PI / 180.0f;
constexpr float TwoPI = PI * 2.0f;

inline float FMod(float x, float y)
{
    float quotient = x / y;
    float whole = (float)TruncToInt(quotient);
    float remainder = x - whole * y;
    remainder += (float)(remainder < 0.0f) * y;
    return remainder;
}
inline float Sin(float x)
{
    x = FMod(x + PI, TwoPI) - PI;
    float xx = x * x * x;
    float t = x - (xx * 0.16666666666f);
    t += (xx *= x * x) * 0.00833333333f;
    t -= (xx *= x * x) * 0.00019841269f;
    t += (xx * x * x) / 362880.0f;
    return t;
}
inline float Cos(float x)
{
    x = FMod(x + PI, TwoPI) - PI;
    float xx = x * x;
    float t = 1.0f - (xx * 0.5f);
    t += (xx *= x * x) * 0.04166666666f;
    t -= (xx *= x * x) * 0.00138888888f;
    t += (xx * x * x) / 40320.0f;
    return t;
}

// Math.hpp:233-239
inline uint PackColorRGBU32(const float* c)
{
    return (uint)(c[0] * 255.0f) | ((uint)(c[1] * 255.0f) << 8) | ((uint)(c[2] * 255.0f) << 16);
}

} // namespace crtmath

// Row-major 4x4, row-vector convention (v' = v * M), same memory layout as the reference's Matrix4.
struct Matrix4 {
    float m[4][4];

    static Matrix4 Identity()
    {
        Matrix4 r; std::memset(&r, 0, sizeof r);
        r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f;
        return r;
    }
    static Matrix4 FromPosition(float x, float y, float z)
    {
        Matrix4 r = Identity(); r.m[3][0] = x; r.m[3][1] = y; r.m[3][2] = z; return r;
    }
    static Matrix4 CreateScale(float x, float y, float z)
    {
        Matrix4 r = Identity(); r.m[0][0] = x; r.m[1][1] = y; r.m[2][2] = z; return r;
    }
    // Matrix.hpp:376-433: out.row[i] = (in1.r0*in2[i][0] + in1.r1*in2[i][1]) + (in1.r2*in2[i][2] + in1.r3*in2[i][3])
    static Matrix4 Multiply(const Matrix4& in1, const Matrix4& in2)
    {
        Matrix4 out;
        for (int i = 0; i < 4; ++i)
            for (int c = 0; c < 4; ++c)
                out.m[i][c] = (in1.m[0][c] * in2.m[i][0] + in1.m[1][c] * in2.m[i][1]) +
                              (in1.m[2][c] * in2.m[i][2] + in1.m[3][c] * in2.m[i][3]);
        return out;
    }
    // Rotation matrix of a unit quaternion (row-vector convention, as DirectXMath's
    // XMMatrixRotationQuaternion which Matrix.hpp:572-613 follows).
    static Matrix4 FromQuaternion(const Quaternion& q)
    {
        const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
        const float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
        const float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
        Matrix4 r = Identity();
        r.m[0][0] = 1.0f - 2.0f * (yy + zz); r.m[0][1] = 2.0f * (xy + wz); r.m[0][2] = 2.0f * (xz - wy);
        r.m[1][0] = 2.0f * (xy - wz); r.m[1][1] = 1.0f - 2.0f * (xx + zz); r.m[1][2] = 2.0f * (yz + wx);
        r.m[2][0] = 2.0f * (xz + wy); r.m[2][1] = 2.0f * (yz - wx); r.m[2][2] = 1.0f - 2.0f * (xx + yy);
        return r;
    }
    // Matrix.hpp:433-440. NOTE: upstream scales by *position* (CreateScale(position)), not by
    // `scale`; kept, because callers of the reference see exactly that.
    static Matrix4 PositionRotationScale(const Vector3f& position, const Quaternion& rotation, const Vector3f& /*scale*/)
    {
        Matrix4 result = Identity();
        result = Multiply(result, FromPosition(position.x, position.y, position.z));
        result = Multiply(result, FromQuaternion(rotation));
        result = Multiply(result, CreateScale(position.x, position.y, position.z));
        return result;
    }

    // Matrix.hpp:292-325: inverse of a TRS matrix (rotation rows may carry scale)
    static Matrix4 InverseTransform(const Matrix4& in)
    {
        Matrix4 out;
        float r[3][4];
        for (int k = 0; k < 3; ++k) { // transposed 3x3, 4th lane = in[2][3]
            r[k][0] = in.m[0][k]; r[k][1] = in.m[1][k]; r[k][2] = in.m[2][k]; r[k][3] = in.m[2][3];
        }
        float rs[4];
        for (int c = 0; c < 4; ++c) {
            float sizeSqr = r[0][c] * r[0][c];
            sizeSqr = sizeSqr + r[1][c] * r[1][c];
            sizeSqr = sizeSqr + r[2][c] * r[2][c];
            rs[c] = (sizeSqr < 1.e-8f) ? 1.0f : (1.0f / sizeSqr);
        }
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 4; ++c) out.m[k][c] = r[k][c] * rs[c];
        const float id3[4] = { 0.f, 0.f, 0.f, 1.f };
        for (int c = 0; c < 4; ++c) {
            float t = out.m[0][c] * in.m[3][0];
            t = t + out.m[1][c] * in.m[3][1];
            t = t + out.m[2][c] * in.m[3][2];
            out.m[3][c] = id3[c] - t;
        }
        return out;
    }

    // Matrix.hpp:327-374: general inverse by 2x2 block adjugates. Lanes written out explicitly.
    static Matrix4 Inverse(const Matrix4& M)
    {
        const float (*a)[4] = M.m;
        // blocks (row-major 2x2): A = rows0-1/cols0-1, B = rows0-1/cols2-3, C = rows2-3/cols0-1, D = rows2-3/cols2-3
        const float A[4] = { a[0][0], a[0][1], a[1][0], a[1][1] };
        const float B[4] = { a[0][2], a[0][3], a[1][2], a[1][3] };
        const float C[4] = { a[2][0], a[2][1], a[3][0], a[3][1] };
        const float D[4] = { a[2][2], a[2][3], a[3][2], a[3][3] };
        const float detA = a[0][0] * a[1][1] - a[0][1] * a[1][0];
        const float detB = a[0][2] * a[1][3] - a[0][3] * a[1][2];
        const float detC = a[2][0] * a[3][1] - a[2][1] * a[3][0];
        const float detD = a[2][2] * a[3][3] - a[2][3] * a[3][2];
        auto adjMul = [](const float* p, const float* q, float* o) { // (p#)*q
            o[0] = p[3] * q[0] - p[1] * q[2]; o[1] = p[3] * q[1] - p[1] * q[3];
            o[2] = p[0] * q[2] - p[2] * q[0]; o[3] = p[0] * q[3] - p[2] * q[1];
        };
        auto mul = [](const float* p, const float* q, float* o) { // p*q
            o[0] = p[0] * q[0] + p[1] * q[2]; o[1] = p[1] * q[3] + p[0] * q[1];
            o[2] = p[2] * q[0] + p[3] * q[2]; o[3] = p[3] * q[3] + p[2] * q[1];
        };
        auto mulAdj = [](const float* p, const float* q, float* o) { // p*(q#)
            o[0] = p[0] * q[3] - p[1] * q[2]; o[1] = p[1] * q[0] - p[0] * q[1];
            o[2] = p[2] * q[3] - p[3] * q[2]; o[3] = p[3] * q[0] - p[2] * q[1];
        };
        float D_C[4], A_B[4], t4[4], X[4], W[4], Y[4], Z[4];
        adjMul(D, C, D_C);
        adjMul(A, B, A_B);
        mul(B, D_C, t4); for (int i = 0; i < 4; ++i) X[i] = detD * A[i] - t4[i];
        mul(C, A_B, t4); for (int i = 0; i < 4; ++i) W[i] = detA * D[i] - t4[i];
        float detM = detA * detD;
        mulAdj(D, A_B, t4); for (int i = 0; i < 4; ++i) Y[i] = detB * C[i] - t4[i];
        mulAdj(A, D_C, t4); for (int i = 0; i < 4; ++i) Z[i] = detC * B[i] - t4[i];
        detM = detM + detB * detC;
        const float tr0 = A_B[0] * D_C[0], tr1 = A_B[1] * D_C[2], tr2 = A_B[2] * D_C[1], tr3 = A_B[3] * D_C[3];
        detM = detM - ((tr0 + tr1) + (tr2 + tr3));
        const float sign[4] = { 1.f, -1.f, -1.f, 1.f };
        float rDet[4];
        for (int i = 0; i < 4; ++i) rDet[i] = sign[i] / detM;
        for (int i = 0; i < 4; ++i) { X[i] *= rDet[i]; Y[i] *= rDet[i]; Z[i] *= rDet[i]; W[i] *= rDet[i]; }
        Matrix4 out;
        out.m[0][0] = X[3]; out.m[0][1] = X[1]; out.m[0][2] = Y[3]; out.m[0][3] = Y[1];
        out.m[1][0] = X[2]; out.m[1][1] = X[0]; out.m[1][2] = Y[2]; out.m[1][3] = Y[0];
        out.m[2][0] = Z[3]; out.m[2][1] = Z[1]; out.m[2][2] = W[3]; out.m[2][3] = W[1];
        out.m[3][0] = Z[2]; out.m[3][1] = Z[0]; out.m[3][2] = W[2]; out.m[3][3] = W[0];
        return out;
    }

    // Matrix.hpp:237-250 (polynomial Sin/Cos as upstream)
    static Matrix4 PerspectiveFovRH(float fov, float width, float height, float zNear, float zFar)
    {
        const float h = crtmath::Cos(0.5f * fov) / crtmath::Sin(0.5f * fov);
        const float w = h * height / width;
        Matrix4 M = Identity();
        M.m[0][0] = w;
        M.m[1][1] = h;
        M.m[2][2] = -(zFar + zNear) / (zFar - zNear);
        M.m[2][3] = -1.0f;
        M.m[3][2] = -(2.0f * zFar * zNear) / (zFar - zNear);
        M.m[3][3] = 0.0f;
        return M;
    }

    // Matrix.hpp:211-235. `front` is a direction (Camera.hpp:109). Upstream normalises with
    // _mm_rsqrt_ps (a vendor-specific estimate, hazard H10); pinned here to 1/sqrtf.
    static Matrix4 LookAtRH(const Vector3f& eye, const Vector3f& front, const Vector3f& up)
    {
        auto cross = [](const Vector3f& p, const Vector3f& q) {
            return Vector3f(p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x);
        };
        auto norm = [](const Vector3f& v) {
            const float r = 1.0f / std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z);
            return Vector3f(r * v.x, r * v.y, r * v.z);
        };
        auto dot = [](const Vector3f& p, const Vector3f& q) { return (p.x * q.x + p.y * q.y) + p.z * q.z; };
        const Vector3f dir(0.0f - front.x, 0.0f - front.y, 0.0f - front.z);
        const Vector3f R0 = norm(cross(up, dir));
        const Vector3f R1 = norm(cross(dir, R0));
        const Vector3f negEye(0.0f - eye.x, 0.0f - eye.y, 0.0f - eye.z);
        Matrix4 M = Identity();
        M.m[0][0] = R0.x; M.m[1][0] = R0.y; M.m[2][0] = R0.z; M.m[3][0] = dot(R0, negEye);
        M.m[0][1] = R1.x; M.m[1][1] = R1.y; M.m[2][1] = R1.z; M.m[3][1] = dot(R1, negEye);
        M.m[0][2] = dir.x; M.m[1][2] = dir.y; M.m[2][2] = dir.z; M.m[3][2] = dot(dir, negEye);
        return M;
    }
};
static_assert(sizeof(Matrix4) == 64, "Matrix4 must be 64 B");

// Camera.hpp:7-136 without window/mouse handling: the caller sets position/Front directly.
struct Camera {
    Matrix4 projection, view, inverseProjection, inverseView;
    float verticalFOV = 65.0f, nearClip = 0.01f, farClip = 500.0f;
    Vector3f position = Vector3f(0.0f, 4.0f, 15.0f);
    Vector3f Front = Vector3f(0.0f, 0.0f, -1.0f);
    int projWidth = 0, projHeight = 0;

    void RecalculateProjection(int width, int height)
    {
        projWidth = width; projHeight = height;
        projection = Matrix4::PerspectiveFovRH(verticalFOV * crtmath::DegToRad, (float)width, (float)height, nearClip, farClip);
        inverseProjection = Matrix4::Inverse(projection);
    }
    void RecalculateView()
    {
        view = Matrix4::LookAtRH(position, Front, Vector3f(0.0f, 1.0f, 0.0f));
        inverseView = Matrix4::Inverse(view);
    }
};
